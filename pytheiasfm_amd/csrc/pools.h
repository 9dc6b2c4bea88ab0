// Process-wide caches of device blocks and pinned host blocks (theia_hip_release_scratch() empties both).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "theia_hip.h"
#include "theia_hip_internal.h"

namespace thip {

// A block can go back to a cache while work that uses it is still queued (a buffer that grows between rounds, the
// destructors on an error return): the cache therefore tags a returned block with an event recorded on the stream the
// returning thread is working on (PoolStreamScope, set by the entry points that enqueue on the shared solver stream) and
// the next taker waits for that event -- in the normal case it completed long ago and the wait is a flag test.  Threads
// without a scope (the BA handle, which synchronises its own stream before it lets go) return blocks untagged.
inline hipStream_t& pool_stream() { static thread_local hipStream_t s = nullptr; return s; }
struct PoolStreamScope {
  hipStream_t prev;
  explicit PoolStreamScope(hipStream_t s) : prev(pool_stream()) { pool_stream() = s; }
  ~PoolStreamScope() { pool_stream() = prev; }
  PoolStreamScope(const PoolStreamScope&) = delete;
  PoolStreamScope& operator=(const PoolStreamScope&) = delete;
};
inline hipEvent_t pool_mark() {
  hipStream_t s = pool_stream();
  if (!s) return nullptr;
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); return nullptr; }
  if (hipEventRecord(e, s) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e); (void)hipStreamSynchronize(s); return nullptr; }
  return e;
}
inline void pool_wait(hipEvent_t e) {
  if (!e) return;
  const hipError_t a = hipEventSynchronize(e), b = hipEventDestroy(e);
  if (a != hipSuccess || b != hipSuccess) {
    (void)hipGetLastError();   // (not this caller's failure: the block is handed out after a device-wide wait instead)
    static const bool report = getenv("THEIA_HIP_DEBUG_STICKY") != nullptr;
    if (report) std::fprintf(stderr, "theia_hip: pool event failed: %s / %s\n", hipGetErrorString(a), hipGetErrorString(b));
    if (a != hipSuccess) (void)hipDeviceSynchronize();
  }
}

// Device blocks of a batch call come from a small process-wide cache: a verification pipeline calls the batch entry
// points back to back with the same shapes, and hipMalloc / hipFree of the GB-sized model workspace cost more than the
// kernels of a 1000-pair chunk.  Blocks are reused when they are at most twice the request; the cache keeps at most
// kPoolLimit bytes (the rest goes back to the runtime), theia_hip_release_scratch() empties it.
struct DevPool {
  struct Block { void* p; size_t bytes; int dev; hipEvent_t ev; };
  std::mutex mu;
  std::vector<Block> free_blocks;
  size_t held = 0;
  static constexpr size_t kPoolLimit = (size_t)6 << 30;
  void* take(size_t bytes, size_t* got) {
    int dev = 0;
    (void)hipGetDevice(&dev);   // a block belongs to the device it was allocated on
    Block taken{nullptr, 0, 0, nullptr};
    {
      std::lock_guard<std::mutex> lk(mu);
      int best = -1;
      for (int i = 0; i < (int)free_blocks.size(); ++i)
        if (free_blocks[i].dev == dev && free_blocks[i].bytes >= bytes && free_blocks[i].bytes <= 2 * bytes + 4096 &&
            (best < 0 || free_blocks[i].bytes < free_blocks[best].bytes)) best = i;
      if (best >= 0) {
        taken = free_blocks[best];
        free_blocks.erase(free_blocks.begin() + best);
        held -= taken.bytes; *got = taken.bytes;
      }
    }
    if (taken.p) { pool_wait(taken.ev); return taken.p; }   // (outside the lock)
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {   // make room and try once more
      release();
      if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    *got = bytes;
    return p;
  }
  void give(void* p, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static const bool off = getenv("THEIA_HIP_NO_POOL") != nullptr;   // diagnostic: every block goes back to the runtime
    hipEvent_t ev = pool_mark();
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!off && held + bytes <= kPoolLimit) { free_blocks.push_back(Block{p, bytes, dev, ev}); held += bytes; return; }
    }
    pool_wait(ev);
    (void)hipFree(p);
  }
  void release() {
    std::vector<Block> blocks;
    { std::lock_guard<std::mutex> lk(mu); blocks.swap(free_blocks); held = 0; }
    for (const Block& b : blocks) { pool_wait(b.ev); (void)hipFree(b.p); }
  }
};
inline DevPool& dev_pool() { static DevPool pool; return pool; }

template <typename T>
struct DBuf {
  T* p = nullptr;
  size_t cap = 0, bytes = 0;
  ~DBuf() { if (p) dev_pool().give(p, bytes); }
  int ensure(size_t count) {
    if (count <= cap) return 0;
    if (p) dev_pool().give(p, bytes);
    p = nullptr; cap = 0; bytes = 0;
    size_t got = 0;
    p = static_cast<T*>(dev_pool().take(std::max<size_t>(count * sizeof(T), 256), &got));
    if (!p) return set_error(THEIA_HIP_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", count * sizeof(T));
    bytes = got; cap = got / sizeof(T);
    return 0;
  }
};

// Pinned host blocks for the per-round transfers (sample indices up, counts / costs / inlier counts down): pageable
// vectors made the D2H copies of a round 4.6 ms against 24 ms of kernels (the DMA engine stages through bounce
// buffers); pinned memory is expensive to allocate, so the blocks are cached like the device blocks above.
struct HostPool {
  struct Block { void* p; size_t bytes; hipEvent_t ev; };
  std::mutex mu;
  std::vector<Block> free_blocks;
  size_t held = 0;
  static constexpr size_t kPoolLimit = (size_t)2 << 30;
  void* take(size_t bytes, size_t* got) {
    Block taken{nullptr, 0, nullptr};
    {
      std::lock_guard<std::mutex> lk(mu);
      int best = -1;
      for (int i = 0; i < (int)free_blocks.size(); ++i)
        if (free_blocks[i].bytes >= bytes && free_blocks[i].bytes <= 2 * bytes + 4096 &&
            (best < 0 || free_blocks[i].bytes < free_blocks[best].bytes)) best = i;
      if (best >= 0) {
        taken = free_blocks[best];
        free_blocks.erase(free_blocks.begin() + best);
        held -= taken.bytes; *got = taken.bytes;
      }
    }
    if (taken.p) { pool_wait(taken.ev); return taken.p; }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    *got = bytes;
    return p;
  }
  void give(void* p, size_t bytes) {
    hipEvent_t ev = pool_mark();
    {
      std::lock_guard<std::mutex> lk(mu);
      if (held + bytes <= kPoolLimit) { free_blocks.push_back(Block{p, bytes, ev}); held += bytes; return; }
    }
    pool_wait(ev);
    (void)hipHostFree(p);
  }
  void release() {
    std::vector<Block> blocks;
    { std::lock_guard<std::mutex> lk(mu); blocks.swap(free_blocks); held = 0; }
    for (const Block& b : blocks) { pool_wait(b.ev); (void)hipHostFree(b.p); }
  }
};
inline HostPool& host_pool() { static HostPool pool; return pool; }

template <typename T>
struct HBuf {   // the std::vector calls the driver used, on a pinned block (contents are NOT preserved by a growing resize)
  T* p = nullptr;
  size_t cap = 0, bytes = 0, n = 0;
  bool pageable = false;   // the pinned allocation failed and the caller accepts plain memory (handle creation at sizes
                           // beyond what the host lets a process pin): copies from it must be synchronised at the call
  HBuf() = default;
  HBuf(const HBuf&) = delete;
  HBuf& operator=(const HBuf&) = delete;
  ~HBuf() { drop(); }
  void drop() {
    if (p) { if (pageable) std::free(p); else host_pool().give(p, bytes); }
    p = nullptr; cap = 0; bytes = 0; pageable = false;
  }
  bool reserve(size_t count, bool allow_pageable = false) {
    if (count <= cap) return true;
    drop();
    size_t got = 0;
    const size_t want = std::max<size_t>(count * sizeof(T), 4096);
    const bool skip_pinned = allow_pageable && getenv("THEIA_HIP_NO_PINNED") != nullptr;   // (test switch for the fallback)
    if (!skip_pinned) p = static_cast<T*>(host_pool().take(want, &got));
    if (!p && allow_pageable) {
      (void)hipGetLastError();
      p = static_cast<T*>(std::malloc(want));
      got = want; pageable = p != nullptr;
    }
    if (!p) return false;
    bytes = got; cap = got / sizeof(T);
    return true;
  }
  bool resize(size_t count, bool allow_pageable = false) { if (!reserve(count, allow_pageable)) return false; n = count; return true; }
  bool assign(size_t count, T v) { if (!resize(count)) return false; for (size_t i = 0; i < count; ++i) p[i] = v; return true; }
  bool pinned() const { return !pageable; }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

}  // namespace thip
