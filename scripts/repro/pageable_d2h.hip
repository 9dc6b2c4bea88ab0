// Stand-alone reproducer for DESIGN.md 3.4 "runtime lesson": hipMemcpyAsync(DeviceToHost) into SMALL PAGEABLE buffers on
// a non-blocking stream, several destinations inside one 4 KB page, followed by hipStreamSynchronize -- the pattern
// theia_hip_ba_run used for its five trace arrays before they went through blocking copies.  Checks (a) that every
// destination holds the device values after the synchronisation, (b) canaries between / around the destinations.
//   hipcc --offload-arch=gfx950 -O2 scripts/repro/pageable_d2h.hip -o scripts/repro/pageable_d2h && ./pageable_d2h [threads]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

__global__ void fill(double* p, int n, double base) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = base + i; }

static std::atomic<long> g_bad_value{0}, g_bad_canary{0}, g_rounds{0};

static void worker(int tid, int rounds, bool same_page) {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int narr = 5, cap = 256;             // five trace arrays of `cap` doubles on the device
  double* d[narr];
  for (int a = 0; a < narr; ++a) hipMalloc(&d[a], cap * sizeof(double));
  // host: one malloc'd block; destination a holds `cnt` doubles at a small offset, canaries everywhere else
  const size_t block = 64 * 1024;
  std::vector<unsigned char> host(block);
  for (int r = 0; r < rounds; ++r) {
    const int cnt = 1 + (r % 9);              // 8 .. 72 bytes per copy (a solve of 1..9 iterations)
    std::memset(host.data(), 0xA5, block);
    size_t off[narr];
    for (int a = 0; a < narr; ++a) off[a] = same_page ? (size_t)(64 + a * 96) : (size_t)(64 + a * 8192);
    const double base = 1000.0 * tid + r;
    for (int a = 0; a < narr; ++a) fill<<<1, 256, 0, st>>>(d[a], cap, base + 0.125 * a);
    for (int a = 0; a < narr; ++a)
      hipMemcpyAsync(host.data() + off[a], d[a], cnt * sizeof(double), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    for (int a = 0; a < narr; ++a) {
      double v[16]; std::memcpy(v, host.data() + off[a], cnt * sizeof(double));
      for (int i = 0; i < cnt; ++i) if (v[i] != base + 0.125 * a + i) g_bad_value++;
    }
    // canaries: every byte outside the five destinations
    std::vector<unsigned char> expect(block, 0xA5);
    for (int a = 0; a < narr; ++a) std::memcpy(expect.data() + off[a], host.data() + off[a], cnt * sizeof(double));
    if (std::memcmp(expect.data(), host.data(), block) != 0) g_bad_canary++;
    g_rounds++;
  }
  for (int a = 0; a < narr; ++a) hipFree(d[a]);
  hipStreamDestroy(st);
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 6;
  const int rounds = argc > 2 ? atoi(argv[2]) : 20000;
  for (int same_page = 1; same_page >= 0; --same_page) {
    g_bad_value = 0; g_bad_canary = 0; g_rounds = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker, t, rounds, same_page != 0);
    for (auto& t : th) t.join();
    printf("%s: %ld rounds on %d threads: wrong values %ld, canary hits %ld\n", same_page ? "five destinations in one page" : "destinations 8 KB apart",
           g_rounds.load(), threads, g_bad_value.load(), g_bad_canary.load());
  }
  return 0;
}
