// host_team.h -- a small persistent team of host threads for the library's host-side loops over independent parts
// (handle creation in ba_solver.hip; the sample streams and acceptance replays of ransac.hip).
#ifndef THEIA_HIP_HOST_TEAM_H_
#define THEIA_HIP_HOST_TEAM_H_
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <pthread.h>
#include <unistd.h>

namespace thip {

// A small persistent team of host threads for these loops: a create() at 3 M observations runs ~15 parallel regions, and
// starting + joining 15 - 31 threads for each of them cost more than some of the regions themselves.  One region at a time
// (a second caller -- entry points are re-entrant across host threads -- falls back to threads of its own); the workers are
// started on first use, re-started after a fork, and take the parts of a region from a shared counter (the results of a
// region never depend on who runs which part).
class HostTeam {
  std::mutex use_;                       // one region at a time
  std::mutex mu_;
  std::condition_variable work_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* job_ = nullptr;
  int nparts_ = 0, wanted_ = 0, active_ = 0;
  std::atomic<int> next_{0};
  uint64_t gen_ = 0;
  pid_t pid_ = 0;
  void worker(int id, uint64_t seen) {    // seen: the generation at the worker's creation (it waits for the next one)
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        work_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (id >= wanted_) continue;     // this region uses fewer workers
        job = job_;
      }
      for (int k = next_.fetch_add(1, std::memory_order_relaxed); k < nparts_; k = next_.fetch_add(1, std::memory_order_relaxed)) (*job)(k);
      std::lock_guard<std::mutex> lk(mu_);
      if (--active_ == 0) done_.notify_one();
    }
  }
 public:
  // false: the team is busy (the caller runs the region with threads of its own)
  bool run(int nparts, unsigned cap, const std::function<void(int)>& fn) {
    std::unique_lock<std::mutex> use(use_, std::try_to_lock);
    if (!use.owns_lock()) return false;
    if (pid_ != getpid()) {              // first use, or the child of a fork: the parent's threads do not exist here, and
      if (!workers_.empty()) {           // their handles can be neither joined nor detached.  The region state is the
        new (&workers_) std::vector<std::thread>();   // parent's too: a worker started against a stale generation would run
        new (&mu_) std::mutex();         // an empty region and count itself out of the next one (and mu_ may have been
        new (&work_) std::condition_variable();       // held by a parent thread at the fork)
        new (&done_) std::condition_variable();
        job_ = nullptr; nparts_ = 0; wanted_ = 0; active_ = 0; gen_ = 0;
        next_.store(0, std::memory_order_relaxed);
      }
      pid_ = getpid();
    }
    const int helpers = (int)cap - 1;
    if ((int)workers_.size() < helpers) {
      std::lock_guard<std::mutex> lk(mu_);   // new workers take the current generation as already seen
      const uint64_t g = gen_;
      while ((int)workers_.size() < helpers) { const int id = (int)workers_.size(); workers_.emplace_back([this, id, g] { worker(id, g); }); }
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &fn; nparts_ = nparts; wanted_ = helpers; active_ = helpers;
      next_.store(0, std::memory_order_relaxed);
      ++gen_;
    }
    work_.notify_all();
    for (int k = next_.fetch_add(1, std::memory_order_relaxed); k < nparts; k = next_.fetch_add(1, std::memory_order_relaxed)) fn(k);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return active_ == 0; });
    job_ = nullptr;
    return true;
  }
};
// Never destroyed: its threads wait for work until the process ends.  The child of a fork gets a FRESH team (pthread_atfork):
// the parent's may have been inside a region at that moment -- use_ held by a thread that does not exist in the child, which
// would leave every host pass of the child serial for good -- and none of its mutexes can be trusted; the old object is
// abandoned, not destroyed.
inline HostTeam*& host_team_slot() { static HostTeam* team = nullptr; return team; }
inline HostTeam& host_team() {
  static std::once_flag once;
  std::call_once(once, [] {
    host_team_slot() = new HostTeam;
    (void)pthread_atfork(nullptr, nullptr, [] { host_team_slot() = new HostTeam; });
  });
  return *host_team_slot();
}


}  // namespace thip
#endif
