// Test infrastructure: stress of obvi::HostPool (csrc/host_util.h) -- back-to-back runs of changing part counts from two caller threads.
// Every part of every run must execute exactly once and nothing may run after its run() has returned (the job record and the counters of a
// run live on the caller's stack).  Prints "ok <runs>" or aborts.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../obvi-slam_amd/csrc/host_util.h"

int main(int argc, char** argv) {
  const int runs = argc > 1 ? std::atoi(argv[1]) : 20000;
  obvi::HostPool pool(6);
  std::atomic<long> bad{0};
  std::atomic<long> on_workers{0};
  auto caller = [&](unsigned seed) {
    const std::thread::id me = std::this_thread::get_id();
    for (int r = 0; r < runs; ++r) {
      seed = seed * 1664525u + 1013904223u;
      const int parts = 2 + (int)((seed >> 16) % 23);
      std::vector<std::atomic<int>> hits(parts);
      for (auto& h : hits) h.store(0);
      std::atomic<int> alive{1};
      pool.run(parts, [&](int i) {
        if (i < 0 || i >= parts || !alive.load()) { bad.fetch_add(1); return; }
        hits[i].fetch_add(1);
        if (std::this_thread::get_id() != me) on_workers.fetch_add(1);
        volatile unsigned spin = 0;
        for (unsigned k = 0; k < 200u + (seed & 1023u); ++k) spin = spin + k;   // a microsecond of work: the workers get their share
      });
      alive.store(0);
      for (int i = 0; i < parts; ++i) if (hits[i].load() != 1) bad.fetch_add(1);
    }
  };
  std::thread a(caller, 1u), b(caller, 2u);
  a.join(); b.join();
  if (bad.load() != 0) { std::printf("FAILED %ld\n", bad.load()); return 1; }
  std::printf("ok %d runs, %ld parts on worker threads\n", 2 * runs, on_workers.load());
  return 0;
}
