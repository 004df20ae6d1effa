// Test infrastructure: the library's HostPool (obvi-slam_amd/csrc/host_util.h) behind a C function, so that its concurrency contract can be
// exercised without a GPU -- calls from several threads at once (one handle per thread: config #5), every part of every call run exactly once,
// nothing of a finished call touched afterwards.  Built by __graft_entry__.build() with hipcc in host-only mode (the header includes the HIP
// runtime API for DevBuf; the pool itself uses none of it).
#include "../obvi-slam_amd/csrc/host_util.h"

#include <atomic>
#include <thread>
#include <vector>

extern "C" int hostpool_stress(int workers, int callers, int runs_per_caller, int max_parts) {
  obvi::HostPool pool(workers);
  std::atomic<int> errors{0};
  std::vector<std::thread> th;
  for (int c = 0; c < callers; ++c)
    th.emplace_back([&, c] {
      unsigned seed = 12345u + 977u * (unsigned)c;
      for (int r = 0; r < runs_per_caller; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int parts = 1 + (int)((seed >> 16) % (unsigned)max_parts);
        std::vector<std::atomic<int>> hits((size_t)parts);
        for (auto& h : hits) h.store(0);
        std::atomic<long long> sum{0};
        const std::function<void(int)> fn = [&](int i) {
          if (i < 0 || i >= parts) { errors.fetch_add(1); return; }
          hits[(size_t)i].fetch_add(1);
          long long s = 0;
          for (int k = 0; k < 200 + 37 * (i % 5); ++k) s += (long long)k * (i + 1);   // a little work of uneven length
          sum.fetch_add(s);
        };
        pool.run(parts, fn);
        for (int i = 0; i < parts; ++i) if (hits[(size_t)i].load() != 1) errors.fetch_add(1);   // each part exactly once, all done when run() returns
        long long want = 0;
        for (int i = 0; i < parts; ++i) for (int k = 0; k < 200 + 37 * (i % 5); ++k) want += (long long)k * (i + 1);
        if (sum.load() != want) errors.fetch_add(1);
      }
    });
  for (auto& t : th) t.join();
  return errors.load();
}
