// host_util.h -- small host helpers of libobvi_ba: device buffers, error plumbing, and the
// symmetric inverse square root the factor constructors of the reference apply to their
// covariances (`cov.inverse().sqrt()`: bounding_box_factor.cpp:31-33, shape_prior_factor.cpp:11,
// independent_object_map_factor.cpp:11, relative_pose_factor.cpp:13).
#ifndef OBVI_HOST_UTIL_H_
#define OBVI_HOST_UTIL_H_

#include <hip/hip_runtime.h>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace obvi {

struct HipError { hipError_t code; const char* what; const char* file; int line; };

#define OBVI_HIP(expr)                                                         \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) throw ::obvi::HipError{e__, #expr, __FILE__, __LINE__}; \
  } while (0)

// Host -> device copies of an upload go through a pinned arena of the handle: the bytes are copied into it and the transfer is
// enqueued from there, so that it neither waits on a pageable staging copy inside the runtime (15-30 us per call, and an upload is
// some forty of them) nor needs the stream synchronised before the caller's buffer or a local vector may go away.  The arena is
// rewound whenever the stream is known to be idle (sync()); a copy that does not fit, or is too long to be worth a second pass
// over its bytes, goes the plain way and marks the arena `spilled`: the function that issued it must synchronise before it returns.
struct StagingArena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  bool spilled = false;
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (base == nullptr || at + bytes > cap) return nullptr;
    used = at + bytes;
    return base + at;
  }
  void rewind() { used = 0; spilled = false; }
};
inline thread_local StagingArena* tl_staging = nullptr;   // the arena of the handle whose API call runs on this thread
constexpr size_t kStagedCopyMaxBytes = (size_t)4 << 20;

inline void h2d_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
  StagingArena* a = tl_staging;
  if (a != nullptr && bytes <= kStagedCopyMaxBytes) {
    if (void* p = a->take(bytes)) {
      std::memcpy(p, src, bytes);
      OBVI_HIP(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s));
      return;
    }
  }
  if (a != nullptr) a->spilled = true;
  OBVI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
}

template <class T>
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p_) { (void)hipFree(p_); p_ = nullptr; } n_ = 0; cap_ = 0; }
  // grow-only allocation; contents are undefined after a growing resize
  void resize(size_t n) {
    if (n > cap_) {
      if (p_) { (void)hipFree(p_); p_ = nullptr; }
      size_t c = n + n / 8 + 16;
      OBVI_HIP(hipMalloc(reinterpret_cast<void**>(&p_), c * sizeof(T)));
      cap_ = c;
    }
    n_ = n;
  }
  void upload(const T* h, size_t n, hipStream_t s) {
    resize(n);
    h2d_async(p_, h, n * sizeof(T), s);
  }
  void upload(const std::vector<T>& h, hipStream_t s) { upload(h.data(), h.size(), s); }
  // Upload without the host-side copy: stage_begin(n) sizes the buffer and hands out n elements of pinned memory from the running call's
  // arena to be filled in place (nullptr when the arena has no room: fill a vector and upload() that), stage_commit starts the copy.
  T* stage_begin(size_t n) {
    resize(n);
    StagingArena* a = tl_staging;
    if (n == 0 || a == nullptr || n * sizeof(T) > kStagedCopyMaxBytes) return nullptr;
    return static_cast<T*>(a->take(n * sizeof(T)));
  }
  void stage_commit(const T* staged, size_t n, hipStream_t s) { if (n) OBVI_HIP(hipMemcpyAsync(p_, staged, n * sizeof(T), hipMemcpyHostToDevice, s)); }
  void download(T* h, size_t n, hipStream_t s) const {
    if (n) OBVI_HIP(hipMemcpyAsync(h, p_, n * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  void zero(hipStream_t s) { if (n_) OBVI_HIP(hipMemsetAsync(p_, 0, n_ * sizeof(T), s)); }
  T* get() const { return p_; }
  size_t size() const { return n_; }
  void swap(DevBuf& o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }

 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// Worker threads of the host's symbolic phase and upload sorts, started once per process: a sliding-window session builds a plan per
// window (a millisecond of work in ranges of points), and starting and joining std::threads for every range cost as much as the
// ranges themselves.  run(parts, fn) calls fn(0) ... fn(parts - 1), each exactly once, on the workers and on the calling thread, and
// returns when all are done.  Calls from different threads (one handle per thread: several sessions of one process, config #5) run
// CONCURRENTLY: every call publishes its own job, a free worker enters any published job that still has parts to hand out, and a
// caller always works on its own job, so no call waits for another one (round 4 serialised them behind one mutex: k handles planning at
// once queued).  Workers spin briefly before they sleep, so that back-to-back calls do not pay a wake-up each.
// A call is ONE immutable job record (function, part count, its own hand-out and completion counters) on the caller's stack, published
// in jobs_ under the mutex.  A worker enters a job only under that mutex (and is counted in job->inside while it holds the pointer);
// run() unpublishes the job under the same mutex and leaves only when every part is done AND no worker is inside any more, so a part
// index never travels from one job to the next and nothing of a finished job is touched after run() has returned.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int i = 0; i < workers; ++i) {
      try { threads_.emplace_back([this] { work(); }); }
      catch (const std::system_error&) { break; }   // fewer workers: the caller takes the rest itself
    }
    keep_near_caller();
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lock(m_); stop_ = true; gen_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int workers() const { return (int)threads_.size(); }
  void run(int parts, const std::function<void(int)>& fn) {
    if (parts <= 1 || threads_.empty()) { for (int i = 0; i < parts; ++i) fn(i); return; }
    Job job(&fn, parts);
    {
      std::lock_guard<std::mutex> lock(m_);
      jobs_.push_back(&job);
      published_.fetch_add(1, std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    take(job);
    while (job.left.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    {                                                                                       // nobody enters from here on
      std::lock_guard<std::mutex> lock(m_);
      for (size_t i = 0; i < jobs_.size(); ++i) if (jobs_[i] == &job) { jobs_.erase(jobs_.begin() + (std::ptrdiff_t)i); break; }
      published_.fetch_sub(1, std::memory_order_release);
    }
    while (job.inside.load(std::memory_order_acquire) != 0) std::this_thread::yield();     // ... and those who did have left
  }

 private:
  struct Job {
    Job(const std::function<void(int)>* f, int n) : fn(f), parts(n), left(n) {}
    const std::function<void(int)>* const fn;
    const int parts;
    std::atomic<int> next{0}, left, inside{0};
  };
  // The workers read what the calling thread wrote a moment ago and the caller then overwrites what they read (one problem per frame in
  // a sliding-window session).  Spread by the scheduler over a two-socket host, every such line crosses the sockets; kept on the caller's
  // block of logical CPUs (same socket, neighbouring L3 slices) it does not -- worth 2-3 % of a session on a 2 x 64-core host.  A library
  // must not change thread placement behind its host's back, so this is opt-in: OBVI_HOST_AFFINITY=1 (run_offline_ba sets it for itself).
  void keep_near_caller() {
#if defined(__linux__)
    const char* v = std::getenv("OBVI_HOST_AFFINITY");
    if (v == nullptr || std::atoi(v) != 1) return;
    const int cpu = sched_getcpu();
    cpu_set_t allowed;
    if (cpu < 0 || threads_.empty() || sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int block = 16, base = cpu - cpu % block;
    cpu_set_t near;
    CPU_ZERO(&near);
    int n = 0;
    for (int c = base; c < base + block; ++c) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &near); ++n; }
    if (n < 2) return;
    for (auto& t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(near), &near);
#endif
  }
  static void take(Job& job) {
    for (;;) {
      const int i = job.next.fetch_add(1, std::memory_order_acq_rel);
      if (i >= job.parts) return;
      (*job.fn)(i);
      job.left.fetch_sub(1, std::memory_order_release);
    }
  }
  void work() {
    uint64_t seen = 0;
    for (;;) {
      for (int spin = 0; spin < 20000 && gen_.load(std::memory_order_acquire) == seen; ++spin) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
      }
      Job* job = nullptr;
      uint64_t picked_at = seen;
      {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        if (stop_) return;
        for (Job* j : jobs_) if (j->next.load(std::memory_order_acquire) < j->parts) { job = j; break; }   // any caller's job with parts left
        picked_at = gen_.load(std::memory_order_acquire);
        if (job == nullptr) { seen = picked_at; continue; }                                               // nothing to do for this generation
        job->inside.fetch_add(1, std::memory_order_acq_rel);   // counted while the pointer is held: run() waits for zero
      }
      take(*job);
      job->inside.fetch_sub(1, std::memory_order_release);
      // another caller's job may be waiting: look again before sleeping -- but only if one is published (the common case is a single caller,
      // whose run() wants the mutex right now to take its job back: fifteen workers queueing for it cost more than the look is worth)
      if (published_.load(std::memory_order_acquire) <= 1) seen = picked_at;   // (a job published since then has moved gen_ on: it is looked at right away)
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> published_{0};   // jobs in jobs_ (read outside the mutex)
  std::vector<Job*> jobs_;   // published jobs, guarded by m_
  bool stop_ = false;
};

// out = (sym(cov))^(-1/2) for an n x n (n <= 9) symmetric positive definite matrix, row-major.
// One-sided view: eigen-decomposition by threshold-free cyclic Jacobi sweeps on a working copy,
// then out = sum_k v_k v_k^T / sqrt(lambda_k).  Returns false when cov is not finite / not SPD.
inline bool sym_inverse_sqrt(const double* cov, int n, double* out) {
  double a[9][9], v[9][9];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      a[i][j] = 0.5 * (cov[i * n + j] + cov[j * n + i]);
      v[i][j] = i == j;
      if (!std::isfinite(a[i][j])) return false;
    }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, tot = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) { tot += a[i][j] * a[i][j]; if (i != j) off += a[i][j] * a[i][j]; }
    if (off <= tot * 1e-32) break;
    for (int p = 0; p + 1 < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
        // rotation angle that annihilates a[p][q]
        const double tau = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::hypot(1.0, tau));
        const double c = 1.0 / std::hypot(1.0, t), s = t * c;
        for (int k = 0; k < n; ++k) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - s * y; a[k][q] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const double x = a[p][k], y = a[q][k]; a[p][k] = c * x - s * y; a[q][k] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const double x = v[k][p], y = v[k][q]; v[k][p] = c * x - s * y; v[k][q] = s * x + c * y; }
      }
  }
  for (int k = 0; k < n; ++k) if (!(a[k][k] > 0.0)) return false;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += v[i][k] * v[j][k] / std::sqrt(a[k][k]);
      out[i * n + j] = acc;
    }
  return true;
}

}  // namespace obvi
#endif  // OBVI_HOST_UTIL_H_
