// obvi_rccl.cpp -- libobvi_rccl.so: the compiled RCCL implementation of libobvi_ba's all-reduce callback
// (include/obvi_rccl.h).  One process per GPU; the collectives are enqueued on the stream the library passes in
// (the handle's stream), so the exchange is ordered with the kernels around it and the host never waits for it.
#include "../../include/obvi_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cctype>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <sys/stat.h>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

static_assert(sizeof(ncclUniqueId) == OBVI_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");

struct obvi_rccl_comm {
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1, device = 0;
  double* bounce = nullptr;     // device buffer for the small host all-reduces
  hipStream_t stream = nullptr; // their stream
  std::string err;
  // Issue order of the data-path collectives (obvi_rccl_allreduce): one communicator is driven from two streams of a handle, which is
  // legal only if every rank enqueues the same collectives in the same host order.  Every call folds (count, op, ordinal of the stream
  // among the streams seen so far) into a running hash; ranks compare (calls, hash) with obvi_rccl_sequence.
  uint64_t seq_calls = 0, seq_hash = 1469598103934665603ull;
  void* seq_streams[8] = {};
  int seq_nstreams = 0;
};

namespace {
constexpr int kBounceDoubles = 256;
int fail(obvi_rccl_comm* c, int code, const char* what, const char* detail) {
  if (c) c->err = std::string(what) + ": " + detail;
  return code;
}
}  // namespace

extern "C" {

int32_t obvi_rccl_nccl_version(void) {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? (int32_t)v : -1;
}

int obvi_rccl_unique_id(char out[OBVI_RCCL_ID_BYTES]) {
  if (!out) return OBVI_ERR_INVALID_ARGUMENT;
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return OBVI_ERR_HIP;
  std::memcpy(out, &id, sizeof(id));
  return OBVI_OK;
}

int obvi_rccl_comm_create(const char id_bytes[OBVI_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, obvi_rccl_comm** out) {
  if (!out) return OBVI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!id_bytes || world < 1 || rank < 0 || rank >= world) return OBVI_ERR_INVALID_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return OBVI_ERR_NO_DEVICE;
  obvi_rccl_comm* c = new (std::nothrow) obvi_rccl_comm();
  if (!c) return OBVI_ERR_HIP;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->bounce), sizeof(double) * kBounceDoubles) != hipSuccess) {
    obvi_rccl_comm_destroy(c);
    return OBVI_ERR_HIP;
  }
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    std::fprintf(stderr, "obvi_rccl: ncclCommInitRank failed: %s\n", ncclGetErrorString(r));
    c->comm = nullptr;
    obvi_rccl_comm_destroy(c);
    return OBVI_ERR_HIP;
  }
  *out = c;
  return OBVI_OK;
}

int obvi_rccl_comm_create_from_file(const char* path_in, int32_t rank, int32_t world, int32_t device, double timeout_s, obvi_rccl_comm** out) {
  if (!path_in || !out) return OBVI_ERR_INVALID_ARGUMENT;
  char id[OBVI_RCCL_ID_BYTES];
  // The file lives only between rank 0's write and the end of the collective initialisation: rank 0 removes whatever an earlier
  // (crashed) run left at `path` before it writes, and removes its own file once ncclCommInitRank has returned -- by then every rank
  // has read it.  Only OBVI_RCCL_JOB is a per-launch tag (the launcher's contract: one string per launch, e.g. rank 0's pid + start time):
  // with it in the file NAME a rank never opens another launch's file and no clocks are compared.  TORCHELASTIC_RUN_ID and MASTER_PORT
  // also go into the name -- they keep concurrent jobs of one host apart -- but torchrun's default run id is the literal "none" and the
  // port is normally a fixed 29500: they repeat from launch to launch, so a file that rank 0 of a crashed launch left behind (killed
  // between its rename and its remove) carries the same name, and a rank that starts before this launch's rank 0 has removed it would
  // take the stale id and hang in ncclCommInitRank.  Without a real per-launch tag a file is therefore still rejected by age (older than
  // the rendezvous time-out; same-host clocks).
  std::string tagged(path_in);
  bool per_launch_tag = false;
  for (const char* name : {"OBVI_RCCL_JOB", "TORCHELASTIC_RUN_ID", "MASTER_PORT"}) {
    const char* v = std::getenv(name);
    if (v == nullptr || *v == 0) continue;
    if (std::strcmp(name, "TORCHELASTIC_RUN_ID") == 0 && std::strcmp(v, "none") == 0) continue;   // torchrun's default: says nothing
    tagged += ".";
    for (const char* ch = v; *ch; ++ch) tagged += (std::isalnum((unsigned char)*ch) || *ch == '-' || *ch == '_') ? *ch : '_';
    per_launch_tag = std::strcmp(name, "OBVI_RCCL_JOB") == 0;
    break;
  }
  const char* path = tagged.c_str();
  if (rank == 0) {
    std::remove(path);
    const int rc = obvi_rccl_unique_id(id);
    if (rc != OBVI_OK) return rc;
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return OBVI_ERR_INVALID_ARGUMENT;
    const bool ok = std::fwrite(id, 1, sizeof(id), f) == sizeof(id);
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return OBVI_ERR_INVALID_ARGUMENT; }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      struct stat st;
      const bool fresh = ::stat(path, &st) == 0 && (per_launch_tag || std::difftime(std::time(nullptr), st.st_mtime) <= std::max(1.0, timeout_s));
      FILE* f = fresh ? std::fopen(path, "rb") : nullptr;
      if (f) {
        const size_t n = std::fread(id, 1, sizeof(id), f);
        std::fclose(f);
        if (n == sizeof(id)) break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return OBVI_ERR_NOT_READY;
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  const int rc = obvi_rccl_comm_create(id, rank, world, device, out);
  if (rank == 0) std::remove(path);   // initialised (or failed) everywhere: nobody needs the id any more, and the next run must not find it
  return rc;
}

void obvi_rccl_comm_destroy(obvi_rccl_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->bounce) (void)hipFree(c->bounce);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int32_t obvi_rccl_comm_rank(const obvi_rccl_comm* c) { return c ? c->rank : -1; }
int32_t obvi_rccl_comm_world(const obvi_rccl_comm* c) {
  if (!c || !c->comm) return -1;
  int n = -1;
  return ncclCommCount(c->comm, &n) == ncclSuccess ? n : -1;
}
const char* obvi_rccl_last_error(const obvi_rccl_comm* c) { return c ? c->err.c_str() : "null communicator"; }

int obvi_rccl_allreduce(void* user, void* device_buf, int64_t count_f64, int32_t op, void* stream) {
  obvi_rccl_comm* c = static_cast<obvi_rccl_comm*>(user);
  if (!c || !c->comm || !device_buf || count_f64 < 0) return OBVI_ERR_INVALID_ARGUMENT;
  if (count_f64 == 0) return 0;
  const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
  {
    int ord = 0;
    while (ord < c->seq_nstreams && c->seq_streams[ord] != stream) ++ord;
    if (ord == c->seq_nstreams && c->seq_nstreams < 8) c->seq_streams[c->seq_nstreams++] = stream;
    for (uint64_t w : {(uint64_t)count_f64, (uint64_t)(uint32_t)op, (uint64_t)ord}) { c->seq_hash ^= w; c->seq_hash *= 1099511628211ull; }
    ++c->seq_calls;
  }
  const ncclResult_t r = ncclAllReduce(device_buf, device_buf, (size_t)count_f64, ncclDouble, rop, c->comm, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) return fail(c, (int)r, "ncclAllReduce", ncclGetErrorString(r));
  return 0;
}

int obvi_rccl_sequence(const obvi_rccl_comm* c, uint64_t* calls, uint64_t* hash) {
  if (!c || !calls || !hash) return OBVI_ERR_INVALID_ARGUMENT;
  *calls = c->seq_calls; *hash = c->seq_hash;
  return OBVI_OK;
}

int obvi_rccl_attach(obvi_ba_handle* h, obvi_rccl_comm* c, const uint8_t* is_shared) {
  if (!h || !c) return OBVI_ERR_INVALID_ARGUMENT;
  const int rc = obvi_ba_set_shared_objects(h, is_shared, c->rank, c->world);
  if (rc != OBVI_OK) return rc;
  return obvi_ba_set_allreduce(h, obvi_rccl_allreduce, c);
}

int obvi_rccl_host_allreduce(obvi_rccl_comm* c, double* host_buf, int32_t count, int32_t op) {
  if (!c || !c->comm || !host_buf || count < 0 || count > kBounceDoubles) return OBVI_ERR_INVALID_ARGUMENT;
  if (count == 0) return OBVI_OK;
  if (hipSetDevice(c->device) != hipSuccess) return OBVI_ERR_HIP;
  if (hipMemcpyAsync(c->bounce, host_buf, sizeof(double) * count, hipMemcpyHostToDevice, c->stream) != hipSuccess) return OBVI_ERR_HIP;
  const int rc = obvi_rccl_allreduce(c, c->bounce, count, op, c->stream);
  if (rc != 0) return OBVI_ERR_HIP;
  if (hipMemcpyAsync(host_buf, c->bounce, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return OBVI_ERR_HIP;
  return hipStreamSynchronize(c->stream) == hipSuccess ? OBVI_OK : OBVI_ERR_HIP;
}

int obvi_rccl_barrier(obvi_rccl_comm* c) {
  double one = 1.0;
  return obvi_rccl_host_allreduce(c, &one, 1, 0);
}

}  // extern "C"

// ---- several handles per rank ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxMembers = 32;
struct GroupBufs { double* p[kMaxMembers]; };
// acc[i] = sum / max over the members' buffers (fixed order: member 0 first); with `fan` the result goes straight back into every buffer
// (a job of one rank: nothing travels between ranks)
__global__ void __launch_bounds__(256) k_group_reduce(GroupBufs b, int n, double* __restrict__ acc, int64_t count, int op, int fan) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < count; i += stride) {
    double v = b.p[0][i];
    for (int m = 1; m < n; ++m) { const double w = b.p[m][i]; v = op == 0 ? v + w : (op == 1 ? fmax(v, w) : fmin(v, w)); }
    if (fan) { for (int m = 0; m < n; ++m) b.p[m][i] = v; } else acc[i] = v;
  }
}
__global__ void __launch_bounds__(256) k_group_fan(GroupBufs b, int n, const double* __restrict__ acc, int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < count; i += stride) { const double v = acc[i]; for (int m = 0; m < n; ++m) b.p[m][i] = v; }
}
struct GroupMember { obvi_rccl_group* g = nullptr; int32_t index = 0; hipEvent_t ready = nullptr; void* buf = nullptr; int64_t count = 0; int32_t op = 0; };
}  // namespace

struct obvi_rccl_group {
  obvi_allreduce_fn inner = nullptr; void* inner_user = nullptr;
  int32_t rank = 0, world = 1, n = 0, device = 0;
  double timeout_s = 120.0;
  hipStream_t stream = nullptr;
  hipEvent_t done[2] = {nullptr, nullptr};
  double* acc = nullptr; int64_t acc_cap = 0;
  std::vector<GroupMember> members;
  std::mutex mu; std::condition_variable cv;
  std::atomic<uint64_t> generation{0};
  int arrived = 0, rc = 0;          // guarded by mu; rc: result of the round that just completed
  bool broken = false;              // a member timed out: every later call fails at once
  std::atomic<uint64_t> collectives{0}, doubles{0};   // read by obvi_rccl_group_stats from any thread
};

extern "C" {

int obvi_rccl_group_create(obvi_allreduce_fn inner, void* inner_user, int32_t rank, int32_t world, int32_t n_members, int32_t device, obvi_rccl_group** out) {
  if (!out) return OBVI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (n_members < 1 || n_members > kMaxMembers || world < 1 || rank < 0 || rank >= world || (inner == nullptr && world != 1)) return OBVI_ERR_INVALID_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return OBVI_ERR_NO_DEVICE;
  obvi_rccl_group* g = new (std::nothrow) obvi_rccl_group();
  if (!g) return OBVI_ERR_HIP;
  g->inner = inner; g->inner_user = inner_user; g->rank = rank; g->world = world; g->n = n_members; g->device = device;
  g->members.resize((size_t)n_members);
  bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < n_members && ok; ++i) {
    g->members[(size_t)i].g = g; g->members[(size_t)i].index = i;
    ok = hipEventCreateWithFlags(&g->members[(size_t)i].ready, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { obvi_rccl_group_destroy(g); return OBVI_ERR_HIP; }
  *out = g;
  return OBVI_OK;
}

int obvi_rccl_group_create_on_comm(obvi_rccl_comm* comm, int32_t n_members, obvi_rccl_group** out) {
  if (comm == nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return OBVI_ERR_NO_DEVICE;
    return obvi_rccl_group_create(nullptr, nullptr, 0, 1, n_members, dev, out);
  }
  // a communicator of one rank needs no collective at all: the group's own sum is the whole exchange
  const bool alone = comm->world == 1;
  return obvi_rccl_group_create(alone ? nullptr : obvi_rccl_allreduce, alone ? nullptr : comm, comm->rank, comm->world, n_members, comm->device, out);
}

void obvi_rccl_group_destroy(obvi_rccl_group* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (auto& m : g->members) if (m.ready) (void)hipEventDestroy(m.ready);
  for (int i = 0; i < 2; ++i) if (g->done[i]) (void)hipEventDestroy(g->done[i]);
  if (g->acc) (void)hipFree(g->acc);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

void obvi_rccl_group_set_timeout(obvi_rccl_group* g, double timeout_s) { if (g && timeout_s > 0.0) g->timeout_s = timeout_s; }
void* obvi_rccl_group_member(obvi_rccl_group* g, int32_t member) { return (g && member >= 0 && member < g->n) ? &g->members[(size_t)member] : nullptr; }

int obvi_rccl_group_attach(obvi_rccl_group* g, int32_t member, obvi_ba_handle* h, const uint8_t* is_shared) {
  if (!g || !h || member < 0 || member >= g->n) return OBVI_ERR_INVALID_ARGUMENT;
  const int rc = obvi_ba_set_shared_objects(h, is_shared, g->rank * g->n + member, g->world * g->n);
  if (rc != OBVI_OK) return rc;
  return obvi_ba_set_allreduce(h, obvi_rccl_group_allreduce, &g->members[(size_t)member]);
}

int obvi_rccl_group_stats(const obvi_rccl_group* g, uint64_t* collectives, uint64_t* doubles) {
  if (!g || !collectives || !doubles) return OBVI_ERR_INVALID_ARGUMENT;
  *collectives = g->collectives.load(std::memory_order_relaxed); *doubles = g->doubles.load(std::memory_order_relaxed);
  return OBVI_OK;
}

int obvi_rccl_group_allreduce(void* user, void* device_buf, int64_t count_f64, int32_t op, void* stream) {
  GroupMember* me = static_cast<GroupMember*>(user);
  if (!me || !me->g || !device_buf || count_f64 < 0 || op < 0 || op > 2) return OBVI_ERR_INVALID_ARGUMENT;
  obvi_rccl_group* g = me->g;
  if (g->n == 1) {
    // one handle on this rank (a rank whose sessions are fused into one problem): nothing to sum, no second stream, no event -- the
    // inter-rank all-reduce goes straight onto the handle's own stream (a cross-stream hop costs tens of microseconds on this runtime,
    // two of them per collective, three collectives per LM step)
    const int rc1 = g->inner != nullptr && count_f64 > 0 ? g->inner(g->inner_user, device_buf, count_f64, op, stream) : 0;
    g->collectives += 1; g->doubles += (uint64_t)count_f64;
    return rc1 != 0 ? OBVI_ERR_HIP : 0;
  }
  if (hipSetDevice(g->device) != hipSuccess) return OBVI_ERR_HIP;
  // this member's buffer is complete once its stream has reached this point
  if (hipEventRecord(me->ready, static_cast<hipStream_t>(stream)) != hipSuccess) return OBVI_ERR_HIP;
  std::unique_lock<std::mutex> lock(g->mu);
  if (g->broken) return OBVI_ERR_NOT_READY;
  me->buf = device_buf; me->count = count_f64; me->op = op;
  const uint64_t my_gen = g->generation.load(std::memory_order_acquire);
  if (++g->arrived < g->n) {
    // not the last one: wait for the round to be enqueued (a short spin first: the members of a lock-step solve arrive microseconds apart)
    lock.unlock();
    for (int spin = 0; spin < 4000 && g->generation.load(std::memory_order_acquire) == my_gen; ++spin) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
    }
    lock.lock();
    if (!g->cv.wait_for(lock, std::chrono::duration<double>(g->timeout_s), [&] { return g->generation.load(std::memory_order_acquire) != my_gen || g->broken; }) || g->broken) {
      g->broken = true;
      g->cv.notify_all();
      return OBVI_ERR_NOT_READY;
    }
  } else {
    // the last member to arrive enqueues the round for everybody
    int rc = 0;
    GroupBufs bufs{};
    for (int m = 0; m < g->n; ++m) {
      const GroupMember& o = g->members[(size_t)m];
      if (o.count != count_f64 || o.op != op) rc = OBVI_ERR_INVALID_ARGUMENT;   // the members are not in the same collective
      bufs.p[m] = static_cast<double*>(o.buf);
    }
    if (rc == 0 && count_f64 > 0) {
      for (int m = 0; m < g->n && rc == 0; ++m) if (hipStreamWaitEvent(g->stream, g->members[(size_t)m].ready, 0) != hipSuccess) rc = OBVI_ERR_HIP;
      const bool travel = g->inner != nullptr;
      if (rc == 0 && travel && count_f64 > g->acc_cap) {
        // grow-only; the old buffer may still be read by the previous round
        if (hipStreamSynchronize(g->stream) != hipSuccess) rc = OBVI_ERR_HIP;
        if (g->acc) (void)hipFree(g->acc);
        g->acc = nullptr; g->acc_cap = 0;
        const int64_t cap = count_f64 + count_f64 / 8 + 64;
        if (rc == 0 && hipMalloc(reinterpret_cast<void**>(&g->acc), sizeof(double) * (size_t)cap) == hipSuccess) g->acc_cap = cap; else rc = OBVI_ERR_HIP;
      }
      if (rc == 0) {
        const unsigned blocks = (unsigned)std::min<int64_t>(2048, (count_f64 + 255) / 256);
        hipLaunchKernelGGL(k_group_reduce, dim3(blocks), dim3(256), 0, g->stream, bufs, g->n, g->acc, count_f64, op, travel ? 0 : 1);
        if (travel) {
          if (g->inner(g->inner_user, g->acc, count_f64, op, g->stream) != 0) rc = OBVI_ERR_HIP;
          else hipLaunchKernelGGL(k_group_fan, dim3(blocks), dim3(256), 0, g->stream, bufs, g->n, g->acc, count_f64);
        }
        if (hipGetLastError() != hipSuccess) rc = OBVI_ERR_HIP;
      }
      if (hipEventRecord(g->done[my_gen & 1], g->stream) != hipSuccess) rc = OBVI_ERR_HIP;
    }
    g->rc = rc;
    g->arrived = 0;
    g->collectives += 1; g->doubles += (uint64_t)count_f64;
    g->generation.store(my_gen + 1, std::memory_order_release);
    g->cv.notify_all();
  }
  const int rc = g->rc;
  lock.unlock();
  // the member's own stream goes on behind the group's round (done[] alternates: round r + 2 re-records an event only after every member has
  // enqueued its wait for round r -- it cannot arrive at r + 2 before)
  if (rc == 0 && count_f64 > 0 && hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->done[my_gen & 1], 0) != hipSuccess) return OBVI_ERR_HIP;
  return rc;
}

}  // extern "C"
