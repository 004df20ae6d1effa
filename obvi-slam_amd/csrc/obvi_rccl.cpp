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
#include <thread>

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
  // has read it.  What tells this launch's file from a leftover is a per-launch tag that the launcher gives every rank: OBVI_RCCL_JOB
  // (any string: a job id, rank 0's pid + start time), else the launcher's own TORCHELASTIC_RUN_ID / MASTER_PORT.  It becomes part of
  // the file NAME, so a rank never opens another launch's file and no clocks are compared.  Only without any tag does a rank fall back
  // to rejecting a file by age (older than the rendezvous time-out) -- same-host clocks, and a leftover younger than the time-out is
  // then still accepted: give launches a tag.
  std::string tagged(path_in);
  bool have_tag = false;
  for (const char* name : {"OBVI_RCCL_JOB", "TORCHELASTIC_RUN_ID", "MASTER_PORT"}) {
    const char* v = std::getenv(name);
    if (v != nullptr && *v != 0) {
      tagged += ".";
      for (const char* ch = v; *ch; ++ch) tagged += (std::isalnum((unsigned char)*ch) || *ch == '-' || *ch == '_') ? *ch : '_';
      have_tag = true;
      break;
    }
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
      const bool fresh = ::stat(path, &st) == 0 && (have_tag || std::difftime(std::time(nullptr), st.st_mtime) <= std::max(1.0, timeout_s));
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
