/*
 * obvi_rccl.h -- compiled RCCL side of the multi-GPU exchange of libobvi_ba (include/obvi_ba.h
 * "multi-GPU"): one process per GPU, one communicator per process, the three per-step all-reduces of
 * obvi_ba_solve forwarded to ncclAllReduce (RCCL over xGMI) on the handle's own HIP stream.
 *
 * There is no reference counterpart: the reference chains sessions sequentially
 * (ltm_trajectory_sequence_executor.py:45-92) and has no collective.  The call this replaces is the
 * `obvi_ba_allreduce_objects(h, rccl_comm)` line of SURVEY.md 8(b); it became a callback
 * (obvi_ba_set_allreduce) so that the library proper does not link RCCL, and this file is the C/C++ host's
 * implementation of that callback -- libobvi_rccl.so links librccl, libobvi_ba.so does not.
 *
 * Plain C ABI: no HIP / RCCL / torch types in the signatures (a stream is a void*, the unique id 128 bytes).
 */
#ifndef OBVI_RCCL_H_
#define OBVI_RCCL_H_

#include <stdint.h>

#include "obvi_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct obvi_rccl_comm obvi_rccl_comm;

enum { OBVI_RCCL_ID_BYTES = 128 };   /* sizeof(ncclUniqueId) */

/* NCCL_VERSION_CODE of the librccl this library resolved at load time (ncclGetVersion), -1 on failure.  A process that also hosts
 * another RCCL user (torch.distributed maps its own librccl.so) compares the two before it forms a communicator here: see
 * INTEGRATION.md "multi-GPU". */
int32_t obvi_rccl_nccl_version(void);
/* rank 0 draws the job's id (ncclGetUniqueId) and hands the bytes to the other ranks by whatever channel the host has
 * (torch.distributed store, MPI, a file: see obvi_rccl_comm_create_from_file). */
int obvi_rccl_unique_id(char out[OBVI_RCCL_ID_BYTES]);
/* ncclCommInitRank on `device`; collective: every rank of the job calls it. */
int obvi_rccl_comm_create(const char id[OBVI_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, obvi_rccl_comm** out);
/* launcher-less rendezvous for a C++ host: rank 0 writes the id to `path` (atomically: temp file + rename), the others
 * poll for it (timeout_s), then obvi_rccl_comm_create.  `path` must be visible to every rank of the node.  A per-launch tag from the
 * environment (OBVI_RCCL_JOB, else TORCHELASTIC_RUN_ID, else MASTER_PORT -- the same value on every rank) is appended to the file name,
 * so that a file left by a crashed launch is never taken for this one. */
int obvi_rccl_comm_create_from_file(const char* path, int32_t rank, int32_t world, int32_t device, double timeout_s, obvi_rccl_comm** out);
void obvi_rccl_comm_destroy(obvi_rccl_comm* comm);
int32_t obvi_rccl_comm_rank(const obvi_rccl_comm* comm);
int32_t obvi_rccl_comm_world(const obvi_rccl_comm* comm);   /* ncclCommCount of the live communicator */
const char* obvi_rccl_last_error(const obvi_rccl_comm* comm);

/* The obvi_allreduce_fn: user = the communicator; op 0 = sum, 1 = max; fp64 in place on `stream` (a hipStream_t);
 * no host synchronisation.  Returns 0, or the ncclResult_t on failure. */
int obvi_rccl_allreduce(void* user, void* device_buf, int64_t count_f64, int32_t op, void* stream);
/* Issue order of the data-path collectives on this communicator so far: number of obvi_rccl_allreduce calls and a running hash of
 * their (count, op, stream ordinal).  Must be equal on every rank at any quiescent point (one communicator is driven from two streams
 * of a handle: legal only if all ranks enqueue in the same host order); ranks compare it with obvi_rccl_host_allreduce (min / max). */
int obvi_rccl_sequence(const obvi_rccl_comm* comm, uint64_t* calls, uint64_t* hash);
/* obvi_ba_set_allreduce(h, obvi_rccl_allreduce, comm) + obvi_ba_set_shared_objects(h, is_shared, rank, world) with the
 * communicator's rank / size. */
int obvi_rccl_attach(obvi_ba_handle* h, obvi_rccl_comm* comm, const uint8_t* is_shared);
/* bench / test plumbing on the same communicator: in-place all-reduce of a small HOST array through a device bounce buffer
 * (op as above, 2 = min), synchronous; and a barrier (all-reduce of one double). */
int obvi_rccl_host_allreduce(obvi_rccl_comm* comm, double* host_buf, int32_t count, int32_t op);
int obvi_rccl_barrier(obvi_rccl_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* OBVI_RCCL_H_ */
