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
 * poll for it (timeout_s), then obvi_rccl_comm_create.  `path` must be visible to every rank of the node.  OBVI_RCCL_JOB -- a string the
 * launcher gives every rank of ONE launch and no other (a job id; rank 0's pid + start time) -- is appended to the file name: a rank then
 * never opens another launch's file and no clocks are compared.  Without it the launcher's TORCHELASTIC_RUN_ID (unless it is torchrun's
 * default "none") / MASTER_PORT still go into the name -- they keep concurrent jobs apart, but they repeat from launch to launch -- and a
 * file older than the rendezvous time-out is rejected as the leftover of a crashed launch (same-host clocks). */
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


/* ---- several handles per rank (sessions over one object map, two per GPU) ---------------------------------------
 * A GROUP stands between k handles of ONE process (same device, one host thread each, all inside obvi_ba_solve at the same time) and the
 * inter-rank all-reduce: per collective the k callbacks rendezvous on the host, the last one to arrive enqueues -- on the group's own
 * stream, behind an event of every member's stream -- a kernel that sums (op 0) or maximises (op 1) the k device buffers, ONE inter-rank
 * all-reduce of the result (`inner`: obvi_rccl_allreduce on a communicator, or any obvi_allreduce_fn; NULL = this process is the whole
 * job), and the copy of the result back into the k buffers; every member's stream then waits for the group's event.  Nothing waits on the
 * host for the device.  The members of a job are numbered rank * k + member of world * k contributors: that is what a member's handle is
 * told by obvi_rccl_group_attach (obvi_ba_set_shared_objects semantics unchanged: object-only factors of a shared object are uploaded by
 * exactly one contributor of the job).  Every member must issue the same sequence of collectives (same counts and ops): the solve of a
 * shared-object job does, because all contributors take the same decisions.  A member that does not arrive within timeout_s (default
 * 120) makes the waiting ones fail with OBVI_ERR_NOT_READY instead of hanging. */
typedef struct obvi_rccl_group obvi_rccl_group;
int obvi_rccl_group_create(obvi_allreduce_fn inner, void* inner_user, int32_t rank, int32_t world, int32_t n_members, int32_t device,
                           obvi_rccl_group** out);
/* the same with a communicator of this library as the inter-rank step (comm NULL: one rank) */
int obvi_rccl_group_create_on_comm(obvi_rccl_comm* comm, int32_t n_members, obvi_rccl_group** out);
void obvi_rccl_group_destroy(obvi_rccl_group* g);
void obvi_rccl_group_set_timeout(obvi_rccl_group* g, double timeout_s);
/* obvi_ba_set_shared_objects(h, is_shared, rank * k + member, world * k) + obvi_ba_set_allreduce(h, obvi_rccl_group_allreduce, <member>) */
int obvi_rccl_group_attach(obvi_rccl_group* g, int32_t member, obvi_ba_handle* h, const uint8_t* is_shared);
/* what obvi_ba_set_allreduce takes for member i: the function below with this `user` */
void* obvi_rccl_group_member(obvi_rccl_group* g, int32_t member);
int obvi_rccl_group_allreduce(void* member, void* device_buf, int64_t count_f64, int32_t op, void* stream);
/* collectives completed by the group so far, and the doubles they carried between ranks (sum of counts: bytes = 8 x) */
int obvi_rccl_group_stats(const obvi_rccl_group* g, uint64_t* collectives, uint64_t* doubles);

#ifdef __cplusplus
}
#endif
#endif /* OBVI_RCCL_H_ */
