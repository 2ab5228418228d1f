/*
 * eg3d_rccl.h — the one exchange step of the path for a C/C++ host: the variable-length
 * all-gather of the edge-point cloud over RCCL (SURVEY 8(b) `eg3d_allgather_edgepoints`, 8(e)).
 * Seeds are sharded over ranks in contiguous ranges (rank order = seed order,
 * plg_matching_from_refpoints.cpp:83-104 has no cross-seed state), every rank runs
 * eg3d_match_resident(..., device_only=1) on its range, then all ranks call
 * eg3d_allgather_edgepoints: counts first (24 B/rank), then an all-gather-v of the seven arrays of the cloud
 *   X | obs_off | key | obs_view | obs_pl | obs_seg | obs_xy
 * as ONE group of ncclSend / ncclRecv pairs: every array of every rank goes straight from the producing
 * context's HBM buffers to its final position in the receiver's result arrays (xGMI is point-to-point: each
 * pair of ranks uses its own link), then a small kernel rebases the observation offsets of ranks > 0. No
 * packing, padding or staging copy: a rank holds the gathered cloud once. Every rank ends with the whole,
 * globally ordered cloud. Lives in its own library
 * (libeg3d_rccl.so, links librccl) so that libeg3d.so has no communication dependency.
 * bench.py --gpus N calls this entry point (through ctypes) on a communicator created with
 * ncclCommInitRank; tests/rccl_two_rank_check.py is the 2-process check for a node with >= 2 GPUs.
 */
#ifndef EG3D_RCCL_H_
#define EG3D_RCCL_H_
#include "eg3d.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eg3d_gather eg3d_gather; /* staging + result buffers of one rank (grow-only) */

/* Communicator of the gather (thin wrappers, so that a host needs no RCCL headers and the communicator is
 * created by the same librccl / HIP runtime the collectives run on): rank 0 makes the 128-byte id and
 * distributes it by any means (bench.py: a torch.distributed broadcast), every rank then calls
 * eg3d_comm_init, which selects `device` and runs ncclCommInitRank. Any ncclComm_t made elsewhere works too. */
#define EG3D_COMM_ID_BYTES 128
int eg3d_comm_unique_id(void* id128);
int eg3d_comm_init(const void* id128, int n_ranks, int rank, int device, void** comm);
/* ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the communicator (any of the pointers may be NULL): the
 * pre-flight line a launcher reads to see that RCCL spans the N ranks it started, one per GPU. */
int eg3d_comm_query(void* comm, int* n_ranks, int* rank, int* device);
void eg3d_comm_destroy(void* comm);

eg3d_gather* eg3d_gather_create(int device);
void eg3d_gather_destroy(eg3d_gather* g);
/* How the payload travels. SENDRECV (default): one group of ncclSend / ncclRecv pairs, every pair of ranks on its own
 * xGMI link. BCAST: the same all-gather-v as one broadcast per (rank, array) — a fallback that uses only RCCL's
 * broadcast collective, for an RCCL build on which the grouped point-to-point transfers misbehave (the environment
 * variable EG3D_GATHER_MODE=bcast, read once by eg3d_gather_create, selects it too). Every rank must use the same
 * mode. Transfers are cut into pieces of at most `bytes` (default 1 GiB) so that no count of a call nears 2^31. */
#define EG3D_GATHER_MODE_SENDRECV 0
#define EG3D_GATHER_MODE_BCAST 1
int eg3d_gather_set_mode(eg3d_gather* g, int mode);
int eg3d_gather_set_chunk_bytes(eg3d_gather* g, uint64_t bytes);

/* nccl_comm: an initialised ncclComm_t of n_ranks ranks (this process = `rank`); hip_stream: the
 * stream the collective is enqueued on (NULL = the null stream). `local` = this rank's
 * eg3d_last_device_output (must be `complete`). On return `out` views the gathered cloud in HBM
 * (valid until the next call on `g`), rank_points / rank_obs (host arrays of n_ranks entries, may
 * be NULL) receive the per-rank counts. Collective: every rank must call it. */
#define EG3D_GATHER_ERR_ARG -1        /* bad arguments (identical on every rank); concat: a part views `g`'s own result buffers */
#define EG3D_GATHER_ERR_HIP -2        /* a device allocation failed on SOME rank (agreed: every rank returns it) */
#define EG3D_GATHER_ERR_RANGE -3      /* (retired: observation offsets are 64-bit) */
#define EG3D_GATHER_ERR_INCOMPLETE -4 /* SOME rank's local result is missing, spans several chunks or views `g`'s result buffers */
#define EG3D_GATHER_ERR_NCCL -5       /* communicator helpers: an RCCL call failed */
#define EG3D_GATHER_ERR_FATAL -6      /* a HIP / RCCL call failed on THIS rank between two collectives: the communicator has
                                         been aborted (ncclCommAbort) so that the peers' calls fail instead of blocking; do not
                                         use or destroy it afterwards */
/* Rank-local conditions known before a collective (incomplete local output, allocation failure) are exchanged
 * as status words in the counts all-gather and in one 8-byte all-gather after the allocations: every rank returns
 * the same error code before the payload exchange instead of leaving its peers blocked. A runtime failure between
 * collectives cannot be agreed on any more: that rank aborts the communicator (EG3D_GATHER_ERR_FATAL), which makes
 * the peers' pending collectives fail. The call returns after the exchange has completed on `hip_stream`
 * (it synchronises the stream): `local` may be overwritten as soon as it returns. */
int eg3d_allgather_edgepoints(eg3d_gather* g, void* nccl_comm, int n_ranks, int rank, void* hip_stream,
                              const eg3d_device_edgepoints* local, eg3d_device_edgepoints* out,
                              uint64_t* rank_points, uint64_t* rank_obs);

/* The same placement and rebasing without the collective: `n_parts` complete clouds resident on this GPU (of several
 * contexts or steps; parts[i] = what eg3d_last_device_output returned, still valid) become one ordered cloud, part
 * order = seed order, observation offsets rebased — every part takes the place a rank's cloud has in the exchange.
 * `out` views buffers of `g` (valid until the next call on `g`). A part must NOT view `g`'s own result buffers (the
 * `out` of an earlier call on the same `g`): such a call is refused with EG3D_GATHER_ERR_ARG — use a second
 * eg3d_gather to append to a running cloud. */
int eg3d_concat_edgepoints(eg3d_gather* g, int n_parts, const eg3d_device_edgepoints* parts, void* hip_stream,
                           eg3d_device_edgepoints* out);

#ifdef __cplusplus
}
#endif
#endif /* EG3D_RCCL_H_ */
