/* nerf_comm.h -- the exchange step of the data-parallel hot path as a C ABI over RCCL (SURVEY 8(b)3:
 * `nerf_comm_init / allreduce / gather_tiles`; 8(e): one summing all-reduce of the flat gradient per step, row
 * bands of an evaluation image gathered on one rank).  Built as libnerf_comm.so (links librccl); libnerf_hip.so
 * does not depend on it.  The reference has no distributed code (grep for distributed|nccl: 0 hits);
 * a caller that is not a torch.distributed program -- the ctypes / cgo / JNI host of INTEGRATION.md -- uses
 * these instead of torch.distributed's process group.
 *
 * Conventions as in nerf_hip.h: extern "C", 0 = ok / negative code + nerf_comm_last_error(), raw device
 * pointers borrowed for the call, explicit hipStream_t (calls are stream-ordered and return at once; the
 * caller overlaps them with compute by giving them their own stream and hipEvents), one process per GPU.
 * Bootstrap: rank 0 calls nerf_comm_get_unique_id and hands the 128 bytes to the other ranks by any side
 * channel (file, socket, MPI, the launcher's store); every rank then calls nerf_comm_init on ITS device. */
#ifndef NERF_COMM_H
#define NERF_COMM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* nerf_comm_stream_t;          /* hipStream_t */
typedef struct nerf_comm* nerf_comm_t;

#define NERF_COMM_F32 0
#define NERF_COMM_BF16 1

int nerf_comm_abi_version(void);
const char* nerf_comm_last_error(void);
/* size of the opaque bootstrap token (ncclUniqueId) and its creation on rank 0 */
int nerf_comm_unique_id_bytes(void);
int nerf_comm_get_unique_id(void* id_out);
/* joins the communicator of `world` ranks on the calling thread's current HIP device (collective) */
int nerf_comm_init(const void* unique_id, int rank, int world, nerf_comm_t* comm_out);
int nerf_comm_rank(nerf_comm_t comm);
int nerf_comm_world(nerf_comm_t comm);
/* in-place sum over the ranks of buf[count] (gradients: the flat 595,844-float decoder gradient in two
 * ranges, the tiny-MLP gradient, one level group of the hash-table gradient; bf16 halves the wire bytes) */
int nerf_comm_allreduce_sum(nerf_comm_t comm, void* buf, int64_t count, int dtype, nerf_comm_stream_t stream);
/* The exchange of the sharded optimiser (reduce-scatter the table gradient, every rank steps its 1/N slice, all-gather
 * the fp16 copy the forward reads; project-nerf_amd/sharded.py).  buf holds `world` equal slices of `per` elements.
 *   nerf_comm_reduce_scatter_sum: slice `rank` of buf <- sum over the ranks of that slice (in place; the other slices are unspecified)
 *   nerf_comm_all_gather        : slice r of buf <- rank r's slice r, on every rank (in place); elem_bytes 2 (fp16 copy) or 4 (fp32 master) */
int nerf_comm_reduce_scatter_sum(nerf_comm_t comm, void* buf, int64_t per, int dtype, nerf_comm_stream_t stream);
int nerf_comm_all_gather(nerf_comm_t comm, void* buf, int64_t per, int elem_bytes, nerf_comm_stream_t stream);
/* evaluation: rank r holds tile [counts[r]] floats (its row band); root receives them back to back in
 * rank order into out[sum(counts)].  counts is a HOST array of `world` entries, the same on every rank;
 * out is ignored on the other ranks. */
int nerf_comm_gather_tiles(nerf_comm_t comm, const float* tile, const int64_t* counts_host, float* out, int root,
                           nerf_comm_stream_t stream);
int nerf_comm_destroy(nerf_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif
