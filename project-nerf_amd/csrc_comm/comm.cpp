// libnerf_comm.so: the exchange step of the data-parallel path over RCCL (include/nerf_comm.h).
// One communicator per process (one process per GPU); every call is enqueued on the caller's stream.
// xGMI is point-to-point: the 2.4 MB decoder gradient is latency-bound, the 52 MB hash-table gradient is sent
// level group by level group (bf16 on the wire) so that a group is in flight while the next is computed.
#include "nerf_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

struct nerf_comm {
  ncclComm_t comm;
  int rank, world, device;
};

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

constexpr int kInval = -22, kComm = -5;

#define COMM_CHECK(call, what)                                                              \
  do {                                                                                      \
    const ncclResult_t r_ = (call);                                                         \
    if (r_ != ncclSuccess) return fail(kComm, "%s: %s", what, ncclGetErrorString(r_));      \
  } while (0)
}  // namespace

extern "C" int nerf_comm_abi_version(void) { return 1; }
extern "C" const char* nerf_comm_last_error(void) { return g_err; }
extern "C" int nerf_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int nerf_comm_get_unique_id(void* id_out) {
  if (!id_out) return fail(kInval, "nerf_comm_get_unique_id: NULL");
  ncclUniqueId id;
  COMM_CHECK(ncclGetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int nerf_comm_init(const void* unique_id, int rank, int world, nerf_comm_t* comm_out) {
  if (!unique_id || !comm_out) return fail(kInval, "nerf_comm_init: NULL");
  if (world < 1 || rank < 0 || rank >= world) return fail(kInval, "nerf_comm_init: rank %d of %d", rank, world);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  nerf_comm* c = new (std::nothrow) nerf_comm;
  if (!c) return fail(-12, "nerf_comm_init: out of memory");
  c->rank = rank;
  c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return fail(kComm, "nerf_comm_init: no HIP device");
  }
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(kComm, "ncclCommInitRank: %s", ncclGetErrorString(r));
  }
  *comm_out = c;
  return 0;
}

extern "C" int nerf_comm_rank(nerf_comm_t comm) { return comm ? comm->rank : fail(kInval, "nerf_comm_rank: NULL"); }
extern "C" int nerf_comm_world(nerf_comm_t comm) { return comm ? comm->world : fail(kInval, "nerf_comm_world: NULL"); }

extern "C" int nerf_comm_allreduce_sum(nerf_comm_t comm, void* buf, int64_t count, int dtype, nerf_comm_stream_t stream) {
  if (!comm) return fail(kInval, "nerf_comm_allreduce_sum: NULL communicator");
  if (count < 0) return fail(kInval, "nerf_comm_allreduce_sum: count=%lld", (long long)count);
  if (dtype != NERF_COMM_F32 && dtype != NERF_COMM_BF16) return fail(kInval, "nerf_comm_allreduce_sum: dtype=%d", dtype);
  if (count == 0) return 0;
  if (!buf) return fail(kInval, "nerf_comm_allreduce_sum: NULL buffer");
  COMM_CHECK(ncclAllReduce(buf, buf, (size_t)count, dtype == NERF_COMM_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, comm->comm,
                           static_cast<hipStream_t>(stream)),
             "ncclAllReduce");
  return 0;
}

// The exchange of the sharded optimiser: buf holds world equal slices of `per` elements; afterwards slice `rank` of
// buf is the sum over the ranks of that slice (the other slices are unspecified) ...
extern "C" int nerf_comm_reduce_scatter_sum(nerf_comm_t comm, void* buf, int64_t per, int dtype, nerf_comm_stream_t stream) {
  if (!comm) return fail(kInval, "nerf_comm_reduce_scatter_sum: NULL communicator");
  if (per < 0) return fail(kInval, "nerf_comm_reduce_scatter_sum: per=%lld", (long long)per);
  if (dtype != NERF_COMM_F32 && dtype != NERF_COMM_BF16) return fail(kInval, "nerf_comm_reduce_scatter_sum: dtype=%d", dtype);
  if (per == 0) return 0;
  if (!buf) return fail(kInval, "nerf_comm_reduce_scatter_sum: NULL buffer");
  const size_t esz = dtype == NERF_COMM_F32 ? 4 : 2;
  char* mine = static_cast<char*>(buf) + (size_t)comm->rank * (size_t)per * esz;     // in place: RCCL allows recv = send + rank * count
  COMM_CHECK(ncclReduceScatter(buf, mine, (size_t)per, dtype == NERF_COMM_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, comm->comm,
                               static_cast<hipStream_t>(stream)),
             "ncclReduceScatter");
  return 0;
}

// ... and every rank's slice `rank` of buf (per elements of elem_bytes = 2 or 4: the fp16 copy of the table, the fp32 master for
// checkpoints) on every rank, in place
extern "C" int nerf_comm_all_gather(nerf_comm_t comm, void* buf, int64_t per, int elem_bytes, nerf_comm_stream_t stream) {
  if (!comm) return fail(kInval, "nerf_comm_all_gather: NULL communicator");
  if (per < 0 || (elem_bytes != 2 && elem_bytes != 4)) return fail(kInval, "nerf_comm_all_gather: per=%lld elem_bytes=%d", (long long)per, elem_bytes);
  if (per == 0) return 0;
  if (!buf) return fail(kInval, "nerf_comm_all_gather: NULL buffer");
  const char* mine = static_cast<const char*>(buf) + (size_t)comm->rank * (size_t)per * (size_t)elem_bytes;
  COMM_CHECK(ncclAllGather(mine, buf, (size_t)per, elem_bytes == 4 ? ncclFloat32 : ncclFloat16, comm->comm, static_cast<hipStream_t>(stream)),
             "ncclAllGather");
  return 0;
}

extern "C" int nerf_comm_gather_tiles(nerf_comm_t comm, const float* tile, const int64_t* counts_host, float* out, int root,
                                      nerf_comm_stream_t stream) {
  if (!comm || !counts_host) return fail(kInval, "nerf_comm_gather_tiles: NULL");
  if (root < 0 || root >= comm->world) return fail(kInval, "nerf_comm_gather_tiles: root=%d of %d", root, comm->world);
  for (int r = 0; r < comm->world; ++r)
    if (counts_host[r] < 0) return fail(kInval, "nerf_comm_gather_tiles: counts[%d]=%lld", r, (long long)counts_host[r]);
  const int64_t mine = counts_host[comm->rank];
  if (mine > 0 && !tile) return fail(kInval, "nerf_comm_gather_tiles: NULL tile");
  if (comm->rank == root && !out) return fail(kInval, "nerf_comm_gather_tiles: NULL out on the root");
  hipStream_t s = static_cast<hipStream_t>(stream);
  COMM_CHECK(ncclGroupStart(), "ncclGroupStart");
  // inside the group an error must not return before ncclGroupEnd(): an open group would swallow every later call
  // of this thread.  The first failure is remembered, the remaining operations are skipped, the group is closed.
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  auto in_group = [&](ncclResult_t r, const char* op) {
    if (first == ncclSuccess && r != ncclSuccess) { first = r; what = op; }
  };
  if (mine > 0) in_group(ncclSend(tile, (size_t)mine, ncclFloat32, root, comm->comm, s), "ncclSend");
  if (comm->rank == root) {
    int64_t off = 0;
    for (int r = 0; r < comm->world && first == ncclSuccess; ++r) {
      if (counts_host[r] > 0) in_group(ncclRecv(out + off, (size_t)counts_host[r], ncclFloat32, r, comm->comm, s), "ncclRecv");
      off += counts_host[r];
    }
  }
  const ncclResult_t end = ncclGroupEnd();
  if (first != ncclSuccess) return fail(kComm, "%s: %s", what, ncclGetErrorString(first));
  if (end != ncclSuccess) return fail(kComm, "ncclGroupEnd: %s", ncclGetErrorString(end));
  return 0;
}

extern "C" int nerf_comm_destroy(nerf_comm_t comm) {
  if (!comm) return 0;
  const ncclResult_t r = ncclCommDestroy(comm->comm);
  delete comm;
  if (r != ncclSuccess) return fail(kComm, "ncclCommDestroy: %s", ncclGetErrorString(r));
  return 0;
}
