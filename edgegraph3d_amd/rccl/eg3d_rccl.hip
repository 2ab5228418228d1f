// eg3d_rccl.hip — RCCL all-gather of the edge-point cloud (include/eg3d_rccl.h). One process per GPU. The
// clouds differ in size per rank, and xGMI is point-to-point: the exchange is an all-gather-v made of ONE group
// of ncclSend / ncclRecv pairs (every pair of ranks on its own link, all links busy at once) whose receives land
// directly at their final positions in the result arrays — no packing, no padding, no staging copy.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/eg3d_rccl.h"
#include "eg3d_gather_plan.h"

namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(n + n / 4, 256);
    if (hipMalloc(&p, want) != hipSuccess) return -1;
    cap = want;
    return 0;
  }
  ~Buf() {
    if (p) (void)hipFree(p);
  }
};

using eg3d_gather_detail::kFieldBytes;

// After the blocks have landed: the observation offsets of rank r's points index r's own arrays; add the
// observations of the ranks before it. bases = [R] point bases followed by [R] observation bases.
__global__ void k_rebase(uint64_t* obs_off, const uint64_t* bases, int n_ranks, uint64_t total_points) {
  const uint64_t* pbase = bases;
  const uint64_t* obase = bases + n_ranks;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_points; i += (uint64_t)gridDim.x * blockDim.x) {
    int r = 0;
    while (r + 1 < n_ranks && pbase[r + 1] <= i) r++;
    if (obase[r]) obs_off[i] += obase[r];
  }
}

}  // namespace

// Largest single ncclSend / ncclRecv / ncclBroadcast: a transfer is cut into pieces of at most this many bytes, the
// same way on both sides (every rank knows every rank's counts), so that no element count of a collective call comes
// near 2^31 whatever the library's internal counters are (a 32 768-seed C4 step moves ~3.5 GB per rank).
#ifndef EG3D_GATHER_CHUNK_BYTES
#define EG3D_GATHER_CHUNK_BYTES (1ull << 30)
#endif

struct eg3d_gather {
  int device = 0;
  int mode = EG3D_GATHER_MODE_SENDRECV;
  uint64_t chunk = EG3D_GATHER_CHUNK_BYTES;
  Buf cnt_dev, X, off, key, view, pl, seg, xy;
  Buf* field(int f) { return f == 0 ? &X : f == 1 ? &off : f == 2 ? &key : f == 3 ? &view : f == 4 ? &pl : f == 5 ? &seg : &xy; }
  // does [p, p + n) overlap one of the result buffers? (a part that views them cannot be an input)
  bool owns(const void* p) const {
    const Buf* all[7] = {&X, &off, &key, &view, &pl, &seg, &xy};
    for (const Buf* b : all)
      if (b->p && (const char*)p >= (const char*)b->p && (const char*)p < (const char*)b->p + b->cap) return true;
    return false;
  }
};

extern "C" eg3d_gather* eg3d_gather_create(int device) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  eg3d_gather* g = new eg3d_gather();
  g->device = device;
  // read ONCE, here: EG3D_GATHER_MODE=bcast selects the fallback exchange; EG3D_GATHER_CHUNK_BYTES is a test knob
  // (small pieces make a small cloud take the multi-piece path)
  if (const char* e = getenv("EG3D_GATHER_MODE")) g->mode = (e[0] == 'b' || e[0] == '1') ? EG3D_GATHER_MODE_BCAST : EG3D_GATHER_MODE_SENDRECV;
  if (const char* e = getenv("EG3D_GATHER_CHUNK_BYTES")) {
    const long long v = atoll(e);
    if (v >= 16) g->chunk = (uint64_t)v;
  }
  return g;
}
extern "C" int eg3d_gather_set_mode(eg3d_gather* g, int mode) {
  if (!g || (mode != EG3D_GATHER_MODE_SENDRECV && mode != EG3D_GATHER_MODE_BCAST)) return EG3D_GATHER_ERR_ARG;
  g->mode = mode;
  return 0;
}
extern "C" int eg3d_gather_set_chunk_bytes(eg3d_gather* g, uint64_t bytes) {
  if (!g || bytes < 16) return EG3D_GATHER_ERR_ARG;
  g->chunk = bytes;
  return 0;
}
extern "C" void eg3d_gather_destroy(eg3d_gather* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  delete g;
}

// ---- communicator helpers: created inside this library so that the communicator, the collectives and the
// device selection all go through the SAME librccl / HIP runtime the gather is linked against
extern "C" int eg3d_comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == EG3D_COMM_ID_BYTES, "ncclUniqueId size");
  if (!id128) return EG3D_GATHER_ERR_ARG;
  return ncclGetUniqueId((ncclUniqueId*)id128) == ncclSuccess ? 0 : EG3D_GATHER_ERR_NCCL;
}
extern "C" int eg3d_comm_init(const void* id128, int n_ranks, int rank, int device, void** comm) {
  if (!id128 || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return EG3D_GATHER_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return EG3D_GATHER_ERR_HIP;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  if (ncclCommInitRank(&c, n_ranks, id, rank) != ncclSuccess) return EG3D_GATHER_ERR_NCCL;
  *comm = (void*)c;
  return 0;
}
// What RCCL itself says about the communicator: ranks it spans, this process's rank in it, the HIP device it is bound to.
// bench.py prints these per rank before the first step ("did RCCL see N ranks, one per GPU"), and refuses to run otherwise.
extern "C" int eg3d_comm_query(void* comm, int* n_ranks, int* rank, int* device) {
  if (!comm) return EG3D_GATHER_ERR_ARG;
  int n = -1, r = -1, d = -1;
  if (ncclCommCount((ncclComm_t)comm, &n) != ncclSuccess) return EG3D_GATHER_ERR_NCCL;
  if (ncclCommUserRank((ncclComm_t)comm, &r) != ncclSuccess) return EG3D_GATHER_ERR_NCCL;
  if (ncclCommCuDevice((ncclComm_t)comm, &d) != ncclSuccess) return EG3D_GATHER_ERR_NCCL;
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  return 0;
}
extern "C" void eg3d_comm_destroy(void* comm) {
  if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

static void fill_out(eg3d_gather* g, uint64_t tp, uint64_t to, eg3d_device_edgepoints* out) {
  out->n_points = tp;
  out->n_obs = to;
  out->X = (const float*)g->X.p;
  out->obs_off = (const uint64_t*)g->off.p;
  out->obs_view = (const int32_t*)g->view.p;
  out->obs_pl = (const uint32_t*)g->pl.p;
  out->obs_seg = (const uint32_t*)g->seg.p;
  out->obs_xy = (const float*)g->xy.p;
  out->key = (const uint32_t*)g->key.p;
  out->complete = 1;
}
static int ensure_result(eg3d_gather* g, uint64_t tp, uint64_t to) {
  for (int f = 0; f < 7; f++)
    if (g->field(f)->ensure(kFieldBytes[f] * ((f < 3 ? tp : to) + 1) + 16)) return -1;
  return 0;
}
static const void* field_ptr(const eg3d_device_edgepoints* e, int f) {
  return f == 0 ? (const void*)e->X : f == 1 ? (const void*)e->obs_off : f == 2 ? (const void*)e->key
         : f == 3 ? (const void*)e->obs_view : f == 4 ? (const void*)e->obs_pl : f == 5 ? (const void*)e->obs_seg
                  : (const void*)e->obs_xy;
}

// A rank-local runtime failure between two collectives cannot be reported to the peers any more: abort the
// communicator so that their pending collectives fail instead of blocking for ever.
#define FATAL_UNLESS(ok)                              \
  do {                                                \
    if (!(ok)) {                                      \
      if (comm) (void)ncclCommAbort(comm);            \
      return EG3D_GATHER_ERR_FATAL;                   \
    }                                                 \
  } while (0)

extern "C" int eg3d_allgather_edgepoints(eg3d_gather* g, void* nccl_comm, int n_ranks, int rank, void* hip_stream,
                                         const eg3d_device_edgepoints* local, eg3d_device_edgepoints* out,
                                         uint64_t* rank_points, uint64_t* rank_obs) {
  // Argument errors that every rank sees identically (same call on every rank) return at once. A condition that
  // is RANK-LOCAL but known before a collective (an incomplete local result, a failed allocation) travels as a
  // status word — in the counts all-gather, and in an 8-byte all-gather after the allocations — so that all ranks
  // agree to give up BEFORE the payload exchange. A runtime failure between collectives aborts the communicator.
  if (!g || !nccl_comm || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return EG3D_GATHER_ERR_ARG;
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  hipStream_t st = (hipStream_t)hip_stream;
  FATAL_UNLESS(hipSetDevice(g->device) == hipSuccess);
  const size_t R = (size_t)n_ranks;
  // control block: [R][3] gathered counts + status | [3] this rank's entry (later its allocation flag) | [R] gathered
  // allocation flags | [2R] point and observation bases
  FATAL_UNLESS(g->cnt_dev.ensure(sizeof(uint64_t) * (6 * R + 4)) == 0);
  uint64_t* cnt = (uint64_t*)g->cnt_dev.p;
  uint64_t* mine_dev = cnt + 3 * R;
  uint64_t* flags_dev = cnt + 3 * R + 3;
  uint64_t* bases_dev = cnt + 4 * R + 3;
  // a local result that views this gather's own result buffers would be overwritten while it is being sent
  const bool usable = local && local->complete && !(local->n_points && g->owns(local->X)) &&
                      !(local->n_obs && g->owns(local->obs_view));
  const uint64_t my_status = usable ? 0u : 1u;
  const uint64_t mine[3] = {my_status ? 0 : local->n_points, my_status ? 0 : local->n_obs, my_status};
  FATAL_UNLESS(hipMemcpyAsync(mine_dev, mine, sizeof(mine), hipMemcpyHostToDevice, st) == hipSuccess);
  FATAL_UNLESS(ncclAllGather(mine_dev, cnt, 3, ncclUint64, comm, st) == ncclSuccess);
  std::vector<uint64_t> h(3 * R), base(2 * R);
  FATAL_UNLESS(hipMemcpyAsync(h.data(), cnt, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost, st) == hipSuccess);
  FATAL_UNLESS(hipStreamSynchronize(st) == hipSuccess);
  uint64_t tp = 0, to = 0;
  const bool all_ok = eg3d_gather_detail::plan(n_ranks, h.data(), base.data(), base.data() + R, &tp, &to);
  for (int r = 0; r < n_ranks; r++) {
    if (rank_points) rank_points[r] = h[3 * r];
    if (rank_obs) rank_obs[r] = h[3 * r + 1];
  }
  if (!all_ok) return EG3D_GATHER_ERR_INCOMPLETE;  // same verdict on every rank
  // ---- every allocation of this call, then agree on their success
  uint64_t alloc_fail = ensure_result(g, tp, to) ? 1 : 0;
  {
    FATAL_UNLESS(hipMemcpyAsync(mine_dev, &alloc_fail, sizeof(uint64_t), hipMemcpyHostToDevice, st) == hipSuccess);
    FATAL_UNLESS(ncclAllGather(mine_dev, flags_dev, 1, ncclUint64, comm, st) == ncclSuccess);
    std::vector<uint64_t> hf(R);
    FATAL_UNLESS(hipMemcpyAsync(hf.data(), flags_dev, sizeof(uint64_t) * R, hipMemcpyDeviceToHost, st) == hipSuccess);
    FATAL_UNLESS(hipStreamSynchronize(st) == hipSuccess);
    for (size_t r = 0; r < R; r++)
      if (hf[r]) return EG3D_GATHER_ERR_HIP;  // same verdict on every rank
  }
  // ---- the exchange, IN PLACE: every field of every rank's cloud travels straight from the producing context's
  // HBM buffers to its final position in the receiving rank's result arrays — one grouped set of point-to-point
  // transfers (xGMI links are point-to-point: each pair of ranks uses its own link, all links busy at once).
  // Nothing is packed or padded, so a rank holds the gathered cloud once and nothing else.
  const uint64_t np = local->n_points, no = local->n_obs;
  const uint64_t CH = g->chunk;
  if (g->mode == EG3D_GATHER_MODE_SENDRECV || n_ranks == 1) {
    for (int f = 0; f < 7; f++) {
      const uint64_t nb = kFieldBytes[f] * (f < 3 ? np : no);
      unsigned char* dst = (unsigned char*)g->field(f)->p + kFieldBytes[f] * (f < 3 ? base[rank] : base[R + rank]);
      if (nb) FATAL_UNLESS(hipMemcpyAsync(dst, field_ptr(local, f), nb, hipMemcpyDeviceToDevice, st) == hipSuccess);
    }
  }
  if (n_ranks > 1 && g->mode == EG3D_GATHER_MODE_SENDRECV) {
    FATAL_UNLESS(ncclGroupStart() == ncclSuccess);
    bool ok = true;
    for (int q = 0; q < n_ranks && ok; q++) {
      if (q == rank) continue;
      for (int f = 0; f < 7 && ok; f++) {
        // zero-size fields are skipped on BOTH sides (the receiver knows the sender's counts); pieces in the same
        // order on both sides
        const uint64_t sb = kFieldBytes[f] * (f < 3 ? np : no);
        const uint64_t rb = kFieldBytes[f] * (f < 3 ? h[3 * q] : h[3 * q + 1]);
        unsigned char* dst = (unsigned char*)g->field(f)->p + kFieldBytes[f] * (f < 3 ? base[q] : base[R + q]);
        const unsigned char* src = (const unsigned char*)field_ptr(local, f);
        for (uint64_t o = 0; o < sb && ok; o += CH)
          ok = ncclSend(src + o, (size_t)std::min<uint64_t>(CH, sb - o), ncclUint8, q, comm, st) == ncclSuccess;
        for (uint64_t o = 0; o < rb && ok; o += CH)
          ok = ncclRecv(dst + o, (size_t)std::min<uint64_t>(CH, rb - o), ncclUint8, q, comm, st) == ncclSuccess;
      }
    }
    const bool ended = ncclGroupEnd() == ncclSuccess;
    FATAL_UNLESS(ok && ended);
  } else if (n_ranks > 1) {
    // Fallback exchange (EG3D_GATHER_MODE=bcast / eg3d_gather_set_mode): the same all-gather-v as R x 7 broadcasts,
    // root q's field going from its context's buffers to its final position on every rank (the root's own copy
    // included: send buffer != receive buffer there). Uses only the library's broadcast collective — the path to take
    // if the grouped point-to-point transfers misbehave on some RCCL build; still nothing packed or padded.
    FATAL_UNLESS(ncclGroupStart() == ncclSuccess);
    bool ok = true;
    for (int q = 0; q < n_ranks && ok; q++)
      for (int f = 0; f < 7 && ok; f++) {
        const uint64_t nb = kFieldBytes[f] * (f < 3 ? h[3 * q] : h[3 * q + 1]);
        unsigned char* dst = (unsigned char*)g->field(f)->p + kFieldBytes[f] * (f < 3 ? base[q] : base[R + q]);
        const unsigned char* src = q == rank ? (const unsigned char*)field_ptr(local, f) : dst;
        for (uint64_t o = 0; o < nb && ok; o += CH)
          ok = ncclBroadcast(src + o, dst + o, (size_t)std::min<uint64_t>(CH, nb - o), ncclUint8, q, comm, st) == ncclSuccess;
      }
    const bool ended = ncclGroupEnd() == ncclSuccess;
    FATAL_UNLESS(ok && ended);
  }
  // ---- rebase the observation offsets of ranks > 0, write the sentinel
  FATAL_UNLESS(hipMemcpyAsync(bases_dev, base.data(), sizeof(uint64_t) * 2 * R, hipMemcpyHostToDevice, st) == hipSuccess);
  if (tp && n_ranks > 1)
    hipLaunchKernelGGL(k_rebase, dim3((unsigned)std::min<uint64_t>(4096, (tp + 255) / 256)), dim3(256), 0, st,
                       (uint64_t*)g->off.p, (const uint64_t*)bases_dev, n_ranks, tp);
  FATAL_UNLESS(hipMemcpyAsync((uint64_t*)g->off.p + tp, &to, sizeof(uint64_t), hipMemcpyHostToDevice, st) == hipSuccess);
  FATAL_UNLESS(hipStreamSynchronize(st) == hipSuccess);  // `base`, `to` leave scope; `local` may be overwritten from here on
  fill_out(g, tp, to, out);
  return 0;
}

// The same placement and rebasing WITHOUT the collective: several clouds resident on this GPU (the results of
// several contexts, or of several steps) become one ordered cloud, part order = seed order — part r takes the
// place rank r's cloud has in the exchange (device-to-device copies instead of the transfers), so the r > 0 path
// of the N-rank exchange runs on a single GPU.
extern "C" int eg3d_concat_edgepoints(eg3d_gather* g, int n_parts, const eg3d_device_edgepoints* parts, void* hip_stream,
                                      eg3d_device_edgepoints* out) {
  if (!g || !parts || !out || n_parts < 1) return EG3D_GATHER_ERR_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  if (hipSetDevice(g->device) != hipSuccess) return EG3D_GATHER_ERR_HIP;
  const size_t R = (size_t)n_parts;
  std::vector<uint64_t> h(3 * R), base(2 * R);
  for (size_t r = 0; r < R; r++) {
    // a part that views this gather's own result buffers (the `out` of an earlier call on `g`) would be freed or
    // overwritten below: refused
    if ((parts[r].n_points && g->owns(parts[r].X)) || (parts[r].n_obs && g->owns(parts[r].obs_view))) return EG3D_GATHER_ERR_ARG;
    h[3 * r] = parts[r].n_points;
    h[3 * r + 1] = parts[r].n_obs;
    h[3 * r + 2] = parts[r].complete ? 0 : 1;
  }
  uint64_t tp = 0, to = 0;
  if (!eg3d_gather_detail::plan(n_parts, h.data(), base.data(), base.data() + R, &tp, &to)) return EG3D_GATHER_ERR_INCOMPLETE;
  if (g->cnt_dev.ensure(sizeof(uint64_t) * (6 * R + 4)) || ensure_result(g, tp, to)) return EG3D_GATHER_ERR_HIP;
  uint64_t* bases_dev = (uint64_t*)g->cnt_dev.p;
  for (size_t r = 0; r < R; r++)
    for (int f = 0; f < 7; f++) {
      const uint64_t nb = kFieldBytes[f] * (f < 3 ? parts[r].n_points : parts[r].n_obs);
      unsigned char* dst = (unsigned char*)g->field(f)->p + kFieldBytes[f] * (f < 3 ? base[r] : base[R + r]);
      if (nb && hipMemcpyAsync(dst, field_ptr(&parts[r], f), nb, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return EG3D_GATHER_ERR_HIP;
    }
  if (hipMemcpyAsync(bases_dev, base.data(), sizeof(uint64_t) * 2 * R, hipMemcpyHostToDevice, st) != hipSuccess)
    return EG3D_GATHER_ERR_HIP;
  if (tp && n_parts > 1)
    hipLaunchKernelGGL(k_rebase, dim3((unsigned)std::min<uint64_t>(4096, (tp + 255) / 256)), dim3(256), 0, st,
                       (uint64_t*)g->off.p, (const uint64_t*)bases_dev, n_parts, tp);
  if (hipMemcpyAsync((uint64_t*)g->off.p + tp, &to, sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return EG3D_GATHER_ERR_HIP;
  fill_out(g, tp, to, out);
  return 0;
}
