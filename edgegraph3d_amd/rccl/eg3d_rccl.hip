// eg3d_rccl.hip — RCCL all-gather of the edge-point cloud (include/eg3d_rccl.h). One process per GPU;
// xGMI is point-to-point, so the cloud travels as ONE large padded message per rank (every link
// busy at once) instead of seven small ones, and is compacted on the device afterwards.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/eg3d_rccl.h"

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

// packed layout of one rank's block, padded to the largest rank: field f starts at off[f]
struct Layout {
  uint64_t mp, mo;     // padded points / observations per rank
  uint64_t off[7];     // X, obs_off, key | obs_view, obs_pl, obs_seg, obs_xy
  uint64_t bytes;
};
static Layout make_layout(uint64_t mp, uint64_t mo) {
  static const uint64_t per[7] = {12, 4, 16, 4, 4, 4, 8};
  Layout L;
  L.mp = mp;
  L.mo = mo;
  uint64_t o = 0;
  for (int f = 0; f < 7; f++) {
    L.off[f] = o;
    o += per[f] * (f < 3 ? mp : mo);
    o = (o + 15) & ~15ull;
  }
  L.bytes = std::max<uint64_t>(o, 16);
  return L;
}

// one block per (rank, field): copy the rank's valid prefix to its place in the ordered cloud
__global__ void k_unpack(const unsigned char* recv, Layout L, const uint64_t* counts /*[R][3]: points, observations, status*/, int n_ranks,
                         float* X, uint32_t* obs_off, uint32_t* key, int32_t* obs_view, uint32_t* obs_pl,
                         uint32_t* obs_seg, float* obs_xy) {
  const int r = blockIdx.y, f = blockIdx.z;
  uint64_t pbase = 0, obase = 0;
  for (int q = 0; q < r; q++) {
    pbase += counts[3 * q];
    obase += counts[3 * q + 1];
  }
  const uint64_t np = counts[3 * r], no = counts[3 * r + 1];
  const unsigned char* src = recv + (size_t)r * L.bytes + L.off[f];
  const uint64_t words = f == 0 ? np * 3 : f == 1 ? np : f == 2 ? np * 4 : f == 6 ? no * 2 : no;
  uint32_t* dst = f == 0   ? (uint32_t*)X + pbase * 3
                  : f == 1 ? obs_off + pbase
                  : f == 2 ? key + pbase * 4
                  : f == 3 ? (uint32_t*)obs_view + obase
                  : f == 4 ? obs_pl + obase
                  : f == 5 ? obs_seg + obase
                           : (uint32_t*)obs_xy + obase * 2;
  const uint32_t add = f == 1 ? (uint32_t)obase : 0u;  // observation offsets index the gathered arrays
  const uint32_t* s32 = (const uint32_t*)src;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = s32[i] + add;
}

}  // namespace

struct eg3d_gather {
  int device = 0;
  Buf send, recv, cnt_dev, X, off, key, view, pl, seg, xy;
  hipEvent_t pack_done = nullptr;
};

extern "C" eg3d_gather* eg3d_gather_create(int device) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  eg3d_gather* g = new eg3d_gather();
  g->device = device;
  if (hipEventCreateWithFlags(&g->pack_done, hipEventDisableTiming) != hipSuccess) g->pack_done = nullptr;
  return g;
}
extern "C" void eg3d_gather_destroy(eg3d_gather* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->pack_done) (void)hipEventDestroy(g->pack_done);
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
extern "C" void eg3d_comm_destroy(void* comm) {
  if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

#define TRY_HIP(e)                   \
  do {                               \
    if ((e) != hipSuccess) return EG3D_GATHER_ERR_HIP; \
  } while (0)
#define TRY_NCCL(e)                   \
  do {                                \
    if ((e) != ncclSuccess) return EG3D_GATHER_ERR_NCCL; \
  } while (0)

extern "C" int eg3d_allgather_edgepoints(eg3d_gather* g, void* nccl_comm, int n_ranks, int rank, void* hip_stream,
                                         const eg3d_device_edgepoints* local, eg3d_device_edgepoints* out,
                                         uint64_t* rank_points, uint64_t* rank_obs) {
  // Argument errors that every rank sees identically (same call on every rank) may return at once;
  // anything RANK-LOCAL (an incomplete local result, a failed allocation) must not: the other ranks
  // would block in the next collective. Such conditions travel as a status word inside the first
  // (counts) all-gather and in a second 8-byte all-gather after the allocations, so that all ranks
  // agree to abort BEFORE the payload collective.
  if (!g || !nccl_comm || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return EG3D_GATHER_ERR_ARG;
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  hipStream_t st = (hipStream_t)hip_stream;
  TRY_HIP(hipSetDevice(g->device));
  // ---- counts + status: [n_points, n_obs, status] per rank. The one allocation made before any
  // collective is this call's 32*(R+1)-byte control block; a rank that cannot even get that has no
  // way of telling its peers (documented in the header as the unrecoverable case).
  const size_t R = (size_t)n_ranks;
  if (g->cnt_dev.ensure(sizeof(uint64_t) * (4 * R + 4))) return EG3D_GATHER_ERR_HIP;
  uint64_t* cnt = (uint64_t*)g->cnt_dev.p;      // [R][3] gathered counts + status
  uint64_t* mine_dev = cnt + 3 * R;             // [3] this rank's entry / [1] its allocation flag
  uint64_t* flags_dev = cnt + 3 * R + 3;        // [R] gathered allocation flags
  const uint64_t my_status = (local && local->complete) ? 0u : 1u;
  const uint64_t mine[3] = {my_status ? 0 : local->n_points, my_status ? 0 : local->n_obs, my_status};
  TRY_HIP(hipMemcpyAsync(mine_dev, mine, sizeof(mine), hipMemcpyHostToDevice, st));
  TRY_NCCL(ncclAllGather(mine_dev, cnt, 3, ncclUint64, comm, st));
  std::vector<uint64_t> h(3 * R);
  TRY_HIP(hipMemcpyAsync(h.data(), cnt, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost, st));
  TRY_HIP(hipStreamSynchronize(st));
  uint64_t mp = 0, mo = 0, tp = 0, to = 0, bad = 0;
  for (int r = 0; r < n_ranks; r++) {
    mp = std::max(mp, h[3 * r]);
    mo = std::max(mo, h[3 * r + 1]);
    tp += h[3 * r];
    to += h[3 * r + 1];
    bad |= h[3 * r + 2];
    if (rank_points) rank_points[r] = h[3 * r];
    if (rank_obs) rank_obs[r] = h[3 * r + 1];
  }
  if (bad) return EG3D_GATHER_ERR_INCOMPLETE;  // same verdict on every rank
  if (to > 0xffffffffull) return EG3D_GATHER_ERR_RANGE;  // observation offsets are 32-bit (same on every rank)
  const Layout L = make_layout(mp, mo);
  // ---- every allocation of this call, then agree on their success
  uint64_t alloc_fail = 0;
  if (g->send.ensure(L.bytes) || g->recv.ensure(L.bytes * R) || g->X.ensure(tp * 12 + 16) || g->off.ensure((tp + 1) * 4) ||
      g->key.ensure(tp * 16 + 16) || g->view.ensure(to * 4 + 16) || g->pl.ensure(to * 4 + 16) ||
      g->seg.ensure(to * 4 + 16) || g->xy.ensure(to * 8 + 16))
    alloc_fail = 1;
  {
    TRY_HIP(hipMemcpyAsync(mine_dev, &alloc_fail, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    TRY_NCCL(ncclAllGather(mine_dev, flags_dev, 1, ncclUint64, comm, st));
    std::vector<uint64_t> hf(R);
    TRY_HIP(hipMemcpyAsync(hf.data(), flags_dev, sizeof(uint64_t) * R, hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    for (size_t r = 0; r < R; r++)
      if (hf[r]) return EG3D_GATHER_ERR_HIP;  // same verdict on every rank
  }
  // ---- pack this rank's block and gather
  unsigned char* sb = (unsigned char*)g->send.p;
  const uint64_t np = local->n_points, no = local->n_obs;
  const void* src[7] = {local->X, local->obs_off, local->key, local->obs_view, local->obs_pl, local->obs_seg, local->obs_xy};
  const uint64_t nbytes[7] = {np * 12, np * 4, np * 16, no * 4, no * 4, no * 4, no * 8};
  for (int f = 0; f < 7; f++)
    if (nbytes[f]) TRY_HIP(hipMemcpyAsync(sb + L.off[f], src[f], nbytes[f], hipMemcpyDeviceToDevice, st));
  if (g->pack_done) TRY_HIP(hipEventRecord(g->pack_done, st));  // `local` may be overwritten once this has fired
  TRY_NCCL(ncclAllGather(sb, g->recv.p, L.bytes, ncclUint8, comm, st));
  // ---- compaction into one ordered cloud
  hipLaunchKernelGGL(k_unpack, dim3(64, n_ranks, 7), dim3(256), 0, st, (const unsigned char*)g->recv.p, L,
                     (const uint64_t*)cnt, n_ranks, (float*)g->X.p, (uint32_t*)g->off.p, (uint32_t*)g->key.p,
                     (int32_t*)g->view.p, (uint32_t*)g->pl.p, (uint32_t*)g->seg.p, (float*)g->xy.p);
  const uint32_t last = (uint32_t)to;
  TRY_HIP(hipMemcpyAsync((uint32_t*)g->off.p + tp, &last, 4, hipMemcpyHostToDevice, st));
  TRY_HIP(hipStreamSynchronize(st));
  out->n_points = tp;
  out->n_obs = to;
  out->X = (const float*)g->X.p;
  out->obs_off = (const uint32_t*)g->off.p;
  out->obs_view = (const int32_t*)g->view.p;
  out->obs_pl = (const uint32_t*)g->pl.p;
  out->obs_seg = (const uint32_t*)g->seg.p;
  out->obs_xy = (const float*)g->xy.p;
  out->key = (const uint32_t*)g->key.p;
  out->complete = 1;
  return 0;
}

// The gather's pack layout and compaction kernel WITHOUT the collective: several clouds resident on this GPU (the
// results of several contexts, or of several steps) become one ordered cloud, part order = seed order. Every
// part is packed into its padded slot exactly as a rank packs its block, then k_unpack rebases and compacts —
// so the r > 0 path of the N-rank exchange runs on a single GPU.
extern "C" int eg3d_concat_edgepoints(eg3d_gather* g, int n_parts, const eg3d_device_edgepoints* parts, void* hip_stream,
                                      eg3d_device_edgepoints* out) {
  if (!g || !parts || !out || n_parts < 1) return EG3D_GATHER_ERR_ARG;
  hipStream_t st = (hipStream_t)hip_stream;
  TRY_HIP(hipSetDevice(g->device));
  const size_t R = (size_t)n_parts;
  std::vector<uint64_t> h(3 * R);
  uint64_t mp = 0, mo = 0, tp = 0, to = 0;
  for (size_t r = 0; r < R; r++) {
    if (!parts[r].complete) return EG3D_GATHER_ERR_INCOMPLETE;
    h[3 * r] = parts[r].n_points;
    h[3 * r + 1] = parts[r].n_obs;
    h[3 * r + 2] = 0;
    mp = std::max(mp, h[3 * r]);
    mo = std::max(mo, h[3 * r + 1]);
    tp += h[3 * r];
    to += h[3 * r + 1];
  }
  if (to > 0xffffffffull) return EG3D_GATHER_ERR_RANGE;
  const Layout L = make_layout(mp, mo);
  if (g->cnt_dev.ensure(sizeof(uint64_t) * (4 * R + 4)) || g->recv.ensure(L.bytes * R) || g->X.ensure(tp * 12 + 16) ||
      g->off.ensure((tp + 1) * 4) || g->key.ensure(tp * 16 + 16) || g->view.ensure(to * 4 + 16) ||
      g->pl.ensure(to * 4 + 16) || g->seg.ensure(to * 4 + 16) || g->xy.ensure(to * 8 + 16))
    return EG3D_GATHER_ERR_HIP;
  TRY_HIP(hipMemcpyAsync(g->cnt_dev.p, h.data(), sizeof(uint64_t) * h.size(), hipMemcpyHostToDevice, st));
  for (size_t r = 0; r < R; r++) {
    unsigned char* slot = (unsigned char*)g->recv.p + r * L.bytes;
    const uint64_t np = parts[r].n_points, no = parts[r].n_obs;
    const void* src[7] = {parts[r].X, parts[r].obs_off, parts[r].key, parts[r].obs_view, parts[r].obs_pl, parts[r].obs_seg,
                          parts[r].obs_xy};
    const uint64_t nbytes[7] = {np * 12, np * 4, np * 16, no * 4, no * 4, no * 4, no * 8};
    for (int f = 0; f < 7; f++)
      if (nbytes[f]) TRY_HIP(hipMemcpyAsync(slot + L.off[f], src[f], nbytes[f], hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(k_unpack, dim3(64, n_parts, 7), dim3(256), 0, st, (const unsigned char*)g->recv.p, L,
                     (const uint64_t*)g->cnt_dev.p, n_parts, (float*)g->X.p, (uint32_t*)g->off.p, (uint32_t*)g->key.p,
                     (int32_t*)g->view.p, (uint32_t*)g->pl.p, (uint32_t*)g->seg.p, (float*)g->xy.p);
  const uint32_t last = (uint32_t)to;
  TRY_HIP(hipMemcpyAsync((uint32_t*)g->off.p + tp, &last, 4, hipMemcpyHostToDevice, st));
  TRY_HIP(hipStreamSynchronize(st));
  out->n_points = tp;
  out->n_obs = to;
  out->X = (const float*)g->X.p;
  out->obs_off = (const uint32_t*)g->off.p;
  out->obs_view = (const int32_t*)g->view.p;
  out->obs_pl = (const uint32_t*)g->pl.p;
  out->obs_seg = (const uint32_t*)g->seg.p;
  out->obs_xy = (const float*)g->xy.p;
  out->key = (const uint32_t*)g->key.p;
  out->complete = 1;
  return 0;
}

extern "C" int eg3d_gather_wait_pack(eg3d_gather* g) {
  if (!g || !g->pack_done) return EG3D_GATHER_ERR_ARG;
  return hipEventSynchronize(g->pack_done) == hipSuccess ? 0 : EG3D_GATHER_ERR_HIP;
}
