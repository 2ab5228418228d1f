// eg3d_api.hip — implementation of the C ABI declared in include/eg3d.h.
//
// Host orchestration of the phase pipeline: HBM-resident scene (cameras, F, polyline CSR, the
// two uniform grids), resident seeds, grow-only device work buffers, rocPRIM/hipCUB exclusive
// scans between phases, HIP events for per-stage timing. No CPU fallback exists: every entry
// point fails with EG3D_ERR_NODEVICE / EG3D_ERR_HIP when no gfx950 device is usable.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eg3d.h"
#include "../../include/eg3d_host.h"
#include "eg3d_host_copy.h"
#include "eg3d_kernels.h"

using namespace eg3d;

static thread_local std::string g_err;
extern "C" const char* eg3d_last_error(void) { return g_err.c_str(); }

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      g_err = std::string(#expr) + ": " + hipGetErrorString(_e);                                    \
      return EG3D_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap && p) return EG3D_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      cap = 0;
      g_err = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
      return EG3D_ERR_HIP;
    }
    cap = want;
    return EG3D_OK;
  }
  // grow, keeping the first `keep` bytes (device-to-device copy on `st`, old block freed once it is done)
  int ensure_keep(size_t bytes, size_t keep, hipStream_t st) {
    if (bytes <= cap && p) return EG3D_OK;
    if (!p || !keep) return ensure(bytes);
    void* q = nullptr;
    const size_t want = bytes + bytes / 2;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) {
      g_err = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
      return EG3D_ERR_HIP;
    }
    e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(q);
      g_err = std::string("growing a device buffer: ") + hipGetErrorString(e);
      return EG3D_ERR_HIP;
    }
    (void)hipFree(p);
    p = q;
    cap = want;
    return EG3D_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

#define BUF_TRY(expr)          \
  do {                         \
    int _r = (expr);           \
    if (_r != EG3D_OK) return _r; \
  } while (0)

// Device allocations shared by a context and its clones (immutable scene / resident seeds): freed
// when the last context referring to them goes away.
struct DevOwner {
  int device = 0;
  std::vector<void*> ptrs;
  ~DevOwner() {
    (void)hipSetDevice(device);
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
  }
};
// Host copies of the grids for eg3d_get_grid (per view CSR with view-local offsets). The grids live on the device (K0 builds
// them there); the copies are made by the first eg3d_get_grid call that asks for a cell size — tests do, the hot path never.
struct HostGrids {
  std::mutex mu;
  bool have[2] = {false, false};
  std::vector<std::vector<uint32_t>> h_off[2], h_ids[2];
  // where to fetch them from (device arrays of the shared, immutable scene)
  int device = 0, n_views = 0;
  const uint32_t* d_off[2] = {nullptr, nullptr};
  const uint32_t* d_ids[2] = {nullptr, nullptr};
  uint32_t cells_per_view[2] = {0, 0};
};

// Test / tuning knobs, read from the environment ONCE when a context is created (eg3d_create; clones
// inherit them) — the hot path never calls getenv:
//   EG3D_K3A_ENGINE_WAVES=n  wavefronts per SIMD the K3a engine launches (default 2 = what its 256-VGPR build allows)
//   EG3D_K3A_ENGINE_LANES=n  lanes of a K3a wavefront that take work (default: 64, fewer for small batches)
#ifndef EG3D_K3B_ENGINE_DEFAULT
#define EG3D_K3B_ENGINE_DEFAULT 0
#endif
//   EG3D_HYP_CAP=n         tests: points per following direction of the hypothesis stage (default 160; a list that would
//                          outgrow it raises EG3D_FLAG_HYP_OVERFLOW and the call returns EG3D_ERR_CAPACITY)
//   EG3D_SLOTS_PER_XCD=n   tests: working slices of the expand stage per XCD (default: what can be resident + margin)
//   EG3D_TRACE_ARENA=1     print the hypothesis arena's use per batch to stderr
//   EG3D_ARENA_CAP0=n      initial hypothesis arena capacity (tests: forces the overflow-and-retry path)
//   EG3D_MAX_SCRATCH_MB=n  tests: cut the chains of a batch into several K3b launches of at most n MB / slice size
//                          chains each (default: one launch takes all chains — their working slices are slots)
//   EG3D_NO_LPT=1          launch chains in identity order instead of longest-first (diagnostic)
//   EG3D_K3B_FULL=1        always run the general build of the expand kernel (default: the build for the scene's class —
//                          polylines of <= 512 vertices and <= 28 views: small, >= 29 views: many views; general otherwise)
//   EG3D_K3B_ENGINE=0|1    (builds with -DEG3D_WITH_K3C_ENGINE only: variants/libeg3d_engine.so) expand stage: 1 = the
//                          lane-per-chain engine (k3c_engine, eg3d_k3c_engine.h), 0 = one wavefront per chain (k3b_expand).
//                          EG3D_K3C_WAVES=n waves per SIMD of the engine's grid, EG3D_K3C_LANES=n lanes of a wave that own a
//                          chain (default: as many waves as fit, then as few lanes as cover the chains). A library built
//                          without the engine refuses EG3D_K3B_ENGINE=1 at eg3d_create.
//   EG3D_PIPELINE_LANES=n  sub-batches of ONE eg3d_match_* call kept in flight on internal contexts (default 0 = by the kind of
//                          call: 3 for a call that copies its cloud to the host, 1 for a device-only call; 1 = the call
//                          runs as a single batch on the context's own stream). EG3D_PIPELINE_UNITS=n: sub-batches the call's
//                          range is cut into (default: chosen from the range, see plan_seed_units). eg3d_set_pipelining
//                          overrides both. Chosen by measurement (profiles/r06_experiments/pipelining_*.json):
//   EG3D_UNIT_RAMP=r       unit i of a seed call gets a share ~ r^i of the range (default 0.6: the LAST unit, whose D2H copy
//                          nothing can hide, is the smallest); EG3D_LANE_PRIORITIES=0|1: lane 0's stream high priority, lane
//                          1 normal, the others low (default 1: the earlier units finish — and cross PCIe — first)
//   EG3D_TEST_FAIL_UNIT=k  tests: the k-th unit (1-based) of every pipelined call fails when its turn to place comes
struct Tunables {
  bool grid_on_host = false;  // EG3D_GRID_ON_HOST=1 (diagnostic / A-B): build the uniform grids with the host builder on threads
                              // (rounds 1-5, and round 6 before K0) instead of on the device
  int lanes = 0, units = 0, test_fail_unit = 0;
  int copy_threads = 0;  // EG3D_COPY_THREADS_PER_LANE: host threads that copy one piece of a cloud from the ring to the caller's
                         // arrays (0 = EG3D_COPY_THREADS shared by the lanes of the call: 16 on one lane, 5 each on three)
  double unit_ramp = 0.6;
  int lane_priorities = 1;
  static constexpr int kHostCallLanes = 3;  // lanes = 0: a host call's default
  int k3a_engine_waves = 0, k3a_engine_lanes = 0;
  int k3b_engine = EG3D_K3B_ENGINE_DEFAULT, k3c_waves = 0, k3c_lanes = 0;
  bool assume_short = false;  // EG3D_K3B_ASSUME_SHORT=1 (tests): start with the few-views builds whatever the view count, so that
                              // the CTR_LONG_REFUSED -> general build retry runs
  bool trace_arena = false;
  bool k3b_full = false;  // EG3D_K3B_FULL=1: always the full expand kernel (diagnostic)
  uint32_t arena_cap0 = 0, hyp_cap = 0;
  uint32_t chain_cap0 = 0, pool_cap0 = 0;  // EG3D_CHAIN_CAP0 / EG3D_POOL_CAP0 (tests): initial points / observation slots per chain,
                                           // small enough to force the relaunch-what-overflowed path several times
  size_t max_scratch = 0;  // 0 = no limit
  uint32_t slots_per_xcd = 0;  // 0 = sized from the occupancy query
  bool use_lpt = true;
  static Tunables from_env() {
    Tunables t;
    if (const char* e = getenv("EG3D_K3A_ENGINE_WAVES")) t.k3a_engine_waves = atoi(e);
    if (const char* e = getenv("EG3D_K3A_ENGINE_LANES")) t.k3a_engine_lanes = atoi(e);
    if (const char* e = getenv("EG3D_K3B_ENGINE")) t.k3b_engine = atoi(e);
    if (const char* e = getenv("EG3D_K3C_WAVES")) t.k3c_waves = atoi(e);
    if (const char* e = getenv("EG3D_K3C_LANES")) t.k3c_lanes = atoi(e);
    if (const char* e = getenv("EG3D_K3B_ASSUME_SHORT")) t.assume_short = e[0] == '1';
    if (const char* e = getenv("EG3D_HYP_CAP")) t.hyp_cap = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("EG3D_ARENA_CAP0")) t.arena_cap0 = (uint32_t)std::max(16, atoi(e));
    if (const char* e = getenv("EG3D_CHAIN_CAP0")) t.chain_cap0 = (uint32_t)std::max(8, atoi(e));
    if (const char* e = getenv("EG3D_POOL_CAP0")) t.pool_cap0 = (uint32_t)std::max(64, atoi(e));
    if (const char* e = getenv("EG3D_MAX_SCRATCH_MB")) t.max_scratch = (size_t)std::max(1, atoi(e)) << 20;
    if (const char* e = getenv("EG3D_NO_LPT")) t.use_lpt = !(e[0] == '1');
    if (const char* e = getenv("EG3D_TRACE_ARENA")) t.trace_arena = e[0] == '1';
    if (const char* e = getenv("EG3D_K3B_FULL")) t.k3b_full = e[0] == '1';
    if (const char* e = getenv("EG3D_SLOTS_PER_XCD")) t.slots_per_xcd = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("EG3D_PIPELINE_LANES")) t.lanes = std::min(16, std::max(0, atoi(e)));
    if (const char* e = getenv("EG3D_TEST_FAIL_UNIT")) t.test_fail_unit = atoi(e);
    if (const char* e = getenv("EG3D_GRID_ON_HOST")) t.grid_on_host = e[0] == '1';
    if (const char* e = getenv("EG3D_COPY_THREADS_PER_LANE")) t.copy_threads = std::min(32, std::max(1, atoi(e)));
    if (const char* e = getenv("EG3D_LANE_PRIORITIES")) t.lane_priorities = atoi(e);
    if (const char* e = getenv("EG3D_UNIT_RAMP")) t.unit_ramp = std::min(16.0, std::max(1.0 / 16.0, atof(e)));
    if (const char* e = getenv("EG3D_PIPELINE_UNITS")) t.units = std::min(4096, std::max(0, atoi(e)));
    return t;
  }
};

struct eg3d_ctx {
  int device = 0;
  Tunables tune;
  hipStream_t stream = nullptr;
  int V = 0, W = 0, H = 0;
  DevScene ds;
  DevBuf b_camP, b_F, b_Fv, b_vpo, b_pvo, b_vtx, b_pls, b_ple, b_g30o, b_g30i, b_g4o, b_g4i, b_bbo, b_bb;
  std::shared_ptr<DevOwner> scene_owner;  // owns b_camP .. b_g4i
  // host copies of the grids for eg3d_get_grid (per view CSR with view-local offsets)
  std::shared_ptr<HostGrids> hg;
  uint32_t gw[2] = {0, 0}, gh[2] = {0, 0};
  uint32_t grid_dropped = 0;
  // resident seeds
  uint32_t n_seeds = 0;
  std::shared_ptr<std::vector<uint32_t>> h_trk;
  DevBuf b_toff, b_tview, b_txy;
  std::shared_ptr<DevOwner> seeds_owner;  // owns b_toff, b_tview, b_txy
  // work buffers
  DevBuf b_sv_seed, b_map_view, b_map_entry, b_map_n, b_raw_cnt, b_raw_off, b_cand_pl, b_start_hits, b_cand_cnt,
      b_start_cnt, b_task_off, b_task_seed, b_task_entry, b_task_hit, b_task_k, b_task_list_off, b_list_cnt, b_list_ptr,
      b_hits, b_tasks, b_nhyp, b_hyp_off, b_res, b_arena, b_ctr, b_cs_task, b_valid, b_chain_off, b_chains,
      b_cscratch, b_couts, b_cpts, b_cobs, b_cpoff, b_cooff, b_scan_tmp, b_scanchk, b_cost, b_cidx, b_cost2, b_order, b_redo[2];
  DevBuf o_X, o_off, o_view, o_pl, o_seg, o_xy, o_key;
  DevBuf f_X, f_off, f_view, f_xy, f_Xo, f_inl;
  DevBuf b_sets_off, b_sets_ids;  // polyline sets of the current eg3d_match_polyline_sets call
  DevBuf b_fscratch, b_queue, b_items;  // K3a following: per-lane staging lists, work-queue heads, the lists to follow
  // K3b: working slices of the resident chains (b_cscratch: 8 XCDs x slots_per_xcd slices), the slot pools,
  // and the staging area finished chains are packed into (sized from the previous launches; grow-only)
  DevBuf b_pools, b_stage_pts, b_stage_obs, b_stage_used;
  uint32_t slots_per_xcd = 0;
  int k3c_per_cu = 0;            // resident blocks per CU of the lane-per-chain engine (occupancy query)
  bool k3b_long_latched = false; // a launch of a few-views build met a solve of > 32 rows: the context runs the general builds from then on
  uint32_t max_pl_vtx = 0;  // vertices of the scene's longest valid polyline
  uint64_t stage_cap_pts = 0, stage_cap_obs = 0;
  hipEvent_t ea[8], eb[8];  // begin/end events per stage: 1 K1, 2 K2, 3 K3a, 4 K3s, 5 K3b, 6 K4, 0 misc, 7 whole call
  hipEvent_t ecopy[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // D2H of the cloud: one per ring buffer (EG3D_D2H_RING <= 7)
  uint32_t chain_cap = 384, pool_cap = 0, hyp_cap = 160;
  uint32_t n_simd = 0;  // SIMDs of the device (4 per CU): sizes the K3a engine's launch
  int wall_clock_khz = 0;  // rate of wall_clock64() on the device (hipDeviceAttributeWallClockRate)
  double arena_per_hyp = 16.0;  // hypothesis arena: points per hypothesis to reserve (learned from overflows)
  void* pinned = nullptr;  // the ring of pinned host buffers the D2H copies of a cloud go through (ensure_d2h_ring)
  size_t pinned_cap = 0, ring_chunk = 0;
  // mailbox for the small read-backs of a step (scan totals, counters): pinned host memory mapped into the
  // GPU's address space, written by k_publish, polled by the calling thread (no driver round trip)
  uint32_t* mbox = nullptr;
  uint32_t* mbox_dev = nullptr;
  uint32_t mbox_seq = 0;
  uint64_t last_np = 0, last_no = 0;
  int last_chunks = 0;
  bool last_accumulated = false;  // the output buffers hold the whole cloud of the last call (device-only calls)
  uint32_t last_nc = 0;
  uint32_t last_nhyp = 0;
  // Internal pipelining of ONE call (run_pipelined): lane 0 is this context, lanes 1.. are clones created on first use
  // (own stream / work buffers, shared scene and seeds). A lane is never handed to the caller.
  std::vector<eg3d_ctx*> lanes;
  bool is_lane = false;
  uint32_t host_calls = 0;  // eg3d_match_* calls with device_only == 0 this context has completed (lanes_for)
  uint64_t last_host_cloud_bytes = 0;  // ... and the size of the last one's cloud
};

// D2H ring of a context: EG3D_D2H_RING pinned buffers of `ring_chunk` bytes (run_stage_b) — 16 MB each for a cloud worth it,
// 2 MB each for small ones (pinning memory costs ~0.2 ms per MB: the big ring is a fifth of a small scene's whole call).
// Allocated by the thread that drives the context, normally right after the expand launch, whose run time hides the pinning
// (the cloud's size is then an estimate from the number of chains; a ring that turns out too small for a big cloud is
// replaced once, at copy time).
#ifndef EG3D_D2H_CHUNK
#define EG3D_D2H_CHUNK ((size_t)16 << 20)
#endif
#define EG3D_D2H_CHUNK_SMALL ((size_t)2 << 20)
#define EG3D_D2H_BIG_CLOUD ((size_t)96 << 20) /* bytes of a unit's cloud from which the 16 MB buffers pay */
#ifndef EG3D_D2H_RING
#define EG3D_D2H_RING 4
#endif
static int ensure_d2h_ring(eg3d_ctx* c, size_t cloud_bytes) {
  const size_t want = cloud_bytes >= EG3D_D2H_BIG_CLOUD ? EG3D_D2H_CHUNK : EG3D_D2H_CHUNK_SMALL;
  if (c->pinned && c->ring_chunk >= want) return EG3D_OK;
  if (c->pinned) (void)hipHostFree(c->pinned);
  c->pinned = nullptr;
  c->ring_chunk = c->pinned_cap = 0;
  HIP_TRY(hipHostMalloc(&c->pinned, want * EG3D_D2H_RING, hipHostMallocDefault));
  c->ring_chunk = want;
  c->pinned_cap = want * EG3D_D2H_RING;
  return EG3D_OK;
}

template <typename T>
static int upload(DevBuf& b, const T* src, size_t n, hipStream_t st) {
  BUF_TRY(b.ensure(sizeof(T) * std::max<size_t>(n, 1)));
  if (n) HIP_TRY(hipMemcpyAsync(b.p, src, sizeof(T) * n, hipMemcpyHostToDevice, st));
  return EG3D_OK;
}

static int scan_exclusive_u32(eg3d_ctx* c, const uint32_t* in, uint32_t* out, size_t n_plus_one) {
  // in[n] must be 0 (or ignored): out[n] = total
  size_t tmp_bytes = 0;
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, in, out, (int)n_plus_one, c->stream));
  BUF_TRY(c->b_scan_tmp.ensure(tmp_bytes));
  HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->b_scan_tmp.p, tmp_bytes, in, out, (int)n_plus_one, c->stream));
  return EG3D_OK;
}
// ---- small read-backs through the mailbox ----------------------------------------------------------
// b_scanchk: [0..3] "scan wrapped" flag words (ORed by k_scan_check, cleared by k_publish), [4..5] a saved
// 64-bit counter.
static int ensure_mailbox(eg3d_ctx* c) {
  if (!c->mbox) {
    void* h = nullptr;
    HIP_TRY(hipHostMalloc(&h, sizeof(uint32_t) * EG3D_MBOX_WORDS, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, sizeof(uint32_t) * EG3D_MBOX_WORDS);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipHostFree(h);
      g_err = "eg3d: hipHostGetDevicePointer failed for the read-back mailbox";
      return EG3D_ERR_HIP;
    }
    c->mbox = (uint32_t*)h;
    c->mbox_dev = (uint32_t*)d;
  }
  if (!c->b_scanchk.p) {
    BUF_TRY(c->b_scanchk.ensure(8 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(c->b_scanchk.p, 0, 8 * sizeof(uint32_t), c->stream));
  }
  return EG3D_OK;
}
struct Readback {
  eg3d_ctx* c;
  PubArgs a{};
  uint32_t off[6] = {0, 0, 0, 0, 0, 0};
  uint32_t used = 2;
  explicit Readback(eg3d_ctx* c_) : c(c_) {}
  int add(const void* dev, uint32_t words) {  // returns the item's index
    const int i = a.n++;
    a.src[i] = (const uint32_t*)dev;
    a.words[i] = words;
    off[i] = used;
    used += words;
    return i;
  }
  void clear_after(uint32_t* dev) { a.clear[a.n_clear++] = dev; }
  const uint32_t* item(int i) const { return c->mbox + off[i]; }
  // Launch the publish kernel behind everything queued on the stream and wait for its data: a short poll of
  // the mailbox (the common case: the GPU is a few microseconds behind), then a blocking wait for long kernels.
  int run() {
    if (a.n > 6 || a.n_clear > 3 || used > EG3D_MBOX_WORDS) {
      g_err = "eg3d: internal: read-back too large";
      return EG3D_ERR_ARG;
    }
    const uint32_t seq = ++c->mbox_seq;
    launch_publish(c->stream, a, c->mbox_dev, seq);
    HIP_TRY(hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
      if (__atomic_load_n(c->mbox, __ATOMIC_ACQUIRE) == seq) return EG3D_OK;
      __builtin_ia32_pause();
      if ((spin & 255u) == 255u &&
          std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500))
        break;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (__atomic_load_n(c->mbox, __ATOMIC_ACQUIRE) != seq) {
      g_err = "eg3d: the read-back mailbox was not written";
      return EG3D_ERR_HIP;
    }
    return EG3D_OK;
  }
};
// Exclusive scan queued on the stream, its wrap check ORed into flag word `slot`; the total is out[n].
static int scan_queue_u32(eg3d_ctx* c, const uint32_t* in, uint32_t* out, size_t n_plus_one, int slot) {
  BUF_TRY(ensure_mailbox(c));
  BUF_TRY(scan_exclusive_u32(c, in, out, n_plus_one));
  launch_scan_check(c->stream, out, n_plus_one, c->b_scanchk.as<uint32_t>() + slot);
  return EG3D_OK;
}
static int wrapped_error(const char* what) {
  g_err = std::string("eg3d: the number of ") + what + " of this batch exceeds 2^32-1; use smaller seed / set ranges";
  return EG3D_ERR_CAPACITY;
}
// Exclusive scan + its total on the host, with overflow detection: phase totals (candidate slots,
// tasks, lists, hits, hypotheses) are 32-bit; a batch whose total does not fit is refused with
// EG3D_ERR_CAPACITY instead of sizing buffers from a wrapped number.
static int scan_total_u32(eg3d_ctx* c, const uint32_t* in, uint32_t* out, size_t n_plus_one, uint32_t& total,
                          const char* what) {
  BUF_TRY(scan_queue_u32(c, in, out, n_plus_one, 0));
  Readback rb(c);
  const int it = rb.add(out + (n_plus_one - 1), 1);
  const int iw = rb.add(c->b_scanchk.as<uint32_t>(), 1);
  rb.clear_after(c->b_scanchk.as<uint32_t>());
  BUF_TRY(rb.run());
  if (*rb.item(iw)) return wrapped_error(what);
  total = *rb.item(it);
  return EG3D_OK;
}

// K0: both uniform grids of the scene on the device (the scene's polylines are already resident: c->ds). Per cell size:
// count the (cell, polyline) pairs of every polyline, exclusive scan, write them as 64-bit keys, radix sort, unique, CSR.
// Temporaries (24 B per pair) are freed before the function returns.
static int build_grids_device(eg3d_ctx* c, uint32_t NP) {
  hipStream_t st = c->stream;
  const int V = c->V;
  DevBuf t_cnt, t_off, t_keys, t_keys2, t_n;
  struct Release {
    DevBuf* b[5];
    ~Release() {
      for (DevBuf* x : b) x->release();
    }
  } rel{{&t_cnt, &t_off, &t_keys, &t_keys2, &t_n}};
  BUF_TRY(t_n.ensure(2 * sizeof(uint32_t)));  // [0] unique keys, [1] samples outside the image
  HIP_TRY(hipMemsetAsync(t_n.p, 0, 2 * sizeof(uint32_t), st));
  for (int which = 0; which < 2; which++) {
    const float cell = which == 0 ? 30.0f : 4.0f;
    const int map_w = (int)std::ceil(c->W / cell), map_h = (int)std::ceil(c->H / cell);  // (as host/grid_build.cpp: float division)
    c->gw[which] = (uint32_t)map_w;
    c->gh[which] = (uint32_t)map_h;
    const unsigned long long total_cells = (unsigned long long)V * (unsigned long long)map_w * (unsigned long long)map_h;
    if (total_cells >= 0xffffffffull) {
      g_err = "eg3d_create: the scene's grids have more than 2^32-2 cells";
      return EG3D_ERR_CAPACITY;
    }
    DevBuf& g_off = which == 0 ? c->b_g30o : c->b_g4o;
    DevBuf& g_ids = which == 0 ? c->b_g30i : c->b_g4i;
    BUF_TRY(t_cnt.ensure(sizeof(uint32_t) * ((size_t)NP + 1)));
    BUF_TRY(t_off.ensure(sizeof(uint32_t) * ((size_t)NP + 1)));
    HIP_TRY(hipMemsetAsync(t_cnt.as<uint32_t>() + NP, 0, sizeof(uint32_t), st));
    launch_k0_pairs(st, false, c->ds, NP, cell, map_w, map_h, t_cnt.as<uint32_t>(), nullptr, nullptr, t_n.as<uint32_t>() + 1);
    uint32_t n_pairs = 0;
    BUF_TRY(scan_total_u32(c, t_cnt.as<uint32_t>(), t_off.as<uint32_t>(), (size_t)NP + 1, n_pairs, "(cell, polyline) entries of the grids"));
    BUF_TRY(g_off.ensure(sizeof(uint32_t) * ((size_t)total_cells + 1)));
    uint32_t n_unique = 0;
    if (n_pairs) {
      BUF_TRY(t_keys.ensure(sizeof(unsigned long long) * (size_t)n_pairs));
      BUF_TRY(t_keys2.ensure(sizeof(unsigned long long) * (size_t)n_pairs));
      launch_k0_pairs(st, true, c->ds, NP, cell, map_w, map_h, nullptr, t_off.as<uint32_t>(), t_keys.as<unsigned long long>(), nullptr);
      int end_bit = EG3D_K0_PL_BITS_HOST;
      while (end_bit < 64 && (total_cells >> (end_bit - EG3D_K0_PL_BITS_HOST)) != 0) end_bit++;
      size_t tmp = 0;
      HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, t_keys.as<unsigned long long>(), t_keys2.as<unsigned long long>(),
                                                (int)n_pairs, 0, end_bit, st));
      BUF_TRY(c->b_scan_tmp.ensure(tmp));
      HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->b_scan_tmp.p, tmp, t_keys.as<unsigned long long>(), t_keys2.as<unsigned long long>(),
                                                (int)n_pairs, 0, end_bit, st));
      tmp = 0;
      HIP_TRY(hipcub::DeviceSelect::Unique(nullptr, tmp, t_keys2.as<unsigned long long>(), t_keys.as<unsigned long long>(),
                                           t_n.as<uint32_t>(), (int)n_pairs, st));
      BUF_TRY(c->b_scan_tmp.ensure(tmp));
      HIP_TRY(hipcub::DeviceSelect::Unique(c->b_scan_tmp.p, tmp, t_keys2.as<unsigned long long>(), t_keys.as<unsigned long long>(),
                                           t_n.as<uint32_t>(), (int)n_pairs, st));
      Readback rb(c);
      const int in = rb.add(t_n.p, 2);
      BUF_TRY(rb.run());
      n_unique = rb.item(in)[0];
      if (which == 1) c->grid_dropped = rb.item(in)[1];  // (both cell sizes have been counted by now)
    }
    BUF_TRY(g_ids.ensure(sizeof(uint32_t) * std::max<size_t>(n_unique, 1)));
    if (n_unique)
      launch_k0_csr(st, t_keys.as<unsigned long long>(), n_unique, (uint32_t)total_cells, g_off.as<uint32_t>(), g_ids.as<uint32_t>());
    else
      HIP_TRY(hipMemsetAsync(g_off.p, 0, sizeof(uint32_t) * ((size_t)total_cells + 1), st));
    c->hg->d_off[which] = g_off.as<uint32_t>();
    c->hg->d_ids[which] = g_ids.as<uint32_t>();
    c->hg->cells_per_view[which] = (uint32_t)(map_w * map_h);
  }
  HIP_TRY(hipStreamSynchronize(st));  // the temporaries go away
  return EG3D_OK;
}

extern "C" int eg3d_dlt_rows(void) { return EG3D_DLT_ROWS; }

extern "C" int eg3d_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static std::vector<DevBuf*> scene_bufs(eg3d_ctx* c) {
  return {&c->b_bbo,  &c->b_bb,  &c->b_camP, &c->b_F,   &c->b_Fv,   &c->b_vpo,  &c->b_pvo, &c->b_vtx,
          &c->b_pls,  &c->b_ple, &c->b_g30o, &c->b_g30i, &c->b_g4o, &c->b_g4i};
}

// Everything about a scene that can be checked on the host, before a device is touched: the library indexes
// device arrays with these offsets and ids, so an inconsistent scene is refused here instead of faulting there.
static int validate_scene(const eg3d_scene* sc) {
  if (sc->n_views < 1 || sc->width < 1 || sc->height < 1 || !sc->cam_P || !sc->F || !sc->F_valid || !sc->view_pl_off ||
      !sc->pl_vtx_off || !sc->pl_start || !sc->pl_end || !sc->pl_valid) {
    g_err = "eg3d_create: bad arguments (null array, no views or empty image)";
    return EG3D_ERR_ARG;
  }
  if (sc->n_views > EG3D_MAX_VIEWS) {
    g_err = "eg3d_create: more than " + std::to_string(EG3D_MAX_VIEWS) + " views";
    return EG3D_ERR_CAPACITY;
  }
  if (sc->view_pl_off[0] != 0) {
    g_err = "eg3d_create: view_pl_off[0] must be 0";
    return EG3D_ERR_ARG;
  }
  for (int v = 0; v < sc->n_views; v++) {
    if (sc->view_pl_off[v + 1] < sc->view_pl_off[v]) {
      g_err = "eg3d_create: view_pl_off is not ascending";
      return EG3D_ERR_ARG;
    }
    if (sc->view_pl_off[v + 1] - sc->view_pl_off[v] > (uint32_t)EG3D_MAX_POLYLINES_PER_VIEW) {
      g_err = "eg3d_create: a view has more than " + std::to_string(EG3D_MAX_POLYLINES_PER_VIEW) + " polylines";
      return EG3D_ERR_CAPACITY;
    }
  }
  const uint32_t np_all = sc->view_pl_off[sc->n_views];
  if (sc->pl_vtx_off[0] != 0) {
    g_err = "eg3d_create: pl_vtx_off[0] must be 0";
    return EG3D_ERR_ARG;
  }
  for (uint32_t p = 0; p < np_all; p++)
    if (sc->pl_vtx_off[p + 1] < sc->pl_vtx_off[p]) {
      g_err = "eg3d_create: pl_vtx_off is not ascending";
      return EG3D_ERR_ARG;
    }
  if (sc->pl_vtx_off[np_all] && !sc->vtx_xy) {
    g_err = "eg3d_create: vtx_xy is null";
    return EG3D_ERR_ARG;
  }
  // Vertex coordinates of valid polylines must be finite and within +-1e7 px. The grid construction samples
  // every segment each ~2.6 px (polyline_graph_2d.cpp:819-835) whether or not it lies inside the image: a stray
  // 1e20 coordinate would keep the reference (and this library's host grid builder) sampling for years, and a
  // NaN makes its cell conversions undefined. Such a scene is refused instead.
  for (uint32_t p = 0; p < np_all; p++) {
    if (!sc->pl_valid[p]) continue;
    for (size_t k = 2 * (size_t)sc->pl_vtx_off[p]; k < 2 * (size_t)sc->pl_vtx_off[p + 1]; k++)
      if (!(std::fabs(sc->vtx_xy[k]) <= 1e7f)) {
        g_err = "eg3d_create: polyline " + std::to_string(p) + " has a vertex coordinate that is not finite or beyond +-1e7 px";
        return EG3D_ERR_ARG;
      }
  }
  return EG3D_OK;
}

extern "C" int eg3d_create(const eg3d_scene* sc, int device, eg3d_ctx** out) {
  if (!sc || !out) {
    g_err = "eg3d_create: bad arguments";
    return EG3D_ERR_ARG;
  }
  BUF_TRY(validate_scene(sc));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_err = "eg3d_create: no HIP device available (this library has no CPU fallback)";
    return EG3D_ERR_NODEVICE;
  }
  if (device < 0 || device >= ndev) {
    g_err = "eg3d_create: device index out of range";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(device));
  eg3d_ctx* c = new eg3d_ctx();
  c->device = device;
  c->tune = Tunables::from_env();
  if (c->tune.hyp_cap) c->hyp_cap = c->tune.hyp_cap;
  c->hg = std::make_shared<HostGrids>();
  if (c->tune.lane_priorities) {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
  } else
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; i++) {
    HIP_TRY(hipEventCreate(&c->ea[i]));
    HIP_TRY(hipEventCreate(&c->eb[i]));
  }
  for (int i = 0; i < 7; i++) HIP_TRY(hipEventCreateWithFlags(&c->ecopy[i], hipEventDisableTiming));
  const int V = sc->n_views;
  c->V = V;
  c->W = sc->width;
  c->H = sc->height;
  const uint32_t NP = sc->view_pl_off[V];
  // An invalid polyline (pl_valid == 0) is one the reference has invalidated: it keeps its id but its
  // coordinates are cleared (polyline_graph_2d.cpp:1047-1058, Q8). The ABI tolerates a caller that
  // leaves vertices on such a polyline; the device must not see them (the polyline-sets path takes
  // raw polyline ids), so they are dropped here: compacted vertex array, zero-length slices.
  std::vector<uint32_t> pvo_c;
  std::vector<float> vtx_c;
  const uint32_t* pvo_up = sc->pl_vtx_off;
  const float* vtx_up = sc->vtx_xy;
  {
    bool stray = false;
    for (uint32_t p = 0; p < NP && !stray; p++) stray = !sc->pl_valid[p] && sc->pl_vtx_off[p + 1] > sc->pl_vtx_off[p];
    if (stray) {
      pvo_c.assign(1, 0);
      for (uint32_t p = 0; p < NP; p++) {
        if (sc->pl_valid[p])
          vtx_c.insert(vtx_c.end(), sc->vtx_xy + 2 * (size_t)sc->pl_vtx_off[p], sc->vtx_xy + 2 * (size_t)sc->pl_vtx_off[p + 1]);
        pvo_c.push_back((uint32_t)(vtx_c.size() / 2));
      }
      if (vtx_c.empty()) vtx_c.assign(2, 0.f);
      pvo_up = pvo_c.data();
      vtx_up = vtx_c.data();
    }
  }
  const uint32_t NV = pvo_up[NP];
  int rc;
#define UP(buf, ptr, n)                                        \
  if ((rc = upload(c->buf, ptr, (size_t)(n), c->stream)) != EG3D_OK) { \
    eg3d_destroy(c);                                           \
    return rc;                                                 \
  }
  UP(b_camP, sc->cam_P, (size_t)V * 16);
  UP(b_F, sc->F, (size_t)V * V * 9);
  UP(b_Fv, sc->F_valid, (size_t)V * V);
  UP(b_vpo, sc->view_pl_off, V + 1);
  UP(b_pvo, pvo_up, NP + 1);
  UP(b_vtx, vtx_up, (size_t)NV * 2);
  UP(b_pls, sc->pl_start, NP);
  UP(b_ple, sc->pl_end, NP);
  {
    // bounding boxes of blocks of EG3D_BB_SEGS segments (the pre-test of the expand stage's closest-point scans)
    std::vector<uint32_t> bbo(NP + 1, 0);
    std::vector<float> bb;
    for (uint32_t p = 0; p < NP; p++) {
      const uint32_t a = pvo_up[p], b = pvo_up[p + 1];
      const uint32_t nseg = b > a + 1 ? b - a - 1 : 0;
      for (uint32_t s0 = 0; s0 < nseg; s0 += EG3D_BB_SEGS) {
        const uint32_t s1 = std::min(nseg, s0 + (uint32_t)EG3D_BB_SEGS);
        float x0 = vtx_up[2 * (size_t)(a + s0)], y0 = vtx_up[2 * (size_t)(a + s0) + 1], x1 = x0, y1 = y0;
        for (uint32_t i = s0 + 1; i <= s1; i++) {
          const float x = vtx_up[2 * (size_t)(a + i)], y = vtx_up[2 * (size_t)(a + i) + 1];
          x0 = std::min(x0, x);
          y0 = std::min(y0, y);
          x1 = std::max(x1, x);
          y1 = std::max(y1, y);
        }
        bb.insert(bb.end(), {x0, y0, x1, y1});
      }
      bbo[p + 1] = (uint32_t)(bb.size() / 4);
    }
    if (bb.empty()) bb.assign(4, 0.f);
    UP(b_bbo, bbo.data(), bbo.size());
    UP(b_bb, bb.data(), bb.size());
    HIP_TRY(hipStreamSynchronize(c->stream));  // `bbo` / `bb` go out of scope
  }
#undef UP
  DevScene& d = c->ds;
  d.n_views = V;
  d.width = sc->width;
  d.height = sc->height;
  d.cam_P = c->b_camP.as<float>();
  d.F = c->b_F.as<double>();
  d.F_valid = c->b_Fv.as<uint8_t>();
  d.view_pl_off = c->b_vpo.as<uint32_t>();
  d.pl_vtx_off = c->b_pvo.as<uint32_t>();
  d.vtx = c->b_vtx.as<f2>();
  d.pl_start = c->b_pls.as<uint32_t>();
  d.pl_end = c->b_ple.as<uint32_t>();
  d.pl_bb_off = c->b_bbo.as<uint32_t>();
  d.pl_bb = c->b_bb.as<float>();
  // grids (row a3), one CSR over (view, cell) per cell size: built on the DEVICE (K0, build_grids_device) from the polylines
  // just uploaded. EG3D_GRID_ON_HOST=1 (diagnostic) runs the host builder instead — the 2 x V builds are independent
  // (polyLine_2d_map.cpp:40-58 constructs one map per view), so they run on a few host threads; serially, as until round 5,
  // this was 2x the whole hot path of a dtu006-sized job (90 ms for C3', 0.87 s for C4).
  c->hg->device = device;
  c->hg->n_views = V;
  if (!c->tune.grid_on_host) {
    if ((rc = build_grids_device(c, NP)) != EG3D_OK) {
      eg3d_destroy(c);
      return rc;
    }
  } else {
    struct GridJob {
      uint32_t w = 0, h = 0, dropped = 0, *o = nullptr, *i = nullptr;
      int rc = 0;
    };
    std::vector<GridJob> jobs((size_t)2 * V);
    std::atomic<int> next{0};
    auto work = [&]() {
      for (int j; (j = next.fetch_add(1)) < 2 * V;) {
        GridJob& g = jobs[(size_t)j];
        // (the 4 px grids first: they are the long jobs)
        g.rc = eg3d_host_build_grid(sc, j % V, j < V ? 4.0f : 30.0f, &g.w, &g.h, &g.o, &g.i, &g.dropped);
      }
    };
    {
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      const int nthr = (int)std::min<unsigned>(std::min<unsigned>(hw, EG3D_COPY_THREADS), (unsigned)(2 * V));
      std::vector<std::thread> th;
      try {
        for (int t = 1; t < nthr; t++) th.emplace_back(work);
      } catch (...) {  // no more threads: the calling thread does what is left
      }
      work();
      for (auto& t : th) t.join();
    }
    bool ok = true;
    for (auto& g : jobs) ok = ok && g.rc == 0;
    // the per-view arrays are copied for eg3d_get_grid; the device CSR over (view, cell) is assembled on the same threads
    struct FreeJobs {
      std::vector<GridJob>& j;
      ~FreeJobs() {
        for (auto& g : j) {
          free(g.o);
          free(g.i);
        }
      }
    } free_jobs{jobs};
    for (int which = 0; which < 2 && ok; which++) {
      c->hg->h_off[which].resize((size_t)V);
      c->hg->h_ids[which].resize((size_t)V);
      for (int v = 0; v < V; v++) {
        const GridJob& g = jobs[(size_t)(which == 0 ? V + v : v)];
        c->hg->h_off[which][(size_t)v].assign(g.o, g.o + (size_t)g.w * g.h + 1);
        c->hg->h_ids[which][(size_t)v].assign(g.i, g.i + g.o[(size_t)g.w * g.h]);
      }
      c->hg->have[which] = true;
    }
    for (int which = 0; which < 2 && ok; which++) {
      std::vector<size_t> id_base((size_t)V + 1, 0), off_base((size_t)V + 1, 0);
      for (int v = 0; v < V; v++) {
        const GridJob& g = jobs[(size_t)(which == 0 ? V + v : v)];
        c->gw[which] = g.w;
        c->gh[which] = g.h;
        c->grid_dropped += g.dropped;
        id_base[(size_t)v + 1] = id_base[(size_t)v] + g.o[(size_t)g.w * g.h];
        off_base[(size_t)v + 1] = off_base[(size_t)v] + (size_t)g.w * g.h;
      }
      if (id_base[(size_t)V] > 0xffffffffull) {
        g_err = "eg3d_create: the grids of this scene hold more than 2^32-1 (cell, polyline) entries";
        eg3d_destroy(c);
        return EG3D_ERR_CAPACITY;
      }
      std::vector<uint32_t> off(off_base[(size_t)V] + 1), ids(std::max<size_t>(id_base[(size_t)V], 1));
      off[0] = 0;
      std::atomic<int> nextv{0};
      auto fill = [&]() {
        for (int v; (v = nextv.fetch_add(1)) < V;) {
          const GridJob& g = jobs[(size_t)(which == 0 ? V + v : v)];
          const size_t nc = (size_t)g.w * g.h;
          const uint32_t base = (uint32_t)id_base[(size_t)v];
          uint32_t* dst = off.data() + off_base[(size_t)v] + 1;
          for (size_t cc = 0; cc < nc; cc++) dst[cc] = base + g.o[cc + 1];
          if (g.o[nc]) memcpy(ids.data() + id_base[(size_t)v], g.i, sizeof(uint32_t) * g.o[nc]);
        }
      };
      {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int nthr = (int)std::min<unsigned>(std::min<unsigned>(hw, EG3D_COPY_THREADS), (unsigned)V);
        std::vector<std::thread> th;
        try {
          for (int t = 1; t < nthr; t++) th.emplace_back(fill);
        } catch (...) {
        }
        fill();
        for (auto& t : th) t.join();
      }
      if (which == 0) {
        if ((rc = upload(c->b_g30o, off.data(), off.size(), c->stream)) == EG3D_OK) rc = upload(c->b_g30i, ids.data(), ids.size(), c->stream);
      } else {
        if ((rc = upload(c->b_g4o, off.data(), off.size(), c->stream)) == EG3D_OK) rc = upload(c->b_g4i, ids.data(), ids.size(), c->stream);
      }
      if (rc == EG3D_OK && hipStreamSynchronize(c->stream) != hipSuccess) {  // `off`/`ids` go out of scope
        g_err = "eg3d_create: uploading the grids failed";
        rc = EG3D_ERR_HIP;
      }
      if (rc != EG3D_OK) ok = false;
    }
    if (!ok) {
      if (rc == EG3D_OK) {
        g_err = "eg3d_create: grid construction failed";
        rc = EG3D_ERR_ARG;
      }
      eg3d_destroy(c);
      return rc;
    }
  }
  d.g30_w = (int)c->gw[0];
  d.g30_h = (int)c->gh[0];
  d.g4_w = (int)c->gw[1];
  d.g4_h = (int)c->gh[1];
  d.g30_off = c->b_g30o.as<uint32_t>();
  d.g30_ids = c->b_g30i.as<uint32_t>();
  d.g4_off = c->b_g4o.as<uint32_t>();
  d.g4_ids = c->b_g4i.as<uint32_t>();
  {
    bool mid = true;
    for (size_t v = 0; v < (size_t)V && mid; v++)
      for (int k = 0; k < 12 && mid; k++) {
        const float p = sc->cam_P[v * 16 + k];
        if (p != 0.0f) {
          const float a = p < 0 ? -p : p;
          mid = a >= 7.888609052210118e-31f && a <= 1.2676506002282294e+30f;  // 2^-100 .. 2^100 (NaN fails)
        }
      }
    d.cams_mid_range = mid ? 1 : 0;
  }
  // the longest valid polyline of the scene (a scene whose polylines all fit the side walks' LDS staging area, and
  // that has few views, runs the smaller build of the expand kernel: launch_k3b)
  c->max_pl_vtx = 0;
  for (uint32_t p = 0; p < NP; p++)
    if (sc->pl_valid[p]) c->max_pl_vtx = std::max(c->max_pl_vtx, sc->pl_vtx_off[p + 1] - sc->pl_vtx_off[p]);
  // observation slots per chain (blocks double when they fill, so budget ~3x the live count);
  // grown automatically when a chain overflows
  // (many-view scenes: points carry ~V/3 observations and blocks are relocated as they double — C4 settles at 196 608; starting
  // at 24 576 as until round 6 cost the first call three extra rounds of the expand stage)
  c->pool_cap = std::min<uint32_t>(131072, 768u * (uint32_t)std::min(V, 128));
  if (c->pool_cap < 6144) c->pool_cap = 6144;
  if (c->tune.pool_cap0) c->pool_cap = c->tune.pool_cap0;
  if (c->tune.chain_cap0) c->chain_cap = c->tune.chain_cap0;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->n_simd = (uint32_t)prop.multiProcessorCount * 4;
  if (hipDeviceGetAttribute(&c->wall_clock_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) c->wall_clock_khz = 0;
  {
    // working-slice slots per XCD: what can be resident (occupancy query x CUs of an XCD) plus a margin —
    // the occupancy API may be one block per CU off, and a pool must never be smaller than the residency
    const int per_cu = k3b_blocks_per_cu();
    if (per_cu < 1) {
      g_err = "eg3d_create: occupancy query of the expand kernel failed";
      eg3d_destroy(c);
      return EG3D_ERR_HIP;
    }
    // XCDs of THIS device: 8 on the whole MI355X (32 CUs each), fewer when the GPU is partitioned (CPX: one XCD of 32
    // CUs appears as a device, DPX / QPX: four / two) — all its blocks then draw from the pools of those XCDs only, so a
    // pool is sized for CUs / XCDs compute units, not for an eighth of whatever the device reports
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    const uint32_t n_xcd = std::min(8u, std::max(1u, cus / 32u));
    // (at least 40: an XCD of 38 CUs on a part with fewer than 8 XCDs would be undersized by the estimate cus / 32 — the
    // pools cost a few tens of MB more on gfx950, where an XCD has 32)
    const uint32_t cus_per_xcd = std::max<uint32_t>((cus + n_xcd - 1u) / n_xcd, 40u);
    c->slots_per_xcd = ((uint32_t)per_cu + 1u) * cus_per_xcd + 16u;
    if (c->tune.slots_per_xcd) c->slots_per_xcd = c->tune.slots_per_xcd;  // tests / experiments
    // (the lane-per-chain engine is an opt-in second form: only a context that asks for it depends on its kernel)
    if (c->tune.k3b_engine != 0) {
#ifdef EG3D_WITH_K3C_ENGINE
      c->k3c_per_cu = k3c_blocks_per_cu();
      if (c->k3c_per_cu < 1) {
        g_err = "eg3d_create: occupancy query of the expand engine failed";
        eg3d_destroy(c);
        return EG3D_ERR_HIP;
      }
#else
      g_err = "eg3d_create: EG3D_K3B_ENGINE=1, but this library was built without the lane-per-chain engine "
              "(-DEG3D_WITH_K3C_ENGINE: edgegraph3d_amd/variants/libeg3d_engine.so)";
      eg3d_destroy(c);
      return EG3D_ERR_ARG;
#endif
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  // from here on the scene buffers belong to the (shareable) owner, not to this context
  c->scene_owner = std::make_shared<DevOwner>();
  c->scene_owner->device = device;
  for (DevBuf* b : scene_bufs(c)) c->scene_owner->ptrs.push_back(b->p);
  *out = c;
  return EG3D_OK;
}

/* A second context on the same device that SHARES the parent's immutable scene (cameras, F,
 * polylines, grids) and its currently resident seeds, with its own stream, events and work
 * buffers: the way to keep several independent batches in flight (one host thread per context). */
extern "C" int eg3d_clone(eg3d_ctx* parent, eg3d_ctx** out) {
  if (!parent || !out) {
    g_err = "eg3d_clone: bad arguments";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(parent->device));
  eg3d_ctx* c = new eg3d_ctx();
  c->device = parent->device;
  c->tune = parent->tune;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; i++) {
    HIP_TRY(hipEventCreate(&c->ea[i]));
    HIP_TRY(hipEventCreate(&c->eb[i]));
  }
  for (int i = 0; i < 7; i++) HIP_TRY(hipEventCreateWithFlags(&c->ecopy[i], hipEventDisableTiming));
  c->V = parent->V;
  c->W = parent->W;
  c->H = parent->H;
  c->ds = parent->ds;
  {
    std::vector<DevBuf*> src = scene_bufs(parent), dst = scene_bufs(c);
    for (size_t i = 0; i < src.size(); i++) *dst[i] = *src[i];
  }
  c->scene_owner = parent->scene_owner;
  c->hg = parent->hg;
  for (int k = 0; k < 2; k++) {
    c->gw[k] = parent->gw[k];
    c->gh[k] = parent->gh[k];
  }
  c->grid_dropped = parent->grid_dropped;
  c->n_seeds = parent->n_seeds;
  c->h_trk = parent->h_trk;
  c->b_toff = parent->b_toff;
  c->b_tview = parent->b_tview;
  c->b_txy = parent->b_txy;
  c->seeds_owner = parent->seeds_owner;
  c->chain_cap = parent->chain_cap;
  c->pool_cap = parent->pool_cap;
  c->hyp_cap = parent->hyp_cap;
  c->n_simd = parent->n_simd;
  c->wall_clock_khz = parent->wall_clock_khz;
  c->arena_per_hyp = parent->arena_per_hyp;
  c->slots_per_xcd = parent->slots_per_xcd;
  c->k3c_per_cu = parent->k3c_per_cu;
  c->k3b_long_latched = parent->k3b_long_latched;
  c->max_pl_vtx = parent->max_pl_vtx;
  c->stage_cap_pts = parent->stage_cap_pts;  // sizing hints only: the clone allocates its own staging area
  c->stage_cap_obs = parent->stage_cap_obs;
  *out = c;
  return EG3D_OK;
}

extern "C" void eg3d_destroy(eg3d_ctx* c) {
  if (!c) return;
  for (size_t l = 1; l < c->lanes.size(); l++) eg3d_destroy(c->lanes[l]);  // (lane 0 is the context itself)
  c->lanes.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (!c->scene_owner)  // creation failed half-way: the scene buffers are still this context's
    for (DevBuf* b : scene_bufs(c)) b->release();
  c->scene_owner.reset();
  c->seeds_owner.reset();
  DevBuf* all[] = {&c->b_sv_seed, &c->b_map_view,
                   &c->b_map_entry, &c->b_map_n, &c->b_raw_cnt, &c->b_raw_off, &c->b_cand_pl, &c->b_start_hits,
                   &c->b_cand_cnt, &c->b_start_cnt, &c->b_task_off, &c->b_task_seed, &c->b_task_entry, &c->b_task_hit,
                   &c->b_task_k, &c->b_task_list_off, &c->b_list_cnt, &c->b_list_ptr, &c->b_hits, &c->b_tasks,
                   &c->b_nhyp, &c->b_hyp_off, &c->b_res, &c->b_arena, &c->b_ctr, &c->b_cs_task,
                   &c->b_valid, &c->b_chain_off, &c->b_chains, &c->b_cscratch, &c->b_couts, &c->b_cpts, &c->b_cobs,
                   &c->b_cpoff, &c->b_cooff, &c->b_scan_tmp, &c->b_scanchk, &c->b_cost, &c->b_cidx, &c->b_cost2, &c->b_order, &c->b_redo[0], &c->b_redo[1], &c->o_X, &c->o_off, &c->o_view, &c->o_pl, &c->o_seg,
                   &c->o_xy, &c->o_key, &c->f_X, &c->f_off, &c->f_view, &c->f_xy, &c->f_Xo, &c->f_inl, &c->b_sets_off, &c->b_sets_ids, &c->b_fscratch, &c->b_queue, &c->b_items,
                   &c->b_pools, &c->b_stage_pts, &c->b_stage_obs, &c->b_stage_used};
  for (DevBuf* b : all) b->release();
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->mbox) (void)hipHostFree(c->mbox);
  for (int i = 0; i < 8; i++) {
    if (c->ea[i]) (void)hipEventDestroy(c->ea[i]);
    if (c->eb[i]) (void)hipEventDestroy(c->eb[i]);
  }
  for (int i = 0; i < 7; i++)
    if (c->ecopy[i]) (void)hipEventDestroy(c->ecopy[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int eg3d_get_grid(eg3d_ctx* c, int view, int which, uint32_t* ncols, uint32_t* nrows,
                             const uint32_t** cell_off, const uint32_t** ids) {
  if (!c || view < 0 || view >= c->V || which < 0 || which > 1) {
    g_err = "eg3d_get_grid: bad arguments";
    return EG3D_ERR_ARG;
  }
  HostGrids& hg = *c->hg;
  {
    std::lock_guard<std::mutex> lk(hg.mu);
    if (!hg.have[which]) {  // first request for this cell size: fetch the device CSR and cut it into per-view CSRs
      HIP_TRY(hipSetDevice(hg.device));
      const size_t cpv = hg.cells_per_view[which], V = (size_t)hg.n_views;
      std::vector<uint32_t> off(V * cpv + 1);
      HIP_TRY(hipMemcpy(off.data(), hg.d_off[which], sizeof(uint32_t) * off.size(), hipMemcpyDeviceToHost));
      std::vector<uint32_t> all(std::max<size_t>(off.back(), 1));
      if (off.back()) HIP_TRY(hipMemcpy(all.data(), hg.d_ids[which], sizeof(uint32_t) * off.back(), hipMemcpyDeviceToHost));
      hg.h_off[which].resize(V);
      hg.h_ids[which].resize(V);
      for (size_t v = 0; v < V; v++) {
        const uint32_t base = off[v * cpv];
        hg.h_off[which][v].resize(cpv + 1);
        for (size_t cc = 0; cc <= cpv; cc++) hg.h_off[which][v][cc] = off[v * cpv + cc] - base;
        hg.h_ids[which][v].assign(all.begin() + base, all.begin() + off[(v + 1) * cpv]);
      }
      hg.have[which] = true;
    }
  }
  *ncols = c->gw[which];
  *nrows = c->gh[which];
  *cell_off = hg.h_off[which][(size_t)view].data();
  *ids = hg.h_ids[which][(size_t)view].data();
  return EG3D_OK;
}

extern "C" int eg3d_upload_seeds(eg3d_ctx* c, const eg3d_seeds* s) {
  if (!c || !s) {
    g_err = "eg3d_upload_seeds: bad arguments";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (!s->trk_off || (s->n_seeds && s->trk_off[s->n_seeds] && (!s->trk_view || !s->trk_xy))) {
    g_err = "eg3d_upload_seeds: null array";
    return EG3D_ERR_ARG;
  }
  const uint32_t n = s->n_seeds;
  if (s->trk_off[0] != 0) {
    g_err = "eg3d_upload_seeds: trk_off[0] must be 0";
    return EG3D_ERR_ARG;
  }
  for (uint32_t i = 0; i < n; i++)
    if (s->trk_off[i + 1] < s->trk_off[i]) {
      g_err = "eg3d_upload_seeds: trk_off is not ascending";
      return EG3D_ERR_ARG;
    }
  const uint32_t m = s->trk_off[n];
  for (uint32_t i = 0; i < m; i++)
    if (s->trk_view[i] < 0 || s->trk_view[i] >= c->V) {
      g_err = "eg3d_upload_seeds: view id out of range";
      return EG3D_ERR_ARG;
    }
  // fresh buffers: the previous ones may still be in use by clones of this context
  c->seeds_owner.reset();
  c->b_toff = DevBuf();
  c->b_tview = DevBuf();
  c->b_txy = DevBuf();
  c->n_seeds = 0;
  auto owner = std::make_shared<DevOwner>();
  owner->device = c->device;
  int rc = upload(c->b_toff, s->trk_off, n + 1, c->stream);
  if (rc == EG3D_OK) rc = upload(c->b_tview, s->trk_view, m, c->stream);
  if (rc == EG3D_OK) rc = upload(c->b_txy, s->trk_xy, (size_t)m * 2, c->stream);
  owner->ptrs = {c->b_toff.p, c->b_tview.p, c->b_txy.p};
  if (rc != EG3D_OK) {
    c->b_toff = c->b_tview = c->b_txy = DevBuf();
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->seeds_owner = owner;
  c->h_trk = std::make_shared<std::vector<uint32_t>>(s->trk_off, s->trk_off + n + 1);
  c->n_seeds = n;
  return EG3D_OK;
}

namespace {

struct BatchState {
  uint32_t b, e, n_seeds, sv_base, n_sv, n_tasks, n_lists, n_hits, n_hyp, n_chains, total_raw;
  uint32_t key0_base;  // added to key[0] of every point (polyline-set path: samples before this batch)
  StageAView a;
  SeedsDev sd;
};

// Stage A: K1 + task enumeration + K2. Leaves everything on the device.
int run_stage_a(eg3d_ctx* c, BatchState& B, eg3d_stage_times* tm) {
  hipStream_t st = c->stream;
  B.n_seeds = B.e - B.b;
  B.sv_base = (*c->h_trk)[B.b];
  B.n_sv = (*c->h_trk)[B.e] - B.sv_base;
  B.sd.trk_off = c->b_toff.as<uint32_t>();
  B.sd.trk_view = c->b_tview.as<int32_t>();
  B.sd.trk_xy = c->b_txy.as<float>();
  const uint32_t n_sv = B.n_sv;
  BUF_TRY(c->b_sv_seed.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_map_view.ensure(sizeof(int32_t) * (n_sv + 1)));
  BUF_TRY(c->b_map_entry.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_map_n.ensure(sizeof(uint32_t) * (B.n_seeds + 1)));
  BUF_TRY(c->b_raw_cnt.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_raw_off.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_cand_cnt.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_start_cnt.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_task_off.ensure(sizeof(uint32_t) * (n_sv + 1)));
  BUF_TRY(c->b_ctr.ensure(sizeof(Counters)));
  HIP_TRY(hipMemsetAsync(c->b_ctr.p, 0, sizeof(Counters), st));
  launch_seed_prep(st, B.sd, B.b, B.n_seeds, B.sv_base, c->b_sv_seed.as<uint32_t>(), c->b_map_view.as<int32_t>(),
                   c->b_map_entry.as<uint32_t>(), c->b_map_n.as<uint32_t>());
  HIP_TRY(hipMemsetAsync(c->b_raw_cnt.as<uint32_t>() + n_sv, 0, sizeof(uint32_t), st));
  launch_k1_count_raw(st, c->ds, B.sd, B.sv_base, n_sv, c->b_sv_seed.as<uint32_t>(), c->b_raw_cnt.as<uint32_t>());
  BUF_TRY(scan_total_u32(c, c->b_raw_cnt.as<uint32_t>(), c->b_raw_off.as<uint32_t>(), n_sv + 1, B.total_raw, "candidate slots"));
  BUF_TRY(c->b_cand_pl.ensure(sizeof(uint32_t) * (B.total_raw + 1)));
  BUF_TRY(c->b_start_hits.ensure(sizeof(Obs) * (B.total_raw + 1)));
  HIP_TRY(hipMemsetAsync(c->b_start_cnt.as<uint32_t>() + n_sv, 0, sizeof(uint32_t), st));
  HIP_TRY(hipEventRecord(c->ea[1], st));
  launch_k1(st, c->ds, B.sd, B.sv_base, n_sv, c->b_sv_seed.as<uint32_t>(), c->b_raw_off.as<uint32_t>(),
            c->b_cand_pl.as<uint32_t>(), c->b_start_hits.as<Obs>(), c->b_cand_cnt.as<uint32_t>(),
            c->b_start_cnt.as<uint32_t>(), c->b_raw_cnt.as<uint32_t>() /* dead after its scan: reused for K1's vertex counts */);
  HIP_TRY(hipEventRecord(c->eb[1], st));
  BUF_TRY(scan_total_u32(c, c->b_start_cnt.as<uint32_t>(), c->b_task_off.as<uint32_t>(), n_sv + 1, B.n_tasks, "tasks"));
  const uint32_t nt = B.n_tasks;
  BUF_TRY(c->b_task_seed.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_entry.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_hit.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_k.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_list_off.ensure(sizeof(uint32_t) * (nt + 1)));
  HIP_TRY(hipMemsetAsync(c->b_task_k.as<uint32_t>() + nt, 0, sizeof(uint32_t), st));
  launch_task_fill(st, B.sd, B.sv_base, n_sv, c->b_sv_seed.as<uint32_t>(), c->b_start_cnt.as<uint32_t>(),
                   c->b_task_off.as<uint32_t>(), c->b_task_seed.as<uint32_t>(), c->b_task_entry.as<uint32_t>(),
                   c->b_task_hit.as<uint32_t>(), c->b_task_k.as<uint32_t>(), c->b_raw_cnt.as<uint32_t>(), c->b_ctr.as<Counters>());
  BUF_TRY(scan_total_u32(c, c->b_task_k.as<uint32_t>(), c->b_task_list_off.as<uint32_t>(), nt + 1, B.n_lists, "hit lists"));
  BUF_TRY(c->b_list_cnt.ensure(sizeof(uint32_t) * (B.n_lists + 1)));
  BUF_TRY(c->b_list_ptr.ensure(sizeof(uint32_t) * (B.n_lists + 1)));
  HIP_TRY(hipMemsetAsync(c->b_list_cnt.as<uint32_t>() + B.n_lists, 0, sizeof(uint32_t), st));
  HIP_TRY(hipEventRecord(c->ea[2], st));
  launch_k2(st, false, c->ds, B.sd, B.b, B.n_seeds, B.sv_base, nt, c->b_task_off.as<uint32_t>(),
            c->b_task_seed.as<uint32_t>(), c->b_task_entry.as<uint32_t>(), c->b_task_hit.as<uint32_t>(), c->b_task_list_off.as<uint32_t>(), c->b_raw_off.as<uint32_t>(),
            c->b_cand_pl.as<uint32_t>(), c->b_cand_cnt.as<uint32_t>(), c->b_start_hits.as<Obs>(),
            c->b_list_cnt.as<uint32_t>(), nullptr, nullptr);
  BUF_TRY(scan_total_u32(c, c->b_list_cnt.as<uint32_t>(), c->b_list_ptr.as<uint32_t>(), B.n_lists + 1, B.n_hits, "epipolar hits"));
  BUF_TRY(c->b_hits.ensure(sizeof(Obs) * (B.n_hits + 1)));
  launch_k2(st, true, c->ds, B.sd, B.b, B.n_seeds, B.sv_base, nt, c->b_task_off.as<uint32_t>(),
            c->b_task_seed.as<uint32_t>(), c->b_task_entry.as<uint32_t>(), c->b_task_hit.as<uint32_t>(), c->b_task_list_off.as<uint32_t>(), c->b_raw_off.as<uint32_t>(),
            c->b_cand_pl.as<uint32_t>(), c->b_cand_cnt.as<uint32_t>(), c->b_start_hits.as<Obs>(),
            c->b_list_cnt.as<uint32_t>(), c->b_list_ptr.as<uint32_t>(), c->b_hits.as<Obs>());
  HIP_TRY(hipEventRecord(c->eb[2], st));
  StageAView& a = B.a;
  a.trk_off = B.sd.trk_off;
  a.trk_view = B.sd.trk_view;
  a.trk_xy = B.sd.trk_xy;
  a.seed_begin = B.b;
  a.sv_base = B.sv_base;
  a.n_tasks = nt;
  a.task_seed = c->b_task_seed.as<uint32_t>();
  a.task_entry = c->b_task_entry.as<uint32_t>();
  a.task_hit = c->b_task_hit.as<uint32_t>();
  a.task_list_off = c->b_task_list_off.as<uint32_t>();
  a.list_ptr = c->b_list_ptr.as<uint32_t>();
  a.list_cnt = c->b_list_cnt.as<uint32_t>();
  a.hits = c->b_hits.as<Obs>();
  (void)tm;
  return EG3D_OK;
}

// Caller-bound output arrays: malloc/realloc'ed runs that are handed to the caller as they are
// (eg3d_free_edgepoints frees them) — no zero-fill on growth, no final copy.
template <typename T>
struct RawVec {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  RawVec() = default;
  RawVec(const RawVec&) = delete;
  RawVec& operator=(const RawVec&) = delete;
  ~RawVec() { free(p); }
  bool reserve(size_t nc) {  // keeps the contents
    if (nc <= cap) return true;
    T* q;
    const size_t bytes = sizeof(T) * nc;
    if (!p && bytes >= (size_t)(4u << 20)) {
      // large result arrays: 2 MiB-aligned and advised as huge pages — the first touch of a fresh 0.45 GB cloud
      // in 4 KiB pages costs 110 k page faults (measured: the host copy ran at 11-15 GB/s whatever the thread
      // count; 60-80 GB/s with huge pages). Untouched capacity costs nothing, so multi-chunk results reserve
      // their estimated total up front: growing such a block later is the slow case.
      void* m = nullptr;
      if (posix_memalign(&m, (size_t)2 << 20, bytes) != 0) return false;
      (void)madvise(m, bytes, MADV_HUGEPAGE);
      q = (T*)m;
    } else {
      q = (T*)realloc(p, bytes);
      if (!q) return false;
    }
    p = q;
    cap = nc;
    return true;
  }
  bool grow_to(size_t want) {  // keeps the contents
    if (want > cap && !reserve(std::max(want, cap + cap / 2 + 64))) return false;
    n = want;
    return true;
  }
  size_t size() const { return n; }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  T* release() {
    T* r = p ? p : (T*)malloc(sizeof(T));
    p = nullptr;
    n = cap = 0;
    return r;
  }
};

static void copy_mt(void* dst, const void* src, size_t bytes) { eg3d::copy_mt(dst, src, bytes); }

// What one unit (a sub-batch of a call: a seed range or a run of polyline sets) adds to the call's totals.
struct HostOut {
  uint64_t n_points = 0, n_obs = 0, n_tasks = 0, n_hyp = 0, n_chains = 0;
  uint32_t flags = 0;
  uint64_t bytes_algorithmic = 0, bytes_vertices = 0;
  float ms[7] = {0, 0, 0, 0, 0, 0, 0};
  uint32_t max_chain_ticks = 0;  // the slowest chain of the expand launches (constant-rate clock)
  int pieces = 0;  // (unit, chunk) pieces placed
};

// Where the units of ONE eg3d_match_* call put their output. Units are computed concurrently on the context's lanes
// (run_pipelined) but PLACED strictly in unit order — the call's cloud is the concatenation of its units' clouds in seed /
// set order, byte for byte what a single batch produces (seeds are independent: plg_matching_from_refpoints.cpp:83-104).
// The order is kept by a turnstile: a unit may place (learn its offsets, reserve its part of the destination) only when
// every earlier unit has placed everything.
//   host calls (device_only == 0): a unit's arrays cross PCIe into its lane's pinned staging area as soon as its k4_emit is
//     done, whatever its turn; once it holds the turn it reserves [p0, p0 + np) x [o0, o0 + no) of the caller-bound arrays,
//     passes the turn on, and copies staging -> destination while the later units are still computing.
//   device-only calls: the unit waits for its turn BEFORE k4_emit, which then writes straight into the owner's output
//     buffers at the unit's offsets (eg3d_last_device_output sees one whole cloud, as before).
struct CallSink {
  eg3d_ctx* owner = nullptr;
  int device_only = 0;
  std::mutex mu;
  std::condition_variable cv;
  uint32_t turn = 0;  // the unit that places next
  bool failed = false;
  int rc = EG3D_OK;
  std::string err;
  // destination of a host call; dst_mu: copiers hold it shared, a (rare) growth of the arrays holds it exclusively
  RawVec<float> X, xy;
  RawVec<uint32_t> pl, seg, key;
  RawVec<uint64_t> off;
  RawVec<int32_t> view;
  std::shared_mutex dst_mu;
  uint64_t n_points = 0, n_obs = 0;          // placed so far
  double weight_total = 0, weight_placed = 0;  // share of the call placed so far (sizes the destination's first allocation)
  int lanes_active = 1;                      // lanes working on the call (shares the host copy threads among them)
  bool keyed_by_sample = false;              // polyline-set calls: key[0] = sample index of the CALL ...
  uint32_t key0_next = 0;                    // ... = samples of the units placed so far + the sample's index in its unit
  HostOut tot;
  void unit_tasks(uint32_t n) {  // (holding the turn)
    if (keyed_by_sample) key0_next += n;
  }

  bool acquire(uint32_t u) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return failed || turn == u; });
    return !failed;
  }
  void release(uint32_t u) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (turn == u) turn = u + 1;
    }
    cv.notify_all();
  }
  void fail(int code, const std::string& text) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!failed) {
        failed = true;
        rc = code;
        err = text;
      }
    }
    cv.notify_all();
  }
  // (holding the turn) reserve np points / no observations of the host destination; w = this piece's share of the call
  bool place_host(uint64_t np, uint64_t no, double w, uint64_t& p0, uint64_t& o0) {
    p0 = n_points;
    o0 = n_obs;
    const uint64_t need_p = p0 + np, need_o = o0 + no;
    if (need_p + 1 > off.cap || need_o > view.cap) {
      // estimated size of the whole call from the share placed so far (virtual until touched; a block that has to grow
      // later is the slow case, so the estimate is generous)
      double f = 1.0;
      if (weight_total > weight_placed + w && weight_placed + w > 0) f = 1.5 * weight_total / (weight_placed + w);
      const size_t rp = (size_t)std::max<double>((double)need_p, f * (double)need_p) + 1;
      const size_t ro = (size_t)std::max<double>((double)need_o, f * (double)need_o) + 1;
      std::unique_lock<std::shared_mutex> lk(dst_mu);
      if (!X.reserve(rp * 3) || !off.reserve(rp + 1) || !key.reserve(rp * 4) || !view.reserve(ro) || !pl.reserve(ro) ||
          !seg.reserve(ro) || !xy.reserve(ro * 2))
        return false;
    }
    n_points = need_p;
    n_obs = need_o;
    weight_placed += w;
    return true;
  }
  void add(const HostOut& h) {
    std::lock_guard<std::mutex> lk(mu);
    tot.n_tasks += h.n_tasks;
    tot.n_hyp += h.n_hyp;
    tot.n_chains += h.n_chains;
    tot.flags |= h.flags;
    tot.bytes_algorithmic += h.bytes_algorithmic;
    tot.pieces += h.pieces;
    tot.max_chain_ticks = std::max(tot.max_chain_ticks, h.max_chain_ticks);
    for (int k = 0; k < 7; k++) tot.ms[k] += h.ms[k];
  }
};

// A unit's pass through the turnstile (it may take the turn several times: once per chunk of an expand stage cut by
// EG3D_MAX_SCRATCH_MB; it passes the turn on exactly once).
struct UnitTurn {
  CallSink& S;
  uint32_t u;
  double weight;       // the unit's share of the call (sum of track lengths / polylines)
  bool held = false, passed = false;
  bool take() {
    if (held) return true;
    if (passed) return false;
    if (S.owner->tune.test_fail_unit == (int)u + 1) {  // (tests: the failure path of the turnstile)
      g_err = "eg3d: unit failure requested by EG3D_TEST_FAIL_UNIT";
      return false;
    }
    held = S.acquire(u);
    return held;
  }
  void pass() {
    if (passed) return;
    if (!held && !S.acquire(u)) {  // (failed call: nobody waits for a turn any more)
      passed = true;
      return;
    }
    S.release(u);
    held = false;
    passed = true;
  }
};

// Stage B: task setup, K3a, K3s, K3b + K4 in chunks. Consumes the StageAView in B (from the seed
// path's stage A or from the polyline-set sampler); the output goes to the call's sink in unit order (T), the
// unit's counts and times to H.
int run_stage_b(eg3d_ctx* c, BatchState& B, UnitTurn& T, HostOut& H) {
  hipStream_t st = c->stream;
  CallSink& S = T.S;
  const int device_only = S.device_only;
  const uint32_t nt = B.n_tasks;
  // ---- task setup + hypothesis offsets
  BUF_TRY(c->b_tasks.ensure(sizeof(TaskDesc) * (nt + 1)));
  BUF_TRY(c->b_nhyp.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_hyp_off.ensure(sizeof(uint32_t) * (nt + 1)));
  HIP_TRY(hipMemsetAsync(c->b_nhyp.as<uint32_t>() + nt, 0, sizeof(uint32_t), st));
  launch_task_setup(st, B.a, c->b_map_view.as<int32_t>(), c->b_map_entry.as<uint32_t>(), c->b_map_n.as<uint32_t>(),
                    c->b_tasks.as<TaskDesc>(), c->b_nhyp.as<uint32_t>());
  BUF_TRY(scan_total_u32(c, c->b_nhyp.as<uint32_t>(), c->b_hyp_off.as<uint32_t>(), nt + 1, B.n_hyp, "hypotheses"));
  // ---- K3a
  BUF_TRY(c->b_res.ensure(sizeof(HypResult) * (B.n_hyp + 1)));
  // K3a engine (eg3d_k3a_engine.h): single-wavefront blocks; the lanes of a wave that take work are limited when there
  // is little of it, so that each working lane gets more of the wave's 64 request slots
  const uint32_t eng_waves_max = c->n_simd * (uint32_t)(c->tune.k3a_engine_waves > 0 ? c->tune.k3a_engine_waves : 2);
  uint32_t eng_lanes = 64;
  if (c->tune.k3a_engine_lanes > 0)
    eng_lanes = (uint32_t)std::min(64, c->tune.k3a_engine_lanes);
  else
    while (eng_lanes > 8 && (uint64_t)eng_waves_max * (eng_lanes / 2) >= B.n_hyp) eng_lanes /= 2;
  const uint32_t eng_waves = std::max<uint32_t>(1, std::min<uint32_t>(eng_waves_max, (B.n_hyp + eng_lanes - 1) / eng_lanes));
  BUF_TRY(c->b_fscratch.ensure(sizeof(HPoint) * std::min<uint32_t>(c->hyp_cap, EG3D_K3A_STAGE_POINTS) * ((size_t)eng_waves * 64)));
  BUF_TRY(c->b_queue.ensure(4 * sizeof(uint32_t)));
  BUF_TRY(c->b_items.ensure(sizeof(uint32_t) * 2 * ((size_t)B.n_hyp + 1)));
  // hypothesis arena: 16 points per hypothesis to start with (the seed path uses 0.4-11.6, the polyline-set path 23-24: its
  // first call overflows once), more once an attempt of this context has overflowed (it is redone with room)
  uint32_t arena_cap = std::max<uint32_t>(1u << 16, std::min<uint64_t>((uint64_t)((double)B.n_hyp * c->arena_per_hyp) + 1, 1ull << 26));
  if (c->tune.arena_cap0) arena_cap = c->tune.arena_cap0;  // tests: force the overflow-and-retry path
  Counters hc;
  BUF_TRY(c->b_cs_task.ensure(sizeof(ChainSeed) * (nt + 1)));
  BUF_TRY(c->b_valid.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_chain_off.ensure(sizeof(uint32_t) * (nt + 1)));
  for (int attempt = 0;; attempt++) {
    BUF_TRY(c->b_arena.ensure(sizeof(HPoint) * (size_t)arena_cap));
    HIP_TRY(hipMemsetAsync(c->b_ctr.p, 0, 2 * sizeof(uint32_t), st));  // arena_used, flags (keep bytes)
    HIP_TRY(hipEventRecord(c->ea[3], st));
    HIP_TRY(hipMemsetAsync(c->b_queue.p, 0, 4 * sizeof(uint32_t), st));
    launch_k3a_engine(st, eng_waves, eng_waves, eng_lanes, c->ds, B.a, c->b_tasks.as<TaskDesc>(), c->b_hyp_off.as<uint32_t>(),
                      B.n_hyp, c->b_res.as<HypResult>(), c->b_fscratch.as<HPoint>(), c->hyp_cap, c->b_arena.as<HPoint>(),
                      arena_cap, c->b_ctr.as<Counters>(), c->b_queue.as<uint32_t>(), c->b_items.as<uint32_t>());
    HIP_TRY(hipEventRecord(c->eb[3], st));
    // ---- K3s and the chain scan are queued right behind K3a; K3a's counters (arena overflow?) and the
    // number of chains come back in ONE read-back. An overflowing attempt is redone from K3a.
    HIP_TRY(hipMemsetAsync(c->b_valid.as<uint32_t>() + nt, 0, sizeof(uint32_t), st));
    HIP_TRY(hipEventRecord(c->ea[4], st));
    launch_k3s(st, nt, c->b_hyp_off.as<uint32_t>(), c->b_res.as<HypResult>(), c->b_cs_task.as<ChainSeed>(),
               c->b_valid.as<uint32_t>());
    BUF_TRY(scan_queue_u32(c, c->b_valid.as<uint32_t>(), c->b_chain_off.as<uint32_t>(), nt + 1, 0));
    Readback rb(c);
    const int ic = rb.add(c->b_ctr.p, sizeof(Counters) / 4);
    const int it = rb.add(c->b_chain_off.as<uint32_t>() + nt, 1);
    const int iw = rb.add(c->b_scanchk.as<uint32_t>(), 1);
    rb.clear_after(c->b_scanchk.as<uint32_t>());
    BUF_TRY(rb.run());
    memcpy(&hc, rb.item(ic), sizeof(Counters));
    if (c->tune.trace_arena)
      fprintf(stderr, "eg3d: hypothesis arena: %u hypotheses, %u points used of %u (%.2f per hypothesis)%s\n", B.n_hyp,
              hc.arena_used, arena_cap, B.n_hyp ? (double)hc.arena_used / B.n_hyp : 0.0,
              (hc.flags & CTR_ARENA_OVERFLOW) ? " OVERFLOW" : "");
    if (!(hc.flags & CTR_ARENA_OVERFLOW)) {
      if (*rb.item(iw)) return wrapped_error("chains");
      B.n_chains = *rb.item(it);
      break;
    }
    if (attempt >= 6) {
      g_err = "eg3d: hypothesis arena overflow";
      return EG3D_ERR_CAPACITY;
    }
    arena_cap = std::max<uint32_t>(arena_cap * 2, hc.arena_used + (hc.arena_used >> 2));
    if (B.n_hyp) c->arena_per_hyp = std::max(c->arena_per_hyp, 1.25 * (double)arena_cap / (double)B.n_hyp);
  }
  H.flags |= (hc.flags & 0xffu);
  BUF_TRY(c->b_chains.ensure(sizeof(ChainSeed) * (B.n_chains + 1)));
  launch_compact_chains(st, nt, c->b_cs_task.as<ChainSeed>(), c->b_valid.as<uint32_t>(), c->b_chain_off.as<uint32_t>(),
                        c->b_chains.as<ChainSeed>());
  HIP_TRY(hipEventRecord(c->eb[4], st));
  // ---- K3b + K4. One launch takes all the chains of the batch: their working slices are slots of a fixed arena
  // (8 XCDs x slots_per_xcd), so nothing grows with the number of chains but the staging area the finished chains
  // are packed into. (EG3D_MAX_SCRATCH_MB, a test knob, still cuts the chains into several launches.)
  const size_t max_scratch = c->tune.max_scratch;
  float ms_expand = 0, ms_emit = 0;
  uint32_t chunk = 0;
  // A launch in which some chains outgrew their working slices is followed by a RELAUNCH OF THOSE CHAINS ONLY with larger
  // slices (redo_n of the chunk's redo_nc chains, listed in b_redo[redo_buf]); the others keep their packed results.
  // (Until round 6 the whole chunk was launched again: the first call on a many-view scene ran its expand stage four times —
  // C4: 6.8 s against 1.75 s warm.)
  uint32_t redo_n = 0, redo_nc = 0;
  int redo_buf = 0;
  for (uint32_t c0 = 0; c0 < B.n_chains; c0 += chunk) {
    // capacities can grow between chunks (overflow -> retry below), so the layout is per chunk
    const ChainLayout L = chain_layout(c->chain_cap, c->pool_cap, (uint32_t)c->V);
    chunk = redo_n ? redo_nc
                   : (uint32_t)std::max<size_t>(1, max_scratch ? std::min<size_t>(B.n_chains - c0, max_scratch / L.total)
                                                               : (size_t)(B.n_chains - c0));
    const uint32_t nc = std::min(chunk, B.n_chains - c0);
    BUF_TRY(ensure_mailbox(c));
    uint32_t* const saved_bytes = c->b_scanchk.as<uint32_t>() + 4;  // device copy of the byte counter before this chunk
    if (!redo_n)
      HIP_TRY(hipMemcpyAsync(saved_bytes, &c->b_ctr.as<Counters>()->bytes, sizeof(unsigned long long),
                             hipMemcpyDeviceToDevice, st));
    // the expand stage's two forms: one wavefront per chain on XCD-affine slots (k3b_expand), or the lane-per-chain
    // engine (k3c_engine): n_waves single-wave blocks whose first eng_lanes lanes own a working slice each
#ifdef EG3D_WITH_K3C_ENGINE
    const bool engine = c->tune.k3b_engine != 0;
#else
    const bool engine = false;
#endif
    uint32_t eng_waves = 0, eng_lanes = 64;
    SlotPools pools;
    pools.base = nullptr;
    pools.stride = pools.ring_n = pools.slots_per_xcd = 0;
#ifdef EG3D_WITH_K3C_ENGINE
    if (engine) {
      const uint32_t per_simd = (uint32_t)(c->tune.k3c_waves > 0 ? c->tune.k3c_waves : std::max(1, c->k3c_per_cu / 4));
      const uint32_t waves_max = c->n_simd * per_simd;
      // owning lanes per wave: 8 by measurement (C3': 2 / 4 / 8 / 16 / 32 / 64 lanes -> 111 / 105 / 103 / 149 / 252 / 397 ms —
      // more lanes buy no parallelism in the lane-private phases and add resident chains, DESIGN_LOG.md round 5)
      eng_lanes = c->tune.k3c_lanes > 0 ? (uint32_t)std::min(64, c->tune.k3c_lanes) : 8u;
      eng_waves = std::max<uint32_t>(1, std::min<uint32_t>(waves_max, (nc + eng_lanes - 1) / eng_lanes));
      // one working slice per owning lane: keep the arena within 16 GB (many-view scenes have slices of ~1 MB)
      while (eng_waves > 64 && L.total * (size_t)eng_waves * eng_lanes > ((size_t)16 << 30)) eng_waves /= 2;
      BUF_TRY(c->b_cscratch.ensure(L.total * (size_t)eng_waves * eng_lanes));
      BUF_TRY(c->b_queue.ensure(4 * sizeof(uint32_t)));
      HIP_TRY(hipMemsetAsync(c->b_queue.p, 0, 4 * sizeof(uint32_t), st));
    } else
#endif
    {
      BUF_TRY(c->b_cscratch.ensure(L.total * 8 * (size_t)c->slots_per_xcd));
      pools.slots_per_xcd = c->slots_per_xcd;
      pools.ring_n = 1;
      while (pools.ring_n <= c->slots_per_xcd) pools.ring_n <<= 1;
      pools.stride = 32 + pools.ring_n;
      BUF_TRY(c->b_pools.ensure(sizeof(uint32_t) * 8 * (size_t)pools.stride));
      pools.base = c->b_pools.as<uint32_t>();
      launch_pool_init(st, pools);
    }
    // staging area: what the previous launches needed, or a first guess (an overflowing launch is repeated once
    // with the exact need, which the output scans report). A relaunch of some chains appends to the area as it is.
    if (!redo_n) {
      const uint64_t guess_pts = 96ull * nc;
      const uint64_t per_pt = (uint64_t)std::min(100, std::max(8, c->V / 2));
      const uint64_t want_pts = std::max<uint64_t>(c->stage_cap_pts, guess_pts);
      // (the first guess is capped at 8 GB of observations: a launch that needs more reports its exact need and is
      // repeated once — better than reserving tens of GB per context on a guess for a many-view scene)
      const uint64_t want_obs = std::max<uint64_t>(c->stage_cap_obs, std::min<uint64_t>(guess_pts * per_pt, (8ull << 30) / sizeof(Obs)));
      BUF_TRY(c->b_stage_pts.ensure(sizeof(StagePt) * (size_t)want_pts));
      BUF_TRY(c->b_stage_obs.ensure(sizeof(Obs) * (size_t)want_obs));
      BUF_TRY(c->b_stage_used.ensure(2 * sizeof(unsigned long long)));
      HIP_TRY(hipMemsetAsync(c->b_stage_used.p, 0, 2 * sizeof(unsigned long long), st));
    }
    StageBuf stage;
    stage.pts = c->b_stage_pts.as<StagePt>();
    stage.obs = c->b_stage_obs.as<Obs>();
    stage.cap_pts = c->b_stage_pts.cap / sizeof(StagePt);
    stage.cap_obs = c->b_stage_obs.cap / sizeof(Obs);
    stage.used = c->b_stage_used.as<unsigned long long>();
    BUF_TRY(c->b_couts.ensure(sizeof(ChainOut) * (nc + 1)));
    BUF_TRY(c->b_cpts.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_cobs.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_cpoff.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_cooff.ensure(sizeof(uint32_t) * (nc + 1)));
    HIP_TRY(hipMemsetAsync(c->b_cpts.as<uint32_t>() + nc, 0, sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(c->b_cobs.as<uint32_t>() + nc, 0, sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(c->b_ctr.p, 0, 2 * sizeof(uint32_t), st));
    // longest-first launch order of this chunk's chains (sort by estimated cost, descending)
    BUF_TRY(c->b_cost.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_cidx.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_cost2.ensure(sizeof(uint32_t) * (nc + 1)));
    BUF_TRY(c->b_order.ensure(sizeof(uint32_t) * (nc + 1)));
    const bool use_lpt = c->tune.use_lpt;
    const uint32_t n_launch = redo_n ? redo_n : nc;
    const uint32_t* const launch_order = redo_n ? c->b_redo[redo_buf].as<uint32_t>() : c->b_order.as<uint32_t>();
    if (!redo_n) {
    launch_chain_cost(st, B.a, c->b_tasks.as<TaskDesc>(), c->b_chains.as<ChainSeed>() + c0, nc, c->b_cost.as<uint32_t>(),
                      c->b_cidx.as<uint32_t>());
    {
      size_t tmp_bytes = 0;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, c->b_cost.as<uint32_t>(),
                                                           c->b_cost2.as<uint32_t>(), c->b_cidx.as<uint32_t>(),
                                                           c->b_order.as<uint32_t>(), (int)nc, 0, 32, st));
      BUF_TRY(c->b_scan_tmp.ensure(tmp_bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(c->b_scan_tmp.p, tmp_bytes, c->b_cost.as<uint32_t>(),
                                                           c->b_cost2.as<uint32_t>(), c->b_cidx.as<uint32_t>(),
                                                           c->b_order.as<uint32_t>(), (int)nc, 0, 32, st));
    }
    if (!use_lpt)  // identity order (diagnostic): chain j runs in block j
      HIP_TRY(hipMemcpyAsync(c->b_order.p, c->b_cidx.p, sizeof(uint32_t) * nc, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipEventRecord(c->ea[5], st));
    // scene class of the launch: solves of more than 32 rows (EG3D_GN_PACK_MAX) need the builds with the solver's
    // long-request path. Few views normally means none (one observation per view), and the smaller builds run; a point
    // that repeats a view can exceed it — the kernel then raises CTR_LONG_REFUSED and the chunk is redone with the general
    // build (k3b_full latched for the context), like the capacity overflows below.
    const bool general = c->tune.k3b_full || c->k3b_long_latched || c->max_pl_vtx > EG3D_STAGE_VTX_HOST;
#ifdef EG3D_WITH_K3C_ENGINE
    if (engine)
      launch_k3c(st, eng_waves, eng_lanes, c->ds, B.a, c->b_tasks.as<TaskDesc>(), c->b_chains.as<ChainSeed>() + c0, nc,
                 c->b_hyp_off.as<uint32_t>(), c->b_res.as<HypResult>(), c->b_arena.as<HPoint>(), c->b_map_view.as<int32_t>(),
                 c->b_map_entry.as<uint32_t>(), c->b_map_n.as<uint32_t>(), L, c->b_cscratch.as<unsigned char>(), stage,
                 c->b_couts.as<ChainOut>(), c->b_cpts.as<uint32_t>(), c->b_cobs.as<uint32_t>(), c->b_ctr.as<Counters>(),
                 c->b_order.as<uint32_t>(), c->b_queue.as<uint32_t>(),
                 (c->tune.k3b_full || c->k3b_long_latched || (c->V > EG3D_SMALL_SCENE_VIEWS_HOST && !c->tune.assume_short)) ? 1 : 0);
    else
#endif
    launch_k3b(st, c->ds, B.a, c->b_tasks.as<TaskDesc>(), c->b_chains.as<ChainSeed>() + c0, n_launch,
               c->b_hyp_off.as<uint32_t>(), c->b_res.as<HypResult>(), c->b_arena.as<HPoint>(),
               c->b_map_view.as<int32_t>(), c->b_map_entry.as<uint32_t>(), c->b_map_n.as<uint32_t>(), L,
               c->b_cscratch.as<unsigned char>(), pools, stage, c->b_couts.as<ChainOut>(), c->b_cpts.as<uint32_t>(),
               c->b_cobs.as<uint32_t>(), c->b_ctr.as<Counters>(), launch_order,
               general ? 1 : (c->V > EG3D_SMALL_SCENE_VIEWS_HOST && !c->tune.assume_short) ? 2 : 0);
    HIP_TRY(hipEventRecord(c->eb[5], st));
    if (!device_only)  // (first host call of this lane: pinned while the expand kernel runs; ~50 points of ~12 observations per chain)
      BUF_TRY(ensure_d2h_ring(c, (size_t)nc * 50u * (36u + 20u * (size_t)std::min(c->V, 12))));
    // the two output scans are queued right behind K3b; its counters (capacity overflow?) and both totals
    // come back in ONE read-back
    BUF_TRY(scan_queue_u32(c, c->b_cpts.as<uint32_t>(), c->b_cpoff.as<uint32_t>(), nc + 1, 0));
    BUF_TRY(scan_queue_u32(c, c->b_cobs.as<uint32_t>(), c->b_cooff.as<uint32_t>(), nc + 1, 1));
    uint32_t np = 0, no = 0;
    unsigned long long stage_used[2] = {0, 0};
    {
      Readback rb(c);
      const int ic = rb.add(c->b_ctr.p, sizeof(Counters) / 4);
      const int ip = rb.add(c->b_cpoff.as<uint32_t>() + nc, 1);
      const int io = rb.add(c->b_cooff.as<uint32_t>() + nc, 1);
      const int iw = rb.add(c->b_scanchk.as<uint32_t>(), 2);
      const int iu = rb.add(c->b_stage_used.p, 4);  // what the launches of this chunk have packed (two 64-bit counters)
      rb.clear_after(c->b_scanchk.as<uint32_t>());
      rb.clear_after(c->b_scanchk.as<uint32_t>() + 1);
      BUF_TRY(rb.run());
      memcpy(&hc, rb.item(ic), sizeof(Counters));
      np = *rb.item(ip);
      no = *rb.item(io);
      memcpy(stage_used, rb.item(iu), sizeof(stage_used));
      const bool overflow = hc.flags & (EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW);
      if (!overflow && rb.item(iw)[0]) return wrapped_error("edge-points of one chunk");
      if (!overflow && rb.item(iw)[1]) return wrapped_error("observations of one chunk");
    }
    if (hc.flags & CTR_LONG_REFUSED) {
      // a point of a few-views scene carries more than 32 observations (a view repeated on a track): valid input — the
      // general build solves it. Latch it for the context, restore the byte counter and redo this chunk.
      if (c->k3b_long_latched) {
        g_err = "eg3d: internal: the general build of the expand kernel refused a solve";
        return EG3D_ERR_HIP;
      }
      c->k3b_long_latched = true;
      if (c->tune.trace_arena) fprintf(stderr, "eg3d: expand launch of %u chains redone: a solve of more than 32 rows -> general build\n", nc);
      HIP_TRY(hipMemcpyAsync(&c->b_ctr.as<Counters>()->bytes, saved_bytes, sizeof(unsigned long long),
                             hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      chunk = 0;  // do not advance
      redo_n = 0;  // (every chain of the chunk again)
      continue;
    }
    if (hc.flags & CTR_SLOT_STARVED) {
      g_err = "eg3d: internal: the expand kernel found no free working slice (slot pool smaller than the residency)";
      return EG3D_ERR_HIP;
    }
    // (the area holds what every launch of the chunk packed — after a relaunch of some chains also their discarded first
    // attempts — so it is the packed totals, not the cloud's, that must fit)
    if (!(hc.flags & (EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW)) &&
        (np > stage.cap_pts || no > stage.cap_obs || stage_used[0] > stage.cap_pts || stage_used[1] > stage.cap_obs)) {
      // the staging area was too small for this launch: its chains were counted but not all packed. Size it
      // for what they need (kept for later calls) and repeat the launch.
      if (c->tune.trace_arena)
        fprintf(stderr, "eg3d: expand launch of %u chains redone: staging area %zu points / %zu observations, needed %u / %u\n", nc,
                (size_t)stage.cap_pts, (size_t)stage.cap_obs, np, no);
      c->stage_cap_pts = std::max<uint64_t>(c->stage_cap_pts, std::max<uint64_t>(np, stage_used[0]) + np / 16 + 64);
      c->stage_cap_obs = std::max<uint64_t>(c->stage_cap_obs, std::max<uint64_t>(no, stage_used[1]) + no / 16 + 64);
      HIP_TRY(hipMemcpyAsync(&c->b_ctr.as<Counters>()->bytes, saved_bytes, sizeof(unsigned long long),
                             hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      chunk = 0;  // do not advance
      redo_n = 0;  // (every chain of the chunk again: what was packed does not fit the area as it is)
      continue;
    }
    if (hc.flags & (EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW)) {
      // a chain outgrew its working slice: enlarge the capacities (kept for later calls) and redo
      // this chunk; results of the overflowing attempt are discarded
      const bool can_grow = ((hc.flags & EG3D_FLAG_CHAIN_OVERFLOW) && c->chain_cap < 8192) ||
                            ((hc.flags & EG3D_FLAG_OBS_OVERFLOW) && c->pool_cap < (1u << 20));
      if (can_grow) {
        if (hc.flags & EG3D_FLAG_CHAIN_OVERFLOW) c->chain_cap *= 2;
        if (hc.flags & EG3D_FLAG_OBS_OVERFLOW) c->pool_cap *= 2;
        H.flags |= hc.flags & 0xffu & ~(uint32_t)(EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW);  // what the chains that keep their results raised
        {
          float t_launch = 0;  // (the launch is over: its counters have been read back)
          HIP_TRY(hipEventElapsedTime(&t_launch, c->ea[5], c->eb[5]));
          ms_expand += t_launch;
        }
        uint32_t n_again = 0;
        if (!engine) {
          // list the chains of this launch that overflowed (and take their share of the byte counter back)
          const int nb = redo_n ? 1 - redo_buf : 0;
          BUF_TRY(c->b_redo[nb].ensure(sizeof(uint32_t) * (nc + 1)));
          BUF_TRY(c->b_queue.ensure(4 * sizeof(uint32_t)));
          HIP_TRY(hipMemsetAsync(c->b_queue.p, 0, 4 * sizeof(uint32_t), st));
          launch_collect_overflow(st, c->b_couts.as<ChainOut>(), launch_order, n_launch, c->b_redo[nb].as<uint32_t>(),
                                  c->b_queue.as<uint32_t>(), c->b_ctr.as<Counters>());
          Readback rb(c);
          const int in = rb.add(c->b_queue.p, 1);
          BUF_TRY(rb.run());
          n_again = *rb.item(in);
          redo_buf = nb;
        }
        if (c->tune.trace_arena)
          fprintf(stderr, "eg3d: expand launch of %u chains: %u outgrew their slices (flags %u); relaunching %s with %u points / %u "
                          "observation slots per chain\n", n_launch, n_again, hc.flags & 3u, n_again ? "those" : "the chunk", c->chain_cap, c->pool_cap);
        if (n_again) {
          redo_n = n_again;
          redo_nc = nc;
        } else {  // (the engine form, or nothing listed: the whole chunk again)
          redo_n = 0;
          HIP_TRY(hipMemcpyAsync(&c->b_ctr.as<Counters>()->bytes, saved_bytes, sizeof(unsigned long long),
                                 hipMemcpyDeviceToDevice, st));
        }
        chunk = 0;  // do not advance
        continue;
      }
    }
    redo_n = 0;
    // ---- K4 + placement of this chunk. Device-only calls keep the WHOLE cloud of the call in the OWNER's output buffers
    // (unit after unit, chunk after chunk, global observation offsets), so that eg3d_last_device_output is complete whatever
    // the cutting — the RCCL gather reads it: the chunk waits for its unit's turn, then k4_emit writes at the call's running
    // offsets. Calls that copy to the host emit into the lane's own buffers (chunk-local offsets, rebased on the host below)
    // and start the D2H at once; the turn is only needed to learn where the chunk goes in the caller's arrays.
    const bool accumulate = device_only != 0;
    const bool last_chunk = c0 + nc >= B.n_chains;
    const double w_piece = T.weight * (double)nc / (double)std::max(1u, B.n_chains);
    eg3d_ctx* const oc = accumulate ? S.owner : c;  // whose output buffers k4_emit writes
    size_t P0 = 0, O0 = 0;
    uint32_t key0 = 0;
    if (accumulate) {
      if (!T.take()) return EG3D_ERR_HIP;  // another unit of the call failed: its error is the call's
      P0 = (size_t)S.n_points;
      O0 = (size_t)S.n_obs;
      key0 = S.key0_next + B.key0_base;
    }
    BUF_TRY(oc->o_X.ensure_keep(sizeof(float) * 3 * (P0 + np + 1), sizeof(float) * 3 * P0, st));
    BUF_TRY(oc->o_off.ensure_keep(sizeof(eg3d_off_t) * (P0 + np + 1), sizeof(eg3d_off_t) * P0, st));
    BUF_TRY(oc->o_key.ensure_keep(sizeof(uint32_t) * 4 * (P0 + np + 1), sizeof(uint32_t) * 4 * P0, st));
    BUF_TRY(oc->o_view.ensure_keep(sizeof(int32_t) * (O0 + no + 1), sizeof(int32_t) * O0, st));
    BUF_TRY(oc->o_pl.ensure_keep(sizeof(uint32_t) * (O0 + no + 1), sizeof(uint32_t) * O0, st));
    BUF_TRY(oc->o_seg.ensure_keep(sizeof(uint32_t) * (O0 + no + 1), sizeof(uint32_t) * O0, st));
    BUF_TRY(oc->o_xy.ensure_keep(sizeof(float) * 2 * (O0 + no + 1), sizeof(float) * 2 * O0, st));
    HIP_TRY(hipEventRecord(c->ea[6], st));
    c->stage_cap_pts = std::max<uint64_t>(c->stage_cap_pts, (uint64_t)np + np / 16);  // sizing hint of the next launch
    c->stage_cap_obs = std::max<uint64_t>(c->stage_cap_obs, (uint64_t)no + no / 16);
    launch_k4(st, c->b_tasks.as<TaskDesc>(), c->b_chains.as<ChainSeed>() + c0, nc, stage,
              c->b_couts.as<ChainOut>(), c->b_cpoff.as<uint32_t>(), c->b_cooff.as<uint32_t>(), P0, O0, key0,
              oc->o_X.as<float>(), oc->o_off.as<eg3d_off_t>(), oc->o_view.as<int32_t>(), oc->o_pl.as<uint32_t>(),
              oc->o_seg.as<uint32_t>(), oc->o_xy.as<float>(), oc->o_key.as<uint32_t>());
    HIP_TRY(hipEventRecord(c->eb[6], st));
    H.flags |= (hc.flags & 0xffu);
    H.bytes_vertices = hc.bytes;  // running total of this batch (K1 + K3b so far)
    H.max_chain_ticks = std::max(H.max_chain_ticks, hc.max_chain_ticks);
    if (accumulate) {
      HIP_TRY(hipStreamSynchronize(st));  // k4_emit has written: the next piece may grow / write the owner's buffers
      S.n_points += np;
      S.n_obs += no;
      if (last_chunk) {
        S.unit_tasks(B.n_tasks);
        T.pass();
      }
    } else if (np) {
      // D2H through the lane's RING of pinned buffers (EG3D_D2H_RING of EG3D_D2H_CHUNK bytes, allocated once — behind the
      // first expand launch, see ensure_d2h_ring): the seven arrays are cut into pieces of at most one buffer; a piece
      // crosses PCIe into a free buffer (async, an event behind it) and is copied on by a few host threads into the
      // caller's pageable memory while the next pieces are crossing. (A pageable hipMemcpy runs at ~2 GB/s and made the
      // copy 3x the compute time on the dtu006-shaped workload. Until round 6 a whole unit was staged at once: pinning
      // that much memory — 0.57 GB for C3' — cost the FIRST call of a context 110-160 ms, twice its kernels.)
      const size_t sz[7] = {sizeof(float) * 3 * np, sizeof(eg3d_off_t) * np,   sizeof(uint32_t) * 4 * np, sizeof(int32_t) * no,
                            sizeof(uint32_t) * no,  sizeof(uint32_t) * no,      sizeof(float) * 2 * no};
      const char* src[7] = {(const char*)c->o_X.p,  (const char*)c->o_off.p, (const char*)c->o_key.p, (const char*)c->o_view.p,
                            (const char*)c->o_pl.p, (const char*)c->o_seg.p, (const char*)c->o_xy.p};
      struct Piece {
        int k;
        size_t at, bytes;
      };
      std::vector<Piece> pieces;
      size_t total = 0;
      // (the big observation arrays first)
      static const int order[7] = {6, 3, 4, 5, 2, 1, 0};
      for (int i = 0; i < 7; i++) total += sz[i];
      BUF_TRY(ensure_d2h_ring(c, total));
      const size_t CH = c->ring_chunk;
      for (int i = 0; i < 7; i++) {
        const int k = order[i];
        for (size_t at = 0; at < sz[k]; at += CH) pieces.push_back({k, at, std::min<size_t>(CH, sz[k] - at)});
      }
#ifdef EG3D_COPY_TIMING
      const auto tc0 = std::chrono::steady_clock::now();
#endif
      size_t issued = 0, done = 0;
      auto issue = [&]() -> int {  // the next piece into its ring buffer
        const Piece& q = pieces[issued];
        const int slot = (int)(issued % EG3D_D2H_RING);
        HIP_TRY(hipMemcpyAsync((char*)c->pinned + (size_t)slot * CH, src[q.k] + q.at, q.bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(c->ecopy[slot], st));
        issued++;
        return EG3D_OK;
      };
      // the first pieces cross PCIe whatever the unit's turn ...
      while (issued < pieces.size() && issued < EG3D_D2H_RING) BUF_TRY(issue());
      // ... where they go is known once every earlier unit has placed its output
      if (!T.take()) return EG3D_ERR_HIP;
      uint64_t p0 = 0, o0 = 0;
      const bool placed = S.place_host(np, no, w_piece, p0, o0);
      const uint32_t key0_host = S.key0_next + B.key0_base;
      if (last_chunk) {
        S.unit_tasks(B.n_tasks);
        T.pass();  // the later units place (and copy) while this one copies
      }
      if (!placed) {
        g_err = "eg3d: out of host memory for the edge-point cloud";
        return EG3D_ERR_ARG;
      }
#ifdef EG3D_COPY_TIMING
      const auto tc1 = std::chrono::steady_clock::now();
#endif
      {
        std::shared_lock<std::shared_mutex> lk(S.dst_mu);
        char* dst[7] = {(char*)(S.X.data() + p0 * 3), (char*)(S.off.data() + p0), (char*)(S.key.data() + p0 * 4),
                        (char*)(S.view.data() + o0),  (char*)(S.pl.data() + o0),  (char*)(S.seg.data() + o0),
                        (char*)(S.xy.data() + o0 * 2)};
        while (done < pieces.size()) {
          const Piece& q = pieces[done];
          const int slot = (int)(done % EG3D_D2H_RING);
          HIP_TRY(hipEventSynchronize(c->ecopy[slot]));
          eg3d::copy_mt(dst[q.k] + q.at, (char*)c->pinned + (size_t)slot * CH, q.bytes,
                        c->tune.copy_threads > 0 ? c->tune.copy_threads : std::max(2, EG3D_COPY_THREADS / std::max(1, S.lanes_active)),
                        (size_t)1 << 20);
          done++;
          if (issued < pieces.size()) BUF_TRY(issue());  // the buffer just emptied takes the next piece
        }
        if (o0) {
          uint64_t* off = S.off.data();
          for (size_t i = p0; i < p0 + np; i++) off[i] += (uint64_t)o0;
        }
        if (key0_host) {  // polyline-set calls: key[0] counts the samples of the whole call
          uint32_t* key = S.key.data();
          for (size_t i = p0; i < p0 + np; i++) key[4 * i] += key0_host;
        }
      }
#ifdef EG3D_COPY_TIMING
      const auto tc2 = std::chrono::steady_clock::now();
      fprintf(stderr, "chunk: p0 %zu o0 %zu np %u no %u nc %u  ", (size_t)p0, (size_t)o0, np, no, nc);
      fprintf(stderr, "copy timing: %.1f MB in %zu pieces  first pieces + turn %.2f ms  D2H + host copy %.2f ms\n", total / 1e6, pieces.size(),
              std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
              std::chrono::duration<double, std::milli>(tc2 - tc1).count());
#endif
    } else if (last_chunk) {
      if (!T.take()) return EG3D_ERR_HIP;
      S.unit_tasks(B.n_tasks);
      T.pass();
    }
    HIP_TRY(hipStreamSynchronize(st));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, c->ea[5], c->eb[5]));
    ms_expand += t;
    HIP_TRY(hipEventElapsedTime(&t, c->ea[6], c->eb[6]));
    ms_emit += t;
    H.n_points += np;
    H.n_obs += no;
    H.pieces++;
    c->last_np = np;  // (the owner's view of the whole call is set when the call ends)
    c->last_no = no;
    c->last_nc = nc;
  }
  if (!B.n_chains) {  // a unit without chains still takes its turn
    if (!T.take()) return EG3D_ERR_HIP;
    S.unit_tasks(B.n_tasks);
    T.pass();
  }
  HIP_TRY(hipStreamSynchronize(st));
  float t = 0;
  for (int k = 1; k <= 4; k++) {
    if ((k == 2 && !B.n_tasks) || (k == 3 && !B.n_hyp) || (k == 4 && !B.n_tasks)) continue;
    HIP_TRY(hipEventElapsedTime(&t, c->ea[k], c->eb[k]));
    H.ms[k] += t;
  }
  H.ms[5] += ms_expand;
  H.ms[6] += ms_emit;
  if (!B.n_chains) {
    HIP_TRY(hipMemcpy(&hc, c->b_ctr.p, sizeof(Counters), hipMemcpyDeviceToHost));
    H.bytes_vertices = hc.bytes;
  }
  H.bytes_algorithmic += H.bytes_vertices;
  H.bytes_vertices = 0;
  H.n_tasks += B.n_tasks;
  H.n_hyp += B.n_hyp;
  c->last_nhyp = B.n_hyp;
  H.n_chains += B.n_chains;
  return EG3D_OK;
}

int run_batch(eg3d_ctx* c, uint32_t b, uint32_t e, UnitTurn& T, HostOut& H) {
  BatchState B;
  memset(&B, 0, sizeof(B));
  B.b = b;
  B.e = e;
  BUF_TRY(run_stage_a(c, B, nullptr));
  BUF_TRY(run_stage_b(c, B, T, H));
  // per seed 12 + k*12 + k*64 + k(k-1)*72 (the vertices touched and the output were
  // added by stage B)
  for (uint32_t sd_i = b; sd_i < e; sd_i++) {
    const uint64_t k = (*c->h_trk)[sd_i + 1] - (*c->h_trk)[sd_i];
    H.bytes_algorithmic += 12 + k * 12 + k * 64 + k * (k - 1) * 72;
  }
  return EG3D_OK;
}

// Pipelines 1-2 extractor, stage A: sample the polylines of sets [set_b, set_e) and
// collect the epipolar hits of every sample; then the common stage B. `sets` is already on the
// device; h_row_off is its host copy of the row offsets.
int run_sets_batch(eg3d_ctx* c, const SetsDev& sets, const uint32_t* h_row_off, uint32_t n_rows_total, uint32_t set_b,
                   uint32_t set_e, UnitTurn& T, HostOut& H) {
  hipStream_t st = c->stream;
  const uint32_t V = (uint32_t)c->V;
  BatchState B;
  memset(&B, 0, sizeof(B));
  B.key0_base = 0;  // (key[0] = sample index of the CALL: the sink adds the samples of the units before this one)
  const uint32_t item_b = h_row_off[(size_t)set_b * V], item_e = h_row_off[(size_t)set_e * V];
  const uint32_t n_items = item_e - item_b;
  BUF_TRY(c->b_ctr.ensure(sizeof(Counters)));
  HIP_TRY(hipMemsetAsync(c->b_ctr.p, 0, sizeof(Counters), st));
  BUF_TRY(c->b_raw_cnt.ensure(sizeof(uint32_t) * (n_items + 1)));
  BUF_TRY(c->b_raw_off.ensure(sizeof(uint32_t) * (n_items + 1)));
  HIP_TRY(hipMemsetAsync(c->b_raw_cnt.as<uint32_t>() + n_items, 0, sizeof(uint32_t), st));
  HIP_TRY(hipEventRecord(c->ea[1], st));
  launch_n1_samples(st, false, c->ds, sets, n_rows_total, item_b, n_items, c->b_raw_cnt.as<uint32_t>(), nullptr, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, c->b_ctr.as<Counters>());
  BUF_TRY(scan_total_u32(c, c->b_raw_cnt.as<uint32_t>(), c->b_raw_off.as<uint32_t>(), n_items + 1, B.n_tasks, "samples"));
  const uint32_t nt = B.n_tasks;
  if ((uint64_t)nt * V > 0x7fffffffull) {
    g_err = "eg3d_match_polyline_sets: too many samples in one batch of sets";
    return EG3D_ERR_CAPACITY;
  }
  BUF_TRY(c->b_start_hits.ensure(sizeof(Obs) * (nt + 1)));  // the samples
  BUF_TRY(c->b_task_seed.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_entry.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_hit.ensure(sizeof(uint32_t) * (nt + 1)));
  BUF_TRY(c->b_task_k.ensure(sizeof(uint32_t) * (nt + 1)));  // first row of the task's set
  BUF_TRY(c->b_task_list_off.ensure(sizeof(uint32_t) * (nt + 1)));
  launch_n1_samples(st, true, c->ds, sets, n_rows_total, item_b, n_items, nullptr, c->b_raw_off.as<uint32_t>(),
                    c->b_start_hits.as<Obs>(), c->b_task_seed.as<uint32_t>(), c->b_task_entry.as<uint32_t>(),
                    c->b_task_hit.as<uint32_t>(), c->b_task_list_off.as<uint32_t>(), c->b_task_k.as<uint32_t>(),
                    c->b_ctr.as<Counters>());
  {
    const uint32_t last = nt * V;
    HIP_TRY(hipMemcpyAsync(c->b_task_list_off.as<uint32_t>() + nt, &last, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `last` leaves scope
  }
  HIP_TRY(hipEventRecord(c->eb[1], st));
  B.n_lists = nt * V;
  BUF_TRY(c->b_list_cnt.ensure(sizeof(uint32_t) * ((size_t)B.n_lists + 1)));
  BUF_TRY(c->b_list_ptr.ensure(sizeof(uint32_t) * ((size_t)B.n_lists + 1)));
  HIP_TRY(hipMemsetAsync(c->b_list_cnt.as<uint32_t>() + B.n_lists, 0, sizeof(uint32_t), st));
  HIP_TRY(hipEventRecord(c->ea[2], st));
  launch_n1_hits(st, false, c->ds, sets, nt, c->b_start_hits.as<Obs>(), c->b_task_k.as<uint32_t>(),
                 c->b_list_cnt.as<uint32_t>(), nullptr, nullptr, c->b_ctr.as<Counters>());
  BUF_TRY(scan_total_u32(c, c->b_list_cnt.as<uint32_t>(), c->b_list_ptr.as<uint32_t>(), (size_t)B.n_lists + 1, B.n_hits,
                         "epipolar hits"));
  BUF_TRY(c->b_hits.ensure(sizeof(Obs) * ((size_t)B.n_hits + 1)));
  launch_n1_hits(st, true, c->ds, sets, nt, c->b_start_hits.as<Obs>(), c->b_task_k.as<uint32_t>(),
                 c->b_list_cnt.as<uint32_t>(), c->b_list_ptr.as<uint32_t>(), c->b_hits.as<Obs>(), c->b_ctr.as<Counters>());
  HIP_TRY(hipEventRecord(c->eb[2], st));
  // identity view map: one row of V entries shared by every sample (StageAView::dense_k)
  {
    std::vector<int32_t> mv(V);
    std::vector<uint32_t> me(V);
    for (uint32_t j = 0; j < V; j++) {
      mv[j] = (int32_t)j;
      me[j] = j;
    }
    BUF_TRY(upload(c->b_map_view, mv.data(), V, st));
    BUF_TRY(upload(c->b_map_entry, me.data(), V, st));
    BUF_TRY(c->b_map_n.ensure(sizeof(uint32_t)));
    HIP_TRY(hipStreamSynchronize(st));
  }
  StageAView& a = B.a;
  a.trk_off = nullptr;
  a.trk_view = nullptr;
  a.trk_xy = nullptr;
  a.seed_begin = 0;
  a.sv_base = 0;
  a.n_tasks = nt;
  a.task_seed = c->b_task_seed.as<uint32_t>();
  a.task_entry = c->b_task_entry.as<uint32_t>();
  a.task_hit = c->b_task_hit.as<uint32_t>();
  a.task_list_off = c->b_task_list_off.as<uint32_t>();
  a.list_ptr = c->b_list_ptr.as<uint32_t>();
  a.list_cnt = c->b_list_cnt.as<uint32_t>();
  a.hits = c->b_hits.as<Obs>();
  a.dense_k = V;
  BUF_TRY(run_stage_b(c, B, T, H));
  // per sample: its coordinates, V camera matrices, V-1 fundamental matrices (the scanned vertices
  // and the output were added by the kernels)
  H.bytes_algorithmic += (uint64_t)nt * (8 + (uint64_t)V * 64 + (uint64_t)(V - 1) * 72);
  return EG3D_OK;
}

template <typename T>
T* dup_to_malloc(const std::vector<T>& v, size_t extra = 0) {
  T* p = (T*)malloc(sizeof(T) * std::max<size_t>(1, v.size() + extra));
  if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

}  // namespace

// ---- one call = units on lanes ----------------------------------------------------------------------------------------
// The reference's parallel entry point runs the seeds of ONE call on its OpenMP team (plg_matching_from_refpoints.cpp:83-104,
// `#pragma omp parallel for`). The analogue here: a call's range is cut into units (contiguous, balanced by the sum of track
// lengths), the units run concurrently on the context's lanes — internal contexts with their own HIP stream, work buffers
// and host thread — and their clouds are concatenated in unit order (CallSink). What that buys on one GPU: the candidate /
// hypothesis stages and the D2H copy of one unit overlap the expand stage of another, and the thin tail of one expand launch
// (a few long chains on an otherwise empty GPU) is filled by the next unit's chains.
static void lane_share_inputs(eg3d_ctx* owner, eg3d_ctx* l) {
  if (l == owner) return;
  l->n_seeds = owner->n_seeds;
  l->h_trk = owner->h_trk;
  l->b_toff = owner->b_toff;
  l->b_tview = owner->b_tview;
  l->b_txy = owner->b_txy;
  l->seeds_owner = owner->seeds_owner;
  // what a lane has learned about the scene's needs serves all of them (capacities only grow)
  l->chain_cap = std::max(l->chain_cap, owner->chain_cap);
  l->pool_cap = std::max(l->pool_cap, owner->pool_cap);
  l->arena_per_hyp = std::max(l->arena_per_hyp, owner->arena_per_hyp);
  l->k3b_long_latched = l->k3b_long_latched || owner->k3b_long_latched;
}
static void lane_return_learned(eg3d_ctx* owner, const eg3d_ctx* l) {
  if (l == owner) return;
  owner->chain_cap = std::max(l->chain_cap, owner->chain_cap);
  owner->pool_cap = std::max(l->pool_cap, owner->pool_cap);
  owner->arena_per_hyp = std::max(l->arena_per_hyp, owner->arena_per_hyp);
  owner->k3b_long_latched = l->k3b_long_latched || owner->k3b_long_latched;
}
static int ensure_lanes(eg3d_ctx* c, int n) {
  if (c->lanes.empty()) c->lanes.push_back(c);
  while ((int)c->lanes.size() < n) {
    eg3d_ctx* l = nullptr;
    BUF_TRY(eg3d_clone(c, &l));
    l->is_lane = true;
    l->tune.lanes = 1;
    if (c->tune.lane_priorities) {
      int least = 0, greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      (void)hipStreamDestroy(l->stream);
      l->stream = nullptr;
      const int mid = (least + greatest) / 2;
      HIP_TRY(hipStreamCreateWithPriority(&l->stream, hipStreamNonBlocking, c->lanes.size() == 1 ? mid : least));
    }
    c->lanes.push_back(l);
  }
  return EG3D_OK;
}

struct UnitRange {
  uint32_t b, e;
  double w;
};
// Cuts [b, e) into contiguous units of (nearly) equal weight; cum[i] = weight of items [0, i) (ascending).
template <typename Cum>
static std::vector<UnitRange> cut_by_weight(uint32_t b, uint32_t e, uint32_t n_units, Cum cum, double ramp = 1.0) {
  std::vector<UnitRange> out;
  if (b >= e) return out;
  n_units = std::max(1u, std::min(n_units, e - b));
  const double w0 = (double)cum(b), w1 = (double)cum(e);
  // unit i gets a share proportional to ramp^i (1 = equal units; < 1 = the later units are smaller)
  double share_all = 0, share_done = 0, r = 1.0;
  for (uint32_t k = 0; k < n_units; k++, r *= ramp) share_all += r;
  r = 1.0;
  uint32_t at = b;
  for (uint32_t k = 1; k <= n_units && at < e; k++) {
    uint32_t end = e;
    share_done += r;
    r *= ramp;
    if (k < n_units) {
      const double target = w0 + (w1 - w0) * share_done / share_all;
      uint32_t lo = at + 1, hi = e;  // first index whose prefix weight reaches the target (at least one item per unit)
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((double)cum(mid) < target) lo = mid + 1; else hi = mid;
      }
      end = std::min(e - (n_units - k), std::max(at + 1, lo));  // leave an item for each unit still to come
    }
    out.push_back({at, end, (double)cum(end) - (double)cum(at)});
    at = end;
  }
  return out;
}

// Runs the units on up to n_lanes lanes (the calling thread drives lane 0). run(lane, unit index, turn, stats) -> status.
template <typename Run>
static int run_pipelined(eg3d_ctx* c, CallSink& S, const std::vector<UnitRange>& units, int n_lanes, Run run) {
  const uint32_t n_units = (uint32_t)units.size();
  n_lanes = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)std::max(1, n_lanes), n_units));
  BUF_TRY(ensure_lanes(c, n_lanes));
  S.lanes_active = n_lanes;
  for (int l = 0; l < n_lanes; l++) lane_share_inputs(c, c->lanes[(size_t)l]);
  for (const UnitRange& u : units) S.weight_total += u.w;
  std::atomic<uint32_t> next{0};
  auto worker = [&](eg3d_ctx* lane) {
    if (hipSetDevice(lane->device) != hipSuccess) {
      S.fail(EG3D_ERR_HIP, "eg3d: hipSetDevice failed on a pipeline thread");
      return;
    }
    for (uint32_t u; (u = next.fetch_add(1)) < n_units;) {
      UnitTurn T{S, u, units[u].w};
      HostOut H;
      const int rc = run(lane, u, T, H);
      if (rc != EG3D_OK) {
        S.fail(rc, g_err);  // (this thread's message; the first failure of the call is the one reported)
        return;
      }
      T.pass();
      S.add(H);
    }
  };
  std::vector<std::thread> th;
  try {
    for (int l = 1; l < n_lanes; l++) th.emplace_back(worker, c->lanes[(size_t)l]);
  } catch (...) {  // no more threads: the lanes that did start (and this thread) take all units
  }
  worker(c);
  for (auto& t : th) t.join();
  for (int l = 0; l < n_lanes; l++) lane_return_learned(c, c->lanes[(size_t)l]);
  if (S.failed) {
    g_err = S.err;
    return S.rc;
  }
  return EG3D_OK;
}

extern "C" void eg3d_free_edgepoints(eg3d_edgepoints* e);
static int finish_match(eg3d_ctx* c, CallSink& S, float total, eg3d_edgepoints* out, eg3d_stage_times* times) {
  const HostOut& H = S.tot;
  const int device_only = S.device_only;
  out->n_points = S.n_points;
  out->n_obs = S.n_obs;
  out->n_tasks = H.n_tasks;
  out->n_hypotheses = H.n_hyp;
  out->n_chains = H.n_chains;
  out->flags = H.flags;
  c->last_chunks = H.pieces;
  c->last_accumulated = device_only != 0;
  if (!device_only) {
    c->host_calls++;
    c->last_host_cloud_bytes = 36 * S.n_points + 20 * S.n_obs;
  }
  if (device_only) {
    c->last_np = S.n_points;
    c->last_no = S.n_obs;
  }
  if (!device_only) {
    const size_t npts = (size_t)S.n_points;
    if (!S.off.reserve(npts + 1)) {
      g_err = "eg3d: out of host memory for the edge-point cloud";
      return EG3D_ERR_ARG;
    }
    S.off[npts] = S.n_obs;
    out->X = S.X.release();
    out->obs_off = S.off.release();
    out->obs_view = S.view.release();
    out->obs_pl = S.pl.release();
    out->obs_seg = S.seg.release();
    out->obs_xy = S.xy.release();
    out->key = S.key.release();
    out->_owner = (void*)1;
  }
  if (times) {
    times->ms_total = total;
    times->ms_candidates = H.ms[1];
    times->ms_epipolar = H.ms[2];
    times->ms_hypotheses = H.ms[3];
    times->ms_select = H.ms[4];
    times->ms_expand = H.ms[5];
    times->ms_emit = H.ms[6];
    times->bytes_algorithmic = H.bytes_algorithmic + 12 * S.n_points + 20 * S.n_obs;
    times->ms_slowest_chain = c->wall_clock_khz > 0 ? (float)((double)H.max_chain_ticks / (double)c->wall_clock_khz) : 0.0f;
  }
  if (H.flags & (EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW | EG3D_FLAG_HYP_OVERFLOW)) {
    g_err = "eg3d: a device-side capacity was exceeded (flags in out->flags)";
    const uint32_t flags = out->flags;
    eg3d_free_edgepoints(out);  // an error return owns nothing the caller would have to free
    out->flags = flags;
    return EG3D_ERR_CAPACITY;
  }
  return EG3D_OK;
}

extern "C" int eg3d_set_pipelining(eg3d_ctx* c, int lanes, int units) {
  if (!c || lanes < 0 || units < 0) {
    g_err = "eg3d_set_pipelining: bad arguments";
    return EG3D_ERR_ARG;
  }
  c->tune.lanes = std::min(16, lanes);
  c->tune.units = std::min(4096, units);
  return EG3D_OK;
}

static int lanes_for(const eg3d_ctx* c, int device_only) {
  if (c->tune.lanes > 0) return c->tune.lanes;
  // by measurement (round 6): cutting a call buys nothing on the device — every expand launch lasts at least as
  // long as its slowest chain and the launches of a call's units run nearly first-in-first-out — but it hides most of the
  // D2H copy of a host call (C3' 64 -> 54 ms). Lanes have a price when they are created (streams, work buffers, pinned
  // rings: +50-90 ms on the call that creates them, seconds on a 200-view scene), so the FIRST host call of a context — a
  // one-shot caller's only call — runs on the context alone; a context that is called again is worth the lanes.
  if (device_only) return 1;
  // every lane owns a full slot arena (the residency of the expand kernel x the slice size): on many-view scenes that is
  // 17 GB per lane (C4) for a 3 % gain with the copy and a loss without — such scenes stay on the context alone
  const size_t arena = chain_layout(c->chain_cap, c->pool_cap, (uint32_t)c->V).total * 8 * (size_t)c->slots_per_xcd;
  if (arena > ((size_t)4 << 30)) return 1;
  // ... and a small cloud has no copy worth hiding (C3-real, 0.3 MB: 3.3 ms uncut, 5.1 ms on three lanes)
  return c->host_calls > 0 && c->last_host_cloud_bytes >= ((uint64_t)4 << 20) ? Tunables::kHostCallLanes : 1;
}

// Units of a seed range. One lane: batches of 16 384 seeds (one expand launch each), as before round 6. Several lanes:
// at least one unit per lane when the range is worth cutting (>= 128 seeds per unit), never more than 16 384 seeds per unit,
// balanced by the sum of track lengths (the same weight the multi-GPU sharding uses).
static std::vector<UnitRange> plan_seed_units(const eg3d_ctx* c, uint32_t b, uint32_t e, int lanes) {
  const uint32_t SEED_BATCH = 16384, MIN_UNIT = 128;
  const uint32_t n = e - b;
  const std::vector<uint32_t>& trk = *c->h_trk;
  auto cum = [&](uint32_t i) { return (double)trk[i] + 1e-3 * (double)i; };  // (+ a little per seed: seeds without tracks still count)
  if (lanes <= 1 && !c->tune.units) {
    std::vector<UnitRange> out;
    for (uint32_t s0 = b; s0 < e; s0 += SEED_BATCH) {
      const uint32_t s1 = std::min(e, s0 + SEED_BATCH);
      out.push_back({s0, s1, cum(s1) - cum(s0)});
    }
    return out;
  }
  uint32_t n_units = c->tune.units ? (uint32_t)c->tune.units : std::min<uint32_t>((uint32_t)lanes, n / MIN_UNIT);
  n_units = std::max(n_units, (n + SEED_BATCH - 1) / SEED_BATCH);
  std::vector<UnitRange> out = cut_by_weight(b, e, std::max(1u, n_units), cum, c->tune.unit_ramp);
  // (balancing by weight can leave a unit of many short tracks above the batch bound: cut such a unit again)
  for (size_t i = 0; i < out.size();) {
    if (out[i].e - out[i].b > SEED_BATCH) {
      const uint32_t mid = out[i].b + (out[i].e - out[i].b) / 2, end = out[i].e;
      out[i] = {out[i].b, mid, cum(mid) - cum(out[i].b)};
      out.insert(out.begin() + (long)i + 1, {mid, end, cum(end) - cum(mid)});
    } else {
      i++;
    }
  }
  return out;
}

extern "C" int eg3d_match_resident(eg3d_ctx* c, uint32_t b, uint32_t e, int device_only, eg3d_edgepoints* out,
                                   eg3d_stage_times* times) {
  if (!c || !out || b > e || e > c->n_seeds) {
    g_err = "eg3d_match_resident: bad arguments (seeds uploaded?)";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  memset(out, 0, sizeof(*out));
  CallSink S;
  S.owner = c;
  S.device_only = device_only;
  c->last_np = c->last_no = 0;
  c->last_chunks = 0;
  const int lanes = lanes_for(c, device_only);
  const std::vector<UnitRange> units = plan_seed_units(c, b, e, lanes);
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipEventRecord(c->ea[7], c->stream));  // the context's own events: nothing to leak on an error path
  int rc = EG3D_OK;
  if (!units.empty())
    rc = run_pipelined(c, S, units, lanes, [&](eg3d_ctx* lane, uint32_t u, UnitTurn& T, HostOut& H) {
      return run_batch(lane, units[u].b, units[u].e, T, H);
    });
  if (rc != EG3D_OK) return rc;
  HIP_TRY(hipEventRecord(c->eb[7], c->stream));
  HIP_TRY(hipEventSynchronize(c->eb[7]));
  float total = 0;
  HIP_TRY(hipEventElapsedTime(&total, c->ea[7], c->eb[7]));
  // (several lanes: the context's own stream saw one of them only — the call's wall time is the total)
  if (units.size() > 1 && lanes > 1)
    total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return finish_match(c, S, total, out, times);
}

extern "C" int eg3d_last_device_output(eg3d_ctx* c, eg3d_device_edgepoints* out) {
  if (!c || !out) {
    g_err = "eg3d_last_device_output: bad arguments";
    return EG3D_ERR_ARG;
  }
  out->n_points = c->last_np;
  out->n_obs = c->last_no;
  out->X = c->o_X.as<float>();
  out->obs_off = c->o_off.as<eg3d_off_t>();
  out->obs_view = c->o_view.as<int32_t>();
  out->obs_pl = c->o_pl.as<uint32_t>();
  out->obs_seg = c->o_seg.as<uint32_t>();
  out->obs_xy = c->o_xy.as<float>();
  out->key = c->o_key.as<uint32_t>();
  out->complete = (c->last_accumulated || c->last_chunks <= 1) ? 1 : 0;
  return EG3D_OK;
}

extern "C" int eg3d_match_refpoints(eg3d_ctx* c, const eg3d_seeds* seeds, uint32_t b, uint32_t e, int device_only,
                                    eg3d_edgepoints* out, eg3d_stage_times* times) {
  int rc = eg3d_upload_seeds(c, seeds);
  if (rc != EG3D_OK) return rc;
  return eg3d_match_resident(c, b, e, device_only, out, times);
}

extern "C" int eg3d_match_polyline_sets(eg3d_ctx* c, const eg3d_polyline_sets* ps, uint32_t set_b, uint32_t set_e,
                                        int device_only, eg3d_edgepoints* out, eg3d_stage_times* times) {
  if (!c || !ps || !out || !ps->row_off || set_b > set_e || set_e > ps->n_sets) {
    g_err = "eg3d_match_polyline_sets: bad arguments";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  memset(out, 0, sizeof(*out));
  const uint32_t V = (uint32_t)c->V;
  const uint32_t n_rows = ps->n_sets * V;
  const uint32_t n_ids = ps->row_off[n_rows];
  {
    std::vector<uint32_t> vpo(V + 1);
    HIP_TRY(hipMemcpy(vpo.data(), c->ds.view_pl_off, sizeof(uint32_t) * (V + 1), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_rows; r++) {
      if (ps->row_off[r + 1] < ps->row_off[r]) {
        g_err = "eg3d_match_polyline_sets: row_off is not ascending";
        return EG3D_ERR_ARG;
      }
      const uint32_t npl = vpo[r % V + 1] - vpo[r % V];
      for (uint32_t k = ps->row_off[r]; k < ps->row_off[r + 1]; k++)
        if (ps->pl_ids[k] >= npl) {
          g_err = "eg3d_match_polyline_sets: polyline id out of range";
          return EG3D_ERR_ARG;
        }
    }
  }
  BUF_TRY(upload(c->b_sets_off, ps->row_off, (size_t)n_rows + 1, c->stream));
  BUF_TRY(upload(c->b_sets_ids, ps->pl_ids, n_ids, c->stream));
  SetsDev sd;
  sd.n_views = V;
  sd.row_off = c->b_sets_off.as<uint32_t>();
  sd.pl_ids = c->b_sets_ids.as<uint32_t>();
  HIP_TRY(hipStreamSynchronize(c->stream));  // the sets are resident before any lane reads them
  CallSink S;
  S.owner = c;
  S.device_only = device_only;
  S.keyed_by_sample = true;
  c->last_np = c->last_no = 0;
  c->last_chunks = 0;
  const int lanes = lanes_for(c, device_only);
  // units = runs of whole sets, bounded by the number of polylines (every sample owns V lists); with several lanes at
  // least one unit per lane, balanced by the number of polylines
  const uint32_t max_items = V >= 64 ? 2048u : 16384u;
  std::vector<UnitRange> units;
  {
    auto cum = [&](uint32_t i) { return (double)ps->row_off[(size_t)i * V] + 1e-3 * (double)i; };
    const uint32_t want = c->tune.units ? (uint32_t)c->tune.units : (uint32_t)lanes;
    for (const UnitRange& r : cut_by_weight(set_b, set_e, want, cum))
      for (uint32_t s0 = r.b; s0 < r.e;) {  // (a unit above the bound is cut into runs of whole sets within it)
        uint32_t s1 = s0 + 1;
        while (s1 < r.e && ps->row_off[(size_t)(s1 + 1) * V] - ps->row_off[(size_t)s0 * V] <= max_items) s1++;
        units.push_back({s0, s1, cum(s1) - cum(s0)});
        s0 = s1;
      }
  }
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipEventRecord(c->ea[7], c->stream));
  int rc = EG3D_OK;
  if (!units.empty())
    rc = run_pipelined(c, S, units, lanes, [&](eg3d_ctx* lane, uint32_t u, UnitTurn& T, HostOut& H) {
      return run_sets_batch(lane, sd, ps->row_off, n_rows, units[u].b, units[u].e, T, H);
    });
  if (rc != EG3D_OK) return rc;
  HIP_TRY(hipEventRecord(c->eb[7], c->stream));
  HIP_TRY(hipEventSynchronize(c->eb[7]));
  float total = 0;
  HIP_TRY(hipEventElapsedTime(&total, c->ea[7], c->eb[7]));
  if (units.size() > 1 && lanes > 1)
    total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return finish_match(c, S, total, out, times);
}

extern "C" void eg3d_free_edgepoints(eg3d_edgepoints* e) {
  if (!e) return;
  free(e->X);
  free(e->obs_off);
  free(e->obs_view);
  free(e->obs_pl);
  free(e->obs_seg);
  free(e->obs_xy);
  free(e->key);
  memset(e, 0, sizeof(*e));
}

extern "C" int eg3d_candidates_run(eg3d_ctx* c, const eg3d_seeds* seeds, uint32_t b, uint32_t e, eg3d_candidates* out) {
  if (!c || !seeds || !out || b > e || e > seeds->n_seeds) {
    g_err = "eg3d_candidates_run: bad arguments";
    return EG3D_ERR_ARG;
  }
  int rc = eg3d_upload_seeds(c, seeds);
  if (rc != EG3D_OK) return rc;
  memset(out, 0, sizeof(*out));
  BatchState B;
  memset(&B, 0, sizeof(B));
  B.b = b;
  B.e = e;
  BUF_TRY(run_stage_a(c, B, nullptr));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const uint32_t n_sv = B.n_sv, nt = B.n_tasks;
  std::vector<uint32_t> raw_off(n_sv + 1), cand_cnt(n_sv + 1), start_cnt(n_sv + 1), cand_pl(B.total_raw + 1),
      task_seed(nt + 1), task_entry(nt + 1), task_hit(nt + 1), task_list_off(nt + 1), list_cnt(B.n_lists + 1),
      list_ptr(B.n_lists + 1);
  std::vector<Obs> start_hits(B.total_raw + 1), hits(B.n_hits + 1);
#define DL(vec, buf, n) \
  if (n) HIP_TRY(hipMemcpy(vec.data(), c->buf.p, sizeof(vec[0]) * (n), hipMemcpyDeviceToHost))
  DL(raw_off, b_raw_off, n_sv + 1);
  DL(cand_cnt, b_cand_cnt, n_sv);
  DL(start_cnt, b_start_cnt, n_sv);
  DL(cand_pl, b_cand_pl, B.total_raw);
  DL(start_hits, b_start_hits, B.total_raw);
  DL(task_seed, b_task_seed, nt);
  DL(task_entry, b_task_entry, nt);
  DL(task_hit, b_task_hit, nt);
  DL(task_list_off, b_task_list_off, nt + 1);
  DL(list_cnt, b_list_cnt, B.n_lists);
  DL(list_ptr, b_list_ptr, B.n_lists + 1);
  DL(hits, b_hits, B.n_hits);
#undef DL
  std::vector<uint32_t> o_cand_off(1, 0), o_cand_pl, o_start_off(1, 0), o_start_pl, o_start_seg, o_task_sv(nt),
      o_hit_pl(B.n_hits), o_hit_seg(B.n_hits);
  std::vector<float> o_start_xy, o_hit_xy((size_t)B.n_hits * 2);
  for (uint32_t sv = 0; sv < n_sv; sv++) {
    for (uint32_t i = 0; i < cand_cnt[sv]; i++) o_cand_pl.push_back(cand_pl[raw_off[sv] + i]);
    o_cand_off.push_back((uint32_t)o_cand_pl.size());
    for (uint32_t i = 0; i < start_cnt[sv]; i++) {
      const Obs& o = start_hits[raw_off[sv] + i];
      o_start_pl.push_back(o.pl);
      o_start_seg.push_back(o.seg);
      o_start_xy.push_back(o.x);
      o_start_xy.push_back(o.y);
    }
    o_start_off.push_back((uint32_t)o_start_pl.size());
  }
  for (uint32_t t = 0; t < nt; t++) o_task_sv[t] = (*c->h_trk)[task_seed[t]] - B.sv_base + task_entry[t];
  for (uint32_t h = 0; h < B.n_hits; h++) {
    o_hit_pl[h] = hits[h].pl;
    o_hit_seg[h] = hits[h].seg;
    o_hit_xy[2 * h] = hits[h].x;
    o_hit_xy[2 * h + 1] = hits[h].y;
  }
  task_hit.resize(nt);
  list_ptr.resize(B.n_lists + 1);
  task_list_off.resize(nt + 1);
  out->n_sv = n_sv;
  out->cand_off = dup_to_malloc(o_cand_off);
  out->cand_pl = dup_to_malloc(o_cand_pl);
  out->start_off = dup_to_malloc(o_start_off);
  out->start_pl = dup_to_malloc(o_start_pl);
  out->start_seg = dup_to_malloc(o_start_seg);
  out->start_xy = dup_to_malloc(o_start_xy);
  out->n_tasks = nt;
  out->task_sv = dup_to_malloc(o_task_sv);
  out->task_hit = dup_to_malloc(task_hit);
  out->task_list_off = dup_to_malloc(task_list_off);
  out->list_off = dup_to_malloc(list_ptr);  // the exclusive scan of the counts is the CSR
  out->hit_pl = dup_to_malloc(o_hit_pl);
  out->hit_seg = dup_to_malloc(o_hit_seg);
  out->hit_xy = dup_to_malloc(o_hit_xy);
  out->_owner = (void*)1;
  return EG3D_OK;
}

extern "C" void eg3d_free_candidates(eg3d_candidates* c) {
  if (!c) return;
  free(c->cand_off);
  free(c->cand_pl);
  free(c->start_off);
  free(c->start_pl);
  free(c->start_seg);
  free(c->start_xy);
  free(c->task_sv);
  free(c->task_hit);
  free(c->task_list_off);
  free(c->list_off);
  free(c->hit_pl);
  free(c->hit_seg);
  free(c->hit_xy);
  memset(c, 0, sizeof(*c));
}

extern "C" int eg3d_gn_filter(eg3d_ctx* c, const float* X, const uint32_t* obs_off, const int32_t* obs_view,
                              const float* obs_xy, uint64_t n, float gn_max_mse, int legacy_abs, float* X_out,
                              uint8_t* inlier, float* ms_kernel) {
  if (!c || !X || !obs_off || !X_out || !inlier) {
    g_err = "eg3d_gn_filter: bad arguments";
    return EG3D_ERR_ARG;
  }
  if (n >= 0xffffffffull) {
    g_err = "eg3d_gn_filter: more than 2^32-2 points in one call";
    return EG3D_ERR_CAPACITY;
  }
  if (n && obs_off[0] != 0) {
    g_err = "eg3d_gn_filter: obs_off[0] must be 0";
    return EG3D_ERR_ARG;
  }
  for (uint64_t i = 0; i < n; i++)
    if (obs_off[i + 1] < obs_off[i]) {
      g_err = "eg3d_gn_filter: obs_off is not ascending";
      return EG3D_ERR_ARG;
    }
  const uint64_t m = n ? obs_off[n] : 0;
  if (m && (!obs_view || !obs_xy)) {
    g_err = "eg3d_gn_filter: null observation arrays";
    return EG3D_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(c->device));
  for (uint64_t i = 0; i < m; i++)
    if (obs_view[i] < 0 || obs_view[i] >= c->V) {
      g_err = "eg3d_gn_filter: view id out of range";
      return EG3D_ERR_ARG;
    }
  hipStream_t st = c->stream;
  BUF_TRY(upload(c->f_X, X, n * 3, st));
  BUF_TRY(upload(c->f_off, obs_off, n + 1, st));
  BUF_TRY(upload(c->f_view, obs_view, m, st));
  BUF_TRY(upload(c->f_xy, obs_xy, m * 2, st));
  BUF_TRY(c->f_Xo.ensure(sizeof(float) * 3 * (n + 1)));
  BUF_TRY(c->f_inl.ensure(n + 1));
  HIP_TRY(hipEventRecord(c->ea[0], st));
  launch_k5(st, c->ds.cam_P, c->V, c->f_X.as<float>(), c->f_off.as<uint32_t>(), c->f_view.as<int32_t>(), c->f_xy.as<float>(),
            n, gn_max_mse, legacy_abs, c->f_Xo.as<float>(), c->f_inl.as<uint8_t>());
  HIP_TRY(hipEventRecord(c->eb[0], st));
  if (n) {
    HIP_TRY(hipMemcpyAsync(X_out, c->f_Xo.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(inlier, c->f_inl.p, n, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (ms_kernel) HIP_TRY(hipEventElapsedTime(ms_kernel, c->ea[0], c->eb[0]));
  return EG3D_OK;
}

#ifdef EG3D_SECTION_TIMING
// Tuning diagnostics, present only in builds made with -DEG3D_SECTION_TIMING (tools/section_timing*.py):
// per-section shader-clock ticks of the most recent k3b_expand launch, summed over the chains
// (sum[16]) and of the slowest chain (slowest[16]).
extern "C" int eg3d_probe_sections(eg3d_ctx* c, double* sum, double* slowest, uint32_t* n_chains) {
  if (!c || !sum || !slowest) return EG3D_ERR_ARG;
  std::vector<ChainOut> co(c->last_nc ? c->last_nc : 1);
  if (c->last_nc) HIP_TRY(hipMemcpy(co.data(), c->b_couts.p, sizeof(ChainOut) * c->last_nc, hipMemcpyDeviceToHost));
  // sum / slowest hold 16 doubles: 0..11 the section ticks, 12..14 the follow diagnostics packed in
  // tsec[11] (sequential N-view steps, look-ahead steps accepted, look-ahead rounds redone)
  for (int k = 0; k < 16; k++) sum[k] = slowest[k] = 0;
  uint64_t worst = 0;
  for (uint32_t j = 0; j < c->last_nc; j++) {
    for (int k = 0; k < 12; k++) sum[k] += (double)co[j].tsec[k];
    sum[12] += (double)(co[j].tsec[11] & 0xffffu);
    sum[13] += (double)((co[j].tsec[11] >> 16) & 0xffffu);
    sum[14] += (double)(co[j].tsec[11] >> 32);
    if (co[j].tsec[7] >= worst) {
      worst = co[j].tsec[7];
      for (int k = 0; k < 12; k++) slowest[k] = (double)co[j].tsec[k];
      slowest[12] = (double)(co[j].tsec[11] & 0xffffu);
      slowest[13] = (double)((co[j].tsec[11] >> 16) & 0xffffu);
      slowest[14] = (double)(co[j].tsec[11] >> 32);
      slowest[15] = (double)co[j].n_points;
    }
  }
  if (n_chains) *n_chains = c->last_nc;
  return EG3D_OK;
}

// the raw per-section sums (16 doubles: Chain::tsec) of the most recent k3b_expand launch — what the LIGHT timing builds
// (-DEG3D_ONE_SECTION=k: section k and the whole chain only) are read through; tools/section_light.py
extern "C" int eg3d_probe_sections_raw(eg3d_ctx* c, double* sum16, uint32_t* n_chains) {
  if (!c || !sum16) return EG3D_ERR_ARG;
  std::vector<ChainOut> co(c->last_nc ? c->last_nc : 1);
  if (c->last_nc) HIP_TRY(hipMemcpy(co.data(), c->b_couts.p, sizeof(ChainOut) * c->last_nc, hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; k++) sum16[k] = 0;
  for (uint32_t j = 0; j < c->last_nc; j++)
    for (int k = 0; k < 16; k++) sum16[k] += (double)co[j].tsec[k];
  if (n_chains) *n_chains = c->last_nc;
  return EG3D_OK;
}

#ifndef EG3D_ONE_SECTION
// Gauss-Newton diagnostics of the expand kernel since the last reset (eg3d_dev_coopgn.h g_gn_dbg)
namespace eg3d { int gn_dbg_read(unsigned long long* out, int reset); }
extern "C" int eg3d_probe_gn(unsigned long long* out128, int reset) { return eg3d::gn_dbg_read(out128, reset); }
#endif

extern "C" int eg3d_probe_hyp_sections(eg3d_ctx* c, double* sum, double* slowest, uint32_t* counts) {
  if (!c || !sum || !slowest || !counts) return EG3D_ERR_ARG;
  const uint32_t n = c->last_nhyp;
  std::vector<HypResult> r(n ? n : 1);
  if (n) HIP_TRY(hipMemcpy(r.data(), c->b_res.p, sizeof(HypResult) * n, hipMemcpyDeviceToHost));
  for (int k = 0; k < 6; k++) sum[k] = slowest[k] = 0;
  for (int k = 0; k < 5; k++) counts[k] = 0;
  counts[0] = n;
  for (uint32_t h = 0; h < n; h++) {
    if (r[h].status & HYP_TRI) counts[1]++;
    if (r[h].status & HYP_D1) counts[2]++;
    if (r[h].status & HYP_D2) counts[3]++;
    if (r[h].status & HYP_COMPAT) counts[4]++;
    // list lengths (the per-section tick fields were retired): sum[0..1] = total n1, n2;
    // slowest[0..1] = longest n1, n2; sum[2] = hypotheses whose lists total >= 32 points
    sum[0] += r[h].n1;
    sum[1] += r[h].n2;
    if (r[h].n1 > slowest[0]) slowest[0] = r[h].n1;
    if (r[h].n2 > slowest[1]) slowest[1] = r[h].n2;
    if (r[h].n1 + r[h].n2 >= 32) sum[2] += 1;
  }
  return EG3D_OK;
}

#endif
#if defined(EG3D_WITH_K3C_ENGINE) && (defined(EG3D_SECTION_TIMING) || defined(EG3D_K3C_TIMING))
// ... and of the lane-per-chain engine (eg3d_k3c_engine.h g_k3c_dbg)
namespace eg3d { int k3c_dbg_read(unsigned long long* out, int reset); }
extern "C" int eg3d_probe_k3c(unsigned long long* out128, int reset) { return eg3d::k3c_dbg_read(out128, reset); }
#endif
  // EG3D_SECTION_TIMING
