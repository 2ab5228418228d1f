"""Randomised, mutated small scenes shared by the device fuzz test (tests/test_gpu_fuzz.py) and its CPU twin
(tests/test_cpu_fuzz.py: the device code compiled for the host against the oracle)."""
import numpy as np

from edgegraph3d_amd import host


def draw(case, salt=0):
    """salt != 0: another random stream for the same case shape (tools/fuzz_campaign.py)."""
    rng = np.random.default_rng(0xE63D + case + 100003 * salt)
    cfg = host.default_config(1)
    cfg.n_views = int(rng.integers(3, 28))
    cfg.n_seeds = int(rng.integers(40, 160))
    cfg.n_curves = int(rng.integers(8, 50))
    cfg.rng_seed = int(rng.integers(1, 2**62))
    cfg.max_track = int(rng.integers(3, 17))
    cfg.obs_noise_px = float(rng.choice([0.0, 0.2, 0.4, 1.0, 2.5]))
    cfg.vtx_noise_px = float(rng.choice([0.0, 0.15, 0.5]))
    cfg.invalid_frac = float(rng.choice([0.0, 0.01, 0.15]))
    cfg.seed_offset_px = float(rng.choice([0.0, 3.0, 6.0, 12.0]))
    if case >= 100:        # many views: lazy pre-solves (>= 64 views), long observation lists (chunked lane groups)
        cfg.n_views = [70, 110, 200][case - 100]
        cfg.n_seeds = 24
        cfg.max_track = 40
    if case % 4 == 1:      # a small image: more seeds near the border, coarser grids
        cfg.width, cfg.height = 640, 480
        cfg.focal, cfg.ppx, cfg.ppy = 1150.0, 330.0, 245.0
    s = host.Synth(cfg)
    sc = s.scene_np()
    off, view, xy = s.seeds_np()
    V = sc["n_views"]
    # ---- scene mutations
    Fv = sc["F_valid"].copy()
    kill = rng.random(Fv.shape) < rng.choice([0.0, 0.05, 0.3])
    Fv[kill] = 0
    sc["F_valid"] = Fv
    NP = len(sc["pl_start"])
    end = sc["pl_end"].copy()
    loops = rng.random(NP) < 0.03
    end[loops] = sc["pl_start"][loops]
    sc["pl_end"] = end
    valid = sc["pl_valid"].copy()
    valid[rng.random(NP) < 0.02] = 0          # invalid, vertices still present (include/eg3d.h allows it)
    sc["pl_valid"] = valid
    # ---- seed mutations
    view, xy = view.copy(), xy.copy()
    for p in range(len(off) - 1):
        a, b = int(off[p]), int(off[p + 1])
        r = rng.random()
        if r < 0.06 and b - a >= 2:
            view[a + 1] = view[a]              # repeated view id: the later observation is the one used
        elif r < 0.10:
            xy[a] = [0.0, float(rng.uniform(0, sc["height"]))]          # on the border
        elif r < 0.13:
            xy[a] = [float(sc["width"]) + 3.0, -2.0]                    # outside
        elif r < 0.15:
            xy[a] = xy[a] + np.float32(0.5) * np.float32(30.0)          # exactly on a 30 px cell boundary region
    keep = np.ones(len(view), bool)
    for p in range(0, len(off) - 1, 7):        # every 7th track cut to two observations (needs >= 3 views: no output)
        a, b = int(off[p]), int(off[p + 1])
        keep[a + 2:b] = False
    new_off = np.zeros_like(off)
    new_off[1:] = np.cumsum([keep[off[p]:off[p + 1]].sum() for p in range(len(off) - 1)])
    seeds = host.SeedsArrays(new_off, view[keep], xy[keep])
    return s, host.SceneArrays(sc), seeds


HOSTILE_KINDS = ["seed_nan", "seed_inf", "seed_huge", "vtx_dup", "cam_zero", "F_nan", "vtx_nan", "vtx_huge"]


def hostile(case, kinds):
    """Hostile numeric inputs: NaN / inf / huge coordinates in seed observations and polyline vertices, zero-length
    segments, a camera of zeros, a NaN fundamental matrix."""
    rng = np.random.default_rng(9000 + case)
    cfg = host.default_config(1)
    cfg.n_views = int(rng.integers(4, 12))
    cfg.n_seeds = int(rng.integers(60, 140))
    cfg.rng_seed = int(rng.integers(1, 2**62))
    s = host.Synth(cfg)
    sc = s.scene_np()
    off, view, xy = s.seeds_np()
    xy = xy.copy()
    vt = sc["vtx_xy"].copy()
    m = len(xy)
    if "seed_nan" in kinds:
        xy[rng.integers(0, m, 6)] = np.nan
    if "seed_inf" in kinds:
        xy[rng.integers(0, m, 4), 0] = np.inf
        xy[rng.integers(0, m, 4), 1] = -np.inf
    if "seed_huge" in kinds:
        xy[rng.integers(0, m, 6)] = [3e9, -7e12]
    if "vtx_dup" in kinds:
        k = rng.integers(1, len(vt) - 1, 200)
        vt[k] = vt[k - 1]
    if "vtx_nan" in kinds:
        vt[rng.integers(0, len(vt), 20)] = np.nan
    if "vtx_huge" in kinds:
        vt[rng.integers(0, len(vt), 20)] = [1e20, -1e20]
    sc["vtx_xy"] = vt
    if "cam_zero" in kinds:
        P = sc["cam_P"].copy()
        P[int(rng.integers(0, sc["n_views"]))] = 0
        sc["cam_P"] = P
    if "F_nan" in kinds:
        F = sc["F"].copy()
        F[int(rng.integers(0, sc["n_views"])), int(rng.integers(0, sc["n_views"]))] = np.nan
        sc["F"] = F
    return s, host.SceneArrays(sc), host.SeedsArrays(off, view, xy)
