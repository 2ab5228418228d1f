"""CPU-side (-m "not gpu") coverage of the product's host logic and of the kernels' per-lane
bodies (through the test-only host simulation) against the oracle."""
import ctypes as C

import numpy as np
import pytest

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import api, host
from oracle import binding as ob
from parity_util import compare_edgepoints
import hostsim_binding as hs


@pytest.mark.parametrize("cfg", [0, 1])
def test_host_grid_builder_matches_oracle(cfg):
    s = host.Synth(cfg)
    o = ob.Oracle(s.scene)
    for v in range(s.n_views):
        for which, cell in ((0, 30.0), (1, 4.0)):
            ncols, nrows, off, ids, dropped = host.build_grid(s.scene, v, cell)
            r = o.grid(v, which)
            assert (ncols, nrows) == (r[0], r[1])
            assert dropped == 0
            assert np.array_equal(off, r[2]) and np.array_equal(ids, r[3])
            assert ids.size > 0


@pytest.mark.parametrize("slot_step", [False, True, 2, 3])  # 2 = the chain state machine of the engine kernel (eg3d_chain_sm.h) with a sequential server, 3 = with the streaming forms of its lane-private loops (what the kernel runs)
@pytest.mark.parametrize("cfg", [0, 1])
def test_kernel_bodies_hostsim_vs_oracle(cfg, slot_step):
    s = host.Synth(cfg)
    o = ob.Oracle(s.scene)
    ref = o.match(s.seeds, 0, s.n_seeds, 1)
    cand = o.candidates_raw(s.seeds, 0, s.n_seeds)
    got = hs.match(s.scene, s.seeds, 0, s.n_seeds, cand, slot_step=slot_step)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"], rep["msgs"]
    assert rep["bitexact_X"] and rep["bitexact_xy"]
    assert got["n_chains"] == ref["stats"]["n_chains"] and got["n_tasks"] == ref["stats"]["n_tasks"]
    assert (got["flags"] & 7) == 0
    assert ref["n_points"] > 100


def test_oracle_threads_do_not_change_output():
    s = host.Synth(1)
    o = ob.Oracle(s.scene)
    a = o.match(s.seeds, 0, s.n_seeds, 1)
    b = o.match(s.seeds, 0, s.n_seeds, 4)
    rep = compare_edgepoints(a, b)
    assert rep["ok"] and rep["bitexact_X"]


def subprocess_nm(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return [l.split()[-1] for l in out.splitlines() if " T " in l]


def test_abi_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports every symbol include/eg3d.h declares."""
    import os
    import re
    if not os.path.exists(api.lib_path()):
        from edgegraph3d_amd import build
        build.build_hip()
    L = C.CDLL(api.lib_path())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "eg3d.h")).read()
    declared = set(re.findall(r"\b(eg3d_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    # no test hooks in the product library: the probes live in tests/probe/libeg3d_probe.so
    exported = subprocess_nm(api.lib_path())
    assert not [n for n in exported if "probe" in n or "internal" in n], exported
    H = host.lib()
    hhdr = open(os.path.join(root, "include", "eg3d_host.h")).read()
    host_syms = set(re.findall(r"\b(eg3d_(?:synth|host|sfm|plg)_[a-z_0-9A-Z]+)\s*\(", hhdr))
    assert len(host_syms) > 24
    for name in host_syms:
        assert hasattr(H, name), name
    # the RCCL gather library (include/eg3d_rccl.h): symbol table only — loading it would pull
    # /opt/rocm's librccl into a process that may already hold the torch wheel's ROCm runtime
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(os.path.dirname(api.lib_path()), "libeg3d_rccl.so")],
                          capture_output=True, text=True, check=True).stdout
    rhdr = open(os.path.join(root, "include", "eg3d_rccl.h")).read()
    rsyms = set(re.findall(r"\b(eg3d_(?:gather|allgather|comm|concat)_[a-z_0-9]+)\s*\(", rhdr))
    assert len(rsyms) == 10  # gather create / destroy / set_mode / set_chunk_bytes, allgather, concat, comm unique_id / init / query / destroy
    for name in rsyms:
        assert re.search(r"\b%s\b" % name, syms), name


def test_no_gpu_means_loud_failure():
    """Without a device the product refuses to run (no CPU fallback)."""
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    s = host.Synth(0)
    with pytest.raises(api.Eg3dError):
        api.Context(s.scene)


def test_inconsistent_scenes_are_refused_before_any_device_is_touched():
    """eg3d_create validates the scene on the host first (the kernels index device arrays with its offsets): a
    descending offset array, a non-zero first offset, a null array, an empty image, a NaN / 1e20 vertex all give
    EG3D_ERR_ARG — here, without a GPU, where a consistent scene gives EG3D_ERR_NODEVICE."""
    import ctypes as C
    from edgegraph3d_amd import api
    L = api.lib()
    L.eg3d_last_error.restype = C.c_char_p
    s = host.Synth(0)
    base = s.scene_np()

    def create(mut):
        sc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
        mut(sc)
        sa = host.SceneArrays(sc)
        h = C.c_void_p()
        rc = L.eg3d_create(C.byref(sa.c), 0, C.byref(h))
        if rc == 0:
            L.eg3d_destroy(h)
        return rc, (L.eg3d_last_error() or b"").decode()

    rc, msg = create(lambda sc: None)
    assert rc in (0, -4), (rc, msg)          # fine scene: created on a GPU box, "no device" here

    def desc_vpo(sc):
        sc["view_pl_off"][1] = sc["view_pl_off"][2] + 1

    def desc_pvo(sc):
        sc["pl_vtx_off"][3] = sc["pl_vtx_off"][5] + 7

    def first_vpo(sc):
        sc["view_pl_off"][0] = 1

    def nan_vtx(sc):
        p = int(np.nonzero(sc["pl_valid"])[0][0])
        sc["vtx_xy"][sc["pl_vtx_off"][p]] = np.nan

    def huge_vtx(sc):
        p = int(np.nonzero(sc["pl_valid"])[0][0])
        sc["vtx_xy"][sc["pl_vtx_off"][p]] = 1e20

    def no_image(sc):
        sc["width"] = 0

    for mut in (desc_vpo, desc_pvo, first_vpo, nan_vtx, huge_vtx, no_image):
        rc, msg = create(mut)
        assert rc == -1, (mut.__name__, rc, msg)
        assert "eg3d_create" in msg


def test_walks_with_vertex_loads_in_flight_equal_the_plain_walks():
    """eg3d_dev_geom.h walk_by_line_pf / walk_by_distance_pf (what the K3a engine walks with: a window of five vertices
    loaded ahead of the segment under test) against walk_by_line / walk_by_distance: status, segment and coordinate
    bits, from every kind of start (first / last segment, on a vertex), towards either end and towards a node that is
    neither (Q15), lines that cross, miss, graze and run quasi-parallel, bounded and unbounded, polylines of 2..300
    vertices with repeated vertices."""
    import ctypes as C
    import hostsim_binding as hs
    L = hs.lib()
    rng = np.random.default_rng(23)
    total = 0
    for case in range(80):
        n = int(rng.integers(2, 300)) if case % 8 else int(rng.integers(2, 7))
        kind = case % 4
        if kind == 0:
            tt = np.linspace(0, rng.uniform(1, 12), n)
            v = np.stack([np.cos(tt) * tt * 20, np.sin(tt) * tt * 20], 1)
        elif kind == 1:
            v = np.cumsum(rng.normal(0, 3, (n, 2)), 0)
            v[rng.integers(0, n, n // 5)] = v[0]
        elif kind == 2:
            v = np.cumsum(np.stack([rng.integers(0, 3, n), rng.integers(0, 3, n)], 1), 0).astype(np.float64)
        else:
            v = np.stack([np.linspace(0, 4 * n, n), rng.normal(0, 0.5, n)], 1)   # nearly straight: quasi-parallel lines
        v = np.ascontiguousarray(v, np.float32)
        nq = 300
        seg = rng.integers(0, max(1, n - 1), nq)
        seg[:3] = [0, max(0, n - 2), max(0, n - 2)]
        tpar = rng.uniform(0, 1, nq).astype(np.float32)
        tpar[:40] = 0.0                                                           # exactly on the segment's first vertex
        a, b = v[seg], v[np.minimum(seg + 1, n - 1)]
        pts = a + tpar[:, None] * (b - a)
        ang = rng.uniform(0, np.pi, nq)
        la, lb = np.cos(ang), np.sin(ang)
        through = v[rng.integers(0, n, nq)] + rng.normal(0, 6, (nq, 2))           # lines through the neighbourhood
        lc = -(la * through[:, 0] + lb * through[:, 1])
        if kind == 3:
            la[:150], lb[:150] = rng.normal(0, 0.02, 150), 1.0                    # quasi-parallel to the stroke
            lc[:150] = rng.uniform(-8, 8, 150)
        dist = rng.choice([10.0, 0.5, 3.0, 50.0, 1e6], nq)
        q = np.ascontiguousarray(np.stack([seg, pts[:, 0], pts[:, 1], la, lb, lc, dist, rng.integers(0, 2, nq)], 1), np.float32)
        for start, end in ((1, 2), (7, 7)):                                       # (7, 7): a closed polyline, both ends one node
            bad = L.hostsim_walk_pf_mismatches(D.np_ptr(v, C.c_float), n, start, end, D.np_ptr(q, C.c_float), nq)
            assert bad == 0, (case, kind, n, start, end, bad)
            total += 6 * nq
    assert total > 200000


def test_closest_point_scan_with_box_pretest_equals_the_plain_scan():
    """eg3d_dev_geom.h polyline_closest_pruned (the expand stage's per-view closest-point scan skips blocks of 8 segments
    whose bounding box is farther than the best distance so far) must return the plain scan's result bit for bit: first
    of equal distances, NaN / infinite queries, zero-length and repeated segments, sub-ranges that cut blocks, huge
    coordinates (where the rounding slack of the bound matters)."""
    import ctypes as C
    import hostsim_binding as hs
    L = hs.lib()
    rng = np.random.default_rng(11)
    total = 0
    for case in range(60):
        n = int(rng.integers(2, 400))
        scale = float(10.0 ** rng.integers(0, 8))          # up to the +-1e7 px the library accepts
        kind = case % 6
        if kind == 0:      # smooth curve
            tt = np.linspace(0, rng.uniform(1, 12), n)
            v = np.stack([np.cos(tt) * tt * 20, np.sin(tt) * tt * 20], 1)
        elif kind == 1:    # random walk with repeated vertices (zero-length segments)
            v = np.cumsum(rng.normal(0, 3, (n, 2)), 0)
            v[rng.integers(0, n, n // 5)] = v[0]
        elif kind == 2:    # two identical passes over the same stroke: exact ties between far-apart segments
            h = np.cumsum(rng.normal(0, 2, ((n + 1) // 2, 2)), 0)
            v = np.concatenate([h, h[::-1]])[:n]
        elif kind == 3:    # integer grid coordinates: many exactly equal distances
            v = rng.integers(-30, 30, (n, 2)).astype(np.float64)
        elif kind == 4:    # axis-aligned staircase
            v = np.cumsum(np.stack([rng.integers(0, 3, n), rng.integers(0, 3, n)], 1), 0).astype(np.float64)
        else:
            v = rng.uniform(-500, 500, (n, 2))
        v = np.ascontiguousarray(v * (scale if kind in (0, 5) else 1.0), np.float32)
        n = len(v)
        q = np.concatenate([v[rng.integers(0, n, 150)] + rng.normal(0, 4, (150, 2)).astype(np.float32),
                            (rng.uniform(-2, 2, (60, 2)) * np.abs(v).max()).astype(np.float32),
                            v[rng.integers(0, n, 40)],                                    # exactly on vertices
                            np.array([[np.nan, 0], [0, np.nan], [np.inf, 1], [-np.inf, -np.inf], [1e30, -1e30], [0, 0]], np.float32)])
        q = np.ascontiguousarray(q, np.float32)
        nseg = n - 1
        ranges = [(0, nseg)] + [tuple(sorted(rng.integers(0, nseg + 1, 2))) for _ in range(4)]
        for s0, s1 in ranges:
            bad = L.hostsim_closest_pruned_mismatches(D.np_ptr(v, C.c_float), n, D.np_ptr(q, C.c_float), len(q), int(s0), int(s1))
            assert bad == 0, (case, kind, n, s0, s1, bad)
            total += len(q)
    assert total > 50000
