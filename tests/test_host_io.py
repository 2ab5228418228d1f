"""Host-side rows either side of the path: OpenMVG JSON round trip (a-IO), dedup + observation
filter (N3) against the oracle, and the C++ reference-surface shim compiles."""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import host
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sfm_lib():
    L = host.lib()
    L.eg3d_sfm_read_json.restype = C.c_void_p
    L.eg3d_sfm_read_json.argtypes = [C.c_char_p]
    L.eg3d_sfm_write_json.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.eg3d_sfm_destroy.argtypes = [C.c_void_p]
    L.eg3d_sfm_n_views.argtypes = [C.c_void_p]
    L.eg3d_sfm_n_points.argtypes = [C.c_void_p]
    L.eg3d_sfm_n_points.restype = C.c_uint64
    L.eg3d_sfm_cam_P.argtypes = [C.c_void_p]
    L.eg3d_sfm_cam_P.restype = D.f32p
    L.eg3d_sfm_seeds.argtypes = [C.c_void_p, C.POINTER(D.Seeds)]
    L.eg3d_sfm_points.argtypes = [C.c_void_p]
    L.eg3d_sfm_points.restype = D.f32p
    L.eg3d_sfm_add_edgepoints.argtypes = [C.c_void_p, C.POINTER(D.EdgePoints), D.u8p]
    return L


def _openmvg_doc(base=10):
    R = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    views, extr = [], []
    for i in range(3):
        views.append({"key": i, "value": {"polymorphic_id": 1073741824, "ptr_wrapper": {"id": 2147483649 + i, "data": {
            "local_path": "/", "filename": "%04d.png" % i, "width": 1600, "height": 1200, "id_view": i,
            "id_intrinsic": 0, "id_pose": base + i}}}})
        extr.append({"key": base + i, "value": {"rotation": R, "center": [100.0 * i, 0.5, -2.25]}})
    structure = [{"key": 7, "value": {"X": [1.5, -2.0, 800.0], "observations": [
        {"key": base, "value": {"id_feat": 3, "x": [801.25, 597.5]}},
        {"key": base + 2, "value": {"id_feat": 9, "x": [551.0, 597.5]}},
        {"key": base + 1, "value": {"id_feat": 1, "x": [676.125, 597.5]}}]}}]
    return {"sfm_data_version": "0.3", "root_path": "/data/imgs", "views": views,
            "intrinsics": [{"key": 0, "value": {"polymorphic_id": 2147483649, "polymorphic_name": "pinhole", "ptr_wrapper": {
                "id": 2147483660, "data": {"width": 1600, "height": 1200, "focal_length": 2890.5, "principal_point": [823.0, 619.0]}}}}],
            "extrinsics": extr, "structure": structure, "control_points": []}


def test_openmvg_json_round_trip():
    L = _sfm_lib()
    with tempfile.TemporaryDirectory() as d:
        p_in, p_out = os.path.join(d, "in.json"), os.path.join(d, "out.json")
        json.dump(_openmvg_doc(), open(p_in, "w"))
        h = L.eg3d_sfm_read_json(p_in.encode())
        assert h
        assert L.eg3d_sfm_n_views(h) == 3 and L.eg3d_sfm_n_points(h) == 1
        P = D.as_np(L.eg3d_sfm_cam_P(h), 48, np.float32).reshape(3, 16)
        # P = K4 [R | -R C]: view 1 has C = (100, 0.5, -2.25)
        assert np.allclose(P[1, :12].reshape(3, 4), [[2890.5, 0, 823, -2890.5 * 100 + 823 * 2.25],
                                                     [0, 2890.5, 619, -2890.5 * 0.5 + 619 * 2.25], [0, 0, 1, 2.25]])
        assert np.all(P[:, 12:] == 0)                      # last row zero (Q6)
        s = D.Seeds()
        L.eg3d_sfm_seeds(h, C.byref(s))
        views = D.as_np(s.trk_view, 3, np.int32)
        assert list(views) == [0, 2, 1]                    # pose keys 10,12,11 -> positions in `extrinsics`
        assert L.eg3d_sfm_write_json(h, p_in.encode(), p_out.encode()) == 0
        out = json.load(open(p_out))
        assert list(out.keys()) == ["sfm_data_version", "root_path", "views", "intrinsics", "extrinsics", "structure", "control_points"]
        assert out["views"] == _openmvg_doc()["views"] and out["intrinsics"] == _openmvg_doc()["intrinsics"]
        assert [e["key"] for e in out["extrinsics"]] == [0, 1, 2]     # rewritten with keys 0..V-1
        assert out["extrinsics"][2]["value"]["center"] == [200.0, 0.5, -2.25]
        st = out["structure"][0]
        assert st["key"] == 0 and st["value"]["X"] == [1.5, -2.0, 800.0]
        assert [o["key"] for o in st["value"]["observations"]] == [0, 2, 1]
        assert all(o["value"]["id_feat"] == 0 for o in st["value"]["observations"])
        assert st["value"]["observations"][2]["value"]["x"] == [676.125, 597.5]
        L.eg3d_sfm_destroy(h)
        # with natural pose keys (id_pose == position, the usual OpenMVG layout) the written file reads back identically
        json.dump(_openmvg_doc(0), open(p_in, "w"))
        h = L.eg3d_sfm_read_json(p_in.encode())
        P = D.as_np(L.eg3d_sfm_cam_P(h), 48, np.float32)
        assert L.eg3d_sfm_write_json(h, p_in.encode(), p_out.encode()) == 0
        h2 = L.eg3d_sfm_read_json(p_out.encode())
        P2 = D.as_np(L.eg3d_sfm_cam_P(h2), 48, np.float32)
        assert np.array_equal(P, P2) and P.any()
        assert json.load(open(p_out))["structure"] == json.load(open(p_out))["structure"]
        L.eg3d_sfm_destroy(h)
        L.eg3d_sfm_destroy(h2)


def test_dedup_and_observation_filter_match_oracle():
    s = host.Synth(1)
    o = ob.Oracle(s.scene)
    e, st = D.EdgePoints(), ob.Stats()
    assert ob.lib().orc_match_refpoints(o._h, s.seeds, 0, s.n_seeds, 1, C.byref(e), C.byref(st)) == 0
    n = int(e.n_points)
    keep_o, keep_h = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    assert ob.lib().orc_filter_close_2d(o._h, C.byref(e), D.np_ptr(keep_o, C.c_uint8)) == 0
    sc = s.scene.contents
    assert host.lib().eg3d_host_filter_close_2d(sc.n_views, sc.width, sc.height, C.byref(e), D.np_ptr(keep_h, C.c_uint8)) == 0
    assert np.array_equal(keep_o, keep_h)
    assert 0 < keep_h.sum() < n                     # the greedy 3 px dedup removes some, keeps some
    assert keep_h[0] == 1                           # the first point is always new
    off = D.as_np(e.obs_off, n + 1, np.uint32)
    a, b = np.ones(n, np.uint8), np.ones(n, np.uint8)
    ta = ob.lib().orc_observation_filter(sc.n_views, D.np_ptr(off, C.c_uint32), n, n // 2, -1, D.np_ptr(a, C.c_uint8))
    tb = host.lib().eg3d_host_observation_filter(sc.n_views, D.np_ptr(off, C.c_uint32), n, n // 2, -1, D.np_ptr(b, C.c_uint8))
    assert ta == tb == 3 and np.array_equal(a, b)
    assert a[: n // 2].all()                        # points before first_edgepoint are never dropped
    ob.lib().orc_free_edgepoints(C.byref(e))


def test_reference_surface_shim_compiles():
    """include/eg3d_refapi.hpp (the reference's call surface over the C ABI) is valid C++17."""
    src = '#include "eg3d_refapi.hpp"\nint main(){ eg3d_ref::SfMData s; eg3d_ref::FundamentalMatrices F; ' \
          'std::vector<eg3d_ref::PolyLineGraph2D> g; (void)sizeof(eg3d_ref::PLGEdgeManager); return s.numPoints_; }\n'
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.cpp")
        open(p, "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), p])


def test_polyline_graph_container_round_trip():
    """EG3DPLG1 container (edgegraph3d_amd/host/plg_file.cpp): the polyline graphs of all views
    written and read back are identical, including invalid / empty polylines and node ids."""
    L = host.lib()
    L.eg3d_plg_write.argtypes = [C.c_char_p, C.POINTER(D.Scene)]
    L.eg3d_plg_read.restype = C.c_void_p
    L.eg3d_plg_read.argtypes = [C.c_char_p]
    L.eg3d_plg_scene.restype = C.POINTER(D.Scene)
    L.eg3d_plg_scene.argtypes = [C.c_void_p]
    L.eg3d_plg_destroy.argtypes = [C.c_void_p]
    s = host.Synth(1)
    sc = s.scene_np()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "plgs.bin").encode()
        assert L.eg3d_plg_write(p, s.scene) == 0
        g = L.eg3d_plg_read(p)
        assert g
        r = L.eg3d_plg_scene(g).contents
        V = int(r.n_views)
        assert V == sc["n_views"] and r.width == sc["width"] and r.height == sc["height"]
        vpo = D.as_np(r.view_pl_off, V + 1, np.uint32)
        NP = int(vpo[-1])
        pvo = D.as_np(r.pl_vtx_off, NP + 1, np.uint32)
        assert np.array_equal(vpo, sc["view_pl_off"]) and np.array_equal(pvo, sc["pl_vtx_off"])
        assert np.array_equal(D.as_np(r.pl_start, NP, np.uint32), sc["pl_start"])
        assert np.array_equal(D.as_np(r.pl_end, NP, np.uint32), sc["pl_end"])
        assert np.array_equal(D.as_np(r.pl_valid, NP, np.uint8), sc["pl_valid"])
        assert (sc["pl_valid"] == 0).any()                     # the generator emits some invalid polylines
        got = D.as_np(r.vtx_xy, 2 * int(pvo[-1]), np.float32).reshape(-1, 2)
        assert np.array_equal(got.view(np.uint32), sc["vtx_xy"].view(np.uint32))
        L.eg3d_plg_destroy(g)
        open(os.path.join(d, "bad.bin"), "wb").write(b"NOTAPLG!" + b"\0" * 12)
        assert not L.eg3d_plg_read(os.path.join(d, "bad.bin").encode())


def test_openmvg_reader_rejects_malformed_files_instead_of_crashing():
    """A variant / truncated OpenMVG file (non-pinhole intrinsic without focal_length, short
    rotation, missing observation coordinates, absurd nesting) makes eg3d_sfm_read_json return NULL."""
    L = _sfm_lib()
    import copy

    def reads(doc, raw=None):
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "x.json")
            open(p, "w").write(raw if raw is not None else json.dumps(doc))
            h = L.eg3d_sfm_read_json(p.encode())
            if h:
                L.eg3d_sfm_destroy(h)
            return bool(h)

    good = _openmvg_doc()
    assert reads(good)
    bad = copy.deepcopy(good)
    del bad["intrinsics"][0]["value"]["ptr_wrapper"]["data"]["focal_length"]
    assert not reads(bad)
    bad = copy.deepcopy(good)
    bad["intrinsics"][0]["value"]["ptr_wrapper"]["data"]["principal_point"] = [823.0]
    assert not reads(bad)
    bad = copy.deepcopy(good)
    bad["extrinsics"][1]["value"]["rotation"] = [[1, 0, 0], [0, 1, 0]]
    assert not reads(bad)
    bad = copy.deepcopy(good)
    del bad["structure"][0]["value"]["observations"][0]["value"]["x"]
    assert not reads(bad)
    bad = copy.deepcopy(good)
    bad["views"][0]["value"]["ptr_wrapper"]["data"]["width"] = "wide"
    assert not reads(bad)
    assert not reads(None, raw="[" * 5000 + "]" * 5000)        # nesting limit, no stack overflow
    assert not reads(None, raw='{"views": [')                  # truncated
    # a raw control character inside a string is not JSON (the reference's rapidjson reader refuses it, reader.h:869);
    # the same string with the character escaped reads
    txt = json.dumps(good)
    assert '"root_path"' in txt
    assert not reads(None, raw=txt.replace('"root_path": "', '"root_path": "\x01', 1))
    assert reads(None, raw=txt.replace('"root_path": "', '"root_path": "\\u0001', 1))


def test_analytic_fundamental_matrices_satisfy_the_epipolar_constraint():
    """N4: eg3d_sfm_analytic_F from the cameras of an SfM scene. F[i][j] maps a point of view i to
    its line in view j (the convention of computeCorrespondEpilineSinglePoint,
    geometric_utilities.cpp:824-843): the projection of any 3-D point into j lies on the line of its
    projection into i. Checked in float64 from the float camera matrices, through the oracle's
    epiline primitive, and against the synthetic generator's own F (same rig => same matrices)."""
    L = _sfm_lib()
    L.eg3d_sfm_create.restype = C.c_void_p
    L.eg3d_sfm_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.eg3d_sfm_set_camera.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, D.f32p, D.f32p, C.c_char_p]
    L.eg3d_sfm_analytic_F.argtypes = [C.c_void_p, D.f64p, D.u8p]
    L.eg3d_synth_camera.argtypes = [C.c_void_p, C.c_int, D.f32p, D.f32p, D.f32p, D.f32p, D.f32p]
    s = host.Synth(1)
    sc = s.scene_np()
    V = sc["n_views"]
    h = L.eg3d_sfm_create(V, sc["width"], sc["height"])
    for v in range(V):
        f, px, py = C.c_float(), C.c_float(), C.c_float()
        R, Cc = np.zeros(9, np.float32), np.zeros(3, np.float32)
        assert L.eg3d_synth_camera(s._h, v, C.byref(f), C.byref(px), C.byref(py), D.np_ptr(R, C.c_float), D.np_ptr(Cc, C.c_float)) == 0
        assert L.eg3d_sfm_set_camera(h, v, f, px, py, D.np_ptr(R, C.c_float), D.np_ptr(Cc, C.c_float), b"v.png") == 0
    P = D.as_np(L.eg3d_sfm_cam_P(h), V * 16, np.float32).reshape(V, 4, 4).astype(np.float64)
    assert np.array_equal(P.astype(np.float32).reshape(V, 16), sc["cam_P"])      # the same rig as the generator's scene
    F = np.zeros((V, V, 9), np.float64)
    Fv = np.zeros((V, V), np.uint8)
    assert L.eg3d_sfm_analytic_F(h, D.np_ptr(F, C.c_double), D.np_ptr(Fv, C.c_uint8)) == 0
    assert np.array_equal(Fv, 1 - np.eye(V, dtype=np.uint8))
    assert np.array_equal(F, sc["F"]) and np.array_equal(Fv, sc["F_valid"])       # generator F == analytic F of its cameras
    rng = np.random.default_rng(11)
    line = np.zeros(3, np.float32)
    worst = 0.0
    for _ in range(200):
        i, j = rng.choice(V, 2, replace=False)
        X = np.append(rng.uniform(-180, 180, 3), 1.0)
        xi, xj = P[i][:3] @ X, P[j][:3] @ X
        xi, xj = xi[:2] / xi[2], xj[:2] / xj[2]
        l = F[i, j].reshape(3, 3) @ np.append(xi, 1.0)
        d = abs(l @ np.append(xj, 1.0)) / np.hypot(l[0], l[1])
        worst = max(worst, d)
        # the oracle's epiline (f64 accumulate, normalised, rounded to f32) of the float-rounded point
        Fij = np.ascontiguousarray(F[i, j])
        assert ob.lib().orc_epiline(D.np_ptr(Fij, C.c_double), np.float32(xi[0]), np.float32(xi[1]), D.np_ptr(line, C.c_float)) == 1
        assert abs(float(line[0]) ** 2 + float(line[1]) ** 2 - 1) < 1e-5
        assert abs(line.astype(np.float64) @ np.append(xj, 1.0)) < 0.05           # px; f32 line coefficients at ~1000 px
    assert worst < 1e-3, worst                                                    # px, float camera matrices
    L.eg3d_sfm_destroy(h)


def test_multithreaded_host_copy_covers_every_byte(tmp_path):
    """edgegraph3d_amd/csrc/eg3d_host_copy.h (the copy of a cloud from pinned staging into the caller's arrays): every
    byte of every size arrives and nothing beyond is touched. Regression for a truncating division that left up to
    threads-1 trailing bytes uncopied (found by the whole-batch C4 parity run)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "hostcopy_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(root, "edgegraph3d_amd", "csrc"),
                           os.path.join(root, "tests", "hostcopy", "hostcopy_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HOSTCOPY-OK" in out.stdout, out.stdout + out.stderr


def test_compare_sfm_json_measures_structural_agreement_of_two_runs(tmp_path):
    """tools/compare_sfm_json.py (the recipe for a user who holds a real reference run: README): two
    before_filtering.json files are compared point by point through their observation lists. Built here from one
    document and a hand-made variant: one edge-point moved by 1e-6 relative, one moved by 1e-3, one with an extra
    observation (same start, grew differently), one missing, one new."""
    import copy
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import compare_sfm_json as cmp
    base = _openmvg_doc(0)

    def pt(key, X, obs):
        return {"key": key, "value": {"X": X, "observations": [{"key": v, "value": {"id_feat": 0, "x": xy}} for v, xy in obs]}}
    edge = [pt(100 + i, [10.0 + i, 20.0, 500.0 + i], [(0, [100.5 + i, 200.25]), (1, [300.0 + i, 210.0]), (2, [50.0, 60.0 + i])])
            for i in range(6)]
    a = copy.deepcopy(base)
    a["structure"] += copy.deepcopy(edge)
    b = copy.deepcopy(base)
    eb = copy.deepcopy(edge)
    eb[1]["value"]["X"][2] *= 1.0 + 1e-6          # within tolerance
    eb[2]["value"]["X"][2] *= 1.0 + 1e-3          # structurally identical, outside tolerance
    eb[3]["value"]["observations"].append({"key": 1, "value": {"id_feat": 0, "x": [7.0, 8.0]}})   # grew differently
    del eb[4]                                     # only in the reference
    eb.append(pt(300, [1.0, 2.0, 3.0], [(2, [9.0, 9.5]), (0, [1.0, 1.5])]))                        # only in ours
    b["structure"] += eb
    pa, pb, pin = str(tmp_path / "a.json"), str(tmp_path / "b.json"), str(tmp_path / "in.json")
    json.dump(a, open(pa, "w"))
    json.dump(b, open(pb, "w"))
    json.dump(base, open(pin, "w"))
    rep = cmp.compare(cmp.read_cloud(pa), cmp.read_cloud(pb), len(cmp.read_cloud(pin)[0]))
    assert rep["input_points"] == 1 and rep["edge_points_ref"] == 6 and rep["edge_points_got"] == 6
    assert rep["structurally_identical"] == 4 and rep["X_bit_equal"] == 2 and rep["X_within_tol"] == 3
    assert 5e-4 < rep["max_rel_dX"] < 2e-3
    assert rep["same_first_observation_different_list"] == 1 and rep["only_in_ref"] == 1 and rep["only_in_got"] == 2
    # without the input file the SfM points are taken to be the common prefix: leading edge-points that agree count as input
    auto = cmp.compare(cmp.read_cloud(pa), cmp.read_cloud(pb))
    assert auto["input_points"] == 2 and auto["structurally_identical"] == 3
    same = cmp.compare(cmp.read_cloud(pa), cmp.read_cloud(pa))
    assert same["input_points"] == 7  # identical files: everything is common prefix
