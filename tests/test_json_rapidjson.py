"""a-IO against the reference's OWN JSON library. The reference reads its OpenMVG file with rapidjson
(external/manifoldReconstructor/src/OpenMvgParser.cpp:39-301) and writes it with rapidjson's PrettyWriter
(src/edgegraph3d/io/output/output_sfm_data.cpp:186-229); the product has its own reader / writer
(edgegraph3d_amd/host/sfm_json.cpp, json_text.hpp). These tests compile a small driver (tests/rapidjson/rj_driver.cpp,
this repo's code) against the header-only rapidjson the reference tree vendors — in the build container only: they
are SKIPPED where /root/reference does not exist, and nothing of that tree is copied here — and check that
  * the product prints doubles exactly as rapidjson's Writer does (Grisu2 + notation), on 200 000 values,
  * its computed table of Grisu's cached powers equals the library's,
  * a number literal copied through from the input file is re-printed as the reference's reader + writer re-print it,
  * a whole written file is BYTE-IDENTICAL to the document output_sfm_data builds, printed by PrettyWriter,
  * the product's reader indexes a hand-written OpenMVG file (ptr_wrapper / polymorphic_id, pose keys that are not
    positions, a view order that differs from the pose order) exactly as OpenMvgParser.cpp does.
(The reference's OpenMvgParser.cpp / output_sfm_data.cpp themselves cannot be compiled here: their headers pull
CGAL and Eigen — SURVEY F3.)"""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import host

RJ = "/root/reference/external"
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(RJ, "rapidjson", "prettywriter.h")),
                                reason="the reference tree (vendored rapidjson) is not present on this machine")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rj") / "rj_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", RJ, "-o", exe, os.path.join(HERE, "rapidjson", "rj_driver.cpp")])
    return exe


def _lib():
    L = host.lib()
    L.eg3d_host_json_double_text.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.eg3d_host_json_number_text.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.eg3d_host_json_cached_power.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
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
    L.eg3d_sfm_add_point.argtypes = [C.c_void_p, D.f32p, C.c_int, D.i32p, D.f32p]
    return L


def test_cached_powers_of_grisu_equal_the_librarys(driver, tmp_path):
    L = _lib()
    out = str(tmp_path / "pw.txt")
    subprocess.check_call([driver, "powers", out])
    ref = [tuple(int(x) for x in ln.split()) for ln in open(out)]
    assert len(ref) == 87
    for i in range(87):
        f, e = C.c_uint64(), C.c_int()
        assert L.eg3d_host_json_cached_power(i, C.byref(f), C.byref(e)) == 0
        assert (f.value, e.value) == ref[i], i


def test_doubles_print_exactly_as_the_librarys_writer(driver, tmp_path):
    L = _lib()
    rng = np.random.default_rng(1)
    n = 100000
    vals = np.concatenate([
        rng.standard_normal(n) * 10.0 ** rng.uniform(-300, 300, n),
        # floats widened to double: what the structure / extrinsics of the written file hold
        (rng.standard_normal(n).astype(np.float32) * np.float32(10) ** rng.uniform(-38, 38, n).astype(np.float32)).astype(np.float64),
        rng.uniform(0, 1600, n).astype(np.float32).astype(np.float64),                         # pixel coordinates
        np.array([2.0 ** 64, 1e21, 1e22, 1e-7, 1e-6, 1e-5, 5e-324, 1.7976931348623157e308, 3.4028234663852886e38, 0.1,
                  float(np.float32(0.1)), 1.0, -1.0, 0.0, -0.0, 123456792.0, 1e15, 1e16, 1e17, 0.5, 2.5e-7])])
    vals = vals[np.isfinite(vals)]
    vals.tofile(str(tmp_path / "d.bin"))
    subprocess.check_call([driver, "doubles", str(tmp_path / "d.bin"), str(tmp_path / "d.txt")])
    want = open(str(tmp_path / "d.txt")).read().split("\n")
    b = C.create_string_buffer(64)
    for v, w in zip(vals, want):
        assert L.eg3d_host_json_double_text(float(v), b, 64) > 0
        assert b.value.decode() == w, repr(v)


def test_number_literals_are_reprinted_as_reader_plus_writer_do(driver, tmp_path):
    """Numbers of `views` / `intrinsics` / `control_points` are copied from the input file: the reference parses them
    (its vendored reader defaults to full precision) and prints what it parsed. Literals whose magnitude underflows
    the double range are out of contract (the vendored library returns garbage for them)."""
    L = _lib()
    rng = np.random.default_rng(2)
    lits = []
    for i in range(40000):
        k = rng.integers(0, 6)
        if k == 0:
            lits.append(str(int(rng.integers(-2 ** 63, 2 ** 63 - 1))))
        elif k == 1:
            lits.append("%.*e" % (int(rng.integers(0, 20)), rng.standard_normal() * 10.0 ** rng.uniform(-290, 290)))
        elif k == 2:
            lits.append("%.*f" % (int(rng.integers(0, 25)), rng.standard_normal() * 10.0 ** rng.uniform(-5, 18)))
        elif k == 3:
            lits.append(repr(float(np.float32(rng.standard_normal() * 10.0 ** rng.uniform(-10, 10)))))
        elif k == 4:
            lits.append(str(int(rng.integers(1, 2 ** 40))) + "".join(str(int(x)) for x in rng.integers(0, 10, int(rng.integers(0, 30)))))
        else:
            lits.append("%d.%sE%+d" % (rng.integers(-99, 99), "".join(str(int(x)) for x in rng.integers(0, 10, int(rng.integers(1, 25)))),
                                      rng.integers(-280, 280)))
    lits += ["0", "-0", "-0.0", "0.0", "1e0", "1E+2", "4294967295", "4294967296", "-2147483648", "-2147483649",
             "18446744073709551615", "18446744073709551616", "-9223372036854775808", "-9223372036854775809",
             "123456789012345678901234567890", "0.000001", "1e-6", "1e21", "1e22", "2892.3295898437500",
             "1.7976931348623157e308", "1.0e3", "12E-2", "0.1e1", "2147483649", "1073741824"]
    open(str(tmp_path / "l.txt"), "w").write("\n".join(lits) + "\n")
    subprocess.check_call([driver, "literals", str(tmp_path / "l.txt"), str(tmp_path / "lo.txt")])
    want = open(str(tmp_path / "lo.txt")).read().split("\n")
    b = C.create_string_buffer(64)
    for s, w in zip(lits, want):
        n = L.eg3d_host_json_number_text(s.encode(), b, 64)
        assert (b.value.decode() if n >= 0 else "ERROR") == w, s


def _hand_written_openmvg():
    """OpenMVG layout with everything the reference's parser has to get right: pose keys that are not positions
    (and not in view order), two intrinsics, ptr_wrapper / polymorphic_id wrappers, numbers spelled in ways the
    writer will normalise, strings with escapes, a control point list that is only copied through."""
    R = [[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
         [[0.99619469809, 0.0, 0.0871557427], [0.0, 1.0, 0.0], [-0.0871557427, 0.0, 0.99619469809]],
         [[0.98480775301, 0.0, -0.17364817766], [0.0, 1.0, 0.0], [0.17364817766, 0.0, 0.98480775301]]]
    views, extr = [], []
    pose_of_view = [12, 10, 11]                 # view i uses pose key pose_of_view[i]
    for i in range(3):
        views.append({"key": i, "value": {"polymorphic_id": 1073741824, "ptr_wrapper": {"id": 2147483649 + i, "data": {
            "local_path": "sub/", "filename": "%04d.png" % i, "width": 1600, "height": 1200, "id_view": i,
            "id_intrinsic": i % 2, "id_pose": pose_of_view[i]}}}})
    for j, key in enumerate([10, 11, 12]):      # array order of the poses: observation keys map to these positions
        extr.append({"key": key, "value": {"rotation": R[j], "center": [100.0 * j + 0.125, 0.5 - j, -2.25]}})
    structure = [
        {"key": 7, "value": {"X": [1.5, -2.0, 800.0], "observations": [
            {"key": 10, "value": {"id_feat": 3, "x": [801.25, 597.5]}},
            {"key": 12, "value": {"id_feat": 9, "x": [551.0, 597.5]}},
            {"key": 11, "value": {"id_feat": 1, "x": [676.125, 597.5]}}]}},
        {"key": 8, "value": {"X": [0.1, 123456.789, -3.3e-5], "observations": [
            {"key": 11, "value": {"id_feat": 0, "x": [12.000000001, 0.333333333333]}},
            {"key": 10, "value": {"id_feat": 0, "x": [1599.999, 1199.0001]}}]}}]
    doc = {"sfm_data_version": "0.3", "root_path": "/data/imgs \\u00e9t\\u00e9\\/\\n", "views": views,
           "intrinsics": [
               {"key": 0, "value": {"polymorphic_id": 2147483649, "polymorphic_name": "pinhole", "ptr_wrapper": {
                   "id": 2147483660, "data": {"width": 1600, "height": 1200, "focal_length": "@F0@",
                                              "principal_point": ["@P0@", 619.0]}}}},
               {"key": 1, "value": {"polymorphic_id": 2147483649, "polymorphic_name": "pinhole", "ptr_wrapper": {
                   "id": 2147483661, "data": {"width": 1600, "height": 1200, "focal_length": 2750.25,
                                              "principal_point": [800.0, 600.0]}}}}],
           "extrinsics": extr, "structure": structure,
           "control_points": [{"key": 0, "value": {"X": ["@C0@", "@C1@", "@C2@"], "weight": "@C3@", "tags": [], "meta": {}}}]}
    text = json.dumps(doc, indent=1)
    # number spellings json.dumps would not produce
    for k, v in (("@F0@", "2892.3295898437500"), ("@P0@", "8.23E2"), ("@C0@", "1.0e3"), ("@C1@", "-0"), ("@C2@", "0.10000000149011612"),
                 ("@C3@", "18446744073709551616")):
        text = text.replace('"%s"' % k, v)
    return text.replace("\\\\u", "\\u").replace("\\\\/", "\\/").replace("\\\\n", "\\n"), pose_of_view


def _f32(x):
    return np.float32(x)


def test_written_file_is_byte_identical_to_the_reference_writers_and_reader_indexes_like_the_parser(driver, tmp_path):
    L = _lib()
    text, pose_of_view = _hand_written_openmvg()
    p_in = str(tmp_path / "in.json")
    open(p_in, "w").write(text)
    doc = json.loads(text)
    h = L.eg3d_sfm_read_json(p_in.encode())
    assert h
    # ---- (b) the reader: the fields as OpenMvgParser.cpp indexes them (driver `index`, same accessors)
    subprocess.check_call([driver, "index", p_in, str(tmp_path / "idx.txt")])
    K, E, Vw, P = {}, [], [], []
    for ln in open(str(tmp_path / "idx.txt")):
        t = ln.split()
        fl = lambda hx: np.array([int(hx, 16)], np.uint32).view(np.float32)[0]
        if t[0] == "K":
            K[int(t[1])] = [fl(x) for x in t[2:5]]
        elif t[0] == "E":
            E.append((int(t[1]), np.array([fl(x) for x in t[2:11]], np.float32).reshape(3, 3), np.array([fl(x) for x in t[11:14]], np.float32)))
        elif t[0] == "V":
            Vw.append((int(t[1]), int(t[2]), int(t[3]), int(t[4])))
        elif t[0] == "P":
            P.append((np.array([fl(x) for x in t[1:4]], np.float32),
                      [(int(o.split(":")[0]), fl(o.split(":")[1]), fl(o.split(":")[2])) for o in t[4:]]))
    assert L.eg3d_sfm_n_views(h) == len(Vw) == 3 and L.eg3d_sfm_n_points(h) == len(P) == 2
    seeds = D.Seeds()
    L.eg3d_sfm_seeds(h, C.byref(seeds))
    off = D.as_np(seeds.trk_off, 3, np.uint32)
    n_obs = int(off[-1])
    views = D.as_np(seeds.trk_view, n_obs, np.int32)
    xy = D.as_np(seeds.trk_xy, 2 * n_obs, np.float32).reshape(n_obs, 2)
    X = D.as_np(L.eg3d_sfm_points(h), 6, np.float32).reshape(2, 3)
    for i, (Xi, obs) in enumerate(P):
        assert np.array_equal(X[i].view(np.uint32), Xi.view(np.uint32))
        assert [o[0] for o in obs] == list(views[off[i]:off[i + 1]])      # position of the pose in `extrinsics`
        assert np.array_equal(np.array([[o[1], o[2]] for o in obs], np.float32).view(np.uint32), xy[off[i]:off[i + 1]].view(np.uint32))
    # camera matrices: K4 * [R t; 0 1] with t = -(R c), every product and sum in float in the order glm evaluates
    # eMatrix * kMatrix and vec3 * mat3 (OpenMvgParser.cpp:100-125, 285)
    Pm = D.as_np(L.eg3d_sfm_cam_P(h), 48, np.float32).reshape(3, 4, 4)
    pose_by_key = {k: (R, c) for k, R, c in E}
    for i, (w, hgt, id_k, id_pose) in enumerate(Vw):
        assert (w, hgt) == (1600, 1200) and id_pose == pose_of_view[i]
        f, px, py = K[id_k]
        R, c = pose_by_key[id_pose]
        t = np.zeros(3, np.float32)
        for r in range(3):
            t[r] = (_f32(-c[0]) * R[r, 0] + _f32(-c[1]) * R[r, 1]) + _f32(-c[2]) * R[r, 2]
        Em = np.zeros((4, 4), np.float32)
        Em[:3, :3] = R
        Em[:3, 3] = t
        Em[3, 3] = 1
        Km = np.zeros((4, 4), np.float32)
        Km[0, 0] = Km[1, 1] = f
        Km[0, 2], Km[1, 2], Km[2, 2] = px, py, 1
        want = np.zeros((4, 4), np.float32)
        for r in range(4):
            for cc in range(4):
                want[r, cc] = ((Km[r, 0] * Em[0, cc] + Km[r, 1] * Em[1, cc]) + Km[r, 2] * Em[2, cc]) + Km[r, 3] * Em[3, cc]
        assert np.array_equal(Pm[i].view(np.uint32), want.view(np.uint32)), i
    # ---- (a) the writer: add points with awkward coordinates, write, and compare with the document
    # output_sfm_data builds (driver `rewrite`: pass-through members from the input, extrinsics of view i = the
    # pose of view i with key i, structure with keys 0..N-1 and id_feat 0, every float as Value(float)), PrettyWriter
    rng = np.random.default_rng(5)
    extra = []
    tricky = np.array([1e-7, 1e21, 1e22, 3.4e38, 1.17549435e-38, 1e-45, 123456792.0, 0.1, 1 / 3, 2.5e-7, -0.0, 16777216.0],
                      np.float32)
    for i in range(2600):   # (ten of the writer's 256-point chunks: their seams are part of the comparison)
        Xp = np.concatenate([tricky, rng.standard_normal(200).astype(np.float32) * np.float32(10) ** rng.uniform(-20, 20, 200).astype(np.float32)])
        Xp = Xp[rng.integers(0, len(Xp), 3)].astype(np.float32)
        k = int(rng.integers(2, 4))
        vws = rng.permutation(3)[:k].astype(np.int32)
        pts = (rng.uniform(0, 1600, (k, 2)) * (10.0 ** rng.integers(-3, 1))).astype(np.float32)
        assert L.eg3d_sfm_add_point(h, D.np_ptr(np.ascontiguousarray(Xp), C.c_float), k, D.np_ptr(np.ascontiguousarray(vws), C.c_int32),
                                    D.np_ptr(np.ascontiguousarray(pts), C.c_float)) == 0
        extra.append((Xp, vws, pts))
    p_out = str(tmp_path / "out.json")
    assert L.eg3d_sfm_write_json(h, p_in.encode(), p_out.encode()) == 0
    blob = struct.pack("<i", 3)
    for i in range(3):
        R, c = pose_by_key[Vw[i][3]]
        blob += R.astype("<f4").tobytes() + c.astype("<f4").tobytes()
    blob += struct.pack("<i", len(P) + len(extra))
    for Xi, obs in P:
        blob += Xi.astype("<f4").tobytes() + struct.pack("<i", len(obs))
        for cam, x, y in obs:
            blob += struct.pack("<iff", cam, x, y)
    for Xp, vws, pts in extra:
        blob += Xp.astype("<f4").tobytes() + struct.pack("<i", len(vws))
        for cam, (x, y) in zip(vws, pts):
            blob += struct.pack("<iff", int(cam), float(x), float(y))
    open(str(tmp_path / "sfm.bin"), "wb").write(blob)
    p_ref = str(tmp_path / "ref.json")
    subprocess.check_call([driver, "rewrite", p_in, str(tmp_path / "sfm.bin"), p_ref])
    got, want = open(p_out, "rb").read(), open(p_ref, "rb").read()
    if got != want:
        for n, (a, b) in enumerate(zip(got.split(b"\n"), want.split(b"\n"))):
            assert a == b, (n, a, b)
    assert got == want
    assert json.loads(got)["control_points"][0]["value"]["weight"] == 18446744073709551616.0
    L.eg3d_sfm_destroy(h)
