#!/usr/bin/env python3
"""For a user who HAS the reference built (OpenCV and all): measure how far this library's cloud agrees with the
reference's on the same input — the only route by which the unpinned part of the oracle (the conventions of
cv::triangulatePoints' SVD, GEMM, determinant / inverse) ever gets pinned.

  1. run the reference's edge_matching on your scene           -> <ref_out>/before_filtering.json
  2. run this library's on the same scene (examples/edge_matching_main, or libeg3d_dlt4x4.so for OpenCV >= 3.2)
                                                               -> <our_out>/before_filtering.json
  3. python tools/compare_sfm_json.py <ref_out>/before_filtering.json <our_out>/before_filtering.json [input.json] [--json out]

Both files are OpenMVG sfm_data JSON (SfM points first, the new edge-points appended: output_sfm_data.cpp:186-229). An
edge-point carries no chain key in the file, so points are matched by what identifies them geometrically: the list of
their 2-D observations (view id, x, y). 2-D coordinates come from walking the polylines in float and do not depend on the
triangulation conventions, so a point that exists on both sides with the same observations is "structurally identical",
and its 3-D coordinates are then compared (bit-equal / within 1e-4 relative, the north star's tolerance). Points whose
FIRST observation matches but whose lists differ grew differently (an accept / reject decision flipped somewhere); the
rest exist on one side only. Expect the orders of magnitude of DESIGN.md 3's table (a different Jacobi convention alone
leaves 62-83 % of the chains identical), not bit equality. CPU only."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import _cdefs as D, host  # noqa: E402


def read_cloud(path):
    """(X [N,3] float32, trk_off [N+1], trk_view, trk_xy [M,2]) of an OpenMVG JSON through the library's own reader."""
    L = host.lib()
    L.eg3d_sfm_read_json.restype = C.c_void_p
    L.eg3d_sfm_read_json.argtypes = [C.c_char_p]
    L.eg3d_sfm_destroy.argtypes = [C.c_void_p]
    L.eg3d_sfm_n_points.argtypes = [C.c_void_p]
    L.eg3d_sfm_n_points.restype = C.c_uint64
    L.eg3d_sfm_seeds.argtypes = [C.c_void_p, C.POINTER(D.Seeds)]
    L.eg3d_sfm_points.argtypes = [C.c_void_p]
    L.eg3d_sfm_points.restype = D.f32p
    h = L.eg3d_sfm_read_json(path.encode())
    if not h:
        raise SystemExit("cannot read %s as OpenMVG sfm_data JSON" % path)
    n = int(L.eg3d_sfm_n_points(h))
    s = D.Seeds()
    L.eg3d_sfm_seeds(h, C.byref(s))
    off = D.as_np(s.trk_off, n + 1, np.uint32).astype(np.int64)
    m = int(off[-1])
    out = (D.as_np(L.eg3d_sfm_points(h), 3 * n, np.float32).reshape(n, 3), off, D.as_np(s.trk_view, m, np.int32),
           D.as_np(s.trk_xy, 2 * m, np.float32).reshape(m, 2))
    L.eg3d_sfm_destroy(h)
    return out


def compare(ref, got, n_input=None, rel_tol=1e-4):
    Xa, oa, va, xa = ref
    Xb, ob, vb, xb = got
    if n_input is None:  # the SfM points both files start with: the longest common prefix of identical points
        n_input = 0
        lim = min(len(Xa), len(Xb))
        while n_input < lim and np.array_equal(Xa[n_input], Xb[n_input]) and oa[n_input + 1] - oa[n_input] == ob[n_input + 1] - ob[n_input] \
                and np.array_equal(va[oa[n_input]:oa[n_input + 1]], vb[ob[n_input]:ob[n_input + 1]]) \
                and np.array_equal(xa[oa[n_input]:oa[n_input + 1]], xb[ob[n_input]:ob[n_input + 1]]):
            n_input += 1

    def sig(off, v, xy, i, first_only=False):
        a, b = int(off[i]), int(off[i + 1])
        if first_only:
            b = min(b, a + 1)
        return v[a:b].tobytes() + xy[a:b].tobytes()

    full_b, first_b = {}, {}
    for i in range(n_input, len(Xb)):
        full_b.setdefault(sig(ob, vb, xb, i), []).append(i)
        first_b.setdefault(sig(ob, vb, xb, i, True), []).append(i)
    rep = {"input_points": int(n_input), "edge_points_ref": int(len(Xa) - n_input), "edge_points_got": int(len(Xb) - n_input),
           "structurally_identical": 0, "X_bit_equal": 0, "X_within_tol": 0, "max_rel_dX": 0.0,
           "same_first_observation_different_list": 0, "only_in_ref": 0, "rel_tol": rel_tol}
    used = set()
    for i in range(n_input, len(Xa)):
        cand = [j for j in full_b.get(sig(oa, va, xa, i), []) if j not in used]
        if cand:
            j = cand[0]
            used.add(j)
            rep["structurally_identical"] += 1
            if np.array_equal(Xa[i].view(np.uint32), Xb[j].view(np.uint32)):
                rep["X_bit_equal"] += 1
            d = float(np.linalg.norm(Xa[i].astype(np.float64) - Xb[j]) / max(1e-12, np.linalg.norm(Xa[i].astype(np.float64))))
            rep["max_rel_dX"] = max(rep["max_rel_dX"], d)
            if d <= rel_tol:
                rep["X_within_tol"] += 1
        elif sig(oa, va, xa, i, True) in first_b:
            rep["same_first_observation_different_list"] += 1
        else:
            rep["only_in_ref"] += 1
    rep["only_in_got"] = rep["edge_points_got"] - len(used) - 0
    e = max(1, rep["edge_points_ref"])
    rep["structurally_identical_frac"] = rep["structurally_identical"] / e
    rep["within_tol_of_identical_frac"] = rep["X_within_tol"] / max(1, rep["structurally_identical"])
    return rep


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) < 2:
        raise SystemExit(__doc__)
    n_input = None
    if len(args) > 2:
        n_input = len(read_cloud(args[2])[0])
    rep = compare(read_cloud(args[0]), read_cloud(args[1]), n_input)
    txt = json.dumps(rep, indent=1)
    print(txt)
    if "--json" in sys.argv:
        open(sys.argv[sys.argv.index("--json") + 1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
