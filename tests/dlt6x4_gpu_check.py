"""Run as a script in its OWN process with EG3D_LIB=<repo>/edgegraph3d_amd/libeg3d_dlt6x4.so (by
tests/test_dlt_forms.py): the library built with the three-rows-per-view (6x4, OpenCV <= 3.1) DLT
system must equal the oracle in the same mode bit for bit, and the committed 6x4 fixture."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from edgegraph3d_amd import api, host  # noqa: E402
from oracle import binding as ob  # noqa: E402
from parity_util import compare_edgepoints  # noqa: E402


def main():
    assert api.lib().eg3d_dlt_rows() == 3, "EG3D_LIB must point at the 6x4 build"
    assert ob.lib().orc_get_dlt_rows() == 3   # the binding follows the product library
    z = np.load(os.path.join(ROOT, "tests", "golden", "synthetic_tiny_v1_dlt6x4.npz"))
    want = {k: z["out_" + k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    want["n_points"], want["n_obs"] = len(want["X"]), len(want["obs_view"])
    s = host.Synth(0)
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    rep = compare_edgepoints(want, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    ctx.close()
    for cfg in (1, 2, 3):
        s = host.Synth(cfg)
        ctx = api.Context(s.scene)
        got = ctx.match_refpoints(s.seeds)
        ref = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, os.cpu_count())
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
        print("config %d: %d edge-points bit-exact in 6x4 mode" % (cfg, got["n_points"]))
        ctx.close()
    # the randomised / hostile scenes of tests/fuzz_scenes.py, in this mode
    from fuzz_scenes import HOSTILE_KINDS, draw, hostile
    n_ok = 0
    for make, cases in ((draw, list(range(0, 40, 3)) + [101]), (lambda c: hostile(c, [HOSTILE_KINDS[c]]), range(6))):
        for case in cases:
            s, sa, seeds = make(case)
            n = len(seeds.trk_off) - 1
            ctx = api.Context(C.byref(sa.c))
            got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
            ref = ob.Oracle(C.byref(sa.c)).match(C.byref(seeds.c), 0, n, os.cpu_count())
            rep = compare_edgepoints(ref, got)
            assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, rep["msgs"][:3])
            ctx.close()
            n_ok += 1
    print("%d randomised / hostile scenes bit-exact in 6x4 mode" % n_ok)


if __name__ == "__main__":
    main()
    print("DLT6X4-OK")
