"""Randomised scenes, device vs oracle: the committed configs exercise a handful of shapes; this draws
small scenes with random rigs, noise levels and image sizes and then MUTATES them the way real inputs
are irregular — view pairs without a fundamental matrix, loop polylines (start == end, Q8), invalid
polylines that kept their vertices, tracks that repeat a view id (Q2), observations on or outside the
image border (Q7), very short tracks — and requires the full path, stage A and the polyline-set path to
agree with the oracle bit for bit on every one. Seeds are fixed: a failure names its case."""
import ctypes as C

import numpy as np
import pytest

from edgegraph3d_amd import api, host
from fuzz_scenes import draw
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu

CASES = list(range(40))


def _oracle(scene_ptr):
    from oracle import binding as ob
    return ob.Oracle(scene_ptr)


@pytest.mark.parametrize("case", CASES)
def test_random_mutated_scene_matches_oracle(case):
    s, sa, seeds = draw(case)
    n = len(seeds.trk_off) - 1
    ctx = api.Context(C.byref(sa.c))
    orc = _oracle(C.byref(sa.c))
    got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
    ref = orc.match(C.byref(seeds.c), 0, n, nthreads=8)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, rep["msgs"][:3])
    assert got["flags"] == ref["flags"], (case, got["flags"], ref["flags"])
    # stage A alone
    ca, cb = ctx.candidates(C.byref(seeds.c), 0, n), orc.candidates(C.byref(seeds.c), 0, n)
    for k in cb:
        x, y = ca[k], cb[k]
        if isinstance(y, np.ndarray):
            bits = (lambda a: a.view(np.uint32) if a.dtype == np.float32 else a)
            assert np.array_equal(bits(x), bits(y)), (case, k)
        else:
            assert x == y, (case, k, x, y)
    # the polyline-set path on the same mutated scene
    n_sets, row_off, ids = s.polyline_sets(2)
    gs = ctx.match_polyline_sets(n_sets, row_off, ids)
    rs = orc.match_polyline_sets(n_sets, row_off, ids, 0, n_sets, 8)
    rep = compare_edgepoints(rs, gs)
    assert rep["ok"] and rep["bitexact_X"], (case, rep["msgs"][:3])
    ctx.close()
