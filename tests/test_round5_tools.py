"""CPU-side checks of the round-5 infrastructure: the build guard reads the kernels' resource figures from the built
library, the structure-tolerant comparator, and the oracle's convention switches (test hook)."""
import os
import sys

import numpy as np

from edgegraph3d_amd import api, build, host
from oracle import binding as ob
from parity_util import chain_ranges, compare_by_chain, compare_edgepoints

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_guard_reads_the_code_objects_and_the_bounds_hold():
    """tools/kernel_resources.py finds every instantiation of the expand kernel in the built library with plausible
    figures, and the committed bounds of build.py hold for it (what build_hip() enforces). Round 6: the product libraries
    no longer carry the lane-per-chain engine's kernels; the variant built with -DEG3D_WITH_K3C_ENGINE does, within its
    own bounds."""
    if not os.path.exists(api.lib_path()):
        build.build_hip()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    res = {n: r for n, r in kr.kernel_resources(api.lib_path()).items() if n.startswith("eg3d::")}
    expand = [n for n in res if "k3b_expand_t<" in n]
    assert len(expand) == 3 and not [n for n in res if "k3c_engine" in n]
    if os.path.exists(build.HIP_LIB_ENGINE):
        eng = {n: r for n, r in kr.kernel_resources(build.HIP_LIB_ENGINE).items() if n.startswith("eg3d::")}
        assert len([n for n in eng if "k3c_engine_t<" in n]) == 2
        assert kr.check_bounds(eng, dict(build.RESOURCE_BOUNDS, **build.ENGINE_RESOURCE_BOUNDS)) == []
    for n in expand:
        assert res[n]["vgpr_count"] == 128 and 0 < res[n]["group_segment_fixed_size"] <= 10240, (n, res[n])
    assert kr.check_bounds(res, build.RESOURCE_BOUNDS) == []
    # a bound that is exceeded is reported, a kernel that disappeared too
    bad = kr.check_bounds(res, {"k3b_expand_t<4, 0, 0>": {"vgpr_spill_count": -1}, "no_such_kernel": {"vgpr_spill_count": 0}})
    assert len(bad) == 2 and "no kernel matches" in bad[1]
    assert len(build.device_source_fingerprint()) == 16


def test_compare_by_chain_matches_chains_by_key_not_by_position():
    s = host.Synth(1)
    o = ob.Oracle(s.scene)
    a = o.match(s.seeds, 0, s.n_seeds, 4)
    rep = compare_by_chain(a, a)
    assert rep["chains_structurally_identical"] == rep["chains_in_both"] == len(chain_ranges(a)) > 10
    assert rep["points_X_bit_equal"] == rep["points_compared"] == a["n_points"] and rep["obs_xy_bit_equal"] == a["n_obs"]
    # drop the first chain of one side: the strict comparison fails at once, the tolerant one loses exactly that chain
    first = chain_ranges(a)[tuple(int(x) for x in a["key"][0][:3])]
    n0, o0 = first[1], int(a["obs_off"][first[1]])
    b = dict(a)
    b["n_points"], b["n_obs"] = a["n_points"] - n0, a["n_obs"] - o0
    b["X"], b["key"] = a["X"][n0:], a["key"][n0:]
    b["obs_off"] = a["obs_off"][n0:] - a["obs_off"][n0]
    for k in ("obs_view", "obs_pl", "obs_seg", "obs_xy"):
        b[k] = a[k][o0:]
    assert not compare_edgepoints(a, b)["ok"]
    rep = compare_by_chain(a, b)
    assert rep["chains_only_ref"] == 1 and rep["chains_only_got"] == 0
    assert rep["chains_structurally_identical"] == rep["chains_in_both"] == len(chain_ranges(a)) - 1
    assert rep["points_X_bit_equal"] == rep["points_compared"] == b["n_points"]


def test_oracle_convention_switches_are_a_hook_and_default_off():
    """orc_set_conventions(mask) changes arithmetic conventions of the unpinned OpenCV routines in the ORACLE only
    (tools/convention_report.py); mask 0 is the restatement the GPU path is held to, and setting it back restores it."""
    L = ob.lib()
    s = host.Synth(1)
    o = ob.Oracle(s.scene)
    base = o.match(s.seeds, 0, s.n_seeds, 2)
    try:
        L.orc_set_conventions(4 | 8)  # hypot + reversed rotation pairs in the DLT's Jacobi SVD
        alt = o.match(s.seeds, 0, s.n_seeds, 2)
    finally:
        L.orc_set_conventions(0)
    again = o.match(s.seeds, 0, s.n_seeds, 2)
    assert compare_edgepoints(base, again)["bitexact_X"]
    rep = compare_by_chain(base, alt)
    assert rep["chains_in_both"] > 10 and rep["share_chains_identical"] > 0.3
    assert rep["points_X_bit_equal"] < rep["points_compared"]          # the start of Gauss-Newton really moved ...
    assert rep["share_points_within_tol"] > 0.99                        # ... and the solutions barely
