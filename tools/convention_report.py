#!/usr/bin/env python3
"""How exposed is the cloud to the conventions of the OpenCV routines the oracle could NOT pin (no OpenCV in this
image: the DLT's Jacobi SVD, the GEMM summation order of the Gauss-Newton normal equations, the epipolar-line
normalisation)? Runs the CPU oracle on C2 and C3' with one convention changed at a time (oracle test hook
orc_set_conventions, oracle/oracle_tri.hpp g_conv) and compares every variant with the restatement as it is, chain by
chain (tests/parity_util.py compare_by_chain). CPU only; writes profiles/r05_convention_sensitivity.json.
usage: tools/convention_report.py [--quick]   (--quick: C2 and the first 1500 seeds of C3')"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from edgegraph3d_amd import host
from oracle import binding as ob
from parity_util import compare_by_chain

VARIANTS = [
    ("gemm_2_accumulators", 1, "J^T J and (H^-1 J^T) r summed with two interleaved accumulators (even / odd terms)"),
    ("gemm_4_accumulators", 2, "... with four interleaved accumulators"),
    ("jacobi_hypot", 4, "Jacobi SVD of the DLT: gamma = hypot(p, beta) instead of sqrt(p*p + beta*beta)"),
    ("jacobi_reverse_pairs", 8, "Jacobi SVD of the DLT: rotation pairs visited in the opposite order"),
    ("epiline_divide", 16, "epipolar line normalised by a / sqrt(nu) instead of a * (1 / sqrt(nu))"),
    ("all_of_the_above", 2 | 4 | 8 | 16, "four accumulators + hypot + reversed pairs + divided epipolar lines"),
]
quick = "--quick" in sys.argv
L = ob.lib()
nt = os.cpu_count() or 1
report = {"what": __doc__.split("usage:")[0].strip(), "dlt_rows": int(L.orc_get_dlt_rows()), "workloads": {}}
for name, cfg, nseeds in (("C2", 2, 0), ("C3'", 3, 1500 if quick else 0)):
    s = host.Synth(cfg)
    n = nseeds or s.n_seeds
    o = ob.Oracle(s.scene)
    L.orc_set_conventions(0)
    base = o.match(s.seeds, 0, n, nt)
    w = {"seeds": n, "points": int(base["n_points"]), "variants": {}}
    for vname, mask, what in VARIANTS:
        L.orc_set_conventions(mask)
        r = o.match(s.seeds, 0, n, nt)
        L.orc_set_conventions(0)
        rep = compare_by_chain(base, r, 1e-4)
        rep["what"] = what
        w["variants"][vname] = rep
        print("%-4s %-22s chains identical %6d / %6d (%.2f %%)  points within 1e-4: %8d / %8d (%.3f %%)  bit-equal %8d  max rel dX %.2e  points %d -> %d"
              % (name, vname, rep["chains_structurally_identical"], rep["chains_in_both"], 100 * rep["share_chains_identical"],
                 rep["points_X_within_tol"], rep["points_compared"], 100 * rep["share_points_within_tol"], rep["points_X_bit_equal"],
                 rep["max_rel_dX"], rep["points_ref"], rep["points_got"]), flush=True)
    report["workloads"][name] = w
out = os.path.join(ROOT, "profiles", "r05_convention_sensitivity%s.json" % ("_quick" if quick else ""))
with open(out, "w") as f:
    json.dump(report, f, indent=1)
print("wrote", out)
