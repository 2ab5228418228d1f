"""CPU twin of tests/test_gpu_fuzz.py: on randomised, mutated scenes (tests/fuzz_scenes.py) the DEVICE code of
stage B compiled for the host (tests/hostsim) must equal the oracle bit for bit. Independent of a GPU, so it runs
in every round's CPU suite; the device run of the same cases (all kernels, through the C ABI) is the gpu test."""
import ctypes as C

import pytest

import hostsim_binding as hs
from fuzz_scenes import HOSTILE_KINDS, draw, hostile
from oracle import binding as ob
from parity_util import compare_edgepoints


@pytest.mark.parametrize("case", [0, 1, 2, 3, 5, 6, 9, 13])
def test_random_mutated_scene_hostsim_vs_oracle(case):
    s, sa, seeds = draw(case)
    n = len(seeds.trk_off) - 1
    o = ob.Oracle(C.byref(sa.c))
    ref = o.match(C.byref(seeds.c), 0, n, 1)
    cand = o.candidates_raw(C.byref(seeds.c), 0, n)
    got = hs.match(C.byref(sa.c), C.byref(seeds.c), 0, n, cand, slot_step=bool(case & 1))
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, rep["msgs"][:3])
    assert got["n_chains"] == ref["stats"]["n_chains"] and got["n_tasks"] == ref["stats"]["n_tasks"]


@pytest.mark.parametrize("case", range(6))
def test_hostile_numeric_inputs_hostsim_vs_oracle(case):
    """NaN / inf / huge seed observations, zero-length segments, a zero camera, a NaN fundamental matrix: the
    device code compiled for the host still equals the oracle (the GPU run of the same cases is the gpu test)."""
    s, sa, seeds = hostile(case, [HOSTILE_KINDS[case]])
    n = len(seeds.trk_off) - 1
    o = ob.Oracle(C.byref(sa.c))
    ref = o.match(C.byref(seeds.c), 0, n, 1)
    cand = o.candidates_raw(C.byref(seeds.c), 0, n)
    got = hs.match(C.byref(sa.c), C.byref(seeds.c), 0, n, cand)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"], (case, rep["msgs"][:3])
