"""ctypes binding of the test-only host simulation (tests/hostsim/libhostsim.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

from edgegraph3d_amd import _cdefs as D

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
        subprocess.check_call(["make", "-s", "-C", d])
        L = C.CDLL(os.path.join(d, "libhostsim.so"))
        L.hostsim_match.argtypes = [C.POINTER(D.Scene), C.POINTER(D.Seeds), C.c_uint32, C.c_uint32,
                                    C.POINTER(D.Candidates), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                    C.POINTER(D.EdgePoints)]
        L.hostsim_free_edgepoints.argtypes = [C.POINTER(D.EdgePoints)]
        L.hostsim_closest_pruned_mismatches.argtypes = [D.f32p, C.c_int, D.f32p, C.c_int, C.c_uint32, C.c_uint32]
        L.hostsim_walk_pf_mismatches.argtypes = [D.f32p, C.c_int, C.c_uint32, C.c_uint32, D.f32p, C.c_int]
        L.hostsim_dist2.restype = C.c_float
        L.hostsim_dist2.argtypes = [C.c_float] * 4
        L.hostsim_seg_closest.restype = C.c_float
        L.hostsim_seg_closest.argtypes = [C.c_float] * 6 + [D.f32p]
        L.hostsim_walk_by_distance.argtypes = [D.f32p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                               C.c_float, C.c_uint32, C.c_float, D.u32p, D.f32p]
        L.hostsim_walk_by_line.argtypes = [D.f32p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                           C.c_uint32, D.f32p, C.c_int, C.c_float, C.c_float, D.u32p, D.f32p]
        L.hostsim_triangulate.argtypes = [D.f32p, D.i32p, D.f32p, C.c_int, D.f32p]
        L.hostsim_gn_filter.argtypes = [D.f32p, D.f32p, D.u32p, D.i32p, D.f32p, C.c_uint64, C.c_float, C.c_int,
                                        D.f32p, D.u8p]
        _LIB = L
    return _LIB


def match(scene_ptr, seeds_ptr, begin, end, cand_struct, hyp_cap=192, chain_cap=768, pool_cap=8192, slot_step=False):
    e = D.EdgePoints()
    rc = lib().hostsim_match(scene_ptr, seeds_ptr, begin, end, C.byref(cand_struct), hyp_cap, chain_cap, pool_cap,
                             int(slot_step), C.byref(e))
    if rc != 0:
        raise RuntimeError("hostsim_match rc=%d" % rc)
    d = D.edgepoints_to_dict(e)
    lib().hostsim_free_edgepoints(C.byref(e))
    return d
