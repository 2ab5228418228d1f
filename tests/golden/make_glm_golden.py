#!/usr/bin/env python3
"""BUILD CONTAINER ONLY. Generates tests/golden/glm_pin_v1.npz: inputs and the outputs the reference's own vendored glm
(/root/reference/external/glm 0.9.6, header-only; driven by tests/glm/glm_driver.cpp of this repository, compiled with the
reference's release flags) computes for the glm expressions the hot path's arithmetic rests on:

    proj      compute_projection                      geometric_utilities.cpp:973-977   (vec4 * mat4, divisions)
    cam       t = -center * R ; P = eMatrix * kMatrix OpenMvgParser.cpp:289, :107-125
    anglecos  compute_anglecos                        geometric_utilities.cpp:579-618   (glm::dot on vec2)
    mindist   minimum_distancesq                      geometric_utilities.cpp:940-954, :555-557

The fixture is DATA (inputs + glm's outputs); tests/test_glm_pin.py checks the oracle's hand-written evaluation orders, the
product's host camera model and (on the GPU) the device primitives against it everywhere, and re-runs the comparison live
on ~1 M fresh cases per function where the reference tree is present.

    python tests/golden/make_glm_golden.py            # writes tests/golden/glm_pin_v1.npz (2048 cases per function)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GLM_INC = "/root/reference/external/glm"
DRIVER_SRC = os.path.join(ROOT, "tests", "glm", "glm_driver.cpp")
DRIVER_BIN = os.path.join(ROOT, "oracle", "_ref", "glm_driver")   # oracle/_ref: built from the reference tree, git-ignored
RECORD = {"proj": (19, 5), "cam": (15, 19), "anglecos": (7, 1), "mindist": (6, 3)}


def have_reference():
    return os.path.isdir(os.path.join(GLM_INC, "glm"))


def build_driver():
    """g++ with the reference's release flags (CMakeLists.txt:48: -O3 -funroll-loops; baseline x86-64, so no FMA)."""
    if not have_reference():
        raise RuntimeError("the reference tree (%s) is not present" % GLM_INC)
    os.makedirs(os.path.dirname(DRIVER_BIN), exist_ok=True)
    if (not os.path.exists(DRIVER_BIN)) or os.path.getmtime(DRIVER_BIN) < os.path.getmtime(DRIVER_SRC):
        subprocess.check_call(["g++", "-O3", "-funroll-loops", "-std=c++11", "-w", "-I", GLM_INC, "-o", DRIVER_BIN, DRIVER_SRC])
    return DRIVER_BIN


def run_glm(mode, inputs):
    """inputs: float32 [n][RECORD[mode][0]] -> glm's outputs, float32 [n][RECORD[mode][1]]."""
    a = np.ascontiguousarray(inputs, np.float32)
    assert a.ndim == 2 and a.shape[1] == RECORD[mode][0]
    p = subprocess.run([build_driver(), mode], input=a.tobytes(), stdout=subprocess.PIPE, check=True)
    return np.frombuffer(p.stdout, np.float32).reshape(a.shape[0], RECORD[mode][1]).copy()


def _rot(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1)
    return R.astype(np.float32)


def cases(mode, n, seed):
    """Seeded inputs: DTU-like magnitudes, then a tail of edge cases (zeros, huge / tiny values, points on the camera
    plane, degenerate segments, vertical lines, NaN / inf) — the last `n_edge` records."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    if mode == "cam":
        fpp = np.stack([rng.uniform(500, 4000, n), rng.uniform(300, 1300, n), rng.uniform(200, 900, n)], 1)
        a = np.concatenate([fpp, _rot(rng, n), rng.normal(0, 500, (n, 3))], 1).astype(f32)
        a[-1, 3:12] = 0
        a[-2, 12:15] = 0
        a[-3, 0] = 1e-30
        a[-4, 12:15] = [1e20, -1e20, 3.0]
        return a
    if mode == "proj":
        cam = cases("cam", n, seed + 1)
        P = run_glm("cam", cam)[:, 3:]
        X = rng.normal(0, 200, (n, 3))
        a = np.concatenate([P, X], 1).astype(f32)
        a[-1, :16] = 0                      # zero camera: 0/0
        a[-2, 16:19] = [1e30, -1e30, 1e30]
        a[-3, 8:12] = [0, 0, 0, 0]          # z row zero: division by zero
        a[-4, 16:19] = [np.nan, 1.0, 2.0]
        a[-5, 16:19] = [np.inf, 1.0, 2.0]
        a[-6, :16] = rng.normal(0, 1e-20, 16)
        a[-7, 12:16] = [1.0, 2.0, 3.0, 4.0]  # a non-zero last row must not matter ([r][c] storage, Q6)
        return a
    if mode == "anglecos":
        seg = rng.uniform(0, 1600, (n, 4))
        k = n // 3
        seg[:k, 2:] = seg[:k, :2] + rng.normal(0, 8, (k, 2))   # short segments, as polylines have
        line = rng.normal(0, 1, (n, 3))
        line[:, 2] *= 1000
        a = np.concatenate([seg, line], 1).astype(f32)
        a[-1, 5] = 0                        # b == 0: the (0, 1) direction
        a[-2, 2:4] = a[-2, 0:2]             # zero-length segment: 0/0
        a[-3, 4] = 0
        a[-4, 4:6] = [1e30, 1e-30]
        a[-5, 4:7] = [np.nan, 1.0, 0.0]
        a[-6, 5] = -0.0
        return a
    if mode == "mindist":
        v = rng.uniform(0, 1600, (n, 2))
        w = v + rng.normal(0, 12, (n, 2))
        p = v + rng.normal(0, 30, (n, 2))
        a = np.concatenate([p, v, w], 1).astype(f32)
        a[-1, 4:6] = a[-1, 2:4]             # v == w
        a[-2, 0:2] = a[-2, 2:4]             # p == v
        a[-3, 0:2] = [np.nan, 1.0]          # NaN: max<float>(0, min<float>(1, NaN))
        a[-4, 4:6] = a[-4, 2:4] + f32(1e-20)
        a[-5, 0:2] = [1e30, -1e30]
        a[-6, 4:6] = [np.inf, 0.0]
        return a
    raise ValueError(mode)


def main():
    n = 2048
    out = {}
    for i, mode in enumerate(("cam", "proj", "anglecos", "mindist")):
        a = cases(mode, n, 0xE63D2018 + 16 * i)
        out[mode + "_in"] = a
        out[mode + "_out"] = run_glm(mode, a)
    path = os.path.join(ROOT, "tests", "golden", "glm_pin_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
