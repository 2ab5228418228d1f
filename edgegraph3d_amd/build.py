"""Build recipes for the native parts of edgegraph3d_amd (in-tree, no pip).

  libeg3d_host.so  g++    host-side utilities (synthetic workload, grid builder, JSON, post steps)
  libeg3d.so       hipcc  the C-ABI library with the gfx950 kernels (include/eg3d.h)

`python -m edgegraph3d_amd.build` builds both. hipcc cross-compiles gfx950 without a GPU.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "edgegraph3d_amd")
HOST_DIR = os.path.join(PKG, "host")
CSRC_DIR = os.path.join(PKG, "csrc")
INC_DIR = os.path.join(ROOT, "include")

HOST_LIB = os.path.join(PKG, "libeg3d_host.so")
HIP_LIB = os.path.join(PKG, "libeg3d.so")

# -ffp-contract=off everywhere: decisions on the path are float threshold tests and the
# arithmetic contract forbids FMA formation on host and device alike.
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-Wall"]
# Optimisation switches of the device code, chosen by measurement on the MI355X. The expand
# kernel is register-bound at its 128 VGPRs (4 waves per SIMD): what lengthens live ranges or adds induction variables costs
# it more than it buys. Each step below was kept because the kernel got faster (C3', one step at a time: -O3 47.1-47.5 ms;
# -O2 46.0-46.3; vectorizers off 45.3-45.4; no machine LICM 45.0-45.4; no loop strength reduction 43.7-44.3; no GVN-PRE,
# no pre-RA machine scheduler 43.6-43.7; C4's step in flight 1737 -> 1673 ms); the other kernels are the same within
# noise. Results are bit-identical (the whole GPU suite runs on this build).
HIP_OPT_FLAGS = ["-O2", "-fno-slp-vectorize", "-fno-vectorize", "-mllvm", "-disable-machine-licm", "-mllvm", "-disable-lsr",
                 "-mllvm", "-enable-pre=false", "-mllvm", "-enable-load-pre=false", "-mllvm", "-enable-misched=false"]
HIP_FLAGS = ["--offload-arch=gfx950"] + HIP_OPT_FLAGS + ["-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc",
             "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _all_sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_host(force=False):
    srcs = _all_sources(HOST_DIR, (".cpp",))
    deps = (srcs + _all_sources(HOST_DIR, (".hpp", ".h")) + _all_sources(INC_DIR, (".h",)) + _all_sources(CSRC_DIR, (".h", ".hpp"))
            + _all_sources(os.path.join(PKG, "rccl"), (".h",)))
    if force or _newer(HOST_LIB, deps):
        _run(["g++"] + HOST_FLAGS + ["-I", INC_DIR, "-I", CSRC_DIR, "-o", HOST_LIB] + srcs + ["-lz"])  # libz: PNG inflate
    return HOST_LIB


OBJ_DIR = os.path.join(PKG, "_obj")  # per-library object files + dependency files (untracked, not sent to the GPU box)


def _obj_stale(obj, dep, stamp):
    """An object is rebuilt when it, its dependency file (written by -MD) or the flag stamp is missing, or when anything
    the dependency file lists is newer."""
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)):
        return True
    t = os.path.getmtime(obj)
    if os.path.getmtime(stamp) > t:
        return True
    text = open(dep).read().replace("\\\n", " ")
    for f in text.split(":", 1)[-1].split():
        if not os.path.exists(f) or os.path.getmtime(f) > t:
            return True
    return False


def build_hip(force=False, out=None, defines=()):
    """hipcc every source to its own object (in parallel, cached by the compiler's own dependency files), link to a
    TEMPORARY file, run the build guard on it, and only then move it to `out`: a library that violates the committed
    resource bounds (or whose figures cannot be read) never reaches the place the next call would find up to date."""
    out = out or HIP_LIB
    srcs = _all_sources(CSRC_DIR, (".hip",))
    host_srcs = [os.path.join(HOST_DIR, "grid_build.cpp")]
    deps = srcs + host_srcs + _all_sources(CSRC_DIR, (".h", ".hpp")) + _all_sources(INC_DIR, (".h",)) + [os.path.abspath(__file__)]  # (the flags live here)
    if not (force or _newer(out, deps)):
        return out
    import concurrent.futures
    import hashlib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("EG3D_EXTRA_HIPFLAGS", "").split()
    cflags = [f for f in HIP_FLAGS if f != "-shared"] + extra + list(defines) + ["-I", INC_DIR, "-I", CSRC_DIR, "-I", HOST_DIR]
    odir = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(out))[0])
    os.makedirs(odir, exist_ok=True)
    stamp = os.path.join(odir, "flags.%s" % hashlib.sha256(" ".join([hipcc] + cflags).encode()).hexdigest()[:16])
    if not os.path.exists(stamp):
        for f in os.listdir(odir):
            os.unlink(os.path.join(odir, f))  # other switches: nothing cached is valid
        open(stamp, "w").close()
    jobs = []
    for src in srcs + host_srcs:
        obj = os.path.join(odir, os.path.basename(src) + ".o")
        if force or _obj_stale(obj, obj + ".d", stamp):
            jobs.append([hipcc] + cflags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(_run, jobs))
    tmp = out + ".tmp%d" % os.getpid()
    try:
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", tmp]
             + [os.path.join(odir, os.path.basename(s) + ".o") for s in srcs + host_srcs])
        check_resources(tmp, record_as=out)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    return out


# The same library with the OTHER form of cv::triangulatePoints' system (two rows per view, 4x4: OpenCV >= 3.2 /
# 4.x; the default build follows the 6x4 form of the OpenCV 3.1 the reference names). Both are kept bit-exact
# against the oracle in the matching mode: the whole `-m gpu` suite runs once per library (tests/conftest.py).
# Selected with EG3D_LIB=<this file>.
HIP_LIB_DLT4X4 = os.path.join(PKG, "libeg3d_dlt4x4.so")


def build_hip_dlt4x4(force=False):
    return build_hip(force, HIP_LIB_DLT4X4, ("-DEG3D_DLT_ROWS=2",))


# The default form + the lane-per-chain engine of the expand stage (eg3d_k3c_engine.h, round 5: bit-exact, 2.4x slower than
# k3b_expand). The product libraries are built WITHOUT it; this variant exists so that the engine's GPU tests
# (tests/test_gpu_engine.py) keep a second complete implementation of rows a10-a16 honest. EG3D_K3B_ENGINE=1 selects it at run time.
HIP_LIB_ENGINE = os.path.join(PKG, "variants", "libeg3d_engine.so")


def build_hip_engine(force=False):
    os.makedirs(os.path.dirname(HIP_LIB_ENGINE), exist_ok=True)
    return build_hip(force, HIP_LIB_ENGINE, ("-DEG3D_WITH_K3C_ENGINE=1",))


PROBE_LIB = os.path.join(ROOT, "tests", "probe", "libeg3d_probe.so")


def build_probe(force=False):
    """TEST-ONLY: device primitives of the product's headers behind a tiny C interface
    (tests/probe/eg3d_probe.h) for the bit-for-bit arithmetic checks; never linked into libeg3d.so."""
    src = os.path.join(ROOT, "tests", "probe", "eg3d_probe.hip")
    deps = [src] + _all_sources(CSRC_DIR, (".h", ".hpp")) + [os.path.abspath(__file__)]
    if force or _newer(PROBE_LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc] + HIP_FLAGS + ["-I", INC_DIR, "-I", CSRC_DIR, "-I", os.path.dirname(src), "-o", PROBE_LIB, src])
    return PROBE_LIB


RCCL_LIB = os.path.join(PKG, "libeg3d_rccl.so")


def build_rccl(force=False):
    """The RCCL all-gather of the edge-point cloud for C/C++ hosts (include/eg3d_rccl.h): its own
    library so that libeg3d.so carries no communication dependency."""
    src = os.path.join(PKG, "rccl", "eg3d_rccl.hip")
    deps = [src] + _all_sources(INC_DIR, (".h",)) + _all_sources(os.path.join(PKG, "rccl"), (".h",))
    if force or _newer(RCCL_LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", INC_DIR, src,
              "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", RCCL_LIB])
    return RCCL_LIB


# ---- build guard -----------------------------------------------------------------------------------------------------
# HIP_OPT_FLAGS holds unversioned LLVM switches that were chosen because they cut the expand kernel's register spills. A
# toolchain bump can silently change what they do; that must show at BUILD time, not weeks later in a bench regression.
# After every HIP build the register / spill / scratch / LDS figures are read from the code object's own metadata
# (tools/kernel_resources.py), written to profiles/kernel_resources_<lib>.json, and compared with the committed bounds
# below (measured figures of the round-5 build + a margin of a few registers). EG3D_NO_BUILD_GUARD=1 skips the check
# (experimental variants built with other switches).
RESOURCE_BOUNDS = {
    # kernel (substring of the demangled name): {metadata field: largest accepted value}
    # (round 6, late: the lane-group DLT, the short redo of a failed look-ahead step, the parallel candidate walks and the
    # solver's convergence pre-check each inline more code around the places where the chain's state is live — 23 / 157 /
    # 61 spilled registers became 101 / 341 / 107, all of them around calls, none inside a loop — and each was kept
    # because the kernel got faster: DESIGN_LOG.md round 6 item 8, profiles/r06_experiments/ab*.txt)
    "k3b_expand_t<4, 0, 0>": {"vgpr_spill_count": 120, "private_segment_fixed_size": 224, "group_segment_fixed_size": 10240},
    "k3b_expand_t<4, 0, 1>": {"vgpr_spill_count": 360, "private_segment_fixed_size": 400, "group_segment_fixed_size": 10240},
    "k3b_expand_t<4, 1, 2>": {"vgpr_spill_count": 140, "private_segment_fixed_size": 256, "group_segment_fixed_size": 10240},
    "k3a_orient": {"vgpr_spill_count": 0, "group_segment_fixed_size": 12800},
    "k3a_follow_spec": {"vgpr_spill_count": 0, "group_segment_fixed_size": 12800},
    "k5_gn_filter": {"vgpr_spill_count": 0},
    "k2_epipolar_hits": {"vgpr_spill_count": 0},
    "k1_seed_candidates": {"vgpr_spill_count": 0},
}

# ... and of the lane-per-chain engine, present only in builds made with -DEG3D_WITH_K3C_ENGINE (build_hip_engine)
ENGINE_RESOURCE_BOUNDS = {
    "k3c_engine_t<false>": {"vgpr_spill_count": 256, "group_segment_fixed_size": 10240},
    "k3c_engine_t<true>": {"vgpr_spill_count": 256, "group_segment_fixed_size": 10240},
}


def check_resources(lib, strict=True, record_as=None):
    """Reads the kernels' resource figures from `lib`, stores them under edgegraph3d_amd/_obj/ (untracked; with
    EG3D_RECORD_KERNEL_RESOURCES=1 also as the committed profiles/kernel_resources_<lib>.json), returns the list of
    violated bounds and raises if strict. `record_as` names the library the figures belong to when `lib` is the temporary
    file of a build in progress. EG3D_NO_BUILD_GUARD=1 skips everything, including the helper tools the reading needs."""
    if os.environ.get("EG3D_NO_BUILD_GUARD") == "1":
        print("  build guard: skipped (EG3D_NO_BUILD_GUARD=1)", flush=True)
        return []
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources as kr
    finally:
        sys.path.pop(0)
    try:
        res = {n: r for n, r in kr.kernel_resources(lib).items() if n.startswith("eg3d::")}
    except Exception as ex:  # objcopy / clang-offload-bundler / llvm-readelf missing, no gfx950 bundle
        raise RuntimeError("build guard: cannot read the kernel resources of %s (%r). The library is NOT installed; "
                           "EG3D_NO_BUILD_GUARD=1 builds without the check." % (lib, ex))
    name = os.path.splitext(os.path.basename(record_as or lib))[0]
    variant = os.sep + "variants" + os.sep in os.path.abspath(record_as or lib)
    outs = []
    if not variant:  # experimental variants are checked / printed, not recorded
        os.makedirs(OBJ_DIR, exist_ok=True)
        outs.append(os.path.join(OBJ_DIR, "kernel_resources_%s.json" % name))
        if os.environ.get("EG3D_RECORD_KERNEL_RESOURCES") == "1":
            outs.append(os.path.join(ROOT, "profiles", "kernel_resources_%s.json" % name))
    import json
    for o in outs:
        try:
            with open(o, "w") as f:
                json.dump(res, f, indent=1, sort_keys=True)
                f.write("\n")
        except OSError:
            pass  # (a read-only tree: the check below still runs)
    for n in sorted(res):
        if any(k in n for k in ("k3a_orient", "k3a_follow", "k3b_expand", "k3c_engine")):
            r = res[n]
            print("  %-28s vgpr %3d  spilled vgpr %3d  spilled sgpr %3d  scratch %4d B  lds %5d B" % (
                n.replace("eg3d::", ""), r["vgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"],
                r["private_segment_fixed_size"], r["group_segment_fixed_size"]), flush=True)
    bounds = dict(RESOURCE_BOUNDS)
    if any("k3c_engine" in n for n in res):
        bounds.update(ENGINE_RESOURCE_BOUNDS)
    bad = kr.check_bounds(res, bounds)
    if bad and strict:
        raise RuntimeError("build guard: kernel resources exceed the committed bounds (edgegraph3d_amd/build.py "
                           "RESOURCE_BOUNDS); the library is NOT installed:\n  " + "\n  ".join(bad))
    return bad


def device_source_fingerprint(defines=()):
    """sha256 over the device CODE (sources with comments and blank lines removed) and the switches it is compiled with:
    what a PMC profile under profiles/ was measured ON. tools/profile_summary.py records it beside the traffic figures;
    bench.py reports `roofline.traffic` only while it still matches (a changed kernel makes the committed figure stale:
    it is then reported as such, not as this run's traffic). Editing a comment does not change it. Of eg3d_api.hip (host
    orchestration: no kernel lives there) only the preprocessor lines count — the macros that choose the kernel build,
    its default form and the slot-pool sizing; EG3D_EXTRA_HIPFLAGS and the `defines` of the build are part of the switches."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in _all_sources(CSRC_DIR, (".h", ".hpp", ".hip")):
        text = open(f, "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
        text = re.sub(r"//[^\n]*", "", text)                     # line comments (no string of these sources holds "//")
        lines = [" ".join(l.split()) for l in text.splitlines() if l.strip()]
        if os.path.basename(f) == "eg3d_api.hip":
            lines = [l for l in lines if l.startswith("#") and not l.startswith("#include")]
        h.update(os.path.basename(f).encode())
        h.update("\n".join(lines).encode())
    h.update(" ".join(HIP_FLAGS + os.environ.get("EG3D_EXTRA_HIPFLAGS", "").split() + list(defines)).encode())
    return h.hexdigest()[:16]


# Environment switches of eg3d_api.hip's Tunables that change WHICH kernel build runs or how it is launched: with any of them
# set, a committed PMC profile does not describe the run (bench.py marks roofline.traffic stale).
KERNEL_CHOICE_ENV = ("EG3D_K3B_ENGINE", "EG3D_K3B_FULL", "EG3D_K3B_ASSUME_SHORT", "EG3D_SLOTS_PER_XCD", "EG3D_MAX_SCRATCH_MB",
                     "EG3D_NO_LPT", "EG3D_K3C_WAVES", "EG3D_K3C_LANES", "EG3D_LIB")


def kernel_choice_env_set():
    return [k for k in KERNEL_CHOICE_ENV if os.environ.get(k)]


def build_oracle(force=False):
    """Builds the CPU oracle (test infrastructure). Building the checker is not using it."""
    d = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "-C", d, "clean"])
    _run(["make", "-C", d])
    return os.path.join(d, "liboracle.so")


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_hip_dlt4x4(force)
    build_hip_engine(force)
    build_probe(force)
    build_rccl(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
