#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of the gfx950 kernels in a built library, read from the code object's own
metadata (the AMDGPU note: .vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size,
.group_segment_fixed_size). `python tools/kernel_resources.py [lib.so] [--json out]`; edgegraph3d_amd/build.py uses
kernel_resources() + check_bounds() as a BUILD GUARD: the optimisation switches of the device code are unversioned LLVM
options, and a toolchain bump that changes what they do shows up here as spills, not weeks later as a bench regression."""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def _demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            pass
    return {n: n for n in names}


def kernel_resources(lib):
    """{demangled kernel name: {field: int}} for every kernel of the gfx950 code object in `lib`."""
    notes = ""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"  # one bundle per translation unit, back to back in the section
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, a in enumerate(starts):
            part, co = os.path.join(d, "part%d.bin" % i), os.path.join(d, "dev%d.co" % i)
            with open(part, "wb") as f:
                f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                                    check=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:  # first key of a kernel entry (keys are sorted)
            cur = {".agpr_count": int(m.group(1))}
            continue
        if cur is None:
            continue
        m = re.match(r"\s+(\.[a-z_]+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == ".name":
            kernels[v.strip("'\"")] = cur
        elif k in FIELDS:
            cur[k] = int(v)
    dm = _demangle(list(kernels))
    out = {}
    for n, r in kernels.items():
        name = re.sub(r"^void ", "", dm[n])
        name = re.sub(r"\(.*$", "", name)
        out[name] = {f.lstrip("."): r.get(f, 0) for f in FIELDS}
    return out


def check_bounds(res, bounds):
    """bounds: {kernel-name substring: {field: max}}; returns the list of violations (strings)."""
    bad = []
    for pat, lim in bounds.items():
        hits = [n for n in res if pat in n]
        if not hits:
            bad.append("no kernel matches %r" % pat)
        for n in hits:
            for f, mx in lim.items():
                if res[n].get(f, 0) > mx:
                    bad.append("%s: %s = %d > %d" % (n, f, res[n][f], mx))
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = args[0] if args else os.path.join(root, "edgegraph3d_amd", "libeg3d.so")
    res = kernel_resources(lib)
    if "--all" not in sys.argv:  # the project's own kernels (the library kernels of rocPRIM / hipCUB have names of a page each)
        res = {n: r for n, r in res.items() if n.startswith("eg3d::")}
    w = max(len(n) for n in res)
    print("%-*s %5s %5s %7s %7s %8s %6s" % (w, "kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for n in sorted(res):
        r = res[n]
        print("%-*s %5d %5d %7d %7d %8d %6d" % (w, n, r["vgpr_count"], r["sgpr_count"], r["vgpr_spill_count"],
                                               r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"]))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
