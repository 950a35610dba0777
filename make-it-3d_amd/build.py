"""Builds make-it-3d_amd/csrc/libmi3d.so for gfx950 with hipcc (cross-compiles without a GPU).

    python make-it-3d_amd/build.py [--force]

No torch, no pybind: the library exposes only the C ABI of include/mi3d.h.
raymarching.hip is compiled with -ffp-contract=off (its fused multiply-adds are explicit, see
csrc/mi3d_common.h); the other translation units use the default contraction.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libmi3d.so")
ARCH = "gfx950"

UNITS = [
    ("raymarching.hip", ["-ffp-contract=off"]),
    ("hashgrid.hip", []),
    ("field.hip", []),
    ("optim.hip", ["-ffp-contract=off"]),
    ("raster.hip", ["-ffp-contract=off"]),
]
HEADERS = ["mi3d_common.h", "mi3d_grid.h", "mi3d_dev.h", "lds_transpose.h", os.path.join("..", "..", "include", "mi3d.h")]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False, out=None, defines=()):
    """`out` / `defines`: a product-grade VARIANT library (same flags plus -D options) beside libmi3d.so, for an A/B of two
    product builds on the GPU box; its objects live in their own directory.  Without them: libmi3d.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = "".join(c for c in "".join(defines) if c.isalnum())
    objdir = os.path.join(CSRC, "build" + ("_" + tag if tag else ""))
    OUT = out or globals()["OUT"]
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, rebuilt = [], False
    for name, extra in UNITS:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs if os.path.exists(h)):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", *defines, *extra, "-c",
                   src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(OUT):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", OUT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
