"""Development build: the product sources compiled with -DMI3D_DEV into tools/bin/libmi3d_dev.so (git-ignored,
travels with gpurun).  The extra entry point mi3d_dev_set(index, value) overrides the tunables of csrc/mi3d_dev.h so
tools/kbench.py can A/B kernel variants and launch geometries in ONE GPU call.  Never loaded by the product: only a
process that sets MI3D_LIB to this file sees it.
    python tools/build_dev.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "make-it-3d_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tools", "bin")
OUT = os.path.join(OUT_DIR, "libmi3d_dev.so")
UNITS = [("raymarching.hip", ["-ffp-contract=off"]), ("hashgrid.hip", []), ("field.hip", []),
         ("optim.hip", ["-ffp-contract=off"]), ("raster.hip", ["-ffp-contract=off"])]
DEV_UNIT = """
extern "C" int mi3d_dev_tunable[32] = {-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1};
extern "C" void mi3d_dev_set(int i, int v) { if (i >= 0 && i < 32) mi3d_dev_tunable[i] = v; }
"""


def build(out=OUT, defines=()):
    """`defines`: extra -D options, e.g. ("-DMI3D_REDUCE_U=16",) -> a variant library for an A/B across two processes
    (python tools/build_dev.py tools/bin/libmi3d_dev_u16.so -DMI3D_REDUCE_U=16)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    dev_src = os.path.join(OUT_DIR, "dev_tunables.cpp")
    open(dev_src, "w").write(DEV_UNIT)
    for name, extra in UNITS + [(dev_src, [])]:
        src = name if os.path.isabs(name) else os.path.join(CSRC, name)
        tag = "" if not defines else "_" + "".join(c for c in "".join(defines) if c.isalnum())
        obj = os.path.join(OUT_DIR, os.path.basename(src).rsplit(".", 1)[0] + tag + ".o")
        lang = ["-x", "hip"] if src.endswith(".cpp") else []
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DMI3D_DEV", *defines, *extra,
                               *lang, "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(build(os.path.abspath(sys.argv[1]), tuple(sys.argv[2:])))
    else:
        print(build())
