"""ORACLE tooling (test infrastructure, NOT product code).

Builds the reference's OWN raymarching kernels - /root/reference/raymarching/src/{raymarching.cu, raymarching.h,
bindings.cpp}, compiled from where they lie - for gfx950, so GPU tests can pin both the product and the C restatement
(oracle/raymarching_ref.c) against the reference itself (the .cu builds through torch's hipify with
-std=c++17).  Runs only where /root/reference exists (the build container); the GPU box uses the prebuilt
oracle/_ref/_raymarching_ref.so, which is git-ignored but travels with gpurun.

hipify writes its translated files NEXT to its inputs and the reference tree is read-only, so the three files are
staged in a temporary directory outside the repository; only the resulting shared object is kept.
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "_raymarching_ref.so")
SRC = "/root/reference/raymarching/src"
FILES = ("raymarching.cu", "raymarching.h", "bindings.cpp")
# The reference's own Python for the path (renderer + field + their helpers; optimizer.py for the Adan pin), staged
# verbatim into the git-ignored oracle/_ref/py/ so the GPU box - which has no /root/reference - can run the reference's
# NeRFNetwork / NeRFRenderer on the drop-in packages (oracle/ref_import.py).  Never committed, never imported by
# the product.
PY_ROOT = "/root/reference"
PY_FILES = ("nerf/renderer.py", "nerf/network_tcnn.py", "nerf/utils.py", "nerf/refine_utils.py", "nerf/unet.py",
            "nerf/provider.py", "activation.py", "encoding.py", "optimizer.py")
PY_OUT = os.path.join(OUT_DIR, "py")


def stage_py(force=False):
    if not os.path.isdir(PY_ROOT):
        raise RuntimeError(f"{PY_ROOT} not present (the reference tree exists only in the build container)")
    for rel in PY_FILES:
        src, dst = os.path.join(PY_ROOT, rel), os.path.join(PY_OUT, rel)
        if os.path.exists(dst) and not force and os.path.getmtime(dst) >= os.path.getmtime(src):
            continue
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
    return PY_OUT



def build(force=False):
    if not os.path.isdir(SRC):
        raise RuntimeError(f"{SRC} not present (the reference tree exists only in the build container)")
    stage_py(force)
    if os.path.exists(OUT) and not force:
        return OUT
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    sys.dont_write_bytecode = True
    from torch.utils import cpp_extension
    work = tempfile.mkdtemp(prefix="mi3d_ref_")
    try:
        for f in FILES:
            shutil.copy(os.path.join(SRC, f), os.path.join(work, f))
        bdir = os.path.join(work, "build")
        os.makedirs(bdir)
        cpp_extension.load(name="_raymarching_ref", sources=[os.path.join(work, "raymarching.cu"), os.path.join(work, "bindings.cpp")],
                           extra_cflags=["-O3", "-std=c++17"], extra_cuda_cflags=["-O3", "-std=c++17"],
                           build_directory=bdir, is_python_module=False, verbose=False)
        so = glob.glob(os.path.join(bdir, "_raymarching_ref*.so"))
        if not so:
            raise RuntimeError("extension build produced no shared object")
        os.makedirs(OUT_DIR, exist_ok=True)
        shutil.copy(so[0], OUT)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return OUT


def load():
    """Import the prebuilt module (GPU box or build container); None if it was never built."""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("_raymarching_ref", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
