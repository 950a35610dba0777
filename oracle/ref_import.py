"""ORACLE tooling (test infrastructure, NOT product code).

Imports the reference's *own* Python modules so the oracle and the product can be pinned against them:
  * in the build container from /root/reference (also used to generate tests/golden/*.npz, make_golden_py.py);
  * on the GPU box - where /root/reference does not exist - from oracle/_ref/py/, a git-ignored staging copy of
    the handful of reference files the path needs, made by oracle/build_ref.py:stage_py() and shipped by gpurun like
    oracle/_ref/_raymarching_ref.so.  Nothing is copied into the repository history.

The reference's import-time-only dependencies that are absent from the image (cv2, trimesh, open3d, mcubes,
imageio, tensorboardX, torchvision, pytorch3d, torch_ema, clip, torchmetrics, contextual_loss, ...) are replaced
by inert stub modules.  Two ways to give the reference its `tinycudann` / `raymarching`:
  encoder="oracle" : tcnn.Encoding = oracle.field_torch.HashGridTorch, a CPU torch restatement (tcnn is CUDA-only and
                     un-pinned: PARITY UNPINNED); `raymarching` is whatever imports (never called on this route);
  encoder="dropin" : the product's drop-in packages make-it-3d_amd/{tinycudann, raymarching} - the reference's
                     NeRFNetwork / NeRFRenderer.run_cuda / update_extra_state then run UNCHANGED on the HIP kernels
                     (the zero-change route).  tests/test_reference_glue_gpu.py compares
                     that against the product's own fast route; bench.py times it as the reference-shaped baseline.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGED = os.path.join(_HERE, "_ref", "py")
REFERENCE = "/root/reference" if os.path.isdir("/root/reference/nerf") else STAGED

_STUB_ROOTS = ("cv2", "trimesh", "open3d", "mcubes", "imageio", "tensorboardX", "torchvision", "pytorch3d",
               "torch_ema", "clip", "torchmetrics", "contextual_loss", "rich", "tqdm", "pandas", "matplotlib",
               "PIL", "raymarching", "xatlas", "nvdiffrast", "sklearn", "scipy", "diffusers", "transformers",
               "kornia", "lpips", "dearpygui", "timm", "skimage")


class _Anything:
    """Callable/attribute sink so `from x import y`, `x.y(...)`, decorators etc. all succeed."""

    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything(self._name)

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self._name + "." + k)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False
TCNN_ORACLE = None  # the stub `tinycudann` module whose Encoding is the CPU torch restatement


def available():
    return os.path.isdir(os.path.join(REFERENCE, "nerf"))


def install():
    """Make `import nerf.renderer`, `import nerf.network_tcnn`, `import activation` resolve to the reference."""
    global _installed, TCNN_ORACLE, _STUB_ROOTS
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference sources not found (neither /root/reference nor {STAGED}: run "
                           "oracle/build_ref.py in the build container)")
    sys.dont_write_bytecode = True  # never write .pyc into the read-only reference tree
    # real scipy / tqdm etc. are fine if importable; only stub what is missing.  `raymarching` resolves to the
    # product's drop-in package when make-it-3d_amd is on sys.path (tests/conftest.py, bench.py put it there).
    import importlib.util as iu
    keep = []
    for r in _STUB_ROOTS:
        if r in ("clip", "matplotlib", "pandas"):
            # always stubbed: `clip` would resolve to an unrelated package; nerf/utils.py imports matplotlib.pyplot and
            # pandas for plotting / logging only, and the real imports cost tens of seconds on a fresh box (matplotlib
            # builds its font cache: 20-30 s of bench.py's reference-shaped leg in round 5)
            if r not in sys.modules:
                keep.append(r)
            continue
        try:
            if iu.find_spec(r) is None:
                keep.append(r)
        except (ImportError, ValueError):
            keep.append(r)
    _STUB_ROOTS = tuple(keep)
    sys.meta_path.insert(0, _StubFinder())

    from oracle import field_torch, oracle as O

    tcnn = types.ModuleType("tinycudann_oracle")

    def Encoding(n_input_dims, encoding_config, dtype=None, seed=1337):
        assert n_input_dims == 3 and encoding_config["otype"] == "HashGrid"
        cfg = O.GridConfig(n_levels=encoding_config["n_levels"],
                           n_features_per_level=encoding_config["n_features_per_level"],
                           log2_hashmap_size=encoding_config["log2_hashmap_size"],
                           base_resolution=encoding_config["base_resolution"],
                           per_level_scale=encoding_config["per_level_scale"])
        return field_torch.HashGridTorch(cfg)

    tcnn.Encoding = Encoding
    TCNN_ORACLE = tcnn
    try:
        have_dropin = iu.find_spec("tinycudann") is not None
    except (ImportError, ValueError):
        have_dropin = False
    if not have_dropin:
        sys.modules["tinycudann"] = tcnn
    # the reference tree has its own raymarching/ package (CUDA JIT build): make sure the name is already bound - to
    # the product's drop-in if importable, else to a stub - before the reference root goes first on sys.path
    if "raymarching" not in sys.modules:
        try:
            importlib.import_module("raymarching")
        except Exception:
            _STUB_ROOTS = _STUB_ROOTS + ("raymarching",)
            importlib.import_module("raymarching")
    sys.path.insert(0, REFERENCE)
    _installed = True


def reference_network(opt, encoder="oracle", **kw):
    """The reference's own nerf.network_tcnn.NeRFNetwork(opt, **kw) with its `tcnn` bound to the CPU restatement
    ("oracle") or to the product's drop-in tinycudann ("dropin")."""
    install()
    import nerf.network_tcnn as ref_net
    if encoder == "oracle":
        ref_net.tcnn = TCNN_ORACLE
    elif encoder == "dropin":
        import raymarching
        import tinycudann
        import nerf.renderer as ref_renderer
        ref_net.tcnn = tinycudann
        ref_renderer.raymarching = raymarching
    else:
        raise ValueError(encoder)
    # (under mi3d.autopatch the module's `NeRFNetwork` is the fused class; the reference's own stays reachable)
    cls = getattr(ref_net, "NeRFNetwork_reference", None) or ref_net.NeRFNetwork
    return cls(opt, **kw)


def default_opt(**over):
    """The subset of main.py:18-92's argparse namespace the renderer/network read."""
    import argparse
    d = dict(bound=1.0, cuda_ray=False, min_near=0.1, density_thresh=10.0, bg_radius=-1, blob_density=5.0,
             blob_radius=0.1, lambda_smooth=0.0, max_depth=10.0, dt_gamma=0.0, max_steps=1024)
    d.update(over)
    return argparse.Namespace(**d)
