"""ORACLE tooling (test infrastructure, NOT product code).

Imports the reference's *own* Python modules from /root/reference in THIS container so the
oracle can be pinned against them and golden vectors generated (tests/golden/make_golden_py.py).
/root/reference does not exist on the GPU box: nothing that runs there may import this module.

The reference's import-time-only dependencies that are absent from the image (cv2, trimesh,
open3d, mcubes, imageio, tensorboardX, torchvision, pytorch3d, torch_ema, clip, torchmetrics,
contextual_loss, ...) are replaced by inert stub modules; `tinycudann` is replaced by a stub whose
`Encoding` is oracle.field_torch.HashGridTorch (tcnn is CUDA-only and un-pinned: PARITY UNPINNED),
and `raymarching` by an empty module (the CUDA extension cannot be built without nvcc).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE = "/root/reference"

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


def available():
    return os.path.isdir(os.path.join(REFERENCE, "nerf"))


def install():
    """Make `import nerf.renderer`, `import nerf.network_tcnn`, `import activation` resolve to the reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    sys.dont_write_bytecode = True  # never write .pyc into the read-only reference tree
    # real scipy / tqdm etc. are fine if importable; only stub what is missing
    import importlib.util as iu
    global _STUB_ROOTS
    keep = []
    for r in _STUB_ROOTS:
        if r in ("raymarching", "clip"):
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

    tcnn = types.ModuleType("tinycudann")

    def Encoding(n_input_dims, encoding_config, dtype=None, seed=1337):
        assert n_input_dims == 3 and encoding_config["otype"] == "HashGrid"
        cfg = O.GridConfig(n_levels=encoding_config["n_levels"],
                           n_features_per_level=encoding_config["n_features_per_level"],
                           log2_hashmap_size=encoding_config["log2_hashmap_size"],
                           base_resolution=encoding_config["base_resolution"],
                           per_level_scale=encoding_config["per_level_scale"])
        return field_torch.HashGridTorch(cfg)

    tcnn.Encoding = Encoding
    sys.modules["tinycudann"] = tcnn
    sys.path.insert(0, REFERENCE)
    _installed = True


def default_opt(**over):
    """The subset of main.py:18-92's argparse namespace the renderer/network read."""
    import argparse
    d = dict(bound=1.0, cuda_ray=False, min_near=0.1, density_thresh=10.0, bg_radius=-1, blob_density=5.0,
             blob_radius=0.1, lambda_smooth=0.0, max_depth=10.0, dt_gamma=0.0, max_steps=1024)
    d.update(over)
    return argparse.Namespace(**d)
