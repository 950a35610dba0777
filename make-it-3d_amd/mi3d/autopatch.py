"""Opt-in import hook: `main.py` reaches the fused field with ZERO edits.

The reference picks its network class at /root/reference/main.py:101-106 - `from nerf.network_tcnn import NeRFNetwork` -
and that class (nerf/network_tcnn.py:37-206) evaluates the field as 13 separate encoder + torch-MLP passes per sample.
With the drop-in `tinycudann` / `raymarching` packages alone it runs unchanged but slowly (its 39 `nn.Linear` weight-gradient
GEMMs reduce over ~11 M rows each: 1.4 of its 1.8 s per C2 step are hipBLASLt, profiles/kernel_stats_r06_reference_shaped.csv).
`mi3d.network.NeRFNetwork` is the same class surface (constructor, parameters, state_dict keys, methods) on the fused
13-point field.  This module swaps the name WHEN `nerf.network_tcnn` IS IMPORTED, without touching the reference's files:

    import mi3d.autopatch                      # anywhere before main.py's line 104 runs, or - with no edit at all -
    PYTHONPATH=<repo>/make-it-3d_amd/autopatch:<repo>/make-it-3d_amd python main.py ...

(the first directory holds a `sitecustomize.py` that does the import; it chains to the interpreter's own sitecustomize).
It is explicit and reversible: nothing happens unless this module is imported; `uninstall()` removes the hook and puts the
reference's class back; the reference's class stays reachable as `nerf.network_tcnn.NeRFNetwork_reference`.  Only stdlib
imports at module level: a process that never imports `nerf.network_tcnn` pays nothing (no torch import)."""
import importlib.abc
import sys

TARGET = "nerf.network_tcnn"
_finder = None


def _patch(module):
    if getattr(module, "_mi3d_autopatched", False):
        return
    from mi3d.network import NeRFNetwork   # (lazy: torch and libmi3d.so load here, when the reference asks for its network)
    module.NeRFNetwork_reference = getattr(module, "NeRFNetwork", None)
    module.NeRFNetwork = NeRFNetwork
    module._mi3d_autopatched = True


class _Loader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec) if hasattr(self.inner, "create_module") else None

    def exec_module(self, module):
        self.inner.exec_module(module)
        _patch(module)

    def __getattr__(self, name):   # (get_source, get_filename, is_package ...: whatever the real loader offers)
        return getattr(self.inner, name)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname != TARGET:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, "find_spec"):
                continue
            spec = f.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None:
                spec.loader = _Loader(spec.loader)
                return spec
        return None


def install():
    """Idempotent.  A `nerf.network_tcnn` that is already imported is patched in place (names bound earlier with
    `from nerf.network_tcnn import NeRFNetwork` keep the reference's class: import this module first)."""
    global _finder
    if _finder is None:
        _finder = _Finder()
        sys.meta_path.insert(0, _finder)
    if TARGET in sys.modules:
        _patch(sys.modules[TARGET])


def uninstall():
    global _finder
    if _finder is not None:
        try:
            sys.meta_path.remove(_finder)
        except ValueError:
            pass
        _finder = None
    m = sys.modules.get(TARGET)
    if m is not None and getattr(m, "_mi3d_autopatched", False):
        if getattr(m, "NeRFNetwork_reference", None) is not None:
            m.NeRFNetwork = m.NeRFNetwork_reference
        m._mi3d_autopatched = False


def installed():
    return _finder is not None


install()
