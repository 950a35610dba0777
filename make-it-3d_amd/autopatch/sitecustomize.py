"""Zero-edit switch for the reference's main.py:

    PYTHONPATH=<repo>/make-it-3d_amd/autopatch:<repo>/make-it-3d_amd python main.py --workspace ws --ref_path ...

Python imports `sitecustomize` at start-up from the first directory of sys.path that has one; this one installs
mi3d.autopatch - `from nerf.network_tcnn import NeRFNetwork` (main.py:101-106) then hands out the fused-field class - and
then runs the sitecustomize it shadows (the interpreter's own, e.g. /usr/lib/python3.10/sitecustomize.py), if any.  Opt-in
by construction: it only exists on a path the user put there."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.dirname(_here)
if _pkg not in sys.path:
    sys.path.insert(sys.path.index(_here) + 1 if _here in sys.path else 0, _pkg)
try:
    import mi3d.autopatch  # noqa: F401  (stdlib-only until nerf.network_tcnn is imported)
except Exception as _e:  # never break interpreter start-up
    sys.stderr.write(f"[mi3d.autopatch] not installed: {_e!r}\n")

# chain to the sitecustomize this file shadows
try:
    import importlib.machinery as _m
    import importlib.util as _u
    _spec = _m.PathFinder.find_spec("sitecustomize", [p for p in sys.path if os.path.abspath(p or ".") != _here])
    if _spec is not None and _spec.loader is not None:
        _mod = _u.module_from_spec(_spec)
        _spec.loader.exec_module(_mod)
except Exception:
    pass
