"""CPU: the opt-in import hook that lets the reference's main.py reach the fused field with zero edits
(make-it-3d_amd/mi3d/autopatch.py, make-it-3d_amd/autopatch/sitecustomize.py).

`main.py:101-106` binds its network class with `from nerf.network_tcnn import NeRFNetwork`.  A fresh interpreter whose
PYTHONPATH starts with <repo>/make-it-3d_amd/autopatch imports the staged reference module that way and must get
mi3d.network.NeRFNetwork - same constructor, same state_dict keys - with the reference's own class still reachable."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG, ROOT


def _run(script, autopatch=True):
    env = dict(os.environ)
    path = ([os.path.join(PKG, "autopatch")] if autopatch else []) + [PKG, ROOT]
    env["PYTHONPATH"] = os.pathsep.join(path)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(script)], env=env, capture_output=True, text=True, timeout=300)


def _have_reference():
    from oracle import ref_import
    return ref_import.available()


def test_sitecustomize_installs_the_hook_without_importing_torch():
    r = _run("""
        import sys
        assert "mi3d.autopatch" in sys.modules, sorted(m for m in sys.modules if "mi3d" in m)
        assert "torch" not in sys.modules          # a process that never imports nerf.network_tcnn pays nothing
        import mi3d.autopatch as ap
        assert ap.installed() and any(type(f).__name__ == "_Finder" for f in sys.meta_path)
        # the interpreter's own sitecustomize still ran if there is one (chained)
        import importlib.machinery as M, os
        here = os.path.dirname(sys.modules["sitecustomize"].__file__)
        other = M.PathFinder.find_spec("sitecustomize", [p for p in sys.path if os.path.abspath(p or ".") != here])
        print("chained" if other is not None else "nothing to chain")
    """)
    assert r.returncode == 0, r.stderr


def test_without_the_directory_nothing_is_installed():
    r = _run("""
        import sys
        assert "mi3d.autopatch" not in sys.modules and "mi3d" not in sys.modules
    """, autopatch=False)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not _have_reference(), reason="reference sources not staged (oracle/build_ref.py)")
def test_main_py_import_line_gets_the_fused_class():
    r = _run("""
        import sys
        from oracle import ref_import
        ref_import.install()                       # the reference tree on sys.path, its import-time-only dependencies stubbed
        from nerf.network_tcnn import NeRFNetwork  # main.py:104, verbatim
        import mi3d.network
        import nerf.network_tcnn as m
        assert NeRFNetwork is mi3d.network.NeRFNetwork, NeRFNetwork
        ref_cls = m.NeRFNetwork_reference
        assert ref_cls.__module__ == "nerf.network_tcnn" and ref_cls is not NeRFNetwork
        opt = ref_import.default_opt(cuda_ray=True)
        ours = NeRFNetwork(opt)                    # main.py:217 `model = NeRFNetwork(opt)`
        theirs = ref_cls(opt)
        assert list(ours.state_dict()) == list(theirs.state_dict()), (list(ours.state_dict()), list(theirs.state_dict()))
        assert [tuple(v.shape) for v in ours.state_dict().values()] == [tuple(v.shape) for v in theirs.state_dict().values()]
        for name in ("render", "density", "forward", "normal", "update_extra_state", "get_params", "export_mesh"):
            assert hasattr(ours, name), name
        groups = ours.get_params(5e-3)
        assert [g["lr"] for g in groups] == [g["lr"] for g in theirs.get_params(5e-3)]
        import mi3d.autopatch as ap
        ap.uninstall()
        assert m.NeRFNetwork is ref_cls and not ap.installed()
        ap.install()
        assert m.NeRFNetwork is mi3d.network.NeRFNetwork
        print("ok")
    """)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]
