"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mi3d.h declares.
(No compute calls - there is no GPU in the build container.)"""
import ctypes
import os
import re

from conftest import PKG, ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "mi3d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mi3d_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mi3d_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = b.build()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 19
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.mi3d_abi_version() == 5


def test_python_binding_covers_the_abi():
    from mi3d import _lib
    lib = _lib.lib()
    declared = set(_declared())
    bound = set(_lib._SIGNATURES) | {"mi3d_abi_version", "mi3d_last_error_string", "mi3d_hashgrid_levels",
                                     "mi3d_grid_scatter_binned_workspace", "mi3d_points_rasterize_workspace"}
    bound |= set(getattr(_lib, "_LATE_SIGNATURES", {}))
    assert declared <= bound, declared - bound
    for n in bound:
        assert hasattr(lib, n)


def test_level_table_host_side(oracle):
    import numpy as np
    import tinycudann as tcnn
    total, offs, res, scl = tcnn.grid_levels(16, 16, float(np.float32(1.3819128274917603)), 19)
    assert total == 6098120 and total * 2 == 12196240  # 48.8 MB of fp32
    assert list(res[:6]) == [16, 23, 31, 43, 59, 81] and res[-1] == 2048
    cfg = oracle.GridConfig()
    assert np.array_equal(offs, cfg.offsets) and np.array_equal(scl, cfg.scales)


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "liboracle" in s:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_scatter_workspace_query_is_host_only_and_sane():
    """mi3d_grid_scatter_binned_workspace is pure host arithmetic (no GPU): monotone in the sample count and in the
    number of stencil points, and at least the bytes of the records it must hold."""
    from mi3d import _lib
    ws = _lib.lib().mi3d_grid_scatter_binned_workspace
    args = (1.0, 2 * 3 ** 0.5 / 1024, 16, 16, 1.3819128274917603, 19)
    full, half_n, one_pt = ws(10_878_592, 13, *args), ws(5_439_296, 13, *args), ws(10_878_592, 1, *args)
    assert 0 < half_n < full and one_pt < full / 8
    # levels 8..15 emit 4 x-pair records of 16 bytes per (evaluation, level)
    assert full > 10_878_592 * 13 * 8 * 4 * 16
    assert full < 300e9 and ws(0, 13, *args) == 0


def test_product_library_reads_no_environment_and_has_no_dev_hooks():
    """Hygiene: libmi3d.so imports neither getenv nor the development tunables (those exist only in the -DMI3D_DEV
    build of tools/build_dev.py), and the kernels carry no debug sentinels."""
    import subprocess
    so = os.path.join(PKG, "csrc", "libmi3d.so")
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms and "mi3d_dev_" not in syms
    for f in ("hashgrid.hip", "field.hip", "raymarching.hip"):
        src = open(os.path.join(PKG, "csrc", f)).read()
        assert "getenv" not in src and "12345" not in src and "static bool" not in src, f
