import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "make-it-3d_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def make_rays(rng, N, bound=1.0):
    """Random camera rays aimed at the scene box, plus a few hand-placed corner cases."""
    o = rng.normal(size=(N, 3)).astype(np.float32)
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o *= rng.uniform(1.1, 2.5, (N, 1)).astype(np.float32) * bound
    tgt = rng.uniform(-0.7, 0.7, (N, 3)).astype(np.float32) * bound
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o, d = o.astype(np.float32), d.astype(np.float32)
    if N >= 4:
        o[0], d[0] = [0.1, 0.2, bound * 1.5], [0, 0, -1]      # axis-parallel: infinite reciprocals
        o[1], d[1] = [bound * 3, bound * 3, bound * 3], [1, 0, 0]  # misses the box
        o[2], d[2] = [0.0, 0.0, 0.0], [0.6, 0.0, 0.8]         # origin inside the box
        o[3], d[3] = [0.05, bound * 1.2, 0.0], [0, -1, 0]
    return o, d


def sphere_bitfield(O, C, H, radius):
    """Analytic occupancy: voxel occupied iff its centre lies within `radius` (cascade c spans [-2^c, 2^c])."""
    co = np.stack(np.meshgrid(*[np.arange(H)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    idx = O.morton3D(co)
    grid = np.zeros((C, H ** 3), np.float32)
    for c in range(C):
        b = min(2 ** c, 2 ** (C - 1))
        ctr = ((co + 0.5) / H * 2 - 1) * b
        grid[c, idx] = np.linalg.norm(ctr, axis=1) < radius
    return O.packbits(grid, 0.5)


def random_bitfield(rng, C, H, density=0.125):
    n = C * H ** 3 // 8
    b = np.full(n, 255, np.uint8)
    k = max(1, int(round(-np.log2(density))))
    for _ in range(k):
        b &= rng.integers(0, 256, n).astype(np.uint8)
    return b


import contextlib


@contextlib.contextmanager
def record_scatter_workspaces():
    """Bytes of record arena every binned scatter of the product was handed, in call order: > 0 = the record path ran
    (k_bin_emit + k_bin_reduce); 0 = the all-atomic fallback.  Tests of the record path assert on it: a missing arena
    would otherwise pass through float atomics without touching the code under test."""
    from mi3d import field_ops
    seen, orig = [], field_ops.scatter_workspace

    def spy(device, needed, cap=None):
        ws = orig(device, needed, cap)
        seen.append(0 if ws is None else ws.numel())
        return ws
    field_ops.scatter_workspace = spy
    try:
        yield seen
    finally:
        field_ops.scatter_workspace = orig


@contextlib.contextmanager
def position_jitter():
    """torch.randn_like(x) for x [m, 3] -> a unit-variance pseudo-random function of x ITSELF (elementwise torch ops on
    bit-identical inputs: bit-identical outputs, whatever the row order).  The smoothness jitter of run_cuda
    (renderer.py:522) is drawn per ROW and the marching waves' slabs arrive in a different order in every run: under this
    patch every sample keeps its jitter whatever row it lands in, so two runs of a render compare term by term."""
    import math
    import torch
    orig = torch.randn_like

    def fake(x, *a, **k):
        if x.dim() == 2 and x.shape[-1] == 3 and x.is_floating_point() and not a and not k:
            p = x.detach().double()
            f64 = dict(dtype=torch.float64, device=x.device)
            s = torch.stack([p @ torch.tensor([12.9898, 78.233, 37.719], **f64),
                             p @ torch.tensor([39.3468, 11.135, 83.155], **f64),
                             p @ torch.tensor([73.156, 52.235, 9.151], **f64)], -1)
            u = torch.frac(torch.sin(s) * 43758.5453123).abs()          # [0, 1)
            return ((u - 0.5) * math.sqrt(12.0)).to(x.dtype)            # unit variance
        return orig(x, *a, **k)
    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig
