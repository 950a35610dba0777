"""CPU: the product's own scalar math (make-it-3d_amd/csrc/mi3d_common.h - the DDA step, Morton codes and hash-grid
indexing the HIP kernels execute) compiled for the HOST and compared bit-for-bit with the oracle, so that index-exactness
is established before a GPU is involved.  tests/host_math/host_math.cpp is the thin extern-C shim."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import PKG, ROOT, make_rays, random_bitfield, sphere_bitfield

HERE = os.path.join(ROOT, "tests", "host_math")


@pytest.fixture(scope="module")
def hm():
    so = os.path.join(HERE, "libhostmath.so")
    src = os.path.join(HERE, "host_math.cpp")
    hdr = os.path.join(PKG, "csrc", "mi3d_common.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                               "-I", os.path.join(PKG, "csrc"), src, "-o", so])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("C_,bound,kind,dt_gamma,max_steps,perturb", [
    (1, 1.0, "ones", 0.0, 256, False), (1, 1.0, "sphere", 0.0, 512, True), (1, 1.0, "random", 1 / 128, 256, True),
    (2, 2.0, "sphere", 0.0, 256, True), (3, 4.0, "random", 1 / 128, 128, False)])
def test_march_bit_exact_with_oracle(hm, oracle, C_, bound, kind, dt_gamma, max_steps, perturb):
    rng = np.random.default_rng(11)
    N, H = 500, 128
    o, d = make_rays(rng, N, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb)
    bits = {"ones": lambda: np.full(C_ * H ** 3 // 8, 255, np.uint8), "sphere": lambda: sphere_bitfield(oracle, C_, H, 0.45 * bound),
            "random": lambda: random_bitfield(rng, C_, H)}[kind]()
    noises = rng.uniform(0, 1, N).astype(np.float32) if perturb else None
    xr, dr, lr, rr, cr = oracle.march_rays_train(o, d, bound, bits, C_, H, nears, fars, noises, align=-1,
                                                 dt_gamma=dt_gamma, max_steps=max_steps, return_counter=True)
    M = N * max_steps
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays, counter = np.zeros((N, 3), np.int32), np.zeros(2, np.int32)
    nz = np.zeros(N, np.float32) if noises is None else noises
    hm.hm_march_train(_p(o), _p(d), _p(bits), C.c_float(bound), C.c_float(dt_gamma), C.c_uint32(max_steps),
                      C.c_uint32(N), C.c_uint32(C_), C.c_uint32(H), C.c_uint32(M), _p(nears), _p(fars), _p(xyzs),
                      _p(dirs), _p(deltas), _p(rays), _p(counter), _p(nz))
    assert np.array_equal(counter, cr)
    # the oracle lays rays out in ray order too (sequential host loop): everything must agree bit for bit
    assert np.array_equal(rays, rr)
    m = int(counter[0])
    assert np.array_equal(xyzs[:m].view(np.uint32), xr[:m].view(np.uint32))
    assert np.array_equal(deltas[:m].view(np.uint32), lr[:m].view(np.uint32))
    assert np.array_equal(dirs[:m].view(np.uint32), dr[:m].view(np.uint32))


def test_morton_bit_exact_with_oracle(hm, oracle):
    rng = np.random.default_rng(2)
    co = rng.integers(0, 1024, (20000, 3)).astype(np.int32)
    out = np.empty(20000, np.int32)
    hm.hm_morton(_p(co), C.c_uint32(20000), _p(out))
    assert np.array_equal(out, oracle.morton3D(co))
    back = np.empty((20000, 3), np.int32)
    hm.hm_morton_invert(_p(out), C.c_uint32(20000), _p(back))
    assert np.array_equal(back, co)


@pytest.mark.parametrize("kw", [{}, dict(n_levels=4, per_level_scale=128 ** (1 / 3)),
                                dict(n_levels=6, log2_hashmap_size=12, base_resolution=4, per_level_scale=1.7)])
def test_grid_corner_indices_bit_exact_with_oracle(hm, oracle, kw):
    import tinycudann as tcnn
    cfg = oracle.GridConfig(**kw)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    x[0] = 0.0; x[1] = 1.0; x[2] = [1.0, 0.0, 0.999999]
    idx_ref, w_ref = oracle.hashgrid_indices(x, cfg)
    total, offs, res, scl = tcnn.grid_levels(cfg.n_levels, cfg.base_resolution, cfg.per_level_scale,
                                             cfg.log2_hashmap_size)
    assert total == cfg.n_entries
    for l in range(cfg.n_levels):
        size = int(offs[l + 1] - offs[l])
        stride, dims = 1, 0
        while dims < 3 and stride <= size:
            stride *= int(res[l]); dims += 1
        hashed = 1 if size < stride else 0
        idx, w = np.empty((3000, 8), np.uint32), np.empty((3000, 8), np.float32)
        hm.hm_grid_corners(_p(x), C.c_uint32(3000), C.c_float(float(scl[l])), C.c_uint32(int(res[l])),
                           C.c_uint32(int(offs[l])), C.c_uint32(size), C.c_uint32(hashed), C.c_uint32(dims), _p(idx),
                           _p(w))
        ref_local = idx_ref[:, l, :] - (0 if idx_ref[:, l, :].max() < size else offs[l])
        assert np.array_equal(idx, ref_local.astype(np.uint32)), l
        assert np.array_equal(w.view(np.uint32), w_ref[:, l, :].view(np.uint32)), l


def test_row12_record_packing_and_the_x_pair_rule(hm):
    """The scatter's 12-byte pair record under autocast (csrc/mi3d_common.h: pack_row12 / unpack_row12_fields, used by
    k_bin_emit / k_bin_reduce): the entry's 13 bits, the flip count and the raw binary16 pair survive exactly, the weight and
    the x fraction come back within 2^-24 (2^-23 at exactly 1, the clamp) - and the rule the record relies on: on a hashed
    power-of-two level the entry of the x + 1 corner is the x corner's with its t low bits flipped, t = 1 + trailing ones of
    cx (x enters the hash with prime 1)."""
    rng = np.random.default_rng(3)
    n = 200_000
    e = rng.integers(0, 8192, n).astype(np.uint32)
    t = rng.integers(0, 14, n).astype(np.uint32)
    w = rng.uniform(0, 1, n).astype(np.float32)
    fx = rng.uniform(0, 1, n).astype(np.float32)
    w[:6] = [0.0, 1.0, 2.0 ** -24, 1 - 2.0 ** -24, 0.5, 2.0 ** -23]
    fx[:6] = [0.0, 2.0 ** -25, 1 - 2.0 ** -24, 0.25, 2.0 ** -23, 0.75]
    raw = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    words = np.empty((n, 3), np.uint32)
    e2, t2 = np.empty(n, np.uint32), np.empty(n, np.uint32)
    w2, fx2 = np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_row12_roundtrip(_p(e), _p(t), _p(w), _p(fx), _p(raw), C.c_uint32(n), _p(words), _p(e2), _p(t2), _p(w2), _p(fx2))
    assert np.array_equal(e2, e) and np.array_equal(t2, t) and np.array_equal(words[:, 1], raw)
    assert np.abs(w2.astype(np.float64) - w).max() <= 2.0 ** -23 and np.abs(w2.astype(np.float64) - w)[w < 1].max() <= 2.0 ** -24 + 1e-12
    assert np.abs(fx2.astype(np.float64) - fx).max() <= 2.0 ** -24 + 1e-12
    assert w2[0] == 0.0 and fx2[0] == 0.0 and fx2[1] == 0.0          # fx_q == 0 marks a single: 2^-25 rounds to it
    assert (w2 < 1.0).all() and (fx2 < 1.0).all()
    # the x-pair rule, for cells of every trailing-ones pattern, on the levels the reference hashes into 2^19 (and a 2^12)
    cells = rng.integers(0, 2048, (100_000, 3)).astype(np.uint32)
    cells[:4096, 0] = np.arange(4096)            # every low-bit pattern of cx
    for size in (1 << 19, 1 << 12):
        e0, e1, rule = (np.empty(100_000, np.uint32) for _ in range(3))
        hm.hm_pair_rule(_p(cells), C.c_uint32(100_000), C.c_uint32(size), _p(e0), _p(e1), _p(rule))
        assert np.array_equal(e1, rule), size
        same_bin = (e0 >> 13) == (e1 >> 13)      # what the emit sends as ONE record (the others leave as two singles)
        assert 0.99 < same_bin.mean() <= 1.0


def test_lds_tile_transposition_offsets(hm):
    """The MLP backward turns its 32 x 32 binary16 tiles round through LDS (csrc/lds_transpose.h, offsets in
    csrc/mi3d_common.h).  Played through on the host: every lane writes its four 8-byte chunks at its write offset, the
    transposing read's exchange (lane t of a 16-lane group, value j <- value t & 3 of what lane 4 j + (t >> 2) of the group
    loaded; tools/tr_probe.hip checks that pattern on the chip) is applied to what every lane loads at its read offset, and
    lane = feature must end up with value q = sample rowmap(q, h') - for both value orders; a half-wave's writes spread over
    all 32 LDS banks (two lanes per bank pair: the rate of a 256-byte-per-instruction write anyway)."""
    wr_d, wr_x, rd = (np.empty(64, np.uint32) for _ in range(3))
    row = C.c_int32(0)
    hm.hm_tr_offsets(_p(wr_d), _p(wr_x), _p(rd), C.byref(row))
    row = row.value
    assert row % 8 == 0 and row >= 64           # 8-byte aligned chunks (the read returns wrong data off alignment)
    rowmap = lambda q, h: (q & 3) + 8 * (q >> 2) + 4 * h
    for kind, wr, step in (("D", wr_d, 16), ("X", wr_x, 8)):
        img = np.full(32 * row // 2, -1, np.int64)  # halfwords; the value = 100 sample + feature
        for lane in range(64):
            s, h = lane & 31, lane >> 5
            for c in range(4):
                at = int(wr[lane]) + step * c
                assert at % 8 == 0
                for e in range(4):
                    q = 4 * c + e
                    assert img[at // 2 + e] == -1   # nobody else's chunk
                    img[at // 2 + e] = 100 * s + (rowmap(q, h) if kind == "D" else 16 * h + q)
        for c in range(4):
            loaded = np.stack([img[(int(rd[lane]) + 8 * row * c) // 2:][:4] for lane in range(64)])   # [lane][4]
            for lane in range(64):
                g, t, f, hp = lane >> 4, lane & 15, lane & 31, lane >> 5
                for j in range(4):
                    got = loaded[16 * g + 4 * j + (t >> 2)][t & 3]
                    assert got == 100 * rowmap(4 * c + j, hp) + f, (kind, lane, c, j)
        # bank spread of one write instruction (32 banks of 4 bytes, a half-wave at a time)
        for half in (0, 1):
            banks = np.concatenate([((wr[32 * half:32 * half + 32].astype(np.int64) // 4) + d) % 32 for d in (0, 1)])
            assert np.bincount(banks, minlength=32).max() == 2
