"""Host-side planning of the two big field kernels, queried through the C ABI without a GPU: the gather's per-XCD
segments (csrc/hashgrid.hip make_encode_plan) and the binned scatter's workspace / reduce layout (plan_for,
plan_reduce_splits)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "make-it-3d_amd"))

PLS = 1.3819128274917603          # 16 levels from 16 to 2048
STEP = 2 * 3 ** 0.5 / 1024        # C2's marching step


@pytest.fixture(scope="module")
def lib():
    from mi3d import _lib as L
    return L


@pytest.mark.parametrize("n,step", [(10_878_464, STEP), (64, STEP), (65, STEP), (1_000_003, 4 * STEP), (5_000_000, 0.0)])
def test_encode_segments_tile_every_level_exactly_once_and_balance(lib, n, step):
    nseg = (C.c_uint32 * 8)()
    seg = (C.c_uint32 * (8 * 16 * 3))()
    lib.call("mi3d_grid_encode_plan", n, 1.0, step, 16, 16, PLS, 19, nseg, seg)
    tiles = (n + 63) // 64
    covered = np.zeros((16, tiles), np.int32)
    s = np.ctypeslib.as_array(seg).reshape(8, 16, 3)
    order = []
    for x in range(8):
        assert nseg[x] <= 16
        for i in range(nseg[x]):
            l, t0, t1 = (int(v) for v in s[x, i])
            assert 0 <= l < 16 and 0 <= t0 < t1 <= tiles
            covered[l, t0:t1] += 1
            order.append((l, t0))
    # levels 0 and 1 (4096 + 12 167 entries = 130 KB) are served from LDS by a kernel of their own (csrc/hashgrid.hip
    # k_grid_encode_planes_lds) and are not in the XCD plan; every other (level, tile) belongs to exactly one XCD
    assert (covered[:2] == 0).all() and (covered[2:] == 1).all()
    assert order == sorted(order)                 # the XCDs walk the (level, tile) list in order: contiguous segments
    if n == 10_878_464 and step == STEP:
        # C2: the XCD that takes the cheap coarse levels takes several of them, the fine levels (3x the cost per tile) are
        # cut across XCDs - so the XCDs get very different tile counts for the same modelled cost
        levels_per_xcd = [len({int(s[x, i, 0]) for i in range(nseg[x])}) for x in range(8)]
        assert max(levels_per_xcd) <= 5 and min(levels_per_xcd) <= 2
        per_xcd_tiles = [sum(int(s[x, i, 2] - s[x, i, 1]) for i in range(nseg[x])) for x in range(8)]
        assert max(per_xcd_tiles) > 1.5 * min(per_xcd_tiles)


def _scatter_plan(lib, n, P, step, workspace):
    out = (C.c_ulonglong * (6 + 7 * 16))()
    lib.call("mi3d_grid_scatter_plan", n, P, 1.0, step, 16, 16, PLS, 19, C.c_size_t(workspace), out)
    head = dict(n_slice=int(out[0]), bytes=int(out[1]), merge=int(out[2]), wgs=int(out[3]), arena=int(out[4]),
                counters=int(out[5]))
    levels = [dict(zip(("bins", "cap", "waves", "row", "split", "wg0", "cnt0"), (int(out[6 + 7 * l + k]) for k in range(7))))
              for l in range(16)]
    return head, levels


def test_scatter_plan_c2(lib):
    n, P = 10_878_464, 13
    need = lib.lib().mi3d_grid_scatter_binned_workspace(n, P, 1.0, STEP, 16, 16, PLS, 19)
    head, lv = _scatter_plan(lib, n, P, STEP, need)
    assert head["n_slice"] == n and head["bytes"] == need            # the size the query promises holds ONE slice
    assert head["merge"] == 7                                        # cells of >= 4 marching steps: levels 0-6
    assert [l["bins"] for l in lv[:6]] == [1, 2, 4, 10, 26, 64] and all(l["bins"] == 64 for l in lv[5:])
    assert [l["row"] for l in lv] == [0] * 7 + [1] * 9               # x-pair records on the fine levels only
    # a workspace of 100 GiB: the samples are cut into the FEWEST equal slices that fit (fp32 planes, 16-byte records: two)
    head2, lv2 = _scatter_plan(lib, n, P, STEP, 100 << 30)
    assert head2["n_slice"] == (n + 1) // 2 and head2["bytes"] <= 100 << 30 < need
    # ... and not the fewest power of two (rounds 2-5 halved: where three slices fit, four ran - every slice has a fixed cost)
    for gib in (56, 40, 31, 24, 12):
        hk, _ = _scatter_plan(lib, n, P, STEP, gib << 30)
        k = -(-n // hk["n_slice"])
        assert hk["n_slice"] == -(-n // k) and hk["bytes"] <= gib << 30
        one_fewer, _ = _scatter_plan(lib, -(-n // (k - 1)), P, STEP, 1 << 50)
        assert one_fewer["bytes"] > gib << 30, (gib, k)
    assert -(-n // _scatter_plan(lib, n, P, STEP, 40 << 30)[0]["n_slice"]) == 3
    # a workspace too small for any slice: the plan of one tile comes back and does not fit (the call takes the atomic path)
    head0, _ = _scatter_plan(lib, n, P, STEP, 1 << 20)
    assert head0["n_slice"] <= 64 and head0["bytes"] > 1 << 20
    for l, r in enumerate(lv2):
        assert r["cap"] % 1 == 0 and r["cap"] >= 64
        assert 1 <= r["split"] <= max(1, r["waves"] // 16)            # every reduce wave gets at least one region
        assert r["wg0"] == sum(q["bins"] * q["split"] for q in lv2[:l])
        assert r["cnt0"] == sum(q["bins"] * q["waves"] for q in lv2[:l])
    assert head2["wgs"] == sum(r["bins"] * r["split"] for r in lv2)
    assert head2["counters"] == sum(r["bins"] * r["waves"] for r in lv2)
    # the average bin gets the base split (4 for a slice of this size); the fine bins carry the work
    assert all(r["split"] >= 4 for r in lv2[8:]) and 2500 <= head2["wgs"] <= 3800
    # region capacity of a fine level: the uniform share of 4 x-pair records per evaluation plus headroom
    share = head2["n_slice"] * P * 4 / (lv2[8]["waves"] * 64)
    assert share < lv2[8]["cap"] < 1.5 * share


def test_scatter_plan_small_pass_and_bad_arguments(lib):
    head, lv = _scatter_plan(lib, 4096, 1, STEP, 1 << 30)
    assert head["n_slice"] == 4096 and all(r["split"] == 1 for r in lv)   # a small pass: one workgroup per bin
    out = (C.c_ulonglong * (6 + 7 * 16))()
    with pytest.raises(lib.Mi3dError):
        lib.call("mi3d_grid_scatter_plan", 0, 13, 1.0, STEP, 16, 16, PLS, 19, C.c_size_t(1 << 30), out)
    with pytest.raises(lib.Mi3dError):
        lib.call("mi3d_grid_scatter_plan", 64, 17, 1.0, STEP, 16, 16, PLS, 19, C.c_size_t(1 << 30), out)


def test_level_routes_of_the_reference_configurations(lib, oracle):
    """Which index route each level takes (csrc/hashgrid.hip level_fast) against the oracle's level table: a level is on
    the dense route exactly when its res^3 entries fit its table (tcnn's rule for not hashing), on the masked-hash route
    when it is hashed into a power-of-two table, and raw positions (tcnn.Encoding's own entry points) never leave the
    general rule."""
    for n_levels, base, log2, pls in ((16, 16, 19, PLS), (4, 16, 19, PLS), (6, 8, 12, 1.5), (8, 16, 10, 2.0)):
        cfg = oracle.GridConfig(n_levels=n_levels, base_resolution=base, log2_hashmap_size=log2, per_level_scale=pls)
        kinds = (C.c_int32 * n_levels)()
        assert lib.call("mi3d_grid_level_routes", n_levels, base, cfg.per_level_scale, log2, 1, kinds) in (0, None)
        sizes = np.diff(cfg.offsets.astype(np.int64))
        for l in range(n_levels):
            res = int(cfg.resolutions[l])
            dense = res ** 3 <= int(sizes[l])
            want = 1 if dense else (2 if int(sizes[l]) & (int(sizes[l]) - 1) == 0 else 0)
            assert kinds[l] == want, (n_levels, base, log2, l, res, int(sizes[l]), kinds[l])
        raw = (C.c_int32 * n_levels)()
        lib.call("mi3d_grid_level_routes", n_levels, base, cfg.per_level_scale, log2, 0, raw)
        assert list(raw) == [0] * n_levels
    k16 = (C.c_int32 * 16)()
    lib.call("mi3d_grid_level_routes", 16, 16, PLS, 19, 1, k16)
    assert list(k16) == [1] * 5 + [2] * 11      # the default grid: levels 0-4 dense, 5-15 hashed into 2^19 entries
