"""ORACLE (test infrastructure, NOT product code): the refine stage's point renderer, restated in numpy.

Follows /root/reference/nerf/refine_utils.py:306-333 `render_point`:
    project with world2cam and K, divide by depth, map pixels to NDC and flip x / y (:307-316)
    -> pytorch3d.renderer.points.rasterize_points(pointcloud, image_size, radius, points_per_pixel)   (:319-320)
    -> alphas = 1 - sqrt(clamp(0.1 * dist / radius^2, 1e-3, 1))                                       (:321-326)
    -> pytorch3d.renderer.compositing.alpha_composite(idx, alphas, features)                          (:328-332)

PARITY UNPINNED for the two pytorch3d calls: pytorch3d is an un-vendored, un-pinned dependency of the reference
(requirements.txt names no version) and is absent from this image, so `rasterize_points` / `alpha_composite` are
restated from the published algorithm (pytorch3d/csrc/rasterize_points/rasterize_points.cu `RasterizePointsNaiveCudaKernel`,
pytorch3d/csrc/compositing/alpha_composite.cu), as remembered:
  * pixel (yi, xi) of the output looks at NDC (PixToNonSquareNdc(W-1-xi, W, H), PixToNonSquareNdc(H-1-yi, H, W)) - +X
    left, +Y up;
  * a point with z < 0 is skipped; it covers the pixel when dx^2 + dy^2 < radius^2 (strict);
  * the K nearest-in-z covering points are kept, ascending z (ties: lower point index first, the visiting order of the
    naive kernel); idx = point index, dists = dx^2 + dy^2, unused slots -1;
  * alpha_composite: front to back, out[c] += T * alpha_k * feat[c, idx_k], T *= 1 - alpha_k, slots with idx < 0 skipped.
The projection and the alpha formula are first-party reference code and are restated line by line.
"""
import numpy as np


def pix_to_ndc(i, S1, S2):
    """pytorch3d rasterization_utils PixToNonSquareNdc: centre of pixel i along an axis of S1 pixels (other axis S2)."""
    f = np.float32
    rng = f(2.0)
    if S1 > S2:
        rng = f(f(S1) * rng) / f(S2)
    off = f(rng / f(2.0))
    return (-off + (rng * np.asarray(i, np.float32) + off) / f(S1)).astype(np.float32)   # fp32, as the CUDA source


def project_points(points, K, world2cam, H, W):
    """refine_utils.py:307-316 -> [P,3] float32 (x_ndc, y_ndc, depth), fp32 arithmetic as torch does it."""
    p = points.astype(np.float32) @ world2cam[:3, :3].T.astype(np.float32) + world2cam[:3, 3].astype(np.float32)
    p = (p @ K.T.astype(np.float32)).astype(np.float32)
    out = p.copy()
    out[:, 0:2] = p[:, 0:2] / p[:, 2:3]
    out[:, 0] = out[:, 0] / np.float32(W) * np.float32(2) - np.float32(1.0)
    out[:, 1] = out[:, 1] / np.float32(H) * np.float32(2) - np.float32(1.0)
    out[:, 0] *= np.float32(-1)
    out[:, 1] *= np.float32(-1)
    return out.astype(np.float32)


def rasterize_points(ndc, H, W, radius, K):
    """idx int32 [H,W,K], zbuf [H,W,K], dists [H,W,K] (fp32; -1 where unused) - brute force, small inputs only."""
    ndc = ndc.astype(np.float32)
    P = ndc.shape[0]
    xs = pix_to_ndc(W - 1 - np.arange(W), W, H).astype(np.float32)
    ys = pix_to_ndc(H - 1 - np.arange(H), H, W).astype(np.float32)
    idx = -np.ones((H, W, K), np.int32)
    zbuf = -np.ones((H, W, K), np.float32)
    dists = -np.ones((H, W, K), np.float32)
    r2 = np.float32(radius) * np.float32(radius)
    order = np.lexsort((np.arange(P), ndc[:, 2]))  # by z, then by index
    px, py, pz = ndc[order, 0], ndc[order, 1], ndc[order, 2]
    ok = pz >= 0
    for yi in range(H):
        dy = (ys[yi] - py).astype(np.float32)
        near_y = ok & (dy * dy < r2)
        cand = np.flatnonzero(near_y)
        if cand.size == 0:
            continue
        for xi in range(W):
            dx = (xs[xi] - px[cand]).astype(np.float32)
            d2 = (dx * dx + dy[cand] * dy[cand]).astype(np.float32)
            hit = np.flatnonzero(d2 < r2)[:K]
            k = hit.size
            if k:
                idx[yi, xi, :k] = order[cand[hit]]
                zbuf[yi, xi, :k] = pz[cand[hit]]
                dists[yi, xi, :k] = d2[hit]
    return idx, zbuf, dists


def point_alphas(dists, radius):
    """refine_utils.py:321-326."""
    d = np.float32(0.1) * dists.astype(np.float32) / np.float32(radius * radius)
    return (np.float32(1) - np.sqrt(np.clip(d, np.float32(1e-3), np.float32(1)))).astype(np.float32)


def alpha_composite(idx, alphas, feats):
    """feats [P,C] -> image [C,H,W]; weights [H,W,K] returned too (w_k = T_k alpha_k, 0 for unused slots)."""
    H, W, K = idx.shape
    C = feats.shape[1]
    out = np.zeros((C, H, W), np.float32)
    wts = np.zeros((H, W, K), np.float32)
    T = np.ones((H, W), np.float32)
    for k in range(K):
        used = idx[..., k] >= 0
        a = np.where(used, alphas[..., k], np.float32(0)).astype(np.float32)
        w = (T * a).astype(np.float32)
        wts[..., k] = w
        f = feats[np.clip(idx[..., k], 0, None)]              # [H,W,C]
        out += (w[..., None] * f * used[..., None]).transpose(2, 0, 1).astype(np.float32)
        T = (T * (np.float32(1) - a)).astype(np.float32)
    return out, wts


def alpha_composite_backward(idx, wts, dout, P):
    """d(out)/d(feats): grad_feats[p, c] = sum over (pixel, k) with idx == p of w_k * dout[c, pixel]."""
    C = dout.shape[0]
    g = np.zeros((P, C), np.float64)
    H, W, K = idx.shape
    d = dout.transpose(1, 2, 0).astype(np.float64)           # [H,W,C]
    for k in range(K):
        used = idx[..., k] >= 0
        np.add.at(g, idx[..., k][used], wts[..., k][used][:, None] * d[used])
    return g


def render_point(points, feats, H, W, K, world2cam, radius, ppp):
    """refine_utils.py:306-333 end to end -> (image [C,H,W], idx, dists, weights)."""
    ndc = project_points(points, K, world2cam, H, W)
    idx, _, dists = rasterize_points(ndc, H, W, radius, ppp)
    alphas = point_alphas(dists, radius)
    img, wts = alpha_composite(idx, alphas, feats.astype(np.float32))
    return img, idx, dists, wts
