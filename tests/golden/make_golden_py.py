"""Generates tests/golden/*.npz from the reference's OWN Python code (run in the build container only,
where /root/reference exists):   python tests/golden/make_golden_py.py

Fixtures (all small; parameters are stored as seeds, not tensors):
  field_head.npz : reference NeRFNetwork.common_forward / normal / forward(3 shadings)
                   (nerf/network_tcnn.py:102-170) on seeded points, 16-level grid + 3x64 MLP, CPU fp32.
                   The tcnn encoder inside it is oracle.field_torch.HashGridTorch (PARITY UNPINNED for tcnn).
  c1_run.npz     : BASELINE.json config 1 - reference NeRFRenderer.run (nerf/renderer.py:332-479) on
                   64x64 rays from the reference's circle_poses + get_rays (nerf/provider.py:217-241,
                   nerf/utils.py:51-116), 64 samples (+32 PDF-upsampled, eval() => deterministic), L=4 hash
                   grid + Linear(8,32)-ReLU-Linear(32,4), bg_color ones.
  rays.npz       : get_rays/circle_poses outputs for three poses at 16x16 (ray-provider parity).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def lively_field(cfg, num_layers, hidden, seed, table_scale):
    """Seeded parameters shared by generator and tests (tests rebuild them from the seed)."""
    fp = O.FieldParams(cfg, num_layers=num_layers, hidden_dim=hidden, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    fp.params = (rng.uniform(-1, 1, cfg.n_params) * table_scale).astype(np.float32)
    return fp


def load_into_reference(net, fp):
    with torch.no_grad():
        net.encoder.params.copy_(torch.from_numpy(fp.params))
        for l, lin in enumerate(net.sigma_net.net):
            lin.weight.copy_(torch.from_numpy(fp.W[l]))
            lin.bias.copy_(torch.from_numpy(fp.B[l]))


def main():
    ref_import.install()
    from nerf.network_tcnn import MLP, NeRFNetwork
    from nerf.provider import circle_poses
    from nerf.utils import get_rays
    from oracle.field_torch import HashGridTorch

    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- field head
    cfg = O.GridConfig()
    fp = lively_field(cfg, 3, 64, seed=11, table_scale=0.3)
    net = ref_import.reference_network(ref_import.default_opt())
    load_into_reference(net, fp)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (192, 3)).astype(np.float32)
    x[:32] *= 0.15            # inside the blob
    x[32:48, 0] = 0.999       # stencil clamps at +bound
    x[48:64, 2] = -1.0        # exactly on -bound
    d = rng.normal(size=(192, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    light = np.array([0.3, -0.5, 0.81], np.float32)
    light /= np.linalg.norm(light)
    out = dict(x=x, d=d, light=light, seed=11, table_scale=0.3)
    with torch.no_grad():
        xt, dt_, lt = torch.from_numpy(x), torch.from_numpy(d), torch.from_numpy(light)
        s, a = net.common_forward(xt)
        out["sigma"], out["albedo"] = s.numpy(), a.numpy()
        out["normal"] = net.normal(xt).numpy()
        for sh in ("albedo", "lambertian", "textureless", "normal"):
            s2, c2, n2 = net(xt, dt_, lt, ratio=0.3, shading=sh)
            out["color_" + sh] = c2.numpy()
    np.savez_compressed(os.path.join(HERE, "field_head.npz"), **out)
    print("field_head: sigma range", out["sigma"].min(), out["sigma"].max())

    # ---------------------------------------------------------------- rays
    rays = {}
    for k, (theta, phi, radius) in enumerate([(80, 30, 1.25), (60, 165, 1.0), (100, -90, 1.5)]):
        _, _, poses = circle_poses("cpu", radius=radius, theta=theta, phi=phi)
        H = W = 16
        fov = 20
        focal = H / (2 * np.tan(np.deg2rad(fov) / 2))
        r = get_rays(poses, np.array([focal, focal, H / 2, W / 2]), H, W, -1)
        rays[f"o{k}"], rays[f"d{k}"], rays[f"s{k}"] = r["rays_o"].numpy(), r["rays_d"].numpy(), r["depth_scale"].numpy()
        rays[f"cfg{k}"] = np.array([theta, phi, radius, H, W, fov], np.float64)
        rays[f"pose{k}"] = poses.numpy()
    np.savez_compressed(os.path.join(HERE, "rays.npz"), **rays)

    # ---------------------------------------------------------------- C1 run()
    c1 = O.GridConfig(n_levels=4, per_level_scale=128 ** (1 / 3))
    fp1 = lively_field(c1, 2, 32, seed=21, table_scale=0.3)
    net1 = ref_import.reference_network(ref_import.default_opt(), num_layers=2, hidden_dim=32)
    net1.encoder = HashGridTorch(c1)
    net1.sigma_net = MLP(c1.n_output_dims, 4, 32, 2, bias=True)
    load_into_reference(net1, fp1)
    net1.eval()
    _, _, poses = circle_poses("cpu", radius=1.25, theta=80, phi=30)
    H = W = 64
    focal = H / (2 * np.tan(np.deg2rad(20) / 2))
    r = get_rays(poses, np.array([focal, focal, H / 2, W / 2]), H, W, -1)
    ro, rd = r["rays_o"], r["rays_d"]
    bg = torch.ones(H * W, 3)
    out = dict(rays_o=ro.numpy(), rays_d=rd.numpy(), seed=21, table_scale=0.3)
    light = torch.tensor([0.1, -0.6, 0.79])
    light = light / light.norm()
    out["light"] = light.numpy()
    with torch.no_grad():
        for tag, ups in (("u0", 0), ("u32", 32)):
            res = net1.run(ro, rd, num_steps=64, upsample_steps=ups, light_d=light, ambient_ratio=1.0,
                           shading="albedo", bg_color=bg, perturb=False)
            out[f"image_{tag}"] = res["image"].numpy()
            out[f"depth_{tag}"] = res["depth"].numpy()
            out[f"ws_{tag}"] = res["weights_sum"].numpy()
            out[f"normal_{tag}"] = res["normal"].numpy()
            print(tag, "ws mean", float(res["weights_sum"].mean()), "image mean", float(res["image"].mean()))
    np.savez_compressed(os.path.join(HERE, "c1_run.npz"), **out)


if __name__ == "__main__":
    main()
