"""Synthetic camera rays for benchmarks and tests: the reference's orbit pose + pinhole ray generator,
restated (nerf/provider.py:217-241 `circle_poses`, nerf/utils.py:51-116 `get_rays` with N=-1, and the intrinsics
of nerf/provider.py:294-297).  Parity with the reference's own functions is pinned by tests/golden/rays.npz."""
import math

import torch


def _unit(v, eps=1e-20):
    # nerf/utils.py:47-48 safe_normalize
    return v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=eps, max=1e32))


def orbit_pose(radius=1.25, theta_deg=80.0, phi_deg=30.0, device="cpu"):
    """Camera-to-world matrix [1,4,4] looking at the origin from spherical (radius, theta from +y, phi about y)."""
    th = torch.tensor([math.radians(theta_deg)], dtype=torch.float32, device=device)
    ph = torch.tensor([math.radians(phi_deg)], dtype=torch.float32, device=device)
    eye = torch.stack([radius * torch.sin(th) * torch.sin(ph), radius * torch.cos(th),
                       radius * torch.sin(th) * torch.cos(ph)], -1)
    fwd = -_unit(eye)
    up0 = torch.tensor([[0.0, -1.0, 0.0]], device=device)
    right = _unit(torch.cross(fwd, up0, dim=-1))
    up = _unit(torch.cross(right, fwd, dim=-1))
    pose = torch.eye(4, dtype=torch.float32, device=device).unsqueeze(0)
    pose[:, :3, :3] = torch.stack((right, up, fwd), -1)
    pose[:, :3, 3] = eye
    return pose


def pinhole_rays(pose, H, W, fov_deg=20.0):
    """Full-image rays. Returns rays_o [B,H*W,3], rays_d [B,H*W,3] (unit), depth_scale [B,H*W] = 1/|pixel dir|."""
    device = pose.device
    focal = H / (2 * math.tan(math.radians(fov_deg) / 2))
    cx, cy = H / 2, W / 2  # sic: the reference passes (H/2, W/2) as (cx, cy)
    B = pose.shape[0]
    u, v = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device),
                          indexing="ij")
    u = u.t().reshape(1, H * W).expand(B, H * W) + 0.5
    v = v.t().reshape(1, H * W).expand(B, H * W) + 0.5
    z = torch.ones_like(u)
    cam = torch.stack(((u - cx) / focal * z, (v - cy) / focal * z, z), -1)
    depth_scale = 1 / cam.pow(2).sum(-1).pow(0.5)
    cam = _unit(cam)
    rays_d = cam @ pose[:, :3, :3].transpose(-1, -2)
    rays_o = pose[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous(), depth_scale


def view_rays(H, W, view=0, radius=1.25, theta_deg=80.0, fov_deg=20.0, device="cpu"):
    """circle_poses(radius 1.25, theta 80, phi 30 + 45*k) -> full-image rays."""
    return pinhole_rays(orbit_pose(radius, theta_deg, 30.0 + 45.0 * view, device), H, W, fov_deg)
