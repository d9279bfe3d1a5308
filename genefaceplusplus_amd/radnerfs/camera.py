"""Camera / conditioning-window helpers the inference loop imports from ``modules.radnerfs.utils``
(genefacepp_infer.py:39-43): get_rays, get_bg_coords, convert_poses, nerf_matrix_to_ngp, get_audio_features, trunc_exp.

Signatures and results follow modules/radnerfs/utils.py:36-60, 71-104, 264-364; none of that module's heavy optional
imports (trimesh, mcubes, lpips, cv2, tensorboardX, imageio) are needed here.  Full-frame ray generation runs as one
HIP kernel (gfpp_get_rays) instead of ~10 torch launches per frame.
"""
import numpy as np
import torch

from .._lib import call, GfppError


class _TruncExp(torch.autograd.Function):
    """utils.py:34-47: forward exp(x) in float32; backward g * exp(clamp(x, -15, 15)) -- the clamp exists ONLY in the backward pass and
    keeps sigma gradients finite once a logit passes ~15."""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def trunc_exp(x):
    """exp evaluated in float32 with the reference's clamped backward (utils.py:34-49)."""
    return _TruncExp.apply(x)


def nerf_matrix_to_ngp(pose, scale=4, offset=(0, 0, 0)):
    """Axis permutation from the NeRF/OpenGL convention to the ngp convention, translation scaled (utils.py:53-60)."""
    p = np.asarray(pose)
    out = np.eye(4, dtype=np.float32)
    for row, src in enumerate((1, 2, 0)):
        out[row, 0] = p[src, 0]
        out[row, 1] = -p[src, 1]
        out[row, 2] = -p[src, 2]
        out[row, 3] = p[src, 3] * scale + offset[row]
    return out


def matrix_to_euler_xyz(m):
    """Euler angles of rotation matrices [...,3,3] for the 'XYZ' convention (what utils.py:164-200 evaluates to)."""
    return torch.stack((torch.atan2(-m[..., 1, 2], m[..., 2, 2]), torch.asin(m[..., 0, 2]), torch.atan2(-m[..., 0, 1], m[..., 0, 0])), dim=-1)


def convert_poses(poses):
    """[B,4,4] -> [B,6] = (euler xyz, translation) (utils.py:264-270)."""
    out = torch.empty(poses.shape[0], 6, dtype=torch.float32, device=poses.device)
    out[:, :3] = matrix_to_euler_xyz(poses[:, :3, :3])
    out[:, 3:] = poses[:, :3, 3]
    return out


def get_bg_coords(H, W, device):
    """[1, H*W, 2] in [-1,1]; component 0 is the ROW coordinate (utils.py:274-279)."""
    rows = torch.arange(H, device=device) / (H - 1) * 2 - 1
    cols = torch.arange(W, device=device) / (W - 1) * 2 - 1
    rr, cc = torch.meshgrid(rows, cols, indexing="ij")
    return torch.stack((rr.reshape(-1), cc.reshape(-1)), dim=-1).unsqueeze(0)


def get_audio_features(features, att_mode, index, smo_win_size=None):
    """Conditioning window around frame ``index`` (utils.py:71-104).  ``smo_win_size`` replaces the reference's global hparams."""
    if smo_win_size is None:
        from . import runtime_hparams
        smo_win_size = runtime_hparams()["smo_win_size"]
    T = features.shape[0]
    if att_mode == 0:
        return features[[index]]
    if att_mode == 1:
        lo = index - smo_win_size
        pad = max(0, -lo)
        window = features[max(lo, 0):index]
        if pad:
            window = torch.cat([torch.zeros(pad, *window.shape[1:], device=window.device, dtype=window.dtype), window], dim=0)
        return window
    if att_mode == 2:
        lo = index - smo_win_size // 2
        hi = index + (smo_win_size - smo_win_size // 2)
        pad_lo, pad_hi = max(0, -lo), max(0, hi - T)
        window = features[max(lo, 0):min(hi, T)]
        if pad_lo:
            window = torch.cat([torch.zeros_like(window[:pad_lo]), window], dim=0)
        if pad_hi:
            window = torch.cat([window, torch.zeros_like(window[:pad_hi])], dim=0)
        return window
    raise NotImplementedError(f"wrong att_mode: {att_mode}")


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
    """Ray generation (utils.py:283-364): every pixel (N = -1, the inference case), N random pixels, random patch_size^2 patches
    (N // patch_size^2 of them), or the pixels of ``rect`` = (xmin, xmax, ymin, ymax) row/col window.

    poses [B,4,4] cam2world on the GPU; returns rays_o/rays_d [B,n,3], inds [B,n], i/j pixel centres.  Random indices are drawn with
    the same torch.randint calls, in the same order, as the reference (so a seeded generator gives the reference's pixels); the rays
    of the selected pixels come from one HIP launch per pose instead of a full-frame meshgrid + gather.
    """
    if not poses.is_cuda:
        raise GfppError("get_rays: poses must be on the GPU (no CPU path)")
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    B = poses.shape[0]
    dev = poses.device
    poses = poses.float().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    if rect is not None:
        xmin, xmax, ymin, ymax = rect
        N = (xmax - xmin) * (ymax - ymin)
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:
            num_patch = N // (patch_size ** 2)
            top = torch.randint(0, H - patch_size, size=[num_patch], device=dev)
            left = torch.randint(0, W - patch_size, size=[num_patch], device=dev)
            corner = torch.stack([top, left], dim=-1)                                                   # [np, 2]
            pi, pj = torch.meshgrid(torch.arange(patch_size, device=dev), torch.arange(patch_size, device=dev), indexing="ij")
            cells = (corner.unsqueeze(1) + torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1).unsqueeze(0)).view(-1, 2)
            sel = cells[:, 0] * W + cells[:, 1]
        elif rect is not None:
            mask = torch.zeros(H, W, dtype=torch.bool, device=dev)
            mask[xmin:xmax, ymin:ymax] = True
            sel = torch.where(mask.view(-1))[0]
        else:
            sel = torch.randint(0, H * W, size=[N], device=dev)                                        # may repeat, like the reference
        sel = sel.to(torch.int64).contiguous()
        n = sel.shape[0]
        rays_o = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
        rays_d = torch.empty(B, n, 3, dtype=torch.float32, device=dev)
        for b in range(B):
            call("gfpp_get_rays_at", poses[b].data_ptr(), fx, fy, cx, cy, H, W, sel.data_ptr(), n, rays_o[b].data_ptr(), rays_d[b].data_ptr(), st)
        inds = sel.unsqueeze(0).expand(B, n) if rect is None else sel.unsqueeze(0)
    else:
        rays_o = torch.empty(B, H * W, 3, dtype=torch.float32, device=dev)
        rays_d = torch.empty(B, H * W, 3, dtype=torch.float32, device=dev)
        for b in range(B):
            call("gfpp_get_rays", poses[b].data_ptr(), fx, fy, cx, cy, H, W, rays_o[b].data_ptr(), rays_d[b].data_ptr(), st)
        inds = torch.arange(H * W, device=dev).expand(B, H * W)
    i = (inds % W).float() + 0.5
    j = torch.div(inds, W, rounding_mode="floor").float() + 0.5
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds, "i": i, "j": j}
