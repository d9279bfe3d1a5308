"""Python-level ray ops with the call signatures of the reference's modules/radnerfs/raymarching/raymarching.py
(near_far_from_aabb :18-48, morton3D :81-104, morton3D_invert :106-126, packbits :128-154, march_rays :347-398,
composite_rays :401-423), executed by hand-written HIP kernels through the C ABI (include/gfpp_radnerf.h).

Inference only: these are plain functions, not autograd Functions (the training variants are SURVEY 8f-2 "next").
Inputs must already live on the GPU; unlike the reference nothing is silently moved (`.cuda()`), and kernels run on
torch's CURRENT stream rather than the legacy default stream.
"""
import torch

from .._lib import call


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (genefaceplusplus_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _f32(t, name):
    """The reference casts to float32 via custom_fwd(cast_inputs=torch.float32)."""
    if t.dtype != torch.float32:
        t = t.float()
    return _req(t.contiguous(), torch.float32, name)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    aabb = _f32(aabb, "aabb")
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    call("gfpp_near_far_from_aabb", rays_o.data_ptr(), rays_d.data_ptr(), aabb.data_ptr(), N, float(min_near),
         nears.data_ptr(), fars.data_ptr(), _stream())
    return nears, fars


def morton3D(coords):
    coords = _req(coords.int().contiguous(), torch.int32, "coords")
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    call("gfpp_morton3D", coords.data_ptr(), N, indices.data_ptr(), _stream())
    return indices


def morton3D_invert(indices):
    indices = _req(indices.int().contiguous(), torch.int32, "indices")
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    call("gfpp_morton3D_invert", indices.data_ptr(), N, coords.data_ptr(), _stream())
    return coords


def packbits(grid, thresh, bitfield=None):
    grid = _f32(grid, "grid")
    C, H3 = grid.shape
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    _req(bitfield, torch.uint8, "bitfield")
    call("gfpp_packbits", grid.data_ptr(), N, float(thresh), bitfield.data_ptr(), _stream())
    return bitfield


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024):
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    _req(rays_alive, torch.int32, "rays_alive")
    _req(rays_t, torch.float32, "rays_t")
    _req(density_bitfield, torch.uint8, "density_bitfield")
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)      # always adds 1..align slots, like the reference
    dev = rays_o.device
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    if perturb:
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev)
    else:
        noises = torch.zeros(n_alive, dtype=torch.float32, device=dev)
    call("gfpp_march_rays", n_alive, n_step, rays_alive.data_ptr(), rays_t.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(),
         float(bound), float(dt_gamma), int(max_steps), int(C), int(H), density_bitfield.data_ptr(), near.data_ptr(),
         far.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), noises.data_ptr(), _stream())
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    sigmas = _f32(sigmas, "sigmas")
    rgbs = _f32(rgbs, "rgbs")
    deltas = _f32(deltas, "deltas")
    for t, n in ((rays_t, "rays_t"), (weights_sum, "weights_sum"), (depth, "depth"), (image, "image")):
        _req(t, torch.float32, n)
    _req(rays_alive, torch.int32, "rays_alive")
    call("gfpp_composite_rays", n_alive, n_step, float(T_thresh), rays_alive.data_ptr(), rays_t.data_ptr(), sigmas.data_ptr(),
         rgbs.data_ptr(), deltas.data_ptr(), weights_sum.data_ptr(), depth.data_ptr(), image.data_ptr(), _stream())
    return tuple()
