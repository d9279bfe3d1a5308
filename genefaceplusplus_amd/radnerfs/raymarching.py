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


# ---------------------------------------------------------------------------------------------------------------------
# training side (reference: raymarching.py:186-345; kernels raymarching.cu:162-820) -- same argument lists and return values
# ---------------------------------------------------------------------------------------------------------------------
def morton3D_dilation(grid):
    """grid [C, H^3] f32 in Morton order -> 6-neighbour max pool (raymarching.py:131-152)."""
    grid = _f32(grid, "grid")
    C, H3 = grid.shape
    H = int(round(H3 ** (1.0 / 3.0)))
    out = torch.empty_like(grid)
    call("gfpp_morton3D_dilation", grid.data_ptr(), C, H, out.data_ptr(), _stream())
    return out


def sph_from_ray(rays_o, rays_d, radius):
    """-> coords [N,2] in [-1,1] (raymarching.py:50-78)."""
    rays_o = _f32(rays_o, "rays_o").view(-1, 3)
    rays_d = _f32(rays_d, "rays_d").view(-1, 3)
    coords = torch.empty(rays_o.shape[0], 2, dtype=torch.float32, device=rays_o.device)
    call("gfpp_sph_from_ray", rays_o.data_ptr(), rays_d.data_ptr(), float(radius), rays_o.shape[0], coords.data_ptr(), _stream())
    return coords


class _MarchRaysTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o = _f32(rays_o, "rays_o").view(-1, 3)
        rays_d = _f32(rays_d, "rays_d").view(-1, 3)
        _req(density_bitfield, torch.uint8, "density_bitfield")
        dev = rays_o.device
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        _req(step_counter, torch.int32, "step_counter")
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
        call("gfpp_march_rays_train", rays_o.data_ptr(), rays_d.data_ptr(), density_bitfield.data_ptr(), float(bound), float(dt_gamma), int(max_steps),
             N, int(C), int(H), M, _f32(nears, "nears").data_ptr(), _f32(fars, "fars").data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(),
             rays.data_ptr(), step_counter.data_ptr(), noises.data_ptr(), _stream())
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())          # D2H copy, as in the reference (raymarching.py:248)
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        ctx.save_for_backward(rays, deltas)
        return xyzs, dirs, deltas, rays

    @staticmethod
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        rays, deltas = ctx.saved_tensors
        N, M = rays.shape[0], grad_xyzs.shape[0]
        grad_rays_o = torch.zeros(N, 3, device=rays.device)
        grad_rays_d = torch.zeros(N, 3, device=rays.device)
        call("gfpp_march_rays_train_backward", _f32(grad_xyzs, "grad_xyzs").data_ptr(), _f32(grad_dirs, "grad_dirs").data_ptr(), rays.data_ptr(),
             deltas.contiguous().data_ptr(), N, M, grad_rays_o.data_ptr(), grad_rays_d.data_ptr(), _stream())
        return (grad_rays_o, grad_rays_d) + (None,) * 13


march_rays_train = _MarchRaysTrain.apply


class _CompositeRaysTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, ambient, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, ambient, deltas = _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs"), _f32(ambient, "ambient"), _f32(deltas, "deltas")
        _req(rays, torch.int32, "rays")
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        ambient_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        call("gfpp_composite_rays_train_forward", sigmas.data_ptr(), rgbs.data_ptr(), ambient.data_ptr(), deltas.data_ptr(), rays.data_ptr(), M, N,
             float(T_thresh), weights_sum.data_ptr(), ambient_sum.data_ptr(), depth.data_ptr(), image.data_ptr(), _stream())
        ctx.save_for_backward(sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image)
        ctx.dims = (M, N, float(T_thresh))
        return weights_sum, ambient_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_ambient_sum, grad_depth, grad_image):
        sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        gw, ga, gi = _f32(grad_weights_sum, "grad_weights_sum"), _f32(grad_ambient_sum, "grad_ambient_sum"), _f32(grad_image, "grad_image")
        grad_sigmas, grad_rgbs, grad_ambient = torch.zeros_like(sigmas), torch.zeros_like(rgbs), torch.zeros_like(ambient)
        call("gfpp_composite_rays_train_backward", gw.data_ptr(), ga.data_ptr(), gi.data_ptr(), sigmas.data_ptr(), rgbs.data_ptr(), ambient.data_ptr(),
             deltas.data_ptr(), rays.data_ptr(), weights_sum.data_ptr(), ambient_sum.data_ptr(), image.data_ptr(), M, N, T_thresh, grad_sigmas.data_ptr(),
             grad_rgbs.data_ptr(), grad_ambient.data_ptr(), _stream())
        return grad_sigmas, grad_rgbs, grad_ambient, None, None, None


composite_rays_train = _CompositeRaysTrain.apply
