"""Drop-in replacements for the reference's four *native extension modules*, so that the reference's own Python shims
(modules/radnerfs/raymarching/raymarching.py, encoders/gridencoder/grid.py, encoders/shencoder/sphere_harmonics.py,
encoders/freqencoder/freq.py) run on MI355X without touching them:

    import genefaceplusplus_amd.compat_ext as ext; ext.install()
    # `import _raymarching_face as _backend` etc. now bind to libgfpp_radnerf.so

Function names, argument order and in-place output convention are those of the pybind modules
(raymarching/src/bindings.cpp:7-20, gridencoder/src/bindings.cpp, shencoder/src/bindings.cpp, freqencoder/src/bindings.cpp);
arguments are torch CUDA(=HIP) tensors, converted to raw pointers for the C ABI.  Kernels run on torch's current stream.
Inference and training entry points; half tables (the reference's autocast) are served through fp32 up-casts where the
ABI is fp32-only.
"""
import sys
import types

import torch

from ._lib import call


def _st():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


# ---- _raymarching_face -------------------------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    call("gfpp_near_far_from_aabb", _p(rays_o), _p(rays_d), _p(aabb), int(N), float(min_near), _p(nears), _p(fars), _st())


def morton3D(coords, N, indices):
    call("gfpp_morton3D", _p(coords), int(N), _p(indices), _st())


def morton3D_invert(indices, N, coords):
    call("gfpp_morton3D_invert", _p(indices), int(N), _p(coords), _st())


def packbits(grid, N, density_thresh, bitfield):
    call("gfpp_packbits", _p(grid), int(N), float(density_thresh), _p(bitfield), _st())


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs,
               deltas, noises):
    call("gfpp_march_rays", int(n_alive), int(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), float(bound), float(dt_gamma),
         int(max_steps), int(C), int(H), _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises), _st())


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
    call("gfpp_composite_rays", int(n_alive), int(n_step), float(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs), _p(deltas),
         _p(weights_sum), _p(depth), _p(image), _st())


def sph_from_ray(rays_o, rays_d, radius, N, coords):                                             # raymarching.h:8
    call("gfpp_sph_from_ray", _p(rays_o), _p(rays_d), float(radius), int(N), _p(coords), _st())


def morton3D_dilation(grid, C, H, grid_dilation):                                                # raymarching.h:12
    call("gfpp_morton3D_dilation", _p(grid), int(C), int(H), _p(grid_dilation), _st())


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):   # raymarching.h:14
    call("gfpp_march_rays_train", _p(rays_o), _p(rays_d), _p(grid), float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H), int(M),
         _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), _p(noises), _st())


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):   # raymarching.h:15
    call("gfpp_march_rays_train_backward", _p(grad_xyzs), _p(grad_dirs), _p(rays), _p(deltas), int(N), int(M), _p(grad_rays_o), _p(grad_rays_d), _st())


def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image):   # raymarching.h:16
    call("gfpp_composite_rays_train_forward", _p(sigmas), _p(rgbs), _p(ambient), _p(deltas), _p(rays), int(M), int(N), float(T_thresh), _p(weights_sum),
         _p(ambient_sum), _p(depth), _p(image), _st())


def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image, M, N,
                                  T_thresh, grad_sigmas, grad_rgbs, grad_ambient):               # raymarching.h:17
    call("gfpp_composite_rays_train_backward", _p(grad_weights_sum), _p(grad_ambient_sum), _p(grad_image), _p(sigmas), _p(rgbs), _p(ambient), _p(deltas),
         _p(rays), _p(weights_sum), _p(ambient_sum), _p(image), int(M), int(N), float(T_thresh), _p(grad_sigmas), _p(grad_rgbs), _p(grad_ambient), _st())


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):   # gridencoder.h:13
    if embeddings.dtype != torch.float32:
        # Under autocast the reference's grid.py:43-44 hands half tables / half grad / half dy_dx to the extension
        # (egs/datasets/May/lm3d_radnerf.yaml:5 `amp: true`).
        ge32 = torch.zeros(grad_embeddings.shape, dtype=torch.float32, device=grad_embeddings.device)
        gi32 = torch.zeros(grad_inputs.shape, dtype=torch.float32, device=grad_inputs.device) if grad_inputs is not None else None
        if int(C) == 2 and grad.dtype == torch.float16:
            # level_dim 2: the half grad goes into the kernels as it is (fp32 accumulation; gridencoder.cu:306-318 accumulates in half); the fp32
            # result is cast into the caller's half tensors
            rows = int(embeddings.shape[0])
            copies = torch.empty(8 * rows * 2 + 64, device=grad_embeddings.device, dtype=torch.float32)
            dy32 = dy_dx.float().contiguous() if dy_dx is not None else None
            call("gfpp_grid_encode_backward_f16", _p(grad.contiguous()), _p(inputs), _p(offsets), _p(ge32), rows, _p(copies), int(B), int(D), 2, int(L), float(S),
                 int(H), _p(dy32), _p(gi32), int(gridtype), int(bool(align_corners)), int(interp), _st(), None, 0)
        else:
            # other level_dims: fp32 accumulation (more accurate than the reference's half atomics, same interface)
            grid_encode_backward(grad.float(), inputs, embeddings.float(), offsets, ge32, B, D, C, L, S, H, dy_dx.float() if dy_dx is not None else None, gi32,
                                 gridtype, align_corners, interp)
        grad_embeddings.add_(ge32.to(grad_embeddings.dtype))
        if grad_inputs is not None:
            grad_inputs.add_(gi32.to(grad_inputs.dtype))
        return
    rows = int(embeddings.shape[0])
    copies = torch.empty(8 * rows * int(C) + 64, device=grad_embeddings.device, dtype=torch.float32)       # gradient copies + level maxima, see gfpp_grid_encode_backward_xcd
    call("gfpp_grid_encode_backward_xcd", _p(grad), _p(inputs), _p(offsets), _p(grad_embeddings), rows, _p(copies), int(B), int(D), int(C), int(L), float(S), int(H),
         _p(dy_dx), _p(grad_inputs), int(gridtype), int(bool(align_corners)), int(interp), _st(), None, 0)


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):   # gridencoder.h:15
    if embeddings.dtype != torch.float32:
        g32 = torch.zeros(grad.shape, dtype=torch.float32, device=grad.device)
        grad_total_variation(inputs, embeddings.float(), g32, offsets, weight, B, D, C, L, S, H, gridtype, align_corners)
        grad.add_(g32.to(grad.dtype))
        return
    call("gfpp_grad_total_variation", _p(inputs), _p(embeddings), _p(grad), _p(offsets), float(weight), int(B), int(D), int(C), int(L), float(S), int(H),
         int(gridtype), int(bool(align_corners)), _st())


# ---- _gridencoder / _shencoder / _freqencoder ------------------------------------------------------------------------
def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    if dy_dx is not None and embeddings.dtype != torch.float32:
        # training under autocast (grid.py:43-52 with calc_grad_inputs): the ABI produces dy_dx in fp32 only -- compute on an fp32 view of
        # the half tables and round once into the caller's half outputs
        o32 = torch.empty(outputs.shape, dtype=torch.float32, device=outputs.device)
        d32 = torch.empty(dy_dx.shape, dtype=torch.float32, device=dy_dx.device)
        grid_encode_forward(inputs, embeddings.float(), offsets, o32, B, D, C, L, S, H, d32, gridtype, align_corners, interp)
        outputs.copy_(o32)
        dy_dx.copy_(d32)
        return
    dtype = {torch.float32: 0, torch.float16: 1}[embeddings.dtype]
    call("gfpp_grid_encode_forward", _p(inputs), _p(embeddings), _p(offsets), _p(outputs), int(B), int(D), int(C), int(L), float(S), int(H),
         _p(dy_dx), int(gridtype), int(bool(align_corners)), int(interp), dtype, _st())


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    call("gfpp_sh_encode_forward", _p(inputs), _p(outputs), int(B), int(D), int(C), _p(dy_dx), _st())


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):                                # shencoder.h:10
    call("gfpp_sh_encode_backward", _p(grad), _p(inputs), int(B), int(D), int(C), _p(dy_dx), _p(grad_inputs), _st())


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    call("gfpp_freq_encode_forward", _p(inputs), int(B), int(D), int(deg), int(C), _p(outputs), _st())


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):                              # freqencoder.h:10
    call("gfpp_freq_encode_backward", _p(grad), _p(outputs), int(B), int(D), int(deg), int(C), _p(grad_inputs), _st())


def _module(name, fns):
    m = types.ModuleType(name)
    m.__dict__.update(fns)
    return m


def install():
    rm = dict(near_far_from_aabb=near_far_from_aabb, morton3D=morton3D, morton3D_invert=morton3D_invert, packbits=packbits,
              march_rays=march_rays, composite_rays=composite_rays, sph_from_ray=sph_from_ray, morton3D_dilation=morton3D_dilation,
              march_rays_train=march_rays_train, march_rays_train_backward=march_rays_train_backward,
              composite_rays_train_forward=composite_rays_train_forward, composite_rays_train_backward=composite_rays_train_backward)
    sys.modules["_raymarching_face"] = _module("_raymarching_face", rm)
    sys.modules["_gridencoder"] = _module("_gridencoder", dict(grid_encode_forward=grid_encode_forward, grid_encode_backward=grid_encode_backward,
                                                                 grad_total_variation=grad_total_variation))
    sys.modules["_shencoder"] = _module("_shencoder", dict(sh_encode_forward=sh_encode_forward, sh_encode_backward=sh_encode_backward))
    sys.modules["_freqencoder"] = _module("_freqencoder", dict(freq_encode_forward=freq_encode_forward, freq_encode_backward=freq_encode_backward))
