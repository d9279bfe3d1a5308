"""Stand-ins for the reference's four native extension modules, backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  The reference's Python shims (modules/radnerfs/raymarching/raymarching.py,
encoders/gridencoder/grid.py, encoders/shencoder/sphere_harmonics.py, encoders/freqencoder/freq.py) do
``import _raymarching_face as _backend`` etc.  ``install()`` registers modules of those names in ``sys.modules`` whose
functions have the pybind signatures of raymarching.h:7-20 / gridencoder.h:12-15 / shencoder.h / freqencoder.h but run
radnerf_oracle.c on CPU torch tensors (in place, like the CUDA originals).  tests/golden/make_golden.py uses this to
execute the reference's own Python control flow without CUDA.
"""
import ctypes
import sys
import types

import numpy as np
import torch

from . import oracle as orc

_FP = ctypes.POINTER(ctypes.c_float)
_IP = ctypes.POINTER(ctypes.c_int32)
_UP = ctypes.POINTER(ctypes.c_uint8)


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.is_contiguous())
    return ctypes.cast(t.data_ptr(), _FP)


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), _IP)


def _up(t):
    assert t.dtype == torch.uint8 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), _UP)


def _u32(v):
    return ctypes.c_uint32(int(v))


# ---- _raymarching_face -------------------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    orc.lib().orc_near_far_from_aabb(_fp(rays_o), _fp(rays_d), _fp(aabb), _u32(N), ctypes.c_float(min_near), _fp(nears), _fp(fars))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
               xyzs, dirs, deltas, noises):
    orc.lib().orc_march_rays(_u32(n_alive), _u32(n_step), _ip(rays_alive), _fp(rays_t), _fp(rays_o), _fp(rays_d),
                             ctypes.c_float(bound), ctypes.c_float(dt_gamma), _u32(max_steps), _u32(C), _u32(H), _up(grid),
                             _fp(nears), _fp(fars), _fp(xyzs), _fp(dirs), _fp(deltas), _fp(noises))


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    sigmas = sigmas.contiguous()
    rgbs = rgbs.contiguous()
    orc.lib().orc_composite_rays(_u32(n_alive), _u32(n_step), ctypes.c_float(T_thresh), _ip(rays_alive), _fp(rays_t),
                                 _fp(sigmas), _fp(rgbs), _fp(deltas), _fp(weights_sum), _fp(depth), _fp(image))


def packbits(grid, N, density_thresh, bitfield):
    orc.lib().orc_packbits(_fp(grid), _u32(N), ctypes.c_float(density_thresh), _up(bitfield))


def morton3D(coords, N, indices):
    orc.lib().orc_morton3D_batch(_ip(coords), _u32(N), _ip(indices))


def morton3D_invert(indices, N, coords):
    orc.lib().orc_morton3D_invert_batch(_ip(indices), _u32(N), _ip(coords))


def morton3D_dilation(grid, C, H, grid_dilation):
    orc.lib().orc_morton3D_dilation(_fp(grid), _u32(C), _u32(H), _fp(grid_dilation))


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    orc.lib().orc_sph_from_ray(_fp(rays_o), _fp(rays_d), ctypes.c_float(radius), _u32(N), _fp(coords))


# training side (raymarching.h:14-17)
def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
    orc.lib().orc_march_rays_train(_fp(rays_o), _fp(rays_d), _up(grid), ctypes.c_float(bound), ctypes.c_float(dt_gamma), _u32(max_steps), _u32(N), _u32(C),
                                   _u32(H), _u32(M), _fp(nears), _fp(fars), _fp(xyzs), _fp(dirs), _fp(deltas), _ip(rays), _ip(counter), _fp(noises))


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):
    orc.lib().orc_march_rays_train_backward(_fp(grad_xyzs.contiguous()), _fp(grad_dirs.contiguous()), _ip(rays), _fp(deltas.contiguous()), _u32(N), _u32(M),
                                            _fp(grad_rays_o), _fp(grad_rays_d))


def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image):
    orc.lib().orc_composite_rays_train_forward(_fp(sigmas), _fp(rgbs), _fp(ambient), _fp(deltas.contiguous()), _ip(rays), _u32(M), _u32(N),
                                               ctypes.c_float(T_thresh), _fp(weights_sum), _fp(ambient_sum), _fp(depth), _fp(image))


def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image,
                                  M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient):
    orc.lib().orc_composite_rays_train_backward(_fp(grad_weights_sum), _fp(grad_ambient_sum), _fp(grad_image), _fp(sigmas), _fp(rgbs), _fp(ambient),
                                                _fp(deltas.contiguous()), _ip(rays), _fp(weights_sum), _fp(ambient_sum), _fp(image), _u32(M), _u32(N),
                                                ctypes.c_float(T_thresh), _fp(grad_sigmas), _fp(grad_rgbs), _fp(grad_ambient))


# ---- _gridencoder / _shencoder / _freqencoder ------------------------------------------------------------
def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    if dy_dx is not None:          # [B, L * D * C], the layout of gridencoder.cu:125,202 (B, L, D, C)
        rc = orc.lib().orc_grid_encode_dydx(_fp(inputs), _fp(embeddings), _ip(offsets), _fp(dy_dx), _u32(B), _u32(D), _u32(C), _u32(L), ctypes.c_float(S),
                                            _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp))
        if rc != 0:
            raise RuntimeError("GridEncoding: unsupported D/C")
    rc = orc.lib().orc_grid_encode_forward(_fp(inputs), _fp(embeddings), _ip(offsets), _fp(outputs), _u32(B), _u32(D), _u32(C),
                                           _u32(L), ctypes.c_float(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)),
                                           _u32(interp))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
    rc = orc.lib().orc_grid_encode_backward(_fp(grad), _fp(inputs), _ip(offsets), _fp(grad_embeddings), _u32(B), _u32(D), _u32(C), _u32(L), ctypes.c_float(S),
                                            _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)), _u32(interp))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C")
    if dy_dx is not None:
        orc.lib().orc_grid_input_backward(_fp(grad), _fp(dy_dx), _fp(grad_inputs), _u32(B), _u32(D), _u32(C), _u32(L))


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    assert D == 3
    rc = orc.lib().orc_sh_encode_forward(_fp(inputs), _fp(outputs), _u32(B), _u32(C))
    if rc != 0:
        raise RuntimeError("SH encoding: degree must be in 1..4 in the oracle")
    if dy_dx is not None:          # [B, D * C^2] = (B, D, C^2), shencoder.cu:125-130
        dy_dx.copy_(torch.from_numpy(orc.sh_encode_dydx(inputs.numpy(), C)).reshape(B, -1))


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    # kernel_sh_backward (shencoder.cu:359-382): grad_inputs[b, d] += sum_c grad[b, c] * dy_dx[b, d, c]
    g = np.einsum("bc,bdc->bd", grad.numpy().astype(np.float64), dy_dx.numpy().reshape(B, D, C * C).astype(np.float64))
    grad_inputs.add_(torch.from_numpy(g.astype(np.float32)))


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
    grad_inputs.copy_(torch.from_numpy(orc.freq_encode_backward(grad.numpy(), outputs.numpy(), D, deg)))


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    rc = orc.lib().orc_grad_total_variation(_fp(inputs), _fp(embeddings), _fp(grad), _ip(offsets), ctypes.c_float(weight), _u32(B), _u32(D), _u32(C), _u32(L),
                                            ctypes.c_float(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C")


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    orc.lib().orc_freq_encode_forward(_fp(inputs), _u32(B), _u32(D), _u32(deg), _u32(C), _fp(outputs))


def _module(name, **fns):
    m = types.ModuleType(name)
    for k, v in fns.items():
        setattr(m, k, v)
    return m


def install():
    """Register the four backend modules (idempotent)."""
    sys.modules["_raymarching_face"] = _module("_raymarching_face", near_far_from_aabb=near_far_from_aabb, march_rays=march_rays,
                                               composite_rays=composite_rays, packbits=packbits, morton3D=morton3D,
                                               morton3D_invert=morton3D_invert, morton3D_dilation=morton3D_dilation, sph_from_ray=sph_from_ray,
                                               march_rays_train=march_rays_train, march_rays_train_backward=march_rays_train_backward,
                                               composite_rays_train_forward=composite_rays_train_forward,
                                               composite_rays_train_backward=composite_rays_train_backward)
    sys.modules["_gridencoder"] = _module("_gridencoder", grid_encode_forward=grid_encode_forward, grid_encode_backward=grid_encode_backward,
                                          grad_total_variation=grad_total_variation)
    sys.modules["_shencoder"] = _module("_shencoder", sh_encode_forward=sh_encode_forward, sh_encode_backward=sh_encode_backward)
    sys.modules["_freqencoder"] = _module("_freqencoder", freq_encode_forward=freq_encode_forward, freq_encode_backward=freq_encode_backward)
