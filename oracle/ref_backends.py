"""Stand-ins for the reference's four native extension modules, backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  The reference's Python shims (modules/radnerfs/raymarching/raymarching.py,
encoders/gridencoder/grid.py, encoders/shencoder/sphere_harmonics.py, encoders/freqencoder/freq.py) do
``import _raymarching_face as _backend`` etc.  ``install()`` registers modules of those names in ``sys.modules`` whose
functions have the pybind signatures of raymarching.h:7-19 / gridencoder.h:12-15 / shencoder.h / freqencoder.h but run
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


# ---- _gridencoder / _shencoder / _freqencoder ------------------------------------------------------------
def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    assert dy_dx is None, "oracle restates the inference path only"
    rc = orc.lib().orc_grid_encode_forward(_fp(inputs), _fp(embeddings), _ip(offsets), _fp(outputs), _u32(B), _u32(D), _u32(C),
                                           _u32(L), ctypes.c_float(S), _u32(H), _u32(gridtype), ctypes.c_int(int(align_corners)),
                                           _u32(interp))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C")


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    assert dy_dx is None and D == 3
    rc = orc.lib().orc_sh_encode_forward(_fp(inputs), _fp(outputs), _u32(B), _u32(C))
    if rc != 0:
        raise RuntimeError("SH encoding: degree must be in 1..4 in the oracle")


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
                                               morton3D_invert=morton3D_invert)
    sys.modules["_gridencoder"] = _module("_gridencoder", grid_encode_forward=grid_encode_forward)
    sys.modules["_shencoder"] = _module("_shencoder", sh_encode_forward=sh_encode_forward)
    sys.modules["_freqencoder"] = _module("_freqencoder", freq_encode_forward=freq_encode_forward)
