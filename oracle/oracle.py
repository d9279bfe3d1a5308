"""CPU oracle for the GeneFace++ motion2video NeRF render path (modules/radnerfs in the reference).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg, never by the product package ``genefaceplusplus_amd``.

PARITY STATUS: pinned.  Kernel level: radnerf_oracle.c reproduces the outputs of the reference's own native
extensions, compiled unmodified for gfx950 (oracle/build_ref.py -> oracle/_ref/) and run on an MI355X
(tests/golden/ref_kernel_golden.npz, tests/test_oracle_ref_kernels_cpu.py; live: tests/test_ref_kernels_gpu.py).
Host level (MLP wiring, render loop, torso pass, ray generation, conditioning nets): tests/golden/make_golden.py
imports the reference's own Python modules from /root/reference, plugs radnerf_oracle.c under their native
extension names, runs them on CPU and commits the outputs as fixtures that this file must reproduce.

All arrays are numpy float32 unless noted; parameters use the reference's state_dict key names.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32 = np.float32
_FP = ctypes.POINTER(ctypes.c_float)
_IP = ctypes.POINTER(ctypes.c_int32)
_UP = ctypes.POINTER(ctypes.c_uint8)


def build():
    """Compile radnerf_oracle.c (gcc, OpenMP).  Idempotent."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_morton3D.restype = ctypes.c_uint32
        _LIB.orc_morton3D.argtypes = [ctypes.c_uint32] * 3
        _LIB.orc_morton3D_invert.restype = ctypes.c_uint32
        _LIB.orc_morton3D_invert.argtypes = [ctypes.c_uint32]
        _LIB.orc_grid_corner_row.restype = ctypes.c_uint32
        _LIB.orc_grid_encode_forward.restype = ctypes.c_int
        _LIB.orc_sh_encode_forward.restype = ctypes.c_int
    return _LIB


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _p(a, T):
    return a.ctypes.data_as(T)


# ---------------------------------------------------------------------------------------------
# native kernels (thin wrappers over radnerf_oracle.c; mirror raymarching.py / grid.py shims)
# ---------------------------------------------------------------------------------------------
def morton3D(x, y, z):
    return int(lib().orc_morton3D(int(x), int(y), int(z)))


def morton3D_invert(i):
    return int(lib().orc_morton3D_invert(int(i)))


def packbits(grid, thresh):
    """raymarching.py packbits: grid [C, H^3] f32 -> bitfield [C*H^3/8] u8."""
    grid = _c(grid, f32).reshape(-1)
    n = grid.size // 8
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_packbits(_p(grid, _FP), ctypes.c_uint32(n), ctypes.c_float(thresh), _p(out, _UP))
    return out


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o = _c(rays_o, f32).reshape(-1, 3)
    rays_d = _c(rays_d, f32).reshape(-1, 3)
    aabb = _c(aabb, f32)
    N = rays_o.shape[0]
    nears = np.empty(N, f32)
    fars = np.empty(N, f32)
    lib().orc_near_far_from_aabb(_p(rays_o, _FP), _p(rays_d, _FP), _p(aabb, _FP), ctypes.c_uint32(N),
                                 ctypes.c_float(min_near), _p(nears, _FP), _p(fars, _FP))
    return nears, fars


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars,
               align=-1, perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """raymarching.py:347-398 (padding: M += align - M % align, outputs zero-initialised)."""
    rays_o = _c(rays_o, f32).reshape(-1, 3)
    rays_d = _c(rays_d, f32).reshape(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), f32)
    dirs = np.zeros((M, 3), f32)
    deltas = np.zeros((M, 2), f32)
    if noises is None:
        if perturb:
            noises = np.random.rand(n_alive).astype(f32)
        else:
            noises = np.zeros(n_alive, f32)
    noises = _c(noises, f32)
    rays_alive = _c(rays_alive, np.int32)
    rays_t = _c(rays_t, f32)
    grid = _c(density_bitfield, np.uint8)
    nears = _c(nears, f32)
    fars = _c(fars, f32)
    lib().orc_march_rays(ctypes.c_uint32(n_alive), ctypes.c_uint32(n_step), _p(rays_alive, _IP), _p(rays_t, _FP),
                         _p(rays_o, _FP), _p(rays_d, _FP), ctypes.c_float(bound), ctypes.c_float(dt_gamma),
                         ctypes.c_uint32(max_steps), ctypes.c_uint32(C), ctypes.c_uint32(H), _p(grid, _UP),
                         _p(nears, _FP), _p(fars, _FP), _p(xyzs, _FP), _p(dirs, _FP), _p(deltas, _FP), _p(noises, _FP))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In place on rays_alive / rays_t / weights_sum / depth / image (must be contiguous numpy arrays)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, f32), (weights_sum, f32), (depth, f32), (image, f32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    sigmas = _c(sigmas, f32)
    rgbs = _c(rgbs, f32)
    deltas = _c(deltas, f32)
    lib().orc_composite_rays(ctypes.c_uint32(n_alive), ctypes.c_uint32(n_step), ctypes.c_float(T_thresh),
                             _p(rays_alive, _IP), _p(rays_t, _FP), _p(sigmas, _FP), _p(rgbs, _FP), _p(deltas, _FP),
                             _p(weights_sum, _FP), _p(depth, _FP), _p(image, _FP))


GRIDTYPE = {"hash": 0, "tiled": 1}
INTERP = {"linear": 0, "smoothstep": 1}


def grid_offsets(input_dim, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, align_corners=False):
    """GridEncoder.__init__ (encoders/gridencoder/grid.py:97-137): offsets [L+1] i32 and per_level_scale."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets = []
    offset = 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), per_level_scale


def grid_level_params(level, S, H):
    sc = ctypes.c_float()
    res = ctypes.c_uint32()
    lib().orc_grid_level_params(ctypes.c_uint32(level), ctypes.c_float(S), ctypes.c_uint32(H), ctypes.byref(sc), ctypes.byref(res))
    return f32(sc.value), int(res.value)


def grid_corner_row(pos_grid, gridtype, align_corners, hashmap_size, resolution):
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    return int(lib().orc_grid_corner_row(ctypes.c_uint32(pg.size), ctypes.c_uint32(gridtype), ctypes.c_int(int(align_corners)),
                                         ctypes.c_uint32(hashmap_size), ctypes.c_uint32(resolution),
                                         pg.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))


def grid_encode_raw(inputs01, embeddings, offsets, S, H, gridtype=0, align_corners=False, interp=0):
    """_backend.grid_encode_forward: inputs in [0,1] [B,D]; returns the level-major [L,B,C] buffer."""
    inputs01 = _c(inputs01, f32)
    B, D = inputs01.shape
    embeddings = _c(embeddings, f32)
    C = embeddings.shape[1]
    offsets = _c(offsets, np.int32)
    L = offsets.shape[0] - 1
    out = np.empty((L, B, C), f32)
    rc = lib().orc_grid_encode_forward(_p(inputs01, _FP), _p(embeddings, _FP), _p(offsets, _IP), _p(out, _FP),
                                       ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L),
                                       ctypes.c_float(S), ctypes.c_uint32(H), ctypes.c_uint32(gridtype),
                                       ctypes.c_int(int(align_corners)), ctypes.c_uint32(interp))
    if rc != 0:
        raise RuntimeError("orc_grid_encode_forward: unsupported D/C")
    return out


def grid_encode(x, embeddings, offsets, per_level_scale, base_resolution=16, gridtype="tiled", align_corners=False,
                interpolation="linear", bound=1):
    """GridEncoder.forward (grid.py:148-164 + :24-63): x in [-bound,bound] [..., D] -> [..., L*C]."""
    x = np.asarray(x, f32)
    prefix = x.shape[:-1]
    u = ((x + f32(bound)) / f32(2 * bound)).astype(f32).reshape(-1, x.shape[-1])
    S = np.log2(per_level_scale)
    out = grid_encode_raw(u, embeddings, offsets, S, base_resolution, GRIDTYPE[gridtype], align_corners, INTERP[interpolation])
    L, B, C = out.shape
    return np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(*prefix, L * C)


def sh_encode(dirs, degree=4):
    dirs = _c(dirs, f32).reshape(-1, 3)
    B = dirs.shape[0]
    out = np.empty((B, degree * degree), f32)
    rc = lib().orc_sh_encode_forward(_p(dirs, _FP), _p(out, _FP), ctypes.c_uint32(B), ctypes.c_uint32(degree))
    if rc != 0:
        raise RuntimeError("orc_sh_encode_forward: degree must be in 1..4")
    return out


def freq_encode(x, degree):
    x = np.asarray(x, f32)
    prefix = x.shape[:-1]
    D = x.shape[-1]
    x2 = _c(x.reshape(-1, D), f32)
    B = x2.shape[0]
    C = D + 2 * D * degree
    out = np.empty((B, C), f32)
    lib().orc_freq_encode_forward(_p(x2, _FP), ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(degree),
                                  ctypes.c_uint32(C), _p(out, _FP))
    return out.reshape(*prefix, C)


# ---------------------------------------------------------------------------------------------
# small dense layers (cond_encoder.py)
# ---------------------------------------------------------------------------------------------
def leaky_relu(x, slope=0.02):
    return np.where(x >= 0, x, x * f32(slope)).astype(f32)


def sigmoid(x):
    x = np.asarray(x, f32)
    return (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)


def linear(x, w, b=None):
    y = np.asarray(x, f32) @ np.asarray(w, f32).T
    if b is not None:
        y = y + np.asarray(b, f32)
    return y.astype(f32)


def conv1d(x, w, b, stride=1, padding=1):
    """x [B,Cin,T], w [Cout,Cin,K] -> [B,Cout,T_out] (torch.nn.Conv1d semantics)."""
    x = np.asarray(x, f32)
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    xp = np.zeros((B, Cin, T + 2 * padding), f32)
    xp[:, :, padding:padding + T] = x
    T_out = (T + 2 * padding - K) // stride + 1
    out = np.zeros((B, Cout, T_out), f32)
    for t in range(T_out):
        patch = xp[:, :, t * stride:t * stride + K]               # [B,Cin,K]
        out[:, :, t] = np.einsum("bck,ock->bo", patch, np.asarray(w, f32))
    return (out + np.asarray(b, f32)[None, :, None]).astype(f32)


def mlp(x, weights):
    """cond_encoder.py:183-202 -- bias-free Linear stack, ReLU between layers."""
    h = np.asarray(x, f32)
    for i, w in enumerate(weights):
        h = h @ np.asarray(w, f32).T
        if i != len(weights) - 1:
            h = np.maximum(h, f32(0))
    return h.astype(f32)


def _mlp_weights(params, prefix):
    ws = []
    i = 0
    while f"{prefix}.net.{i}.weight" in params:
        ws.append(params[f"{prefix}.net.{i}.weight"])
        i += 1
    return ws


_AUDIONET_STRIDES = {1: [1, 1, 1, 1], 2: [2, 1, 1, 1], 3: [2, 2, 1, 1], 4: [2, 2, 1, 1], 16: [2, 2, 2, 2]}


def audio_net(x, params, prefix="cond_prenet", win_size=1):
    """AudioNet.forward (cond_encoder.py:98-143): x [b, t_window, c] -> [b, dim_aud]."""
    if win_size not in _AUDIONET_STRIDES:
        raise ValueError("unsupported win_size")
    strides = _AUDIONET_STRIDES[win_size]
    h = np.asarray(x, f32).transpose(0, 2, 1)
    for i, s in zip((0, 2, 4, 6), strides):
        h = leaky_relu(conv1d(h, params[f"{prefix}.encoder_conv.{i}.weight"], params[f"{prefix}.encoder_conv.{i}.bias"], s, 1))
    h = h[:, :, 0]  # squeeze(-1)
    h = leaky_relu(linear(h, params[f"{prefix}.encoder_fc1.0.weight"], params[f"{prefix}.encoder_fc1.0.bias"]))
    return linear(h, params[f"{prefix}.encoder_fc1.2.weight"], params[f"{prefix}.encoder_fc1.2.bias"])


def audio_att_net(x, params, prefix="cond_att_net"):
    """AudioAttNet.forward (cond_encoder.py:146-180): x [seq, c] -> [c]."""
    x = np.asarray(x, f32)
    seq, c = x.shape
    y = x.T[None]                                                   # [1, c, seq]
    for i in (0, 2, 4, 6, 8):
        y = leaky_relu(conv1d(y, params[f"{prefix}.attentionConvNet.{i}.weight"], params[f"{prefix}.attentionConvNet.{i}.bias"], 1, 1))
    y = linear(y.reshape(1, seq), params[f"{prefix}.attentionNet.0.weight"], params[f"{prefix}.attentionNet.0.bias"])
    y = np.exp(y - y.max(axis=1, keepdims=True))
    y = (y / y.sum(axis=1, keepdims=True)).astype(f32).reshape(seq, 1)
    return (y * x).sum(axis=0).astype(f32)


def cal_cond_feat(cond, params, hp, eye_area_percent=None):
    """RADNeRF.cal_cond_feat (radnerf.py:88-106)."""
    feat = audio_net(cond, params, "cond_prenet", hp["cond_win_size"])
    if hp.get("add_eye_blink_cond", False):
        eap = f32(0.0) if eye_area_percent is None else f32(np.asarray(eye_area_percent, f32).reshape(-1)[0])
        blink = (params["blink_embedding.weight"][0].astype(f32) * eap).reshape(1, -1)
        blink = linear(blink, params["blink_encoder.0.weight"], params["blink_encoder.0.bias"])
        blink = linear(blink, params["blink_encoder.1.weight"], params["blink_encoder.1.bias"])
        k = hp["eye_blink_dim"]
        feat = feat.copy()
        feat[..., :k] = feat[..., :k] + blink
    if hp["with_att"]:
        feat = audio_att_net(feat, params, "cond_att_net")
    return feat.astype(f32)


# ---------------------------------------------------------------------------------------------
# camera helpers (modules/radnerfs/utils.py)
# ---------------------------------------------------------------------------------------------
def nerf_matrix_to_ngp(pose, scale=4, offset=(0, 0, 0)):
    """utils.py:53-60."""
    return np.array([
        [pose[1, 0], -pose[1, 1], -pose[1, 2], pose[1, 3] * scale + offset[0]],
        [pose[2, 0], -pose[2, 1], -pose[2, 2], pose[2, 3] * scale + offset[1]],
        [pose[0, 0], -pose[0, 1], -pose[0, 2], pose[0, 3] * scale + offset[2]],
        [0, 0, 0, 1]], dtype=f32)


def convert_poses(poses):
    """utils.py:264-270 (matrix_to_euler_angles, convention XYZ, :164-200) -> [B,6]."""
    poses = np.asarray(poses, f32)
    M = poses[:, :3, :3]
    out = np.empty((poses.shape[0], 6), f32)
    out[:, 0] = np.arctan2(-M[:, 1, 2], M[:, 2, 2])
    out[:, 1] = np.arcsin(M[:, 0, 2])
    out[:, 2] = np.arctan2(-M[:, 0, 1], M[:, 0, 0])
    out[:, 3:] = poses[:, :3, 3]
    return out


def get_bg_coords(H, W):
    """utils.py:274-279 -> [1, H*W, 2]; [...,0] is the row coordinate."""
    X = (np.arange(H, dtype=np.int64).astype(f32) / f32(H - 1) * f32(2) - f32(1)).astype(f32)
    Y = (np.arange(W, dtype=np.int64).astype(f32) / f32(W - 1) * f32(2) - f32(1)).astype(f32)
    xs, ys = np.meshgrid(X, Y, indexing="ij")
    return np.stack([xs.reshape(-1), ys.reshape(-1)], axis=-1)[None].astype(f32)


def get_rays(poses, intrinsics, H, W):
    """utils.py:283-364 with N=-1 (all pixels, row-major h*W+w)."""
    poses = np.asarray(poses, f32)
    B = poses.shape[0]
    fx, fy, cx, cy = [f32(v) for v in intrinsics]
    jj, ii = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    i = (ii.reshape(1, H * W) + f32(0.5)).astype(f32)
    j = (jj.reshape(1, H * W) + f32(0.5)).astype(f32)
    zs = np.ones_like(i)
    xs = ((i - cx) / fx * zs).astype(f32)
    ys = ((j - cy) / fy * zs).astype(f32)
    directions = np.stack((xs, ys, zs), axis=-1)
    directions = (directions / np.linalg.norm(directions, axis=-1, keepdims=True)).astype(f32)
    directions = np.broadcast_to(directions, (B, H * W, 3))
    rays_d = (directions @ poses[:, :3, :3].transpose(0, 2, 1)).astype(f32)
    rays_o = np.broadcast_to(poses[:, None, :3, 3], rays_d.shape).astype(f32)
    return {"rays_o": rays_o, "rays_d": rays_d, "i": i, "j": j,
            "inds": np.broadcast_to(np.arange(H * W), (B, H * W))}


def get_audio_features(features, att_mode, index, smo_win_size):
    """utils.py:71-104."""
    if att_mode == 0:
        return features[[index]]
    if att_mode == 1:
        left = index - smo_win_size
        pad_left = 0
        if left < 0:
            pad_left = -left
            left = 0
        auds = features[left:index]
        if pad_left > 0:
            auds = np.concatenate([np.zeros((pad_left,) + auds.shape[1:], auds.dtype), auds], axis=0)
        return auds
    if att_mode == 2:
        left = index - smo_win_size // 2
        right = index + (smo_win_size - smo_win_size // 2)
        pad_left = pad_right = 0
        if left < 0:
            pad_left = -left
            left = 0
        if right > features.shape[0]:
            pad_right = right - features.shape[0]
            right = features.shape[0]
        auds = features[left:right]
        if pad_left > 0:
            auds = np.concatenate([np.zeros_like(auds[:pad_left]), auds], axis=0)
        if pad_right > 0:
            auds = np.concatenate([auds, np.zeros_like(auds[:pad_right])], axis=0)
        return auds
    raise NotImplementedError(f"wrong att_mode: {att_mode}")


# ---------------------------------------------------------------------------------------------
# head NeRF (radnerf.py + renderer.py)
# ---------------------------------------------------------------------------------------------
class GridSpec:
    """Static description of one GridEncoder instance (grid.py:97-137)."""

    def __init__(self, input_dim, gridtype, desired_resolution, log2_hashmap_size=16, interpolation="linear",
                 num_levels=16, level_dim=2, base_resolution=16, align_corners=False):
        self.input_dim = input_dim
        self.gridtype = gridtype
        self.interpolation = interpolation
        self.base_resolution = base_resolution
        self.align_corners = align_corners
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.offsets, self.per_level_scale = grid_offsets(input_dim, num_levels, level_dim, 2, base_resolution,
                                                          log2_hashmap_size, desired_resolution, align_corners)

    def encode(self, x, embeddings, bound=1):
        return grid_encode(x, embeddings, self.offsets, self.per_level_scale, self.base_resolution, self.gridtype,
                           self.align_corners, self.interpolation, bound)


def head_grid_specs(hp):
    """radnerf.py:58, 72."""
    gt = {"tiledgrid": "tiled", "hashgrid": "hash"}[hp["grid_type"]]
    pos = GridSpec(3, gt, hp["desired_resolution"] * hp["bound"], hp["log2_hashmap_size"], hp["grid_interpolation_type"])
    amb = GridSpec(hp["ambient_coord_dim"], gt, hp["desired_resolution"], hp["log2_hashmap_size"], hp["grid_interpolation_type"])
    return pos, amb


def torso_grid_spec():
    """radnerf_torso.py:32 -- always tiled, linear, log2_hashmap_size 16, res 2048."""
    return GridSpec(2, "tiled", 2048, 16, "linear")


def head_forward(position, direction, cond_feat, ind_code, params, hp):
    """RADNeRF.forward (radnerf.py:108-141) -> sigma [M], color [M,3], ambient_pos [M,D_amb]."""
    position = np.asarray(position, f32)
    M = position.shape[0]
    pos_spec, amb_spec = head_grid_specs(hp)
    cond = np.broadcast_to(np.asarray(cond_feat, f32).reshape(1, -1), (M, np.asarray(cond_feat).size))
    pos_feat = pos_spec.encode(position, params["position_embedder.embeddings"], bound=hp["bound"])
    ambient_logit = mlp(np.concatenate([pos_feat, cond], axis=1), _mlp_weights(params, "ambient_net"))
    ambient_pos = np.tanh(ambient_logit).astype(f32)
    ambient_feat = amb_spec.encode(ambient_pos, params["ambient_embedder.embeddings"], bound=1)
    h = mlp(np.concatenate([pos_feat, ambient_feat], axis=-1), _mlp_weights(params, "sigma_net"))
    sigma = np.exp(h[..., 0]).astype(f32)
    geo_feat = h[..., 1:]
    direction_feat = sh_encode(direction, 4)
    if ind_code is not None:
        ind = np.broadcast_to(np.asarray(ind_code, f32).reshape(1, -1), (M, np.asarray(ind_code).size))
        color_inp = np.concatenate([direction_feat, geo_feat, ind], axis=-1)
    else:
        color_inp = np.concatenate([direction_feat, geo_feat], axis=-1)
    color = sigmoid(mlp(color_inp, _mlp_weights(params, "color_net")))
    return sigma, color, ambient_pos


def head_density(position, cond_feat, params, hp):
    """RADNeRF.density (radnerf.py:143-166)."""
    position = np.asarray(position, f32)
    M = position.shape[0]
    pos_spec, amb_spec = head_grid_specs(hp)
    cond = np.broadcast_to(np.asarray(cond_feat, f32).reshape(1, -1), (M, np.asarray(cond_feat).size))
    pos_feat = pos_spec.encode(position, params["position_embedder.embeddings"], bound=hp["bound"])
    ambient_pos = np.tanh(mlp(np.concatenate([pos_feat, cond], axis=1), _mlp_weights(params, "ambient_net"))).astype(f32)
    ambient_feat = amb_spec.encode(ambient_pos, params["ambient_embedder.embeddings"], bound=1)
    h = mlp(np.concatenate([pos_feat, ambient_feat], axis=-1), _mlp_weights(params, "sigma_net"))
    return {"sigma": np.exp(h[..., 0]).astype(f32), "geo_feat": h[..., 1:]}


def march_composite_loop(rays_o, rays_d, nears, fars, cond_feat, ind_code, params, hp, dt_gamma, max_steps, T_thresh,
                         density_scale=1, trace=None):
    """The inference branch of NeRFRenderer.render (renderer.py:341-384; identical copies at
    radnerf_torso.py:128-151, radnerf_torso_sr.py:158-181).  Returns weights_sum [N], depth [N], image [N,3]
    (before background) and appends (n_alive, n_step) per trip to ``trace``."""
    N = rays_o.shape[0]
    cascade = 1 + math.ceil(math.log2(hp["bound"]))
    weights_sum = np.zeros(N, f32)
    depth = np.zeros(N, f32)
    image = np.zeros((N, 3), f32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step = 0
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, hp["bound"],
                                        params["density_bitfield"], cascade, hp["grid_size"], nears, fars, 128, False,
                                        dt_gamma, max_steps)
        sigmas, rgbs, _ = head_forward(xyzs, dirs, cond_feat, ind_code, params, hp)
        sigmas = (f32(density_scale) * sigmas).astype(f32)
        if trace is not None:
            trace.append((n_alive, n_step))
        composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        step += n_step
    return weights_sum, depth, image


def _finish(image, weights_sum, depth, nears, fars, bg_color, prefix):
    """renderer.py:385-397."""
    image = image + (f32(1) - weights_sum)[:, None] * bg_color
    image = np.clip(image.reshape(*prefix, 3), 0, 1).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        depth = (np.maximum(depth - nears, f32(0)) / (fars - nears)).astype(f32)
    return image, depth.reshape(*prefix)


def render_head(rays_o, rays_d, cond, params, hp, bg_color=None, dt_gamma=0, max_steps=1024, T_thresh=1e-4,
                eye_area_percent=None, trace=None):
    """NeRFRenderer.render, eval branch (renderer.py:286-399) for RADNeRF."""
    rays_o = np.asarray(rays_o, f32)
    prefix = rays_o.shape[:-1]
    rays_o = np.ascontiguousarray(rays_o.reshape(-1, 3))
    rays_d = np.ascontiguousarray(np.asarray(rays_d, f32).reshape(-1, 3))
    nears, fars = near_far_from_aabb(rays_o, rays_d, params["aabb_infer"], hp["min_near"])
    cond_feat = cal_cond_feat(cond, params, hp, eye_area_percent)
    ind_code = params["individual_embeddings"][0] if hp["individual_embedding_dim"] > 0 else None
    weights_sum, depth, image = march_composite_loop(rays_o, rays_d, nears, fars, cond_feat, ind_code, params, hp,
                                                     dt_gamma, max_steps, T_thresh, trace=trace)
    if bg_color is None:
        bg_color = f32(1)
    else:
        bg_color = np.asarray(bg_color, f32).reshape(-1, 3)
    image_out, depth_out = _finish(image, weights_sum, depth, nears, fars, bg_color, prefix)
    return {"rgb_map": image_out, "depth_map": depth_out, "weights_sum": weights_sum, "head_image": image,
            "nears": nears, "fars": fars, "cond_feat": cond_feat}


# ---------------------------------------------------------------------------------------------
# torso pass (radnerf_torso.py / radnerf_torso_sr.py)
# ---------------------------------------------------------------------------------------------
def grid_sample_2d(grid2d, coords):
    """F.grid_sample(grid.view(1,1,H,W), coords.view(1,-1,1,2), align_corners=True) (bilinear, zeros padding).
    coords[:,0] is consumed as x (width index), coords[:,1] as y (radnerf_torso.py:168)."""
    Hh, Ww = grid2d.shape
    x = np.asarray(coords[:, 0], f32)
    y = np.asarray(coords[:, 1], f32)
    ix = (((x + f32(1)) / f32(2)) * f32(Ww - 1)).astype(f32)
    iy = (((y + f32(1)) / f32(2)) * f32(Hh - 1)).astype(f32)
    ix0 = np.floor(ix)
    iy0 = np.floor(iy)
    ix1 = ix0 + 1
    iy1 = iy0 + 1
    w_nw = (ix1 - ix) * (iy1 - iy)
    w_ne = (ix - ix0) * (iy1 - iy)
    w_sw = (ix1 - ix) * (iy - iy0)
    w_se = (ix - ix0) * (iy - iy0)

    def tap(iy_, ix_):
        ok = (ix_ >= 0) & (ix_ <= Ww - 1) & (iy_ >= 0) & (iy_ <= Hh - 1)
        v = grid2d[np.clip(iy_, 0, Hh - 1).astype(np.int64), np.clip(ix_, 0, Ww - 1).astype(np.int64)]
        return np.where(ok, v, f32(0)).astype(f32)

    out = tap(iy0, ix0) * w_nw + tap(iy0, ix1) * w_ne + tap(iy1, ix0) * w_sw + tap(iy1, ix1) * w_se
    return out.astype(f32)


def forward_torso(x, poses, c, params, hp, image=None, weights_sum=None, lm68=None, sr_variant=False):
    """RADNeRFTorso.forward_torso (radnerf_torso.py:51-84) or, with sr_variant=True,
    RADNeRFTorsowithSR.forward_torso (radnerf_torso_sr.py:75-114).  Returns alpha [P,1], color [P,3], dx [P,2]."""
    x = (np.asarray(x, f32) * f32(hp["torso_shrink"])).astype(f32)
    P = x.shape[0]
    enc_x = freq_encode(x, 10)
    parts = [enc_x]
    if sr_variant:
        lm = np.asarray(lm68, f32).reshape(1, 68, 2)[:, [5, 6, 7, 8, 9, 10, 11]].reshape(1, -1)
        enc_lm68 = freq_encode(lm, 4)
        if c is not None:
            parts.append(np.broadcast_to(np.asarray(c, f32).reshape(1, -1), (P, np.asarray(c).size)))
        parts.append(np.broadcast_to(enc_lm68, (P, enc_lm68.shape[1])))
    else:
        enc_pose = freq_encode(np.asarray(poses, f32).reshape(1, 6), 4)
        parts.append(np.broadcast_to(enc_pose, (P, enc_pose.shape[1])))
        if c is not None:
            parts.append(np.broadcast_to(np.asarray(c, f32).reshape(1, -1), (P, np.asarray(c).size)))
    h = np.concatenate(parts, axis=-1).astype(f32)
    if hp["torso_head_aware"]:
        if image is None:
            image = np.zeros((P, 3), f32)
            weights_sum = np.zeros((P, 1), f32)
        e = np.concatenate([np.asarray(image, f32), np.asarray(weights_sum, f32).reshape(P, 1)], axis=-1)
        e = leaky_relu(linear(e, params["head_color_weights_encoder.0.weight"], params["head_color_weights_encoder.0.bias"]))
        e = leaky_relu(linear(e, params["head_color_weights_encoder.2.weight"], params["head_color_weights_encoder.2.bias"]))
        e = linear(e, params["head_color_weights_encoder.4.weight"], params["head_color_weights_encoder.4.bias"])
        h = np.concatenate([h, e], axis=-1)
    dx = mlp(h, _mlp_weights(params, "torso_deform_net"))
    xd = np.clip(x + dx, -1, 1).astype(f32)
    feat = torso_grid_spec().encode(xd, params["torso_embedder.embeddings"], bound=1)
    h2 = mlp(np.concatenate([feat, h], axis=-1), _mlp_weights(params, "torso_canonicial_net"))
    alpha = sigmoid(h2[..., :1])
    color = sigmoid(h2[..., 1:])
    return alpha, color, dx


def render_torso(rays_o, rays_d, cond, bg_coords, poses, params, hp, bg_color=None, dt_gamma=0, max_steps=1024,
                 T_thresh=1e-4, lm68=None, eye_area_percent=None, sr_variant=False, head_aware_coin=True, trace=None):
    """RADNeRFTorso.render (radnerf_torso.py:86-199), eval branch; sr_variant=True follows
    RADNeRFTorsowithSR.render (radnerf_torso_sr.py:116-231) up to (not including) the SR network.
    ``head_aware_coin`` stands for the reference's ``random.random() < 0.5`` (radnerf_torso.py:177)."""
    rays_o = np.asarray(rays_o, f32)
    prefix = rays_o.shape[:-1]
    rays_o = np.ascontiguousarray(rays_o.reshape(-1, 3))
    rays_d = np.ascontiguousarray(np.asarray(rays_d, f32).reshape(-1, 3))
    bg_coords = np.ascontiguousarray(np.asarray(bg_coords, f32).reshape(-1, 2))
    N = rays_o.shape[0]
    nears, fars = near_far_from_aabb(rays_o, rays_d, params["aabb_infer"], hp["min_near"])
    # non-SR RADNeRFTorso.render calls cal_cond_feat(cond) without eye_area_percent (radnerf_torso.py:106)
    cond_feat = cal_cond_feat(cond, params, hp, eye_area_percent if sr_variant else None)
    ind_code = params["individual_embeddings"][0] if hp["individual_embedding_dim"] > 0 else None
    weights_sum, depth, image = march_composite_loop(rays_o, rays_d, nears, fars, cond_feat, ind_code, params, hp,
                                                     dt_gamma, max_steps, T_thresh, trace=trace)
    if bg_color is None:
        bg_color = f32(1)
    else:
        bg_color = np.asarray(bg_color, f32).reshape(-1, 3)
    torso_code = params["torso_individual_codes"][0] if hp["torso_individual_embedding_dim"] > 0 else None
    gs = hp["grid_size"]
    density_thresh_torso = min(hp["density_thresh_torso"], 0)  # mean_density_torso is not persisted (=0)
    occupancy = grid_sample_2d(np.asarray(params["density_grid_torso"], f32).reshape(gs, gs), bg_coords)
    mask = occupancy > density_thresh_torso
    torso_alpha = np.zeros((N, 1), f32)
    torso_color = np.zeros((N, 3), f32)
    deform = None
    if mask.any():
        if hp["torso_head_aware"] and (sr_variant or head_aware_coin):
            a, col, deform = forward_torso(bg_coords[mask], poses, torso_code, params, hp, image[mask],
                                           weights_sum[:, None][mask], lm68=lm68, sr_variant=sr_variant)
        else:
            a, col, deform = forward_torso(bg_coords[mask], poses, torso_code, params, hp, None, None, lm68=lm68,
                                           sr_variant=sr_variant)
        torso_alpha[mask] = a
        torso_color[mask] = col
    torso_bg = (torso_color * torso_alpha + bg_color * (f32(1) - torso_alpha)).astype(f32)
    image_out, depth_out = _finish(image, weights_sum, depth, nears, fars, torso_bg, prefix)
    return {"rgb_map": image_out, "depth_map": depth_out, "torso_alpha_map": torso_alpha, "torso_rgb_map": torso_bg,
            "deform": deform, "weights_sum": weights_sum, "head_image": image, "mask": mask}


# ---------------------------------------------------------------------------------------------
# training-side kernels (radnerf_oracle.c, second half; reference raymarching.cu:162-820, gridencoder.cu:198-368, 505-609)
# ---------------------------------------------------------------------------------------------
def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, M=None, noises=None, dt_gamma=0.0, max_steps=1024):
    """-> xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] i32, counter [2] i32 (rays visited in index order)."""
    rays_o, rays_d = _c(rays_o, f32).reshape(-1, 3), _c(rays_d, f32).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * max_steps if M is None else int(M)
    xyzs, dirs, deltas = np.zeros((M, 3), f32), np.zeros((M, 3), f32), np.zeros((M, 2), f32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32)
    noises = np.zeros(N, f32) if noises is None else _c(noises, f32)
    nears, fars, grid = _c(nears, f32), _c(fars, f32), _c(density_bitfield, np.uint8)
    lib().orc_march_rays_train(_p(rays_o, _FP), _p(rays_d, _FP), _p(grid, _UP), ctypes.c_float(bound), ctypes.c_float(dt_gamma), ctypes.c_uint32(max_steps),
                               ctypes.c_uint32(N), ctypes.c_uint32(C), ctypes.c_uint32(H), ctypes.c_uint32(M), _p(nears, _FP), _p(fars, _FP), _p(xyzs, _FP),
                               _p(dirs, _FP), _p(deltas, _FP), _p(rays, _IP), _p(counter, _IP), _p(noises, _FP))
    return xyzs, dirs, deltas, rays, counter


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas):
    grad_xyzs, grad_dirs, deltas, rays = _c(grad_xyzs, f32), _c(grad_dirs, f32), _c(deltas, f32), _c(rays, np.int32)
    N, M = rays.shape[0], grad_xyzs.shape[0]
    go, gd = np.zeros((N, 3), f32), np.zeros((N, 3), f32)
    lib().orc_march_rays_train_backward(_p(grad_xyzs, _FP), _p(grad_dirs, _FP), _p(rays, _IP), _p(deltas, _FP), ctypes.c_uint32(N), ctypes.c_uint32(M),
                                        _p(go, _FP), _p(gd, _FP))
    return go, gd


def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, ambient, deltas, rays = _c(sigmas, f32), _c(rgbs, f32), _c(ambient, f32), _c(deltas, f32), _c(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, amb, depth, image = np.zeros(N, f32), np.zeros(N, f32), np.zeros(N, f32), np.zeros((N, 3), f32)
    lib().orc_composite_rays_train_forward(_p(sigmas, _FP), _p(rgbs, _FP), _p(ambient, _FP), _p(deltas, _FP), _p(rays, _IP), ctypes.c_uint32(M), ctypes.c_uint32(N),
                                           ctypes.c_float(T_thresh), _p(ws, _FP), _p(amb, _FP), _p(depth, _FP), _p(image, _FP))
    return ws, amb, depth, image


def composite_rays_train_backward(grad_ws, grad_amb, grad_image, sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image, T_thresh=1e-4):
    a = [_c(v, f32) for v in (grad_ws, grad_amb, grad_image, sigmas, rgbs, ambient, deltas)]
    rays = _c(rays, np.int32)
    weights_sum, ambient_sum, image = _c(weights_sum, f32), _c(ambient_sum, f32), _c(image, f32)
    M, N = a[3].shape[0], rays.shape[0]
    gs, gr, ga = np.zeros(M, f32), np.zeros((M, 3), f32), np.zeros(M, f32)
    lib().orc_composite_rays_train_backward(_p(a[0], _FP), _p(a[1], _FP), _p(a[2], _FP), _p(a[3], _FP), _p(a[4], _FP), _p(a[5], _FP), _p(a[6], _FP), _p(rays, _IP),
                                            _p(weights_sum, _FP), _p(ambient_sum, _FP), _p(image, _FP), ctypes.c_uint32(M), ctypes.c_uint32(N),
                                            ctypes.c_float(T_thresh), _p(gs, _FP), _p(gr, _FP), _p(ga, _FP))
    return gs, gr, ga


def morton3D_dilation(grid, C, H):
    grid = _c(grid, f32).reshape(C, H ** 3)
    out = np.zeros_like(grid)
    lib().orc_morton3D_dilation(_p(grid, _FP), ctypes.c_uint32(C), ctypes.c_uint32(H), _p(out, _FP))
    return out


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _c(rays_o, f32).reshape(-1, 3), _c(rays_d, f32).reshape(-1, 3)
    out = np.zeros((rays_o.shape[0], 2), f32)
    lib().orc_sph_from_ray(_p(rays_o, _FP), _p(rays_d, _FP), ctypes.c_float(radius), ctypes.c_uint32(rays_o.shape[0]), _p(out, _FP))
    return out


def _grid_args(x01, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation):
    x01, embeddings, offsets = _c(x01, f32), _c(embeddings, f32), _c(offsets, np.int32)
    B, D = x01.shape
    C, L = embeddings.shape[1], offsets.shape[0] - 1
    S = f32(np.log2(per_level_scale))
    gt = {"hash": 0, "tiled": 1}[gridtype]
    it = {"linear": 0, "smoothstep": 1}[interpolation]
    return x01, embeddings, offsets, B, D, C, L, S, gt, it


def grid_encode_dydx(x01, embeddings, offsets, per_level_scale, base_resolution=16, gridtype="tiled", align_corners=False, interpolation="linear"):
    """x01 in [0,1] -> dy_dx [B, L, D, C]."""
    x01, embeddings, offsets, B, D, C, L, S, gt, it = _grid_args(x01, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation)
    out = np.zeros((B, L, D, C), f32)
    rc = lib().orc_grid_encode_dydx(_p(x01, _FP), _p(embeddings, _FP), _p(offsets, _IP), _p(out, _FP), ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C),
                                    ctypes.c_uint32(L), ctypes.c_float(S), ctypes.c_uint32(base_resolution), ctypes.c_uint32(gt), ctypes.c_int(int(align_corners)),
                                    ctypes.c_uint32(it))
    assert rc == 0
    return out


def grid_encode_backward(grad, x01, embeddings, offsets, per_level_scale, base_resolution=16, gridtype="tiled", align_corners=False, interpolation="linear",
                         dy_dx=None):
    """grad [L,B,C] -> grad_embeddings [rows, C] (and grad_inputs [B,D] when dy_dx is given)."""
    x01, embeddings, offsets, B, D, C, L, S, gt, it = _grid_args(x01, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation)
    grad = _c(grad, f32)
    ge = np.zeros_like(embeddings)
    rc = lib().orc_grid_encode_backward(_p(grad, _FP), _p(x01, _FP), _p(offsets, _IP), _p(ge, _FP), ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C),
                                        ctypes.c_uint32(L), ctypes.c_float(S), ctypes.c_uint32(base_resolution), ctypes.c_uint32(gt), ctypes.c_int(int(align_corners)),
                                        ctypes.c_uint32(it))
    assert rc == 0
    gi = None
    if dy_dx is not None:
        dy_dx = _c(dy_dx, f32)
        gi = np.zeros((B, D), f32)
        lib().orc_grid_input_backward(_p(grad, _FP), _p(dy_dx, _FP), _p(gi, _FP), ctypes.c_uint32(B), ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L))
    return ge, gi


def grad_total_variation(x01, embeddings, grad, offsets, weight, per_level_scale, base_resolution=16, gridtype="tiled", align_corners=False):
    x01, embeddings, offsets, B, D, C, L, S, gt, _ = _grid_args(x01, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, "linear")
    grad = _c(grad, f32).copy()
    rc = lib().orc_grad_total_variation(_p(x01, _FP), _p(embeddings, _FP), _p(grad, _FP), _p(offsets, _IP), ctypes.c_float(weight), ctypes.c_uint32(B),
                                        ctypes.c_uint32(D), ctypes.c_uint32(C), ctypes.c_uint32(L), ctypes.c_float(S), ctypes.c_uint32(base_resolution),
                                        ctypes.c_uint32(gt), ctypes.c_int(int(align_corners)))
    assert rc == 0
    return grad


def grid_encode_levels(x01, embeddings, offsets, per_level_scale, base_resolution=16, gridtype="tiled", align_corners=False, interpolation="linear"):
    """Level-major forward [L,B,C] for inputs already in [0,1] (the layout the backward kernels' `grad` uses)."""
    return grid_encode_raw(x01, embeddings, offsets, np.log2(per_level_scale), base_resolution, GRIDTYPE[gridtype], align_corners, INTERP[interpolation])


# ---------------------------------------------------------------------------------------------
# SH / frequency encoder gradients (shencoder.cu:125-382, freqencoder.cu:63-93) -- numpy restatements
# ---------------------------------------------------------------------------------------------
def _sh4_monomials():
    """Component k of the degree-<=4 real SH basis as {(i, j, l): coeff} over x^i y^j z^l (the polynomials of kernel_sh,
    shencoder.cu:44-68).  Derivatives are then taken symbolically, so the table below is the only hand-written input."""
    a, b, e = 0.48860251190291987, 1.0925484305920792, 0.54627421529603959
    f, g, h = 0.59004358992664352, 2.8906114426405538, 0.45704579946446572
    k, m = 0.3731763325901154, 1.4453057213202769
    return [
        {(0, 0, 0): 0.28209479177387814},
        {(0, 1, 0): -a}, {(0, 0, 1): a}, {(1, 0, 0): -a},
        {(1, 1, 0): b}, {(0, 1, 1): -b}, {(0, 0, 2): 0.94617469575755997, (0, 0, 0): -0.31539156525251999}, {(1, 0, 1): -b},
        {(2, 0, 0): e, (0, 2, 0): -e},
        {(0, 3, 0): f, (2, 1, 0): -3 * f}, {(1, 1, 1): g}, {(0, 1, 0): h, (0, 1, 2): -5 * h}, {(0, 0, 3): 5 * k, (0, 0, 1): -3 * k},
        {(1, 0, 0): h, (1, 0, 2): -5 * h}, {(2, 0, 1): m, (0, 2, 1): -m}, {(1, 2, 0): 3 * f, (3, 0, 0): -f},
    ]


def sh_encode_dydx(dirs, degree=4):
    """-> [B, 3, degree^2]: d feature / d(x, y, z), coordinates treated as independent (shencoder.cu:125-352)."""
    d = np.asarray(dirs, np.float64).reshape(-1, 3)
    n = degree * degree
    out = np.zeros((d.shape[0], 3, n), np.float64)
    for c, poly in enumerate(_sh4_monomials()[:n]):
        for powers, coeff in poly.items():
            for axis in range(3):
                if powers[axis] == 0:
                    continue
                p = list(powers)
                p[axis] -= 1
                out[:, axis, c] += coeff * powers[axis] * d[:, 0] ** p[0] * d[:, 1] ** p[1] * d[:, 2] ** p[2]
    return out.astype(f32)


def sh_encode_backward(grad, dirs, degree=4):
    """kernel_sh_backward (shencoder.cu:359-382): grad [B, degree^2] -> grad_inputs [B, 3]."""
    return np.einsum("bc,bdc->bd", np.asarray(grad, np.float64), sh_encode_dydx(dirs, degree).astype(np.float64)).astype(f32)


def freq_encode_backward(grad, outputs, D, degree):
    """kernel_freq_backward (freqencoder.cu:63-93): grad, outputs [B, C], C = D + 2 D degree -> grad_inputs [B, D]."""
    g = np.asarray(grad, f32).reshape(-1, D + 2 * D * degree)
    o = np.asarray(outputs, f32).reshape(g.shape)
    res = g[:, :D].copy()
    for k in range(degree):
        s = D + 2 * D * k
        res = res + f32(2.0 ** k) * (g[:, s:s + D] * o[:, s + D:s + 2 * D] - g[:, s + D:s + 2 * D] * o[:, s:s + D])
    return res.astype(f32)


def render_case(case, trace=None):
    """A whole frame of a test case (genefaceplusplus_amd.synthetic.frame_case and its relatives: variant, hp, sd, HW, pose, intr, cond, lm68,
    eye_area_percent, bg_color, T_thresh) through render_head / render_torso, rays from this module's get_rays."""
    hp, sd, HW = case["hp"], case["sd"], case["HW"]
    rays = get_rays(case["pose"], case["intr"], HW, HW)
    kw = dict(bg_color=case["bg_color"], dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"], T_thresh=case["T_thresh"],
              eye_area_percent=case["eye_area_percent"], trace=trace)
    if case["variant"] in ("may_head", "may_head_sr", "audio_head"):
        return render_head(rays["rays_o"], rays["rays_d"], case["cond"], sd, hp, **kw)
    return render_torso(rays["rays_o"], rays["rays_d"], case["cond"], get_bg_coords(HW, HW), convert_poses(case["pose"]),
                        sd, hp, lm68=case["lm68"], sr_variant=(case["variant"] == "may_torso_sr"), **kw)
