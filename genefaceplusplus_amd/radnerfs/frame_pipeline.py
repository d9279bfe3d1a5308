"""Host side of the fused frame pipeline (Section B of include/gfpp_radnerf.h).

``FramePipeline`` is built once per model: it re-packs the MLP weights into MFMA fragment order, uploads the per-level
grid tables, fills the C descriptor structs, and owns the per-resolution device workspaces.  ``render_head`` /
``render_head_torso`` then issue a fixed sequence of C-ABI calls on torch's current stream -- no allocation (after the first
frame of a resolution), no host synchronisation, hence capturable in a hipGraph (see ``GraphedFrame``).
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib, tuning
from .._lib import call, GfppError

c_p = ctypes.c_void_p
c_u32 = ctypes.c_uint32
c_f = ctypes.c_float


class GridLevel(ctypes.Structure):
    _fields_ = [("scale", c_f), ("resolution", c_u32), ("offset", c_u32), ("size", c_u32),
                ("sy", c_u32), ("sz", c_u32), ("mask", c_u32), ("flags", c_u32)]


class GridDesc(ctypes.Structure):
    _fields_ = [("table", c_p), ("levels", c_p), ("dtype", ctypes.c_int32), ("D", c_u32), ("L", c_u32), ("gridtype", c_u32),
                ("interp", c_u32), ("align_corners", c_u32), ("levels_host", c_p), ("row_padded", c_u32)]


class HeadModel(ctypes.Structure):
    _fields_ = [("aabb", c_f * 6), ("min_near", c_f), ("bound", c_f), ("density_scale", c_f), ("cascade", c_u32), ("grid_size", c_u32),
                ("density_bitfield", c_p), ("pos_grid", GridDesc), ("amb_grid", GridDesc),
                ("amb_w0", c_p), ("amb_w0_cond", c_p), ("amb_w1", c_p), ("amb_w2", c_p),
                ("sig_w0", c_p), ("sig_w1", c_p), ("sig_w2_geo", c_p), ("sig_w2_sig", c_p),
                ("col_w0", c_p), ("col_w0_ind", c_p), ("col_w1", c_p), ("cond_dim", c_u32), ("ind_dim", c_u32),
                ("lp_weights", c_p), ("lp_skinny", c_p), ("lp_dtype", ctypes.c_int32), ("occ_aabb", c_f * 6),
                ("pos_grid_blk", GridDesc), ("amb_grid_blk", GridDesc)]


class FrameWs(ctypes.Structure):
    _fields_ = [("N", c_u32), ("nears", c_p), ("fars", c_p), ("ray_state", c_p),
                ("alive", c_p * 2), ("counters", c_p), ("frame_consts", c_p), ("sample_t", c_p), ("sample_cnt", c_p),
                ("sample_stride", c_u32), ("phase_cycles", c_p), ("separate_trips", c_u32),
                ("gcounters", c_p), ("N_global", c_u32), ("trip_first", c_u32), ("trip_count", c_u32), ("full_grid_trips", c_u32),
                ("snapshots", c_p), ("defer_resolve", c_u32), ("resolve_max_steps", c_u32), ("clip_job", c_p), ("clip_lane", c_u32),
                ("clip_sub", c_u32), ("clip_advance", c_u32), ("n_frames", c_u32), ("frame_consts_stride", c_u32), ("timeouts", c_p), ("row_rays", c_u32)]


class CondModel(ctypes.Structure):
    _fields_ = [("smo", c_u32), ("t_win", c_u32), ("c_in", c_u32), ("dim_aud", c_u32), ("strides", c_u32 * 4),
                ("conv_w", c_p * 4), ("conv_b", c_p * 4), ("fc_w", c_p * 2), ("fc_b", c_p * 2),
                ("blink_dim", c_u32), ("blink_emb", c_p), ("blink_w", c_p * 2), ("blink_b", c_p * 2),
                ("with_att", c_u32), ("att_conv_w", c_p * 5), ("att_conv_b", c_p * 5), ("att_fc_w", c_p), ("att_fc_b", c_p),
                ("center_tap_only", c_u32), ("blob", c_p), ("blob_floats", c_u32)]


def cond_train_model(m):
    """(descriptor, parameters, fill) of cal_cond_feat's networks for the training kernels (gfpp_cond_feat_train_forward / _backward): the descriptor points at
    the parameters THEMSELVES (fp32, contiguous: an optimiser step is seen without re-packing), `parameters` lists them in the order their gradients come back,
    fill(addresses) makes the descriptor that says where each gradient goes.  None if the window is outside what the one-workgroup kernels cover."""
    from .cond_nets import _STRIDES
    hp, pre = m.hparams, m.cond_prenet
    dims = (int(m.smo_win_size), int(m.cond_win_size), int(m.cond_in_dim), int(m.cond_out_dim))
    if dims[3] > 64 or dims[0] > 64 or dims[0] * 64 * dims[1] > 8192 or pre.win_size not in _STRIDES:
        return None
    slots, plist = [], []

    def f(name, i, p):
        slots.append((name, i))
        plist.append(p)

    for i in range(4):
        conv = pre.encoder_conv[2 * i]
        f("conv_w", i, conv.weight), f("conv_b", i, conv.bias)
    for i, j in enumerate((0, 2)):
        f("fc_w", i, pre.encoder_fc1[j].weight), f("fc_b", i, pre.encoder_fc1[j].bias)
    blink = int(hp["eye_blink_dim"]) if hp.get("add_eye_blink_cond", False) else 0
    if blink:
        f("blink_emb", None, m.blink_embedding.weight)
        for i in range(2):
            f("blink_w", i, m.blink_encoder[i].weight), f("blink_b", i, m.blink_encoder[i].bias)
    if m.with_att:
        att = m.cond_att_net
        for i in range(5):
            conv = att.attentionConvNet[2 * i]
            f("att_conv_w", i, conv.weight), f("att_conv_b", i, conv.bias)
        f("att_fc_w", None, att.attentionNet[0].weight), f("att_fc_b", None, att.attentionNet[0].bias)

    def fill(addresses):
        cm = CondModel()
        cm.smo, cm.t_win, cm.c_in, cm.dim_aud = dims
        for i, s_ in enumerate(_STRIDES[pre.win_size]):
            cm.strides[i] = s_
        cm.blink_dim, cm.with_att = blink, int(bool(m.with_att))
        for (name, i), addr in zip(slots, addresses):
            if i is None:
                setattr(cm, name, addr)
            else:
                getattr(cm, name)[i] = addr
        return cm

    return fill([p.data_ptr() for p in plist]), plist, fill


class TorsoModel(ctypes.Structure):
    _fields_ = [("density_grid", c_p), ("grid_size", c_u32), ("density_thresh", c_f), ("torso_shrink", c_f), ("variant", c_u32),
                ("code_dim", c_u32), ("const_dim", c_u32), ("head_aware", c_u32), ("grid", GridDesc),
                ("def_w0_x", c_p), ("def_w0_c", c_p), ("def_w0_h", c_p), ("def_w1", c_p), ("def_w2", c_p),
                ("can_w0_g", c_p), ("can_w0_x", c_p), ("can_w0_c", c_p), ("can_w0_h", c_p), ("can_w1", c_p), ("can_w2", c_p),
                ("ha_w0", c_p), ("ha_b0", c_p), ("ha_w1", c_p), ("ha_b1", c_p), ("ha_w2", c_p), ("ha_b2", c_p),
                ("lp_weights", c_p), ("lp_skinny", c_p), ("lp_dtype", ctypes.c_int32)]


_lib.register("gfpp_torso_frame", [ctypes.POINTER(TorsoModel), ctypes.POINTER(FrameWs), c_p, c_p, c_p, c_p, c_f, c_u32, c_p, c_p, c_p, c_p,
                                   c_p, c_p, c_p])
_lib.register("gfpp_cond_feat", [ctypes.POINTER(CondModel), c_p, c_p, c_p, c_p])
_lib.register("gfpp_cond_feat_batch", [ctypes.POINTER(CondModel), c_p, c_u32, c_p, c_u32, c_p, c_u32, c_u32, c_p])
_lib.register("gfpp_cond_feat_train_floats", [ctypes.POINTER(CondModel), ctypes.c_int], restype=ctypes.c_uint32)
_lib.register("gfpp_cond_feat_train_forward", [ctypes.POINTER(CondModel), c_p, c_p, c_p, c_p, c_p])
_lib.register("gfpp_cond_feat_train_backward", [ctypes.POINTER(CondModel), ctypes.POINTER(CondModel), c_p, c_p, c_p, c_p, c_p, c_p])
_lib.register("gfpp_torso_frame_lp", [ctypes.POINTER(TorsoModel), ctypes.POINTER(FrameWs), c_p, c_p, c_p, c_p, c_f, c_u32, c_p, c_p, c_p, c_p,
                                      c_p, c_p, c_p])
_lib.register("gfpp_torso_fold_batch", [ctypes.POINTER(TorsoModel), c_p, c_u32, c_p, c_u32, c_p, c_p])
_lib.register("gfpp_torso_group_lp", [ctypes.POINTER(TorsoModel), ctypes.POINTER(FrameWs), c_p, c_p, c_p, c_p, c_p, c_u32, c_p, c_f, c_u32, c_u32, c_p, c_p, c_p, c_p, c_p, c_p, c_p])
_lib.register("gfpp_torso_mask", [ctypes.POINTER(TorsoModel), c_p, c_u32, c_p, c_p])
_lib.register("gfpp_occupancy_bounds", [c_p, c_u32, c_u32, c_f, c_p, c_p])
_lib.register("gfpp_grid_level_table", [c_u32, c_f, c_u32, c_p, c_p])
_lib.register("gfpp_grid_levels_fill", [c_u32, c_u32, c_f, c_u32, c_u32, ctypes.c_int, c_p, c_u32, c_p])
_lib.register("gfpp_head_frame_begin", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_p, c_p, c_p])
_lib.register("gfpp_head_frame_march", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_fold", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_p])
_lib.register("gfpp_head_frame_fold_batch", [ctypes.POINTER(HeadModel), c_p, c_u32, c_p, c_p, c_u32, c_p])
_lib.register("gfpp_head_frame_premarch", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_p])
_lib.register("gfpp_head_frame_begin_premarch", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_p])
_lib.register("gfpp_head_frame_trips", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_trips_lp", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_march_lp", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_persist_lp", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_persist", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_f, c_u32, c_f, c_p])
_lib.register("gfpp_head_frame_resolve", [ctypes.POINTER(FrameWs), c_u32, c_p])
_lib.register("gfpp_head_group_begin", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_u32, c_f, c_f, c_f, c_f, c_u32, c_u32, c_p, c_p, c_f, c_u32, c_p])
_lib.register("gfpp_head_group_resolve", [ctypes.POINTER(FrameWs), c_u32, c_p])
_lib.register("gfpp_head_eval_samples", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_u32, c_p, c_p, c_p, c_p])
_lib.register("gfpp_head_eval_samples_lp", [ctypes.POINTER(HeadModel), ctypes.POINTER(FrameWs), c_p, c_p, c_u32, c_p, c_p, c_p, c_p])
_lib.register("gfpp_head_frame_finish", [ctypes.POINTER(FrameWs), c_p, c_f, c_p, c_p, c_p])


# ---------------------------------------------------------------------------------------------------------------------
# weight packing (layout documented at gfpp_head_model in include/gfpp_radnerf.h)
# ---------------------------------------------------------------------------------------------------------------------
def _rr(r):
    return (r & 3) + 8 * (r >> 2)


def activation_pairs(width=128):
    """K-step -> (input index fed by lanes 0-31, by lanes 32-63) when the inputs are the previous layer's accumulators."""
    return [(32 * m + _rr(r), 32 * m + _rr(r) + 4) for m in range(width // 32) for r in range(16)]


def encoder_pairs(base, count):
    """Inputs produced by an encoder whose `count` values are split between the two half-waves."""
    half = count // 2
    return [(base + s, base + half + s) for s in range(half)]


def pack_mfma(weight, pairs, shift=0):
    """weight [128, in] (nn.Linear layout) -> float32 [S/4, 4, 64, 4]; `shift` offsets activation indices into `weight`."""
    W = weight.detach().float()
    assert W.shape[0] == 128, "the MFMA path is built for 128-wide layers (4 tiles of 32 rows)"
    S = len(pairs)
    assert S % 4 == 0
    idx = torch.tensor(pairs, dtype=torch.long, device=W.device) + shift          # [S, 2]
    Wk = W[:, idx]                                                                  # [128, S, 2]
    P = Wk.view(4, 32, S, 2).permute(2, 0, 3, 1)                                    # [S, m, half, i]
    P = P.reshape(S // 4, 4, 4, 64).permute(0, 2, 3, 1)                             # [S/4, m, lane, sub]
    return P.contiguous()


def pack_valu(weight):
    """weight [C, 128] -> float32 [2, C, 64] with V[h][c][16*m+r] = W[c][32*m + rr(r) + 4*h]."""
    W = weight.detach().float()
    C = W.shape[0]
    cols = torch.tensor([[32 * m + _rr(r) + 4 * h for m in range(4) for r in range(16)] for h in range(2)], dtype=torch.long, device=W.device)
    return W[:, cols].permute(1, 0, 2).contiguous().view(2, C, 64)


# ---- 16-bit operand image (layout documented at gfpp_head_model.lp_weights) -----------------------------------------------
GFPP_F16, GFPP_BF16 = 1, 2
LP_DTYPES = {"fp16": (GFPP_F16, torch.float16), "bf16": (GFPP_BF16, torch.bfloat16)}


def lp_cols_encoder(base):
    """32 encoder features as 2 steps of K = 16: col[s][h][e].  Half-wave h holds the levels h, h+2, ... (encode_half_lp), so its
    k-th value (k = 8 s + e) is level 2 (k // 2) + h, channel k % 2 = feature 2 level + channel."""
    return [[[base + 2 * (2 * ((8 * s + e) // 2) + h) + (e % 2) for e in range(8)] for h in range(2)] for s in range(2)]


def lp_cols_act(base=0):
    """128 previous-layer activations as 8 steps: step s takes accumulator tile s>>1, registers 8(s&1)..+7."""
    return [[[base + 32 * (s >> 1) + _rr(8 * (s & 1) + e) + 4 * h for e in range(8)] for h in range(2)] for s in range(8)]


def lp_cols_sh():
    return [[[8 * h + e for e in range(8)] for h in range(2)]]


def pack_lp(weight, cols):
    """weight [128, K] (nn.Linear layout, any float dtype) -> [S, 4, 64, 8] with P[s][m][lane][e] = W[32m + (lane&31)][cols[s][lane>>5][e]]."""
    W = weight.detach().double()
    assert W.shape[0] == 128
    idx = torch.tensor(cols, dtype=torch.long, device=W.device)            # [S, 2, 8]
    S = idx.shape[0]
    Wk = W[:, idx]                                                          # [128, S, 2, 8]
    P = Wk.view(4, 32, S, 2, 8).permute(2, 0, 3, 1, 4)                      # [S, m, h, i, e]
    return P.reshape(S, 4, 64, 8).contiguous()


def lp_ambient_dtype(dtype):
    """Operand type of ambient_net inside a 16-bit mode (csrc/frame_head_lp.hip::LpAmbient): its output is a coordinate of the second hash grid, which 8-bit
    significands displace by several cells of the finest level -- the bf16 mode multiplies ambient_net as f16 (steps 0..9 of the weight image, skinny rows 0..2)."""
    return torch.float16 if dtype == torch.bfloat16 else dtype


def _as_bits(t, dtype):
    """round to `dtype`; 16-bit types are returned as their bit patterns (an image may mix f16 and bf16 parts), wider ones (layout tests) as they are."""
    t = t.to(dtype).contiguous()
    return t.view(torch.int16) if t.element_size() == 2 else t


def lp_weight_image(model, dtype):
    """The five wide layers of the head as one [31, 4, 64, 8] tensor of 16-bit patterns (steps: amb0 2 | amb1 8 | sig0 4 | sig1 8 | colour 9): `dtype`, except
    ambient_net's ten steps, which are lp_ambient_dtype(dtype)."""
    A0, A1, _ = (l.weight.detach().double() for l in model.ambient_net.net)
    S0, S1, S2 = (l.weight.detach().double() for l in model.sigma_net.net)
    C0 = model.color_net.net[0].weight.detach().double()
    merged = torch.cat([C0[:, :16], C0[:, 16:144] @ S2[1:129, :]], dim=1)    # [128, 144]: SH columns | geo columns through sigma_net.2
    parts = [pack_lp(A0[:, :32], lp_cols_encoder(0)), pack_lp(A1, lp_cols_act()),
             pack_lp(S0, lp_cols_encoder(0) + lp_cols_encoder(32)), pack_lp(S1, lp_cols_act()),
             pack_lp(merged, lp_cols_sh() + lp_cols_act(16))]
    adt = lp_ambient_dtype(dtype)
    img = torch.cat([_as_bits(p, adt if k < 2 else dtype) for k, p in enumerate(parts)], dim=0)
    assert img.shape == (31, 4, 64, 8)
    return img.contiguous()


def pack_frag(weight, cols, tiles):
    """weight [M <= 32*tiles, K] -> [S, tiles, 64, 8] float64 fragments; cols[s][h][e] = input column, -1 = padding (zero)."""
    W = weight.detach().double()
    Wp = torch.zeros(32 * tiles, W.shape[1] + 1, dtype=torch.float64, device=W.device)
    Wp[:W.shape[0], :W.shape[1]] = W
    idx = torch.tensor(cols, dtype=torch.long, device=W.device)              # [S, 2, 8]
    idx = torch.where(idx < 0, torch.full_like(idx, W.shape[1]), idx)
    S = idx.shape[0]
    Wk = Wp[:, idx]                                                          # [32*tiles, S, 2, 8]
    return Wk.view(tiles, 32, S, 2, 8).permute(2, 0, 3, 1, 4).reshape(S, tiles, 64, 8).contiguous()


def _act_cols(n_steps, base=0):
    """previous-layer activations as operands: step s, half h, element e -> row 32 (s>>1) + rr(8 (s&1) + e) + 4 h."""
    return [[[base + 32 * (s >> 1) + _rr(8 * (s & 1) + e) + 4 * h for e in range(8)] for h in range(2)] for s in range(n_steps)]


def torso_lp_images(m, dtype):
    """(weights [28, 64, 8], skinny [128 x 2]) of `dtype` for gfpp_torso_frame_lp; layout documented at gfpp_torso_model.lp_weights."""
    hp = m.hparams
    D0, D1, D2 = (l.weight.detach().double() for l in m.torso_deform_net.net)
    K0, K1, K2 = (l.weight.detach().double() for l in m.torso_canonicial_net.net)
    head_aware = bool(hp["torso_head_aware"])
    const_dim = D0.shape[1] - 42 - (16 if head_aware else 0)
    dev = D0.device
    freq = lambda base: [[[(base + 16 * s + 8 * h + e) if 16 * s + 8 * h + e < 42 else -1 for e in range(8)] for h in range(2)] for s in range(3)]
    ha_in = lambda base: [[[(base + _rr(e) + 4 * h) if head_aware else -1 for e in range(8)] for h in range(2)]]
    grid = [[[2 * (2 * ((8 * s + e) // 2) + h) + (e % 2) for e in range(8)] for h in range(2)] for s in range(2)]
    frags = []
    if head_aware:
        enc = m.head_color_weights_encoder
        H0, H1, H2 = (enc[i].weight.detach().double() for i in (0, 2, 4))
        frags.append(pack_frag(H0, [[[e if (h == 0 and e < 4) else -1 for e in range(8)] for h in range(2)]], 1))
        frags.append(pack_frag(H1, [[[_rr(e) + 4 * h for e in range(8)] for h in range(2)]], 1))
        frags.append(pack_frag(H2, _act_cols(2), 1))
    else:
        frags.append(torch.zeros(4, 1, 64, 8, dtype=torch.float64, device=dev))
    frags.append(pack_frag(D0, freq(0) + ha_in(42 + const_dim), 2))
    frags.append(pack_frag(D1, _act_cols(4), 2))
    frags.append(pack_frag(K0, grid + freq(32) + ha_in(32 + 42 + const_dim), 1))
    frags.append(pack_frag(K1, _act_cols(2), 1))
    img = torch.cat([f.reshape(-1, 64, 8) for f in frags], dim=0)
    assert img.shape == (28, 64, 8), img.shape
    a4 = torch.tensor(_act_cols(4), dtype=torch.long, device=dev)              # [4, 2, 8]
    a2 = torch.tensor(_act_cols(2), dtype=torch.long, device=dev)
    sk_def = D2[:, a4].permute(2, 0, 1, 3).reshape(2, 2, 32)                   # [h, row, 8 s + e]
    sk_can = K2[:, a2].permute(2, 0, 1, 3).reshape(2, 4, 16)
    skinny = torch.cat([sk_def.reshape(-1), sk_can.reshape(-1)])
    assert skinny.numel() == 256
    return img.to(dtype).contiguous(), skinny.to(dtype).contiguous()


def lp_skinny_image(model, dtype):
    """[2, 7, 64] 16-bit patterns: the skinny output rows in the operand order of the preceding layer's activations; rows 0-2 (ambient_net.2) in
    lp_ambient_dtype(dtype), rows 3-6 in `dtype`."""
    A2 = model.ambient_net.net[2].weight.detach().double()
    S2 = model.sigma_net.net[2].weight.detach().double()
    C1 = model.color_net.net[1].weight.detach().double()
    rows = torch.zeros(7, 128, dtype=torch.float64, device=A2.device)
    rows[:A2.shape[0]] = A2
    rows[3] = S2[0]
    rows[4:7] = C1
    cols = torch.tensor(lp_cols_act(), dtype=torch.long, device=A2.device)      # [8 steps, 2 halves, 8]
    img = rows[:, cols].permute(2, 0, 1, 3).reshape(2, 7, 64)                    # [7, 8, 2, 8] -> [half, row, 64]
    return torch.cat([_as_bits(img[:, :3], lp_ambient_dtype(dtype)), _as_bits(img[:, 3:], dtype)], dim=1).contiguous()


def corner_block_table(emb, offsets, levels):
    """The 16-bit corner-block copy of a 2-channel grid table (csrc/grid_device.h: level_block_issue / level_block_finish; gfpp_head_model.pos_grid_blk /
    amb_grid_blk -- what the 16-bit head kernels read since round 4; layout pinned by tests/test_block_table_cpu.py): row r of level l holds both
    channels of the four corners (r, r + 1, r + sy, r + sy + 1) of the x-y cell that starts at r -- index arithmetic modulo the level size, the neighbours
    gfpp_grid_levels_fill's `sy` / `mask` give -- as 8 halves: c0(x,y) c0(x+1,y) | c1(x,y) c1(x+1,y) | c0(x,y+1) c0(x+1,y+1) | c1(x,y+1) c1(x+1,y+1).
    emb [rows, 2] float tensor, offsets [L+1] ints (unpadded), levels: sequence of objects with sy / mask / size.  Returns [rows, 8] float16."""
    parts = []
    peak = float(emb.abs().max()) if emb.numel() else 0.0
    if not peak < 65504.0:
        raise GfppError(f"corner_block_table: the grid table holds |values| up to {peak:g}, beyond the fp16 range of the 16-bit kernels' table copy "
                        "(render this model with precision='fp32')")
    for l, lv in enumerate(levels):
        T = emb[int(offsets[l]):int(offsets[l + 1])].float()
        size, sy, mask = int(lv.size), int(lv.sy), int(lv.mask)
        assert T.shape == (size, 2)
        r = torch.arange(size, device=T.device, dtype=torch.int64)
        wrap = (lambda k: k & mask) if mask != 0xFFFFFFFF else (lambda k: k % size)      # (no power of two: valid cells never leave the level, the rest is never read)
        i00, i10, i01, i11 = r, wrap(r + 1), wrap(r + sy), wrap(r + sy + 1)
        parts.append(torch.stack([T[i00, 0], T[i10, 0], T[i00, 1], T[i10, 1], T[i01, 0], T[i11, 0], T[i01, 1], T[i11, 1]], dim=1).to(torch.float16))
    return torch.cat(parts, dim=0).contiguous()


def supports(model):
    """The MFMA kernels are specialised for the shipped architecture family (hidden 128, 3/3/2 layers, 16x2 grids)."""
    hp = model.hparams
    ok = (hp["hidden_dim_ambient"] == 128 and hp["hidden_dim_sigma"] == 128 and hp["hidden_dim_color"] == 128 and hp["geo_feat_dim"] == 128
          and hp["num_layers_ambient"] == 3 and hp["num_layers_sigma"] == 3 and hp["num_layers_color"] == 2
          and hp["ambient_coord_dim"] in (2, 3) and model.position_embedder.num_levels == 16 and model.position_embedder.level_dim == 2)
    return ok


# HIP multiplexes a process's streams onto a handful of hardware queues (4 by default, round-robin at creation); two streams that land on the same
# queue serialise.  Every renderer / pipeline of a process therefore shares ONE set of streams per (device, role, lane) instead of drawing fresh
# ones from torch's pool: a second model in the same process (bench.py's SR mode, several identities on one GPU) keeps the lane <-> queue mapping
# of the first (measured: a second ClipRenderer with its own streams ran 30 % slower than alone).
_STREAMS = {}


def shared_stream(device, role, lane=0):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), role, int(lane))
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class fingerprint_memo:
    """`with fingerprint_memo(model):` -- FramePipeline._fingerprint(model) walks the module tree once inside the block (ClipRenderer.start() asks for it three times:
    the clip's conditioning cache, pipeline() of frame_consts_rows, pipeline() of group_supported; nothing can change the parameters in between: one host thread, no
    optimizer step inside start()).  ~0.05 ms of a short job's 0.2 ms start-up."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.outer = self.model.__dict__.get("_gfpp_fingerprint_memo")
        if self.outer is None:
            self.model.__dict__["_gfpp_fingerprint_memo"] = [None]
        return self

    def __exit__(self, *exc):
        if self.outer is None:
            self.model.__dict__.pop("_gfpp_fingerprint_memo", None)
        return False


class FoldedConsts:
    """A frame's 256 folded constants, computed ahead of the frame (FramePipeline.fold_rows): head_pass points the workspace at them instead of running
    the conditioning networks and the fold."""

    def __init__(self, consts):
        self.consts = consts


class GraphedFrame:
    """One frame of C-ABI launches captured in a hipGraph (torch.cuda.CUDAGraph is a hipGraph on ROCm) and replayed.

    The per-frame inputs (rays, conditioning window, landmarks, pose, background) are copied into static device buffers, the graph
    is replayed, and the result tensors are the graph's static outputs: they are overwritten by the next frame of the same shape,
    which is how the reference's caller uses them (it moves every frame to the host right away, genefacepp_infer.py:465-469).
    Host cost per frame: a handful of small device-to-device copies + one graph launch instead of ~25 launches and their Python."""

    def __init__(self, fn, inputs, copy_inputs=True, before_capture=None):
        """copy_inputs=False: `inputs` already ARE the static buffers (the caller refreshes them itself before `graph.replay()`).
        before_capture: called after the warm-up runs and before the capture (launch parameters that are learnt from a rendered frame)."""
        self.fn = fn
        self.static = {k: (v.detach().clone() if torch.is_tensor(v) and copy_inputs else v) for k, v in inputs.items()}
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(2):                       # warm-up: allocates the workspaces, packs weights, loads code objects
                fn(**self.static)
        torch.cuda.current_stream().wait_stream(stream)
        if before_capture is not None:
            before_capture()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (RCCL's watchdog in multi-GPU runs, a video writer) may keep calling into HIP while this
        # thread captures; only this thread's calls have to be capturable
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.out = fn(**self.static)

    def matches(self, inputs):
        for k, v in inputs.items():
            sv = self.static.get(k)
            if torch.is_tensor(v) != torch.is_tensor(sv):
                return False
            if torch.is_tensor(v):
                if v.shape != sv.shape or v.dtype != sv.dtype:
                    return False
            elif v != sv:
                return False
        return True

    def __call__(self, inputs):
        for k, v in inputs.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.out


#: header struct name -> ctypes mirror (checked against the library's own sizeof the first time a pipeline is built, _lib.check_struct)
STRUCT_MIRRORS = {"grid_level": GridLevel, "grid_desc": GridDesc, "head_model": HeadModel, "frame_ws": FrameWs, "cond_model": CondModel,
                  "torso_model": TorsoModel}
_layout_checked = False


def check_layout():
    global _layout_checked
    if not _layout_checked:
        for name, mirror in STRUCT_MIRRORS.items():
            _lib.check_struct(name, mirror)
        _layout_checked = True


class FramePipeline:
    def __init__(self, model):
        check_layout()
        if not supports(model):
            raise GfppError("fused pipeline: unsupported architecture (needs hidden 128, layers 3/3/2, 16x2 grids); "
                            "set model.executor = 'staged'")
        self.device = model.density_bitfield.device
        if self.device.type != "cuda":
            raise GfppError("fused pipeline: the model must live on the GPU (there is no CPU path)")
        _lib.lib()
        self._keep = []            # device tensors referenced by raw pointers in the descriptors
        self._lp_images = {}
        self.precision = "fp32"
        self._graphs = {}
        self._side_stream = {}
        self._torso_pixels = {}
        #: which of several independent workspaces (and side streams) the next frame uses: frames of different lanes may be in flight at
        #: the same time on different streams (clip.ClipRenderer(lanes=2)); weights and tables are shared
        self.lane = 0
        self._versions = self._fingerprint(model)
        self.head = self._build_head(model)
        self.torso = self._build_torso(model) if hasattr(model, "torso_deform_net") else None
        self.cond = self._build_cond(model)
        self._ws = {}

    # -- change detection ----------------------------------------------------------------------------------------
    @staticmethod
    def _fingerprint(model):
        """(address, version) of every parameter and buffer.  Called on every model.pipeline() -- several times per rendered frame -- so the module tree is
        not walked through nn.Module.parameters() / buffers() (recursive generators with a de-duplication set: ~0.25 ms for this model, a fifth of a frame at
        the per-frame API's rate, and most of ClipRenderer.start()'s host time): the list of sub-modules is cached on the model and validated by identity of
        every module's children (a replaced or added sub-module rebuilds it); parameters and buffers are read from the modules' own dictionaries, so a replaced
        Parameter object, a moved tensor (.to()) and an in-place update (optimizer step, load_state_dict) all change the fingerprint as before."""
        memo = model.__dict__.get("_gfpp_fingerprint_memo")
        if memo is not None:                        # inside `with fingerprint_memo(model)`: one walk for a whole ClipRenderer.start()
            if memo[0] is None:
                memo[0] = FramePipeline._fingerprint_walk(model)
            return memo[0]
        return FramePipeline._fingerprint_walk(model)

    @staticmethod
    def _fingerprint_walk(model):
        cache = model.__dict__.get("_gfpp_module_list")
        if cache is None or any(tuple(m._modules.values()) != kids for m, kids in cache):
            cache = [(m, tuple(m._modules.values())) for m in model.modules()]
            model.__dict__["_gfpp_module_list"] = cache
        out = []
        for m, _ in cache:
            for t in m._parameters.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
            for t in m._buffers.values():
                if t is not None:
                    out.append((t.data_ptr(), t._version))
        return tuple(out)

    def matches(self, model):
        return model.density_bitfield.device == self.device and self._fingerprint(model) == self._versions

    # -- descriptors ---------------------------------------------------------------------------------------------
    def _hold(self, t):
        t = t.contiguous()
        self._keep.append(t)
        return t.data_ptr()

    def _grid_desc(self, enc):
        if enc.level_dim != 2 or enc.num_levels != 16:
            raise GfppError("fused pipeline: grids must have 16 levels x 2 channels")
        L = enc.num_levels
        off = np.ascontiguousarray(enc.offsets.cpu().numpy().astype(np.int32))
        lv = (GridLevel * L)()
        call("gfpp_grid_levels_fill", int(enc.input_dim), L, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), int(enc.gridtype_id),
             int(enc.align_corners), off.ctypes.data, 1, lv)
        levels = torch.from_numpy(np.frombuffer(lv, dtype=np.uint8).reshape(L, ctypes.sizeof(GridLevel)).copy()).to(self.device)
        d = GridDesc()
        # padded copy: every level is followed by a repeat of its first row (see gfpp_grid_desc.row_padded)
        emb = enc.embeddings.detach().float()
        parts = []
        for l in range(L):
            parts += [emb[off[l]:off[l + 1]], emb[off[l]:off[l] + 1]]
        d.table = self._hold(torch.cat(parts, dim=0))
        self._keep.append(lv)
        d.levels_host = ctypes.addressof(lv)
        d.row_padded = 1
        d.levels = self._hold(levels)
        d.dtype = 0
        d.D, d.L = enc.input_dim, L
        d.gridtype, d.interp, d.align_corners = enc.gridtype_id, enc.interp_id, int(enc.align_corners)
        return d

    def _grid_desc_block(self, enc):
        """The grid as a 16-bit corner-block table for the 16-bit head kernels (gfpp_head_model.*_grid_blk: f16, row_padded = 2), or an empty descriptor if a
        level needs the hash / a true modulo (the kernels' generic lookup then reads the fp32 table)."""
        L = enc.num_levels
        off = np.ascontiguousarray(enc.offsets.cpu().numpy().astype(np.int32))
        lv = (GridLevel * L)()
        call("gfpp_grid_levels_fill", int(enc.input_dim), L, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), int(enc.gridtype_id),
             int(enc.align_corners), off.ctypes.data, 0, lv)
        d = GridDesc()
        # GFPP_LP_BLOCK_TABLE=0 (round-4 advisory: an opt-out for A/B and parity debugging): no 16-bit copy of the tables -- the 16-bit kernels then run their generic
        # lookup on the fp32 tables (fp32 corner values and weights, 2 060 B per sample), the instantiation hash-grid models use anyway
        if any(int(lv[l].flags) & 1 for l in range(L)) or not tuning.HOST["lp_block_table"]:                 # GFPP_LEVEL_SLOW
            return d
        levels = torch.from_numpy(np.frombuffer(lv, dtype=np.uint8).reshape(L, ctypes.sizeof(GridLevel)).copy()).to(self.device)
        d.table = self._hold(corner_block_table(enc.embeddings.detach(), off, lv))
        self._keep.append(lv)
        d.levels_host = ctypes.addressof(lv)
        d.row_padded = 2
        d.levels = self._hold(levels)
        d.dtype = 1                       # GFPP_F16
        d.D, d.L = enc.input_dim, L
        d.gridtype, d.interp, d.align_corners = enc.gridtype_id, enc.interp_id, int(enc.align_corners)
        return d

    def _build_head(self, m):
        hm = HeadModel()
        aabb = m.aabb_infer.detach().cpu().numpy().astype(np.float32)
        for i in range(6):
            hm.aabb[i] = float(aabb[i])
        hm.min_near, hm.bound, hm.density_scale = float(m.min_near), float(m.bound), float(m.density_scale)
        hm.cascade, hm.grid_size = int(m.cascade), int(m.grid_size)
        hm.density_bitfield = self._hold(m.density_bitfield)
        # bounds of the occupied cells: the pre-march stops a ray where it leaves them (same samples, far fewer empty cells walked)
        for i in range(6):
            hm.occ_aabb[i] = 0.0
        if hm.density_bitfield % 4 == 0:
            out6 = torch.empty(6, dtype=torch.float32, device=self.device)
            call("gfpp_occupancy_bounds", hm.density_bitfield, hm.cascade, hm.grid_size, hm.bound, out6.data_ptr(), torch.cuda.current_stream().cuda_stream)
            for i, v in enumerate(out6.cpu().tolist()):
                hm.occ_aabb[i] = v
        hm.pos_grid = self._grid_desc(m.position_embedder)
        hm.amb_grid = self._grid_desc(m.ambient_embedder)
        hm.pos_grid_blk, hm.amb_grid_blk = GridDesc(), GridDesc()         # built with the first 16-bit precision (set_precision): the fp32 mode never reads them
        act = activation_pairs()
        enc32 = encoder_pairs(0, 32)
        A0, A1, A2 = (l.weight for l in m.ambient_net.net)
        S0, S1, S2 = (l.weight for l in m.sigma_net.net)
        C0, C1 = (l.weight for l in m.color_net.net)
        hm.amb_w0 = self._hold(pack_mfma(A0, enc32))
        hm.amb_w0_cond = self._hold(A0.detach().float()[:, 32:])
        hm.amb_w1 = self._hold(pack_mfma(A1, act))
        hm.amb_w2 = self._hold(pack_valu(A2))
        hm.sig_w0 = self._hold(pack_mfma(S0, enc32 + encoder_pairs(32, 32)))
        hm.sig_w1 = self._hold(pack_mfma(S1, act))
        hm.sig_w2_geo = self._hold(pack_mfma(S2[1:], act))
        hm.sig_w2_sig = self._hold(pack_valu(S2[:1]))
        hm.col_w0 = self._hold(pack_mfma(C0, encoder_pairs(0, 16) + [(16 + a, 16 + b) for a, b in act]))
        ind_dim = int(m.individual_embedding_dim)
        hm.col_w0_ind = self._hold(C0.detach().float()[:, 144:]) if ind_dim > 0 else None
        hm.col_w1 = self._hold(pack_valu(C1))
        hm.cond_dim, hm.ind_dim = int(A0.shape[1] - 32), ind_dim
        hm.lp_weights, hm.lp_skinny, hm.lp_dtype = None, None, 0
        return hm

    def set_precision(self, model, precision):
        """'fp32' (exact-fp32 MFMA) | 'fp16' | 'bf16' (16-bit MFMA operands, fp32 accumulation; weights repacked on first use)."""
        if precision == "fp32":
            self.precision = "fp32"
            if self.torso is not None and self.fp32_torso == "mfma":
                # the torso MLPs on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 = an fp32 fma chain): the same fragment image as the 16-bit modes, in fp32
                if "torso_fp32" not in self._lp_images:
                    w, sk = torso_lp_images(model, torch.float32)
                    self._lp_images["torso_fp32"] = (w.to(self.device), sk.to(self.device))
                self.torso.lp_weights = self._lp_images["torso_fp32"][0].data_ptr()
                self.torso.lp_skinny = self._lp_images["torso_fp32"][1].data_ptr()
                self.torso.lp_dtype = 0                     # GFPP_F32
            return
        if precision not in LP_DTYPES:
            raise GfppError(f"precision must be 'fp32', 'fp16' or 'bf16', got {precision!r}")
        if precision not in self._lp_images:
            dt = LP_DTYPES[precision][1]
            self._lp_images[precision] = (lp_weight_image(model, dt).to(self.device), lp_skinny_image(model, dt).to(self.device))
        self.head.lp_weights = self._lp_images[precision][0].data_ptr()
        self.head.lp_skinny = self._lp_images[precision][1].data_ptr()
        if "block_grids" not in self._lp_images:
            # the 16-bit head kernels read the two tables as 16-bit corner-block copies (2 x the fp32 bytes: 14.5 MB per 3-D grid); the fp32 kernels keep pos_grid / amb_grid
            self._lp_images["block_grids"] = (self._grid_desc_block(model.position_embedder), self._grid_desc_block(model.ambient_embedder))
        self.head.pos_grid_blk, self.head.amb_grid_blk = self._lp_images["block_grids"]
        if self.torso is not None:
            key = "torso_" + precision
            if key not in self._lp_images:
                w, sk = torso_lp_images(model, LP_DTYPES[precision][1])
                self._lp_images[key] = (w.to(self.device), sk.to(self.device))
            self.torso.lp_weights = self._lp_images[key][0].data_ptr()
            self.torso.lp_skinny = self._lp_images[key][1].data_ptr()
            self.torso.lp_dtype = LP_DTYPES[precision][0]
        self.head.lp_dtype = LP_DTYPES[precision][0]
        self.precision = precision

    def _build_cond(self, m):
        """Descriptor of cal_cond_feat's networks (None if the shape is outside what the one-workgroup kernel covers)."""
        from .cond_nets import _STRIDES
        hp = m.hparams
        pre = m.cond_prenet
        cm = CondModel()
        cm.smo, cm.t_win, cm.c_in, cm.dim_aud = int(m.smo_win_size), int(m.cond_win_size), int(m.cond_in_dim), int(m.cond_out_dim)
        if cm.dim_aud > 64 or cm.smo > 64 or cm.smo * 64 * cm.t_win > 8192:
            return None
        # all weights go into ONE contiguous blob (the kernel pulls it into LDS in one burst); every piece starts 16-byte aligned
        pieces = []

        def f(t):
            t = t.detach().float().reshape(-1)
            pad = (-t.numel()) % 4
            pieces.append((t, pad))
            return len(pieces) - 1

        center = cm.t_win == 1
        cm.center_tap_only = int(center)
        slots = {}
        for i, s_ in enumerate(_STRIDES[pre.win_size]):
            cm.strides[i] = s_
            conv = pre.encoder_conv[2 * i]
            slots[("conv_w", i)] = f(conv.weight[:, :, 1] if center else conv.weight)
            slots[("conv_b", i)] = f(conv.bias)
        for i, j in enumerate((0, 2)):
            slots[("fc_w", i)], slots[("fc_b", i)] = f(pre.encoder_fc1[j].weight), f(pre.encoder_fc1[j].bias)
        if hp.get("add_eye_blink_cond", False):
            cm.blink_dim = int(hp["eye_blink_dim"])
            slots[("blink_emb", None)] = f(m.blink_embedding.weight[0])
            for i in range(2):
                slots[("blink_w", i)], slots[("blink_b", i)] = f(m.blink_encoder[i].weight), f(m.blink_encoder[i].bias)
        cm.with_att = int(bool(m.with_att))
        if m.with_att:
            att = m.cond_att_net
            for i in range(5):
                conv = att.attentionConvNet[2 * i]
                slots[("att_conv_w", i)], slots[("att_conv_b", i)] = f(conv.weight), f(conv.bias)
            slots[("att_fc_w", None)], slots[("att_fc_b", None)] = f(att.attentionNet[0].weight), f(att.attentionNet[0].bias)
        blob = torch.cat([torch.cat([t, t.new_zeros(pad)]) for t, pad in pieces]).contiguous()
        self._keep.append(blob)
        offs, at = [], 0
        for t, pad in pieces:
            offs.append(at)
            at += t.numel() + pad
        base = blob.data_ptr()
        for (name, i), k in slots.items():
            addr = base + 4 * offs[k]
            if i is None:
                setattr(cm, name, addr)
            else:
                getattr(cm, name)[i] = addr
        cm.blob, cm.blob_floats = base, int(blob.numel())
        return cm

    def cond_feat(self, cond, eye_area_percent=None):
        """cal_cond_feat on the device in one launch -> [cond_out_dim] (or [smo, cond_out_dim] without the attention net)."""
        cond = self._dev_f32(cond, "cond")
        cm = self.cond
        if tuple(cond.shape) != (cm.smo, cm.t_win, cm.c_in):
            raise GfppError(f"cond must be [{cm.smo}, {cm.t_win}, {cm.c_in}], got {tuple(cond.shape)}")
        eye = None
        if cm.blink_dim and eye_area_percent is not None:
            eye = self._dev_f32(eye_area_percent.reshape(-1)[:1], "eye_area_percent")
        out = torch.empty(cm.dim_aud if cm.with_att else (cm.smo, cm.dim_aud), dtype=torch.float32, device=self.device)
        call("gfpp_cond_feat", ctypes.byref(cm), cond.data_ptr(), eye.data_ptr() if eye is not None else None, out.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        return out

    def cond_feat_rows(self, rows, cond_at, eye_at, count):
        """cal_cond_feat for `count` frames whose driving signals are the rows of one packed matrix (`rows` [F, row_floats] f32 on the device; the window
        starts at column `cond_at`, the eye value -- or None -- at column `eye_at`) in one launch -> [count, cond_out] (flattened [smo * dim] rows without
        the attention net)."""
        cm = self.cond
        if rows.dtype != torch.float32 or not rows.is_contiguous() or rows.dim() != 2 or rows.shape[0] < count:
            raise GfppError("cond_feat_rows: rows must be a contiguous float32 [F, row_floats] matrix")
        width = cm.dim_aud if cm.with_att else cm.smo * cm.dim_aud
        out = torch.empty(count, width, dtype=torch.float32, device=self.device)
        stride = int(rows.shape[1])
        use_eye = cm.blink_dim and eye_at is not None
        call("gfpp_cond_feat_batch", ctypes.byref(cm), rows.data_ptr() + 4 * int(cond_at), stride, (rows.data_ptr() + 4 * int(eye_at)) if use_eye else None,
             stride if use_eye else 0, out.data_ptr(), width, int(count), torch.cuda.current_stream().cuda_stream)
        return out

    def fold_rows(self, cond_feats, ind_code):
        """gfpp_head_frame_fold for every row of cond_feats [F, >= cond_dim] (one launch) -> frame constants [F, 256]."""
        if cond_feats.dtype != torch.float32 or not cond_feats.is_contiguous() or cond_feats.dim() != 2 or cond_feats.shape[1] < self.head.cond_dim:
            raise GfppError("fold_rows: cond_feats must be a contiguous float32 [F, >= cond_dim] matrix")
        F = int(cond_feats.shape[0])
        out = torch.empty(F, 256, dtype=torch.float32, device=self.device)
        ind = self._dev_f32(ind_code.reshape(-1), "ind_code") if ind_code is not None else None
        for first in range(0, F, 65535):
            n = min(65535, F - first)
            call("gfpp_head_frame_fold_batch", ctypes.byref(self.head), cond_feats[first:].data_ptr(), int(cond_feats.shape[1]), ind.data_ptr() if ind is not None else None,
                 out[first:].data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        return out

    def _build_torso(self, m):
        hp = m.hparams
        tm = TorsoModel()
        tm.density_grid = self._hold(m.density_grid_torso.detach().float())
        tm.grid_size = int(m.grid_size)
        tm.density_thresh = float(min(m.density_thresh_torso, m.mean_density_torso))
        tm.torso_shrink = float(hp["torso_shrink"])
        tm.variant = 1 if m.landmark_conditioned else 0
        tm.code_dim = int(m.torso_individual_embedding_dim)
        tm.const_dim = (126 if m.landmark_conditioned else 54) + tm.code_dim
        tm.head_aware = int(bool(hp["torso_head_aware"]))
        tm.grid = self._grid_desc(m.torso_embedder)
        D0, D1, D2 = (l.weight.detach().float() for l in m.torso_deform_net.net)
        K0, K1, K2 = (l.weight.detach().float() for l in m.torso_canonicial_net.net)
        if D0.shape[0] != 64 or K0.shape[0] != 32 or D0.shape[1] != 42 + tm.const_dim + 16 * tm.head_aware:
            raise GfppError("fused pipeline: unexpected torso MLP shapes")
        c0, c1 = 42, 42 + tm.const_dim
        kt = lambda w: w.t().contiguous()           # [out,in] -> k-major [in,out]
        tm.def_w0_x = self._hold(kt(D0[:, :c0]))
        tm.def_w0_c = self._hold(D0[:, c0:c1])
        tm.def_w1 = self._hold(kt(D1))
        tm.def_w2 = self._hold(kt(D2))
        tm.can_w0_g = self._hold(kt(K0[:, :32]))
        tm.can_w0_x = self._hold(kt(K0[:, 32:32 + c0]))
        tm.can_w0_c = self._hold(K0[:, 32 + c0:32 + c1])
        tm.can_w1 = self._hold(kt(K1))
        tm.can_w2 = self._hold(kt(K2))
        if tm.head_aware:
            tm.def_w0_h = self._hold(kt(D0[:, c1:]))
            tm.can_w0_h = self._hold(kt(K0[:, 32 + c1:]))
            enc = m.head_color_weights_encoder
            for i, name in zip((0, 2, 4), ("0", "1", "2")):
                setattr(tm, "ha_w" + name, self._hold(kt(enc[i].weight.detach().float())))
                setattr(tm, "ha_b" + name, self._hold(enc[i].bias.detach().float()))
        tm.lp_weights, tm.lp_skinny, tm.lp_dtype = None, None, 0
        return tm

    # -- per-resolution workspace ------------------------------------------------------------------------------------
    def workspace(self, N):
        ent = self._ws.get((N, self.lane))
        if ent is None:
            dev = self.device
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            t = {"nears": f(N), "fars": f(N), "ray_state": f(N, 8),       # one 32-byte record per ray (gfpp_frame_ws.ray_state)
                 "alive0": torch.empty(N, dtype=torch.int32, device=dev), "alive1": torch.empty(N, dtype=torch.int32, device=dev),
                 "counters": torch.zeros(192, dtype=torch.int32, device=dev), "frame_consts": f(256),
                 "out_image": f(N, 3), "out_depth": f(N)}
            self._ws_bytes = sum(v.numel() * v.element_size() for v in t.values())
            ws = FrameWs()
            ws.N = N
            for k in ("nears", "fars", "ray_state", "counters", "frame_consts"):
                setattr(ws, k, t[k].data_ptr())
            ws.alive[0], ws.alive[1] = t["alive0"].data_ptr(), t["alive1"].data_ptr()
            ws.phase_cycles = None
            ws.sample_t, ws.sample_cnt, ws.sample_stride = None, None, 0
            # lanes other than 0 only exist when several frames are in flight: no multi-trip launches then (see gfpp_frame_ws.separate_trips)
            ws.separate_trips = 0
            ws.gcounters, ws.N_global, ws.trip_first, ws.trip_count = None, 0, 0, 0
            ws.full_grid_trips = 0
            ws.snapshots = None
            ws.defer_resolve, ws.resolve_max_steps, ws.clip_job, ws.clip_lane = 0, 0, None, 0
            ws.clip_sub, ws.clip_advance, ws.n_frames, ws.frame_consts_stride = 0, 0, 0, 0
            ws.row_rays = 0
            t["timeouts"] = torch.zeros(1, dtype=torch.int32, device=dev)     # sticky: no kernel resets it (gfpp_frame_ws.timeouts)
            ws.timeouts = t["timeouts"].data_ptr()
            ent = (ws, t)
            self._ws[(N, self.lane)] = ent
        if self.frames_in_flight <= 1:
            ent[0].full_grid_trips = 0
            ent[0].separate_trips = self.separate_trips or 0
        elif ent[0].full_grid_trips and self.frames_in_flight * 32 <= self.cu_count:
            # calibrated (calibrate_trip_launches): the trips the frames normally need are launches of their own, the rest is ONE multi-trip launch
            # on 32 workgroups.  Launches of that size cannot starve each other at their device-wide barriers (all lanes' together fit the device
            # several times), and the usual one finds nothing left and returns before any barrier
            ent[0].separate_trips = ent[0].full_grid_trips
        else:
            # every trip a launch of its own: two full-width multi-trip launches spinning at their barriers could keep each other's workgroups
            # from ever becoming resident (see gfpp_frame_ws.separate_trips)
            ent[0].separate_trips = 0xFFFF
        return ent

    @property
    def cu_count(self):
        return torch.cuda.get_device_properties(self.device).multi_processor_count

    def calibrate_trip_launches(self, N, margin=tuning.HOST["trip_margin"]):
        """Several frames in flight issue every possible trip as a launch of its own (16 for the shipped max_steps); most of them find nothing left
        (the step budget of renderer.py:364 is used up after ~6 trips) and cost ~2 us each of every frame.  Call this after a frame of the clip has
        been rendered on this lane (synchronises): the trips beyond the ones that frame used (+ margin) then become ONE multi-trip launch on a small
        grid (gfpp_frame_ws.full_grid_trips / separate_trips).  Results never depend on it: a later frame that needs more trips is rendered by the
        small grid.  Returns the number of full-grid trips."""
        ws, t = self.workspace(N)
        if self.lp_kernel == "persist" and (self.precision != "fp32" or self.fp32_kernel == "wave"):
            ws.full_grid_trips = 0                       # one launch per frame: nothing to calibrate
            return 0
        torch.cuda.synchronize(self.device)
        used = int((t["counters"][64:127] > 0).sum().item())
        ws.full_grid_trips = max(1, used + int(margin)) if self.frames_in_flight > 1 else 0
        return int(ws.full_grid_trips)

    #: exact-fp32 mode: 'wave' = autonomous wavefronts over pre-marched samples (gfpp_head_frame_trips), 'tile' = the workgroup-synchronous
    #: kernel that marches inside the trip (gfpp_head_frame_march); same bits per sample
    fp32_kernel = "wave"

    #: exact-fp32 mode, torso pass: 'mfma' = the MFMA kernel with fp32 fragments (gfpp_torso_frame_lp, lp_dtype GFPP_F32; round 3), 'valu' = one
    #: thread per pixel on the vector ALU (gfpp_torso_frame, 170 us per 512^2 frame; the A/B partner)
    fp32_torso = tuning.HOST["fp32_torso"]

    #: slab test + state reset + pre-march as one launch (gfpp_head_frame_begin_premarch); False = the two separate launches (tests compare them)
    fuse_begin = tuning.HOST["fuse_begin"]

    #: 16-bit kernel: trips with a launch of their own before the multi-trip launch (None / 0 = the library default, 6); tests vary it
    separate_trips = None

    #: 'persist' = the whole loop as ONE launch with workgroup-local trips (gfpp_head_frame_persist_lp, the production path of the 16-bit modes since
    #: round 3; gfpp_head_frame_persist for the exact-fp32 mode since round 4), 'trips' = one launch per trip (gfpp_head_frame_trips_lp / _trips, the A/B
    #: partners; also taken for max_steps > 24 or more than 2^22 rays)
    lp_kernel = tuning.HOST["lp_kernel"]

    #: set > 1 by a caller that keeps frames of several lanes in flight at once (ClipRenderer)
    frames_in_flight = 1

    # -- frames ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _dev_f32(t, name):
        if not t.is_cuda:
            raise GfppError(f"{name} must be on the GPU")
        return t.detach().float().contiguous()

    #: (device pointer of a gfpp_clip_job, lane) while a clip renderer issues / captures a frame: the torso kernel of the 16-bit modes then stores the
    #: uint8 frame itself and advances the job's cursor (sets clip_job_consumed); None otherwise
    clip_job = None
    clip_job_consumed = False
    #: resolve-in-consumer and uint8-store-in-torso (GFPP_FUSE_TAIL: "0" none, "1" both, "resolve" / "store" one of them).  Measured same-box (round 3,
    #: 512^2 bf16, two frames in flight): none 2 860-2 906 frames/s, resolve only 2 830-2 883 (neutral: the 6 us launch it saves is hidden by the other
    #: frame in flight), store fused 2 640-2 660 (the torso kernel itself gets 45 us longer) -- so the default keeps the two small launches
    fuse_tail = tuning.HOST["fuse_tail"]

    def head_pass(self, rays_o, rays_d, cond_feat, ind_code, dt_gamma, max_steps, T_thresh, shard=None, defer_resolve=False):
        """near/far + constant folding + the whole march/evaluate/composite loop; leaves the result in the workspace.

        shard = (process group, rays of the whole frame): `rays_o/rays_d` are ONE TILE of a frame that several GPUs render together.  The
        sample budget of a ray depends on the frame-wide alive count (renderer.py:364), so the trips are issued one by one and the alive counts
        of all tiles are summed (one int32 all_reduce per trip, RCCL) into `gcounters`, which the next trip's kernels read -- every ray then
        gets exactly the samples it gets in a single-GPU frame.

        `cond_feat` is a tensor, or a callable returning it: the callable (the conditioning networks) is then issued on a side
        stream together with the constant folding, next to the slab test and the pre-march on the main stream, which do not depend
        on it (fork / join, also inside a captured graph)."""
        rays_o = self._dev_f32(rays_o, "rays_o")
        rays_d = self._dev_f32(rays_d, "rays_d")
        N = rays_o.shape[0]
        ws, t = self.workspace(N)
        ind = self._dev_f32(ind_code.reshape(-1), "ind_code") if ind_code is not None else None
        ind_ptr = ind.data_ptr() if ind is not None else None
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        lp = self.precision != "fp32"
        premarched = lp or self.fp32_kernel == "wave"
        if premarched and ws.sample_stride < int(max_steps) + 7:
            stride = (int(max_steps) + 7 + 7) // 8 * 8
            t["sample_t"] = torch.empty(N, stride, dtype=torch.float32, device=self.device)
            t["sample_cnt"] = torch.empty(N, dtype=torch.int32, device=self.device)
            ws.sample_t, ws.sample_cnt, ws.sample_stride = t["sample_t"].data_ptr(), t["sample_cnt"].data_ptr(), stride

        def fold(cf, stream_ptr):
            cf = self._dev_f32(cf.reshape(-1), "cond_feat")
            if cf.numel() != self.head.cond_dim:
                raise GfppError(f"cond_feat must have {self.head.cond_dim} values, got {cf.numel()}")
            call("gfpp_head_frame_fold", ctypes.byref(self.head), ctypes.byref(ws), cf.data_ptr(), ind_ptr, stream_ptr)
            return cf

        ws.frame_consts = t["frame_consts"].data_ptr()
        folded = isinstance(cond_feat, FoldedConsts)
        if folded:
            if cond_feat.consts.numel() != 256 or cond_feat.consts.dtype != torch.float32 or not cond_feat.consts.is_contiguous():
                raise GfppError("FoldedConsts: 256 contiguous float32 values")
            ws.frame_consts = cond_feat.consts.data_ptr()
        side = None
        if callable(cond_feat):
            side = self._side_stream.get(self.lane)
            if side is None:
                side = self._side_stream[self.lane] = shared_stream(self.device, "side", self.lane)
            side.wait_stream(main)                      # fork: everything the caller queued so far (input copies) is visible
            with torch.cuda.stream(side):
                t["cond_feat"] = fold(cond_feat(), side.cuda_stream)
        if premarched and self.fuse_begin:
            # slab test + state reset + pre-march in one launch (the rays are read once)
            call("gfpp_head_frame_begin_premarch", ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), float(dt_gamma),
                 int(max_steps), st)
            if side is None and not folded:
                fold(cond_feat, st)
        else:
            call("gfpp_head_frame_begin", ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), None, None, st)
            if side is None and not folded:
                fold(cond_feat, st)
            if premarched:
                call("gfpp_head_frame_premarch", ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), float(dt_gamma),
                     int(max_steps), st)
        if side is not None:
            main.wait_stream(side)                      # join
        trips = "gfpp_head_frame_trips_lp" if lp else ("gfpp_head_frame_trips" if premarched else "gfpp_head_frame_march")
        persist = premarched and self.lp_kernel == "persist" and int(max_steps) <= 24 and N <= (1 << 22)
        ws.defer_resolve, ws.resolve_max_steps = 0, 0
        t["deferred"] = 0
        if persist:
            trips = "gfpp_head_frame_persist_lp" if lp else "gfpp_head_frame_persist"
            if "snapshots" not in t:
                t["snapshots"] = torch.empty(N, 7, 5, dtype=torch.float32, device=self.device)
            ws.snapshots = t["snapshots"].data_ptr()
            if defer_resolve and shard is None and self.fuse_tail in ("1", "resolve"):
                # the consumer of the ray records (the 16-bit torso kernel) picks budget and snapshot per ray itself: one launch less on the frame's critical path
                ws.defer_resolve, ws.resolve_max_steps = 1, int(max_steps)
                t["deferred"] = int(max_steps)
        if shard is None:
            call(trips, ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), float(dt_gamma), int(max_steps), float(T_thresh), st)
            return ws, t
        import torch.distributed as dist
        group, n_frame = shard
        if not premarched:
            raise GfppError("ray-tile sharding runs on the pre-marched trip kernels (fp32_kernel='wave' or a 16-bit precision)")
        g = t.get("gcounters")
        if g is None:
            g = t["gcounters"] = torch.zeros(64, dtype=torch.int32, device=self.device)
        if persist:
            # workgroup-local trips: the tiles need nothing from each other while they render; what the frame-wide loop control needs -- the step
            # budget and the alive counts -- is a function of the histogram of the rays' end points, summed over the tiles with ONE all_reduce
            ws.gcounters, ws.N_global = g.data_ptr(), int(n_frame)
            try:
                call(trips, ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), float(dt_gamma), int(max_steps), float(T_thresh), st)
                g.copy_(t["counters"][128:192])
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                call("gfpp_head_frame_resolve", ctypes.byref(ws), int(max_steps), st)
            finally:
                ws.gcounters, ws.N_global = None, 0
            return ws, t
        g.zero_()
        g[:1].fill_(int(n_frame))
        ws.gcounters, ws.N_global = g.data_ptr(), int(n_frame)
        try:
            for trip in range(int(max_steps)):
                ws.trip_first, ws.trip_count = trip, 1
                call(trips, ctypes.byref(self.head), ctypes.byref(ws), rays_o.data_ptr(), rays_d.data_ptr(), float(dt_gamma), int(max_steps), float(T_thresh), st)
                if trip + 1 < int(max_steps):
                    nxt = t["counters"][trip + 1:trip + 2].clone()
                    dist.all_reduce(nxt, op=dist.ReduceOp.SUM, group=group)      # frame-wide n_alive of the next trip
                    g[trip + 1:trip + 2].copy_(nxt)
        finally:
            ws.gcounters, ws.N_global, ws.trip_first, ws.trip_count = None, 0, 0, 0
        return ws, t

    def eval_samples(self, position, direction, cond_feat, ind_code):
        """RADNeRF.forward on a sample list through the trip kernels' own evaluate_block (gfpp_head_eval_samples[_lp], current precision)
        -> sigma [M], color [M,3], ambient [M, amb_D]."""
        position = self._dev_f32(position, "position").reshape(-1, 3)
        direction = self._dev_f32(direction, "direction").reshape(-1, 3)
        M = position.shape[0]
        ws, _ = self.workspace(1)
        cf = self._dev_f32(cond_feat.reshape(-1), "cond_feat")
        if cf.numel() != self.head.cond_dim:
            raise GfppError(f"cond_feat must have {self.head.cond_dim} values, got {cf.numel()}")
        ind = self._dev_f32(ind_code.reshape(-1), "ind_code") if ind_code is not None else None
        st = torch.cuda.current_stream().cuda_stream
        call("gfpp_head_frame_fold", ctypes.byref(self.head), ctypes.byref(ws), cf.data_ptr(), ind.data_ptr() if ind is not None else None, st)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        sigma, color, ambient = f(M), f(M, 3), f(M, int(self.head.amb_grid.D))
        call("gfpp_head_eval_samples" if self.precision == "fp32" else "gfpp_head_eval_samples_lp", ctypes.byref(self.head), ctypes.byref(ws),
             position.data_ptr(), direction.data_ptr(), M, sigma.data_ptr(), color.data_ptr(), ambient.data_ptr(), st)
        return sigma, color, ambient

    def render_head(self, rays_o, rays_d, cond_feat, ind_code, dt_gamma, max_steps, T_thresh, bg_color, shard=None):
        ws, t = self.head_pass(rays_o, rays_d, cond_feat, ind_code, dt_gamma, max_steps, T_thresh, shard=shard)
        st = torch.cuda.current_stream().cuda_stream
        bg_ptr, bg_scalar, _bg_keep = self._bg(bg_color, ws.N)
        out_image, out_depth = torch.empty_like(t["out_image"]), torch.empty_like(t["out_depth"])
        call("gfpp_head_frame_finish", ctypes.byref(ws), bg_ptr, bg_scalar, out_image.data_ptr(), out_depth.data_ptr(), st)
        return {"image": out_image, "depth": out_depth}

    def _bg(self, bg_color, N):
        if bg_color is None:
            return None, 1.0, None
        if torch.is_tensor(bg_color):
            bg = self._dev_f32(bg_color, "bg_color").reshape(-1, 3)
            if bg.shape[0] == 1:
                bg = bg.expand(N, 3).contiguous()
            return bg.data_ptr(), 1.0, bg
        return None, float(bg_color), None

    def render_head_torso(self, rays_o, rays_d, cond_feat, ind_code, bg_coords, poses, torso_code, lm68, dt_gamma, max_steps, T_thresh,
                          bg_color, use_head_for_torso, shard=None):
        """Head pass + torso pass + compositing -> dict(image [N,3], depth [N], torso_alpha [N,1], torso_bg [N,3],
        deform_dense [N,2], torso_mask [N] u8, deform=None)."""
        if self.torso is None:
            raise GfppError("this model has no torso networks")
        ws, t = self.head_pass(rays_o, rays_d, cond_feat, ind_code, dt_gamma, max_steps, T_thresh, shard=shard, defer_resolve=self.precision != "fp32")
        N = ws.N
        dev = self.device
        bg_coords = self._dev_f32(bg_coords, "bg_coords").reshape(-1, 2)
        if self.torso.variant == 1:
            if lm68 is None:
                raise GfppError("RADNeRFTorsowithSR.render needs lm68")
            cond_in = self._dev_f32(lm68.reshape(-1), "lm68")
            if cond_in.numel() != 136:
                raise GfppError("lm68 must hold 68 x 2 values")
        else:
            cond_in = self._dev_f32(poses.reshape(-1), "poses")
            if cond_in.numel() != 6:
                raise GfppError("poses must hold 6 values")
        code = self._dev_f32(torso_code.reshape(-1), "torso_code") if torso_code is not None else None
        bg_ptr, bg_scalar, _bg_keep = self._bg(bg_color, N)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {"image": f(N, 3), "depth": f(N), "torso_alpha": f(N, 1), "torso_bg": f(N, 3), "deform_dense": f(N, 2),
               "torso_mask": torch.empty(N, dtype=torch.uint8, device=dev), "deform": None}
        valu = self.precision == "fp32" and (self.fp32_torso != "mfma" or not self.torso.lp_weights or self.torso.lp_dtype != 0)
        ws.clip_job, ws.clip_lane = None, 0
        if self.clip_job is not None and not valu and self.fuse_tail in ("1", "store"):
            ws.clip_job, ws.clip_lane = int(self.clip_job[0]), int(self.clip_job[1])
            self.clip_job_consumed = True
        call("gfpp_torso_frame" if valu else "gfpp_torso_frame_lp", ctypes.byref(self.torso), ctypes.byref(ws),
             bg_coords.data_ptr(), cond_in.data_ptr(),
             code.data_ptr() if code is not None else None, bg_ptr, bg_scalar, int(bool(use_head_for_torso)), out["image"].data_ptr(),
             out["depth"].data_ptr(), out["torso_alpha"].data_ptr(), out["torso_bg"].data_ptr(), out["deform_dense"].data_ptr(),
             out["torso_mask"].data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    # -- frame groups: K consecutive frames of a clip through ONE persistent head launch (gfpp_frame_ws.n_frames) ------------------------------------
    GROUP_MAX = 4           # kPMaxFrames of csrc/frame_head_lp.hip
    #: the K torso passes of a group (+ resolve + the clip job's uint8 stores) as ONE launch of persistent workgroups (gfpp_torso_group_lp, round 5);
    #: GFPP_GROUP_TORSO=0: one gfpp_torso_frame_lp (+ store) per frame behind a resolve launch, the A/B partner (same bits)
    group_torso = tuning.HOST["group_torso"]

    def group_supported(self, N, K, max_steps):
        """Frame groups run on the persistent 16-bit launch (what a clip renders with unless told otherwise); torso models need the MFMA torso kernel's weight images."""
        return (self.precision != "fp32" and self.lp_kernel == "persist" and 2 <= int(K) <= self.GROUP_MAX and int(max_steps) <= 24
                and int(K) * int(N) <= (1 << 22) and (self.torso is None or bool(self.torso.lp_weights)))

    def torso_pixels(self, bg_coords):
        """(mask [N] uint8, ascending int32 indices of the masked pixels): WHERE the torso field is evaluated -- the occupancy grid sampled at the pixel coordinates
        (radnerf_torso.py:166-169), constants of the model and the resolution -- computed once per coordinate tensor (gfpp_torso_mask + one compaction; synchronises:
        the first frame of a resolution, never inside a captured graph because the capture's warm-up frames come first)."""
        # keyed on the CALLER's tensor (address, version, dtype, strides): a half or strided coordinate tensor is converted once and the fp32 copy kept with the
        # entry -- keyed on the converted tensor, every call of such a caller made a fresh copy, missed, synchronised and churned the four entries (round-5 advisory)
        key = (bg_coords.data_ptr(), int(bg_coords.numel()), bg_coords._version, bg_coords.dtype, tuple(bg_coords.stride()))
        hit = self._torso_pixels.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise GfppError("torso_pixels: the masked-pixel list of this resolution must exist before a graph is captured (render one frame group first)")
            caller = bg_coords
            bg_coords = self._dev_f32(bg_coords, "bg_coords").reshape(-1, 2)
            N = int(bg_coords.shape[0])
            mask = torch.empty(N, dtype=torch.uint8, device=self.device)
            call("gfpp_torso_mask", ctypes.byref(self.torso), bg_coords.data_ptr(), N, mask.data_ptr(), torch.cuda.current_stream().cuda_stream)
            idx = torch.nonzero(mask, as_tuple=False).reshape(-1).to(torch.int32).contiguous()
            while len(self._torso_pixels) >= 4:                             # a caller that hands over a fresh coordinate tensor per call must not grow this for ever
                self._torso_pixels.pop(next(iter(self._torso_pixels)))
            hit = self._torso_pixels[key] = (mask, idx, bg_coords, caller)  # (keeps the keyed tensor's address alive; [2] = its contiguous fp32 form)
        return hit[0], hit[1], hit[2]

    def group_workspace(self, N, K, max_steps):
        """The workspaces of K frames of N rays BEHIND EACH OTHER in every per-ray array (what gfpp_head_frame_persist_lp needs to render them with one
        launch) -> (group record, [the K frames' own records], tensors).  `tensors['rays_o' / 'rays_d']` [K, N, 3] are where the caller puts the rays."""
        key = ("group", int(N), int(K), self.lane)
        ent = self._ws.get(key)
        stride = (int(max_steps) + 7 + 7) // 8 * 8
        if ent is not None and ent[2]["sample_t"].shape[1] >= stride:
            return ent
        dev = self.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        t = {"nears": f(K * N), "fars": f(K * N), "ray_state": f(K * N, 8), "counters": torch.zeros(K, 192, dtype=torch.int32, device=dev),
             "sample_t": f(K * N, stride), "sample_cnt": torch.empty(K * N, dtype=torch.int32, device=dev), "snapshots": f(K * N, 7, 5),
             "rays_o": f(K, N, 3), "rays_d": f(K, N, 3), "timeouts": torch.zeros(1, dtype=torch.int32, device=dev)}

        def record(k, n_frames):
            ws = FrameWs()
            ws.N = N
            ws.nears, ws.fars = t["nears"][k * N:].data_ptr(), t["fars"][k * N:].data_ptr()
            ws.ray_state, ws.counters = t["ray_state"][k * N:].data_ptr(), t["counters"][k:].data_ptr()
            ws.sample_t, ws.sample_cnt, ws.sample_stride = t["sample_t"][k * N:].data_ptr(), t["sample_cnt"][k * N:].data_ptr(), stride
            ws.snapshots = t["snapshots"][k * N:].data_ptr()
            ws.alive[0], ws.alive[1] = None, None
            ws.frame_consts, ws.phase_cycles, ws.gcounters = None, None, None
            ws.separate_trips = ws.N_global = ws.trip_first = ws.trip_count = ws.full_grid_trips = 0
            ws.defer_resolve, ws.resolve_max_steps, ws.clip_job, ws.clip_lane, ws.clip_sub, ws.clip_advance = 0, 0, None, 0, 0, 0
            ws.n_frames, ws.frame_consts_stride, ws.timeouts = n_frames, 0, t["timeouts"].data_ptr()
            ws.row_rays = 0
            return ws
        ent = (record(0, K), [record(k, 0) for k in range(K)], t)
        self._ws[key] = ent
        return ent

    def _group_head_launch(self, consts, N, dt_gamma, max_steps, T_thresh, poses, camera, who):
        """The two halves of a frame group's head pass as closures: begin() = per frame slab test + pre-march (+ the rays, when `poses` / `camera` are given: ONE
        prologue launch for the K frames), launch() = the ONE persistent head launch over the rays of all K frames.  (Two halves: the torso's per-frame constant
        folds are issued between them.)"""
        K = len(consts)
        gws, frames, t = self.group_workspace(N, K, max_steps)
        st = torch.cuda.current_stream().cuda_stream
        c = [x.consts if isinstance(x, FoldedConsts) else x for x in consts]
        for x in c:
            if x.numel() != 256 or x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
                raise GfppError(f"{who}: every frame needs 256 contiguous float32 folded constants on the GPU")
        step = (c[1].data_ptr() - c[0].data_ptr()) // 4
        if step < 256 or any(c[k].data_ptr() - c[0].data_ptr() != 4 * step * k for k in range(K)):
            raise GfppError(f"{who}: the frames' constants must be equally spaced views (frame_consts_stride)")
        pstep = 0
        if poses is not None:
            pstep = (poses[1].data_ptr() - poses[0].data_ptr()) // 4
            if pstep < 16 or any(poses[k].data_ptr() - poses[0].data_ptr() != 4 * pstep * k or poses[k].numel() != 16 for k in range(K)):
                raise GfppError(f"{who}: the frames' poses must be equally spaced [4, 4] views")

        def begin():
            if poses is not None:
                fx, fy, cx, cy, H, W = camera
                gws.row_rays = int(W)                  # the rays are generated in pixel order: the head launch may give every XCD its own image columns
                call("gfpp_head_group_begin", ctypes.byref(self.head), ctypes.byref(gws), poses[0].data_ptr(), int(pstep), float(fx), float(fy), float(cx), float(cy), int(H),
                     int(W), t["rays_o"].data_ptr(), t["rays_d"].data_ptr(), float(dt_gamma), int(max_steps), st)
            else:
                gws.row_rays = 0
                for k in range(K):
                    call("gfpp_head_frame_begin_premarch", ctypes.byref(self.head), ctypes.byref(frames[k]), t["rays_o"][k].data_ptr(), t["rays_d"][k].data_ptr(),
                         float(dt_gamma), int(max_steps), st)

        def launch():
            gws.frame_consts, gws.frame_consts_stride = c[0].data_ptr(), step
            call("gfpp_head_frame_persist_lp", ctypes.byref(self.head), ctypes.byref(gws), t["rays_o"].data_ptr(), t["rays_d"].data_ptr(), float(dt_gamma), int(max_steps),
                 float(T_thresh), st)
        return begin, launch

    def render_group_head(self, consts, N, dt_gamma, max_steps, T_thresh, bg_color, after_frame=None, poses=None, camera=None):
        """Head-only models (RADNeRF / RADNeRFwithSR): K = len(consts) frames of N rays through ONE persistent head launch, then one resolve launch for the group and
        per frame the head-only epilogue (renderer.py:385-397) (+ `after_frame(k, out)`: the SR stage, the uint8 store).  Rays: from `poses` + `camera` inside the
        prologue launch, or as the caller has put them into group_workspace()['rays_o' / 'rays_d'].  Every frame is the bits of its own render_head.  Returns the K
        result dicts {'image' [N, 3], 'depth' [N]}."""
        K = len(consts)
        if self.torso is not None or not self.group_supported(N, K, max_steps):
            raise GfppError("render_group_head: needs a head-only model, a 16-bit precision, lp_kernel='persist', 2 <= K <= 4, max_steps <= 24")
        gws, frames, t = self.group_workspace(N, K, max_steps)
        st = torch.cuda.current_stream().cuda_stream
        begin, launch = self._group_head_launch(consts, N, dt_gamma, max_steps, T_thresh, poses, camera, "render_group_head")
        begin()
        launch()
        call("gfpp_head_group_resolve", ctypes.byref(gws), int(max_steps), st)       # every frame against its own histogram, one launch
        bg_ptr, bg_scalar, _bg_keep = self._bg(bg_color, N)
        outs = []
        for k in range(K):
            ws = frames[k]
            ws.defer_resolve, ws.resolve_max_steps = 0, 0
            ws.clip_job, ws.clip_lane, ws.clip_sub, ws.clip_advance = None, 0, 0, 0
            out = {"image": torch.empty(N, 3, dtype=torch.float32, device=self.device), "depth": torch.empty(N, dtype=torch.float32, device=self.device)}
            call("gfpp_head_frame_finish", ctypes.byref(ws), bg_ptr, bg_scalar, out["image"].data_ptr(), out["depth"].data_ptr(), st)
            if after_frame is not None:
                after_frame(k, out)
            outs.append(out)
        return outs

    def render_group_head_torso(self, consts, ind_code, bg_coords, torso_inputs, torso_code, dt_gamma, max_steps, T_thresh, bg_color, use_head_for_torso,
                                after_frame=None, poses=None, camera=None):
        """K frames (K = len(consts)) whose rays the caller has put into group_workspace()['rays_o' / 'rays_d']: per frame slab test + pre-march, then ONE
        persistent head launch over the rays of all K frames, then per frame resolve + torso pass (+ `after_frame(k, out)`: the SR stage, the uint8 store).
        consts[k]: the frame's 256 folded constants (FoldedConsts of equally spaced views, e.g. of the clip's rows); torso_inputs[k]: lm68 [136] or poses [6].
        poses + camera = (fx, fy, cx, cy, H, W): the rays are GENERATED here from the frames' cam2world matrices (equally spaced [4, 4] views) in the same launch
        as the slab test and the pre-march (gfpp_head_group_begin) instead of being read from the workspace.
        Every frame is the bits of its own render_head_torso.  Returns the K result dicts."""
        K = len(consts)
        N = int(bg_coords.reshape(-1, 2).shape[0])
        if self.torso is None or not self.group_supported(N, K, max_steps):
            raise GfppError("render_group_head_torso: needs a 16-bit precision, lp_kernel='persist', a torso model, 2 <= K <= 4, max_steps <= 24")
        gws, frames, t = self.group_workspace(N, K, max_steps)
        st = torch.cuda.current_stream().cuda_stream
        dev = self.device
        begin, launch = self._group_head_launch(consts, N, dt_gamma, max_steps, T_thresh, poses, camera, "render_group_head_torso")
        begin()
        code = self._dev_f32(torso_code.reshape(-1), "torso_code") if torso_code is not None else None
        folded = None
        if self.group_torso and self.torso.lp_dtype in (GFPP_F16, GFPP_BF16):
            # the torso MLPs' per-frame constant columns folded into biases for all K frames, AHEAD of the head launch (they do not depend on it)
            want = 136 if self.torso.variant == 1 else 6
            ins = [self._dev_f32(x.reshape(-1), "lm68 / poses") for x in torso_inputs]
            if any(x.numel() != want for x in ins):
                raise GfppError("render_group_head_torso: torso_inputs must hold lm68 [136] (landmark-conditioned torso) or poses [6]")
            tstep = (ins[1].data_ptr() - ins[0].data_ptr()) // 4
            if tstep < want or any(ins[k].data_ptr() - ins[0].data_ptr() != 4 * tstep * k for k in range(K)):
                ins = [torch.stack(ins)]                  # not equally spaced views (a caller outside the clip renderer): one small copy
                tstep = want
            folded = torch.empty(K, 96, dtype=torch.float32, device=dev)
            call("gfpp_torso_fold_batch", ctypes.byref(self.torso), ins[0].data_ptr(), int(tstep), code.data_ptr() if code is not None else None, K, folded.data_ptr(), st)
        launch()
        bg_caller = bg_coords
        if folded is None:
            bg_coords = self._dev_f32(bg_coords, "bg_coords").reshape(-1, 2)
        bg_ptr, bg_scalar, _bg_keep = self._bg(bg_color, N)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        if folded is not None:
            # ONE launch for the K torso passes: resolve (budget + snapshot per ray), torso field, compositing and -- inside a clip job -- the uint8 stores and the
            # cursor's advance (gfpp_torso_group_lp; every value the bits of the per-frame kernel)
            gws.clip_job, gws.clip_lane, gws.clip_sub, gws.clip_advance = None, 0, 0, 0
            if self.clip_job is not None:
                gws.clip_job, gws.clip_lane, gws.clip_advance = int(self.clip_job[0]), int(self.clip_job[1]), K * int(self.clip_job[2])
                self.clip_job_consumed = True
            stack = {"image": f(K * N, 3), "depth": f(K * N), "torso_alpha": f(K * N, 1), "torso_bg": f(K * N, 3), "deform_dense": f(K * N, 2),
                     "torso_mask": torch.empty(K * N, dtype=torch.uint8, device=dev)}
            try:
                mask_static, masked_idx, bg_coords = self.torso_pixels(bg_caller)
                call("gfpp_torso_group_lp", ctypes.byref(self.torso), ctypes.byref(gws), bg_coords.data_ptr(), folded.data_ptr(), code.data_ptr() if code is not None else None,
                     mask_static.data_ptr(), masked_idx.data_ptr() if masked_idx.numel() else None, int(masked_idx.numel()),
                     bg_ptr, bg_scalar, int(bool(use_head_for_torso)), int(max_steps), stack["image"].data_ptr(), stack["depth"].data_ptr(), stack["torso_alpha"].data_ptr(),
                     stack["torso_bg"].data_ptr(), stack["deform_dense"].data_ptr(), stack["torso_mask"].data_ptr(), st)
            finally:
                gws.clip_job, gws.clip_lane, gws.clip_advance = None, 0, 0
            outs = []
            for k in range(K):
                out = {name: v[k * N:(k + 1) * N] for name, v in stack.items()}
                out["deform"] = None
                if after_frame is not None:
                    after_frame(k, out)
                outs.append(out)
            return outs
        outs = []
        defer = self.fuse_tail in ("1", "resolve")          # the torso kernel picks budget and snapshot per ray itself (one launch less per frame)
        if not defer:
            call("gfpp_head_group_resolve", ctypes.byref(gws), int(max_steps), st)       # every frame against its own histogram, one launch
        store = self.clip_job is not None and self.fuse_tail in ("1", "store")      # ... and writes the uint8 frame into the clip job's slot
        for k in range(K):
            ws = frames[k]
            ws.defer_resolve, ws.resolve_max_steps = (1, int(max_steps)) if defer else (0, 0)
            ws.clip_job, ws.clip_lane, ws.clip_sub, ws.clip_advance = None, 0, 0, 0
            if store:
                ws.clip_job, ws.clip_lane, ws.clip_sub = int(self.clip_job[0]), int(self.clip_job[1]), k
                ws.clip_advance = K * int(self.clip_job[2]) if k == K - 1 else 0xFFFFFFFF
                self.clip_job_consumed = True
            cond_in = self._dev_f32(torso_inputs[k].reshape(-1), "lm68 / poses")
            if cond_in.numel() != (136 if self.torso.variant == 1 else 6):
                raise GfppError("render_group_head_torso: torso_inputs must hold lm68 [136] (landmark-conditioned torso) or poses [6]")
            out = {"image": f(N, 3), "depth": f(N), "torso_alpha": f(N, 1), "torso_bg": f(N, 3), "deform_dense": f(N, 2),
                   "torso_mask": torch.empty(N, dtype=torch.uint8, device=dev), "deform": None}
            call("gfpp_torso_frame_lp", ctypes.byref(self.torso), ctypes.byref(ws), bg_coords.data_ptr(), cond_in.data_ptr(),
                 code.data_ptr() if code is not None else None, bg_ptr, bg_scalar, int(bool(use_head_for_torso)), out["image"].data_ptr(),
                 out["depth"].data_ptr(), out["torso_alpha"].data_ptr(), out["torso_bg"].data_ptr(), out["deform_dense"].data_ptr(),
                 out["torso_mask"].data_ptr(), st)
            if after_frame is not None:
                after_frame(k, out)
            outs.append(out)
        return outs

    def graphed(self, key, fn, inputs):
        """Run `fn(**inputs)` through a per-`key` captured graph (captured on first use, re-captured if shapes change)."""
        key = (key, self.precision)
        g = self._graphs.get(key)
        if g is None or not g.matches(inputs):
            g = GraphedFrame(fn, inputs)
            self._graphs[key] = g
        return g(inputs)

    MAX_TRIPS = 63

    def enable_phase_cycles(self, N, on=True):
        """Profiling aid of the 16-bit kernel: per-trip shader cycles by phase (see gfpp_frame_ws.phase_cycles)."""
        ws, t = self.workspace(N)
        if on:
            t["phase_cycles"] = torch.zeros(64, 8, dtype=torch.int64, device=self.device)
            ws.phase_cycles = t["phase_cycles"].data_ptr()
        else:
            ws.phase_cycles = None
            ws.sample_t, ws.sample_cnt, ws.sample_stride = None, None, 0
            t.pop("phase_cycles", None)
        return t.get("phase_cycles")

    def check_barriers(self):
        """Raise if a device-wide barrier of a multi-trip launch (the trip-launch path of the 16-bit modes) timed out in ANY frame rendered since the last
        check: besides poisoning counters[127] (which the next frame's begin kernel zeroes again) the kernel adds one to the workspace's sticky word
        (gfpp_frame_ws.timeouts), which nothing resets but this function -- so a frame in the middle of a clip that lacks trips is not delivered
        silently.  One 4-byte read per workspace; synchronises."""
        bad = []
        for key, ent in self._ws.items():
            if key[0] == "group":                  # frame groups run on the persistent launch only: no barrier
                continue
            (n, lane), t = key, ent[1]
            count = int(t["timeouts"].item())
            if count > 0:
                bad.append((n, lane, count))
                t["timeouts"].zero_()
        if bad:
            raise GfppError(f"a device-wide barrier of the multi-trip launch timed out (workspaces (rays, lane, launches) = {bad}): frames rendered there "
                            f"since the last check are incomplete.  Other work held compute units for too long; render with lp_kernel='persist' (no "
                            f"barrier) or separate_trips >= max_steps")

    def budget(self, N):
        """The persistent 16-bit launch's own record of the last frame: histogram of the rays' end points [32], evaluated samples, workgroup rounds
        (sum, max); synchronises."""
        c = self.workspace(N)[1]["counters"].cpu().numpy()
        return {"hist": c[128:160].copy(), "samples": int(c[168]), "rounds_sum": int(c[169]), "rounds_max": int(c[170]), "samples_max_wg": int(c[171]),
                # thread 0's shader clock per workgroup, summed over the workgroups, in units of 1024 cycles: fetch (+ ingest) | compaction | evaluate |
                # composite + list upkeep; and the longest workgroup
                "kcycles": {"fetch": int(c[172]), "compact": int(c[173]), "evaluate": int(c[174]), "composite": int(c[175]), "longest_wg": int(c[176]),
                            "ingest_of_fetch": int(c[177])}}

    def trip_counters(self, N):
        """(alive rays at the start of each trip, samples evaluated by each trip) of the last frame; synchronises.  After the persistent 16-bit
        launch the alive counts are the ones gfpp_head_frame_resolve reconstructed (what the trip launches would have left), and the samples of the
        whole frame are reported under trip 0 (a workgroup's local trips are not the reference's)."""
        ws, t = self.workspace(N)
        if t.get("deferred"):
            # the frame's consumer resolved budget and snapshots on the fly; the reconstructed alive counts are written by the resolve entry (idempotent)
            call("gfpp_head_frame_resolve", ctypes.byref(ws), int(t["deferred"]), torch.cuda.current_stream().cuda_stream)
        c = t["counters"].cpu().numpy()
        return c[:64], c[64:127]
