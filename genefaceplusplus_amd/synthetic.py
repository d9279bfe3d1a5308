"""Synthetic "May"-shaped model, scene and per-frame driving inputs.

No May checkpoint / dataset ships with the reference mount (README lists them as downloads), so parity tests and
bench.py use random-init weights of the *exact* May architecture and checkpoint layout, generated here with a
fixed numpy RNG.  This module only creates *inputs* (numpy arrays keyed by the reference's ``state_dict`` names);
it contains no renderer code.

The same arrays are loaded (strict=True) into the reference's own nn.Modules by tests/golden/make_golden.py, into
the product modules of this package, and handed to the CPU oracle -- which is what makes the three comparable.
"""
import math

import numpy as np

from .configs import may_hparams

f32 = np.float32


# --------------------------------------------------------------------------------------------- shapes
def grid_offsets(input_dim, num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048,
                 align_corners=False):
    """Level table sizes as GridEncoder.__init__ computes them (encoders/gridencoder/grid.py:97-137)."""
    per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def _expand_bits(v):
    v = np.asarray(v, dtype=np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z):
    """x -> bit 0, y -> bit 1, z -> bit 2 (raymarching.cu:56-71)."""
    return _expand_bits(x) | (_expand_bits(y) << np.uint32(1)) | (_expand_bits(z) << np.uint32(2))


def pack_bitfield(density_grid, thresh):
    """bit i of byte n <-> density_grid.flat[8n+i] > thresh (raymarching.cu:267-289)."""
    occ = (np.asarray(density_grid, f32).reshape(-1, 8) > f32(thresh)).astype(np.uint8)
    return (occ << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)


# --------------------------------------------------------------------------------------------- weights
def _linear(rng, out_f, in_f, bias):
    b = 1.0 / math.sqrt(in_f)
    w = rng.uniform(-b, b, (out_f, in_f)).astype(f32)
    return (w, rng.uniform(-b, b, (out_f,)).astype(f32)) if bias else (w, None)


def _conv(rng, out_c, in_c, k):
    b = 1.0 / math.sqrt(in_c * k)
    return rng.uniform(-b, b, (out_c, in_c, k)).astype(f32), rng.uniform(-b, b, (out_c,)).astype(f32)


#: per-layer gains applied on top of the nn.Linear default init so that the random-init model has a non-degenerate
#: output range (default init shrinks the signal ~3x per bias-free layer: colours would all be 0.5 +- 0.001).
DEFAULT_GAINS = {"ambient_net": 3.0, "sigma_net": 2.0, "color_net": 8.0, "torso_deform_net": 1.2,
                 "torso_canonicial_net": 4.0}


def _table(rng, offsets, scale, decay, base_resolution=16):
    """Grid table ~ U(-a_l, a_l) per level with a_l = scale * (rows_0 / rows_l) ** (decay / D-ish): coarse levels carry
    the large amplitudes, fine levels small ones -- like a trained multiresolution grid, and it keeps the (random) field's
    Lipschitz constant level-independent so that fp32 round-off in the ambient coordinates is not amplified 2048x."""
    n = int(offsets[-1])
    tab = rng.uniform(-1.0, 1.0, (n, 2))
    L = len(offsets) - 1
    for l in range(L):
        res = base_resolution * (2048.0 / base_resolution) ** (l / (L - 1))
        tab[offsets[l]:offsets[l + 1]] *= scale * (base_resolution / res) ** decay
    return tab.astype(f32)


def _mlp(sd, rng, prefix, dim_in, dim_out, dim_hidden, num_layers, gain=1.0):
    for l in range(num_layers):
        i = dim_in if l == 0 else dim_hidden
        o = dim_out if l == num_layers - 1 else dim_hidden
        sd[f"{prefix}.net.{l}.weight"] = (_linear(rng, o, i, False)[0] * f32(gain)).astype(f32)


def cond_input_dim(hp):
    """radnerf.py:17-32: 44 (esperanto) / 29 (deepspeech) audio features, or 3-D landmarks (68 / 131 / 468 points)."""
    ct = hp.get("cond_type", "idexp_lm3d_normalized")
    if ct in ("esperanto", "deepspeech"):
        return {"esperanto": 44, "deepspeech": 29}[ct]
    return {"lm68": 68 * 3, "lm131": 131 * 3, "lm468": 468 * 3}[hp.get("nerf_keypoint_mode", "lm68")]


def synthetic_state_dict(hp=None, variant="may_torso", seed=9999, table_scale=1.0, sigma_gain=6.0,
                         ellipsoid=(0.30, 0.22, 0.35), gains=None, table_decay=1.0):
    """Random-init parameters + buffers with the reference's key names / shapes / dtypes.

    variant: 'may_head' (RADNeRF), 'may_torso' (RADNeRFTorso), 'may_head_sr', 'may_torso_sr' (without sr_net.*).
    table_scale, table_decay: grid tables ~ U(-a_l, a_l), a_l = table_scale * (16 / res_l) ** table_decay  (the reference's
                init is U(-1e-4, 1e-4) at every level, grid.py:141-143, which gives degenerate all-zero features).
    sigma_gain: row 0 of sigma_net.net.2.weight is made non-negative and scaled so sigma*dt spans a useful range
                 -- with bias-free layers a plain random row gives alpha ~ 0.027 everywhere.
    """
    hp = may_hparams(variant) if hp is None else hp
    g = dict(DEFAULT_GAINS)
    g.update(gains or {})
    rng = np.random.default_rng(seed)
    sd = {}
    bound = float(hp["bound"])
    H = int(hp["grid_size"])
    cascade = 1 + math.ceil(math.log2(hp["bound"]))
    sd["aabb_train"] = np.array([-bound, -bound / 2, -bound, bound, bound / 2, bound], f32)
    sd["aabb_infer"] = sd["aabb_train"].copy()
    if hp["individual_embedding_dim"] > 0:
        sd["individual_embeddings"] = (rng.standard_normal((hp["individual_embedding_num"], hp["individual_embedding_dim"])) * 0.1).astype(f32)

    # occupancy: an ellipsoid at the origin, stored in Morton order per cascade level
    density_grid = np.zeros((cascade, H ** 3), f32)
    idx = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    for c in range(cascade):
        half = min(2.0 ** c, bound)
        cx = ((X.astype(f32) + 0.5) / H * 2 - 1) * half
        cy = ((Y.astype(f32) + 0.5) / H * 2 - 1) * half
        cz = ((Z.astype(f32) + 0.5) / H * 2 - 1) * half
        inside = (cx / ellipsoid[0]) ** 2 + (cy / ellipsoid[1]) ** 2 + (cz / ellipsoid[2]) ** 2 <= 1.0
        density_grid[c, morton3d(X, Y, Z).reshape(-1)] = np.where(inside, f32(20.0), f32(0.0)).reshape(-1)
    sd["density_grid"] = density_grid
    sd["density_bitfield"] = pack_bitfield(density_grid, hp["density_thresh"])
    sd["step_counter"] = np.zeros((16, 2), np.int32)

    # conditioning nets (cond_encoder.py:98-180)
    cond_in = cond_input_dim(hp)
    cond_out = hp["cond_out_dim"] // 2 * 2
    for i, (o, c) in zip((0, 2, 4, 6), ((32, cond_in), (32, 32), (64, 32), (64, 64))):
        sd[f"cond_prenet.encoder_conv.{i}.weight"], sd[f"cond_prenet.encoder_conv.{i}.bias"] = _conv(rng, o, c, 3)
    sd["cond_prenet.encoder_fc1.0.weight"], sd["cond_prenet.encoder_fc1.0.bias"] = _linear(rng, 64, 64, True)
    sd["cond_prenet.encoder_fc1.2.weight"], sd["cond_prenet.encoder_fc1.2.bias"] = _linear(rng, cond_out, 64, True)
    if hp.get("add_eye_blink_cond", False):
        sd["blink_embedding.weight"] = rng.standard_normal((1, cond_out // 2)).astype(f32)
        sd["blink_encoder.0.weight"], sd["blink_encoder.0.bias"] = _linear(rng, cond_out // 2, cond_out // 2, True)
        sd["blink_encoder.1.weight"], sd["blink_encoder.1.bias"] = _linear(rng, hp["eye_blink_dim"], cond_out // 2, True)
    if hp["with_att"]:
        for i, (o, c) in zip((0, 2, 4, 6, 8), ((16, cond_out), (8, 16), (4, 8), (2, 4), (1, 2))):
            sd[f"cond_att_net.attentionConvNet.{i}.weight"], sd[f"cond_att_net.attentionConvNet.{i}.bias"] = _conv(rng, o, c, 3)
        smo = hp["smo_win_size"]
        sd["cond_att_net.attentionNet.0.weight"], sd["cond_att_net.attentionNet.0.bias"] = _linear(rng, smo, smo, True)

    # head NeRF (radnerf.py:55-86)
    off3, _ = grid_offsets(3, log2_hashmap_size=hp["log2_hashmap_size"], desired_resolution=hp["desired_resolution"] * hp["bound"])
    sd["position_embedder.offsets"] = off3
    sd["position_embedder.embeddings"] = _table(rng, off3, table_scale, table_decay)
    _mlp(sd, rng, "ambient_net", 32 + cond_out, hp["ambient_coord_dim"], hp["hidden_dim_ambient"], hp["num_layers_ambient"], g["ambient_net"])
    offa, _ = grid_offsets(hp["ambient_coord_dim"], log2_hashmap_size=hp["log2_hashmap_size"], desired_resolution=hp["desired_resolution"])
    sd["ambient_embedder.offsets"] = offa
    sd["ambient_embedder.embeddings"] = _table(rng, offa, table_scale, table_decay)
    _mlp(sd, rng, "sigma_net", 64, 1 + hp["geo_feat_dim"], hp["hidden_dim_sigma"], hp["num_layers_sigma"], g["sigma_net"])
    last = f"sigma_net.net.{hp['num_layers_sigma'] - 1}.weight"
    sd[last][0] = np.abs(sd[last][0]) * f32(sigma_gain)
    _mlp(sd, rng, "color_net", 16 + hp["geo_feat_dim"] + hp["individual_embedding_dim"], 3, hp["hidden_dim_color"], hp["num_layers_color"], g["color_net"])
    if variant == "may_head_sr":
        sd["lambda_ambient"] = np.array([hp.get("lambda_ambient") or 1.0], f32)

    if variant in ("may_torso", "may_torso_sr"):
        g2 = np.zeros((H, H), f32)
        g2[:, int(0.6 * H):] = 1.0            # lower 40 % of image rows (the grid is stored transposed, radnerf_torso.py:225)
        sd["density_grid_torso"] = g2.reshape(-1)
        tdim = hp["torso_individual_embedding_dim"]
        if tdim > 0:
            sd["torso_individual_codes"] = (rng.standard_normal((hp["individual_embedding_num"], tdim)) * 0.1).astype(f32)
        off2, _ = grid_offsets(2, log2_hashmap_size=16, desired_resolution=2048)
        sd["torso_embedder.offsets"] = off2
        sd["torso_embedder.embeddings"] = _table(rng, off2, table_scale, table_decay)
        cond_cols = 14 + 14 * 2 * 4 if variant == "may_torso_sr" else 6 + 6 * 2 * 4
        deform_in = 42 + cond_cols + tdim
        canon_in = 32 + 42 + cond_cols + tdim
        if hp["torso_head_aware"]:
            for i, (o, c) in zip((0, 2, 4), ((16, 4), (32, 16), (16, 32))):
                sd[f"head_color_weights_encoder.{i}.weight"], sd[f"head_color_weights_encoder.{i}.bias"] = _linear(rng, o, c, True)
            deform_in += 16
            canon_in += 16
        _mlp(sd, rng, "torso_deform_net", deform_in, 2, 64, 3, g["torso_deform_net"])
        _mlp(sd, rng, "torso_canonicial_net", canon_in, 4, 32, 3, g["torso_canonicial_net"])
    return sd


# --------------------------------------------------------------------------------------------- camera / driving inputs
def intrinsics_for(H, W):
    """tasks/radnerfs/dataset_utils.py:216-230: focal 1015 at centre 112 (bfm.py:35-36) rescaled to H/2 => 2320 @512."""
    cx, cy = H / 2.0, W / 2.0
    fl = 1015.0 * (cx / 112.0)
    return np.array([fl, fl, cx, cy], f32)


def synthetic_pose(frame_idx=0, max_yaw_deg=5.0, distance=4.0, seed=0):
    """ngp-convention cam2world [4,4]: camera ``distance`` units along +y looking at the origin, yaw jitter about z."""
    rng = np.random.default_rng(seed)
    yaws = rng.uniform(-max_yaw_deg, max_yaw_deg, size=frame_idx + 1)
    th = math.radians(float(yaws[frame_idx]))
    base = np.array([[-1, 0, 0, 0],
                     [0, 0, -1, distance],
                     [0, -1, 0, 0],
                     [0, 0, 0, 1]], dtype=np.float64)
    c, s = math.cos(th), math.sin(th)
    rz = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    return (rz @ base).astype(f32)


def synthetic_frame_inputs(hp, frame_idx=0, seed=0):
    """cond window [smo, cond_win_size, cond_in] ([smo,1,204] for the lm3d configs), lm68 [136], eye_area_percent [1,1] for one frame."""
    rng = np.random.default_rng((seed + 1) * 100003 + frame_idx)
    cond_in = cond_input_dim(hp)
    cond = np.clip(rng.standard_normal((hp["smo_win_size"], hp.get("cond_win_size", 1), cond_in)), -1.5, 1.5).astype(f32)
    lm68 = rng.uniform(0.3, 0.7, (136,)).astype(f32)
    eye = np.array([[0.3]], f32)
    return {"cond": cond, "lm68": lm68, "eye_area_percent": eye}


#: the state-dict family a variant's parameters follow (synthetic_state_dict's own variant switch)
SD_FAMILY = {"audio_head": "may_head", "audio_torso": "may_torso"}


def frame_case(variant, HW, frame_idx=0, hp_over=None, **sd_kw):
    """One synthetic model + one frame of driving inputs as numpy arrays: what the parity tests hand to the CPU oracle and to the product alike."""
    hp = may_hparams(variant)
    hp.update(hp_over or {})
    sd = synthetic_state_dict(hp, SD_FAMILY.get(variant, variant), **sd_kw)
    fi = synthetic_frame_inputs(hp, frame_idx)
    pose = synthetic_pose(frame_idx)[None]
    return {"variant": variant, "hp": hp, "sd": sd, "HW": HW, "pose": pose, "intr": intrinsics_for(HW, HW), **fi,
            "bg_color": np.full((1, HW * HW, 3), 0.5, np.float32), "T_thresh": 0.01}


# ---------------------------------------------------------------------------------------------------------------------
# super-resolution net of the *_sr models (radnerf_sr.py:14-43): same key names / shapes / dtypes as the reference's
# Superresolution(channels=3).state_dict(), deterministic values.  Kept apart from synthetic_state_dict so that the NeRF
# fixtures do not depend on it.
# ---------------------------------------------------------------------------------------------------------------------
SR_LAYERS = (  # name, out, in, kernel, resolution (None = no noise: ToRGB), up
    ("block0.conv0", 128, 3, 3, 256), ("block0.conv1", 128, 128, 3, 256), ("block0.torgb", 3, 128, 1, None),
    ("block1.conv0", 64, 128, 3, 512), ("block1.conv1", 64, 64, 3, 512), ("block1.torgb", 3, 64, 1, None))


def sr_resample_filter():
    """upfirdn2d.setup_filter([1,3,3,1]) (upfirdn2d.py:74-121): outer product, normalised to sum 1."""
    f = np.array([1, 3, 3, 1], f32)
    f2 = np.outer(f, f)
    return (f2 / f2.sum()).astype(f32)


def synthetic_sr_state(seed=4321, prefix="sr_net.", w_dim=16):
    rng = np.random.default_rng(seed)
    sd = {}
    filt = sr_resample_filter()
    sd[prefix + "resample_filter"] = filt.copy()
    for blk in ("block0", "block1"):
        sd[prefix + blk + ".resample_filter"] = filt.copy()
    for name, o, i, k, res in SR_LAYERS:
        p = prefix + name
        sd[p + ".weight"] = rng.standard_normal((o, i, k, k)).astype(f32)
        sd[p + ".bias"] = (0.1 * rng.standard_normal(o)).astype(f32)
        sd[p + ".affine.weight"] = rng.standard_normal((i, w_dim)).astype(f32)
        sd[p + ".affine.bias"] = (1.0 + 0.1 * rng.standard_normal(i)).astype(f32)
        if res is not None:
            sd[p + ".noise_strength"] = np.array(0.05 + 0.05 * rng.random(), f32)
            sd[p + ".resample_filter"] = filt.copy()
            sd[p + ".noise_const"] = rng.standard_normal((res, res)).astype(f32)
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# the on-disk boundary: a checkpoint directory in the reference's layout
# ---------------------------------------------------------------------------------------------------------------------
def write_checkpoint(work_dir, variant="may_torso", hp=None, steps=250000, seed=9999, extra_hparams=None, state_dict=None):
    """Write ``work_dir/model_ckpt_steps_<steps>.ckpt`` + ``work_dir/config.yaml`` the way the reference's trainer leaves them, so that
    the reference's own loader takes them: ``set_hparams(f"{dir}/config.yaml")`` (utils/commons/hparams.py:80-190: the saved file is the
    flat, fully-resolved hparams dict, hparams.py:167-170) and ``load_ckpt(model, dir, model_name='model', strict=True)``
    (utils/commons/ckpt_utils.py:29-76: newest ``model_ckpt_steps_*.ckpt``, ``checkpoint['state_dict']['model']``).

    File format = ``Trainer._atomic_save`` / ``dump_checkpoint`` (utils/commons/trainer.py:542-567): a LEGACY-pickle ``torch.save``
    (``_use_new_zipfile_serialization=False``) of ``{'epoch', 'global_step', 'checkpoint_callback_best', 'optimizer_states',
    'state_dict': {<task child name>: <its state_dict>}}`` written to ``.part`` and renamed.  The weights are the deterministic synthetic
    ones of ``synthetic_state_dict`` (+ ``synthetic_sr_state`` for the *_sr variants) unless ``state_dict`` is given.
    Returns (ckpt_path, config_path)."""
    import os
    import torch
    import yaml
    hp = dict(may_hparams(variant) if hp is None else hp)
    hp.update(extra_hparams or {})
    sd = state_dict
    if sd is None:
        sd = dict(synthetic_state_dict(hp, variant, seed=seed))
        if hp.get("with_sr"):
            sd.update(synthetic_sr_state())
    tensors = {k: (v.detach().cpu().clone() if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items()}
    os.makedirs(work_dir, exist_ok=True)
    ckpt = {"epoch": 0, "global_step": int(steps), "checkpoint_callback_best": None, "optimizer_states": [],
            "state_dict": {"model": tensors}}
    ckpt_path = os.path.join(work_dir, f"model_ckpt_steps_{int(steps)}.ckpt")
    torch.save(ckpt, ckpt_path + ".part", _use_new_zipfile_serialization=False)
    os.replace(ckpt_path + ".part", ckpt_path)
    config_path = os.path.join(work_dir, "config.yaml")
    plain = {k: (list(v) if isinstance(v, tuple) else v) for k, v in hp.items()}
    plain["work_dir"] = work_dir
    with open(config_path, "w") as f:
        yaml.safe_dump(plain, f)
    return ckpt_path, config_path


def read_checkpoint(work_dir_or_file, model_name="model", steps=None):
    """What ``get_last_checkpoint`` + the key handling of ``load_ckpt`` do (utils/commons/ckpt_utils.py:7-50), for callers that do not run the
    reference's loader: newest ``model_ckpt_steps_*.ckpt`` of a directory (or the file itself), ``state_dict[model_name]`` (nested layout) or
    the ``'<model_name>.'``-prefixed keys (flat layout).  Returns (state_dict, path)."""
    import glob
    import os
    import re
    import torch
    if os.path.isfile(work_dir_or_file):
        path = work_dir_or_file
    else:
        pattern = f"{work_dir_or_file}/model_ckpt_steps_{'*' if steps is None else steps}.ckpt"
        paths = sorted(glob.glob(pattern), key=lambda x: -int(re.findall(r".*steps_(\d+)\.ckpt", x)[0]))
        if not paths:
            raise FileNotFoundError(f"| ckpt not found in {work_dir_or_file}.")
        path = paths[0]
    try:
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)      # the reference's files are legacy pickles of plain dicts
    except TypeError:                                                               # pragma: no cover  (torch < 1.13)
        checkpoint = torch.load(path, map_location="cpu")
    sd = checkpoint["state_dict"]
    if any("." in k for k in sd):
        sd = {k[len(model_name) + 1:]: v for k, v in sd.items() if k.startswith(model_name + ".")}
    else:
        sd = sd[model_name]
    return sd, path


# ---------------------------------------------------------------------------------------------------------------------
# compact state files: a state_dict as one .npz whose grid tables / occupancy grids are stored as float16
# (tools/make_trained_checkpoint.py rounds those arrays to float16-representable values BEFORE its final evaluation, so the
# file is lossless for them) -- the form in which the trained procedural field travels as a test fixture (tests/golden/trained/)
# ---------------------------------------------------------------------------------------------------------------------
COMPACT_F16_SUFFIXES = (".embeddings", "density_grid", "density_grid_torso", ".noise_const", "individual_embeddings", "torso_individual_codes")


def compact_f16_key(key):
    return key.endswith(COMPACT_F16_SUFFIXES)


def save_compact_state(path, state_dict):
    """state_dict (tensors or arrays) -> np.savez_compressed; keys of COMPACT_F16_SUFFIXES as float16 (must already be representable)."""
    import torch
    arrs = {}
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        if compact_f16_key(k) and a.dtype == np.float32:
            h = a.astype(np.float16)
            if not np.array_equal(h.astype(f32), a):
                raise ValueError(f"save_compact_state: {k} is not float16-representable (round it before the final evaluation)")
            a = h
        arrs[k] = a
    np.savez_compressed(path, **arrs)


def load_compact_state(path):
    """-> {key: numpy array} with the float16-stored arrays widened back to float32 (exact)."""
    with np.load(path) as z:
        return {k: (z[k].astype(f32) if z[k].dtype == np.float16 else z[k]) for k in z.files}
