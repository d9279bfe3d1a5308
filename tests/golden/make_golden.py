#!/usr/bin/env python
"""Generate tests/golden/* by running the REFERENCE's own Python modules (read-only mount /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

What is executed from the reference, unmodified: utils/commons/hparams.py (yaml chain), modules/radnerfs/
{renderer,radnerf,radnerf_torso,radnerf_torso_sr,cond_encoder,utils}.py, raymarching/raymarching.py and the three encoder
shims (grid.py, sphere_harmonics.py, freq.py).  What is substituted:
  * the four CUDA extension modules (_raymarching_face, _gridencoder, _shencoder, _freqencoder) -> oracle/ref_backends.py,
    i.e. radnerf_oracle.c on CPU tensors (the CUDA sources cannot be built or run here);
  * ``Tensor.cuda()`` -> identity (the shims force-move inputs to CUDA);
  * heavy optional imports of utils.py that the render path never calls (trimesh, mcubes, lpips, tensorboardX, cv2, imageio);
  * modules.radnerfs.radnerf_sr.Superresolution -> identity upsampler (the SR network is outside the first scope; only the
    pre-SR ``rgb_map`` is recorded).
So these fixtures pin everything ABOVE the native-kernel boundary (MLP wiring, loop control, torso pass, ray generation,
conditioning nets, checkpoint key layout, yaml chain); the kernels themselves stay pinned by SURVEY section 8c invariants only.

Outputs: may_hparams.json, state_manifest.json, ref_python_golden.npz
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import ref_backends  # noqa: E402
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.configs import may_hparams, VARIANT_YAML  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def prepare_reference_imports():
    ref_backends.install()
    for name in ("trimesh", "mcubes", "lpips", "tensorboardX", "cv2", "imageio"):
        _stub(name)

    class _IdentitySR(torch.nn.Module):
        def __init__(self, channels=3):
            super().__init__()

        def forward(self, rgb):
            return torch.nn.functional.interpolate(rgb, scale_factor=2, mode="nearest")

    _stub("modules.radnerfs.radnerf_sr", Superresolution=_IdentitySR)
    torch.Tensor.cuda = lambda self, *a, **k: self
    os.chdir(REF)


def to_t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def main():
    prepare_reference_imports()
    from utils.commons.hparams import set_hparams, hparams as global_hparams
    from modules.radnerfs.radnerf import RADNeRF
    from modules.radnerfs.radnerf_torso import RADNeRFTorso
    from modules.radnerfs.radnerf_torso_sr import RADNeRFTorsowithSR
    from modules.radnerfs import utils as ref_utils
    from modules.radnerfs.encoders.gridencoder import GridEncoder

    out = {}
    # ------------------------------------------------------------------ 1. yaml chain
    hp_json = {}
    for variant, yaml_path in VARIANT_YAML.items():
        ref_hp = set_hparams(config=yaml_path, exp_name="", print_hparams=False, global_hparams=False)
        mine = may_hparams(variant)
        for k, v in mine.items():
            assert k in ref_hp and ref_hp[k] == v, (variant, k, v, ref_hp.get(k))
        # keys that change the model must not be missing from our subset
        for k in ("with_sr", "add_eye_blink_cond", "eye_blink_dim", "torso_head_aware", "smo_win_size"):
            assert ref_hp.get(k) == mine.get(k), (variant, k)
        hp_json[variant] = {k: ref_hp[k] for k in mine}
    with open(os.path.join(HERE, "may_hparams.json"), "w") as f:
        json.dump(hp_json, f, indent=1, sort_keys=True)

    # ------------------------------------------------------------------ 2. checkpoint layout + models
    manifest = {}
    models = {}
    for variant, cls in (("may_head", RADNeRF), ("may_torso", RADNeRFTorso), ("may_torso_sr", RADNeRFTorsowithSR)):
        ref_hp = set_hparams(config=VARIANT_YAML[variant], exp_name="", print_hparams=False, global_hparams=True)
        torch.manual_seed(ref_hp["seed"])
        model = cls(ref_hp).eval()
        ref_sd = model.state_dict()
        manifest[variant] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()
                             if not k.startswith("sr_net.")}
        sd = syn.synthetic_state_dict(may_hparams(variant), variant)
        missing = set(manifest[variant]) - set(sd)
        extra = set(sd) - set(manifest[variant])
        assert not missing and not extra, (variant, missing, extra)
        for k, v in sd.items():
            assert list(v.shape) == manifest[variant][k][0], (k, v.shape, manifest[variant][k])
            assert str(v.dtype) == manifest[variant][k][1], (k, v.dtype, manifest[variant][k])
        model.load_state_dict(to_t(sd), strict=True)
        models[variant] = (model, sd, dict(ref_hp))
    with open(os.path.join(HERE, "state_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)

    # ------------------------------------------------------------------ 3. grid encoder table layout
    for D in (2, 3):
        enc = GridEncoder(input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                          desired_resolution=2048, gridtype="tiled")
        out[f"grid_offsets_D{D}"] = enc.offsets.numpy().astype(np.int32)
        out[f"grid_per_level_scale_D{D}"] = np.array([enc.per_level_scale], np.float64)

    # ------------------------------------------------------------------ 4. camera helpers
    pose = syn.synthetic_pose(3)
    H = W = 16
    intr = syn.intrinsics_for(H, W)
    rays = ref_utils.get_rays(torch.from_numpy(pose)[None], intr, H, W, N=-1)
    out["rays_pose"] = pose
    out["rays_o_16"] = rays["rays_o"].numpy().copy()
    out["rays_d_16"] = rays["rays_d"].numpy().copy()
    out["bg_coords_16"] = ref_utils.get_bg_coords(16, 16, "cpu").numpy()
    out["convert_poses"] = ref_utils.convert_poses(torch.from_numpy(pose)[None]).numpy()
    c2w = np.array([[0.9, -0.1, 0.2, 0.01], [0.1, 0.95, -0.05, -0.02], [-0.2, 0.07, 0.97, 0.8], [0, 0, 0, 1]], np.float32)
    out["c2w"] = c2w
    out["ngp_pose"] = ref_utils.nerf_matrix_to_ngp(c2w, scale=4, offset=[0, 0, 0])
    feats = torch.arange(10 * 1 * 4, dtype=torch.float32).reshape(10, 1, 4)
    global_hparams["smo_win_size"] = 5
    out["audio_features_in"] = feats.numpy()
    for idx in (0, 1, 5, 9):
        out[f"audio_features_mode2_{idx}"] = ref_utils.get_audio_features(feats, 2, idx).numpy()
    out["trunc_exp"] = ref_utils.trunc_exp(torch.tensor([-3.0, 0.0, 2.5, 20.0])).numpy()

    # ------------------------------------------------------------------ 5. cond nets + per-sample forward + frames
    rng = np.random.default_rng(7)
    with torch.no_grad():
        for variant in ("may_head", "may_torso", "may_torso_sr"):
            model, sd, ref_hp = models[variant]
            global_hparams.clear()
            global_hparams.update(ref_hp)
            hp = may_hparams(variant)
            fi = syn.synthetic_frame_inputs(hp, 0)
            cond = torch.from_numpy(fi["cond"])
            eye = torch.from_numpy(fi["eye_area_percent"])
            cond_feat = model.cal_cond_feat(cond, eye_area_percent=eye)
            out[f"{variant}.cond_feat"] = cond_feat.numpy().copy()

            if variant == "may_head":
                P = (rng.uniform(-1, 1, (256, 3)) * np.array([0.35, 0.3, 0.4])).astype(np.float32)
                P[:4] = np.array([[1.0, 0.5, -1.0], [0.0, 0.0, 0.0], [-1.0, -0.5, 1.0], [0.3, -0.2, 0.1]], np.float32)
                Dn = rng.standard_normal((256, 3)).astype(np.float32)
                Dn /= np.linalg.norm(Dn, axis=1, keepdims=True)
                sigma, color, amb = model(torch.from_numpy(P), torch.from_numpy(Dn), cond_feat, model.individual_embeddings[0])
                out["fwd.position"] = P
                out["fwd.direction"] = Dn
                out["fwd.sigma"] = sigma.numpy().copy()
                out["fwd.color"] = color.numpy().copy()
                out["fwd.ambient"] = amb.numpy().copy()
                dens = model.density(torch.from_numpy(P), cond_feat)
                out["fwd.density_sigma"] = dens["sigma"].numpy().copy()
                out["fwd.geo_feat_sum"] = dens["geo_feat"].numpy().sum(axis=1)

            HW = 256 if variant == "may_torso_sr" else 64
            pose = syn.synthetic_pose(0)
            rays = ref_utils.get_rays(torch.from_numpy(pose)[None], syn.intrinsics_for(HW, HW), HW, HW, N=-1)
            bg_coords = ref_utils.get_bg_coords(HW, HW, "cpu")
            poses6 = ref_utils.convert_poses(torch.from_numpy(pose)[None])
            bg_color = torch.full((1, HW * HW, 3), 0.5)
            kw = dict(ref_hp)   # the caller splats the whole hparams dict (genefacepp_infer.py:461-463)
            import random
            random.seed(0)      # RADNeRFTorso (non-SR) flips a host coin only when torso_head_aware (false for may_torso)
            res = model.render(rays["rays_o"], rays["rays_d"], cond, bg_coords, poses6, index=0, staged=False,
                               bg_color=bg_color, lm68=torch.from_numpy(fi["lm68"]), perturb=False, force_all_rays=False,
                               T_thresh=0.01, eye_area_percent=eye, **kw)
            rgb = res["rgb_map"].numpy()
            if variant == "may_torso_sr":
                rgb = np.transpose(rgb, (0, 2, 3, 1)).reshape(1, HW * HW, 3)   # [1,3,256,256] -> [1,N,3]
            depth = res["depth_map"].numpy().reshape(1, -1)
            sel = np.arange(0, HW * HW, 7 if HW == 256 else 1)
            out[f"{variant}.render.sel"] = sel.astype(np.int32)
            out[f"{variant}.render.rgb"] = rgb[:, sel].copy()
            out[f"{variant}.render.depth"] = depth[:, sel].copy()
            out[f"{variant}.render.rgb_sum"] = rgb.astype(np.float64).sum(axis=(0, 1))
            if "torso_alpha_map" in res:
                ta = res["torso_alpha_map"].numpy().reshape(-1)
                out[f"{variant}.render.torso_alpha"] = ta[sel].copy()
                out[f"{variant}.render.torso_alpha_sum"] = np.array([ta.astype(np.float64).sum()])
                out[f"{variant}.render.deform_abs_sum"] = np.array([np.abs(res["deform"].numpy()).astype(np.float64).sum()])
            print(variant, "rendered", HW, "rgb mean", rgb.mean(axis=(0, 1)))

    np.savez_compressed(os.path.join(HERE, "ref_python_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
