"""The call sequence of the reference's caller, restated for boxes without the reference tree (the GPU box): what
``GeneFace2Infer.load_secc2video`` (inference/genefacepp_infer.py:163-191) and ``forward_secc2video`` (:433-486) do between a checkpoint
directory and the uint8 frames, statement by statement, on this package's classes.  tests/ref_caller.py runs the reference's own file on the
CPU; this restatement lets the same sequence run on the real kernels."""
import os

import numpy as np
import torch
import yaml

from genefaceplusplus_amd import radnerfs, synthetic as syn


def load_secc2video(torso_model_dir, device):
    """:163-191 -- set_hparams(f"{dir}/config.yaml") (the saved file is the flat dict: utils/commons/hparams.py:167-170), model class by
    hparams['with_sr'], load_ckpt(model, dir, model_name='model', strict=True) (utils/commons/ckpt_utils.py:29-76), torch.compile(model);
    then __init__'s ``.to(device).eval()`` (:128) on the wrapper."""
    with open(os.path.join(torso_model_dir, "config.yaml")) as f:
        hparams = yaml.safe_load(f)
    model = radnerfs.RADNeRFTorsowithSR(hparams) if hparams.get("with_sr") else radnerfs.RADNeRFTorso(hparams)
    state_dict, path = syn.read_checkpoint(torso_model_dir, model_name="model")
    model.load_state_dict(state_dict, strict=True)
    try:
        model = torch.compile(model)
    except Exception:                                # the reference prints the traceback and goes on with the plain module (:187-190)
        import traceback
        traceback.print_exc()
    model.to(device).eval()
    return model, hparams


def forward_secc2video(model, hparams, batch, T_thresh, autocast=True):
    """:433-486, the `low_memory_usage=False` branch + the uint8 conversion of :505: per frame ``render(rays_o[i], rays_d[i], cond_inp[i],
    bg_coords, poses[i], index=i, staged=False, bg_color=bg_color, lm68=lm68s[i], perturb=False, force_all_rays=False, T_thresh=...,
    eye_area_percent=eye_area_percent[i], **hparams)`` under ``torch.cuda.amp.autocast``, `.cpu()` per frame."""
    num_frames = len(batch["poses"])
    pred_rgb_lst = []
    with torch.no_grad(), torch.autocast("cuda", enabled=autocast):
        for i in range(num_frames):
            model_out = model.render(batch["rays_o"][i], batch["rays_d"][i], batch["cond_wins"][i], batch["bg_coords"], batch["poses"][i], index=i, staged=False,
                                     bg_color=batch["bg_img"], lm68=batch["lm68"][i], perturb=False, force_all_rays=False, T_thresh=T_thresh,
                                     eye_area_percent=batch["eye_area_percent"][i], **hparams)
            if hparams.get("with_sr", False):
                pred_rgb = model_out["sr_rgb_map"][0].cpu()
            else:
                pred_rgb = model_out["rgb_map"][0].reshape([512, 512, 3]).permute(2, 0, 1).cpu()
            pred_rgb_lst.append(pred_rgb)
    pred_rgbs = torch.stack(pred_rgb_lst).cpu()
    pred_rgbs = pred_rgbs * 2 - 1
    imgs = pred_rgbs.clamp(-1, 1)
    return ((imgs.permute(0, 2, 3, 1) + 1) / 2 * 255).int().cpu().numpy().astype(np.uint8)


def make_batch(dataset, hp, n_frames, device, get_rays, convert_poses):
    """the render part of the batch as prepare_batch_from_inp lays it out (:246-275, 411-431), from synthetic driving signals"""
    batch = {"rays_o": [], "rays_d": [], "poses": [], "cond_wins": [], "lm68": [], "eye_area_percent": []}
    for i in range(n_frames):
        ngp_pose = torch.from_numpy(syn.synthetic_pose(i))[None]
        rays = get_rays(ngp_pose.to(device), dataset.intrinsics, dataset.H, dataset.W, N=-1)
        batch["rays_o"].append(rays["rays_o"].to(device))
        batch["rays_d"].append(rays["rays_d"].to(device))
        batch["poses"].append(convert_poses(ngp_pose).to(device))
        fi = syn.synthetic_frame_inputs(hp, i)
        batch["cond_wins"].append(torch.from_numpy(fi["cond"]).to(device))
        batch["lm68"].append(torch.from_numpy(fi["lm68"]))
        batch["eye_area_percent"].append(torch.from_numpy(fi["eye_area_percent"]))
    batch["lm68"] = torch.stack(batch["lm68"]).to(device)
    batch["eye_area_percent"] = torch.stack(batch["eye_area_percent"]).to(device)
    batch["bg_img"] = dataset.bg_img.reshape([1, -1, 3]).to(device)
    batch["bg_coords"] = dataset.bg_coords.to(device)
    return batch
