#!/usr/bin/env python
"""Fit the package's own models to a procedural talking-head clip with the package's own training path, and write the result as a checkpoint
in the reference's layout -- the "weights that are not random" every other number of this repository can be re-taken on.

    python tools/make_trained_checkpoint.py [--family plain|sr|both] [--out gpurun_out/trained] [--head-steps 4000] [--torso-steps 2500] [--sr-steps 400]

Stages, as the reference trains a person (docs/train_and_infer/guide.md; tasks/radnerfs/radnerf.py:101-176, radnerf_torso.py:27-150,
radnerf_torso_sr.py:64-222), on `genefaceplusplus_amd.procedural.ProceduralClip` instead of a video:

  head   RADNeRF (family `plain`, 512^2) / RADNeRFwithSR (family `sr`, 256^2): mark_untrained_grid, then per step one random frame, n_rays random
         pixels, training-mode render() (march_rays_train -> networks under autograd -> composite_rays_train, csrc/train*.hip) with the torso
         treated as background (bg_color = bg_torso_img), loss = MSE + lambda_weights_entropy * entropy(weights_sum) + ramped lambda_ambient *
         |ambient| outside the face mask; update_extra_state() every `update_extra_interval` steps; Adam (eps 1e-15) with the reference's three
         parameter groups (networks lr, grids 10 x lr, attention net 5 x lr), torch.autocast(fp16) + GradScaler (`amp: true`);
  torso  RADNeRFTorso / RADNeRFTorsowithSR initialised from the head model (strict=False), everything without 'torso' in its name frozen;
         loss = MSE(torso_rgb_map, bg_torso_img) + lambda_weights_entropy * entropy(torso_alpha) (torso_train_mode 1); torso occupancy refreshed
         every `update_extra_interval` steps;
  sr     (family `sr`) the StyleGAN2 super-resolution net on the finished NeRF's 256^2 frames against the 512^2 target (MSE; the reference adds
         LPIPS, whose VGG weights are a download).

Differences from the reference's schedule, all forced by the budget of seconds instead of hours: the exponential learning-rate decay 0.1 ** (step /
250 000) and the ambient ramp min(step / 250 000, 1) run over THIS run's step count; no lip fine-tuning stage (LPIPS).

Before the final evaluation the grid tables, occupancy grids, noise buffers and per-frame codes are rounded to float16-representable values
(what the 16-bit render modes read anyway), so that the compact fixture (synthetic.save_compact_state) is lossless.

Outputs under --out: <variant>/model_ckpt_steps_<n>.ckpt + config.yaml (reference layout, synthetic.write_checkpoint), <variant>.npz (compact
fixture), fit_log.json (loss / PSNR curves, final full-frame PSNR against the target in every render precision).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from genefaceplusplus_amd import radnerfs, synthetic as syn          # noqa: E402
from genefaceplusplus_amd.configs import may_hparams                 # noqa: E402
from genefaceplusplus_amd.procedural import ProceduralClip           # noqa: E402
from genefaceplusplus_amd.radnerfs import camera                      # noqa: E402

TRAIN_HP = {"lr": 5e-4, "optimizer_adam_beta1": 0.9, "optimizer_adam_beta2": 0.999, "n_rays": 65536, "update_extra_interval": 16,
            "lambda_weights_entropy": 1e-4, "lambda_ambient": 0.1, "torso_train_mode": 1}
VAL_EVERY = 8               # every 8th frame of the clip is held out of training and is what the PSNR figures are taken on


def psnr(a, b):
    return float(-10.0 * torch.log10(((a.float() - b.float()) ** 2).mean().clamp(min=1e-12)))


def entropy(alphas):
    a = alphas.clamp(1e-5, 1 - 1e-5)
    return torch.mean(-a * torch.log2(a) - (1 - a) * torch.log2(1 - a))


class Fit:
    def __init__(self, family, dev, args):
        self.family, self.dev, self.args = family, dev, args
        self.sr = family == "sr"
        self.head_variant, self.torso_variant = ("may_head_sr", "may_torso_sr") if self.sr else ("may_head", "may_torso")
        self.HW = 256 if self.sr else 512
        self.intr = syn.intrinsics_for(self.HW, self.HW)
        self.clip = ProceduralClip(T=args.frames, seed=args.seed)
        self.train_frames = [k for k in range(self.clip.T) if k % VAL_EVERY != VAL_EVERY - 1]
        self.val_frames = [k for k in range(self.clip.T) if k % VAL_EVERY == VAL_EVERY - 1]
        self.bg_coords_full = camera.get_bg_coords(self.HW, self.HW, dev)
        self.poses_dev = torch.from_numpy(self.clip.ngp_poses).to(dev)
        self.log = {"family": family, "frames": self.clip.T, "val_frames": len(self.val_frames), "stages": {}}

    # -- one frame's training sample (what RADNeRFDataset.__getitem__ yields, dataset_utils.py:304-432) ------------------------------------------
    def sample(self, k, n_rays):
        hp = self.hp
        # (the *_sr tasks train on ALL pixels of the 256^2 frame, in order: dataset_utils.py:322-323, the SR net needs the image)
        rays = camera.get_rays(self.poses_dev[k:k + 1], self.intr, self.HW, self.HW, N=-1 if self.sr else n_rays)
        inds = rays["inds"][0]
        bgc = self.bg_coords_full[0, inds]
        tgt = self.clip.target(k, rays["rays_o"][0], rays["rays_d"][0], bgc)
        return {"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "bg_coords": bgc[None], "pose": camera.convert_poses(self.poses_dev[k:k + 1]),
                "cond": self.clip.cond_window(k, hp["smo_win_size"]).to(self.dev), "eye": self.clip.eye_area_percents[k].to(self.dev).reshape(1, 1),
                "lm68": self.clip.lm68s[k].reshape(-1).to(self.dev), "idx": k, **tgt}

    # -- stage 1 ----------------------------------------------------------------------------------------------------------------------------------
    def train_head(self):
        args, dev = self.args, self.dev
        self.hp = hp = dict(may_hparams(self.head_variant), **TRAIN_HP)
        if self.sr:
            # the shipped yamls disagree (lm3d_radnerf_sr.yaml: eye_blink_dim 2, lm3d_radnerf_torso_sr.yaml: 4) and the torso task loads the head model's
            # state into its own (radnerf_torso_sr.py:69-72, a size mismatch raises even with strict=False): the head a torso_sr model was built on had 4
            hp["eye_blink_dim"] = may_hparams(self.torso_variant)["eye_blink_dim"]
        self.head_hp = {k: v for k, v in hp.items() if k not in TRAIN_HP}
        model = getattr(radnerfs, radnerfs.CLASSES[self.head_variant])(hp).to(dev)
        model.executor = "fused"
        model.train()
        if self.sr:
            model.on_train_nerf()
        model.conds = self.clip.conds
        model.mark_untrained_grid(self.poses_dev, self.intr)
        named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        grids = [p for k, p in named if "position_embedder" in k or "ambient_embedder" in k]
        att = [p for k, p in named if "cond_att_net" in k]
        nets = [p for k, p in named if not ("position_embedder" in k or "ambient_embedder" in k or "cond_att_net" in k)]
        betas = (hp["optimizer_adam_beta1"], hp["optimizer_adam_beta2"])
        opt = torch.optim.Adam([{"params": nets, "lr": hp["lr"]}, {"params": grids, "lr": hp["lr"] * 10}, {"params": att, "lr": hp["lr"] * 5}], betas=betas, eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", enabled=True)
        steps = args.head_steps
        curve, run = [], []
        rng = np.random.default_rng(args.seed + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(steps):
            if step % hp["update_extra_interval"] == 0:
                model.update_extra_state()
            lr = max(hp["lr"] * 0.1 ** (step / steps), 1e-5)
            for g, m in zip(opt.param_groups, (1, 10, 5)):
                g["lr"] = lr * m
            k = int(rng.choice(self.train_frames))
            s = self.sample(k, hp["n_rays"])
            with torch.autocast("cuda", dtype=torch.float16):
                out = model.render(s["rays_o"], s["rays_d"], s["cond"], s["bg_coords"], s["pose"], index=k, bg_color=s["bg_torso"][None], perturb=True,
                                   force_all_rays=False, eye_area_percent=s["eye"], dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"])
                pred = out["rgb_map"].float()
                if pred.dim() == 4:                                      # RADNeRFwithSR: [1,3,256,256] view of the all-pixels sample (radnerf_sr.py:185-186)
                    pred = pred.permute(0, 2, 3, 1).reshape(1, -1, 3)
                mse = torch.mean((pred - s["gt"][None]) ** 2)
                ent = entropy(out["weights_sum"].float())
                nonface = ~s["face_mask"]
                amb = (out["ambient"].float().abs() * nonface).sum() / (nonface.sum() + 1)
                loss = mse + hp["lambda_weights_entropy"] * ent + min(step / steps, 1.0) * hp["lambda_ambient"] * amb
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            run.append(mse.detach())
            if (step + 1) % args.log_every == 0:
                m = float(torch.stack(run).mean())
                run = []
                curve.append({"step": step + 1, "train_mse": m, "train_psnr": -10 * math.log10(max(m, 1e-12)), "mean_density": float(model.mean_density),
                              "occupied_cells": int((model.density_grid > min(model.mean_density, model.density_thresh)).sum()),
                              "samples_per_step": int(model.step_counter[(model.local_step - 1) % 16, 0])})
                if (step + 1) % (args.log_every * 5) == 0 or step + 1 == steps:
                    curve[-1]["val_psnr_fp32"] = self.eval_psnr(model, self.val_frames[:4], "fp32", head_only=True)["mean"]
                    model.train()
                print(f"[{self.family} head] {curve[-1]}", flush=True)
        torch.cuda.synchronize()
        self.log["stages"]["head"] = {"variant": self.head_variant, "steps": steps, "seconds": round(time.perf_counter() - t0, 1),
                                      "ms_per_step_incl_targets_and_logging": round(1e3 * (time.perf_counter() - t0) / steps, 2), "curve": curve}
        self.head = model
        return model

    # -- stage 2 ----------------------------------------------------------------------------------------------------------------------------------
    def train_torso(self):
        args, dev = self.args, self.dev
        self.hp = hp = dict(may_hparams(self.torso_variant), **TRAIN_HP)
        model = getattr(radnerfs, radnerfs.CLASSES[self.torso_variant])(hp).to(dev)
        missing = model.load_state_dict(self.head.state_dict(), strict=False)
        assert all("torso" in k or "head_color_weights_encoder" in k or "lm68" in k for k in missing.missing_keys), missing.missing_keys
        model.density_bitfield = self.head.density_bitfield.clone()
        model.mean_density, model.mean_count = self.head.mean_density, self.head.mean_count
        model.executor = "fused"
        model.train()
        if self.sr:
            model.on_train_torso_nerf()
        else:
            for k, p in model.named_parameters():
                if "torso" not in k:
                    p.requires_grad_(False)
        model.poses = torch.from_numpy(self.clip.ngp_poses)
        model.lm68s = self.clip.lm68s
        model.conds = self.clip.conds
        named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        grids = [p for k, p in named if "torso_embedder" in k]
        nets = [p for k, p in named if "torso_embedder" not in k]
        betas = (hp["optimizer_adam_beta1"], hp["optimizer_adam_beta2"])
        opt = torch.optim.Adam([{"params": nets, "lr": hp["lr"]}, {"params": grids, "lr": hp["lr"] * 10}], betas=betas, eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", enabled=True)
        steps = args.torso_steps
        curve, run = [], []
        rng = np.random.default_rng(args.seed + 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(steps):
            if step % hp["update_extra_interval"] == 0:
                model.update_extra_state()
            lr = max(hp["lr"] * 0.1 ** (step / steps), 1e-5)
            for g, m in zip(opt.param_groups, (1, 10)):
                g["lr"] = lr * m
            k = int(rng.choice(self.train_frames))
            s = self.sample(k, hp["n_rays"])
            with torch.autocast("cuda", dtype=torch.float16):
                kw = dict(lm68=s["lm68"], eye_area_percent=s["eye"]) if self.sr else {}
                out = model.render(s["rays_o"], s["rays_d"], s["cond"], s["bg_coords"], s["pose"], index=k, bg_color=s["bg"][None], perturb=True, force_all_rays=False,
                                   dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"], **kw)
                pred = out["torso_rgb_map"].float()
                if self.sr:                                              # [1,3,256,256] view of the n_rays = 256^2 = all-pixels sample (radnerf_torso_sr.py:153-154)
                    pred = pred.permute(0, 2, 3, 1).reshape(1, -1, 3)
                mse = torch.mean((pred - s["bg_torso"][None]) ** 2)
                loss = mse + hp["lambda_weights_entropy"] * entropy(out["torso_alpha_map"].float())
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            run.append(mse.detach())
            if (step + 1) % args.log_every == 0:
                m = float(torch.stack(run).mean())
                run = []
                curve.append({"step": step + 1, "train_torso_mse": m, "train_torso_psnr": -10 * math.log10(max(m, 1e-12)), "mean_density_torso": float(model.mean_density_torso),
                              "torso_cells_over_thresh": int((model.density_grid_torso > min(model.mean_density_torso, model.density_thresh_torso)).sum())})
                print(f"[{self.family} torso] {curve[-1]}", flush=True)
        torch.cuda.synchronize()
        self.log["stages"]["torso"] = {"variant": self.torso_variant, "steps": steps, "seconds": round(time.perf_counter() - t0, 1), "curve": curve}
        model.requires_grad_(True)
        self.model = model
        return model

    # -- stage 3 ----------------------------------------------------------------------------------------------------------------------------------
    def train_sr(self):
        args, dev = self.args, self.dev
        model = self.model.eval()
        model.on_train_superresolution()
        sr = model.sr_net
        hp = self.hp
        opt = torch.optim.Adam([p for p in sr.parameters() if p.requires_grad], lr=hp["lr"] * 4, betas=(0.9, 0.99), eps=1e-15)
        intr512 = syn.intrinsics_for(512, 512)
        rng = np.random.default_rng(args.seed + 3)
        curve, run = [], []
        t0 = time.perf_counter()
        for step in range(args.sr_steps):
            k = int(rng.choice(self.train_frames))
            with torch.no_grad():
                sr.eval()
                low = self.render_frame(model, k, "fp32")["rgb_map"].float().clone()          # [1,3,256,256]
                gt512 = self.clip.frame(k, 512, intr512, dev)["gt"].reshape(1, 512, 512, 3).permute(0, 3, 1, 2)
            sr.train()
            pred = sr(low, noise_mode="random")
            loss = torch.mean((pred - gt512) ** 2)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            run.append(loss.detach())
            if (step + 1) % max(1, args.log_every // 5) == 0:
                m = float(torch.stack(run).mean())
                run = []
                curve.append({"step": step + 1, "sr_mse": m, "sr_psnr": -10 * math.log10(max(m, 1e-12))})
                print(f"[{self.family} sr] {curve[-1]}", flush=True)
        sr.eval()
        model.requires_grad_(True)
        torch.cuda.synchronize()
        self.log["stages"]["sr"] = {"steps": args.sr_steps, "seconds": round(time.perf_counter() - t0, 1), "curve": curve}
        self.head.sr_net.load_state_dict(sr.state_dict())

    # -- inference-path evaluation against the target ----------------------------------------------------------------------------------------------
    def render_frame(self, model, k, precision, head_only=False, tgt=None):
        dev, hp = self.dev, model.hparams
        model.eval()
        model.precision = precision
        tgt = tgt if tgt is not None else self.clip.frame(k, self.HW, self.intr, dev)
        bg = tgt["bg_torso"] if head_only else tgt["bg"]
        with torch.no_grad():
            return model.render(tgt["rays_o"], tgt["rays_d"], self.clip.cond_window(k, hp["smo_win_size"]).to(dev), self.bg_coords_full, camera.convert_poses(self.poses_dev[k:k + 1]),
                                index=0, bg_color=bg[None], perturb=False, force_all_rays=False, T_thresh=0.01, dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"],
                                lm68=self.clip.lm68s[k].reshape(-1).to(dev), eye_area_percent=self.clip.eye_area_percents[k].to(dev).reshape(1, 1), sr_noise_mode="const")

    def eval_psnr(self, model, frames, precision, head_only=False):
        vals, vals_sr = [], []
        for k in frames:
            tgt = self.clip.frame(k, self.HW, self.intr, self.dev)
            res = self.render_frame(model, k, precision, head_only, tgt)
            rgb = res["rgb_map"].float()
            if rgb.dim() == 4:
                rgb = rgb.permute(0, 2, 3, 1)
            vals.append(psnr(rgb.reshape(-1, 3), tgt["gt"]))
            if "sr_rgb_map" in res and not head_only:
                gt512 = self.clip.frame(k, 512, syn.intrinsics_for(512, 512), self.dev)["gt"]
                vals_sr.append(psnr(res["sr_rgb_map"].float().permute(0, 2, 3, 1).reshape(-1, 3), gt512))
        out = {"mean": round(float(np.mean(vals)), 2), "min": round(float(np.min(vals)), 2), "frames": len(vals)}
        if vals_sr:
            out["sr_mean"] = round(float(np.mean(vals_sr)), 2)
        return out

    # -- rounding + writing ----------------------------------------------------------------------------------------------------------------------
    def finalize(self, out_dir):
        for model in (self.head, self.model):
            model.eval()
            with torch.no_grad():
                for k, t in list(model.state_dict().items()):
                    if syn.compact_f16_key(k) and t.dtype == torch.float32:
                        t.copy_(t.clamp(-65504, 65504).half().float())
            model._pipeline = None
        final = {}
        for name, model, head_only in ((self.head_variant, self.head, True), (self.torso_variant, self.model, False)):
            final[name] = {p: {"val": self.eval_psnr(model, self.val_frames, p, head_only), "train": self.eval_psnr(model, self.train_frames[::16], p, head_only)}
                           for p in ("fp32", "fp16", "bf16")}
            model.precision = "auto"
            hp = self.head_hp if head_only else may_hparams(name)
            steps = self.args.head_steps + (0 if head_only else self.args.torso_steps)
            work = os.path.join(out_dir, name)
            syn.write_checkpoint(work, name, hp, steps=steps, state_dict=model.state_dict())
            syn.save_compact_state(os.path.join(out_dir, name + ".npz"), model.state_dict())
            final[name]["npz_bytes"] = os.path.getsize(os.path.join(out_dir, name + ".npz"))
            final[name]["occupied_cells"] = int(np.unpackbits(model.density_bitfield.cpu().numpy()).sum())
            print(f"[{self.family}] {name}: {final[name]}", flush=True)
        self.log["final_psnr_vs_target"] = final
        return self.log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="both", choices=["plain", "sr", "both"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trained"))
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--head-steps", type=int, default=4000)
    ap.add_argument("--torso-steps", type=int, default=2500)
    ap.add_argument("--sr-steps", type=int, default=400)
    ap.add_argument("--log-every", type=int, default=100)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "training runs on the HIP kernels (there is no CPU path)"
    dev = torch.device("cuda:0")
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(args.seed)
    import random
    random.seed(args.seed)
    logs = []
    for fam in (("plain", "sr") if args.family == "both" else (args.family,)):
        fit = Fit(fam, dev, args)
        fit.train_head()
        fit.train_torso()
        if fam == "sr" and args.sr_steps > 0:
            fit.train_sr()
        logs.append(fit.finalize(args.out))
        with open(os.path.join(args.out, "fit_log.json"), "w") as f:
            json.dump({"args": vars(args), "device": torch.cuda.get_device_name(0), "runs": logs}, f, indent=1)
    print("done:", args.out)


if __name__ == "__main__":
    main()
