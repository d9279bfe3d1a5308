#!/usr/bin/env python
"""tests/golden/trained/fit_log.json (written by tools/make_trained_checkpoint.py) -> profiles/r06_trained_fit.md: the loss / PSNR-vs-target curves of every stage."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "trained", "fit_log.json")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_trained_fit.md")
log = json.load(open(src))
a = log["args"]
out = ["# The trained procedural field: fit curves", "",
       f"`python tools/make_trained_checkpoint.py --family both` on {log['device']} (gpurun), seed {a['seed']}, {a['frames']} frames of "
       "`genefaceplusplus_amd.procedural.ProceduralClip` (every 8th held out), head / torso / SR stages = "
       f"{a['head_steps']} / {a['torso_steps']} / {a['sr_steps']} steps.  " + log.get("note", ""), "",
       "Every step runs the package's own training path: `march_rays_train` -> grid encoders + MLPs under autograd (`csrc/train*.hip`, fused whole-MLP launches under "
       "autocast) -> `composite_rays_train`, `update_extra_state` every 16 steps, Adam with the reference's parameter groups, `GradScaler`.  Convergence of this loop IS the "
       "end-to-end test of the backward kernels: a wrong table gradient, weight gradient or composite backward does not reach 40 dB.", "",
       "PSNR figures below: `train_psnr` = running mean over the step's sampled pixels (training-mode render, perturbed samples); `val` = full held-out frames through the "
       "INFERENCE path (fused persistent launch) against the analytic target.", ""]
for run in log["runs"]:
    out += [f"## family `{run['family']}` ({'256x256 + super-resolution' if run['family'] == 'sr' else '512x512'})", ""]
    st = run["stages"]
    h = st["head"]
    out += [f"### head stage: `{h['variant']}`, {h['steps']} steps in {h['seconds']} s ({h['ms_per_step_incl_targets_and_logging']} ms per step including the analytic targets, "
            "the occupancy refreshes and the validation renders)", "",
            "| step | train MSE | train PSNR (dB) | val PSNR fp32 (dB) | occupied cells of 128^3 | samples per step (65 536 rays) | mean density |", "|---|---|---|---|---|---|---|"]
    for c in h["curve"]:
        if c["step"] % 500 == 0 or c["step"] in (100, 200, 300):
            out.append(f"| {c['step']} | {c['train_mse']:.5f} | {c['train_psnr']:.2f} | {c.get('val_psnr_fp32', '')} | {c['occupied_cells']} | {c['samples_per_step']} | {c['mean_density']:.3f} |")
    t = st["torso"]
    out += ["", f"### torso stage: `{t['variant']}`, {t['steps']} steps in {t['seconds']} s (head frozen; loss on `torso_rgb_map` vs the torso-over-background target)", "",
            "| step | torso MSE | torso PSNR (dB) | mean torso density | torso cells over threshold (of 16 384) |", "|---|---|---|---|---|"]
    for c in t["curve"]:
        if c["step"] % 500 == 0 or c["step"] in (100, 200, 300):
            out.append(f"| {c['step']} | {c['train_torso_mse']:.5f} | {c['train_torso_psnr']:.2f} | {c['mean_density_torso']:.3f} | {c['torso_cells_over_thresh']} |")
    if "sr" in st:
        s = st["sr"]
        out += ["", f"### super-resolution stage: {s['steps']} steps in {s['seconds']} s (StyleGAN2 blocks in autograd-visible torch ops, MSE against the 512^2 target)", "",
                "| step | MSE | PSNR (dB) |", "|---|---|---|"]
        for c in s["curve"]:
            if c["step"] % 100 == 0 or c["step"] == 20:
                out.append(f"| {c['step']} | {c['sr_mse']:.5f} | {c['sr_psnr']:.2f} |")
    out += ["", "### final: full frames through the inference path vs the analytic target (after rounding tables / grids to float16-representable values)", "",
            "| model | precision | val mean (dB) | val min (dB) | train-frame mean (dB) | SR 512^2 val mean (dB) | occupied cells | compact file (bytes) |", "|---|---|---|---|---|---|---|---|"]
    for name, f in run["final_psnr_vs_target"].items():
        for p in ("fp32", "fp16", "bf16"):
            out.append(f"| {name} | {p} | {f[p]['val']['mean']} | {f[p]['val']['min']} | {f[p]['train']['mean']} | {f[p]['val'].get('sr_mean', '')} | {f['occupied_cells']} | {f['npz_bytes']} |")
    out.append("")
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst)
