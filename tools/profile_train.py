#!/usr/bin/env python
"""One (or a few) training steps of the May head NeRF -- the target process of the training-side rocprofv3 pass, and a samples/s meter.

    python tools/profile_train.py [n_rays=65536] [steps=6] [amp]

`amp`: the step under torch.autocast(fp16) with a GradScaler, the reference's training configuration (egs/datasets/May/lm3d_radnerf.yaml:5 `amp: true`;
utils/commons/trainer.py wraps the step the same way): half grid tables / features / table-gradient accumulators, half MLP GEMMs.

A step = what tasks/radnerfs/radnerf.py does per batch: training-mode render() of n_rays random rays of a 512x512 frame (march_rays_train ->
networks under autograd -> composite_rays_train), MSE loss against a synthetic target, backward, Adam update (the reference's lr 5e-4;
base.yaml:36).  Prints rays/s and samples/s (samples = occupied samples marched per step)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import frame_case, build_model  # noqa: E402
from genefaceplusplus_amd.radnerfs import camera  # noqa: E402

amp = "amp" in sys.argv[1:]
argv = [a for a in sys.argv[1:] if a != "amp"]
n_rays = int(argv[0]) if len(argv) > 0 else 65536
steps = int(argv[1]) if len(argv) > 1 else 6
dev = torch.device("cuda:0")
case = frame_case("may_head", 512)
model = build_model(case, dev, "fused")
model.train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.99), eps=1e-15)
pose = torch.from_numpy(case["pose"]).to(dev)
cond = torch.from_numpy(case["cond"]).to(dev)
bg_coords = camera.get_bg_coords(512, 512, dev)
hp = case["hp"]
samples = []
times = []
scaler = torch.amp.GradScaler("cuda", enabled=amp)
torch.manual_seed(0)
loss = None


def one_step():
    global loss
    rays = camera.get_rays(pose, case["intr"], 512, 512, N=n_rays)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        res = model.render(rays["rays_o"], rays["rays_d"], cond, bg_coords, camera.convert_poses(pose), index=0, dt_gamma=hp["dt_gamma"],
                           bg_color=torch.full((1, rays["rays_o"].shape[1], 3), 0.5, device=dev), perturb=True, force_all_rays=False, max_steps=hp["max_steps"])
        target = torch.rand_like(res["rgb_map"].float())
        loss = ((res["rgb_map"].float() - target) ** 2).mean() + 1e-3 * res["ambient"].float().mean()
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for it in range(steps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one_step()
    torch.cuda.synchronize()
    if it >= 2:
        times.append(time.perf_counter() - t0)
        samples.append(int(model.step_counter[(model.local_step - 1) % 16, 0]))
t = float(np.mean(times))
line = (f"profile_train{' [amp]' if amp else ''}: {n_rays} rays/step, {np.mean(samples):.0f} samples/step, {t * 1e3:.2f} ms/step -> {n_rays / t / 1e6:.2f} Mrays/s, {np.mean(samples) / t / 1e6:.2f} Msamples/s "
        f"(loss {float(loss):.5f})")
# the same steps as a trainer issues them: back to back, the host running ahead of the GPU (one synchronisation around the lot, none per step)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(steps):
    one_step()
torch.cuda.synchronize()
tp = (time.perf_counter() - t0) / steps
print(line + f"; back to back (no synchronisation per step): {tp * 1e3:.2f} ms/step")
