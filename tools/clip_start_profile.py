#!/usr/bin/env python
"""Where the fixed cost of a SHORT clip job goes (the driver's bench times 20 frames: start-up is a fifth of it).  Host time of ClipRenderer.start(), GPU time of
its launches, and the job time for n frames with 1 / 4 frames per graph launch.   python tools/clip_start_profile.py [variant] [hw] [precision] [frames]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import frame_case, build_model
from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.clip import ClipRenderer

variant = sys.argv[1] if len(sys.argv) > 1 else "may_torso"
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
case = frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = precision
hp = case["hp"]
F = n + 5
fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
         "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
for group in (1, 4):
    cr = ClipRenderer(model, HW, HW, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=0.01, group=group, render_kwargs=dict(hp, use_head_for_torso=True))
    clip = cr.prepare(batch, dev)
    out = torch.empty(n, *cr.out_hw, 3, dtype=torch.uint8, device=dev)
    cr.render_to_device(clip, range(5), out=out[:5])
    torch.cuda.synchronize()
    rows = []
    for rep in range(6):
        cr._cond_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cr.start(clip, range(5, 5 + n), out)
        t1 = time.perf_counter()
        cr.issue()
        cr.join()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)))
    r = np.array(rows[1:])
    print(f"group {cr.group}: start() host {r[:, 0].mean():.3f} ms, issue+join host {r[:, 1].mean():.3f} ms, job {r[:, 2].mean():.3f} ms for {n} frames "
          f"= {n / r[:, 2].mean() * 1e3:.0f} frames/s (min job {r[:, 2].min():.3f} ms)")
    # the same job with the conditioning cached (what a second job over the same clip pays)
    rows = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cr.start(clip, range(5, 5 + n), out)
        t1 = time.perf_counter()
        cr.issue()
        cr.join()
        torch.cuda.synchronize()
        rows.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
    r = np.array(rows[1:])
    print(f"   conditioning cached: start() host {r[:, 0].mean():.3f} ms, job {r[:, 1].mean():.3f} ms")
