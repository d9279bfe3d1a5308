#!/usr/bin/env python
"""Frame-loop experiments on one GPU: frames/s of a resident clip (512x512 head+torso, bf16) by who issues the graph launches (C loop / Python loop),
how many frames are in flight (lanes) and how far the issuing thread may run ahead of the GPU.  Same-call A/B: everything in one process."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.clip import ClipRenderer  # noqa: E402
from helpers import frame_case, build_model  # noqa: E402

variant, HW, F = (sys.argv[1] if len(sys.argv) > 1 else "may_torso"), int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 240
dev = torch.device("cuda:0")
case = frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = "bf16"
hp = case["hp"]
fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
         "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
for lanes in (1, 2, 3, 4):
    r = ClipRenderer(model, HW, HW, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=0.01, use_graph=True, lanes=lanes)
    clip = r.prepare(batch, dev)
    out = torch.empty(F, *r.out_hw, 3, dtype=torch.uint8, device=dev)
    r.render_to_device(clip, range(8), out=out[:8])
    torch.cuda.synchronize()
    for mode, ahead in (("c", 0), ("c", 1), ("c", 2), ("c", 4), ("python", 0)):
        r.replay_mode, r.max_ahead = mode, ahead
        best = 0.0
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r.render_to_device(clip, out=out)
            t_issue = time.perf_counter() - t
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            best = max(best, F / dt)
        print(f"lanes={lanes} replay={mode:6s} max_ahead={ahead}: {best:8.1f} frames/s   (issue {1e3 * t_issue / F:.3f} ms/frame)", flush=True)
