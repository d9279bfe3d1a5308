#!/usr/bin/env python
"""Render a short synthetic clip through ClipRenderer (nothing else) -- the target process of the counter passes of one WORKLOAD (tools/pmc_workload.sh):
    profile_clip.py <variant> <hw> <precision> [frames per graph launch] [frames]
<variant> = may_torso | may_torso_sr | may_head ... (random-init synthetic field) or trained_<variant> (the fitted procedural field, tests/golden/trained/, with the
procedural clip's own driving signals and background).  Plain launches (no graph), one lane: every kernel of a frame (group) is its own dispatch in the trace, in order."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import frame_case, build_model, trained_case
from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.clip import ClipRenderer

variant = sys.argv[1] if len(sys.argv) > 1 else "may_torso"
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
group = int(sys.argv[4]) if len(sys.argv) > 4 else 1
F = int(sys.argv[5]) if len(sys.argv) > 5 else 3 * max(group, 1)
dev = torch.device("cuda:0")
trained = variant.startswith("trained_")
case = trained_case(variant[len("trained_"):], HW) if trained else frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = precision
hp = case["hp"]
if trained:
    batch = case["clip"].clip_batch(hp["smo_win_size"], range(F))
else:
    fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
    batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
             "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
cr = ClipRenderer(model, HW, HW, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=0.01, use_graph=False, lanes=1, group=group,
                  render_kwargs=dict(hp, use_head_for_torso=True))
clip = cr.prepare(batch, dev)
out = cr.render_to_device(clip)
torch.cuda.synchronize()
print("frames", tuple(out.shape), "frames per launch", cr.group, "mean", float(out.float().mean()))
