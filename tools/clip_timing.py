#!/usr/bin/env python
"""Where a clip's time goes: device-resident stack vs pinned-host delivery, ring depth, graph vs eager (run on the GPU box)."""
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
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
case = frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = "fp16"
hp = case["hp"]
fi = [syn.synthetic_frame_inputs(hp, i) for i in range(F)]
batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
         "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
lanes_list = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2]
for graph, ring, lanes in [(g, r, l) for g in (True, False) for l in lanes_list for r in ((4, 8) if g else (4,))]:
    if True:
        r = ClipRenderer(model, HW, HW, case["intr"], bg_img=torch.from_numpy(case["bg_color"]), T_thresh=0.01, ring=ring, use_graph=graph, lanes=lanes)
        clip = r.prepare(batch, dev)
        r.render_to_device(clip, range(4))
        torch.cuda.synchronize()
        t = time.perf_counter()
        r.render_to_device(clip)
        t_launch = time.perf_counter() - t
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t
        n = [0]
        t = time.perf_counter()
        r.render_to_host(clip, sink=lambda i, a: n.__setitem__(0, n[0] + 1))
        t_host = time.perf_counter() - t
        del clip
        print(f"graph={graph} lanes={lanes} ring={ring}: device-resident {F / t_dev:8.1f} fps (host launch loop alone {1e3 * t_launch / F:.3f} ms/frame), "
              f"to pinned host {F / t_host:8.1f} fps")
