#!/usr/bin/env python
"""Render a few synthetic 512x512 head+torso frames (nothing else) -- the target process for rocprofv3 runs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import frame_case, build_model, product_render

variant = sys.argv[1] if len(sys.argv) > 1 else "may_torso"
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
precision = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda:0")
case = frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = precision
for _ in range(frames):
    product_render(model, case, dev, "hip")
torch.cuda.synchronize()
if precision != "fp32" and os.environ.get("GFPP_PHASES"):
    pc = model.pipeline().enable_phase_cycles(HW * HW)
    product_render(model, case, dev, "hip")
    torch.cuda.synchronize()
    pc = pc.cpu().numpy()
    if __import__("genefaceplusplus_amd.tuning", fromlist=["LIB"]).LIB["trip_pool"]:
        print("trip: k-cycles summed over wavefronts: gather / compaction barriers / evaluate / wait for last block / composite | wavefront rounds, blocks, "
              "longest workgroup round (cycles); per block = evaluate / blocks")
        for k in range(16):
            if pc[k].sum():
                print(k, [int(v // 1000) for v in pc[k][:5]], int(pc[k][5]), int(pc[k][6]), int(pc[k][7]), "per block", int(pc[k][2] // max(pc[k][6], 1)))
    else:
        print("trip: cycles summed over wavefronts (k = 1e3) copy / samples / evaluate / composite | pos-enc / amb-MLP / amb-enc / sigma+colour")
        for k in range(16):
            if pc[k].sum():
                print(k, [int(v // 1000) for v in pc[k]])
alive, smp = model.pipeline().trip_counters(HW * HW)
print("alive", alive[:8], "samples", smp[:8], "total", smp.sum())
