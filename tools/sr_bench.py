#!/usr/bin/env python
"""Time of the super-resolution stage alone (gfpp_sr_forward: 4 launches, 256^2 -> 512^2), HIP events, for kernel experiments.   python tools/sr_bench.py [reps] [noise mode: const | random | none]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "const"
dev = torch.device("cuda:0")
sd = syn.synthetic_sr_state(prefix="")
net = Superresolution(channels=3)
net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
net = net.to(dev).eval()
x = torch.rand(1, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(5):
        y = net(x, noise_mode=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = net(x, noise_mode=mode)
    e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"sr_bench[{mode}]: {us:.1f} us per forward ({77.3e9 / us / 1e6:.0f} TFLOP/s of f16 MFMA work, 19.3 + 38.7 + 19.3 GFLOP), checksum {float(y.double().sum()):.6e}")
