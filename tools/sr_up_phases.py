#!/usr/bin/env python
"""Where a workgroup of the polyphase up-sampling launch (k_sr_up_poly) spends its time: thread 0's shader-clock stamps at the phase boundaries (the kernel's
profiling instantiation, gfpp_sr_ws.up_prof), averaged over the 704 workgroups of a forward.   python tools/sr_up_phases.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution  # noqa: E402

dev = torch.device("cuda:0")
net = Superresolution(channels=3)
net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in syn.synthetic_sr_state(prefix="").items()}, strict=True)
net = net.to(dev).eval()
x = torch.rand(1, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(20):
        net(x, noise_mode="random", clamp01=True)
    net.up_prof = torch.zeros(704, 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        net(x, noise_mode="random", clamp01=True)
    torch.cuda.synchronize()
t = net.up_prof.cpu().numpy().astype(np.float64)
names = ["first loads (patch 0 + nine taps)", "products, K slice 0 (+ 3 barriers)", "patch slice 1 (+ last tap group)", "products, K slice 1", "T -> LDS", "G loads", "FIR GEMM + epilogue + stores"]
d = np.diff(t, axis=1)
print(f"workgroups {t.shape[0]}; cycles of thread 0 per phase (mean / p10 / p90), share of the workgroup's {d.sum(1).mean():.0f} cycles:")
for k, n in enumerate(names):
    print(f"  {n:42s} {d[:, k].mean():9.0f} {np.percentile(d[:, k], 10):9.0f} {np.percentile(d[:, k], 90):9.0f}   {100 * d[:, k].mean() / d.sum(1).mean():5.1f} %")
span = t[:, 7].max() - t[:, 0].min()
print(f"launch span (first start to last end, one clock domain per XCD: indicative) {span:.0f} cycles; workgroup starts: p50 {np.percentile(t[:, 0] - t[:, 0].min(), 50):.0f}, max {(t[:, 0] - t[:, 0].min()).max():.0f}")
