#!/usr/bin/env python
"""Phase timeline of the super-resolution kernels: a -DGFPP_SR_PROF=1 build (tools/build_variant.sh srprof superres.hip -DGFPP_SR_PROF=1, selected with
GFPP_LIB_PATH) stamps the 100 MHz wall clock at every workgroup's phase boundaries; this prints, per layer, when workgroups start / end relative to the
launch's first start and how long each phase lasts (median / p90 over workgroups, microseconds).   GFPP_LIB_PATH=... python tools/sr_phase.py [noise mode]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
buf = torch.zeros(3 * 2048 * 16, dtype=torch.int64, device=dev)
os.environ["GFPP_SR_PROF_PTR"] = hex(buf.data_ptr())
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "random"
sd = syn.synthetic_sr_state(prefix="")
net = Superresolution(channels=3)
net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
net = net.to(dev).eval()
x = torch.rand(1, 3, 256, 256, device=dev)
names = ["block0.conv0+conv1+torgb (256 wg)", "block1.conv0 up (512 wg)", "block1.conv1+torgb (1024 wg)"]
counts = [256, 512, 1024]
with torch.no_grad():
    for _ in range(5):
        net(x, noise_mode=mode)
    torch.cuda.synchronize()
    for rep in range(3):
        buf.zero_()
        net(x, noise_mode=mode)
        torch.cuda.synchronize()
        raw = buf.cpu().numpy().reshape(3, 2048, 16).astype(np.float64)
        t = raw / 100.0     # microseconds (slots 0-7; slots 8-10 are shader-clock cycle sums of the tap loop's sections)
        print(f"--- forward {rep}")
        frame0 = t[0, :256, 0].min()
        for l in range(3):
            w = t[l, :counts[l]]
            w = w[w[:, 0] > 0]            # (the resident-weights last layer launches one workgroup per CU; its marks 1-3 are those of a workgroup's LAST patch)
            s0 = w[:, 0].min()
            q = lambda v: f"{np.median(v):6.2f} / {np.percentile(v, 90):6.2f}"
            print(f"{names[l]}: first start at {s0 - frame0:7.2f} us of the forward; starts spread {w[:, 0].max() - s0:6.2f}; last end {w[:, 4].max() - s0:6.2f}")
            print(f"    prologue {q(w[:, 1] - w[:, 0])} | tap loop {q(w[:, 2] - w[:, 1])} | epilogue math {q(w[:, 3] - w[:, 2])} | row stores {q(w[:, 4] - w[:, 3])} | whole {q(w[:, 4] - w[:, 0])}")
            if (w[:, 5] > 0).all() and (w[:, 6] > 0).all():
                print(f"    of the tap loop: second K slice's patch {q(w[:, 6] - w[:, 5])}; first half taps {q(w[:, 5] - w[:, 1])}; second half taps {q(w[:, 2] - w[:, 6])}")
            cyc = raw[l, :counts[l]][raw[l, :counts[l], 0] > 0][:, 8:12]
            if (cyc > 0).any():
                print(f"    tap loop sections of wavefront 0 (k-cycles, median): operand reads + MFMAs {np.median(cyc[:, 0]) / 1e3:.1f} | wait for the next chunk {np.median(cyc[:, 1]) / 1e3:.1f} | barrier {np.median(cyc[:, 2]) / 1e3:.1f} | staging the chunk after (+ slice reload) {np.median(cyc[:, 3]) / 1e3:.1f}")
            # how many workgroups run at once, sampled
            ends = np.sort(w[:, 4] - s0)
            print("    workgroups finished by time: " + ", ".join(f"{int(p * 100)}% {ends[int(p * (len(ends) - 1))]:.1f}" for p in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)))
