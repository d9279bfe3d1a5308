#!/usr/bin/env python
"""Phase timeline of the 16-bit torso kernel (k_torso_lp): a -DGFPP_TORSO_PROF=1 build (tools/build_variant.sh torsoprof frame_torso_lp.hip -DGFPP_TORSO_PROF=1, selected with
GFPP_LIB_PATH) stamps the 100 MHz wall clock at every workgroup's phase boundaries; per-frame render() calls, the last launch's stamps are printed: spread of the workgroups'
starts, and per phase the median / p90 over the workgroups that had torso pixels.   GFPP_LIB_PATH=... python tools/torso_phase.py [variant] [hw] [precision]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
dev = torch.device("cuda:0")
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["GFPP_TORSO_PROF_PTR"] = hex(buf.data_ptr())
from helpers import frame_case, build_model, product_render  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "may_torso"
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16"
case = frame_case(variant, HW)
model = build_model(case, dev, "fused")
model.precision = precision
model.use_graph = False
if hasattr(model, "sr_net"):
    model.sr_net.ready = False
for rep in range(6):
    buf.zero_()
    product_render(model, case, dev, "product")
    torch.cuda.synchronize()
    if rep < 3:
        continue
    t = buf.cpu().numpy().reshape(4096, 8).astype(np.float64) / 100.0      # microseconds
    n_wg = (HW * HW + 255) // 256
    w = t[:n_wg]
    s0 = w[:, 0].min()
    busy = w[:, 3] > 0
    q = lambda v: f"{np.median(v):6.2f} / {np.percentile(v, 90):6.2f}"
    print(f"--- frame {rep}: {n_wg} workgroups, {int(busy.sum())} with torso pixels; starts spread {w[:, 0].max() - s0:.2f} us; last end {w[:, 5].max() - s0:.2f} us")
    print(f"    all: occupancy test + vote {q(w[:, 1] - w[:, 0])} | ray records {q(w[:, 2] - w[:, 1])} | whole {q(w[:, 5] - w[:, 0])}")
    b = w[busy]
    print(f"    with torso pixels: weights + fold {q(b[:, 3] - b[:, 2])} | MLP passes {q(b[:, 4] - b[:, 3])} | compositing + stores {q(b[:, 5] - b[:, 4])} | whole {q(b[:, 5] - b[:, 0])}")
    e = np.sort(w[:, 5] - s0)
    print("    workgroups finished by time: " + ", ".join(f"{int(p * 100)}% {e[int(p * (len(e) - 1))]:.1f}" for p in (0.1, 0.5, 0.9, 1.0)))
