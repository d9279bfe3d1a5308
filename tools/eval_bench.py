#!/usr/bin/env python
"""Throughput of the 16-bit radiance-field block (evaluate_block_lp) in isolation: gfpp_head_eval_samples_lp on the occupied samples of one
512x512 frame in ray order (value-independent work, unlike the trips whose schedule depends on sigma).  Prints us per call and shader cycles
per 32-sample block and wavefront.  Used with experiment builds of the library (-DGFPP_ABLATE=n) to see where a block's time goes.
    python tools/eval_bench.py [precision] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import frame_case, build_model  # noqa: E402
from genefaceplusplus_amd.radnerfs import raymarching as rm, camera  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
case = frame_case("may_torso", 512)
model = build_model(case, dev, "fused")
model.precision = precision
hp = case["hp"]
N = 512 * 512
pose = torch.from_numpy(case["pose"]).to(dev)
rays = camera.get_rays(pose, case["intr"], 512, 512)
ro, rd = rays["rays_o"].view(-1, 3).contiguous(), rays["rays_d"].view(-1, 3).contiguous()
nears, fars = rm.near_far_from_aabb(ro, rd, model.aabb_infer, model.min_near)
alive = torch.arange(N, dtype=torch.int32, device=dev)
xyzs, dirs, deltas = rm.march_rays(N, 8, alive, nears.clone(), ro, rd, model.bound, model.density_bitfield, model.cascade, model.grid_size, nears, fars, -1, False,
                                   hp["dt_gamma"], hp["max_steps"])
keep = deltas[:, 0] > 0
P, D = xyzs[keep].contiguous(), dirs[keep].contiguous()
grid = int(os.environ.get("GFPP_EVAL_GRID", "0")) or torch.cuda.get_device_properties(0).multi_processor_count
waves = int(os.environ.get("GFPP_EVAL_WAVES", "8"))
if "GFPP_EVAL_GRID" in os.environ or "GFPP_EVAL_WAVES" in os.environ:
    keep_n = min(P.shape[0], grid * waves * 32 * 12)          # 12 blocks per active wavefront
    P, D = P[:keep_n].contiguous(), D[:keep_n].contiguous()
M = P.shape[0]
if os.environ.get("GFPP_EVAL_BANDED"):
    # L2-locality experiment: workgroup b runs on XCD b % 8 and takes the 256-sample chunks b, b + 256, ...; permute the ray-ordered stream so
    # that the chunks of XCD x all come from image band x (GFPP_EVAL_BANDED=8 -> 8 contiguous bands; =64 -> 64 strips interleaved over the XCDs)
    nb = int(os.environ["GFPP_EVAL_BANDED"])
    chunks = torch.arange(M, device=dev).split(256)
    n_ch = len(chunks)
    band_of = [(i * nb // n_ch) % 8 for i in range(n_ch)]                     # spatial strip -> XCD
    by_xcd = [[c for c, b in zip(chunks, band_of) if b == x] for x in range(8)]
    order = []
    k = 0
    while any(by_xcd):
        x = k % 8                                                              # chunk position k is processed by workgroup k % 256 -> XCD (k % 256) % 8 = k % 8
        if by_xcd[x]:
            order.append(by_xcd[x].pop(0))
        else:
            order.append(next(l for l in by_xcd if l).pop(0))
        k += 1
    perm = torch.cat(order)
    P, D = P[perm].contiguous(), D[perm].contiguous()
cf = torch.randn(64, device=dev) * 0.1
ind = model.individual_embeddings[0]
pipe = model.pipeline()
with torch.no_grad():
    for _ in range(3):
        out = pipe.eval_samples(P, D, cf, ind)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = pipe.eval_samples(P, D, cf, ind)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
blocks = (M + 31) // 32
cus = torch.cuda.get_device_properties(0).multi_processor_count
clock_ghz = float(os.environ.get("GFPP_CLOCK_GHZ", "2.4"))
per_wave = blocks / (min(grid, cus) * waves)
print(f"eval_bench {precision} grid {min(grid, cus)} waves {waves}: {M} samples, {ms * 1e3:.1f} us per call, {M / ms / 1e6:.1f} Msamples/ms, "
      f"{ms * 1e-3 * clock_ghz * 1e9 / per_wave:.0f} cycles per block per wavefront at {clock_ghz} GHz, "
      f"algorithmic {M * 2060 / ms / 1e9 * 1e3:.0f} GB/s, {M * 128768 / ms / 1e12 * 1e3:.1f} TFLOP/s; sigma checksum {float(out[0].double().sum()):.6e}")
