#!/usr/bin/env python
"""Micro-benchmark of the stand-alone grid-encode kernel (the "hash-grid stage" of the north star) on MI355X.
Inputs: (i) 2^22 uniform points (cache hostile), (ii) the marcher's real sample stream of a 512x512 synthetic frame
(coherent along rays / across neighbouring pixels), fp32 and fp16 tables.  Prints achieved algorithmic GB/s
(B x (4 D + L 2^D C s_tab + L C s_out)) against the 8 TB/s HBM peak."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams
from genefaceplusplus_amd.radnerfs import raymarching as rm, camera
from genefaceplusplus_amd.radnerfs.encoders import grid_encode_raw


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    dev = torch.device("cuda:0")
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    off3 = torch.from_numpy(sd["position_embedder.offsets"]).to(dev)
    emb3 = torch.from_numpy(sd["position_embedder.embeddings"]).to(dev)
    off2, pls = syn.grid_offsets(2)
    off2 = torch.from_numpy(off2).to(dev)
    emb2 = torch.rand(int(off2[-1]), 2, device=dev)
    _, pls3 = syn.grid_offsets(3)
    out = []
    # real sample stream: all occupied samples of trip 0..n of a 512^2 frame (n_step = 8 from every ray's near)
    HW = 512
    pose = torch.from_numpy(syn.synthetic_pose(0)).to(dev)[None]
    rays = camera.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
    ro, rd = rays["rays_o"][0].contiguous(), rays["rays_d"][0].contiguous()
    aabb = torch.from_numpy(sd["aabb_infer"]).to(dev)
    nears, fars = rm.near_far_from_aabb(ro, rd, aabb, 0.05)
    alive = torch.arange(HW * HW, dtype=torch.int32, device=dev)
    xyzs, _, deltas = rm.march_rays(HW * HW, 8, alive, nears.clone(), ro, rd, 1, torch.from_numpy(sd["density_bitfield"]).to(dev), 1, 128, nears, fars,
                                    -1, False, 1 / 256, 16)
    real = ((xyzs[deltas[:, 0] > 0] + 1) / 2).contiguous()
    cases = [("uniform", torch.rand(1 << 22, 3, device=dev), emb3, off3, pls3, 3),
             ("ray_stream", real, emb3, off3, pls3, 3),
             ("uniform_2d", torch.rand(1 << 22, 2, device=dev), emb2, off2, pls, 2)]
    for name, u, emb, off, p, D in cases:
        for dt, sz in ((torch.float32, 4), (torch.float16, 2)):
            e = emb.to(dt)
            B = u.shape[0]
            t = timeit(lambda: grid_encode_raw(u, e, off, p, 16, 1, False, 0))
            bytes_pt = 4 * D + 16 * (2 ** D) * 2 * sz + 16 * 2 * sz
            gbps = B * bytes_pt / t / 1e9
            out.append({"input": name, "tables": str(dt).replace("torch.", ""), "points": B, "ms": round(t * 1e3, 4), "bytes_per_point": bytes_pt,
                        "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4), "Gpoints_per_s": round(B / t / 1e9, 3)})
            print(json.dumps(out[-1]))
    return out


if __name__ == "__main__":
    main()
