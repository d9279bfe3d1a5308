"""Worker of tests/test_dist_gpu.py (launched by torch.distributed.run, one rank per GPU, RCCL): renders a short clip frame-parallel, gathers it
to the writer rank AND to every rank, and compares both with the same clip rendered by this rank alone -- byte for byte."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from genefaceplusplus_amd import synthetic as syn, frames
    from genefaceplusplus_amd.clip import ClipRenderer, render_clip_distributed
    from helpers import frame_case, build_model
    HW, F = 64, 11                                              # ragged: 11 frames over `world` ranks
    case = frame_case("may_torso", HW)
    model = build_model(case, dev, "fused")
    model.precision = "fp16"
    fi = [syn.synthetic_frame_inputs(case["hp"], i) for i in range(F)]
    batch = {"ngp_poses": np.stack([syn.synthetic_pose(i) for i in range(F)]).astype(np.float32), "cond_wins": np.stack([f["cond"] for f in fi]),
             "lm68": np.stack([f["lm68"] for f in fi]), "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi])}
    cr = ClipRenderer(model, HW, HW, syn.intrinsics_for(HW, HW), bg_img=torch.full((1, HW * HW, 3), 0.5, device=dev), T_thresh=0.01, use_graph=True)
    clip = cr.prepare(batch, dev)
    alone = cr.render_to_device(clip, range(F)).clone()
    res = {"rank": rank, "world_seen": dist.get_world_size(), "backend": dist.get_backend()}
    for inter in (False, True):
        every = render_clip_distributed(cr, clip, F, interleaved=inter)
        writer = render_clip_distributed(cr, clip, F, interleaved=inter, dst=0)
        res[f"all_equal_{inter}"] = bool(torch.equal(every, alone))
        res[f"writer_equal_{inter}"] = bool(torch.equal(writer, alone)) if rank == 0 else (writer is None)
    # every rank rendered the same bytes for the same frame (weights replicated, kernels deterministic)
    digest = torch.tensor([float(alone.float().sum())], device=dev, dtype=torch.float64)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    res["replicas_agree"] = bool(lo.item() == hi.item())
    sys.stdout.write("\nDISTRESULT " + json.dumps(res) + "\n"); sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
