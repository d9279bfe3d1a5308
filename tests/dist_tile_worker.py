"""Worker of tests/test_dist_gpu.py::test_ray_tile_sharding_*: every rank renders ONE frame together with the others (frames.render_frame_tiled, one
int32 all_reduce per trip for the frame-wide alive count) and compares it with the same frame rendered alone -- bit for bit.  Backend from argv:
gloo (ranks may share one GPU) or nccl (one GPU per rank)."""
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
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev_id = local % torch.cuda.device_count()
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from genefaceplusplus_amd import frames
    from genefaceplusplus_amd.radnerfs import camera
    from helpers import frame_case, build_model
    res = {"rank": rank, "world": world}
    import random
    for variant, HW, precision, head_aware in (("may_head", 48, "fp32", False), ("may_torso", 37, "fp16", False), ("may_torso", 64, "bf16", False),
                                               ("may_torso", 40, "fp16", True)):
        case = frame_case(variant, HW, hp_over={"torso_head_aware": True} if head_aware else None)
        model = build_model(case, dev, "fused")
        model.precision = precision
        model.use_graph = False
        pose = torch.from_numpy(case["pose"]).to(dev)
        r = camera.get_rays(pose, case["intr"], HW, HW)
        args = (r["rays_o"], r["rays_d"], torch.from_numpy(case["cond"]).to(dev), camera.get_bg_coords(HW, HW, "cpu").to(dev), camera.convert_poses(pose))
        kw = dict(index=0, perturb=False, T_thresh=0.01, max_steps=16, dt_gamma=case["hp"]["dt_gamma"])
        bg = torch.from_numpy(case["bg_color"]).to(dev)
        with torch.no_grad():
            alone = model.render(*args, bg_color=bg, **kw)
            alone = {k: v.clone() for k, v in alone.items() if torch.is_tensor(v)}
            alive_alone = model.pipeline().trip_counters(HW * HW)[0].copy()
        if head_aware:
            # the head-aware torso draws a coin per frame (radnerf_torso.py:177): the ranks' Python RNGs are deliberately out of step here, the tiled
            # renderer must make ONE draw for the frame (group rank 0's) -- compare with the frame rendered alone under BOTH outcomes
            both = {}
            for coin in (False, True):
                with torch.no_grad():
                    a = model.render(*args, bg_color=bg, use_head_for_torso=coin, **kw)
                both[coin] = {k: v.clone() for k, v in a.items() if torch.is_tensor(v)}
            res["head_aware_branches_differ"] = not torch.equal(both[False]["rgb_map"], both[True]["rgb_map"])
            random.seed(1000 + 7 * rank)
            tiled = frames.render_frame_tiled(model, *args, bg_color=bg, **kw)
            match = [c for c in (False, True) if torch.equal(tiled["rgb_map"].reshape(-1, 3), both[c]["rgb_map"].reshape(-1, 3))
                     and torch.equal(tiled["torso_alpha_map"].reshape(-1), both[c]["torso_alpha_map"].reshape(-1))]
            res[f"{variant}_{HW}_headaware_{precision}"] = len(match) == 1 and res["head_aware_branches_differ"]
            continue
        tiled = frames.render_frame_tiled(model, *args, bg_color=bg, **kw)
        ok = bool(torch.equal(tiled["rgb_map"].reshape(-1, 3), alone["rgb_map"].reshape(-1, 3)) and torch.equal(tiled["depth_map"].reshape(-1), alone["depth_map"].reshape(-1)))
        if "torso_alpha_map" in alone:
            ok = ok and bool(torch.equal(tiled["torso_alpha_map"].reshape(-1), alone["torso_alpha_map"].reshape(-1)))
        # the frame-wide alive counts the tiles agreed on == the single-GPU loop's
        lo, hi = frames.ray_tile(HW * HW, rank, world)
        pipe = model.pipeline()
        if pipe.lp_kernel == "persist" and (precision != "fp32" or pipe.fp32_kernel == "wave"):
            # one launch per tile + ONE all_reduce of the end-point histogram: the resolve step leaves the frame-wide alive counts in the tile's counters
            g = pipe.workspace(hi - lo)[1]["counters"].cpu().numpy()
            hist = pipe.workspace(hi - lo)[1]["gcounters"].cpu().numpy()
            ok = ok and int(hist[:32].sum()) == HW * HW
        else:
            g = pipe.workspace(hi - lo)[1]["gcounters"].cpu().numpy()
        res[f"{variant}_{HW}_{precision}"] = ok and bool(np.array_equal(g[:16], alive_alone[:16]))
        res[f"{variant}_{HW}_{precision}_trips"] = int((alive_alone[:16] > 0).sum())
    sys.stdout.write("\nTILERESULT " + json.dumps(res) + "\n"); sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
