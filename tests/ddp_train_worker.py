#!/usr/bin/env python
"""Test helper (child interpreters of tests/test_train_ddp_cpu.py; not shipped): the product's training-mode render() under
torch.nn.parallel.DistributedDataParallel on two gloo ranks, with the C-ABI calls redirected to the CPU oracle like tests/product_on_oracle.py.

Person-specific training in the reference is single-process or DDP (tasks/run.py -> utils/commons/trainer.py: one process per GPU, NCCL);
on MI355X the same wrapper runs over RCCL.  What this checks is that the drop-in classes survive the wrapper: custom autograd Functions
(grid encoder, ray marcher, compositor) fire DDP's gradient hooks, buffers (density grid / bitfield) broadcast, parameters that a step does
not touch are tolerated -- i.e. after backward both ranks hold the mean of the two ranks' single-process gradients.
Usage: ddp_train_worker.py <rank> <world> <port>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import product_on_oracle as poo  # noqa: E402
import oracle.oracle as orc  # noqa: E402


def step(model, hp, syn, rank, HW):
    pose = syn.synthetic_pose(3 * rank)[None]                     # every rank its own view
    r = orc.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
    fi = syn.synthetic_frame_inputs(hp, rank)
    res = model(torch.from_numpy(r["rays_o"]), torch.from_numpy(r["rays_d"]), torch.from_numpy(fi["cond"]),
                torch.from_numpy(orc.get_bg_coords(HW, HW)), torch.from_numpy(orc.convert_poses(pose)), index=0, dt_gamma=hp["dt_gamma"],
                bg_color=torch.full((1, HW * HW, 3), 0.5), perturb=False, force_all_rays=True, max_steps=hp["max_steps"],
                eye_area_percent=torch.from_numpy(fi["eye_area_percent"]))
    target = torch.full_like(res["rgb_map"], 0.25)
    return ((res["rgb_map"] - target) ** 2).mean() + 1e-3 * res["ambient"].mean() + 1e-2 * res["weights_sum"].mean()


class RenderModule(torch.nn.Module):
    """DDP drives forward(); the trainer's call is model.render(...) (tasks/radnerfs/radnerf.py:run_model)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, *a, **k):
        return self.net.render(*a, **k)


def main(rank, world, port):
    from genefaceplusplus_amd import synthetic as syn, radnerfs
    from genefaceplusplus_amd.configs import may_hparams
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    is_cuda = torch.Tensor.is_cuda
    HW = 12
    hp = may_hparams("may_head")
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in syn.synthetic_state_dict(hp, "may_head").items()}

    def fresh():
        m = radnerfs.RADNeRF(hp)
        m.load_state_dict(sd, strict=True)
        return m.train()

    # ---- single-process gradients of this rank's batch, then their mean over the ranks (what DDP must produce) -----------------
    poo.patch()
    alone = RenderModule(fresh())
    step(alone, hp, syn, rank, HW).backward()
    torch.Tensor.is_cuda = is_cuda                               # (the harness's CPU exception must not leak into DDP's own device checks)
    want = {}
    for name, p in alone.named_parameters():
        g = torch.zeros_like(p) if p.grad is None else p.grad.clone()
        dist.all_reduce(g)
        want[name] = g / world
    # ---- the same step under DDP -------------------------------------------------------------------------------------------------
    ddp = torch.nn.parallel.DistributedDataParallel(RenderModule(fresh()), find_unused_parameters=True)
    poo.patch()
    loss = step(ddp, hp, syn, rank, HW)
    loss.backward()
    torch.Tensor.is_cuda = is_cuda
    worst, touched = 0.0, 0
    for name, p in ddp.module.named_parameters():
        ref = want[name]
        got = torch.zeros_like(p) if p.grad is None else p.grad
        denom = float(ref.norm())
        if denom > 0:
            touched += 1
            worst = max(worst, float((got - ref).norm()) / denom)
        else:
            worst = max(worst, float(got.abs().max()))
    # both ranks must hold the same gradients afterwards
    flat = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in ddp.module.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    sys.stdout.write("\nDDPRESULT {\"rank\": %d, \"worst_rel\": %.3e, \"touched\": %d, \"ranks_agree\": %s, \"loss\": %.6f}\n"
                     % (rank, worst, touched, "true" if same else "false", float(loss.detach())))
    sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
