#!/usr/bin/env python
"""Training-side golden vectors from the REFERENCE's own Python (build container only; /root/reference is read-only and absent on the GPU box):

    python tests/golden/make_golden_train.py

Executed from the reference, unmodified: NeRFRenderer.render in training mode (modules/radnerfs/renderer.py:286-399, the `self.training`
branch :319-340), RADNeRF.forward (radnerf.py:108-141), mark_untrained_grid (renderer.py:131-199), the autograd shims
raymarching.py:_march_rays_train / _composite_rays_train and grid.py:_grid_encode (forward + backward).  Substituted exactly as in
make_golden.py: the four CUDA extensions -> oracle/ref_backends.py (radnerf_oracle.c on CPU tensors), Tensor.cuda() -> identity.
So this pins the reference's training-time control flow and autograd wiring around the kernels; the kernels stay the oracle's restatements.

Output: ref_python_train_golden.npz -- the forward results of one 24x24 training render and the gradients of a photometric loss.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import prepare_reference_imports, to_t, REPO  # noqa: E402

sys.path.insert(0, REPO)
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.configs import may_hparams, VARIANT_YAML  # noqa: E402

HW = 24


def main():
    prepare_reference_imports()
    from utils.commons.hparams import set_hparams
    from modules.radnerfs.radnerf import RADNeRF
    from modules.radnerfs import utils as ref_utils

    ref_hp = set_hparams(config=VARIANT_YAML["may_head"], exp_name="", print_hparams=False, global_hparams=True)
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    model = RADNeRF(ref_hp)
    model.load_state_dict(to_t(sd), strict=True)
    model.train()

    pose = torch.from_numpy(syn.synthetic_pose(0))[None]
    intr = syn.intrinsics_for(HW, HW)
    rays = ref_utils.get_rays(pose, intr, HW, HW, N=-1)
    fi = syn.synthetic_frame_inputs(hp, 0)
    cond = torch.from_numpy(fi["cond"])
    bg = torch.full((1, HW * HW, 3), 0.5)
    out = {}
    res = model.render(rays["rays_o"], rays["rays_d"], cond, ref_utils.get_bg_coords(HW, HW, "cpu"), ref_utils.convert_poses(pose), index=0,
                       dt_gamma=hp["dt_gamma"], bg_color=bg, perturb=False, force_all_rays=True, max_steps=hp["max_steps"],
                       eye_area_percent=torch.from_numpy(fi["eye_area_percent"]))
    for k in ("weights_sum", "ambient", "rgb_map", "depth_map"):
        out["fwd." + k] = res[k].detach().numpy().copy()
    out["fwd.step_counter"] = model.step_counter.numpy().copy()
    out["fwd.local_step"] = np.array([model.local_step])
    torch.manual_seed(0)
    target = torch.rand(1, HW * HW, 3)
    out["target"] = target.numpy().copy()
    loss = ((res["rgb_map"] - target) ** 2).mean() + 1e-3 * res["ambient"].mean() + 1e-2 * res["weights_sum"].mean()
    out["loss"] = np.array([float(loss)])
    loss.backward()
    named = dict(model.named_parameters())
    for name in ("ambient_net.net.0.weight", "ambient_net.net.2.weight", "sigma_net.net.0.weight", "sigma_net.net.2.weight", "color_net.net.0.weight",
                 "color_net.net.1.weight", "cond_prenet.encoder_fc1.2.weight", "cond_att_net.attentionNet.0.weight", "individual_embeddings"):
        g = named[name].grad
        out["grad." + name] = (g[:4].numpy().copy() if name == "individual_embeddings" else g.numpy().copy())
    for name in ("position_embedder.embeddings", "ambient_embedder.embeddings"):
        g = named[name].grad.numpy()
        out["gradsum." + name] = np.array([g.astype(np.float64).sum(), np.abs(g).astype(np.float64).sum()])
        nz = np.flatnonzero(np.abs(g).sum(axis=1))[:256]
        out["gradrows." + name] = nz.astype(np.int64)
        out["gradvals." + name] = g[nz].copy()

    # mark_untrained_grid: cells no camera sees get density -1 (renderer.py:131-199)
    model.density_grid.zero_()
    poses = np.stack([syn.synthetic_pose(i) for i in range(3)]).astype(np.float32)
    model.mark_untrained_grid(poses, intr)
    out["mark.poses"] = poses
    out["mark.untrained"] = np.packbits(model.density_grid.numpy() < 0)
    # update_extra_state: occupancy-grid refresh from the density field (renderer.py:201-284).  It draws random.randint once and
    # torch.rand_like once per cascade; with both generators seeded the whole refresh is reproducible on CPU.
    import random
    model.load_state_dict(to_t(sd), strict=True)               # back to the initial grid / bitfield
    model.mean_density = model.iter_density = 0
    rng = np.random.default_rng(5)
    conds = np.clip(rng.standard_normal((40, 1, 204)), -1.5, 1.5).astype(np.float32)
    model.conds = torch.from_numpy(conds)
    out["upd.conds"] = conds
    random.seed(3)
    torch.manual_seed(3)
    with torch.no_grad():
        model.update_extra_state(decay=0.95)
    grid = model.density_grid.numpy()
    out["upd.grid_sample"] = grid[0, ::997].copy()
    out["upd.grid_sum"] = np.array([grid.astype(np.float64).sum(), float((grid > 0).sum())])
    out["upd.bitfield"] = model.density_bitfield.numpy().copy()
    out["upd.scalars"] = np.array([model.mean_density, model.iter_density, model.mean_count, model.local_step], np.float64)
    # ------------------------------------------------------------------ torso model: training step (head frozen) + 2-D grid refresh
    from modules.radnerfs.radnerf_torso import RADNeRFTorso
    ref_hp_t = set_hparams(config=VARIANT_YAML["may_torso"], exp_name="", print_hparams=False, global_hparams=True)
    hp_t = may_hparams("may_torso")
    sd_t = syn.synthetic_state_dict(hp_t, "may_torso")
    torso = RADNeRFTorso(ref_hp_t)
    torso.load_state_dict(to_t(sd_t), strict=True)
    torso.train()
    fi_t = syn.synthetic_frame_inputs(hp_t, 0)
    random.seed(11)
    res = torso.render(rays["rays_o"], rays["rays_d"], torch.from_numpy(fi_t["cond"]), ref_utils.get_bg_coords(HW, HW, "cpu"), ref_utils.convert_poses(pose),
                       index=0, dt_gamma=hp_t["dt_gamma"], bg_color=bg, perturb=False, force_all_rays=True, max_steps=hp_t["max_steps"])
    for k in ("weights_sum", "ambient", "rgb_map", "depth_map", "torso_alpha_map", "torso_rgb_map"):
        out["torso.fwd." + k] = res[k].detach().numpy().copy()
    out["torso.fwd.deform"] = res["deform"].detach().numpy().copy()
    loss = ((res["rgb_map"] - target) ** 2).mean() + 1e-2 * res["torso_alpha_map"].mean() + 1e-3 * res["deform"].abs().mean()
    out["torso.loss"] = np.array([float(loss.detach())])
    loss.backward()
    named = dict(torso.named_parameters())
    for name in ("torso_deform_net.net.0.weight", "torso_deform_net.net.2.weight", "torso_canonicial_net.net.0.weight", "torso_canonicial_net.net.2.weight"):
        out["torso.grad." + name] = named[name].grad.numpy().copy()
    gte = named["torso_embedder.embeddings"].grad.numpy()
    out["torso.gradsum.torso_embedder.embeddings"] = np.array([gte.astype(np.float64).sum(), np.abs(gte).astype(np.float64).sum()])
    out["torso.grad.torso_individual_codes"] = named["torso_individual_codes"].grad[:4].numpy().copy()
    out["torso.head_has_grad"] = np.array([int(named["sigma_net.net.0.weight"].grad is not None)])
    poses_t = np.stack([syn.synthetic_pose(i) for i in range(6)]).astype(np.float32)
    torso.poses = torch.from_numpy(poses_t)
    out["torso.upd.poses"] = poses_t
    random.seed(4)
    torch.manual_seed(4)
    with torch.no_grad():
        torso.update_extra_state(decay=0.95)
    out["torso.upd.grid"] = torso.density_grid_torso.numpy().copy()
    out["torso.upd.mean"] = np.array([torso.mean_density_torso], np.float64)
    # ------------------------------------------------------------------ training-time ray sampling (utils.py:283-364 with N > 0 / patch / rect)
    H2 = W2 = 40
    intr2 = syn.intrinsics_for(H2, W2)
    for tag, kw in (("rand", dict(N=300)), ("patch", dict(N=4 * 64, patch_size=8)), ("rect", dict(rect=(5, 9, 10, 30))), ("clip", dict(N=10 ** 9))):
        torch.manual_seed(21)
        rr = ref_utils.get_rays(pose, intr2, H2, W2, **kw)
        out[f"rays.{tag}.inds"] = rr["inds"].numpy().astype(np.int64)
        out[f"rays.{tag}.rays_d"] = rr["rays_d"].numpy().copy()
        out[f"rays.{tag}.rays_o"] = rr["rays_o"].numpy().copy()
        out[f"rays.{tag}.ij"] = np.stack([rr["i"].numpy(), rr["j"].numpy()])
    np.savez_compressed(os.path.join(HERE, "ref_python_train_golden.npz"), **out)
    print({k: (v.shape, float(np.abs(v).sum())) for k, v in out.items() if k.startswith(("fwd", "loss"))})


if __name__ == "__main__":
    main()
