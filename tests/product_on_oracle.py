#!/usr/bin/env python
"""Test helper (run in a child interpreter by tests/test_oracle_train_cpu.py; not shipped): run the PRODUCT's Python training path on CPU tensors with its C-ABI calls redirected to the CPU oracle,
and compare one training step with tests/golden/ref_python_train_golden.npz.  Separates wiring differences (visible here) from kernel /
numerics differences (only visible on the GPU).  Usage: python tests/product_on_oracle.py"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle.oracle as orc  # noqa: E402

F, I, U = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
fp = lambda p: ctypes.cast(p, F)
ip = lambda p: ctypes.cast(p, I)
up = lambda p: ctypes.cast(p, U)
u32 = lambda v: ctypes.c_uint32(int(v))
cf = lambda v: ctypes.c_float(float(v))


def dispatch(name, *a):
    L = orc.lib()
    if name == "gfpp_near_far_from_aabb":
        L.orc_near_far_from_aabb(fp(a[0]), fp(a[1]), fp(a[2]), u32(a[3]), cf(a[4]), fp(a[5]), fp(a[6]))
    elif name == "gfpp_march_rays_train":
        L.orc_march_rays_train(fp(a[0]), fp(a[1]), up(a[2]), cf(a[3]), cf(a[4]), u32(a[5]), u32(a[6]), u32(a[7]), u32(a[8]), u32(a[9]), fp(a[10]), fp(a[11]),
                               fp(a[12]), fp(a[13]), fp(a[14]), ip(a[15]), ip(a[16]), fp(a[17]))
    elif name == "gfpp_composite_rays_train_forward":
        L.orc_composite_rays_train_forward(fp(a[0]), fp(a[1]), fp(a[2]), fp(a[3]), ip(a[4]), u32(a[5]), u32(a[6]), cf(a[7]), fp(a[8]), fp(a[9]), fp(a[10]), fp(a[11]))
    elif name == "gfpp_composite_rays_train_backward":
        L.orc_composite_rays_train_backward(fp(a[0]), fp(a[1]), fp(a[2]), fp(a[3]), fp(a[4]), fp(a[5]), fp(a[6]), ip(a[7]), fp(a[8]), fp(a[9]), fp(a[10]),
                                            u32(a[11]), u32(a[12]), cf(a[13]), fp(a[14]), fp(a[15]), fp(a[16]))
    elif name == "gfpp_grid_encode_forward":
        inputs, emb, offsets, out, B, D, C, Lv, S, H, dy_dx, gridtype, ac, interp, dtype = a[:15]
        assert dtype == 0
        if dy_dx:
            assert L.orc_grid_encode_dydx(fp(inputs), fp(emb), ip(offsets), fp(dy_dx), u32(B), u32(D), u32(C), u32(Lv), cf(S), u32(H), u32(gridtype),
                                          ctypes.c_int(ac), u32(interp)) == 0
        assert L.orc_grid_encode_forward(fp(inputs), fp(emb), ip(offsets), fp(out), u32(B), u32(D), u32(C), u32(Lv), cf(S), u32(H), u32(gridtype),
                                         ctypes.c_int(ac), u32(interp)) == 0
    elif name == "gfpp_grid_encode_backward":
        grad, inputs, emb, offsets, grad_emb, B, D, C, Lv, S, H, dy_dx, grad_inputs, gridtype, ac, interp = a[:16]
        assert L.orc_grid_encode_backward(fp(grad), fp(inputs), ip(offsets), fp(grad_emb), u32(B), u32(D), u32(C), u32(Lv), cf(S), u32(H), u32(gridtype),
                                          ctypes.c_int(ac), u32(interp)) == 0
        if dy_dx:
            L.orc_grid_input_backward(fp(grad), fp(dy_dx), fp(grad_inputs), u32(B), u32(D), u32(C), u32(Lv))
    elif name == "gfpp_grid_encode_backward_xcd":      # same gradient; the XCD-private scratch copies are a device-side detail
        grad, inputs, offsets, grad_emb, rows, copies, B, D, C, Lv, S, H, dy_dx, grad_inputs, gridtype, ac, interp = a[:17]
        assert L.orc_grid_encode_backward(fp(grad), fp(inputs), ip(offsets), fp(grad_emb), u32(B), u32(D), u32(C), u32(Lv), cf(S), u32(H), u32(gridtype),
                                          ctypes.c_int(ac), u32(interp)) == 0
        if dy_dx:
            L.orc_grid_input_backward(fp(grad), fp(dy_dx), fp(grad_inputs), u32(B), u32(D), u32(C), u32(Lv))
    elif name == "gfpp_grid_encode_input_backward":    # the device recomputes the derivative from the table; the oracle takes its two reference steps
        grad, grad_dtype, inputs, emb, offsets, grad_inputs, B, D, C, Lv, S, H, gridtype, ac, interp = a[:15]
        assert grad_dtype == 0
        dy_dx = np.zeros(int(B) * int(Lv) * int(D) * int(C), np.float32)
        assert L.orc_grid_encode_dydx(fp(inputs), fp(emb), ip(offsets), dy_dx.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), u32(B), u32(D), u32(C), u32(Lv), cf(S),
                                      u32(H), u32(gridtype), ctypes.c_int(ac), u32(interp)) == 0
        L.orc_grid_input_backward(fp(grad), dy_dx.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), fp(grad_inputs), u32(B), u32(D), u32(C), u32(Lv))
    elif name == "gfpp_sh_encode_forward":
        assert not a[5]
        assert L.orc_sh_encode_forward(fp(a[0]), fp(a[1]), u32(a[2]), u32(a[4])) == 0
    elif name == "gfpp_morton3D":
        L.orc_morton3D_batch(ip(a[0]), u32(a[1]), ip(a[2]))
    elif name == "gfpp_morton3D_dilation":
        L.orc_morton3D_dilation(fp(a[0]), u32(a[1]), u32(a[2]), fp(a[3]))
    elif name == "gfpp_packbits":
        L.orc_packbits(fp(a[0]), u32(a[1]), cf(a[2]), up(a[3]))
    elif name == "gfpp_march_rays":
        L.orc_march_rays(u32(a[0]), u32(a[1]), ip(a[2]), fp(a[3]), fp(a[4]), fp(a[5]), cf(a[6]), cf(a[7]), u32(a[8]), u32(a[9]), u32(a[10]), up(a[11]),
                         fp(a[12]), fp(a[13]), fp(a[14]), fp(a[15]), fp(a[16]), fp(a[17]))
    elif name == "gfpp_composite_rays":
        L.orc_composite_rays(u32(a[0]), u32(a[1]), cf(a[2]), ip(a[3]), fp(a[4]), fp(a[5]), fp(a[6]), fp(a[7]), fp(a[8]), fp(a[9]), fp(a[10]))
    elif name == "gfpp_freq_encode_forward":
        L.orc_freq_encode_forward(fp(a[0]), u32(a[1]), u32(a[2]), u32(a[3]), u32(a[4]), fp(a[5]))
    elif name in ("gfpp_get_rays", "gfpp_get_rays_at"):
        # the oracle restates ray generation in numpy (oracle.get_rays, utils.py:352-363); pick the listed pixels
        pose = np.ctypeslib.as_array(fp(a[0]), shape=(16,)).reshape(4, 4).copy()
        H, W = int(a[5]), int(a[6])
        full = orc.get_rays(pose[None], (a[1], a[2], a[3], a[4]), H, W)
        if name == "gfpp_get_rays":
            sel, ro_p, rd_p = np.arange(H * W), a[7], a[8]
        else:
            n = int(a[8])
            sel = np.ctypeslib.as_array(ctypes.cast(a[7], ctypes.POINTER(ctypes.c_int64)), shape=(n,)).copy()
            ro_p, rd_p = a[9], a[10]
        np.ctypeslib.as_array(fp(ro_p), shape=(len(sel), 3))[:] = full["rays_o"][0][sel]
        np.ctypeslib.as_array(fp(rd_p), shape=(len(sel), 3))[:] = full["rays_d"][0][sel]
    else:
        raise NotImplementedError(name)
    return 0


def patch():
    from genefaceplusplus_amd.radnerfs import raymarching, encoders, camera
    for mod in (raymarching, encoders, camera):
        mod.call = dispatch
    raymarching._stream = encoders._stream = lambda: None
    import types
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=None)
    torch.Tensor.is_cuda = property(lambda self: True)          # the product refuses CPU tensors; this harness is the one exception


def infer():
    """The product's reference-shaped ('staged') executor, eval mode, against tests/golden/ref_python_golden.npz (the reference's own
    inference render over the oracle kernels): head, torso and torso-SR frames."""
    import random
    from genefaceplusplus_amd import synthetic as syn, radnerfs
    from genefaceplusplus_amd.configs import may_hparams
    patch()
    g = np.load(os.path.join(HERE, "golden", "ref_python_golden.npz"))
    worst = 0.0
    classes = {"may_head": "RADNeRF", "may_torso": "RADNeRFTorso", "may_torso_sr": "RADNeRFTorsowithSR"}
    with torch.no_grad():
        for variant, cls in classes.items():
            hp = may_hparams(variant)
            sd = dict(syn.synthetic_state_dict(hp, variant))
            model = getattr(radnerfs, cls)(hp)
            if hasattr(model, "sr_net"):
                sd.update(syn.synthetic_sr_state())
                model.sr_net.ready = False          # the golden run used a stand-in for the SR net and recorded the pre-SR image only
            model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
            model.eval()
            model.executor = "staged"
            model.return_deform = True
            fi = syn.synthetic_frame_inputs(hp, 0)
            # conditioning nets (plain torch in both code bases) and the per-sample API
            cf = model.cal_cond_feat(torch.from_numpy(fi["cond"]), eye_area_percent=torch.from_numpy(fi["eye_area_percent"]))
            worst = max(worst, float(np.abs(cf.numpy() - g[f"{variant}.cond_feat"]).max()))
            if variant == "may_head":
                P, Dn = torch.from_numpy(g["fwd.position"]), torch.from_numpy(g["fwd.direction"])
                sigma, color, amb = model(P, Dn, cf, model.individual_embeddings[0])
                dens = model.density(P, cf)
                api = {"sigma": float(np.abs(sigma.numpy() - g["fwd.sigma"]).max()), "color": float(np.abs(color.numpy() - g["fwd.color"]).max()),
                       "ambient": float(np.abs(amb.numpy() - g["fwd.ambient"]).max()),
                       "density": float(np.abs(dens["sigma"].numpy() - g["fwd.density_sigma"]).max()),
                       "geo_feat": float(np.abs(dens["geo_feat"].numpy().sum(axis=1) - g["fwd.geo_feat_sum"]).max())}
                print("forward()/density()", api)
                worst = max(worst, *api.values())
            HW = 256 if variant == "may_torso_sr" else 64
            pose = syn.synthetic_pose(0)[None]
            r = orc.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
            random.seed(0)
            res = model.render(torch.from_numpy(r["rays_o"]), torch.from_numpy(r["rays_d"]), torch.from_numpy(fi["cond"]),
                               torch.from_numpy(orc.get_bg_coords(HW, HW)), torch.from_numpy(orc.convert_poses(pose)), index=0, staged=False,
                               bg_color=torch.full((1, HW * HW, 3), 0.5), lm68=torch.from_numpy(fi["lm68"]), perturb=False, force_all_rays=False,
                               T_thresh=0.01, eye_area_percent=torch.from_numpy(fi["eye_area_percent"]), **hp)
            rgb = res["rgb_map"].numpy()
            if variant == "may_torso_sr":
                rgb = np.transpose(rgb, (0, 2, 3, 1)).reshape(1, HW * HW, 3)
            sel = g[f"{variant}.render.sel"]
            errs = {"rgb": float(np.abs(rgb[:, sel] - g[f"{variant}.render.rgb"]).max())}
            dref = g[f"{variant}.render.depth"]
            ok = np.isfinite(dref)
            errs["depth"] = float(np.abs(res["depth_map"].numpy().reshape(1, -1)[:, sel][ok] - dref[ok]).max())
            if "torso_alpha_map" in res:
                errs["torso_alpha"] = float(np.abs(res["torso_alpha_map"].numpy().reshape(-1)[sel] - g[f"{variant}.render.torso_alpha"]).max())
                errs["deform_abs_sum"] = abs(float(np.abs(res["deform"].numpy()).astype(np.float64).sum()) - float(g[f"{variant}.render.deform_abs_sum"][0]))
            print(variant, errs)
            worst = max(worst, *errs.values())
    print("worst", worst)
    return worst


def main():
    from genefaceplusplus_amd import synthetic as syn, radnerfs
    from genefaceplusplus_amd.configs import may_hparams
    from genefaceplusplus_amd.radnerfs import camera
    patch()
    g = np.load(os.path.join(HERE, "golden", "ref_python_train_golden.npz"))
    HW = 24
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    model = radnerfs.RADNeRF(hp)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    model.train()
    pose = syn.synthetic_pose(0)[None]
    r = orc.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
    fi = syn.synthetic_frame_inputs(hp, 0)
    res = model.render(torch.from_numpy(r["rays_o"]), torch.from_numpy(r["rays_d"]), torch.from_numpy(fi["cond"]),
                       torch.from_numpy(orc.get_bg_coords(HW, HW)), torch.from_numpy(orc.convert_poses(pose)), index=0, dt_gamma=hp["dt_gamma"],
                       bg_color=torch.full((1, HW * HW, 3), 0.5), perturb=False, force_all_rays=True, max_steps=hp["max_steps"],
                       eye_area_percent=torch.from_numpy(fi["eye_area_percent"]))
    worst = 0.0
    for k in ("weights_sum", "ambient", "rgb_map"):
        err = float(np.abs(res[k].detach().numpy() - g["fwd." + k]).max())
        worst = max(worst, err)
        print("fwd", k, err)
    target = torch.from_numpy(g["target"])
    loss = ((res["rgb_map"] - target) ** 2).mean() + 1e-3 * res["ambient"].mean() + 1e-2 * res["weights_sum"].mean()
    print("loss", float(loss.detach()), float(g["loss"][0]))
    loss.backward()
    named = dict(model.named_parameters())
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
    for key in g.files:
        if key.startswith("grad."):
            name = key[5:]
            got = named[name].grad.detach().numpy()
            got = got[:4] if name == "individual_embeddings" else got
            worst = max(worst, rel(got, g[key]))
            print(f"{name:45s} rel err {rel(got, g[key]):.3e}   |ref| {np.linalg.norm(g[key]):.3e}")
        elif key.startswith("gradsum."):
            name = key[8:]
            got = named[name].grad.detach().numpy()
            worst = max(worst, rel(got[g['gradrows.' + name]], g['gradvals.' + name]))
            print(f"{name:45s} abs-sum {np.abs(got).astype(np.float64).sum():.6e} vs {g[key][1]:.6e}   rows rel err {rel(got[g['gradrows.' + name]], g['gradvals.' + name]):.3e}")
    # occupancy-grid upkeep, same call sequence as tests/golden/make_golden_train.py
    import random
    model.density_grid.zero_()
    model.mark_untrained_grid(g["mark.poses"], syn.intrinsics_for(HW, HW))
    same = np.array_equal(np.packbits(model.density_grid.numpy() < 0), g["mark.untrained"])
    print("mark_untrained_grid identical", same)
    worst = max(worst, 0.0 if same else 1.0)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    model.mean_density = model.iter_density = 0
    model.conds = torch.from_numpy(g["upd.conds"])
    random.seed(3)
    torch.manual_seed(3)
    with torch.no_grad():
        model.update_extra_state(decay=0.95)
    grid = model.density_grid.numpy()
    errs = {"grid_sample": float(np.abs(grid[0, ::997] - g["upd.grid_sample"]).max()),
            "grid_sum": abs(float(grid.astype(np.float64).sum()) - g["upd.grid_sum"][0]), "occupied": abs(float((grid > 0).sum()) - g["upd.grid_sum"][1]),
            "bitfield": float((model.density_bitfield.numpy() != g["upd.bitfield"]).sum()),
            "scalars": float(np.abs(np.array([model.mean_density, model.iter_density, model.mean_count, model.local_step], np.float64) - g["upd.scalars"]).max())}
    print("update_extra_state", errs)
    worst = max(worst, *errs.values())
    # ---- torso model: one training step with the head frozen, then the 2-D grid refresh (radnerf_torso.py:86-244)
    hp_t = may_hparams("may_torso")
    sd_t = syn.synthetic_state_dict(hp_t, "may_torso")
    torso = radnerfs.RADNeRFTorso(hp_t)
    torso.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_t.items()}, strict=True)
    torso.train()
    torso.return_deform = True
    fi_t = syn.synthetic_frame_inputs(hp_t, 0)
    random.seed(11)
    res = torso.render(torch.from_numpy(r["rays_o"]), torch.from_numpy(r["rays_d"]), torch.from_numpy(fi_t["cond"]), torch.from_numpy(orc.get_bg_coords(HW, HW)),
                       torch.from_numpy(orc.convert_poses(pose)), index=0, dt_gamma=hp_t["dt_gamma"], bg_color=torch.full((1, HW * HW, 3), 0.5), perturb=False,
                       force_all_rays=True, max_steps=hp_t["max_steps"])
    errs = {}
    for k in ("weights_sum", "ambient", "rgb_map", "depth_map", "torso_alpha_map", "torso_rgb_map", "deform"):
        a, b = res[k].detach().numpy(), g["torso.fwd." + k]
        ok = np.isfinite(b)
        errs[k] = float(np.abs(a.reshape(b.shape)[ok] - b[ok]).max())
    loss = ((res["rgb_map"] - target) ** 2).mean() + 1e-2 * res["torso_alpha_map"].mean() + 1e-3 * res["deform"].abs().mean()
    errs["loss"] = abs(float(loss.detach()) - float(g["torso.loss"][0]))
    loss.backward()
    named = dict(torso.named_parameters())
    for key in g.files:
        if key.startswith("torso.grad."):
            name = key[len("torso.grad."):]
            got = named[name].grad.detach().numpy()
            got = got[:4] if name == "torso_individual_codes" else got
            errs["grad " + name] = rel(got, g[key])
    gte = named["torso_embedder.embeddings"].grad.numpy()
    errs["grad torso_embedder abs-sum"] = abs(float(np.abs(gte).astype(np.float64).sum()) - g["torso.gradsum.torso_embedder.embeddings"][1])
    errs["head frozen"] = float(int(named["sigma_net.net.0.weight"].grad is not None) != int(g["torso.head_has_grad"][0]))
    torso.poses = torch.from_numpy(g["torso.upd.poses"])
    random.seed(4)
    torch.manual_seed(4)
    with torch.no_grad():
        torso.update_extra_state(decay=0.95)
    errs["torso grid"] = float(np.abs(torso.density_grid_torso.numpy() - g["torso.upd.grid"]).max())
    errs["torso mean density"] = abs(torso.mean_density_torso - float(g["torso.upd.mean"][0]))
    print("torso", errs)
    worst = max(worst, *errs.values())
    # ---- training-time ray sampling: same torch.randint call sequence as the reference, so a seeded generator gives the same pixels
    H2 = W2 = 40
    pose_t = torch.from_numpy(pose)
    errs = {}
    for tag, kw in (("rand", dict(N=300)), ("patch", dict(N=4 * 64, patch_size=8)), ("rect", dict(rect=(5, 9, 10, 30))), ("clip", dict(N=10 ** 9))):
        torch.manual_seed(21)
        rr = camera.get_rays(pose_t, syn.intrinsics_for(H2, W2), H2, W2, **kw)
        errs[tag + " inds"] = float((rr["inds"].numpy().astype(np.int64) != g[f"rays.{tag}.inds"]).sum()) if rr["inds"].shape == g[f"rays.{tag}.inds"].shape else 1.0
        errs[tag + " rays_d"] = float(np.abs(rr["rays_d"].numpy() - g[f"rays.{tag}.rays_d"]).max())
        errs[tag + " rays_o"] = float(np.abs(rr["rays_o"].numpy() - g[f"rays.{tag}.rays_o"]).max())
        errs[tag + " ij"] = float(np.abs(np.stack([rr["i"].numpy(), rr["j"].numpy()]) - g[f"rays.{tag}.ij"]).max())
    print("get_rays", errs)
    worst = max(worst, *[0.0 if v <= 3e-7 else v for v in errs.values()])        # directions: numpy vs torch division order, 1-2 ulp
    print("worst", worst)
    return worst


if __name__ == "__main__":
    sys.exit(0 if (infer() if "--infer" in sys.argv else main()) <= 1e-6 else 1)
