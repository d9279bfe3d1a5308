"""Shared scaffolding for frame-level parity tests: one synthetic May-shaped model + one frame of driving inputs,
rendered by the CPU oracle and by the product on the GPU from identical numpy arrays."""
import os

import numpy as np
import torch

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams, CLASSES

from genefaceplusplus_amd.synthetic import frame_case, SD_FAMILY      # noqa: F401  (the case builder lives with the other synthetic inputs)


TRAINED_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained")
_TRAINED_CLIP = {}


def trained_case(variant, HW, frame_idx=7):
    """A frame of the TRAINED procedural field (tools/make_trained_checkpoint.py -> tests/golden/trained/<torso variant>.npz, fit curves in fit_log.json beside
    it): the package's own training path fitted the May architecture to genefaceplusplus_amd.procedural.ProceduralClip; this is that checkpoint with the
    clip's own driving signals and background for frame `frame_idx` (7 = a held-out frame).  Head-only variants take the head part of the torso file."""
    from genefaceplusplus_amd.procedural import ProceduralClip, head_only_state
    sr = variant.endswith("_sr")
    torso_variant = "may_torso_sr" if sr else "may_torso"
    sd = syn.load_compact_state(os.path.join(TRAINED_DIR, torso_variant + ".npz"))
    hp = may_hparams(variant)
    head_only = variant.startswith("may_head")
    if head_only:
        sd = head_only_state(sd)
        hp["eye_blink_dim"] = may_hparams(torso_variant).get("eye_blink_dim", hp.get("eye_blink_dim"))      # the head a torso model was built on (see the tool)
        if sr:
            sd["lambda_ambient"] = np.array([1.0], np.float32)
    clip = _TRAINED_CLIP.setdefault(256, ProceduralClip(T=256, seed=0))
    k = frame_idx
    bg = clip.background_image(HW).numpy()
    if head_only:        # the head stage's background is the torso layer over the static background (tasks/radnerfs/radnerf.py:117)
        rows = torch.arange(HW) / (HW - 1) * 2 - 1
        vv, uu = torch.meshgrid(rows, rows, indexing="ij")
        trgb, ta = clip.torso(k, uu.reshape(-1), vv.reshape(-1))
        bg = (trgb * ta[:, None] + torch.from_numpy(bg[0]) * (1 - ta[:, None])).numpy()[None]
    return {"variant": variant, "hp": hp, "sd": sd, "HW": HW, "pose": clip.ngp_poses[k:k + 1].copy(), "intr": syn.intrinsics_for(HW, HW),
            "cond": clip.cond_window(k, hp["smo_win_size"]).numpy(), "lm68": clip.lm68s[k].reshape(-1).numpy(),
            "eye_area_percent": clip.eye_area_percents[k].reshape(1, 1).numpy(), "bg_color": np.ascontiguousarray(bg, np.float32), "T_thresh": 0.01,
            "clip": clip, "frame_idx": k}


def nonconvex_occupancy(case, kind, seed=7):
    """Replace the scene's convex ellipsoid by an occupancy with HOLES along the rays (occupied -> empty -> occupied), the shape a trained
    checkpoint's bitfield has after update_extra_state (renderer.py:202-284: min(mean_density, density_thresh) on a dilated, decayed probe).
      'speckle': the kernel-level cases' grid (tests/ref_kernel_cases.py:42-54: an ellipsoid + 2 % random cells, random densities), every cascade;
      'shell'  : a hollow ellipsoid shell with a detached blob in front of it and one behind (rays meet 2-3 separate occupied runs).
    Returns the case with density_grid / density_bitfield replaced (arrays copied, nothing else touched)."""
    import math
    from ref_kernel_cases import make_density, morton3D_np, packbits_np
    hp = case["hp"]
    H = int(hp["grid_size"])
    C = 1 + math.ceil(math.log2(hp["bound"]))
    if kind == "speckle":
        grid = make_density(C, H, seed)
    elif kind == "shell":
        ii, jj, kk = np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij")
        m = morton3D_np(ii.ravel(), jj.ravel(), kk.ravel())
        grid = np.zeros((C, H ** 3), np.float32)
        for c in range(C):
            b = float(min(2 ** c, hp["bound"]))
            x, y, z = (((a.ravel() + 0.5) / H * 2 - 1) * b for a in (ii, jj, kk))
            r_out = (x / 0.42) ** 2 + (y / 0.30) ** 2 + (z / 0.46) ** 2
            r_in = (x / 0.30) ** 2 + (y / 0.17) ** 2 + (z / 0.34) ** 2
            blob_front = (x - 0.05) ** 2 + (y - 0.42) ** 2 + (z + 0.08) ** 2 < 0.06 ** 2          # towards the camera (+y)
            blob_back = (x + 0.10) ** 2 + (y + 0.40) ** 2 + (z - 0.05) ** 2 < 0.08 ** 2
            grid[c, m] = np.where(((r_out <= 1.0) & (r_in > 1.0)) | blob_front | blob_back, 20.0, 0.0).astype(np.float32)
    else:
        raise ValueError(kind)
    case = dict(case)
    case["sd"] = dict(case["sd"])
    case["sd"]["density_grid"] = grid
    case["sd"]["density_bitfield"] = packbits_np(grid, hp["density_thresh"])
    return case


def pose_at(distance=4.0, yaw_deg=0.0, shift=(0.0, 0.0, 0.0), away=False):
    """ngp cam2world [1,4,4]: the bench camera (on +y, looking at the origin along -y) moved around -- closer / inside the AABB, turned about z,
    translated sideways (part of the frame misses the box), or turned away from it (every ray misses: the NaN-depth convention)."""
    import math
    base = np.array([[-1, 0, 0, 0], [0, 0, -1, distance], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    if away:
        base[:3, :3] = base[:3, :3] @ np.diag([-1.0, 1.0, -1.0])       # half a turn about the camera's own y axis: view direction +y, away from the box
    th = math.radians(yaw_deg)
    c, s = math.cos(th), math.sin(th)
    rz = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    pose = rz @ base
    pose[:3, 3] += np.asarray(shift, np.float64)
    return pose.astype(np.float32)[None]


def oracle_render(orc, case, trace=None):
    return orc.render_case(case, trace=trace)


def build_model(case, device, executor):
    from genefaceplusplus_amd import radnerfs
    model = getattr(radnerfs, CLASSES[case["variant"]])(case["hp"])
    sd = dict(case["sd"])
    if hasattr(model, "sr_net"):
        for k, v in syn.synthetic_sr_state().items():  # sr_net.* with the reference's layout (tests/golden/sr_state_manifest.json) unless the case brings its own
            sd.setdefault(k, v)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(device).eval()
    model.executor = executor
    return model


def product_render(model, case, device, rays_from="oracle", orc=None):
    """Call render() exactly like inference/genefacepp_infer.py:461-463 does (whole hparams dict splatted in)."""
    from genefaceplusplus_amd.radnerfs import camera
    HW = case["HW"]
    pose = torch.from_numpy(case["pose"]).to(device)
    if rays_from == "oracle":
        r = orc.get_rays(case["pose"], case["intr"], HW, HW)
        rays_o, rays_d = torch.from_numpy(r["rays_o"]).to(device), torch.from_numpy(r["rays_d"]).to(device)
    else:
        r = camera.get_rays(pose, case["intr"], HW, HW)
        rays_o, rays_d = r["rays_o"], r["rays_d"]
    with torch.no_grad():
        return model.render(rays_o, rays_d, torch.from_numpy(case["cond"]).to(device), camera.get_bg_coords(HW, HW, device),
                            camera.convert_poses(pose), index=0, staged=False, bg_color=torch.from_numpy(case["bg_color"]).to(device),
                            lm68=torch.from_numpy(case["lm68"]).to(device), perturb=False, force_all_rays=False,
                            T_thresh=case["T_thresh"], eye_area_percent=torch.from_numpy(case["eye_area_percent"]).to(device),
                            **case["hp"])


def compare_frames(res, ref, variant, HW, rgb_tol=2e-4, depth_tol=1e-3, frac=5e-4):
    """SURVEY 8c tolerance: max-abs <= 2e-4 on rgb / 1e-3 on depth, allowing <= 0.05 % of pixels to exceed (rays whose
    transmittance crosses T_thresh within rounding)."""
    rgb = res["rgb_map"].float().cpu().numpy()
    if variant in ("may_torso_sr", "may_head_sr"):
        rgb = np.transpose(rgb, (0, 2, 3, 1))
    rgb = rgb.reshape(-1, 3)
    err = np.abs(rgb - ref["rgb_map"].reshape(-1, 3)).max(axis=1)
    stats = {"rgb_max": float(err.max()), "rgb_frac_over": float((err > rgb_tol).mean())}
    assert stats["rgb_frac_over"] <= frac, stats
    d = res["depth_map"].float().cpu().numpy().reshape(-1)
    dref = ref["depth_map"].reshape(-1)
    ok = np.isfinite(dref)
    derr = np.abs(d[ok] - dref[ok])
    stats["depth_frac_over"] = float((derr > depth_tol).mean()) if derr.size else 0.0      # (every ray may miss the box: no finite depth at all)
    assert stats["depth_frac_over"] <= frac, stats
    # rays that miss the box: near = far = FLT_MAX and the reference's (depth - near).clamp(0) / (far - near) is 0 / 0 (renderer.py:396)
    stats["depth_nan"] = int(np.isnan(dref).sum())
    assert np.array_equal(np.isnan(d), np.isnan(dref)), ("NaN pattern of depth_map", int(np.isnan(d).sum()), stats["depth_nan"])
    if "torso_alpha_map" in ref:
        ta = res["torso_alpha_map"].float().cpu().numpy().reshape(-1)
        taerr = np.abs(ta - ref["torso_alpha_map"].reshape(-1))
        stats["alpha_max"] = float(taerr.max())
        assert (taerr > rgb_tol).mean() <= frac, stats
        # the torso field per pixel: composited torso colour (rgb * alpha + bg * (1 - alpha)), one torso sample per pixel
        tb = res["torso_rgb_map"].float().cpu().numpy()
        if variant == "may_torso_sr":
            tb = np.transpose(tb, (0, 2, 3, 1))
        tberr = np.abs(tb.reshape(-1, 3) - ref["torso_rgb_map"].reshape(-1, 3)).max(axis=1)
        stats["torso_rgb_max"] = float(tberr.max())
        assert (tberr > rgb_tol).mean() <= frac, stats
    return stats
