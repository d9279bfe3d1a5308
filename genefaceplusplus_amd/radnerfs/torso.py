"""Head + torso two-pass models: ``RADNeRFTorso`` and ``RADNeRFTorsowithSR``.

Drop-ins for the reference's modules/radnerfs/radnerf_torso.py:17-199 and radnerf_torso_sr.py:17-244 (constructor,
parameter/buffer names incl. the ``torso_canonicial_net`` spelling, ``forward_torso`` / ``render`` signatures and result
keys).  The torso is a 2-D deformation field over the background pixel grid: frequency-encode the pixel, predict an
offset, look the displaced pixel up in a 2-D tiled grid, decode (alpha, rgb), then composite
head over torso over background.
"""
import random

from .._lib import GfppError

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import raymarching
from .cond_nets import MLP
from .encoders import get_encoder
from .head import RADNeRF
from .camera import convert_poses

_CHIN_LANDMARKS = [5, 6, 7, 8, 9, 10, 11]   # the 7 of 68 landmarks the SR variant conditions on (radnerf_torso_sr.py:86)


def _head_aware_encoder():
    return nn.Sequential(nn.Linear(4, 16, bias=True), nn.LeakyReLU(0.02, True), nn.Linear(16, 32, bias=True),
                         nn.LeakyReLU(0.02, True), nn.Linear(32, 16, bias=True))


class _TorsoBase(RADNeRF):
    """State and compositing shared by both torso variants."""

    #: True for RADNeRFTorsowithSR: landmark conditioning, no pose encoding in the MLP input, no host coin flip
    landmark_conditioned = False
    #: fused executor only: also return result['deform'] ([P,2], masked pixels) like the reference; costs one host sync
    return_deform = False

    def __init__(self, hparams):
        super().__init__(hparams)
        self.register_buffer("density_grid_torso", torch.zeros(self.grid_size ** 2))
        self.mean_density_torso = 0      # plain attribute, NOT in the checkpoint => 0 at inference (radnerf_torso.py:22)
        self.density_thresh_torso = hparams["density_thresh_torso"]
        self.torso_individual_embedding_num = hparams["individual_embedding_num"]
        self.torso_individual_embedding_dim = hparams["torso_individual_embedding_dim"]
        if self.torso_individual_embedding_dim > 0:
            self.torso_individual_codes = nn.Parameter(torch.randn(self.torso_individual_embedding_num, self.torso_individual_embedding_dim) * 0.1)

    def _build_torso_nets(self, hparams, cond_dim):
        self.torso_pose_embedder, self.pose_embedding_dim = get_encoder("frequency", input_dim=6, multires=4)
        self.torso_deform_pos_embedder, self.torso_deform_pos_dim = get_encoder("frequency", input_dim=2, multires=10)
        self.torso_embedder, self.torso_in_dim = get_encoder("tiledgrid", input_dim=2, num_levels=16, level_dim=2, base_resolution=16,
                                                             log2_hashmap_size=16, desired_resolution=2048)
        deform_in = self.torso_deform_pos_dim + cond_dim + self.torso_individual_embedding_dim
        canon_in = self.torso_in_dim + deform_in
        if hparams["torso_head_aware"]:
            self.head_color_weights_encoder = _head_aware_encoder()
            deform_in += 16
            canon_in += 16
        self.torso_deform_net = MLP(deform_in, 2, 64, 3)
        self.torso_canonicial_net = MLP(canon_in, 4, 32, 3)

    # -- the per-pixel torso field, stand-alone API (render() uses the fused kernel) ---------------------------
    def _frame_constant_columns(self, poses, c, lm68):
        raise NotImplementedError

    def _forward_torso(self, x, poses, c, image, weights_sum, lm68):
        hp = self.hparams
        x = x * hp["torso_shrink"]
        n = x.shape[0]
        cols = [self.torso_deform_pos_embedder(x)] + [v.reshape(1, -1).expand(n, -1) for v in self._frame_constant_columns(poses, c, lm68)]
        h = torch.cat(cols, dim=-1)
        if hp["torso_head_aware"]:
            if image is None:
                image = torch.zeros(n, 3, dtype=h.dtype, device=h.device)
                weights_sum = torch.zeros(n, 1, dtype=h.dtype, device=h.device)
            h = torch.cat([h, self.head_color_weights_encoder(torch.cat([image, weights_sum], dim=-1))], dim=-1)
        dx = self.torso_deform_net(h)
        moved = (x + dx).clamp(-1, 1).float()
        feat = self.torso_embedder(moved, bound=1)
        out = self.torso_canonicial_net(torch.cat([feat.to(h.dtype), h], dim=-1))
        return torch.sigmoid(out[..., :1]), torch.sigmoid(out[..., 1:]), dx

    def _torso_code(self, index):
        if self.torso_individual_embedding_dim <= 0:
            return None
        return self.torso_individual_codes[index if self.training else 0]

    def _probe_pose_and_landmarks(self, pick):
        raise NotImplementedError

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Torso stage: only the 2-D torso occupancy grid is refreshed, the head grid stays frozen (radnerf_torso.py:201-244).
        One jittered probe per pixel cell under a random training pose, 5x5 max-dilated, blended with a decaying maximum."""
        G = self.grid_size
        dev = self.density_grid_torso.device
        pick = random.randint(0, self.poses.shape[0] - 1)
        pose, lm68 = self._probe_pose_and_landmarks(pick)
        code = self.torso_individual_codes[[pick]] if self.torso_individual_embedding_dim > 0 else None
        probe = torch.zeros_like(self.density_grid_torso)
        half_cell = 1 / G
        for x0 in range(0, G, S):
            for y0 in range(0, G, S):
                xs = torch.arange(x0, min(x0 + S, G), dtype=torch.int32, device=dev)
                ys = torch.arange(y0, min(y0 + S, G), dtype=torch.int32, device=dev)
                coords = torch.stack(torch.meshgrid(xs, ys, indexing="ij"), dim=-1).reshape(-1, 2)
                cell = (coords[:, 1] * G + coords[:, 0]).long()          # row-major in (y, x): the grid is sampled as an image
                pts = (2 * coords.float() / (G - 1) - 1) * (1 - half_cell)
                pts += (torch.rand_like(pts) * 2 - 1) * half_cell
                alpha, _, _ = self._forward_torso(pts, pose, code, None, None, lm68)
                probe[cell] = alpha.squeeze(1).float()
        probe = F.max_pool2d(probe.view(1, 1, G, G), kernel_size=5, stride=1, padding=2).view(-1)
        self.density_grid_torso = torch.maximum(self.density_grid_torso * decay, probe)
        self.mean_density_torso = torch.mean(self.density_grid_torso).item()

    def _torso_mask(self, bg_coords):
        thresh = min(self.density_thresh_torso, self.mean_density_torso)
        occ = F.grid_sample(self.density_grid_torso.view(1, 1, self.grid_size, self.grid_size), bg_coords.view(1, -1, 1, 2),
                            align_corners=True).view(-1)
        return occ > thresh

    def _render_staged(self, rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, max_steps, T_thresh, lm68,
                       eye_area_percent, use_head_for_torso, force_all_rays=False):
        N = rays_o.shape[0]
        dev = rays_o.device
        extra = {}
        with torch.no_grad():       # the head field is frozen while the torso trains (radnerf_torso.py:93)
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
            cond_feat = self.cal_cond_feat(cond, eye_area_percent=eye_area_percent)
            if self.training:
                weights_sum, ambient_sum, depth, image, _ = self._march_eval_composite_train(
                    rays_o, rays_d, nears, fars, cond_feat, self._individual_code(index), dt_gamma, max_steps, perturb, force_all_rays)
                extra = {"weights_sum": weights_sum, "ambient": ambient_sum}
            else:
                weights_sum, depth, image = self._march_eval_composite_staged(rays_o, rays_d, nears, fars, cond_feat, self._individual_code(index),
                                                                              dt_gamma, max_steps, T_thresh, perturb)
        if bg_color is None:
            bg_color = 1
        code = self._torso_code(index)
        mask = self._torso_mask(bg_coords)
        torso_alpha = torch.zeros(N, 1, device=dev)
        torso_color = torch.zeros(N, 3, device=dev)
        deform = None
        if mask.any():
            if use_head_for_torso is None:          # radnerf_torso.py:177-180: the coin is drawn only when there is a masked pixel
                use_head_for_torso = random.random() < 0.5
            if self.hparams["torso_head_aware"] and use_head_for_torso:
                a, col, deform = self._forward_torso(bg_coords[mask], poses, code, image[mask], weights_sum.unsqueeze(-1)[mask], lm68)
            else:
                a, col, deform = self._forward_torso(bg_coords[mask], poses, code, None, None, lm68)
            torso_alpha[mask] = a.float()
            torso_color[mask] = col.float()
        torso_bg = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
        image = (image + (1 - weights_sum).unsqueeze(-1) * torso_bg)
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"image": image.clamp(0, 1), "depth": depth, "torso_alpha": torso_alpha, "torso_bg": torso_bg, "deform": deform, **extra}

    def _render_common(self, rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, max_steps, T_thresh, lm68,
                       eye_area_percent, use_head_for_torso, post=None, post_key=(), force_all_rays=False, shard=None):
        """`post(out)`: extra device work on the pipeline's result dict (the SR stage), issued inside the frame so that it is part of the
        captured graph; `post_key` distinguishes graphs captured with different post work."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        bg_coords = bg_coords.contiguous().view(-1, 2)
        fused = self.executor == "fused" and not self.training and self._fused_ok(perturb, max_steps)
        if shard is not None and not fused:
            raise GfppError("ray_shard needs the fused executor at inference (the staged loop has no frame-wide alive count)")
        if fused:
            if use_head_for_torso is None:
                use_head_for_torso = random.random() < 0.5
            ind_code, torso_code = self._individual_code(index), self._torso_code(index)

            def frame(rays_o, rays_d, cond, eye, bg_coords, poses, lm68, bg_color):
                def cond_feat():                                                        # runs on the pipeline's side stream
                    with torch.no_grad():
                        return self.cal_cond_feat(cond, eye_area_percent=eye)
                if self._clip_cond_feat is not None:                                    # the clip renderer's precomputed constants of this frame
                    from .frame_pipeline import FoldedConsts
                    cond_feat = FoldedConsts(self._clip_cond_feat)
                o = self.pipeline().render_head_torso(rays_o, rays_d, cond_feat, ind_code, bg_coords, poses, torso_code, lm68, dt_gamma,
                                                      max_steps, T_thresh, bg_color, use_head_for_torso, shard=shard)
                if post is not None:
                    post(o)
                return o
            inputs = {"rays_o": rays_o, "rays_d": rays_d, "cond": cond, "eye": eye_area_percent, "bg_coords": bg_coords, "poses": poses,
                      "lm68": lm68, "bg_color": bg_color}
            if self.use_graph and not torch.is_grad_enabled() and shard is None:
                out = self.pipeline().graphed(("torso", float(dt_gamma), int(max_steps), float(T_thresh), bool(use_head_for_torso)) + tuple(post_key),
                                              frame, inputs)
            else:
                out = frame(**inputs)
            if self.return_deform:
                # the reference returns dx of the masked pixels only ([P,2]); compacting needs a host sync, hence opt-in
                out["deform"] = out["deform_dense"][out["torso_mask"].bool()]
            return out
        out = self._render_staged(rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, max_steps, T_thresh, lm68,
                                  eye_area_percent, use_head_for_torso, force_all_rays)
        if post is not None:
            post(out)
        return out


    # -- K consecutive frames of a clip through one persistent head launch (clip.ClipRenderer, frame groups) ---------------------------------------
    def group_supported(self, N, K, max_steps, perturb=False):
        return (self.executor == "fused" and not self.training and self._fused_ok(perturb, max_steps) and self.pipeline().group_supported(N, K, max_steps))

    def render_group(self, consts, bg_coords, poses, lm68s, index=0, dt_gamma=0, bg_color=None, max_steps=1024, T_thresh=1e-4, upscale_torso=False,
                     sr_noise_mode="random", after_frame=None, ngp_poses=None, camera=None, **kwargs):
        """render() for K = len(consts) frames whose RAYS the caller has put into pipeline().group_workspace(N, K, max_steps)[2]['rays_o' / 'rays_d'] and
        whose conditioning is given as folded constants (consts[k]: 256 values, RADNeRF.frame_consts_rows) -- the frame loop of
        inference/genefacepp_infer.py:460-469 taken K frames at a time.  Every frame is the bits of its own render() call (per sample and per ray nothing
        changes; the super-resolution noise of 'random' mode is drawn per launch either way).  poses [K, 1, 6] / lm68s [K, 136]: per frame.
        after_frame(k, result dict of frame k): issued right behind the frame's last kernel (the clip renderer's uint8 store).
        ngp_poses (K equally spaced [4, 4] cam2world views) + camera (fx, fy, cx, cy, H, W): generate the rays on the device inside the group's prologue launch
        instead of reading them from the workspace.  Returns the K result dicts with render()'s keys."""
        K = len(consts)
        N = int(bg_coords.reshape(-1, 2).shape[0])
        if not self.group_supported(N, K, max_steps):
            raise GfppError("render_group: not available for this model / precision / executor (see FramePipeline.group_supported)")
        ind_code, torso_code = self._individual_code(index), self._torso_code(index)
        use_head = True if self.landmark_conditioned else (kwargs.get("use_head_for_torso") if self.hparams["torso_head_aware"] else False)
        if use_head is None:
            use_head = random.random() < 0.5               # one coin per frame group (render(): one per frame; only torso_head_aware non-SR models draw it)
        sr = getattr(self, "sr_net", None)
        side = sr.input_resolution if sr is not None else None
        results = [None] * K

        def finish(k, out):
            if sr is not None:
                rgb = out["image"].reshape(1, side, side, 3).permute(0, 3, 1, 2)
                res = {"torso_alpha_map": out["torso_alpha"], "torso_rgb_map": out["torso_bg"].reshape(1, side, side, 3).permute(0, 3, 1, 2),
                       "depth_map": out["depth"].view(1, N), "rgb_map": rgb}
                if sr.ready:
                    res["sr_rgb_map"] = sr(rgb, noise_mode=sr_noise_mode, clamp01=True, clip_sub=k)
                    if upscale_torso:
                        res["sr_torso_rgb_map"] = sr(res["torso_rgb_map"], noise_mode=sr_noise_mode, clamp01=True)
            else:
                res = {"torso_alpha_map": out["torso_alpha"], "torso_rgb_map": out["torso_bg"], "depth_map": out["depth"].view(1, N), "rgb_map": out["image"].view(1, N, 3)}
            results[k] = res
            if after_frame is not None:
                after_frame(k, res)
        torso_inputs = lm68s if self.landmark_conditioned else poses
        self.pipeline().render_group_head_torso(consts, ind_code, bg_coords, torso_inputs, torso_code, dt_gamma, max_steps, T_thresh, bg_color, use_head,
                                                after_frame=finish, poses=ngp_poses, camera=camera)
        return results


class RADNeRFTorso(_TorsoBase):
    _render_passes_eye = False        # radnerf_torso.py:106 calls cal_cond_feat(cond) without eye_area_percent

    """Pose-conditioned torso (non-SR configs, 512x512 rays)."""

    def __init__(self, hparams):
        super().__init__(hparams)
        self._build_torso_nets(hparams, cond_dim=6 + 6 * 2 * 4)

    def _frame_constant_columns(self, poses, c, lm68):
        cols = [self.torso_pose_embedder(poses)]
        if c is not None:
            cols.append(c)
        return cols

    def forward_torso(self, x, poses, c=None, image=None, weights_sum=None):
        """x [P,2] in [-1,1], poses [1,6] -> alpha [P,1], color [P,3], dx [P,2] (radnerf_torso.py:51-84)."""
        return self._forward_torso(x, poses, c, image, weights_sum, None)

    def _probe_pose_and_landmarks(self, pick):
        return convert_poses(self.poses[[pick]]).to(self.density_grid_torso.device), None

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False,
               max_steps=1024, T_thresh=1e-4, **kwargs):
        prefix = rays_o.shape[:-1]
        # The reference flips a host coin when torso_head_aware, but only inside `if mask.any()` (radnerf_torso.py:177-180).  The staged
        # executor (which knows the mask on the host) draws it exactly there: `use_head=None` = "draw when needed".  The fused executor never
        # learns the mask on the host (no sync), so it draws once per frame: the RNG streams differ only on frames whose torso mask is empty.
        # `use_head_for_torso=` (not a reference argument) lets a caller that renders ONE frame on several ranks make the draw once for all of them
        # (frames.render_frame_tiled).
        use_head = kwargs.get("use_head_for_torso") if self.hparams["torso_head_aware"] else False
        # NB: this variant calls cal_cond_feat(cond) without eye_area_percent (radnerf_torso.py:106)
        out = self._render_common(rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, max_steps, T_thresh, None,
                                  None, use_head, force_all_rays=force_all_rays, shard=kwargs.get("ray_shard"))
        res = {"torso_alpha_map": out["torso_alpha"], "torso_rgb_map": out["torso_bg"], "depth_map": out["depth"].view(*prefix),
               "rgb_map": out["image"].view(*prefix, 3)}
        if out["deform"] is not None:
            res["deform"] = out["deform"]
        for k in ("weights_sum", "ambient"):        # training only (radnerf_torso.py:126-128)
            if k in out:
                res[k] = out[k]
        return res


class RADNeRFTorsowithSR(_TorsoBase):
    """Landmark-conditioned, head-aware torso of the released May checkpoint (256x256 rays + super-resolution)."""

    landmark_conditioned = True

    def __init__(self, hparams):
        super().__init__(hparams)
        self.lm68_embedder, self.lm68_embedding_dim = get_encoder("frequency", input_dim=7 * 2, multires=4)
        self._build_torso_nets(hparams, cond_dim=self.lm68_embedding_dim)
        from .superres import Superresolution
        self.sr_net = Superresolution(channels=3)

    # -- training-stage switches the reference trainer calls (tasks/radnerfs/radnerf_torso_sr.py:192; radnerf_torso_sr.py:58-73) -----------
    def on_train_torso_nerf(self):
        self.requires_grad_(False)
        if self.torso_individual_embedding_dim > 0:
            self.torso_individual_codes.requires_grad_(True)
        self.torso_pose_embedder.requires_grad_(True)
        self.torso_deform_pos_embedder.requires_grad_(True)
        self.torso_embedder.requires_grad_(True)
        if self.hparams["torso_head_aware"]:
            self.head_color_weights_encoder.requires_grad_(True)
        self.torso_deform_net.requires_grad_(True)
        self.torso_canonicial_net.requires_grad_(True)
        self.sr_net.requires_grad_(False)

    def on_train_superresolution(self):
        self.requires_grad_(False)
        self.sr_net.requires_grad_(True)

    def _frame_constant_columns(self, poses, c, lm68):
        # the reference also encodes `poses` here but never uses the result (radnerf_torso_sr.py:84,89-96)
        chin = lm68.reshape(1, 68, 2)[:, _CHIN_LANDMARKS].reshape(1, -1)
        cols = [c] if c is not None else []
        cols.append(self.lm68_embedder(chin))
        return cols

    def forward_torso(self, x, poses, c=None, image=None, weights_sum=None, lm68=None):
        """(radnerf_torso_sr.py:75-114)."""
        return self._forward_torso(x, poses, c, image, weights_sum, lm68)

    def _probe_pose_and_landmarks(self, pick):
        dev = self.density_grid_torso.device         # radnerf_torso_sr.py:253-255
        return convert_poses(self.poses[[pick]]).to(dev), self.lm68s[[pick]].to(dev).reshape(1, 68 * 2)

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False,
               max_steps=1024, T_thresh=1e-4, upscale_torso=False, lm68=None, eye_area_percent=None, **kwargs):
        sr_noise = kwargs.get("sr_noise_mode", "random")     # the reference always renders with the layers' default, 'random'
        side = self.sr_net.input_resolution          # 256: the reference hard-codes [1,256,256,3] (radnerf_torso_sr.py:219,229)

        def superresolve(o):
            if self.sr_net.ready:
                o["sr_rgb"] = self.sr_net(o["image"].reshape(1, side, side, 3).permute(0, 3, 1, 2), noise_mode=sr_noise, clamp01=True, clip_sub=0)
                if upscale_torso:
                    o["sr_torso_rgb"] = self.sr_net(o["torso_bg"].reshape(1, side, side, 3).permute(0, 3, 1, 2), noise_mode=sr_noise, clamp01=True)

        out = self._render_common(rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, max_steps, T_thresh, lm68,
                                  eye_area_percent, True, post=superresolve, post_key=("sr", sr_noise, bool(upscale_torso)),
                                  force_all_rays=force_all_rays)
        rgb = out["image"].reshape(1, side, side, 3).permute(0, 3, 1, 2)
        torso_bg = out["torso_bg"].reshape(1, side, side, 3).permute(0, 3, 1, 2)
        res = {"torso_alpha_map": out["torso_alpha"], "torso_rgb_map": torso_bg, "depth_map": out["depth"].view(*rays_o.shape[:-1]),
               "rgb_map": rgb}
        if out["deform"] is not None:
            res["deform"] = out["deform"]
        for k in ("weights_sum", "ambient"):
            if k in out:
                res[k] = out[k]
        if "sr_rgb" in out:
            res["sr_rgb_map"] = out["sr_rgb"]
        if "sr_torso_rgb" in out:
            res["sr_torso_rgb_map"] = out["sr_torso_rgb"]
        return res


class RADNeRFwithSR(RADNeRF):
    """Head-only model of the *_sr configs (reference: modules/radnerfs/radnerf_sr.py:45-210, which repeats RADNeRF's
    networks and adds ``sr_net`` + the ``lambda_ambient`` scalar).  Renders 256x256 rays, then the SR stage (csrc/superres.hip) -> 512x512."""

    def __init__(self, hparams):
        super().__init__(hparams)
        from .superres import Superresolution
        self.sr_net = Superresolution(channels=3)
        self.lambda_ambient = nn.Parameter(torch.tensor([1.0]), requires_grad=False)

    # -- training-stage switches (radnerf_sr.py:116-122) ----------------------------------------------------------------------------------
    def on_train_nerf(self):
        self.requires_grad_(True)
        self.sr_net.requires_grad_(False)

    def on_train_superresolution(self):
        self.requires_grad_(False)
        self.sr_net.requires_grad_(True)

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False,
               max_steps=1024, T_thresh=1e-4, cond_mask=None, eye_area_percent=None, **kwargs):
        res = super().render(rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, force_all_rays, max_steps, T_thresh,
                             cond_mask, eye_area_percent=eye_area_percent, **kwargs)
        side = self.sr_net.input_resolution
        rgb = res["rgb_map"].reshape(1, side, side, 3).permute(0, 3, 1, 2)
        res["rgb_map"] = rgb
        if self.sr_net.ready:
            # the reference always renders with the layers' default noise ('random', radnerf_sr.py:30-43); `sr_noise_mode` is our test hook
            res["sr_rgb_map"] = self.sr_net(rgb.clone(), noise_mode=kwargs.get("sr_noise_mode", "random"), clamp01=True, clip_sub=0)
        return res
