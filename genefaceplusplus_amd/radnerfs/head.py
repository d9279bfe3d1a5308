"""Head radiance field: ``NeRFRenderer`` (state + render entry point) and ``RADNeRF`` (networks).

Drop-in for the reference's modules/radnerfs/renderer.py:66-399 and radnerf.py:13-166: same constructor
(``Model(hparams: dict)``), attributes (``.hparams``), parameter/buffer names, ``render / forward / density /
cal_cond_feat`` signatures and result keys.  What differs is the execution: ``render()`` hands the frame to the fused
HIP pipeline (frame_pipeline.py: device-side loop control, no host syncs); the reference-shaped ``staged`` executor below
(one march / evaluate / composite round trip per loop iteration, MLPs through rocBLAS) is kept as a debugging aid and as
the "unfused" baseline for measurements.
"""
import copy
import ctypes
import os
import warnings
import math
import random

import numpy as np
import torch
import torch.nn as nn

from . import raymarching
from .. import _lib, tuning
from .._lib import GfppError
from .cond_nets import AudioNet, AudioAttNet, MLP, SplitFirstColumn
from .encoders import get_encoder
from .camera import trunc_exp, get_audio_features

_COND_DIMS = {"esperanto": 44, "deepspeech": 29}
_KEYPOINT_DIMS = {"lm68": 68 * 3, "lm131": 131 * 3, "lm468": 468 * 3}


class NeRFRenderer(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.bound = hparams["bound"]
        self.cascade = 1 + math.ceil(math.log2(hparams["bound"]))
        self.grid_size = hparams["grid_size"]
        self.density_scale = 1
        self.min_near = hparams["min_near"]
        self.density_thresh = hparams["density_thresh"]
        self.cuda_ray = hparams["cuda_ray"]

        b = float(self.bound)
        aabb = torch.tensor([-b, -b / 2, -b, b, b / 2, b], dtype=torch.float32)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

        self.individual_embedding_num = hparams["individual_embedding_num"]
        self.individual_embedding_dim = hparams["individual_embedding_dim"]
        if self.individual_embedding_dim > 0:
            self.individual_embeddings = nn.Parameter(torch.randn(self.individual_embedding_num, self.individual_embedding_dim) * 0.1)

        cells = self.grid_size ** 3
        self.register_buffer("density_grid", torch.zeros(self.cascade, cells))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * cells // 8, dtype=torch.uint8))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        # not persisted, exactly like the reference (renderer.py:97-104)
        self.mean_density = 0
        self.iter_density = 0
        self.mean_count = 0
        self.local_step = 0
        #: 'fused' (default) or 'staged'; see module docstring
        self.executor = "fused"
        #: arithmetic of the five wide head layers in the fused executor: 'fp32' (exact-fp32 MFMA), 'fp16' / 'bf16' (16-bit MFMA
        #: operands, fp32 accumulation), or 'auto' = follow the caller's torch.autocast context like nn.Linear does in the
        #: reference (genefacepp_infer.py renders under autocast -> fp16; no autocast -> fp32)
        self.precision = "auto"
        #: replay each frame from a captured hipGraph (frame_pipeline.GraphedFrame).  The returned tensors are then the graph's
        #: static outputs, valid until the next render() of the same shape -- fine for the reference's caller, opt-in otherwise
        self.use_graph = False
        self._pipeline = None

    # -- to be provided by the model ------------------------------------------------------------------------
    def cal_cond_feat(self, cond, **kwargs):
        raise NotImplementedError()

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.step_counter.zero_()
        self.mean_density = self.iter_density = self.mean_count = self.local_step = 0

    # -- shared pieces of render() ------------------------------------------------------------------------
    def _individual_code(self, index):
        if self.individual_embedding_dim <= 0:
            return None
        return self.individual_embeddings[index if self.training else 0]

    # -- occupancy grid upkeep (training side) ------------------------------------------------------------------
    def _cell_block(self, lo, hi):
        """Integer cell coordinates [n,3] of the sub-cube lo..hi (per axis) and their Morton codes."""
        dev = self.density_bitfield.device
        axes = [torch.arange(a, b, dtype=torch.int32, device=dev) for a, b in zip(lo, hi)]
        coords = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)
        return coords, raymarching.morton3D(coords).long()

    def _cell_blocks(self, S):
        G = self.grid_size
        for x in range(0, G, S):
            for y in range(0, G, S):
                for z in range(0, G, S):
                    yield self._cell_block((x, y, z), (min(x + S, G), min(y + S, G), min(z + S, G)))

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells no training camera sees get density -1 and are never marched (renderer.py:131-199).
        poses [B,4,4] camera-to-world, intrinsic (fx, fy, cx, cy)."""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        fx, fy, cx, cy = intrinsic
        dev = self.density_grid.device
        poses = poses.to(dev)
        seen = torch.zeros_like(self.density_grid)
        for coords, cell in self._cell_blocks(S):
            unit = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)              # [1,n,3] in [-1,1]
            for cas in range(self.cascade):
                bound = min(2 ** cas, self.bound)
                half_cell = bound / self.grid_size
                world = unit * (bound - half_cell)
                for head in range(0, poses.shape[0], S):
                    cam = poses[head:head + S]
                    # world -> camera: (p - t) R  (R is camera-to-world, so right-multiplying applies its transpose)
                    p = (world - cam[:, :3, 3].unsqueeze(1)) @ cam[:, :3, :3]
                    inside = (p[..., 2] > 0) \
                        & (p[..., 0].abs() < cx / fx * p[..., 2] + half_cell * 2) \
                        & (p[..., 1].abs() < cy / fy * p[..., 2] + half_cell * 2)
                    seen[cas, cell] += inside.sum(0).to(seen.dtype)
        self.density_grid[seen == 0] = -1

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Refresh the occupancy grid from the current density field, then re-pack the bitfield the marcher reads
        (renderer.py:201-284).  ``self.conds`` ([T, t_win, C], set by the training task) supplies a random conditioning
        window; densities are probed at one jittered point per cell, dilated, and blended in with a decaying maximum."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        pick = random.randint(0, self.conds.shape[0] - 1)
        hp = getattr(self, "hparams", None)           # the reference reads the global hparams here; the model's own copy is the same dict
        window = get_audio_features(self.conds, 2, pick, smo_win_size=hp["smo_win_size"] if hp else None)
        cond_feat = self.cal_cond_feat(window.to(dev))
        probe = torch.zeros_like(self.density_grid)
        for coords, cell in self._cell_blocks(S):
            unit = 2 * coords.float() / (self.grid_size - 1) - 1
            for cas in range(self.cascade):
                bound = min(2 ** cas, self.bound)
                half_cell = bound / self.grid_size
                pts = unit * (bound - half_cell)
                pts += (torch.rand_like(pts) * 2 - 1) * half_cell
                sigma = self.density(pts, cond_feat)["sigma"].reshape(-1).detach().to(probe.dtype)
                probe[cas, cell] = sigma * self.density_scale
        probe = raymarching.morton3D_dilation(probe)
        both = (self.density_grid >= 0) & (probe >= 0)
        self.density_grid[both] = torch.maximum(self.density_grid[both] * decay, probe[both])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        self.density_bitfield = raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh), self.density_bitfield)
        seen_steps = min(16, self.local_step)
        if seen_steps > 0:
            self.mean_count = int(self.step_counter[:seen_steps, 0].sum().item() / seen_steps)
        self.local_step = 0

    def _march_eval_composite_train(self, rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, max_steps, perturb,
                                    force_all_rays):
        """Training-time pass (renderer.py:319-340): one packed sample list for the whole batch of rays, evaluated by the
        autograd-visible networks, composited by the differentiable kernel pair."""
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                force_all_rays, dt_gamma, max_steps)
        sigmas, rgbs, ambient = self(xyzs, dirs, cond_feat, ind_code)
        sigmas = self.density_scale * sigmas
        weights_sum, ambient_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ambient.abs().sum(-1), deltas, rays)
        return weights_sum, ambient_sum, depth, image, xyzs

    _warned = set()

    def _fused_ok(self, perturb=False, max_steps=16, cond_mask=None):
        """Can this call run on the fused frame pipeline?  Otherwise render() falls back to the reference-shaped `staged` executor (same
        results, one host-visible trip per loop iteration, ~40x slower) -- and says so once per reason."""
        from .frame_pipeline import supports
        why = None
        if not supports(self):
            why = "architecture outside the fused kernels' family (hidden 128, layers 3/3/2, 16x2 grids)"
        elif perturb:
            why = "perturb=True"
        elif cond_mask is not None:
            why = "cond_mask given"
        elif max_steps > 63:
            why = f"max_steps={max_steps} > 63"
        if why is None:
            return True
        if why not in NeRFRenderer._warned:
            NeRFRenderer._warned.add(why)
            warnings.warn(f"genefaceplusplus_amd: render() uses the staged executor ({why}); the fused hipGraph path covers "
                          f"perturb=False, cond_mask=None, max_steps<=63 on the shipped architecture", RuntimeWarning, stacklevel=3)
        return False

    def pipeline(self):
        """Lazily build the fused frame pipeline (packs weights for the HIP kernels; rebuilt if parameters move)."""
        from .frame_pipeline import FramePipeline
        if self._pipeline is None or not self._pipeline.matches(self):
            self._pipeline = FramePipeline(self)
        want = self.resolved_precision()
        if self._pipeline.precision != want:
            self._pipeline.set_precision(self, want)
        return self._pipeline

    def resolved_precision(self):
        if self.precision != "auto":
            return self.precision
        if torch.is_autocast_enabled():
            dtype = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            return "bf16" if dtype == torch.bfloat16 else "fp16"
        return "fp32"

    def _march_eval_composite_staged(self, rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, max_steps, T_thresh,
                                     perturb=False, cond_mask=None, trace=None):
        """Reference-shaped loop (renderer.py:341-384): one host-visible trip per iteration."""
        N = rays_o.shape[0]
        dev = rays_o.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        depth = torch.zeros(N, dtype=torch.float32, device=dev)
        image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        rays_alive = torch.arange(N, dtype=torch.int32, device=dev)
        rays_t = nears.clone()
        step = 0
        while step < max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128,
                                                        perturb if step == 0 else False, dt_gamma, max_steps)
            if cond_mask is not None:
                sigmas, rgbs, _ = self(xyzs, dirs, cond_feat, ind_code, cond_mask=cond_mask[rays_alive])
            else:
                sigmas, rgbs, _ = self(xyzs, dirs, cond_feat, ind_code)
            sigmas = self.density_scale * sigmas
            if trace is not None:
                trace.append((n_alive, n_step))
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
            rays_alive = rays_alive[rays_alive >= 0].contiguous()
            step += n_step
        return weights_sum, depth, image

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False,
               force_all_rays=False, max_steps=1024, T_thresh=1e-4, cond_mask=None, eye_area_percent=None, **kwargs):
        """Head-only frame: rays [B,N,3] (B == 1) -> {'rgb_map' [B,N,3], 'depth_map' [B,N]} (renderer.py:286-399)."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        ind_code = self._individual_code(index)
        if self.training:
            cond_feat = self.cal_cond_feat(cond, eye_area_percent=eye_area_percent)
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
            weights_sum, ambient_sum, depth, image, xyzs = self._march_eval_composite_train(
                rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, max_steps, perturb, force_all_rays)
            if bg_color is None:
                bg_color = 1
            image = (image + (1 - weights_sum).unsqueeze(-1) * bg_color).view(*prefix, 3).clamp(0, 1)
            depth = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
            return {"weights_sum": weights_sum, "ambient": ambient_sum, "position": xyzs, "depth_map": depth, "rgb_map": image}
        shard = kwargs.get("ray_shard")          # (process group, rays of the whole frame): this call renders one ray tile (frames.render_frame_tiled)
        if self.executor == "fused" and self._fused_ok(perturb, max_steps, cond_mask):
            def frame(rays_o, rays_d, cond, eye, bg_color):
                # (a clip renderer that computed the conditioning of all frames up front hands the frame's constants in: clip.ClipRenderer, frame_consts_rows)
                pre = self._clip_cond_feat
                if pre is not None:
                    from .frame_pipeline import FoldedConsts
                cond_feat = FoldedConsts(pre) if pre is not None else (lambda: self.cal_cond_feat(cond, eye_area_percent=eye))     # runs on the pipeline's side stream
                return self.pipeline().render_head(rays_o, rays_d, cond_feat, ind_code, dt_gamma, max_steps, T_thresh, bg_color, shard=shard)
            inputs = {"rays_o": rays_o, "rays_d": rays_d, "cond": cond, "eye": eye_area_percent, "bg_color": bg_color}
            if self.use_graph and not torch.is_grad_enabled() and shard is None:
                out = self.pipeline().graphed(("head", float(dt_gamma), int(max_steps), float(T_thresh)), frame, inputs)
            else:
                out = frame(**inputs)
            return {"depth_map": out["depth"].view(*prefix), "rgb_map": out["image"].view(*prefix, 3)}
        if shard is not None:
            raise GfppError("ray_shard needs the fused executor (the staged loop has no frame-wide alive count)")
        cond_feat = self.cal_cond_feat(cond, eye_area_percent=eye_area_percent)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        weights_sum, depth, image = self._march_eval_composite_staged(rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma,
                                                                      max_steps, T_thresh, perturb, cond_mask)
        if bg_color is None:
            bg_color = 1
        image = (image + (1 - weights_sum).unsqueeze(-1) * bg_color).view(*prefix, 3).clamp(0, 1)
        depth = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
        return {"depth_map": depth, "rgb_map": image}


#: a training step's conditioning networks: "fused" = one forward and one backward launch of the library's own kernels (RADNeRF._fused_train_cond_feat; shapes
#: they do not cover stay eager); GFPP_TRAIN_COND=eager: torch's layers, the A/B partner
COND_TRAIN = tuning.HOST["train_cond"]


class _CondFeatTrain(torch.autograd.Function):
    """cal_cond_feat of a training step through gfpp_cond_feat_train_forward / _backward: (cond, eye, descriptor, *parameters) -> cond_feat; the parameters'
    gradients come back in the descriptor's order."""

    @staticmethod
    def forward(ctx, cond, eye, desc, *params):
        cm, plist, fill = desc
        st = torch.cuda.current_stream().cuda_stream
        saved = torch.empty(int(_lib.lib().gfpp_cond_feat_train_floats(ctypes.byref(cm), 0)), dtype=torch.float32, device=cond.device)
        out = torch.empty(cm.dim_aud if cm.with_att else (cm.smo, cm.dim_aud), dtype=torch.float32, device=cond.device)
        _lib.call("gfpp_cond_feat_train_forward", ctypes.byref(cm), cond.data_ptr(), eye.data_ptr() if eye is not None else None, out.data_ptr(), saved.data_ptr(), st)
        ctx.save_for_backward(cond, eye, saved)
        ctx.desc = desc
        return out

    @staticmethod
    def backward(ctx, gout):
        cond, eye, saved = ctx.saved_tensors
        cm, plist, fill = ctx.desc
        st = torch.cuda.current_stream().cuda_stream
        flat = torch.empty(sum(p.numel() for p in plist), dtype=torch.float32, device=cond.device)
        grads, at = [], 0
        for p in plist:
            grads.append(flat[at:at + p.numel()].view(p.shape))
            at += p.numel()
        gm = fill([g.data_ptr() for g in grads])
        scratch = torch.empty(int(_lib.lib().gfpp_cond_feat_train_floats(ctypes.byref(cm), 1)), dtype=torch.float32, device=cond.device)
        _lib.call("gfpp_cond_feat_train_backward", ctypes.byref(cm), ctypes.byref(gm), cond.data_ptr(), eye.data_ptr() if eye is not None else None, saved.data_ptr(),
                  gout.float().contiguous().data_ptr(), scratch.data_ptr(), st)
        # (one storage per gradient: AccumulateGrad may keep what it is handed as p.grad, and views of one flat buffer would alias every conditioning-net
        # gradient to a single allocation at unaligned offsets -- a few hundred KB of copies per step)
        return (None, None, None, *[g.clone() if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads)])


class RADNeRF(NeRFRenderer):
    def __init__(self, hparams):
        super().__init__(hparams)
        self.hparams = copy.deepcopy(hparams)
        cond_type = hparams["cond_type"]
        if cond_type in _COND_DIMS:
            self.cond_in_dim = _COND_DIMS[cond_type]
        elif cond_type == "idexp_lm3d_normalized":
            mode = hparams.get("nerf_keypoint_mode", "lm68")
            if mode not in _KEYPOINT_DIMS:
                raise NotImplementedError()
            self.cond_in_dim = _KEYPOINT_DIMS[mode]
        else:
            raise NotImplementedError()

        self.cond_out_dim = hparams["cond_out_dim"] // 2 * 2
        self.cond_win_size = hparams["cond_win_size"]
        self.smo_win_size = hparams["smo_win_size"]
        self.cond_prenet = AudioNet(self.cond_in_dim, self.cond_out_dim, win_size=self.cond_win_size)
        if hparams.get("add_eye_blink_cond", False):
            half = self.cond_out_dim // 2
            self.blink_embedding = nn.Embedding(1, half)
            self.blink_encoder = nn.Sequential(nn.Linear(half, half), nn.Linear(half, hparams["eye_blink_dim"]))
        self.with_att = hparams["with_att"]
        if self.with_att:
            self.cond_att_net = AudioAttNet(self.cond_out_dim, seq_len=self.smo_win_size)

        self.grid_type = hparams["grid_type"]
        self.grid_interpolation_type = hparams["grid_interpolation_type"]
        grid_kw = dict(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=hparams["log2_hashmap_size"],
                       interpolation=self.grid_interpolation_type)
        self.position_embedder, self.position_embedding_dim = get_encoder(
            self.grid_type, input_dim=3, desired_resolution=hparams["desired_resolution"] * self.bound, **grid_kw)
        self.num_layers_ambient = hparams["num_layers_ambient"]
        self.hidden_dim_ambient = hparams["hidden_dim_ambient"]
        self.ambient_coord_dim = hparams["ambient_coord_dim"]
        self.ambient_net = MLP(self.position_embedding_dim + self.cond_out_dim, self.ambient_coord_dim, self.hidden_dim_ambient,
                               self.num_layers_ambient)
        self.ambient_embedder, self.ambient_embedding_dim = get_encoder(
            self.grid_type, input_dim=self.ambient_coord_dim, desired_resolution=hparams["desired_resolution"], **grid_kw)

        self.num_layers_sigma = hparams["num_layers_sigma"]
        self.hidden_dim_sigma = hparams["hidden_dim_sigma"]
        self.geo_feat_dim = hparams["geo_feat_dim"]
        self.sigma_net = MLP(self.position_embedding_dim + self.ambient_embedding_dim, 1 + self.geo_feat_dim, self.hidden_dim_sigma,
                             self.num_layers_sigma)

        self.num_layers_color = hparams["num_layers_color"]
        self.hidden_dim_color = hparams["hidden_dim_color"]
        self.direction_embedder, self.direction_embedding_dim = get_encoder("spherical_harmonics")
        self.color_net = MLP(self.direction_embedding_dim + self.geo_feat_dim + self.individual_embedding_dim, 3, self.hidden_dim_color,
                             self.num_layers_color)
        self.dropout = nn.Dropout(p=hparams["cond_dropout_rate"], inplace=False)

    # -- conditioning ---------------------------------------------------------------------------------------
    #: whether this class's render() hands eye_area_percent to cal_cond_feat (RADNeRFTorso does not: radnerf_torso.py:106)
    _render_passes_eye = True
    #: set by clip.ClipRenderer around a frame: that frame's row of frame_consts_rows (256 folded constants)
    _clip_cond_feat = None

    def frame_consts_rows(self, rows, cond_at, eye_at, count, index=0):
        """Everything a frame's head pass needs from its conditioning window, for `count` frames in two launches: cal_cond_feat (one workgroup per frame,
        FramePipeline.cond_feat_rows -- with the eye value only where this class's render() passes it) and the fold of the result into the first
        layers' per-frame constants (fold_rows) -> [count, 256] rows that render() takes as `_clip_cond_feat`; None when the fused kernels do not
        serve this model.  Same kernels as inside a frame: the frame is bit-equal."""
        from .frame_pipeline import supports
        if self.executor != "fused" or self.training or not supports(self):
            return None
        pipe = self.pipeline()
        if pipe.cond is None:
            return None
        use_eye = self._render_passes_eye and self.hparams.get("add_eye_blink_cond", False)
        feats = pipe.cond_feat_rows(rows, cond_at, eye_at if use_eye else None, count)
        if feats.shape[1] != pipe.head.cond_dim:
            return None
        return pipe.fold_rows(feats, self._individual_code(index))

    # -- K consecutive frames of a clip through one persistent head launch (clip.ClipRenderer, frame groups): head-only models ----------------------
    def group_supported(self, N, K, max_steps, perturb=False):
        return (self.executor == "fused" and not self.training and self._fused_ok(perturb, max_steps) and self.pipeline().group_supported(N, K, max_steps))

    def render_group(self, consts, bg_coords, poses, lm68s=None, index=0, dt_gamma=0, bg_color=None, max_steps=1024, T_thresh=1e-4, sr_noise_mode="random",
                     after_frame=None, ngp_poses=None, camera=None, **kwargs):
        """render() for K = len(consts) frames whose conditioning is given as folded constants (consts[k]: 256 values, frame_consts_rows) -- the frame loop of
        inference/genefacepp_infer.py:460-469 taken K frames at a time with a head-only model (RADNeRF, RADNeRFwithSR).  Rays: generated on the device from
        ngp_poses (K equally spaced [4, 4] cam2world views) + camera (fx, fy, cx, cy, H, W), or put by the caller into pipeline().group_workspace(N, K, max_steps)
        [2]['rays_o' / 'rays_d'].  Every frame is the bits of its own render() call (the super-resolution noise of 'random' mode is drawn per launch either way).
        after_frame(k, result dict of frame k): issued right behind the frame's last kernel.  `poses` / `lm68s` (the torso models' inputs) are not used.
        Returns the K result dicts with render()'s keys."""
        K = len(consts)
        N = int(bg_coords.reshape(-1, 2).shape[0])
        if not self.group_supported(N, K, max_steps):
            raise GfppError("render_group: not available for this model / precision / executor (see FramePipeline.group_supported)")
        sr = getattr(self, "sr_net", None)
        side = sr.input_resolution if sr is not None else None
        results = [None] * K

        def finish(k, out):
            res = {"depth_map": out["depth"].view(1, N), "rgb_map": out["image"].view(1, N, 3)}
            if sr is not None:
                rgb = out["image"].reshape(1, side, side, 3).permute(0, 3, 1, 2)
                res["rgb_map"] = rgb
                if sr.ready:
                    res["sr_rgb_map"] = sr(rgb.clone(), noise_mode=sr_noise_mode, clamp01=True, clip_sub=k)
            results[k] = res
            if after_frame is not None:
                after_frame(k, res)
        self.pipeline().render_group_head(consts, N, dt_gamma, max_steps, T_thresh, bg_color, after_frame=finish, poses=ngp_poses, camera=camera)
        return results

    def cal_cond_feat(self, cond, eye_area_percent=None):
        """cond [smo_win, t_window, cond_in] -> cond_feat [cond_out] (radnerf.py:88-106)."""
        hp = self.hparams
        if (self.executor == "fused" and not self.training and torch.is_tensor(cond) and cond.is_cuda and not torch.is_grad_enabled()
                and (eye_area_percent is None or torch.is_tensor(eye_area_percent))):
            from .frame_pipeline import supports
            if supports(self):
                pipe = self.pipeline()
                if pipe.cond is not None:
                    return pipe.cond_feat(cond, eye_area_percent if hp.get("add_eye_blink_cond", False) else None)
        feat = self._fused_train_cond_feat(cond, eye_area_percent)
        return feat if feat is not None else self._cond_feat_eager(cond, eye_area_percent)

    def _cond_feat_eager(self, cond, eye_area_percent=None):
        hp = self.hparams
        feat = self.cond_prenet(cond)
        if hp.get("add_eye_blink_cond", False):
            if eye_area_percent is None:
                eye_area_percent = torch.zeros(1, 1, dtype=feat.dtype)
            k = hp["eye_blink_dim"]
            blink = self.blink_embedding.weight[0].reshape(1, -1) * eye_area_percent.reshape(1, 1).to(feat.device)
            blink = self.blink_encoder(blink)
            feat = torch.cat([feat[..., :k] + blink.expand(feat.shape[0], k), feat[..., k:]], dim=-1)
        if self.with_att:
            feat = self.cond_att_net(feat)
        return feat

    def _cond_modules(self):
        mods = [self.cond_prenet]
        if self.hparams.get("add_eye_blink_cond", False):
            mods += [self.blink_embedding, self.blink_encoder]
        if self.with_att:
            mods.append(self.cond_att_net)
        return mods

    def _cond_train_call_ok(self, cond, eye_area_percent):
        if not (self.training and torch.is_grad_enabled() and torch.is_tensor(cond) and cond.device.type == "cuda" and not cond.requires_grad
                and (eye_area_percent is None or (torch.is_tensor(eye_area_percent) and not eye_area_percent.requires_grad))
                and not torch.cuda.is_current_stream_capturing()):
            return None
        params = [p for m in self._cond_modules() for p in m.parameters()]
        if not params or not all(p.requires_grad and p.device == cond.device for p in params):        # (the launches run on cond's device: same device index, not only "cuda")
            return None
        return params

    def _fused_train_cond_feat(self, cond, eye_area_percent):
        """A training step's cal_cond_feat as ONE forward and ONE backward launch (gfpp_cond_feat_train_forward / _backward, csrc/cond_nets.hip), fp32.  None:
        not a CUDA training call, a shape outside the one-workgroup kernels, or GFPP_TRAIN_COND=eager.  (torch's eager layers captured as two graphs --
        torch.cuda.make_graphed_callables -- were the intermediate step: 5.9 -> 5.55 ms per May step, ~100 kernel nodes still 1.3 ms of GPU time; this path: 4.6 ms.)"""
        if COND_TRAIN != "fused":
            return None
        params = self._cond_train_call_ok(cond, eye_area_percent)
        if params is None or any(p.dtype != torch.float32 or not p.is_contiguous() for p in params):
            return None
        from .frame_pipeline import cond_train_model
        desc = cond_train_model(self)
        if desc is None or tuple(cond.shape) != (desc[0].smo, desc[0].t_win, desc[0].c_in):
            return None
        eye = None
        if desc[0].blink_dim and eye_area_percent is not None:
            eye = eye_area_percent.reshape(-1)[:1].to(device=cond.device, dtype=torch.float32)
        return _CondFeatTrain.apply(cond.float().contiguous(), eye, desc, *desc[1])

    # -- per-sample evaluation (stand-alone API; render() uses the fused kernels) ----------------------------
    def _sigma_trunk(self, position, cond_feat):
        n = position.shape[0]
        pos_feat = self.position_embedder(position, bound=self.bound)
        ambient_in = torch.cat([pos_feat, cond_feat.reshape(1, -1).expand(n, -1).to(pos_feat.dtype)], dim=1)
        ambient_pos = torch.tanh(self.ambient_net(ambient_in).float())
        ambient_feat = self.ambient_embedder(ambient_pos, bound=1)
        sigma_in = torch.cat([pos_feat, ambient_feat], dim=-1)
        hp = self.sigma_net.forward_padded(sigma_in)
        if hp is not None:          # training under amp: the whole MLP as one launch per direction, its output / gradient left zero-padded (no [M, 129] pad and slice copies)
            logit, geo_feat = SplitFirstColumn.apply(hp, self.geo_feat_dim)
            return trunc_exp(logit), geo_feat, ambient_pos
        h = self.sigma_net(sigma_in)
        return trunc_exp(h[..., 0]), h[..., 1:], ambient_pos

    def _fused_eval_ok(self, position, cond_mask):
        if not (self.executor == "fused" and not self.training and not torch.is_grad_enabled() and cond_mask is None and position.is_cuda):
            return False
        from .frame_pipeline import supports
        return supports(self)

    def forward(self, position, direction, cond_feat, individual_code, cond_mask=None):
        """-> sigma [M] f32, color [M,3], ambient_pos [M, ambient_coord_dim] f32 (radnerf.py:108-141).  At inference (no grad, fused
        executor) the samples go through the trip kernels' own evaluate_block (gfpp_head_eval_samples[_lp], precision as for render())."""
        if self._fused_eval_ok(position, cond_mask):
            return self.pipeline().eval_samples(position, direction, cond_feat, individual_code)
        sigma, geo_feat, ambient_pos = self._sigma_trunk(position, cond_feat)
        parts = [self.direction_embedder(direction).to(geo_feat.dtype), geo_feat]
        if individual_code is not None:
            parts.append(individual_code.reshape(1, -1).expand(position.shape[0], -1).to(geo_feat.dtype))
        width = sum(int(t.shape[-1]) for t in parts)
        fused = self.color_net.fused_widths(geo_feat, cols=width) if geo_feat.dim() == 2 else None
        if fused is not None and fused[0] > width:
            # the zero columns up to the fused launch's input width join the concatenation that is made anyway (instead of a second pass that pads it)
            parts.append(torch.zeros(1, 1, dtype=geo_feat.dtype, device=geo_feat.device).expand(position.shape[0], fused[0] - width))
        color = torch.sigmoid(self.color_net(torch.cat(parts, dim=-1)))
        return sigma, color, ambient_pos

    def density(self, position, cond_feat, e=None, cond_mask=None):
        """-> {'sigma', 'geo_feat'} (radnerf.py:143-166)."""
        assert self.hparams.get("to_heatmap", False) is False
        sigma, geo_feat, _ = self._sigma_trunk(position, cond_feat)
        return {"sigma": sigma, "geo_feat": geo_feat}
