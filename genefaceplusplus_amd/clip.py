"""Clip renderer: the frame loop of the reference's caller (inference/genefacepp_infer.py:246-269 ray pre-materialisation,
:436-469 per-frame render + `.cpu()` + uint8 conversion), restated around the fused pipeline.

What changes against the reference's loop, and why:

* rays are generated on the device from the 64-byte pose inside each frame (gfpp_get_rays) instead of being materialised for
  the whole clip up front -- the reference keeps 6.3 MB of rays per 512x512 frame resident (3.2 GB for a 512-frame clip);
* float -> uint8 HWC happens on the device (gfpp_rgb_to_u8), so a frame leaves the GPU as 786 432 B instead of 3 MB of fp32;
* ray generation, the two NeRF passes, (super-resolution,) and the uint8 conversion are ONE captured hipGraph per frame;
* frames leave through a ring of pinned host buffers filled by asynchronous copies on a second stream: the host never blocks
  on the frame it has just launched, only on the one `ring` frames back, which is when the consumer (video writer) gets it.

The per-frame device work is untouched by this file: it calls `model.render()` with exactly the arguments the reference passes.
"""
import numpy as np
import torch

from . import frames
from ._lib import GfppError, call
from .radnerfs import camera
from .radnerfs.frame_pipeline import GraphedFrame


class ClipRenderer:
    """Renders a clip with `model` (RADNeRF / RADNeRFTorso / RADNeRFTorsowithSR / RADNeRFwithSR on a GPU, eval mode).

    H, W: ray grid (512x512, or 256x256 for the *_sr models whose output is 512x512); intrinsics (fx, fy, cx, cy);
    bg_img [1, H*W, 3] float in [0,1] or None (white), as `sample['bg_img']` in the reference.
    """

    def __init__(self, model, H, W, intrinsics, bg_img=None, T_thresh=1e-4, ring=4, use_graph=True, render_kwargs=None):
        dev = model.density_bitfield.device
        if dev.type != "cuda":
            raise GfppError("ClipRenderer: the model must live on the GPU (there is no CPU path)")
        self.model, self.H, self.W, self.device = model, H, W, dev
        self.intrinsics = tuple(float(v) for v in intrinsics)
        self.T_thresh = float(T_thresh)
        self.use_graph = use_graph
        self.render_kwargs = dict(model.hparams) if render_kwargs is None else dict(render_kwargs)
        self.with_sr = hasattr(model, "sr_net") and model.sr_net.ready
        scale = 1
        if self.with_sr:
            if H != model.sr_net.input_resolution or W != H:
                raise GfppError(f"ClipRenderer: the super-resolution models render {model.sr_net.input_resolution}^2 rays")
            scale = 2                                   # Superresolution: 256 -> 512 (radnerf_sr.py:14-43)
        self.out_hw = (H * scale, W * scale)
        self.bg_coords = camera.get_bg_coords(H, W, dev)
        self.bg_img = None if bg_img is None else bg_img.to(dev).float().reshape(1, H * W, 3).contiguous()
        self.rays_o = torch.empty(1, H * W, 3, dtype=torch.float32, device=dev)
        self.rays_d = torch.empty(1, H * W, 3, dtype=torch.float32, device=dev)
        self.frame_u8 = torch.empty(*self.out_hw, 3, dtype=torch.uint8, device=dev)
        self.ring = max(2, int(ring))
        self._dev_ring = [torch.empty_like(self.frame_u8) for _ in range(self.ring)]
        self._host_ring = [torch.empty(*self.out_hw, 3, dtype=torch.uint8).pin_memory() for _ in range(self.ring)]
        self._ready = [torch.cuda.Event() for _ in range(self.ring)]
        self._done = [torch.cuda.Event() for _ in range(self.ring)]
        self._copy_stream = torch.cuda.Stream(device=dev)
        self._graph = None

    # -- one frame of device work --------------------------------------------------------------------------------------------
    def _frame(self, pose, pose6, cond, lm68, eye):
        fx, fy, cx, cy = self.intrinsics
        call("gfpp_get_rays", pose.data_ptr(), fx, fy, cx, cy, self.H, self.W, self.rays_o.data_ptr(), self.rays_d.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        kw = dict(self.render_kwargs)
        kw.update(index=0, staged=False, bg_color=self.bg_img, lm68=lm68, perturb=False, force_all_rays=False, T_thresh=self.T_thresh,
                  eye_area_percent=eye)
        res = self.model.render(self.rays_o, self.rays_d, cond, self.bg_coords, pose6, **kw)
        if self.with_sr:
            rgb = res["sr_rgb_map"].permute(0, 2, 3, 1)        # [1,3,h,w] view of NHWC memory
        else:
            rgb = res["rgb_map"]
        frames.to_uint8_hwc(rgb.reshape(*self.out_hw, 3), self.frame_u8)
        return {"u8": self.frame_u8}

    def _launch(self, clip, i):
        """Run frame i of a prepared clip; returns the static uint8 frame buffer (overwritten by the next launch)."""
        if not self.use_graph:
            with torch.no_grad():
                return self._frame(**self._views(clip["packed"][i], clip["layout"]))["u8"]
        key = (clip["layout"], self.model.resolved_precision())
        if self._graph is None or self._graph_key != key:
            # all driving signals of a frame travel as ONE small row (a few KB): one device-to-device copy per frame feeds the graph
            self._static_in = clip["packed"][i].clone()
            views = self._views(self._static_in, clip["layout"])
            inner, self.model.use_graph = self.model.use_graph, False       # this graph already contains the model's launches
            try:
                self._graph = GraphedFrame(self._frame, views, copy_inputs=False)
            finally:
                self.model.use_graph = inner
            self._graph_key = key
        self._static_in.copy_(clip["packed"][i], non_blocking=True)
        self._graph.graph.replay()
        return self.frame_u8

    @staticmethod
    def _views(row, layout):
        out, at = {}, 0
        for name, shape in layout:
            n = int(np.prod(shape))
            out[name] = row[at:at + n].view(*shape)
            at += n + (-n) % 4                      # fields are padded to 16 bytes (see prepare)
        return out

    @staticmethod
    def prepare(batch, device):
        """Move a clip's driving signals to the device once (a few KB per frame), packed one row per frame.
        batch: 'ngp_poses' [F,4,4] (cam2world in ngp convention, i.e. dataset.poses), 'cond_wins' [F,smo,t_win,C],
        optional 'lm68' [F,136], 'eye_area_percent' [F,1,1]."""
        pose = torch.as_tensor(batch["ngp_poses"], dtype=torch.float32).to(device).contiguous()
        F = pose.shape[0]
        cond = torch.as_tensor(batch["cond_wins"], dtype=torch.float32).to(device)
        lm = batch.get("lm68")
        lm = (torch.zeros(F, 136) if lm is None else torch.as_tensor(lm, dtype=torch.float32).reshape(F, 136)).to(device)
        eye = batch.get("eye_area_percent")
        eye = (torch.zeros(F, 1, 1) if eye is None else torch.as_tensor(eye, dtype=torch.float32).reshape(F, 1, 1)).to(device)
        parts = [("pose", pose, (4, 4)), ("pose6", camera.convert_poses(pose), (1, 6)), ("cond", cond, tuple(cond.shape[1:])),
                 ("lm68", lm, (136,)), ("eye", eye, (1, 1))]
        # each field starts on a 16-byte boundary (the kernels read some of them with wide loads)
        cols, layout = [], []
        for name, t, shape in parts:
            flat = t.reshape(F, -1)
            pad = (-flat.shape[1]) % 4
            cols.append(torch.nn.functional.pad(flat, (0, pad)))
            layout.append((name, shape, flat.shape[1] + pad))
        packed = torch.cat(cols, dim=1).contiguous()
        return {"packed": packed, "layout": tuple((n, s) for n, s, _ in layout), "strides": tuple(w for _, _, w in layout), "frames": F}

    # -- public API -----------------------------------------------------------------------------------------------------------
    def render_to_device(self, clip, frame_indices=None, out=None):
        """Render frames (all, or the given indices) into a uint8 stack [F,h,w,3] that stays on the GPU (the multi-GPU path
        gathers these with frames.gather_clip).  No host synchronisation."""
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        if out is None:
            out = torch.empty(len(idx), *self.out_hw, 3, dtype=torch.uint8, device=self.device)
        for k, i in enumerate(idx):
            out[k].copy_(self._launch(clip, i), non_blocking=True)
        return out

    def render_to_host(self, clip, sink=None, frame_indices=None):
        """Render frames and hand each one to `sink(frame_index, uint8 ndarray [h,w,3])` in order, as soon as its copy has landed
        in pinned memory (the array is only valid during the call -- the ring slot is reused).  With sink=None the frames are
        collected and returned as one ndarray [F,h,w,3].  This is the PCIe-inclusive path."""
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        collected = None
        if sink is None:
            collected = np.empty((len(idx), *self.out_hw, 3), np.uint8)

            def sink(k, arr, _pos=[0]):
                collected[_pos[0]] = arr
                _pos[0] += 1
        main = torch.cuda.current_stream()

        def retire(k):
            slot = k % self.ring
            self._done[slot].synchronize()
            sink(idx[k], self._host_ring[slot].numpy())

        for k, i in enumerate(idx):
            slot = k % self.ring
            if k >= self.ring:
                retire(k - self.ring)
            u8 = self._launch(clip, i)
            self._dev_ring[slot].copy_(u8, non_blocking=True)
            self._ready[slot].record(main)
            self._copy_stream.wait_event(self._ready[slot])
            with torch.cuda.stream(self._copy_stream):
                self._host_ring[slot].copy_(self._dev_ring[slot], non_blocking=True)
                self._done[slot].record(self._copy_stream)
        for k in range(max(0, len(idx) - self.ring), len(idx)):
            retire(k)
        return collected


def render_clip_distributed(renderer, clip, n_frames=None, interleaved=False, group=None):
    """Frame-parallel clip over the ranks of `group`: every rank renders its shard into a device uint8 stack and
    the one exchange step is the all_gather of those stacks (RCCL over xGMI).  Returns the whole clip [F,h,w,3] on every rank."""
    import torch.distributed as dist
    F = clip["frames"] if n_frames is None else n_frames
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return renderer.render_to_device(clip, range(F))
    mine = frames.shard_frames(F, dist.get_rank(group), dist.get_world_size(group), interleaved)
    local = renderer.render_to_device(clip, mine)
    return frames.gather_clip(local, F, interleaved, group)
