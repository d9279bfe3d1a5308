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
from .radnerfs.frame_pipeline import GraphedFrame, shared_stream


class ClipRenderer:
    """Renders a clip with `model` (RADNeRF / RADNeRFTorso / RADNeRFTorsowithSR / RADNeRFwithSR on a GPU, eval mode).

    H, W: ray grid (512x512, or 256x256 for the *_sr models whose output is 512x512); intrinsics (fx, fy, cx, cy);
    bg_img [1, H*W, 3] float in [0,1] or None (white), as `sample['bg_img']` in the reference.
    """

    def __init__(self, model, H, W, intrinsics, bg_img=None, T_thresh=1e-4, ring=4, use_graph=True, render_kwargs=None, lanes=None,
                 calibrate_trips=True):
        """lanes: how many frames are in flight at once.  With 2, consecutive frames alternate between two streams (each with its own
        workspace and graph; weights and tables are shared), so one frame's prologue (slab test, pre-march, conditioning nets: small
        launches that leave most CUs idle) and its late, sparse trips overlap the other frame's full-width launches.  None = 3
        (round 2: +10 % over two lanes at 512^2, +12 % for the 256^2 frames of the super-resolution models; a fourth lane loses 13 %).
        calibrate_trips: with several lanes every possible trip of the render loop is a launch of its own (gfpp_frame_ws.separate_trips); when a
        lane's graph is captured, the trips beyond the ones its warm-up frame needed (+ 1) are given a small grid, because a launch that finds
        nothing left still needs a whole CU per workgroup (+5 % frames/s at 512^2; results never depend on it, a later frame that needs more
        trips is rendered by the small grid)."""
        dev = model.density_bitfield.device
        if dev.type != "cuda":
            raise GfppError("ClipRenderer: the model must live on the GPU (there is no CPU path)")
        self.model, self.H, self.W, self.device = model, H, W, dev
        self.rays_per_frame = H * W
        self.calibrate = bool(calibrate_trips)
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
        # computed on the host like the reference's dataset does (dataset_utils.py:240, `.cuda()` afterwards): the GPU's division by a host scalar is a
        # multiplication by the reciprocal, 1 ulp away, which the torso field turns into single-LSB differences of the uint8 frame
        self.bg_coords = camera.get_bg_coords(H, W, "cpu").to(dev)
        self.bg_img = None if bg_img is None else bg_img.to(dev).float().reshape(1, H * W, 3).contiguous()
        fused = getattr(model, "executor", "fused") == "fused"
        if lanes is None:
            lanes = 3        # measured (round 2, after the trip kernel got faster): 512^2 frames 2 300 / 2 530 / 2 010 frames/s with 2 / 3 / 4 lanes
        self.lanes = max(1, int(lanes)) if fused else 1        # the staged executor synchronises with the host every trip: nothing to overlap
        self._lane = [{"rays_o": torch.empty(1, H * W, 3, dtype=torch.float32, device=dev),
                       "rays_d": torch.empty(1, H * W, 3, dtype=torch.float32, device=dev),
                       "u8": torch.empty(*self.out_hw, 3, dtype=torch.uint8, device=dev),
                       "stream": shared_stream(dev, "lane", _i) if self.lanes > 1 else None,
                       "graph": None, "key": None, "static_in": None} for _i in range(self.lanes)]
        self.ring = max(2, int(ring), self.lanes)
        self._dev_ring = [torch.empty(*self.out_hw, 3, dtype=torch.uint8, device=dev) for _ in range(self.ring)]
        self._host_ring = [torch.empty(*self.out_hw, 3, dtype=torch.uint8).pin_memory() for _ in range(self.ring)]
        self._ready = [torch.cuda.Event() for _ in range(self.ring)]
        self._done = [torch.cuda.Event() for _ in range(self.ring)]
        self._copy_stream = shared_stream(dev, "copy")

    # -- one frame of device work --------------------------------------------------------------------------------------------
    def _enter_lane(self, lane):
        """Point the model's per-frame mutable state (pipeline workspace, side stream, SR activations) at this lane."""
        if getattr(self.model, "executor", "fused") == "fused":
            pipe = self.model.pipeline()
            pipe.lane, pipe.frames_in_flight = lane, self.lanes
        if self.with_sr:
            self.model.sr_net.lane = lane

    def _leave_lane(self):
        if getattr(self.model, "executor", "fused") == "fused":
            pipe = self.model.pipeline()
            pipe.lane, pipe.frames_in_flight = 0, 1
        if self.with_sr:
            self.model.sr_net.lane = 0

    def _frame(self, lane, pose, pose6, cond, lm68, eye):
        L = self._lane[lane]
        fx, fy, cx, cy = self.intrinsics
        call("gfpp_get_rays", pose.data_ptr(), fx, fy, cx, cy, self.H, self.W, L["rays_o"].data_ptr(), L["rays_d"].data_ptr(),
             torch.cuda.current_stream().cuda_stream)
        kw = dict(self.render_kwargs)
        kw.update(index=0, staged=False, bg_color=self.bg_img, lm68=lm68, perturb=False, force_all_rays=False, T_thresh=self.T_thresh,
                  eye_area_percent=eye)
        res = self.model.render(L["rays_o"], L["rays_d"], cond, self.bg_coords, pose6, **kw)
        if self.with_sr:
            rgb = res["sr_rgb_map"].permute(0, 2, 3, 1)        # [1,3,h,w] view of NHWC memory
        else:
            rgb = res["rgb_map"]
        frames.to_uint8_hwc(rgb.reshape(*self.out_hw, 3), L["u8"])
        return {"u8": L["u8"]}

    def _launch(self, clip, i, lane=0):
        """Issue frame i of a prepared clip on the CURRENT stream with lane `lane`'s buffers; returns that lane's static uint8 frame
        buffer (overwritten by the lane's next launch)."""
        L = self._lane[lane]
        inner, self.model.use_graph = self.model.use_graph, False           # a frame here is one graph of its own (or plain launches)
        self._enter_lane(lane)
        try:
            if not self.use_graph:
                with torch.no_grad():
                    return self._frame(lane, **self._views(clip["packed"][i], clip["layout"]))["u8"]
            key = (clip["layout"], self.model.resolved_precision())
            if L["graph"] is None or L["key"] != key:
                # all driving signals of a frame travel as ONE small row (a few KB): one device-to-device copy per frame feeds the graph
                L["static_in"] = clip["packed"][i].clone()
                views = self._views(L["static_in"], clip["layout"])
                L["graph"] = GraphedFrame(lambda **v: self._frame(lane, **v), views, copy_inputs=False,
                                          before_capture=(lambda: self._calibrate_trip_launches()) if self.calibrate else None)
                L["graph"].fn = None     # only needed for the capture; keeping it would tie the renderer into a reference cycle, and a cycle
                L["key"] = key           # is freed by the garbage collector at a random time -- destroying a hipGraph during someone's capture fails
            L["static_in"].copy_(clip["packed"][i], non_blocking=True)
            L["graph"].graph.replay()
            return L["u8"]
        finally:
            self._leave_lane()
            self.model.use_graph = inner

    def _calibrate_trip_launches(self):
        """Between the warm-up frames and the capture of a lane's graph: the trips beyond the ones the warm-up frame needed (+ 1) get a small grid
        (FramePipeline.calibrate_trip_launches; results never depend on it)."""
        pipe = self.model.pipeline()
        if getattr(pipe, "calibrate_trip_launches", None) is not None and self.lanes > 1:
            pipe.calibrate_trip_launches(self.rays_per_frame)

    def _fork(self):
        """Lane streams start after everything queued on the caller's stream."""
        main = torch.cuda.current_stream()
        for L in self._lane:
            if L["stream"] is not None:
                L["stream"].wait_stream(main)
        return main

    def _join(self, main):
        for L in self._lane:
            if L["stream"] is not None:
                main.wait_stream(L["stream"])

    def _on_lane(self, lane):
        st = self._lane[lane]["stream"]
        return torch.cuda.stream(st) if st is not None else _NullContext()

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
    def render_to_device(self, clip, frame_indices=None, out=None, after_caller_stream=True):
        """Render frames (all, or the given indices) into a uint8 stack [F,h,w,3] that stays on the GPU (the multi-GPU path
        gathers these with frames.gather_clip).  No host synchronisation; the stack is complete in the caller's stream order.
        after_caller_stream=False: the frames do not wait for work queued on the caller's stream (use it for the second and later chunks
        of a clip rendered chunk by chunk, so that frames keep overlapping across chunk boundaries; `clip` and `out` must already exist)."""
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        if out is None:
            out = torch.empty(len(idx), *self.out_hw, 3, dtype=torch.uint8, device=self.device)
        main = self._fork() if after_caller_stream else torch.cuda.current_stream()
        for k, i in enumerate(idx):
            lane = k % self.lanes
            with self._on_lane(lane):
                out[k].copy_(self._launch(clip, i, lane), non_blocking=True)
        self._join(main)
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

        def retire(k):
            slot = k % self.ring
            self._done[slot].synchronize()
            sink(idx[k], self._host_ring[slot].numpy())

        main = self._fork()
        # the device->host copies need a stream (and a hardware queue) of their own: with it, two rendering lanes are the measured optimum
        # (1 900-2 100 frames/s delivered; three lanes + the copy stream: 1 250-1 400)
        host_lanes = min(self.lanes, 2)
        for k, i in enumerate(idx):
            slot, lane = k % self.ring, k % host_lanes
            if k >= self.ring:
                retire(k - self.ring)
            with self._on_lane(lane):
                st = torch.cuda.current_stream()
                st.wait_event(self._done[slot])             # the slot's previous device->host copy has drained (no-op the first time round)
                self._dev_ring[slot].copy_(self._launch(clip, i, lane), non_blocking=True)
                self._ready[slot].record(st)
            self._copy_stream.wait_event(self._ready[slot])
            with torch.cuda.stream(self._copy_stream):
                self._host_ring[slot].copy_(self._dev_ring[slot], non_blocking=True)
                self._done[slot].record(self._copy_stream)
        for k in range(max(0, len(idx) - self.ring), len(idx)):
            retire(k)
        self._join(main)
        return collected


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def render_clip_distributed(renderer, clip, n_frames=None, interleaved=False, group=None, dst=None):
    """Frame-parallel clip over the ranks of `group`: every rank renders its shard into a device uint8 stack and
    the one exchange step is the gather of those stacks (RCCL over xGMI).  dst=None: all_gather, the whole clip [F,h,w,3] on every rank;
    dst=r: gather to the writer rank r only (None elsewhere) -- what a single video writer needs (genefacepp_infer.py:454-518)."""
    import torch.distributed as dist
    F = clip["frames"] if n_frames is None else n_frames
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return renderer.render_to_device(clip, range(F))
    mine = frames.shard_frames(F, dist.get_rank(group), dist.get_world_size(group), interleaved)
    local = renderer.render_to_device(clip, mine)
    return frames.gather_clip(local, F, interleaved, group, dst=dst)
