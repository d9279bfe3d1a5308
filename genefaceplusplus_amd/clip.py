"""Clip renderer: the frame loop of the reference's caller (inference/genefacepp_infer.py:246-269 ray pre-materialisation,
:436-469 per-frame render + `.cpu()` + uint8 conversion), restated around the fused pipeline.

What changes against the reference's loop, and why:

* rays are generated on the device from the 64-byte pose inside each frame (gfpp_get_rays) instead of being materialised for
  the whole clip up front -- the reference keeps 6.3 MB of rays per 512x512 frame resident (3.2 GB for a 512-frame clip);
* float -> uint8 HWC happens on the device (gfpp_rgb_to_u8), so a frame leaves the GPU as 786 432 B instead of 3 MB of fp32;
* ray generation, the two NeRF passes, (super-resolution,) and the uint8 conversion are ONE captured hipGraph per frame;
* frames leave through a ring of pinned host buffers filled by asynchronous copies on a second stream: the host never blocks
  on the frame it has just launched, only on the one `ring` frames back, which is when the consumer (video writer) gets it.

* the loop itself runs below Python (round 3): a frame's graph fetches its own row of driving signals and stores its own uint8 frame through a
  device-side cursor (gfpp_clip_job), so a frame needs nothing from the host but its graph launch, and those are issued from C for a whole run
  of frames (gfpp_graph_replay) -- the reference's `for i in range(num_frames)` costs ~0.2 ms of Python per frame next to 0.3-0.4 ms of GPU time.

The per-frame device work is untouched by this file: it calls `model.render()` with exactly the arguments the reference passes.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, frames, tuning
from ._lib import GfppError, call
from .radnerfs import camera
from .radnerfs.frame_pipeline import GraphedFrame, shared_stream


class ClipJob(ctypes.Structure):
    """gfpp_clip_job (include/gfpp_radnerf.h): the device-resident record through which a frame's graph finds its inputs and its output slot."""
    _fields_ = [("packed", ctypes.c_void_p), ("order", ctypes.c_void_p), ("out", ctypes.c_void_p), ("frame_bytes", ctypes.c_uint64),
                ("row_floats", ctypes.c_uint32), ("n", ctypes.c_uint32), ("lanes", ctypes.c_uint32), ("ring_frames", ctypes.c_uint32),
                ("cursor", ctypes.c_uint32 * 8), ("ticket", ctypes.c_uint32 * 8)]


_lib.register("gfpp_clip_fetch", [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p])
_lib.register("gfpp_clip_store_u8", [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p])
_lib.register("gfpp_clip_fetch_at", [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p])
_lib.register("gfpp_clip_fetch_group", [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p])
_lib.register("gfpp_clip_store_u8_at", [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p])
_lib.register("gfpp_graph_replay", [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                    ctypes.c_uint32])
STRUCT_MIRRORS = {"clip_job": ClipJob}


class ClipRenderer:
    """Renders a clip with `model` (RADNeRF / RADNeRFTorso / RADNeRFTorsowithSR / RADNeRFwithSR on a GPU, eval mode).

    H, W: ray grid (512x512, or 256x256 for the *_sr models whose output is 512x512); intrinsics (fx, fy, cx, cy);
    bg_img [1, H*W, 3] float in [0,1] or None (white), as `sample['bg_img']` in the reference.
    """

    def __init__(self, model, H, W, intrinsics, bg_img=None, T_thresh=1e-4, ring=4, use_graph=True, render_kwargs=None, lanes=None,
                 calibrate_trips=True, group=None):
        """lanes: how many frames are in flight at once.  With 2, consecutive frames alternate between two streams (each with its own
        workspace and graph; weights and tables are shared), so one frame's prologue and epilogue (slab test + pre-march, conditioning nets, torso
        pass, uint8 store) overlap the tail of the other frame's head pass.  None = 2 for the 16-bit modes (the head pass is one launch that holds
        every CU: a third frame only queues behind it), 3 for the exact-fp32 mode (one launch per trip) and for the super-resolution models (numbers
        at the assignment below).
        calibrate_trips (trip-launch paths only: lp_kernel='trips', fp32_kernel='tile'): with several lanes every possible trip of the render loop is a launch of
        its own (gfpp_frame_ws.separate_trips); when a lane's graph is captured, the trips beyond the ones its warm-up frame needed (+ 1) are given
        a small grid, because a launch that finds nothing left still needs a whole CU per workgroup (results never depend on it, a later frame that
        needs more trips is rendered by the small grid).
        group: frames per graph launch (round 4).  K > 1: a lane takes K consecutive frames of the clip at a time and renders them with ONE persistent
        head launch (RADNeRF*.render_group, gfpp_frame_ws.n_frames) -- every frame the bits of its own launch, the fixed costs of a launch paid once
        per K frames.  None: 4 (measured, frames/s with K = 1 / 2 / 3 / 4: 256^2 rays + SR 4 806 / 5 107 / 5 205 / 5 286 -- a workgroup's share of ONE
        such frame is ~110 occupied rays, 27 sample blocks for 8 wavefronts --; 512^2 3 684 / 3 778 / 3 861 / 3 846; a 20-frame job 3 090 / - / 2 920 / 3 140);
        models / precisions without group support (fp32, lp_kernel='trips', more than 2^22 rays per group) and clips without precomputed
        conditioning render frame by frame whatever is asked."""
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
            # measured, 512^2 frames: round 2 (one launch per trip) 2 300 / 2 530 / 2 010 frames/s with 2 / 3 / 4 lanes; round 3 (the head pass is ONE launch
            # that holds every CU for ~0.29 ms) 2 535 / 2 956 / 2 769 / 2 693 with 1 / 2 / 3 / 4, and 2 980 / 2 995 with 2 / 3 once the conditioning left the
            # frames: a third frame's launches only queue behind the head pass.  The super-resolution models (256^2 rays: a 0.12 ms head pass + four SR
            # launches) do gain from a third frame: 3 860 -> 4 050 frames/s
            # (round 4, frame groups of 4: 512^2 3 836 / 3 839 / 3 852 frames/s with 2 / 3 / 4 lanes -- still nothing to gain; SR models 4 822 / 5 153 / 5 375)
            lanes = 4 if self.with_sr else (3 if getattr(model, "precision", "auto") == "fp32" else 2)
        self.lanes = max(1, int(lanes)) if fused else 1        # the staged executor synchronises with the host every trip: nothing to overlap
        self._lane = [{"rays_o": torch.empty(1, H * W, 3, dtype=torch.float32, device=dev),
                       "rays_d": torch.empty(1, H * W, 3, dtype=torch.float32, device=dev),
                       "u8": torch.empty(*self.out_hw, 3, dtype=torch.uint8, device=dev),
                       "stream": shared_stream(dev, "lane", _i) if self.lanes > 1 else None,
                       "graph": None, "key": None, "static_in": None} for _i in range(self.lanes)]
        if group is None:
            group = tuning.HOST["clip_group"]
        self.group_wanted = max(1, min(int(group), 4)) if fused else 1
        self.group = 1                                       # what the captured graphs render with (decided with the first clip: _ensure_graphs)
        self.ring = max(2, int(ring), self.lanes)
        self._copy_stream = shared_stream(dev, "copy")
        # the job record (gfpp_clip_job) in device memory + a small ring of pinned staging copies (a staging slot is rewritten only after the
        # asynchronous upload that read it has completed)
        _lib.check_struct("clip_job", ClipJob)
        self._job_dev = torch.zeros(ctypes.sizeof(ClipJob), dtype=torch.uint8, device=dev)
        self._job_stage = [torch.zeros(ctypes.sizeof(ClipJob), dtype=torch.uint8).pin_memory() for _ in range(8)]
        self._job_stage_ev = [None] * 8
        self._order_stage, self._order_stage_ev = [None] * 8, [None] * 8
        self._job_calls = 0
        self._job = None                          # host-side state of the running job
        self._host_bufs = None

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

    def _frame(self, lane, pose, pose6, cond, lm68, eye, cond_feat=None):
        """One frame on the current stream: the job's next row of this lane -> the lane's static input -> rays -> model.render() -> uint8 into
        the job's output slot (the views `pose` .. `eye` are views of the lane's static input; `cond_feat`: the frame's precomputed conditioning row)."""
        L = self._lane[lane]
        st = torch.cuda.current_stream().cuda_stream
        call("gfpp_clip_fetch", self._job_dev.data_ptr(), lane, L["static_in"].data_ptr(), int(L["static_in"].numel()), st)
        fx, fy, cx, cy = self.intrinsics
        call("gfpp_get_rays", pose.data_ptr(), fx, fy, cx, cy, self.H, self.W, L["rays_o"].data_ptr(), L["rays_d"].data_ptr(), st)
        kw = dict(self.render_kwargs)
        kw.update(index=0, staged=False, bg_color=self.bg_img, lm68=lm68, perturb=False, force_all_rays=False, T_thresh=self.T_thresh,
                  eye_area_percent=eye)
        pipe = self.model.pipeline() if getattr(self.model, "executor", "fused") == "fused" else None
        if pipe is not None and not self.with_sr:
            pipe.clip_job, pipe.clip_job_consumed = (self._job_dev.data_ptr(), lane), False       # the torso kernel stores the uint8 frame itself when it can
        sr = getattr(getattr(self.model, "_orig_mod", self.model), "sr_net", None) if self.with_sr and tuning.HOST["sr_store_u8"] else None
        if sr is not None:
            sr.clip_store, sr.clip_consumed = (self._job_dev.data_ptr(), lane, 1, self.lanes), False   # ... and so does the SR stage's last layer
        target = getattr(self.model, "_orig_mod", self.model)     # (a torch.compile wrapper keeps attributes set on it to itself)
        target._clip_cond_feat = cond_feat
        try:
            res = self.model.render(L["rays_o"], L["rays_d"], cond, self.bg_coords, pose6, **kw)
        finally:
            target._clip_cond_feat = None
            stored = pipe is not None and pipe.clip_job_consumed
            if pipe is not None:
                pipe.clip_job, pipe.clip_job_consumed = None, False
            if sr is not None:
                stored = stored or sr.clip_consumed
                sr.clip_store, sr.clip_consumed = None, False
        if stored:
            return {}
        if self.with_sr:
            rgb = res["sr_rgb_map"].permute(0, 2, 3, 1)        # [1,3,h,w] view of NHWC memory
        else:
            rgb = res["rgb_map"]
        rgb = rgb.reshape(*self.out_hw, 3)
        if not rgb.is_contiguous() or rgb.dtype != torch.float32:
            rgb = rgb.float().contiguous()
        call("gfpp_clip_store_u8", self._job_dev.data_ptr(), lane, rgb.data_ptr(), int(rgb.numel()), torch.cuda.current_stream().cuda_stream)
        return {}

    def _frame_group(self, lane, rows):
        """K frames on the current stream (rows: the views of the lane's K static input rows): per frame fetch + rays, then model.render_group -- ONE
        persistent head launch for all K -- with each frame's uint8 store issued right behind its last kernel."""
        L = self._lane[lane]
        K = len(rows)
        st = torch.cuda.current_stream().cuda_stream
        model = getattr(self.model, "_orig_mod", self.model)
        kw = dict(self.render_kwargs)
        max_steps = int(kw.get("max_steps", 1024))
        _g, _f, t = model.pipeline().group_workspace(self.rays_per_frame, K, max_steps)
        fx, fy, cx, cy = self.intrinsics
        # the K rows of driving signals in one launch; the rays are generated inside the group's prologue launch (gfpp_head_group_begin)
        call("gfpp_clip_fetch_group", self._job_dev.data_ptr(), lane, K, L["g_static_in"].data_ptr(), int(L["g_static_in"].shape[1]), st)

        pipe = model.pipeline()
        if not self.with_sr:
            pipe.clip_job, pipe.clip_job_consumed = (self._job_dev.data_ptr(), lane, self.lanes), False      # (GFPP_FUSE_TAIL: the torso kernel may store the uint8 frames itself)

        sr = getattr(model, "sr_net", None) if self.with_sr and tuning.HOST["sr_store_u8"] else None
        if sr is not None:
            sr.clip_store, sr.clip_consumed = (self._job_dev.data_ptr(), lane, K, self.lanes), False

        def store(k, res):
            if pipe.clip_job_consumed or (sr is not None and sr.clip_consumed):
                return
            rgb = res["sr_rgb_map"].permute(0, 2, 3, 1) if self.with_sr else res["rgb_map"]
            rgb = rgb.reshape(*self.out_hw, 3)
            if not rgb.is_contiguous() or rgb.dtype != torch.float32:
                rgb = rgb.float().contiguous()
            advance = K * self.lanes if k == K - 1 else 0xFFFFFFFF
            call("gfpp_clip_store_u8_at", self._job_dev.data_ptr(), lane, k, advance, rgb.data_ptr(), int(rgb.numel()), torch.cuda.current_stream().cuda_stream)
        kw.pop("index", None)
        kw.update(bg_color=self.bg_img, T_thresh=self.T_thresh)
        try:
            model.render_group([v["cond_feat"] for v in rows], self.bg_coords, [v["pose6"] for v in rows], [v["lm68"] for v in rows], index=0, after_frame=store,
                               ngp_poses=[v["pose"] for v in rows], camera=(fx, fy, cx, cy, self.H, self.W), **kw)
        finally:
            pipe.clip_job, pipe.clip_job_consumed = None, False
            if sr is not None:
                sr.clip_store, sr.clip_consumed = None, False
        return {}

    # -- the job: which frames, where to ----------------------------------------------------------------------------------------------------------
    def _upload_job(self, job):
        slot = self._job_calls % len(self._job_stage)
        self._job_calls += 1
        if self._job_stage_ev[slot] is not None:
            self._job_stage_ev[slot].synchronize()
        stage = self._job_stage[slot]
        ctypes.memmove(stage.data_ptr(), ctypes.addressof(job), ctypes.sizeof(ClipJob))
        self._job_dev.copy_(stage, non_blocking=True)
        ev = self._job_stage_ev[slot] = torch.cuda.Event()
        ev.record()

    def _upload_order(self, idx):
        """The job's frame order as an int32 device tensor, through a small ring of REUSED pinned staging buffers (a fresh `pin_memory()` per job is a host
        allocation of pinned memory -- the slowest single step of start() -- and a short job's start-up is a twentieth of the driver's 20-frame line)."""
        n = len(idx)
        slot = self._job_calls % len(self._job_stage)
        st = self._order_stage[slot]
        if st is None or st.numel() < n:
            st = self._order_stage[slot] = torch.empty(max(256, 2 * n), dtype=torch.int32).pin_memory()
        ev = self._order_stage_ev[slot]
        if ev is not None:
            ev.synchronize()                         # the asynchronous upload that last read this buffer has completed
        st[:n] = torch.as_tensor(idx, dtype=torch.int32)
        order = torch.empty(n, dtype=torch.int32, device=self.device)
        order.copy_(st[:n], non_blocking=True)
        ev = self._order_stage_ev[slot] = torch.cuda.Event()
        ev.record()
        return order

    def _ensure_graphs(self, clip):
        """Capture the lanes' frame graphs (first use, or another clip layout / precision).  The warm-up runs and the capture execute the frame with
        an EMPTY job (n = 0: fetch and store do nothing, the static input keeps a copy of the clip's first row), so they cannot disturb a job."""
        target = getattr(self.model, "_orig_mod", self.model)
        group = self.group_wanted
        if group > 1 and not (any(n == "cond_feat" for n, _ in clip["layout"]) and hasattr(target, "group_supported")
                              and target.group_supported(self.rays_per_frame, group, int(self.render_kwargs.get("max_steps", 1024)))):
            group = 1
        key = (clip["layout"], self.model.resolved_precision(), group)
        todo = [k for k, L in enumerate(self._lane) if L["static_in"] is None or L["key"] != key or (self.use_graph and L["graph"] is None)]
        self.group = group
        if not todo:
            return
        dry = ClipJob()
        dry.lanes, dry.ring_frames = self.lanes, 1
        self._upload_job(dry)
        inner, self.model.use_graph = self.model.use_graph, False           # a frame here is one graph of its own (or plain launches)
        try:
            for lane in todo:
                L = self._lane[lane]
                L["static_in"] = clip["packed"][0].clone()
                L["views"] = self._views(L["static_in"], clip["layout"])
                L["key"] = key
                L["graph"] = None
                if group > 1:
                    # K rows of driving signals per launch; a row's fields are views of it (the frames' folded constants: equally spaced, one per row)
                    L["g_static_in"] = clip["packed"][:1].repeat(group, 1).contiguous()
                    L["g_views"] = [self._views(L["g_static_in"][k], clip["layout"]) for k in range(group)]
                if self.use_graph:
                    self._enter_lane(lane)
                    try:
                        if group > 1:
                            g = GraphedFrame(lambda **_v: self._frame_group(lane, L["g_views"]), {"rows": L["g_static_in"]}, copy_inputs=False)
                        else:
                            g = GraphedFrame(lambda **v: self._frame(lane, **v), L["views"], copy_inputs=False,
                                             before_capture=(lambda: self._calibrate_trip_launches()) if self.calibrate else None)
                    finally:
                        self._leave_lane()
                    g.fn = None           # only needed for the capture; keeping it would tie the renderer into a reference cycle, and a cycle
                    L["graph"] = g        # is freed by the garbage collector at a random time -- destroying a hipGraph during someone's capture fails
        finally:
            self.model.use_graph = inner
        torch.cuda.current_stream().synchronize()
        self._execs = None

    def start(self, clip, frame_indices, out, ring_frames=None):
        """Begin a job: the frames `frame_indices` of `clip` (any order, repeats allowed) are rendered into `out` [ring_frames or len, h, w, 3] uint8
        (device), position k of the job into slot k % ring_frames.  Everything queued on the caller's stream so far is visible to the frames.
        Follow with issue(count) -- any number of times, until all frames are issued -- and join()."""
        idx = list(frame_indices)
        if out.dtype != torch.uint8 or not out.is_contiguous() or tuple(out.shape[1:]) != (*self.out_hw, 3):
            raise GfppError(f"ClipRenderer.start: out must be a contiguous uint8 [F, {self.out_hw[0]}, {self.out_hw[1]}, 3] stack")
        ring_given = ring_frames is not None and int(ring_frames) < len(idx)
        ring_frames = len(idx) if ring_frames is None else int(ring_frames)
        if out.shape[0] < min(ring_frames, max(len(idx), 1)):
            raise GfppError("ClipRenderer.start: out is smaller than the job's ring")
        if self._job is not None and self._job["issued"]:
            # frames of the previous job may still be running on the lanes (a caller that never joined): the new job record, its rows and its
            # output ring must not reach the device before they are done.  A stream-level wait, no host synchronisation
            self._join(torch.cuda.current_stream())
        target = getattr(self.model, "_orig_mod", self.model)
        if hasattr(target, "modules"):
            from .radnerfs.frame_pipeline import fingerprint_memo
            with fingerprint_memo(target):
                clip = self._with_cond_features(clip)
                self._ensure_graphs(clip)
        else:
            clip = self._with_cond_features(clip)
            self._ensure_graphs(clip)
        order = self._upload_order(idx if idx else [0])
        job = ClipJob()
        job.packed, job.order, job.out = clip["packed"].data_ptr(), order.data_ptr(), out.data_ptr()
        job.frame_bytes = self.out_hw[0] * self.out_hw[1] * 3
        job.row_floats, job.n, job.lanes, job.ring_frames = int(clip["packed"].shape[1]), len(idx), self.lanes, max(ring_frames, 1)
        if job.frame_bytes != out[0].numel():
            raise GfppError("ClipRenderer.start: a slot of `out` is not one frame")
        for l in range(8):
            job.cursor[l] = l * self.group                  # groups of `group` consecutive positions are dealt to the lanes round-robin
        self._upload_job(job)
        main = self._fork()
        self._job = {"n": len(idx), "issued": 0, "order": order, "out": out, "main": main, "clip": clip, "ring_given": ring_given}       # (keeps the extended rows alive)
        return self

    #: conditioning features of all frames in one launch at the start of a job instead of 16 dependent layers inside every frame (GFPP_CLIP_PRECOND=0: per frame)
    precompute_cond = tuning.HOST["clip_precond"]

    def _with_cond_features(self, clip):
        """The clip with one more field per row: the 256 per-frame constants the head pass gets from the frame's conditioning window (cal_cond_feat +
        fold; RADNeRF.frame_consts_rows: two launches for the whole clip, the bits of the per-frame kernels).  Computed at every start() on the caller's
        stream -- ~50 us, and the weights may have changed since the last job."""
        fn = getattr(self.model, "frame_consts_rows", None)
        if not self.precompute_cond or fn is None or clip["frames"] == 0:
            return clip
        # a caller that starts one job per chunk of the same clip (render_to_host in a loop, bench's chunked gather) gets the extended rows from a one-entry
        # cache: the constants depend on the clip's rows and on the weights (parameter versions), nothing else -- recomputing them per start() is O(F) work
        # and a full copy of the rows per chunk, O(F^2 / chunk) per clip
        target = getattr(self.model, "_orig_mod", self.model)
        precision = getattr(self.model, "resolved_precision", None)
        from .radnerfs.frame_pipeline import FramePipeline
        weights = FramePipeline._fingerprint(target) if hasattr(target, "modules") else ()
        stamp = (clip["packed"].data_ptr(), clip["packed"]._version, tuple(clip["packed"].shape), precision() if callable(precision) else None, weights)
        hit = getattr(self, "_cond_cache", None)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        at, cols = 0, {}
        for (name, _shape), width in zip(clip["layout"], clip["strides"]):
            cols[name] = at
            at += width
        with torch.no_grad():
            feats = fn(clip["packed"], cols["cond"], cols["eye"], clip["frames"])
        if feats is None:
            return clip
        width = feats.shape[1]
        pad = (-width) % 4
        if pad:
            feats = torch.nn.functional.pad(feats, (0, pad))
        ext = {"packed": torch.cat([clip["packed"], feats], dim=1).contiguous(), "layout": clip["layout"] + (("cond_feat", (width,)),),
               "strides": clip["strides"] + (width + pad,), "frames": clip["frames"], "_source": clip["packed"]}      # (_source keeps the keyed tensor's address alive)
        self._cond_cache = (stamp, ext)
        return ext

    def _exec_arrays(self):
        if getattr(self, "_execs", None) is None:
            execs = (ctypes.c_void_p * self.lanes)(*[int(L["graph"].graph.raw_cuda_graph_exec()) for L in self._lane])
            streams = (ctypes.c_void_p * self.lanes)(*[(L["stream"].cuda_stream if L["stream"] is not None else torch.cuda.current_stream().cuda_stream)
                                                      for L in self._lane])
            self._execs = (execs, streams)
        return self._execs

    #: (the SR stage's random noise is drawn inside its kernels since round 3: no graph of this renderer uses torch's generator, so every graph can be
    #: launched from C -- torch's own replay() would only add the generator bookkeeping)
    replay_mode = tuning.HOST["clip_replay"]        # 'c' | 'python' (experiments)
    max_ahead = tuning.HOST["clip_max_ahead"]   # frames per lane the issuing thread may queue ahead of the GPU (0: no limit)

    def _replay_from_c(self):
        return self.replay_mode == "c" and self.use_graph and hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec")

    _warned_round_up = False

    @property
    def chunk_multiple(self):
        """Frames per graph launch once a job has started (1 before): issue() works in whole multiples of this."""
        return self.group

    def issue(self, count=None):
        """Issue the next `count` frames of the job (all that are left by default): frame k runs on lane k % lanes -- with frame groups (self.group = K > 1)
        the K frames [g K, g K + K) are one launch on lane g % lanes, a `count` that is no multiple of K is rounded up to whole groups (the number of frames really
        issued is returned), and a last, partial group renders its missing positions for nothing (they fetch and store nothing).  No host synchronisation."""
        J = self._job
        count = J["n"] - J["issued"] if count is None else min(int(count), J["n"] - J["issued"])
        if count <= 0:
            return 0
        K = self.group
        # whole frame groups only: a count that is no multiple of K is rounded UP (the caller gets at least what it asked for and the return value / the next call
        # account for the overshoot; `chunk_multiple` tells callers that size their buffers per chunk what to round to)
        launches, first = -(-count // K), (J["issued"] // K) % self.lanes
        asked, count = count, min(launches * K, J["n"] - J["issued"])
        if count != asked and J.get("ring_given") and not ClipRenderer._warned_round_up:
            # (round-5 advisory) a caller that sized its own ring to its chunks gets up to K - 1 more frames stored per call than it asked for
            import warnings
            ClipRenderer._warned_round_up = True
            warnings.warn(f"ClipRenderer.issue({asked}): rounded up to {count} (whole groups of {K} frames per launch); a job started with its own ring_frames must "
                          f"leave room for the extra frames or issue multiples of chunk_multiple = {K}", RuntimeWarning, stacklevel=2)
        if self._replay_from_c():
            execs, streams = self._exec_arrays()
            if self.lanes == 1:
                streams[0] = torch.cuda.current_stream().cuda_stream
            call("gfpp_graph_replay", execs, streams, self.lanes, first, launches, int(self.max_ahead))
        else:
            inner, self.model.use_graph = self.model.use_graph, False
            try:
                for k in range(launches):
                    lane = (first + k) % self.lanes
                    with self._on_lane(lane):
                        if self.use_graph:
                            self._lane[lane]["graph"].graph.replay()
                        else:
                            self._enter_lane(lane)
                            with torch.no_grad():
                                if K > 1:
                                    self._frame_group(lane, self._lane[lane]["g_views"])
                                else:
                                    self._frame(lane, **self._lane[lane]["views"])
            finally:
                self._leave_lane()
                self.model.use_graph = inner
        J["issued"] += count
        return count

    def join(self):
        """The caller's stream waits for every frame issued so far (the lanes keep running: more frames may be issued afterwards)."""
        self._join(torch.cuda.current_stream())

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
    def render_to_device(self, clip, frame_indices=None, out=None, after_caller_stream=None):
        """Render frames (all, or the given indices) into a uint8 stack [F,h,w,3] that stays on the GPU (the multi-GPU path
        gathers these with frames.gather_clip).  No host synchronisation; the stack is complete in the caller's stream order.
        (= start + issue + join; callers that exchange finished chunks while later frames render use those three directly.)
        `after_caller_stream` (rounds 1-2) is gone: the lanes always start behind everything queued on the caller's stream."""
        if after_caller_stream is not None:
            import warnings
            warnings.warn("ClipRenderer.render_to_device: after_caller_stream is ignored since round 3 (the lanes always wait for the caller's stream)",
                          DeprecationWarning, stacklevel=2)
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        if out is None:
            out = torch.empty(len(idx), *self.out_hw, 3, dtype=torch.uint8, device=self.device)
        if not idx:
            return out
        self.start(clip, idx, out)
        self.issue()
        self.join()
        return out

    def render_to_host(self, clip, sink=None, frame_indices=None, chunk=16):
        """Render frames and hand each one to `sink(frame_index, uint8 ndarray [h,w,3])` in order, once its chunk has landed in pinned memory (the
        array is only valid during the call -- the buffer is reused two chunks later).  With sink=None the frames are collected and returned as one
        ndarray [F,h,w,3].  This is the PCIe-inclusive path: frames are stored into a device ring of two chunks, a finished chunk leaves with ONE
        asynchronous copy on a stream of its own while the next chunk renders, and the consumer runs while both are in flight."""
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        collected = None
        if sink is None:
            collected = np.empty((len(idx), *self.out_hw, 3), np.uint8)

            def sink(k, arr, _pos=[0]):
                collected[_pos[0]] = arr
                _pos[0] += 1
        if not idx:
            return collected
        M = max(1, int(chunk))
        M = -(-M // self.group_wanted) * self.group_wanted          # whole frame groups per chunk (harmless when the clip falls back to single frames)
        if self._host_bufs is None or self._host_bufs["M"] != M:
            self._host_bufs = {"M": M, "ring": torch.empty(2 * M, *self.out_hw, 3, dtype=torch.uint8, device=self.device),
                               "host": [torch.empty(M, *self.out_hw, 3, dtype=torch.uint8).pin_memory() for _ in range(2)],
                               "copied": [torch.cuda.Event(), torch.cuda.Event()], "rendered": [torch.cuda.Event() for _ in range(self.lanes)]}
        B = self._host_bufs
        self.start(clip, idx, B["ring"], ring_frames=2 * M)
        bounds = [(c, min(c + M, len(idx))) for c in range(0, len(idx), M)]

        def deliver(c):
            lo, hi = bounds[c]
            B["copied"][c & 1].synchronize()
            if self._barriers_possible():
                self.check()                     # BEFORE the chunk reaches the sink: a frame of it may lack trips (sticky word, FramePipeline.check_barriers)
            host = B["host"][c & 1].numpy()
            for k in range(lo, hi):
                sink(idx[k], host[k - lo])

        for c, (lo, hi) in enumerate(bounds):
            buf = c & 1
            if c >= 2:                                   # this half of the device ring is free once its previous chunk has left
                for L in self._lane:
                    (L["stream"] if L["stream"] is not None else torch.cuda.current_stream()).wait_event(B["copied"][buf])
            self.issue(hi - lo)
            for L, ev in zip(self._lane, B["rendered"]):
                st = L["stream"] if L["stream"] is not None else torch.cuda.current_stream()
                ev.record(st)
                self._copy_stream.wait_event(ev)
            with torch.cuda.stream(self._copy_stream):
                B["host"][buf][:hi - lo].copy_(B["ring"][buf * M:buf * M + hi - lo], non_blocking=True)
                B["copied"][buf].record(self._copy_stream)
            if c >= 1:
                deliver(c - 1)
        deliver(len(bounds) - 1)
        self.join()
        self.check()
        return collected

    def _barriers_possible(self):
        """Only the trip-launch path of the 16-bit modes has device-wide barriers (lp_kernel='trips', or max_steps > 24 / more than 2^22 rays)."""
        if getattr(self.model, "executor", "fused") != "fused":
            return False
        pipe = self.model.pipeline()
        return pipe.precision != "fp32" and (pipe.lp_kernel != "persist" or int(self.render_kwargs.get("max_steps", 1024)) > 24 or self.rays_per_frame > (1 << 22))      # (render()'s own default)

    def check(self):
        """After the frames of a job have completed: raise GfppError if any of them was rendered by a launch whose device-wide barrier timed out
        (FramePipeline.check_barriers; only the trip-launch path has one).  render_to_host calls it before returning; callers of render_to_device /
        start-issue-join call it once per clip, after their own synchronisation."""
        if getattr(self.model, "executor", "fused") == "fused":
            self.model.pipeline().check_barriers()


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
