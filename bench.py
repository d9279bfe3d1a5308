#!/usr/bin/env python
"""bench.py -- rendered 512x512 head+torso frames/s of the GeneFace++ motion2video NeRF path on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame (one pass of the hot path over one frame of synthetic driving input) per rank.  Frames are
independent units, so ranks render disjoint frames with no data-path collective ("weak" scaling: per-GPU work fixed); the
only exchange is the RCCL gather of the finished uint8 frames to the writer rank (--gather all: all_gather), inside the timed region.  Inputs (rays, conditioning
windows, poses, background) are resident in HBM before the timed region starts, exactly like the reference keeps them
(inference/genefacepp_infer.py:246-275).  Rank 0 prints ONE JSON line.

The headline `value` is BASELINE.json configs[2] literally: "May head+torso two-pass render, 1 MI355X, bf16 MLP, hipGraph-captured
per-frame" (--precision bf16, the default: MLP layers on bf16 MFMA operands with fp32 accumulation).  `modes` carries the same measurement
for fp16 (what the reference's own torch.autocast(fp16) inference computes in) and for the exact-fp32 parity mode, plus a >= 2 000-frame
run with mean +- std (`modes.long_run`); `configs` carries the other single-GPU BASELINE configurations: configs[1] (May head-only, fp32,
single-frame latency p50 / p99) and configs[0] (64x64 crop, 1 024 rays per step, on the CPU oracle).

Extra objects in the JSON line:
  roofline      dominant kernel = the fused head trip kernel (sample fetch + 2 grid encodes + MLPs + composite).
                16-bit modes: bound "hbm" -- the north star's yardstick for the hash-grid stage; achieved = evaluated samples x the gather
                bytes the mode REQUESTS (SURVEY 8d's formula at the precision of the tables read: 1 036 B with the 16-bit corner-block
                tables, 2 060 B for hash-addressed models on fp32 tables) / time of the head launches (HIP events on the launch stream)
                vs the 8 TB/s HBM peak; `frac_fp32_equiv` = the same launches at 2 060 B (the unit of rounds 1-3, continuity only);
                `traffic` = fabric bytes per launch from the FETCH_SIZE PMC pass committed under profiles/;
                the MFMA fraction (128 768 FLOP/sample vs 2.5 PFLOP/s) is reported beside it.
                fp32 mode: bound "mfma", 161 536 FLOP/sample vs 157.3 TFLOP/s (fp32-input MFMA, exact fp32)
  modes         frames/s of the other precision modes (short runs of the same pipeline)
  grid_stage    the stand-alone hash/tiled-grid kernel on 2^22 uniform points: achieved = B x 1164 B / t vs 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (kind "port": the reference has no CPU path) on a bounded sample, rank 0 at N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_parts import cpu_baseline, spawn_ranks, dist_evidence, ranks_ok, run_identities, run_ray_tiles   # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--hw", type=int, default=512, help="frame side (rays = hw*hw)")
    ap.add_argument("--variant", default="may_torso", choices=["may_head", "may_torso", "may_torso_sr"])
    ap.add_argument("--executor", default="fused", choices=["fused", "staged"])
    ap.add_argument("--precision", default="bf16", choices=["fp32", "fp16", "bf16"],
                    help="arithmetic of the MLP layers (head + torso): 16-bit MFMA operands with fp32 accumulation (fp16 = what the reference's "
                         "autocast inference computes in, bf16 = BASELINE config 3), or exact-fp32 MFMA (the parity mode)")
    ap.add_argument("--no-modes", action="store_true", help="skip the short runs of the other two precision modes")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python instead of replaying the per-frame hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid-stage", action="store_true")
    ap.add_argument("--gather-every", type=int, default=5, help="multi-GPU: all_gather the finished uint8 frames every this many frames, "
                                                                "overlapped with the rendering of the next chunk")
    ap.add_argument("--lanes", type=int, default=None, help="frames in flight per GPU (default: 2 in the 16-bit modes since the head pass is ONE launch, 3 in fp32): consecutive frames alternate between this many streams, each with "
                                                         "its own workspace and hipGraph (weights / tables shared), so one frame's small prologue launches and "
                                                         "sparse late trips overlap the other's full-width launches; 1 = strictly one frame at a time")
    ap.add_argument("--gather", default="writer", choices=["writer", "all"],
                    help="multi-GPU exchange step: 'writer' = finished frames go to rank 0 only (the reference has ONE consumer, the video writer, "
                         "genefacepp_infer.py:454-518), 'all' = all_gather to every rank")
    ap.add_argument("--identities", type=int, default=1,
                    help="BASELINE configs[4]: this many person-specific models at once; the ranks are split into contiguous blocks (frames.identity_groups), "
                         "frame-parallel inside a block, driving signals broadcast once.  With --gpus 1 the identities share the one GPU")
    ap.add_argument("--shard", default="frames", choices=["frames", "rays"],
                    help="multi-GPU partition: 'frames' = frame-parallel (the default: independent frames, no data-path collective, weak scaling); 'rays' = "
                         "latency mode, ALL ranks render each frame together as ray tiles (one int32 all_reduce per trip for the frame-wide alive count + "
                         "an all_gather of the tiles per frame; strong scaling of one frame)")
    ap.add_argument("--shard-order", default="contiguous", choices=["contiguous", "interleaved"],
                    help="multi-GPU frame partition: 'contiguous' = rank r renders the block [r F / G, (r + 1) F / G) of the clip (SURVEY 8e, DESIGN section 6: keeps video order, "
                         "one gather per chunk reassembles the clip), 'interleaved' = frame i on rank i %% G")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[0] / configs[1] entries")
    ap.add_argument("--long-run-frames", type=int, default=2000, help="frames of modes.long_run (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (control-flow checks of the N>1 path on one GPU)")
    ap.add_argument("--ckpt-dir", default=None, help="a checkpoint directory in the reference's layout (model_ckpt_steps_*.ckpt + config.yaml, utils/commons/ckpt_utils.py:29-76, "
                                                     "e.g. checkpoints/motion2video_nerf/may_torso): bench THESE weights instead of the synthetic ones (SURVEY 8d, last bullet)")
    ap.add_argument("--data-dir", default=None, help="with --ckpt-dir: the directory holding trainval_dataset.npy (data/binary/videos/May, tasks/radnerfs/dataset_utils.py:160-296): "
                                                     "poses, landmarks, background and intrinsics of the clip come from it; without it the synthetic driving signals are used")
    ap.add_argument("--trained", default=None, choices=["plain", "sr"],
                    help="bench the TRAINED procedural field instead of the random-init one: tests/golden/trained/may_torso[_sr].npz (tools/make_trained_checkpoint.py) is "
                         "written out as a checkpoint directory in the reference's layout and loaded through --ckpt-dir's path; poses, conditioning windows, landmarks "
                         "and background are the procedural clip's own")
    ap.add_argument("--no-trained", action="store_true", help="skip the configs.trained_* entries of the default run")
    ap.add_argument("--ckpt-parity", type=int, default=2, help="with --ckpt-dir: this many frames are also rendered by the CPU oracle from the same weights and compared (0 = skip)")
    return ap.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    if world > 1 and args.dist_backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: RCCL needs one GPU per rank, this node shows {torch.cuda.device_count()} "
                         f"(--dist-backend gloo lets the ranks share a GPU for control-flow checks)")
    local_dev = local_rank % torch.cuda.device_count()          # (== local_rank on a real node; lets the gloo check share one GPU)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    dinfo = dist_evidence(dev, args.dist_backend) if world > 1 else None

    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    from genefaceplusplus_amd.radnerfs import camera
    from genefaceplusplus_amd import radnerfs, frames
    from genefaceplusplus_amd.configs import CLASSES

    if args.identities > 1:
        return run_identities(args, rank, world, dev, dinfo)
    if args.shard == "rays" and world > 1:
        return run_ray_tiles(args, rank, world, dev, dinfo)

    pclip = None
    if args.trained:
        import tempfile
        from genefaceplusplus_amd.procedural import ProceduralClip
        tv = "may_torso_sr" if args.trained == "sr" else "may_torso"
        fixture = os.path.join(ROOT, "tests", "golden", "trained", tv + ".npz")
        args.ckpt_dir = os.path.join(tempfile.mkdtemp(prefix="gfpp_trained_"), tv)
        syn.write_checkpoint(args.ckpt_dir, tv, may_hparams(tv), steps=6500, state_dict=syn.load_compact_state(fixture))
        pclip = ProceduralClip(T=256, seed=0)
    HW, K, W = args.hw, args.steps, args.warmup
    real = None              # --ckpt-dir: {"ckpt": path, "dataset": RADNeRFDataset | None}
    if args.ckpt_dir:
        # a real checkpoint in the reference's layout (last bullet): the loader's own steps -- flat config.yaml (utils/commons/hparams.py:167-170),
        # newest model_ckpt_steps_*.ckpt, state_dict['model'], strict load (utils/commons/ckpt_utils.py:29-76, genefacepp_infer.py:163-191)
        import yaml
        with open(os.path.join(args.ckpt_dir, "config.yaml")) as f:
            hp = yaml.safe_load(f)
        sd_t, ckpt_path = syn.read_checkpoint(args.ckpt_dir, model_name="model")
        args.variant = "may_torso_sr" if hp.get("with_sr") else "may_torso"
        HW = args.hw = 256 if hp.get("with_sr") else 512
        sd = {k: v.numpy() for k, v in sd_t.items()}
        real = {"ckpt": ckpt_path, "dataset": None}
    else:
        hp = may_hparams(args.variant)
        sd = syn.synthetic_state_dict(hp, args.variant)
        if args.variant == "may_torso_sr":
            sd = dict(sd)
            sd.update(syn.synthetic_sr_state())
    N = HW * HW
    if args.variant == "may_torso_sr":
        assert HW == 256, "the *_sr models render 256x256 rays"
    model = getattr(radnerfs, CLASSES[args.variant])(hp)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    model.executor = args.executor
    model.precision = args.precision
    model.use_graph = not args.no_graph and args.executor == "fused"

    # ---- this rank's frames (frame-parallel sharding): a contiguous block of the clip per rank (--shard-order interleaved: global frame = step * world + rank) ----
    total = K + W
    my_frames = frames.shard_frames(total * world, rank, world, interleaved=(args.shard_order == "interleaved"))
    intr = syn.intrinsics_for(HW, HW)
    bg_coords = camera.get_bg_coords(HW, HW, "cpu").to(dev)          # host-computed like the reference dataset (dataset_utils.py:240)
    bg_color = torch.full((1, N, 3), 0.5, device=dev)
    fi_all = [syn.synthetic_frame_inputs(hp, fidx) for fidx in my_frames]
    batch = {"ngp_poses": np.stack([syn.synthetic_pose(fidx) for fidx in my_frames]).astype(np.float32),
             "cond_wins": np.stack([f["cond"] for f in fi_all]), "lm68": np.stack([f["lm68"] for f in fi_all]),
             "eye_area_percent": np.stack([f["eye_area_percent"] for f in fi_all])}
    if pclip is not None:
        batch = pclip.clip_batch(hp["smo_win_size"], my_frames)
        bg_color = pclip.background_image(HW, dev)
        fi_all = [{"cond": batch["cond_wins"][j], "lm68": batch["lm68"][j].reshape(-1), "eye_area_percent": np.asarray(batch["eye_area_percent"][j]).reshape(1, 1)}
                  for j in range(len(my_frames))]
    if real is not None and args.data_dir:
        # the clip's own poses / landmarks / eye values / background / intrinsics (tasks/radnerfs/dataset_utils.py:160-296), frames cycled to the bench length
        from genefaceplusplus_amd.dataset import RADNeRFDataset
        ds = RADNeRFDataset("trainval", dict(hp, binary_data_dir=os.path.dirname(os.path.abspath(args.data_dir))), data_dir=args.data_dir, training=False, device=dev,
                            allow_bfm68_fallback=True)
        real["dataset"] = ds
        assert (ds.H, ds.W) == (HW, HW), f"the dataset's frames are {ds.H}x{ds.W}, the model renders {HW}x{HW} rays"
        full = ds.clip_batch()
        take = [f % full["ngp_poses"].shape[0] for f in my_frames]
        batch = {k: np.asarray(v.cpu() if torch.is_tensor(v) else v)[take] for k, v in full.items()}
        intr = tuple(float(v) for v in ds.intrinsics)
        bg_color = ds.bg_img.reshape(1, -1, 3).to(dev).float()
        batch.setdefault("lm68", np.zeros((len(take), 136), np.float32))
        batch.setdefault("eye_area_percent", np.zeros((len(take), 1, 1), np.float32))
        fi_all = [{"cond": batch["cond_wins"][j], "lm68": batch["lm68"][j].reshape(-1), "eye_area_percent": np.asarray(batch["eye_area_percent"][j]).reshape(1, 1)}
                  for j in range(len(take))]
    # the clip renderer = the caller's frame loop (genefacepp_infer.py:246-269, 460-469): pose -> rays on the device -> model.render() -> uint8 HWC
    # on the device, one hipGraph per frame.  The driving signals of all frames are resident in HBM before the timed region starts.
    from genefaceplusplus_amd.clip import ClipRenderer
    cr = ClipRenderer(model, HW, HW, intr, bg_img=bg_color, T_thresh=0.01, use_graph=model.use_graph, lanes=args.lanes)
    clip = cr.prepare(batch, dev)
    HWO = cr.out_hw[0]                                          # the *_sr models render 256^2 rays and super-resolve to 512^2
    out_u8 = torch.empty(K, HWO, HWO, 3, dtype=torch.uint8, device=dev)
    # multi-GPU: finished frames are all_gathered in chunks while the next chunk renders (RCCL runs on its own stream)
    chunk = max(1, min(K, args.gather_every)) if world > 1 else K
    chunk = -(-chunk // cr.group_wanted) * cr.group_wanted          # whole frame groups per chunk (ClipRenderer.issue)

    def chunk_bounds(n):
        """[begin, end) of a job's exchange chunks.  The last chunk's exchange has nothing to hide behind, so it is made as small as the frame loop allows: ONE frame
        group -- the un-hidable tail of a job is the transfer of `group_wanted` frames per rank whatever the job's length (reported as gather_tail_frames)."""
        if world == 1:
            return [(0, n)]
        tail = min(n, cr.group_wanted)
        body = [(c, min(c + chunk, n - tail)) for c in range(0, n - tail, chunk)]
        return body + [(n - tail, n)]

    def receive_buffers(bnds):
        if world > 1 and args.gather == "all":
            return [torch.empty(world * (e - b), HWO, HWO, 3, dtype=torch.uint8, device=dev) for b, e in bnds]
        if world > 1 and rank == 0:          # the writer rank receives one stack per rank and chunk; nobody else receives anything
            return [[torch.empty(e - b, HWO, HWO, 3, dtype=torch.uint8, device=dev) for _ in range(world)] for b, e in bnds]
        return None
    bounds = chunk_bounds(K)
    gathered = receive_buffers(bounds)

    def exchange(c, b, e, async_op, stack=None, recv=None):
        """The one exchange step of the frame-parallel clip: chunk c's finished frames leave for the writer (or for everybody)."""
        stack = out_u8 if stack is None else stack
        recv = gathered if recv is None else recv
        if args.gather == "all":
            return dist.all_gather_into_tensor(recv[c], stack[b:e], async_op=async_op)
        return dist.gather(stack[b:e], recv[c] if rank == 0 else None, dst=0, async_op=async_op)

    def run_job(frame_idx, stack, bnds, recv, alone=False):
        """ONE timed job: barrier + synchronise, start the clip job, per chunk issue / join / (async) exchange, wait for the exchanges, synchronise, barrier.
        Returns this rank's figures; `alone`: no exchange and no barriers (rank 0 rendering by itself while the others wait: the efficiency denominator)."""
        torch.cuda.synchronize()
        if world > 1 and not alone:
            dist.barrier()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        pend = []
        cr._cond_cache = None                    # the conditioning networks + constant fold of the timed frames run INSIDE the timed region (no cached outputs from the warm-up)
        cr.start(clip, frame_idx, stack)         # ONE job: every frame's graph finds its inputs / output slot through the device-side cursor
        for c, (b, e) in enumerate(bnds):
            cr.issue(e - b)                       # the frame loop proper: graph launches issued from C (gfpp_graph_replay), lanes round-robin
            cr.join()                             # caller's stream waits for these frames; the lanes go on with the next chunk
            if world > 1 and not alone:
                pend.append(exchange(c, b, e, True, stack, recv))
        t_iss = time.perf_counter() - t_begin    # host time to queue every frame (no synchronisation yet): the launch-rate ceiling of the frame loop
        ev_r, ev_g = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev_r.record()                             # caller's stream, after the join of the lanes: every frame of this rank is rendered
        for work in pend:
            work.wait()
        ev_g.record()                             # ... and every chunk's exchange has completed (what the rendering did not hide = exposed)
        torch.cuda.synchronize()
        t_loc = time.perf_counter() - t_begin    # this rank's own time (before the closing barrier)
        if world > 1 and not alone:
            dist.barrier()
        torch.cuda.synchronize()
        return {"elapsed": time.perf_counter() - t_begin, "local": t_loc, "issue": t_iss, "exposed_ms": ev_r.elapsed_time(ev_g)}

    # reference-shaped per-frame API with pre-materialised rays (what genefacepp_infer.py calls today): used by `modes` below
    inputs = []
    for j in range(min(len(my_frames), W + 4)):
        pose = torch.from_numpy(batch["ngp_poses"][j]).to(dev)[None]
        rays = camera.get_rays(pose, intr, HW, HW)
        inputs.append({"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "poses": camera.convert_poses(pose),
                       "cond": torch.from_numpy(fi_all[j]["cond"]).to(dev), "lm68": torch.from_numpy(fi_all[j]["lm68"]).to(dev),
                       "eye": torch.from_numpy(fi_all[j]["eye_area_percent"]).to(dev)})
    scratch_u8 = torch.empty(HWO, HWO, 3, dtype=torch.uint8, device=dev)

    def render(i, slot=None, timed=False):
        x = inputs[i % len(inputs)]
        with torch.no_grad():
            res = model.render(x["rays_o"], x["rays_d"], x["cond"], bg_coords, x["poses"], index=i, staged=False, bg_color=bg_color,
                               lm68=x["lm68"], perturb=False, force_all_rays=False, T_thresh=0.01, eye_area_percent=x["eye"], **hp)
        rgb = res["rgb_map"]
        if args.variant == "may_torso_sr":
            rgb = res["sr_rgb_map"].permute(0, 2, 3, 1)              # [1,3,512,512] view of NHWC memory
        if slot is not None:
            frames.to_uint8_hwc(rgb.reshape(HWO, HWO, 3), scratch_u8)
        return res

    # warm-up: W frames, and one collective of the timed size so that RCCL's lazy channel set-up is not inside the timed region
    # (the W warm-up frames are cycled until every lane has replayed its frame-group graph once: with W = 5 and groups of four frames only lane 0 would have, and the
    # first replay of the other lanes' graphs -- instantiation, code upload -- would fall into the timed region; `config.warmup_frames_rendered`)
    n_warm = max(W, cr.lanes * cr.group_wanted) if W > 0 else 0
    warm_idx = [i % W for i in range(n_warm)]
    cr.render_to_device(clip, warm_idx, out=out_u8[:n_warm] if n_warm <= K else None)
    torch.cuda.synchronize()
    t_warm = time.perf_counter()                  # (after the first pass: that one captures the graphs)
    # ... and until the GPU has rendered for WARM_MS: a handful of frames (1-2 ms) end before the part has left its idle power state, and the K timed frames would be
    # rendered on the clock ramp (measured: 20 timed frames after 5 / 40 / 200 warm-up frames = 3 410 / 3 595 / 3 750 frames/s; a clip of 2 000 frames: 4 280)
    WARM_MS = 60.0
    while W > 0 and n_warm < 1024:
        torch.cuda.synchronize()
        if 1e3 * (time.perf_counter() - t_warm) >= WARM_MS:
            break
        cr.render_to_device(clip, warm_idx, out=out_u8[:len(warm_idx)] if len(warm_idx) <= K else None)
        n_warm += len(warm_idx)
    gather_note = None
    if world > 1:
        try:
            exchange(0, bounds[0][0], bounds[0][1], False)
            torch.cuda.synchronize()
            ok = torch.ones(1, device=dev)
        except Exception as exc:                     # a backend without gather() for device tensors (gloo): every rank must take the same path
            ok = torch.zeros(1, device=dev)
            gather_note = f"gather-to-writer unavailable on this backend ({type(exc).__name__}); fell back to all_gather"
        if args.gather == "writer":
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                args.gather = "all"
                gathered = receive_buffers(bounds)
                gather_note = gather_note or "gather-to-writer failed on another rank; fell back to all_gather"
                exchange(0, bounds[0][0], bounds[0][1], False)
        if len(bounds) > 1:                            # ... and one of the tail chunk's size
            exchange(len(bounds) - 1, bounds[-1][0], bounds[-1][1], False)
    if hasattr(model, "sr_net"):
        model.sr_net.reseed(20260930)            # every lane's in-kernel noise restarts here: the timed frames' noise fields are reproducible (timed_frames_check)
    job = run_job(range(W, W + K), out_u8, bounds, gathered)          # barrier + synchronise | EXACTLY the K timed frames | synchronise + barrier
    elapsed, t_local, t_issue = job["elapsed"], job["local"], job["issue"]
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
        per_rank[rank] = K / t_local
        dist.all_reduce(per_rank)
        exposed = torch.tensor([job["exposed_ms"]], dtype=torch.float64, device=dev)
        dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
        frame_bytes = HWO * HWO * 3
        dinfo.update({"gather": args.gather, "gather_chunks": len(bounds), "frames_per_chunk_and_rank": chunk, "gather_tail_frames": bounds[-1][1] - bounds[-1][0],
                      "gathered_MB": round(world * K * frame_bytes / 1e6, 2),       # bytes that arrived at the writer (gather 'all': at every rank)
                      "gather_ms_exposed": round(float(exposed.item()), 3),
                      "per_rank_fps": [round(float(v), 2) for v in per_rank.tolist()]})
        if gathered is not None:                   # the writer (gather 'all': every rank) checks that it really holds every rank's frames
            got = gathered[-1] if args.gather == "all" else torch.cat(gathered[-1])
            b, e = bounds[-1]
            dinfo["writer_holds_own_frames"] = bool(torch.equal(got[rank * (e - b):(rank + 1) * (e - b)], out_u8[b:e]))
            dinfo["writer_frames_nonzero_per_rank"] = [bool(got[r * (e - b):(r + 1) * (e - b)].any().item()) for r in range(world)]
        # a 20-step window per rank (4.4 ms at 512^2) also holds the barrier skew, RCCL's per-operation latency and the tail chunk: the steady-state figure of the
        # same job shape, >= 2 000 frames per rank in blocks of 100, next to it -- with rank 0's own rate alone (no exchange, the other ranks waiting) as the
        # denominator of the weak-scaling efficiency, measured in the same process minutes apart
        if args.long_run_frames > 0:
            per = max(cr.group_wanted * 2, (min(100, args.long_run_frames) // cr.group_wanted) * cr.group_wanted)
            n_blocks = max(1, args.long_run_frames // per)
            avail = list(range(W, len(my_frames)))
            idx = (avail * (per // len(avail) + 1))[:per]
            lr_stack = torch.empty(per, HWO, HWO, 3, dtype=torch.uint8, device=dev)
            lr_bounds = chunk_bounds(per)
            lr_recv = receive_buffers(lr_bounds)
            run_job(idx, lr_stack, lr_bounds, lr_recv)                # one untimed block: the receive buffers' first touch, RCCL channels of this message size
            tot = {"elapsed": 0.0, "local": 0.0, "exposed_ms": 0.0}
            blocks = []
            for _ in range(n_blocks):
                j = run_job(idx, lr_stack, lr_bounds, lr_recv)
                for k in tot:
                    tot[k] += j[k]
                blocks.append(j["elapsed"])
            agg = torch.tensor([tot["elapsed"], tot["exposed_ms"]], dtype=torch.float64, device=dev)
            dist.all_reduce(agg, op=dist.ReduceOp.MAX)
            pr = torch.zeros(world, dtype=torch.float64, device=dev)
            pr[rank] = n_blocks * per / tot["local"]
            dist.all_reduce(pr)
            alone = 0.0
            if rank == 0:                                             # the same blocks by rank 0 alone (no exchange); the others wait at the barrier below
                run_job(idx, lr_stack, lr_bounds, lr_recv, alone=True)
                alone = sum(run_job(idx, lr_stack, lr_bounds, lr_recv, alone=True)["elapsed"] for _ in range(n_blocks))
            dist.barrier()
            value_lr = world * n_blocks * per / float(agg[0].item())
            dinfo["long_run"] = {"value": round(value_lr, 2), "unit": "frames/s", "frames_per_rank": n_blocks * per, "block_frames": per, "blocks": n_blocks,
                                 "per_rank_fps": [round(float(v), 2) for v in pr.tolist()], "gather_ms_exposed_per_block": round(float(agg[1].item()) / n_blocks, 3),
                                 "gather_chunks_per_block": len(lr_bounds), "gather_tail_frames": lr_bounds[-1][1] - lr_bounds[-1][0],
                                 "rank0_alone_fps": round(n_blocks * per / alone, 2) if rank == 0 else None,
                                 "efficiency_vs_rank0_alone": round(value_lr / (world * n_blocks * per / alone), 4) if rank == 0 else None,
                                 "what": "the timed job's shape (barrier | start, issue / join / async exchange per chunk, wait | barrier) per block of frames, blocks back to back; "
                                         "value = all ranks' frames / the slowest rank's summed block times"}
            del lr_stack, lr_recv
        # (c) a multi-rank line whose ranks were not all there, or shared devices under RCCL, is not a measurement
        dinfo["ranks_ok"] = ranks_ok(dinfo, world, args.dist_backend)
    elapsed = float(t_max.item())

    result = None
    if rank == 0:
        fps = world * K / elapsed
        if dinfo is not None and not dinfo["ranks_ok"]:
            fps = float("nan")
        result = {"metric": "rendered frames/sec at 512x512 (head+torso)", "value": (round(fps, 3) if fps == fps else None), "unit": "frames/s", "n_gpus": world,
                  "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
                  "vs_baseline": None, "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16 (ambient_net f16)"}[args.precision],
                  "dtype_note": "MLP layers on MFMA: f32 = exact-fp32 MFMA; f16 / bf16 = 16-bit operands with fp32 accumulation; marcher, grid interpolation, "
                                "activations' transcendental parts and compositing are fp32 in every mode.  bf16 mode: ambient_net (two wide layers + three rows, 48 of a "
                                "block's 148 MFMAs) multiplies f16 operands -- its output is a coordinate of the second hash grid, which 8-bit significands displace by up to "
                                "five cells of the finest level (tools/lp_emulate.py); sigma_net, the merged geo / colour layer, colour rows and the torso MLPs are bf16",
                  "data": "synthetic",
                  "config": {"workload": f"{args.variant}: May-shaped head+torso NeRF, {HW}x{HW} = {N} rays/frame"
                                         + (" + StyleGAN2 super-resolution to 512x512 (random noise inputs, like the reference)" if args.variant == "may_torso_sr" else "")
                                         + ", max_steps 16, T_thresh 0.01, "
                                         f"random-init weights of the May architecture (seed 9999), ellipsoid occupancy, synthetic poses/landmarks",
                             "frames_per_gpu": K, "shard_order": args.shard_order, "parallelism": f"frame-parallel x{world}" + ((" + RCCL " + ("gather to the writer rank" if args.gather == "writer" else "all_gather")
                                                                           + f" of uint8 frames every {chunk} frames, overlapped with rendering") if world > 1 else ""),
                             "frame_loop": "genefaceplusplus_amd.clip.ClipRenderer: at the start of the timed job the conditioning networks and the constant fold of all its "
                                           "frames (two launches, inside the timed region); then per frame ONE graph launch issued from C (gfpp_graph_replay); inside the "
                                           "graph: fetch the frame's row of driving signals and folded constants by a device-side cursor -> rays on device -> model.render() "
                                           "-> uint8 HWC into the output stack",
                             "frames_in_flight": cr.lanes, "host_issue_ms_per_frame": round(1e3 * t_issue / K, 4),
                             "warmup_frames_rendered": n_warm,
                             "warmup_note": f"the {W} warm-up frames, cycled through every lane's graph and repeated until the GPU has rendered for 60 ms (idle power state left "
                                            "behind); the timed region is exactly the K frames of `steps`",
                             **({"gather_note": gather_note} if gather_note else {}),
                             **({"dist": dinfo} if dinfo else {}),
                             "executor": args.executor,
                             "launch": "hipGraph replay per frame" if model.use_graph else "eager"}}

    # ---- everything else the line carries: tools/bench_parts.py (importable sub-benchmarks over this run's state) ---------------------------------------------------
    parts = None
    if rank == 0:
        import types
        from tools import bench_parts
        parts = bench_parts.build(types.SimpleNamespace(args=args, dev=dev, world=world, rank=rank, model=model, hp=hp, inputs=inputs, render=render, HW=HW, HWO=HWO, N=N,
                                                        K=K, W=W, cr=cr, clip=clip, out_u8=out_u8, my_frames=my_frames, bg_coords=bg_coords, bg_color=bg_color, intr=intr,
                                                        batch=batch, fi_all=fi_all))
        parts.timed_frames_check(result)                 # the timed frames themselves: rendered, and rendered RIGHT (a failed check sets `value` to null)

    if rank == 0 and real is not None:
        result["data"] = "real checkpoint" + (" + real driving signals" if real["dataset"] is not None else " + synthetic driving signals")
        if pclip is not None:
            result["data"] = "procedural clip (analytic target); weights FITTED to it by this package's training path (tools/make_trained_checkpoint.py), read back from a reference-layout checkpoint"
            # the timed frames against the analytic target the field was fitted to
            psnrs = []
            for k in sorted({0, K // 2, K - 1}):
                gt = pclip.frame(my_frames[W + k] % pclip.T, HWO, syn.intrinsics_for(HWO, HWO), dev)["gt"].reshape(HWO, HWO, 3)
                mse = float(((((out_u8[k].float() + 0.5) / 255.0) - gt) ** 2).mean().item())
                psnrs.append(round(10.0 * float(np.log10(1.0 / max(mse, 1e-20))), 2))
            result["config"]["timed_frames_psnr_vs_analytic_target_db"] = psnrs
        result["config"]["workload"] = (f"{args.variant} from {real['ckpt']} ({HW}x{HW} rays" + (" + super-resolution to 512x512" if args.variant == "may_torso_sr" else "")
                                        + f", max_steps {hp.get('max_steps')}, T_thresh 0.01)" + (f", driving signals of {args.data_dir}" if real["dataset"] is not None else ""))
        if args.ckpt_parity > 0:
            # the same weights through the CPU oracle (test infrastructure, used here as the checker only): SURVEY 8c's tolerance per frame
            try:
                from oracle import oracle as orc
                orc.build()
                checks = []
                for j in range(min(args.ckpt_parity, len(inputs))):
                    x = inputs[j]
                    pose_np = batch["ngp_poses"][j:j + 1]
                    rays = orc.get_rays(pose_np, np.asarray(intr, np.float32), HW, HW)
                    kwo = dict(bg_color=bg_color.cpu().numpy(), dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"], T_thresh=0.01,
                               eye_area_percent=np.asarray(fi_all[j]["eye_area_percent"], np.float32))
                    ref = orc.render_torso(rays["rays_o"], rays["rays_d"], np.asarray(fi_all[j]["cond"], np.float32), orc.get_bg_coords(HW, HW), orc.convert_poses(pose_np), sd, hp,
                                           lm68=np.asarray(fi_all[j]["lm68"], np.float32), sr_variant=(args.variant == "may_torso_sr"), **kwo)
                    # the product renders the ORACLE's rays here (as the parity tests do): the two ray generators agree to 1-2 ulp in the directions (tests/test_kernels_gpu.py::
                    # test_get_rays), which a sharp trained field turns into ~5e-4 of the pixels beyond 2e-4 -- the renderer's parity is what this entry is about
                    model.precision = "fp32"
                    with torch.no_grad():
                        got = model.render(torch.from_numpy(rays["rays_o"]).to(dev), torch.from_numpy(rays["rays_d"]).to(dev), x["cond"], bg_coords, x["poses"], index=j, staged=False,
                                           bg_color=bg_color, lm68=x["lm68"], perturb=False, force_all_rays=False, T_thresh=0.01, eye_area_percent=x["eye"],
                                           **hp)["rgb_map"].float().cpu().numpy()
                    model.precision = args.precision
                    if args.variant == "may_torso_sr":
                        got = np.transpose(got, (0, 2, 3, 1))
                    err = np.abs(got.reshape(-1, 3) - ref["rgb_map"].reshape(-1, 3)).max(axis=1)
                    checks.append({"frame": j, "rgb_max_abs": float(err.max()), "frac_over_2e-4": float((err > 2e-4).mean())})
                result["config"]["ckpt_parity_fp32_vs_oracle"] = checks
            except Exception as exc:
                result["config"]["ckpt_parity_fp32_vs_oracle"] = {"error": str(exc)}

    if rank == 0 and args.executor == "fused":
        result["roofline"] = parts.head_roofline(model, hp, inputs[W], N, args.variant, frames_per_launch=cr.group, ms_per_frame_period=1e3 * elapsed / K if world == 1 else None,
                                                 pmc_tag=("trained_" if pclip is not None else "") + f"{args.variant}_{HW}_{args.precision}")
    if rank == 0 and world == 1 and not args.no_modes:
        parts.other_modes(result)
    if rank == 0 and world == 1 and not args.no_configs and args.variant == "may_torso" and HW == 512:
        parts.baseline_configs(result)
    if rank == 0 and world == 1 and not args.no_trained and not args.no_configs and real is None and args.variant == "may_torso" and HW == 512:
        parts.trained_fields(result)
    if rank == 0 and "roofline" in result:
        parts.attach_traffic(result["roofline"])
        sr = result.get("configs", {}).get("may_torso_sr_256", {}).get("roofline")
        if sr:
            parts.attach_traffic(sr)
    if rank == 0 and not args.no_grid_stage:
        parts.grid_stage(result)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(args.variant if args.variant != "may_torso_sr" else "may_torso_sr",
                                                  hw_sample=HW, hw_full=HW)
        except Exception as exc:  # the oracle is optional equipment for the bench, never for the product
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
