"""Frame-parallel sharding of a clip across the GPUs of one node (new work -- the reference renders
frames serially on one device, inference/genefacepp_infer.py:460-485).

Frames are independent units (each needs only its own pose, conditioning window, landmarks; the weights are replicated), so
the data path has no collective.  The single exchange step is the all_gather of finished uint8 frames (786 432 B per 512x512
frame) over RCCL / xGMI -- backend "nccl" on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist

from ._lib import GfppError, call


def shard_frames(n_frames, rank, world, interleaved=False):
    """Frame indices owned by `rank`.

    contiguous (default): blocks of ceil(n/world) frames keep video order inside a rank (what a clip renderer wants);
    interleaved: frame i -> rank i % world (what a streaming server wants: every rank works on the "current" second)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    if interleaved:
        return list(range(rank, n_frames, world))
    per = -(-n_frames // world)
    return list(range(min(rank * per, n_frames), min((rank + 1) * per, n_frames)))


def frame_owner(frame_idx, n_frames, world, interleaved=False):
    if interleaved:
        return frame_idx % world
    per = -(-n_frames // world)
    return frame_idx // per


def to_uint8_hwc(rgb, out=None):
    """float [..,3] in [0,1] -> uint8 (truncating, like `(x * 255.).int()` in genefacepp_infer.py:468); one HIP launch."""
    rgb = rgb.contiguous()
    if out is None:
        out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
    if not rgb.is_cuda:
        raise GfppError("to_uint8_hwc: the frame must be on the GPU (there is no CPU path)")
    call("gfpp_rgb_to_u8", rgb.data_ptr(), rgb.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out


def gather_clip(local_frames, n_frames, interleaved=False, group=None, dst=None):
    """Exchange step of the frame-parallel clip: the per-rank uint8 frame stacks [F_local, H, W, 3] are reassembled in frame order.

    dst=None: all_gather -- every rank gets the clip.  dst=<group rank>: gather to that rank only -- the reference has ONE consumer of the
    frames (the video writer, genefacepp_infer.py:454-518), so the other ranks need not receive world x F frames each: over xGMI's
    point-to-point links every rank then sends its stack once, to the writer (returns None on the other ranks).
    Ranks may own different numbers of frames (last block shorter): stacks are padded to the longest."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_frames
    rank = dist.get_rank(group)
    counts = [len(shard_frames(n_frames, r, world, interleaved)) for r in range(world)]
    longest = max(counts)
    if local_frames.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local_frames.shape[0]} frames, expected {counts[rank]}")
    if local_frames.shape[0] < longest:
        pad = torch.zeros(longest - local_frames.shape[0], *local_frames.shape[1:], dtype=local_frames.dtype, device=local_frames.device)
        local_frames = torch.cat([local_frames, pad], dim=0)
    if dst is None:
        parts = [torch.empty_like(local_frames) for _ in range(world)]
        dist.all_gather(parts, local_frames.contiguous(), group=group)
    else:
        parts = [torch.empty_like(local_frames) for _ in range(world)] if rank == dst else None
        dist.gather(local_frames.contiguous(), parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    clip = torch.empty(n_frames, *local_frames.shape[1:], dtype=local_frames.dtype, device=local_frames.device)
    for r in range(world):
        idx = shard_frames(n_frames, r, world, interleaved)
        if idx:
            clip[torch.tensor(idx, device=clip.device)] = parts[r][:len(idx)]
    return clip


# ---- one frame shared between GPUs: ray tiles (north star: "within a frame, ray tiles"; SURVEY 8e caveat) -----------------------------------

def ray_tile(n_rays, rank, world):
    """Contiguous slice [lo, hi) of the frame's rays rendered by `rank` (whole image rows when n_rays / world is a multiple of the width)."""
    per = -(-n_rays // world)
    return min(rank * per, n_rays), min((rank + 1) * per, n_rays)


def render_frame_tiled(model, rays_o, rays_d, cond, bg_coords, poses, group=None, bg_color=None, **render_kwargs):
    """Latency mode: ONE frame rendered by all ranks of `group`, each taking a contiguous tile of the rays, then all_gather of the tiles.

    Every rank passes the same full-frame inputs (rays [1,N,3], bg_coords [1,N,2], bg_color [1,N,3] or None).  Through the torso pass a pixel
    only needs its own head colour / alpha, so tiles are independent except for the loop control: n_step = clamp(N // n_alive, 1, 8)
    (renderer.py:364) uses the FRAME-wide alive count, which the pipeline all-reduces once per trip (`ray_shard`) -- the result equals the
    single-GPU frame bit for bit.  Returns {'rgb_map' [1,N,3], 'depth_map' [1,N]} (+ 'torso_alpha_map' [N,1] for the torso models) on every rank.
    Not for the *_sr models: the SR convolutions need the whole 256^2 image (gather first, then super-resolve on one rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    N = rays_o.shape[-2]
    lo, hi = ray_tile(N, rank, world)
    sl = slice(lo, hi)
    kw = dict(render_kwargs)
    if world > 1:
        from ._lib import GfppError
        if hasattr(model, "sr_net"):
            raise GfppError("render_frame_tiled: the *_sr models are not tileable (their render() does not take ray_shard and the SR convolutions "
                            "need the whole image): render frame-parallel instead")
        if (world - 1) * (-(-N // world)) >= N:
            raise GfppError(f"render_frame_tiled: {N} rays over {world} ranks leaves a rank without rays (tiles are ceil(N / world) rays)")
        kw["ray_shard"] = (group, N)
        if getattr(model, "hparams", {}).get("torso_head_aware") and hasattr(model, "torso_embedder") and kw.get("use_head_for_torso") is None:
            # the head-aware torso draws a coin per call (radnerf_torso.py:135-139): ONE draw for the frame, made on the group's rank 0
            import random
            coin = torch.tensor([1 if random.random() < 0.5 else 0], dtype=torch.int32, device=rays_o.device)
            dist.broadcast(coin, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            kw["use_head_for_torso"] = bool(coin.item())
    bg = bg_color[..., sl, :] if torch.is_tensor(bg_color) and bg_color.shape[-2] == N else bg_color
    with torch.no_grad():
        res = model.render(rays_o[..., sl, :].contiguous(), rays_d[..., sl, :].contiguous(), cond, bg_coords[..., sl, :].contiguous(), poses, bg_color=bg, **kw)
    if world == 1:
        return res
    per = -(-N // world)
    out = {}
    for key, width in (("rgb_map", 3), ("depth_map", 1), ("torso_alpha_map", 1)):
        if key not in res:
            continue
        tile = res[key].reshape(-1, width).float()
        pad = torch.zeros(per, width, device=tile.device)
        pad[:tile.shape[0]] = tile
        full = torch.empty(world * per, width, device=tile.device)
        dist.all_gather_into_tensor(full, pad, group=group)
        full = full[:N]
        out[key] = full.view(1, N, 3) if key == "rgb_map" else (full.view(1, N) if key == "depth_map" else full.view(N, 1))
    return out


# ---- several identities on one node (BASELINE config 5: 4 person-specific models on 8 GPUs, 2 GPUs each) --------

def identity_groups(world, n_identities):
    """Contiguous rank blocks, one per identity: identity i owns ranks[i].  Contiguous so that the ranks of one identity are xGMI
    neighbours on one node and its frame gather stays inside the block.  world must be a multiple of n_identities."""
    if n_identities < 1 or world % n_identities:
        raise ValueError(f"{world} ranks cannot be split evenly over {n_identities} identities")
    per = world // n_identities
    return [list(range(i * per, (i + 1) * per)) for i in range(n_identities)]


def make_identity_groups(n_identities):
    """Create one sub-communicator per identity (every rank must call this: `new_group` is collective).
    Returns (my_identity, my_group, all_rank_lists)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ranks = identity_groups(world, n_identities)
    mine, my_group = None, None
    for i, block in enumerate(ranks):
        g = dist.new_group(ranks=block)
        if rank in block:
            mine, my_group = i, g
    return mine, my_group, ranks


def share_driving_signals(tensors, src=0):
    """The upstream audio2motion result (expression / landmark windows, blink values: tens of KB) is computed once on `src` and broadcast to every
    rank of the job, instead of once per identity (the reference runs it per inference call, genefacepp_infer.py:298-431).
    `tensors`: dict name -> tensor, same shapes on every rank (allocate empties on the receivers)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for name in sorted(tensors):
            dist.broadcast(tensors[name], src=src)
    return tensors


def gather_identity_clip(local_frames, n_frames, my_group, interleaved=False, dst=None):
    """Frame gather inside one identity's rank block (the only collective of the multi-identity job besides the one broadcast);
    dst = rank INSIDE the block that writes the identity's video (None: every rank of the block gets the clip)."""
    return gather_clip(local_frames, n_frames, interleaved, group=my_group, dst=dst)
