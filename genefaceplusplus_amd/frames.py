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
