"""Frame-parallel sharding + clip gather on the `gloo` backend, world_size 2 (the N>1 path of bench.py / frames.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genefaceplusplus_amd import frames


def test_shard_frames_partitions():
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            for inter in (False, True):
                parts = [frames.shard_frames(n, r, world, inter) for r in range(world)]
                flat = sorted(i for p in parts for i in p)
                assert flat == list(range(n))
                for r, p in enumerate(parts):
                    assert all(frames.frame_owner(i, n, world, inter) == r for i in p)
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= (1 if inter else -(-n // world))
    with pytest.raises(ValueError):
        frames.shard_frames(4, 2, 2)


def test_to_uint8_has_no_cpu_path():
    from genefaceplusplus_amd._lib import GfppError
    with pytest.raises(GfppError):
        frames.to_uint8_hwc(torch.tensor([[0.0, 0.5, 1.0]]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, inter, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = frames.shard_frames(n_frames, rank, world, inter)
    # a "rendered frame" whose every pixel encodes its global frame index
    local = torch.stack([torch.full((4, 4, 3), i % 256, dtype=torch.uint8) for i in mine]) if mine else torch.zeros(0, 4, 4, 3, dtype=torch.uint8)
    clip = frames.gather_clip(local, n_frames, inter)
    ok = clip.shape == (n_frames, 4, 4, 3) and all(int(clip[i, 0, 0, 0]) == i % 256 for i in range(n_frames))
    # gather-to-writer: only rank `dst` receives (the reference has a single video writer), byte-identical to the all_gather result
    for dst in range(world):
        w = frames.gather_clip(local, n_frames, inter, dst=dst)
        ok = ok and ((w is None) if rank != dst else bool(torch.equal(w, clip)))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, bool(ok), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,inter", [(7, False), (8, True), (5, True)])
def test_gather_clip_world2_gloo(n_frames, inter):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, inter, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in results) == [0, 1]
    assert all(ok for _, ok, _ in results) and all(t == 2.0 for _, _, t in results)


def test_identity_groups_partition():
    assert frames.identity_groups(8, 4) == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert frames.identity_groups(2, 2) == [[0], [1]]
    assert frames.identity_groups(4, 1) == [[0, 1, 2, 3]]
    with pytest.raises(ValueError):
        frames.identity_groups(8, 3)


def _identity_worker(rank, world, port, n_ident, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ident, group, blocks = frames.make_identity_groups(n_ident)
    # shared upstream result: computed on rank 0 only, then identical everywhere
    sig = {"cond": torch.arange(12, dtype=torch.float32).reshape(3, 4) if rank == 0 else torch.zeros(3, 4),
           "eye": torch.full((3, 1), 0.25) if rank == 0 else torch.zeros(3, 1)}
    frames.share_driving_signals(sig, src=0)
    shared_ok = bool(torch.equal(sig["cond"], torch.arange(12, dtype=torch.float32).reshape(3, 4)) and float(sig["eye"].sum()) == 0.75)
    # per-identity conditioning of the shared landmarks (postnet.IdentityConditioner): every rank of an identity's block must arrive at the same
    # windows, and they must be what one process computes for that person
    from genefaceplusplus_amd.postnet import IdentityConditioner
    lm = {"idexp_lm3d": 0.3 * torch.randn(n_frames, 68, 3, generator=torch.Generator().manual_seed(5)) if rank == 0 else torch.zeros(n_frames, 68, 3)}
    frames.share_driving_signals(lm, src=0)
    person = IdentityConditioner(0.3 * torch.randn(300, 68, 3, generator=torch.Generator().manual_seed(100 + ident)))
    wins = person.cond_wins(lm["idexp_lm3d"], 3)
    want = person.cond_wins(0.3 * torch.randn(n_frames, 68, 3, generator=torch.Generator().manual_seed(5)), 3)
    shared_ok = shared_ok and bool(torch.equal(wins, want)) and tuple(wins.shape) == (n_frames, 3, 1, 204)
    # every identity renders the whole clip, frame-parallel inside its block; a frame's pixels encode (identity, frame)
    local_rank, local_world = blocks[ident].index(rank), len(blocks[ident])
    mine = frames.shard_frames(n_frames, local_rank, local_world)
    local = torch.stack([torch.full((2, 2, 3), 16 * ident + i, dtype=torch.uint8) for i in mine]) if mine else torch.zeros(0, 2, 2, 3, dtype=torch.uint8)
    clip = frames.gather_identity_clip(local, n_frames, group)
    clip_ok = clip.shape[0] == n_frames and all(int(clip[i, 0, 0, 0]) == 16 * ident + i for i in range(n_frames))
    w = frames.gather_identity_clip(local, n_frames, group, dst=0)          # the block's rank 0 writes this identity's video
    clip_ok = clip_ok and ((w is None) if local_rank != 0 else bool(torch.equal(w, clip)))
    q.put((rank, ident, shared_ok, bool(clip_ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_ident", [(2, 2), (4, 2)])
def test_multi_identity_groups_gloo(world, n_ident):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_identity_worker, args=(r, world, port, n_ident, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = world // n_ident
    for rank, ident, shared_ok, clip_ok in results:
        assert ident == rank // per and shared_ok and clip_ok


class _FakeRenderer:
    """Stands in for clip.ClipRenderer in the CPU test of the distributed clip path: a 'frame' is a tiny image holding its index."""

    def render_to_device(self, clip, frame_indices=None, out=None):
        idx = list(range(clip["frames"])) if frame_indices is None else list(frame_indices)
        return torch.stack([torch.full((2, 3, 3), i, dtype=torch.uint8) for i in idx]) if idx else torch.zeros(0, 2, 3, 3, dtype=torch.uint8)


def _clip_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genefaceplusplus_amd.clip import render_clip_distributed
    out = render_clip_distributed(_FakeRenderer(), {"frames": n_frames})
    ok = out.shape == (n_frames, 2, 3, 3) and all(int(out[i, 0, 0, 0]) == i for i in range(n_frames))
    inter = render_clip_distributed(_FakeRenderer(), {"frames": n_frames}, interleaved=True)
    ok = ok and all(int(inter[i, 1, 2, 2]) == i for i in range(n_frames))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_render_clip_distributed_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clip_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results)
