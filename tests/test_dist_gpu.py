"""The N>1 path on real devices.  (1) RCCL, one rank per GPU (skipped on a 1-GPU box): the distributed clip equals the single-GPU clip byte
for byte, through both exchange steps (all_gather / gather-to-writer).  (2) bench.py's own N=2 control flow on ONE GPU over gloo (two ranks
sharing cuda:0): the JSON contract of a multi-rank run."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _tagged(stdout, tag):
    """The ranks share one stdout and their lines can land on top of each other: pick the (flat) JSON objects out by tag, not by line."""
    import re
    return [json.loads(m) for m in re.findall(tag + r" (\{[^{}]*\})", stdout)]


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(n, script_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL); the driver's multi-GPU node runs it")
def test_rccl_distributed_clip_equals_single_gpu_clip():
    n = min(torch.cuda.device_count(), 4)
    r = _torchrun(n, [os.path.join(ROOT, "tests", "dist_gpu_worker.py")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    results = _tagged(r.stdout, "DISTRESULT")
    assert sorted(x["rank"] for x in results) == list(range(n))
    for x in results:
        assert x["world_seen"] == n and x["backend"] == "nccl" and x["replicas_agree"]
        assert x["all_equal_False"] and x["all_equal_True"] and x["writer_equal_False"] and x["writer_equal_True"], x


def _bench(args, timeout=900):
    """The driver's command shape: `python bench.py --gpus N ...` with no launcher around it (bench.py starts its own ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("gather", ["all", "writer"])
def test_bench_two_ranks_on_one_gpu_gloo(gather):
    r = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--hw", "128", "--dist-backend", "gloo", "--gather", gather, "--long-run-frames", "24",
                "--no-modes", "--no-cpu-baseline", "--no-grid-stage", "--no-configs"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [line for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
    ev = d["config"]["dist"]                       # the line evidences its own ranks
    assert ev["launcher"] == "self" and ev["backend"] == "gloo" and ev["world_size"] == 2 and ev["ranks_seen"] == 2
    assert len(ev["device_uuids"]) == 2 and len(ev["per_rank_fps"]) == 2 and all(v > 0 for v in ev["per_rank_fps"])
    assert ev["gathered_MB"] == round(2 * 6 * 128 * 128 * 3 / 1e6, 2) and ev["gather_ms_exposed"] >= 0
    assert ev["writer_holds_own_frames"] and all(ev["writer_frames_nonzero_per_rank"])
    # the steady-state pass of the N > 1 line (round-5 review, item 4): same job shape in blocks, all ranks' frames / the slowest rank's time, rank 0 alone beside it
    lr = ev["long_run"]
    assert lr["frames_per_rank"] == 24 and lr["blocks"] == 1 and lr["value"] > 0 and len(lr["per_rank_fps"]) == 2 and all(v > 0 for v in lr["per_rank_fps"])
    assert lr["rank0_alone_fps"] > 0 and 0 < lr["efficiency_vs_rank0_alone"] and lr["gather_tail_frames"] == ev["gather_tail_frames"] == 4
    assert ev["ranks_ok"] is True and lr["gather_ms_exposed_per_block"] >= 0


def test_bench_under_an_external_launcher_still_works():
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--hw", "128", "--dist-backend", "gloo", "--long-run-frames", "0",
                      "--no-modes", "--no-cpu-baseline", "--no-grid-stage", "--no-configs"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["dist"]["launcher"].startswith("external") and d["config"]["dist"]["ranks_seen"] == 2


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="1-GPU box only")
def test_bench_rccl_on_too_few_gpus_fails_loudly():
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--hw", "64", "--no-modes", "--no-cpu-baseline", "--no-grid-stage", "--no-configs"], timeout=300)
    assert r.returncode != 0 and "one GPU per rank" in r.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL); the driver's multi-GPU node runs it")
def test_bench_rccl_two_gpus_self_launched():
    r = _bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--long-run-frames", "200", "--no-modes", "--no-cpu-baseline", "--no-grid-stage", "--no-configs"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    ev = d["config"]["dist"]
    assert ev["backend"].startswith("rccl") and ev["ranks_seen"] == 2 and ev["distinct_devices"] == 2 and ev["writer_holds_own_frames"] and ev["ranks_ok"]
    assert ev["long_run"]["frames_per_rank"] == 200 and ev["long_run"]["efficiency_vs_rank0_alone"] > 0.5


def test_bench_identities_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--identities", "2", "--steps", "4", "--warmup", "2", "--hw", "128"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["config"]["identities"] == 2 and d["config"]["frames_total"] == 8 and d["value"] > 0


def _tile_results(n, backend):
    r = _torchrun(n, [os.path.join(ROOT, "tests", "dist_tile_worker.py"), backend])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = _tagged(r.stdout, "TILERESULT")
    assert sorted(x["rank"] for x in out) == list(range(n))
    for x in out:
        for k, v in x.items():
            if k.endswith(("fp32", "fp16", "bf16")):
                assert v is True, (k, x)
            if k.endswith("_trips"):
                assert v >= 4, (k, x)


@pytest.mark.parametrize("n", [2, 3])
def test_ray_tile_sharding_equals_single_gpu_frame_gloo(n):
    """ONE frame rendered by n ranks (ray tiles, frame-wide alive count all-reduced per trip) == the single-GPU frame, bit for bit; the ranks share
    this box's GPU over gloo (the RCCL variant below needs one GPU per rank)."""
    _tile_results(n, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL)")
def test_ray_tile_sharding_equals_single_gpu_frame_rccl():
    _tile_results(min(torch.cuda.device_count(), 4), "nccl")


def test_bench_ray_tile_mode_two_ranks_gloo():
    r = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--hw", "128", "--dist-backend", "gloo", "--shard", "rays"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["dist"]["ranks_seen"] == 2


def test_bench_identities_two_ranks_gloo():
    r = _bench(["--gpus", "2", "--identities", "2", "--steps", "4", "--warmup", "2", "--hw", "128", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["identities"] == 2 and d["config"]["dist"]["ranks_seen"] == 2 and d["value"] > 0


def test_bench_eight_ranks_on_one_gpu_gloo():
    """The driver's 8-GPU command shape on the 1-GPU box: `python bench.py --gpus 8 --dist-backend gloo --steps 20` -- eight ranks share the one GPU, so
    this checks control flow only: every rank seen, each rendering its own block of 20 frames, the timed window bracketed by barriers, the finished frames
    gathered to the writer in chunks with the exposed gather time reported (what a scaling efficiency computed from a 20-step run rests on)."""
    r = _bench(["--gpus", "8", "--steps", "20", "--warmup", "2", "--hw", "96", "--dist-backend", "gloo", "--long-run-frames", "16", "--no-modes", "--no-cpu-baseline",
                "--no-grid-stage", "--no-configs"], timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [line for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    ev = d["config"]["dist"]
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["scaling"] == "weak"
    assert ev["ranks_seen"] == 8 and ev["world_size"] == 8 and len(ev["per_rank_fps"]) == 8 and all(v > 0 for v in ev["per_rank_fps"])
    assert ev["gather"] == "writer" and ev["gather_ms_exposed"] >= 0 and ev["gathered_MB"] == round(8 * 20 * 96 * 96 * 3 / 1e6, 2)
    assert ev["writer_holds_own_frames"] and all(ev["writer_frames_nonzero_per_rank"]) and len(ev["writer_frames_nonzero_per_rank"]) == 8
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]            # whole-job frames / the slowest rank's time
    lr = ev["long_run"]                                                                  # ... and the steady-state pass of the same shape
    assert lr["frames_per_rank"] == 16 and len(lr["per_rank_fps"]) == 8 and lr["value"] > 0 and lr["rank0_alone_fps"] > 0 and ev["gather_tail_frames"] == 4


def test_bench_four_identities_on_eight_ranks_gloo():
    """BASELINE configs[4] literally: 4 person-specific models on 8 ranks (2 each), shared driving signals broadcast once (gloo, one GPU: control flow)."""
    r = _bench(["--gpus", "8", "--identities", "4", "--steps", "4", "--warmup", "2", "--hw", "64", "--dist-backend", "gloo"], timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([line for line in r.stdout.splitlines() if line.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["config"]["identities"] == 4 and d["config"]["dist"]["ranks_seen"] == 8 and d["config"]["frames_total"] == 32 and d["value"] > 0
