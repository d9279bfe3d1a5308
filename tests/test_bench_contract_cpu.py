"""The bench line committed with the round's profiles carries every field the driver's contract names (bench.py docstring)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(HERE, "..", "profiles", "r03_bench_final.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] in ("f16", "bf16", "f32") and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]          # whole-job throughput == frames / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_per_frame"] == 1 and "limiter" in r and r["l2"]["peak"] > r["peak"]      # one persistent launch; the level that serves the stream is labelled
    assert d["config"]["host_issue_ms_per_frame"] <= 0.05                          # the frame loop is issued from C
    assert "roofline" in d["configs"]["may_torso_sr_256"]                          # the released checkpoint's own configuration carries its own roofline
    assert d["dtype"] == "bf16" and "bf16" not in d["modes"]                       # BASELINE configs[2] literally is the headline
    assert d["modes"]["long_run"]["frames"] >= 2000 and d["modes"]["long_run"]["block_std"] >= 0
    assert d["configs"]["may_head_fp32_latency"]["frames"] >= 200 and d["configs"]["may_head_fp32_latency"]["latency_ms_p99"] >= d["configs"]["may_head_fp32_latency"]["latency_ms_p50"]
    assert d["configs"]["crop64_cpu_oracle"]["rays_per_step"] == 1024
    assert d["grid_stage_ray_stream"]["frac"] is None                              # a cache-served stream carries no HBM fraction
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher (the driver's command shape) starts two ranks itself; on this GPU-less box both fail loudly
    ("needs MI355X GPUs") and the job's exit code says so -- no AssertionError about WORLD_SIZE any more."""
    import subprocess
    import sys
    root = os.path.join(HERE, "..")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" not in r.stderr
    assert r.stderr.count("bench.py needs MI355X GPUs") >= 1
    assert not [line for line in r.stdout.splitlines() if line.startswith("{")]
