"""Training path under DistributedDataParallel (person-specific training, one process per GPU over RCCL): two gloo ranks on the CPU,
kernels served by the oracle (tests/ddp_train_worker.py).  After backward every rank must hold the mean of the ranks' single-process gradients."""
import json
import os
import re
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_training_step_under_ddp_world2_gloo():
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_train_worker.py"), str(r), "2", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    results = [json.loads(m) for so, _ in outs for m in re.findall(r"DDPRESULT (\{[^{}]*\})", so)]
    assert sorted(r["rank"] for r in results) == [0, 1]
    for r in results:
        assert r["ranks_agree"], r
        assert r["touched"] >= 20, r                     # grids, MLPs, conditioning nets all received gradients
        assert r["worst_rel"] <= 1e-5, r                 # DDP's bucketed all-reduce sums in a different order than the plain all_reduce
    assert results[0]["loss"] != results[1]["loss"]      # the ranks really rendered different views
