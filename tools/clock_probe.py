#!/usr/bin/env python
"""Shader clock under load: a one-wavefront probe kernel samples s_memtime vs the 100 MHz s_memrealtime on a side stream while the main stream
runs (a) nothing, (b) the 16-bit radiance-block kernel on the whole chip (tools/eval_bench.py's workload), (c) the same with fewer active waves.
Prints the effective shader frequency of each window.   python tools/clock_probe.py [precision]"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import frame_case, build_model  # noqa: E402

src = os.path.join(ROOT, "tools", "probe", "clock_probe.hip")
so = "/tmp/clock_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
case = frame_case("may_torso", 512)
model = build_model(case, dev, "fused")
model.precision = precision
pipe = model.pipeline()
rng = np.random.default_rng(0)
M = 1 << 20
P = torch.from_numpy((rng.uniform(-1, 1, (M, 3)) * np.array([0.3, 0.22, 0.35])).astype(np.float32)).to(dev)
D = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
cf, ind = torch.randn(64, device=dev) * 0.1, model.individual_embeddings[0]
side = torch.cuda.Stream()
n = 2000


def window(label, work, spin=2):
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        lib.clock_probe_launch(buf.data_ptr(), n, spin, side.cuda_stream)
    t0 = time.perf_counter()
    work()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(n, 2).astype(np.float64)
    dt_ref = (a[-1, 1] - a[0, 1]) / 100e6
    per = (a[1:, 1] - a[:-1, 1]) / 100e6                 # real seconds per sample = (spin * 127 * 64 + loop overhead) shader cycles
    span = slice(n // 8, n // 2)                          # while the main-stream work is running (it starts right after the probe)
    cyc = spin * 127 * 64
    print(f"{label:42s} probe {dt_ref * 1e3:6.2f} ms | {per[span].mean() * 1e6:.3f} us per {cyc}-cycle sleep -> <= {cyc / per[span].mean() / 1e9:.3f} GHz (p5 {cyc / np.percentile(per[span], 95) / 1e9:.3f}, p95 "
          f"{cyc / np.percentile(per[span], 5) / 1e9:.3f}); s_memtime rate {(a[-1, 0] - a[0, 0]) / (a[-1, 1] - a[0, 1]) * 100e6 / 1e9:.3f} GHz")


def evals(k):
    def f():
        with torch.no_grad():
            for _ in range(k):
                pipe.eval_samples(P, D, cf, ind)
    return f


with torch.no_grad():
    pipe.eval_samples(P, D, cf, ind)
window("idle GPU", lambda: time.sleep(0.01))
window("radiance blocks, whole chip, 8 waves/CU", evals(60))
os.environ["GFPP_EVAL_WAVES"] = "4"
window("radiance blocks, whole chip, 4 waves/CU", evals(40))
os.environ["GFPP_EVAL_WAVES"] = "1"
window("radiance blocks, whole chip, 1 wave/CU", evals(25))
del os.environ["GFPP_EVAL_WAVES"]
os.environ["GFPP_EVAL_GRID"] = "32"
window("radiance blocks, 32 CUs, 8 waves/CU", evals(8))
del os.environ["GFPP_EVAL_GRID"]
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
window("rocBLAS bf16 GEMM 8192^3 x30", lambda: [torch.mm(x, x) for _ in range(30)])
