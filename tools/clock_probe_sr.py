#!/usr/bin/env python
"""Shader clock while the super-resolution stage runs: tools/probe/clock_probe.hip's one-wavefront probe (fixed number of shader cycles of s_sleep per sample,
100 MHz s_memrealtime stamps, s_memtime next to them) on a side stream, SR forwards on the main stream.   python tools/clock_probe_sr.py"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genefaceplusplus_amd import synthetic as syn  # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution  # noqa: E402

src = os.path.join(ROOT, "tools", "probe", "clock_probe.hip")
so = "/tmp/clock_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
dev = torch.device("cuda:0")
sd = syn.synthetic_sr_state(prefix="")
net = Superresolution(channels=3)
net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
net = net.to(dev).eval()
x = torch.rand(1, 3, 256, 256, device=dev)
side = torch.cuda.Stream()
n = 2000


def window(label, work, spin=2):
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        lib.clock_probe_launch(buf.data_ptr(), n, spin, side.cuda_stream)
    work()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(n, 2).astype(np.float64)
    per = (a[1:, 1] - a[:-1, 1]) / 100e6
    span = slice(n // 8, n // 2)
    cyc = spin * 127 * 64
    mt = (a[1:, 0] - a[:-1, 0])[span].sum() / per[span].sum()
    print(f"{label:46s} {per[span].mean() * 1e6:.3f} us per {cyc}-cycle sleep -> <= {cyc / per[span].mean() / 1e9:.3f} GHz (p5 {cyc / np.percentile(per[span], 95) / 1e9:.3f}, p95 "
          f"{cyc / np.percentile(per[span], 5) / 1e9:.3f}); s_memtime ticks at {mt / 1e9:.3f} GHz in that span")


def forwards(k, mode="random"):
    def f():
        with torch.no_grad():
            for _ in range(k):
                net(x, noise_mode=mode)
    return f


forwards(5)()
window("idle GPU", lambda: time.sleep(0.01))
window("SR forwards back to back", forwards(120))
xb = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
window("rocBLAS f16 GEMM 8192^3 x30", lambda: [torch.mm(xb, xb) for _ in range(30)])
