"""Which hardware function of a wavefront goes wrong beside the polyphase SR launch?  Runs tools/probe/canary.hip (self-checking MFMA chains, vector ALU, LDS, global
loads, shuffles, dot products -- each repetition compared with the wavefront's own first) on the main stream while a second stream keeps SR forwards in flight.

STATUS: written at the end of round 6 and compiled for gfx950; GPU access closed before it ran once (docs/LAB_NOTEBOOK.md).  Treat it as unverified.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o build/probe/libcanary.so tools/probe/canary.hip      (here; the .so travels with gpurun)
    python tools/canary.py [launches] [iters per launch] [workgroups]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from genefaceplusplus_amd import synthetic as syn, tuning                      # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution             # noqa: E402

TESTS = ("mfma chain, register operands", "mfma chain, A from LDS", "vector ALU chain", "LDS write / read back", "16-byte global loads", "lane shuffle",
         "dot2 + permlane32_swap", "mfma chain on an accumulator that starts as an LDS-read bias")


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    lib = ctypes.CDLL(os.path.join(ROOT, "build", "probe", "libcanary.so"))
    lib.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.canary_tab_value.restype = ctypes.c_float
    lib.canary_tab_value.argtypes = [ctypes.c_uint32]
    dev = torch.device("cuda:0")
    n_tab = 1 << 20
    i = np.arange(n_tab, dtype=np.uint64)
    tab = ((((i * 2654435761) & 0xFFFFFFFF) >> 20).astype(np.float32) * np.float32(1.0 / 4096.0))
    assert tab[12345] == lib.canary_tab_value(12345)
    tab = torch.from_numpy(tab).to(dev)
    sd = syn.synthetic_sr_state(prefix="")
    net = Superresolution(channels=3)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    x = torch.rand(1, 3, 256, 256, device=dev)
    side = torch.cuda.Stream()
    per = int(os.environ.get("SR_PER_LAUNCH", "4"))
    with torch.no_grad():
        for load in ("alone", "sr poly=0 on a second stream", "sr poly=1 on a second stream"):
            res = torch.zeros(8, 4, dtype=torch.int64, device=dev)
            for _ in range(launches):
                if load != "alone":
                    with torch.cuda.stream(side), tuning.tuned(sr_up_poly=int(load[8])):
                        for _ in range(per):
                            net(x, noise_mode="const")
                rc = lib.canary_launch(torch.cuda.current_stream().cuda_stream, res.data_ptr(), tab.data_ptr(), n_tab, iters, wgs)
                assert rc == 0, rc
            torch.cuda.synchronize()
            r = res.cpu().numpy()
            print(f"{load}: {launches} launches x {wgs} workgroups x 4 wavefronts x {iters} repetitions")
            for t, name in enumerate(TESTS):
                if r[t, 0]:
                    print(f"   {name}: {int(r[t, 0])} wavefront-repetitions differ; lanes (OR) {int(r[t, 1]) & 0xFFFFFFFFFFFFFFFF:016x}; registers (OR) {int(r[t, 3]):04x}; last at repetition {int(r[t, 2])}")
            if not r[:, 0].any():
                print("   every repetition of every test equals the first")


main()
