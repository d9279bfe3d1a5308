"""Run the reference's OWN native kernels (oracle/_ref: raymarching.cu, gridencoder.cu, shencoder.cu, freqencoder.cu compiled unmodified for
gfx950 by oracle/build_ref.py) on an MI355X over the seeded `small` case table of tests/ref_kernel_cases.py and write their outputs as the
committed fixture tests/golden/ref_kernel_golden.npz.  Needs a GPU:

    python oracle/build_ref.py                                    # here, where /root/reference is mounted
    gpurun -- 'python tests/golden/make_golden_ref_kernels.py'    # writes gpurun_out/ref_kernel_golden.npz + ref_kernel_report.json
    cp gpurun_out/ref_kernel_golden.npz tests/golden/

The report lists, per case and scale, the max abs difference ref-vs-oracle and ref-vs-product (no assertions here; the tests assert).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_kernel_cases as rkc  # noqa: E402
from oracle import build_ref, ref_backends  # noqa: E402
from genefaceplusplus_amd import compat_ext  # noqa: E402

NAMES = ("_raymarching_face", "_gridencoder", "_shencoder", "_freqencoder")


def backends():
    ref = {n: build_ref.load(n) for n in NAMES}
    ref_backends.install()
    orc = {n: sys.modules[n] for n in NAMES}
    compat_ext.install()
    hip = {n: sys.modules[n] for n in NAMES}
    return ref, orc, hip


def diff(a, b):
    out = {}
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.shape != y.shape:
            out[str(k)] = f"shape {x.shape} vs {y.shape}"
        elif x.size == 0:
            out[str(k)] = 0.0
        else:
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            out[str(k)] = [float(np.nanmax(d)), float((d > 0).mean())]
    return out


def main():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    ref, orc, hip = backends()
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    report, store = {}, {}
    for scale in ("small", "full"):
        for case in rkc.cases(scale):
            r = rkc.run_case(case, ref, dev)
            if scale == "small":
                r2 = rkc.run_case(case, ref, dev)
                for k, v in r.items():
                    store[f"{case.name}/{k}"] = v
                rerun = diff(r, r2)
            else:
                rerun = None
            entry = {"rerun": rerun}
            try:
                entry["orc"] = diff(rkc.run_case(case, orc, "cpu", f32_only=True), r)
            except Exception as e:  # noqa: BLE001
                entry["orc"] = "ERR " + repr(e)[:200]
            try:
                entry["hip"] = diff(rkc.run_case(case, hip, dev), r)
            except Exception as e:  # noqa: BLE001
                entry["hip"] = "ERR " + repr(e)[:200]
            report[f"{scale}/{case.name}"] = entry
            print(scale, case.name, json.dumps(entry), flush=True)
    np.savez_compressed(os.path.join(outdir, "ref_kernel_golden.npz"), **store)
    with open(os.path.join(outdir, "ref_kernel_report.json"), "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "cases": report}, f, indent=1)


if __name__ == "__main__":
    main()
