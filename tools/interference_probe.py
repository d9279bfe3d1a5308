"""Does a kernel's result depend on WHAT ELSE runs on the device?  One frame of a model rendered again and again with plain launches on the main stream (no graph, no
clip renderer) while a second stream keeps SR forwards (or a plain GEMM) in flight; every render is compared bit for bit with the first.

    python tools/interference_probe.py [renders] [variant] [HW] [precision]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model, frame_case, product_render                    # noqa: E402
from genefaceplusplus_amd import synthetic as syn, tuning                      # noqa: E402
from genefaceplusplus_amd.radnerfs.superres import Superresolution             # noqa: E402


def main():
    renders = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    variant = sys.argv[2] if len(sys.argv) > 2 else "may_torso"
    HW = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    precision = sys.argv[4] if len(sys.argv) > 4 else "fp16"
    dev = torch.device("cuda:0")
    case = frame_case(variant, HW)
    model = build_model(case, dev, "fused")
    model.precision = precision
    case["hp"] = dict(case["hp"], use_head_for_torso=True)
    sd = syn.synthetic_sr_state(prefix="")
    net = Superresolution(channels=3)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    net.lane = 5                                    # (its own activations)
    x = torch.rand(1, 3, 256, 256, device=dev)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    side = torch.cuda.Stream()
    keys = ("rgb_map", "depth_map", "torso_alpha_map", "torso_rgb_map")

    def render():
        r = product_render(model, case, dev, rays_from="product")
        return {k: r[k].clone() for k in keys if k in r and torch.is_tensor(r[k])}

    with torch.no_grad():
        want = render()
        torch.cuda.synchronize()
        for load in ("alone", "gemm on a second stream", "sr poly=0 on a second stream", "sr poly=1 on a second stream"):
            bad = {}
            for it in range(renders):
                if load != "alone":
                    with torch.cuda.stream(side):
                        if load.startswith("gemm"):
                            a @ a
                        else:
                            with tuning.tuned(sr_up_poly=int(load[8])):
                                for _ in range(int(os.environ.get("SR_PER_RENDER", "3"))):
                                    net(x, noise_mode="const")
                got = render()
                for k in want:
                    if not torch.equal(got[k], want[k]):
                        bad[k] = bad.get(k, 0) + 1
            torch.cuda.synchronize()
            print(f"{variant} {HW} {precision}  {load}: of {renders} renders, maps that differ from the first: {bad or 'none'}", flush=True)


main()
