#!/usr/bin/env python
"""Generate tests/golden/sr_golden.npz + sr_state_manifest.json by running the REFERENCE's own Superresolution module
(modules/radnerfs/radnerf_sr.py:14-43 on top of modules/eg3ds/*) on the CPU (where its blocks run in fp32: force_fp32,
superresolution.py:204-206) with the deterministic weights of genefaceplusplus_amd.synthetic.synthetic_sr_state.

Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_golden_sr.py

The class is executed from the reference's source text; only the surrounding module (radnerf_sr.py imports the CUDA-only renderer
stack at import time) is not imported.  Recorded: the state-dict layout, and for two inputs x two noise modes the 512x512 output
at a set of crops (corners, edges, centre) plus per-channel sums -- small enough to commit, dense enough to pin every layer."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from genefaceplusplus_amd import synthetic as syn  # noqa: E402

CROPS = [(0, 0), (0, 480), (480, 0), (480, 480), (240, 240), (101, 390)]   # top-left corners of 32x32 crops


def sr_inputs():
    rng = np.random.default_rng(11)
    a = rng.random((1, 3, 256, 256)).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, 256, dtype=np.float32), np.linspace(0, 1, 256, dtype=np.float32), indexing="ij")
    b = np.stack([0.5 + 0.5 * np.sin(9 * xx + 3 * yy), yy * xx, 0.25 + 0.5 * (np.cos(17 * yy) > 0)], 0)[None].astype(np.float32)
    return {"noise": a, "smooth": b}


def main():
    src = open(os.path.join(REF, "modules/radnerfs/radnerf_sr.py")).read()
    ns = {}
    exec("import torch\nfrom modules.eg3ds.models.superresolution import *\n" + src[src.index("class Superresolution"):src.index("class RADNeRFwithSR")], ns)
    model = ns["Superresolution"](channels=3).eval()
    ref_sd = model.state_dict()
    manifest = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()}
    sd = syn.synthetic_sr_state(prefix="")
    assert set(sd) == set(manifest), (set(sd) ^ set(manifest))
    for k, v in sd.items():
        assert list(v.shape) == manifest[k][0] and str(v.dtype) == manifest[k][1], (k, v.shape, v.dtype, manifest[k])
        if k.endswith("resample_filter"):
            np.testing.assert_array_equal(v, ref_sd[k].numpy())           # setup_filter([1,3,3,1])
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    with open(os.path.join(HERE, "sr_state_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)

    out = {"crops": np.array(CROPS, np.int32)}
    with torch.no_grad():
        for name, x in sr_inputs().items():
            for mode in ("const", "none"):
                y = model(torch.from_numpy(x).clone(), noise_mode=mode).numpy()
                assert y.shape == (1, 3, 512, 512) and y.dtype == np.float32
                out[f"{name}.{mode}.crops"] = np.stack([y[0, :, r:r + 32, c:c + 32] for r, c in CROPS])
                out[f"{name}.{mode}.sum"] = y.astype(np.float64).sum(axis=(0, 2, 3))
                out[f"{name}.{mode}.abs_sum"] = np.abs(y).astype(np.float64).sum(axis=(0, 2, 3))
                print(name, mode, "mean", y.mean(axis=(0, 2, 3)), "range", y.min(), y.max())
    # one TRAINING step of the reference module (noise_mode 'random' from a seeded generator: four randn draws, one per layer, in
    # execution order): output, and gradients of a photometric loss w.r.t. the input and a handful of parameters
    model.train()
    x = torch.from_numpy(sr_inputs()["smooth"]).clone().requires_grad_(True)
    torch.manual_seed(13)
    # Superresolution.forward (radnerf_sr.py:30-43) spelled out with two separate clones for `x` and `rgb`: on CPU the blocks run in fp32, so
    # `x.to(dtype)` is the input itself and the block's in-place `img.add_(y)` would overwrite a tensor autograd still needs (on the GPU the
    # fp16 cast makes the copy).  The blocks themselves are the reference's, unmodified.
    ws = torch.ones([1, 14, model.w_dim])[:, -1:, :].repeat(1, 3, 1)
    feat, img = model.block0(x.clone(), x.clone(), ws, noise_mode="random")
    feat, y = model.block1(feat, img, ws, noise_mode="random")
    torch.manual_seed(14)
    target = torch.rand(1, 3, 512, 512)
    loss = ((y - target) ** 2).mean()
    loss.backward()
    yn = y.detach().numpy()
    out["train.crops"] = np.stack([yn[0, :, r:r + 32, c:c + 32] for r, c in CROPS])
    out["train.sum"] = yn.astype(np.float64).sum(axis=(0, 2, 3))
    out["train.loss"] = np.array([float(loss.detach())])
    out["train.grad_input_crop"] = x.grad.numpy()[0, :, 100:132, 60:92].copy()
    out["train.grad_input_abs_sum"] = np.array([np.abs(x.grad.numpy()).astype(np.float64).sum()])
    named = dict(model.named_parameters())
    for name in ("block0.conv0.weight", "block0.conv1.affine.bias", "block0.conv1.noise_strength", "block0.torgb.weight", "block1.conv0.weight",
                 "block1.conv0.affine.weight", "block1.conv1.bias", "block1.torgb.bias"):
        out["train.grad." + name] = named[name].grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "sr_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
