#!/usr/bin/env python
"""Generate tests/golden/postnet_golden.npz by running the REFERENCE's own LLE functions (modules/postnet/lle.py:8-93) and the landmark
branch of its conditioning preparation (inference/genefacepp_infer.py:335-344, 389-396, 407, 421-423; keypoint_mode 'lm68', no postnet model,
blink_mode 'none') on seeded inputs, on the CPU.

Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_golden_postnet.py

The LLE functions are imported from the reference as they are.  The surrounding lines of genefacepp_infer.py sit inside a method of the
inference class (which loads checkpoints and the 3DMM at construction), so they are executed here in the reference's order with the reference's
own calls: torch mean / std / quantile, compute_LLE_projection, and get_audio_features executed from the source text of modules/radnerfs/utils.py
(that module imports the CUDA-only training stack at import time).  Not reproduced: the round trip through the 3DMM's mean shape (:399, 405 --
algebraically the identity; the key_mean_shape asset is not in the mount) and the blink injection."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from modules.postnet.lle import compute_LLE_projection, find_k_nearest_neighbors, solve_LLE_projection_batch  # noqa: E402

SMO = 3


def reference_get_audio_features():
    src = open(os.path.join(REF, "modules/radnerfs/utils.py")).read()
    body = src[src.index("def get_audio_features"):src.index("@torch.jit.script", src.index("def get_audio_features"))]
    ns = {"torch": torch, "hparams": {"smo_win_size": SMO}}
    exec(body, ns)
    return ns["get_audio_features"]


def inputs():
    g = torch.Generator().manual_seed(20240922)
    # a low-dimensional "person manifold" plus noise, and predictions near it: the situation the projection is made for
    basis = torch.randn(12, 68 * 3, generator=g)
    ds = (torch.randn(400, 12, generator=g) @ basis * 0.3 + 0.05 * torch.randn(400, 68 * 3, generator=g)).reshape(400, 68, 3)
    pred = (torch.randn(30, 12, generator=g) @ basis * 0.3 + 0.2 * torch.randn(30, 68 * 3, generator=g)).reshape(30, 68, 3)
    return ds, pred


def main():
    ds, pred = inputs()
    out = {"ds": ds.numpy(), "pred": pred.numpy(), "smo_win_size": np.int32(SMO)}
    feats, base = pred.reshape(-1, 204), ds.reshape(-1, 204)
    for K in (10, 4, 1):
        ind = find_k_nearest_neighbors(feats, base, K)
        fuse, err, w = compute_LLE_projection(feats, base, K)
        out[f"knn_K{K}"] = ind.numpy()
        out[f"fuse_K{K}"] = fuse.numpy()
        out[f"weights_K{K}"] = w.numpy()
        if err is not None:
            out[f"errors_K{K}"] = err.numpy()
    fuse2, err2, w2 = solve_LLE_projection_batch(feats[:5], base[:50].reshape(5, 10, 204))
    out["solve_fuse"], out["solve_errors"], out["solve_weights"] = fuse2.numpy(), err2.numpy(), w2.numpy()

    # genefacepp_infer.py:335-344
    idexp_lm3d_ds = ds
    mean = idexp_lm3d_ds.mean(dim=0, keepdim=True)
    std = idexp_lm3d_ds.std(dim=0, keepdim=True)
    normalized_ds = (idexp_lm3d_ds - mean) / std
    lower = torch.quantile(normalized_ds, q=0.03, dim=0)
    upper = torch.quantile(normalized_ds, q=0.97, dim=0)
    gaf = reference_get_audio_features()
    for lle_percent in (0.2, 1.0, 0.0):
        # :389-396
        idexp_lm3d = pred.reshape([-1, 68 * 3]).clone()
        ds_lle = idexp_lm3d_ds.reshape([-1, 68 * 3])
        feat_fuse, _, _ = compute_LLE_projection(feats=idexp_lm3d[:, :68 * 3], feat_database=ds_lle[:, :68 * 3], K=10)
        idexp_lm3d[:, :68 * 3] = lle_percent * feat_fuse + (1 - lle_percent) * idexp_lm3d[:, :68 * 3]
        idexp_lm3d = idexp_lm3d.reshape([-1, 68, 3])
        normalized = (idexp_lm3d - mean) / std
        # :407
        normalized = torch.clamp(normalized, min=lower, max=upper)
        # :421-423
        cond_win = normalized.reshape([len(normalized), 1, -1])
        cond_wins = torch.stack([gaf(cond_win, att_mode=2, index=idx) for idx in range(len(cond_win))])
        tag = str(lle_percent).replace(".", "p")
        out[f"normalized_{tag}"] = normalized.numpy()
        out[f"cond_wins_{tag}"] = cond_wins.numpy()
    out["mean"], out["std"], out["lower"], out["upper"] = mean.numpy(), std.numpy(), lower.numpy(), upper.numpy()
    path = os.path.join(HERE, "postnet_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
