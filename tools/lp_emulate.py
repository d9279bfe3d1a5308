"""CPU emulation of the 16-bit head kernels' rounding points (csrc/frame_head_lp.hip), layer group by layer group.

Which operands of RADNeRF.forward (radnerf.py:108-141) can carry 8 significant bits (bf16) before a FRAME leaves SURVEY 8c's 16-bit tolerance
(PSNR >= 45 dB against the fp32 render, max-abs <= 2e-2)?  The kernels round an activation exactly where it becomes an MFMA operand and accumulate
in fp32; this script restates that on the CPU oracle's own render loop (oracle.head_forward replaced by the emulation below, everything else --
marcher, compositing, loop control -- the oracle's) and renders the non-convex scenes of tests/test_render_scenes_gpu.py in a grid of per-group
operand types:

    amb   ambient_net (two wide layers + the three skinny rows): its output is a COORDINATE of the second hash grid
    sig   sigma_net's two wide layers
    dens  the density row (sigma_net.net.2 row 0): the logit's absolute error is sigma's relative error
    col   merged geo / colour layer + colour rows

operand types: f32 | f16 | bf16 | bf16x2 (activations as hi + lo, two MFMAs) | bf16x3 (weights split too, three MFMAs).
Test infrastructure (imports oracle/): never imported by the package.  Usage: python tools/lp_emulate.py [--hw 128] [--scene shell]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc  # noqa: E402

f32 = np.float32


def rnd(x, kind):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=f32))
    if kind == "f16":
        return t.to(torch.float16).float().numpy()
    if kind == "bf16":
        return t.to(torch.bfloat16).float().numpy()
    return t.numpy()


def lin(x, W, kind):
    """x [M, K] @ W[out, K].T with the operands rounded as `kind` says, fp32 accumulation (emulated in fp64, rounded once)."""
    x = np.asarray(x, f32)
    W = np.asarray(W, f32)
    if kind == "f32":
        return (x.astype(np.float64) @ W.astype(np.float64).T).astype(f32)
    if kind in ("f16", "bf16"):
        return (rnd(x, kind).astype(np.float64) @ rnd(W, kind).astype(np.float64).T).astype(f32)
    xh = rnd(x, "bf16")
    xl = rnd(x - xh, "bf16")
    Wh = rnd(W, "bf16")
    if kind == "bf16x2":
        return ((xh.astype(np.float64) + xl.astype(np.float64)) @ Wh.astype(np.float64).T).astype(f32)
    if kind == "bf16x3":
        Wl = rnd(W - Wh, "bf16")
        return ((xh.astype(np.float64) + xl.astype(np.float64)) @ Wh.astype(np.float64).T + xh.astype(np.float64) @ Wl.astype(np.float64).T).astype(f32)
    raise ValueError(kind)


def make_forward(cfg, table_kind="f16"):
    """head_forward with the kernels' rounding points.  cfg: {'amb','sig','dens','col'} -> operand type."""
    cache = {}

    def tables(params):
        key = id(params["position_embedder.embeddings"])
        if key not in cache:
            cache[key] = (rnd(params["position_embedder.embeddings"], table_kind), rnd(params["ambient_embedder.embeddings"], table_kind))
        return cache[key]

    def forward(position, direction, cond_feat, ind_code, params, hp):
        position = np.asarray(position, f32)
        pos_spec, amb_spec = orc.head_grid_specs(hp)
        tpos, tamb = tables(params)
        cond = np.asarray(cond_feat, f32).reshape(-1)
        A0, A1, A2 = orc._mlp_weights(params, "ambient_net")
        S0, S1, S2 = orc._mlp_weights(params, "sigma_net")
        C0, C1 = orc._mlp_weights(params, "color_net")
        pos_feat = pos_spec.encode(position, tpos, bound=hp["bound"])
        # ambient_net: the conditioning columns are a per-frame fp32 bias (k_fold_constants)
        bias_a = (A0[:, 32:].astype(np.float64) @ cond.astype(np.float64)).astype(f32)
        ka = cfg["amb"] if isinstance(cfg["amb"], tuple) else (cfg["amb"],) * 3        # per layer: amb0, amb1, the three skinny rows
        h = np.maximum(lin(pos_feat, A0[:, :32], ka[0]) + bias_a, 0)
        h = np.maximum(lin(h, A1, ka[1]), 0)
        amb = np.tanh(lin(h, A2, ka[2])).astype(f32)
        amb_feat = amb_spec.encode(amb, tamb, bound=1)
        h = np.maximum(lin(np.concatenate([pos_feat, amb_feat], 1), S0, cfg["sig"]), 0)
        h = np.maximum(lin(h, S1, cfg["sig"]), 0)
        sigma = np.exp(lin(h, S2[:1], cfg["dens"])[:, 0]).astype(f32)
        # merged geo / colour layer (host product in fp64), individual code folded into an fp32 bias
        merged = np.concatenate([C0[:, :16].astype(np.float64), C0[:, 16:144].astype(np.float64) @ S2[1:129].astype(np.float64)], 1).astype(f32)
        sh = orc.sh_encode(direction, 4)
        bias_c = 0.0
        if ind_code is not None:
            bias_c = (C0[:, 144:].astype(np.float64) @ np.asarray(ind_code, np.float64).reshape(-1)).astype(f32)
        hc = np.maximum(lin(np.concatenate([sh, h], 1), merged, cfg["col"]) + bias_c, 0)
        color = orc.sigmoid(lin(hc, C1, cfg["col"]))
        return sigma, color, amb
    return forward


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / max(mse, 1e-20))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--scene", default="shell", choices=["shell", "speckle", "ellipsoid"])
    ap.add_argument("--variant", default="may_head")
    ap.add_argument("--T", type=float, default=0.01)
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
    from helpers import frame_case, nonconvex_occupancy, oracle_render
    case = frame_case(args.variant, args.hw)
    if args.scene != "ellipsoid":
        case = nonconvex_occupancy(case, args.scene)
    case["T_thresh"] = args.T
    true_forward = orc.head_forward
    t0 = time.time()
    ref = oracle_render(orc, case)["rgb_map"].reshape(-1, 3)
    print(f"fp32 reference rendered in {time.time() - t0:.1f} s ({args.scene}, {args.variant}, {args.hw}^2, T_thresh {args.T})", flush=True)
    grid = {
        "all f16": dict(amb="f16", sig="f16", dens="f16", col="f16"),
        "all bf16": dict(amb="bf16", sig="bf16", dens="bf16", col="bf16"),
        "amb f32, rest bf16": dict(amb="f32", sig="bf16", dens="bf16", col="bf16"),
        "amb f16, rest bf16": dict(amb="f16", sig="bf16", dens="bf16", col="bf16"),
        "amb bf16x3, rest bf16": dict(amb="bf16x3", sig="bf16", dens="bf16", col="bf16"),
        "amb bf16x2, rest bf16": dict(amb="bf16x2", sig="bf16", dens="bf16", col="bf16"),
        "amb (bf16,bf16,f16)": dict(amb=("bf16", "bf16", "f16"), sig="bf16", dens="bf16", col="bf16"),
        "amb (bf16,f16,f16)": dict(amb=("bf16", "f16", "f16"), sig="bf16", dens="bf16", col="bf16"),
        "amb (f16,f16,bf16)": dict(amb=("f16", "f16", "bf16"), sig="bf16", dens="bf16", col="bf16"),
        "amb (bf16x2,bf16x2,bf16x3)": dict(amb=("bf16x2", "bf16x2", "bf16x3"), sig="bf16", dens="bf16", col="bf16"),
        "amb f16 + dens f16": dict(amb="f16", sig="bf16", dens="f16", col="bf16"),
        "amb f16 + sig f16 + dens f16": dict(amb="f16", sig="f16", dens="f16", col="bf16"),
        "amb f16 + col f16": dict(amb="f16", sig="bf16", dens="bf16", col="f16"),
        "sig,dens,col f16, amb bf16": dict(amb="bf16", sig="f16", dens="f16", col="f16"),
        "amb f16, sig bf16x2, dens bf16x2": dict(amb="f16", sig="bf16x2", dens="bf16x2", col="bf16"),
        "amb f16, sig bf16x3, dens bf16x3": dict(amb="f16", sig="bf16x3", dens="bf16x3", col="bf16"),
    }
    if args.configs:
        grid = {k: v for k, v in grid.items() if any(s in k for s in args.configs.split(";"))}
    try:
        for name, cfg in grid.items():
            orc.head_forward = make_forward(cfg)
            t0 = time.time()
            rgb = oracle_render(orc, case)["rgb_map"].reshape(-1, 3)
            err = np.abs(rgb - ref).max(axis=1)
            print(f"{name:38s} psnr {psnr(rgb, ref):6.2f} dB  max {err.max():.4f}  frac>2e-2 {float((err > 2e-2).mean()):.5f}  frac>5e-2 {float((err > 5e-2).mean()):.5f}"
                  f"  ({time.time() - t0:.0f} s)", flush=True)
    finally:
        orc.head_forward = true_forward


if __name__ == "__main__":
    main()
