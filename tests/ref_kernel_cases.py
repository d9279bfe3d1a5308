"""Seeded input cases for pinning the native kernels against the reference's OWN extensions (oracle/_ref).

One table, three back ends with the reference's pybind signatures (raymarching.h:7-19, gridencoder.h:12-15, shencoder.h,
freqencoder.h):

    ref  oracle/_ref/*.so      the reference's .cu files compiled unmodified for gfx950 (oracle/build_ref.py)   -- GPU
    hip  compat_ext            the product's C ABI (libgfpp_radnerf.so)                                         -- GPU
    orc  oracle/ref_backends   the CPU restatement radnerf_oracle.c                                             -- CPU

`tests/golden/make_golden_ref_kernels.py` runs the `small` table through `ref` on an MI355X and commits the outputs
(tests/golden/ref_kernel_golden.npz); tests/test_oracle_ref_kernels_cpu.py checks `orc` against that fixture on CPU, and
tests/test_ref_kernels_gpu.py checks ref vs orc vs hip live (both scales).  Inputs are never stored: they are regenerated
here from the seeds, in numpy only.

Comparison specs per output: "exact" (bit for bit), ("close", atol, rtol), or a named canonicalisation for outputs whose
layout depends on the order of atomics (march_rays_train) or on a threshold tie (composite_rays).
"""
import zlib

import numpy as np

f32 = np.float32

SCALES = {"small": 1, "full": 8}


# ---- numpy helpers (independent of the oracle) -----------------------------------------------------------------------
def _spread3(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3D_np(x, y, z):
    return (_spread3(x) | (_spread3(y) << np.uint32(1)) | (_spread3(z) << np.uint32(2))).astype(np.uint32)


def make_density(C, H, seed):
    """Morton-ordered density grid [C, H^3]: an ellipsoid (world space, cascade c spans [-2^c, 2^c]) plus speckle."""
    rng = np.random.default_rng(seed)
    ii, jj, kk = np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij")
    m = morton3D_np(ii.ravel(), jj.ravel(), kk.ravel())
    grid = np.zeros((C, H ** 3), f32)
    for c in range(C):
        b = float(2 ** c)
        x = ((ii.ravel() + 0.5) / H * 2 - 1) * b
        y = ((jj.ravel() + 0.5) / H * 2 - 1) * b
        z = ((kk.ravel() + 0.5) / H * 2 - 1) * b
        inside = (x / 0.45) ** 2 + (y / 0.35) ** 2 + (z / 0.55) ** 2 < 1.0
        speck = rng.random(H ** 3) < 0.02
        dens = np.where(inside | speck, rng.uniform(11, 40, H ** 3), rng.uniform(0, 9.9, H ** 3)).astype(f32)
        grid[c, m] = dens
    return grid


def packbits_np(grid, thresh):
    flat = (grid.reshape(-1) > thresh).astype(np.uint8).reshape(-1, 8)
    return (flat << np.arange(8, dtype=np.uint8)).sum(axis=1).astype(np.uint8)


def camera_rays(n, seed, spread=0.25):
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (n, 3)) * 0.05 + np.array([0, 4.0, 0])).astype(f32)
    d = rng.standard_normal((n, 3)) * spread * np.array([1, 0, 1]) + np.array([0, -1, 0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(f32)


def edge_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (n, 3)) * 0.3 + np.array([0, 4.0, 0])).astype(f32)
    d = rng.standard_normal((n, 3)) * 0.2 + np.array([0, -1, 0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(f32)
    d[0] = [0, -1, 0]; d[1] = [1, 0, 0]; d[2] = [0, 0, 1]; o[3] = [0, 0, 0]; d[4] = [0, 1, 0]
    o[5] = [-1.0, 4.0, 0.0]; d[5] = [0, -1, 0]
    return o, d


def slab_near_far(o, d, aabb, min_near):
    """Plain numpy slab test (only used to make plausible nears/fars *inputs*; the kernel itself is case `near_far`)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1) / d).astype(f32)
        t0 = ((aabb[:3] - o) * inv).astype(f32)
        t1 = ((aabb[3:] - o) * inv).astype(f32)
    near = np.nanmax(np.minimum(t0, t1), axis=1)
    far = np.nanmin(np.maximum(t0, t1), axis=1)
    miss = far < near
    near = np.maximum(near, min_near)
    big = np.finfo(f32).max
    return np.where(miss, big, near).astype(f32), np.where(miss, big, far).astype(f32)


def grid_offsets(D, L, C, per_level_scale, H, log2_hashmap, align_corners=False):
    """GridEncoder.__init__ (grid.py:117-132): per-level row counts, rounded up to multiples of 8."""
    max_params = 2 ** log2_hashmap
    offs, off = [], 0
    for i in range(L):
        res = int(np.ceil(H * per_level_scale ** i))
        n = min(max_params, (res if align_corners else res + 1) ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return np.array(offs, np.int32)


class Case:
    def __init__(self, name, module, fn, args, outs, canon=None, skip=()):
        self.name, self.module, self.fn, self.args, self.outs, self.canon, self.skip = name, module, fn, args, outs, canon, tuple(skip)

    def __repr__(self):
        return self.name


def _z(*shape, dtype=f32):
    return np.zeros(shape, dtype)


# ---- canonicalisers ----------------------------------------------------------------------------------------------------
def canon_march_train(outs, args):
    """march_rays_train allocates each ray's sample range with atomicAdd (raymarching.cu:446-447): offsets depend on the
    order the rays got there.  Canonical form: rays sorted by ray index, samples re-packed in that order."""
    xyzs, dirs, deltas, rays, counter = outs[12], outs[13], outs[14], outs[15], outs[16]
    M = xyzs.shape[0]
    order = np.argsort(rays[:, 0], kind="stable")
    r = rays[order]
    cx, cd, cdl, counts = [], [], [], []
    for idx, off, cnt in r:
        ok = off + cnt <= M
        cnt = cnt if ok else 0
        counts.append(cnt)
        cx.append(xyzs[off:off + cnt]); cd.append(dirs[off:off + cnt]); cdl.append(deltas[off:off + cnt])
    return {"ray_index": r[:, 0].copy(), "ray_count": np.array(counts, np.int32),
            "xyzs": np.concatenate(cx), "dirs": np.concatenate(cd), "deltas": np.concatenate(cdl), "counter": counter.copy()}


# ---- the table ---------------------------------------------------------------------------------------------------------
def cases(scale="small"):
    k = SCALES[scale]
    out = []
    aabb1 = np.array([-1, -0.5, -1, 1, 0.5, 1], f32)

    # -- _raymarching_face: helpers --
    o, d = edge_rays(1000 * k, 1)
    out.append(Case("near_far", "_raymarching_face", "near_far_from_aabb", [o, d, aabb1, len(o), 0.05, _z(len(o)), _z(len(o))],
                    {5: "exact", 6: "exact"}))
    o2, d2 = camera_rays(500 * k, 2, spread=0.6)
    o2 = (o2 * f32(0.3)).astype(f32)
    out.append(Case("sph_from_ray", "_raymarching_face", "sph_from_ray", [o2, d2, 2.0, len(o2), _z(len(o2), 2)], {4: ("close", 2e-6, 0)}))
    rng = np.random.default_rng(3)
    coords = rng.integers(0, 128, (1024 * k, 3)).astype(np.int32)
    out.append(Case("morton3D", "_raymarching_face", "morton3D", [coords, len(coords), _z(len(coords), dtype=np.int32)], {2: "exact"}))
    codes = rng.integers(0, 128 ** 3, 1024 * k).astype(np.int32)
    out.append(Case("morton3D_invert", "_raymarching_face", "morton3D_invert", [codes, len(codes), _z(len(codes), 3, dtype=np.int32)], {2: "exact"}))
    dens = make_density(2, 32, 4)
    out.append(Case("packbits", "_raymarching_face", "packbits", [dens, dens.size // 8, 10.0, _z(dens.size // 8, dtype=np.uint8)], {3: "exact"}))
    out.append(Case("morton3D_dilation", "_raymarching_face", "morton3D_dilation", [dens, 2, 32, _z(*dens.shape)], {3: "exact"}))

    # -- march_rays / composite_rays (inference) --
    H = 128
    bit1 = packbits_np(make_density(1, H, 5), 10.0)
    bit2 = packbits_np(make_density(2, H, 6), 10.0)
    n_rays = 256 * k if scale == "small" else 8192
    for tag, C, bound, bits, aabb, dt_gamma in (("c1", 1, 1.0, bit1, aabb1, 1 / 256), ("c2", 2, 2.0, bit2, np.array([-2, -2, -2, 2, 2, 2], f32), 1 / 256),
                                                ("c2_dt0", 2, 2.0, bit2, np.array([-2, -2, -2, 2, 2, 2], f32), 0.0)):
        o, d = camera_rays(n_rays, 7 + C)
        nears, fars = slab_near_far(o, d, aabb, f32(0.05))
        rng = np.random.default_rng(8 + C)
        alive = np.sort(rng.choice(n_rays, n_rays // 2, replace=False)).astype(np.int32)
        rays_t = (nears + rng.uniform(0, 0.5, n_rays)).astype(f32)
        for n_step in ((1, 2, 3, 8) if tag == "c1" else (2, 8)):
            for perturb in ((False, True) if n_step == 2 else (False,)):
                n_alive = len(alive)
                M = n_alive * n_step
                M += 128 - M % 128
                noises = rng.random(n_alive).astype(f32) if perturb else _z(n_alive)
                out.append(Case(f"march_rays_{tag}_s{n_step}{'_noise' if perturb else ''}", "_raymarching_face", "march_rays",
                                [n_alive, n_step, alive, rays_t, o, d, bound, dt_gamma, 16, C, H, bits, nears, fars, _z(M, 3), _z(M, 3), _z(M, 2), noises],
                                {14: "exact", 15: "exact", 16: "exact"}))
    for n_step in (1, 4):
        rng = np.random.default_rng(20 + n_step)
        N, n_alive = 1000 * k, 400 * k
        alive = np.sort(rng.choice(N, n_alive, replace=False)).astype(np.int32)
        M = n_alive * n_step
        sig = np.exp(rng.uniform(-2, 6, M)).astype(f32)
        rgb = rng.uniform(0, 1, (M, 3)).astype(f32)
        deltas = np.stack([np.full(M, 0.027063, f32), rng.uniform(3.5, 4.5, M).astype(f32)], 1)
        deltas[rng.random(M) < 0.1] = 0
        ws = rng.uniform(0, 0.995, N).astype(f32)
        out.append(Case(f"composite_rays_s{n_step}", "_raymarching_face", "composite_rays",
                        [n_alive, n_step, 0.01, alive, rng.uniform(3.5, 4.5, N).astype(f32), sig, rgb, deltas, ws, rng.uniform(0, 4, N).astype(f32),
                         rng.uniform(0, 1, (N, 3)).astype(f32)],
                        {3: "alive", 4: "by_alive", 8: ("by_alive_close", 2e-6), 9: ("by_alive_close", 2e-5), 10: ("by_alive_close", 2e-6)}))

    # -- march_rays_train / composite_rays_train --
    for tag, C, bound, bits, aabb, perturb in (("c1", 1, 1.0, bit1, aabb1, False), ("c2_noise", 2, 2.0, bit2, np.array([-2, -2, -2, 2, 2, 2], f32), True)):
        N = 300 * k
        o, d = camera_rays(N, 30 + C)
        nears, fars = slab_near_far(o, d, aabb, f32(0.05))
        rng = np.random.default_rng(31 + C)
        max_steps = 64
        M = N * max_steps
        noises = rng.random(N).astype(f32) if perturb else _z(N)
        out.append(Case(f"march_rays_train_{tag}", "_raymarching_face", "march_rays_train",
                        [o, d, bits, bound, 1 / 256, max_steps, N, C, H, M, nears, fars, _z(M, 3), _z(M, 3), _z(M, 2), _z(N, 3, dtype=np.int32),
                         _z(2, dtype=np.int32), noises],
                        {12: None, 13: None, 14: None, 15: None, 16: None}, canon=canon_march_train))
    # a packed sample layout for the training compositor: rays with 0..24 samples, contiguous ranges in ray order
    rng = np.random.default_rng(40)
    N = 300 * k
    cnt = rng.integers(0, 25, N).astype(np.int32)
    cnt[:4] = [0, 1, 24, 0]
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    M = int(cnt.sum())
    perm = rng.permutation(N).astype(np.int32)
    rays = np.stack([perm, off, cnt], 1).astype(np.int32)
    sig = np.exp(rng.uniform(-2, 5, M)).astype(f32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(f32)
    amb = rng.uniform(0, 0.1, M).astype(f32)
    dl = np.stack([np.full(M, 0.027063, f32), rng.uniform(3.5, 4.5, M).astype(f32)], 1)
    out.append(Case("composite_rays_train_forward", "_raymarching_face", "composite_rays_train_forward",
                    [sig, rgb, amb, dl, rays, M, N, 1e-4, _z(N), _z(N), _z(N), _z(N, 3)],
                    {8: ("close", 2e-6, 1e-6), 9: ("close", 2e-6, 1e-6), 10: ("close", 2e-5, 1e-6), 11: ("close", 2e-6, 1e-6)}))
    out.append(Case("march_rays_train_backward", "_raymarching_face", "march_rays_train_backward",
                    [rng.standard_normal((M, 3)).astype(f32), rng.standard_normal((M, 3)).astype(f32), rays, dl, N, M, _z(N, 3), _z(N, 3)],
                    {6: "exact", 7: "exact"}))
    # the backward consumes the forward's sums; they are produced by the CPU oracle so that every back end gets identical inputs
    from oracle import oracle as orc
    ws, ambs, _, img = orc.composite_rays_train_forward(sig, rgb, amb, dl, rays, 1e-4)
    out.append(Case("composite_rays_train_backward", "_raymarching_face", "composite_rays_train_backward",
                    [rng.standard_normal(N).astype(f32), rng.standard_normal(N).astype(f32), rng.standard_normal((N, 3)).astype(f32), sig, rgb, amb, dl, rays,
                     ws, ambs, img, M, N, 1e-4, _z(M), _z(M, 3), _z(M)],
                    {14: ("close", 5e-4, 1e-3), 15: ("close", 1e-5, 1e-4), 16: ("close", 1e-5, 1e-4)}))

    # -- _gridencoder --
    GT = {"hash": 0, "tiled": 1}
    IT = {"linear": 0, "smoothstep": 1}
    B = 128 if scale == "small" else 8192
    for D, gt, it, C, dt, ac in ((3, "tiled", "linear", 2, f32, False), (2, "tiled", "linear", 2, f32, False), (3, "hash", "linear", 2, f32, False),
                                 (3, "tiled", "smoothstep", 2, f32, False), (2, "hash", "smoothstep", 4, f32, False), (3, "hash", "linear", 1, f32, False),
                                 (3, "tiled", "linear", 8, f32, False), (3, "tiled", "linear", 2, f32, True), (3, "hash", "linear", 2, np.float16, False),
                                 (2, "tiled", "linear", 2, np.float16, False)):
        L, Hb, pls = 16, 16, 2 ** (7 / 15)
        offs = grid_offsets(D, L, C, pls, Hb, 16, ac)
        rng = np.random.default_rng(zlib.crc32(f"{D}{gt}{it}{C}{ac}".encode()))
        emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(dt)
        u = rng.uniform(0, 1, (B, D)).astype(f32)
        u[0] = 0.0; u[1] = 1.0; u[2] = -0.001; u[3, 0] = 1.0001; u[4] = 0.5
        S = float(np.log2(pls))
        tag = f"D{D}_{gt}_{it}_C{C}_{'f16' if dt is np.float16 else 'f32'}{'_ac' if ac else ''}"
        exact = "exact" if dt is f32 else ("close", 4e-3, 4e-3)
        out.append(Case(f"grid_forward_{tag}", "_gridencoder", "grid_encode_forward",
                        [u, emb, offs, _z(L, B, C, dtype=dt), B, D, C, L, S, Hb, None, GT[gt], ac, IT[it]], {3: exact}))
        # fp32: the derivative is the same fmaf chain on both sides.  Half tables: the reference accumulates in half (gridencoder.cu:163,186)
        # and dy_dx reaches |scale * 2| ~ 4096, where one half ulp is 4
        dy = "exact" if dt is f32 else ("close", 8.0, 4e-3)
        out.append(Case(f"grid_forward_dydx_{tag}", "_gridencoder", "grid_encode_forward",
                        [u, emb, offs, _z(L, B, C, dtype=dt), B, D, C, L, S, Hb, _z(B, L * D * C, dtype=dt), GT[gt], ac, IT[it]], {3: exact, 10: dy}))
        # backward / TV write table-sized outputs: the committed fixture uses 2^10-row levels, the live GPU comparison the shipping 2^16
        offs_b = offs if scale != "small" else grid_offsets(D, L, C, pls, Hb, 10, ac)
        emb_b = emb if scale != "small" else rng.uniform(-1, 1, (int(offs_b[-1]), C)).astype(dt)
        grad = (rng.standard_normal((L, B, C)) * (1e-2 if dt is np.float16 else 1.0)).astype(dt)
        dydx_in = rng.standard_normal((B, L * D * C)).astype(dt)
        gtol = ("close", 2e-5, 2e-5) if dt is f32 else ("close", 2e-3, 2e-2)
        out.append(Case(f"grid_backward_{tag}", "_gridencoder", "grid_encode_backward",
                        [grad, u, emb_b, offs_b, _z(*emb_b.shape, dtype=dt), B, D, C, L, S, Hb, dydx_in, _z(B, D, dtype=dt), GT[gt], ac, IT[it]],
                        {4: gtol, 12: ("close", 2e-4, 2e-5) if dt is f32 else ("close", 2e-1, 2e-2)}))
        # (grad_total_variation reads `inputs` with the table dtype, gridencoder.cu:642: with half tables the reference itself raises on fp32 inputs)
        if it == "linear" and not ac and dt is f32:
            g0 = rng.standard_normal(emb_b.shape).astype(dt)
            out.append(Case(f"grid_tv_{tag}", "_gridencoder", "grad_total_variation",
                            [u, emb_b, g0, offs_b, 1e-3 if dt is f32 else 1.0, B, D, C, L, S, Hb, GT[gt], ac],
                            {2: ("close", 2e-6, 2e-6) if dt is f32 else ("close", 2e-2, 2e-2)}))

    # -- _shencoder / _freqencoder --
    rng = np.random.default_rng(50)
    dirs = rng.standard_normal((700 * k, 3)).astype(f32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    nB = len(dirs)
    for deg in (1, 2, 3, 4):
        out.append(Case(f"sh_forward_deg{deg}", "_shencoder", "sh_encode_forward", [dirs, _z(nB, deg * deg), nB, 3, deg, None], {1: ("close", 1e-6, 0)}))
    out.append(Case("sh_forward_dydx_deg4", "_shencoder", "sh_encode_forward", [dirs, _z(nB, 16), nB, 3, 4, _z(nB, 48)], {1: ("close", 1e-6, 0), 5: ("close", 4e-6, 0)}))
    dy_in = rng.standard_normal((nB, 48)).astype(f32)
    out.append(Case("sh_backward_deg4", "_shencoder", "sh_encode_backward", [rng.standard_normal((nB, 16)).astype(f32), dirs, nB, 3, 4, dy_in, _z(nB, 3)],
                    {6: ("close", 2e-5, 1e-5)}))
    for D, deg, sc in ((2, 10, 0.8), (6, 4, 4.0), (14, 4, 1.0)):
        x = (rng.uniform(-1, 1, (300 * k, D)) * sc).astype(f32)
        Cc = D + 2 * D * deg
        # __sinf (freqencoder.cu:56) is a fast intrinsic with absolute error that grows with |argument| (up to 2^9 * 0.8 here)
        out.append(Case(f"freq_forward_D{D}_deg{deg}", "_freqencoder", "freq_encode_forward", [x, len(x), D, deg, Cc, _z(len(x), Cc)], {5: ("close", 5e-4, 0)}))
        outs_in = rng.uniform(-1, 1, (len(x), Cc)).astype(f32)
        out.append(Case(f"freq_backward_D{D}_deg{deg}", "_freqencoder", "freq_encode_backward",
                        [rng.standard_normal((len(x), Cc)).astype(f32), outs_in, len(x), D, deg, Cc, _z(len(x), D)], {6: "exact"}))
    return out


# ---- running and comparing ---------------------------------------------------------------------------------------------
def run_case(case, mods, device, f32_only=False):
    """Call `case` on a back end.  `mods` maps module name -> object with the pybind functions.  Returns {arg index: numpy}
    (or the canonicalised dict).  `f32_only`: the CPU oracle has fp32 tables only -- half arguments are up-cast for the call."""
    import torch
    targs, halves = [], []
    for i, a in enumerate(case.args):
        if isinstance(a, np.ndarray):
            t = torch.from_numpy(a.copy())
            if f32_only and t.dtype == torch.float16:
                t = t.float(); halves.append(i)
            targs.append(t.to(device))
        elif isinstance(a, (float, np.floating)):
            targs.append(float(a))
        elif isinstance(a, (bool, np.bool_)):
            targs.append(bool(a))
        elif isinstance(a, (int, np.integer)):
            targs.append(int(a))
        else:
            targs.append(a)
    getattr(mods[case.module], case.fn)(*targs)
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    res = {}
    for i in case.outs:
        t = targs[i]
        if i in halves:
            t = t.half()
        res[i] = t.cpu().numpy()
    if case.canon is not None:
        res = case.canon(res, targs)
    return res


def compare(case, got, ref, label=""):
    """Assert `got` matches `ref` under the case's specs; returns {key: max abs diff}."""
    stats = {}
    if case.canon is not None:
        for key in ref:
            a, b = np.asarray(got[key]), np.asarray(ref[key])
            assert a.shape == b.shape, (case.name, key, a.shape, b.shape, label)
            stats[key] = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
            np.testing.assert_array_equal(a, b, err_msg=f"{case.name}:{key} {label}")
        return stats
    alive_idx = [i for i, s in case.outs.items() if s == "alive"]
    ok_rays = None
    if alive_idx:
        i = alive_idx[0]
        a, b = got[i], ref[i]
        mism = a != b
        stats["alive_mismatch_frac"] = float(mism.mean())
        assert mism.mean() < 2e-3, (case.name, label, stats)          # rays whose transmittance sits within rounding of T_thresh
        alive0 = case.args[3]
        N = case.args[4].shape[0]
        ok_rays = ~np.isin(np.arange(N), alive0[mism])
    for i, spec in case.outs.items():
        a, b = np.asarray(got[i]), np.asarray(ref[i])
        assert a.shape == b.shape and a.dtype == b.dtype, (case.name, i, a.shape, b.shape, a.dtype, b.dtype, label)
        if spec == "alive":
            continue
        if isinstance(spec, str) and spec == "by_alive":
            a, b, spec = a[ok_rays], b[ok_rays], "exact"
        elif isinstance(spec, tuple) and spec[0] == "by_alive_close":
            a, b, spec = a[ok_rays], b[ok_rays], ("close", spec[1], 0)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        stats[i] = float(np.nanmax(d)) if d.size else 0.0
        if spec == "exact":
            np.testing.assert_array_equal(a, b, err_msg=f"{case.name}: arg {i} {label}")
        else:
            _, atol, rtol = spec
            np.testing.assert_allclose(a.astype(np.float64), b.astype(np.float64), atol=atol, rtol=rtol, err_msg=f"{case.name}: arg {i} {label}")
    return stats
