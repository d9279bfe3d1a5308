"""Training-side oracle restatements (oracle/radnerf_oracle.c, second half) against self-derived invariants: the reference ships no
test for these kernels, so they are pinned the same way as the inference kernels -- closed forms, linearity, finite
differences of the (already pinned) forward kernels."""
import numpy as np
import pytest

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams

f32 = np.float32


@pytest.fixture(scope="module")
def orc(oracle_mod):
    return oracle_mod


@pytest.fixture(scope="module")
def scene():
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    rng = np.random.default_rng(3)
    N = 64
    o = np.tile(np.array([[0.0, 4.0, 0.0]], f32), (N, 1))
    d = np.stack([rng.uniform(-0.08, 0.08, N), -np.ones(N), rng.uniform(-0.1, 0.1, N)], 1).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return hp, sd, o, d


def test_march_train_equals_inference_marcher(orc, scene):
    """With one launch budget >= the ray's samples, march_rays_train must emit exactly the samples of march_rays (same stepping code)."""
    hp, sd, o, d = scene
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], f32)
    nears, fars = orc.near_far_from_aabb(o, d, aabb, 0.05)
    xyzs, dirs, deltas, rays, counter = orc.march_rays_train(o, d, 1.0, sd["density_bitfield"], 1, 128, nears, fars, dt_gamma=hp["dt_gamma"], max_steps=64)
    N = o.shape[0]
    assert counter[1] == N and counter[0] == rays[:, 2].sum()
    alive = np.arange(N, dtype=np.int32)
    x2, d2, t2 = orc.march_rays(N, 64, alive, nears.copy(), o, d, 1.0, sd["density_bitfield"], 1, 128, nears, fars, -1, False, hp["dt_gamma"], 64)
    x2, t2 = x2.reshape(N, 64, 3), t2.reshape(N, 64, 2)
    assert rays[:, 2].max() > 5
    for n in range(N):
        ray, off, cnt = rays[n]
        assert ray == n                                          # the oracle visits rays in order
        np.testing.assert_array_equal(xyzs[off:off + cnt], x2[n, :cnt])
        np.testing.assert_array_equal(deltas[off:off + cnt], t2[n, :cnt])
        assert (t2[n, cnt:, 0] == 0).all()
        np.testing.assert_array_equal(dirs[off:off + cnt], np.tile(d[n], (cnt, 1)))


def test_march_train_overflow_and_backward(orc, scene):
    hp, sd, o, d = scene
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], f32)
    nears, fars = orc.near_far_from_aabb(o, d, aabb, 0.05)
    full = orc.march_rays_train(o, d, 1.0, sd["density_bitfield"], 1, 128, nears, fars, dt_gamma=hp["dt_gamma"], max_steps=64)
    total = int(full[4][0])
    M = total // 2
    xyzs, dirs, deltas, rays, counter = orc.march_rays_train(o, d, 1.0, sd["density_bitfield"], 1, 128, nears, fars, M=M, dt_gamma=hp["dt_gamma"], max_steps=64)
    assert counter[0] == total                                   # the counter keeps counting, rays beyond M write nothing (:455)
    np.testing.assert_array_equal(rays, full[3])
    fits = rays[:, 1] + rays[:, 2] <= M
    assert fits.any() and (~fits).any()
    # backward: d(xyz)/d(o) = 1, d(xyz)/d(d) = t_end, d(dir)/d(d) = 1, summed over the ray's samples
    rng = np.random.default_rng(0)
    gx, gd = rng.standard_normal(full[0].shape).astype(f32), rng.standard_normal(full[1].shape).astype(f32)
    go, gdd = orc.march_rays_train_backward(gx, gd, full[3], full[2])
    for n in range(o.shape[0]):
        _, off, cnt = full[3][n]
        np.testing.assert_allclose(go[n], gx[off:off + cnt].sum(0), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gdd[n], (gx[off:off + cnt] * full[2][off:off + cnt, 1:2] + gd[off:off + cnt]).sum(0), rtol=1e-4, atol=1e-4)


def test_composite_train_closed_form_and_gradient(orc):
    """Constant sigma/rgb: weights_sum = 1 - e^{-sigma dt k}; backward == finite differences of the forward (fp64 recomputation)."""
    K, dt, sigma = 12, f32(0.03), f32(7.0)
    rays = np.array([[0, 0, K], [1, K, 0]], np.int32)             # second ray has no samples -> zeros
    sig = np.full(K, sigma, f32)
    rgb = np.tile(np.array([[0.2, 0.5, 0.9]], f32), (K, 1))
    amb = np.linspace(0.1, 0.2, K).astype(f32)
    deltas = np.stack([np.full(K, dt, f32), 3.5 + dt * np.arange(1, K + 1, dtype=f32)], 1)
    ws, asum, depth, img = orc.composite_rays_train_forward(sig, rgb, amb, deltas, rays, T_thresh=1e-4)
    np.testing.assert_allclose(ws[0], 1 - np.exp(-sigma * dt * K), rtol=1e-5)
    np.testing.assert_allclose(img[0], ws[0] * rgb[0], rtol=1e-5)
    np.testing.assert_allclose(asum[0], amb.sum(), rtol=1e-6)
    assert ws[1] == 0 and (img[1] == 0).all() and depth[1] == 0
    # early stop AFTER the update (T *= 1 - alpha; if T < thresh break): with a large threshold exactly one sample is used
    ws1, asum1, _, _ = orc.composite_rays_train_forward(sig, rgb, amb, deltas, rays, T_thresh=0.9)
    np.testing.assert_allclose(ws1[0], 1 - np.exp(-sigma * dt), rtol=1e-5)
    np.testing.assert_allclose(asum1[0], amb[0], rtol=1e-6)

    rng = np.random.default_rng(1)
    sig = rng.uniform(0.5, 20, K).astype(f32)
    rgb = rng.random((K, 3)).astype(f32)
    ws, asum, depth, img = orc.composite_rays_train_forward(sig, rgb, amb, deltas, rays)
    gws, gamb, gimg = np.array([0.7, 0.0], f32), np.array([0.3, 0.0], f32), np.array([[1.0, -2.0, 0.5], [0, 0, 0]], f32)
    gs, gr, ga = orc.composite_rays_train_backward(gws, gamb, gimg, sig, rgb, amb, deltas, rays, ws, asum, img)

    def loss(s64, c64):
        T, acc_ws, acc = 1.0, 0.0, np.zeros(3)
        for k in range(K):
            a = 1 - np.exp(-s64[k] * float(dt))
            w = a * T
            acc += w * c64[k]
            acc_ws += w
            T *= 1 - a
        return float(gws[0]) * acc_ws + float(gimg[0] @ acc)

    s64, c64 = sig.astype(np.float64), rgb.astype(np.float64)
    for k in (0, 3, K - 1):
        e = np.zeros(K); e[k] = 1e-5
        np.testing.assert_allclose(gs[k], (loss(s64 + e, c64) - loss(s64 - e, c64)) / 2e-5, rtol=2e-3, atol=1e-5)
        ec = np.zeros((K, 3)); ec[k, 1] = 1e-5
        np.testing.assert_allclose(gr[k, 1], (loss(s64, c64 + ec) - loss(s64, c64 - ec)) / 2e-5, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(ga, np.full(K, 0.3, f32))


def test_grid_backward_and_dydx(orc):
    """The forward is linear in the table: <grad, forward(E)> differentiated w.r.t. E is the backward scatter; dy_dx == finite differences."""
    rng = np.random.default_rng(2)
    for D, gridtype, interp in ((3, "tiled", "linear"), (2, "tiled", "linear"), (3, "hash", "smoothstep")):
        off, pls = syn.grid_offsets(D, log2_hashmap_size=12, desired_resolution=256)
        E = rng.standard_normal((int(off[-1]), 2)).astype(f32)
        B = 37
        x = rng.random((B, D)).astype(f32)
        x[0] = 1.5                                               # out of range: no gradient, zero dy_dx
        grad = rng.standard_normal((16, B, 2)).astype(f32)
        ge, _ = orc.grid_encode_backward(grad, x, E, off, pls, 16, gridtype, False, interp)
        # linearity: sum(grad * forward(E)) == sum(ge * E)
        fwd = orc.grid_encode_levels(x, E, off, pls, 16, gridtype, False, interp)
        np.testing.assert_allclose(float((grad.astype(np.float64) * fwd).sum()), float((ge.astype(np.float64) * E).sum()), rtol=2e-4)
        dy = orc.grid_encode_dydx(x, E, off, pls, 16, gridtype, False, interp)
        assert (dy[0] == 0).all()
        eps = 1e-4
        for b in (1, 5, 20):
            for d in range(D):
                xp, xm = x.copy(), x.copy()
                xp[b, d] += eps; xm[b, d] -= eps
                fd = (orc.grid_encode_levels(xp, E, off, pls, 16, gridtype, False, interp)[:, b] -
                      orc.grid_encode_levels(xm, E, off, pls, 16, gridtype, False, interp)[:, b]) / (2 * eps)
                coarse = slice(0, 6)                              # fine levels cross cell borders within +-eps
                np.testing.assert_allclose(dy[b, coarse, d], fd[coarse], rtol=5e-2, atol=5e-2)
        _, gi = orc.grid_encode_backward(grad, x, E, off, pls, 16, gridtype, False, interp, dy_dx=dy)
        np.testing.assert_allclose(gi, np.einsum("lbc,bldc->bd", grad.astype(np.float64), dy.astype(np.float64)), rtol=1e-4, atol=1e-4)


def test_dilation_tv_sph(orc):
    H, C = 16, 1
    rng = np.random.default_rng(4)
    dense = rng.random((H, H, H)).astype(f32)
    idx = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    mort = syn.morton3d(X, Y, Z)
    grid = np.zeros((C, H ** 3), f32)
    grid[0, mort.reshape(-1)] = dense.reshape(-1)
    out = orc.morton3D_dilation(grid, C, H)
    pad = np.pad(dense, 1, constant_values=-np.inf)
    ref = np.maximum.reduce([pad[1:-1, 1:-1, 1:-1], pad[2:, 1:-1, 1:-1], pad[:-2, 1:-1, 1:-1], pad[1:-1, 2:, 1:-1], pad[1:-1, :-2, 1:-1],
                             pad[1:-1, 1:-1, 2:], pad[1:-1, 1:-1, :-2]])
    np.testing.assert_array_equal(out[0, mort.reshape(-1)], ref.reshape(-1))
    # sph_from_ray: a ray from the origin along +x exits the unit sphere at theta = pi/2, phi = 0
    c = orc.sph_from_ray(np.zeros((1, 3), f32), np.array([[1.0, 0, 0]], f32), 1.0)
    np.testing.assert_allclose(c, [[0.0, 0.0]], atol=1e-6)
    # TV gradient: a constant table has zero variation
    off, pls = syn.grid_offsets(2, log2_hashmap_size=12, desired_resolution=256)
    E = np.ones((int(off[-1]), 2), f32)
    g = orc.grad_total_variation(rng.random((50, 2)).astype(f32), E, np.zeros_like(E), off, 1.0, pls, 16, "tiled", False)
    assert np.abs(g).max() == 0


def test_sh_and_freq_gradients_match_finite_differences(orc=None):
    import oracle.oracle as orc
    rng = np.random.default_rng(11)
    d = rng.standard_normal((40, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for degree in (1, 2, 3, 4):
        jac = orc.sh_encode_dydx(d, degree)
        eps = 1e-3
        for axis in range(3):
            step = np.zeros(3)
            step[axis] = eps
            num = (orc.sh_encode((d + step).astype(f32), degree).astype(np.float64) - orc.sh_encode((d - step).astype(f32), degree)) / (2 * eps)
            np.testing.assert_allclose(jac[:, axis], num, atol=2e-3)
        g = rng.standard_normal((40, degree * degree)).astype(f32)
        np.testing.assert_allclose(orc.sh_encode_backward(g, d, degree), np.einsum("bc,bdc->bd", g, jac), rtol=1e-5, atol=1e-5)
    # frequency encoding: d/dx [x, sin(2^k x), cos(2^k x)] = [1, 2^k cos, -2^k sin]
    x = rng.uniform(-1, 1, (30, 2)).astype(f32)
    for deg in (1, 4, 10):
        out = orc.freq_encode(x, deg)
        g = rng.standard_normal(out.shape).astype(f32)
        want = g[:, :2].astype(np.float64).copy()
        for k in range(deg):
            s = 2 + 4 * k
            want += 2.0 ** k * (g[:, s:s + 2] * np.cos(2.0 ** k * x.astype(np.float64)) - g[:, s + 2:s + 4] * np.sin(2.0 ** k * x.astype(np.float64)))
        np.testing.assert_allclose(orc.freq_encode_backward(g, out, 2, deg), want, rtol=2e-4, atol=2e-3 * 2.0 ** deg / 1024 + 1e-4)


def test_training_render_composition_matches_reference_python(orc=None):
    """tests/golden/ref_python_train_golden.npz comes from the reference's own NeRFRenderer.render (training branch) and autograd shims run on
    CPU over the oracle kernels (tests/golden/make_golden_train.py).  The oracle-side composition that the GPU tests compare the product with
    must reproduce its forward results, and mark_untrained_grid's visibility rule its mask."""
    import os
    import oracle.oracle as orc
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.configs import may_hparams
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_train_golden.npz"))
    HW = 24
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    pose = syn.synthetic_pose(0)[None]
    r = orc.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
    o, d = r["rays_o"].reshape(-1, 3), r["rays_d"].reshape(-1, 3)
    fi = syn.synthetic_frame_inputs(hp, 0)
    nears, fars = orc.near_far_from_aabb(o, d, sd["aabb_train"], hp["min_near"])
    xyzs, dirs, deltas, rays, counter = orc.march_rays_train(o, d, hp["bound"], sd["density_bitfield"], 1, hp["grid_size"], nears, fars,
                                                             dt_gamma=hp["dt_gamma"], max_steps=hp["max_steps"])
    M = int(counter[0])
    assert int(g["fwd.step_counter"][0, 0]) == M and int(g["fwd.step_counter"][0, 1]) == HW * HW and int(g["fwd.local_step"][0]) == 1
    cond_feat = orc.cal_cond_feat(fi["cond"], sd, hp, eye_area_percent=fi["eye_area_percent"])
    sig, rgb, amb = orc.head_forward(xyzs[:M], dirs[:M], cond_feat, sd["individual_embeddings"][0], sd, hp)
    ws, amb_sum, depth, image = orc.composite_rays_train_forward(sig, rgb, np.abs(amb).sum(-1), deltas[:M], rays)
    img, dep = orc._finish(image, ws, depth, nears, fars, np.full((HW * HW, 3), 0.5, f32), (1, HW * HW))
    np.testing.assert_allclose(ws, g["fwd.weights_sum"], atol=2e-5)
    np.testing.assert_allclose(amb_sum, g["fwd.ambient"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(img, g["fwd.rgb_map"], atol=2e-5)
    ok = np.isfinite(g["fwd.depth_map"])
    np.testing.assert_allclose(dep[ok], g["fwd.depth_map"][ok], atol=1e-4)
    # mark_untrained_grid: a cell is trained iff some camera sees it (renderer.py:171-190)
    G = hp["grid_size"]
    fx, fy, cx, cy = syn.intrinsics_for(HW, HW)
    untrained = np.unpackbits(g["mark.untrained"])[:G ** 3].astype(bool)
    rng = np.random.default_rng(1)
    pick = rng.integers(0, G, (3000, 3))
    code = np.array([orc.morton3D(*c) for c in pick], np.int64)
    half = 1.0 / G
    world = (2 * pick.astype(f32) / (G - 1) - 1) * f32(1 - half)
    seen = np.zeros(len(pick), bool)
    for P in g["mark.poses"]:
        cam = (world - P[:3, 3]) @ P[:3, :3]
        seen |= (cam[:, 2] > 0) & (np.abs(cam[:, 0]) < cx / fx * cam[:, 2] + half * 2) & (np.abs(cam[:, 1]) < cy / fy * cam[:, 2] + half * 2)
    np.testing.assert_array_equal(untrained[code], ~seen)


def test_product_training_wiring_equals_reference_python():
    """The product's own training-mode Python (NeRFRenderer.render, RADNeRF.forward, the autograd wrappers of raymarching.py / encoders.py) run
    on CPU tensors with its C-ABI calls redirected to the oracle reproduces the reference's golden step -- forward results and every recorded
    gradient -- to the last bit: the two Python layers are wired identically around the kernels.  (Child interpreter: the helper patches
    Tensor.is_cuda, the one place where the product's refusal of CPU tensors is overridden.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "product_on_oracle.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "worst 0.0" in out.stdout, out.stdout[-2000:]
