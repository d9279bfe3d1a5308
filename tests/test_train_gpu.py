"""Training-side HIP kernels (csrc/train.hip) vs the CPU oracle, through the same Python wrappers the reference's training
loop would use (genefaceplusplus_amd.radnerfs.raymarching / encoders).  Sample positions, counts and per-ray sums are compared per ray
(the reference hands out point ranges with atomics, so slot order is unspecified); table gradients are atomic sums (tolerance)."""
import numpy as np
import pytest
import torch

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene():
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    rng = np.random.default_rng(3)
    N = 3000
    o = np.tile(np.array([[0.0, 4.0, 0.0]], f32), (N, 1))
    d = np.stack([rng.uniform(-0.1, 0.1, N), -np.ones(N), rng.uniform(-0.12, 0.12, N)], 1).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return hp, sd, o, d


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_march_rays_train_and_backward(dev, oracle_mod, scene):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    hp, sd, o, d = scene
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], f32)
    nears, fars = oracle_mod.near_far_from_aabb(o, d, aabb, 0.05)
    ref = oracle_mod.march_rays_train(o, d, 1.0, sd["density_bitfield"], 1, 128, nears, fars, dt_gamma=hp["dt_gamma"], max_steps=32)
    ro, rd = _t(o, dev).requires_grad_(True), _t(d, dev).requires_grad_(True)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, _t(sd["density_bitfield"], dev), 1, 128, _t(nears, dev), _t(fars, dev), counter, -1,
                                                   False, 128, True, hp["dt_gamma"], 32)
    c = counter.cpu().numpy()
    assert c[0] == ref[4][0] and c[1] == ref[4][1] == o.shape[0]
    assert xyzs.shape[0] % 128 == 0 and xyzs.shape[0] >= c[0]
    r = rays.cpu().numpy()
    by_ray = {int(a): (int(b), int(k)) for a, b, k in r}
    assert len(by_ray) == o.shape[0]
    X, D, T = xyzs.detach().cpu().numpy(), dirs.detach().cpu().numpy(), deltas.detach().cpu().numpy()
    for n in range(0, o.shape[0], 7):
        _, off_r, cnt_r = ref[3][n]
        off, cnt = by_ray[n]
        assert cnt == cnt_r
        np.testing.assert_array_equal(X[off:off + cnt], ref[0][off_r:off_r + cnt])       # bit-exact positions
        np.testing.assert_array_equal(T[off:off + cnt], ref[2][off_r:off_r + cnt])
        np.testing.assert_array_equal(D[off:off + cnt], ref[1][off_r:off_r + cnt])
    # backward through the autograd function
    gx, gd = torch.randn_like(xyzs), torch.randn_like(dirs)
    torch.autograd.backward([xyzs, dirs], [gx, gd])
    go_ref, gd_ref = oracle_mod.march_rays_train_backward(gx.cpu().numpy(), gd.cpu().numpy(), r, T)
    # the oracle indexes gradients by rays-row like the reference (row n of `rays` <-> output row n)
    np.testing.assert_allclose(ro.grad.cpu().numpy(), go_ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rd.grad.cpu().numpy(), gd_ref, rtol=1e-4, atol=1e-4)


def test_composite_rays_train_forward_backward(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    rng = np.random.default_rng(5)
    N = 500
    counts = rng.integers(0, 24, N)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    perm = rng.permutation(N)
    rays = np.stack([perm, offs, counts], 1).astype(np.int32)                # scrambled ray ids: outputs are indexed by rays[:,0]
    M = int(counts.sum()) + 17
    sig = rng.uniform(0.0, 60.0, M).astype(f32)
    rgb = rng.random((M, 3)).astype(f32)
    amb = rng.random(M).astype(f32)
    deltas = np.stack([np.full(M, 0.027, f32), rng.uniform(3.5, 4.5, M).astype(f32)], 1)
    ws_r, as_r, dp_r, im_r = oracle_mod.composite_rays_train_forward(sig, rgb, amb, deltas, rays, 1e-4)
    ts, tr, ta = (_t(v, dev).requires_grad_(True) for v in (sig, rgb, amb))
    ws, asum, depth, img = rm.composite_rays_train(ts, tr, ta, _t(deltas, dev), _t(rays, dev), 1e-4)
    for got, ref in ((ws, ws_r), (asum, as_r), (depth, dp_r), (img, im_r)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-6)          # __expf vs expf
    gws, gas, gim = torch.randn_like(ws), torch.randn_like(asum), torch.randn_like(img)
    torch.autograd.backward([ws, asum, img], [gws, gas, gim])
    gs_r, gr_r, ga_r = oracle_mod.composite_rays_train_backward(gws.cpu().numpy(), gas.cpu().numpy(), gim.cpu().numpy(), sig, rgb, amb, deltas, rays,
                                                                ws_r, as_r, im_r, 1e-4)
    np.testing.assert_allclose(ts.grad.cpu().numpy(), gs_r, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(tr.grad.cpu().numpy(), gr_r, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ga_r, rtol=0, atol=0)


@pytest.mark.parametrize("D,gridtype,interp", [(3, "tiled", "linear"), (2, "tiled", "linear"), (3, "hash", "smoothstep")])
def test_grid_encoder_gradients(dev, oracle_mod, D, gridtype, interp):
    from genefaceplusplus_amd.radnerfs.encoders import GridEncoder
    rng = np.random.default_rng(6)
    enc = GridEncoder(input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=14, desired_resolution=512, gridtype=gridtype,
                      interpolation=interp).to(dev)
    E = rng.standard_normal(tuple(enc.embeddings.shape)).astype(f32)
    with torch.no_grad():
        enc.embeddings.copy_(_t(E, dev))
    B = 4000
    x = rng.uniform(-1, 1, (B, D)).astype(f32)
    x[:3] = 1.7                                                         # out of range rows
    off = enc.offsets.cpu().numpy()
    xt = _t(x, dev).requires_grad_(True)
    y = enc(xt, bound=1)
    u = ((x + f32(1)) / f32(2)).astype(f32)
    fwd = oracle_mod.grid_encode_levels(u, E, off, enc.per_level_scale, 16, gridtype, False, interp)
    np.testing.assert_array_equal(y.detach().cpu().numpy(), np.ascontiguousarray(fwd.transpose(1, 0, 2)).reshape(B, 32))
    g = rng.standard_normal((B, 32)).astype(f32)
    y.backward(_t(g, dev))
    grad_lbc = np.ascontiguousarray(g.reshape(B, 16, 2).transpose(1, 0, 2))
    dy = oracle_mod.grid_encode_dydx(u, E, off, enc.per_level_scale, 16, gridtype, False, interp)
    ge, gi = oracle_mod.grid_encode_backward(grad_lbc, u, E, off, enc.per_level_scale, 16, gridtype, False, interp, dy_dx=dy)
    got_e = enc.embeddings.grad.cpu().numpy()
    scale = np.abs(ge).max()
    assert np.abs(got_e - ge).max() <= 2e-5 * scale + 1e-5               # atomic summation order
    # d/dx: the module maps x -> (x + 1)/2, so dL/dx = 0.5 * dL/du
    np.testing.assert_allclose(xt.grad.cpu().numpy(), 0.5 * gi, rtol=2e-4, atol=2e-3)
    # total-variation gradient on top of the existing grad
    before = enc.embeddings.grad.clone()
    pts = rng.uniform(-1, 1, (3000, D)).astype(f32)
    enc.grad_total_variation(weight=1e-3, inputs=_t(pts, dev), bound=1)
    tv_ref = oracle_mod.grad_total_variation(((pts + f32(1)) / f32(2)).astype(f32), E, np.zeros_like(E), off, 1e-3, enc.per_level_scale, 16, gridtype, False)
    delta = (enc.embeddings.grad - before).cpu().numpy()
    # delta is a difference of fp32 gradients of magnitude |before|: allow their rounding on top of the atomic-order tolerance
    assert np.abs(delta - tv_ref).max() <= 2e-5 * np.abs(tv_ref).max() + 5e-7 * float(before.abs().max())


def test_dilation_and_sph(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    rng = np.random.default_rng(7)
    grid = rng.random((2, 32 ** 3)).astype(f32)
    np.testing.assert_array_equal(rm.morton3D_dilation(_t(grid, dev)).cpu().numpy(), oracle_mod.morton3D_dilation(grid, 2, 32))
    o = rng.uniform(-0.3, 0.3, (1000, 3)).astype(f32)
    d = rng.standard_normal((1000, 3)).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    np.testing.assert_allclose(rm.sph_from_ray(_t(o, dev), _t(d, dev), 1.0).cpu().numpy(), oracle_mod.sph_from_ray(o, d, 1.0), rtol=1e-5, atol=2e-6)


def test_compat_ext_exposes_training_api(dev):
    """The reference's own shims look the training functions up by name on the extension modules (raymarching.py:9-12, grid.py:9-12)."""
    import sys
    import genefaceplusplus_amd.compat_ext as ext
    ext.install()
    rmod, gmod = sys.modules["_raymarching_face"], sys.modules["_gridencoder"]
    for n in ("sph_from_ray", "morton3D_dilation", "march_rays_train", "march_rays_train_backward", "composite_rays_train_forward",
              "composite_rays_train_backward"):
        assert callable(getattr(rmod, n))
    for n in ("grid_encode_forward", "grid_encode_backward", "grad_total_variation"):
        assert callable(getattr(gmod, n))
    g = torch.rand(1, 16 ** 3, device=dev)
    out = torch.empty_like(g)
    rmod.morton3D_dilation(g, 1, 16, out)
    assert float((out >= g).float().mean()) == 1.0
