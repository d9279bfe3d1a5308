"""HIP kernels (through the C ABI) against the CPU oracle, kernel by kernel.  Needs an MI355X.

Bars: bit-exact for integer / index work (Morton codes, bitfield, marcher sample positions and counts, alive flags)
and for the fp32 grid lookup (same fmaf sequence on both sides); stated tolerances where the reference itself uses
fast-math intrinsics (__expf in compositing, __sinf in the frequency encoding).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_library_loaded_is_in_tree():
    from genefaceplusplus_amd import _lib
    _lib.lib()
    assert _lib.LIB_PATH.endswith("genefaceplusplus_amd/libgfpp_radnerf.so")


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.uniform(-1, 1, (n, 3)) * np.array([0.3, 0.3, 0.3]) + np.array([0, 4.0, 0])).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32) * np.array([0.2, 0.2, 0.2], np.float32) + np.array([0, -1, 0], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # edge cases: axis-parallel rays (zero components => inf reciprocals), rays that miss, origin inside the box
    d[0] = [0, -1, 0]
    d[1] = [1, 0, 0]
    d[2] = [0, 0, 1]
    o[3] = [0, 0, 0]
    d[4] = [0, 1, 0]
    o[5] = [-1.0, 4.0, 0.0]; d[5] = [0, -1, 0]     # exactly on a slab plane: (aabb - o) * inf = nan path
    return o, d.astype(np.float32)


def test_near_far(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    o, d = _rays(5000, 1)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    n_ref, f_ref = oracle_mod.near_far_from_aabb(o, d, aabb, 0.05)
    n, f = rm.near_far_from_aabb(t(o, dev), t(d, dev), t(aabb, dev), 0.05)
    np.testing.assert_array_equal(n.cpu().numpy(), n_ref)
    np.testing.assert_array_equal(f.cpu().numpy(), f_ref)
    assert (n_ref == np.finfo(np.float32).max).any() and (n_ref < 10).any()


def test_morton_and_packbits(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    rng = np.random.default_rng(2)
    coords = rng.integers(0, 128, (4096, 3)).astype(np.int32)
    idx = rm.morton3D(t(coords, dev))
    ref = np.array([oracle_mod.morton3D(*c) for c in coords], np.int32)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    back = rm.morton3D_invert(idx)
    np.testing.assert_array_equal(back.cpu().numpy(), coords)
    grid = rng.uniform(0, 20, (1, 128 ** 3 // 16)).astype(np.float32)
    bits = rm.packbits(t(grid, dev), 10.0)
    np.testing.assert_array_equal(bits.cpu().numpy(), oracle_mod.packbits(grid, 10.0))


@pytest.mark.parametrize("n_step", [1, 2, 3, 8])
def test_march_rays_bit_exact(dev, oracle_mod, n_step):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    hp = may_hparams("may_head")
    sd = syn.synthetic_state_dict(hp, "may_head")
    HW = 96
    pose = syn.synthetic_pose(1)[None]
    rays = oracle_mod.get_rays(pose, syn.intrinsics_for(HW, HW), HW, HW)
    o, d = rays["rays_o"][0], rays["rays_d"][0]
    nears, fars = oracle_mod.near_far_from_aabb(o, d, sd["aabb_infer"], hp["min_near"])
    N = o.shape[0]
    rng = np.random.default_rng(3)
    alive = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int32)
    rays_t = nears + rng.uniform(0, 0.6, N).astype(np.float32)     # continue from arbitrary positions
    ref = oracle_mod.march_rays(len(alive), n_step, alive, rays_t, o, d, 1, sd["density_bitfield"], 1, 128, nears, fars, 128, False,
                                hp["dt_gamma"], 16)
    got = rm.march_rays(len(alive), n_step, t(alive, dev), t(rays_t, dev), t(o, dev), t(d, dev), 1, t(sd["density_bitfield"], dev), 1, 128,
                        t(nears, dev), t(fars, dev), 128, False, hp["dt_gamma"], 16)
    for g, r, name in zip(got, ref, ("xyzs", "dirs", "deltas")):
        assert g.shape == r.shape
        np.testing.assert_array_equal(g.cpu().numpy(), r, err_msg=name)
    assert (ref[2][:, 0] > 0).sum() > 1000


def test_march_rays_all_ones_and_all_zero_bitfield(dev, oracle_mod):
    """SURVEY 8c invariant 7."""
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    o = np.tile(np.array([[0.1, 4.0, -0.2]], np.float32), (256, 1))
    d = np.tile(np.array([[0.0, -1.0, 0.0]], np.float32), (256, 1))
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = oracle_mod.near_far_from_aabb(o, d, aabb, 0.05)
    alive = np.arange(256, dtype=np.int32)
    ones = np.full(128 ** 3 // 8, 255, np.uint8)
    x, _, dl = rm.march_rays(256, 8, t(alive, dev), t(nears, dev), t(o, dev), t(d, dev), 1, t(ones, dev), 1, 128, t(nears, dev), t(fars, dev),
                             -1, False, 1 / 256, 16)
    dt = np.float32(2 * np.float32(1.7320508075688772) / np.float32(128))
    dl = dl.cpu().numpy().reshape(256, 8, 2)
    assert np.all(dl[:, :, 0] == dt)
    tk = nears[0]
    for k in range(8):
        tk = np.float32(tk + dt)
        assert dl[0, k, 1] == tk
    zeros = np.zeros(128 ** 3 // 8, np.uint8)
    x, _, dl = rm.march_rays(256, 8, t(alive, dev), t(nears, dev), t(o, dev), t(d, dev), 1, t(zeros, dev), 1, 128, t(nears, dev), t(fars, dev),
                             -1, False, 1 / 256, 16)
    assert float(dl.abs().sum()) == 0.0 and float(x.abs().sum()) == 0.0


@pytest.mark.parametrize("n_step", [1, 4])
def test_composite_rays(dev, oracle_mod, n_step):
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    rng = np.random.default_rng(5)
    N, n_alive = 4000, 1500
    alive = np.sort(rng.choice(N, n_alive, replace=False)).astype(np.int32)
    M = n_alive * n_step
    sig = np.exp(rng.uniform(-2, 6, M)).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.stack([np.full(M, 0.027063, np.float32), rng.uniform(3.5, 4.5, M).astype(np.float32)], 1)
    deltas[rng.random(M) < 0.1] = 0          # exhausted rays
    ws = rng.uniform(0, 0.995, N).astype(np.float32)
    dep = rng.uniform(0, 4, N).astype(np.float32)
    img = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    rt = rng.uniform(3.5, 4.5, N).astype(np.float32)
    ref = [alive.copy(), rt.copy(), ws.copy(), dep.copy(), img.copy()]
    oracle_mod.composite_rays(n_alive, n_step, ref[0], ref[1], sig, rgb, deltas, ref[2], ref[3], ref[4], 0.01)
    g = [t(alive, dev), t(rt, dev), t(ws, dev), t(dep, dev), t(img, dev)]
    rm.composite_rays(n_alive, n_step, g[0], g[1], t(sig, dev), t(rgb, dev), t(deltas, dev), g[2], g[3], g[4], 0.01)
    got = [x.cpu().numpy() for x in g]
    # alive flags: exact except rays whose transmittance sits within rounding of the threshold
    mism = got[0] != ref[0]
    assert mism.mean() < 2e-3
    ok = ~np.isin(np.arange(N), alive[mism])
    np.testing.assert_allclose(got[2][ok], ref[2][ok], atol=2e-6)       # v_exp_f32 vs libm expf
    np.testing.assert_allclose(got[3][ok], ref[3][ok], atol=2e-5)
    np.testing.assert_allclose(got[4][ok], ref[4][ok], atol=2e-6)
    np.testing.assert_array_equal(got[1][ok], ref[1][ok])


def _grid_case(oracle_mod, D, gridtype, rng, C=2, log2_hashmap=16, desired=2048):
    off, pls = oracle_mod.grid_offsets(D, 16, C, 2, 16, log2_hashmap, desired)
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float32)
    return off, pls, emb


@pytest.mark.parametrize("D,gridtype,interp,C", [(3, "tiled", "linear", 2), (2, "tiled", "linear", 2), (3, "hash", "linear", 2),
                                                 (3, "tiled", "smoothstep", 2), (2, "hash", "smoothstep", 4), (3, "hash", "linear", 1),
                                                 (3, "tiled", "linear", 8)])
def test_grid_encode_fp32_bit_exact(dev, oracle_mod, D, gridtype, interp, C):
    from genefaceplusplus_amd.radnerfs.encoders import grid_encode_raw
    rng = np.random.default_rng(11)
    off, pls, emb = _grid_case(oracle_mod, D, gridtype, rng, C)
    B = 20011                                                   # ragged: not a multiple of the block size
    u = rng.uniform(0, 1, (B, D)).astype(np.float32)
    u[0] = 0.0; u[1] = 1.0                                       # box corners
    u[2] = -0.001; u[3, 0] = 1.0001                              # out of range -> zeros
    u[4] = 0.5
    S = np.log2(pls)
    sc, res = oracle_mod.grid_level_params(7, S, 16)
    u[5] = np.float32((np.float32(3.0) - np.float32(0.5)) / sc)  # lands (nearly) on a lattice vertex of level 7
    ref = oracle_mod.grid_encode_raw(u, emb, off, S, 16, oracle_mod.GRIDTYPE[gridtype], False, oracle_mod.INTERP[interp])
    got = grid_encode_raw(t(u, dev), t(emb, dev), t(off, dev), pls, 16, oracle_mod.GRIDTYPE[gridtype], False, oracle_mod.INTERP[interp])
    got = got.cpu().numpy()
    assert got.shape == ref.shape == (16, B, C)
    assert np.all(got[:, 2] == 0) and np.all(got[:, 3] == 0)
    np.testing.assert_array_equal(got, ref)


def test_grid_encode_empty_and_half(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs.encoders import grid_encode_raw
    rng = np.random.default_rng(12)
    off, pls, emb = _grid_case(oracle_mod, 3, "tiled", rng)
    out = grid_encode_raw(torch.empty(0, 3, device=dev), t(emb, dev), t(off, dev), pls, 16, 1, False, 0)
    assert out.shape == (16, 0, 2)
    u = rng.uniform(0, 1, (5000, 3)).astype(np.float32)
    ref = oracle_mod.grid_encode_raw(u, emb.astype(np.float16).astype(np.float32), off, np.log2(pls), 16, 1, False, 0)
    got = grid_encode_raw(t(u, dev), t(emb, dev).half(), t(off, dev), pls, 16, 1, False, 0)
    assert got.dtype == torch.float16
    # fp16 tables, fp32 accumulate, one rounding on store (the reference accumulates in half: looser still)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref, atol=1e-3)


def test_grid_unsupported_raises(dev):
    from genefaceplusplus_amd.radnerfs.encoders import grid_encode_raw
    from genefaceplusplus_amd._lib import GfppError
    emb = torch.zeros(64, 3, device=dev)
    off = torch.tensor([0, 64], dtype=torch.int32, device=dev)
    with pytest.raises(GfppError):
        grid_encode_raw(torch.rand(8, 3, device=dev), emb, off, 2.0, 16, 0, False, 0)     # C = 3
    with pytest.raises(GfppError):
        grid_encode_raw(torch.rand(8, 5, device=dev), torch.zeros(64, 2, device=dev), off, 2.0, 16, 0, False, 0)   # D = 5


def test_sh_and_freq(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs.encoders import SHEncoder, FreqEncoder
    rng = np.random.default_rng(13)
    d = rng.standard_normal((3001, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for deg in (1, 2, 3, 4):
        got = SHEncoder(degree=deg)(t(d, dev)).cpu().numpy()
        np.testing.assert_allclose(got, oracle_mod.sh_encode(d, deg), atol=1e-6)
    assert np.all(got[:, 0] == np.float32(0.28209479177387814))
    for D, deg, scale in ((2, 10, 0.8), (6, 4, 4.0), (14, 4, 1.0)):
        x = (rng.uniform(-1, 1, (777, D)) * scale).astype(np.float32)
        got = FreqEncoder(input_dim=D, degree=deg)(t(x, dev)).cpu().numpy()
        ref = oracle_mod.freq_encode(x, deg)
        assert got.shape == ref.shape == (777, D + 2 * D * deg)
        np.testing.assert_array_equal(got[:, :D], x)
        np.testing.assert_allclose(got, ref, atol=2e-6)


def test_get_rays(dev, oracle_mod):
    from genefaceplusplus_amd.radnerfs.camera import get_rays, get_bg_coords, convert_poses
    pose = syn.synthetic_pose(2)
    H = W = 48
    intr = syn.intrinsics_for(H, W)
    ref = oracle_mod.get_rays(pose[None], intr, H, W)
    got = get_rays(t(pose, dev)[None], intr, H, W)
    np.testing.assert_array_equal(got["rays_o"].cpu().numpy(), ref["rays_o"])
    np.testing.assert_allclose(got["rays_d"].cpu().numpy(), ref["rays_d"], atol=2e-7)
    # torch on the GPU divides by a host scalar as a multiplication by its reciprocal (1 ulp from numpy's true division);
    # the reference computes bg_coords with the very same torch expression on its GPU
    np.testing.assert_allclose(get_bg_coords(H, W, dev).cpu().numpy(), oracle_mod.get_bg_coords(H, W), atol=1.5e-7)
    np.testing.assert_allclose(convert_poses(t(pose, dev)[None]).cpu().numpy(), oracle_mod.convert_poses(pose[None]), atol=1e-6)
    # sampled variants (training, utils.py:310-343): the selected rays are bit-identical to the same pixels of the full frame, the index draw
    # follows the reference's torch.randint call sequence, patches are patch_size^2 contiguous pixels, rect selects a row/col window
    full_o, full_d = got["rays_o"][0], got["rays_d"][0]
    torch.manual_seed(5)
    r = get_rays(t(pose, dev)[None], intr, H, W, N=300)
    torch.manual_seed(5)
    want = torch.randint(0, H * W, size=[300], device=dev)
    assert torch.equal(r["inds"][0], want) and r["rays_d"].shape == (1, 300, 3)
    assert torch.equal(r["rays_d"][0], full_d[want]) and torch.equal(r["rays_o"][0], full_o[want])
    assert torch.equal(r["i"][0], (want % W).float() + 0.5) and torch.equal(r["j"][0], (want // W).float() + 0.5)
    r = get_rays(t(pose, dev)[None], intr, H, W, N=4 * 64, patch_size=8)
    idx = r["inds"][0].view(4, 8, 8)
    assert torch.equal(idx, idx[:, :1, :1] + (torch.arange(8, device=dev).view(1, 8, 1) * W + torch.arange(8, device=dev).view(1, 1, 8)))
    assert int(idx.max()) < H * W and torch.equal(r["rays_d"][0], full_d[r["inds"][0]])
    r = get_rays(t(pose, dev)[None], intr, H, W, rect=(5, 9, 10, 30))
    rows, cols = torch.meshgrid(torch.arange(5, 9, device=dev), torch.arange(10, 30, device=dev), indexing="ij")
    assert torch.equal(r["inds"][0], (rows * W + cols).reshape(-1)) and torch.equal(r["rays_d"][0], full_d[r["inds"][0]])
    assert get_rays(t(pose, dev)[None], intr, H, W, N=10 ** 9)["rays_d"].shape == (1, H * W, 3)       # N is clipped to H*W


@pytest.mark.parametrize("variant", ["may_head", "may_torso_sr"])
def test_cond_feat_kernel_matches_oracle(dev, oracle_mod, variant):
    """gfpp_cond_feat (one launch) vs the oracle's cal_cond_feat (radnerf.py:88-106) and vs the PyTorch modules it replaces."""
    from helpers import frame_case, build_model
    case = frame_case(variant, 64)
    model = build_model(case, dev, "fused")
    cond = torch.from_numpy(case["cond"]).to(dev)
    eye = torch.from_numpy(case["eye_area_percent"]).to(dev)
    ref = oracle_mod.cal_cond_feat(case["cond"], case["sd"], case["hp"], case["eye_area_percent"])
    with torch.no_grad():
        got = model.cal_cond_feat(cond, eye_area_percent=eye)
        model.executor = "staged"
        torch_path = model.cal_cond_feat(cond, eye_area_percent=eye)
    assert got.shape == torch_path.shape
    np.testing.assert_allclose(got.cpu().numpy().reshape(-1), np.asarray(ref).reshape(-1), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(got.cpu().numpy(), torch_path.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("noise_mode", ["const", "none"])
def test_superresolution_matches_oracle(dev, noise_mode):
    """gfpp_sr_forward (folded weights, f16 MFMA convolutions) vs the fp32 oracle of the reference's Superresolution.  The reference runs
    these blocks in fp16 on the GPU; stated tolerance vs fp32: PSNR >= 50 dB over the output range and max-abs <= 4e-2."""
    from oracle import sr_oracle
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    net = Superresolution(channels=3)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    rng = np.random.default_rng(5)
    yy, xx = np.meshgrid(np.linspace(0, 1, 256, dtype=np.float32), np.linspace(0, 1, 256, dtype=np.float32), indexing="ij")
    x = np.stack([0.5 + 0.5 * np.sin(9 * xx + 3 * yy), yy * xx, rng.random((256, 256), dtype=np.float32)], 0)[None].astype(np.float32)
    ref = sr_oracle.superresolution(x, sd, prefix="", noise_mode=noise_mode)
    with torch.no_grad():
        got = net(torch.from_numpy(x).to(dev), noise_mode=noise_mode).cpu().numpy()
    assert got.shape == ref.shape == (1, 3, 512, 512)
    err = np.abs(got - ref)
    span = float(ref.max() - ref.min())
    psnr = 10 * np.log10(span ** 2 / float(np.mean((got - ref) ** 2)))
    print("sr", noise_mode, "max", float(err.max()), "mean", float(err.mean()), "psnr", psnr, "range", float(ref.min()), float(ref.max()))
    assert psnr >= 50.0 and err.max() <= 4e-2, (psnr, float(err.max()))


def test_rgb_to_uint8_truncates_like_the_caller(dev):
    """gfpp_rgb_to_u8 == `(pred_rgb * 255.).int() ... astype(np.uint8)` of inference/genefacepp_infer.py:468 (truncation, values in [0,1])."""
    from genefaceplusplus_amd import frames
    x = torch.tensor([[0.0, 0.5, 1.0], [0.999, 0.0039, 0.00392157]], device=dev)
    assert frames.to_uint8_hwc(x).cpu().tolist() == [[0, 127, 255], [254, 0, 1]]
    rng = np.random.default_rng(4)
    for n in (1, 3, 4, 5, 4099):                                    # tails that are not a multiple of the 4-value vector width
        v = rng.random((n, 3)).astype(np.float32)
        got = frames.to_uint8_hwc(t(v, dev)).cpu().numpy()
        np.testing.assert_array_equal(got, (v * np.float32(255.0)).astype(np.int32).astype(np.uint8))


def test_empty_inputs_are_no_ops(dev):
    """Zero rays / points / samples: every wrapper returns correctly shaped empties and launches nothing (the reference's shims get here when
    a frame has no alive ray or a batch has no masked pixel)."""
    from genefaceplusplus_amd.radnerfs import raymarching as rm
    from genefaceplusplus_amd.radnerfs.encoders import GridEncoder, SHEncoder, FreqEncoder
    z3 = torch.zeros(0, 3, device=dev)
    aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32, device=dev)
    nears, fars = rm.near_far_from_aabb(z3, z3, aabb, 0.05)
    assert nears.shape == (0,) and fars.shape == (0,)
    assert rm.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=dev)).shape == (0,)
    assert rm.morton3D_invert(torch.zeros(0, dtype=torch.int32, device=dev)).shape == (0, 3)
    bitfield = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=dev)
    xyzs, dirs, deltas = rm.march_rays(0, 4, torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, device=dev), z3, z3, 1.0, bitfield, 1, 128,
                                       nears, fars, 128, False, 0.0, 16)
    assert xyzs.shape[1] == 3 and float(xyzs.abs().sum()) == 0.0 and float(deltas.abs().sum()) == 0.0
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048, gridtype="tiled").to(dev)
    assert enc(z3, bound=1).shape == (0, 32)
    assert SHEncoder(degree=4)(z3).shape == (0, 16)
    assert FreqEncoder(input_dim=2, degree=10)(torch.zeros(0, 2, device=dev)).shape == (0, 42)
    torch.cuda.synchronize()


def test_superresolution_random_noise_is_drawn_in_the_kernels(dev):
    """noise_mode 'random' (the reference's default: a fresh unit-normal field per layer and frame, networks_stylegan2.py:329-331) without a generator
    launch: Philox4x32-10 inside the SR kernels, keyed by the seed, counted by (pixel, layer, frame).  What must hold: a frame differs from the next
    and from the noise-free one; the perturbation is what `noise_strength * N(0, 1)` through the rest of the net gives with the oracle's own normals
    (same first two moments); the same seed reproduces the same frames; clamp01 is the caller's clamp."""
    from oracle import sr_oracle
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    net = Superresolution(channels=3)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.random((1, 3, 256, 256)).astype(np.float32)).to(dev)
    with torch.no_grad():
        net.reseed(1234)
        a0 = net(x, noise_mode="random").cpu().numpy()
        a1 = net(x, noise_mode="random").cpu().numpy()
        none = net(x, noise_mode="none").cpu().numpy()
        net.reseed(1234)
        b0 = net(x, noise_mode="random").cpu().numpy()
        b1 = net(x, noise_mode="random").cpu().numpy()
        net.reseed(99)
        c0 = net(x, noise_mode="random").cpu().numpy()
        clamped = net(x, noise_mode="none", clamp01=True).cpu().numpy()
    np.testing.assert_array_equal(a0, b0)
    np.testing.assert_array_equal(a1, b1)
    assert not np.array_equal(a0, a1) and not np.array_equal(a0, c0)
    # launches frozen in a captured graph take a new seed too (round-3 advisory): the seed WORD lives in device memory, the launch argument the graph
    # baked in is XOR-ed with it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        net(x, noise_mode="random")
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        out = net(x, noise_mode="random")
    net.reseed(1234)
    graph.replay()
    g0 = out.cpu().numpy().copy()
    graph.replay()
    g1 = out.cpu().numpy().copy()
    net.reseed(99)
    graph.replay()
    np.testing.assert_array_equal(out.cpu().numpy(), c0)
    np.testing.assert_array_equal(g0, a0)
    np.testing.assert_array_equal(g1, a1)
    np.testing.assert_array_equal(clamped, np.clip(none, 0.0, 1.0))
    # moments of the perturbation against the oracle run with numpy normals of the same law (4 independent fields)
    ref_none = sr_oracle.superresolution(x.cpu().numpy(), sd, prefix="", noise_mode="none")
    nrng = np.random.default_rng(3)
    fields = {name: nrng.standard_normal((res, res)).astype(np.float32) for name, res in (("block0.conv0", 256), ("block0.conv1", 256), ("block1.conv0", 512), ("block1.conv1", 512))}
    ref_rand = sr_oracle.superresolution(x.cpu().numpy(), sd, prefix="", noise_mode="random", noise_random=fields)
    d_ref, d_got = (ref_rand - ref_none).reshape(-1), (a0 - none).reshape(-1)
    print("perturbation std: oracle", float(d_ref.std()), "kernels", float(d_got.std()), "means", float(d_ref.mean()), float(d_got.mean()))
    assert abs(d_got.std() / d_ref.std() - 1.0) <= 0.05
    assert abs(d_got.mean() - d_ref.mean()) <= 0.05 * d_ref.std()
    # the two frames' perturbations: independent noise (what they share is the deterministic part of the response to noise of this strength --
    # E[f(x + n)] - f(x) -- which the oracle's two draws share just as well)
    d1 = (a1 - none).reshape(-1)
    nrng2 = np.random.default_rng(4)
    fields2 = {name: nrng2.standard_normal(v.shape).astype(np.float32) for name, v in fields.items()}
    d_ref2 = (sr_oracle.superresolution(x.cpu().numpy(), sd, prefix="", noise_mode="random", noise_random=fields2) - ref_none).reshape(-1)
    c_got, c_ref = float(np.corrcoef(d_got, d1)[0, 1]), float(np.corrcoef(d_ref, d_ref2)[0, 1])
    print("correlation of two frames' perturbations: kernels", c_got, "oracle", c_ref)
    assert abs(c_got - c_ref) <= 0.03


@pytest.mark.parametrize("noise_mode", ["const", "random", "none"])
def test_superresolution_first_layer_inside_the_second_equals_its_own_launch(dev, monkeypatch, noise_mode):
    """Block 0's first convolution computed into the halo patch of the second one (k_sr_conv3<..., FIRST>) against its own launch (k_sr_first,
    gfpp_tuning.sr_fuse_first = 0): same fragments, same MFMA order, same epilogue -- the 512^2 image bit for bit, image borders included."""
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.random((1, 3, 256, 256)).astype(np.float32)).to(dev)
    outs = []
    from genefaceplusplus_amd import tuning
    for fuse in (0, 1):
        net = Superresolution(channels=3)
        net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
        net = net.to(dev).eval()
        with torch.no_grad(), tuning.tuned(sr_fuse_first=fuse):
            net.reseed(77)
            outs.append([net(x, noise_mode=noise_mode).cpu().numpy() for _ in range(2)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert (noise_mode == "random") == (not np.array_equal(outs[0][0], outs[0][1]))


@pytest.mark.parametrize("noise_mode", ["const", "random", "none"])
def test_superresolution_last_layer_with_resident_weights_equals_the_per_patch_launch(dev, monkeypatch, noise_mode):
    """Block 1's last convolution as one workgroup per CU walking over its patches with the layer's 72 KB of weights resident in LDS (k_sr_final_resident)
    against one workgroup per patch with streamed weight chunks (k_sr_conv3<64, 2, final>, gfpp_tuning.sr_final_resident = 0): same fragments, same tap and step order,
    same epilogue -- the 512^2 image bit for bit over two frames (the frame counter advances the same way), image borders included."""
    from genefaceplusplus_amd import synthetic as syn
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    rng = np.random.default_rng(13)
    x = torch.from_numpy(rng.random((1, 3, 256, 256)).astype(np.float32)).to(dev)
    outs = []
    from genefaceplusplus_amd import tuning
    for resident in (0, 1):
        net = Superresolution(channels=3)
        net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
        net = net.to(dev).eval()
        with torch.no_grad(), tuning.tuned(sr_final_resident=resident):
            net.reseed(78)
            outs.append([net(x, noise_mode=noise_mode).cpu().numpy() for _ in range(3)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert (noise_mode == "random") == (not np.array_equal(outs[0][0], outs[0][1]))


def test_occupancy_bounds_enclose_exactly_the_set_cells(dev):
    """gfpp_occupancy_bounds against a numpy walk over the set bits (Morton order, raymarching.cu:56-88; two cascades), plus the empty bitfield."""
    import ctypes
    from genefaceplusplus_amd import _lib
    from genefaceplusplus_amd.radnerfs import frame_pipeline  # noqa: F401  (registers the entry)
    H, C, bound = 64, 2, 2.0
    rng = np.random.default_rng(3)
    cells = np.zeros((C, H, H, H), bool)                      # [level][x][y][z]
    cells[0, 20:41, 25:33, 0:7] = rng.random((21, 8, 7)) < 0.3
    cells[0, 20, 25, 3] = cells[0, 40, 32, 6] = True
    cells[1, 30:35, 31:34, 60:64] = True
    def spread(v):
        v = v.astype(np.uint64)
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    lv, x, y, z = np.nonzero(cells)
    idx = lv.astype(np.uint64) * np.uint64(H ** 3) + (spread(x) | (spread(y) << np.uint64(1)) | (spread(z) << np.uint64(2)))
    bits = np.zeros(C * H ** 3, np.uint8)
    bits[idx.astype(np.int64)] = 1
    packed = np.packbits(bits, bitorder="little")
    lo, hi, margin = np.full(3, np.inf), np.full(3, -np.inf), 0.0
    for l in range(C):
        sel = lv == l
        if not sel.any():
            continue
        mb = min(2.0 ** l, bound)
        cw = 2.0 * mb / H
        for k, n in enumerate((x[sel], y[sel], z[sel])):
            lo[k] = min(lo[k], -bound if n.min() == 0 else n.min() * cw - mb)
            hi[k] = max(hi[k], bound if n.max() == H - 1 else (n.max() + 1) * cw - mb)
        margin = max(margin, cw)
    out = torch.empty(6, dtype=torch.float32, device=dev)
    field = torch.from_numpy(packed).to(dev)
    _lib.call("gfpp_occupancy_bounds", field.data_ptr(), C, H, bound, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    np.testing.assert_allclose(out.cpu().numpy(), np.concatenate([lo - margin, hi + margin]), rtol=0, atol=1e-6)
    field.zero_()
    _lib.call("gfpp_occupancy_bounds", field.data_ptr(), C, H, bound, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    o = out.cpu().numpy()
    assert np.all(o[:3] > o[3:])


@pytest.mark.parametrize("noise_mode", ["const", "none", "random"])
def test_superresolution_polyphase_upsampling_layer_matches_the_composed_convolution(dev, noise_mode):
    """Round 6: block 1's up-sampling layer as transposed convolution in polyphase form + the FIR as a second GEMM (k_sr_up_poly: 9 tap products per pixel and an
    f16 intermediate like the reference's own) against the composed 3 x 3 convolution with 4 x 64 output channels (k_sr_conv3<128, up>: 36 tap products, no
    intermediate; gfpp_tuning.sr_up_poly = 0).  Not the same bits -- the intermediate is rounded to f16 and the sums associate differently --: the 512^2 images
    agree to >= 60 dB and 1e-2, and BOTH are inside the oracle's bars (PSNR >= 50 dB, 4e-2); with in-kernel noise the same seed draws the same field in both."""
    from oracle import sr_oracle
    from genefaceplusplus_amd import tuning
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    rng = np.random.default_rng(17)
    yy, xx = np.meshgrid(np.linspace(0, 1, 256, dtype=np.float32), np.linspace(0, 1, 256, dtype=np.float32), indexing="ij")
    x = np.stack([0.5 + 0.5 * np.sin(11 * xx + 2 * yy), yy * (1 - xx), rng.random((256, 256), dtype=np.float32)], 0)[None].astype(np.float32)
    outs = {}
    for poly in (0, 1):
        net = Superresolution(channels=3)
        net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
        net = net.to(dev).eval()
        with torch.no_grad(), tuning.tuned(sr_up_poly=poly):
            net.reseed(91)
            outs[poly] = net(torch.from_numpy(x).to(dev), noise_mode=noise_mode).cpu().numpy()
    span = float(outs[0].max() - outs[0].min())
    diff = np.abs(outs[0] - outs[1])
    psnr = 10 * np.log10(span ** 2 / max(float(np.mean((outs[0] - outs[1]) ** 2)), 1e-20))
    print("poly vs composed", noise_mode, "max", float(diff.max()), "psnr", psnr)
    assert psnr >= 60.0 and diff.max() <= 1e-2 * span, (psnr, float(diff.max()), span)
    if noise_mode != "random":
        ref = sr_oracle.superresolution(x, sd, prefix="", noise_mode=noise_mode)
        for poly in (0, 1):
            err = np.abs(outs[poly] - ref)
            rs = float(ref.max() - ref.min())
            p = 10 * np.log10(rs ** 2 / float(np.mean((outs[poly] - ref) ** 2)))
            print("  vs oracle, poly =", poly, "max", float(err.max()), "psnr", p)
            assert p >= 50.0 and err.max() <= 4e-2
