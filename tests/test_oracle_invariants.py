"""Self-derived known-answer tests for the CPU oracle.  The reference has no test
suite, so these invariants -- read off the kernel sources -- are what pins the oracle's kernel-level semantics."""
import numpy as np
import pytest

from genefaceplusplus_amd import synthetic as syn
from genefaceplusplus_amd.configs import may_hparams

F32MAX = np.finfo(np.float32).max
DT = np.float32(np.float32(2 * np.float32(1.7320508075688772)) / np.float32(128))   # 2*sqrt(3)/128 = 0.027063...


def test_1_morton_bit_interleave(oracle_mod):
    orc = oracle_mod
    assert orc.morton3D(1, 0, 0) == 1 and orc.morton3D(0, 1, 0) == 2 and orc.morton3D(0, 0, 1) == 4
    assert orc.morton3D(127, 127, 127) == 128 ** 3 - 1
    rng = np.random.default_rng(0)
    for x, y, z in rng.integers(0, 128, (200, 3)):
        m = orc.morton3D(x, y, z)
        assert (orc.morton3D_invert(m), orc.morton3D_invert(m >> 1), orc.morton3D_invert(m >> 2)) == (x, y, z)
        ref = 0
        for b in range(7):
            ref |= ((int(x) >> b) & 1) << (3 * b) | ((int(y) >> b) & 1) << (3 * b + 1) | ((int(z) >> b) & 1) << (3 * b + 2)
        assert m == ref
    # numpy generator used for the synthetic scene agrees
    assert int(syn.morton3d(np.uint32(5), np.uint32(9), np.uint32(77))) == orc.morton3D(5, 9, 77)


def test_2_bitfield_layout(oracle_mod):
    grid = np.zeros(64, np.float32)
    grid[[0, 9, 63]] = 11.0
    grid[10] = 10.0                           # not > thresh
    bits = oracle_mod.packbits(grid.reshape(1, -1), 10.0)
    assert list(bits) == [1, 2, 0, 0, 0, 0, 0, 128]
    np.testing.assert_array_equal(bits, syn.pack_bitfield(grid, 10.0))


def test_3_near_far(oracle_mod):
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    o = np.array([[0, 4, 0], [0, 4, 0], [3, 4, 0], [0, 0.2, 0]], np.float32)
    d = np.array([[0, -1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0]], np.float32)
    n, f = oracle_mod.near_far_from_aabb(o, d, aabb, 0.05)
    assert n[0] == 3.5 and f[0] == 4.5                     # through the box centre
    assert n[2] == F32MAX and f[2] == F32MAX               # misses the x slab
    assert n[3] == np.float32(0.05) and f[3] == np.float32(0.7)     # origin inside: near clamped to min_near
    # ray pointing away: the slab intervals are negative; near is clamped, far stays negative => marcher does nothing
    assert f[1] < 0


def _grid(oracle_mod, D, gridtype="tiled"):
    off, pls = oracle_mod.grid_offsets(D, 16, 2, 2, 16, 16, 2048)
    rng = np.random.default_rng(1)
    emb = rng.uniform(-1, 1, (int(off[-1]), 2)).astype(np.float32)
    return off, pls, emb


def test_4_grid_known_answers(oracle_mod):
    orc = oracle_mod
    off, pls, emb = _grid(orc, 3)
    S = np.log2(pls)
    res = [orc.grid_level_params(l, S, 16)[1] for l in range(16)]
    assert res == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    assert orc.grid_level_params(15, S, 16)[0] == np.float32(2047.0)
    sizes = np.diff(off)
    assert list(sizes[:3]) == [4920, 13824, 32768] and all(s == 65536 for s in sizes[3:])
    off2, _ = orc.grid_offsets(2, 16, 2, 2, 16, 16, 2048)
    assert list(np.diff(off2)[:9]) == [296, 576, 1024, 1936, 3600, 6728, 12776, 24032, 45800]
    # a point exactly on a lattice vertex of level 0 returns that table row: u*15 + 0.5 = g + 0 => u = (g - 0.5)/15
    g = np.array([3, 7, 11])
    u = ((g - 0.5) / 15.0).astype(np.float32)
    pos = np.float32(u) * np.float32(15.0) + np.float32(0.5)
    if np.all(pos == np.floor(pos)):
        out = orc.grid_encode_raw(u[None], emb, off, S, 16, 1, False, 0)
        row = int(g[0] + g[1] * 17 + g[2] * 17 * 17) % 4920
        np.testing.assert_array_equal(out[0, 0], emb[row])
    # out-of-range input -> zeros on every level
    out = orc.grid_encode_raw(np.array([[0.5, 1.0001, 0.5], [-1e-6, 0.2, 0.2]], np.float32), emb, off, S, 16, 1, False, 0)
    assert np.all(out == 0)
    # level-15 tiled 3-D index ignores z (stride exceeds the level size after two dimensions)
    r1 = orc.grid_corner_row([100, 200, 5], 1, False, 65536, 2048)
    r2 = orc.grid_corner_row([100, 200, 1999], 1, False, 65536, 2048)
    assert r1 == r2 == (100 + 200 * 2049) % 65536
    # ... but the hash does not; prime_0 = 1
    h1 = orc.grid_corner_row([100, 200, 5], 0, False, 65536, 2048)
    assert h1 == ((100 * 1) ^ (200 * 2654435761 % 2 ** 32) ^ (5 * 805459861 % 2 ** 32)) % 65536
    # level 0 wraps modulo the ROUNDED size 4920, not 17^3 = 4913
    assert orc.grid_corner_row([16, 16, 16], 1, False, 4920, 16) == (16 + 16 * 17 + 16 * 289) % 4920


def test_5_sh_constants(oracle_mod):
    d = np.array([[0.3, -0.5, 0.8124038]], np.float32)
    sh = oracle_mod.sh_encode(d, 4)[0]
    assert sh[0] == np.float32(0.28209479177387814)
    np.testing.assert_allclose(sh[1:4], np.float32(0.48860251190291987) * np.array([-d[0, 1], d[0, 2], -d[0, 0]]), rtol=1e-6)
    assert oracle_mod.sh_encode(d, 2).shape == (1, 4)


def test_6_freq_layout(oracle_mod):
    x = np.array([[0.25, -0.5]], np.float32)
    out = oracle_mod.freq_encode(x, 3)[0]
    assert out.shape == (2 + 2 * 2 * 3,)
    exp = [0.25, -0.5]
    for f in range(3):
        exp += [np.sin(0.25 * 2 ** f), np.sin(-0.5 * 2 ** f), np.cos(0.25 * 2 ** f), np.cos(-0.5 * 2 ** f)]
    np.testing.assert_allclose(out, np.array(exp, np.float32), atol=2e-7)


def test_7_march_all_ones_all_zeros(oracle_mod):
    orc = oracle_mod
    o = np.array([[0.1, 4.0, -0.2]], np.float32)
    d = np.array([[0.0, -1.0, 0.0]], np.float32)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = orc.near_far_from_aabb(o, d, aabb, 0.05)
    alive = np.zeros(1, np.int32)
    ones = np.full(128 ** 3 // 8, 255, np.uint8)
    xyz, dirs, dl = orc.march_rays(1, 8, alive, nears, o, d, 1, ones, 1, 128, nears, fars, -1, False, 1 / 256, 16)
    assert DT == np.float32(0.027063293)
    t = nears[0]
    for k in range(8):
        assert dl[k, 0] == DT
        np.testing.assert_array_equal(xyz[k], np.array([0.1, np.float32(4.0 + t * -1.0), -0.2], np.float32))
        t = np.float32(t + DT)
        assert dl[k, 1] == t
    zeros = np.zeros(128 ** 3 // 8, np.uint8)
    xyz, dirs, dl = orc.march_rays(1, 8, alive, nears, o, d, 1, zeros, 1, 128, nears, fars, -1, False, 1 / 256, 16)
    assert not dl.any() and not xyz.any()
    # 128-padding always adds 1..128 slots
    assert orc.march_rays(1, 8, alive, nears, o, d, 1, zeros, 1, 128, nears, fars, 128, False, 1 / 256, 16)[0].shape[0] == 128
    assert orc.march_rays(16, 8, np.zeros(16, np.int32), nears, o, d, 1, zeros, 1, 128, nears, fars, 128, False, 1 / 256, 16)[0].shape[0] == 256


def test_8_composite_closed_form_and_termination(oracle_mod):
    orc = oracle_mod
    n_step, sigma, dt = 8, 20.0, float(DT)
    alive = np.zeros(1, np.int32)
    rays_t = np.array([3.5], np.float32)
    ws, dep, img = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros((1, 3), np.float32)
    deltas = np.stack([np.full(n_step, DT), 3.5 + DT * np.arange(1, n_step + 1)], 1).astype(np.float32)
    orc.composite_rays(1, n_step, alive, rays_t, np.full(n_step, sigma, np.float32), np.full((n_step, 3), 0.25, np.float32), deltas, ws, dep,
                       img, 1e-4)
    np.testing.assert_allclose(ws[0], 1 - np.exp(-sigma * dt * n_step), rtol=1e-5)
    np.testing.assert_allclose(img[0], 0.25 * ws[0], rtol=1e-5)
    assert alive[0] == 0 and rays_t[0] == deltas[-1, 1]
    # termination uses the PRE-sample transmittance: with T_thresh = 0.5 the sample that takes T below 0.5 is still
    # added, and so is the next one (whose pre-sample T is the first to be < 0.5); then the ray dies
    alive[:] = 0; ws[:] = 0; dep[:] = 0; img[:] = 0; rays_t[:] = 3.5
    orc.composite_rays(1, n_step, alive, rays_t, np.full(n_step, sigma, np.float32), np.ones((n_step, 3), np.float32), deltas, ws, dep, img, 0.5)
    a = 1 - np.exp(-sigma * dt)
    k_stop = next(k for k in range(n_step) if (1 - a) ** k < 0.5)         # first sample whose pre-sample T < 0.5
    np.testing.assert_allclose(ws[0], 1 - (1 - a) ** (k_stop + 1), rtol=1e-5)
    assert alive[0] == -1 and rays_t[0] == np.float32(3.5)
    # a slot with delta == 0 contributes nothing and kills the ray
    alive[:] = 0; ws[:] = 0; dep[:] = 0; img[:] = 0
    d0 = deltas.copy(); d0[2:] = 0
    orc.composite_rays(1, n_step, alive, rays_t, np.full(n_step, sigma, np.float32), np.ones((n_step, 3), np.float32), d0, ws, dep, img, 1e-4)
    np.testing.assert_allclose(ws[0], 1 - (1 - a) ** 2, rtol=1e-5)
    assert alive[0] == -1


def test_9_loop_budget(oracle_mod):
    """Samples per ray <= 23 and n_step is a pure function of the n_alive sequence."""
    from helpers import frame_case, oracle_render
    tr = []
    oracle_render(oracle_mod, frame_case("may_head", 48), trace=tr)
    N = 48 * 48
    assert tr[0] == (N, 1)
    for na, ns in tr:
        assert ns == max(min(N // na, 8), 1)
    assert sum(ns for _, ns in tr[:-1]) < 16 <= 23 and sum(ns for _, ns in tr) <= 23


def test_10_frame_level(oracle_mod):
    from helpers import frame_case, oracle_render
    orc = oracle_mod
    # sigma -> 0 everywhere: the head contributes nothing; output = background (head) ...
    case = frame_case("may_head", 32, sigma_gain=1e-9)
    last = "sigma_net.net.2.weight"
    case["sd"][last][0] = -1e7 * np.abs(case["sd"][last][0]) - 1e4        # exp(very negative) = 0
    res = oracle_render(orc, case)
    np.testing.assert_allclose(res["rgb_map"], 0.5, atol=1e-6)
    assert res["weights_sum"].max() < 1e-6
    # ... and with an empty torso grid the torso alpha is identically zero
    case = frame_case("may_torso", 32)
    case["sd"]["density_grid_torso"][:] = 0
    res = oracle_render(orc, case)
    assert not res["mask"].any() and not res["torso_alpha_map"].any()
    np.testing.assert_array_equal(res["torso_rgb_map"], np.full_like(res["torso_rgb_map"], 0.5))
