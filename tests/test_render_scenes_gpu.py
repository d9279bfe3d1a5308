"""Frame-level parity of the fused path on the scenes a TRAINED checkpoint produces, not only the convex synthetic ellipsoid (round-3 review, item 1):

  (a) non-convex occupancy -- rays that go occupied -> empty -> occupied: the kernel-level speckle grid (tests/ref_kernel_cases.py:42-54), a hollow
      shell with detached blobs, and a bitfield the product's own update_extra_state leaves (renderer.py:202-284), cascades 1 and 2, 512^2 and 256^2;
  (b) T_thresh in {0.05 (the CLI's --fast, genefacepp_infer.py:591-592), 0.0 (never terminates by transmittance), 0.5};
  (c) grid tables at the reference's init scale U(-1e-4, 1e-4) (grid.py:141-143);
  (d) cameras that clip or miss the AABB (NaN-depth convention of rays whose near = far = FLT_MAX, renderer.py:396).

Every case runs through product_render on the fused executor: fp32 against the oracle inside SURVEY 8c's tolerance WITH the reference's (n_alive, n_step)
sequence, fp16 / bf16 inside their stated bars, and the persistent launch (k_begin_premarch's sample_t store + occ_aabb cut, the workgroup-local trips, the
histogram budget) bit-equal to the trip launches including the alive counts it reconstructs."""
import numpy as np
import pytest
import torch

from helpers import frame_case, oracle_render, build_model, product_render, compare_frames, nonconvex_occupancy, pose_at, trained_case

pytestmark = pytest.mark.gpu

BARS = {"fp16": 2e-2, "bf16": 2e-2}
# Frame bars per 16-bit precision: (PSNR vs the fp32 oracle, share of pixels beyond BARS) = SURVEY 8c's 45 dB / 2e-2 / 0.05 % for BOTH modes.  fp16 -- the
# reference's own inference precision (autocast) -- measures 56-67 dB on these scenes.  bf16 (BASELINE configs[2]'s arithmetic) measured 43.1-47.3 dB here in
# round 4 and ran under a lowered bar (42 dB / 5e-2 / 0.1 %); tools/lp_emulate.py located the loss in ONE layer group -- ambient_net, whose output is a coordinate
# of the second hash grid -- and since round 5 the bf16 mode multiplies that group as f16 (csrc/frame_head_lp.hip::LpAmbient; CPU emulation of the same scenes:
# 57-59 dB), so the lowered bar is gone.
FRAME_BARS = {"fp16": (45.0, 5e-4), "bf16": (45.0, 5e-4)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / max(mse, 1e-20))


def _rgb(res, variant):
    rgb = res["rgb_map"].float().cpu().numpy()
    if variant in ("may_torso_sr", "may_head_sr"):
        rgb = np.transpose(rgb, (0, 2, 3, 1))
    return rgb.reshape(-1, 3)


def _model(case, dev, precision, kernel=None):
    model = build_model(case, dev, "fused")
    model.precision = precision
    model.use_graph = False
    if hasattr(model, "sr_net"):
        model.sr_net.ready = False                    # head + torso passes are compared here; the SR stage has its own tests
    if kernel is not None:
        model.pipeline().lp_kernel = kernel
    return model


def _outputs(res):
    return {k: v.detach().cpu().numpy().copy() for k, v in res.items() if torch.is_tensor(v)}


def check_all_modes(dev, orc, case, tag, frac16=None, psnr16=None, alive16_rel=2e-3, precisions=("fp16", "bf16")):
    """fp32 vs oracle (+ identical trip schedule); 16-bit modes vs oracle inside their bars; persist == trips bit for bit (+ alive counts) in every precision."""
    variant, HW = case["variant"], case["HW"]
    N = HW * HW
    trace = []
    ref = oracle_render(orc, case, trace=trace)
    rref = ref["rgb_map"].reshape(-1, 3)
    n_alive_ref = [n for n, _ in trace]
    n_step_ref = [s for _, s in trace]

    model = _model(case, dev, "fp32")
    res = product_render(model, case, dev, "oracle", orc)
    stats = compare_frames(res, ref, variant, HW)
    alive, samples = model.pipeline().trip_counters(N)
    got_alive = [int(a) for a in alive[:len(trace)]]
    got_step = [max(min(N // a, 8), 1) for a in got_alive if a > 0]
    print(tag, "fp32", stats, "trips", trace)
    assert got_step == n_step_ref, (got_step, n_step_ref)                       # the step budget of every ray follows from this sequence
    # ... and the reference's own alive counts, trip for trip -- up to the rays whose transmittance crosses T_thresh within the fp32 summation order of the
    # MLP layers (SURVEY 8c allows 0.05 % of the pixels for them; measured: at most 1 ray of 262 144 ends one trip earlier or later)
    worst = max(abs(a - b) for a, b in zip(got_alive, n_alive_ref))
    print(tag, "fp32 n_alive: largest difference to the reference's loop", worst, "rays")
    assert worst <= max(2, int(2e-5 * N)), (got_alive, n_alive_ref)
    # the fp32 frame above came from ONE launch (gfpp_head_frame_persist, round 4): the trip launches give the same bits and the same counts
    assert model.pipeline().lp_kernel == "persist" and int(model.pipeline().budget(N)["hist"].sum()) == N
    trips_model = _model(case, dev, "fp32", "trips")
    res_trips = _outputs(product_render(trips_model, case, dev, "oracle", orc))
    for k, v in _outputs(res).items():
        np.testing.assert_array_equal(v, res_trips[k], err_msg=f"{tag} fp32 {k}")
    np.testing.assert_array_equal(alive[:26], trips_model.pipeline().trip_counters(N)[0][:26])

    for precision in precisions:
        outs = {}
        for kernel in ("persist", "trips"):
            m = _model(case, dev, precision, kernel)
            r = product_render(m, case, dev, "oracle", orc)
            torch.cuda.synchronize()
            outs[kernel] = _outputs(r)
            a, s = m.pipeline().trip_counters(N)
            outs[kernel]["_alive"] = a[:26].copy()
            if kernel == "persist":
                assert m.pipeline().lp_kernel == "persist"
                b = m.pipeline().budget(N)
                assert int(b["hist"].sum()) == N, (int(b["hist"].sum()), N)
                rgb = _rgb(r, variant)
                err = np.abs(rgb - rref).max(axis=1)
                st = {"psnr": _psnr(rgb, rref), "rgb_max": float(err.max()), "frac_over": float((err > BARS[precision]).mean())}
                print(tag, precision, st, "alive", [int(x) for x in a[:len(trace) + 1]])
                want_psnr, want_frac = FRAME_BARS[precision]
                if psnr16 is not None:
                    want_psnr, want_frac = min(want_psnr, psnr16), max(want_frac, frac16)
                assert st["psnr"] >= want_psnr and st["frac_over"] <= want_frac, (precision, st)
                # the reconstructed schedule is the reference's up to the rays whose transmittance crosses T_thresh within the 16-bit rounding
                for k, n in enumerate(n_alive_ref):
                    assert abs(int(a[k]) - n) <= max(2, alive16_rel * N), (precision, k, int(a[k]), n)
        for k in outs["trips"]:
            np.testing.assert_array_equal(outs["persist"][k], outs["trips"][k], err_msg=f"{tag} {precision} {k}")


# ---- (a) non-convex occupancy --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["speckle", "shell"])
@pytest.mark.parametrize("variant,HW,over", [("may_torso", 512, None), ("may_torso_sr", 256, None), ("may_head", 160, {"bound": 2})],
                         ids=["torso512", "torso_sr256", "head160_cascade2"])
def test_nonconvex_occupancy(dev, oracle_mod, variant, HW, over, kind):
    case = nonconvex_occupancy(frame_case(variant, HW, hp_over=over), kind)
    occ = np.unpackbits(case["sd"]["density_bitfield"]).mean()
    assert 0.005 < occ < 0.2, occ
    check_all_modes(dev, oracle_mod, case, f"{kind}/{variant}/{HW}")


@pytest.mark.parametrize("variant,HW,over", [("may_torso", 256, None), ("may_torso", 512, None), ("may_head", 128, {"bound": 2})], ids=["torso256", "torso512", "head128_cascade2"])
def test_occupancy_left_by_update_extra_state(dev, oracle_mod, variant, HW, over):
    """The bitfield a checkpoint carries: the product's own update_extra_state (renderer.py:202-284 -- jittered probe of the density field per cell,
    morton3D_dilation, decaying maximum, threshold min(mean_density, density_thresh), packbits) on the synthetic field, twice; then the frame through
    every mode.  The random field's level set is as non-convex as an occupancy gets: isolated cells and holes everywhere inside the box."""
    import random
    case = frame_case(variant, HW, hp_over=over)
    # (the torso classes' update_extra_state refreshes the TORSO grid with the head frozen, radnerf_torso.py:201-240: the head's bitfield in a torso checkpoint
    # is the one the head stage's NeRFRenderer.update_extra_state left -- so the head model of the same weights produces it)
    head_case = frame_case("may_head", HW, hp_over=over)
    model = build_model(head_case, dev, "fused")
    for k in ("position_embedder.embeddings", "ambient_embedder.embeddings", "sigma_net.net.0.weight", "density_bitfield"):
        np.testing.assert_array_equal(head_case["sd"][k], case["sd"][k])
    rng = np.random.default_rng(11)
    model.conds = torch.from_numpy(np.clip(rng.standard_normal((40, 1, model.cond_in_dim)), -1.5, 1.5).astype(np.float32))
    random.seed(3)
    torch.manual_seed(3)
    model.density_grid.zero_()
    for _ in range(2):
        model.update_extra_state(decay=0.95)
    grid = model.density_grid.cpu().numpy().copy()
    bits = model.density_bitfield.cpu().numpy().copy()
    thresh = min(model.mean_density, model.density_thresh)
    np.testing.assert_array_equal(bits, oracle_mod.packbits(grid, thresh))
    occ = np.unpackbits(bits).mean()
    print("update_extra_state bitfield: occupied", occ, "mean density", model.mean_density, "threshold", thresh)
    assert 0.02 < occ, occ
    case["sd"] = dict(case["sd"], density_grid=grid, density_bitfield=bits)
    check_all_modes(dev, oracle_mod, case, f"extra_state/{variant}/{HW}")


# ---- (b) T_thresh --------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T_thresh", [0.05, 0.0, 0.5])
@pytest.mark.parametrize("variant,HW,kind", [("may_torso", 256, None), ("may_torso", 128, "shell"), ("may_torso_sr", 256, "speckle")],
                         ids=["ellipsoid256", "shell128", "sr_speckle256"])
def test_T_thresh_values(dev, oracle_mod, variant, HW, kind, T_thresh):
    """0.05 is what `--fast` renders with (genefacepp_infer.py:566,591-592); 0.0 never ends a ray by transmittance (every ray runs to the step budget: the
    snapshot path of the persistent launch); 0.5 ends most rays after their first opaque sample.  The quantity T_thresh moves is e, the first sample whose
    pre-sample transmittance is below it (raymarching.cu:1006) -- what the histogram budget is built on."""
    case = frame_case(variant, HW)
    if kind:
        case = nonconvex_occupancy(case, kind)
    case["T_thresh"] = T_thresh
    # T = 0.5: a ray whose transmittance passes 0.5 within the 16-bit rounding of sigma ends one sample earlier or later and that sample carries up to half
    # the pixel: the per-pixel bars hold for all but a few rays per thousand, the PSNR bar for the frame
    # (round 6: the frame bar at T = 0.5 is the survey's 45 dB too -- measured 53-58 dB --; what cannot hold there is the share of pixels beyond 2e-2: 0.09-0.22 % measured
    # against the survey's 0.05 %, the rays whose transmittance passes 0.5 within the rounding; bar 0.3 %)
    loose = T_thresh >= 0.5
    check_all_modes(dev, oracle_mod, case, f"T{T_thresh}/{variant}/{HW}/{kind}", frac16=3e-3 if loose else None, psnr16=45.0 if loose else None,
                    alive16_rel=1e-2 if loose else 2e-3)


def test_fast_threshold_at_full_size(dev, oracle_mod):
    """BASELINE's 512 x 512 head + torso frame at the `--fast` threshold, every mode."""
    case = frame_case("may_torso", 512)
    case["T_thresh"] = 0.05
    check_all_modes(dev, oracle_mod, case, "T0.05/may_torso/512")


# ---- (c) tables at the reference's init scale ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,HW", [("may_torso", 128), ("may_torso_sr", 256)])
def test_reference_init_scale_tables(dev, oracle_mod, variant, HW):
    """GridEncoder.reset_parameters: U(-1e-4, 1e-4) at every level (grid.py:141-143) -- what an untrained model renders with: features ~1e-4, sigma ~ 1,
    colours ~ 0.5, no ray ever terminates (the thin-scene schedule)."""
    case = frame_case(variant, HW, table_scale=1e-4, table_decay=0.0)
    tab = case["sd"]["position_embedder.embeddings"]
    assert float(np.abs(tab).max()) <= 1e-4 and float(np.abs(tab[-65536:]).max()) > 5e-5       # the finest level too
    check_all_modes(dev, oracle_mod, case, f"init_tables/{variant}/{HW}")


# ---- (d) cameras that clip / miss the box -------------------------------------------------------------------------------------------------
CAMERAS = {
    "inside_box": dict(distance=0.3),                               # the camera sits inside the AABB (|y| < 0.5): near = min_near
    "close": dict(distance=0.9, yaw_deg=12.0),
    "half_misses": dict(distance=4.0, shift=(1.05, 0.0, 0.0)),      # translated sideways: the columns beyond x = 1 miss the box
    "yaw40": dict(distance=4.0, yaw_deg=40.0),                      # rays cross the thin slab diagonally
    "all_miss": dict(distance=4.0, shift=(3.0, 0.0, 0.0)),          # every ray passes beside the box: nears = fars = FLT_MAX everywhere, depth NaN, image = torso / background
    "box_behind": dict(distance=4.0, away=True),                    # looking away from the box: the slab test has no `far < 0` exit (raymarching.cu:91-145), so near = min_near > far -- no sample, depth 0
    "far_camera": dict(distance=12.0),                              # the whole box and the space beside it in view: hits through the ellipsoid in the middle, misses at the sides
    "grazing_top": dict(distance=4.0, shift=(0.0, 0.0, 0.98)),      # the optical axis runs along the top face of the box
}


@pytest.mark.parametrize("cam", sorted(CAMERAS))
@pytest.mark.parametrize("variant,HW,kind", [("may_torso", 128, None), ("may_head", 96, "speckle")], ids=["torso128", "head96_speckle"])
def test_cameras_that_clip_or_miss_the_box(dev, oracle_mod, variant, HW, kind, cam):
    case = frame_case(variant, HW)
    if kind:
        case = nonconvex_occupancy(case, kind)
    case["pose"] = pose_at(**CAMERAS[cam])
    rays = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
    nears, fars = oracle_mod.near_far_from_aabb(rays["rays_o"][0], rays["rays_d"][0], case["sd"]["aabb_infer"], case["hp"]["min_near"])
    miss = float((nears > 1e30).mean())
    print(cam, "rays that miss the box", miss, "near range", float(nears[nears < 1e30].min()) if miss < 1 else None)
    if cam == "all_miss":
        assert miss == 1.0
    if cam == "box_behind":
        assert miss == 0.0 and float(fars.max()) < 0.0 < float(nears.min())
    if cam == "far_camera":
        assert 0.05 < miss < 0.8
    if cam == "half_misses":
        assert 0.2 < miss < 0.8
    if cam == "inside_box":
        assert abs(float(nears.min()) - case["hp"]["min_near"]) < 1e-6
    # (since round 6 in every precision.  bf16's alive counts: a camera INSIDE the box marches every ray through the whole field, and the rays whose transmittance
    # sits at T_thresh within bf16's 8-bit significand of sigma are 0.4 % of the frame there (67 of 16 384 on trip 3; frame 56 dB, no pixel over the bar))
    check_all_modes(dev, oracle_mod, case, f"camera/{cam}/{variant}", alive16_rel=6e-3)


# ---- (e) a TRAINED field ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,HW", [("may_torso", 512), ("may_torso_sr", 256), ("may_head", 512), ("may_head_sr", 256)])
def test_trained_field(dev, oracle_mod, variant, HW):
    """Weights that are not random (round-5 review, first item): the May architecture fitted to the procedural clip by the package's own training path
    (tools/make_trained_checkpoint.py; tests/golden/trained/, curves in fit_log.json), rendered at the resolution its class is served at, a held-out frame
    with the clip's own driving signals and per-pixel background.  Every mode against the oracle inside SURVEY 8c with the reference's schedule
    (check_all_modes), and the fp32 / 16-bit frames against the ANALYTIC target the field was trained on."""
    case = trained_case(variant, HW)
    check_all_modes(dev, oracle_mod, case, f"trained/{variant}/{HW}")
    r = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
    tgt = case["clip"].target(case["frame_idx"], torch.from_numpy(r["rays_o"][0]), torch.from_numpy(r["rays_d"][0]), torch.from_numpy(oracle_mod.get_bg_coords(HW, HW))[0])
    gt = tgt["gt"].numpy()
    for precision in ("fp32", "fp16", "bf16"):
        res = product_render(_model(case, dev, precision), case, dev, "oracle", oracle_mod)
        p = _psnr(_rgb(res, variant), gt)
        print(f"trained/{variant}/{HW}", precision, "PSNR against the procedural target", round(p, 2), "dB")
        assert p >= 36.0, (precision, p)
