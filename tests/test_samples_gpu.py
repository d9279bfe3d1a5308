"""Per-sample parity of the MFMA radiance MLPs, below the compositing that frame-level tests average over.

gfpp_head_eval_samples[_lp] runs the trip kernels' own evaluate_block / evaluate_block_lp on a caller-supplied sample list.  Checked against
  (1) the reference's OWN RADNeRF.forward (fwd.* of tests/golden/ref_python_golden.npz, produced by tests/golden/make_golden.py from
      /root/reference/modules/radnerfs/radnerf.py:108-141 on CPU fp32), and
  (2) the CPU oracle on 20 000 samples (incl. positions outside the box => zero features) for every kernel instantiation the library ships:
      tiled / hash tables (the SLOW template), linear / smoothstep, ambient D = 3 / 2, bound 1 / 2.
Tolerances (stated, per precision): fp32 = exact-fp32 MFMA, differs from BLAS only by summation order; fp16 / bf16 round the MLP operands to
11 / 8 significant bits (accumulation fp32).  sigma = exp(h): its relative error is the absolute error of the logit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import frame_case, build_model

# Bars per precision over ALL samples: (max relative error of sigma = max abs error of the density logit, max abs error of colour, of the
# ambient coordinate), and the same three as MEANS -- a permuted weight row or a mis-folded bias moves most samples by O(0.1..1), far above
# the mean bars, while the max bars bound the worst rounding case.  Measured on the MI355X (random-init weights with gains of 3..6 per layer,
# harsher than a trained field): fp32 1.3e-5..3.5e-5 / 1.2e-5 / 5e-7; fp16 1.5e-2..4.1e-2 / 1.4e-2 / 6e-4; bf16 with ambient_net on bf16 operands (round 4)
# 0.10..0.41 / 0.10 / 5e-3.
BARS = {"fp32": ((1e-4, 5e-5, 2e-6), (1e-5, 2e-6, 2e-7)),
        "fp16": ((1e-1, 4e-2, 2e-3), (1e-2, 3e-3, 3e-4)),
        # round 5: ambient_net runs on f16 operands inside the bf16 mode (LpAmbient): the ambient coordinate is the fp16 mode's, and sigma / colour lose the
        # part of their error that came from the features of displaced ambient cells -- the fp16 bars hold (round 4's bf16 bars: 8e-1 / 2e-1 / 1.5e-2 max,
        # 8e-2 / 2.5e-2 / 2.5e-3 mean; CPU emulation of the new mode, tools/lp_emulate.py's rounding points: 2.8e-2 / 1.3e-2 / 5.6e-4 max)
        "bf16": ((1e-1, 4e-2, 2e-3), (1e-2, 3e-3, 3e-4))}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _check(got, ref, precision, tag):
    (rt, ca, aa), (rm, cm, am) = BARS[precision]
    sg, cg, ag = (x.float().cpu().numpy() for x in got)
    sr, cr, ar = ref
    assert sg.shape == sr.shape and cg.shape == cr.shape and ag.shape == ar.shape, tag
    rel = np.abs(sg - sr) / np.maximum(np.abs(sr), 1e-30)
    stats = {"sigma_rel": float(rel.max()), "color": float(np.abs(cg - cr).max()), "ambient": float(np.abs(ag - ar).max()),
             "sigma_rel_mean": float(rel.mean()), "color_mean": float(np.abs(cg - cr).mean()), "ambient_mean": float(np.abs(ag - ar).mean())}
    print(tag, precision, stats)
    assert stats["sigma_rel"] <= rt and stats["color"] <= ca and stats["ambient"] <= aa, (tag, precision, stats)
    assert stats["sigma_rel_mean"] <= rm and stats["color_mean"] <= cm and stats["ambient_mean"] <= am, (tag, precision, stats)
    return stats


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_forward_matches_the_reference_python_golden(dev, golden, precision):
    case = frame_case("may_head", 64)
    model = build_model(case, dev, "fused")
    model.precision = precision
    P, Dn = torch.from_numpy(golden["fwd.position"]).to(dev), torch.from_numpy(golden["fwd.direction"]).to(dev)
    cond_feat = torch.from_numpy(golden["may_head.cond_feat"]).to(dev)
    with torch.no_grad():
        got = model(P, Dn, cond_feat, model.individual_embeddings[0])
    _check(got, (golden["fwd.sigma"], golden["fwd.color"], golden["fwd.ambient"]), precision, "reference-python golden")
    if precision == "fp32":
        # the golden's 4 hand-placed positions include a corner of the box and the origin (radnerf.py forward has no clamp)
        assert np.isfinite(got[0].cpu().numpy()).all()


def test_fused_forward_equals_the_torch_layer_path(dev, golden):
    """Same module, executor 'staged' (nn.Linear on rocBLAS + the stand-alone grid kernels) vs 'fused' (evaluate_block): wiring check
    that does not involve the oracle at all."""
    case = frame_case("may_head", 64)
    model = build_model(case, dev, "fused")
    model.precision = "fp32"
    P, Dn = torch.from_numpy(golden["fwd.position"]).to(dev), torch.from_numpy(golden["fwd.direction"]).to(dev)
    cf = torch.from_numpy(golden["may_head.cond_feat"]).to(dev)
    with torch.no_grad():
        a = model(P, Dn, cf, model.individual_embeddings[0])
        model.executor = "staged"
        b = model(P, Dn, cf, model.individual_embeddings[0])
    _check(a, tuple(x.float().cpu().numpy() for x in b), "fp32", "fused vs torch layers")


INSTANCES = [
    ("tiled_linear_amb3", "may_head", {}),
    ("hash_linear_amb3", "may_head", {"grid_type": "hashgrid"}),
    ("tiled_smoothstep_amb3", "may_head", {"grid_interpolation_type": "smoothstep"}),
    ("tiled_linear_amb2_audio", "audio_head", {}),
    ("hash_smoothstep_amb2", "audio_head", {"grid_type": "hashgrid", "grid_interpolation_type": "smoothstep"}),
    ("tiled_linear_bound2", "may_head", {"bound": 2}),
]


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("tag,variant,over", INSTANCES, ids=[i[0] for i in INSTANCES])
def test_every_instantiation_vs_oracle(dev, oracle_mod, tag, variant, over, precision):
    case = frame_case(variant, 64, hp_over=over)
    hp, sd = case["hp"], case["sd"]
    model = build_model(case, dev, "fused")
    model.precision = precision
    rng = np.random.default_rng(17)
    M = 20000 + 13                                                   # ragged: not a multiple of the 32-sample block
    b = float(hp["bound"])
    P = (rng.uniform(-1, 1, (M, 3)) * np.array([0.45, 0.4, 0.5]) * b).astype(np.float32)
    P[:3] = [[b, b / 2, -b], [0, 0, 0], [-b, -b, b]]
    P[3] = [1.2 * b, 0.1, 0.0]                                       # outside the box: zero position features (gridencoder.cu:110-135)
    Dn = rng.standard_normal((M, 3)).astype(np.float32)
    Dn /= np.linalg.norm(Dn, axis=1, keepdims=True)
    cond_feat = oracle_mod.cal_cond_feat(case["cond"], sd, hp, case["eye_area_percent"])
    ind = sd["individual_embeddings"][0]
    ref = oracle_mod.head_forward(P, Dn, cond_feat, ind, sd, hp)
    with torch.no_grad():
        got = model(torch.from_numpy(P).to(dev), torch.from_numpy(Dn).to(dev), torch.from_numpy(np.asarray(cond_feat)).to(dev),
                    model.individual_embeddings[0])
    assert got[2].shape[1] == hp["ambient_coord_dim"]
    _check(got, ref, precision, tag)


def test_eval_samples_empty_and_unfolded_inputs(dev):
    case = frame_case("may_head", 64)
    model = build_model(case, dev, "fused")
    model.precision = "fp32"
    z = torch.zeros(0, 3, device=dev)
    with torch.no_grad():
        s, c, a = model(z, z, torch.zeros(64, device=dev), model.individual_embeddings[0])
    assert s.shape == (0,) and c.shape == (0, 3) and a.shape == (0, 3)
    from genefaceplusplus_amd._lib import GfppError
    with pytest.raises(GfppError):
        with torch.no_grad():
            model(torch.zeros(4, 3, device=dev), torch.zeros(4, 3, device=dev), torch.zeros(7, device=dev), None)      # cond_feat of the wrong length
