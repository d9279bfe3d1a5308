"""Super-resolution stage: the oracle restatement (oracle/sr_oracle.py) against golden vectors produced by the
reference's own Superresolution module (tests/golden/make_golden_sr.py), and the product's state-dict layout against the reference's."""
import json
import os

import numpy as np
import pytest

from genefaceplusplus_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "sr_golden.npz"))


def _inputs():
    rng = np.random.default_rng(11)
    a = rng.random((1, 3, 256, 256)).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, 256, dtype=np.float32), np.linspace(0, 1, 256, dtype=np.float32), indexing="ij")
    b = np.stack([0.5 + 0.5 * np.sin(9 * xx + 3 * yy), yy * xx, 0.25 + 0.5 * (np.cos(17 * yy) > 0)], 0)[None].astype(np.float32)
    return {"noise": a, "smooth": b}


@pytest.mark.parametrize("name", ["noise", "smooth"])
@pytest.mark.parametrize("mode", ["const", "none"])
def test_sr_oracle_reproduces_reference(golden, name, mode):
    from oracle import sr_oracle
    sd = syn.synthetic_sr_state(prefix="sr_net.")
    y = sr_oracle.superresolution(_inputs()[name], sd, noise_mode=mode)
    assert y.shape == (1, 3, 512, 512) and y.dtype == np.float32
    crops = np.stack([y[0, :, r:r + 32, c:c + 32] for r, c in golden["crops"]])
    np.testing.assert_allclose(crops, golden[f"{name}.{mode}.crops"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y.astype(np.float64).sum(axis=(0, 2, 3)), golden[f"{name}.{mode}.sum"], rtol=1e-5, atol=0.5)
    np.testing.assert_allclose(np.abs(y).astype(np.float64).sum(axis=(0, 2, 3)), golden[f"{name}.{mode}.abs_sum"], rtol=1e-5)


def test_sr_state_layout_matches_reference():
    """sr_net.* keys / shapes / dtypes of the product's SR models == the reference's Superresolution(channels=3).state_dict()."""
    import torch
    from genefaceplusplus_amd import radnerfs
    from genefaceplusplus_amd.configs import may_hparams
    manifest = json.load(open(os.path.join(HERE, "golden", "sr_state_manifest.json")))
    for cls, variant in ((radnerfs.RADNeRFTorsowithSR, "may_torso_sr"),):
        model = cls(may_hparams(variant))
        mine = {k[len("sr_net."):]: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items() if k.startswith("sr_net.")}
        assert mine == manifest
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in syn.synthetic_sr_state().items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all(not k.startswith("sr_net.") for k in missing)


def test_host_side_weight_folding_matches_the_oracle_layers():
    """The product never runs modulation / demodulation / transposed convolution / FIR filtering on the device: they are folded into plain 3x3
    weights on the host (superres.py::effective_weight, _compose_up_weights).  Plain torch convolutions with the folded weights must
    reproduce the oracle's modulated_conv2d, for the plain layers and for the x2 layer (4 phases x 64 channels, depth-to-space)."""
    import torch
    import torch.nn.functional as F
    from oracle import sr_oracle
    from genefaceplusplus_amd.radnerfs.superres import Superresolution, _compose_up_weights
    sd = syn.synthetic_sr_state(prefix="")
    sr = Superresolution(channels=3)
    sr.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    rng = np.random.default_rng(2)
    filt = torch.from_numpy(sd["resample_filter"]).float()
    for name, layer in (("block0.conv1", sr.block0.conv1), ("block1.conv0", sr.block1.conv0), ("block1.conv1", sr.block1.conv1)):
        x = torch.from_numpy(rng.standard_normal((1, layer.in_channels, 10, 12)).astype(np.float32))
        styles = sr_oracle.styles_from_ones(sd[name + ".affine.weight"], sd[name + ".affine.bias"])
        ref = sr_oracle.modulated_conv2d(x, sd[name + ".weight"], styles, None, layer.up, 1, filt, True)
        w = layer.effective_weight()
        if layer.up == 1:
            got = F.conv2d(x.double(), w, padding=1)
        else:
            wc = _compose_up_weights(w, filt)                                             # [4 * Cout, Cin, 3, 3], channel = Cout * (2 py + px) + o
            y = F.conv2d(x.double(), wc, padding=1).reshape(1, 2, 2, layer.out_channels, 10, 12)
            got = y.permute(0, 3, 4, 1, 5, 2).reshape(1, layer.out_channels, 20, 24)      # (2 y + py, 2 x + px)
        np.testing.assert_allclose(got.numpy(), ref.double().numpy(), rtol=2e-4, atol=2e-4)
    # ToRGB: modulation only, 1x1
    x = torch.from_numpy(rng.standard_normal((1, 64, 6, 6)).astype(np.float32))
    styles = sr_oracle.styles_from_ones(sd["block1.torgb.affine.weight"], sd["block1.torgb.affine.bias"]) * (1.0 / np.sqrt(64))
    ref = sr_oracle.modulated_conv2d(x, sd["block1.torgb.weight"], styles, None, 1, 0, filt, False)
    got = F.conv2d(x.double(), sr.block1.torgb.effective_weight().double().t().reshape(3, 64, 1, 1))   # stored [in, out]
    np.testing.assert_allclose(got.numpy(), ref.double().numpy(), rtol=2e-4, atol=2e-4)


def test_training_path_of_the_sr_stage_matches_oracle_and_differentiates():
    """Superresolution._forward_autograd (plain torch ops, what training mode runs) == the oracle (itself pinned on the reference's golden
    vectors), and its gradients agree with central differences."""
    import torch
    from oracle import sr_oracle
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    sr = Superresolution(channels=3)
    sr.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    x = _inputs()["smooth"]
    for mode in ("const", "none"):
        with torch.no_grad():
            got = sr._forward_autograd(torch.from_numpy(x), mode).numpy()
        ref = sr_oracle.superresolution(x, syn.synthetic_sr_state(prefix="sr_net."), noise_mode=mode)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-4)
    # gradients (float64, small image, no noise): d loss / d (input, one conv weight, one affine bias) vs central differences
    sr = sr.double()
    torch.manual_seed(0)
    xs = torch.rand(1, 3, 8, 8, dtype=torch.float64, requires_grad=True)
    target = torch.rand(1, 3, 16, 16, dtype=torch.float64)
    loss_fn = lambda: ((sr._forward_autograd(xs, "none") - target) ** 2).mean()
    loss_fn().backward()
    for t in (xs, sr.block1.conv0.weight, sr.block0.conv1.affine.bias, sr.block1.torgb.weight):
        g = t.grad.clone()
        assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
        d = g / g.norm()
        eps = 1e-5
        with torch.no_grad():
            t.add_(eps * d); up = float(loss_fn()); t.sub_(2 * eps * d); down = float(loss_fn()); t.add_(eps * d)
        assert abs((up - down) / (2 * eps) - float((g * d).sum())) <= 1e-5 * max(1.0, float(g.norm()))


def test_training_path_of_the_sr_stage_matches_the_reference_module(golden):
    """One training step of the reference's own Superresolution blocks (tests/golden/make_golden_sr.py: noise_mode 'random' from a seeded
    generator, photometric loss, backward) against the product's training path: output, input gradient, parameter gradients.  The noise fields
    are drawn with the same four torch.randn calls, so the seeded generator yields the same noise."""
    import torch
    from genefaceplusplus_amd.radnerfs.superres import Superresolution
    sd = syn.synthetic_sr_state(prefix="")
    sr = Superresolution(channels=3)
    sr.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    sr.train()
    x = torch.from_numpy(_inputs()["smooth"]).clone().requires_grad_(True)
    torch.manual_seed(13)
    y = sr._forward_autograd(x, "random")
    torch.manual_seed(14)
    target = torch.rand(1, 3, 512, 512)
    loss = ((y - target) ** 2).mean()
    loss.backward()
    yn = y.detach().numpy()
    crops = np.stack([yn[0, :, r:r + 32, c:c + 32] for r, c in golden["crops"]])
    np.testing.assert_allclose(crops, golden["train.crops"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(yn.astype(np.float64).sum(axis=(0, 2, 3)), golden["train.sum"], rtol=1e-5, atol=0.5)
    assert abs(float(loss.detach()) - float(golden["train.loss"][0])) <= 1e-5
    np.testing.assert_allclose(x.grad.numpy()[0, :, 100:132, 60:92], golden["train.grad_input_crop"], rtol=2e-3, atol=2e-9)
    assert abs(float(np.abs(x.grad.numpy()).astype(np.float64).sum()) - float(golden["train.grad_input_abs_sum"][0])) <= 1e-3 * float(golden["train.grad_input_abs_sum"][0])
    named = dict(sr.named_parameters())
    for key in golden.files:
        if key.startswith("train.grad."):
            name = key[len("train.grad."):]
            got, want = named[name].grad.numpy().astype(np.float64), golden[key].astype(np.float64)
            assert np.linalg.norm(got - want) <= 1e-3 * np.linalg.norm(want) + 1e-9, (name, np.linalg.norm(got - want), np.linalg.norm(want))
