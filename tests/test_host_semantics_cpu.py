"""Host-side semantics the reference's trainer relies on (round-1 advisor findings), CPU only."""
import torch

from genefaceplusplus_amd.configs import may_hparams
from genefaceplusplus_amd.radnerfs.camera import trunc_exp


def test_trunc_exp_backward_is_clamped_like_the_reference():
    """utils.py:34-47: forward exp(x) (fp32), backward g * exp(clamp(x, -15, 15))."""
    x = torch.tensor([-20.0, -1.0, 0.0, 3.0, 15.0, 20.0, 60.0], requires_grad=True)
    y = trunc_exp(x)
    assert y.dtype == torch.float32 and torch.equal(y.detach(), torch.exp(x.detach()))
    y.backward(torch.ones_like(y))
    want = torch.exp(x.detach().clamp(-15, 15))
    assert torch.allclose(x.grad, want) and torch.isfinite(x.grad).all()
    assert float(x.grad[5]) == float(torch.exp(torch.tensor(15.0)))            # x = 20: clamped, not e^20
    h = torch.tensor([1.0, 2.0], dtype=torch.float16, requires_grad=True)      # half logits (autocast): result is fp32 like the reference's
    assert trunc_exp(h).dtype == torch.float32


def test_training_stage_switches_exist_and_toggle_the_right_parameters():
    """tasks/radnerfs/radnerf_torso_sr.py:192 calls model.on_train_torso_nerf(); radnerf_sr.py:116-122, radnerf_torso_sr.py:58-73."""
    from genefaceplusplus_amd.radnerfs import RADNeRFTorsowithSR, RADNeRFwithSR
    m = RADNeRFTorsowithSR(may_hparams("may_torso_sr"))
    m.on_train_torso_nerf()
    on = {n for n, p in m.named_parameters() if p.requires_grad}
    assert on and all(n.startswith(("torso_", "head_color_weights_encoder.")) for n in on), sorted(on)[:5]
    assert not any(n.startswith("sr_net.") for n in on) and any(n.startswith("torso_deform_net.") for n in on)
    m.on_train_superresolution()
    on = {n for n, p in m.named_parameters() if p.requires_grad}
    assert on and all(n.startswith("sr_net.") for n in on)
    h = RADNeRFwithSR(may_hparams("may_head_sr"))
    h.on_train_nerf()
    assert all(p.requires_grad != n.startswith("sr_net.") for n, p in h.named_parameters())
    h.on_train_superresolution()
    assert all(p.requires_grad == n.startswith("sr_net.") for n, p in h.named_parameters())
