"""Host-side pieces of the round-5 training path that need no GPU: the padded sigma split, the decision which calls take the whole-MLP launches, the descriptor the
conditioning-network kernels are handed (slot order = gradient order, pointers at the parameters themselves)."""
import ctypes

import torch

from genefaceplusplus_amd.configs import may_hparams
from genefaceplusplus_amd.radnerfs import cond_nets


def test_split_first_column_is_the_two_slices_with_one_padded_gradient():
    torch.manual_seed(0)
    h = torch.randn(7, 160, requires_grad=True)
    ref = h.detach().clone().requires_grad_(True)
    a, b = cond_nets.SplitFirstColumn.apply(h, 128)
    ra, rb = ref[:, 0], ref[:, 1:129]
    assert torch.equal(a, ra) and torch.equal(b, rb)
    ga, gb = torch.randn(7), torch.randn(7, 128)
    (a * ga).sum().backward(retain_graph=True)
    (b * gb).sum().backward()
    ((ra * ga).sum() + (rb * gb).sum()).backward()
    assert torch.equal(h.grad, ref.grad) and float(h.grad[:, 129:].abs().max()) == 0.0
    # only one of the two outputs used: the other half of the gradient is zeros
    h2 = torch.randn(3, 160, requires_grad=True)
    _, b2 = cond_nets.SplitFirstColumn.apply(h2, 128)
    b2.sum().backward()
    assert float(h2.grad[:, 0].abs().max()) == 0.0 and float((h2.grad[:, 1:129] - 1).abs().max()) == 0.0


def test_fused_training_launches_are_for_cuda_fp16_autocast_batches_only():
    m = cond_nets.MLP(64, 129, 128, 3)
    x = torch.randn(cond_nets.WGRAD_MIN_ROWS, 64)
    assert m.fused_widths(x) is None and m.forward_padded(x) is None          # CPU tensor
    assert m(x).shape == (cond_nets.WGRAD_MIN_ROWS, 129)                       # ... takes the layer-by-layer path
    # the width tables: what the kernels are instantiated for (csrc/train_mlp_fused.hip::fm_shape_ok)
    assert cond_nets._FUSED_IN == (64, 96, 160) and cond_nets._FUSED_OUT == (32, 160) and cond_nets._FUSED_HIDDEN == 128


def test_conditioning_descriptor_points_at_the_parameters_in_gradient_order():
    from genefaceplusplus_amd import radnerfs
    from genefaceplusplus_amd.radnerfs.frame_pipeline import cond_train_model, CondModel
    for variant, blink in (("may_head", 0), ("may_head_sr", 2), ("audio_head", 0)):
        hp = may_hparams(variant)
        model = (radnerfs.RADNeRFwithSR if hp.get("with_sr") else radnerfs.RADNeRF)(hp)
        cm, plist, fill = cond_train_model(model)
        assert isinstance(cm, CondModel) and cm.center_tap_only == 0 and not cm.blob and cm.blink_dim == blink and cm.with_att == 1
        assert (cm.smo, cm.t_win, cm.c_in, cm.dim_aud) == (hp["smo_win_size"], hp["cond_win_size"], model.cond_in_dim, model.cond_out_dim)
        # every conditioning parameter exactly once, none of the field's
        want = {id(p) for n, p in model.named_parameters() if n.startswith(("cond_prenet", "cond_att_net", "blink_"))}
        assert {id(p) for p in plist} == want and len(plist) == len(want)
        assert cm.conv_w[0] == plist[0].data_ptr() and cm.conv_b[0] == plist[1].data_ptr() and cm.att_fc_b == plist[-1].data_ptr()
        # a gradient descriptor: the same slots, other addresses
        fake = [0x1000 * (i + 1) for i in range(len(plist))]
        gm = fill(fake)
        assert gm.conv_w[0] == fake[0] and gm.att_fc_b == fake[-1] and ctypes.sizeof(gm) == ctypes.sizeof(cm)


def test_swap23_row_order_makes_operand_quads_natural_feature_order():
    """The layout invariant csrc/train_mlp_fused.hip is built on (lp_mfma_device.h's 32x32x16 conventions): after a layer, lane (j, h) holds in acc[t][r] the MFMA row
    32 t + (r & 3) + 8 (r >> 2) + 4 h of column j, and packs acc[s >> 1][8 (s & 1) + e] into operand (step s, element e).  With the weight image putting MFMA row i of
    tile t on feature 32 t + swap23(i), that operand element IS feature 16 s + 8 h + e: an operand register quad = 16 contiguous bytes of a row-major [M, features] matrix."""
    swap23 = lambda i: (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)
    assert sorted(swap23(i) for i in range(32)) == list(range(32))
    for s in range(8):
        for h in range(2):
            for e in range(8):
                t, r = s >> 1, 8 * (s & 1) + e
                mfma_row = (r & 3) + 8 * (r >> 2) + 4 * h                # row of tile t this lane's acc[t][r] holds
                assert 32 * t + swap23(mfma_row) == 16 * s + 8 * h + e
    # and the transposing-read kernel's pitch rule: the 8 rows x 64 bytes one ds_read_b64_tr_b16 instruction touches fall into distinct banks
    for C in (32, 64, 96, 128, 160):
        pitch = C * 2 + 64 if (C * 2) % 128 == 0 else C * 2
        assert pitch % 256 in (64, 192) and pitch % 16 == 0
        banks = set()
        for row in list(range(4)) + list(range(8, 12)):
            for word in range(16):                                       # 64 bytes = 16 banks per row
                banks.add(((row * pitch) // 4 + word) % 64)
        assert len(banks) == 64
