"""Host-side weight packing of the fused head kernels (frame_pipeline.py) checked against a numpy emulation of the MFMA register layouts
they are built for (MI355X_MICROARCH / cdna_hip_programming guides: A operand row = lane & 31, B operand column = lane & 31, the two
half-waves split K; accumulator register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31).  The emulation moves data
exactly like the kernels do -- a lane's own accumulators become its operands of the next layer -- so a wrong column table or K permutation
shows up as a wrong MLP output, with no GPU involved."""
import numpy as np
import torch

from genefaceplusplus_amd.configs import may_hparams
from genefaceplusplus_amd.radnerfs import frame_pipeline as fp


def _model():
    from genefaceplusplus_amd import radnerfs
    torch.manual_seed(3)
    m = radnerfs.RADNeRF(may_hparams("may_head"))
    for p in m.parameters():
        if p.dim() == 2 and p.shape[0] <= 256:
            torch.nn.init.normal_(p, std=0.2)
    return m.double()


def _row(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def _mfma16(P, B, acc):
    """One layer of K=16 steps.  P [S,4,64,8] fragments, B [S,64,8] operands (lane-major), acc [4,64,16] (tile, lane, register)."""
    S = P.shape[0]
    for m in range(4):
        # dense view of this tile: D[row, j] += sum_{s, h, e} P[s, m, row + 32 h, e] * B[s, j + 32 h, e]
        A = P[:, m].reshape(S, 2, 32, 8)                       # [s, h, row, e]
        X = B.reshape(S, 2, 32, 8)                             # [s, h, j, e]
        D = np.einsum("shre,shje->rj", A, X)
        for lane in range(64):
            j, hi = lane & 31, lane >> 5
            for r in range(16):
                acc[m, lane, r] += D[_row(r, hi), j]
    return acc


def _operands_from_acc(acc, act=lambda v: np.maximum(v, 0.0)):
    """relu_pack: lane keeps its own 64 accumulators: step s = tile s >> 1, registers 8 (s & 1) .. + 7."""
    B = np.zeros((8, 64, 8))
    for s in range(8):
        B[s] = act(acc[s >> 1, :, 8 * (s & 1):8 * (s & 1) + 8])
    return B


def _bias_to_acc(bias):
    acc = np.zeros((4, 64, 16))
    for m in range(4):
        for lane in range(64):
            for r in range(16):
                acc[m, lane, r] = bias[32 * m + _row(r, lane >> 5)]
    return acc


def _encoder_operands(feat, cols):
    """feat [n_feat, 32 samples]; cols[s][h][e] -> B [S, 64, 8]: lane (j, h) holds feature cols[s][h][e] of sample j."""
    S = len(cols)
    B = np.zeros((S, 64, 8))
    for s in range(S):
        for lane in range(64):
            for e in range(8):
                B[s, lane, e] = feat[cols[s][lane >> 5][e], lane & 31]
    return B


def _skinny(img, row, B):
    """skinny_dot: lane (j, h) sums img[h, row, 8 s + e] * B[s, lane, e]; the two half-waves are added."""
    part = np.einsum("sle,lse->l", B, img[(np.arange(64) >> 5)][:, row].reshape(64, 8, 8))
    return part[:32] + part[32:]


def test_16bit_weight_image_reproduces_the_head_mlps():
    m = _model()
    rng = np.random.default_rng(0)
    x_pos, x_amb = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    cond, ind, sh = rng.standard_normal(64), rng.standard_normal(4), rng.standard_normal((16, 32))
    A0, A1, A2 = (l.weight.detach().numpy() for l in m.ambient_net.net)
    S0, S1, S2 = (l.weight.detach().numpy() for l in m.sigma_net.net)
    C0, C1 = (l.weight.detach().numpy() for l in m.color_net.net)
    relu = lambda v: np.maximum(v, 0.0)
    # dense reference (radnerf.py:108-141 with the ambient grid features given)
    a2 = relu(A1 @ relu(A0[:, :32] @ x_pos + (A0[:, 32:] @ cond)[:, None]))
    amb_ref = A2 @ a2
    s2 = relu(S1 @ relu(S0 @ np.concatenate([x_pos, x_amb])))
    h = S2 @ s2
    c1 = relu(C0 @ np.concatenate([sh, h[1:], np.repeat(ind[:, None], 32, axis=1)]))
    rgb_ref = C1 @ c1

    W = fp.lp_weight_image(m, torch.float64).numpy()            # [31, 4, 64, 8]: amb0 2 | amb1 8 | sig0 4 | sig1 8 | colour 9
    K = fp.lp_skinny_image(m, torch.float64).numpy()            # [2, 7, 64]
    assert W.shape == (31, 4, 64, 8) and K.shape == (2, 7, 64)
    bpos = _encoder_operands(x_pos, fp.lp_cols_encoder(0))
    bamb = _encoder_operands(x_amb, fp.lp_cols_encoder(0))
    acc = _mfma16(W[0:2], bpos, _bias_to_acc(A0[:, 32:] @ cond))
    acc = _mfma16(W[2:10], _operands_from_acc(acc), np.zeros((4, 64, 16)))
    bh = _operands_from_acc(acc)
    amb = np.stack([_skinny(K, r, bh) for r in range(3)])
    np.testing.assert_allclose(amb, amb_ref, rtol=1e-9, atol=1e-9)
    acc = _mfma16(W[10:12], bpos, np.zeros((4, 64, 16)))
    acc = _mfma16(W[12:14], bamb, acc)
    acc = _mfma16(W[14:22], _operands_from_acc(acc), np.zeros((4, 64, 16)))
    bh = _operands_from_acc(acc)
    np.testing.assert_allclose(_skinny(K, 3, bh), h[0], rtol=1e-9, atol=1e-9)            # density logit
    bsh = _encoder_operands(sh, fp.lp_cols_sh())
    acc = _mfma16(W[22:31], np.concatenate([bsh, bh]), _bias_to_acc(C0[:, 144:] @ ind))  # merged sigma_net.2 x color_net.0
    bh = _operands_from_acc(acc)
    rgb = np.stack([_skinny(K, 4 + r, bh) for r in range(3)])
    np.testing.assert_allclose(rgb, rgb_ref, rtol=1e-8, atol=1e-8)


def _skinny_tile_row(i):
    """skinny_tile_chunk (frame_head_lp.hip, the skinny layers on the matrix pipe since round 4): image row that lane row i of the gathered 32-row tile reads."""
    i &= 15
    low = i & 3
    return 4 + min(low, 2) if i & 8 else low


def test_skinny_rows_as_one_gathered_mfma_tile():
    """The experiment build runs the skinny layers as MFMA chains: every lane takes its A operand from the skinny image through a row map, all three
    layers use the same tile, and each reads its rows from accumulator registers 0..6 -- in BOTH half-waves.  Emulated with the kernels' lane layout."""
    m = _model()
    rng = np.random.default_rng(5)
    K = fp.lp_skinny_image(m, torch.float64).numpy()            # [2, 7, 64]
    A2, S2row, C1 = m.ambient_net.net[2].weight.detach().numpy(), m.sigma_net.net[2].weight.detach().numpy()[0], m.color_net.net[1].weight.detach().numpy()
    dense = np.concatenate([np.concatenate([A2, np.zeros((3 - A2.shape[0], 128))]), S2row[None], C1])      # the 7 image rows, dense
    x = rng.standard_normal((128, 32))                          # activations of 32 samples
    cols = fp.lp_cols_act()
    B = _encoder_operands(x, cols)                              # what relu_pack leaves in the lanes
    tile = np.zeros((8, 1, 64, 8))
    for s in range(8):
        for lane in range(64):
            tile[s, 0, lane] = K[lane >> 5, _skinny_tile_row(lane & 31), 8 * s:8 * s + 8]
    acc = np.zeros((1, 64, 16))
    for s in range(8):                                          # (_mfma16 over one row tile)
        A = tile[s, 0].reshape(2, 32, 8)
        X = B[s].reshape(2, 32, 8)
        D = np.einsum("hre,hje->rj", A, X)
        for lane in range(64):
            for r in range(16):
                acc[0, lane, r] += D[_row(r, lane >> 5), lane & 31]
    want = dense @ x                                            # [7, 32]
    for lane in range(64):
        np.testing.assert_allclose(acc[0, lane, :7], want[:, lane & 31], rtol=1e-10, atol=1e-10)


def test_column_tables_are_permutations_held_by_the_right_half_wave():
    act = fp.lp_cols_act()
    flat = sorted(c for s in act for h in s for c in h)
    assert flat == list(range(128))
    for s in act:
        for h, cols in enumerate(s):
            assert all(((c % 32) >> 2) & 1 == h for c in cols)       # rows 4..7, 12..15, ... live in lanes 32-63
    enc = fp.lp_cols_encoder(0)
    assert sorted(c for s in enc for h in s for c in h) == list(range(32))
    for s in enc:
        for h, cols in enumerate(s):
            assert all((c // 2) % 2 == h for c in cols)               # half-wave h encodes the levels h, h+2, ...


def test_fp32_fragment_packing_reproduces_a_dense_layer():
    """pack_mfma for v_mfma_f32_32x32x2_f32: A lane l = (row l & 31, k = l >> 5), 4 K-steps per 16-byte chunk."""
    rng = np.random.default_rng(1)
    Wt = torch.from_numpy(rng.standard_normal((128, 128)))
    x = rng.standard_normal((128, 32))
    pairs = fp.activation_pairs(128)
    P = fp.pack_mfma(Wt, pairs).numpy()                          # [S/4, 4, 64, 4]
    S = len(pairs)
    D = np.zeros((128, 32))
    for s in range(S):
        for m in range(4):
            a = P[s // 4, m, :, s % 4]                           # per lane: W[32 m + (l & 31)][pairs[s][l >> 5]]
            for h in range(2):
                D[32 * m:32 * m + 32] += np.outer(a[32 * h:32 * h + 32], x[pairs[s][h]])
    np.testing.assert_allclose(D, Wt.numpy() @ x, rtol=1e-5, atol=1e-5)
    # the skinny VALU rows use the same accumulator order
    V = fp.pack_valu(torch.from_numpy(rng.standard_normal((3, 128)))).numpy()
    assert V.shape == (2, 3, 64)
