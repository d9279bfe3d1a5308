"""The host side of the polyphase up-sampling launch (k_sr_up_poly, opt-in: include/gfpp_radnerf.h gfpp_tuning.sr_up_poly) without a GPU: the two operand tables
radnerfs/superres.py builds for it -- the 3 x 3 weights in tap-group order (gfpp_sr_model.w_up_poly) and the FIR as a GEMM operand (gfpp_sr_model.up_fir_g) -- are
unpacked again and run through a numpy restatement of the kernel's data flow (polyphase products on the 18 x 14 position grid of a 16 x 12 patch, T channel-major
`[py][my][px][mx]`, an output row = two runs of 80 T entries x the coefficient table), and the result must be conv2d_resample(up = 2) of the same weights:
stride-2 transposed convolution + the 4 x 4 FIR with gain 4 (conv2d_resample.py:117-133), here in torch fp64 on the CPU."""
import numpy as np
import torch
import torch.nn.functional as F

from genefaceplusplus_amd.radnerfs import superres

PW, PH, GW, GH = 16, 12, 18, 14                 # kUpPW, kUpPH, kUpGW, kUpGH of csrc/superres.hip
SHIFT = ((0, 0),) * 4 + ((0, -1),) * 2 + ((-1, 0),) * 2 + ((-1, -1),)          # (dy, dx) of the input a tap multiplies, taps in _POLY_TAPS order
PHASE = (0, 1, 2, 3, 0, 2, 0, 1, 0)             # kPh: which of T_ee, T_eo, T_oe, T_oo (2 py + px) a tap feeds


def _reference(x, w_eff, fir2):
    """x [H, W, 128], w_eff [64, 128, 3, 3] -> [2H, 2W, 64]."""
    t = F.conv_transpose2d(x.permute(2, 0, 1)[None], w_eff.transpose(0, 1), stride=2)
    t = F.pad(t, [1, 1, 1, 1])
    y = F.conv2d(t, (fir2 * 4.0).flip([0, 1])[None, None].repeat(64, 1, 1, 1), groups=64)
    return y[0].permute(1, 2, 0)


def _kernel_data_flow(x, frags, gtab):
    H, W = x.shape[:2]
    # fragments [nt][ks][tap][s][lane = 32 h + j][e] -> W[tap][n = 32 nt + j][c = 64 ks + 16 s + 8 h + e]
    w = frags.double().numpy().reshape(2, 2, 9, 4, 2, 32, 8).transpose(2, 0, 5, 1, 3, 4, 6).reshape(9, 64, 128)
    # table [pair][s][lane][e] -> coefficient of run entry k = 16 s + 8 h + e for output column j
    g = gtab.double().numpy().reshape(2, 5, 2, 32, 8).transpose(0, 1, 2, 4, 3).reshape(2, 80, 32)
    xn = x.numpy()

    def at(yy, xx):
        return xn[yy, xx] if 0 <= yy < H and 0 <= xx < W else np.zeros(128)
    out = np.zeros((2 * H, 2 * W, 64))
    for y0 in range(0, H, PH):
        for x0 in range(0, W, PW):
            T = np.zeros((64, 2, GH, 2, GW))                         # [ch][py][my][px][mx]
            for my in range(GH):
                for mx in range(GW):
                    for tap in range(9):
                        dy, dx = SHIFT[tap]
                        if my + dy < 0 or mx + dx < 0:
                            continue                                  # the kernel reads a clamped position there: entries nobody uses
                        T[:, PHASE[tap] >> 1, my, PHASE[tap] & 1, mx] += w[tap] @ at(y0 - 1 + my + dy, x0 - 1 + mx + dx)
            flat = np.concatenate([T.reshape(64, -1), np.zeros((64, 20))], axis=1)       # a channel's padding (kUpChStride): coefficient 0 x 0
            for il in range(2 * PH):
                if 2 * y0 + il >= 2 * H:
                    break
                Y, av = il >> 1, il & 1
                my = Y + 1
                run0, run1 = (0 * GH + my) * 2 * GW, (1 * GH + (my if av else my - 1)) * 2 * GW
                row = flat[:, run0:run0 + 80] @ g[1 if av else 0] + flat[:, run1:run1 + 80] @ g[0 if av else 1]      # [64, 32]
                out[2 * y0 + il, 2 * x0:2 * x0 + 32] = row.T
    return out


def test_polyphase_tables_reproduce_conv2d_resample_up2():
    g = torch.Generator().manual_seed(11)
    H, W = 28, 32                                # three patch rows (the last one partial), two patch columns
    x = (torch.randint(-8, 9, (H, W, 128), generator=g).double() / 8.0)
    w_eff = (torch.randint(-16, 17, (64, 128, 3, 3), generator=g).double() / 64.0)         # exact in f16: the test isolates the layout
    f1 = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64) / 8.0
    frags = superres._pack_up_poly(w_eff)
    gtab = superres._fir_gemm_table([float(v * 2.0) for v in f1])          # per-axis gain 2 (gfpp_sr_model.fir)
    assert tuple(frags.shape) == (2, 2, 9, 4, 64, 8) and tuple(gtab.shape) == (2, 5, 64, 8)
    got = _kernel_data_flow(x, frags, gtab)
    want = _reference(x, w_eff, torch.outer(f1, f1)).numpy()
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), float(np.abs(got - want).max())


def test_fir_table_is_exact_and_sparse():
    """The coefficients are products of 0.25 and 0.75: exact in f16 (the table asserts that itself), 16 non-zero taps per output value: per column 4 x 2 rows x 2 pairs."""
    f1 = [0.25, 0.75, 0.75, 0.25]
    gtab = superres._fir_gemm_table(f1).double().numpy().reshape(2, 5, 2, 32, 8).transpose(0, 1, 2, 4, 3).reshape(2, 80, 32)
    assert ((gtab != 0).sum(axis=1) == 8).all()                      # a run = 2 T rows x 4 column taps
    assert np.allclose(gtab.sum(axis=1), 2.0)                        # (g1 + g3) = (g0 + g2) = 1 of a row pair x the four column taps' sum 2: gain 2 per axis
    assert (gtab[:, 72:] == 0).all()                                 # the eight padding entries of a run
