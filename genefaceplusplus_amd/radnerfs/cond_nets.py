"""Per-frame conditioning networks and the bias-free MLP container.

State-dict layout (names, shapes) is the reference's modules/radnerfs/cond_encoder.py:98-202 so checkpoints load
with strict=True:  AudioNet -> ``encoder_conv.{0,2,4,6}``, ``encoder_fc1.{0,2}``; AudioAttNet ->
``attentionConvNet.{0,2,4,6,8}``, ``attentionNet.0``; MLP -> ``net.{i}.weight`` (no bias).

These nets see a [smo_win, 1, 204] window once per frame (~0.1 MFLOP): they stay in PyTorch.  The
per-sample MLPs (ambient / sigma / colour / torso) are evaluated by the fused HIP kernels, which read the weights out of
the MLP containers below; ``MLP.forward`` (plain torch GEMMs) is kept for the stand-alone ``forward()/density()`` API.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_SLOPE = 0.02
# window length -> strides of the four k=3 convolutions (the table of cond_encoder.py:103-114, whose `== [5, 8]` branch
# is dead, so 5 and 8 are rejected like in the reference)
_STRIDES = {1: (1, 1, 1, 1), 2: (2, 1, 1, 1), 3: (2, 2, 1, 1), 4: (2, 2, 1, 1), 16: (2, 2, 2, 2)}


def _conv_stack(channels, strides):
    layers = []
    for (cin, cout), s in zip(zip(channels[:-1], channels[1:]), strides):
        layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=s, padding=1, bias=True), nn.LeakyReLU(_SLOPE, True)]
    return nn.Sequential(*layers)


class AudioNet(nn.Module):
    """[b, t_window, dim_in] -> [b, dim_aud]."""

    def __init__(self, dim_in=29, dim_aud=64, win_size=16):
        super().__init__()
        if win_size not in _STRIDES:
            raise ValueError("unsupported win_size")
        self.win_size = win_size
        self.dim_aud = dim_aud
        self.encoder_conv = _conv_stack((dim_in, 32, 32, 64, 64), _STRIDES[win_size])
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(_SLOPE, True), nn.Linear(64, dim_aud))

    def forward(self, x):
        h = self.encoder_conv(x.permute(0, 2, 1)).squeeze(-1)
        return self.encoder_fc1(h)


class AudioAttNet(nn.Module):
    """Attention smoother over the window: [seq_len, c] -> [c]."""

    def __init__(self, in_out_dim=64, seq_len=8):
        super().__init__()
        self.seq_len = seq_len
        self.in_out_dim = in_out_dim
        self.attentionConvNet = _conv_stack((in_out_dim, 16, 8, 4, 2, 1), (1, 1, 1, 1, 1))
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):
        scores = self.attentionConvNet(x[:, :self.in_out_dim].t().unsqueeze(0))
        w = self.attentionNet(scores.view(1, self.seq_len)).view(self.seq_len, 1)
        return (w * x).sum(dim=0)


_lib.register("gfpp_linear_weight_grad", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p])
#: rows from which a training-mode Linear layer takes the split-M weight-gradient kernel (below, the BLAS call is fine)
WGRAD_MIN_ROWS = 8192


class _LinearNoBias(torch.autograd.Function):
    """y = x W^T for the per-sample MLPs in training (cond_encoder.py:183-202).  Forward and the input gradient are plain GEMMs ([M, I] x [I, O]
    shapes the BLAS handles well); the weight gradient dW = dY^T X -- an O x I output with a reduction over the step's ~3 x 10^5 samples, 27-46 % of a
    training step through the BLAS heuristics -- is gfpp_linear_weight_grad.  Under autocast the operands are cast to half like F.linear's are."""

    @staticmethod
    def forward(ctx, x, weight):
        if torch.is_autocast_enabled():
            # (the caller -- MLP.forward -- only comes here under fp16 autocast or none: bf16 autocast takes the plain nn.Linear path)
            x, w = x.to(torch.float16), weight.to(torch.float16)
        else:
            x, w = x.float(), weight.float()
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.to(x.dtype).contiguous()
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            M, O, I = x.shape[0], w.shape[0], w.shape[1]
            # the kernel reads 16-byte vectors: a contiguous VIEW into a larger buffer (rows sliced off a batch) may start anywhere
            if x.data_ptr() % 16:
                x = x.clone()
            if gy.data_ptr() % 16:
                gy = gy.clone()
            gw = torch.empty(O, I, dtype=torch.float32, device=x.device)
            partial = torch.empty(512, O, I, dtype=torch.float32, device=x.device)
            _lib.call("gfpp_linear_weight_grad", gy.data_ptr(), x.data_ptr(), M, O, I, 1 if x.dtype == torch.float16 else 0, partial.data_ptr(), gw.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        return gx, gw


class MLP(nn.Module):
    """num_layers bias-free Linear layers, ReLU between them."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        dims = [dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        last = self.num_layers - 1
        # training batches (a step's samples x features): the layers' weight gradients through the split-M kernel
        own = (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS and max(l.out_features for l in self.net) <= 256
               and max(l.in_features for l in self.net) <= 160 and x.dtype in (torch.float32, torch.float16))
        if own and torch.is_autocast_enabled():
            # the split-M kernel has f16 and f32 operand forms: under bf16 autocast the layers run as nn.Linear does (F.linear casts to the autocast dtype)
            dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            own = dt == torch.float16
        for i, layer in enumerate(self.net):
            x = _LinearNoBias.apply(x, layer.weight) if own and layer.weight.requires_grad else layer(x)
            if i != last:
                x = F.relu(x, inplace=True)
        return x
