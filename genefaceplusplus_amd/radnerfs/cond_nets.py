"""Per-frame conditioning networks and the bias-free MLP container.

State-dict layout (names, shapes) is the reference's modules/radnerfs/cond_encoder.py:98-202 so checkpoints load
with strict=True:  AudioNet -> ``encoder_conv.{0,2,4,6}``, ``encoder_fc1.{0,2}``; AudioAttNet ->
``attentionConvNet.{0,2,4,6,8}``, ``attentionNet.0``; MLP -> ``net.{i}.weight`` (no bias).

These nets see a [smo_win, 1, 204] window once per frame (~0.1 MFLOP): they stay in PyTorch.  The
per-sample MLPs (ambient / sigma / colour / torso) are evaluated by the fused HIP kernels, which read the weights out of
the MLP containers below; ``MLP.forward`` (plain torch GEMMs) is kept for the stand-alone ``forward()/density()`` API.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, tuning

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
_u32 = ctypes.c_uint32
_lib.register("gfpp_mlp_train_pack", [ctypes.POINTER(ctypes.c_void_p), _u32, _u32, _u32, _u32, _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
_lib.register("gfpp_linear_weight_grad_scratch_floats", [ctypes.c_uint32, ctypes.c_uint32], restype=ctypes.c_uint64)
_lib.register("gfpp_mlp_train_image_bytes", [_u32, _u32, _u32, ctypes.c_int], restype=ctypes.c_uint32)
_lib.register("gfpp_mlp_train_forward", [ctypes.c_void_p, ctypes.c_void_p, _u32, _u32, _u32, _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
_lib.register("gfpp_mlp_train_backward", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _u32, _u32, _u32, _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
#: whole-MLP training launches under fp16 autocast (csrc/train_mlp_fused.hip); GFPP_TRAIN_FUSED_MLP=0: layer by layer (the A/B partner, and what other shapes use)
FUSED_MLP = tuning.HOST["train_fused_mlp"]
_FUSED_IN, _FUSED_OUT, _FUSED_HIDDEN = (64, 96, 160), (32, 160), 128


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
            partial = torch.empty(int(_lib.lib().gfpp_linear_weight_grad_scratch_floats(O, I)), dtype=torch.float32, device=x.device)
            _lib.call("gfpp_linear_weight_grad", gy.data_ptr(), x.data_ptr(), M, O, I, 1 if x.dtype == torch.float16 else 0, partial.data_ptr(), gw.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        return gx, gw


class _FusedMLP(torch.autograd.Function):
    """The whole MLP over a training batch under fp16 autocast: one forward launch (hidden activations kept), one backward launch for the input-gradient
    chain (gfpp_mlp_train_forward / _backward), the layers' weight gradients through the split-M kernel on the matrices those two wrote.  Same arithmetic as
    autocast's layer-by-layer graph: half operands, fp32 accumulation, every layer output rounded to half, relu's backward on the rounded activation."""

    @staticmethod
    def forward(ctx, x, in_pad, out_pad, padded_out, *weights):
        M, I = x.shape                         # I == in_pad: the caller built x with its zero columns already (their gradient comes back as zeros)
        NL, O = len(weights), weights[-1].shape[0]
        dev, st = x.device, torch.cuda.current_stream().cuda_stream
        xh = x.to(torch.float16)
        if I != in_pad:
            xh = F.pad(xh, (0, in_pad - I))
        xh = xh.contiguous()
        if xh.data_ptr() % 16:
            xh = xh.clone()
        ws = [w.detach().float().contiguous() for w in weights]
        ptrs = (ctypes.c_void_p * NL)(*[w.data_ptr() for w in ws])
        size = [int(_lib.lib().gfpp_mlp_train_image_bytes(NL, in_pad, out_pad, b)) for b in (0, 1)]
        fwd = torch.empty(size[0] // 2, dtype=torch.float16, device=dev)
        bwd = torch.empty(size[1] // 2, dtype=torch.float16, device=dev)
        _lib.call("gfpp_mlp_train_pack", ptrs, NL, int(weights[0].shape[1]), _FUSED_HIDDEN, O, in_pad, out_pad, fwd.data_ptr(), bwd.data_ptr(), st)
        acts = torch.empty(NL - 1, M, _FUSED_HIDDEN, dtype=torch.float16, device=dev)
        out = torch.empty(M, out_pad, dtype=torch.float16, device=dev)
        _lib.call("gfpp_mlp_train_forward", xh.data_ptr(), fwd.data_ptr(), M, in_pad, _FUSED_HIDDEN, NL, out_pad, acts.data_ptr(), out.data_ptr(), st)
        ctx.save_for_backward(xh, acts, bwd)
        ctx.shape = (I, O if not padded_out else out_pad, in_pad, out_pad, [tuple(w.shape) for w in weights], x.dtype)
        return out if padded_out else out[:, :O]

    @staticmethod
    def backward(ctx, gy):
        xh, acts, bwd = ctx.saved_tensors
        I, O, in_pad, out_pad, shapes, x_dtype = ctx.shape
        NL, M = len(shapes), xh.shape[0]
        dev, st = xh.device, torch.cuda.current_stream().cuda_stream
        gyp = gy.to(torch.float16)
        if O != out_pad:
            gyp = F.pad(gyp, (0, out_pad - O))
        gyp = gyp.contiguous()
        if gyp.data_ptr() % 16:
            gyp = gyp.clone()
        G = torch.empty(NL - 1, M, _FUSED_HIDDEN, dtype=torch.float16, device=dev)
        gx = torch.empty(M, in_pad, dtype=torch.float16, device=dev) if ctx.needs_input_grad[0] else None
        _lib.call("gfpp_mlp_train_backward", gyp.data_ptr(), acts.data_ptr(), bwd.data_ptr(), M, in_pad, _FUSED_HIDDEN, NL, out_pad, G.data_ptr(),
                  gx.data_ptr() if gx is not None else None, st)
        # the split-M kernel's slices: ONE scratch for the layers (the launches are stream-ordered), sized by the library for the largest of them
        widths = [in_pad] + [_FUSED_HIDDEN] * (NL - 1)
        outs_w = [_FUSED_HIDDEN] * (NL - 1) + [out_pad]
        partial = torch.empty(max(int(_lib.lib().gfpp_linear_weight_grad_scratch_floats(o, i)) for o, i in zip(outs_w, widths)), dtype=torch.float32, device=dev)
        gws = []
        for l in range(NL):
            if not ctx.needs_input_grad[4 + l]:
                gws.append(None)
                continue
            dy = gyp if l == NL - 1 else G[l]
            xin = xh if l == 0 else acts[l - 1]
            gw = torch.empty(dy.shape[1], xin.shape[1], dtype=torch.float32, device=dev)
            _lib.call("gfpp_linear_weight_grad", dy.data_ptr(), xin.data_ptr(), M, dy.shape[1], xin.shape[1], 1, partial.data_ptr(), gw.data_ptr(), st)
            o, i = shapes[l]
            gws.append(gw[:o, :i].contiguous() if (o, i) != tuple(gw.shape) else gw)
        gxo = None
        if gx is not None:
            gxo = (gx[:, :I] if I != in_pad else gx).to(x_dtype)
        return (gxo, None, None, None, *gws)


class SplitFirstColumn(torch.autograd.Function):
    """(h[:, 0], h[:, 1:1 + n]) of a zero-padded row-major [M, 1 + n + pad] matrix (sigma_net's output: density logit + geo_feat) whose backward assembles the
    padded gradient in ONE concatenation -- autograd's two slice backwards are two zero fills, two copies and an add over [M, 129], and the fused MLP's backward
    then pads the sum again."""

    @staticmethod
    def forward(ctx, h, n):
        ctx.n, ctx.width, ctx.dtype = n, h.shape[1], h.dtype
        return h[:, 0], h[:, 1:1 + n]

    @staticmethod
    def backward(ctx, g0, g1):
        M = (g0 if g0 is not None else g1).shape[0]
        dev = (g0 if g0 is not None else g1).device
        g0 = torch.zeros(M, 1, dtype=ctx.dtype, device=dev) if g0 is None else g0.to(ctx.dtype).unsqueeze(1)
        g1 = torch.zeros(M, ctx.n, dtype=ctx.dtype, device=dev) if g1 is None else g1.to(ctx.dtype)
        parts = [g0, g1]
        if ctx.width > 1 + ctx.n:
            parts.append(torch.zeros(M, ctx.width - 1 - ctx.n, dtype=ctx.dtype, device=dev))
        return torch.cat(parts, dim=1), None


class MLP(nn.Module):
    """num_layers bias-free Linear layers, ReLU between them."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        dims = [dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:]))

    def fused_widths(self, x, cols=None):
        """(in_pad, out_pad) if this call runs as the whole-MLP training launches (csrc/train_mlp_fused.hip): gradients on, a CUDA [M, features] batch of at least
        WGRAD_MIN_ROWS rows under fp16 autocast, hidden 128, 2 or 3 trainable layers; else None.  x may already carry the zero columns up to in_pad.
        cols: ask for an input that is still to be assembled (x then only lends its rows, device and dtype)."""
        if not (FUSED_MLP and torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS and torch.is_autocast_enabled()
                and x.dtype in (torch.float32, torch.float16)):
            return None
        dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
        if dt != torch.float16 or self.dim_hidden != _FUSED_HIDDEN or self.num_layers not in (2, 3) or self.dim_in > _FUSED_IN[-1] or self.dim_out > _FUSED_OUT[-1] \
                or not all(l.weight.requires_grad for l in self.net):
            return None
        in_pad = next(p for p in _FUSED_IN if self.dim_in <= p)
        out_pad = next(p for p in _FUSED_OUT if self.dim_out <= p)
        return (in_pad, out_pad) if (x.shape[1] if cols is None else cols) in (self.dim_in, in_pad) else None

    def forward_padded(self, x):
        """The fused training launches with the output left zero-padded ([M, out_pad], columns >= dim_out are zeros): the caller slices it (SplitFirstColumn) and
        the gradient arrives padded too -- no pad / slice copies of [M, 129] matrices.  None when fused_widths() says the call takes the layer-by-layer path."""
        w = self.fused_widths(x)
        return None if w is None else _FusedMLP.apply(x, w[0], w[1], True, *[l.weight for l in self.net])

    def forward(self, x):
        last = self.num_layers - 1
        # training batches (a step's samples x features): the layers' weight gradients through the split-M kernel
        own = (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS and max(l.out_features for l in self.net) <= 256
               and max(l.in_features for l in self.net) <= 160 and x.dtype in (torch.float32, torch.float16))
        if own and torch.is_autocast_enabled():
            # the split-M kernel has f16 and f32 operand forms: under bf16 autocast the layers run as nn.Linear does (F.linear casts to the autocast dtype)
            dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
            own = dt == torch.float16
            w = self.fused_widths(x) if own else None
            if w is not None:
                return _FusedMLP.apply(x, w[0], w[1], False, *[l.weight for l in self.net])
        for i, layer in enumerate(self.net):
            x = _LinearNoBias.apply(x, layer.weight) if own and layer.weight.requires_grad else layer(x)
            if i != last:
                x = F.relu(x, inplace=True)
        return x
