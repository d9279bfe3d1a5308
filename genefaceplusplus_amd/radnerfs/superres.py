"""Super-resolution stage of the *_sr models: drop-in for the reference's ``Superresolution`` (modules/radnerfs/radnerf_sr.py:14-43
on top of modules/eg3ds/models/superresolution.py:159-258 and networks_stylegan2.py:37-94, 286-478).

Same constructor, ``state_dict()`` key set / shapes / dtypes (``block{0,1}.{conv0,conv1,torgb}.{weight,bias,affine.*,noise_const,
noise_strength,resample_filter}``, pinned by tests/golden/sr_state_manifest.json) and ``forward(rgb, noise_mode=...)`` signature.
Execution differs: ``Superresolution`` feeds ``ws = ones`` (radnerf_sr.py:32-33), so every style vector is a constant of the
checkpoint -- modulation, demodulation and, for the up-sampling layer, the transposed convolution + FIR filter are folded into plain
3x3 convolution weights once (fp64, re-done when the parameters change), and a frame is four implicit-GEMM MFMA launches
(csrc/superres.hip) instead of ~40 PyTorch / cuDNN / plugin launches.  Activations are f16 like the reference's GPU path
(use_fp16=True); images and accumulation fp32.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .._lib import call, GfppError

c_p = ctypes.c_void_p
c_f = ctypes.c_float


def _as_i64(u):
    """An unsigned 64-bit value as the int64 torch stores."""
    u &= 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


class SrModel(ctypes.Structure):
    _fields_ = [("w_first", c_p), ("w_b0c1", c_p), ("w_up", c_p), ("w_b1c1", c_p), ("bias", c_p * 4), ("noise_strength", c_f * 4),
                ("rgb0_w", c_p), ("rgb0_b", c_p), ("rgb1_w", c_p), ("rgb1_b", c_p), ("fir", c_f * 4), ("conv_clamp", c_f), ("w_up_poly", c_p), ("up_fir_g", c_p)]


class SrWs(ctypes.Structure):
    _fields_ = [("x0", c_p), ("x1", c_p), ("x2", c_p), ("img256", c_p), ("rng_state", c_p), ("rng_seed", ctypes.c_uint64), ("clamp01", ctypes.c_uint32),
                ("clip_job", c_p), ("clip_lane", ctypes.c_uint32), ("clip_sub", ctypes.c_uint32), ("clip_advance", ctypes.c_uint32), ("up_prof", c_p)]


_lib.register("gfpp_sr_forward", [ctypes.POINTER(SrModel), ctypes.POINTER(SrWs), c_p, c_p, c_p, c_p])
STRUCT_MIRRORS = {"sr_model": SrModel, "sr_ws": SrWs}


def _setup_filter():
    f = torch.tensor([1.0, 3.0, 3.0, 1.0])
    f = torch.outer(f, f)
    return f / f.sum()


class _Affine(nn.Module):
    """FullyConnectedLayer(w_dim, in_channels, bias_init=1) (networks_stylegan2.py:99-133); only ever applied to ws = ones."""

    def __init__(self, w_dim, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_features, w_dim))
        self.bias = nn.Parameter(torch.ones(out_features))
        self.weight_gain = 1.0 / np.sqrt(w_dim)

    def styles_for_ones(self):
        return (self.weight.detach().double() * self.weight_gain).sum(dim=1) + self.bias.detach().double()

    def styles_autograd(self):
        """the same, as part of the autograd graph (training)"""
        return (self.weight * self.weight_gain).sum(dim=1) + self.bias


class _SynthesisLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, up=1):
        super().__init__()
        self.in_channels, self.out_channels, self.resolution, self.up = in_channels, out_channels, resolution, up
        self.affine = _Affine(w_dim, in_channels)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, 3, 3))
        self.register_buffer("noise_const", torch.randn(resolution, resolution))
        self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.register_buffer("resample_filter", _setup_filter())

    def effective_weight(self):
        """modulate with the constant styles, demodulate (networks_stylegan2.py:60-70) -> [out, in, 3, 3] float64."""
        w = self.weight.detach().double() * self.affine.styles_for_ones().reshape(1, -1, 1, 1)
        d = (w.square().sum(dim=[1, 2, 3]) + 1e-8).rsqrt()
        return w * d.reshape(-1, 1, 1, 1)


    def forward_autograd(self, x, noise, conv_clamp):
        """SynthesisLayer.forward (networks_stylegan2.py:321-344) in plain differentiable torch ops: modulate / demodulate the weight
        (:60-70), correlate -- for up = 2 a stride-2 transposed convolution followed by the 4x4 FIR filter with gain 4
        (conv2d_resample.py:117-133) --, add noise, bias, leaky ReLU(0.2) x sqrt(2), clamp."""
        w = self.weight * self.affine.styles_autograd().reshape(1, -1, 1, 1)
        w = w * (w.square().sum(dim=[1, 2, 3]) + 1e-8).rsqrt().reshape(-1, 1, 1, 1)
        if self.up == 1:
            x = F.conv2d(x, w, padding=1)
        else:
            f = self.resample_filter.to(x.dtype)
            x = F.conv_transpose2d(x, w.transpose(0, 1), stride=2)                     # [H, W] -> [2H + 1, 2W + 1]
            x = F.pad(x, [1, 1, 1, 1])
            x = F.conv2d(x, (f * 4.0).flip([0, 1])[None, None].repeat(x.shape[1], 1, 1, 1), groups=x.shape[1])   # -> [2H, 2W]
        if noise is not None:
            x = x + noise * self.noise_strength
        x = F.leaky_relu(x + self.bias.reshape(1, -1, 1, 1), 0.2) * float(np.sqrt(2.0))
        return x.clamp(-conv_clamp, conv_clamp)


class _ToRGB(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim):
        super().__init__()
        self.affine = _Affine(w_dim, in_channels)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, 1, 1))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.weight_gain = 1.0 / np.sqrt(in_channels)

    def effective_weight(self):
        """modulated, NOT demodulated (networks_stylegan2.py:363-366) -> [in, out] float64."""
        s = self.affine.styles_for_ones() * self.weight_gain
        return (self.weight.detach().double()[:, :, 0, 0] * s.reshape(1, -1)).t().contiguous()


    def forward_autograd(self, x, conv_clamp):
        """ToRGBLayer.forward (networks_stylegan2.py:363-368): modulated 1x1, no demodulation, linear, clamp."""
        w = self.weight * (self.affine.styles_autograd() * self.weight_gain).reshape(1, -1, 1, 1)
        return (F.conv2d(x, w) + self.bias.reshape(1, -1, 1, 1)).clamp(-conv_clamp, conv_clamp)


class _Block(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, up):
        super().__init__()
        self.register_buffer("resample_filter", _setup_filter())
        self.conv0 = _SynthesisLayer(in_channels, out_channels, w_dim, resolution, up=up)
        self.conv1 = _SynthesisLayer(out_channels, out_channels, w_dim, resolution)
        self.torgb = _ToRGB(out_channels, 3, w_dim)


def _pack_conv3(w_eff, nt):
    """[Cout, Cin, 3, 3] -> f16 fragments [passes, 9, Cin/16, nt, 64, 8]: lane (i, h) of tile t holds W[pass*32nt + 32t + i][16 s + 8 h + e][tap]."""
    cout, cin = w_eff.shape[:2]
    per_pass = 32 * nt
    assert cout % per_pass == 0 and cin % 16 == 0
    w = w_eff.reshape(cout // per_pass, nt, 32, cin // 16, 2, 8, 9)          # [pass, t, i, s, h, e, tap]
    w = w.permute(0, 6, 3, 1, 4, 2, 5)                                       # [pass, tap, s, t, h, i, e]
    return w.reshape(cout // per_pass, 9, cin // 16, nt, 64, 8).to(torch.float16).contiguous()


_POLY_TAPS = ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (1, 2), (2, 0), (2, 1), (2, 2))      # (ky, kx), grouped by the input shift they multiply (csrc/superres.hip::k_sr_up_poly)


def _pack_up_poly(w_eff):
    """[64, 128, 3, 3] -> f16 fragments [2 nt][2 ks][9 taps][4 steps][64 lanes][8]: lane (j, h) of step s holds W[32 nt + j][64 ks + 16 s + 8 h + e][ky][kx]."""
    cout, cin = w_eff.shape[:2]
    assert (cout, cin) == (64, 128)
    w = w_eff.reshape(2, 32, 2, 4, 2, 8, 3, 3)                              # [nt, j, ks, s, h, e, ky, kx]
    taps = torch.stack([w[..., ky, kx] for ky, kx in _POLY_TAPS], dim=0)    # [tap, nt, j, ks, s, h, e]
    frag = taps.permute(1, 3, 0, 4, 5, 2, 6)                                # [nt, ks, tap, s, h, j, e]
    return frag.reshape(2, 2, 9, 4, 64, 8).to(torch.float16).contiguous()


def _fir_gemm_table(fir):
    """The FIR of the polyphase up-sampling layer as a GEMM operand (gfpp_sr_model.up_fir_g): [2 pairs][5 steps][64 lanes][8] f16."""
    g = torch.zeros(2, 5, 64, 8, dtype=torch.float64)
    for pair, rows in enumerate(((1, 3), (0, 2))):
        for s in range(5):
            for lane in range(64):
                j, h = lane & 31, lane >> 5
                for e in range(8):
                    k = 16 * s + 8 * h + e
                    r, k36 = divmod(k, 36)
                    if r >= 2:
                        continue
                    px, mx = divmod(k36, 18)
                    t = 2 * (mx - 1) + px - (j - 1)
                    if 0 <= t < 4:
                        g[pair, s, lane, e] = float(fir[rows[r]]) * float(fir[t])
    h16 = g.to(torch.float16)
    assert torch.equal(h16.double(), g), "the FIR products must be exact in f16"
    return h16.contiguous()


def _pack_first(w_eff):
    """[128, 3, 3, 3] -> [2 steps, 4 tiles, 64, 8] f16 with k = 3 tap + channel (27 used of 32)."""
    flat = torch.zeros(128, 32, dtype=torch.float64, device=w_eff.device)
    flat[:, :27] = w_eff.permute(0, 2, 3, 1).reshape(128, 27)               # [n][ky][kx][c] -> k = (3 ky + kx) 3 + c
    w = flat.reshape(4, 32, 2, 2, 8).permute(2, 0, 3, 1, 4)                  # [t, i, s, h, e] -> [s, t, h, i, e]
    return w.reshape(2, 4, 64, 8).to(torch.float16).contiguous()


def _compose_up_weights(w_eff, filt):
    """The up = 2 path of conv2d_resample (conv2d_resample.py:117-133: conv_transpose2d stride 2, then the FIR filter with gain 4) as ONE
    3x3 convolution at the low resolution with 4 x Cout output channels, channel = 64 phase + o, phase = 2 py + px <-> output pixel
    (2y + py, 2x + px).  Both operations are linear and shift invariant up to the stride, so the composed taps are read off the response
    to a unit impulse, once per 3x3 basis kernel (fp64)."""
    cout, cin = w_eff.shape[:2]
    f = filt.double()
    resp = torch.zeros(3, 3, 32, 32, dtype=torch.float64)
    x = torch.zeros(1, 1, 16, 16, dtype=torch.float64)
    x[0, 0, 8, 8] = 1.0
    for ky in range(3):
        for kx in range(3):
            k = torch.zeros(1, 1, 3, 3, dtype=torch.float64)
            k[0, 0, ky, kx] = 1.0
            # conv2d_resample(up=2, padding=1, flip_weight=False), see oracle/sr_oracle.py::conv2d_resample for the padding arithmetic
            y = F.conv_transpose2d(x, k.transpose(0, 1), stride=2, padding=0)            # 33 x 33
            y = F.pad(y, [1, 1, 1, 1])
            y = F.conv2d(y, (f * 4.0).flip([0, 1])[None, None])                           # 32 x 32
            resp[ky, kx] = y[0, 0]
    # output (2y + py, 2x + px) from input (y + dy, x + dx):  resp[2 (8 - dy) + py][2 (8 - dx) + px]
    C = torch.zeros(3, 3, 2, 2, 3, 3, dtype=torch.float64)                                # [ky, kx, py, px, dy+1, dx+1]
    covered = torch.zeros(32, 32, dtype=torch.bool)
    for py in range(2):
        for px in range(2):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    r, c = 2 * (8 - dy) + py, 2 * (8 - dx) + px
                    C[:, :, py, px, dy + 1, dx + 1] = resp[:, :, r, c]
                    covered[r, c] = True
    assert float(resp[:, :, ~covered].abs().max()) == 0.0, "the composed kernel must fit a 3x3 neighbourhood"
    w = torch.einsum("oikl,klpqab->pqoiab", w_eff.double().cpu(), C)                      # [py, px, o, i, 3, 3]
    return w.reshape(4 * cout, cin, 3, 3).to(w_eff.device)


class Superresolution(nn.Module):
    ready = True

    def __init__(self, channels=3, img_resolution=512, sr_antialias=True):
        super().__init__()
        assert img_resolution == 512
        assert channels == 3, "the radnerfs models instantiate Superresolution(channels=3)"
        self.sr_antialias = sr_antialias
        self.input_resolution = 256
        self.w_dim = 16
        self.conv_clamp = 256.0
        self.block0 = _Block(channels, 128, self.w_dim, 256, up=1)
        self.block1 = _Block(128, 64, self.w_dim, 512, up=2)
        self.register_buffer("resample_filter", _setup_filter())
        self._packed = None

    # -- packing (once per parameter version) -------------------------------------------------------------------------------------
    def _fingerprint(self):
        from .frame_pipeline import FramePipeline
        return FramePipeline._fingerprint(self)         # (address, version) of every parameter and buffer without walking the module tree through generators

    def _pack(self):
        fp = self._fingerprint()
        if self._packed is not None and self._packed["fp"] == fp:
            return self._packed
        dev = self.resample_filter.device
        if dev.type != "cuda":
            raise GfppError("Superresolution: the module must live on the GPU (there is no CPU path)")
        _lib.lib()
        b0, b1 = self.block0, self.block1
        keep = {
            "w_first": _pack_first(b0.conv0.effective_weight()).to(dev),
            "w_b0c1": _pack_conv3(b0.conv1.effective_weight(), 4).to(dev),
            "w_up": _pack_conv3(_compose_up_weights(b1.conv0.effective_weight(), self.resample_filter.detach().cpu()), 4).to(dev),
            "w_b1c1": _pack_conv3(b1.conv1.effective_weight(), 2).to(dev),
            "w_up_poly": _pack_up_poly(b1.conv0.effective_weight()).to(dev),
            "bias": [l.bias.detach().float().contiguous() for l in (b0.conv0, b0.conv1, b1.conv0, b1.conv1)],
            "rgb0_w": b0.torgb.effective_weight().float().contiguous(), "rgb0_b": b0.torgb.bias.detach().float().contiguous(),
            "rgb1_w": b1.torgb.effective_weight().float().contiguous(), "rgb1_b": b1.torgb.bias.detach().float().contiguous(),
        }
        m = SrModel()
        for k in ("w_first", "w_b0c1", "w_up", "w_b1c1", "w_up_poly", "rgb0_w", "rgb0_b", "rgb1_w", "rgb1_b"):
            setattr(m, k, keep[k].data_ptr())
        for i, l in enumerate((b0.conv0, b0.conv1, b1.conv0, b1.conv1)):
            m.bias[i] = keep["bias"][i].data_ptr()
            m.noise_strength[i] = float(l.noise_strength)
        f2 = self.resample_filter.detach().double().cpu()
        f1 = f2.sum(dim=0)                                # the filter is the outer product of its marginals (setup_filter, upfirdn2d.py:106)
        if not torch.allclose(torch.outer(f1, f1), f2, atol=1e-7):
            raise GfppError("Superresolution: resample_filter must be separable")
        for i in range(4):
            m.fir[i] = float(f1[i] * 2.0)                 # gain up^2 = 4 -> 2 per axis
        keep["up_fir_g"] = _fir_gemm_table([float(f1[i] * 2.0) for i in range(4)]).to(dev)
        m.up_fir_g = keep["up_fir_g"].data_ptr()
        m.conv_clamp = float(self.conv_clamp)
        self._packed = {"fp": fp, "keep": keep, "model": m, "ws": {}}
        return self._packed

    #: which activation workspace forward() uses; frames of different lanes may be in flight on different streams (clip.ClipRenderer)
    lane = 0
    #: set by clip.ClipRenderer around a frame (group): (device pointer of the gfpp_clip_job, lane, frames of the group, lanes).  A forward(clip_sub=k) then leaves frame
    #: k of the group as uint8 in the job's output slot from inside the last layer's epilogue (no fp32 image, no store launch) and sets `clip_consumed`
    clip_store = None
    clip_consumed = False

    def _workspace(self, P):
        ent = P["ws"].get(self.lane)
        if ent is None:
            dev = self.resample_filter.device
            bufs = {"x0": torch.empty(256, 256, 128, dtype=torch.float16, device=dev), "x1": torch.empty(256, 256, 128, dtype=torch.float16, device=dev),
                    "x2": torch.empty(512, 512, 64, dtype=torch.float16, device=dev), "img256": torch.empty(256, 256, 3, dtype=torch.float32, device=dev)}
            ws = SrWs()
            for k, v in bufs.items():
                setattr(ws, k, v.data_ptr())
            # noise_mode 'random' is drawn inside the kernels: frame counter + ticket word of this lane, key = torch's seed mixed with the lane
            # [frame counter, ticket, seed word]: the key is rng_seed (a launch argument, frozen in captured graphs) XOR the seed word (device memory)
            bufs["rng_state"] = torch.zeros(3, dtype=torch.int64, device=dev)
            bufs["rng_seed"] = self._lane_seed(torch.initial_seed(), self.lane)
            if getattr(self, "_reseed", None) is not None:
                bufs["rng_state"][2] = _as_i64(bufs["rng_seed"] ^ self._lane_seed(self._reseed, self.lane))
            ent = P["ws"][self.lane] = (ws, bufs)
        return ent

    @staticmethod
    def _lane_seed(seed, lane):
        return (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019 * (int(lane) + 1)) & 0xFFFFFFFFFFFFFFFF

    def reseed(self, seed):
        """Restart the in-kernel noise of every lane from `seed` (frame counter 0): the same seed gives the same frames -- also for launches that are
        frozen in a captured graph (the seed WORD lives in device memory; the `rng_seed` launch argument a graph bakes in stays what it was, the key is
        their XOR).  NB: the 'random' noise of the HIP path is NOT governed by torch.manual_seed (torch's generator is never touched: nothing for a graph
        replay to re-seed); it starts from torch.initial_seed() at the time a lane's workspace is created, and from `seed` after reseed(seed).
        Call it outside a graph capture (it writes device memory)."""
        if self._packed is not None:
            for lane, (ws, bufs) in self._packed["ws"].items():
                state = torch.tensor([0, 0, _as_i64(bufs["rng_seed"] ^ self._lane_seed(seed, lane))], dtype=torch.int64)
                bufs["rng_state"].copy_(state.to(bufs["rng_state"].device))
        self._reseed = int(seed)

    # -- training: the same network in autograd-visible torch ops (the convolutions go to MIOpen) ----------------------------------------
    def _forward_autograd(self, rgb, noise_mode):
        """radnerf_sr.py:30-43 over superresolution.py:219-245 (block0, no up-sampling) and networks_stylegan2.py:446-472 (block1, x2),
        architecture 'skip', ws = ones.  Used in training mode only; inference runs the folded-weight HIP kernels."""
        layers = (self.block0.conv0, self.block0.conv1, self.block1.conv0, self.block1.conv1)
        if noise_mode == "const":
            noises = [l.noise_const for l in layers]
        elif noise_mode == "random":
            noises = [torch.randn(l.resolution, l.resolution, device=rgb.device) for l in layers]
        else:
            noises = [None] * 4
        c = self.conv_clamp
        x = self.block0.conv0.forward_autograd(rgb, noises[0], c)
        x = self.block0.conv1.forward_autograd(x, noises[1], c)
        img = rgb + self.block0.torgb.forward_autograd(x, c)
        x = self.block1.conv0.forward_autograd(x, noises[2], c)
        x = self.block1.conv1.forward_autograd(x, noises[3], c)
        # upfirdn2d.upsample2d of the running image (upfirdn2d.py:330-355): zero-stuff x2, pad (2, 1), 4x4 FIR with gain 4
        f = self.resample_filter.to(img.dtype)
        B, C, H, W = img.shape
        up = torch.zeros(B, C, H, 2, W, 2, dtype=img.dtype, device=img.device)
        up[:, :, :, 0, :, 0] = img
        up = F.pad(up.reshape(B, C, 2 * H, 2 * W), [2, 1, 2, 1])
        up = F.conv2d(up, (f * 4.0).flip([0, 1])[None, None].repeat(C, 1, 1, 1), groups=C)
        return up + self.block1.torgb.forward_autograd(x, c)

    # -- forward ------------------------------------------------------------------------------------------------------------------
    def forward(self, rgb, noise_mode="random", clamp01=False, clip_sub=None, **block_kwargs):
        """rgb [1,3,256,256] in [0,1] -> [1,3,512,512] fp32 (radnerf_sr.py:30-43).  noise_mode: 'random' (the reference's default: a fresh
        unit normal field per layer, scaled by the learned noise_strength), 'const' (the stored noise_const buffers) or 'none'.
        clamp01 (not a reference argument): clamp the result to [0, 1] inside the last kernel -- what the callers do next (radnerf_torso_sr.py:221,231)."""
        assert noise_mode in ("random", "const", "none")
        if rgb.dim() != 4 or rgb.shape[0] != 1 or rgb.shape[1] != 3:
            raise GfppError(f"Superresolution: expected rgb [1,3,H,W], got {tuple(rgb.shape)}")
        if not rgb.is_cuda:
            raise GfppError("Superresolution: input must be on the GPU (there is no CPU path)")
        if rgb.shape[-1] < self.input_resolution:
            rgb = F.interpolate(rgb, size=(self.input_resolution, self.input_resolution), mode="bilinear", align_corners=False,
                                antialias=self.sr_antialias)
        if rgb.shape[-1] != self.input_resolution or rgb.shape[-2] != self.input_resolution:
            raise GfppError("Superresolution: input must be 256x256 (or smaller, then it is interpolated up like in the reference)")
        if torch.is_grad_enabled() and (rgb.requires_grad or any(p.requires_grad for p in self.parameters())) and self.training:
            out = self._forward_autograd(rgb.float(), noise_mode)
            return out.clamp(0, 1) if clamp01 else out
        P = self._pack()
        x = rgb.detach().float().permute(0, 2, 3, 1).contiguous()               # NHWC view of the NeRF image: no copy when it came from render()
        layers = (self.block0.conv0, self.block0.conv1, self.block1.conv0, self.block1.conv1)
        if noise_mode == "const":
            noises = [l.noise_const.detach().float().contiguous() for l in layers]
        else:
            # 'random': the reference draws a fresh unit-normal field per layer and frame with torch.randn (networks_stylegan2.py:329-331); here the
            # kernels draw it themselves (Philox4x32-10 keyed by torch's seed, counter = pixel / layer / frame) -- no generator launch, no 2.6 MB of
            # noise written and read back, and nothing in a captured graph that torch would have to re-seed per replay
            noises = None
        arr = (c_p * 4)(*[n.data_ptr() for n in noises]) if noises is not None else None
        ws, bufs = self._workspace(P)
        ws.rng_state = bufs["rng_state"].data_ptr() if noise_mode == "random" else None
        ws.rng_seed = bufs["rng_seed"]
        ws.clamp01 = 1 if clamp01 else 0
        ws.clip_job = None
        ws.up_prof = self.up_prof.data_ptr() if getattr(self, "up_prof", None) is not None else None      # (tools/sr_up_phases.py)
        if self.clip_store is not None and clip_sub is not None and clamp01:
            from .. import tuning
            if tuning.LIB["sr_final_resident"]:
                job, lane, frames, lanes = self.clip_store
                ws.clip_job, ws.clip_lane, ws.clip_sub = job, lane, int(clip_sub)
                ws.clip_advance = (frames * lanes if frames > 1 else 0) if clip_sub == frames - 1 else 0xFFFFFFFF
                self.clip_consumed = True
        out = torch.empty(512, 512, 3, dtype=torch.float32, device=x.device)
        call("gfpp_sr_forward", ctypes.byref(P["model"]), ctypes.byref(ws), x.data_ptr(), arr, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out.permute(2, 0, 1).unsqueeze(0)                                # [1,3,512,512] view
