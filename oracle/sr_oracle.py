"""CPU restatement of the super-resolution stage of the *_sr models -- TEST INFRASTRUCTURE, never imported by the product.

Follows, with file:line citations into /root/reference:
  modules/radnerfs/radnerf_sr.py:14-43                      Superresolution (two blocks, ws = ones, 256 -> 512)
  modules/eg3ds/models/superresolution.py:159-258           SynthesisBlockNoUp
  modules/eg3ds/models/networks_stylegan2.py:37-94          modulated_conv2d
  modules/eg3ds/models/networks_stylegan2.py:99-133         FullyConnectedLayer (the style affine)
  modules/eg3ds/models/networks_stylegan2.py:286-344        SynthesisLayer
  modules/eg3ds/models/networks_stylegan2.py:349-371        ToRGBLayer
  modules/eg3ds/models/networks_stylegan2.py:375-478        SynthesisBlock
  modules/eg3ds/torch_utils/ops/conv2d_resample.py:47-147   conv2d_resample (up = 2 path: transposed conv + FIR)
  modules/eg3ds/torch_utils/ops/upfirdn2d.py:169-217        _upfirdn2d_ref;  :330-355 upsample2d
  modules/eg3ds/torch_utils/ops/bias_act.py:95-125          _bias_act_ref (lrelu alpha 0.2, gain sqrt 2, clamp)
The convolutions themselves are torch's CPU conv2d / conv_transpose2d in fp32 (the reference calls cuDNN through the same torch
functions; SURVEY.md 8c: "their CPU counterparts in torch are the oracle for those ops").  The reference runs both blocks in fp16 on
the GPU (use_fp16=True) and in fp32 on the CPU (force_fp32); this restatement is the fp32 one.
Pinned against the reference's own modules by tests/golden/make_golden_sr.py -> tests/golden/sr_golden.npz."""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def styles_from_ones(affine_w, affine_b, w_dim=16):
    """FullyConnectedLayer(w_dim, in, bias_init=1) applied to ws = ones (networks_stylegan2.py:115-128, radnerf_sr.py:32-33)."""
    w = _t(affine_w) * (1.0 / np.sqrt(w_dim))
    return torch.addmm(_t(affine_b).unsqueeze(0), torch.ones(1, w_dim), w.t())[0]          # [in]


def upfirdn2d_ref(x, f, up=1, padding=(0, 0, 0, 0), gain=1.0):
    """_upfirdn2d_ref with down = 1, flip_filter = False (upfirdn2d.py:169-217).  x [B,C,H,W], f [fh,fw]."""
    B, C, H, W = x.shape
    px0, px1, py0, py1 = padding
    x = x.reshape(B, C, H, 1, W, 1)
    x = F.pad(x, [0, up - 1, 0, 0, 0, up - 1])
    x = x.reshape(B, C, H * up, W * up)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    f = (f * gain).flip([0, 1])
    return F.conv2d(x, f[None, None].repeat(C, 1, 1, 1), groups=C)


def upsample2d(x, f, up=2):
    """upfirdn2d.upsample2d (upfirdn2d.py:330-355)."""
    fh, fw = f.shape
    p = [(fw + up - 1) // 2, (fw - up) // 2, (fh + up - 1) // 2, (fh - up) // 2]
    return upfirdn2d_ref(x, f, up=up, padding=p, gain=up * up)


def conv2d_resample(x, w, f, up, padding):
    """conv2d_resample for the two cases the SR net uses (conv2d_resample.py:47-147): up == 1 -> plain correlation with symmetric
    padding; up == 2 (flip_weight False) -> transposed strided convolution followed by the FIR filter."""
    if up == 1:
        return F.conv2d(x, w, padding=padding)
    kh, kw = w.shape[2:]
    fh, fw = f.shape
    px0 = px1 = py0 = py1 = padding
    px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2; py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    wt = w.transpose(0, 1)
    px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
    pxt = max(min(-px0, -px1), 0); pyt = max(min(-py0, -py1), 0)
    # _conv2d_wrapper(..., transpose=True, flip_weight=(not False)) -> no flip
    x = F.conv_transpose2d(x, wt, stride=up, padding=[pyt, pxt])
    return upfirdn2d_ref(x, f, padding=(px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt), gain=up ** 2)


def modulated_conv2d(x, weight, styles, noise, up, padding, f, demodulate):
    """fused_modconv path with batch 1 (networks_stylegan2.py:37-94); fp32, so the fp16 pre-normalisation branch is not taken."""
    w = _t(weight) * styles.reshape(1, -1, 1, 1)
    if demodulate:
        d = (w.square().sum(dim=[1, 2, 3]) + 1e-8).rsqrt()
        w = w * d.reshape(-1, 1, 1, 1)
    x = conv2d_resample(x, w, f, up, padding)
    if noise is not None:
        x = x + noise
    return x


def bias_act(x, b, act, gain=None, clamp=None):
    """_bias_act_ref (bias_act.py:95-125)."""
    x = x + _t(b).reshape(1, -1, 1, 1)
    if act == "lrelu":
        x = F.leaky_relu(x, 0.2)
        g = np.sqrt(2) if gain is None else gain
        x = x * float(g)
    if clamp is not None:
        x = x.clamp(-clamp, clamp)
    return x


def synthesis_layer(x, sd, p, up, noise_mode, f, conv_clamp=256.0, noise_random=None):
    """SynthesisLayer.forward (networks_stylegan2.py:321-344).  noise_mode 'const' | 'none' | 'random' (then noise_random [res,res]
    is the unit normal field the caller drew)."""
    styles = styles_from_ones(sd[p + ".affine.weight"], sd[p + ".affine.bias"])
    noise = None
    if noise_mode == "const":
        noise = _t(sd[p + ".noise_const"]) * float(sd[p + ".noise_strength"])
    elif noise_mode == "random":
        noise = _t(noise_random) * float(sd[p + ".noise_strength"])
    x = modulated_conv2d(x, sd[p + ".weight"], styles, noise, up, 1, f, True)
    return bias_act(x, sd[p + ".bias"], "lrelu", clamp=conv_clamp)


def to_rgb(x, sd, p, conv_clamp=256.0):
    """ToRGBLayer.forward (networks_stylegan2.py:363-368): no demodulation, weight_gain on the styles, linear activation."""
    w = sd[p + ".weight"]
    styles = styles_from_ones(sd[p + ".affine.weight"], sd[p + ".affine.bias"]) * (1.0 / np.sqrt(w.shape[1] * w.shape[2] * w.shape[3]))
    x = modulated_conv2d(x, w, styles, None, 1, 0, None, False)
    return bias_act(x, sd[p + ".bias"], "linear", clamp=conv_clamp)


def superresolution(rgb, sd, prefix="sr_net.", noise_mode="const", noise_random=None):
    """Superresolution.forward (radnerf_sr.py:30-43) for a [1,3,256,256] input in [0,1] -> [1,3,512,512] float32 (numpy in / out).
    noise_random: dict layer name -> [res,res] unit normal field, for noise_mode == 'random'."""
    with torch.no_grad():
        f = _t(sd[prefix + "resample_filter"])
        img = _t(rgb).clone()
        x = img
        nr = noise_random or {}
        # block0: SynthesisBlockNoUp(3 -> 128 @ 256), architecture 'skip' (superresolution.py:219-245)
        x = synthesis_layer(x, sd, prefix + "block0.conv0", 1, noise_mode, f, noise_random=nr.get("block0.conv0"))
        x = synthesis_layer(x, sd, prefix + "block0.conv1", 1, noise_mode, f, noise_random=nr.get("block0.conv1"))
        img = img + to_rgb(x, sd, prefix + "block0.torgb")
        # block1: SynthesisBlock(128 -> 64 @ 512) (networks_stylegan2.py:446-472)
        x = synthesis_layer(x, sd, prefix + "block1.conv0", 2, noise_mode, f, noise_random=nr.get("block1.conv0"))
        x = synthesis_layer(x, sd, prefix + "block1.conv1", 1, noise_mode, f, noise_random=nr.get("block1.conv1"))
        img = upsample2d(img, f) + to_rgb(x, sd, prefix + "block1.torgb")
        return img.numpy()
