"""Input encodings of the radiance fields: multiresolution grid, spherical harmonics, frequency.

Same constructor arguments, attributes, parameter/buffer names (``embeddings``, ``offsets``) and output layouts as the
reference's encoders/gridencoder/grid.py:96-164, encoders/shencoder/sphere_harmonics.py:61-87,
encoders/freqencoder/freq.py:54-78 and the factory encoders/encoding.py:6-35 -- so reference checkpoints load
unchanged -- but forward() runs our HIP kernels through the C ABI.  The grid encoder is differentiable (table and input gradients,
total-variation gradient); SH and frequency encodings are forward only (their inputs never require grad on the radnerfs paths).
"""
import numpy as np
import torch
import torch.nn as nn

from .._lib import call, GfppError

_GRIDTYPE = {"hash": 0, "tiled": 1}
_INTERP = {"linear": 0, "smoothstep": 1}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def grid_encode_raw(inputs01, embeddings, offsets, per_level_scale, base_resolution, gridtype_id, align_corners, interp_id):
    """Level-major lookup: inputs01 [B,D] f32 in [0,1] -> [L,B,C] in the table's dtype."""
    if not inputs01.is_cuda:
        raise GfppError("grid_encode: inputs must be on the GPU (no CPU path)")
    inputs01 = inputs01.float().contiguous()
    B, D = inputs01.shape
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    if embeddings.dtype == torch.float32:
        dtype = 0
    elif embeddings.dtype == torch.float16:
        dtype = 1
    else:
        raise GfppError(f"grid tables must be float32 or float16, got {embeddings.dtype}")
    out = torch.empty(L, B, C, device=inputs01.device, dtype=embeddings.dtype)
    S = float(np.log2(per_level_scale))
    call("gfpp_grid_encode_forward", inputs01.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(), out.data_ptr(), B, D, C, L, S,
         int(base_resolution), None, int(gridtype_id), int(bool(align_corners)), int(interp_id), dtype, _stream())
    return out


class _GridEncodeFn(torch.autograd.Function):
    """Differentiable lookup (reference: _grid_encode, grid.py:24-94): gradients w.r.t. the table always, w.r.t. the inputs when they
    require grad.  The reference's forward then also writes dy_dx [B, L*D*C] for the backward pass; with level_dim 2 the derivative is recomputed
    from the table there instead (gfpp_grid_encode_input_backward: the same 2^D corners, no 116 MB intermediate), other level_dims keep dy_dx.

    Under autocast with an even level_dim the reference casts the table to half for the call (grid.py:41-44) -- features, their gradient and the
    table gradient's accumulators are then half -- and that is the reference's training configuration (`amp: true`).  Same here (`half=True`): the
    half lookup kernel forward, a half grad into gfpp_grid_encode_backward_f16 backward; accumulation and the parameter's gradient are fp32 (more
    accurate than the reference's half atomics, same interface), and so is the input gradient."""

    @staticmethod
    def forward(ctx, inputs01, embeddings, offsets, per_level_scale, base_resolution, gridtype_id, align_corners, interp_id):
        inputs01 = inputs01.float().contiguous()
        emb = embeddings.float().contiguous()
        B, D = inputs01.shape
        L, C = offsets.shape[0] - 1, emb.shape[1]
        S = float(np.log2(per_level_scale))
        half = bool(torch.is_autocast_enabled()) and C == 2
        out = torch.empty(L, B, C, device=inputs01.device, dtype=torch.half if half else torch.float32)
        # decide from autograd's own bookkeeping: `inputs01` may be a fresh no-grad copy after the cast above (a non-fp32 or non-contiguous
        # input), whose requires_grad flag says nothing about the caller's tensor
        want_dx = bool(ctx.needs_input_grad[0])
        # level_dim 2: the input gradient is recomputed from the table in the backward pass (gfpp_grid_encode_input_backward); otherwise dy_dx is kept
        dy_dx = torch.empty(B, L * D * C, device=inputs01.device, dtype=torch.float32) if want_dx and C != 2 else None
        if half:
            call("gfpp_grid_encode_forward", inputs01.data_ptr(), emb.to(torch.half).data_ptr(), offsets.data_ptr(), out.data_ptr(), B, D, C, L, S,
                 int(base_resolution), None, int(gridtype_id), int(bool(align_corners)), int(interp_id), 1, _stream())
        else:
            call("gfpp_grid_encode_forward", inputs01.data_ptr(), emb.data_ptr(), offsets.data_ptr(), out.data_ptr(), B, D, C, L, S, int(base_resolution),
                 dy_dx.data_ptr() if dy_dx is not None else None, int(gridtype_id), int(bool(align_corners)), int(interp_id), 0, _stream())
        ctx.save_for_backward(inputs01, offsets, dy_dx, emb if want_dx and C == 2 else None)
        ctx.dims = (B, D, C, L, S, int(base_resolution), int(gridtype_id), int(bool(align_corners)), int(interp_id), int(emb.shape[0]), half, want_dx)
        return out

    @staticmethod
    def backward(ctx, grad):
        inputs01, offsets, dy_dx, emb = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, ac, interp, rows, half, want_dx = ctx.dims
        grad_emb = torch.zeros(rows, C, device=grad.device, dtype=torch.float32)
        grad_inputs = torch.zeros_like(inputs01) if want_dx else None
        # eight scratch copies of the table gradient (gfpp_grid_encode_backward_xcd: LDS accumulation per level range, no device atomic per corner)
        copies = torch.empty(8 * rows * C + 64, device=grad.device, dtype=torch.float32)     # (+ 64 words: the levels' gradient maxima)
        bins, bins_bytes = table_gradient_bins(rows, C, L, B, grad.device)                   # per-range point lists (round 6)
        grad = grad.to(torch.half if half else torch.float32).contiguous()
        call("gfpp_grid_encode_backward_f16" if half else "gfpp_grid_encode_backward_xcd", grad.data_ptr(), inputs01.data_ptr(), offsets.data_ptr(),
             grad_emb.data_ptr(), rows, copies.data_ptr(), B, D, C, L, S, H, dy_dx.data_ptr() if dy_dx is not None else None,
             grad_inputs.data_ptr() if dy_dx is not None else None, gridtype, ac, interp, _stream(), bins.data_ptr() if bins is not None else None, bins_bytes)
        if want_dx and dy_dx is None:
            call("gfpp_grid_encode_input_backward", grad.data_ptr(), 1 if half else 0, inputs01.data_ptr(), emb.data_ptr(), offsets.data_ptr(), grad_inputs.data_ptr(),
                 B, D, C, L, S, H, gridtype, ac, interp, _stream())
        return grad_inputs, grad_emb, None, None, None, None, None, None


def table_gradient_bins(rows, C, L, B, device):
    """The `bins` scratch of gfpp_grid_encode_backward_xcd / _f16 (per-range point lists: one uint32 per point and range of every level), or (None, 0) when the
    lists would not pay or not fit: below ~32 k points a range's walk over all of them is short anyway, and the lists of a table with very many ranges
    (a 2^19-row hash grid: 1 000+ ranges) would take gigabytes."""
    from .._lib import lib
    if B < 32768:
        return None, 0
    n = int(lib().gfpp_grid_backward_bins_bytes(int(rows), int(C), int(L), int(B)))
    if n > (1 << 30):
        return None, 0
    return torch.empty(n // 4, dtype=torch.int32, device=device), n


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _GRIDTYPE[gridtype]
        self.interpolation = interpolation
        self.interp_id = _INTERP[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        rows = []
        for lvl in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale ** lvl))
            n = min(self.max_params, (res if align_corners else res + 1) ** input_dim)
            rows.append(int(np.ceil(n / 8) * 8))          # level sizes are rounded up to multiples of 8
        offsets = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder(hip): input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base={self.base_resolution} per_level_scale={self.per_level_scale:.4f} rows={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} interpolation={self.interpolation}")

    def table(self):
        """Table in the dtype the lookup will use: half under autocast when level_dim is even (grid.py:43-44)."""
        emb = self.embeddings
        if torch.is_autocast_enabled() and self.level_dim % 2 == 0:
            emb = emb.to(torch.half)
        return emb

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        if torch.is_grad_enabled() and (self.embeddings.requires_grad or flat.requires_grad):
            out = _GridEncodeFn.apply(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, self.gridtype_id,
                                      self.align_corners, self.interp_id)
        else:
            out = grid_encode_raw(flat, self.table(), self.offsets, self.per_level_scale, self.base_resolution, self.gridtype_id,
                                  self.align_corners, self.interp_id)
        B = flat.shape[0]
        return out.permute(1, 0, 2).reshape(B, self.output_dim).view(prefix + [self.output_dim])

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """Adds the total-variation gradient of the cells visited by `inputs` (or B random points) to embeddings.grad (grid.py:166-189)."""
        D, C, L = self.input_dim, self.embeddings.shape[1], self.offsets.shape[0] - 1
        if inputs is None:
            inputs = torch.rand(B, D, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).reshape(-1, D)
        inputs = inputs.float().contiguous()
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        emb = self.embeddings.detach().float().contiguous()
        call("gfpp_grad_total_variation", inputs.data_ptr(), emb.data_ptr(), self.embeddings.grad.data_ptr(), self.offsets.data_ptr(), float(weight),
             inputs.shape[0], D, C, L, float(np.log2(self.per_level_scale)), int(self.base_resolution), self.gridtype_id, int(bool(self.align_corners)),
             _stream())


class _SHEncodeFn(torch.autograd.Function):
    """sphere_harmonics.py:14-58: forward keeps dy_dx [B, 3 * degree^2] when the directions need a gradient."""

    @staticmethod
    def forward(ctx, flat, degree):
        B = flat.shape[0]
        out = torch.empty(B, degree ** 2, dtype=torch.float32, device=flat.device)
        dy_dx = torch.empty(B, 3 * degree ** 2, dtype=torch.float32, device=flat.device) if flat.requires_grad else None
        call("gfpp_sh_encode_forward", flat.data_ptr(), out.data_ptr(), B, 3, degree, dy_dx.data_ptr() if dy_dx is not None else None, _stream())
        ctx.save_for_backward(flat, dy_dx)
        ctx.degree = degree
        return out

    @staticmethod
    def backward(ctx, grad):
        flat, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None
        grad = grad.float().contiguous()
        grad_inputs = torch.zeros_like(flat)
        call("gfpp_sh_encode_backward", grad.data_ptr(), flat.data_ptr(), flat.shape[0], 3, ctx.degree, dy_dx.data_ptr(), grad_inputs.data_ptr(), _stream())
        return grad_inputs, None


class _FreqEncodeFn(torch.autograd.Function):
    """freq.py:14-53: the backward reads sin / cos back from the forward outputs."""

    @staticmethod
    def forward(ctx, flat, degree, output_dim):
        B, D = flat.shape
        out = torch.empty(B, output_dim, dtype=torch.float32, device=flat.device)
        call("gfpp_freq_encode_forward", flat.data_ptr(), B, D, degree, output_dim, out.data_ptr(), _stream())
        ctx.save_for_backward(out)
        ctx.dims = (B, D, degree, output_dim)
        return out

    @staticmethod
    def backward(ctx, grad):
        (out,) = ctx.saved_tensors
        B, D, degree, C = ctx.dims
        grad = grad.float().contiguous()
        grad_inputs = torch.empty(B, D, dtype=torch.float32, device=grad.device)
        call("gfpp_freq_encode_backward", grad.data_ptr(), out.data_ptr(), B, D, degree, C, grad_inputs.data_ptr(), _stream())
        return grad_inputs, None, None


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3:
            raise AssertionError("SH encoder only support input dim == 3")
        if not 1 <= degree <= 4:
            raise GfppError("SH encoder: this build supports degree 1..4 (the reference allows up to 8; the render path uses 4)")
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, 3).float().contiguous()
        if not flat.is_cuda:
            raise GfppError("sh_encode: inputs must be on the GPU (no CPU path)")
        if torch.is_grad_enabled() and flat.requires_grad:
            out = _SHEncodeFn.apply(flat, self.degree)
        else:
            out = torch.empty(flat.shape[0], self.output_dim, dtype=torch.float32, device=flat.device)
            call("gfpp_sh_encode_forward", flat.data_ptr(), out.data_ptr(), flat.shape[0], 3, self.degree, None, _stream())
        return out.view(prefix + [self.output_dim])


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def forward(self, inputs, **kwargs):
        prefix = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim).float().contiguous()
        if not flat.is_cuda:
            raise GfppError("freq_encode: inputs must be on the GPU (no CPU path)")
        if torch.is_grad_enabled() and flat.requires_grad:
            out = _FreqEncodeFn.apply(flat, self.degree, self.output_dim)
        else:
            out = torch.empty(flat.shape[0], self.output_dim, dtype=torch.float32, device=flat.device)
            call("gfpp_freq_encode_forward", flat.data_ptr(), flat.shape[0], self.input_dim, self.degree, self.output_dim,
                 out.data_ptr(), _stream())
        return out.view(prefix + [self.output_dim])


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048, align_corners=False, interpolation="linear", **kwargs):
    """Factory with the reference's names (encoders/encoding.py:6-35). Returns (encoder, output_dim)."""
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        enc = FreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == "spherical_harmonics":
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                          log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                          gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners,
                          interpolation=interpolation, **kwargs)
    else:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, spherical_harmonics, hashgrid, tiledgrid]")
    return enc, enc.output_dim
