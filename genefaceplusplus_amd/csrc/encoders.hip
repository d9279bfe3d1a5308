// encoders.hip -- stand-alone encoder kernels behind the reference's `_gridencoder`, `_shencoder` and `_freqencoder`
// extension APIs (forward only).
#include "grid_device.h"
#include "sh_device.h"

namespace gfpp {

constexpr int kEncBlock = 256;

struct LevelScales {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

// grid = (ceil(B/256), L): one thread evaluates one (point, level) pair and writes C contiguous outputs of the
// level-major [L,B,C] buffer, so a wavefront's stores are one contiguous 64*C*sizeof(T) segment.
template <int D, int C, typename T>
__global__ __launch_bounds__(kEncBlock) void k_grid_encode(const float *__restrict__ inputs, const T *__restrict__ table,
                                                          const int32_t *__restrict__ offsets, T *__restrict__ outputs, uint32_t B,
                                                          LevelScales ls, uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * kEncBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float u[D];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        u[d] = inputs[(size_t)b * D + d];
        inside = inside && !(u[d] < 0.0f || u[d] > 1.0f);
    }
    float out[C];
    if (inside) {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t size = (uint32_t)offsets[level + 1] - off;
        grid_level_lookup<D, C, T>(u, table, off, size, ls.scale[level], ls.resolution[level], gridtype, align_corners, interp, out);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = 0.0f;
    }
    T *dst = outputs + ((size_t)level * B + b) * C;
    if constexpr (sizeof(T) == 4) {
        if constexpr (C == 2) *reinterpret_cast<float2 *>(dst) = make_float2(out[0], out[1]);
        else if constexpr (C == 4) *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        else {
#pragma unroll
            for (int c = 0; c < C; ++c) dst[c] = out[c];
        }
    } else {
        if constexpr (C == 2) *reinterpret_cast<__half2 *>(dst) = __floats2half2_rn(out[0], out[1]);
        else {
#pragma unroll
            for (int c = 0; c < C; ++c) dst[c] = __float2half(out[c]);
        }
    }
}

__global__ __launch_bounds__(kEncBlock) void k_sh_encode(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B, uint32_t degree) {
    const uint32_t b = blockIdx.x * kEncBlock + threadIdx.x;
    if (b >= B) return;
    float sh[16];
    sh_basis4(inputs[3ull * b], inputs[3ull * b + 1], inputs[3ull * b + 2], sh);
    const uint32_t n = degree * degree;
    float *o = outputs + (size_t)b * n;
    for (uint32_t i = 0; i < n; ++i) o[i] = sh[i];
}

// dy_dx [B, 3, degree^2]: rows d/dx, d/dy, d/dz of the features (shencoder.cu:125-352)
__global__ __launch_bounds__(kEncBlock) void k_sh_dydx(const float *__restrict__ inputs, float *__restrict__ dy_dx, uint32_t B, uint32_t degree) {
    const uint32_t b = blockIdx.x * kEncBlock + threadIdx.x;
    if (b >= B) return;
    float dx[16], dy[16], dz[16];
    sh_basis4_grad(inputs[3ull * b], inputs[3ull * b + 1], inputs[3ull * b + 2], dx, dy, dz);
    const uint32_t n = degree * degree;
    float *o = dy_dx + (size_t)b * 3 * n;
    for (uint32_t i = 0; i < n; ++i) { o[i] = dx[i]; o[n + i] = dy[i]; o[2 * n + i] = dz[i]; }
}

// grad_inputs[b, d] += sum_c grad[b, c] * dy_dx[b, d, c]   (kernel_sh_backward, shencoder.cu:359-382)
__global__ __launch_bounds__(kEncBlock) void k_sh_backward(const float *__restrict__ grad, const float *__restrict__ dy_dx, uint32_t B, uint32_t n,
                                                           float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kEncBlock + threadIdx.x;
    if (t >= B * 3u) return;
    const uint32_t b = t / 3u;
    const float *g = grad + (size_t)b * n, *j = dy_dx + (size_t)t * n;
    float acc = grad_inputs[t];
    for (uint32_t c = 0; c < n; ++c) acc = fmaf(g[c], j[c], acc);
    grad_inputs[t] = acc;
}

// grad_inputs[b, d] = grad[b, d] + sum_f 2^f (grad_sin * cos - grad_cos * sin), with sin / cos read back from the forward outputs
// (kernel_freq_backward, freqencoder.cu:63-93)
__global__ __launch_bounds__(kEncBlock) void k_freq_backward(const float *__restrict__ grad, const float *__restrict__ outputs, uint32_t B, uint32_t D,
                                                             uint32_t deg, uint32_t C, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kEncBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *g = grad + (size_t)b * C, *o = outputs + (size_t)b * C;
    float acc = g[d];
    for (uint32_t f = 0; f < deg; ++f) {
        const uint32_t s = D + 2u * D * f + d, c = s + D;
        acc += scalbnf(1.0f, (int)f) * (g[s] * o[c] - g[c] * o[s]);
    }
    grad_inputs[t] = acc;
}

// One thread per output element (coalesced stores of the [B,C] row-major output).
__global__ __launch_bounds__(kEncBlock) void k_freq_encode(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t C, float *__restrict__ outputs) {
    const size_t t = (size_t)blockIdx.x * kEncBlock + threadIdx.x;
    if (t >= (size_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (size_t)b * C);
    const float *in = inputs + (size_t)b * D;
    outputs[t] = (c < D) ? in[c] : freq_feature(in[c % D], c / D - 1);
}

template <int D, typename T>
static int launch_grid(const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t C, uint32_t L,
                       const LevelScales &ls, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st) {
    const dim3 grid(div_up(B, kEncBlock), L), block(kEncBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_encode<D, 1, T>), grid, block, 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, ls, gridtype, ac, interp); break;
        case 2: hipLaunchKernelGGL((k_grid_encode<D, 2, T>), grid, block, 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, ls, gridtype, ac, interp); break;
        case 4: hipLaunchKernelGGL((k_grid_encode<D, 4, T>), grid, block, 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, ls, gridtype, ac, interp); break;
        case 8: hipLaunchKernelGGL((k_grid_encode<D, 8, T>), grid, block, 0, st, inputs, (const T *)emb, offsets, (T *)outputs, B, ls, gridtype, ac, interp); break;
        default: set_error("gfpp_grid_encode_forward: level_dim must be 1, 2, 4 or 8 (got %u)", C); return GFPP_EUNSUPPORTED;
    }
    return check_launch("gfpp_grid_encode_forward");
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                                      int align_corners, uint32_t interp, int dtype, gfpp_stream_t stream) {
    if (B == 0 || L == 0) return 0;
    if (!inputs || !embeddings || !offsets || !outputs) { set_error("gfpp_grid_encode_forward: null pointer"); return GFPP_EINVAL; }
    if (dy_dx) {   // the optional second output of the reference's grid_encode_forward (gridencoder.cu:198-243)
        if (dtype != GFPP_F32) { set_error("gfpp_grid_encode_forward: dy_dx is available for fp32 tables only"); return GFPP_EUNSUPPORTED; }
        const int rc = gfpp_grid_encode_dydx(inputs, (const float *)embeddings, offsets, (float *)dy_dx, B, D, C, L, S, H, gridtype, align_corners, interp, stream);
        if (rc) return rc;
    }
    if (L > kMaxLevels || gridtype > 1 || interp > 1) { set_error("gfpp_grid_encode_forward: L<=32, gridtype in {0,1}, interp in {0,1}"); return GFPP_EINVAL; }
    LevelScales ls;
    for (uint32_t l = 0; l < L; ++l) {
        const float sc = fmaf(exp2f((float)l * S), (float)H, -1.0f);
        ls.scale[l] = sc;
        ls.resolution[l] = (uint32_t)ceil((double)sc) + 1u;
    }
    const hipStream_t st = (hipStream_t)stream;
    const bool ac = align_corners != 0;
    if (dtype == GFPP_F32) {
        if (D == 2) return launch_grid<2, float>(inputs, embeddings, offsets, outputs, B, C, L, ls, gridtype, ac, interp, st);
        if (D == 3) return launch_grid<3, float>(inputs, embeddings, offsets, outputs, B, C, L, ls, gridtype, ac, interp, st);
    } else if (dtype == GFPP_F16) {
        if (D == 2) return launch_grid<2, __half>(inputs, embeddings, offsets, outputs, B, C, L, ls, gridtype, ac, interp, st);
        if (D == 3) return launch_grid<3, __half>(inputs, embeddings, offsets, outputs, B, C, L, ls, gridtype, ac, interp, st);
    } else {
        set_error("gfpp_grid_encode_forward: dtype must be GFPP_F32 or GFPP_F16");
        return GFPP_EUNSUPPORTED;
    }
    set_error("gfpp_grid_encode_forward: input_dim must be 2 or 3 (got %u)", D);
    return GFPP_EUNSUPPORTED;
}

GFPP_API int gfpp_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree, float *dy_dx,
                                    gfpp_stream_t stream) {
    if (B == 0) return 0;
    if (!inputs || !outputs) { set_error("gfpp_sh_encode_forward: null pointer"); return GFPP_EINVAL; }
    if (D != 3 || degree < 1 || degree > 4) { set_error("gfpp_sh_encode_forward: needs D=3, 1<=degree<=4"); return GFPP_EUNSUPPORTED; }
    hipLaunchKernelGGL(k_sh_encode, dim3(div_up(B, kEncBlock)), dim3(kEncBlock), 0, (hipStream_t)stream, inputs, outputs, B, degree);
    if (dy_dx) hipLaunchKernelGGL(k_sh_dydx, dim3(div_up(B, kEncBlock)), dim3(kEncBlock), 0, (hipStream_t)stream, inputs, dy_dx, B, degree);
    return check_launch("gfpp_sh_encode_forward");
}

GFPP_API int gfpp_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree, const float *dy_dx,
                                     float *grad_inputs, gfpp_stream_t stream) {
    (void)inputs;
    if (B == 0) return 0;
    if (!grad || !dy_dx || !grad_inputs) { set_error("gfpp_sh_encode_backward: null pointer"); return GFPP_EINVAL; }
    if (D != 3 || degree < 1 || degree > 4) { set_error("gfpp_sh_encode_backward: needs D=3, 1<=degree<=4"); return GFPP_EUNSUPPORTED; }
    hipLaunchKernelGGL(k_sh_backward, dim3(div_up(B * 3u, kEncBlock)), dim3(kEncBlock), 0, (hipStream_t)stream, grad, dy_dx, B, degree * degree, grad_inputs);
    return check_launch("gfpp_sh_encode_backward");
}

GFPP_API int gfpp_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs,
                                      gfpp_stream_t stream) {
    if (B == 0) return 0;
    if (!inputs || !outputs) { set_error("gfpp_freq_encode_forward: null pointer"); return GFPP_EINVAL; }
    if (C != D + 2 * D * deg || D == 0) { set_error("gfpp_freq_encode_forward: C must equal D + 2*D*deg"); return GFPP_EINVAL; }
    const size_t total = (size_t)B * C;
    hipLaunchKernelGGL(k_freq_encode, dim3((uint32_t)((total + kEncBlock - 1) / kEncBlock)), dim3(kEncBlock), 0, (hipStream_t)stream, inputs, B, D, C, outputs);
    return check_launch("gfpp_freq_encode_forward");
}

GFPP_API int gfpp_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *grad_inputs,
                                       gfpp_stream_t stream) {
    if (B == 0) return 0;
    if (!grad || !outputs || !grad_inputs) { set_error("gfpp_freq_encode_backward: null pointer"); return GFPP_EINVAL; }
    if (C != D + 2 * D * deg || D == 0) { set_error("gfpp_freq_encode_backward: C must equal D + 2*D*deg"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_freq_backward, dim3(div_up(B * D, kEncBlock)), dim3(kEncBlock), 0, (hipStream_t)stream, grad, outputs, B, D, deg, C, grad_inputs);
    return check_launch("gfpp_freq_encode_backward");
}
