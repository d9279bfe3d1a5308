// cond_nets.hip -- RADNeRF.cal_cond_feat (radnerf.py:88-106) as ONE launch of one workgroup.
//
// The reference runs AudioNet (cond_encoder.py:98-143: four k=3 Conv1d + LeakyReLU over the t-window, two Linear), the blink
// branch (radnerf.py:97-103) and AudioAttNet (cond_encoder.py:146-180: five k=3 Conv1d over the smoothing window, Linear + softmax,
// weighted sum) as ~70 PyTorch launches on a [smo, t_window, c_in] window -- ~0.1 MFLOP, i.e. pure launch latency (0.6 ms of
// host time per frame).  Here every layer is a set of lane-split dot products inside one 1024-thread workgroup, activations ping-pong
// between two LDS buffers, and the 64 output values land in device memory where gfpp_head_frame_begin reads them: no host round
// trip.  fp32, fmaf chains; the summation order differs from a sequential dot product (lane-split, four chains per lane).
#include <hip/hip_runtime.h>

#include "gfpp_common.h"

namespace gfpp {

constexpr int kCondThreads = 1024;
constexpr int kCondLds = 40000;  // floats of LDS: two activation buffers of `widest` floats each, then (if it fits) the whole weight blob

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.02f * v; }

// Lanes per output element: the largest power of two <= min(64, threads / outputs), so that a layer with few outputs (all of them
// here) still uses the whole workgroup: each output's dot product is split over G adjacent lanes and reduced with shuffles.
__device__ __forceinline__ uint32_t lanes_per_output(uint32_t n_out) {
    uint32_t g = 64;
    while (g > 1 && g * n_out > (uint32_t)kCondThreads) g >>= 1;
    return g;
}

__device__ __forceinline__ float group_sum(float s, uint32_t G) {
    for (uint32_t off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, (int)off);
    return s;
}

// out[b][co][t] = act(bias[co] + sum_ci sum_k w[co][ci][k] * in[b][ci][t*stride + k - 1])   (zero padding 1), b < B
// `in` is [B][Cin][Lin] when in_channels_last == false, or [B][Lin][Cin] (the layout cond arrives in) when true.
__device__ void conv1d_k3(const float *__restrict__ in, bool in_channels_last, const float *__restrict__ w, const float *__restrict__ bias,
                          float *__restrict__ out, uint32_t B, uint32_t Cin, uint32_t Cout, uint32_t Lin, uint32_t Lout, uint32_t stride, bool act) {
    const uint32_t total = B * Cout * Lout, G = lanes_per_output(total), per_pass = (uint32_t)kCondThreads / G;
    const uint32_t sub = threadIdx.x % G, slot = threadIdx.x / G;
    for (uint32_t first = 0; first < total; first += per_pass) {
        const uint32_t idx = first + slot;
        float s = 0.0f;
        uint32_t co = 0;
        if (idx < total) {
            const uint32_t t = idx % Lout, b = idx / (Lout * Cout);
            co = (idx / Lout) % Cout;
            // taps that fall inside the input: k in [k_lo, k_hi)
            const int t0 = (int)(t * stride) - 1;
            const int k_lo = t0 < 0 ? -t0 : 0, k_hi = (int)Lin - t0 < 3 ? (int)Lin - t0 : 3;
            const float *wc = w + (size_t)co * Cin * 3;
            // input element (ci, position p): base + ci * cs + p * ps
            const float *xb = in + (size_t)b * Cin * Lin;
            const uint32_t cs = in_channels_last ? 1u : Lin, ps = in_channels_last ? Cin : 1u;
            float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // four independent chains: the loads of four channels are in flight together
            uint32_t ci = sub;
            for (; ci + 3u * G < Cin; ci += 4u * G) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t c = ci + (uint32_t)q * G;
                    for (int k = k_lo; k < k_hi; ++k) s4[q] = fmaf(wc[c * 3 + (uint32_t)k], xb[c * cs + (uint32_t)(t0 + k) * ps], s4[q]);
                }
            }
            for (; ci < Cin; ci += G)
                for (int k = k_lo; k < k_hi; ++k) s4[0] = fmaf(wc[ci * 3 + (uint32_t)k], xb[ci * cs + (uint32_t)(t0 + k) * ps], s4[0]);
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        s = group_sum(s, G);
        if (idx < total && sub == 0) {
            s += bias ? bias[co] : 0.0f;
            out[idx] = act ? leaky(s) : s;
        }
    }
}

// out[b][o] = act(bias[o] + sum_i w[o][i] * in[b][i])
__device__ void linear(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out, uint32_t B,
                       uint32_t In, uint32_t Out, bool act) {
    const uint32_t total = B * Out, G = lanes_per_output(total), per_pass = (uint32_t)kCondThreads / G;
    const uint32_t sub = threadIdx.x % G, slot = threadIdx.x / G;
    for (uint32_t first = 0; first < total; first += per_pass) {
        const uint32_t idx = first + slot;
        float s = 0.0f;
        uint32_t o = 0;
        if (idx < total) {
            o = idx % Out;
            const uint32_t b = idx / Out;
            const float *wr = w + (size_t)o * In, *xr = in + (size_t)b * In;
            float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            uint32_t i = sub;
            for (; i + 3u * G < In; i += 4u * G) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s4[q] = fmaf(wr[i + (uint32_t)q * G], xr[i + (uint32_t)q * G], s4[q]);
            }
            for (; i < In; i += G) s4[0] = fmaf(wr[i], xr[i], s4[0]);
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        s = group_sum(s, G);
        if (idx < total && sub == 0) {
            s += bias ? bias[o] : 0.0f;
            out[idx] = act ? leaky(s) : s;
        }
    }
}

// (one workgroup per window: gfpp_cond_feat_batch launches `count` of them, the strides say where window / eye value / result of workgroup k are)
__global__ __launch_bounds__(kCondThreads) void k_cond_feat(gfpp_cond_model m, const float *__restrict__ cond, const float *__restrict__ eye_area,
                                                           float *__restrict__ cond_feat, uint32_t widest, uint32_t cond_stride, uint32_t eye_stride,
                                                           uint32_t out_stride) {
    cond += (size_t)blockIdx.x * cond_stride;
    if (eye_area) eye_area += (size_t)blockIdx.x * eye_stride;
    cond_feat += (size_t)blockIdx.x * out_stride;
    __shared__ float lds[kCondLds];
    __shared__ float small[64];
    float *const buf0 = lds, *const buf1 = lds + widest;
    float *buf[2] = {buf0, buf1};
    float *staged = nullptr;
    // Every layer is a short dependent step, so a weight fetch from L2 per layer (~1-2 us each, 16 layers) would be the whole run time:
    // when all weights are one contiguous blob that fits, pull it into LDS with one burst of independent loads and run from there.
    if (m.blob && 2u * widest + m.blob_floats <= (uint32_t)kCondLds) {
        float *lw = lds + 2u * widest;
        const float4 *src4 = reinterpret_cast<const float4 *>(m.blob);
        float4 *dst4 = reinterpret_cast<float4 *>(lw);
        for (uint32_t i = threadIdx.x; i < m.blob_floats / 4u; i += kCondThreads) dst4[i] = src4[i];
        __syncthreads();
        staged = lw;
    }
    // weight pointer -> where to read it: the LDS copy if staged (kernel-argument pointers are global-address-space to the compiler,
    // so the LDS alias must be formed from the LDS base, not by offsetting the argument)
    auto W = [&](const float *p) -> const float * { return staged ? static_cast<const float *>(staged + (p - m.blob)) : p; };
    const uint32_t B = m.smo;
    // ---- AudioNet: conv stack over the t-window ---------------------------------------------------------------------
    const uint32_t ch[5] = {m.c_in, 32u, 32u, 64u, 64u};
    uint32_t L = m.t_win;
    const float *src = cond;
    int cur = 0;
    for (int l = 0; l < 4; ++l) {
        const uint32_t Lout = (L + 2u - 3u) / m.strides[l] + 1u;
        if (m.center_tap_only) linear(src, W(m.conv_w[l]), W(m.conv_b[l]), buf[cur], B, ch[l], ch[l + 1], true);   // t_win == 1: only the centre tap sees data
        else conv1d_k3(src, l == 0, W(m.conv_w[l]), W(m.conv_b[l]), buf[cur], B, ch[l], ch[l + 1], L, Lout, m.strides[l], true);
        __syncthreads();
        src = buf[cur];
        cur ^= 1;
        L = Lout;
    }
    // L == 1 here (checked on the host): src is [B][64]
    linear(src, W(m.fc_w[0]), W(m.fc_b[0]), buf[cur], B, 64u, 64u, true);
    __syncthreads();
    src = buf[cur];
    cur ^= 1;
    linear(src, W(m.fc_w[1]), W(m.fc_b[1]), buf[cur], B, 64u, m.dim_aud, false);
    __syncthreads();
    float *feat = buf[cur];   // [B][dim_aud]
    cur ^= 1;
    // ---- blink branch: feat[:, :k] += blink_encoder(blink_embedding * eye_area) ---------------------------------------
    if (m.blink_dim) {
        const float eap = eye_area ? eye_area[0] : 0.0f;
        const uint32_t half = m.dim_aud / 2u;
        float *e0 = buf[cur], *e1 = buf[cur] + 64;
        for (uint32_t i = threadIdx.x; i < half; i += kCondThreads) e0[i] = W(m.blink_emb)[i] * eap;
        __syncthreads();
        linear(e0, W(m.blink_w[0]), W(m.blink_b[0]), e1, 1u, half, half, false);
        __syncthreads();
        linear(e1, W(m.blink_w[1]), W(m.blink_b[1]), small, 1u, half, m.blink_dim, false);
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < B * m.blink_dim; idx += kCondThreads) {
            const uint32_t k = idx % m.blink_dim, b = idx / m.blink_dim;
            feat[b * m.dim_aud + k] += small[k];
        }
        __syncthreads();
    }
    if (!m.with_att) {
        // radnerf.py:104-106: without the attention net the per-window features are returned as they are ([smo, dim_aud])
        for (uint32_t i = threadIdx.x; i < B * m.dim_aud; i += kCondThreads) cond_feat[i] = feat[i];
        return;
    }
    // ---- AudioAttNet: scores = conv stack over the window axis of feat^T, softmax(Linear(scores)), weighted sum --------
    // input [1][C = dim_aud][L = smo] is feat transposed: element (c, t) = feat[t][c] == the "channels last" layout with B = 1
    const uint32_t ach[6] = {m.dim_aud, 16u, 8u, 4u, 2u, 1u};
    const float *asrc = feat;
    float *ping = buf[cur], *pong = buf[cur] + widest / 2u;
    for (int l = 0; l < 5; ++l) {
        conv1d_k3(asrc, l == 0, W(m.att_conv_w[l]), W(m.att_conv_b[l]), ping, 1u, ach[l], ach[l + 1], B, B, 1u, true);
        __syncthreads();
        asrc = ping;
        float *t = ping; ping = pong; pong = t;
    }
    // asrc: scores [smo]
    linear(asrc, W(m.att_fc_w), W(m.att_fc_b), small, 1u, B, B, false);
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = small[0];
        for (uint32_t i = 1; i < B; ++i) mx = fmaxf(mx, small[i]);
        float sum = 0.0f;
        for (uint32_t i = 0; i < B; ++i) { small[i] = expf(small[i] - mx); sum += small[i]; }
        for (uint32_t i = 0; i < B; ++i) small[i] = small[i] / sum;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < m.dim_aud; c += kCondThreads) {
        float s = 0.0f;
        for (uint32_t t = 0; t < B; ++t) s = fmaf(small[t], feat[t * m.dim_aud + c], s);
        cond_feat[c] = s;
    }
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_cond_feat_batch(const gfpp_cond_model *model, const float *cond, uint32_t cond_stride, const float *eye_area, uint32_t eye_stride,
                                  float *cond_feat, uint32_t out_stride, uint32_t count, gfpp_stream_t stream) {
    if (count == 0) return 0;
    if (!model || !cond || !cond_feat) { set_error("gfpp_cond_feat: null argument"); return GFPP_EINVAL; }
    const gfpp_cond_model &m = *model;
    if (m.smo == 0 || m.t_win == 0 || m.c_in == 0 || m.dim_aud == 0 || m.dim_aud > 64) { set_error("gfpp_cond_feat: bad dimensions"); return GFPP_EINVAL; }
    uint32_t L = m.t_win, widest = 0;
    const uint32_t ch[5] = {m.c_in, 32u, 32u, 64u, 64u};
    for (int l = 0; l < 4; ++l) {
        if (m.strides[l] == 0 || !m.conv_w[l] || !m.conv_b[l]) { set_error("gfpp_cond_feat: incomplete AudioNet"); return GFPP_EINVAL; }
        L = (L + 2u - 3u) / m.strides[l] + 1u;
        if (m.smo * ch[l + 1] * L > widest) widest = m.smo * ch[l + 1] * L;
    }
    if (L != 1) { set_error("gfpp_cond_feat: the conv stack must reduce the t-window to length 1 (got %u), as AudioNet's squeeze(-1) needs", L); return GFPP_EUNSUPPORTED; }
    if (m.smo * 16u * 2u > widest) widest = m.smo * 16u * 2u;   // ping + pong of the attention stack
    if (widest < 256u) widest = 256u;
    widest = (widest + 3u) & ~3u;
    if (m.center_tap_only && m.t_win != 1) { set_error("gfpp_cond_feat: centre-tap weights are only valid for t_win == 1"); return GFPP_EINVAL; }
    if (m.blob && ((uintptr_t)m.blob & 15u || m.blob_floats % 4u)) { set_error("gfpp_cond_feat: the weight blob must be 16-byte aligned, a multiple of 4 floats"); return GFPP_EINVAL; }
    if (2u * widest > (uint32_t)kCondLds || m.smo > 64 || (m.blink_dim && (m.dim_aud / 2u > 64 || m.blink_dim > 64))) {
        set_error("gfpp_cond_feat: window too large for the one-workgroup kernel");
        return GFPP_EUNSUPPORTED;
    }
    if (!m.fc_w[0] || !m.fc_w[1] || (m.blink_dim && (!m.blink_emb || !m.blink_w[0] || !m.blink_w[1])) || (m.with_att && !m.att_fc_w)) {
        set_error("gfpp_cond_feat: missing weights");
        return GFPP_EINVAL;
    }
    hipLaunchKernelGGL(k_cond_feat, dim3(count), dim3(kCondThreads), 0, (hipStream_t)stream, m, cond, eye_area, cond_feat, widest, cond_stride, eye_stride, out_stride);
    return check_launch("gfpp_cond_feat");
}

GFPP_API int gfpp_cond_feat(const gfpp_cond_model *model, const float *cond, const float *eye_area, float *cond_feat, gfpp_stream_t stream) {
    return gfpp_cond_feat_batch(model, cond, 0, eye_area, 0, cond_feat, 0, 1, stream);
}
