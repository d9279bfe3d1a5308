// cond_nets.hip -- RADNeRF.cal_cond_feat (radnerf.py:88-106) as ONE launch of one workgroup.
//
// The reference runs AudioNet (cond_encoder.py:98-143: four k=3 Conv1d + LeakyReLU over the t-window, two Linear), the blink
// branch (radnerf.py:97-103) and AudioAttNet (cond_encoder.py:146-180: five k=3 Conv1d over the smoothing window, Linear + softmax,
// weighted sum) as ~70 PyTorch launches on a [smo, t_window, c_in] window -- ~0.1 MFLOP, i.e. pure launch latency (0.6 ms of
// host time per frame).  Here every layer is a loop over output elements inside one 256-thread workgroup, activations ping-pong
// between two LDS buffers, and the 64 output values land in device memory where gfpp_head_frame_begin reads them: no host round
// trip.  fp32 with explicit fmaf chains in (channel, tap) order.
#include <hip/hip_runtime.h>

#include "gfpp_common.h"

namespace gfpp {

constexpr int kCondThreads = 256;
constexpr int kCondBuf = 8192;   // floats per LDS activation buffer: smo * channels * length of the widest hidden layer

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.02f * v; }

// out[b][co][t] = act(bias[co] + sum_ci sum_k w[co][ci][k] * in[b][ci][t*stride + k - 1])   (zero padding 1), b < B
// `in` is [B][Cin][Lin] when in_channels_last == false, or [B][Lin][Cin] (the layout cond arrives in) when true.
__device__ void conv1d_k3(const float *__restrict__ in, bool in_channels_last, const float *__restrict__ w, const float *__restrict__ bias,
                          float *__restrict__ out, uint32_t B, uint32_t Cin, uint32_t Cout, uint32_t Lin, uint32_t Lout, uint32_t stride, bool act) {
    const uint32_t total = B * Cout * Lout;
    for (uint32_t idx = threadIdx.x; idx < total; idx += kCondThreads) {
        const uint32_t t = idx % Lout, co = (idx / Lout) % Cout, b = idx / (Lout * Cout);
        float s = bias ? bias[co] : 0.0f;
        const float *wc = w + (size_t)co * Cin * 3;
        for (uint32_t ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int p = (int)(t * stride) + k - 1;
                if (p < 0 || p >= (int)Lin) continue;
                const float x = in_channels_last ? in[((size_t)b * Lin + p) * Cin + ci] : in[((size_t)b * Cin + ci) * Lin + p];
                s = fmaf(wc[ci * 3 + k], x, s);
            }
        }
        out[idx] = act ? leaky(s) : s;
    }
}

// out[b][o] = act(bias[o] + sum_i w[o][i] * in[b][i])
__device__ void linear(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out, uint32_t B,
                       uint32_t In, uint32_t Out, bool act) {
    for (uint32_t idx = threadIdx.x; idx < B * Out; idx += kCondThreads) {
        const uint32_t o = idx % Out, b = idx / Out;
        float s = bias ? bias[o] : 0.0f;
        for (uint32_t i = 0; i < In; ++i) s = fmaf(w[(size_t)o * In + i], in[(size_t)b * In + i], s);
        out[idx] = act ? leaky(s) : s;
    }
}

__global__ __launch_bounds__(kCondThreads) void k_cond_feat(gfpp_cond_model m, const float *__restrict__ cond, const float *__restrict__ eye_area,
                                                           float *__restrict__ cond_feat) {
    __shared__ float buf[2][kCondBuf];
    __shared__ float small[64];
    const uint32_t B = m.smo;
    // ---- AudioNet: conv stack over the t-window ---------------------------------------------------------------------
    const uint32_t ch[5] = {m.c_in, 32u, 32u, 64u, 64u};
    uint32_t L = m.t_win;
    const float *src = cond;
    int cur = 0;
    for (int l = 0; l < 4; ++l) {
        const uint32_t Lout = (L + 2u - 3u) / m.strides[l] + 1u;
        conv1d_k3(src, l == 0, m.conv_w[l], m.conv_b[l], buf[cur], B, ch[l], ch[l + 1], L, Lout, m.strides[l], true);
        __syncthreads();
        src = buf[cur];
        cur ^= 1;
        L = Lout;
    }
    // L == 1 here (checked on the host): src is [B][64]
    linear(src, m.fc_w[0], m.fc_b[0], buf[cur], B, 64u, 64u, true);
    __syncthreads();
    src = buf[cur];
    cur ^= 1;
    linear(src, m.fc_w[1], m.fc_b[1], buf[cur], B, 64u, m.dim_aud, false);
    __syncthreads();
    float *feat = buf[cur];   // [B][dim_aud]
    cur ^= 1;
    // ---- blink branch: feat[:, :k] += blink_encoder(blink_embedding * eye_area) ---------------------------------------
    if (m.blink_dim) {
        const float eap = eye_area ? eye_area[0] : 0.0f;
        const uint32_t half = m.dim_aud / 2u;
        float *e0 = buf[cur], *e1 = buf[cur] + 64;
        for (uint32_t i = threadIdx.x; i < half; i += kCondThreads) e0[i] = m.blink_emb[i] * eap;
        __syncthreads();
        linear(e0, m.blink_w[0], m.blink_b[0], e1, 1u, half, half, false);
        __syncthreads();
        linear(e1, m.blink_w[1], m.blink_b[1], small, 1u, half, m.blink_dim, false);
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
    float *ping = buf[cur], *pong = buf[cur] + kCondBuf / 2;
    for (int l = 0; l < 5; ++l) {
        conv1d_k3(asrc, l == 0, m.att_conv_w[l], m.att_conv_b[l], ping, 1u, ach[l], ach[l + 1], B, B, 1u, true);
        __syncthreads();
        asrc = ping;
        float *t = ping; ping = pong; pong = t;
    }
    // asrc: scores [smo]
    linear(asrc, m.att_fc_w, m.att_fc_b, small, 1u, B, B, false);
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

GFPP_API int gfpp_cond_feat(const gfpp_cond_model *model, const float *cond, const float *eye_area, float *cond_feat, gfpp_stream_t stream) {
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
    if (widest > (uint32_t)kCondBuf || m.smo * 16u > (uint32_t)kCondBuf / 2u || m.smo > 64 || (m.blink_dim && (m.dim_aud / 2u > 64 || m.blink_dim > 64))) {
        set_error("gfpp_cond_feat: window too large for the one-workgroup kernel");
        return GFPP_EUNSUPPORTED;
    }
    if (!m.fc_w[0] || !m.fc_w[1] || (m.blink_dim && (!m.blink_emb || !m.blink_w[0] || !m.blink_w[1])) || (m.with_att && !m.att_fc_w)) {
        set_error("gfpp_cond_feat: missing weights");
        return GFPP_EINVAL;
    }
    hipLaunchKernelGGL(k_cond_feat, dim3(1), dim3(kCondThreads), 0, (hipStream_t)stream, m, cond, eye_area, cond_feat);
    return check_launch("gfpp_cond_feat");
}
