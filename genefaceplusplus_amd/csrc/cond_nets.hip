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

// ---- training: the same networks with every activation kept, and their backward pass, one launch each ---------------------------------------------------------
// A training step runs cal_cond_feat under autograd: ~120 eager launches of 2-16 us on a [<= 8, <= 16, <= 204] window, every convolution call ~0.1 ms of host
// time in MIOpen -- 2.0 of a May step's 5.5 ms (tools/profile_train.py); as two captured graphs the ~100 kernel nodes still take 0.6 + 0.7 ms of GPU time,
// dependency after dependency.  Here: one workgroup, fp32, activations in a caller-provided buffer (L2-resident), a thread per output element and plain sums --
// the whole backward pass is ~2 MFLOP.  Gradients are OVERWRITTEN (autograd accumulates them into .grad itself).
struct CondLayout {
    uint32_t L[5];          // lengths of the t axis before / after each AudioNet convolution
    uint32_t a[5];          // a[l]: output of convolution l - 1 (l = 1..4), [B][ch[l]][L[l]]
    uint32_t f1, feat, e0, e1, e2, s[6], w, saved_total;
    // backward-only
    uint32_t dfeat, g0, g1, vec, scratch_total;
};

__host__ __device__ inline void cond_layout(const gfpp_cond_model &m, CondLayout &o) {
    const uint32_t ch[5] = {m.c_in, 32u, 32u, 64u, 64u}, ach[6] = {m.dim_aud, 16u, 8u, 4u, 2u, 1u};
    const uint32_t B = m.smo, half = m.dim_aud / 2u;
    uint32_t at = 0, widest = 64u * B > m.dim_aud * B ? 64u * B : m.dim_aud * B;
    o.L[0] = m.t_win;
    o.a[0] = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        o.L[l + 1] = (o.L[l] - 1u) / m.strides[l] + 1u;
        o.a[l + 1] = at;
        at += B * ch[l + 1] * o.L[l + 1];
        if (B * ch[l + 1] * o.L[l + 1] > widest) widest = B * ch[l + 1] * o.L[l + 1];
    }
    o.f1 = at; at += B * 64u;
    o.feat = at; at += B * m.dim_aud;
    o.e0 = at; at += half;
    o.e1 = at; at += half;
    o.e2 = at; at += 64u;
    o.s[0] = o.feat;
#pragma unroll
    for (int l = 0; l < 5; ++l) { o.s[l + 1] = at; at += ach[l + 1] * B; }
    o.w = at; at += 64u;
    o.saved_total = at;
    uint32_t bt = 0;
    o.dfeat = bt; bt += B * m.dim_aud;
    o.g0 = bt; bt += widest;
    o.g1 = bt; bt += widest;
    o.vec = bt; bt += 8u * 64u;
    o.scratch_total = bt;
}

// the derivative of the leaky relu from its OUTPUT (slope > 0: output and pre-activation have the same sign), in place
__device__ void leaky_bwd(float *__restrict__ g, const float *__restrict__ out, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += kCondThreads) g[i] = out[i] > 0.0f ? g[i] : 0.02f * g[i];
}

// dW[co][ci][k] = sum_b sum_t dP[b][co][t] X[b][ci][t stride + k - 1];  db[co] = sum_b sum_t dP[b][co][t]          (X as in conv1d_k3)
__device__ void conv_bwd_w(const float *__restrict__ dP, const float *__restrict__ X, bool x_channels_last, float *__restrict__ dW, float *__restrict__ db, uint32_t B,
                           uint32_t Cin, uint32_t Cout, uint32_t Lin, uint32_t Lout, uint32_t stride) {
    const uint32_t cs = x_channels_last ? 1u : Lin, ps = x_channels_last ? Cin : 1u;
    for (uint32_t idx = threadIdx.x; idx < Cout * Cin * 3u; idx += kCondThreads) {
        const uint32_t k = idx % 3u, ci = (idx / 3u) % Cin, co = idx / (3u * Cin);
        float s = 0.0f;
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t t = 0; t < Lout; ++t) {
                const int p = (int)(t * stride + k) - 1;
                if (p >= 0 && p < (int)Lin) s = fmaf(dP[(b * Cout + co) * Lout + t], X[(size_t)b * Cin * Lin + ci * cs + (uint32_t)p * ps], s);
            }
        dW[idx] = s;
    }
    for (uint32_t co = threadIdx.x; co < Cout; co += kCondThreads) {
        float s = 0.0f;
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t t = 0; t < Lout; ++t) s += dP[(b * Cout + co) * Lout + t];
        db[co] = s;
    }
}

// dX[b][ci][p] (= or +=) sum_co sum_k W[co][ci][k] dP[b][co][t],  t stride + k - 1 == p
__device__ void conv_bwd_x(const float *__restrict__ dP, const float *__restrict__ W, float *__restrict__ dX, bool dx_channels_last, bool accumulate, uint32_t B,
                           uint32_t Cin, uint32_t Cout, uint32_t Lin, uint32_t Lout, uint32_t stride) {
    const uint32_t cs = dx_channels_last ? 1u : Lin, ps = dx_channels_last ? Cin : 1u;
    for (uint32_t idx = threadIdx.x; idx < B * Cin * Lin; idx += kCondThreads) {
        const uint32_t p = idx % Lin, ci = (idx / Lin) % Cin, b = idx / (Lin * Cin);
        float s = 0.0f;
        for (uint32_t k = 0; k < 3u; ++k) {
            const int q = (int)p + 1 - (int)k;                       // = t stride
            if (q < 0 || q % (int)stride != 0) continue;
            const uint32_t t = (uint32_t)q / stride;
            if (t >= Lout) continue;
            for (uint32_t co = 0; co < Cout; ++co) s = fmaf(W[((size_t)co * Cin + ci) * 3u + k], dP[(b * Cout + co) * Lout + t], s);
        }
        float *dst = dX + (size_t)b * Cin * Lin + ci * cs + p * ps;
        *dst = accumulate ? *dst + s : s;
    }
}

// dW[o][i] = sum_b dY[b][o] X[b][i];  db[o] = sum_b dY[b][o]
__device__ void linear_bwd_w(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ dW, float *__restrict__ db, uint32_t B, uint32_t In, uint32_t Out) {
    for (uint32_t idx = threadIdx.x; idx < Out * In; idx += kCondThreads) {
        const uint32_t i = idx % In, o = idx / In;
        float s = 0.0f;
        for (uint32_t b = 0; b < B; ++b) s = fmaf(dY[b * Out + o], X[b * In + i], s);
        dW[idx] = s;
    }
    if (db)
        for (uint32_t o = threadIdx.x; o < Out; o += kCondThreads) {
            float s = 0.0f;
            for (uint32_t b = 0; b < B; ++b) s += dY[b * Out + o];
            db[o] = s;
        }
}

// dX[b][i] = sum_o W[o][i] dY[b][o]
__device__ void linear_bwd_x(const float *__restrict__ dY, const float *__restrict__ W, float *__restrict__ dX, uint32_t B, uint32_t In, uint32_t Out) {
    for (uint32_t idx = threadIdx.x; idx < B * In; idx += kCondThreads) {
        const uint32_t i = idx % In, b = idx / In;
        float s = 0.0f;
        for (uint32_t o = 0; o < Out; ++o) s = fmaf(W[(size_t)o * In + i], dY[b * Out + o], s);
        dX[idx] = s;
    }
}

__global__ __launch_bounds__(kCondThreads) void k_cond_feat_train_fwd(gfpp_cond_model m, const float *__restrict__ cond, const float *__restrict__ eye_area,
                                                                     float *__restrict__ cond_feat, float *__restrict__ save) {
    CondLayout lay;
    cond_layout(m, lay);
    const uint32_t B = m.smo, ch[5] = {m.c_in, 32u, 32u, 64u, 64u}, ach[6] = {m.dim_aud, 16u, 8u, 4u, 2u, 1u};
    const float *src = cond;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        conv1d_k3(src, l == 0, m.conv_w[l], m.conv_b[l], save + lay.a[l + 1], B, ch[l], ch[l + 1], lay.L[l], lay.L[l + 1], m.strides[l], true);
        __syncthreads();
        src = save + lay.a[l + 1];
    }
    linear(src, m.fc_w[0], m.fc_b[0], save + lay.f1, B, 64u, 64u, true);
    __syncthreads();
    float *feat = save + lay.feat;
    linear(save + lay.f1, m.fc_w[1], m.fc_b[1], feat, B, 64u, m.dim_aud, false);
    __syncthreads();
    if (m.blink_dim) {
        const float eap = eye_area ? eye_area[0] : 0.0f;
        const uint32_t half = m.dim_aud / 2u;
        for (uint32_t i = threadIdx.x; i < half; i += kCondThreads) save[lay.e0 + i] = m.blink_emb[i] * eap;
        __syncthreads();
        linear(save + lay.e0, m.blink_w[0], m.blink_b[0], save + lay.e1, 1u, half, half, false);
        __syncthreads();
        linear(save + lay.e1, m.blink_w[1], m.blink_b[1], save + lay.e2, 1u, half, m.blink_dim, false);
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < B * m.blink_dim; idx += kCondThreads) feat[(idx / m.blink_dim) * m.dim_aud + idx % m.blink_dim] += save[lay.e2 + idx % m.blink_dim];
        __syncthreads();
    }
    if (!m.with_att) {
        for (uint32_t i = threadIdx.x; i < B * m.dim_aud; i += kCondThreads) cond_feat[i] = feat[i];
        return;
    }
    const float *asrc = feat;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        conv1d_k3(asrc, l == 0, m.att_conv_w[l], m.att_conv_b[l], save + lay.s[l + 1], 1u, ach[l], ach[l + 1], B, B, 1u, true);
        __syncthreads();
        asrc = save + lay.s[l + 1];
    }
    float *w = save + lay.w;
    linear(asrc, m.att_fc_w, m.att_fc_b, w, 1u, B, B, false);
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = w[0];
        for (uint32_t i = 1; i < B; ++i) mx = fmaxf(mx, w[i]);
        float sum = 0.0f;
        for (uint32_t i = 0; i < B; ++i) { w[i] = expf(w[i] - mx); sum += w[i]; }
        for (uint32_t i = 0; i < B; ++i) w[i] = w[i] / sum;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < m.dim_aud; c += kCondThreads) {
        float s = 0.0f;
        for (uint32_t t = 0; t < B; ++t) s = fmaf(w[t], feat[t * m.dim_aud + c], s);
        cond_feat[c] = s;
    }
}

// g: a gfpp_cond_model whose pointers say where each parameter's gradient goes (same shapes as the parameters)
__global__ __launch_bounds__(kCondThreads) void k_cond_feat_train_bwd(gfpp_cond_model m, gfpp_cond_model g, const float *__restrict__ cond, const float *__restrict__ eye_area,
                                                                     const float *__restrict__ save, const float *__restrict__ dout, float *__restrict__ tmp) {
    CondLayout lay;
    cond_layout(m, lay);
    const uint32_t B = m.smo, D = m.dim_aud, ch[5] = {m.c_in, 32u, 32u, 64u, 64u}, ach[6] = {m.dim_aud, 16u, 8u, 4u, 2u, 1u};
    auto G = [](const float *p) { return const_cast<float *>(p); };
    const float *feat = save + lay.feat;
    float *dfeat = tmp + lay.dfeat, *ga = tmp + lay.g0, *gb = tmp + lay.g1, *vec = tmp + lay.vec;
    if (!m.with_att) {
        for (uint32_t i = threadIdx.x; i < B * D; i += kCondThreads) dfeat[i] = dout[i];
        __syncthreads();
    } else {
        const float *w = save + lay.w;
        float *dw = vec, *dz = vec + 64;
        // out[c] = sum_t w[t] feat[t][c]
        for (uint32_t t = threadIdx.x; t < B; t += kCondThreads) {
            float s = 0.0f;
            for (uint32_t c = 0; c < D; ++c) s = fmaf(dout[c], feat[t * D + c], s);
            dw[t] = s;
        }
        for (uint32_t i = threadIdx.x; i < B * D; i += kCondThreads) dfeat[i] = w[i / D] * dout[i % D];
        __syncthreads();
        // softmax, then Linear(scores)
        if (threadIdx.x < B) {
            float dot = 0.0f;
            for (uint32_t j = 0; j < B; ++j) dot = fmaf(w[j], dw[j], dot);
            dz[threadIdx.x] = w[threadIdx.x] * (dw[threadIdx.x] - dot);
        }
        __syncthreads();
        const float *scores = save + lay.s[5];
        linear_bwd_w(dz, scores, G(g.att_fc_w), G(g.att_fc_b), 1u, B, B);
        linear_bwd_x(dz, m.att_fc_w, ga, 1u, B, B);                      // d scores [1][B] = [B][1 channel][L = B] post-activation gradient of the last convolution
        __syncthreads();
        // the five k = 3 convolutions over the window axis, last to first
        float *dcur = ga, *dnext = gb;
#pragma unroll
        for (int l = 4; l >= 0; --l) {
            leaky_bwd(dcur, save + lay.s[l + 1], ach[l + 1] * B);
            __syncthreads();
            conv_bwd_w(dcur, save + lay.s[l], l == 0, G(g.att_conv_w[l]), G(g.att_conv_b[l]), 1u, ach[l], ach[l + 1], B, B, 1u);
            if (l > 0) conv_bwd_x(dcur, m.att_conv_w[l], dnext, false, false, 1u, ach[l], ach[l + 1], B, B, 1u);
            else conv_bwd_x(dcur, m.att_conv_w[0], dfeat, true, true, 1u, ach[0], ach[1], B, B, 1u);      // its input is feat^T: the gradient joins dfeat
            __syncthreads();
            float *t = dcur; dcur = dnext; dnext = t;
        }
    }
    // ---- blink branch: feat[:, :k] += blink_encoder(blink_embedding * eye_area)
    if (m.blink_dim) {
        const float eap = eye_area ? eye_area[0] : 0.0f;
        const uint32_t half = D / 2u, K = m.blink_dim;
        float *de2 = vec + 128, *de1 = vec + 192, *de0 = vec + 256;
        for (uint32_t k = threadIdx.x; k < K; k += kCondThreads) {
            float s = 0.0f;
            for (uint32_t b = 0; b < B; ++b) s += dfeat[b * D + k];
            de2[k] = s;
        }
        __syncthreads();
        linear_bwd_w(de2, save + lay.e1, G(g.blink_w[1]), G(g.blink_b[1]), 1u, half, K);
        linear_bwd_x(de2, m.blink_w[1], de1, 1u, half, K);
        __syncthreads();
        linear_bwd_w(de1, save + lay.e0, G(g.blink_w[0]), G(g.blink_b[0]), 1u, half, half);
        linear_bwd_x(de1, m.blink_w[0], de0, 1u, half, half);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < half; i += kCondThreads) G(g.blink_emb)[i] = de0[i] * eap;
        __syncthreads();
    }
    // ---- AudioNet: two Linear layers, four convolutions over the t axis
    linear_bwd_w(dfeat, save + lay.f1, G(g.fc_w[1]), G(g.fc_b[1]), B, 64u, D);
    linear_bwd_x(dfeat, m.fc_w[1], ga, B, 64u, D);
    __syncthreads();
    leaky_bwd(ga, save + lay.f1, B * 64u);
    __syncthreads();
    linear_bwd_w(ga, save + lay.a[4], G(g.fc_w[0]), G(g.fc_b[0]), B, 64u, 64u);
    linear_bwd_x(ga, m.fc_w[0], gb, B, 64u, 64u);                        // d a[4] [B][64][L = 1]
    __syncthreads();
    float *dcur = gb, *dnext = ga;
#pragma unroll
    for (int l = 3; l >= 0; --l) {
        leaky_bwd(dcur, save + lay.a[l + 1], B * ch[l + 1] * lay.L[l + 1]);
        __syncthreads();
        conv_bwd_w(dcur, l == 0 ? cond : save + lay.a[l], l == 0, G(g.conv_w[l]), G(g.conv_b[l]), B, ch[l], ch[l + 1], lay.L[l], lay.L[l + 1], m.strides[l]);
        if (l > 0) conv_bwd_x(dcur, m.conv_w[l], dnext, false, false, B, ch[l], ch[l + 1], lay.L[l], lay.L[l + 1], m.strides[l]);
        __syncthreads();
        float *t = dcur; dcur = dnext; dnext = t;
    }
}

static int cond_train_check(const char *who, const gfpp_cond_model &m) {
    if (m.smo == 0 || m.t_win == 0 || m.c_in == 0 || m.dim_aud == 0 || m.dim_aud > 64 || m.smo > 64 || m.blink_dim > 64) { set_error("%s: bad dimensions", who); return GFPP_EUNSUPPORTED; }
    if (m.center_tap_only || m.blob) { set_error("%s: takes the parameters as they are (Conv1d [out, in, 3]), no centre-tap copies, no blob", who); return GFPP_EINVAL; }
    uint32_t L = m.t_win;
    for (int l = 0; l < 4; ++l) {
        if (m.strides[l] == 0 || !m.conv_w[l] || !m.conv_b[l]) { set_error("%s: incomplete AudioNet", who); return GFPP_EINVAL; }
        L = (L - 1u) / m.strides[l] + 1u;
    }
    if (L != 1) { set_error("%s: the conv stack must reduce the t-window to length 1 (got %u)", who, L); return GFPP_EUNSUPPORTED; }
    if (!m.fc_w[0] || !m.fc_w[1] || !m.fc_b[0] || !m.fc_b[1] || (m.blink_dim && (!m.blink_emb || !m.blink_w[0] || !m.blink_w[1] || !m.blink_b[0] || !m.blink_b[1]))
        || (m.with_att && (!m.att_fc_w || !m.att_fc_b))) {
        set_error("%s: missing weights", who);
        return GFPP_EINVAL;
    }
    for (int l = 0; l < 5 && m.with_att; ++l)
        if (!m.att_conv_w[l] || !m.att_conv_b[l]) { set_error("%s: incomplete AudioAttNet", who); return GFPP_EINVAL; }
    return 0;
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API uint32_t gfpp_cond_feat_train_floats(const gfpp_cond_model *model, int scratch) {
    if (!model || cond_train_check("gfpp_cond_feat_train_floats", *model)) return 0;
    CondLayout lay;
    cond_layout(*model, lay);
    return scratch ? lay.scratch_total : lay.saved_total;
}

GFPP_API int gfpp_cond_feat_train_forward(const gfpp_cond_model *model, const float *cond, const float *eye_area, float *cond_feat, float *saved, gfpp_stream_t stream) {
    const char *who = "gfpp_cond_feat_train_forward";
    if (!model || !cond || !cond_feat || !saved) { set_error("%s: null argument", who); return GFPP_EINVAL; }
    int rc = cond_train_check(who, *model);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cond_feat_train_fwd, dim3(1), dim3(kCondThreads), 0, (hipStream_t)stream, *model, cond, eye_area, cond_feat, saved);
    return check_launch(who);
}

GFPP_API int gfpp_cond_feat_train_backward(const gfpp_cond_model *model, const gfpp_cond_model *grads, const float *cond, const float *eye_area, const float *saved,
                                           const float *grad_out, float *scratch, gfpp_stream_t stream) {
    const char *who = "gfpp_cond_feat_train_backward";
    if (!model || !grads || !cond || !saved || !grad_out || !scratch) { set_error("%s: null argument", who); return GFPP_EINVAL; }
    int rc = cond_train_check(who, *model);
    if (rc) return rc;
    gfpp_cond_model g = *grads;
    g.smo = model->smo; g.t_win = model->t_win; g.c_in = model->c_in; g.dim_aud = model->dim_aud; g.blink_dim = model->blink_dim; g.with_att = model->with_att;
    g.center_tap_only = 0; g.blob = nullptr; g.blob_floats = 0;
    for (int l = 0; l < 4; ++l) g.strides[l] = model->strides[l];
    rc = cond_train_check(who, g);                                       // every gradient has a place
    if (rc) return rc;
    hipLaunchKernelGGL(k_cond_feat_train_bwd, dim3(1), dim3(kCondThreads), 0, (hipStream_t)stream, *model, g, cond, eye_area, saved, grad_out, scratch);
    return check_launch(who);
}

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
