// lp_mfma_device.h -- 16-bit-operand MFMA building blocks shared by the head (frame_head_lp.hip) and torso (frame_torso_lp.hip) kernels:
// operand traits for f16 / bf16, an LDS-fed layer of K = 16 steps, relu / leaky-relu + repack of the accumulators into the next
// layer's operands, and skinny output rows as packed dot products.
//
// Layout conventions (v_mfma_f32_32x32x16_*): a layer out[32 T] = W[32 T, 16 NS] x is T row tiles x NS steps; the weight image holds
// 16-byte fragments [step][tile][lane]: lane (i = lane & 31, h = lane >> 5) carries W[32 t + i][col(step, h, 0..7)].  The samples /
// pixels are the 32 columns; lane (j, h) supplies the 8 K-values of its half for column j.  After a layer, lane (j, h) holds in
// acc[t][r] the output row 32 t + (r & 3) + 8 (r >> 2) + 4 h of column j, i.e. 16 T values = 2 T operand registers-quads of the next
// layer (step s = 2 t + (r >> 3), element e = r & 7).
#pragma once

#include <hip/hip_runtime.h>

namespace gfpp {

typedef float v16f __attribute__((ext_vector_type(16)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename H>
struct LpTraits;
template <>
struct LpTraits<_Float16> {
    typedef f16x8 vec;
    static constexpr bool kPackedMax = true;
    typedef _Float16 pair __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float dot2(pair a, pair b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
    static __device__ __forceinline__ vec relu(vec t) { return __builtin_elementwise_max(t, (vec)(_Float16)0.0f); }   // v_pk_max_f16
};
template <>
struct LpTraits<__bf16> {
    typedef bf16x8 vec;
    static constexpr bool kPackedMax = true;
    typedef __bf16 pair __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float dot2(pair a, pair b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
    // there is no packed bf16 max, but a bf16 is negative exactly when its bit pattern is a negative int16: clamp the patterns at zero
    // (v_pk_max_i16; -0.0 and negative NaNs become +0.0)
    static __device__ __forceinline__ vec relu(vec t) {
        return __builtin_bit_cast(vec, __builtin_elementwise_max(__builtin_bit_cast(s16x8, t), (s16x8)(short)0));
    }
};

// Exact-fp32 variant of the same fragment layout (the torso's fp32 mode): a "16-wide" step is eight v_mfma_f32_32x32x2_f32 -- an fp32 fma chain,
// no rounding of operands -- instruction e taking element e of both fragments (its K pair is (h = 0, h = 1) of that element).
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <>
struct LpTraits<float> {
    typedef f32x8 vec;
    static constexpr bool kPackedMax = false;
    typedef float pair __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float dot2(pair a, pair b, float c) { return fmaf(a[1], b[1], fmaf(a[0], b[0], c)); }
    static __device__ __forceinline__ vec relu(vec t) { return __builtin_elementwise_max(t, (vec)0.0f); }
};

// acc[t] += W[32 t.., 16 s..] * b[s] for NS steps; A operands come from the LDS-resident weight image, kAhead steps ahead of
// their use (the reads of all wavefronts of the workgroup queue up in the LDS; one step = T MFMAs does not cover the latency).
// The sched_barriers pin the order: read-ahead first, then this step's MFMAs.
template <typename H, int NS, int T>
__device__ __forceinline__ void mfma_layer_lds(v16f (&acc)[T], const typename LpTraits<H>::vec *__restrict__ wl, const typename LpTraits<H>::vec (&b)[NS],
                                               int lane) {
    typedef typename LpTraits<H>::vec vec;
    constexpr int kAhead = NS >= 3 ? 2 : 1;
    const vec *p = wl + lane;
    vec ring[kAhead + 1][T];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
        if (k < NS) {
#pragma unroll
            for (int t = 0; t < T; ++t) ring[k][t] = p[(k * T + t) * 64];
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + kAhead < NS) {
#pragma unroll
            for (int t = 0; t < T; ++t) ring[(s + kAhead) % (kAhead + 1)][t] = p[((s + kAhead) * T + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = LpTraits<H>::mfma(ring[s % (kAhead + 1)][t], b[s], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// act(accumulators) -> the next layer's 2 T operand registers-quads.  ACT: 0 none, 1 relu, 2 leaky relu (slope 0.02)
template <typename H, int T, int ACT>
__device__ __forceinline__ void act_pack(const v16f (&acc)[T], typename LpTraits<H>::vec (&b)[2 * T]) {
#pragma unroll
    for (int s = 0; s < 2 * T; ++s) {
        if constexpr (ACT == 1 && LpTraits<H>::kPackedMax) {
            // round first, clamp the packed pairs afterwards: relu(round(x)) == round(relu(x))
            // two values per conversion instruction (v_cvt_pk_*): spelled as 2-vectors, the compiler does not pair scalar bf16 conversions
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef typename LpTraits<H>::pair pair;
            typename LpTraits<H>::vec t;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 v = {acc[s >> 1][8 * (s & 1) + e], acc[s >> 1][8 * (s & 1) + e + 1]};
                const pair p = __builtin_convertvector(v, pair);
                t[e] = p[0];
                t[e + 1] = p[1];
            }
            b[s] = LpTraits<H>::relu(t);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = acc[s >> 1][8 * (s & 1) + e];
                b[s][e] = (H)(ACT == 1 ? fmaxf(v, 0.0f) : (ACT == 2 ? (v >= 0.0f ? v : v * 0.02f) : v));
            }
        }
    }
}

// ---- activation packing UNDER the next layer's MFMAs (round 5) -------------------------------------------------------------------------------------------
// act_pack between two layers is 64 vector instructions (one per value: convert a pair, clamp a pair) that a wavefront ran with its matrix pipe idle, and the next
// layer's 32 MFMAs (32 cycles each) then ran with its vector ALU idle: in the ISA of a 32-sample block 370 of ~1 300 vector instructions sat in such clusters
// between the layers (p48 / p40 / p56 in the instruction trace, docs/LAB_NOTEBOOK.md).  A matrix instruction occupies the pipe for 8 passes while the wavefront
// may issue independent vector instructions, so the operand of step s + 2 is packed -- two vector instructions behind each of the four MFMAs of step s, pinned by
// sched_group_barrier -- while the pipe works; only the operands of the first two steps are packed up front.  Same conversions and clamps, same order of the
// MFMAs per accumulator: the same bits as act_pack + mfma_layer_lds.
template <typename H>
__device__ __forceinline__ typename LpTraits<H>::pair relu_pair(float a, float b);
template <>
__device__ __forceinline__ LpTraits<_Float16>::pair relu_pair<_Float16>(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef LpTraits<_Float16>::pair pair;
    const pair p = __builtin_convertvector((f32x2){a, b}, pair);
    return __builtin_elementwise_max(p, (pair)(_Float16)0.0f);
}
template <>
__device__ __forceinline__ LpTraits<__bf16>::pair relu_pair<__bf16>(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef LpTraits<__bf16>::pair pair;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const pair p = __builtin_convertvector((f32x2){a, b}, pair);
    return __builtin_bit_cast(pair, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), (s16x2)(short)0));
}

// relu + round of the eight accumulator values that are operand step s of the next layer: tile s >> 1, registers 8 (s & 1) .. + 7 (act_pack<H, 4, 1>, one vector)
template <typename H>
__device__ __forceinline__ typename LpTraits<H>::vec relu_pack_step(const v16f (&acc)[4], int s) {
    typename LpTraits<H>::vec t;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const typename LpTraits<H>::pair p = relu_pair<H>(acc[s >> 1][8 * (s & 1) + e], acc[s >> 1][8 * (s & 1) + e + 1]);
        t[e] = p[0];
        t[e + 1] = p[1];
    }
    return t;
}

// acc[t] = W[32 t.., 16 s..] * relu(prev) over the 8 steps of a 128 -> 128 layer (acc starts at ZERO: the first step's MFMAs take C = 0), the operand of step
// s + 2 packed from `prev` behind the MFMAs of step s.  keep != nullptr: the eight packed operands are also handed back (the layer's input feeds a second consumer).
template <typename H>
__device__ __forceinline__ void mfma_layer_lds_fused(v16f (&acc)[4], const typename LpTraits<H>::vec *__restrict__ wl, const v16f (&prev)[4], int lane,
                                                     typename LpTraits<H>::vec (*keep)[8]) {
    typedef typename LpTraits<H>::vec vec;
    constexpr int NS = 8, T = 4, kAhead = 2;
    const vec *p = wl + lane;
    vec ring[kAhead + 1][T];
#pragma unroll
    for (int k = 0; k < kAhead; ++k)
#pragma unroll
        for (int t = 0; t < T; ++t) ring[k][t] = p[(k * T + t) * 64];
    vec b[3];
    b[0] = relu_pack_step<H>(prev, 0);
    b[1] = relu_pack_step<H>(prev, 1);
    const v16f zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + kAhead < NS) {
#pragma unroll
            for (int t = 0; t < T; ++t) ring[(s + kAhead) % (kAhead + 1)][t] = p[((s + kAhead) * T + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        const vec bs = b[s % 3];
        if (keep) (*keep)[s] = bs;
        vec nb;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[t] = LpTraits<H>::mfma(ring[s % (kAhead + 1)][t], bs, s == 0 ? zero : acc[t]);
            if (s + 2 < NS) {
                const int q = s + 2, e = 2 * t;
                const typename LpTraits<H>::pair pr = relu_pair<H>(prev[q >> 1][8 * (q & 1) + e], prev[q >> 1][8 * (q & 1) + e + 1]);
                nb[e] = pr[0];
                nb[e + 1] = pr[1];
            }
        }
        if (s + 2 < NS) {
            b[(s + 2) % 3] = nb;
            // one MFMA, then the two vector instructions of one operand pair, four times
#pragma unroll
            for (int t = 0; t < T; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Sum of the two half-waves' values, in every lane: v_permlane32_swap exchanges lanes 32..63 of one register with lanes 0..31 of another
// (gfx950), no LDS round trip like ds_bpermute (__shfl_xor).  a = b = t  ->  a = {t_lo, t_lo}, b = {t_hi, t_hi}.
__device__ __forceinline__ float half_wave_sum(float t) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Skinny output rows on packed 16-bit dot products: out[c] = sum over this lane's 8 NV activations (operand registers b) of w * x,
// fp32 accumulation (v_dot2c_f32_{f16,bf16}); the two half-waves are added.  `wrow`: LDS, rows of 4 NV 32-bit words, row index
// hi * rows_per_half + c.  (Operand pairs are taken with shufflevector: bit-casting vector ELEMENTS to pairs is miscompiled by
// ROCm 7.2's clang -- every element collapses to element 0.)
template <int C, int NV, typename H>
__device__ __forceinline__ void skinny_dot(const void *__restrict__ wrow, int rows_per_half, const typename LpTraits<H>::vec (&b)[NV], int hi,
                                           float (&out)[C]) {
    typedef typename LpTraits<H>::vec vec;
    // The weights of a row are NV 16-byte LDS reads (two distinct addresses per wavefront).  All reads of a row are issued together and the
    // next row's are in flight while this row's dot products run (the sched_barriers pin that order: left alone, the compiler issues
    // read - wait - 4 dot2 - read - wait ..., i.e. NV x C exposed LDS round trips per block).
    vec w[2][NV];
    const vec *p0 = reinterpret_cast<const vec *>(wrow) + (hi * rows_per_half) * NV;   // (a row = NV operand vectors: 4 NV words of 16-bit pairs)
#pragma unroll
    for (int s = 0; s < NV; ++s) w[0][s] = p0[s];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (c + 1 < C) {
            const vec *p = reinterpret_cast<const vec *>(wrow) + (hi * rows_per_half + c + 1) * NV;
#pragma unroll
            for (int s = 0; s < NV; ++s) w[(c + 1) & 1][s] = p[s];
        }
        __builtin_amdgcn_sched_barrier(0);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const vec ww = w[c & 1][s], x = b[s];
            s0 = LpTraits<H>::dot2(__builtin_shufflevector(ww, ww, 0, 1), __builtin_shufflevector(x, x, 0, 1), s0);
            s1 = LpTraits<H>::dot2(__builtin_shufflevector(ww, ww, 2, 3), __builtin_shufflevector(x, x, 2, 3), s1);
            s0 = LpTraits<H>::dot2(__builtin_shufflevector(ww, ww, 4, 5), __builtin_shufflevector(x, x, 4, 5), s0);
            s1 = LpTraits<H>::dot2(__builtin_shufflevector(ww, ww, 6, 7), __builtin_shufflevector(x, x, 6, 7), s1);
        }
        __builtin_amdgcn_sched_barrier(0);
        out[c] = half_wave_sum(s0 + s1);
    }
}

}  // namespace gfpp
