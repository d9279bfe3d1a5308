// train_mlp.hip -- the weight gradient of the bias-free Linear layers of the radiance MLPs (cond_encoder.py:183-202 `MLP`, used by ambient_net /
// sigma_net / color_net, radnerf.py:60-100) over one training batch:
//
//     dW [O, I] = dY^T [O, M] x X [M, I],      M = the step's samples (~3 x 10^5), O, I <= 256 x 160
//
// i.e. a 128 x 128 output with a reduction 300 000 long.  The BLAS heuristics pick an output-tiled kernel without split-K for this shape (measured:
// 0.7 ms per layer in fp32, 0.8-2.2 ms in half -- eight layers were 27 % / 46 % of a training step); what the shape wants is the opposite: every
// workgroup streams its own slice of the M rows ONCE, keeps the whole O x I output in MFMA accumulators, and the slices are summed at the end.
//
// Both operands have the reduction index as their SLOW axis in memory, while an MFMA lane supplies eight consecutive k of one row: a chunk of rows is
// staged row-major in LDS (coalesced 16-byte loads, one chunk ahead in registers) and the fragments are read from there transposed (eight 2- or 4-byte
// LDS reads per fragment; wavefront w owns the output row tiles w and w + 4 and all column tiles, so a k-step is <= 7 fragments for <= 10 MFMAs).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "gfpp_common.h"
#include "lp_mfma_device.h"

namespace gfpp {

constexpr int kWgThreads = 256;
constexpr int kWgMaxTO = 8, kWgMaxTI = 5;   // <= 256 output rows (two row tiles per wavefront), <= 160 columns
constexpr uint32_t kWgMaxSlices = 512;      // workgroups = slices of the M rows

template <typename G>
struct WgCfg;
template <>
struct WgCfg<_Float16> { static constexpr int rows = 64; };
template <>
struct WgCfg<float> { static constexpr int rows = 32; };

struct WgArgs {
    const void *gy, *x;     // [M, O], [M, I] row-major
    float *partial;         // [slices, O, I]
    uint32_t M, O, I, TO, TI, chunks_per_wg, n_chunks;
};

// A chunk = the rows [m0, m0 + R) of a row-major [M, C] matrix = ONE contiguous byte range, whatever C is (3, 129, 148 ... columns): it is copied as
// 16-byte vectors from the aligned address below its start -- first into registers (issued before the previous chunk is computed, so the loads fly
// under its MFMAs), then into LDS with the same flat layout (pitch = C, element (r, c) at shift + r C + c).  Vectors that start beyond the chunk's
// last valid row are zero; the one that straddles the end of the matrix is assembled element by element (no read past the matrix).
template <typename G, int MAXV>
struct WgChunk {
    uint4 v[MAXV];
    __device__ __forceinline__ void load(const G *__restrict__ src, uint32_t C, uint32_t M, uint32_t m0, uint32_t rows, int tid) {
        const size_t g0 = (size_t)m0 * C * sizeof(G), ga = g0 & ~(size_t)15;
        const uint32_t mend = m0 + rows < M ? m0 + rows : M;
        const size_t g1 = (size_t)mend * C * sizeof(G);
        const char *p = reinterpret_cast<const char *>(src);
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            const size_t at = ga + ((size_t)q * kWgThreads + (size_t)tid) * 16u;
            if (at + 16u <= g1) {
                v[q] = *reinterpret_cast<const uint4 *>(p + at);
            } else {
                // the vector that straddles the end of the matrix (at most one per chunk load): element by element, nothing is read past the last row --
                // the buffer may sit at the very end of an allocation
                G tmp[16 / sizeof(G)];
#pragma unroll
                for (uint32_t e = 0; e < 16u / sizeof(G); ++e) tmp[e] = at + (e + 1u) * sizeof(G) <= g1 ? *reinterpret_cast<const G *>(p + at + e * sizeof(G)) : (G)0.0f;
                v[q] = *reinterpret_cast<const uint4 *>(tmp);
            }
        }
    }
    __device__ __forceinline__ void store(G *lds, int tid) const {
#pragma unroll
        for (int q = 0; q < MAXV; ++q) reinterpret_cast<uint4 *>(lds)[q * kWgThreads + tid] = v[q];
    }
};

template <typename G>
__global__ __launch_bounds__(kWgThreads) void k_linear_wgrad(WgArgs a) {
    typedef typename LpTraits<G>::vec vec;
    constexpr int R = WgCfg<G>::rows, STEPS = R / 16;
    // vectors per thread: R rows x <= 256 (dY) / <= 160 (X) columns + the 16 bytes of misalignment
    constexpr int VY = (R * 256 * (int)sizeof(G) + 16) / (kWgThreads * 16) + 1, VX = (R * 160 * (int)sizeof(G) + 16) / (kWgThreads * 16) + 1;
    __shared__ __attribute__((aligned(16))) uint4 s_y[VY * kWgThreads], s_x[VX * kWgThreads];
    const G *sy = reinterpret_cast<const G *>(s_y), *sx = reinterpret_cast<const G *>(s_x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    v16f acc[2][kWgMaxTI];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < kWgMaxTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][t][r] = 0.0f;
    const G *gy = static_cast<const G *>(a.gy), *x = static_cast<const G *>(a.x);
    const uint32_t c0 = blockIdx.x * a.chunks_per_wg, c1 = c0 + a.chunks_per_wg < a.n_chunks ? c0 + a.chunks_per_wg : a.n_chunks;
    WgChunk<G, VY> ry;
    WgChunk<G, VX> rx;
    if (c0 < c1) {
        ry.load(gy, a.O, a.M, c0 * (uint32_t)R, (uint32_t)R, tid);
        rx.load(x, a.I, a.M, c0 * (uint32_t)R, (uint32_t)R, tid);
    }
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t m0 = c * (uint32_t)R;
        ry.store(const_cast<G *>(sy), tid);
        rx.store(const_cast<G *>(sx), tid);
        __syncthreads();
        if (c + 1 < c1) {                                          // the next chunk's rows: in flight under this chunk's MFMAs
            ry.load(gy, a.O, a.M, m0 + (uint32_t)R, (uint32_t)R, tid);
            rx.load(x, a.I, a.M, m0 + (uint32_t)R, (uint32_t)R, tid);
        }
        const uint32_t shy = (uint32_t)((((size_t)m0 * a.O * sizeof(G)) & 15u) / sizeof(G)), shx = (uint32_t)((((size_t)m0 * a.I * sizeof(G)) & 15u) / sizeof(G));
        const uint32_t valid = a.M - m0 < (uint32_t)R ? a.M - m0 : (uint32_t)R;      // rows of this chunk that exist
        // a fragment = this lane's column of the operand at 8 consecutive rows: unconditional LDS reads (a column beyond the matrix reads column 0 and
        // is zeroed afterwards; rows beyond M exist only in the job's last chunk, which takes the per-element path)
        auto fragment = [&](const G *sm, uint32_t shift, uint32_t C, uint32_t col, uint32_t row) {
            const bool col_ok = col < C;
            const G *p = sm + shift + row * C + (col_ok ? col : 0u);
            vec f;
            if (valid == (uint32_t)R) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = p[e * C];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = row + e < valid;
                    const G v = p[ok ? e * C : 0u];
                    f[e] = ok ? v : (G)0.0f;
                }
            }
            if (!col_ok) f = (vec)(G)0.0f;
            return f;
        };
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const uint32_t row = (uint32_t)(16 * s + 8 * h);
            vec B[kWgMaxTI];
#pragma unroll
            for (int t = 0; t < kWgMaxTI; ++t)
                if ((uint32_t)t < a.TI) B[t] = fragment(sx, shx, a.I, 32u * t + (uint32_t)i, row);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t to = (uint32_t)wave + 4u * k;
                if (to < a.TO) {                                   // wavefront-uniform
                    const vec A = fragment(sy, shy, a.O, 32u * to + (uint32_t)i, row);
#pragma unroll
                    for (int t = 0; t < kWgMaxTI; ++t)
                        if ((uint32_t)t < a.TI) acc[k][t] = LpTraits<G>::mfma(A, B[t], acc[k][t]);
                }
            }
        }
        __syncthreads();
    }
    // lane (j = i, h) holds rows (r & 3) + 8 (r >> 2) + 4 h of column j of every tile
    float *out = a.partial + (size_t)blockIdx.x * a.O * a.I;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t to = (uint32_t)wave + 4u * k;
        if (to >= a.TO) continue;
#pragma unroll
        for (int t = 0; t < kWgMaxTI; ++t) {
            if ((uint32_t)t >= a.TI) continue;
            const uint32_t col = 32u * t + (uint32_t)i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t o = 32u * to + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h);
                if (o < a.O && col < a.I) out[(size_t)o * a.I + col] = acc[k][t][r];
            }
        }
    }
}

// ---- zero-padded half matrices (widths multiples of 32: what the fused MLP launches write, train_mlp_fused.hip) -------------------------------------------------
// The kernel above gathers a fragment -- one column at 8 consecutive rows -- with eight 2-byte LDS reads, (2 + TI) x 8 of them per 16-row step for <= 10 MFMAs, with
// one wavefront per SIMD (194 VGPRs + 160 AGPRs): measured 143-146 us per 128 x 128 layer over 3 x 10^5 rows, ~1 TB/s, 22 % of a fused training step's kernel time.
// gfx950's transposing LDS read (ds_read_b64_tr_b16: the 16 lanes of a quarter wavefront read a 4-row x 16-column block, lane p the four halves at its own
// address = row p / 4, columns 4 (p % 4) ..; lane l gets column l of the block, rows 0 .. 3) delivers the same fragment in TWO reads, if the rows start on 8-byte
// boundaries -- true for these matrices.  Chunks of 64 rows are staged row-major with a row pitch = 64 or 192 (mod 256) bytes (the 8 rows x 64 bytes one read
// instruction touches then fall into distinct banks), double-buffered (one barrier per chunk, the chunk after next in flight in registers), eight wavefronts
// as 4 row-tile groups x 2 column-tile groups with <= 2 x 3 accumulator tiles each: two wavefronts per SIMD.
constexpr int kWtThreads = 512, kWtRows = 64;
__host__ __device__ constexpr int wt_pitch(int C) { return (C * 2) % 128 == 0 ? C * 2 + 64 : C * 2; }
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 wt_fp16x4;
typedef _Float16 wt_f16x4 __attribute__((ext_vector_type(4)));

// this lane's fragment of the 16-row step whose first row starts at `at` (LDS byte address incl. the lane's own offset): rows +0..3 and +4..7 of its half
template <int P>
__device__ __forceinline__ f16x8 wt_fragment(const unsigned char *at) {
    typedef __attribute__((address_space(3))) wt_fp16x4 *lds_ptr;
    const wt_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_ptr)(at));
    const wt_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_ptr)(at + 4 * P));
    return __builtin_shufflevector(__builtin_bit_cast(wt_f16x4, lo), __builtin_bit_cast(wt_f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int TO, int TI>
__global__ __launch_bounds__(kWtThreads, kWtThreads / 256) void k_linear_wgrad_tr(WgArgs a) {
    constexpr int CO = 32 * TO, CI = 32 * TI, PY = wt_pitch(CO), PX = wt_pitch(CI);
    constexpr int RVY = CO / 8, RVX = CI / 8;                                    // 16-byte vectors per row
    constexpr int NVY = (kWtRows * RVY + kWtThreads - 1) / kWtThreads, NVX = (kWtRows * RVX + kWtThreads - 1) / kWtThreads;
    constexpr int NRT = (TO + 3) / 4, NCT = (TI + 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_y[2][kWtRows * PY], s_x[2][kWtRows * PX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5, p = lane & 15, gi = (lane >> 4) & 1;
    const int rg = wave & 3, cg = wave >> 2;
    v16f acc[NRT][NCT];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int k = 0; k < NCT; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][k][e] = 0.0f;
    const uint4 *gy = static_cast<const uint4 *>(a.gy), *gx = static_cast<const uint4 *>(a.x);
    const uint32_t c0 = blockIdx.x * a.chunks_per_wg, c1 = c0 + a.chunks_per_wg < a.n_chunks ? c0 + a.chunks_per_wg : a.n_chunks;
    uint4 ry[NVY], rx[NVX];
    auto load = [&](uint32_t c) {
        const uint32_t m0 = c * (uint32_t)kWtRows;
#pragma unroll
        for (int q = 0; q < NVY; ++q) {
            const int v = q * kWtThreads + tid;
            const uint32_t m = m0 + (uint32_t)(v / RVY);
            ry[q] = (v < kWtRows * RVY && m < a.M) ? gy[(size_t)m * RVY + (uint32_t)(v % RVY)] : uint4{0u, 0u, 0u, 0u};       // rows beyond the matrix: zeros
        }
#pragma unroll
        for (int q = 0; q < NVX; ++q) {
            const int v = q * kWtThreads + tid;
            const uint32_t m = m0 + (uint32_t)(v / RVX);
            rx[q] = (v < kWtRows * RVX && m < a.M) ? gx[(size_t)m * RVX + (uint32_t)(v % RVX)] : uint4{0u, 0u, 0u, 0u};
        }
    };
    auto store = [&](int b) {
#pragma unroll
        for (int q = 0; q < NVY; ++q) {
            const int v = q * kWtThreads + tid;
            if (v < kWtRows * RVY) *reinterpret_cast<uint4 *>(&s_y[b][(v / RVY) * PY + (v % RVY) * 16]) = ry[q];
        }
#pragma unroll
        for (int q = 0; q < NVX; ++q) {
            const int v = q * kWtThreads + tid;
            if (v < kWtRows * RVX) *reinterpret_cast<uint4 *>(&s_x[b][(v / RVX) * PX + (v % RVX) * 16]) = rx[q];
        }
    };
    if (c0 < c1) {
        load(c0);
        store(0);
        if (c0 + 1 < c1) load(c0 + 1);
    }
    __syncthreads();
    // the lane's place in a step's block: row 8 h + p / 4, columns 16 gi + 4 (p % 4) .. of a 32-column tile
    const int off_y = (8 * h + (p >> 2)) * PY + (16 * gi + 4 * (p & 3)) * 2, off_x = (8 * h + (p >> 2)) * PX + (16 * gi + 4 * (p & 3)) * 2;
    for (uint32_t c = c0; c < c1; ++c) {
        const int b = (int)((c - c0) & 1u);
        if (c + 1 < c1) {
            store(b ^ 1);                                          // chunk c + 1 (requested one iteration ago); its buffer was last read before the previous barrier
            if (c + 2 < c1) load(c + 2);
        }
#pragma unroll
        for (int s = 0; s < kWtRows / 16; ++s) {
            f16x8 A[NRT], B[NCT];
#pragma unroll
            for (int r = 0; r < NRT; ++r)
                if (rg + 4 * r < TO) A[r] = wt_fragment<PY>(&s_y[b][0] + off_y + s * 16 * PY + (rg + 4 * r) * 64);
#pragma unroll
            for (int k = 0; k < NCT; ++k)
                if (cg + 2 * k < TI) B[k] = wt_fragment<PX>(&s_x[b][0] + off_x + s * 16 * PX + (cg + 2 * k) * 64);
#pragma unroll
            for (int r = 0; r < NRT; ++r)
#pragma unroll
                for (int k = 0; k < NCT; ++k)
                    if (rg + 4 * r < TO && cg + 2 * k < TI) acc[r][k] = LpTraits<_Float16>::mfma(A[r], B[k], acc[r][k]);
        }
        __syncthreads();
    }
    // lane (j = i, h) holds rows (r & 3) + 8 (r >> 2) + 4 h of column j of every tile
    float *out = a.partial + (size_t)blockIdx.x * CO * CI;
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
        const int rt = rg + 4 * r;
        if (rt >= TO) continue;
#pragma unroll
        for (int k = 0; k < NCT; ++k) {
            const int ct = cg + 2 * k;
            if (ct >= TI) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) out[(size_t)(32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h) * CI + 32 * ct + i] = acc[r][k][e];
        }
    }
}

template <int TO>
static bool wt_pick_in(uint32_t ti, const WgArgs &a, uint32_t slices, hipStream_t st) {
    switch (ti) {
    case 2: hipLaunchKernelGGL((k_linear_wgrad_tr<TO, 2>), dim3(slices), dim3(kWtThreads), 0, st, a); return true;
    case 3: hipLaunchKernelGGL((k_linear_wgrad_tr<TO, 3>), dim3(slices), dim3(kWtThreads), 0, st, a); return true;
    case 4: hipLaunchKernelGGL((k_linear_wgrad_tr<TO, 4>), dim3(slices), dim3(kWtThreads), 0, st, a); return true;
    case 5: hipLaunchKernelGGL((k_linear_wgrad_tr<TO, 5>), dim3(slices), dim3(kWtThreads), 0, st, a); return true;
    }
    return false;
}
static bool wt_pick(uint32_t to, uint32_t ti, const WgArgs &a, uint32_t slices, hipStream_t st) {
    if (to == 1) return wt_pick_in<1>(ti, a, slices, st);
    if (to == 4) return wt_pick_in<4>(ti, a, slices, st);
    if (to == 5) return wt_pick_in<5>(ti, a, slices, st);
    return false;
}

// grad_weight = the sum of the slices, in a FIXED order (a quarter of the slices per wavefront in four chains, the quarters added through LDS): no atomics, no
// zero fill in front of it, and the weight gradient of a step is reproducible bit for bit (round 4 added 32-slice groups into a cleared matrix with device atomics)
__global__ __launch_bounds__(256) void k_linear_wgrad_reduce(const float *__restrict__ partial, uint32_t slices, uint32_t n, float *__restrict__ gw) {
    __shared__ float part[4][64];
    const uint32_t lane = threadIdx.x & 63u, q = threadIdx.x >> 6;
    const uint32_t e = blockIdx.x * 64u + lane;
    const uint32_t per = (slices + 3u) / 4u, w0 = q * per < slices ? q * per : slices, w1 = w0 + per < slices ? w0 + per : slices;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < n) {
        const float *p = partial + e;
        uint32_t w = w0;
        for (; w + 3u < w1; w += 4u) {
            s0 += p[(size_t)w * n];
            s1 += p[(size_t)(w + 1u) * n];
            s2 += p[(size_t)(w + 2u) * n];
            s3 += p[(size_t)(w + 3u) * n];
        }
        for (; w < w1; ++w) s0 += p[(size_t)w * n];
    }
    part[q][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && e < n) gw[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API unsigned long long gfpp_linear_weight_grad_scratch_floats(uint32_t O, uint32_t I) {
    if (O == 0 || I == 0 || div_up(O, 32) > (uint32_t)kWgMaxTO || div_up(I, 32) > (uint32_t)kWgMaxTI) return 0ull;
    return (unsigned long long)kWgMaxSlices * O * I;
}

GFPP_API int gfpp_linear_weight_grad(const void *grad_out, const void *input, uint32_t M, uint32_t O, uint32_t I, int dtype, float *partial, float *grad_weight,
                                     gfpp_stream_t stream) {
    const char *who = "gfpp_linear_weight_grad";
    if (!grad_out || !input || !partial || !grad_weight || M == 0 || O == 0 || I == 0) { set_error("%s: null argument or empty matrix", who); return GFPP_EINVAL; }
    if (dtype != GFPP_F32 && dtype != GFPP_F16) { set_error("%s: dtype must be GFPP_F32 or GFPP_F16", who); return GFPP_EUNSUPPORTED; }
    WgArgs a;
    a.gy = grad_out; a.x = input; a.partial = partial; a.M = M; a.O = O; a.I = I;
    a.TO = div_up(O, 32); a.TI = div_up(I, 32);
    if (a.TO > (uint32_t)kWgMaxTO || a.TI > (uint32_t)kWgMaxTI) { set_error("%s: built for out_features <= 256 and in_features <= 160 (got %u, %u)", who, O, I); return GFPP_EUNSUPPORTED; }
    const hipStream_t st = (hipStream_t)stream;
    if (((uintptr_t)grad_out | (uintptr_t)input) & 15u) { set_error("%s: grad_out and input must be 16-byte aligned", who); return GFPP_EINVAL; }
    // zero-padded half matrices (both widths multiples of 32, the fused MLP's): fragments by transposing LDS reads, one workgroup per CU (gfpp_tuning.wgrad_tr = 0: the
    // generic kernel, its A/B partner)
    const bool tr_off = tuning().wgrad_tr == 0;
    const bool tr = dtype == GFPP_F16 && !tr_off && O % 32 == 0 && I % 32 == 0 && (a.TO == 1 || a.TO == 4 || a.TO == 5) && a.TI >= 2;
    const uint32_t R = tr ? (uint32_t)kWtRows : dtype == GFPP_F16 ? (uint32_t)WgCfg<_Float16>::rows : (uint32_t)WgCfg<float>::rows;
    a.n_chunks = div_up(M, R);
    a.chunks_per_wg = div_up(a.n_chunks, kWgMaxSlices);          // (the transposing kernel: 80 KB of LDS and 82 VGPRs, two workgroups per CU)
    const uint32_t slices = div_up(a.n_chunks, a.chunks_per_wg);
    if (tr) wt_pick(a.TO, a.TI, a, slices, st);
    else if (dtype == GFPP_F16) hipLaunchKernelGGL(k_linear_wgrad<_Float16>, dim3(slices), dim3(kWgThreads), 0, st, a);
    else hipLaunchKernelGGL(k_linear_wgrad<float>, dim3(slices), dim3(kWgThreads), 0, st, a);
    int rc = check_launch(who);
    if (rc) return rc;
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3(div_up(O * I, 64)), dim3(256), 0, st, partial, slices, O * I, grad_weight);
    return check_launch(who);
}
