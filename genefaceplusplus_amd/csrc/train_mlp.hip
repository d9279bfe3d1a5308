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

// grad_weight (zeroed) += the slices, 32 per workgroup row
constexpr uint32_t kWgReduceGroup = 32;
__global__ __launch_bounds__(256) void k_linear_wgrad_reduce(const float *__restrict__ partial, uint32_t slices, uint32_t n, float *__restrict__ gw) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t w0 = blockIdx.y * kWgReduceGroup, w1 = w0 + kWgReduceGroup < slices ? w0 + kWgReduceGroup : slices;
    float s = 0.0f;
    for (uint32_t w = w0; w < w1; ++w) s += partial[(size_t)w * n + e];
    atomicAdd(&gw[e], s);
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_linear_weight_grad(const void *grad_out, const void *input, uint32_t M, uint32_t O, uint32_t I, int dtype, float *partial, float *grad_weight,
                                     gfpp_stream_t stream) {
    const char *who = "gfpp_linear_weight_grad";
    if (!grad_out || !input || !partial || !grad_weight || M == 0 || O == 0 || I == 0) { set_error("%s: null argument or empty matrix", who); return GFPP_EINVAL; }
    if (dtype != GFPP_F32 && dtype != GFPP_F16) { set_error("%s: dtype must be GFPP_F32 or GFPP_F16", who); return GFPP_EUNSUPPORTED; }
    WgArgs a;
    a.gy = grad_out; a.x = input; a.partial = partial; a.M = M; a.O = O; a.I = I;
    a.TO = div_up(O, 32); a.TI = div_up(I, 32);
    if (a.TO > (uint32_t)kWgMaxTO || a.TI > (uint32_t)kWgMaxTI) { set_error("%s: built for out_features <= 256 and in_features <= 160 (got %u, %u)", who, O, I); return GFPP_EUNSUPPORTED; }
    const uint32_t R = dtype == GFPP_F16 ? (uint32_t)WgCfg<_Float16>::rows : (uint32_t)WgCfg<float>::rows;
    a.n_chunks = div_up(M, R);
    a.chunks_per_wg = div_up(a.n_chunks, kWgMaxSlices);
    const uint32_t slices = div_up(a.n_chunks, a.chunks_per_wg);
    const hipStream_t st = (hipStream_t)stream;
    if (((uintptr_t)grad_out | (uintptr_t)input) & 15u) { set_error("%s: grad_out and input must be 16-byte aligned", who); return GFPP_EINVAL; }
    if (hipMemsetAsync(grad_weight, 0, (size_t)O * I * sizeof(float), st) != hipSuccess) { set_error("%s: cannot clear grad_weight", who); return GFPP_EINVAL; }
    if (dtype == GFPP_F16) hipLaunchKernelGGL(k_linear_wgrad<_Float16>, dim3(slices), dim3(kWgThreads), 0, st, a);
    else hipLaunchKernelGGL(k_linear_wgrad<float>, dim3(slices), dim3(kWgThreads), 0, st, a);
    int rc = check_launch(who);
    if (rc) return rc;
    hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3(div_up(O * I, 256), div_up(slices, kWgReduceGroup)), dim3(256), 0, st, partial, slices, O * I, grad_weight);
    return check_launch(who);
}
