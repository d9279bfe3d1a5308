// head_eval_device.h -- device helpers shared by the fp32 (frame_head.hip) and 16-bit (frame_head_lp.hip) head trip kernels:
// grid-encoder halves, fragment-order bias load, accumulator -> operand moves, skinny VALU output rows, wave scan.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>

#include "grid_device.h"
#include "march_device.h"
#include "sh_device.h"
#include "lp_mfma_device.h"   // v16f

namespace gfpp {


constexpr int kTile = 128;     // sample slots per workgroup tile (4 wavefronts x 32 columns)
constexpr int kThreads = 256;
constexpr int kMaxTrips = 63;      // counters[0..63] alive counts, counters[64..127] evaluated samples
constexpr int kCounterWords = 192; // gfpp_frame_ws.counters: + [128..191] the budget array of the persistent 16-bit launch (histogram, sample / round counts)
constexpr int kBudgetBase = 128, kBudgetSamples = 40, kBudgetRounds = 41, kBudgetRoundsMax = 42, kBudgetSamplesMax = 43, kBudgetCycles = 44;   // [44..47] phase cycles / 1024 (sums), [48] longest workgroup, [49] the ingest steps' share of [44]

struct GridDev {
    const void *table;
    const gfpp_grid_level *levels;
    uint32_t gridtype, interp, align_corners;
    uint32_t fast;     // no level needs the hash / a true modulo and the table is the padded copy: the straight-line lookup (level_fast_issue / _finish)
};

inline GridDev make_grid_dev(const gfpp_grid_desc &d) {
    uint32_t slow = 0;
    if (d.levels_host)
        for (uint32_t l = 0; l < d.L && l < 32; ++l) slow |= d.levels_host[l].flags & GFPP_LEVEL_SLOW;
    return GridDev{d.table, d.levels, d.gridtype, d.interp, d.align_corners, (d.levels_host && !slow && d.row_padded) ? 1u : 0u};
}

__device__ __forceinline__ void load_bias(v16f (&acc)[4], const float *__restrict__ bias_frag, int hi) {
    const float4 *p = reinterpret_cast<const float4 *>(bias_frag + hi * 64);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = p[m * 4 + q];
            acc[m][4 * q] = v.x; acc[m][4 * q + 1] = v.y; acc[m][4 * q + 2] = v.z; acc[m][4 * q + 3] = v.w;
        }
    }
}

__device__ __forceinline__ void zero_acc(v16f (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
}

template <bool RELU>
__device__ __forceinline__ void acc_to_b(const v16f (&acc)[4], float (&b)[64]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) b[m * 16 + r] = RELU ? fmaxf(acc[m][r], 0.0f) : acc[m][r];
}

// Skinny output layer on the VALU: out[c] = sum over this lane's 64 activations, then the two half-waves are added.
template <int C>
__device__ __forceinline__ void valu_rows(const float *__restrict__ wv, const float (&b)[64], int hi, float (&out)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float4 *p = reinterpret_cast<const float4 *>(wv + (hi * C + c) * 64);
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = p[q];
            s = fmaf(v.x, b[4 * q], s); s = fmaf(v.y, b[4 * q + 1], s); s = fmaf(v.z, b[4 * q + 2], s); s = fmaf(v.w, b[4 * q + 3], s);
        }
        out[c] = s + __shfl_xor(s, 32);
    }
}

// This lane's half of a 16-level, 2-channel grid encoding: levels hi*8 .. hi*8+7 -> 16 features.
template <int D>
__device__ __forceinline__ void encode_half(const float (&u)[D], const GridDev &g, int hi, bool valid, float (&f)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 0.0f;
    if (!valid) return;
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) inside = inside & !(u[d] < 0.0f || u[d] > 1.0f);
    if (!inside) return;
    if (g.fast) {
        // (wave-uniform) the straight-line lookup of the 16-bit kernels -- same corner order, weight products and fma chain as grid_level_lookup, so the
        // same bits -- with the gathers of two levels in flight before the first is interpolated; the generic lookup below waits level by level
        const float *table = reinterpret_cast<const float *>(g.table);
        const bool ac = g.align_corners != 0;
        auto levels = [&](auto smooth_tag) {
            constexpr bool SM = decltype(smooth_tag)::value;
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 2) {
                LevelGathers<D> lg[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const gfpp_grid_level &d = g.levels[hi * 8 + i0 + k];
                    level_fast_issue<D, SM>(u, table, LevelU{d.scale, d.sy, d.sz, d.mask, d.offset}, ac, lg[k]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float o[2];
                    level_fast_finish<D>(lg[k], o);
                    f[2 * (i0 + k)] = o[0];
                    f[2 * (i0 + k) + 1] = o[1];
                }
            }
        };
        if (g.interp == 1) levels(std::true_type{}); else levels(std::false_type{});
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const gfpp_grid_level lv = g.levels[hi * 8 + i];
        float o[2];
        grid_level_lookup<D, 2, float>(u, reinterpret_cast<const float *>(g.table), lv.offset, lv.size, lv.scale, lv.resolution, g.gridtype,
                                       g.align_corners != 0, g.interp, o);
        f[2 * i] = o[0];
        f[2 * i + 1] = o[1];
    }
}

// LDS traffic of one wavefront is executed in program order; this only stops the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

}  // namespace gfpp
