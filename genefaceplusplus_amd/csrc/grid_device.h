// grid_device.h -- multiresolution hash / tiled grid lookup, device side.  Shared by the stand-alone encoder kernel
// (encoders.hip) and the fused head / torso kernels.
//
// Index semantics follow the reference's get_grid_index / fast_hash / kernel_grid
// (modules/radnerfs/encoders/gridencoder/src/gridencoder.cu:50-84, 137-190): uint32 wrap-around arithmetic, the
// `stride <= hashmap_size` early exit that drops trailing dimensions at fine tiled levels, modulo by the
// (8-rounded) level size.  Per-level scale/resolution are computed ONCE on the host (exp2f + ceil in fp32) and passed
// by value, instead of every thread re-deriving them with a device exp2f of unspecified rounding.
#pragma once

#include <hip/hip_fp16.h>

#include "gfpp_common.h"

namespace gfpp {

constexpr int kMaxLevels = 32;

struct GridLevels {
    float scale[kMaxLevels];        // exp2f(level*S)*H - 1
    uint32_t resolution[kMaxLevels];  // ceil(scale) + 1
    uint32_t offset[kMaxLevels];      // first table row of the level
    uint32_t size[kMaxLevels];        // rows in the level (hashmap_size)
    uint32_t L;
};

// Host: fill scale / resolution (offset/size come from the device-side offsets array or a host copy).
inline void fill_level_scales(GridLevels &g, uint32_t L, float S, uint32_t H) {
    g.L = L;
    for (uint32_t l = 0; l < L; ++l) {
        const float sc = fmaf(exp2f((float)l * S), (float)H, -1.0f);
        g.scale[l] = sc;
        g.resolution[l] = (uint32_t)ceil((double)sc) + 1u;
    }
}

template <int D>
__device__ __forceinline__ uint32_t grid_row(const uint32_t (&pg)[D], uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                             uint32_t resolution) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (stride <= hashmap_size) {
            index += pg[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1u);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) index ^= pg[d] * primes[d];
    }
    return index % hashmap_size;
}

template <typename T>
struct TableIO;
template <>
struct TableIO<float> {
    template <int C>
    static __device__ __forceinline__ void load(const float *row, float (&v)[C]) {
        if constexpr (C == 2) { const float2 t = *reinterpret_cast<const float2 *>(row); v[0] = t.x; v[1] = t.y; }
        else if constexpr (C == 4) { const float4 t = *reinterpret_cast<const float4 *>(row); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else {
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = row[c];
        }
    }
};
template <>
struct TableIO<__half> {
    template <int C>
    static __device__ __forceinline__ void load(const __half *row, float (&v)[C]) {
        if constexpr (C == 2) { const float2 t = __half22float2(*reinterpret_cast<const __half2 *>(row)); v[0] = t.x; v[1] = t.y; }
        else {
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = __half2float(row[c]);
        }
    }
};

// Two x-adjacent table rows in one load.  Rows r and r+1 are contiguous in memory (C values each); the pair is only
// C*sizeof(T)-aligned, so the vector type carries a reduced alignment (gfx950 global loads handle it natively).
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef _Float16 f16x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <int C, typename T>
__device__ __forceinline__ void load_row_pair(const T *__restrict__ p, float (&lo)[C], float (&hi)[C]) {
    if constexpr (C == 2 && sizeof(T) == 4) {
        const f32x4_a8 v = *reinterpret_cast<const f32x4_a8 *>(p);
        lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3];
    } else if constexpr (C == 1 && sizeof(T) == 4) {
        const f32x2_a4 v = *reinterpret_cast<const f32x2_a4 *>(p);
        lo[0] = v[0]; hi[0] = v[1];
    } else if constexpr (C == 2 && sizeof(T) == 2) {
        const f16x4_a4 v = *reinterpret_cast<const f16x4_a4 *>(p);
        lo[0] = (float)v[0]; lo[1] = (float)v[1]; hi[0] = (float)v[2]; hi[1] = (float)v[3];
    } else {
        TableIO<T>::template load<C>(p, lo);
        TableIO<T>::template load<C>(p + C, hi);
    }
}

// d-linear (or smoothstep) interpolation of one level for one point; `u` in [0,1]^D.  Accumulates in fp32
// (for f16 tables the reference accumulates in half, gridencoder.cu:163 -- documented tolerance, not bit parity).
//
// The 2^D corners are visited in the reference's order (corner bit d selects +1 along dimension d), but the two corners
// that differ only in x are fetched together: dimension 0 always enters the linear index with stride 1
// (gridencoder.cu:70-75), so unless the level is hashed or the row wraps at the end of the level they are adjacent
// table rows -- one 16-byte gather instead of two 8-byte ones, which halves the number of cache lines the texture
// path has to look up (the bound of this kernel is tag-lookup rate, not bytes).
// OFF32: rows are addressed by 32-bit byte offsets against the (wave-uniform) table pointer -- for callers whose tables are known to be below 4 GiB (the fused
// head kernels: 16 levels x 2 channels); a 64-bit per-lane address per level cost them spilled registers.
template <int D, int C, typename T, bool OFF32 = false>
__device__ __forceinline__ void grid_level_lookup(const float (&u)[D], const T *__restrict__ table, uint32_t level_offset,
                                                  uint32_t hashmap_size, float scale, uint32_t resolution, uint32_t gridtype,
                                                  bool align_corners, uint32_t interp, float (&out)[C]) {
    float frac[D];
    uint32_t base[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(u[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(pos);
        base[d] = (uint32_t)fl;
        float f = pos - (float)base[d];
        if (interp == 1) f = f * f * fmaf(-2.0f, f, 3.0f);
        frac[d] = f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = 0.0f;
    auto row_ptr = [&](uint32_t row) -> const T * {
        if constexpr (OFF32) return reinterpret_cast<const T *>(reinterpret_cast<const char *>(table) + (level_offset + row) * (uint32_t)(C * sizeof(T)));
        else return table + ((size_t)level_offset + row) * C;
    };

    // is the level addressed by the hash?  (stride after all D dimensions > level size, hash grid type)
    // does the tiled index drop the last dimension?  (stride already > level size before dimension D-1 is reached:
    // gridencoder.cu:72 stops the loop, so corners that differ only in that dimension alias the same row)
    uint32_t stride = 1;
    bool last_dim_dropped = false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (stride <= hashmap_size) stride *= align_corners ? resolution : (resolution + 1u);
        else if (d == D - 1) last_dim_dropped = true;
    }
    const bool hashed = gridtype == 0 && stride > hashmap_size;
    const bool reuse_upper = D == 3 && last_dim_dropped && !hashed;

    constexpr int kPairs = 1 << (D - 1);
    float keep0[kPairs / 2 > 0 ? kPairs / 2 : 1][C], keep1[kPairs / 2 > 0 ? kPairs / 2 : 1][C];
#pragma unroll
    for (int pair = 0; pair < kPairs; ++pair) {
        // weights of the two corners (x bit clear / set), products in the reference's order d = 0, 1, ...
        float w0 = 1.0f - frac[0], w1 = frac[0];
        uint32_t pg[D];
        pg[0] = base[0];
#pragma unroll
        for (int d = 1; d < D; ++d) {
            if (pair & (1 << (d - 1))) { w0 *= frac[d]; w1 *= frac[d]; pg[d] = base[d] + 1u; }
            else { w0 *= 1.0f - frac[d]; w1 *= 1.0f - frac[d]; pg[d] = base[d]; }
        }
        float v0[C], v1[C];
        const bool upper = D == 3 && pair >= kPairs / 2;
        if (upper && reuse_upper) {
            // same rows as pair - kPairs/2 (z ignored by the index): reuse the values, keep the accumulation order
#pragma unroll
            for (int c = 0; c < C; ++c) { v0[c] = keep0[pair - kPairs / 2][c]; v1[c] = keep1[pair - kPairs / 2][c]; }
        } else {
            const uint32_t row0 = grid_row<D>(pg, gridtype, align_corners, hashmap_size, resolution);
            if (!hashed && row0 + 1u < hashmap_size) {
                load_row_pair<C, T>(row_ptr(row0), v0, v1);
            } else {
                pg[0] = base[0] + 1u;
                const uint32_t row1 = grid_row<D>(pg, gridtype, align_corners, hashmap_size, resolution);
                TableIO<T>::template load<C>(row_ptr(row0), v0);
                TableIO<T>::template load<C>(row_ptr(row1), v1);
            }
            if (D == 3 && !upper) {
#pragma unroll
                for (int c = 0; c < C; ++c) { keep0[pair][c] = v0[c]; keep1[pair][c] = v1[c]; }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = fmaf(w1, v1[c], fmaf(w0, v0[c], out[c]));
    }
}

// ---- resolved index arithmetic for the fused kernels --------------------------------------------------------------------
// gfpp_grid_level carries, besides the reference's per-level quantities, the index arithmetic resolved on the host
// (gfpp_grid_levels_fill): sy / sz = stride of the y / z coordinate in the level's linear index, 0 where the reference's
// `stride <= hashmap_size` test (gridencoder.cu:72) drops the dimension; mask = size - 1 where the level size is a power
// of two, ~0 where the index provably stays below the size, so `index % hashmap_size` (gridencoder.cu:82) is one AND.
// Levels that need the hash (or a true modulo) are flagged GFPP_LEVEL_SLOW and go through grid_level_lookup above.
// The user of these fields is level_fast_uniform in frame_head_lp.hip.

// ---- grid encoding, straight-line ---------------------------------------------------------------------------------------
// One level of the lookup with the index arithmetic resolved on the host (gfpp_grid_levels_fill) and no branch: tables are the
// per-level padded copy (row `size` repeats row 0), so the x+1 neighbour of the last row needs no wrap-around case, and a dropped
// z coordinate (sz == 0) simply fetches the same rows again.  Same corner order, weight products and fma chain as
// grid_level_lookup => bit-identical features.
struct LevelU {   // one level's descriptor in scalar registers
    float scale;
    uint32_t sy, sz, mask, offset;
};

// (two halves: level_fast_issue puts the level's 2^(D-1) gathers in flight and keeps the fractions, level_fast_finish interpolates -- so that a caller
// can issue the gathers of several levels before it consumes the first: left as one function, the compiler waits for a level's four gathers before it
// even computes the next level's addresses, i.e. sixteen dependent memory round trips per block and grid instead of sixteen overlapped ones)
template <int D>
struct LevelGathers {
    float frac[D];
    f32x4_a8 v[1 << (D - 1)];
};

template <int D, bool SMOOTH>
__device__ __forceinline__ void level_fast_issue(const float (&u)[D], const float *__restrict__ table, const LevelU &lv, bool align_corners,
                                                 LevelGathers<D> &g) {
    uint32_t base[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        // pos >= 0 here (u is clamped to [0,1]): truncation IS floor, and v_fract_f32 returns pos - floor(pos) exactly (the subtraction is exact
        // in fp32 and < 1) -- two instructions instead of floor + convert + subtract, same bits
        const float pos = fmaf(u[d], lv.scale, align_corners ? 0.0f : 0.5f);
        base[d] = (uint32_t)pos;
        float f = __builtin_amdgcn_fractf(pos);
        if constexpr (SMOOTH) f = f * f * fmaf(-2.0f, f, 3.0f);
        g.frac[d] = f;
    }
    // byte offsets in 32 bits against the wave-uniform table pointer (global_load with an SGPR base): one v_add_lshl_u32 per gather instead of a
    // 64-bit per-lane address; the padded tables are far below 4 GiB
    const char *lt = reinterpret_cast<const char *>(table);
    const uint32_t y0 = __umul24(base[1], lv.sy), y1 = y0 + lv.sy;
    uint32_t z0 = 0, z1 = 0;
    if constexpr (D == 3) { z0 = __umul24(base[2], lv.sz); z1 = z0 + lv.sz; }
    constexpr int kPairs = 1 << (D - 1);
#pragma unroll
    for (int pair = 0; pair < kPairs; ++pair) {
        uint32_t row = base[0] + ((pair & 1) ? y1 : y0);
        if constexpr (D == 3) row += (pair & 2) ? z1 : z0;   // sz == 0 (z dropped by the tiled index): the same rows again, an L1 hit
        row &= lv.mask;
        g.v[pair] = *reinterpret_cast<const f32x4_a8 *>(lt + ((row + lv.offset) << 3));
    }
}

template <int D>
__device__ __forceinline__ void level_fast_finish(const LevelGathers<D> &g, float (&out)[2]) {
    // corner weights and accumulation on packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32): (w0, w1) = ((1 - fx) * yf * zf, fx * yf * zf) with the
    // products in grid_level_lookup's order, so the (1 - fx, fx) * yf halves are shared by the two z planes -- 6 packed multiplies per 3-D level
    // instead of 16 scalar ones, same bits
    constexpr int kPairs = 1 << (D - 1);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 wx = {1.0f - g.frac[0], g.frac[0]};
    f32x2 wxy[2] = {wx * (1.0f - g.frac[1]), wx * g.frac[1]};
    f32x2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int pair = 0; pair < kPairs; ++pair) {
        f32x2 w = wxy[pair & 1];
        if constexpr (D == 3) w = w * ((pair & 2) ? g.frac[2] : 1.0f - g.frac[2]);
        const f32x2 c0 = {g.v[pair][0], g.v[pair][1]}, c1 = {g.v[pair][2], g.v[pair][3]};
        acc = __builtin_elementwise_fma(f32x2{w[0], w[0]}, c0, acc);
        acc = __builtin_elementwise_fma(f32x2{w[1], w[1]}, c1, acc);
    }
    out[0] = acc[0];
    out[1] = acc[1];
}


// ---- corner-block tables (the 16-bit head kernels of frame_head_lp.hip since round 4; gfpp_head_model.pos_grid_blk / amb_grid_blk) ------------------------
// The straight-line lookup above fetches a level's 2^D corners with 2^(D-1) gathers of 16 bytes from 2^(D-1) different cache lines (rows r and r + 1 are
// adjacent, the y and z neighbours are whole strides away) -- 64 gathers from 64 lines per sample and 16-level 3-D grid, and the vector L1's tag rate is what a
// lane-divergent gather costs.  A corner-block table trades memory for that: row r of a level holds, in 16-bit floats, BOTH channels of the FOUR corners
// (r, r + 1, r + sy, r + sy + 1) of the x-y cell that starts at r (index arithmetic is linear modulo the level size, so the neighbours are functions of r
// alone) -- 16 bytes, 16-byte aligned: one gather per z plane.  A 3-D level is 2 gathers from 2 lines (1 line where the tiled index drops z: the same row
// twice), a 2-D level is 1.  The reference's own autocast path reads a half table too (grid.py:43-47); the table is 2 x the fp32 one (16 B instead of 8 B per row).
// Row layout (8 halves): c0(x,y) c0(x+1,y) | c1(x,y) c1(x+1,y) | c0(x,y+1) c0(x+1,y+1) | c1(x,y+1) c1(x+1,y+1): a dword is one channel's x pair, the operand
// of a packed dot product with the pair of corner weights (v_dot2_f32_f16: 16-bit products, fp32 accumulation).
typedef _Float16 gfpp_h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));

template <int D>
struct BlockGathers {
    float frac[D];
    u32x4_a16 v[D == 3 ? 2 : 1];
};

template <int D, bool SMOOTH>
__device__ __forceinline__ void level_block_issue(const float (&u)[D], const void *__restrict__ table, const LevelU &lv, bool align_corners,
                                                  BlockGathers<D> &g) {
    uint32_t base[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(u[d], lv.scale, align_corners ? 0.0f : 0.5f);
        base[d] = (uint32_t)pos;
        float f = __builtin_amdgcn_fractf(pos);
        if constexpr (SMOOTH) f = f * f * fmaf(-2.0f, f, 3.0f);
        g.frac[d] = f;
    }
    const char *lt = reinterpret_cast<const char *>(table);
    uint32_t row = base[0] + __umul24(base[1], lv.sy);
    if constexpr (D == 3) row += __umul24(base[2], lv.sz);
    row &= lv.mask;
    // (measured alternative, dropped: 32-byte x-y-z rows for the levels that keep z -- one cache line per level, tables 3 x the fp32 bytes: head pass 0.256 ms
    // against 0.263 for these rows, the same frames/s)
    g.v[0] = *reinterpret_cast<const u32x4_a16 *>(lt + ((row + lv.offset) << 4));
    if constexpr (D == 3) {
        const uint32_t row1 = (row + lv.sz) & lv.mask;          // sz == 0 (z dropped by the tiled index): the same row again, an L1 hit
        g.v[1] = *reinterpret_cast<const u32x4_a16 *>(lt + ((row1 + lv.offset) << 4));
    }
}

template <int D>
__device__ __forceinline__ void level_block_finish(const BlockGathers<D> &g, float (&out)[2]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 wx = {1.0f - g.frac[0], g.frac[0]};
    // the four x-y corner weights of the cell once, in fp16 (they are the same in both z planes); the z interpolation happens on the two planes' fp32 sums:
    // 2 packed multiplies + 2 conversions + 8 dot products + 2 packed blend operations per 3-D level (6 + 4 + 8 with the z weight folded into the corner weights)
    const gfpp_h2 h0 = __builtin_convertvector(wx * (1.0f - g.frac[1]), gfpp_h2), h1 = __builtin_convertvector(wx * g.frac[1], gfpp_h2);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    f32x2 plane[D == 3 ? 2 : 1];
#pragma unroll
    for (int z = 0; z < (D == 3 ? 2 : 1); ++z) {
        // (pairs taken with shufflevector from the whole row: bit-casting single vector ELEMENTS to pairs is miscompiled by ROCm 7.2's clang -- every
        // element collapses to element 0, see skinny_dot)
        const h8 row = __builtin_bit_cast(h8, g.v[z]);
        const gfpp_h2 c0y0 = __builtin_shufflevector(row, row, 0, 1), c1y0 = __builtin_shufflevector(row, row, 2, 3);
        const gfpp_h2 c0y1 = __builtin_shufflevector(row, row, 4, 5), c1y1 = __builtin_shufflevector(row, row, 6, 7);
        float o0 = __builtin_amdgcn_fdot2(c0y0, h0, 0.0f, false), o1 = __builtin_amdgcn_fdot2(c1y0, h0, 0.0f, false);
        o0 = __builtin_amdgcn_fdot2(c0y1, h1, o0, false);
        o1 = __builtin_amdgcn_fdot2(c1y1, h1, o1, false);
        plane[z] = f32x2{o0, o1};
    }
    f32x2 o = plane[0];
    if constexpr (D == 3) o = __builtin_elementwise_fma(f32x2{g.frac[2], g.frac[2]}, plane[1] - plane[0], plane[0]);      // p0 + fz (p1 - p0)
    out[0] = o[0];
    out[1] = o[1];
}

}  // namespace gfpp
