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

// d-linear (or smoothstep) interpolation of one level for one point; `u` in [0,1]^D.  Accumulates in fp32
// (for f16 tables the reference accumulates in half, gridencoder.cu:163 -- documented tolerance, not bit parity).
template <int D, int C, typename T>
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
    const T *level_table = table + (size_t)level_offset * C;
#pragma unroll
    for (int corner = 0; corner < (1 << D); ++corner) {
        float w = 1.0f;
        uint32_t pg[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (corner & (1 << d)) { w *= frac[d]; pg[d] = base[d] + 1u; }
            else { w *= 1.0f - frac[d]; pg[d] = base[d]; }
        }
        const uint32_t row = grid_row<D>(pg, gridtype, align_corners, hashmap_size, resolution);
        float v[C];
        TableIO<T>::template load<C>(level_table + (size_t)row * C, v);
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = fmaf(w, v[c], out[c]);
    }
}

}  // namespace gfpp
