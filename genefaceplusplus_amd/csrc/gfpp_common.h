// gfpp_common.h -- shared device helpers and host-side launch checking for libgfpp_radnerf.so (gfx950 only).
//
// Floating-point policy: this library is compiled with -ffp-contract=off.  Wherever the reference source has an
// a*b+c pattern (nvcc fuses those by default) the kernels call fmaf() explicitly, so that the marcher's voxel
// decisions and the grid encoder's lattice indices are a deterministic function of the source, not of the optimiser.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gfpp_radnerf.h"

#define GFPP_API extern "C" __attribute__((visibility("default")))

namespace gfpp {

void set_error(const char *fmt, ...);
// the launch tuning of the library (gfpp_set_tuning, raymarch.hip): read at every issue, never getenv()
const gfpp_tuning &tuning();
// compute units of the current device, asked for once per process (round-5 advisory: several launch paths queried the runtime at every launch)
inline int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- Morton code: x -> bit 0, y -> bit 1, z -> bit 2 of every 3-bit group ------------------------------------
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
// the same code for coordinates below 256: the 10-bit spread's first stage is the identity (march_device.h: the marcher's probe of one-cascade models)
__device__ __forceinline__ uint32_t spread3_8(uint32_t v) {
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3_8(uint32_t x, uint32_t y, uint32_t z) {
    return (spread3_8(z) << 2) | ((spread3_8(y) << 1) | spread3_8(x));
}
__device__ __forceinline__ uint32_t compact3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

}  // namespace gfpp
