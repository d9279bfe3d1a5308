// sh_device.h -- view-direction and positional encodings, device side.
//   sh_basis4   : real spherical harmonics up to degree 4 (16 values); constants and sign conventions as the
//                 reference's kernel_sh (modules/radnerfs/encoders/shencoder/src/shencoder.cu:44-68).
//   freq_feature: one sin/cos feature of the frequency encoding (encoders/freqencoder/src/freqencoder.cu:46-56):
//                 column `col` = 2*octave + is_cos, value sin(x * 2^octave + is_cos * pi/2).
//                 The reference uses the __sinf fast intrinsic; the fp32 paths use the correctly-rounded-ish ocml sinf
//                 (documented tolerance against either), the 16-bit torso kernel the hardware sine (freq_feature_fast).
#pragma once

#include "gfpp_common.h"

namespace gfpp {

__device__ __forceinline__ void sh_basis4(float x, float y, float z, float (&o)[16]) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = fmaf(0.94617469575755997f, z2, -0.31539156525251999f);
    o[7] = -1.0925484305920792f * xz;
    o[8] = fmaf(0.54627421529603959f, x2, -(0.54627421529603959f * y2));
    o[9] = 0.59004358992664352f * y * fmaf(-3.0f, x2, y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * fmaf(-5.0f, z2, 1.0f);
    o[12] = 0.3731763325901154f * z * fmaf(5.0f, z2, -3.0f);
    o[13] = 0.45704579946446572f * x * fmaf(-5.0f, z2, 1.0f);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * fmaf(3.0f, y2, -x2);
}

// d sh_basis4 / d(x, y, z), the three Cartesian coordinates treated as independent variables (the reference's write_sh_dx/dy/dz,
// shencoder.cu:128-352, does the same): term-by-term derivatives of the polynomials above.
__device__ __forceinline__ void sh_basis4_grad(float x, float y, float z, float (&dx)[16], float (&dy)[16], float (&dz)[16]) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float a = 0.48860251190291987f, b = 1.0925484305920792f, f3 = 1.7701307697799306f, f6 = 3.5402615395598611f;
    const float g = 2.8906114426405538f, h = 0.45704579946446572f, m = 1.4453057213202769f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dx[i] = 0.0f; dy[i] = 0.0f; dz[i] = 0.0f; }
    dy[1] = -a;
    dz[2] = a;
    dx[3] = -a;
    dx[4] = b * y;  dy[4] = b * x;
    dy[5] = -b * z; dz[5] = -b * y;
    dz[6] = 1.8923493915151199f * z;
    dx[7] = -b * z; dz[7] = -b * x;
    dx[8] = b * x;  dy[8] = -b * y;
    dx[9] = -f6 * xy;            dy[9] = f3 * (y2 - x2);
    dx[10] = g * yz;             dy[10] = g * xz;              dz[10] = g * xy;
    dy[11] = h * fmaf(-5.0f, z2, 1.0f);                        dz[11] = -10.0f * h * yz;
    dz[12] = fmaf(5.597644988851731f, z2, -1.1195289977703462f);
    dx[13] = h * fmaf(-5.0f, z2, 1.0f);                        dz[13] = -10.0f * h * xz;
    dx[14] = g * xz;             dy[14] = -g * yz;             dz[14] = m * (x2 - y2);
    dx[15] = f3 * (y2 - x2);     dy[15] = f6 * xy;
}

__device__ __forceinline__ float freq_feature(float x, uint32_t col) {
    const uint32_t octave = col >> 1;
    const float phase = (float)(col & 1u) * (3.141592653589793f / 2);
    return sinf(scalbnf(x, (int)octave) + phase);
}

// The same feature on the hardware sine (v_sin_f32 on the argument in revolutions -- what the reference's __sinf compiles to on its GPU, freqencoder.cu:56), for
// features that become 16-bit MFMA operands: |argument| <= 2^9 here, absolute error ~1e-5, two orders of magnitude below the operand rounding.  ocml's sinf
// carries a Payne-Hanek reduction for large arguments that the compiler inlines at every call: ~140 vector instructions per feature against 4.
__device__ __forceinline__ float freq_feature_fast(float x, uint32_t col) {
    const uint32_t octave = col >> 1;
    const float phase = (float)(col & 1u) * (3.141592653589793f / 2);
    return __sinf(scalbnf(x, (int)octave) + phase);
}

}  // namespace gfpp
