// sh_device.h -- view-direction and positional encodings, device side.
//   sh_basis4   : real spherical harmonics up to degree 4 (16 values); constants and sign conventions as the
//                 reference's kernel_sh (modules/radnerfs/encoders/shencoder/src/shencoder.cu:44-68).
//   freq_feature: one sin/cos feature of the frequency encoding (encoders/freqencoder/src/freqencoder.cu:46-56):
//                 column `col` = 2*octave + is_cos, value sin(x * 2^octave + is_cos * pi/2).
//                 The reference uses the __sinf fast intrinsic; we use the correctly-rounded-ish ocml sinf
//                 (documented tolerance against either).
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

__device__ __forceinline__ float freq_feature(float x, uint32_t col) {
    const uint32_t octave = col >> 1;
    const float phase = (float)(col & 1u) * (3.141592653589793f / 2);
    return sinf(scalbnf(x, (int)octave) + phase);
}

}  // namespace gfpp
