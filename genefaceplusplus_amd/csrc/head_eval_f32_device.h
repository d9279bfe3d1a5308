// head_eval_f32_device.h -- the exact-fp32 evaluation of one 32-sample block (RADNeRF.forward, radnerf.py:108-141) on v_mfma_f32_32x32x2_f32, its weight view and
// the argument record of the fp32 trip kernels.  Shared by frame_head.hip (one launch per trip: k_head_trip / _w / _wp, k_head_eval) and frame_head_lp.hip (the
// whole frame as one persistent launch: k_head_frame_persist<AMB_D, float, ...>) -- the same function, so the two are bit-identical per sample.
#pragma once

#include "head_eval_device.h"

namespace gfpp {

struct HeadWeights {
    const float4 *amb_w0, *amb_w1, *sig_w0, *sig_w1, *sig_w2_geo, *col_w0;
    const float *amb_w2, *sig_w2_sig, *col_w1;
};

struct TripArgs {
    MarchParams mp;
    GridDev pos, amb;
    HeadWeights w;
    const uint8_t *bitfield;
    const float *rays_o, *rays_d, *fars;
    float *state;               // [N, kRayRec] ray records (march_device.h)
    const int32_t *alive_in;
    int32_t *alive_out;
    int32_t *counters;
    const int32_t *gcounters;   // frame-wide alive counts per trip (== counters unless this launch renders one ray tile of a shared frame)
    uint32_t N_global;          // rays of the whole frame (== N on one GPU)
    const float *frame_consts;  // [0,128): ambient bias frag, [128,256): colour bias frag
    float T_thresh, density_scale;
    uint32_t N, trip, max_steps;
    // wave-autonomous kernel only (k_head_trip_w): the frame's pre-marched sample lists (k_premarch in frame_head_lp.hip)
    const float *sample_t;
    const uint32_t *sample_cnt;
    uint32_t sample_stride;
    // per-sample evaluation entry only (k_head_eval): tanh(ambient_net) of sample c goes to dbg_ambient[c * AMB_D ...]
    float *dbg_ambient;
};

__device__ __forceinline__ v16f mfma32(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// acc[m] (4 output tiles of 32 rows) += W_packed * b ;  NS rank-2 steps, b[s] is this lane's B element of step s.
// Weight fragments stream from L2 one quad (4 steps x 4 tiles = 16 MFMAs ~ 1k cycles) ahead of their use; the
// sched_barrier keeps the compiler from hoisting a whole layer's loads (which would spill the accumulators).
template <int NS>
__device__ __forceinline__ void mfma_layer(v16f (&acc)[4], const float4 *__restrict__ w, const float (&b)[NS], int lane) {
    static_assert(NS % 4 == 0, "steps are packed in quads");
    const uint32_t l = (uint32_t)lane;   // uniform base (SGPR pair) + 32-bit lane offset => global_load saddr form
    float4 a0 = w[l], a1 = w[64u + l], a2 = w[128u + l], a3 = w[192u + l];
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) {
        float4 n0 = a0, n1 = a1, n2 = a2, n3 = a3;
        if (q + 1 < NS / 4) {
            const float4 *wq = w + (q + 1) * 256;
            n0 = wq[l];
            n1 = wq[64u + l];
            n2 = wq[128u + l];
            n3 = wq[192u + l];
        }
        acc[0] = mfma32(a0.x, b[4 * q], acc[0]); acc[1] = mfma32(a1.x, b[4 * q], acc[1]);
        acc[2] = mfma32(a2.x, b[4 * q], acc[2]); acc[3] = mfma32(a3.x, b[4 * q], acc[3]);
        acc[0] = mfma32(a0.y, b[4 * q + 1], acc[0]); acc[1] = mfma32(a1.y, b[4 * q + 1], acc[1]);
        acc[2] = mfma32(a2.y, b[4 * q + 1], acc[2]); acc[3] = mfma32(a3.y, b[4 * q + 1], acc[3]);
        acc[0] = mfma32(a0.z, b[4 * q + 2], acc[0]); acc[1] = mfma32(a1.z, b[4 * q + 2], acc[1]);
        acc[2] = mfma32(a2.z, b[4 * q + 2], acc[2]); acc[3] = mfma32(a3.z, b[4 * q + 2], acc[3]);
        acc[0] = mfma32(a0.w, b[4 * q + 3], acc[0]); acc[1] = mfma32(a1.w, b[4 * q + 3], acc[1]);
        acc[2] = mfma32(a2.w, b[4 * q + 3], acc[2]); acc[3] = mfma32(a3.w, b[4 * q + 3], acc[3]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1; a2 = n2; a3 = n3;
    }
}

// Evaluate RADNeRF.forward for the 32 occupied samples [first, first+32) of the tile.
template <int AMB_D, typename Tile, bool DBG = false>
__device__ __forceinline__ void evaluate_block(const TripArgs &a, Tile &sh, uint32_t first, uint32_t n_step, int lane_in) {
    int lane = lane_in;
    // launder the lane id: keeps the (tile-loop-invariant) per-lane weight addresses from being hoisted out of the
    // tile loop and spilled
    asm volatile("" : "+v"(lane));
    const int j = lane & 31, hi = lane >> 5;
    const uint32_t c = first + j;
    const bool valid = c < sh.n_valid;
    const uint32_t slot = valid ? sh.order[c] : 0u;
    const uint32_t ray_local = slot / n_step;

    // ---- position grid (each half-wave does 8 of the 16 levels) ------------------------------------------------
    float u3[3];
    const float b2 = 2.0f * a.mp.bound;
    u3[0] = (sh.px[slot] + a.mp.bound) / b2;
    u3[1] = (sh.py[slot] + a.mp.bound) / b2;
    u3[2] = (sh.pz[slot] + a.mp.bound) / b2;
    float fpos[16];
    encode_half<3>(u3, a.pos, hi, valid, fpos);

    v16f acc[4];
    float bs[64];

    // ---- ambient net -----------------------------------------------------------------------------------------
    load_bias(acc, a.frame_consts, hi);
    mfma_layer<16>(acc, a.w.amb_w0, fpos, lane);
    acc_to_b<true>(acc, bs);
    zero_acc(acc);
    mfma_layer<64>(acc, a.w.amb_w1, bs, lane);
    acc_to_b<true>(acc, bs);
    float amb[AMB_D];
    valu_rows<AMB_D>(a.w.amb_w2, bs, hi, amb);
    float ua[AMB_D];
#pragma unroll
    for (int d = 0; d < AMB_D; ++d) {
        const float th = tanhf(amb[d]);
        if constexpr (DBG) { if (a.dbg_ambient && valid && hi == 0) a.dbg_ambient[(size_t)c * AMB_D + d] = th; }
        ua[d] = (th + 1.0f) / 2.0f;
    }
    float famb[16];
    encode_half<AMB_D>(ua, a.amb, hi, valid, famb);

    // ---- sigma net -------------------------------------------------------------------------------------------
    zero_acc(acc);
    {
        float b32[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) { b32[i] = fpos[i]; b32[16 + i] = famb[i]; }
        mfma_layer<32>(acc, a.w.sig_w0, b32, lane);
    }
    acc_to_b<true>(acc, bs);
    zero_acc(acc);
    mfma_layer<64>(acc, a.w.sig_w1, bs, lane);
    acc_to_b<true>(acc, bs);
    float logit[1];
    valu_rows<1>(a.w.sig_w2_sig, bs, hi, logit);
    const float sigma = a.density_scale * expf(logit[0]);
    zero_acc(acc);
    mfma_layer<64>(acc, a.w.sig_w2_geo, bs, lane);   // geo_feat: no activation

    // ---- colour net --------------------------------------------------------------------------------------------
    {
        float b72[72];
        float shv[16];
        sh_basis4(sh.dx[ray_local], sh.dy[ray_local], sh.dz[ray_local], shv);
#pragma unroll
        for (int s = 0; s < 8; ++s) b72[s] = hi ? shv[8 + s] : shv[s];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) b72[8 + m * 16 + r] = acc[m][r];
        load_bias(acc, a.frame_consts + 128, hi);
        mfma_layer<72>(acc, a.w.col_w0, b72, lane);
    }
    acc_to_b<true>(acc, bs);
    float rgb[3];
    valu_rows<3>(a.w.col_w1, bs, hi, rgb);

    if (valid && hi == 0) {
        sh.sigma[slot] = sigma;
        sh.cr[slot] = 1.0f / (1.0f + expf(-rgb[0]));
        sh.cg[slot] = 1.0f / (1.0f + expf(-rgb[1]));
        sh.cb[slot] = 1.0f / (1.0f + expf(-rgb[2]));
    }
}

}  // namespace gfpp
