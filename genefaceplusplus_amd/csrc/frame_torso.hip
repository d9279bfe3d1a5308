// frame_torso.hip -- fused torso pass + final compositing for gfx950.
//
// One thread per pixel of the frame:  bilinear occupancy test against the 2-D torso density grid -> (masked pixels only)
// frequency encoding of the pixel coordinate, optional head-aware encoder of (head rgb, head alpha), deformation MLP,
// 2-D tiled-grid lookup at the displaced coordinate, canonical MLP, sigmoid -> torso over background -> head over torso,
// clamp, depth normalisation.  Everything the reference does between radnerf_torso.py:156-197 (radnerf_torso_sr.py:
// 186-231) and forward_torso, in one launch with no boolean-mask gathers (each of which is a host sync there).
//
// The MLPs here are tiny (hidden 64 / 32, ~13 k FMA per pixel): weights are read through wave-uniform pointers (scalar
// loads, SGPR operands of v_fmac), k-major so that one input feeds a contiguous row of outputs; inputs that are constant
// over the frame (pose or landmark encodings, the individual code) are folded into bias vectors in the block prologue.
#include <hip/hip_runtime.h>

#include "grid_device.h"
#include "sh_device.h"

namespace gfpp {

constexpr int kTorsoThreads = 256;
constexpr int kMaxConst = 160;

struct TorsoArgs {
    // geometry / inputs
    const float *bg_coords;     // [N,2]
    const float *density_grid;  // [G*G]
    const float *cond_in;       // poses [6] (variant 0) or lm68 [136] (variant 1)
    const float *code;          // [code_dim] or null
    const float *state;         // [N,8] ray records of the head pass: {weights_sum, depth, r, g, b (premultiplied head colour), ...} (march_device.h::kRayRec)
    const float *nears, *fars;  // [N]
    const float *bg_color;      // [N,3] or null
    float bg_scalar, shrink, thresh;
    uint32_t N, G, variant, code_dim, const_dim, head_aware, use_head;
    // grid
    const float *table;
    const gfpp_grid_level *levels;
    // weights (k-major unless noted)
    const float *def_w0_x, *def_w0_c, *def_w0_h, *def_w1, *def_w2;
    const float *can_w0_g, *can_w0_x, *can_w0_c, *can_w0_h, *can_w1, *can_w2;
    const float *ha_w0, *ha_b0, *ha_w1, *ha_b1, *ha_w2, *ha_b2;
    // outputs
    float *out_image, *out_depth, *torso_alpha, *torso_bg, *deform;
    uint8_t *mask_out;
};

__device__ __forceinline__ float leaky(float v) { return v >= 0.0f ? v : v * 0.02f; }
__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// F.grid_sample(bilinear, zeros padding, align_corners=True) of a G x G single-channel grid; cx indexes columns.
__device__ __forceinline__ float bilinear_occupancy(const float *__restrict__ grid, uint32_t G, float cx, float cy) {
    const float ix = ((cx + 1.0f) / 2.0f) * (float)(G - 1);
    const float iy = ((cy + 1.0f) / 2.0f) * (float)(G - 1);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    auto tap = [&](float yy, float xx) -> float {
        const bool ok = xx >= 0.0f && xx <= (float)(G - 1) && yy >= 0.0f && yy <= (float)(G - 1);
        return ok ? grid[(uint32_t)yy * G + (uint32_t)xx] : 0.0f;
    };
    return tap(y0, x0) * wnw + tap(y0, x1) * wne + tap(y1, x0) * wsw + tap(y1, x1) * wse;
}

// acc[j] += x * Wt[j], j < OUT, Wt wave-uniform
template <int OUT>
__device__ __forceinline__ void axpy_row(float (&acc)[OUT], float x, const float *__restrict__ wt) {
#pragma unroll
    for (int j = 0; j < OUT; ++j) acc[j] = fmaf(x, wt[j], acc[j]);
}

__global__ __launch_bounds__(kTorsoThreads) void k_torso(TorsoArgs a) {
    __shared__ float s_const[kMaxConst];
    __shared__ float s_bdef[64];
    __shared__ float s_bcan[32];
    const int tid = threadIdx.x;

    // ---- block prologue: per-frame constant columns and their folded biases ------------------------------------------------
    // variant 0 (RADNeRFTorso):       const = [freq(pose[6], 4 octaves) (54), code]
    // variant 1 (RADNeRFTorsowithSR): const = [code, freq(chin landmarks 5..11 of lm68 (14 values), 4 octaves) (126)]
    {
        const uint32_t enc_D = a.variant == 0 ? 6u : 14u;
        const uint32_t enc_C = enc_D + 2u * enc_D * 4u;
        const uint32_t enc_at = a.variant == 0 ? 0u : a.code_dim;
        const uint32_t code_at = a.variant == 0 ? enc_C : 0u;
        for (uint32_t c = tid; c < enc_C; c += kTorsoThreads) {
            const uint32_t d = c % enc_D;
            const float v = a.variant == 0 ? a.cond_in[d] : a.cond_in[10 + d];   // landmarks 5..11 -> flat 10..23
            s_const[enc_at + c] = c < enc_D ? v : freq_feature(v, c / enc_D - 1);
        }
        for (uint32_t c = tid; c < a.code_dim; c += kTorsoThreads) s_const[code_at + c] = a.code[c];
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.0f;
        for (uint32_t k = 0; k < a.const_dim; ++k) s = fmaf(a.def_w0_c[(size_t)tid * a.const_dim + k], s_const[k], s);
        s_bdef[tid] = s;
    } else if (tid < 96) {
        const int j = tid - 64;
        float s = 0.0f;
        for (uint32_t k = 0; k < a.const_dim; ++k) s = fmaf(a.can_w0_c[(size_t)j * a.const_dim + k], s_const[k], s);
        s_bcan[j] = s;
    }
    __syncthreads();

    const uint32_t n = blockIdx.x * kTorsoThreads + tid;
    if (n >= a.N) return;

    const float cx = a.bg_coords[2ull * n], cy = a.bg_coords[2ull * n + 1];
    const float occ = bilinear_occupancy(a.density_grid, a.G, cx, cy);
    const bool masked = occ > a.thresh;
    const float hr = a.state[8ull * n + 2], hg = a.state[8ull * n + 3], hb = a.state[8ull * n + 4];
    const float wsum = a.state[8ull * n];

    float alpha = 0.0f, tr = 0.0f, tg = 0.0f, tb = 0.0f, ddx = 0.0f, ddy = 0.0f;
    if (masked) {
        const float x0 = cx * a.shrink, x1 = cy * a.shrink;
        // frequency encoding of the pixel coordinate: [x0, x1, sin(2^0 x0), sin(2^0 x1), cos(2^0 x0), cos(2^0 x1), sin(2^1 x0), ...]
        float ex[42];
        ex[0] = x0; ex[1] = x1;
#pragma unroll
        for (int c = 2; c < 42; ++c) ex[c] = freq_feature((c & 1) ? x1 : x0, (uint32_t)(c / 2 - 1));

        float ha[16];
        if (a.head_aware) {
            // Linear(4,16)+LeakyReLU -> Linear(16,32)+LeakyReLU -> Linear(32,16) on (head rgb, head alpha) (zeros when unused)
            const float in4[4] = {a.use_head ? hr : 0.0f, a.use_head ? hg : 0.0f, a.use_head ? hb : 0.0f, a.use_head ? wsum : 0.0f};
            float h1[16], h2[32];
#pragma unroll
            for (int j = 0; j < 16; ++j) h1[j] = a.ha_b0[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) axpy_row<16>(h1, in4[k], a.ha_w0 + k * 16);
#pragma unroll
            for (int j = 0; j < 32; ++j) h2[j] = a.ha_b1[j];
#pragma unroll
            for (int k = 0; k < 16; ++k) axpy_row<32>(h2, leaky(h1[k]), a.ha_w1 + k * 32);
#pragma unroll
            for (int j = 0; j < 16; ++j) ha[j] = a.ha_b2[j];
#pragma unroll
            for (int k = 0; k < 32; ++k) axpy_row<16>(ha, leaky(h2[k]), a.ha_w2 + k * 16);
        }

        // ---- deformation MLP: (42 + const + 16) -> 64 -> 64 -> 2 -----------------------------------------------------------
        float h[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) h[j] = s_bdef[j];
#pragma unroll
        for (int k = 0; k < 42; ++k) axpy_row<64>(h, ex[k], a.def_w0_x + k * 64);
        if (a.head_aware) {
#pragma unroll
            for (int k = 0; k < 16; ++k) axpy_row<64>(h, ha[k], a.def_w0_h + k * 64);
        }
        float g[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) g[j] = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) axpy_row<64>(g, fmaxf(h[k], 0.0f), a.def_w1 + k * 64);
        float dxy[2] = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 64; ++k) axpy_row<2>(dxy, fmaxf(g[k], 0.0f), a.def_w2 + k * 2);
        ddx = dxy[0]; ddy = dxy[1];

        // ---- 2-D tiled grid at the displaced, clamped coordinate --------------------------------------------------------------
        float u[2];
        u[0] = (clampf(x0 + ddx, -1.0f, 1.0f) + 1.0f) / 2.0f;
        u[1] = (clampf(x1 + ddy, -1.0f, 1.0f) + 1.0f) / 2.0f;
        float feat[32];
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            const gfpp_grid_level lv = a.levels[l];
            float o[2];
            grid_level_lookup<2, 2, float>(u, a.table, lv.offset, lv.size, lv.scale, lv.resolution, 1u, false, 0u, o);
            feat[2 * l] = o[0];
            feat[2 * l + 1] = o[1];
        }

        // ---- canonical MLP: (32 + 42 + const + 16) -> 32 -> 32 -> 4 ---------------------------------------------------------------
        float c1[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) c1[j] = s_bcan[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) axpy_row<32>(c1, feat[k], a.can_w0_g + k * 32);
#pragma unroll
        for (int k = 0; k < 42; ++k) axpy_row<32>(c1, ex[k], a.can_w0_x + k * 32);
        if (a.head_aware) {
#pragma unroll
            for (int k = 0; k < 16; ++k) axpy_row<32>(c1, ha[k], a.can_w0_h + k * 32);
        }
        float c2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) c2[j] = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) axpy_row<32>(c2, fmaxf(c1[k], 0.0f), a.can_w1 + k * 32);
        float o4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 32; ++k) axpy_row<4>(o4, fmaxf(c2[k], 0.0f), a.can_w2 + k * 4);
        alpha = sigmoidf(o4[0]);
        tr = sigmoidf(o4[1]); tg = sigmoidf(o4[2]); tb = sigmoidf(o4[3]);
    }

    // ---- torso over background, head over torso (radnerf_torso.py:186-197) ---------------------------------------------------------
    const float T = 1.0f - wsum;
    const float tcol[3] = {tr, tg, tb}, hcol[3] = {hr, hg, hb};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float bg = a.bg_color ? a.bg_color[3ull * n + c] : a.bg_scalar;
        const float tbg = tcol[c] * alpha + bg * (1.0f - alpha);
        a.torso_bg[3ull * n + c] = tbg;
        a.out_image[3ull * n + c] = clampf(hcol[c] + T * tbg, 0.0f, 1.0f);
    }
    a.torso_alpha[n] = alpha;
    a.deform[2ull * n] = ddx;
    a.deform[2ull * n + 1] = ddy;
    a.mask_out[n] = masked ? 1 : 0;
    a.out_depth[n] = fmaxf(a.state[8ull * n + 1] - a.nears[n], 0.0f) / (a.fars[n] - a.nears[n]);
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_torso_frame(const gfpp_torso_model *m, const gfpp_frame_ws *ws, const float *bg_coords, const float *cond_in,
                              const float *code, const float *bg_color, float bg_scalar, uint32_t use_head, float *out_image,
                              float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask, gfpp_stream_t stream) {
    if (!m || !ws || !bg_coords || !cond_in || !out_image || !out_depth || !torso_alpha || !torso_bg || !deform || !mask) {
        set_error("gfpp_torso_frame: null argument");
        return GFPP_EINVAL;
    }
    if (m->grid.D != 2 || m->grid.L != 16 || m->grid.dtype != GFPP_F32 || m->grid.gridtype != 1 || m->grid.interp != 0 || m->grid.align_corners) {
        set_error("gfpp_torso_frame: the torso grid must be a 16-level fp32 2-D tiled grid with linear interpolation");
        return GFPP_EUNSUPPORTED;
    }
    const uint32_t enc_c = m->variant == 0 ? 54u : 126u;
    if (m->variant > 1 || m->const_dim != enc_c + m->code_dim || m->const_dim > (uint32_t)kMaxConst || (m->code_dim && !code)) {
        set_error("gfpp_torso_frame: inconsistent constant-column layout");
        return GFPP_EINVAL;
    }
    TorsoArgs a;
    a.bg_coords = bg_coords; a.density_grid = m->density_grid; a.cond_in = cond_in; a.code = code;
    a.state = ws->ray_state; a.nears = ws->nears; a.fars = ws->fars;
    a.bg_color = bg_color; a.bg_scalar = bg_scalar; a.shrink = m->torso_shrink; a.thresh = m->density_thresh;
    a.N = ws->N; a.G = m->grid_size; a.variant = m->variant; a.code_dim = m->code_dim; a.const_dim = m->const_dim;
    a.head_aware = m->head_aware; a.use_head = use_head;
    a.table = (const float *)m->grid.table; a.levels = m->grid.levels;
    a.def_w0_x = m->def_w0_x; a.def_w0_c = m->def_w0_c; a.def_w0_h = m->def_w0_h; a.def_w1 = m->def_w1; a.def_w2 = m->def_w2;
    a.can_w0_g = m->can_w0_g; a.can_w0_x = m->can_w0_x; a.can_w0_c = m->can_w0_c; a.can_w0_h = m->can_w0_h; a.can_w1 = m->can_w1; a.can_w2 = m->can_w2;
    a.ha_w0 = m->ha_w0; a.ha_b0 = m->ha_b0; a.ha_w1 = m->ha_w1; a.ha_b1 = m->ha_b1; a.ha_w2 = m->ha_w2; a.ha_b2 = m->ha_b2;
    a.out_image = out_image; a.out_depth = out_depth; a.torso_alpha = torso_alpha; a.torso_bg = torso_bg; a.deform = deform; a.mask_out = mask;
    hipLaunchKernelGGL(k_torso, dim3(div_up(ws->N, kTorsoThreads)), dim3(kTorsoThreads), 0, (hipStream_t)stream, a);
    return check_launch("gfpp_torso_frame");
}
