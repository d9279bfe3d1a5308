// frame_torso_lp.hip -- torso pass + final compositing with the torso MLPs on 16-bit MFMA operands (fp32 accumulation).
//
// Same function as frame_torso.hip (radnerf_torso.py:156-197 / radnerf_torso_sr.py:186-231 + forward_torso), for the 16-bit
// precision modes of the head kernel.  The fp32 kernel spends its time streaming ~13 k weights per pixel through scalar loads into
// v_fmac operands; here a wavefront compacts the masked pixels of its 64-pixel span and pushes them, 32 at a time, through
//   head-aware encoder 4 -> 16 -> 32 -> 16 (LeakyReLU)      4 MFMAs
//   deformation MLP   (42 freq + 16 head-aware) -> 64 -> 64  16 MFMAs, -> 2 as packed dot products
//   2-D tiled grid at the displaced coordinate (each half-wave 8 of the 16 levels, straight-line lookups on the padded table)
//   canonical MLP     (32 grid + 42 freq + 16 head-aware) -> 32 -> 32   8 MFMAs, -> 4 as packed dot products
// with the 28 KB weight image resident in LDS.  Columns that are constant over the frame (pose / landmark encoding, individual code)
// are folded into fp32 bias vectors in the block prologue, exactly as in the fp32 kernel.  Everything outside the MLPs (occupancy
// test, frequency features, grid interpolation, sigmoid, compositing, depth) is fp32 and shared with frame_torso.hip's semantics.
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "grid_device.h"
#include "lp_mfma_device.h"
#include "march_device.h"
#include "sh_device.h"


namespace gfpp {

constexpr int kTlThreads = 256;
constexpr int kTlWaves = kTlThreads / 64;
constexpr int kTlMaxConst = 160;
// weight image: [step][tile][lane] 16-byte fragments, layers in this order (steps x tiles)
constexpr int kTlHa0 = 0;                    // 1 x 1   4 -> 16 (rows padded to 32)
constexpr int kTlHa1 = kTlHa0 + 1;           // 1 x 1  16 -> 32
constexpr int kTlHa2 = kTlHa1 + 1;           // 2 x 1  32 -> 16 (padded)
constexpr int kTlDef0 = kTlHa2 + 2;          // 4 x 2  [freq 42 | pad 6 | head-aware 16] -> 64
constexpr int kTlDef1 = kTlDef0 + 8;         // 4 x 2  64 -> 64
constexpr int kTlCan0 = kTlDef1 + 8;         // 6 x 1  [grid 32 | freq 42 | pad 6 | head-aware 16] -> 32
constexpr int kTlCan1 = kTlCan0 + 6;         // 2 x 1  32 -> 32
constexpr int kTlFrags = kTlCan1 + 2;        // 28 fragments-of-64-lanes
constexpr int kTlSkinnyVecs = 2 * 2 * 4 + 2 * 4 * 2;      // operand vectors of 8 values: def2 [2 halves][2 rows][4] + can2 [2][4][2]
constexpr int kTlSkinnyDef = 0, kTlSkinnyCan = 2 * 2 * 4;

struct TorsoLpArgs {
    const float *bg_coords, *density_grid, *cond_in, *code, *state /* [N,8] ray records of the head pass, march_device.h::kRayRec */, *nears, *fars, *bg_color;
    float bg_scalar, shrink, thresh;
    uint32_t N, G, variant, code_dim, const_dim, head_aware, use_head;
    const float *table;
    const gfpp_grid_level *levels;
    const float *def_w0_c, *can_w0_c;                      // fp32 [64][const_dim], [32][const_dim]: folded in the prologue
    const float *ha_b0, *ha_b1, *ha_b2;                    // fp32 biases of the head-aware encoder
    const void *w16;                                       // kTlFrags * 64 fragments (operand vectors of 8 values: f16 / bf16 / f32)
    const void *skinny16;                                  // kTlSkinnyVecs operand vectors
    float *out_image, *out_depth, *torso_alpha, *torso_bg, *deform;
    uint8_t *mask_out;
    BudgetView bv;            // hist != null: the head pass was the persistent launch with the resolve deferred to this kernel
    gfpp_clip_job *job;       // != null: also store the frame as uint8 into the clip job's slot of lane `lane` and advance its cursor
    uint32_t lane, sub, advance;   // the frame takes job position cursor[lane] + sub; the launch moves the cursor by `advance` (0: the job's `lanes`; ~0: not at all)
};

template <typename H>
struct TorsoLpShared {
    typename LpTraits<H>::vec w[kTlFrags * 64];            // 28 672 B (16-bit operands) / 57 344 B (exact fp32)
    typename LpTraits<H>::vec skinny[kTlSkinnyVecs];       //    512 B / 1 024 B
    float consts[kTlMaxConst];
    float bdef[64], bcan[32], bha0[32], bha1[32], bha2[32];
    gfpp_grid_level lv[16];
    float res[kTlWaves][6][64];        // per wavefront: alpha, r, g, b, dx, dy by local pixel
    uint8_t order[kTlWaves][64];
};

__device__ __forceinline__ float tl_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float tl_bilinear_occupancy(const float *__restrict__ grid, uint32_t G, float cx, float cy) {
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

// bias vector (natural order, LDS) -> accumulator fragment: acc[t][r] = b[32 t + (r&3) + 8 (r>>2) + 4 h]
template <int T>
__device__ __forceinline__ void tl_load_bias(v16f (&acc)[T], const float *__restrict__ b, int hi) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi];
}

// One level of the 2-D tiled grid, index arithmetic resolved on the host, padded table (see frame_head_lp.hip) -- in two halves, so that the caller can request
// the rows of ALL its levels before it interpolates the first: as one function per level the compiler waited for a level's two rows before it computed the next
// level's addresses -- eight memory round trips in a row per 32-pixel pass (tools/torso_phase.py: 5 us per pass).  Same arithmetic, same order.
struct TlRows {
    float frac[2];
    f32x4_a8 v0, v1;
};
__device__ __forceinline__ void tl_level2_issue(const float (&u)[2], const float *__restrict__ table, const gfpp_grid_level &lv, TlRows &g) {
    uint32_t base[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float pos = fmaf(u[d], lv.scale, 0.5f);
        const float fl = floorf(pos);
        base[d] = (uint32_t)fl;
        g.frac[d] = pos - fl;
    }
    const float *lt = table + 2ull * lv.offset;
    const uint32_t y0 = __umul24(base[1], lv.sy), y1 = y0 + lv.sy;
    const uint32_t r0 = (base[0] + y0) & lv.mask, r1 = (base[0] + y1) & lv.mask;
    g.v0 = *reinterpret_cast<const f32x4_a8 *>(lt + 2ull * r0);
    g.v1 = *reinterpret_cast<const f32x4_a8 *>(lt + 2ull * r1);
}
__device__ __forceinline__ void tl_level2_finish(const TlRows &g, float (&out)[2]) {
    out[0] = 0.0f;
    out[1] = 0.0f;
    {
        const float w0 = (1.0f - g.frac[0]) * (1.0f - g.frac[1]), w1 = g.frac[0] * (1.0f - g.frac[1]);
        out[0] = fmaf(w1, g.v0[2], fmaf(w0, g.v0[0], out[0]));
        out[1] = fmaf(w1, g.v0[3], fmaf(w0, g.v0[1], out[1]));
    }
    {
        const float w0 = (1.0f - g.frac[0]) * g.frac[1], w1 = g.frac[0] * g.frac[1];
        out[0] = fmaf(w1, g.v1[2], fmaf(w0, g.v1[0], out[0]));
        out[1] = fmaf(w1, g.v1[3], fmaf(w0, g.v1[1], out[1]));
    }
}

// What a 32-pixel pass of the torso MLPs reads besides its pixels: the LDS-resident weight image and the frame's folded biases.
template <typename H>
struct TorsoPassCtx {
    const typename LpTraits<H>::vec *W, *skinny;
    const float *bdef, *bcan, *bha0, *bha1, *bha2;
    const gfpp_grid_level *lv;
    const float *table;
    float shrink;
    uint32_t head_aware, use_head;
};

// The torso field of 32 pixels (one per column; lane (j, h) supplies column j's pixel coordinate px, py and head colour / alpha hr .. wsum in BOTH half-waves):
// head-aware encoder, deformation MLP, 2-D grid, canonical MLP -> o4 = pre-sigmoid alpha, r, g, b and dxy = the deformation, valid in the lanes of both halves.
template <typename H>
__device__ __forceinline__ void torso_eval_cols(const TorsoPassCtx<H> &c, float px, float py, float hr, float hg, float hb, float wsum, int lane, float (&o4)[4],
                                                float (&dxy)[2]) {
    typedef typename LpTraits<H>::vec vec;
    const int hi = lane >> 5;
    const vec *W = c.W;
    const float x0 = px * c.shrink, x1 = py * c.shrink;

    // frequency features of the pixel coordinate, this half-wave's 24 of the 48 operand slots (42 used):
    // slot k = 16 s + 8 h + e  <->  ex[k] = k < 2 ? x_k : freq_feature(x_{k&1}, k/2 - 1)
    vec bex[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * s + 8 * hi + e;
            const float xk = (k & 1) ? x1 : x0;
            float v = k < 2 ? xk : (k < 42 ? freq_feature(xk, (uint32_t)(k / 2 - 1)) : 0.0f);
            if constexpr (sizeof(H) == 2) v = k < 2 ? xk : (k < 42 ? freq_feature_fast(xk, (uint32_t)(k / 2 - 1)) : 0.0f);   // (the exact-fp32 instantiation keeps sinf)
            bex[s][e] = (H)v;
        }

    // head-aware encoder of (head rgb, head alpha): Linear 4->16, LeakyReLU, 16->32, LeakyReLU, 32->16
    vec bha[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) bha[0][e] = (H)0.0f;
    if (c.head_aware) {
        const float i0 = c.use_head ? hr : 0.0f, i1 = c.use_head ? hg : 0.0f;
        const float i2 = c.use_head ? hb : 0.0f, i3 = c.use_head ? wsum : 0.0f;
        vec bin[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) bin[0][e] = (H)0.0f;
        if (hi == 0) { bin[0][0] = (H)i0; bin[0][1] = (H)i1; bin[0][2] = (H)i2; bin[0][3] = (H)i3; }
        v16f acc1[1];
        vec b2[2];
        tl_load_bias<1>(acc1, c.bha0, hi);
        mfma_layer_lds<H, 1, 1>(acc1, W + kTlHa0 * 64, bin, lane);
        act_pack<H, 1, 2>(acc1, b2);            // rows 0..15 live in b2[0]; b2[1] (rows 16..31) is padding
        vec b1[1] = {b2[0]};
        tl_load_bias<1>(acc1, c.bha1, hi);
        mfma_layer_lds<H, 1, 1>(acc1, W + kTlHa1 * 64, b1, lane);
        act_pack<H, 1, 2>(acc1, b2);
        tl_load_bias<1>(acc1, c.bha2, hi);
        mfma_layer_lds<H, 2, 1>(acc1, W + kTlHa2 * 64, b2, lane);
        act_pack<H, 1, 0>(acc1, b2);
        bha[0] = b2[0];
    }

    // deformation MLP
    {
        v16f acc2[2];
        vec bin[4] = {bex[0], bex[1], bex[2], bha[0]};
        tl_load_bias<2>(acc2, c.bdef, hi);
        mfma_layer_lds<H, 4, 2>(acc2, W + kTlDef0 * 64, bin, lane);
        vec bh[4];
        act_pack<H, 2, 1>(acc2, bh);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
        mfma_layer_lds<H, 4, 2>(acc2, W + kTlDef1 * 64, bh, lane);
        act_pack<H, 2, 1>(acc2, bh);
        skinny_dot<2, 4, H>(c.skinny + kTlSkinnyDef, 2, bh, hi, dxy);
    }

    // 2-D tiled grid at the displaced, clamped coordinate; half-wave h encodes the levels h, h+2, ...
    vec bgrid[2];
    {
        float u[2];
        u[0] = (clampf(x0 + dxy[0], -1.0f, 1.0f) + 1.0f) / 2.0f;
        u[1] = (clampf(x1 + dxy[1], -1.0f, 1.0f) + 1.0f) / 2.0f;
        float f[16];
        TlRows rows[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tl_level2_issue(u, c.table, c.lv[2 * i + hi], rows[i]);
        __builtin_amdgcn_sched_barrier(0);             // all sixteen rows are requested before the first is used
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float o[2];
            tl_level2_finish(rows[i], o);
            f[2 * i] = o[0];
            f[2 * i + 1] = o[1];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) bgrid[s][e] = (H)f[8 * s + e];
    }

    // canonical MLP
    {
        v16f acc1[1];
        vec bin[6] = {bgrid[0], bgrid[1], bex[0], bex[1], bex[2], bha[0]};
        tl_load_bias<1>(acc1, c.bcan, hi);
        mfma_layer_lds<H, 6, 1>(acc1, W + kTlCan0 * 64, bin, lane);
        vec bh[2];
        act_pack<H, 1, 1>(acc1, bh);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[0][r] = 0.0f;
        mfma_layer_lds<H, 2, 1>(acc1, W + kTlCan1 * 64, bh, lane);
        act_pack<H, 1, 1>(acc1, bh);
        skinny_dot<4, 2, H>(c.skinny + kTlSkinnyCan, 4, bh, hi, o4);
    }
}

// One pass of a wavefront's 64-pixel span: its compacted masked pixels [first, first + 32) (order: compact index -> local pixel) through torso_eval_cols; alpha, r, g,
// b, dx, dy of each go to res[0..5][local pixel] (the wavefront's own LDS rows).  cx, cy, hr .. wsum: this lane's pixel coordinate and head colour / alpha (the pass
// fetches its columns' values with lane shuffles).
template <typename H>
__device__ __forceinline__ void torso_pass(const TorsoPassCtx<H> &c, float *__restrict__ res, const uint8_t *__restrict__ order, uint32_t first, uint32_t n_m,
                                           float cx, float cy, float hr, float hg, float hb, float wsum, int lane) {
    const int j = lane & 31, hi = lane >> 5;
    const uint32_t col = first + (uint32_t)j;
    const bool valid = col < n_m;
    const int src = valid ? (int)order[col] : 0;   // local pixel this column evaluates
    float o4[4], dxy[2];
    torso_eval_cols<H>(c, __shfl(cx, src), __shfl(cy, src), __shfl(hr, src), __shfl(hg, src), __shfl(hb, src), __shfl(wsum, src), lane, o4, dxy);
    if (valid && hi == 0) {
        res[0 * 64 + src] = tl_sigmoid(o4[0]);
        res[1 * 64 + src] = tl_sigmoid(o4[1]);
        res[2 * 64 + src] = tl_sigmoid(o4[2]);
        res[3 * 64 + src] = tl_sigmoid(o4[3]);
        res[4 * 64 + src] = dxy[0];
        res[5 * 64 + src] = dxy[1];
    }
}

template <typename H>
__global__ __launch_bounds__(kTlThreads) void k_torso_lp(TorsoLpArgs a) {
    typedef typename LpTraits<H>::vec vec;
    __shared__ TorsoLpShared<H> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- this thread's pixel: occupancy test first (most workgroups of a frame have no torso pixel and skip the weights) ----------
    const uint32_t n = blockIdx.x * kTlThreads + tid;
    const bool in_frame = n < a.N;
    float cx = 0.0f, cy = 0.0f, hr = 0.0f, hg = 0.0f, hb = 0.0f, wsum = 0.0f, hdepth = 0.0f;
    bool masked = false;
    // the step budget of the head pass (renderer.py:359-364,384 replayed on the histogram) when its resolve step was left to this kernel
    __shared__ uint32_t s_budget;
    if (tid == 0) s_budget = a.bv.hist ? budget_from_hist(a.bv.hist, a.bv.N_global, a.bv.max_steps, nullptr) : 0u;
    if (in_frame) {
        cx = a.bg_coords[2ull * n]; cy = a.bg_coords[2ull * n + 1];
        masked = tl_bilinear_occupancy(a.density_grid, a.G, cx, cy) > a.thresh;
    }
    const bool block_has_work = __syncthreads_or(masked ? 1 : 0) != 0;
    if (in_frame) {
        const RayAccum head = ray_state_final(a.state, a.bv, s_budget, n);
        hr = head.r; hg = head.g; hb = head.b; wsum = head.wsum; hdepth = head.depth;
    }

    float alpha = 0.0f, tr = 0.0f, tg = 0.0f, tb = 0.0f, ddx = 0.0f, ddy = 0.0f;
    if (block_has_work) {
        // ---- prologue: weights -> LDS, per-frame constant columns folded into biases (same arithmetic as frame_torso.hip) ----------
        for (int i = tid; i < kTlFrags * 64; i += kTlThreads) sh.w[i] = reinterpret_cast<const vec *>(a.w16)[i];
        for (int i = tid; i < kTlSkinnyVecs; i += kTlThreads) sh.skinny[i] = reinterpret_cast<const vec *>(a.skinny16)[i];
        if (tid < 16 * 8) reinterpret_cast<uint32_t *>(&sh.lv[0])[tid] = reinterpret_cast<const uint32_t *>(a.levels)[tid];
        if (tid < 32) {
            sh.bha0[tid] = (a.head_aware && tid < 16) ? a.ha_b0[tid] : 0.0f;
            sh.bha1[tid] = a.head_aware ? a.ha_b1[tid] : 0.0f;
            sh.bha2[tid] = (a.head_aware && tid < 16) ? a.ha_b2[tid] : 0.0f;
        }
        {
            const uint32_t enc_D = a.variant == 0 ? 6u : 14u;
            const uint32_t enc_C = enc_D + 2u * enc_D * 4u;
            const uint32_t enc_at = a.variant == 0 ? 0u : a.code_dim;
            const uint32_t code_at = a.variant == 0 ? enc_C : 0u;
            for (uint32_t c = tid; c < enc_C; c += kTlThreads) {
                const uint32_t d = c % enc_D;
                const float v = a.variant == 0 ? a.cond_in[d] : a.cond_in[10 + d];   // landmarks 5..11 -> flat 10..23
                sh.consts[enc_at + c] = c < enc_D ? v : freq_feature(v, c / enc_D - 1);
            }
            for (uint32_t c = tid; c < a.code_dim; c += kTlThreads) sh.consts[code_at + c] = a.code[c];
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.0f;
            for (uint32_t k = 0; k < a.const_dim; ++k) s = fmaf(a.def_w0_c[(size_t)tid * a.const_dim + k], sh.consts[k], s);
            sh.bdef[tid] = s;
        } else if (tid < 96) {
            const int q = tid - 64;
            float s = 0.0f;
            for (uint32_t k = 0; k < a.const_dim; ++k) s = fmaf(a.can_w0_c[(size_t)q * a.const_dim + k], sh.consts[k], s);
            sh.bcan[q] = s;
        }
        // (measured, dropped: the row's weights 32 at a time before the fma chain -- the landmark variant's 134 columns 8.5 -> 7.0 us of prologue, the pose
        // variant's 62 columns 5.0 -> 6.2; tools/torso_phase.py)
        __syncthreads();

        // ---- compaction of the wavefront's masked pixels -------------------------------------------------------------------------
        const unsigned long long ballot = __ballot(masked);
        const uint32_t n_m = (uint32_t)__popcll(ballot);
        if (masked) sh.order[wave][__popcll(ballot & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        TorsoPassCtx<H> pc{sh.w, sh.skinny, sh.bdef, sh.bcan, sh.bha0, sh.bha1, sh.bha2, sh.lv, a.table, a.shrink, a.head_aware, a.use_head};
        for (uint32_t first = 0; first < n_m; first += 32) torso_pass<H>(pc, &sh.res[wave][0][0], sh.order[wave], first, n_m, cx, cy, hr, hg, hb, wsum, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (masked) {
            const float *r = &sh.res[wave][0][0];
            alpha = r[lane]; tr = r[64 + lane]; tg = r[128 + lane]; tb = r[192 + lane]; ddx = r[256 + lane]; ddy = r[320 + lane];
        }
    }
    uint32_t packed = 0;                // this pixel's uint8 r | g << 8 | b << 16 (clip job only)
    if (in_frame) {
        // ---- torso over background, head over torso (radnerf_torso.py:186-197) -----------------------------------------------------
        const float T = 1.0f - wsum;
        const float tcol[3] = {tr, tg, tb}, hcol[3] = {hr, hg, hb};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float bg = a.bg_color ? a.bg_color[3ull * n + c] : a.bg_scalar;
            const float tbg = tcol[c] * alpha + bg * (1.0f - alpha);
            const float v = clampf(hcol[c] + T * tbg, 0.0f, 1.0f);
            a.torso_bg[3ull * n + c] = tbg;
            a.out_image[3ull * n + c] = v;
            packed |= (uint32_t)(uint8_t)clampf(v * 255.0f, 0.0f, 255.0f) << (8 * c);   // gfpp_rgb_to_u8's conversion (genefacepp_infer.py:468)
        }
        a.torso_alpha[n] = alpha;
        a.deform[2ull * n] = ddx;
        a.deform[2ull * n + 1] = ddy;
        a.mask_out[n] = masked ? 1 : 0;
        a.out_depth[n] = fmaxf(hdepth - a.nears[n], 0.0f) / (a.fars[n] - a.nears[n]);
    }
    if (a.job) {
        // the uint8 frame, fused: four consecutive pixels hold 12 bytes = three dwords, assembled inside the lane quad and written as dwords (a wavefront
        // writes 192 contiguous bytes with ONE store instruction; per-pixel byte stores took 45 us per 512^2 frame, the whole torso pass takes 30)
        const int q0 = lane & ~3, ql = lane & 3;
        const uint32_t w0 = (uint32_t)__shfl((int)packed, q0), w1 = (uint32_t)__shfl((int)packed, q0 + 1);
        const uint32_t w2 = (uint32_t)__shfl((int)packed, q0 + 2), w3 = (uint32_t)__shfl((int)packed, q0 + 3);
        const uint32_t pos = a.job->cursor[a.lane] + a.sub;
        if (in_frame && pos < a.job->n) {
            uint8_t *frame = a.job->out + (size_t)(pos % a.job->ring_frames) * a.job->frame_bytes;
            const uint32_t nq = n & ~3u;
            if (nq + 3u < a.N && (a.job->frame_bytes & 3ull) == 0ull) {     // (odd frame sizes: byte stores keep every access aligned)
                const uint32_t d = ql == 0 ? (w0 | (w1 << 24)) : (ql == 1 ? ((w1 >> 8) | (w2 << 16)) : ((w2 >> 16) | (w3 << 8)));
                if (ql < 3) *reinterpret_cast<uint32_t *>(frame + 3ull * nq + 4u * (uint32_t)ql) = d;
            } else {
                frame[3ull * n] = (uint8_t)packed; frame[3ull * n + 1] = (uint8_t)(packed >> 8); frame[3ull * n + 2] = (uint8_t)(packed >> 16);
            }
        }
    }
    if (a.job) {
        // the lane's cursor moves on when every workgroup has read it: the last one to get here advances it (as k_clip_store_u8 does)
        __syncthreads();
        if (tid == 0 && a.advance != 0xFFFFFFFFu) {
            const uint32_t pos = a.job->cursor[a.lane];
            if (atomicAdd(&a.job->ticket[a.lane], 1u) == gridDim.x - 1u) {
                a.job->ticket[a.lane] = 0u;
                a.job->cursor[a.lane] = pos + (a.advance ? a.advance : a.job->lanes);
            }
        }
    }
}


// ---- a frame GROUP's torso passes as TWO launches (round 5) -----------------------------------------------------------------------------------------------
// k_torso_lp is one workgroup per 256 pixels with every workgroup resident at once: a launch lasts as long as ONE workgroup's dependent chain -- occupancy test,
// the 28 KB weight image into LDS, the fold of the frame's constant columns (62-134 dependent fmas per bias), two 32-pixel passes -- 27-30 us whatever the frame
// size, with the device a few per cent busy, and a group of K frames paid it K times plus K uint8 stores, a resolve launch and the gaps between them.  Two facts
// let the group do better.  (1) WHICH pixels the torso field is evaluated at does not depend on the frame: the mask is the occupancy grid sampled at the pixel
// coordinates (radnerf_torso.py:166-169), both constants of the model and the resolution -- so the masked pixels are listed ONCE (gfpp_torso_mask + a stream
// compaction on the host side of the ABI) and every 32-pixel pass of every frame is full and known in advance: k_torso_mlp_group deals the K x ceil(M / 32) passes
// out to persistent wavefronts, weights once per workgroup, no occupancy test, no compaction, no shuffles.  (2) Everything else of the pass -- resolve (the frame's
// step budget from its histogram, snapshot selection per ray), torso over background, head over torso, depth, the uint8 frame of the clip job and the cursor's
// advance -- is a stream over all pixels: k_torso_compose_group, one thread per pixel at full occupancy.  The constant fold happens once per FRAME
// (k_torso_fold, issued ahead of the head launch).  Per pixel the same instructions as k_torso_lp (torso_eval_cols, the compositing expressions): every output is
// the bits of the per-frame launch.  (A first version -- ONE launch of persistent wavefronts walking over 64-pixel spans with the occupancy test inside -- took
// 113 us for four 512^2 frames, as long as the four launches it replaced: eight dependent global round trips per span and 0 or 4 masked spans per wavefront.)
struct TorsoFoldArgs {
    const float *cond_in;       // frame f's lm68 [136] / pose [6] at cond_in + f * cond_stride
    uint32_t cond_stride;
    const float *code, *def_w0_c, *can_w0_c;
    uint32_t variant, code_dim, const_dim;
    float *out;                 // [frames, 96]: bdef [64] | bcan [32]
};

__global__ __launch_bounds__(128) void k_torso_fold(TorsoFoldArgs f) {
    __shared__ float consts[kTlMaxConst];
    const uint32_t tid = threadIdx.x;
    const float *cond_in = f.cond_in + (size_t)blockIdx.x * f.cond_stride;
    // (the prologue of k_torso_lp, expression for expression)
    const uint32_t enc_D = f.variant == 0 ? 6u : 14u;
    const uint32_t enc_C = enc_D + 2u * enc_D * 4u;
    const uint32_t enc_at = f.variant == 0 ? 0u : f.code_dim;
    const uint32_t code_at = f.variant == 0 ? enc_C : 0u;
    for (uint32_t c = tid; c < enc_C; c += 128u) {
        const uint32_t d = c % enc_D;
        const float v = f.variant == 0 ? cond_in[d] : cond_in[10 + d];
        consts[enc_at + c] = c < enc_D ? v : freq_feature(v, c / enc_D - 1);
    }
    for (uint32_t c = tid; c < f.code_dim; c += 128u) consts[code_at + c] = f.code[c];
    __syncthreads();
    if (tid < 96u) {
        const float *w = tid < 64u ? f.def_w0_c + (size_t)tid * f.const_dim : f.can_w0_c + (size_t)(tid - 64u) * f.const_dim;
        float s = 0.0f;
        for (uint32_t k = 0; k < f.const_dim; ++k) s = fmaf(w[k], consts[k], s);
        f.out[(size_t)blockIdx.x * 96u + tid] = s;
    }
}

constexpr uint32_t kTgMaxFrames = 4;      // = kPMaxFrames of the head launch (frame_head_lp.hip)
constexpr int kTgCounterWords = 192, kTgBudgetBase = 128, kTgBudgetSamples = 40;     // gfpp_frame_ws.counters (head_eval_device.h: kCounterWords, kBudgetBase, kBudgetSamples)
// two free words of a frame's budget array (zeroed with it by the group's prologue): the frame's step budget B and the clip job position of the group's first frame,
// computed / read ONCE by the MLP launch and handed to the compose launch -- a first version computed B (two dozen dependent loads of the histogram) in every one
// of the compose launch's 4 096 workgroups and took a ticket per workgroup on one word to find the last reader of the cursor: 219 us for four frames
constexpr int kTgSlotBudget = 56, kTgSlotJobPos = 57;

// The torso mask of a resolution: mask[n] = occupancy(bg_coords[n]) > thresh -- k_torso_lp's own test, once per model and resolution.
__global__ __launch_bounds__(256) void k_torso_mask(const float *__restrict__ bg_coords, const float *__restrict__ density_grid, uint32_t G, float thresh, uint32_t N,
                                                    uint8_t *__restrict__ mask) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    mask[n] = tl_bilinear_occupancy(density_grid, G, bg_coords[2ull * n], bg_coords[2ull * n + 1]) > thresh ? 1 : 0;
}

struct TorsoGroupArgs {
    TorsoLpArgs a;              // the frames' shared inputs; every per-ray array is the STACK of the K frames (frame f at + f * N), outputs likewise
    uint32_t frames, max_steps;
    const float *folded;        // [frames, 96] (k_torso_fold)
    int32_t *counters;          // [frames, 192]: the head launch's histograms; the reconstructed alive counts are written here (compose kernel)
    const float *snaps;         // [frames * N, 7, 5]
    const uint8_t *mask;        // [N] (k_torso_mask)
    const int32_t *masked;      // [n_masked] pixel indices with mask = 1, ascending
    uint32_t n_masked, passes_per_frame;
};

template <typename H>
struct TorsoMlpShared {
    typename LpTraits<H>::vec w[kTlFrags * 64];
    typename LpTraits<H>::vec skinny[kTlSkinnyVecs];
    float bdef[kTgMaxFrames][64], bcan[kTgMaxFrames][32], bha0[32], bha1[32], bha2[32];
    gfpp_grid_level lv[16];
    uint32_t budget[kTgMaxFrames];
};

template <typename H>
__global__ __launch_bounds__(kTlThreads, 3) void k_torso_mlp_group(TorsoGroupArgs g) {
    typedef typename LpTraits<H>::vec vec;
    const TorsoLpArgs &a = g.a;
    __shared__ TorsoMlpShared<H> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    for (int i = tid; i < kTlFrags * 64; i += kTlThreads) sh.w[i] = reinterpret_cast<const vec *>(a.w16)[i];
    for (int i = tid; i < kTlSkinnyVecs; i += kTlThreads) sh.skinny[i] = reinterpret_cast<const vec *>(a.skinny16)[i];
    if (tid < 16 * 8) reinterpret_cast<uint32_t *>(&sh.lv[0])[tid] = reinterpret_cast<const uint32_t *>(a.levels)[tid];
    if (tid < 32) {
        sh.bha0[tid] = (a.head_aware && tid < 16) ? a.ha_b0[tid] : 0.0f;
        sh.bha1[tid] = a.head_aware ? a.ha_b1[tid] : 0.0f;
        sh.bha2[tid] = (a.head_aware && tid < 16) ? a.ha_b2[tid] : 0.0f;
    }
    for (uint32_t i = (uint32_t)tid; i < g.frames * 96u; i += kTlThreads) {
        const uint32_t f = i / 96u, k = i - f * 96u;
        const float v = g.folded[i];
        if (k < 64u) sh.bdef[f][k] = v; else sh.bcan[f][k - 64u] = v;
    }
    const bool need_head = a.head_aware && a.use_head;           // the head's colour / alpha of the pixel are inputs of the torso field (head-aware encoder)
    if ((uint32_t)tid < g.frames) {
        // renderer.py:359-364,384 replayed on the frame's histogram (k_group_budget_resolve's header).  Workgroup 0 also leaves the alive counts the trip launches would
        // have counted (FramePipeline.trip_counters), the budget and -- thread 0 -- the lane's job position for the compose launch, which then needs no such chain
        int32_t *c = g.counters + (size_t)tid * kTgCounterWords;
        const bool first = blockIdx.x == 0;
        const uint32_t B = (need_head || first) ? budget_from_hist(c + kTgBudgetBase, a.N, g.max_steps, first ? c : nullptr) : 0u;
        sh.budget[tid] = B;
        if (first) {
            c[kTgBudgetBase + kTgSlotBudget] = (int32_t)B;
            if (tid == 0) {
                c[64] = c[kTgBudgetBase + kTgBudgetSamples];
                c[kTgBudgetBase + kTgSlotJobPos] = a.job ? (int32_t)a.job->cursor[a.lane] : 0;
            }
        }
    }
    __syncthreads();
    TorsoPassCtx<H> pc{sh.w, sh.skinny, nullptr, nullptr, sh.bha0, sh.bha1, sh.bha2, sh.lv, a.table, a.shrink, a.head_aware, a.use_head};
    const uint32_t n_passes = g.frames * g.passes_per_frame, n_waves = gridDim.x * kTlWaves;
    for (uint32_t p = blockIdx.x * kTlWaves + (uint32_t)wave; p < n_passes; p += n_waves) {
        const uint32_t f = p / g.passes_per_frame, col = (p - f * g.passes_per_frame) * 32u + (uint32_t)j;
        const bool valid = col < g.n_masked;
        const uint32_t n = (uint32_t)g.masked[valid ? col : 0u];
        const size_t fn = (size_t)f * a.N + n;
        const float px = a.bg_coords[2ull * n], py = a.bg_coords[2ull * n + 1];
        float hr = 0.0f, hg = 0.0f, hb = 0.0f, wsum = 0.0f;
        if (need_head) {
            const BudgetView bv{g.counters + (size_t)f * kTgCounterWords + kTgBudgetBase, g.snaps, a.N, g.max_steps};
            const RayAccum head = ray_state_final(a.state, bv, sh.budget[f], (uint32_t)fn);
            hr = head.r; hg = head.g; hb = head.b; wsum = head.wsum;
        }
        pc.bdef = sh.bdef[f];
        pc.bcan = sh.bcan[f];
        float o4[4], dxy[2];
        torso_eval_cols<H>(pc, px, py, hr, hg, hb, wsum, lane, o4, dxy);
        if (valid && hi == 0) {
            // alpha, the torso's own colour (k_torso_compose_group blends it over the background in place) and the deformation of the masked pixel
            a.torso_alpha[fn] = tl_sigmoid(o4[0]);
            a.torso_bg[3ull * fn] = tl_sigmoid(o4[1]);
            a.torso_bg[3ull * fn + 1] = tl_sigmoid(o4[2]);
            a.torso_bg[3ull * fn + 2] = tl_sigmoid(o4[3]);
            a.deform[2ull * fn] = dxy[0];
            a.deform[2ull * fn + 1] = dxy[1];
        }
    }
}

// grid (ceil(N / 256), frames): one thread per pixel of a frame; the frame's step budget and the group's job position come from the MLP launch (no per-workgroup
// chain of dependent loads in front of the stream, no ticket behind it: nobody but this launch's first thread touches the lane's cursor, and the next group's
// fetch kernel follows in stream order)
__global__ __launch_bounds__(256) void k_torso_compose_group(TorsoGroupArgs g) {
    const TorsoLpArgs &a = g.a;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, f = blockIdx.y;
    const uint32_t n = blockIdx.x * 256u + tid;
    const bool in_frame = n < a.N;
    const size_t fn = (size_t)f * a.N + n;
    const uint32_t budget = (uint32_t)g.counters[(size_t)f * kTgCounterWords + kTgBudgetBase + kTgSlotBudget];
    const uint32_t job_pos0 = (uint32_t)g.counters[kTgBudgetBase + kTgSlotJobPos];
    uint32_t packed = 0;
    if (in_frame) {
        const bool masked = g.mask[n] != 0;
        const BudgetView bv{g.counters + (size_t)f * kTgCounterWords + kTgBudgetBase, g.snaps, a.N, g.max_steps};
        const RayAccum head = ray_state_final(a.state, bv, budget, (uint32_t)fn);
        float alpha = 0.0f, tcol[3] = {0.0f, 0.0f, 0.0f}, ddx = 0.0f, ddy = 0.0f;
        if (masked) {
            alpha = a.torso_alpha[fn];
            tcol[0] = a.torso_bg[3ull * fn]; tcol[1] = a.torso_bg[3ull * fn + 1]; tcol[2] = a.torso_bg[3ull * fn + 2];
            ddx = a.deform[2ull * fn]; ddy = a.deform[2ull * fn + 1];
        }
        // torso over background, head over torso (radnerf_torso.py:186-197) -- k_torso_lp's expressions
        const float T = 1.0f - head.wsum;
        const float hcol[3] = {head.r, head.g, head.b};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float bg = a.bg_color ? a.bg_color[3ull * n + c] : a.bg_scalar;
            const float tbg = tcol[c] * alpha + bg * (1.0f - alpha);
            const float v = clampf(hcol[c] + T * tbg, 0.0f, 1.0f);
            a.torso_bg[3ull * fn + c] = tbg;
            a.out_image[3ull * fn + c] = v;
            packed |= (uint32_t)(uint8_t)clampf(v * 255.0f, 0.0f, 255.0f) << (8 * c);
        }
        a.torso_alpha[fn] = alpha;
        a.deform[2ull * fn] = ddx;
        a.deform[2ull * fn + 1] = ddy;
        a.mask_out[fn] = masked ? 1 : 0;
        a.out_depth[fn] = fmaxf(head.depth - a.nears[fn], 0.0f) / (a.fars[fn] - a.nears[fn]);
    }
    if (a.job) {
        // the uint8 frame (k_torso_lp's store): four consecutive pixels hold 12 bytes = three dwords, assembled inside the lane quad
        const int q0 = (int)(lane & ~3u), ql = (int)(lane & 3u);
        const uint32_t w0 = (uint32_t)__shfl((int)packed, q0), w1 = (uint32_t)__shfl((int)packed, q0 + 1);
        const uint32_t w2 = (uint32_t)__shfl((int)packed, q0 + 2), w3 = (uint32_t)__shfl((int)packed, q0 + 3);
        const uint32_t pos = job_pos0 + f;
        if (in_frame && pos < a.job->n) {
            uint8_t *frame = a.job->out + (size_t)(pos % a.job->ring_frames) * a.job->frame_bytes;
            const uint32_t nq = n & ~3u;
            if (nq + 3u < a.N && (a.job->frame_bytes & 3ull) == 0ull) {
                const uint32_t d = ql == 0 ? (w0 | (w1 << 24)) : (ql == 1 ? ((w1 >> 8) | (w2 << 16)) : ((w2 >> 16) | (w3 << 8)));
                if (ql < 3) *reinterpret_cast<uint32_t *>(frame + 3ull * nq + 4u * (uint32_t)ql) = d;
            } else {
                frame[3ull * n] = (uint8_t)packed; frame[3ull * n + 1] = (uint8_t)(packed >> 8); frame[3ull * n + 2] = (uint8_t)(packed >> 16);
            }
        }
        if (blockIdx.x == 0 && f == 0 && tid == 0 && a.advance != 0xFFFFFFFFu) a.job->cursor[a.lane] = job_pos0 + (a.advance ? a.advance : a.job->lanes);
    }
}

}  // namespace gfpp

using namespace gfpp;

// The model / workspace checks and the argument record shared by the per-frame and the group entry.
static int torso_lp_args(const char *who, const gfpp_torso_model *m, const gfpp_frame_ws *ws, const float *bg_coords, const float *cond_in, const float *code,
                         const float *bg_color, float bg_scalar, uint32_t use_head, TorsoLpArgs &a) {
    if (!m->lp_weights || !m->lp_skinny || (m->lp_dtype != GFPP_F16 && m->lp_dtype != GFPP_BF16 && m->lp_dtype != GFPP_F32)) {
        set_error("%s: the model carries no MFMA weight image (lp_weights / lp_skinny / lp_dtype)", who);
        return GFPP_EINVAL;
    }
    if (m->grid.D != 2 || m->grid.L != 16 || m->grid.dtype != GFPP_F32 || m->grid.gridtype != 1 || m->grid.interp != 0 || m->grid.align_corners ||
        !m->grid.row_padded || !m->grid.levels_host) {
        set_error("%s: the torso grid must be a 16-level fp32 2-D tiled grid (padded table copy) with linear interpolation", who);
        return GFPP_EUNSUPPORTED;
    }
    for (int l = 0; l < 16; ++l)
        if (m->grid.levels_host[l].flags & GFPP_LEVEL_SLOW) { set_error("%s: level %d needs the generic lookup", who, l); return GFPP_EUNSUPPORTED; }
    const uint32_t enc_c = m->variant == 0 ? 54u : 126u;
    if (m->variant > 1 || m->const_dim != enc_c + m->code_dim || m->const_dim > (uint32_t)kTlMaxConst || (m->code_dim && !code)) {
        set_error("%s: inconsistent constant-column layout", who);
        return GFPP_EINVAL;
    }
    a.bg_coords = bg_coords; a.density_grid = m->density_grid; a.cond_in = cond_in; a.code = code;
    a.state = ws ? ws->ray_state : nullptr; a.nears = ws ? ws->nears : nullptr; a.fars = ws ? ws->fars : nullptr;
    a.bg_color = bg_color; a.bg_scalar = bg_scalar; a.shrink = m->torso_shrink; a.thresh = m->density_thresh;
    a.N = ws ? ws->N : 0u; a.G = m->grid_size; a.variant = m->variant; a.code_dim = m->code_dim; a.const_dim = m->const_dim;
    a.head_aware = m->head_aware; a.use_head = use_head;
    a.table = (const float *)m->grid.table; a.levels = m->grid.levels;
    a.def_w0_c = m->def_w0_c; a.can_w0_c = m->can_w0_c;
    a.ha_b0 = m->ha_b0; a.ha_b1 = m->ha_b1; a.ha_b2 = m->ha_b2;
    a.w16 = m->lp_weights; a.skinny16 = m->lp_skinny;
    a.bv = BudgetView{nullptr, nullptr, 0u, 0u};
    a.job = nullptr; a.lane = 0; a.sub = 0; a.advance = 0;
    return 0;
}

GFPP_API int gfpp_torso_frame_lp(const gfpp_torso_model *m, const gfpp_frame_ws *ws, const float *bg_coords, const float *cond_in,
                                 const float *code, const float *bg_color, float bg_scalar, uint32_t use_head, float *out_image,
                                 float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask, gfpp_stream_t stream) {
    if (!m || !ws || !bg_coords || !cond_in || !out_image || !out_depth || !torso_alpha || !torso_bg || !deform || !mask) {
        set_error("gfpp_torso_frame_lp: null argument");
        return GFPP_EINVAL;
    }
    TorsoLpArgs a;
    const int rc = torso_lp_args("gfpp_torso_frame_lp", m, ws, bg_coords, cond_in, code, bg_color, bg_scalar, use_head, a);
    if (rc) return rc;
    a.out_image = out_image; a.out_depth = out_depth; a.torso_alpha = torso_alpha; a.torso_bg = torso_bg; a.deform = deform; a.mask_out = mask;
    if (ws->defer_resolve) {
        if (!ws->snapshots || !ws->counters || ws->resolve_max_steps == 0 || ws->resolve_max_steps > 24u) { set_error("gfpp_torso_frame_lp: defer_resolve needs snapshots, counters and resolve_max_steps"); return GFPP_EINVAL; }
        a.bv = BudgetView{ws->gcounters ? ws->gcounters : ws->counters + 128, ws->snapshots, ws->gcounters ? ws->N_global : ws->N, ws->resolve_max_steps};
    }
    a.job = ws->clip_job; a.lane = ws->clip_lane; a.sub = ws->clip_sub; a.advance = ws->clip_advance;
    if (a.job && a.lane >= 8) { set_error("gfpp_torso_frame_lp: clip_lane must be < 8"); return GFPP_EINVAL; }
    const dim3 grid(div_up(ws->N, kTlThreads)), block(kTlThreads);
    if (m->lp_dtype == GFPP_BF16) hipLaunchKernelGGL(k_torso_lp<__bf16>, grid, block, 0, (hipStream_t)stream, a);
    else if (m->lp_dtype == GFPP_F32) hipLaunchKernelGGL(k_torso_lp<float>, grid, block, 0, (hipStream_t)stream, a);   // exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
    else hipLaunchKernelGGL(k_torso_lp<_Float16>, grid, block, 0, (hipStream_t)stream, a);
    return check_launch("gfpp_torso_frame_lp");
}

GFPP_API int gfpp_torso_fold_batch(const gfpp_torso_model *m, const float *cond_in, uint32_t cond_stride, const float *code, uint32_t frames, float *folded,
                                   gfpp_stream_t stream) {
    if (!m || !cond_in || !folded) { set_error("gfpp_torso_fold_batch: null argument"); return GFPP_EINVAL; }
    if (frames == 0) return 0;
    const uint32_t enc_c = m->variant == 0 ? 54u : 126u;
    if (m->variant > 1 || m->const_dim != enc_c + m->code_dim || m->const_dim > (uint32_t)kTlMaxConst || (m->code_dim && !code) || !m->def_w0_c || !m->can_w0_c) {
        set_error("gfpp_torso_fold_batch: inconsistent constant-column layout");
        return GFPP_EINVAL;
    }
    const TorsoFoldArgs f{cond_in, cond_stride, code, m->def_w0_c, m->can_w0_c, m->variant, m->code_dim, m->const_dim, folded};
    hipLaunchKernelGGL(k_torso_fold, dim3(frames), dim3(128), 0, (hipStream_t)stream, f);
    return check_launch("gfpp_torso_fold_batch");
}

GFPP_API int gfpp_torso_mask(const gfpp_torso_model *m, const float *bg_coords, uint32_t N, uint8_t *mask, gfpp_stream_t stream) {
    if (!m || !bg_coords || !mask || !m->density_grid) { set_error("gfpp_torso_mask: null argument"); return GFPP_EINVAL; }
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_torso_mask, dim3(div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, bg_coords, m->density_grid, m->grid_size, m->density_thresh, N, mask);
    return check_launch("gfpp_torso_mask");
}

static int torso_group_wgs_per_cu() {
    const int n = tuning().torso_group_wgs;                     // persistent workgroups of the MLP launch per CU (0 = default; experiments)
    return n <= 0 ? 3 : (n > 4 ? 4 : n);
}

GFPP_API int gfpp_torso_group_lp(const gfpp_torso_model *m, const gfpp_frame_ws *ws, const float *bg_coords, const float *folded, const float *code,
                                 const uint8_t *mask_static, const int32_t *masked_idx, uint32_t n_masked, const float *bg_color, float bg_scalar, uint32_t use_head,
                                 uint32_t max_steps, float *out_image, float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask,
                                 gfpp_stream_t stream) {
    if (!m || !ws || !bg_coords || !folded || !mask_static || (n_masked && !masked_idx) || !out_image || !out_depth || !torso_alpha || !torso_bg || !deform || !mask) {
        set_error("gfpp_torso_group_lp: null argument");
        return GFPP_EINVAL;
    }
    const uint32_t frames = ws->n_frames > 1u ? ws->n_frames : 1u;
    if (frames > kTgMaxFrames || ws->N == 0 || n_masked > ws->N || !ws->ray_state || !ws->nears || !ws->fars || !ws->counters || !ws->snapshots || ws->gcounters) {
        set_error("gfpp_torso_group_lp: needs a frame-group workspace (n_frames <= %u; ray_state, nears, fars, counters [n_frames, 192], snapshots) and n_masked <= N", kTgMaxFrames);
        return GFPP_EINVAL;
    }
    if (max_steps == 0 || max_steps > 24u) { set_error("gfpp_torso_group_lp: max_steps must be in 1..24"); return GFPP_EUNSUPPORTED; }
    if (m->lp_dtype != GFPP_F16 && m->lp_dtype != GFPP_BF16) { set_error("gfpp_torso_group_lp: 16-bit weight images only"); return GFPP_EUNSUPPORTED; }
    TorsoGroupArgs g;
    // (cond_in is not read by the group kernels: the constant columns arrive folded)
    const int rc = torso_lp_args("gfpp_torso_group_lp", m, ws, bg_coords, nullptr, code, bg_color, bg_scalar, use_head, g.a);
    if (rc) return rc;
    g.a.out_image = out_image; g.a.out_depth = out_depth; g.a.torso_alpha = torso_alpha; g.a.torso_bg = torso_bg; g.a.deform = deform; g.a.mask_out = mask;
    g.a.job = ws->clip_job; g.a.lane = ws->clip_lane; g.a.sub = 0; g.a.advance = ws->clip_advance;
    if (g.a.job && g.a.lane >= 8) { set_error("gfpp_torso_group_lp: clip_lane must be < 8"); return GFPP_EINVAL; }
    g.frames = frames; g.max_steps = max_steps;
    g.folded = folded; g.counters = ws->counters; g.snaps = ws->snapshots;
    g.mask = mask_static; g.masked = masked_idx; g.n_masked = n_masked;
    g.passes_per_frame = div_up(n_masked, 32u);
    const hipStream_t st = (hipStream_t)stream;
    {
        // (always launched, with one workgroup when no pixel is masked: it also computes the frames' step budgets and reads the job position for the compose launch)
        const int cus = cu_count();
        uint32_t grid = (uint32_t)cus * (uint32_t)torso_group_wgs_per_cu();
        const uint32_t need = div_up(frames * g.passes_per_frame, (uint32_t)kTlWaves);
        if (grid > need) grid = need;
        if (grid == 0u) grid = 1u;
        if (m->lp_dtype == GFPP_BF16) hipLaunchKernelGGL(k_torso_mlp_group<__bf16>, dim3(grid), dim3(kTlThreads), 0, st, g);
        else hipLaunchKernelGGL(k_torso_mlp_group<_Float16>, dim3(grid), dim3(kTlThreads), 0, st, g);
        const int rc2 = check_launch("gfpp_torso_group_lp (mlp)");
        if (rc2) return rc2;
    }
    hipLaunchKernelGGL(k_torso_compose_group, dim3(div_up(ws->N, 256u), frames), dim3(256), 0, st, g);
    return check_launch("gfpp_torso_group_lp");
}
