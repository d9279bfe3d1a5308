// march_device.h -- per-ray device routines shared by the stand-alone kernels (raymarch.hip) and the fused frame
// pipeline (frame_head.hip): slab test, occupancy-guided stepping, and the per-sample compositing update.
//
// Semantics follow the reference kernels kernel_near_far_from_aabb (raymarching.cu:91-145), kernel_march_rays
// (:827-929) and kernel_composite_rays (:942-1029); structure and code are our own.
#pragma once

#include <float.h>

#include "gfpp_common.h"

namespace gfpp {

struct RayBox {
    float near, far;
};

// Slab test of one ray against an axis-aligned box.  A miss reports near = far = FLT_MAX.
__device__ __forceinline__ RayBox ray_box(float ox, float oy, float oz, float dx, float dy, float dz, const float *__restrict__ aabb,
                                          float min_near) {
    const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
    RayBox r;
    float n = (aabb[0] - ox) * rdx, f = (aabb[3] - ox) * rdx;
    if (n > f) { float t = n; n = f; f = t; }
    float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
    if (ny > fy) { float t = ny; ny = fy; fy = t; }
    if (n > fy || ny > f) { r.near = r.far = FLT_MAX; return r; }
    if (ny > n) n = ny;
    if (fy < f) f = fy;
    float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
    if (nz > fz) { float t = nz; nz = fz; fz = t; }
    if (n > fz || nz > f) { r.near = r.far = FLT_MAX; return r; }
    if (nz > n) n = nz;
    if (fz < f) f = fz;
    if (n < min_near) n = min_near;
    r.near = n;
    r.far = f;
    return r;
}

// Constants of one march launch (identical for every ray).
struct MarchParams {
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf;
    uint32_t C, H;
    float half_H;       // 0.5 H
    uint32_t H_pow2;    // H is a power of two: the voxel coordinate's fp64 product is an exact scaling, done in fp32 (same bits)
    float rbound;       // 1 / bound, correctly rounded (the host's division): what the reference's `1 / mip_bound` gives in the outermost shell
};

__host__ __device__ inline MarchParams make_march_params(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    MarchParams p;
    p.bound = bound;
    p.dt_gamma = dt_gamma;
    p.dt_max = 2 * 1.7320508075688772f * (float)(1u << (C - 1)) / (float)H;
    const float alt = 2 * 1.7320508075688772f / (float)max_steps;
    p.dt_min = p.dt_max < alt ? p.dt_max : alt;
    p.rH = 1.0f / (float)H;
    p.H3 = (float)(H * H * H);
    p.Hf = (float)H;
    p.Cf = (float)C;
    p.C = C;
    p.H = H;
    p.half_H = 0.5f * (float)H;
    p.H_pow2 = (H & (H - 1u)) == 0u && H >= 2u ? 1u : 0u;
    p.rbound = 1.0f / bound;
    return p;
}

__device__ __forceinline__ int cascade_of(float x, float y, float z, float dt, const MarchParams &p) {
    if (p.C == 1) return 0;
    int e_pos, e_dt;
    frexpf(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), &e_pos);
    frexpf((float)((double)(dt * p.Hf) * 0.5), &e_dt);
    const int lp = (int)fminf(p.Cf - 1.0f, fmaxf(0.0f, (float)e_pos));
    const int ld = (int)fminf(p.Cf - 1.0f, fmaxf(0.0f, (float)e_dt));
    return lp > ld ? lp : ld;
}

// Voxel coordinate along one axis: fp64 product (the reference multiplies by the double literal 0.5), rounded to
// fp32, clamped, truncated.
__device__ __forceinline__ int voxel_of(float v, float rbound, const MarchParams &p) {
    // H a power of two (every shipped configuration: 128): 0.5 * a * H only changes a's exponent, in fp64 as in fp32 -- the fp32 product has the
    // bits of the rounded fp64 one, without three conversions and two quarter-rate multiplies per axis and step
    const float a = fmaf(v, rbound, 1.0f);
    const float g = p.H_pow2 ? a * p.half_H : (float)(0.5 * (double)a * (double)p.H);
    return (int)clampf(g, 0.0f, (float)(p.H - 1));
}

// Distance (in t) from position v to the face of voxel n that the ray leaves through, along one axis.
__device__ __forceinline__ float exit_distance(int n, float v, float d, float rd, float mip_bound, const MarchParams &p) {
    const float face = ((float)n + 0.5f + 0.5f * copysignf(1.0f, d)) * p.rH;
    return fmaf(fmaf(face, 2.0f, -1.0f), mip_bound, -v) * rd;
}

// One sample produced by the marcher.
struct Sample {
    float x, y, z, dt, t_end;
    float t0;   // t at which the sample was taken (t_end = t0 + dt)
};

// Advance one ray from t, emitting up to n_step occupied samples through `emit(step, Sample)`.
// Returns the number of samples emitted; `t` is left at the position after the last emitted sample (or >= far).
// ONE_SMALL_SHELL (the caller has checked p.C == 1 and p.H <= 256: every shipped model): a leaner probe with the same bits -- 1 / mip_bound without the IEEE
// division sequence (the only shell's bound is the scene bound, whose correctly rounded reciprocal comes from the host), the cell index is the Morton code itself
// (no float round trip through `level * H^3`), and the 8-bit Morton spread (no 32-bit multiplies): 164 -> 125 instructions per probe.  Measured and dropped in
// round 4, when the pre-march still ran UNDER the other lane's head launch; since the head launch holds every vector register of a SIMD (2 x 249-251) the frame
// group's prologue is on the critical path between two head launches (round 5 kernel trace: 71-79 us of every 880 us period) and the probe is what it executes.
template <bool ONE_SMALL_SHELL = false, typename Emit>
__device__ __forceinline__ uint32_t march_one_ray(float ox, float oy, float oz, float dx, float dy, float dz, float &t, float far,
                                                  uint32_t n_step, const uint8_t *__restrict__ bitfield, const MarchParams &p,
                                                  Emit &&emit) {
    const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        const float x = clampf(fmaf(t, dx, ox), -p.bound, p.bound);
        const float y = clampf(fmaf(t, dy, oy), -p.bound, p.bound);
        const float z = clampf(fmaf(t, dz, oz), -p.bound, p.bound);
        const float dt = clampf(t * p.dt_gamma, p.dt_min, p.dt_max);
        const int level = ONE_SMALL_SHELL ? 0 : cascade_of(x, y, z, dt, p);
        const float mip_bound = ONE_SMALL_SHELL ? fminf(1.0f, p.bound) : fminf(scalbnf(1.0f, level), p.bound);
        // (one cascade means bound <= 1: the shell's bound is the scene bound)
        const float mip_rbound = ONE_SMALL_SHELL ? (p.bound <= 1.0f ? p.rbound : 1.0f) : 1.0f / mip_bound;
        const int nx = voxel_of(x, mip_rbound, p), ny = voxel_of(y, mip_rbound, p), nz = voxel_of(z, mip_rbound, p);
        // the reference forms this index in fp32 (exact below 2^24); with one cascade the level is 0 and the index is the Morton code itself
        const uint32_t cell = ONE_SMALL_SHELL ? morton3_8((uint32_t)nx, (uint32_t)ny, (uint32_t)nz)
                                              : (uint32_t)fmaf((float)level, p.H3, (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const bool occupied = (bitfield[cell >> 3] >> (cell & 7u)) & 1u;
        if (occupied) {
            const float t0 = t;
            t += dt;
            Sample s{x, y, z, dt, t, t0};
            emit(step, s);
            ++step;
        } else {
            const float tx = exit_distance(nx, x, dx, rdx, mip_bound, p);
            const float ty = exit_distance(ny, y, dy, rdy, mip_bound, p);
            const float tz = exit_distance(nz, z, dz, rdz, mip_bound, p);
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                t += clampf(t * p.dt_gamma, p.dt_min, p.dt_max);
            } while (t < tt);
        }
    }
    return step;
}

// The same walk when the step is a CONSTANT and longer than a voxel can hold a ray: every chain point is probed, nothing else is computed.
//
// With dt_min == dt_max (make_march_params: max_steps <= H / 2^(C-1) -- the shipped max_steps 16 against H = 128) the marcher's `clamp(t dt_gamma, dt_min, dt_max)`
// is the constant dt_max whatever t is, so t walks ONE chain t_{k+1} = t_k + dt_max (the same fp32 addition in the occupied branch and in the empty branch's
// skip loop, raymarching.cu:873-912): which points of the chain are emitted depends on the grid, the chain does not.  The empty branch skips the chain points
// with t < tt = t_k + e, e = the distance to the voxel's exit face.  dt_max is DEFINED as the voxel diagonal (2 sqrt(3) 2^(C-1) / H), and e <= h / max_a |d_a|
// (h = the voxel's side, one axis at least has |d_a| >= the largest component), so with max_a |d_a| > (1 + 1e-4) / sqrt(3) the exit lies more than 1e-4 dt
// before the next chain point -- orders of magnitude beyond the rounding of the reference's fp32 expressions for e and tt (a few ulp of t): the skip loop runs
// exactly once, i.e. the reference probes every chain point too.  (Rays inside that margin of an exact voxel diagonal take the general walk.)  What is left of a
// probe is position, voxel, Morton code, bit: ~40 vector instructions instead of 125 -- k_group_begin is vector-ALU bound on exactly this loop.
__device__ __forceinline__ bool march_fixed_step_ok(float dx, float dy, float dz, const MarchParams &p) {
    return p.C == 1u && p.H <= 256u && p.dt_min == p.dt_max && fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz))) > 0.57741f;
}
template <typename Emit>
__device__ __forceinline__ uint32_t march_one_ray_fixed_step(float ox, float oy, float oz, float dx, float dy, float dz, float &t, float far, uint32_t n_step,
                                                             const uint8_t *__restrict__ bitfield, const MarchParams &p, Emit &&emit) {
    const float dt = p.dt_max;                          // == clampf(t * p.dt_gamma, p.dt_min, p.dt_max) for every t
    const float mip_rbound = p.bound <= 1.0f ? p.rbound : 1.0f;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        const float x = clampf(fmaf(t, dx, ox), -p.bound, p.bound);
        const float y = clampf(fmaf(t, dy, oy), -p.bound, p.bound);
        const float z = clampf(fmaf(t, dz, oz), -p.bound, p.bound);
        const int nx = voxel_of(x, mip_rbound, p), ny = voxel_of(y, mip_rbound, p), nz = voxel_of(z, mip_rbound, p);
        const uint32_t cell = morton3_8((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
        const float t0 = t;
        t += dt;
        if ((bitfield[cell >> 3] >> (cell & 7u)) & 1u) {
            Sample s{x, y, z, dt, t, t0};
            emit(step, s);
            ++step;
        }
    }
    return step;
}

// Running compositing state of one ray.
struct RayAccum {
    float wsum, depth, r, g, b;
};

// Per-ray state of the fused frame pipeline: ONE 32-byte record per ray {weights_sum, depth, r, g, b, t (or the sample cursor of the pre-marched
// kernels), -, -} instead of five arrays indexed by ray id.  A trip touches the rays of its alive list, i.e. scattered ids: with separate arrays
// every 4-byte store dirtied its own 32-byte sector (measured 4-8x write amplification, profiles/r01_fp16_pmc.md); a record is read with one
// 16-byte + one 8-byte load and written back as exactly one sector.
constexpr uint32_t kRayRec = 8;   // floats per record
__device__ __forceinline__ RayAccum ray_state_load(const float *__restrict__ state, uint32_t ray) {
    const float4 a = *reinterpret_cast<const float4 *>(state + (size_t)kRayRec * ray);
    return RayAccum{a.x, a.y, a.z, a.w, state[(size_t)kRayRec * ray + 4]};
}
__device__ __forceinline__ void ray_state_store(float *__restrict__ state, uint32_t ray, const RayAccum &acc, float t_or_cursor_bits) {
    *reinterpret_cast<float4 *>(state + (size_t)kRayRec * ray) = float4{acc.wsum, acc.depth, acc.r, acc.g};
    *reinterpret_cast<float2 *>(state + (size_t)kRayRec * ray + 4) = float2{acc.b, t_or_cursor_bits};
}
__device__ __forceinline__ float ray_state_t(const float *__restrict__ state, uint32_t ray) { return state[(size_t)kRayRec * ray + 5]; }

// Fold one sample into the accumulators.  Returns true if the ray must stop AFTER this sample (the test uses the
// transmittance from before the sample, as the reference does).
__device__ __forceinline__ bool composite_sample(RayAccum &a, float sigma, float dt, float t_end, float cr, float cg, float cb,
                                                 float T_thresh) {
    const float alpha = 1.0f - __expf(-sigma * dt);
    const float T = 1.0f - a.wsum;
    const float w = alpha * T;
    a.wsum += w;
    a.depth = fmaf(w, t_end, a.depth);
    a.r = fmaf(w, cr, a.r);
    a.g = fmaf(w, cg, a.g);
    a.b = fmaf(w, cb, a.b);
    return T < T_thresh;
}


// renderer.py:359-364,384 replayed on the histogram: alive rays at the start of every trip and the step budget.  Every thread that calls this
// computes the same numbers from <= max_steps + 8 cached loads.
__device__ __forceinline__ uint32_t budget_from_hist(const int32_t *__restrict__ hist, uint32_t N_global, uint32_t max_steps, int32_t *counters_out) {
    uint32_t S = 0, gone = 0, alive = N_global, trip = 0;
    while (S < max_steps && alive > 0u) {
        if (counters_out) counters_out[trip] = (int32_t)alive;
        uint32_t n = N_global / alive;
        n = n < 1u ? 1u : (n > 8u ? 8u : n);
        for (uint32_t j = S; j < S + n; ++j) gone += (uint32_t)hist[j];     // rays whose m lies inside this window are dead after it
        S += n;
        alive = N_global - gone;
        ++trip;
    }
    if (counters_out) counters_out[trip] = (int32_t)alive;   // what the last trip appended for a next one (0 when the loop ended for lack of rays)
    return S;
}


// What a consumer of the ray records needs to finish the persistent launch's job on the fly (gfpp_frame_ws.defer_resolve): the budget histogram and
// the snapshots.  hist == nullptr: the records are final already.
struct BudgetView {
    const int32_t *hist;     // [32] rays by end point (frame-wide)
    const float *snaps;      // [N, 7, 5]
    uint32_t N_global, max_steps;
};

// The final {weights_sum, depth, r, g, b} of ray n: its record, or -- if it composited more samples than the step budget B allows -- the snapshot
// after B samples (k_head_budget_resolve does the same in place).
__device__ __forceinline__ RayAccum ray_state_final(const float *__restrict__ state, const BudgetView &bv, uint32_t B, uint32_t n) {
    RayAccum acc = ray_state_load(state, n);
    if (bv.hist) {
        const uint32_t done = __float_as_uint(state[(size_t)kRayRec * n + 7]);
        if (done > B && B >= bv.max_steps && done <= bv.max_steps + 7u) {
            const float *sp = bv.snaps + ((size_t)n * 7u + (B - bv.max_steps)) * 5u;
            acc = RayAccum{sp[0], sp[1], sp[2], sp[3], sp[4]};
        }
    }
    return acc;
}

}  // namespace gfpp
