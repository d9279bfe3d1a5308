// train.hip -- training-side kernels behind the reference's `_raymarching_face` and `_gridencoder` extension APIs:
//   march_rays_train (+ backward), composite_rays_train forward / backward, morton3D_dilation, sph_from_ray        raymarching.cu:162-820
//   grid_encode dy_dx, grid_encode_backward (table gradient + input gradient), grad_total_variation                 gridencoder.cu:198-368, 505-609
// One thread per ray / per (point, level); 256-thread workgroups.  Arithmetic follows the reference expression by expression with the
// same explicit-fmaf policy as the inference kernels, so the CPU restatement used by the tests agrees bit for bit wherever no
// atomic summation order is involved (table gradients are sums of atomics: equal up to fp32 reassociation).
#include <hip/hip_runtime.h>

#include "grid_device.h"
#include "march_device.h"

namespace gfpp {

constexpr int kTrBlock = 256;

// ---- march_rays_train: two passes per ray (count, then write), ranges handed out with atomics -----------------------------------
__global__ __launch_bounds__(kTrBlock) void k_march_rays_train(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                              const uint8_t *__restrict__ bitfield, MarchParams mp, uint32_t max_steps, uint32_t N,
                                                              uint32_t M, const float *__restrict__ nears, const float *__restrict__ fars,
                                                              float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                              int32_t *__restrict__ rays, int32_t *__restrict__ counter,
                                                              const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = n < N;             // every lane of a wavefront takes part in the range allocation below
    float ox = 0.0f, oy = 0.0f, oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 1.0f, far = 0.0f, t0 = 0.0f;
    uint32_t num_steps = 0;
    if (active) {
        const float *o = rays_o + 3ull * n, *d = rays_d + 3ull * n;
        ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
        far = fars[n];
        t0 = nears[n];
        t0 = fmaf(clampf(t0 * mp.dt_gamma, mp.dt_min, mp.dt_max), noises[n], t0);
        float t = t0;
        num_steps = march_one_ray(ox, oy, oz, dx, dy, dz, t, far, max_steps, bitfield, mp, [](uint32_t, const Sample &) {});
    }
    // The reference takes two device atomics per RAY (raymarching.cu:446-447).  Here a wavefront sums its rays' sample counts (shuffle scan) and
    // takes ONE pair of atomics: its 64 rays get consecutive ray slots and back-to-back sample ranges, which is also what makes the second pass's
    // writes and the compositor's reads of neighbouring rays land next to each other.  Same contract as the reference: rays[k] = (ray id,
    // offset, count) in an order that depends on scheduling; a ray whose range would end beyond M keeps its entry and writes nothing.
    uint32_t incl = num_steps;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    const uint32_t wave_total = (uint32_t)__shfl((int)incl, 63);
    const unsigned long long act = __ballot(active);
    uint32_t base_pt = 0, base_ray = 0;
    if (lane == 0) {
        base_pt = (uint32_t)atomicAdd(&counter[0], (int)wave_total);
        base_ray = (uint32_t)atomicAdd(&counter[1], (int)__popcll(act));
    }
    base_pt = (uint32_t)__shfl((int)base_pt, 0);
    base_ray = (uint32_t)__shfl((int)base_ray, 0);
    if (!active) return;
    const uint32_t point_index = base_pt + incl - num_steps;
    const uint32_t ray_index = base_ray + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
    rays[3ull * ray_index] = (int32_t)n;
    rays[3ull * ray_index + 1] = (int32_t)point_index;
    rays[3ull * ray_index + 2] = (int32_t)num_steps;
    if (num_steps == 0 || point_index + num_steps > M) return;
    float *px = xyzs + 3ull * point_index, *pd = dirs + 3ull * point_index, *pt = deltas + 2ull * point_index;
    float t = t0;
    march_one_ray(ox, oy, oz, dx, dy, dz, t, far, num_steps, bitfield, mp, [&](uint32_t s, const Sample &smp) {
        px[3 * s] = smp.x; px[3 * s + 1] = smp.y; px[3 * s + 2] = smp.z;
        pd[3 * s] = dx; pd[3 * s + 1] = dy; pd[3 * s + 2] = dz;
        pt[2 * s] = smp.dt; pt[2 * s + 1] = smp.t_end;
    });
}

__global__ __launch_bounds__(kTrBlock) void k_march_rays_train_backward(const float *__restrict__ grad_xyzs, const float *__restrict__ grad_dirs,
                                                                       const int32_t *__restrict__ rays, const float *__restrict__ deltas, uint32_t N,
                                                                       uint32_t M, float *__restrict__ grad_rays_o, float *__restrict__ grad_rays_d) {
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    float go[3], gd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { go[k] = grad_rays_o[3ull * n + k]; gd[k] = grad_rays_d[3ull * n + k]; }
    for (uint32_t s = 0; s < num_steps; ++s) {
        const float *gx = grad_xyzs + 3ull * (offset + s), *gdir = grad_dirs + 3ull * (offset + s);
        const float tend = deltas[2ull * (offset + s) + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            go[k] += gx[k];
            gd[k] += fmaf(gx[k], tend, gdir[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { grad_rays_o[3ull * n + k] = go[k]; grad_rays_d[3ull * n + k] = gd[k]; }
}

// ---- composite_rays_train ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTrBlock) void k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                                 const float *__restrict__ ambient, const float *__restrict__ deltas,
                                                                 const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh,
                                                                 float *__restrict__ weights_sum, float *__restrict__ ambient_sum,
                                                                 float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, ws = 0.0f, d = 0.0f, amb = 0.0f;
    if (!(num_steps == 0 || offset + num_steps > M)) {
        for (uint32_t s = 0; s < num_steps; ++s) {
            const size_t k = offset + s;
            const float alpha = 1.0f - __expf(-sigmas[k] * deltas[2 * k]);
            const float weight = alpha * T;
            r = fmaf(weight, rgbs[3 * k], r);
            g = fmaf(weight, rgbs[3 * k + 1], g);
            b = fmaf(weight, rgbs[3 * k + 2], b);
            d = fmaf(weight, deltas[2 * k + 1], d);
            ws += weight;
            amb += ambient[k];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
        }
    }
    weights_sum[index] = ws;
    ambient_sum[index] = amb;
    depth[index] = d;
    image[3ull * index] = r; image[3ull * index + 1] = g; image[3ull * index + 2] = b;
}

__global__ __launch_bounds__(kTrBlock) void k_composite_train_bwd(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad_ambient_sum,
                                                                 const float *__restrict__ grad_image, const float *__restrict__ sigmas,
                                                                 const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                                 const int32_t *__restrict__ rays, const float *__restrict__ weights_sum,
                                                                 const float *__restrict__ image, uint32_t M, uint32_t N, float T_thresh,
                                                                 float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs,
                                                                 float *__restrict__ grad_ambient) {
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3ull * n], offset = (uint32_t)rays[3ull * n + 1], num_steps = (uint32_t)rays[3ull * n + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index], gamb = grad_ambient_sum[index];
    const float gi0 = grad_image[3ull * index], gi1 = grad_image[3ull * index + 1], gi2 = grad_image[3ull * index + 2];
    const float r_final = image[3ull * index], g_final = image[3ull * index + 1], b_final = image[3ull * index + 2], ws_final = weights_sum[index];
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t s = 0; s < num_steps; ++s) {
        const size_t k = offset + s;
        const float c0 = rgbs[3 * k], c1 = rgbs[3 * k + 1], c2 = rgbs[3 * k + 2], dt = deltas[2 * k];
        const float alpha = 1.0f - __expf(-sigmas[k] * dt);
        const float weight = alpha * T;
        r = fmaf(weight, c0, r);
        g = fmaf(weight, c1, g);
        b = fmaf(weight, c2, b);
        T *= 1.0f - alpha;
        grad_rgbs[3 * k] = gi0 * weight; grad_rgbs[3 * k + 1] = gi1 * weight; grad_rgbs[3 * k + 2] = gi2 * weight;
        grad_ambient[k] = gamb;
        float acc = gi0 * fmaf(T, c0, -(r_final - r));
        acc = fmaf(gi1, fmaf(T, c1, -(g_final - g)), acc);
        acc = fmaf(gi2, fmaf(T, c2, -(b_final - b)), acc);
        acc = fmaf(gws, 1.0f - ws_final, acc);
        grad_sigmas[k] = dt * acc;
        if (T < T_thresh) break;
    }
}

// ---- density-grid upkeep ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTrBlock) void k_morton_dilation(const float *__restrict__ grid, uint32_t C, uint32_t H, float *__restrict__ out) {
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    if (n >= C * H3) return;
    const uint32_t c = n / H3, ind = n - c * H3;
    const uint32_t x = compact3(ind), y = compact3(ind >> 1), z = compact3(ind >> 2);
    const float *g = grid + (size_t)c * H3;
    float res = grid[n];
    if (x + 1 < H) res = fmaxf(res, g[morton3(x + 1, y, z)]);
    if (x > 0) res = fmaxf(res, g[morton3(x - 1, y, z)]);
    if (y + 1 < H) res = fmaxf(res, g[morton3(x, y + 1, z)]);
    if (y > 0) res = fmaxf(res, g[morton3(x, y - 1, z)]);
    if (z + 1 < H) res = fmaxf(res, g[morton3(x, y, z + 1)]);
    if (z > 0) res = fmaxf(res, g[morton3(x, y, z - 1)]);
    out[n] = res;
}

__global__ __launch_bounds__(kTrBlock) void k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius, uint32_t N,
                                                          float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kTrBlock + threadIdx.x;
    if (n >= N) return;
    constexpr float RPI = 0.3183098861837907f;
    const float ox = rays_o[3ull * n], oy = rays_o[3ull * n + 1], oz = rays_o[3ull * n + 2];
    const float dx = rays_d[3ull * n], dy = rays_d[3ull * n + 1], dz = rays_d[3ull * n + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float Cc = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
    const float t = (-B + sqrtf(fmaf(B, B, -(A * Cc)))) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[2ull * n] = fmaf(2.0f * theta, RPI, -1.0f);
    coords[2ull * n + 1] = phi * RPI;
}

// ---- grid encoder: per-(point, level) fractional position shared by dy_dx / backward / TV ---------------------------------------------
struct TrLevels {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

template <int D>
__device__ __forceinline__ bool tr_locate(const float *__restrict__ in, float scale, bool align_corners, uint32_t interp, float (&pos)[D],
                                          float (&deriv)[D], uint32_t (&pg)[D]) {
    // all coordinates first, then the range test without short-circuit: `a && b` on values that are loaded inside the expression compiles to a branch
    // per coordinate with the load behind it -- D memory round trips in a row per point
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = in[d];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) inside = inside & !(x[d] < 0.0f || x[d] > 1.0f);
    if (!inside) return false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float p = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        p -= (float)pg[d];
        if (interp == 1) { deriv[d] = 6.0f * p * (1.0f - p); p = p * p * fmaf(-2.0f, p, 3.0f); }
        else deriv[d] = 1.0f;
        pos[d] = p;
    }
    return true;
}

// A level's index arithmetic resolved once (what gfpp_grid_levels_fill does on the host for the frame kernels): the loop of get_grid_index (gridencoder.cu:66-83)
// keeps a dimension while stride <= size, and `index % size` is an AND for a power-of-two size and nothing at all where every index provably stays below the size.
// grid_row per corner -- three multiplies, the dimension tests and a 32-bit modulo by a run-time value, eight times per point and level -- was most of the training
// kernels' per-point instructions; hash-addressed / true-modulo levels (slow) keep it.  Wave-uniform: the fields live in scalar registers.
struct TrIndex {
    uint32_t st[3], mask;
    bool slow;
};
template <int D>
__device__ __forceinline__ TrIndex tr_index(uint32_t size, uint32_t res, uint32_t gridtype, bool align_corners) {
    TrIndex ix;
    const uint32_t r1 = align_corners ? res : res + 1u;
    uint32_t stride = 1;
    ix.st[0] = ix.st[1] = ix.st[2] = 0u;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (stride <= size) { ix.st[d] = stride; stride *= r1; }
    const bool pow2 = size != 0u && (size & (size - 1u)) == 0u, hashed = gridtype == 0u && stride > size, exact = stride <= size;
    ix.slow = hashed || (!pow2 && (!exact || align_corners));
    ix.mask = pow2 ? size - 1u : 0xFFFFFFFFu;
    return ix;
}
template <int D>
__device__ __forceinline__ uint32_t tr_base_row(const TrIndex &ix, const uint32_t (&pg)[D]) {
    uint32_t base = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) base += pg[d] * ix.st[d];
    return base;
}
// row of corner idx (bit d set: the +1 neighbour along d) of the cell at pg -- the value grid_row returns for it
template <int D>
__device__ __forceinline__ uint32_t tr_corner_row(const TrIndex &ix, uint32_t base, const uint32_t (&pg)[D], int idx, uint32_t gridtype, bool align_corners, uint32_t size,
                                                  uint32_t res) {
    if (!ix.slow) {
        uint32_t row = base;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (idx & (1 << d)) row += ix.st[d];
        return row & ix.mask;
    }
    uint32_t pl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pl[d] = pg[d] + ((idx >> d) & 1u);
    return grid_row<D>(pl, gridtype, align_corners, size, res);
}

// dy_dx [B, L, D, C] (gridencoder.cu:198-243)
template <int D, int C>
__global__ __launch_bounds__(kTrBlock) void k_grid_dydx(const float *__restrict__ inputs, const float *__restrict__ table, const int32_t *__restrict__ offsets,
                                                       float *__restrict__ dy_dx, uint32_t B, uint32_t L, TrLevels lv, uint32_t gridtype,
                                                       bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * kTrBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float *out = dy_dx + ((size_t)b * L + level) * D * C;
    float pos[D], deriv[D];
    uint32_t pg[D];
    const float scale = lv.scale[level];
    if (!tr_locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, pos, deriv, pg)) {
#pragma unroll
        for (int i = 0; i < D * C; ++i) out[i] = 0.0f;
        return;
    }
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
    const float *grid = table + (size_t)off * C;
#pragma unroll
    for (int gd = 0; gd < D; ++gd) {
        float rg[C];
#pragma unroll
        for (int c = 0; c < C; ++c) rg[c] = 0.0f;
#pragma unroll
        for (int idx = 0; idx < (1 << (D - 1)); ++idx) {
            float w = scale;
            uint32_t pl[D];
#pragma unroll
            for (int nd = 0; nd < D - 1; ++nd) {
                const int d = (nd >= gd) ? nd + 1 : nd;
                if ((idx & (1 << nd)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1u; }
            }
            pl[gd] = pg[gd];
            const uint32_t il = grid_row<D>(pl, gridtype, align_corners, size, res);
            pl[gd] = pg[gd] + 1u;
            const uint32_t ir = grid_row<D>(pl, gridtype, align_corners, size, res);
#pragma unroll
            for (int c = 0; c < C; ++c) rg[c] = fmaf(w * (grid[(size_t)ir * C + c] - grid[(size_t)il * C + c]), deriv[gd], rg[c]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[gd * C + c] = rg[c];
    }
}

// table gradient: grad [L, B, C] scattered into grad_table with atomics (gridencoder.cu:247-340).
// Two kernels share the levels.  A coarse level is a small table that EVERY point hits (level 0 of the 3-D grid: 4 920 rows for ~10^6 samples x 8
// corners), so device atomics on it serialise on a few thousand addresses: levels whose table fits the LDS (<= kLdsGradFloats values) are
// accumulated per workgroup in LDS (ds_add_f32) over a long run of points and flushed with one device atomic per touched value; the fine levels,
// where collisions are rare, keep the direct scatter.  Which kernel owns a level is decided on the device from offsets[] (both are launched
// over all levels; the wrong one returns at once).
//
// Where the device atomics land: a gradient table that all eight XCDs scatter into ping-pongs its cache lines between the XCDs' L2s (measured:
// 10.8 G atomics/s for the 14 fine levels).  With `xcd_copies` != NULL every workgroup adds into the private copy of the XCD it runs on
// (HW_REG_XCC_ID, so the copy choice never depends on a dispatch-order assumption) -- a line of copy k is only ever owned by XCD k's L2 -- and
// k_grid_reduce_xcd folds the eight copies into grad_table afterwards.
constexpr uint32_t kXcds = 8;
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & (kXcds - 1u); }   // hwreg(HW_REG_XCC_ID, 0, 4)

constexpr uint32_t kLdsGradFloats = 32768;     // 128 KiB of the CU's 160 KiB
constexpr uint32_t kLdsGradPoints = 8192;      // points per workgroup of the LDS kernel

template <int D, int C, typename G>
__device__ __forceinline__ void grid_backward_point(const G *__restrict__ grad, const float *__restrict__ inputs, uint32_t b, uint32_t B, uint32_t level, float scale,
                                                    uint32_t size, uint32_t res, uint32_t gridtype, bool align_corners, uint32_t interp, float *gg) {
    float pos[D], deriv[D];
    uint32_t pg[D];
    if (!tr_locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, pos, deriv, pg)) return;
    float gc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gc[c] = (float)grad[((size_t)level * B + b) * C + c];
#pragma unroll
    for (int idx = 0; idx < (1 << D); ++idx) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if ((idx & (1 << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        const uint32_t row = grid_row<D>(pl, gridtype, align_corners, size, res);
#pragma unroll
        for (int c = 0; c < C; ++c) atomicAdd(&gg[(size_t)row * C + c], w * gc[c]);
    }
}

template <int D, int C, typename G = float>
__global__ __launch_bounds__(kTrBlock) void k_grid_backward(const G *__restrict__ grad, const float *__restrict__ inputs,
                                                           const int32_t *__restrict__ offsets, float *__restrict__ grad_table, uint32_t B, uint32_t L,
                                                           TrLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t lds_floats,
                                                           float *__restrict__ xcd_copies, uint32_t total_floats) {
    // Points arrive ray-major (consecutive samples of one ray are neighbours in space and hit the SAME table rows at the coarse and middle levels):
    // handing neighbouring points to neighbouring lanes makes the 64 atomics of a wavefront instruction queue up on a handful of addresses.  The
    // point index is therefore transposed -- lane l of the i-th wavefront takes point l * ceil(B / 64) + i -- so that one instruction's atomics go
    // to 64 different places.
    const uint32_t i = blockIdx.x * kTrBlock + threadIdx.x;
    const uint32_t chunk = (B + 63u) / 64u;
    const uint32_t b = (i & 63u) * chunk + (i >> 6);
    if ((i >> 6) >= chunk || b >= B) return;
    const uint32_t level = blockIdx.y;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
    if (size * C <= lds_floats) return;                        // owned by k_grid_backward_lds
    float *dst = xcd_copies ? xcd_copies + (size_t)xcc_id() * total_floats : grad_table;
    grid_backward_point<D, C, G>(grad, inputs, b, B, level, lv.scale[level], size, res, gridtype, align_corners, interp, dst + (size_t)off * C);
}

// grad_table[i] += sum over the eight XCD copies
__global__ __launch_bounds__(kTrBlock) void k_grid_reduce_xcd(const float *__restrict__ copies, float *__restrict__ grad_table, uint32_t total_floats) {
    const uint32_t i = blockIdx.x * kTrBlock + threadIdx.x;
    if (i >= total_floats) return;
    float s = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < kXcds; ++k) s += copies[(size_t)k * total_floats + i];
    if (s != 0.0f) grad_table[i] += s;
}

template <int D, int C, typename G = float>
__global__ __launch_bounds__(kTrBlock) void k_grid_backward_lds(const G *__restrict__ grad, const float *__restrict__ inputs,
                                                               const int32_t *__restrict__ offsets, float *__restrict__ grad_table, uint32_t B, uint32_t L,
                                                               TrLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp,
                                                               float *__restrict__ xcd_copies, uint32_t total_floats) {
    extern __shared__ float acc[];
    const uint32_t level = blockIdx.y;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
    const uint32_t n = size * C;
    if (n > kLdsGradFloats) return;                            // owned by k_grid_backward
    for (uint32_t i = threadIdx.x; i < n; i += kTrBlock) acc[i] = 0.0f;
    __syncthreads();
    const uint32_t first = blockIdx.x * kLdsGradPoints, last = first + kLdsGradPoints < B ? first + kLdsGradPoints : B;
    for (uint32_t b = first + threadIdx.x; b < last; b += kTrBlock)
        grid_backward_point<D, C, G>(grad, inputs, b, B, level, lv.scale[level], size, res, gridtype, align_corners, interp, acc);
    __syncthreads();
    float *gg = (xcd_copies ? xcd_copies + (size_t)xcc_id() * total_floats : grad_table) + (size_t)off * C;
    for (uint32_t i = threadIdx.x; i < n; i += kTrBlock) {
        const float v = acc[i];
        if (v != 0.0f) atomicAdd(&gg[i], v);
    }
}

// ---- every level without device atomics: range passes -------------------------------------------------------------------------------------------------
// The direct scatter above is bound by the device's atomic rate (measured 14.6 G float atomics/s with XCD-private copies: 4.6 ms for the 67 M atomics
// of one May grid, 45 % of a training step).  Here a workgroup owns ONE range of a level's table and one eighth of the points: it walks its points,
// recomputes the corners (cheap: ~200 instructions per point and level) and adds the ones that fall into its range into LDS accumulators; the range
// then leaves as plain coalesced stores into the slice's private copy of the gradient (the `xcd_copies` scratch: slice s owns copy s;
// k_grid_reduce_xcd sums the copies).  A 2^16-row level is eight ranges, i.e. its corners are located eight times -- still ~10x cheaper than an atomic
// each.  grad: fp32 or half ([L, B, C]).
//
// The LDS accumulators are 64-bit FIXED POINT: ds_add_f32 turned out to run at ~1 lane per 5 cycles on gfx950 (the range kernel took 877 us with float
// LDS atomics, 167 us with the adds replaced by plain stores, 177 us with ds_add_u64), integer LDS atomics at full rate.  A level's values are scaled by
// 2^(36 - e) with 2^e > max |grad| of that level (k_grid_grad_levelmax), so one contribution is below 2^36 and 2^27 of them fit (16 M points hitting
// one entry with all eight corners); what is kept of a contribution reaches 36 bits below the level's largest gradient (fp32 keeps 24 bits below each value; the reference's half accumulators under amp 11,
// and nothing below 6e-8).  A non-finite gradient anywhere in the level makes the whole level NaN, so that a GradScaler still sees the overflow.
constexpr uint32_t kRgThreads = 1024;
constexpr uint32_t kRgSlices = kXcds;          // point slices = gradient copies
constexpr uint32_t kRgValues = kLdsGradFloats * sizeof(float) / sizeof(long long);   // accumulators of one range (16 384)

// max |grad| per level as float bits (non-negative floats order like unsigned integers; NaN bit patterns are above +inf's): out[level], zeroed before
template <typename G>
__global__ __launch_bounds__(kTrBlock) void k_grid_grad_levelmax(const G *__restrict__ grad, uint32_t per_level, uint32_t *__restrict__ out) {
    const uint32_t level = blockIdx.y;
    const G *g = grad + (size_t)level * per_level;
    uint32_t m = 0;
    auto take = [&](G v) {
        const uint32_t bits = __float_as_uint(fabsf((float)v));
        m = bits > m ? bits : m;
    };
    // 16-byte vectors over the aligned interior of the level (a level starts wherever B C elements put it), single elements in front of and behind it: the
    // element-wise loop read 2 bytes per lane and took 56 us for a May step's 19 MB
    constexpr uint32_t VE = 16u / (uint32_t)sizeof(G);
    const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(g) & 15u) / sizeof(G));
    uint32_t head = mis ? VE - mis : 0u;
    head = head < per_level ? head : per_level;
    const uint32_t nv = (per_level - head) / VE, tail0 = head + nv * VE;
    const uint32_t t = blockIdx.x * kTrBlock + threadIdx.x, nt = gridDim.x * kTrBlock;
    if (t < head) take(g[t]);
    if (t < per_level - tail0) take(g[tail0 + t]);
    const uint4 *gv = reinterpret_cast<const uint4 *>(g + head);
    for (uint32_t i = t; i < nv; i += nt) {
        const uint4 q = gv[i];
        G e[VE];
        __builtin_memcpy(e, &q, 16);
#pragma unroll
        for (uint32_t k = 0; k < VE; ++k) take(e[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)m, o);
        m = other > m ? other : m;
    }
    // one atomic per WORKGROUP: the levels' maxima share a cache line, and an atomic per wavefront (4 096 of them on one L2 line) was most of this kernel's 50 us
    __shared__ uint32_t wave_max[kTrBlock / 64];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (uint32_t w = 1; w < kTrBlock / 64; ++w) m = wave_max[w] > m ? wave_max[w] : m;
        if (m) atomicMax(&out[level], m);
    }
}

// ---- which points a range needs: per level, every point is put on the list of each range that one of its corners falls into (round 6) ---------------------------
// The range kernel below walks, per (level, range), ALL points and keeps the corners that land in its range: a 2^16-row level is eight ranges, so a point's cell is
// located eight times there although its eight corners touch two or three of them (rows r, r + 1, r + sy, r + sy + 1 of one z plane are at most a row stride apart,
// the other plane is sz rows further, modulo the level size).  One pass per level finds the ranges of a point's corners once (a bit mask of <= 32 ranges) and appends
// the point's index to those ranges' lists; the range kernel then walks its list.  Lists have room for every point (a point is on a list at most once) and are
// filled through one LDS counter per range and workgroup and one device atomic per range and workgroup; their order is whatever the atomics give -- the range
// kernel's accumulators are integers, the sums do not depend on it.  Levels of more than 32 ranges (tables beyond 2^18 rows at level_dim 2) are marked
// "no list" (count 0xFFFFFFFF) and walked the old way.
constexpr uint32_t kBinMaxRanges = 32;
template <int D, int C>
__global__ __launch_bounds__(1024) void k_grid_bin_points(const float *__restrict__ inputs, const int32_t *__restrict__ offsets, uint32_t B, TrLevels lv, uint32_t gridtype,
                                                          bool align_corners, uint32_t *__restrict__ counts, uint32_t *__restrict__ lists) {
    __shared__ uint32_t s_cnt[kBinMaxRanges], s_base[kBinMaxRanges];
    const uint32_t level = blockIdx.y, b = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
    constexpr uint32_t rows_per = kLdsGradFloats * (uint32_t)sizeof(float) / (uint32_t)sizeof(long long) / C;
    const uint32_t nr = (size + rows_per - 1u) / rows_per;
    if (nr > kBinMaxRanges) {
        if (blockIdx.x == 0 && threadIdx.x < kBinMaxRanges) counts[level * kBinMaxRanges + threadIdx.x] = 0xFFFFFFFFu;
        return;
    }
    uint32_t item0 = 0;                                            // the level's first range among all ranges (the range kernel's item numbering); scalar loop
    for (uint32_t l = 0; l < level; ++l) item0 += ((uint32_t)(offsets[l + 1] - offsets[l]) + rows_per - 1u) / rows_per;
    if (threadIdx.x < kBinMaxRanges) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t mask = 0u, rank[1 << D];
    if (b < B) {
        float pos[D], deriv[D];
        uint32_t pg[D];
        if (tr_locate<D>(inputs + (size_t)b * D, lv.scale[level], align_corners, 0u, pos, deriv, pg)) {      // (the cell does not depend on the interpolation)
            const TrIndex ix = tr_index<D>(size, res, gridtype, align_corners);
            const uint32_t base = tr_base_row<D>(ix, pg);
#pragma unroll
            for (int idx = 0; idx < (1 << D); ++idx) mask |= 1u << (tr_corner_row<D>(ix, base, pg, idx, gridtype, align_corners, size, res) / rows_per);
        }
    }
    {
        uint32_t m = mask;
#pragma unroll
        for (int k = 0; k < (1 << D); ++k) {
            if (m) {
                const uint32_t r = (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1u;
                rank[k] = atomicAdd(&s_cnt[r], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < nr && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[level * kBinMaxRanges + threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    {
        uint32_t m = mask;
#pragma unroll
        for (int k = 0; k < (1 << D); ++k) {
            if (m) {
                const uint32_t r = (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1u;
                lists[(size_t)(item0 + r) * B + s_base[r] + rank[k]] = b;
            }
        }
    }
}

template <int D, int C, typename G>
__global__ __launch_bounds__(kRgThreads) void k_grid_backward_ranges(const G *__restrict__ grad, const float *__restrict__ inputs, const int32_t *__restrict__ offsets,
                                                                     uint32_t B, uint32_t L, TrLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp,
                                                                     float *__restrict__ copies, uint32_t total_floats, const uint32_t *__restrict__ levelmax,
                                                                     const uint32_t *__restrict__ bin_counts, const uint32_t *__restrict__ bin_lists) {
    extern __shared__ long long rg_acc[];
    long long *acc = rg_acc;
    uint32_t item = blockIdx.x / kRgSlices;
    const uint32_t item_global = item;
    const uint32_t slice = blockIdx.x % kRgSlices;
    uint32_t level = 0, range = 0, off = 0, size = 0;
    bool found = false;
    for (; level < L; ++level) {                                   // (scalar loop: <= 32 levels)
        off = (uint32_t)offsets[level];
        size = (uint32_t)offsets[level + 1] - off;
        const uint32_t nr = (size * C + kRgValues - 1u) / kRgValues;
        if (item < nr) { range = item; found = true; break; }
        item -= nr;
    }
    if (!found) return;                                            // the grid is sized by an upper bound of the ranges
    const uint32_t rows_per = kRgValues / C, row0 = range * rows_per;
    const uint32_t nrows = size - row0 < rows_per ? size - row0 : rows_per, n = nrows * C;
    for (uint32_t i = threadIdx.x; i < n; i += kRgThreads) acc[i] = 0ll;
    __syncthreads();
    // fixed-point scale of this level: contributions |w g| <= max |g| < 2^e  ->  |w g| 2^(36 - e) < 2^36
    const uint32_t maxbits = levelmax[level];
    const bool finite = maxbits < 0x7F800000u;
    int e = maxbits ? (int)(maxbits >> 23) - 126 : 0;                // 2^e > max |g|
    e = e > 100 ? 100 : (e < -100 ? -100 : e);                       // (keeps both scale factors inside fp32's range; gradients below 2^-100 round to zero)
    const float to_fixed_a = ldexpf(1.0f, 16 - e), to_fixed_b = 1048576.0f;   // two exact power-of-two factors (their product can exceed fp32's range)
    // the points of this workgroup: its eighth of the range's LIST (k_grid_bin_points), or of all points where there is none
    const uint32_t listed = bin_counts ? bin_counts[level * kBinMaxRanges + (range < kBinMaxRanges ? range : 0u)] : 0xFFFFFFFFu;
    const bool by_list = listed != 0xFFFFFFFFu;
    const uint32_t n_pts = by_list ? listed : B;
    const uint32_t *list = by_list ? bin_lists + (size_t)item_global * B : nullptr;
    const uint32_t per = (n_pts + kRgSlices - 1u) / kRgSlices, first = slice * per, last = first + per < n_pts ? first + per : n_pts;
    const float scale = lv.scale[level];
    const uint32_t res = lv.resolution[level];
    const TrIndex ix = tr_index<D>(size, res, gridtype, align_corners);          // resolved once per workgroup
    if (finite) {
        for (uint32_t i = first + threadIdx.x; i < last; i += kRgThreads) {
            const uint32_t b = by_list ? list[i] : i;
            float pos[D], deriv[D];
            uint32_t pg[D];
            float gc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) gc[c] = (float)grad[((size_t)level * B + b) * C + c] * to_fixed_a;   // (in flight together with the coordinates)
            if (!tr_locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, pos, deriv, pg)) continue;
            const uint32_t base = tr_base_row<D>(ix, pg);
#pragma unroll
            for (int idx = 0; idx < (1 << D); ++idx) {
                const uint32_t r = tr_corner_row<D>(ix, base, pg, idx, gridtype, align_corners, size, res) - row0;      // (wraps for rows below the range)
                if (r < nrows) {
                    float w = to_fixed_b;                           // the corner's weight: the same products in the same order as before (same bits)
#pragma unroll
                    for (int d = 0; d < D; ++d) w *= (idx & (1 << d)) ? pos[d] : 1.0f - pos[d];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[r * C + c]), (unsigned long long)__float2ll_rn(w * gc[c]));
                    }
                }
            }
        }
    }
    __syncthreads();
    float *dst = copies + (size_t)slice * total_floats + (size_t)(off + row0) * C;
    const float back_a = ldexpf(1.0f, e - 16), back_b = 1.0f / 1048576.0f;
    for (uint32_t i = threadIdx.x; i < n; i += kRgThreads) dst[i] = finite ? ((float)acc[i] * back_b) * back_a : __uint_as_float(0x7FC00000u);
}

// input gradient from dy_dx (gridencoder.cu:342-368)
template <typename G>
__global__ __launch_bounds__(kTrBlock) void k_grid_input_backward(const G *__restrict__ grad, const float *__restrict__ dy_dx, float *__restrict__ grad_inputs,
                                                                 uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    const uint32_t t = blockIdx.x * kTrBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; ++l)
        for (uint32_t c = 0; c < C; ++c) r = fmaf((float)grad[((size_t)l * B + b) * C + c], dd[(l * D + d) * C + c], r);
    grad_inputs[t] = r;
}

// The input gradient WITHOUT a materialised dy_dx (gridencoder.cu:198-243 + 342-368 in one pass): dL/dx[b, d] = sum over levels and channels of
// grad[l, b, c] * d feature[l, b, c] / d x[d].  The reference writes dy_dx [B, L, D, C] in the forward pass (116 MB for a May step's ambient grid) and
// reads it back here; the derivative along every axis is a pairing of the SAME 2^D corner values the forward pass interpolates, so one thread per point
// walks the levels, gathers the corners once per level and keeps the D sums in registers.
template <int D, int C, typename G>
__global__ __launch_bounds__(kTrBlock) void k_grid_input_grad(const G *__restrict__ grad, const float *__restrict__ inputs, const float *__restrict__ table,
                                                             const int32_t *__restrict__ offsets, float *__restrict__ grad_inputs, uint32_t B, uint32_t L, TrLevels lv,
                                                             uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * kTrBlock + threadIdx.x;
    if (b >= B) return;
    float sum[D];
#pragma unroll
    for (int d = 0; d < D; ++d) sum[d] = 0.0f;
    for (uint32_t level = 0; level < L; ++level) {
        float pos[D], deriv[D];
        uint32_t pg[D];
        const float scale = lv.scale[level];
        if (!tr_locate<D>(inputs + (size_t)b * D, scale, align_corners, interp, pos, deriv, pg)) break;   // out of range at one level = at every level
        const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
        const float *grid = table + (size_t)off * C;
        float v[1 << D][C];
        const TrIndex ix = tr_index<D>(size, res, gridtype, align_corners);      // level-uniform (scalar registers)
        const uint32_t base = tr_base_row<D>(ix, pg);
#pragma unroll
        for (int idx = 0; idx < (1 << D); ++idx) {
            const uint32_t row = tr_corner_row<D>(ix, base, pg, idx, gridtype, align_corners, size, res);
#pragma unroll
            for (int c = 0; c < C; ++c) v[idx][c] = grid[(size_t)row * C + c];
        }
        float gc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) gc[c] = (float)grad[((size_t)level * B + b) * C + c];
#pragma unroll
        for (int gd = 0; gd < D; ++gd) {
            float rg[C];
#pragma unroll
            for (int c = 0; c < C; ++c) rg[c] = 0.0f;
#pragma unroll
            for (int idx = 0; idx < (1 << D); ++idx) {
                if (idx & (1 << gd)) continue;                     // the pair (idx, idx | 1 << gd): left and right neighbour along gd
                float w = scale;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (d == gd) continue;
                    w *= (idx & (1 << d)) ? pos[d] : 1.0f - pos[d];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) rg[c] = fmaf(w * (v[idx | (1 << gd)][c] - v[idx][c]), deriv[gd], rg[c]);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) sum[gd] = fmaf(gc[c], rg[c], sum[gd]);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) grad_inputs[(size_t)b * D + d] = sum[d];
}

// total-variation gradient (gridencoder.cu:505-597)
template <int D, int C>
__global__ __launch_bounds__(kTrBlock) void k_grad_tv(const float *__restrict__ inputs, const float *__restrict__ table, float *__restrict__ grad,
                                                     const int32_t *__restrict__ offsets, float weight, uint32_t B, uint32_t L, TrLevels lv,
                                                     uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * kTrBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const float *in = inputs + (size_t)b * D;
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; ++d) inside = inside && !(in[d] < 0.0f || in[d] > 1.0f);
    if (!inside) return;
    const uint32_t off = (uint32_t)offsets[level], size = (uint32_t)offsets[level + 1] - off, res = lv.resolution[level];
    const float *grid = table + (size_t)off * C;
    float *gg = grad + (size_t)off * C;
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pg[d] = (uint32_t)floorf(fmaf(in[d], lv.scale[level], align_corners ? 0.0f : 0.5f));
    float results[C], idelta[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { results[c] = 0.0f; idelta[c] = 0.0f; }
    const uint32_t row = grid_row<D>(pg, gridtype, align_corners, size, res);
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const uint32_t cur = pg[d];
        if (cur < res) {
            pg[d] = cur + 1u;
            const uint32_t rr = grid_row<D>(pg, gridtype, align_corners, size, res);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float gv = grid[(size_t)row * C + c] - grid[(size_t)rr * C + c];
                results[c] += gv;
                idelta[c] = fmaf(gv, gv, idelta[c]);
            }
        }
        if (cur > 0) {
            pg[d] = cur - 1u;
            const uint32_t rl = grid_row<D>(pg, gridtype, align_corners, size, res);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float gv = grid[(size_t)row * C + c] - grid[(size_t)rl * C + c];
                results[c] += gv;
                idelta[c] = fmaf(gv, gv, idelta[c]);
            }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) atomicAdd(&gg[(size_t)row * C + c], w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f)));
}

static int tr_levels(TrLevels &lv, uint32_t L, float S, uint32_t H) {
    if (L == 0 || L > (uint32_t)kMaxLevels) return GFPP_EINVAL;
    GridLevels g;
    fill_level_scales(g, L, S, H);
    for (uint32_t l = 0; l < L; ++l) { lv.scale[l] = g.scale[l]; lv.resolution[l] = g.resolution[l]; }
    return 0;
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps,
                                   uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                                   float *deltas, int32_t *rays, int32_t *counter, const float *noises, gfpp_stream_t stream) {
    if (N == 0) return 0;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !rays || !counter || !noises) { set_error("gfpp_march_rays_train: null pointer"); return GFPP_EINVAL; }
    if (C < 1 || C > 8 || max_steps == 0) { set_error("gfpp_march_rays_train: cascade must be 1..8, max_steps > 0"); return GFPP_EINVAL; }
    const MarchParams mp = make_march_params(bound, dt_gamma, max_steps, C, H);
    hipLaunchKernelGGL(k_march_rays_train, dim3(div_up(N, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, rays_o, rays_d, grid, mp, max_steps, N, M, nears,
                       fars, xyzs, dirs, deltas, rays, counter, noises);
    return check_launch("gfpp_march_rays_train");
}

GFPP_API int gfpp_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const int32_t *rays, const float *deltas, uint32_t N, uint32_t M,
                                            float *grad_rays_o, float *grad_rays_d, gfpp_stream_t stream) {
    if (N == 0) return 0;
    if (!grad_xyzs || !grad_dirs || !rays || !deltas || !grad_rays_o || !grad_rays_d) { set_error("gfpp_march_rays_train_backward: null pointer"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_march_rays_train_backward, dim3(div_up(N, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, grad_xyzs, grad_dirs, rays, deltas, N, M,
                       grad_rays_o, grad_rays_d);
    return check_launch("gfpp_march_rays_train_backward");
}

GFPP_API int gfpp_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ambient, const float *deltas, const int32_t *rays, uint32_t M,
                                               uint32_t N, float T_thresh, float *weights_sum, float *ambient_sum, float *depth, float *image,
                                               gfpp_stream_t stream) {
    if (N == 0) return 0;
    if (!sigmas || !rgbs || !ambient || !deltas || !rays || !weights_sum || !ambient_sum || !depth || !image) { set_error("gfpp_composite_rays_train_forward: null pointer"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(div_up(N, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh,
                       weights_sum, ambient_sum, depth, image);
    return check_launch("gfpp_composite_rays_train_forward");
}

GFPP_API int gfpp_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_ambient_sum, const float *grad_image, const float *sigmas,
                                                const float *rgbs, const float *ambient, const float *deltas, const int32_t *rays, const float *weights_sum,
                                                const float *ambient_sum, const float *image, uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                                float *grad_rgbs, float *grad_ambient, gfpp_stream_t stream) {
    (void)ambient; (void)ambient_sum;
    if (N == 0) return 0;
    if (!grad_weights_sum || !grad_ambient_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs || !grad_ambient) {
        set_error("gfpp_composite_rays_train_backward: null pointer");
        return GFPP_EINVAL;
    }
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(div_up(N, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, grad_weights_sum, grad_ambient_sum, grad_image, sigmas,
                       rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient);
    return check_launch("gfpp_composite_rays_train_backward");
}

GFPP_API int gfpp_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation, gfpp_stream_t stream) {
    if (!grid || !grid_dilation || C == 0 || H == 0 || H > 1024) { set_error("gfpp_morton3D_dilation: bad arguments"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_morton_dilation, dim3(div_up(C * H * H * H, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, grid, C, H, grid_dilation);
    return check_launch("gfpp_morton3D_dilation");
}

GFPP_API int gfpp_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, gfpp_stream_t stream) {
    if (N == 0) return 0;
    if (!rays_o || !rays_d || !coords) { set_error("gfpp_sph_from_ray: null pointer"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_sph_from_ray, dim3(div_up(N, kTrBlock)), dim3(kTrBlock), 0, (hipStream_t)stream, rays_o, rays_d, radius, N, coords);
    return check_launch("gfpp_sph_from_ray");
}

#define GFPP_DISPATCH_DC(KERNEL, ...)                                                                                                  \
    do {                                                                                                                                \
        const dim3 grid_(div_up(B, kTrBlock), L), block_(kTrBlock);                                                                     \
        if (D == 2 && C == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid_, block_, 0, st, __VA_ARGS__);                                    \
        else if (D == 3 && C == 2) hipLaunchKernelGGL((KERNEL<3, 2>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 2 && C == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 3 && C == 1) hipLaunchKernelGGL((KERNEL<3, 1>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 2 && C == 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 3 && C == 4) hipLaunchKernelGGL((KERNEL<3, 4>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 2 && C == 8) hipLaunchKernelGGL((KERNEL<2, 8>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else if (D == 3 && C == 8) hipLaunchKernelGGL((KERNEL<3, 8>), grid_, block_, 0, st, __VA_ARGS__);                               \
        else { set_error("grid encoder (training): input_dim must be 2 or 3 and level_dim 1, 2, 4 or 8 (got %u, %u)", D, C); return GFPP_EUNSUPPORTED; } \
    } while (0)

GFPP_API int gfpp_grid_encode_dydx(const float *inputs, const float *embeddings, const int32_t *offsets, float *dy_dx, uint32_t B, uint32_t D, uint32_t C,
                                   uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream) {
    if (B == 0) return 0;
    if (!inputs || !embeddings || !offsets || !dy_dx || gridtype > 1 || interp > 1) { set_error("gfpp_grid_encode_dydx: bad arguments"); return GFPP_EINVAL; }
    TrLevels lv;
    if (tr_levels(lv, L, S, H)) { set_error("gfpp_grid_encode_dydx: 1 <= L <= 32"); return GFPP_EINVAL; }
    const hipStream_t st = (hipStream_t)stream;
    GFPP_DISPATCH_DC(k_grid_dydx, inputs, embeddings, offsets, dy_dx, B, L, lv, gridtype, align_corners != 0, interp);
    return check_launch("gfpp_grid_encode_dydx");
}

// One (D, C) instantiation of the three table-gradient kernels for grad type G
template <int D, int C, typename G>
static int grid_backward_launch(const char *who, const G *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, float *xcd_copies,
                                uint32_t total_floats, uint32_t B, uint32_t L, const TrLevels &lv, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t st,
                                void *bins, unsigned long long bins_bytes) {
    // levels whose table fits kLdsGradFloats go through the LDS-privatised kernel (128 KiB of dynamic LDS); a device that cannot reserve that much
    // (64 KiB parts) scatters every level directly instead (lds_floats = 0) -- slower, same result
    const int lds_bytes = (int)(kLdsGradFloats * sizeof(float));
    // per call, not once per process: the attribute is per device and a process may train on several (it costs a table write)
    bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_backward_lds<D, C, G>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess;
    bool ranges = lds_ok && xcd_copies != nullptr;
    if (ranges && tuning().grid_bwd_scatter) ranges = false;       // A/B: device atomics for the levels beyond the LDS (the round-2 path)
    if (ranges) ranges = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_grid_backward_ranges<D, C, G>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess;
    if (!lds_ok) (void)hipGetLastError();
    int rc = 0;
    if (lds_ok && !ranges) {
        hipLaunchKernelGGL((k_grid_backward_lds<D, C, G>), dim3(div_up(B, kLdsGradPoints), L), dim3(kTrBlock), (size_t)lds_bytes, st, grad, inputs, offsets,
                           grad_embeddings, B, L, lv, gridtype, ac, interp, xcd_copies, total_floats);
        rc = check_launch(who);
        if (rc) return rc;
    }
    // (the range passes store every value of every copy; only the scatter path adds into them)
    if (!ranges && xcd_copies && hipMemsetAsync(xcd_copies, 0, (size_t)kXcds * total_floats * sizeof(float), st) != hipSuccess) { set_error("%s: cannot clear the gradient copies", who); return GFPP_EINVAL; }
    if (ranges) {
        // per-level max |grad| -> the levels' fixed-point scales (kept behind the eight copies: the scratch has 64 spare words)
        uint32_t *levelmax = reinterpret_cast<uint32_t *>(xcd_copies + (size_t)kXcds * total_floats);
        if (hipMemsetAsync(levelmax, 0, 64 * sizeof(uint32_t), st) != hipSuccess) { set_error("%s: cannot clear the level maxima", who); return GFPP_EINVAL; }
        hipLaunchKernelGGL((k_grid_grad_levelmax<G>), dim3(32, L), dim3(kTrBlock), 0, st, grad, B * (uint32_t)C, levelmax);
        rc = check_launch(who);
        if (rc) return rc;
        // sum over levels of ceil(size C / V) <= total / V + L: workgroups beyond the actual ranges return at once
        const uint32_t items = total_floats / kRgValues + L;
        // the points of every range as lists (k_grid_bin_points), when the caller brought the scratch for them: [L][32] counters, then `items` lists of B indices
        uint32_t *bin_counts = nullptr, *bin_lists = nullptr;
        if (bins && tuning().grid_bwd_bins && L <= (uint32_t)kMaxLevels && bins_bytes >= gfpp_grid_backward_bins_bytes(total_floats / (uint32_t)C, (uint32_t)C, L, B)) {
            bin_counts = static_cast<uint32_t *>(bins);
            bin_lists = bin_counts + (size_t)kMaxLevels * kBinMaxRanges;
            if (hipMemsetAsync(bin_counts, 0, (size_t)kMaxLevels * kBinMaxRanges * sizeof(uint32_t), st) != hipSuccess) { set_error("%s: cannot clear the list counters", who); return GFPP_EINVAL; }
            hipLaunchKernelGGL((k_grid_bin_points<D, C>), dim3(div_up(B, 1024u), L), dim3(1024), 0, st, inputs, offsets, B, lv, gridtype, ac, bin_counts, bin_lists);
            rc = check_launch(who);
            if (rc) return rc;
        }
        hipLaunchKernelGGL((k_grid_backward_ranges<D, C, G>), dim3(items * kRgSlices), dim3(kRgThreads), (size_t)lds_bytes, st, grad, inputs, offsets, B, L, lv,
                           gridtype, ac, interp, xcd_copies, total_floats, levelmax, bin_counts, bin_lists);
    } else {
        hipLaunchKernelGGL((k_grid_backward<D, C, G>), dim3(div_up(B, kTrBlock), L), dim3(kTrBlock), 0, st, grad, inputs, offsets, grad_embeddings, B, L, lv,
                           gridtype, ac, interp, lds_ok ? kLdsGradFloats : 0u, xcd_copies, total_floats);
    }
    return check_launch(who);
}

static int grid_backward_impl(const char *who, const void *grad, int grad_dtype, const float *inputs, const int32_t *offsets, float *grad_embeddings,
                              uint32_t rows_total, float *xcd_copies, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                              float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream, void *bins = nullptr,
                              unsigned long long bins_bytes = 0) {
    if (B == 0) return 0;
    if (!grad || !inputs || !offsets || !grad_embeddings || gridtype > 1 || interp > 1 || ((dy_dx == nullptr) != (grad_inputs == nullptr))) {
        set_error("%s: bad arguments (dy_dx and grad_inputs go together)", who);
        return GFPP_EINVAL;
    }
    TrLevels lv;
    if (tr_levels(lv, L, S, H)) { set_error("%s: 1 <= L <= 32", who); return GFPP_EINVAL; }
    const hipStream_t st = (hipStream_t)stream;
    const uint32_t total_floats = rows_total * C;
    const bool ac = align_corners != 0;
    int rc;
    if (grad_dtype == GFPP_F16) {
        const _Float16 *g = static_cast<const _Float16 *>(grad);
        if (C != 2 || (D != 2 && D != 3)) { set_error("%s: half gradients are built for level_dim 2, input_dim 2 or 3 (got %u, %u)", who, C, D); return GFPP_EUNSUPPORTED; }
        rc = D == 2 ? grid_backward_launch<2, 2, _Float16>(who, g, inputs, offsets, grad_embeddings, xcd_copies, total_floats, B, L, lv, gridtype, ac, interp, st, bins, bins_bytes)
                    : grid_backward_launch<3, 2, _Float16>(who, g, inputs, offsets, grad_embeddings, xcd_copies, total_floats, B, L, lv, gridtype, ac, interp, st, bins, bins_bytes);
    } else {
        const float *g = static_cast<const float *>(grad);
#define GFPP_BWD_ONE(DD, CC) rc = grid_backward_launch<DD, CC, float>(who, g, inputs, offsets, grad_embeddings, xcd_copies, total_floats, B, L, lv, gridtype, ac, interp, st, bins, bins_bytes)
        if (D == 2 && C == 2) GFPP_BWD_ONE(2, 2);
        else if (D == 3 && C == 2) GFPP_BWD_ONE(3, 2);
        else if (D == 2 && C == 1) GFPP_BWD_ONE(2, 1);
        else if (D == 3 && C == 1) GFPP_BWD_ONE(3, 1);
        else if (D == 2 && C == 4) GFPP_BWD_ONE(2, 4);
        else if (D == 3 && C == 4) GFPP_BWD_ONE(3, 4);
        else if (D == 2 && C == 8) GFPP_BWD_ONE(2, 8);
        else if (D == 3 && C == 8) GFPP_BWD_ONE(3, 8);
        else { set_error("grid encoder (training): input_dim must be 2 or 3 and level_dim 1, 2, 4 or 8 (got %u, %u)", D, C); return GFPP_EUNSUPPORTED; }
#undef GFPP_BWD_ONE
    }
    if (rc) return rc;
    if (xcd_copies) {
        hipLaunchKernelGGL(k_grid_reduce_xcd, dim3(div_up(total_floats, kTrBlock)), dim3(kTrBlock), 0, st, xcd_copies, grad_embeddings, total_floats);
        rc = check_launch(who);
        if (rc) return rc;
    }
    if (!dy_dx) return 0;
    if (grad_dtype == GFPP_F16) hipLaunchKernelGGL(k_grid_input_backward<_Float16>, dim3(div_up(B * D, kTrBlock)), dim3(kTrBlock), 0, st, static_cast<const _Float16 *>(grad), dy_dx, grad_inputs, B, D, C, L);
    else hipLaunchKernelGGL(k_grid_input_backward<float>, dim3(div_up(B * D, kTrBlock)), dim3(kTrBlock), 0, st, static_cast<const float *>(grad), dy_dx, grad_inputs, B, D, C, L);
    return check_launch(who);
}

GFPP_API unsigned long long gfpp_grid_backward_bins_bytes(uint32_t rows_total, uint32_t C, uint32_t L, uint32_t B) {
    const unsigned long long items = (unsigned long long)rows_total * C / kRgValues + L;
    return ((unsigned long long)kMaxLevels * kBinMaxRanges + items * B) * sizeof(uint32_t);
}

GFPP_API int gfpp_grid_encode_backward_f16(const void *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t rows_total,
                                           float *xcd_copies, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                           float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream, void *bins,
                                           unsigned long long bins_bytes) {
    if (!xcd_copies || rows_total == 0) { set_error("gfpp_grid_encode_backward_f16: needs the [8, rows_total * C] fp32 scratch"); return GFPP_EINVAL; }
    return grid_backward_impl("gfpp_grid_encode_backward_f16", grad, GFPP_F16, inputs, offsets, grad_embeddings, rows_total, xcd_copies, B, D, C, L, S, H, dy_dx,
                              grad_inputs, gridtype, align_corners, interp, stream, bins, bins_bytes);
}

GFPP_API int gfpp_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets, float *grad_embeddings, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                                       int align_corners, uint32_t interp, gfpp_stream_t stream) {
    (void)embeddings;
    return grid_backward_impl("gfpp_grid_encode_backward", grad, GFPP_F32, inputs, offsets, grad_embeddings, 0, nullptr, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype,
                              align_corners, interp, stream);
}

GFPP_API int gfpp_grid_encode_backward_xcd(const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t rows_total,
                                           float *xcd_copies, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                           float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream, void *bins,
                                           unsigned long long bins_bytes) {
    if (!xcd_copies || rows_total == 0) { set_error("gfpp_grid_encode_backward_xcd: needs the [8, rows_total * C] scratch"); return GFPP_EINVAL; }
    return grid_backward_impl("gfpp_grid_encode_backward_xcd", grad, GFPP_F32, inputs, offsets, grad_embeddings, rows_total, xcd_copies, B, D, C, L, S, H, dy_dx, grad_inputs,
                              gridtype, align_corners, interp, stream, bins, bins_bytes);
}

GFPP_API int gfpp_grid_encode_input_backward(const void *grad, int grad_dtype, const float *inputs, const float *embeddings, const int32_t *offsets,
                                             float *grad_inputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                             int align_corners, uint32_t interp, gfpp_stream_t stream) {
    const char *who = "gfpp_grid_encode_input_backward";
    if (B == 0) return 0;
    if (!grad || !inputs || !embeddings || !offsets || !grad_inputs || gridtype > 1 || interp > 1) { set_error("%s: bad arguments", who); return GFPP_EINVAL; }
    if (grad_dtype != GFPP_F32 && grad_dtype != GFPP_F16) { set_error("%s: grad must be fp32 or half", who); return GFPP_EUNSUPPORTED; }
    if (C != 2 || (D != 2 && D != 3)) { set_error("%s: built for level_dim 2, input_dim 2 or 3 (got %u, %u): use dy_dx + gfpp_grid_encode_backward", who, C, D); return GFPP_EUNSUPPORTED; }
    TrLevels lv;
    if (tr_levels(lv, L, S, H)) { set_error("%s: 1 <= L <= 32", who); return GFPP_EINVAL; }
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid(div_up(B, kTrBlock)), block(kTrBlock);
    const bool ac = align_corners != 0;
    if (grad_dtype == GFPP_F16) {
        const _Float16 *g = static_cast<const _Float16 *>(grad);
        if (D == 2) hipLaunchKernelGGL((k_grid_input_grad<2, 2, _Float16>), grid, block, 0, st, g, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, interp);
        else hipLaunchKernelGGL((k_grid_input_grad<3, 2, _Float16>), grid, block, 0, st, g, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, interp);
    } else {
        const float *g = static_cast<const float *>(grad);
        if (D == 2) hipLaunchKernelGGL((k_grid_input_grad<2, 2, float>), grid, block, 0, st, g, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, interp);
        else hipLaunchKernelGGL((k_grid_input_grad<3, 2, float>), grid, block, 0, st, g, inputs, embeddings, offsets, grad_inputs, B, L, lv, gridtype, ac, interp);
    }
    return check_launch(who);
}

GFPP_API int gfpp_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                                       uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, gfpp_stream_t stream) {
    if (B == 0) return 0;
    if (!inputs || !embeddings || !grad || !offsets || gridtype > 1) { set_error("gfpp_grad_total_variation: bad arguments"); return GFPP_EINVAL; }
    TrLevels lv;
    if (tr_levels(lv, L, S, H)) { set_error("gfpp_grad_total_variation: 1 <= L <= 32"); return GFPP_EINVAL; }
    const hipStream_t st = (hipStream_t)stream;
    GFPP_DISPATCH_DC(k_grad_tv, inputs, embeddings, grad, offsets, weight, B, L, lv, gridtype, align_corners != 0);
    return check_launch("gfpp_grad_total_variation");
}
