/*
 * radnerf_oracle.c -- CPU restatement of the GeneFace++ radnerfs native kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: pinned at kernel level against the reference's own native code.  oracle/build_ref.py compiles the
 * reference's raymarching.cu / gridencoder.cu / shencoder.cu / freqencoder.cu UNMODIFIED (from /root/reference, for gfx950,
 * outputs in oracle/_ref/); tests/golden/make_golden_ref_kernels.py ran them on an MI355X over the seeded case table of
 * tests/ref_kernel_cases.py and committed their outputs (tests/golden/ref_kernel_golden.npz).  This file reproduces them
 * (tests/test_oracle_ref_kernels_cpu.py, CPU; tests/test_ref_kernels_gpu.py, live three-way with the product):
 * bit for bit for Morton codes, bitfields, the slab test, every marcher variant, the packed training march, fp32 grid features
 * and dy_dx, march / frequency-encoder backward; ulp-scale tolerances for __expf / __sinf / atomics / half accumulation.
 * Caveat: "the reference's code" here means its source under clang's FMA contraction for gfx950, not nvcc's for sm_xx.
 * Additionally pinned by (1) the self-derived invariants of SURVEY.md section 8c (tests/test_oracle_invariants.py) and
 * (2) the reference's own *Python* layer run on top of this file (tests/golden/make_golden.py).
 *
 * Floating-point policy (the reference is compiled by nvcc with default -fmad=true): every
 * a*b+c pattern in the reference source is written here as an explicit fmaf(); this file must be
 * compiled with -ffp-contract=off so that nothing else is fused.  For the shipped geometry
 * (cascade 1, H = 128, base resolution 16: all powers of two) the only place where fused vs
 * unfused arithmetic can differ in the marcher is the sample position o + t*d.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * helpers -- modules/radnerfs/raymarching/src/raymarching.cu:19-81
 * ---------------------------------------------------------------------------------------- */
static const float ORC_SQRT3 = 1.7320508075688772f; /* raymarching.cu:19 */

static inline float orc_clampf(float x, float lo, float hi) { /* raymarching.cu:34-36 */
    return fminf(hi, fmaxf(lo, x));
}
static inline float orc_signf(float x) { return copysignf(1.0f, x); } /* raymarching.cu:30-32 */

static inline int orc_mip_from_pos(float x, float y, float z, float max_cascade) { /* :42-47 */
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)exponent));
}
static inline int orc_mip_from_dt(float dt, float H, float max_cascade) { /* :49-54 */
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)exponent));
}
static inline uint32_t orc_expand_bits(uint32_t v) { /* :56-63 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
ORC_API uint32_t orc_morton3D(uint32_t x, uint32_t y, uint32_t z) { /* :65-71 */
    return orc_expand_bits(x) | (orc_expand_bits(y) << 1) | (orc_expand_bits(z) << 2);
}
ORC_API uint32_t orc_morton3D_invert(uint32_t x) { /* :73-81 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:214-241 (kernel_morton3D / kernel_morton3D_invert): coords [N,3] i32 <-> indices [N] */
ORC_API void orc_morton3D_batch(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)orc_morton3D((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}
ORC_API void orc_morton3D_invert_batch(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t i = (uint32_t)indices[n];
        coords[3 * n] = (int32_t)orc_morton3D_invert(i);
        coords[3 * n + 1] = (int32_t)orc_morton3D_invert(i >> 1);
        coords[3 * n + 2] = (int32_t)orc_morton3D_invert(i >> 2);
    }
}

/* raymarching.cu:267-289 kernel_packbits: grid [N*8] float -> bitfield [N] u8, bit i <-> grid[8n+i] > thresh */
ORC_API void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++)
            bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? ((uint8_t)1 << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------------------------------------
 * near_far_from_aabb -- raymarching.cu:91-145
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                    uint32_t N, float min_near, float *nears, float *fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float t;

        float near = (aabb[0] - ox) * rdx;
        float far = (aabb[3] - ox) * rdx;
        if (near > far) { t = near; near = far; far = t; }

        float near_y = (aabb[1] - oy) * rdy;
        float far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }

        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;

        float near_z = (aabb[2] - oz) * rdz;
        float far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }

        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;

        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* ------------------------------------------------------------------------------------------
 * march_rays (inference) -- raymarching.cu:827-929.  Outputs must be zero-initialised by the
 * caller (raymarching.py:384-386); slot layout n*n_step + s.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive,
                            const float *rays_t, const float *rays_o_, const float *rays_d_,
                            float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                            const uint8_t *grid, const float *nears, const float *fars,
                            float *xyzs_, float *dirs_, float *deltas_, const float *noises) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        const float *rays_o = rays_o_ + (size_t)index * 3;
        const float *rays_d = rays_d_ + (size_t)index * 3;
        float *xyzs = xyzs_ + (size_t)n * n_step * 3;
        float *dirs = dirs_ + (size_t)n * n_step * 3;
        float *deltas = deltas_ + (size_t)n * n_step * 2;

        const float ox = rays_o[0], oy = rays_o[1], oz = rays_o[2];
        const float dx = rays_d[0], dy = rays_d[1], dz = rays_d[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float rH = 1 / (float)H;
        const float H3 = (float)(H * H * H);

        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;

        const float dt_max = 2 * ORC_SQRT3 * (float)(1 << (C - 1)) / (float)H;     /* :866 */
        const float dt_min = fminf(dt_max, 2 * ORC_SQRT3 / (float)max_steps);       /* :867 */

        uint32_t step = 0;
        t = fmaf(orc_clampf(t * dt_gamma, dt_min, dt_max), noise, t);               /* :873 */

        while (t < far && step < n_step) {
            const float x = orc_clampf(fmaf(t, dx, ox), -bound, bound);             /* :877-879 */
            const float y = orc_clampf(fmaf(t, dy, oy), -bound, bound);
            const float z = orc_clampf(fmaf(t, dz, oz), -bound, bound);

            const float dt = orc_clampf(t * dt_gamma, dt_min, dt_max);

            const int lvl_p = orc_mip_from_pos(x, y, z, (float)C);
            const int lvl_d = orc_mip_from_dt(dt, (float)H, (float)C);
            const int level = lvl_p > lvl_d ? lvl_p : lvl_d;

            const float mip_bound = fminf(scalbnf(1.0f, level), bound);
            const float mip_rbound = 1 / mip_bound;

            /* :890-892 -- the literal 0.5 is a double: fp64 product, rounded to fp32 by clamp(),
             * then truncated to int. */
            const int nx = (int)orc_clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
            const int ny = (int)orc_clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
            const int nz = (int)orc_clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));

            /* :894 -- index arithmetic happens in float (H3 is const float) */
            const uint32_t gidx = (uint32_t)fmaf((float)level, H3, (float)orc_morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            const int occ = grid[gidx / 8] & (1 << (gidx % 8));

            if (occ) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                t += dt;
                deltas[0] = dt;
                deltas[1] = t;
                xyzs += 3; dirs += 3; deltas += 2;
                step++;
            } else {
                /* :919-921 */
                const float tx = fmaf(fmaf(((float)nx + 0.5f + 0.5f * orc_signf(dx)) * rH, 2.0f, -1.0f), mip_bound, -x) * rdx;
                const float ty = fmaf(fmaf(((float)ny + 0.5f + 0.5f * orc_signf(dy)) * rH, 2.0f, -1.0f), mip_bound, -y) * rdy;
                const float tz = fmaf(fmaf(((float)nz + 0.5f + 0.5f * orc_signf(dz)) * rH, 2.0f, -1.0f), mip_bound, -z) * rdz;
                const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
                do {
                    t += orc_clampf(t * dt_gamma, dt_min, dt_max);
                } while (t < tt);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * composite_rays (inference) -- raymarching.cu:942-1029.  The reference uses the __expf fast
 * intrinsic (:984); libm expf here => tolerance, not bit equality, on the composited values.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                float *rays_t_, const float *sigmas_, const float *rgbs_,
                                const float *deltas_, float *weights_sum_, float *depth_, float *image_) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float *sigmas = sigmas_ + (size_t)n * n_step;
        const float *rgbs = rgbs_ + (size_t)n * n_step * 3;
        const float *deltas = deltas_ + (size_t)n * n_step * 2;

        float t = rays_t_[index];
        float weight_sum = weights_sum_[index];
        float d = depth_[index];
        float r = image_[(size_t)index * 3], g = image_[(size_t)index * 3 + 1], b = image_[(size_t)index * 3 + 2];

        uint32_t step = 0;
        while (step < n_step) {
            if (deltas[0] == 0) break;
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t = deltas[1];
            d = fmaf(weight, t, d);
            r = fmaf(weight, rgbs[0], r);
            g = fmaf(weight, rgbs[1], g);
            b = fmaf(weight, rgbs[2], b);
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; deltas += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t_[index] = t;

        weights_sum_[index] = weight_sum;
        depth_[index] = d;
        image_[(size_t)index * 3] = r; image_[(size_t)index * 3 + 1] = g; image_[(size_t)index * 3 + 2] = b;
    }
}

/* ------------------------------------------------------------------------------------------
 * grid encoder forward -- modules/radnerfs/encoders/gridencoder/src/gridencoder.cu:50-196
 * inputs [B,D] in [0,1]; embeddings [sum,C] f32; offsets [L+1]; outputs [L,B,C] (level-major)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t orc_fast_hash(const uint32_t *pos_grid, uint32_t D) { /* gridencoder.cu:50-63 */
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}
static inline uint32_t orc_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                                      uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid) { /* :66-84 */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = orc_fast_hash(pos_grid, D);
    return (index % hashmap_size) * C + ch;
}
/* level scale / resolution as the device computes them (gridencoder.cu:138-139) */
ORC_API void orc_grid_level_params(uint32_t level, float S, uint32_t H, float *scale, uint32_t *resolution) {
    const float sc = fmaf(exp2f((float)level * S), (float)H, -1.0f);
    *scale = sc;
    *resolution = (uint32_t)ceil((double)sc) + 1;
}

ORC_API int orc_grid_encode_forward(const float *inputs_, const float *embeddings, const int32_t *offsets,
                                    float *outputs_, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                    uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D < 1 || D > 7 || C < 1 || C > 8) return -1;
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t level = 0; level < (int64_t)L; level++) {
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
            const float *inputs = inputs_ + (size_t)b * D;
            float *outputs = outputs_ + (size_t)level * B * C + (size_t)b * C;

            int flag_oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (inputs[d] < 0 || inputs[d] > 1) flag_oob = 1;
            if (flag_oob) {
                for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = 0;
                continue;
            }

            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            float scale; uint32_t resolution;
            orc_grid_level_params((uint32_t)level, S, H, &scale, &resolution);

            float pos[7]; uint32_t pos_grid[7];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = pos[d] * pos[d] * fmaf(-2.0f, pos[d], 3.0f); /* smoothstep :40-42 */
            }

            float results[8] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pos_grid_local[7];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pos_grid_local[d] = pos_grid[d]; }
                    else { w *= pos[d]; pos_grid_local[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid_local);
                for (uint32_t ch = 0; ch < C; ch++) results[ch] = fmaf(w, grid[index + ch], results[ch]);
            }
            for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = results[ch];
        }
    }
    return 0;
}

/* exposes the raw table row index of one corner (for the index-level known-answer tests) */
ORC_API uint32_t orc_grid_corner_row(uint32_t D, uint32_t gridtype, int align_corners, uint32_t hashmap_size,
                                     uint32_t resolution, const uint32_t *pos_grid) {
    return orc_grid_index(D, 1, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
}

/* ------------------------------------------------------------------------------------------
 * spherical harmonics, degree <= 4 -- modules/radnerfs/encoders/shencoder/src/shencoder.cu:28-68
 * inputs [B,3], outputs [B,degree^2]
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_sh_encode_forward(const float *inputs, float *outputs_, uint32_t B, uint32_t degree) {
    if (degree < 1 || degree > 4) return -1;
    const uint32_t C2 = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        float *o = outputs_ + (size_t)b * C2;
        const float x = inputs[3 * b], y = inputs[3 * b + 1], z = inputs[3 * b + 2];
        const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        o[0] = 0.28209479177387814f;
        if (degree <= 1) continue;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        if (degree <= 2) continue;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = fmaf(0.94617469575755997f, z2, -0.31539156525251999f);
        o[7] = -1.0925484305920792f * xz;
        o[8] = fmaf(0.54627421529603959f, x2, -(0.54627421529603959f * y2));
        if (degree <= 3) continue;
        o[9] = 0.59004358992664352f * y * fmaf(-3.0f, x2, y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * fmaf(-5.0f, z2, 1.0f);
        o[12] = 0.3731763325901154f * z * fmaf(5.0f, z2, -3.0f);
        o[13] = 0.45704579946446572f * x * fmaf(-5.0f, z2, 1.0f);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * fmaf(3.0f, y2, -x2);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * frequency encoder -- modules/radnerfs/encoders/freqencoder/src/freqencoder.cu:30-58
 * outputs [B,C], C = D + 2*D*deg; layout [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...];
 * cos is sin(x + pi/2) (:55-56).  Reference uses __sinf (fast math) => tolerance.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs) {
    (void)deg;
    const float PI = 3.14159265358979323846f;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; t++) {
        const uint32_t b = (uint32_t)(t / C);
        const uint32_t c = (uint32_t)(t - (int64_t)b * C);
        const float *in = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = in[c];
        } else {
            const uint32_t col = c / D - 1;
            const uint32_t d = c % D;
            const uint32_t freq = col / 2;
            const float phase_shift = (float)(col % 2) * (PI / 2);
            outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
    }
}

/* ==========================================================================================
 * TRAINING-SIDE KERNELS
 * ========================================================================================== */

/* one step of the marcher shared by the two passes of kernel_march_rays_train (raymarching.cu:400-441 and :461-515, which repeat
 * the code of kernel_march_rays).  Returns 1 if the cell at t is occupied (then *x,y,z,dt are the sample), else advances *t past
 * the empty voxel and returns 0. */
static inline int orc_train_probe(float ox, float oy, float oz, float dx, float dy, float dz, float rdx, float rdy, float rdz,
                                  float bound, float dt_gamma, float dt_min, float dt_max, uint32_t C, uint32_t H, const uint8_t *grid,
                                  float *t, float *x_, float *y_, float *z_, float *dt_) {
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const float x = orc_clampf(fmaf(*t, dx, ox), -bound, bound);
    const float y = orc_clampf(fmaf(*t, dy, oy), -bound, bound);
    const float z = orc_clampf(fmaf(*t, dz, oz), -bound, bound);
    const float dt = orc_clampf(*t * dt_gamma, dt_min, dt_max);
    const int lvl_p = orc_mip_from_pos(x, y, z, (float)C);
    const int lvl_d = orc_mip_from_dt(dt, (float)H, (float)C);
    const int level = lvl_p > lvl_d ? lvl_p : lvl_d;
    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = (int)orc_clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
    const int ny = (int)orc_clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
    const int nz = (int)orc_clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, (float)(H - 1));
    const uint32_t gidx = (uint32_t)fmaf((float)level, H3, (float)orc_morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    if (grid[gidx / 8] & (1 << (gidx % 8))) {
        *x_ = x; *y_ = y; *z_ = z; *dt_ = dt;
        return 1;
    }
    const float tx = fmaf(fmaf(((float)nx + 0.5f + 0.5f * orc_signf(dx)) * rH, 2.0f, -1.0f), mip_bound, -x) * rdx;
    const float ty = fmaf(fmaf(((float)ny + 0.5f + 0.5f * orc_signf(dy)) * rH, 2.0f, -1.0f), mip_bound, -y) * rdy;
    const float tz = fmaf(fmaf(((float)nz + 0.5f + 0.5f * orc_signf(dz)) * rH, 2.0f, -1.0f), mip_bound, -z) * rdz;
    const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        *t += orc_clampf(*t * dt_gamma, dt_min, dt_max);
    } while (*t < tt);
    return 0;
}

/* kernel_march_rays_train -- raymarching.cu:352-518.  rays [N,3] i32 = (ray id, first point, point count); counter[0] = points,
 * counter[1] = rays.  The reference hands out point ranges with atomicAdd in whatever order the hardware schedules the rays; this
 * restatement visits the rays in index order, i.e. ONE of the valid outcomes (compare per ray, not per slot). */
ORC_API void orc_march_rays_train(const float *rays_o_, const float *rays_d_, const uint8_t *grid, float bound, float dt_gamma,
                                  uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                  const float *fars, float *xyzs_, float *dirs_, float *deltas_, int32_t *rays, int32_t *counter,
                                  const float *noises) {
    const float dt_max = 2 * ORC_SQRT3 * (float)(1 << (C - 1)) / (float)H;
    const float dt_min = fminf(dt_max, 2 * ORC_SQRT3 / (float)max_steps);
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o_[3 * n], oy = rays_o_[3 * n + 1], oz = rays_o_[3 * n + 2];
        const float dx = rays_d_[3 * n], dy = rays_d_[3 * n + 1], dz = rays_d_[3 * n + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float far = fars[n];
        float t0 = nears[n];
        t0 = fmaf(orc_clampf(t0 * dt_gamma, dt_min, dt_max), noises[n], t0);                       /* :392 */
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {                                                   /* first pass :400-441 */
            if (orc_train_probe(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, bound, dt_gamma, dt_min, dt_max, C, H, grid, &t, &x, &y, &z, &dt)) {
                num_steps++;
                t += dt;
            }
        }
        const uint32_t point_index = (uint32_t)counter[0];                                           /* :446-447 (atomicAdd) */
        counter[0] += (int32_t)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1];
        counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;
        float *xyzs = xyzs_ + (size_t)point_index * 3, *dirs = dirs_ + (size_t)point_index * 3, *deltas = deltas_ + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        while (t < far && step < num_steps) {                                                        /* second pass :461-515 */
            if (orc_train_probe(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, bound, dt_gamma, dt_min, dt_max, C, H, grid, &t, &x, &y, &z, &dt)) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                t += dt;
                deltas[0] = dt;
                deltas[1] = t;
                xyzs += 3; dirs += 3; deltas += 2;
                step++;
            }
        }
    }
}

/* kernel_march_rays_train_backward -- raymarching.cu:535-583.  grad_rays_o/d are ACCUMULATED into (the caller zero-fills). */
ORC_API void orc_march_rays_train_backward(const float *grad_xyzs_, const float *grad_dirs_, const int32_t *rays, const float *deltas_,
                                           uint32_t N, uint32_t M, float *grad_rays_o_, float *grad_rays_d_) {
    for (uint32_t n = 0; n < N; n++) {
        float *go = grad_rays_o_ + 3 * (size_t)n, *gd = grad_rays_d_ + 3 * (size_t)n;      /* indexed by n, NOT by rays[n*3] (as the reference) */
        const uint32_t offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float *gx = grad_xyzs_ + (size_t)offset * 3, *gdir = grad_dirs_ + (size_t)offset * 3, *deltas = deltas_ + (size_t)offset * 2;
        for (uint32_t step = 0; step < num_steps; step++) {
            for (int k = 0; k < 3; k++) {
                go[k] += gx[k];
                gd[k] += fmaf(gx[k], deltas[1], gdir[k]);
            }
            gx += 3; gdir += 3; deltas += 2;
        }
    }
}

/* kernel_composite_rays_train_forward -- raymarching.cu:603-688 (T *= 1 - alpha, post-update threshold test: NOT the inference rule) */
ORC_API void orc_composite_rays_train_forward(const float *sigmas_, const float *rgbs_, const float *ambient_, const float *deltas_,
                                              const int32_t *rays, uint32_t M, uint32_t N, float T_thresh, float *weights_sum,
                                              float *ambient_sum, float *depth, float *image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; ambient_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *sigmas = sigmas_ + offset, *rgbs = rgbs_ + (size_t)offset * 3, *ambient = ambient_ + offset, *deltas = deltas_ + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0, amb = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);          /* __expf in the reference (fast intrinsic) */
            const float weight = alpha * T;
            r = fmaf(weight, rgbs[0], r);
            g = fmaf(weight, rgbs[1], g);
            b = fmaf(weight, rgbs[2], b);
            d = fmaf(weight, deltas[1], d);
            ws += weight;
            amb += ambient[0];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; ambient++; deltas += 2;
        }
        weights_sum[index] = ws; ambient_sum[index] = amb; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* kernel_composite_rays_train_backward -- raymarching.cu:711-810 */
ORC_API void orc_composite_rays_train_backward(const float *grad_weights_sum_, const float *grad_ambient_sum_, const float *grad_image_,
                                               const float *sigmas_, const float *rgbs_, const float *ambient_, const float *deltas_,
                                               const int32_t *rays, const float *weights_sum_, const float *ambient_sum_, const float *image_,
                                               uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas_, float *grad_rgbs_,
                                               float *grad_ambient_) {
    (void)ambient_; (void)ambient_sum_;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum_[index], gamb = grad_ambient_sum_[index];
        const float *gi = grad_image_ + (size_t)index * 3;
        const float r_final = image_[index * 3], g_final = image_[index * 3 + 1], b_final = image_[index * 3 + 2], ws_final = weights_sum_[index];
        const float *sigmas = sigmas_ + offset, *rgbs = rgbs_ + (size_t)offset * 3, *deltas = deltas_ + (size_t)offset * 2;
        float *gs = grad_sigmas_ + offset, *gr = grad_rgbs_ + (size_t)offset * 3, *ga = grad_ambient_ + offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);
            const float weight = alpha * T;
            r = fmaf(weight, rgbs[0], r);
            g = fmaf(weight, rgbs[1], g);
            b = fmaf(weight, rgbs[2], b);
            ws += weight;
            T *= 1.0f - alpha;
            gr[0] = gi[0] * weight; gr[1] = gi[1] * weight; gr[2] = gi[2] * weight;
            ga[0] = gamb;
            float acc = gi[0] * fmaf(T, rgbs[0], -(r_final - r));
            acc = fmaf(gi[1], fmaf(T, rgbs[1], -(g_final - g)), acc);
            acc = fmaf(gi[2], fmaf(T, rgbs[2], -(b_final - b)), acc);
            acc = fmaf(gws, 1 - ws_final, acc);
            gs[0] = deltas[0] * acc;
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; deltas += 2; gs++; gr += 3; ga++;
        }
    }
}

/* kernel_morton3D_dilation -- raymarching.cu:304-335: 6-neighbour max pool of a Morton-ordered [C, H^3] grid */
ORC_API void orc_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *out) {
    const uint32_t H3 = H * H * H;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)C * H3; n++) {
        const uint32_t c = (uint32_t)(n / H3), ind = (uint32_t)(n - (int64_t)c * H3);
        const uint32_t x = orc_morton3D_invert(ind), y = orc_morton3D_invert(ind >> 1), z = orc_morton3D_invert(ind >> 2);
        const float *g = grid + (size_t)c * H3;
        float res = grid[n];
        if (x + 1 < H) res = fmaxf(res, g[orc_morton3D(x + 1, y, z)]);
        if (x > 0) res = fmaxf(res, g[orc_morton3D(x - 1, y, z)]);
        if (y + 1 < H) res = fmaxf(res, g[orc_morton3D(x, y + 1, z)]);
        if (y > 0) res = fmaxf(res, g[orc_morton3D(x, y - 1, z)]);
        if (z + 1 < H) res = fmaxf(res, g[orc_morton3D(x, y, z + 1)]);
        if (z > 0) res = fmaxf(res, g[orc_morton3D(x, y, z - 1)]);
        out[n] = res;
    }
}

/* kernel_sph_from_ray -- raymarching.cu:162-199 */
ORC_API void orc_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords) {
    const float RPI = 0.3183098861837907f;
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float Cc = fmaf(oz, oz, fmaf(oy, oy, ox * ox)) - radius * radius;
        const float t = (-B + sqrtf(fmaf(B, B, -(A * Cc)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[2 * n] = fmaf(2 * theta, RPI, -1.0f);
        coords[2 * n + 1] = phi * RPI;
    }
}

/* grid encoder: dy_dx of the forward (gridencoder.cu:198-243), layout [B, L, D, C] */
ORC_API int orc_grid_encode_dydx(const float *inputs_, const float *embeddings, const int32_t *offsets, float *dy_dx_, uint32_t B, uint32_t D,
                                 uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D < 1 || D > 7 || C < 1 || C > 8) return -1;
#pragma omp parallel for schedule(static) collapse(2)
    for (int64_t level = 0; level < (int64_t)L; level++) {
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
            const float *inputs = inputs_ + (size_t)b * D;
            float *dy_dx = dy_dx_ + (size_t)b * D * L * C + (size_t)level * D * C;
            int flag_oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (inputs[d] < 0 || inputs[d] > 1) flag_oob = 1;
            if (flag_oob) {                                              /* :119-133: zeros */
                for (uint32_t i = 0; i < D * C; i++) dy_dx[i] = 0;
                continue;
            }
            const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
            float scale; uint32_t resolution;
            orc_grid_level_params((uint32_t)level, S, H, &scale, &resolution);
            float pos[7], pos_deriv[7]; uint32_t pos_grid[7];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) { pos_deriv[d] = 6 * pos[d] * (1 - pos[d]); pos[d] = pos[d] * pos[d] * fmaf(-2.0f, pos[d], 3.0f); }  /* :155-160 */
                else pos_deriv[d] = 1.0f;
            }
            for (uint32_t gd = 0; gd < D; gd++) {
                float results_grad[8] = {0};
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float w = scale;
                    uint32_t pgl[7];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                        else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                    }
                    pgl[gd] = pos_grid[gd];
                    const uint32_t il = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                    pgl[gd] = pos_grid[gd] + 1;
                    const uint32_t ir = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                    for (uint32_t ch = 0; ch < C; ch++) results_grad[ch] = fmaf(w * (grid[ir + ch] - grid[il + ch]), pos_deriv[gd], results_grad[ch]);
                }
                for (uint32_t ch = 0; ch < C; ch++) dy_dx[gd * C + ch] = results_grad[ch];
            }
        }
    }
    return 0;
}

/* kernel_grid_backward -- gridencoder.cu:247-340: grad [L,B,C] -> grad_embeddings (+=).  Serial over points (atomicAdd order in the
 * reference is unspecified; fp32 sums may differ in the last bits). */
ORC_API int orc_grid_encode_backward(const float *grad_, const float *inputs_, const int32_t *offsets, float *grad_embeddings, uint32_t B,
                                     uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     uint32_t interp) {
    if (D < 1 || D > 7 || C < 1 || C > 8) return -1;
    for (uint32_t level = 0; level < L; level++) {
        float *grad_grid = grad_embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        orc_grid_level_params(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float *inputs = inputs_ + (size_t)b * D;
            const float *grad = grad_ + (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (inputs[d] < 0 || inputs[d] > 1) oob = 1;
            if (oob) continue;
            float pos[7]; uint32_t pos_grid[7];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = pos[d] * pos[d] * fmaf(-2.0f, pos[d], 3.0f);
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[7];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t c = 0; c < C; c++) grad_grid[index + c] += w * grad[c];
            }
        }
    }
    return 0;
}

/* kernel_input_backward -- gridencoder.cu:342-368: grad_inputs[b][d] = sum_l sum_ch grad[l][b][ch] * dy_dx[b][l][d][ch] */
ORC_API void orc_grid_input_backward(const float *grad, const float *dy_dx_, float *grad_inputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float *dy_dx = dy_dx_ + (size_t)b * L * D * C;
        float result = 0;
        for (uint32_t l = 0; l < L; l++)
            for (uint32_t ch = 0; ch < C; ch++) result = fmaf(grad[(size_t)l * B * C + (size_t)b * C + ch], dy_dx[l * D * C + d * C + ch], result);
        grad_inputs[t] = result;
    }
}

/* kernel_grad_tv -- gridencoder.cu:505-597: total-variation gradient at the cells visited by `inputs`, accumulated into grad */
ORC_API int orc_grad_total_variation(const float *inputs_, const float *embeddings, float *grad_, const int32_t *offsets, float weight,
                                     uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    if (D < 1 || D > 7 || C < 1 || C > 8) return -1;
    for (uint32_t level = 0; level < L; level++) {
        const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        float *grad = grad_ + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        orc_grid_level_params(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float *inputs = inputs_ + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (inputs[d] < 0 || inputs[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pos_grid[7];
            for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f));
            float results[8] = {0}, idelta[8] = {0};
            const uint32_t index = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
            const float w = weight / (float)(2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur_d = pos_grid[d];
                if (cur_d < resolution) {
                    pos_grid[d] = cur_d + 1;
                    const uint32_t ir = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[ir + ch];
                        results[ch] += gv;
                        idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                if (cur_d > 0) {
                    pos_grid[d] = cur_d - 1;
                    const uint32_t il = orc_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = grid[index + ch] - grid[il + ch];
                        results[ch] += gv;
                        idelta[ch] = fmaf(gv, gv, idelta[ch]);
                    }
                }
                pos_grid[d] = cur_d;
            }
            for (uint32_t ch = 0; ch < C; ch++) grad[index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
    return 0;
}
