/*
 * gfpp_radnerf.h -- C ABI of libgfpp_radnerf.so, the MI355X (gfx950) native backend of the GeneFace++
 * motion2video NeRF render path (modules/radnerfs in the reference).
 *
 * Conventions (all entry points):
 *   - plain C: raw DEVICE pointers + sizes, no torch / ATen types;
 *   - every pointer must be device memory of the current HIP device, contiguous, 4-byte aligned;
 *   - outputs are caller-allocated and written in place (same contract as the reference's pybind functions:
 *     "arguments are at::Tensor already allocated by the caller");
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The reference launches on the
 *     legacy default stream with no error check; here the caller chooses the stream and every launch is checked;
 *   - return value: 0 on success, a positive hipError_t if the launch failed, a negative GFPP_E* code
 *     for invalid arguments.  gfpp_last_error() returns a thread-local message for the last non-zero return.
 *   - nothing here allocates, frees or synchronises, so every call is hipGraph-capturable.
 *
 * Section A mirrors, one to one, the reference's native extension API (the functions its Python shims call):
 *     _raymarching_face : modules/radnerfs/raymarching/src/raymarching.h:7-19, bindings.cpp:7-20
 *     _gridencoder      : modules/radnerfs/encoders/gridencoder/src/gridencoder.h:12-15
 *     _shencoder        : modules/radnerfs/encoders/shencoder/src/shencoder.h
 *     _freqencoder      : modules/radnerfs/encoders/freqencoder/src/freqencoder.h
 * Section B is the fused frame pipeline behind RADNeRF*.render() (modules/radnerfs/renderer.py:286-399,
 * radnerf_torso.py:86-199, radnerf_torso_sr.py:116-244), which has no native counterpart in the reference
 * (there it is ~500 PyTorch/extension launches and <=16 device->host syncs per frame).
 */
#ifndef GFPP_RADNERF_H
#define GFPP_RADNERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFPP_ABI_VERSION 1

#define GFPP_EINVAL (-1)       /* bad argument (null pointer, zero size where not allowed, ...) */
#define GFPP_EUNSUPPORTED (-2) /* unsupported D / C / degree / dtype combination (reference: std::runtime_error) */

typedef void *gfpp_stream_t; /* hipStream_t */

/* dtype codes for grid tables / encoder outputs */
#define GFPP_F32 0
#define GFPP_F16 1

int gfpp_abi_version(void);
const char *gfpp_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Section A.1 -- _raymarching_face
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces near_far_from_aabb (raymarching.h:7; kernel raymarching.cu:91-145).
 * rays_o, rays_d [N,3] f32; aabb [6] f32 (xmin,ymin,zmin,xmax,ymax,zmax); nears, fars [N] f32.
 * A ray missing the box gets near = far = FLT_MAX; near is clamped up to min_near. */
int gfpp_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                            float *nears, float *fars, gfpp_stream_t stream);

/* replaces morton3D / morton3D_invert (raymarching.h:9-10; raymarching.cu:214-241).
 * coords [N,3] i32 <-> indices [N] i32; x -> bit 0, y -> bit 1, z -> bit 2 of each 3-bit group. */
int gfpp_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, gfpp_stream_t stream);
/* replaces morton3D_invert (raymarching.h:10; raymarching.cu:228-241). */
int gfpp_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, gfpp_stream_t stream);

/* replaces packbits (raymarching.h:11; raymarching.cu:267-289).
 * grid [N*8] f32 -> bitfield [N] u8, bit i of byte n <=> grid[8n+i] > density_thresh. */
int gfpp_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, gfpp_stream_t stream);

/* replaces march_rays (raymarching.h:18; kernel raymarching.cu:827-929), argument order of the C++ binding.
 * rays_alive [n_alive] i32; rays_t, nears, fars [N] f32; rays_o, rays_d [N,3]; grid = density bitfield [C*H^3/8] u8;
 * xyzs, dirs [M,3], deltas [M,2] f32 with M >= n_alive*n_step, ZERO-INITIALISED by the caller (raymarching.py:384-386);
 * noises [n_alive] f32.  Slot layout n*n_step + s. */
int gfpp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                    const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                    const float *noises, gfpp_stream_t stream);

/* replaces composite_rays (raymarching.h:19; kernel raymarching.cu:942-1029).
 * sigmas [M], rgbs [M,3], deltas [M,2] f32; in/out: rays_alive [n_alive] (set to -1 when the ray terminates),
 * rays_t, weights_sum, depth [N], image [N,3]. */
int gfpp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                        const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                        float *image, gfpp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Section A.2 -- _gridencoder / _shencoder / _freqencoder (forward only; backward/TV are training-only)
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces grid_encode_forward (gridencoder.h:12; kernel gridencoder.cu:87-196).
 * inputs [B,D] f32 in [0,1]; embeddings [offsets[L],C] f32 or f16; offsets [L+1] i32 (device);
 * outputs [L,B,C] (level-major, same dtype as embeddings); D in {2,3}, C in {1,2,4,8}; S = log2(per_level_scale);
 * gridtype 0 = hash, 1 = tiled; interp 0 = linear, 1 = smoothstep.  dy_dx must be NULL (inference). */
int gfpp_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B,
                             uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, gfpp_stream_t stream);

/* replaces sh_encode_forward (shencoder.h; kernel shencoder.cu:28-68, degree <= 4 part).
 * inputs [B,3] f32, outputs [B,degree^2] f32; dy_dx must be NULL. */
int gfpp_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree, float *dy_dx,
                           gfpp_stream_t stream);

/* replaces freq_encode_forward (freqencoder.h; kernel freqencoder.cu:30-58).
 * inputs [B,D] f32, outputs [B,C] f32, C = D + 2*D*deg; layout [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]. */
int gfpp_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs,
                             gfpp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Section A.3 -- ray generation (modules/radnerfs/utils.py:283-364 get_rays with N = -1; pure PyTorch in the
 * reference, ~10 launches per frame and 6.3 MB/frame kept resident for a whole clip)
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces get_rays with N = -1 (modules/radnerfs/utils.py:283-364, pixel centres :302-304, directions :352-363).
 * pose [4,4] f32 row-major cam2world (device); rays_o, rays_d [H*W,3] f32, pixel index h*W + w,
 * dir = normalize(((w+0.5-cx)/fx, (h+0.5-cy)/fy, 1)) @ R^T. */
int gfpp_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                  float *rays_d, gfpp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GFPP_RADNERF_H */
