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
 *     _raymarching_face : modules/radnerfs/raymarching/src/raymarching.h:7-20, bindings.cpp:7-20
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

#define GFPP_ABI_VERSION 8

#define GFPP_EINVAL (-1)       /* bad argument (null pointer, zero size where not allowed, ...) */
#define GFPP_EUNSUPPORTED (-2) /* unsupported D / C / degree / dtype combination (reference: std::runtime_error) */

typedef void *gfpp_stream_t; /* hipStream_t */

/* dtype codes for grid tables / encoder outputs */
#define GFPP_F32 0
#define GFPP_F16 1
#define GFPP_BF16 2 /* MFMA operand type of the 16-bit head kernel only */

int gfpp_abi_version(void);
const char *gfpp_last_error(void);
/* sizeof() of a struct of this header as the library was compiled, by name without the gfpp_ prefix ("frame_ws", "head_model", "torso_model",
 * "cond_model", "grid_desc", "grid_level", "sr_model", "sr_ws"); 0 for an unknown name.  A binding that mirrors the structs (ctypes, cgo, JNI)
 * checks its own layout against this at load time instead of corrupting memory on a mismatch. */
unsigned gfpp_struct_size(const char *name);

/* Launch tuning of the whole library: ONE record, set once by the host (genefaceplusplus_amd/_lib.py::set_tuning; before round 6 these were getenv() reads
 * inside the launch paths).  Every field's 0 / default is the shipped, measured optimum; the other values are the A/B partners the tests and tools use.
 * gfpp_set_tuning copies the record (size-checked); it is read at every issue -- a captured graph keeps what it was captured with. */
typedef struct gfpp_tuning {
    uint32_t size;               /* sizeof(gfpp_tuning) of the caller */
    int32_t trip_pool;           /* fp32 trip kernel: 1 = workgroup sample pool (default), 0 = one tile per wavefront */
    int32_t lp_separate_trips;   /* 16-bit trip launches before the multi-trip launch takes over: -1 = default (5) */
    int32_t occ_clip;            /* pre-march stops at the occupancy bounds: 1 (default) / 0 */
    uint32_t barrier_spins;      /* multi-trip launch's barrier time-out in spins: 0 = default (1 << 22); tests force a time-out with a small value */
    uint32_t persist_caps;       /* local n_step caps of the persistent launch by workgroup round, 4 bits each: 0 = default (4,4,4,8,8,8,8,8) */
    int32_t persist_xcd;         /* XCD-local tile ownership of the persistent launch: 0 = off (default; measured: traffic down, time unchanged), 1, 2 */
    int32_t torso_group_wgs;     /* persistent workgroups per CU of the torso MLP group launch: 0 = default */
    int32_t sr_fuse_first;       /* block 0's first SR convolution inside the second one's halo load: 1 (default) / 0 */
    int32_t sr_final_resident;   /* last SR layer with LDS-resident weights: 1 (default) / 0 = one workgroup per patch */
    int32_t sr_up_poly;          /* SR up-sampling layer in polyphase form + FIR GEMM (k_sr_up_poly) when the model carries w_up_poly: 0 (default) = the composed 3 x 3
                                  * convolution.  OFF by default although it is 6 % faster per forward: while its MFMA phase shares a CU with OTHER kernels' wavefronts
                                  * (a second clip lane's torso / pre-march launch) those kernels return different bits now and then -- 16 pixels of a torso pass off
                                  * by 1e-3 .. 5e-2 in 2-40 % of the frames, never with the composed launch (tools/interference_probe.py, tools/clip_interference.py,
                                  * docs/LAB_NOTEBOOK.md round 6).  Safe where nothing else runs beside it (one lane, plain launches). */
    int32_t grid_bwd_scatter;    /* table gradient: 0 = LDS ranges (default), 1 = device atomics (the round-2 path) */
    int32_t wgrad_tr;            /* transposing-read weight gradients: 1 (default) / 0 */
    int32_t grid_bwd_bins;       /* table gradient: ranges walk per-range point lists when the caller brings the `bins` scratch: 1 (default) / 0 = all points per range */
    int32_t march_fixed_step;    /* pre-march: rays whose step is constant and longer than a voxel probe every chain point without the exit-face arithmetic
                                  * (march_device.h::march_one_ray_fixed_step, same bits): 1 (default) / 0 = the general walk for every ray */
} gfpp_tuning;
int gfpp_set_tuning(const gfpp_tuning *t);   /* NULL restores the defaults */
int gfpp_get_tuning(gfpp_tuning *out);       /* out->size must be set */

/* ------------------------------------------------------------------------------------------------------------
 * Section A.1 -- _raymarching_face
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces near_far_from_aabb (raymarching.h:7; kernel raymarching.cu:91-145).
 * rays_o, rays_d [N,3] f32; aabb [6] f32 (xmin,ymin,zmin,xmax,ymax,zmax); nears, fars [N] f32.
 * A ray missing the box gets near = far = FLT_MAX; near is clamped up to min_near. */
int gfpp_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                            float *nears, float *fars, gfpp_stream_t stream);

/* Conservative bounds of the occupied cells of a density bitfield -> out6 (device, 6 floats: lo xyz, hi xyz; lo > hi when no bit is set): every
 * set bit (bit layout of kernel_packbits, raymarching.cu:268-300; cell = level * H^3 + Morton code, raymarching.cu:56-88) contributes the world extent
 * its cell has in kernel_march_rays' position -> cell mapping (raymarching.cu:880-894; the first / last cell of an axis reaches to the scene bound,
 * positions are clamped into them), the result is widened by one cell.  For gfpp_head_model.occ_aabb; bitfield must be 4-byte aligned. */
int gfpp_occupancy_bounds(const uint8_t *bitfield, uint32_t cascade, uint32_t grid_size, float bound, float *out6, gfpp_stream_t stream);

/* replaces morton3D / morton3D_invert (raymarching.h:9-10; raymarching.cu:214-241).
 * coords [N,3] i32 <-> indices [N] i32; x -> bit 0, y -> bit 1, z -> bit 2 of each 3-bit group. */
int gfpp_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, gfpp_stream_t stream);
/* replaces morton3D_invert (raymarching.h:10; raymarching.cu:228-241). */
int gfpp_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, gfpp_stream_t stream);

/* replaces packbits (raymarching.h:11; raymarching.cu:267-289).
 * grid [N*8] f32 -> bitfield [N] u8, bit i of byte n <=> grid[8n+i] > density_thresh. */
int gfpp_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, gfpp_stream_t stream);

/* replaces march_rays (raymarching.h:19; kernel raymarching.cu:827-929), argument order of the C++ binding.
 * rays_alive [n_alive] i32; rays_t, nears, fars [N] f32; rays_o, rays_d [N,3]; grid = density bitfield [C*H^3/8] u8;
 * xyzs, dirs [M,3], deltas [M,2] f32 with M >= n_alive*n_step, ZERO-INITIALISED by the caller (raymarching.py:384-386);
 * noises [n_alive] f32.  Slot layout n*n_step + s. */
int gfpp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                    const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                    const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                    const float *noises, gfpp_stream_t stream);

/* replaces composite_rays (raymarching.h:20; kernel raymarching.cu:942-1029).
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

/* replaces sh_encode_forward (shencoder.h:9; kernel shencoder.cu:28-352, degree <= 4 part).
 * inputs [B,3] f32, outputs [B,degree^2] f32; dy_dx NULL or [B,3,degree^2] f32 (rows d/dx, d/dy, d/dz of the features). */
int gfpp_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree, float *dy_dx,
                           gfpp_stream_t stream);

/* replaces sh_encode_backward (shencoder.h:10; kernel shencoder.cu:359-382): grad [B,degree^2], dy_dx from the forward,
 * grad_inputs [B,3] += grad . dy_dx (the caller zero-initialises, sphere_harmonics.py:47-49). `inputs` is unused, as in the reference. */
int gfpp_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree, const float *dy_dx,
                            float *grad_inputs, gfpp_stream_t stream);

/* replaces freq_encode_forward (freqencoder.h; kernel freqencoder.cu:30-58).
 * inputs [B,D] f32, outputs [B,C] f32, C = D + 2*D*deg; layout [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]. */
int gfpp_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs,
                             gfpp_stream_t stream);

/* replaces freq_encode_backward (freqencoder.h:10; kernel freqencoder.cu:63-93): grad, outputs [B,C] (outputs = the forward result,
 * its sin / cos columns are the derivative factors), grad_inputs [B,D] = grad_x + sum_f 2^f (grad_sin cos - grad_cos sin). */
int gfpp_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                              float *grad_inputs, gfpp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Section A.3 -- ray generation (modules/radnerfs/utils.py:283-364 get_rays with N = -1; pure PyTorch in the
 * reference, ~10 launches per frame and 6.3 MB/frame kept resident for a whole clip)
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces get_rays with N = -1 (modules/radnerfs/utils.py:283-364, pixel centres :302-304, directions :352-363).
 * pose [4,4] f32 row-major cam2world (device); rays_o, rays_d [H*W,3] f32, pixel index h*W + w,
 * dir = normalize(((w+0.5-cx)/fx, (h+0.5-cy)/fy, 1)) @ R^T. */
int gfpp_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                  float *rays_d, gfpp_stream_t stream);

/* replaces get_rays with N > 0 or rect (modules/radnerfs/utils.py:310-343 pick the pixel indices, :352-363 the directions): the rays of the
 * listed pixels only.  inds [n_rays] i64 (= h*W + w, what torch.randint / torch.where produce), rays_o, rays_d [n_rays,3] f32. */
int gfpp_get_rays_at(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t *inds,
                     uint32_t n_rays, float *rays_o, float *rays_d, gfpp_stream_t stream);

/* replaces the per-frame `(pred_rgb * 255.).int() ... astype(np.uint8)` host conversion of the caller
 * (inference/genefacepp_infer.py:468): rgb [n_values] f32 in [0,1] -> out [n_values] u8, truncating. rgb 16-byte aligned. */
int gfpp_rgb_to_u8(const float *rgb, uint64_t n_values, uint8_t *out, gfpp_stream_t stream);

/* ---- the caller's frame loop below Python (inference/genefacepp_infer.py:460-469: `for i in range(num_frames): render(...)`, then the uint8
 * conversion of :468 / :505) ---------------------------------------------------------------------------------------------------------------
 * A clip's driving signals are device-resident, one packed row per frame (clip.ClipRenderer.prepare).  With the job record below -- also in
 * DEVICE memory, written by the host once per render call -- a frame's captured graph fetches its own inputs and stores its own output:
 * gfpp_clip_fetch copies row order[cursor[lane]] of `packed` into the graph's static input buffer, gfpp_clip_store_u8 converts the frame to uint8
 * straight into slot cursor[lane] % ring_frames of `out` and advances the lane's cursor by `lanes`.  Per frame the host then issues nothing but
 * the graph launch, and gfpp_graph_replay issues those from C for a whole run of frames (lane = frame % lanes, one stream per lane). */
typedef struct gfpp_clip_job {
    const float *packed;      /* [frames, row_floats] f32 */
    const int32_t *order;     /* [n] i32: clip frame index of the k-th frame of this job */
    uint8_t *out;             /* [ring_frames, frame_bytes] u8 */
    uint64_t frame_bytes;
    uint32_t row_floats;
    uint32_t n;               /* frames of the job (positions >= n fetch and store nothing) */
    uint32_t lanes;           /* frames in flight: lane l renders the positions l, l + lanes, ... */
    uint32_t ring_frames;     /* positions wrap around in `out` (= n: no wrap) */
    uint32_t cursor[8];       /* per lane: position of the lane's next frame (host: cursor[l] = l) */
    uint32_t ticket[8];       /* per lane: workgroups of the storing launch that are done (host: 0); the last one advances the cursor */
} gfpp_clip_job;

/* job (DEVICE pointer) -> static_in [row_floats] f32 (device); replaces the per-frame indexing `cond_inp[i], poses[i], lm68s[i]`
 * of inference/genefacepp_infer.py:461-463 */
int gfpp_clip_fetch(const gfpp_clip_job *job, uint32_t lane, float *static_in, uint32_t row_floats, gfpp_stream_t stream);

/* the conversion of gfpp_rgb_to_u8 (inference/genefacepp_infer.py:468) written to the job's output slot of this lane's frame; the launch's last
 * workgroup then advances cursor[lane] by `lanes`.  rgb [n_values] f32 16-byte aligned; n_values must equal the job's frame_bytes (any size: slots that
 * are not 4-byte aligned are written byte by byte).
 * (gfpp_torso_frame_lp does the same itself when ws->clip_job is set: no separate launch on the frame's critical path.) */
int gfpp_clip_store_u8(gfpp_clip_job *job, uint32_t lane, const float *rgb, uint64_t n_values, gfpp_stream_t stream);

/* (ABI 6) The same two for a frame GROUP -- K consecutive positions of a lane rendered by one graph launch (the caller's loop of
 * inference/genefacepp_infer.py:460-469 taken K frames at a time): frame `sub` of the group takes position cursor[lane] + sub; the store adds `advance` to the
 * lane's cursor when it is done (0: the job's `lanes`; 0xFFFFFFFF: nothing -- every frame of a group but the last, which passes K * lanes).  Groups are dealt
 * to the lanes round-robin: the host starts cursor[l] at l * K.  Frame sizes need not be multiples of four bytes. */
int gfpp_clip_fetch_at(const gfpp_clip_job *job, uint32_t lane, uint32_t sub, float *static_in, uint32_t row_floats, gfpp_stream_t stream);
/* ... and the rows of all `count` frames of the lane's next group in one launch: row k -> static_in + k * row_floats (the indexing
 * `cond_inp[i], poses[i], lm68s[i]` of inference/genefacepp_infer.py:461-463 for `count` consecutive i) */
int gfpp_clip_fetch_group(const gfpp_clip_job *job, uint32_t lane, uint32_t count, float *static_in, uint32_t row_floats, gfpp_stream_t stream);
int gfpp_clip_store_u8_at(gfpp_clip_job *job, uint32_t lane, uint32_t sub, uint32_t advance, const float *rgb, uint64_t n_values, gfpp_stream_t stream);

/* The frame loop of inference/genefacepp_infer.py:460-469 for `count` frames: frame k is one launch of the captured graph of lane
 * (first_lane + k) % lanes on that lane's stream.  execs: [lanes] hipGraphExec_t, streams: [lanes] hipStream_t (host arrays).
 * max_ahead: 0 = queue everything at once; d > 0 = the issuing thread keeps at most d frames per lane queued ahead of the GPU. */
int gfpp_graph_replay(void *const *execs, void *const *streams, uint32_t lanes, uint32_t first_lane, uint32_t count, uint32_t max_ahead);

/* ------------------------------------------------------------------------------------------------------------
 * Section B -- fused frame pipeline (no native counterpart in the reference; replaces the Python loop of
 * NeRFRenderer.render, modules/radnerfs/renderer.py:341-397, its copies radnerf_torso.py:128-151 /
 * radnerf_torso_sr.py:158-181, RADNeRF.forward radnerf.py:108-141 evaluated per trip, and the torso pass
 * radnerf_torso.py:156-197 / radnerf_torso_sr.py:186-231 incl. forward_torso :51-84 / :75-114).
 *
 * Loop control lives on the device: trip k reads the alive-ray count that trip k-1 produced, derives
 * n_step = clamp(N / n_alive, 1, 8) and the cumulative step exactly like renderer.py:354-384, and exits by itself when
 * the reference's loop would have ended -- so a frame is a fixed sequence of launches with no host synchronisation.
 *
 * All descriptor structs live in HOST memory and hold DEVICE pointers; they are read at call time only.
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces the per-thread level setup of kernel_grid (gridencoder.cu:137-139) with a host-side table (no GPU needed):
 * per-level scale = exp2f(level*S)*H - 1 and resolution = ceil(scale) + 1, computed in
 * fp32 exactly as the device code of gridencoder.cu:138-139 states them. */
int gfpp_grid_level_table(uint32_t L, float S, uint32_t H, float *scale_out, uint32_t *resolution_out);

#define GFPP_LEVEL_SLOW 1u /* flags: the level is addressed by the hash, or needs a true modulo -> generic lookup */

typedef struct gfpp_grid_level { /* one entry per level, uploaded to device memory by the caller */
    float scale;
    uint32_t resolution;
    uint32_t offset; /* first table row of the level */
    uint32_t size;   /* rows in the level (the reference's hashmap_size) */
    /* index arithmetic of get_grid_index (gridencoder.cu:66-84) resolved per level, see gfpp_grid_levels_fill */
    uint32_t sy;    /* stride of coordinate 1 in the linear index, 0 if the `stride <= hashmap_size` test drops it */
    uint32_t sz;    /* stride of coordinate 2, 0 if dropped (or D == 2) */
    uint32_t mask;  /* index % size == index & mask: size - 1 for power-of-two sizes, 0xFFFFFFFF where index < size always */
    uint32_t flags; /* GFPP_LEVEL_SLOW */
} gfpp_grid_level;

/* Fills all fields of `levels[0..L)` (HOST memory) from the encoder's hyper-parameters and its `offsets` array (HOST copy of
 * GridEncoder.offsets, L+1 entries; grid.py:121-132): scale / resolution as gfpp_grid_level_table, offset / size from the
 * offsets, and the resolved index arithmetic of get_grid_index (gridencoder.cu:66-84) for D in {2,3}. */
int gfpp_grid_levels_fill(uint32_t D, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, const int32_t *offsets,
                          uint32_t row_pad, gfpp_grid_level *levels); /* row_pad: rows of padding after every level (0 or 1) */

typedef struct gfpp_grid_desc {
    const void *table;             /* [rows, 2] f32 or f16 (level_dim is 2 on this path) */
    const gfpp_grid_level *levels; /* [L] in device memory */
    int32_t dtype;                 /* GFPP_F32 | GFPP_F16 */
    uint32_t D;                    /* 2 or 3 */
    uint32_t L;                    /* 16 */
    uint32_t gridtype;             /* 0 hash, 1 tiled */
    uint32_t interp;               /* 0 linear, 1 smoothstep */
    uint32_t align_corners;
    const gfpp_grid_level *levels_host; /* the same [L] entries in HOST memory (the 16-bit kernel passes them as kernel arguments) */
    uint32_t row_padded;           /* 1: `table` is the padded copy -- level l starts at row offsets[l] + l and is followed by one row that
                                    * repeats its first row, so that the x+1 neighbour of a level's last row (index modulo, gridencoder.cu:82)
                                    * is the next row in memory; levels[l].offset already includes the padding */
} gfpp_grid_desc;

/* Per-frame conditioning: replaces RADNeRF.cal_cond_feat (radnerf.py:88-106 = AudioNet cond_encoder.py:98-143, the blink branch
 * radnerf.py:97-103, AudioAttNet cond_encoder.py:146-180).  All weights in PyTorch layout (Conv1d [out,in,3], Linear [out,in]), fp32,
 * device memory.  One launch of one workgroup; cond_feat stays on the device for gfpp_head_frame_begin. */
typedef struct gfpp_cond_model {
    uint32_t smo;        /* smo_win_size: windows per frame (the batch of AudioNet, the sequence of AudioAttNet) */
    uint32_t t_win;      /* cond_win_size: time steps per window (1 for the lm3d configs, 16 for the audio ones) */
    uint32_t c_in;       /* 204 (lm68 x 3), 29 or 44 */
    uint32_t dim_aud;    /* cond_out_dim (<= 64) */
    uint32_t strides[4]; /* cond_encoder.py:103-114 */
    const float *conv_w[4], *conv_b[4]; /* channels c_in -> 32 -> 32 -> 64 -> 64 */
    const float *fc_w[2], *fc_b[2];     /* 64 -> 64 -> dim_aud */
    uint32_t blink_dim;                 /* eye_blink_dim, 0 = no blink branch */
    const float *blink_emb;             /* blink_embedding.weight[0], [dim_aud/2] */
    const float *blink_w[2], *blink_b[2];
    uint32_t with_att;
    const float *att_conv_w[5], *att_conv_b[5]; /* channels dim_aud -> 16 -> 8 -> 4 -> 2 -> 1 over the smo axis */
    const float *att_fc_w, *att_fc_b;           /* [smo, smo] */
    /* optional accelerators (0 / NULL = plain): */
    uint32_t center_tap_only; /* t_win == 1 only: conv_w[l] hold the centre taps W[:, :, 1] as [out, in] matrices (the outer taps only see padding) */
    const float *blob;        /* all weight / bias pointers above point into this one contiguous, 16-byte aligned array: it is copied */
    uint32_t blob_floats;     /* to LDS in one burst when it fits (multiple of 4) */
} gfpp_cond_model;

/* replaces RADNeRF.cal_cond_feat (radnerf.py:88-106).  cond [smo, t_win, c_in] f32; eye_area [1] f32 or NULL (= 0, like
 * eye_area_percent=None); cond_feat [dim_aud] f32 out ([smo, dim_aud] when with_att == 0). */
int gfpp_cond_feat(const gfpp_cond_model *model, const float *cond, const float *eye_area, float *cond_feat, gfpp_stream_t stream);

/* RADNeRF.cal_cond_feat (radnerf.py:88-106) for `count` windows in ONE launch, one workgroup per window (the same kernel as gfpp_cond_feat, so the
 * same bits): window k at cond + k cond_stride, its eye value at eye_area + k eye_stride (eye_area NULL = 0 for all), its result at cond_feat +
 * k out_stride (strides in floats).  The caller's frame loop computes it per frame (genefacepp_infer.py:461-463 -> render -> cal_cond_feat); a clip
 * renderer that holds the driving signals of all frames takes the 16 dependent layers (~40 us on one CU) out of every frame this way. */
int gfpp_cond_feat_batch(const gfpp_cond_model *model, const float *cond, uint32_t cond_stride, const float *eye_area, uint32_t eye_stride,
                         float *cond_feat, uint32_t out_stride, uint32_t count, gfpp_stream_t stream);

/* RADNeRF.cal_cond_feat (radnerf.py:88-106) inside a TRAINING step (tasks/radnerfs/radnerf.py:101-176 trains AudioNet / AudioAttNet with the field): the same
 * networks with every activation kept (forward) and their whole backward pass, one launch of one workgroup each -- autograd runs them as ~120 eager launches,
 * each convolution call ~0.1 ms of host time in MIOpen.  fp32 throughout (autocast would run the layers in half).
 *   model: the parameters AS THEY ARE (Conv1d weights [out, in, 3], center_tap_only = 0, blob = NULL), fp32 device pointers.
 *   saved: gfpp_cond_feat_train_floats(model, 0) floats, written by the forward pass, read by the backward pass; scratch: (model, 1) floats.
 *   grads: a gfpp_cond_model whose pointers say where each parameter's gradient goes (same shapes; the dimensions are taken from `model`); every one is OVERWRITTEN.
 *   grad_out: [dim_aud] ([smo, dim_aud] without the attention net).  cond itself gets no gradient (a driving signal). */
uint32_t gfpp_cond_feat_train_floats(const gfpp_cond_model *model, int scratch);
int gfpp_cond_feat_train_forward(const gfpp_cond_model *model, const float *cond, const float *eye_area, float *cond_feat, float *saved, gfpp_stream_t stream);
int gfpp_cond_feat_train_backward(const gfpp_cond_model *model, const gfpp_cond_model *grads, const float *cond, const float *eye_area, const float *saved,
                                  const float *grad_out, float *scratch, gfpp_stream_t stream);

/* MLP weight packing for the MFMA kernels ("fragment order", fp32):
 *   a dense layer out[128] = W[128,K] x  is evaluated as a sequence of K/2 rank-2 updates with
 *   v_mfma_f32_32x32x2_f32; update `s` consumes the input pair (k0[s], k1[s]).  Packed array P[s/4][m][lane][s%4] =
 *   W[32*m + (lane & 31)][ (lane < 32) ? k0[s] : k1[s] ],  m = 0..3 (32-row output tiles), lane = 0..63.
 *   Pair orders (rr(r) = (r&3) + 8*(r>>2)):
 *     encoder features (32 values in two halves of 16):  (s, 16+s), s = 0..15   [SH, 16 values: (s, 8+s), s = 0..7]
 *     previous-layer activations (128):  step 16*m'+r -> (32*m'+rr(r), 32*m'+rr(r)+4), m' = 0..3, r = 0..15
 *   which is exactly how the accumulator registers of one layer line up as B operands of the next, so activations
 *   never leave the register file.  Skinny layers (3 or 1 outputs) are packed for VALU dot products:
 *   V[h][c][16*m'+r] = W[c][32*m'+rr(r)+4*h], h = 0,1.  Bias vectors in the same order: Bf[h][16*m+r] = b[32*m+rr(r)+4*h]. */
typedef struct gfpp_head_model {
    float aabb[6];
    float min_near;
    float bound;
    float density_scale;
    uint32_t cascade;   /* C */
    uint32_t grid_size; /* H */
    const uint8_t *density_bitfield;
    gfpp_grid_desc pos_grid; /* D = 3 */
    gfpp_grid_desc amb_grid; /* D = ambient_coord_dim (2 or 3) */
    /* ambient_net 96->128->128->amb_D (cond columns folded into a per-frame bias) */
    const float *amb_w0;      /* packed, 16 steps (position features) */
    const float *amb_w0_cond; /* [128, cond_dim] row-major = W0[:, 32:] */
    const float *amb_w1;      /* packed, 64 steps */
    const float *amb_w2;      /* VALU pack [2][amb_D][64] */
    /* sigma_net 64->128->128->129 */
    const float *sig_w0;     /* packed, 32 steps: 16 position + 16 ambient feature pairs */
    const float *sig_w1;     /* packed, 64 steps */
    const float *sig_w2_geo; /* packed, 64 steps: rows 1..128 of the last layer */
    const float *sig_w2_sig; /* VALU pack [2][1][64]: row 0 (density logit) */
    /* color_net 148->128->3 (individual-code columns folded into a per-frame bias) */
    const float *col_w0;     /* packed, 72 steps: 8 SH pairs + 64 geo-feature pairs */
    const float *col_w0_ind; /* [128, ind_dim] row-major = W0[:, 144:] (NULL if ind_dim == 0) */
    const float *col_w1;     /* VALU pack [2][3][64] */
    uint32_t cond_dim;       /* 64 */
    uint32_t ind_dim;        /* 4 */
    /* 16-bit operand image for gfpp_head_frame_march_lp (NULL if not built): the five wide layers as K = 16 MFMA steps,
     * 31 steps x 4 row tiles x 64 lanes x 8 halves (126 976 B), L[step][m][lane][e] = W[32*m + (lane & 31)][col(step, lane >> 5, e)]:
     *   steps  0- 1  ambient_net.0, position features:            col = 2*(2*((8*s + e)/2) + h) + e%2  (half-wave h encodes levels h, h+2, ..)
     *   steps  2- 9  ambient_net.1, activations:                  col = act(s) = 32*(s>>1) + rr(8*(s&1) + e) + 4*h
     *   steps 10-13  sigma_net.0:  s<2 position features as above, s>=2 ambient features 32 + (the same formula with s-2)
     *   steps 14-21  sigma_net.1, activations
     *   steps 22-30  MERGED colour layer  [ C0[:, :16] | C0[:, 16:144] @ S2[1:129, :] ]  (color_net.0 x sigma_net.2 geo rows,
     *                no activation lies between them, radnerf.py:126-137): step 22 SH col = 8*h + e, steps 23-30 16 + act(s-23)
     * with h = lane >> 5, rr(r) = (r&3) + 8*(r>>2).  lp_dtype = GFPP_F16 (what the reference's autocast inference uses) or
     * GFPP_BF16.  The folded biases and all accumulation stay fp32.
     * (ABI 7) With lp_dtype = GFPP_BF16 the steps 0-9 (ambient_net.0 / .1) and the rows 0-2 of lp_skinny (ambient_net.2) are F16 bit patterns: ambient_net's
     * output is a COORDINATE of the second hash grid, and 8-bit significands displace it by up to five cells of the finest level (the one layer group whose
     * rounding kept bf16 frames below SURVEY 8c's 45 dB; csrc/frame_head_lp.hip::LpAmbient, tools/lp_emulate.py).  Everything behind the ambient grid is bf16. */
    const void *lp_weights;
    /* the three skinny output layers for the same kernel, 16-bit, [2 half-waves][7 rows][64]: rows 0-2 ambient_net.2 (zero rows beyond
     * ambient_coord_dim), row 3 sigma_net.2 row 0 (density logit), rows 4-6 color_net.1; entry k = 8*s + e of half h = W[row][act(s)] as above,
     * so that a row is a sequence of packed dot products against the operand registers of the preceding layer (1 792 B) */
    const void *lp_skinny;
    int32_t lp_dtype;
    /* (ABI 5) conservative world-space bounds of the occupied cells of density_bitfield, lo xyz | hi xyz (gfpp_occupancy_bounds).  The pre-march
     * kernels stop a ray where it leaves these bounds -- kernel_march_rays (raymarching.cu:828-940) would only skip empty cells from there to `far`
     * -- and give a ray that misses them no sample; the sample times are the bits of the full march.  hi <= lo on an axis (e.g. all zero): unknown,
     * every ray is marched to `far`. */
    float occ_aabb[6];
    /* (ABI 6) 16-bit CORNER-BLOCK copies of the two tables for the 16-bit kernels (gfpp_head_frame_persist_lp / _trips_lp / gfpp_head_eval_samples_lp), used when
     * no level of either grid is GFPP_LEVEL_SLOW (hash-addressed or true modulo; such models are read through pos_grid / amb_grid by the generic lookup):
     * dtype GFPP_F16, row_padded = 2, `levels` filled with row_pad = 0.  Row r of level l (16 bytes, 8 halves) holds both channels of the four corners
     * (r, r + 1, r + sy, r + sy + 1) of the x-y cell that starts at r, indices modulo the level size:
     *     c0(x,y) c0(x+1,y) | c1(x,y) c1(x+1,y) | c0(x,y+1) c0(x+1,y+1) | c1(x,y+1) c1(x+1,y+1)
     * so that a level costs one 16-byte gather per z plane (32 gathers from <= 25 cache lines per sample and 3-D grid instead of 64 from 64) and the
     * interpolation is four packed dot products per plane (v_dot2_f32_f16: fp16 corner weights, fp32 accumulation).  The reference's autocast inference reads
     * a half table too (grid.py:43-47).  The exact-fp32 kernels never read these.  BOTH left empty (table == NULL): the caller's opt-out -- the 16-bit kernels then
     * read pos_grid / amb_grid through the generic lookup (fp32 corner values and weights, 2 060 B per sample), as for hash-grid models. */
    gfpp_grid_desc pos_grid_blk;
    gfpp_grid_desc amb_grid_blk;
} gfpp_head_model;

/* per-frame device workspace (caller-allocated, reusable across frames) */
typedef struct gfpp_frame_ws {
    uint32_t N;          /* rays in the frame */
    float *nears;        /* [N] */
    float *fars;         /* [N] */
    float *ray_state;    /* [N,8] f32: ONE 32-byte record per ray {weights_sum, depth, r, g, b (premultiplied head colour), t or sample cursor, -, -}
                          * -- the running state that raymarching.cu:942-1029 keeps in the separate rays_t / weights_sum / depth / image arrays.
                          * A trip touches scattered ray ids: a record is one 16-byte + one 8-byte access and one dirty 32-byte sector per ray,
                          * where five arrays cost five sectors (measured 4-8x write amplification) */
    int32_t *alive[2];   /* [N] each: ping-pong lists of alive ray ids */
    int32_t *counters;   /* [192] i32 (ABI 4; 128 before): counters[k] = rays alive at the start of trip k; counters[64+k] = samples trip k evaluated;
                          * counters[127] = barrier word of the 16-bit kernel's multi-trip launch (negative = a barrier timed out);
                          * counters[128 + m], m < 32 = the budget histogram of gfpp_head_frame_persist_lp: rays whose compositing ends at
                          * sample index m = min(samples the ray owns, index of the first sample whose pre-sample transmittance is below
                          * T_thresh) -- the whole (n_alive, n_step) sequence of renderer.py:359-364,384 is a function of it;
                          * counters[168] = samples that launch evaluated, counters[169] / [170] = workgroup rounds (sum / max), [171] = most samples of one
                          * workgroup, [172..175] = shader cycles / 1024 by phase (fetch, compaction, evaluate, composite; summed over workgroups), [176] longest workgroup.
                          * All 192 are zeroed by gfpp_head_frame_begin / gfpp_head_frame_begin_premarch (counters[0] = N). */
    float *frame_consts; /* [256] f32: folded biases of ambient_net.0 and color_net.0 in fragment order */
    float *sample_t;        /* 16-bit kernel only: [N, sample_stride] f32, t of every occupied sample of each ray in march order */
    uint32_t *sample_cnt;   /* 16-bit kernel only: [N] u32 */
    uint32_t sample_stride; /* >= max_steps + 7 */
    uint64_t *phase_cycles; /* optional (NULL = off), [64][8] u64, caller-zeroed: gfpp_head_frame_march_lp adds, per trip, the shader
                             * cycles its wavefronts spent in {weight copy, sample fetch, evaluate, composite} and, splitting evaluate,
                             * {position encode, ambient MLP, ambient encode, sigma + colour MLP} -- a profiling aid */
    uint32_t separate_trips; /* 16-bit kernel: how many trips get a launch of their own before ONE multi-trip launch (device-wide barrier
                              * between its trips) takes the rest; 0 = default (5: with the shipped schedule the sixth trip uses up the step budget, so the multi-trip launch runs it and returns without a barrier).  Set it >= max_steps when frames are in flight on
                              * several streams at once (one workspace each): two multi-trip launches spinning at their barriers could
                              * keep each other's workgroups from ever becoming resident. */
    /* Ray-tile sharding of ONE frame over several GPUs (renderer.py:364: n_step = clamp(N // n_alive, 1, 8) is a function of the FRAME-wide
     * alive count, so a rank that renders only a tile of the rays needs the global numbers to give every ray the reference's sample budget):
     * gcounters == NULL: single-GPU frame (everything above).  Otherwise this workspace holds one tile of N rays of a frame of N_global rays;
     * gcounters [64] i32 holds the frame-wide alive counts per trip -- the caller sets gcounters[0] = N_global and, after issuing trip k
     * (trip_first = k, trip_count = 1), all-reduces counters[k+1] of all ranks into gcounters[k+1] before issuing trip k+1.  The trip
     * kernels take n_step / the loop exit from gcounters and their own work list from counters. */
    const int32_t *gcounters;
    uint32_t N_global;
    uint32_t trip_first;   /* gfpp_head_frame_trips / _trips_lp issue the trips [trip_first, trip_first + trip_count) only; trip_count == 0: all */
    uint32_t trip_count;
    uint32_t full_grid_trips; /* 0: every trip launch covers the whole device.  k > 0: trips >= k are launched on a small grid (32 workgroups of the
                               * 16-bit kernel, 64 of the fp32 kernel) -- for callers with several frames in flight that know from an earlier frame
                               * that the loop normally ends after k trips (renderer.py:364: n_step grows as rays die, the step budget is used up
                               * after ~6 trips of 16).  With separate_trips = k the 16-bit entry then issues k full launches and ONE multi-trip launch
                               * of 32 workgroups for the rest, which normally finds nothing left and returns before any barrier: launches of that
                               * size from a few streams cannot starve each other (together they fit the device several times), and the ten
                               * launches per frame that would find nothing left (~2 us each) are gone.  Results do not depend on it: a late
                               * trip that does have work is rendered by the small grid, just more slowly. */
    float *snapshots;      /* gfpp_head_frame_persist_lp only: [N, 7, 5] f32 -- {weights_sum, depth, r, g, b} of a ray after max_steps .. max_steps + 6
                            * composited samples (only rays that get that far write it; gfpp_head_frame_resolve reads it) */
    uint32_t defer_resolve;   /* 1: gfpp_head_frame_persist_lp issues no resolve launch; the consumer of the ray records, gfpp_torso_frame_lp
                               * (16-bit / MFMA torso kernel only; gfpp_head_frame_finish and gfpp_torso_frame do NOT resolve), picks budget and
                               * snapshot per ray on the fly (it reads counters[128..] / gcounters and `snapshots`, with `resolve_max_steps`).  counters[k] then stay unset until gfpp_head_frame_resolve is called. */
    uint32_t resolve_max_steps; /* max_steps of the head pass, for the consumers' on-the-fly resolve */
    gfpp_clip_job *clip_job;  /* NULL, or (DEVICE pointer) the clip job this frame belongs to: gfpp_torso_frame_lp then also writes the frame as uint8
                               * into the job's output slot of lane `clip_lane` and advances that lane's cursor (= gfpp_clip_store_u8 fused) */
    uint32_t clip_lane;
    /* (ABI 6) */
    uint32_t clip_sub;        /* this frame is the clip_sub-th of its lane's frame GROUP: it takes job position cursor[clip_lane] + clip_sub */
    uint32_t clip_advance;    /* what the storing launch adds to the lane's cursor when it is done: 0 = `lanes` (one frame per graph launch); a group of K
                               * frames per launch sets 0xFFFFFFFF (no advance) on all but its last frame and K * lanes on the last */
    uint32_t n_frames;        /* gfpp_head_frame_persist_lp only.  0 / 1: one frame.  K in 2..4: this workspace describes K frames of N rays each whose arrays
                               * lie BEHIND EACH OTHER -- rays_o / rays_d [K N, 3], nears / fars [K N], ray_state [K N, 8], sample_t [K N, stride],
                               * sample_cnt [K N], snapshots [K N, 7, 5], counters [K, 192], frame_consts [K, frame_consts_stride] -- and ONE launch renders all of them: a
                               * workgroup pools the samples of its rays of all K frames, so the fixed costs of a launch (weight image into LDS, partly
                               * filled sample blocks of every local round, the launch's tail) are paid once per K frames.  Per sample and per ray nothing
                               * changes (same block evaluation, same compositing order): every frame is the bits of its own launch.  Each frame keeps its own
                               * histogram / counters and is resolved on its own (gfpp_head_frame_resolve or the consumer's on-the-fly resolve, with the
                               * frame's own gfpp_frame_ws).  Used by the clip renderer for small frames (256^2 rays: 0.117 ms per frame alone). */
    uint32_t frame_consts_stride; /* frame groups: floats between the folded constants of consecutive frames (0 = 256: a [K, 256] array) -- the clip renderer's
                                   * constants sit inside the frames' rows of driving signals */
    int32_t *timeouts;        /* optional [1] i32 that NO kernel of this library resets: a device-wide barrier of the multi-trip launch
                               * (gfpp_head_frame_trips_lp) that times out adds 1 -- unlike counters[127], which the next frame's begin kernel zeroes, so a
                               * time-out in the middle of a clip stays visible until the caller has looked (FramePipeline.check_barriers) */
    uint32_t row_rays;        /* (ABI 7) 0, or W: ray n is pixel (n / W, n % W) of its frame (gfpp_head_group_begin sets up exactly this order).  Known, and a
                               * multiple of 64 with one workgroup per CU on a multiple of 8 CUs, the persistent head launch gives every XCD its own image COLUMNS
                               * (8-ray tile column c goes to XCD c % 8, i.e. to the workgroups b with b % 8 == c % 8): the x-y cells of the position grid a
                               * workgroup gathers -- at the levels whose index drops z, all of them -- then come from 1/8 of the table rows, so that an XCD's 4 MiB
                               * L2 holds them.  A pure speed choice: which workgroup renders a ray never changes its bits. */
} gfpp_frame_ws;

/* Starts a frame (replaces renderer.py:302-350 = raymarching.cu:91-145 slab test + the torch.zeros/arange/clone state
 * setup, and the per-sample cond_feat.repeat / individual_code.repeat + cat of radnerf.py:115-136):
 * slab test for every ray (= near_far_from_aabb), zeroes the accumulators, rays_t = near, resets the trip
 * counters (counters[0] = N), and folds cond_feat [cond_dim] / ind_code [ind_dim] into the two per-frame bias vectors. */
int gfpp_head_frame_begin(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                          const float *cond_feat, const float *ind_code, gfpp_stream_t stream);

/* The constant-folding half of gfpp_head_frame_begin on its own (the per-sample cond_feat.repeat / individual_code.repeat + cat of
 * radnerf.py:115-136 become two bias vectors): pass cond_feat == NULL to gfpp_head_frame_begin and call this once cond_feat exists,
 * e.g. on a second stream next to the slab test and the pre-march, which do not depend on the conditioning networks. */
int gfpp_head_frame_fold(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *cond_feat, const float *ind_code,
                         gfpp_stream_t stream);

/* gfpp_head_frame_fold for `count` frames in one launch (same kernel, same bits): frame k's conditioning vector at cond_feats + k cond_stride (floats),
 * its 256 constants at frame_consts + 256 k; then gfpp_frame_ws.frame_consts of a frame may point at its row.  What is folded: the conditioning
 * columns of ambient_net's first layer and the individual-code columns of color_net's (radnerf.py:108-141: those inputs are the same for every sample of
 * a frame).  For callers that hold the driving signals of a whole clip (with gfpp_cond_feat_batch). */
int gfpp_head_frame_fold_batch(const gfpp_head_model *model, const float *cond_feats, uint32_t cond_stride, const float *ind_code, float *frame_consts,
                               uint32_t count, gfpp_stream_t stream);

/* Runs the whole march -> evaluate -> composite loop of renderer.py:354-384 (kernels raymarching.cu:827-929, :942-1029;
 * networks radnerf.py:108-141; encoders gridencoder.cu:87-196, shencoder.cu:28-68): `max_steps` fused trip launches, each of
 * which marches its alive rays (kernel_march_rays semantics), evaluates RADNeRF.forward on MFMA for the occupied samples
 * only, composites (kernel_composite_rays semantics) and compacts the survivors for the next trip. */
int gfpp_head_frame_march(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                          float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* The trip launches of the exact-fp32 mode over the frame's pre-marched sample lists (renderer.py:354-384 loop; per-sample arithmetic
 * radnerf.py:108-141 on exact-fp32 MFMA, compositing raymarching.cu:942-1029): call gfpp_head_frame_begin, gfpp_head_frame_premarch,
 * then this.  Same results as gfpp_head_frame_march sample for sample; wavefronts own their tiles (no workgroup barriers). */
int gfpp_head_frame_trips(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                          float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* Same loop as gfpp_head_frame_march (renderer.py:354-384, raymarching.cu:827-929, :942-1029, radnerf.py:108-141) with the five
 * wide layers on 16-bit MFMA operands (model->lp_weights), fp32 accumulation -- the precision class of the reference's own
 * inference path (inference/genefacepp_infer.py:433-486 renders under torch.autocast: nn.Linear in fp16).  March, grid interpolation, tanh / exp / sigmoid, SH and compositing are fp32 and
 * identical to the fp32 entry point, so sample positions and the trip schedule do not depend on the mode except through
 * sigma (rays whose transmittance crosses T_thresh).  Stated tolerance vs the fp32 oracle: PSNR >= 45 dB, max-abs <= 2e-2
 * (SURVEY.md 8c).  One 512-thread workgroup per CU keeps the weight image resident in LDS. */
int gfpp_head_frame_march_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                             float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* First stage of gfpp_head_frame_march_lp on its own: marches every ray of the frame once through the occupancy bitfield
 * (kernel_march_rays semantics, raymarching.cu:827-929) and stores the t of every occupied sample in ws->sample_t / sample_cnt.
 * Needs only gfpp_head_frame_begin's slab test.  Follow with gfpp_head_frame_trips_lp. */
int gfpp_head_frame_premarch(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                             float dt_gamma, uint32_t max_steps, gfpp_stream_t stream);

/* gfpp_head_frame_begin (without the fold) and gfpp_head_frame_premarch in ONE launch: near/far of every ray (near_far_from_aabb,
 * raymarching.cu:91-145), ray-state and counter reset, then the pre-march from that near (raymarching.cu:827-929) -- the rays are read
 * once.  Same results as the two calls.  The caller folds the conditioning with gfpp_head_frame_fold (any stream, before the trips). */
int gfpp_head_frame_begin_premarch(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                   float dt_gamma, uint32_t max_steps, gfpp_stream_t stream);

/* Second stage: the trip launches of gfpp_head_frame_march_lp, consuming the pre-marched lists (renderer.py:354-384 loop). */
int gfpp_head_frame_trips_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                             float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* The whole loop of renderer.py:354-384 for the 16-bit mode as ONE launch without any device-wide dependency (replaces the six-plus launches
 * of gfpp_head_frame_trips_lp; same per-sample arithmetic, results bit-identical to it).  What makes that possible: along a ray the
 * compositing recurrence (raymarching.cu:978-1022) is sequential and the marcher carries nothing but t (raymarching.cu:857), so the state of a
 * ray after its first k samples does not depend on how the loop cut those samples into trips; the trip schedule only decides (a) the total
 * step budget B = sum of n_step over the trips that run (a ray composites its first min(c, e + 1, B) samples, c = occupied samples it owns, e =
 * first sample whose pre-sample transmittance is < T_thresh) and (b) the n_alive sequence, and both are functions of the histogram of
 * m = min(c, e) over the rays: a ray is alive after the window ending at sample S  <=>  m >= S.  So every workgroup takes its own tiles of 32
 * rays (dealt out through a multiplicative permutation), keeps the weight image in LDS for the whole frame and loops LOCALLY: next samples of
 * its alive rays -> workgroup sample pool -> evaluate (the trip kernels' evaluate_block_lp) -> composite -> local compaction, with a local
 * n_step.  A ray that composites more than max_steps samples snapshots its state after samples max_steps .. max_steps + 6 (B lies in
 * [max_steps, max_steps + 7]); gfpp_head_frame_resolve then replays renderer.py:359-364,384 on the histogram, writes counters[k] (alive rays
 * at the start of trip k, as the trip launches would have left them) and gives every ray with more than B composited samples its snapshot.
 * Needs gfpp_head_frame_begin_premarch (or _begin + _premarch) and gfpp_head_frame_fold first; max_steps <= 24, N <= 2^22.
 * ws->gcounters != NULL (one ray tile of a frame shared between GPUs): the resolve step is NOT issued -- the caller sums counters[128..191] of
 * all tiles into gcounters[0..63] (ONE all-reduce per frame instead of one per trip) and then calls gfpp_head_frame_resolve. */
int gfpp_head_frame_persist_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                               float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* The same one-launch loop in the exact-fp32 parity mode (round 4): the fp32 trip kernels' own block evaluation (v_mfma_f32_32x32x2_f32 on the fp32 tables and the
 * fragment-ordered fp32 weights of gfpp_head_model) inside the persistent structure -- results bit-identical to gfpp_head_frame_trips, counters[k] as the trip
 * launches leave them (renderer.py:354-384; RADNeRF.forward radnerf.py:108-141).  One frame per launch (ws->n_frames <= 1); same preconditions, same resolve
 * step and the same ws->gcounters convention as gfpp_head_frame_persist_lp. */
int gfpp_head_frame_persist(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                            float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream);

/* Second step of gfpp_head_frame_persist_lp (issued by it unless ws->gcounters is set): replays the loop control of renderer.py:359-364,384
 * (n_step = clamp(N // n_alive, 1, 8), step += n_step, exit at max_steps or when nobody is alive) on the histogram -- ws->gcounters[0..63] if set
 * (the frame-wide sums, with ws->N_global rays), else ws->counters[128..191] -- which yields the budget B and counters[k]; then the snapshot
 * selection for rays that composited more than B samples (raymarching.cu:978-1022 stops them at the end of the last trip's window). */
int gfpp_head_frame_resolve(const gfpp_frame_ws *ws, uint32_t max_steps, gfpp_stream_t stream);

/* (ABI 6) The prologue and the resolve step of a frame GROUP (ws->n_frames = K frames of N = H * W rays behind each other in every array, see gfpp_frame_ws.n_frames),
 * each as ONE launch.  gfpp_head_group_begin = for every frame f: the rays of get_rays (modules/radnerfs/utils.py:283-364, the expressions of gfpp_get_rays) from the
 * frame's cam2world matrix at poses + f * pose_stride (floats; e.g. a field of the frame's row of driving signals), stored to rays_o / rays_d [K N, 3] for the head
 * launch, + what gfpp_head_frame_begin_premarch does (slab test = raymarching.cu:91-145, state / counter reset, pre-march = raymarching.cu:827-929 once per ray).
 * gfpp_head_group_resolve = gfpp_head_frame_resolve for every frame against its own histogram (counters [K, 192]).  Same bits as the per-frame entries. */
int gfpp_head_group_begin(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *poses, uint32_t pose_stride, float fx, float fy, float cx, float cy,
                          uint32_t H, uint32_t W, float *rays_o, float *rays_d, float dt_gamma, uint32_t max_steps, gfpp_stream_t stream);
int gfpp_head_group_resolve(const gfpp_frame_ws *ws, uint32_t max_steps, gfpp_stream_t stream);

/* Head-only epilogue (renderer.py:385-397): image = clamp(image + (1 - weights_sum) * bg, 0, 1),
 * depth = clamp(depth - near, 0) / (far - near).  bg_color [N,3] or NULL (then bg_scalar is used; reference default 1). */
int gfpp_head_frame_finish(const gfpp_frame_ws *ws, const float *bg_color, float bg_scalar, float *out_image, float *out_depth,
                           gfpp_stream_t stream);

/* Per-sample evaluation of the radiance field = RADNeRF.forward (modules/radnerfs/radnerf.py:108-141) on a caller-supplied
 * sample list, run by the SAME device code as the trip kernels (evaluate_block / evaluate_block_lp): this is the hook the per-sample
 * parity tests use (and what RADNeRF.forward()/density() call at inference).  The per-frame constants must have been folded into
 * ws->frame_consts first (gfpp_head_frame_fold: cond_feat and the individual code).  positions [M,3] in [-bound, bound] (outside
 * => zero grid features, grid.py:152 + gridencoder.cu:110-135), directions [M,3] unit vectors.
 * Out: sigma [M] (= exp(h0), utils.py:36-49 forward; NOT multiplied by density_scale, like RADNeRF.forward), color [M,3] (sigmoid), ambient [M, amb_D] (tanh; may be NULL).
 * `_lp`: 16-bit MFMA operands per model->lp_dtype (fp32 accumulate); the other entry is the exact-fp32 path. */
int gfpp_head_eval_samples(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *positions, const float *directions,
                           uint32_t M, float *sigma, float *color, float *ambient, gfpp_stream_t stream);
int gfpp_head_eval_samples_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *positions, const float *directions,
                              uint32_t M, float *sigma, float *color, float *ambient, gfpp_stream_t stream);

/* Torso field + final compositing.  Weight matrices are stored K-MAJOR ([in][out], i.e. nn.Linear.weight transposed) so
 * that one input feeds a contiguous row of outputs; the *_c blocks (columns multiplying per-frame constants) stay row-major
 * [out][const_dim] and are folded into biases on the device at the start of every launch.
 * Input column order of torso_deform_net (and, after the 32 grid features, of torso_canonicial_net):
 *   variant 0, RADNeRFTorso        (radnerf_torso.py:51-84):   [freq(x,10) 42 | freq(pose,4) 54 | code | head-aware 16]
 *   variant 1, RADNeRFTorsowithSR  (radnerf_torso_sr.py:75-114): [freq(x,10) 42 | code | freq(chin lm,4) 126 | head-aware 16] */
typedef struct gfpp_torso_model {
    const float *density_grid;  /* [G*G] f32, density_grid_torso */
    uint32_t grid_size;         /* G = 128 */
    float density_thresh;       /* min(density_thresh_torso, mean_density_torso) -- 0 for a loaded checkpoint */
    float torso_shrink;         /* 0.8 */
    uint32_t variant;           /* 0 pose-conditioned, 1 landmark-conditioned */
    uint32_t code_dim;          /* torso_individual_embedding_dim (8) */
    uint32_t const_dim;         /* 54 + code_dim or 126 + code_dim */
    uint32_t head_aware;        /* hparams['torso_head_aware'] */
    gfpp_grid_desc grid;        /* D = 2, tiled, linear */
    const float *def_w0_x;      /* [42][64]  */
    const float *def_w0_c;      /* [64][const_dim] row-major */
    const float *def_w0_h;      /* [16][64] or NULL */
    const float *def_w1;        /* [64][64] */
    const float *def_w2;        /* [64][2] */
    const float *can_w0_g;      /* [32][32] grid-feature columns */
    const float *can_w0_x;      /* [42][32] */
    const float *can_w0_c;      /* [32][const_dim] row-major */
    const float *can_w0_h;      /* [16][32] or NULL */
    const float *can_w1;        /* [32][32] */
    const float *can_w2;        /* [32][4] */
    const float *ha_w0, *ha_b0; /* head_color_weights_encoder: [4][16],[16] */
    const float *ha_w1, *ha_b1; /* [16][32],[32] */
    const float *ha_w2, *ha_b2; /* [32][16],[16] */
    /* 16-bit operand image for gfpp_torso_frame_lp (NULL if not built): 28 fragments of [tile][lane] x 8 halves, K = 16 MFMA steps,
     * F[step][t][lane][e] = W[32*t + (lane & 31)][col(step, lane >> 5, e)] (zero where the column is padding), layers in this order:
     *   head-aware encoder  ha0 1 step x 1 tile (4 -> 16, inputs in elements 0..3 of half 0), ha1 1 x 1 (16 -> 32), ha2 2 x 1 (32 -> 16);
     *   torso_deform_net.0  4 x 2: steps 0-2 = frequency features of the pixel, slot 16*s + 8*h + e (42 used), step 3 = head-aware 16;
     *   torso_deform_net.1  4 x 2 over the 64 activations;  torso_canonicial_net.0  6 x 1: steps 0-1 = 32 grid features (half-wave h holds
     *   levels h, h+2, ..), steps 2-4 = frequency features, step 5 = head-aware;  torso_canonicial_net.1  2 x 1.
     * Activation columns follow the accumulator order act(s) of gfpp_head_model.lp_weights.  lp_skinny: torso_deform_net.2 as
     * [2 halves][2 rows][32] and torso_canonicial_net.2 as [2][4][16], 16-bit, in the operand order of the preceding activations. */
    const void *lp_weights;
    const void *lp_skinny;
    int32_t lp_dtype;
} gfpp_torso_model;

/* Replaces the torso pass + epilogue of RADNeRFTorso.render / RADNeRFTorsowithSR.render (radnerf_torso.py:156-197,
 * radnerf_torso_sr.py:186-231) on top of a finished head pass (ws->image / weights_sum / depth / nears / fars):
 * occupancy mask = F.grid_sample(density_grid_torso)(bg_coords) > thresh; forward_torso on the masked pixels;
 * torso_bg = rgb*alpha + bg*(1-alpha); image = clamp(head + (1-weights_sum)*torso_bg, 0, 1); depth normalised.
 * cond_in = poses [6] (variant 0) or lm68 [136] (variant 1); code [code_dim]; use_head != 0 feeds (head rgb, alpha) to the
 * head-aware encoder (else zeros, like passing image=None).  Outputs: out_image, torso_bg [N,3]; out_depth, torso_alpha [N];
 * deform [N,2] (zero outside the mask); mask [N] u8. */
int gfpp_torso_frame(const gfpp_torso_model *model, const gfpp_frame_ws *ws, const float *bg_coords, const float *cond_in,
                     const float *code, const float *bg_color, float bg_scalar, uint32_t use_head, float *out_image,
                     float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask, gfpp_stream_t stream);

/* gfpp_torso_frame with the torso MLPs (radnerf_torso.py:51-84, radnerf_torso_sr.py:75-114) on 16-bit MFMA operands, fp32 accumulation;
 * same arguments and outputs.  Occupancy test, frequency features, grid interpolation, sigmoid, compositing and depth stay fp32. */
int gfpp_torso_frame_lp(const gfpp_torso_model *model, const gfpp_frame_ws *ws, const float *bg_coords, const float *cond_in,
                        const float *code, const float *bg_color, float bg_scalar, uint32_t use_head, float *out_image,
                        float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask, gfpp_stream_t stream);

/* (ABI 7) The torso passes of a frame GROUP (radnerf_torso.py:156-197 / radnerf_torso_sr.py:186-231 for K consecutive frames of the caller's loop,
 * genefacepp_infer.py:460-469) as two launches whatever K is.
 * gfpp_torso_mask: mask[n] = occupancy grid sampled at bg_coords[n] > density_thresh (radnerf_torso.py:166-169) -- constants of the model and the resolution, so the
 *   caller computes it ONCE and keeps the ascending list of the masked pixels (`masked_idx`, a stream compaction of `mask`).
 * gfpp_torso_fold_batch: the per-frame constant columns of torso_deform_net.0 / torso_canonicial_net.0 (frequency-encoded pose or chin landmarks + individual code)
 *   folded into bias vectors, one workgroup per frame -- the arithmetic of gfpp_torso_frame_lp's prologue: frame f reads cond_in + f * cond_stride (poses [6] or
 *   lm68 [136]) and writes folded[f] = bdef [64] | bcan [32].  It does not depend on the head pass: issue it ahead of it.
 * gfpp_torso_group_lp: `ws` is a frame-group record (gfpp_frame_ws.n_frames = K, every per-ray array the stack of the K frames) whose head pass was
 *   gfpp_head_frame_persist_lp WITHOUT a resolve step.  Launch 1: the torso field at the listed pixels of all K frames, 32 per pass, dealt out to persistent
 *   wavefronts (no occupancy test, no compaction: every pass is full and known in advance).  Launch 2, one thread per pixel: step budget from each frame's histogram
 *   and snapshot selection per ray (= gfpp_head_group_resolve; it also writes counters[f][k], the alive counts of the reference's loop), torso over background,
 *   head over torso, depth, and -- when ws->clip_job is set -- the uint8 store of frame f into job position cursor[clip_lane] + f and the cursor's advance by
 *   ws->clip_advance (= gfpp_clip_store_u8_at).  Outputs are stacks over the frames: out_image / torso_bg [K N, 3], out_depth / torso_alpha [K N], deform [K N, 2],
 *   mask [K N].  Every value is the bits of the per-frame entry (same per-pixel code).  16-bit weight images only (model->lp_dtype GFPP_F16 / GFPP_BF16). */
int gfpp_torso_mask(const gfpp_torso_model *model, const float *bg_coords, uint32_t N, uint8_t *mask, gfpp_stream_t stream);
int gfpp_torso_fold_batch(const gfpp_torso_model *model, const float *cond_in, uint32_t cond_stride, const float *code, uint32_t frames, float *folded,
                          gfpp_stream_t stream);
int gfpp_torso_group_lp(const gfpp_torso_model *model, const gfpp_frame_ws *ws, const float *bg_coords, const float *folded, const float *code,
                        const uint8_t *mask_static, const int32_t *masked_idx, uint32_t n_masked, const float *bg_color, float bg_scalar, uint32_t use_head,
                        uint32_t max_steps, float *out_image, float *out_depth, float *torso_alpha, float *torso_bg, float *deform, uint8_t *mask,
                        gfpp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Section A.3 -- training-side entry points of _raymarching_face and _gridencoder.  Same conventions as Section A:
 * caller-allocated outputs, written in place; fp32.
 * ---------------------------------------------------------------------------------------------------------- */

/* replaces march_rays_train (raymarching.h:14; kernel raymarching.cu:352-518): per ray, count the occupied samples (<= max_steps), reserve
 * a range with atomicAdd(counter[0], count) / a row with atomicAdd(counter[1], 1), write rays [N,3] = (ray, first point, count) and, if the
 * range fits below M, the samples xyzs/dirs [M,3], deltas [M,2] = (dt, t + dt).  noises [N]: perturbation of the start, 0 = none. */
int gfpp_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                          float *deltas, int32_t *rays, int32_t *counter, const float *noises, gfpp_stream_t stream);

/* replaces march_rays_train_backward (raymarching.h:15; raymarching.cu:535-583): grad_rays_o/d [N,3] += sums over each ray's samples. */
int gfpp_march_rays_train_backward(const float *grad_xyzs, const float *grad_dirs, const int32_t *rays, const float *deltas, uint32_t N, uint32_t M,
                                   float *grad_rays_o, float *grad_rays_d, gfpp_stream_t stream);

/* replaces composite_rays_train_forward (raymarching.h:16; raymarching.cu:603-688): T *= 1 - alpha, stop when T < T_thresh AFTER the update. */
int gfpp_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *ambient, const float *deltas, const int32_t *rays, uint32_t M,
                                      uint32_t N, float T_thresh, float *weights_sum, float *ambient_sum, float *depth, float *image, gfpp_stream_t stream);

/* replaces composite_rays_train_backward (raymarching.h:17; raymarching.cu:711-810). */
int gfpp_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_ambient_sum, const float *grad_image, const float *sigmas,
                                       const float *rgbs, const float *ambient, const float *deltas, const int32_t *rays, const float *weights_sum,
                                       const float *ambient_sum, const float *image, uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas,
                                       float *grad_rgbs, float *grad_ambient, gfpp_stream_t stream);

/* replaces morton3D_dilation (raymarching.h:12; raymarching.cu:304-335): 6-neighbour max pool of a Morton-ordered [C, H^3] grid. */
int gfpp_morton3D_dilation(const float *grid, uint32_t C, uint32_t H, float *grid_dilation, gfpp_stream_t stream);

/* replaces sph_from_ray (raymarching.h:8; raymarching.cu:162-199): far intersection with a sphere -> (theta, phi) in [-1,1]^2, coords [N,2]. */
int gfpp_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, gfpp_stream_t stream);

/* the dy_dx half of grid_encode_forward (gridencoder.h:12; gridencoder.cu:198-243): d features / d inputs, [B, L, D, C]; D in {2,3}, C in {1,2,4}. */
int gfpp_grid_encode_dydx(const float *inputs, const float *embeddings, const int32_t *offsets, float *dy_dx, uint32_t B, uint32_t D, uint32_t C,
                          uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream);

/* replaces grid_encode_backward (gridencoder.h:13; gridencoder.cu:247-368): grad [L,B,C] -> grad_embeddings (+=, atomics) and, when dy_dx and
 * grad_inputs are given (both or neither), grad_inputs [B,D]. */
int gfpp_grid_encode_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets, float *grad_embeddings, uint32_t B,
                              uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp, gfpp_stream_t stream);

/* Weight gradient of a bias-free Linear layer over a training batch: grad_weight [O, I] fp32 = grad_out^T [O, M] x input [M, I] -- what autograd's
 * backward of the reference's `MLP` layers (cond_encoder.py:183-202: nn.Linear(..., bias=False), used for ambient_net / sigma_net / color_net,
 * radnerf.py:60-100) computes with a BLAS call per layer.  M = the samples of the step: the output is small and the reduction very long, so the M
 * rows are cut into <= 512 slices, one workgroup each with the whole O x I output in MFMA accumulators (fp32 accumulation; dtype GFPP_F16: half
 * operands as under `amp: true`, GFPP_F32: exact-fp32 MFMA), and the slices are summed.  partial: scratch [512, O, I] fp32.  O <= 256, I <= 160
 * (GFPP_EUNSUPPORTED beyond: the caller keeps its BLAS call).  grad_weight is overwritten.  grad_out and input: 16-byte aligned, read as 16-byte vectors;
 * nothing is read past their last element (the vector that straddles the end is assembled element by element). */
int gfpp_linear_weight_grad(const void *grad_out, const void *input, uint32_t M, uint32_t O, uint32_t I, int dtype, float *partial, float *grad_weight,
                            gfpp_stream_t stream);
/* floats of the `partial` scratch gfpp_linear_weight_grad needs for an O x I layer (cond_encoder.py:183-202's layers): the slice count is the library's, not the
 * binding's, to know; 0 for a shape the kernels do not cover. */
unsigned long long gfpp_linear_weight_grad_scratch_floats(uint32_t O, uint32_t I);

/* A whole `MLP` (cond_encoder.py:183-202: bias-free Linear layers, ReLU between them; ambient_net / sigma_net / color_net, radnerf.py:60-100) over a
 * training batch under `amp: true` as ONE forward and ONE backward launch -- what autograd runs as a BLAS GEMM + relu + cast kernels per layer and
 * direction (radnerf.py:108-141 under torch.autocast, tasks/radnerfs/radnerf.py:101-176).  Half operands, fp32 accumulation, every layer output rounded
 * to half (autocast's F.linear), relu's backward on the saved rounded activation.  All matrices are plain row-major half, 16-byte aligned, with the
 * feature axes zero-padded: input [M, in_pad], hidden [n_layers - 1][M, 128], output [M, out_pad].
 *   hidden 128; n_layers 2 or 3; in_pad 64 / 96 / 160; out_pad 32 / 160 (GFPP_EUNSUPPORTED otherwise: the caller keeps the per-layer path).
 *   gfpp_mlp_train_pack      the forward and the transposed (backward) weight images from the fp32 parameters (weights[l]: device pointer to
 *                            nn.Linear.weight [out_l, in_l]; `weights` itself is a HOST array of n_layers pointers), one launch;
 *                            gfpp_mlp_train_image_bytes: the size of either image.
 *   gfpp_mlp_train_forward   x -> hidden_acts (relu(hidden), kept for the backward pass) and out.
 *   gfpp_mlp_train_backward  grad_out (+ hidden_acts) -> grad_hidden [n_layers - 1][M, 128] (G_l = (W_{l+1}^T G_{l+1}) where the activation passed: the
 *                            `grad_out` operand of gfpp_linear_weight_grad for layer l) and grad_x [M, in_pad] (may be null).
 * The weight gradients are gfpp_linear_weight_grad(G_l or grad_out, act_{l-1} or x). */
int gfpp_mlp_train_pack(const float *const *weights, uint32_t n_layers, uint32_t in_features, uint32_t hidden, uint32_t out_features, uint32_t in_pad,
                        uint32_t out_pad, void *fwd_image, void *bwd_image, gfpp_stream_t stream);
uint32_t gfpp_mlp_train_image_bytes(uint32_t n_layers, uint32_t in_pad, uint32_t out_pad, int backward);
int gfpp_mlp_train_forward(const void *x, const void *fwd_image, uint32_t M, uint32_t in_pad, uint32_t hidden, uint32_t n_layers, uint32_t out_pad,
                           void *hidden_acts, void *out, gfpp_stream_t stream);
int gfpp_mlp_train_backward(const void *grad_out, const void *hidden_acts, const void *bwd_image, uint32_t M, uint32_t in_pad, uint32_t hidden,
                            uint32_t n_layers, uint32_t out_pad, void *grad_hidden, void *grad_x, gfpp_stream_t stream);

/* The input gradient of the lookup without a materialised dy_dx (gridencoder.cu:198-243 kernel_grid's dy_dx branch + gridencoder.cu:342-368
 * kernel_input_backward in one pass): grad [L,B,2] (fp32 or half, grad_dtype) -> grad_inputs [B,D] fp32, the derivative recomputed from the fp32 table.
 * The reference keeps dy_dx [B, L*D*C] from the forward pass (116 MB per May step for the ambient grid); callers that hold one can still pass it to
 * gfpp_grid_encode_backward.  level_dim 2, input_dim 2 or 3 (GFPP_EUNSUPPORTED otherwise). */
int gfpp_grid_encode_input_backward(const void *grad, int grad_dtype, const float *inputs, const float *embeddings, const int32_t *offsets,
                                    float *grad_inputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                    int align_corners, uint32_t interp, gfpp_stream_t stream);

/* grid_encode_backward with a HALF grad (gridencoder.h:13; gridencoder.cu:247-368 instantiated for at::Half -- what `amp: true`, the reference's training
 * configuration egs/datasets/May/lm3d_radnerf.yaml:5, runs: grid.py:43-44 casts the table to half, so features and their gradient are half).  grad:
 * [L,B,2] half.  Accumulation is fp32 (the reference adds __half2 atomics into a half gradient, gridencoder.cu:306-318, which autograd then casts to the
 * fp32 parameter; here the fp32 gradient comes out directly): same kernels and scratch as gfpp_grid_encode_backward_xcd, which see below.  dy_dx (fp32,
 * gfpp_grid_encode_dydx) and grad_inputs [B,D] fp32 go together.  level_dim 2 only (GFPP_EUNSUPPORTED otherwise; the reference forces fp32 for odd
 * level_dim too). */
int gfpp_grid_encode_backward_f16(const void *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t rows_total,
                                  float *xcd_copies, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                  float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream, void *bins,
                                  unsigned long long bins_bytes);

/* The same gradient (gridencoder.cu:247-368) without a device atomic per corner (no reference counterpart).  `xcd_copies` is caller-provided scratch of
 * 8 x rows_total x C + 64 floats (cleared by the call): eight private copies of the table gradient that are summed into grad_embeddings (+=) at the
 * end, and the levels' gradient maxima behind them.  A workgroup owns one RANGE of 16 384 values of a level's table and one eighth of the points; it
 * recomputes the corners of its points, adds those that fall into its range into LDS accumulators and stores the range into the eighth's copy -- a
 * 2^16-row level is located eight times, which is ~10x cheaper than the 67 M device atomics of a May grid were (4.6 ms per call, 45 % of a training step
 * in round 2).  The LDS accumulators are 64-bit fixed point scaled by the level's largest |grad| (float LDS atomics run ~50x slower than integer ones
 * on gfx950): a contribution is kept down to 2^-36 of that maximum; a non-finite grad makes its level's gradient NaN.  gfpp_tuning.grid_bwd_scatter selects the
 * round-2 path (LDS-privatised coarse levels + XCD-private device atomics).  rows_total = embeddings.shape[0].
 * `bins` (ABI 8; may be NULL): gfpp_grid_backward_bins_bytes(rows_total, C, L, B) bytes of scratch.  With it one pass per level first puts every point on the LIST
 * of each range that one of its corners falls into (a point's eight corners touch two or three of a fine level's eight ranges), and a range's workgroups walk
 * their list instead of all points: a 2^16-row level's cells are located ~2.3 times instead of eight.  Same accumulators, same sums (integer adds: the order
 * the lists come out in does not matter).  gfpp_tuning.grid_bwd_bins = 0 ignores the scratch (the A/B partner). */
int gfpp_grid_encode_backward_xcd(const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t rows_total,
                                  float *xcd_copies, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                                  float *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, gfpp_stream_t stream, void *bins,
                                  unsigned long long bins_bytes);
/* bytes of the `bins` scratch of the two calls above (gridencoder.cu:247-339 has no counterpart: its scatter needs none) */
unsigned long long gfpp_grid_backward_bins_bytes(uint32_t rows_total, uint32_t C, uint32_t L, uint32_t B);

/* replaces grad_total_variation (gridencoder.h:15; gridencoder.cu:505-609): TV gradient of the cells visited by `inputs`, grad (+=, atomics). */
int gfpp_grad_total_variation(const float *inputs, const float *embeddings, float *grad, const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, gfpp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Super-resolution stage of the *_sr models: replaces Superresolution.forward (modules/radnerfs/radnerf_sr.py:30-43 =
 * SynthesisBlockNoUp superresolution.py:159-258 + SynthesisBlock networks_stylegan2.py:375-478, layers :286-371, modulated_conv2d
 * :37-94, conv2d_resample.py:47-147, upfirdn2d.py:330-355, bias_act.py:95-125).  ws = ones there, so the styles are constants: the
 * caller folds modulation / demodulation (and, for block1.conv0, transposed convolution + FIR) into the weights, packed as f16 MFMA
 * fragments  F[pass][tap 9][step Cin/16][tile NT][lane][e] = W_eff[pass*32*NT + 32*t + (lane & 31)][16*s + 8*(lane >> 5) + e][tap / 3][tap % 3].
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct gfpp_sr_model {
    const void *w_first;   /* block0.conv0  3 -> 128: [2 steps][4 tiles][64][8] f16, k = 3*tap + channel (27 of 32 used) */
    const void *w_b0c1;    /* block0.conv1  128 -> 128: NT = 4, 1 pass */
    const void *w_up;      /* block1.conv0  128 -> 4 phases x 64 (output pixel (2y + py, 2x + px), channel = 64*(2*py + px) + o): NT = 4, 2 passes */
    const void *w_b1c1;    /* block1.conv1  64 -> 64: NT = 2, 1 pass */
    const float *bias[4];  /* per layer, in the order above: [128], [128], [64], [64] */
    float noise_strength[4];
    const float *rgb0_w, *rgb0_b; /* block0.torgb: modulated 1x1 weights [128][3], bias [3] */
    const float *rgb1_w, *rgb1_b; /* block1.torgb: [64][3], [3] */
    float fir[4];          /* 1-D taps of the (separable) resample filter times the per-axis gain 2: [1,3,3,1]/8 * 2 */
    float conv_clamp;      /* 256 */
    /* ABI 8: block1.conv0 in POLYPHASE form (NULL: only the composed `w_up` exists): the modulated / demodulated 3 x 3 weights themselves, [2 output-channel halves]
     * [2 input-channel halves][9 taps][4 K steps][64 lanes][8] f16 with lane (j, h) of step s holding W[32 nt + j][64 ks + 16 s + 8 h + e][ky][kx], taps in the
     * order (ky, kx) = (0,0) (0,1) (1,0) (1,1) | (0,2) (1,2) | (2,0) (2,1) | (2,2) -- grouped by the shift of the input they multiply.  The layer then runs as
     * transposed convolution (9 tap products per pixel instead of the composed form's 36) + the FIR as a second small GEMM (gfpp_tuning.sr_up_poly). */
    const void *w_up_poly;
    /* ... and the FIR as the B operand of that GEMM: [2 row-tap pairs][5 K steps][64 lanes][8] f16.  A "run" is two consecutive T rows of one row phase (2 x 36
     * entries [px][mx], mx = 0 the patch's halo column) + 8 entries of padding; lane (j, h) of step s holds the coefficients of run entries 16 s + 8 h .. + 7 for
     * output column j of the patch: (pair 0: fir[1] for the run's first row, fir[3] for its second; pair 1: fir[0], fir[2]) x (fir[t] with t = 2 (mx - 1) + px -
     * (j - 1) when 0 <= t < 4, else 0); padding entries 0.  With fir = [1,3,3,1]/4 every product is exact in f16. */
    const void *up_fir_g;
} gfpp_sr_model;

typedef struct gfpp_sr_ws { /* caller-allocated device workspace */
    void *x0;      /* [256][256][128] f16 */
    void *x1;      /* [256][256][128] f16 */
    void *x2;      /* [512][512][64]  f16 */
    float *img256; /* [256][256][3]   f32 */
    /* noise_mode 'random' generated inside the kernels (networks_stylegan2.py:329-331 draws a fresh unit-normal field per layer and frame with
     * torch.randn): rng_state != NULL and noise == NULL -> every output pixel of every layer gets normal(Philox4x32-10(key = rng_seed ^ rng_state[2],
     * counter = (pixel, layer, frame))), Box-Muller; frame = rng_state[0], which the last launch of the call increments (rng_state[1] is its ticket
     * word; both zero-initialised by the caller, one triple per stream that runs this workspace).  rng_state[2] (ABI 6) lives in DEVICE memory so that a
     * caller can re-seed a workspace whose launches are frozen in a captured graph (the `rng_seed` argument is baked into the graph's kernel arguments). */
    uint64_t *rng_state; /* [3] u64, or NULL */
    uint64_t rng_seed;
    uint32_t clamp01;    /* 1: rgb_out = clamp(result, 0, 1) (the caller's `.clamp(0, 1)` of radnerf_torso_sr.py:221,231 folded in) */
    /* ABI 8: the clip job this forward belongs to (DEVICE pointer, or NULL).  The last layer's epilogue then writes the frame as uint8 -- `(x * 255.).int()` of
     * genefacepp_infer.py:468, clamped -- straight into the job's output slot of position cursor[clip_lane] + clip_sub instead of writing rgb_out as fp32, and the
     * launch's last workgroup advances the lane's cursor by clip_advance (0: by job->lanes; 0xFFFFFFFF: not at all -- the earlier frames of a group): no
     * gfpp_clip_store_u8 launch behind the frame.  Needs the resident last layer (gfpp_tuning.sr_final_resident). */
    gfpp_clip_job *clip_job;
    uint32_t clip_lane, clip_sub, clip_advance;
    uint64_t *up_prof;   /* NULL (production), or [704 workgroups][8] u64: the polyphase up-sampling launch runs its profiling instantiation and leaves thread 0's shader-clock
                          * stamps at its phase boundaries there (tools/sr_up_phases.py) */
} gfpp_sr_ws;

/* replaces Superresolution.forward (radnerf_sr.py:30-43).  rgb_in [256][256][3] f32 (NHWC, values in [0,1]) -> rgb_out [512][512][3] f32.  noise: 4 device pointers to
 * the per-layer noise fields [256^2], [256^2], [512^2], [512^2] f32 (noise_const, or caller-drawn unit normals), or NULL: noise_mode 'none', or -- with
 * ws->rng_state set -- noise_mode 'random' drawn inside the kernels. */
int gfpp_sr_forward(const gfpp_sr_model *model, const gfpp_sr_ws *ws, const float *rgb_in, const float *const noise[4], float *rgb_out,
                    gfpp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GFPP_RADNERF_H */
