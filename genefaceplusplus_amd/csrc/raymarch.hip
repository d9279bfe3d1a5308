// raymarch.hip -- stand-alone ray kernels behind the reference's `_raymarching_face` extension API
// (near_far_from_aabb, morton3D(_invert), packbits, march_rays, composite_rays) plus on-device ray generation.
// One thread per ray; 256-thread workgroups (4 wavefronts) so that a 262 144-ray frame is 1024 workgroups.
#include <stdarg.h>
#include <string.h>

#include "march_device.h"

namespace gfpp {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                    const float *__restrict__ aabb, uint32_t N, float min_near,
                                                    float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float *o = rays_o + 3ull * n, *d = rays_d + 3ull * n;
    const RayBox rb = ray_box(o[0], o[1], o[2], d[0], d[1], d[2], aabb, min_near);
    nears[n] = rb.near;
    fars[n] = rb.far;
}

__global__ __launch_bounds__(kBlock) void k_morton(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[3ull * n], (uint32_t)coords[3ull * n + 1], (uint32_t)coords[3ull * n + 2]);
}

__global__ __launch_bounds__(kBlock) void k_morton_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const uint32_t code = (uint32_t)indices[n];
    coords[3ull * n] = (int32_t)compact3(code);
    coords[3ull * n + 1] = (int32_t)compact3(code >> 1);
    coords[3ull * n + 2] = (int32_t)compact3(code >> 2);
}

// One thread packs one output byte from 8 consecutive floats (two 16-byte loads).
__global__ __launch_bounds__(kBlock) void k_packbits(const float4 *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4 a = grid[2ull * n], b = grid[2ull * n + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

__global__ __launch_bounds__(kBlock) void k_march(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                 const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                 const float *__restrict__ rays_d, MarchParams p, const uint8_t *__restrict__ bitfield,
                                                 const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                                                 float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t ray = (uint32_t)rays_alive[n];
    const float *o = rays_o + 3ull * ray, *d = rays_d + 3ull * ray;
    const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
    float t = rays_t[ray];
    t = fmaf(clampf(t * p.dt_gamma, p.dt_min, p.dt_max), noises[n], t);
    float *px = xyzs + 3ull * n * n_step, *pd = dirs + 3ull * n * n_step, *pl = deltas + 2ull * n * n_step;
    march_one_ray(ox, oy, oz, dx, dy, dz, t, fars[ray], n_step, bitfield, p, [&](uint32_t s, const Sample &smp) {
        px[3 * s] = smp.x; px[3 * s + 1] = smp.y; px[3 * s + 2] = smp.z;
        pd[3 * s] = dx; pd[3 * s + 1] = dy; pd[3 * s + 2] = dz;
        pl[2 * s] = smp.dt; pl[2 * s + 1] = smp.t_end;
    });
}

__global__ __launch_bounds__(kBlock) void k_composite(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                                                     float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                     const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                     float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t ray = (uint32_t)rays_alive[n];
    const float *sg = sigmas + (size_t)n * n_step, *cl = rgbs + 3ull * n * n_step, *dl = deltas + 2ull * n * n_step;
    RayAccum a{weights_sum[ray], depth[ray], image[3ull * ray], image[3ull * ray + 1], image[3ull * ray + 2]};
    float t = rays_t[ray];
    uint32_t s = 0;
    for (; s < n_step; ++s) {
        const float dt = dl[2 * s];
        if (dt == 0.0f) break;  // never-written slot: the ray ran out of samples
        t = dl[2 * s + 1];
        if (composite_sample(a, sg[s], dt, t, cl[3 * s], cl[3 * s + 1], cl[3 * s + 2], T_thresh)) break;
    }
    if (s < n_step) rays_alive[n] = -1;
    else rays_t[ray] = t;
    weights_sum[ray] = a.wsum;
    depth[ray] = a.depth;
    image[3ull * ray] = a.r; image[3ull * ray + 1] = a.g; image[3ull * ray + 2] = a.b;
}

__global__ __launch_bounds__(kBlock) void k_get_rays(const float *__restrict__ pose, float fx, float fy, float cx, float cy, uint32_t H,
                                                    uint32_t W, float *__restrict__ rays_o, float *__restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= H * W) return;
    const uint32_t h = n / W, w = n - h * W;
    const float xs = ((float)w + 0.5f - cx) / fx;
    const float ys = ((float)h + 0.5f - cy) / fy;
    const float norm = sqrtf(fmaf(xs, xs, fmaf(ys, ys, 1.0f)));
    const float ux = xs / norm, uy = ys / norm, uz = 1.0f / norm;
    // rays_d = dir @ R^T  ==  R @ dir
    for (int r = 0; r < 3; ++r) {
        rays_d[3ull * n + r] = fmaf(pose[4 * r + 2], uz, fmaf(pose[4 * r + 1], uy, pose[4 * r] * ux));
        rays_o[3ull * n + r] = pose[4 * r + 3];
    }
}

// the same for a list of pixel indices (training: random / patch / rect sampling, utils.py:310-343); inds are h * W + w, int64 like torch.randint's
__global__ __launch_bounds__(kBlock) void k_get_rays_at(const float *__restrict__ pose, float fx, float fy, float cx, float cy, uint32_t W,
                                                       const long long *__restrict__ inds, uint32_t n_rays, float *__restrict__ rays_o,
                                                       float *__restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_rays) return;
    const uint32_t pix = (uint32_t)inds[n];
    const uint32_t h = pix / W, w = pix - h * W;
    const float xs = ((float)w + 0.5f - cx) / fx;
    const float ys = ((float)h + 0.5f - cy) / fy;
    const float norm = sqrtf(fmaf(xs, xs, fmaf(ys, ys, 1.0f)));
    const float ux = xs / norm, uy = ys / norm, uz = 1.0f / norm;
    for (int r = 0; r < 3; ++r) {
        rays_d[3ull * n + r] = fmaf(pose[4 * r + 2], uz, fmaf(pose[4 * r + 1], uy, pose[4 * r] * ux));
        rays_o[3ull * n + r] = pose[4 * r + 3];
    }
}

// float -> uint8 with truncation (x * 255 then int cast), 4 values per thread
__global__ __launch_bounds__(kBlock) void k_rgb_to_u8(const float *__restrict__ rgb, size_t n, uint8_t *__restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(rgb + i);
        uchar4 o;
        o.x = (uint8_t)clampf(v.x * 255.0f, 0.0f, 255.0f); o.y = (uint8_t)clampf(v.y * 255.0f, 0.0f, 255.0f);
        o.z = (uint8_t)clampf(v.z * 255.0f, 0.0f, 255.0f); o.w = (uint8_t)clampf(v.w * 255.0f, 0.0f, 255.0f);
        *reinterpret_cast<uchar4 *>(out + i) = o;
    } else {
        for (size_t k = i; k < n; ++k) out[k] = (uint8_t)clampf(rgb[k] * 255.0f, 0.0f, 255.0f);
    }
}

#define GFPP_REQUIRE_EARLY(cond, what)                            \
    do {                                                          \
        if (!(cond)) {                                            \
            gfpp::set_error("%s: invalid argument (%s)", what, #cond); \
            return GFPP_EINVAL;                                   \
        }                                                         \
    } while (0)

// ---- clip job: a frame's graph fetches its inputs and stores its output by a device-side cursor ---------------------------------------------
__global__ __launch_bounds__(kBlock) void k_clip_fetch(const gfpp_clip_job *__restrict__ job, uint32_t lane, uint32_t sub, float *__restrict__ static_in,
                                                       uint32_t row_floats) {
    // (blockIdx.y: the frames of a group, one row of static_in each)
    const uint32_t pos = job->cursor[lane] + sub + blockIdx.y;
    if (pos >= job->n) return;
    const float *row = job->packed + (size_t)job->order[pos] * job->row_floats;
    float *dst = static_in + (size_t)blockIdx.y * row_floats;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < row_floats; i += gridDim.x * kBlock) dst[i] = row[i];
}

__global__ __launch_bounds__(kBlock) void k_clip_store_u8(gfpp_clip_job *__restrict__ job, uint32_t lane, uint32_t sub, uint32_t advance,
                                                          const float *__restrict__ rgb, size_t n) {
    // a grid-stride loop over a FEW workgroups: the launch ends with one ticket per workgroup on the same word, and 768 of those (one float4 per thread)
    // took longer than the 3.9 MB they guard (9.8 us per launch in the trace)
    const uint32_t pos = job->cursor[lane] + sub;
    if (pos < job->n) {
        uint8_t *out = job->out + (size_t)(pos % job->ring_frames) * job->frame_bytes;
        // whole float4 / uchar4 quads where the slot is 4-byte aligned (every even frame size); a scalar tail / fallback otherwise (frames of odd H x W)
        const size_t quads = (job->frame_bytes & 3ull) == 0ull ? n / 4 : 0;
        for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += (size_t)gridDim.x * kBlock) {
            const float4 v = *reinterpret_cast<const float4 *>(rgb + 4 * q);
            uchar4 o;
            o.x = (uint8_t)clampf(v.x * 255.0f, 0.0f, 255.0f); o.y = (uint8_t)clampf(v.y * 255.0f, 0.0f, 255.0f);
            o.z = (uint8_t)clampf(v.z * 255.0f, 0.0f, 255.0f); o.w = (uint8_t)clampf(v.w * 255.0f, 0.0f, 255.0f);
            *reinterpret_cast<uchar4 *>(out + 4 * q) = o;
        }
        for (size_t k = 4 * quads + (size_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (size_t)gridDim.x * kBlock)
            out[k] = (uint8_t)clampf(rgb[k] * 255.0f, 0.0f, 255.0f);
    }
    // the cursor moves on when every workgroup of the launch has read it: the last one to get here advances it (a frame group: only its last frame's store)
    __syncthreads();
    if (advance != 0xFFFFFFFFu && threadIdx.x == 0 && atomicAdd(&job->ticket[lane], 1u) == gridDim.x - 1u) {
        job->ticket[lane] = 0u;
        job->cursor[lane] = job->cursor[lane] + (advance ? advance : job->lanes);
    }
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_clip_fetch_at(const gfpp_clip_job *job, uint32_t lane, uint32_t sub, float *static_in, uint32_t row_floats, gfpp_stream_t stream) {
    GFPP_REQUIRE_EARLY(job && static_in && lane < 8 && row_floats > 0, "gfpp_clip_fetch");
    hipLaunchKernelGGL(k_clip_fetch, dim3((row_floats + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, job, lane, sub, static_in, row_floats);
    return check_launch("gfpp_clip_fetch");
}

GFPP_API int gfpp_clip_fetch_group(const gfpp_clip_job *job, uint32_t lane, uint32_t count, float *static_in, uint32_t row_floats, gfpp_stream_t stream) {
    GFPP_REQUIRE_EARLY(job && static_in && lane < 8 && row_floats > 0 && count >= 1 && count <= 16, "gfpp_clip_fetch_group");
    hipLaunchKernelGGL(k_clip_fetch, dim3((row_floats + kBlock - 1) / kBlock, count), dim3(kBlock), 0, (hipStream_t)stream, job, lane, 0u, static_in, row_floats);
    return check_launch("gfpp_clip_fetch_group");
}

GFPP_API int gfpp_clip_fetch(const gfpp_clip_job *job, uint32_t lane, float *static_in, uint32_t row_floats, gfpp_stream_t stream) {
    return gfpp_clip_fetch_at(job, lane, 0u, static_in, row_floats, stream);
}

GFPP_API int gfpp_clip_store_u8_at(gfpp_clip_job *job, uint32_t lane, uint32_t sub, uint32_t advance, const float *rgb, uint64_t n_values, gfpp_stream_t stream) {
    GFPP_REQUIRE_EARLY(job && rgb && lane < 8 && n_values > 0 && ((uintptr_t)rgb & 15u) == 0, "gfpp_clip_store_u8");
    const uint64_t threads = (n_values + 3) / 4;
    uint64_t blocks = (threads + kBlock - 1) / kBlock;
    if (blocks > 128) blocks = 128;
    hipLaunchKernelGGL(k_clip_store_u8, dim3((uint32_t)blocks), dim3(kBlock), 0, (hipStream_t)stream, job, lane, sub, advance, rgb, (size_t)n_values);
    return check_launch("gfpp_clip_store_u8");
}

GFPP_API int gfpp_clip_store_u8(gfpp_clip_job *job, uint32_t lane, const float *rgb, uint64_t n_values, gfpp_stream_t stream) {
    return gfpp_clip_store_u8_at(job, lane, 0u, 0u, rgb, n_values, stream);
}

GFPP_API int gfpp_graph_replay(void *const *execs, void *const *streams, uint32_t lanes, uint32_t first_lane, uint32_t count, uint32_t max_ahead) {
    GFPP_REQUIRE_EARLY(execs && streams && lanes > 0 && lanes <= 8 && max_ahead <= 16, "gfpp_graph_replay");
    // max_ahead > 0: at most that many frames of a lane are queued ahead of the GPU (the issuing thread waits on the lane's frame max_ahead back)
    hipEvent_t ev[8][16];
    if (max_ahead)
        for (uint32_t l = 0; l < lanes; ++l)
            for (uint32_t d = 0; d < max_ahead; ++d)
                if (hipEventCreateWithFlags(&ev[l][d], hipEventDisableTiming) != hipSuccess) { set_error("gfpp_graph_replay: cannot create events"); return GFPP_EINVAL; }
    int rc = 0;
    for (uint32_t k = 0; k < count && rc == 0; ++k) {
        const uint32_t lane = (first_lane + k) % lanes, turn = k / lanes;
        if (max_ahead && turn >= max_ahead) (void)hipEventSynchronize(ev[lane][turn % max_ahead]);
        const hipError_t err = hipGraphLaunch((hipGraphExec_t)execs[lane], (hipStream_t)streams[lane]);
        if (err != hipSuccess) { set_error("gfpp_graph_replay: hipGraphLaunch failed at frame %u (%s)", k, hipGetErrorString(err)); rc = (int)err; break; }
        if (max_ahead) (void)hipEventRecord(ev[lane][turn % max_ahead], (hipStream_t)streams[lane]);
    }
    if (max_ahead)
        for (uint32_t l = 0; l < lanes; ++l)
            for (uint32_t d = 0; d < max_ahead; ++d) (void)hipEventDestroy(ev[l][d]);
    return rc;
}

namespace gfpp {
static gfpp_tuning default_tuning() {
    gfpp_tuning t{};
    t.size = (uint32_t)sizeof(gfpp_tuning);
    t.trip_pool = 1; t.lp_separate_trips = -1; t.occ_clip = 1; t.barrier_spins = 0; t.persist_caps = 0; t.persist_xcd = 0; t.torso_group_wgs = 0;
    t.sr_fuse_first = 1; t.sr_final_resident = 1; t.sr_up_poly = 0; t.grid_bwd_scatter = 0; t.wgrad_tr = 1; t.grid_bwd_bins = 1; t.march_fixed_step = 1;
    return t;
}
static gfpp_tuning g_tuning = default_tuning();
const gfpp_tuning &tuning() { return g_tuning; }
}  // namespace gfpp

GFPP_API int gfpp_set_tuning(const gfpp_tuning *t) {
    if (!t) { gfpp::g_tuning = gfpp::default_tuning(); return 0; }
    if (t->size != sizeof(gfpp_tuning)) { gfpp::set_error("gfpp_set_tuning: record of %u bytes, this library's gfpp_tuning has %u", t->size, (unsigned)sizeof(gfpp_tuning)); return GFPP_EINVAL; }
    gfpp::g_tuning = *t;
    return 0;
}
GFPP_API int gfpp_get_tuning(gfpp_tuning *out) {
    if (!out || out->size != sizeof(gfpp_tuning)) { gfpp::set_error("gfpp_get_tuning: out->size must be sizeof(gfpp_tuning)"); return GFPP_EINVAL; }
    *out = gfpp::g_tuning;
    return 0;
}

GFPP_API int gfpp_abi_version(void) { return GFPP_ABI_VERSION; }
GFPP_API const char *gfpp_last_error(void) { return gfpp::g_err; }
GFPP_API unsigned gfpp_struct_size(const char *name) {
    if (!name) return 0;
    const struct { const char *n; unsigned s; } table[] = {
        {"frame_ws", (unsigned)sizeof(gfpp_frame_ws)},       {"head_model", (unsigned)sizeof(gfpp_head_model)}, {"torso_model", (unsigned)sizeof(gfpp_torso_model)},
        {"cond_model", (unsigned)sizeof(gfpp_cond_model)},   {"grid_desc", (unsigned)sizeof(gfpp_grid_desc)},   {"grid_level", (unsigned)sizeof(gfpp_grid_level)},
        {"sr_model", (unsigned)sizeof(gfpp_sr_model)},       {"sr_ws", (unsigned)sizeof(gfpp_sr_ws)},         {"clip_job", (unsigned)sizeof(gfpp_clip_job)},
        {"tuning", (unsigned)sizeof(gfpp_tuning)},
    };
    for (const auto &e : table)
        if (strcmp(e.n, name) == 0) return e.s;
    return 0;
}

#define GFPP_REQUIRE(cond, what)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            gfpp::set_error("%s: invalid argument (%s)", what, #cond); \
            return GFPP_EINVAL;                                   \
        }                                                         \
    } while (0)

GFPP_API int gfpp_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                                     float *nears, float *fars, gfpp_stream_t stream) {
    if (N == 0) return 0;
    GFPP_REQUIRE(rays_o && rays_d && aabb && nears && fars, "gfpp_near_far_from_aabb");
    hipLaunchKernelGGL(k_near_far, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch("gfpp_near_far_from_aabb");
}

// ---- bounds of the occupied cells ---------------------------------------------------------------------------------------------------------
// One workgroup walks the bitfield (2 Mbit for one cascade of 128^3: 64 K words), every set bit widens per-thread bounds by the world extent of
// its cell; LDS reduction; thread 0 writes lo xyz, hi xyz widened by one cell of the coarsest occupied level.  Build-time work (once per bitfield).
__global__ __launch_bounds__(1024) void k_occupancy_bounds(const uint32_t *__restrict__ words, uint32_t n_words, uint32_t H, uint32_t log2_H3, float bound, float *__restrict__ out6) {
    __shared__ float s_lo[3][1024], s_hi[3][1024];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float margin = 0.0f;
    for (uint32_t w = threadIdx.x; w < n_words; w += 1024u) {
        uint32_t bits = words[w];
        while (bits) {
            const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            const uint32_t cell = w * 32u + b, level = cell >> log2_H3, code = cell & ((1u << log2_H3) - 1u);
            const float mip_bound = fminf(scalbnf(1.0f, (int)level), bound), cw = 2.0f * mip_bound / (float)H;
            const uint32_t n[3] = {compact3(code), compact3(code >> 1), compact3(code >> 2)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // positions are clamped into the first / last cell of an axis (voxel_of): those cells reach to the scene bound
                const float a = n[k] == 0u ? -bound : fmaf((float)n[k], cw, -mip_bound), z = n[k] + 1u >= H ? bound : fmaf((float)(n[k] + 1u), cw, -mip_bound);
                lo[k] = fminf(lo[k], a);
                hi[k] = fmaxf(hi[k], z);
            }
            margin = fmaxf(margin, cw);
        }
    }
    __shared__ float s_margin[1024];
    for (int k = 0; k < 3; ++k) { s_lo[k][threadIdx.x] = lo[k]; s_hi[k][threadIdx.x] = hi[k]; }
    s_margin[threadIdx.x] = margin;
    __syncthreads();
    for (uint32_t stride = 512u; stride > 0u; stride >>= 1) {
        if (threadIdx.x < stride) {
            for (int k = 0; k < 3; ++k) {
                s_lo[k][threadIdx.x] = fminf(s_lo[k][threadIdx.x], s_lo[k][threadIdx.x + stride]);
                s_hi[k][threadIdx.x] = fmaxf(s_hi[k][threadIdx.x], s_hi[k][threadIdx.x + stride]);
            }
            s_margin[threadIdx.x] = fmaxf(s_margin[threadIdx.x], s_margin[threadIdx.x + stride]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const bool any = s_hi[k][0] >= s_lo[k][0];
        out6[k] = any ? s_lo[k][0] - s_margin[0] : 1.0f;          // empty bitfield: lo > hi
        out6[3 + k] = any ? s_hi[k][0] + s_margin[0] : -1.0f;
    }
}

GFPP_API int gfpp_occupancy_bounds(const uint8_t *bitfield, uint32_t cascade, uint32_t grid_size, float bound, float *out6, gfpp_stream_t stream) {
    GFPP_REQUIRE(bitfield && out6, "gfpp_occupancy_bounds");
    uint32_t log2_H = 0;
    while ((1u << log2_H) < grid_size) ++log2_H;
    if ((1u << log2_H) != grid_size || grid_size < 4u || grid_size > 256u || cascade == 0u || cascade > 8u || !(bound > 0.0f)) {
        set_error("gfpp_occupancy_bounds: grid_size must be a power of two in 4..256, cascade in 1..8, bound > 0");
        return GFPP_EINVAL;
    }
    const uint32_t n_words = cascade * grid_size * grid_size * grid_size / 32u;
    hipLaunchKernelGGL(k_occupancy_bounds, dim3(1), dim3(1024), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(bitfield), n_words, grid_size, 3u * log2_H, bound, out6);
    return check_launch("gfpp_occupancy_bounds");
}

GFPP_API int gfpp_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, gfpp_stream_t stream) {
    if (N == 0) return 0;
    GFPP_REQUIRE(coords && indices, "gfpp_morton3D");
    hipLaunchKernelGGL(k_morton, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, coords, N, indices);
    return check_launch("gfpp_morton3D");
}

GFPP_API int gfpp_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, gfpp_stream_t stream) {
    if (N == 0) return 0;
    GFPP_REQUIRE(coords && indices, "gfpp_morton3D_invert");
    hipLaunchKernelGGL(k_morton_invert, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, indices, N, coords);
    return check_launch("gfpp_morton3D_invert");
}

GFPP_API int gfpp_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, gfpp_stream_t stream) {
    if (N == 0) return 0;
    GFPP_REQUIRE(grid && bitfield, "gfpp_packbits");
    GFPP_REQUIRE(((uintptr_t)grid & 15u) == 0, "gfpp_packbits");
    hipLaunchKernelGGL(k_packbits, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)grid, N, density_thresh, bitfield);
    return check_launch("gfpp_packbits");
}

GFPP_API int gfpp_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                             const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                             const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                             const float *noises, gfpp_stream_t stream) {
    (void)nears;
    if (n_alive == 0) return 0;
    GFPP_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas && noises, "gfpp_march_rays");
    GFPP_REQUIRE(n_step >= 1 && C >= 1 && C <= 8 && H >= 1 && H <= 1024 && max_steps >= 1, "gfpp_march_rays");
    const MarchParams p = make_march_params(bound, dt_gamma, max_steps, C, H);
    hipLaunchKernelGGL(k_march, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, p, grid, fars, xyzs, dirs, deltas, noises);
    return check_launch("gfpp_march_rays");
}

GFPP_API int gfpp_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                                 const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                                 float *image, gfpp_stream_t stream) {
    if (n_alive == 0) return 0;
    GFPP_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "gfpp_composite_rays");
    GFPP_REQUIRE(n_step >= 1, "gfpp_composite_rays");
    hipLaunchKernelGGL(k_composite, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_alive, n_step, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return check_launch("gfpp_composite_rays");
}

GFPP_API int gfpp_get_rays(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, float *rays_o,
                           float *rays_d, gfpp_stream_t stream) {
    if (H == 0 || W == 0) return 0;
    GFPP_REQUIRE(pose && rays_o && rays_d, "gfpp_get_rays");
    hipLaunchKernelGGL(k_get_rays, dim3(div_up(H * W, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pose, fx, fy, cx, cy, H, W, rays_o, rays_d);
    return check_launch("gfpp_get_rays");
}

GFPP_API int gfpp_get_rays_at(const float *pose, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, const int64_t *inds,
                              uint32_t n_rays, float *rays_o, float *rays_d, gfpp_stream_t stream) {
    if (n_rays == 0) return 0;
    GFPP_REQUIRE(pose && inds && rays_o && rays_d && H && W, "gfpp_get_rays_at");
    hipLaunchKernelGGL(k_get_rays_at, dim3(div_up(n_rays, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pose, fx, fy, cx, cy, W,
                       (const long long *)inds, n_rays, rays_o, rays_d);
    return check_launch("gfpp_get_rays_at");
}

GFPP_API int gfpp_rgb_to_u8(const float *rgb, uint64_t n_values, uint8_t *out, gfpp_stream_t stream) {
    if (n_values == 0) return 0;
    GFPP_REQUIRE(rgb && out, "gfpp_rgb_to_u8");
    GFPP_REQUIRE(((uintptr_t)rgb & 15u) == 0 && ((uintptr_t)out & 3u) == 0, "gfpp_rgb_to_u8");
    const uint64_t threads = (n_values + 3) / 4;
    hipLaunchKernelGGL(k_rgb_to_u8, dim3((uint32_t)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, rgb, (size_t)n_values, out);
    return check_launch("gfpp_rgb_to_u8");
}
