// frame_head.hip -- fused head-NeRF frame pipeline for gfx950.
//
// One "trip" kernel = one iteration of the reference's render loop (renderer.py:354-384), entirely on the device:
//   phase 1  march      : one thread per alive ray emits up to n_step occupied samples into LDS (march_device.h)
//   phase 2  evaluate   : occupied samples are compacted inside the tile; each wavefront takes 32 of them and runs
//                         position grid -> ambient MLP -> tanh -> ambient grid -> sigma MLP -> exp -> SH -> colour MLP
//                         -> sigmoid with v_mfma_f32_32x32x2_f32 (exact fp32 fma chains).  Weights are the A operand
//                         (pre-packed in fragment order, see gfpp_radnerf.h), the 32 samples are the B columns, so the
//                         accumulator registers of one layer ARE the B operands of the next: activations never leave
//                         the register file and nothing is written to HBM between encode and composite.
//   phase 3  composite  : one thread per ray folds its samples into (weight_sum, depth, rgb), decides alive/dead, and the
//                         survivors are appended to the next trip's list (one atomicAdd per tile).
// Loop control (n_alive -> n_step, cumulative step, exit) is recomputed by every workgroup from the counters that the
// previous trips left in memory, so the host never synchronises.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "head_eval_device.h"
#include "head_eval_f32_device.h"

namespace gfpp {

struct TileShared {
    float px[kTile], py[kTile], pz[kTile], dt[kTile], tend[kTile];  // by slot = ray_local * n_step + s
    float sigma[kTile], cr[kTile], cg[kTile], cb[kTile];            // by slot
    float dx[kTile], dy[kTile], dz[kTile];                          // by ray_local
    uint32_t order[kTile];                                          // compact index -> slot
    uint32_t wave_tot[4];
    uint32_t n_valid;
    uint32_t out_base;
};

template <int AMB_D>
__global__ __launch_bounds__(kThreads, 2) void k_head_trip(TripArgs a) {
    __shared__ TileShared sh;
    // ---- loop state, recomputed from the per-trip counters (renderer.py:354-384) ----------------------------------
    uint32_t step_before = 0;
    for (uint32_t k = 0; k < a.trip; ++k) {
        const uint32_t na = (uint32_t)a.gcounters[k];
        if (na == 0) return;
        uint32_t ns = a.N_global / na;
        ns = ns < 1u ? 1u : (ns > 8u ? 8u : ns);
        step_before += ns;
    }
    // sample budget from the FRAME-wide alive count (renderer.py:364), work list from this launch's own (they coincide on one GPU)
    const uint32_t n_alive_frame = (uint32_t)a.gcounters[a.trip];
    if (n_alive_frame == 0 || step_before >= a.max_steps) return;
    uint32_t n_step = a.N_global / n_alive_frame;
    n_step = n_step < 1u ? 1u : (n_step > 8u ? 8u : n_step);
    const uint32_t n_alive = (uint32_t)a.counters[a.trip];
    if (n_alive == 0) return;

    const uint32_t rays_per_tile = kTile / n_step;
    const uint32_t n_tiles = (n_alive + rays_per_tile - 1) / rays_per_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- phase 1: march ---------------------------------------------------------------------------------------
        const uint32_t n = tile * rays_per_tile + tid;
        const bool has_ray = (uint32_t)tid < rays_per_tile && n < n_alive;
        uint32_t ray = 0, cnt = 0;
        float t = 0.0f;
        if (has_ray) {
            ray = a.trip == 0 ? n : (uint32_t)a.alive_in[n];
            const float *o = a.rays_o + 3ull * ray, *d = a.rays_d + 3ull * ray;
            const float dx = d[0], dy = d[1], dz = d[2];
            sh.dx[tid] = dx; sh.dy[tid] = dy; sh.dz[tid] = dz;
            t = ray_state_t(a.state, ray);
            const uint32_t base = tid * n_step;
            cnt = march_one_ray(o[0], o[1], o[2], dx, dy, dz, t, a.fars[ray], n_step, a.bitfield, a.mp, [&](uint32_t s, const Sample &smp) {
                sh.px[base + s] = smp.x; sh.py[base + s] = smp.y; sh.pz[base + s] = smp.z;
                sh.dt[base + s] = smp.dt; sh.tend[base + s] = smp.t_end;
            });
        }
        // tile-level compaction of the occupied samples (rays live in waves 0 and 1 only)
        const uint32_t incl = wave_inclusive_scan(cnt, lane);
        if (lane == 63) sh.wave_tot[wave] = incl;
        __syncthreads();
        {
            uint32_t before = 0;
            for (int w = 0; w < wave; ++w) before += sh.wave_tot[w];
            uint32_t pos = before + incl - cnt;
            for (uint32_t s = 0; s < cnt; ++s) sh.order[pos + s] = tid * n_step + s;
            if (tid == kThreads - 1) {
                sh.n_valid = before + incl;
                if (before + incl) atomicAdd(&a.counters[64 + a.trip], (int)(before + incl));   // evaluated samples of this trip
            }
        }
        __syncthreads();

        // ---- phase 2: evaluate the radiance field on the occupied samples, 32 per wavefront -------------------------
        const uint32_t n_valid = sh.n_valid;
        for (uint32_t first = wave * 32; first < n_valid; first += 4 * 32) evaluate_block<AMB_D>(a, sh, first, n_step, lane);
        __syncthreads();

        // ---- phase 3: composite, ray state update, survivor compaction -------------------------------------------------
        bool survives = false;
        if (has_ray) {
            RayAccum acc = ray_state_load(a.state, ray);
            const uint32_t base = tid * n_step;
            uint32_t s = 0;
            float t_last = t;
            for (; s < cnt; ++s) {
                const uint32_t k = base + s;
                t_last = sh.tend[k];
                if (composite_sample(acc, sh.sigma[k], sh.dt[k], t_last, sh.cr[k], sh.cg[k], sh.cb[k], a.T_thresh)) break;
            }
            // the reference declares the ray dead when it stops before n_step samples (terminated, or ran out of samples)
            survives = (s == n_step);
            ray_state_store(a.state, ray, acc, survives ? t_last : t);
        }
        const unsigned long long ballot = __ballot(survives);
        const uint32_t wave_cnt = (uint32_t)__popcll(ballot);
        if (lane == 0) sh.wave_tot[wave] = wave_cnt;
        __syncthreads();
        if (tid == 0) {
            const uint32_t total = sh.wave_tot[0] + sh.wave_tot[1] + sh.wave_tot[2] + sh.wave_tot[3];
            sh.out_base = total ? (uint32_t)atomicAdd(&a.counters[a.trip + 1], (int)total) : 0u;
        }
        __syncthreads();
        if (survives) {
            uint32_t before = sh.out_base;
            for (int w = 0; w < wave; ++w) before += sh.wave_tot[w];
            const uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
            a.alive_out[before + rank] = (int32_t)ray;
        }
        __syncthreads();
    }
}

// ---- the same trip with autonomous wavefronts -------------------------------------------------------------------------------------
// k_head_trip above synchronises its four wavefronts around every phase of a 128-slot tile (one 32-sample block each between barriers)
// and walks the occupancy bitfield inside the trip.  This variant takes the structure of the 16-bit kernel (frame_head_lp.hip): samples
// come from the frame's pre-marched lists, a wavefront owns a tile of up to 64 rays / 128 slots from fetch to survivor append, and there
// is no workgroup barrier at all -- one wavefront's grid gathers and compositing overlap the other wavefronts' MFMA chains.  The per-sample
// arithmetic is evaluate_block, unchanged, so every sample gets the same bits as in k_head_trip.
constexpr int kWSlots = 128, kWRays = 64;
struct WaveTile {
    float px[kWSlots], py[kWSlots], pz[kWSlots], dt[kWSlots], tend[kWSlots];   // by slot = ray_local * n_step + s
    float sigma[kWSlots], cr[kWSlots], cg[kWSlots], cb[kWSlots];               // by slot
    float dx[kWRays], dy[kWRays], dz[kWRays];                                  // by ray_local
    uint32_t order[kWSlots];                                                   // compact index -> slot
    uint32_t n_valid;
};

template <int AMB_D>
__global__ __launch_bounds__(kThreads, 2) void k_head_trip_w(TripArgs a) {
    __shared__ WaveTile tiles[kThreads / 64];
    uint32_t step_before = 0;
    for (uint32_t k = 0; k < a.trip; ++k) {
        const uint32_t na = (uint32_t)a.gcounters[k];
        if (na == 0) return;
        uint32_t ns = a.N_global / na;
        ns = ns < 1u ? 1u : (ns > 8u ? 8u : ns);
        step_before += ns;
    }
    // sample budget from the FRAME-wide alive count (renderer.py:364), work list from this launch's own (they coincide on one GPU)
    const uint32_t n_alive_frame = (uint32_t)a.gcounters[a.trip];
    if (n_alive_frame == 0 || step_before >= a.max_steps) return;
    uint32_t n_step = a.N_global / n_alive_frame;
    n_step = n_step < 1u ? 1u : (n_step > 8u ? 8u : n_step);
    const uint32_t n_alive = (uint32_t)a.counters[a.trip];
    if (n_alive == 0) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t kWaves = kThreads / 64;
    const uint32_t waves_total = gridDim.x * kWaves;
    // tile size by cost: (tiles per wavefront) x (blocks per tile + per-tile overhead); see k_head_trip_lp
    uint32_t rays_per_tile = (uint32_t)kWSlots / n_step < (uint32_t)kWRays ? (uint32_t)kWSlots / n_step : (uint32_t)kWRays;
    {
        uint32_t best = 0xFFFFFFFFu, best_rpt = rays_per_tile;
        for (uint32_t rpt = rays_per_tile; rpt * n_step >= 32u || rpt == rays_per_tile; rpt >>= 1) {
            const uint32_t tiles_n = (n_alive + rpt - 1) / rpt;
            const uint32_t crit = ((tiles_n + waves_total - 1) / waves_total) * (10u * ((rpt * n_step + 31u) / 32u) + 1u);
            if (crit < best) { best = crit; best_rpt = rpt; }
            if (rpt == 1u) break;
        }
        rays_per_tile = best_rpt;
    }
    const uint32_t n_tiles = (n_alive + rays_per_tile - 1) / rays_per_tile;
    WaveTile &wt = tiles[wave];
    const uint32_t gw = blockIdx.x * kWaves + wave;
    uint32_t evaluated = 0;
    // every ray that is still alive took the full n_step samples in each earlier trip (a shorter take declares it dead), so its cursor into
    // the pre-marched list is the loop's cumulative step count
    const uint32_t used = step_before;
    for (uint32_t tile = gw; tile < n_tiles; tile += waves_total) {
        // ---- phase 1: this trip's samples of every ray (one lane per ray) --------------------------------------------------------------
        const uint32_t n = tile * rays_per_tile + lane;
        const bool has_ray = (uint32_t)lane < rays_per_tile && n < n_alive;
        uint32_t ray = 0, cnt = 0;
        if (has_ray) {
            ray = a.trip == 0 ? n : (uint32_t)a.alive_in[n];
            const uint32_t avail = a.sample_cnt[ray] - used;
            cnt = avail < n_step ? avail : n_step;
            const float *o = a.rays_o + 3ull * ray, *d = a.rays_d + 3ull * ray;
            const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
            wt.dx[lane] = dx; wt.dy[lane] = dy; wt.dz[lane] = dz;
            const float *ts = a.sample_t + (size_t)ray * a.sample_stride + used;
            const uint32_t base = lane * n_step;
            for (uint32_t s = 0; s < cnt; ++s) {
                // the same expressions as march_one_ray (raymarching.cu:873-882, 905-913) evaluated at the stored t
                const float t0 = ts[s];
                const float dt = clampf(t0 * a.mp.dt_gamma, a.mp.dt_min, a.mp.dt_max);
                wt.px[base + s] = clampf(fmaf(t0, dx, ox), -a.mp.bound, a.mp.bound);
                wt.py[base + s] = clampf(fmaf(t0, dy, oy), -a.mp.bound, a.mp.bound);
                wt.pz[base + s] = clampf(fmaf(t0, dz, oz), -a.mp.bound, a.mp.bound);
                wt.dt[base + s] = dt;
                wt.tend[base + s] = t0 + dt;
            }
        }
        const uint32_t incl = wave_inclusive_scan(cnt, lane);
        const uint32_t n_valid = (uint32_t)__shfl((int)incl, 63);
        {
            const uint32_t pos = incl - cnt;
            for (uint32_t s = 0; s < cnt; ++s) wt.order[pos + s] = lane * n_step + s;
            if (lane == 0) wt.n_valid = n_valid;
        }
        wave_sync();
        // ---- phase 2: evaluate, 32 samples per pass ---------------------------------------------------------------------------------------
        for (uint32_t first = 0; first < n_valid; first += 32) evaluate_block<AMB_D>(a, wt, first, n_step, lane);
        evaluated += n_valid;
        wave_sync();
        // ---- phase 3: composite, survivor compaction ------------------------------------------------------------------------------------------
        bool survives = false;
        if (has_ray) {
            RayAccum acc = ray_state_load(a.state, ray);
            const uint32_t base = lane * n_step;
            uint32_t s = 0;
            for (; s < cnt; ++s) {
                const uint32_t k = base + s;
                if (composite_sample(acc, wt.sigma[k], wt.dt[k], wt.tend[k], wt.cr[k], wt.cg[k], wt.cb[k], a.T_thresh)) break;
            }
            survives = (s == n_step);
            ray_state_store(a.state, ray, acc, 0.0f);        // (this kernel's cursor is the loop's cumulative step count, not stored per ray)
        }
        const unsigned long long ballot = __ballot(survives);
        const uint32_t total = (uint32_t)__popcll(ballot);
        uint32_t out_base = 0;
        if (lane == 0 && total) out_base = (uint32_t)atomicAdd(&a.counters[a.trip + 1], (int)total);
        out_base = (uint32_t)__shfl((int)out_base, 0);
        if (survives) a.alive_out[out_base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull))] = (int32_t)ray;
        wave_sync();
    }
    if (lane == 0 && evaluated) atomicAdd(&a.counters[64 + a.trip], (int)evaluated);
}

// The same trip with the samples of a whole workgroup pooled -- the fp32 twin of k_head_trip_pool (frame_head_lp.hip, where the reasons and
// the measurements are): every workgroup takes an equal share of the alive rays (wavefront tiles dealt out through a multiplicative
// permutation), the occupied samples of its four tiles are compacted together so that every 32-sample block but the last is full, the
// blocks are dealt out round-robin, and the survivors of the workgroup are appended with ONE atomic.  k_head_trip_w stays as the A/B
// partner (GFPP_TRIP_POOL=0); per sample nothing changes (same evaluate_block, same order along a ray).
constexpr int kWaveCount = kThreads / 64;
constexpr int kPoolW = kWaveCount * kWSlots;   // 512 sample slots of one workgroup round
constexpr uint32_t kNoRayW = 0xFFu;
struct WavePool {
    float px[kPoolW], py[kPoolW], pz[kPoolW], dt[kPoolW], tend[kPoolW];   // by slot = ray_local * n_step + s
    float sigma[kPoolW], cr[kPoolW], cg[kPoolW], cb[kPoolW];              // by slot
    float dx[kPoolW], dy[kPoolW], dz[kPoolW];                             // by ray_local (wavefront w owns [w * rw, (w + 1) * rw))
    uint32_t ray[kPoolW];                                                 // by ray_local
    uint16_t order[kPoolW];                                               // compact index -> slot, over the whole workgroup
    uint8_t cnt[kPoolW];                                                  // samples the local ray takes in this trip; kNoRayW: no such ray
    uint32_t wave_valid[kWaveCount], wave_surv[kWaveCount];
    uint32_t n_valid, out_base;
};

template <int AMB_D>
__global__ __launch_bounds__(kThreads, 2) void k_head_trip_wp(TripArgs a) {
    __shared__ WavePool pool;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the alive counts of every trip up to this one are final: one parallel fetch (lane k takes trip k; trips < 64)
    const uint32_t fetched = (uint32_t)lane <= a.trip ? (uint32_t)a.gcounters[lane] : 0u;
    uint32_t step_before = 0;
    for (uint32_t k = 0; k < a.trip; ++k) {
        const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)fetched, (int)k);
        if (na == 0) return;
        uint32_t ns = a.N_global / na;
        ns = ns < 1u ? 1u : (ns > 8u ? 8u : ns);
        step_before += ns;
    }
    const uint32_t n_alive_frame = (uint32_t)__builtin_amdgcn_readlane((int)fetched, (int)a.trip);
    if (n_alive_frame == 0 || step_before >= a.max_steps) return;
    uint32_t n_step = a.N_global / n_alive_frame;
    n_step = n_step < 1u ? 1u : (n_step > 8u ? 8u : n_step);
    const uint32_t n_alive = a.gcounters == a.counters ? n_alive_frame : (uint32_t)a.counters[a.trip];
    if (n_alive == 0) return;
    const uint32_t used = step_before;   // samples every alive ray has consumed so far

    // equal shares: `rounds` rounds of gridDim.x * 4 wavefront tiles of rw rays (rw * n_step <= 128 slots per wavefront)
    const uint32_t tiles_per_round = gridDim.x * kWaveCount;
    const uint32_t rw_max = (uint32_t)kWSlots / n_step;
    const uint32_t rounds = (n_alive + tiles_per_round * rw_max - 1) / (tiles_per_round * rw_max);
    const uint32_t rw = (n_alive + tiles_per_round * rounds - 1) / (tiles_per_round * rounds);
    const uint32_t n_tiles = (n_alive + rw - 1) / rw;
    const uint32_t mult = n_tiles % 1237u ? 1237u : 1u;   // q -> q * mult mod n_tiles is a permutation of the tiles (1237 is prime)

    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t q0 = (r * gridDim.x + blockIdx.x) * kWaveCount;
        if (q0 >= n_tiles) break;   // nothing for this workgroup in this round (the same decision in all its wavefronts)
        const uint32_t q = q0 + (uint32_t)wave;
        const bool has_tile = q < n_tiles;
        const uint32_t tile = has_tile ? (uint32_t)(((unsigned long long)q * mult) % n_tiles) : 0u;

        // ---- phase 1: this trip's samples of the wavefront's rw rays (one or two rays per lane) into the pool -----------------------
        uint32_t my_valid = 0, cnt_r[2], pos_r[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t i = (uint32_t)(sub * 64 + lane);
            const uint32_t local = (uint32_t)wave * rw + i;
            const uint32_t n = tile * rw + i;
            const bool in_tile = i < rw;
            const bool has_ray = in_tile && has_tile && n < n_alive;
            uint32_t cnt = 0;
            if (has_ray) {
                const uint32_t ray = a.trip == 0 ? n : (uint32_t)a.alive_in[n];
                const uint32_t avail = a.sample_cnt[ray] - used;
                cnt = avail < n_step ? avail : n_step;
                pool.ray[local] = ray;
                const float *o = a.rays_o + 3ull * ray, *d = a.rays_d + 3ull * ray;
                const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
                pool.dx[local] = dx; pool.dy[local] = dy; pool.dz[local] = dz;
                const float *ts = a.sample_t + (size_t)ray * a.sample_stride + used;
                const uint32_t base = local * n_step;
                for (uint32_t s = 0; s < cnt; ++s) {
                    // the same expressions as march_one_ray (raymarching.cu:873-882, 905-913) evaluated at the stored t
                    const float t0 = ts[s];
                    const float dt = clampf(t0 * a.mp.dt_gamma, a.mp.dt_min, a.mp.dt_max);
                    pool.px[base + s] = clampf(fmaf(t0, dx, ox), -a.mp.bound, a.mp.bound);
                    pool.py[base + s] = clampf(fmaf(t0, dy, oy), -a.mp.bound, a.mp.bound);
                    pool.pz[base + s] = clampf(fmaf(t0, dz, oz), -a.mp.bound, a.mp.bound);
                    pool.dt[base + s] = dt;
                    pool.tend[base + s] = t0 + dt;
                }
            }
            if (in_tile) pool.cnt[local] = (uint8_t)(has_ray ? cnt : kNoRayW);
            const uint32_t incl = wave_inclusive_scan(cnt, lane);
            cnt_r[sub] = cnt;
            pos_r[sub] = my_valid + incl - cnt;
            my_valid += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) pool.wave_valid[wave] = my_valid;
        __syncthreads();
        // compaction over the workgroup: wavefront tiles in order, rays in order inside a tile
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int v = 0; v < kWaveCount; ++v) {
            const uint32_t x = pool.wave_valid[v];
            before += v < wave ? x : 0u;
            total += x;
        }
        if (tid == 0) pool.n_valid = total;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t slot0 = ((uint32_t)wave * rw + (uint32_t)(sub * 64 + lane)) * n_step;
            for (uint32_t s = 0; s < cnt_r[sub]; ++s) pool.order[before + pos_r[sub] + s] = (uint16_t)(slot0 + s);
        }
        __syncthreads();

        // ---- phase 2: the pooled blocks, dealt out round-robin --------------------------------------------------------------------------
        for (uint32_t first = 32u * (uint32_t)wave; first < total; first += 32u * kWaveCount) evaluate_block<AMB_D>(a, pool, first, n_step, lane);
        __syncthreads();

        // ---- phase 3: composite, ray state update (the owner of the ray); survivor compaction over the workgroup, one append ----------
        unsigned long long alive_bits[2] = {0ull, 0ull};
        uint32_t ray_r[2] = {0u, 0u};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t i = (uint32_t)(sub * 64 + lane);
            if ((uint32_t)(sub * 64) >= rw) break;   // wavefront-uniform
            const uint32_t local = (uint32_t)wave * rw + i;
            const uint32_t cnt = i < rw ? (uint32_t)pool.cnt[local] : kNoRayW;
            bool survives = false;
            if (cnt != kNoRayW) {
                const uint32_t ray = pool.ray[local];
                ray_r[sub] = ray;
                RayAccum acc = ray_state_load(a.state, ray);
                const uint32_t base = local * n_step;
                uint32_t s = 0;
                for (; s < cnt; ++s) {
                    const uint32_t k = base + s;
                    if (composite_sample(acc, pool.sigma[k], pool.dt[k], pool.tend[k], pool.cr[k], pool.cg[k], pool.cb[k], a.T_thresh)) break;
                }
                survives = (s == n_step);
                ray_state_store(a.state, ray, acc, 0.0f);
            }
            alive_bits[sub] = __ballot(survives);
        }
        const uint32_t surv0 = (uint32_t)__popcll(alive_bits[0]), surv1 = (uint32_t)__popcll(alive_bits[1]);
        if (lane == 0) pool.wave_surv[wave] = surv0 + surv1;
        __syncthreads();
        if (tid == 0) {
            uint32_t sum = 0;
#pragma unroll
            for (int v = 0; v < kWaveCount; ++v) sum += pool.wave_surv[v];
            pool.out_base = sum ? (uint32_t)atomicAdd(&a.counters[a.trip + 1], (int)sum) : 0u;
            if (total) atomicAdd(&a.counters[64 + a.trip], (int)total);   // evaluated samples of this trip
        }
        __syncthreads();
        {
            uint32_t out = pool.out_base;
#pragma unroll
            for (int v = 0; v < kWaveCount; ++v) out += v < wave ? pool.wave_surv[v] : 0u;
            const unsigned long long below = (1ull << lane) - 1ull;
            if ((alive_bits[0] >> lane) & 1ull) a.alive_out[out + (uint32_t)__popcll(alive_bits[0] & below)] = (int32_t)ray_r[0];
            if ((alive_bits[1] >> lane) & 1ull) a.alive_out[out + surv0 + (uint32_t)__popcll(alive_bits[1] & below)] = (int32_t)ray_r[1];
        }
        // (no barrier here: the next round's phase 1 writes only the wavefront's own rays and slots, and its first barrier comes before
        // wave_valid / n_valid / wave_surv / out_base are read or written again)
    }
}

// ---- frame begin: slab test + state reset + constant folding ------------------------------------------------------
// ---- per-sample evaluation (RADNeRF.forward, radnerf.py:108-141) with the trip kernels' own arithmetic ---------------------------------
// A wavefront takes 32 caller-supplied (position, direction) pairs, lays them out as a one-sample-per-ray tile and runs evaluate_block --
// the very function the trips run -- so a per-sample parity test of this entry pins the MFMA layer wiring of the frame path.
struct EvalArgs {
    TripArgs t;
    const float *positions, *directions;
    float *sigma, *color, *ambient;
    uint32_t M;
};

template <int AMB_D>
__global__ __launch_bounds__(kThreads, 2) void k_head_eval(EvalArgs e) {
    __shared__ WaveTile tiles[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31;
    WaveTile &wt = tiles[wave];
    for (uint32_t base = (blockIdx.x * (kThreads / 64) + wave) * 32u; base < e.M; base += gridDim.x * (kThreads / 64) * 32u) {
        const uint32_t idx = base + (uint32_t)j;
        const bool ok = idx < e.M;
        if (lane < 32) {
            wt.px[j] = ok ? e.positions[3ull * idx] : 0.0f;
            wt.py[j] = ok ? e.positions[3ull * idx + 1] : 0.0f;
            wt.pz[j] = ok ? e.positions[3ull * idx + 2] : 0.0f;
            wt.dx[j] = ok ? e.directions[3ull * idx] : 0.0f;
            wt.dy[j] = ok ? e.directions[3ull * idx + 1] : 0.0f;
            wt.dz[j] = ok ? e.directions[3ull * idx + 2] : 1.0f;
            wt.order[j] = (uint32_t)j;
        }
        if (lane == 0) wt.n_valid = e.M - base < 32u ? e.M - base : 32u;
        wave_sync();
        TripArgs a = e.t;
        a.dbg_ambient = e.ambient ? e.ambient + (size_t)base * AMB_D : nullptr;
        evaluate_block<AMB_D, WaveTile, true>(a, wt, 0, 1, lane);
        wave_sync();
        if (lane < 32 && ok) {
            e.sigma[idx] = wt.sigma[j];
            e.color[3ull * idx] = wt.cr[j];
            e.color[3ull * idx + 1] = wt.cg[j];
            e.color[3ull * idx + 2] = wt.cb[j];
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(kThreads) void k_frame_begin(const float *__restrict__ rays_o, const float *__restrict__ rays_d, uint32_t N,
                                                         float min_near, float ax0, float ay0, float az0, float ax1, float ay1, float az1,
                                                         float *__restrict__ nears, float *__restrict__ fars, float *__restrict__ state,
                                                         int32_t *__restrict__ counters) {
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < kCounterWords) counters[threadIdx.x] = threadIdx.x == 0 ? (int32_t)N : 0;
    if (n >= N) return;
    const float aabb[6] = {ax0, ay0, az0, ax1, ay1, az1};
    const float *o = rays_o + 3ull * n, *d = rays_d + 3ull * n;
    const RayBox rb = ray_box(o[0], o[1], o[2], d[0], d[1], d[2], aabb, min_near);
    nears[n] = rb.near;
    fars[n] = rb.far;
    // word 5: t = near for the kernel that marches inside the trips; the pre-marched 16-bit kernel keeps its sample cursor in the same word
    // (written at the end of trip 0, never read before)
    *reinterpret_cast<float4 *>(state + (size_t)kRayRec * n) = float4{0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<float4 *>(state + (size_t)kRayRec * n + 4) = float4{0.0f, rb.near, rb.near, 0.0f};
}

// out[row] = sum_k W[row, k] * v[k] for 128 rows, written in bias-fragment order Bf[h][16*m+r] <- row 32*m+rr(r)+4*h.
__global__ __launch_bounds__(128) void k_fold_constants(const float *__restrict__ w_cond, const float *__restrict__ cond, uint32_t cond_dim,
                                                       const float *__restrict__ w_ind, const float *__restrict__ ind, uint32_t ind_dim,
                                                       float *__restrict__ frame_consts, uint32_t cond_stride) {
    const int row = threadIdx.x;  // 0..127
    // (blockIdx.y = frame of a batch: its conditioning row and its 256 constants; the individual code is the same for all)
    cond += (size_t)blockIdx.y * cond_stride;
    frame_consts += (size_t)blockIdx.y * 256u;
    const float *w = blockIdx.x == 0 ? w_cond : w_ind;
    const float *v = blockIdx.x == 0 ? cond : ind;
    const uint32_t K = blockIdx.x == 0 ? cond_dim : ind_dim;
    float s = 0.0f;
    if (w && v)
        for (uint32_t k = 0; k < K; ++k) s = fmaf(w[(size_t)row * K + k], v[k], s);
    const int m = row >> 5, i = row & 31;
    const int h = (i >> 2) & 1;
    const int r = (i & 3) + 4 * (i >> 3);
    frame_consts[blockIdx.x * 128 + h * 64 + m * 16 + r] = s;
}

__global__ __launch_bounds__(kThreads) void k_head_finish(uint32_t N, const float *__restrict__ nears, const float *__restrict__ fars,
                                                         const float *__restrict__ state, const float *__restrict__ bg_color, float bg_scalar,
                                                         float *__restrict__ out_image, float *__restrict__ out_depth) {
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const RayAccum acc = ray_state_load(state, n);
    const float T = 1.0f - acc.wsum;
    const float img[3] = {acc.r, acc.g, acc.b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float bg = bg_color ? bg_color[3ull * n + c] : bg_scalar;
        out_image[3ull * n + c] = clampf(img[c] + T * bg, 0.0f, 1.0f);
    }
    out_depth[n] = fmaxf(acc.depth - nears[n], 0.0f) / (fars[n] - nears[n]);
}

static bool grid_ok(const gfpp_grid_desc &g, uint32_t D) {
    return g.table && g.levels && g.D == D && g.L == 16 && g.gridtype <= 1 && g.interp <= 1 && g.dtype == GFPP_F32;
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_grid_level_table(uint32_t L, float S, uint32_t H, float *scale_out, uint32_t *resolution_out) {
    if (!scale_out || !resolution_out || L > (uint32_t)kMaxLevels) { set_error("gfpp_grid_level_table: bad arguments"); return GFPP_EINVAL; }
    GridLevels g;
    fill_level_scales(g, L, S, H);
    for (uint32_t l = 0; l < L; ++l) { scale_out[l] = g.scale[l]; resolution_out[l] = g.resolution[l]; }
    return 0;
}

GFPP_API int gfpp_grid_levels_fill(uint32_t D, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, const int32_t *offsets,
                                   uint32_t row_pad, gfpp_grid_level *levels) {
    if (!offsets || !levels || L > (uint32_t)kMaxLevels || D < 2 || D > 3 || gridtype > 1) { set_error("gfpp_grid_levels_fill: bad arguments"); return GFPP_EINVAL; }
    GridLevels g;
    fill_level_scales(g, L, S, H);
    for (uint32_t l = 0; l < L; ++l) {
        gfpp_grid_level &lv = levels[l];
        lv.scale = g.scale[l];
        lv.resolution = g.resolution[l];
        lv.offset = (uint32_t)offsets[l] + l * row_pad;
        lv.size = (uint32_t)(offsets[l + 1] - offsets[l]);
        const uint32_t r1 = align_corners ? lv.resolution : lv.resolution + 1u;
        // the loop of get_grid_index: a dimension enters the index only while stride <= hashmap_size (uint32 arithmetic)
        uint32_t stride = 1, st[3] = {0, 0, 0};
        for (uint32_t d = 0; d < D; ++d) {
            if (stride <= lv.size) { st[d] = stride; stride *= r1; }
        }
        lv.sy = st[1];
        lv.sz = st[2];
        const bool pow2 = lv.size != 0 && (lv.size & (lv.size - 1u)) == 0;
        const bool hashed = gridtype == 0 && stride > lv.size;
        // a non-power-of-two size needs no modulo iff every index stays below it: sum over kept dims of res*stride < stride_total <= size
        const bool exact = stride <= lv.size;
        lv.mask = pow2 ? lv.size - 1u : 0xFFFFFFFFu;
        lv.flags = (hashed || (!pow2 && (!exact || align_corners)) || lv.size >= (1u << 24) || r1 >= (1u << 12)) ? GFPP_LEVEL_SLOW : 0u;
    }
    return 0;
}

GFPP_API int gfpp_head_frame_begin(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                   const float *cond_feat, const float *ind_code, gfpp_stream_t stream) {
    if (!model || !ws || !rays_o || !rays_d) { set_error("gfpp_head_frame_begin: null argument"); return GFPP_EINVAL; }
    if (!ws->nears || !ws->fars || !ws->ray_state || !ws->counters || !ws->frame_consts || ws->N == 0) {
        set_error("gfpp_head_frame_begin: incomplete workspace");
        return GFPP_EINVAL;
    }
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_frame_begin, dim3(div_up(ws->N, kThreads)), dim3(kThreads), 0, st, rays_o, rays_d, ws->N, model->min_near, model->aabb[0],
                       model->aabb[1], model->aabb[2], model->aabb[3], model->aabb[4], model->aabb[5], ws->nears, ws->fars, ws->ray_state, ws->counters);
    int rc = check_launch("gfpp_head_frame_begin(init)");
    if (rc || !cond_feat) return rc;   // cond_feat == NULL: the caller folds later with gfpp_head_frame_fold (possibly on another stream)
    return gfpp_head_frame_fold(model, ws, cond_feat, ind_code, stream);
}

GFPP_API int gfpp_head_frame_fold(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *cond_feat, const float *ind_code,
                                  gfpp_stream_t stream) {
    if (!model || !ws || !cond_feat || !ws->frame_consts) { set_error("gfpp_head_frame_fold: null argument"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_fold_constants, dim3(2, 1), dim3(128), 0, (hipStream_t)stream, model->amb_w0_cond, cond_feat, model->cond_dim, model->col_w0_ind,
                       model->ind_dim ? ind_code : nullptr, model->ind_dim, ws->frame_consts, 0u);
    return check_launch("gfpp_head_frame_fold");
}

GFPP_API int gfpp_head_frame_fold_batch(const gfpp_head_model *model, const float *cond_feats, uint32_t cond_stride, const float *ind_code, float *frame_consts,
                                        uint32_t count, gfpp_stream_t stream) {
    if (count == 0) return 0;
    if (!model || !cond_feats || !frame_consts || count > 65535u) { set_error("gfpp_head_frame_fold_batch: null argument, or more than 65 535 frames"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_fold_constants, dim3(2, count), dim3(128), 0, (hipStream_t)stream, model->amb_w0_cond, cond_feats, model->cond_dim, model->col_w0_ind,
                       model->ind_dim ? ind_code : nullptr, model->ind_dim, frame_consts, cond_stride);
    return check_launch("gfpp_head_frame_fold_batch");
}

GFPP_API int gfpp_head_frame_march(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                   float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    if (!model || !ws || !rays_o || !rays_d) { set_error("gfpp_head_frame_march: null argument"); return GFPP_EINVAL; }
    if (max_steps == 0 || max_steps > (uint32_t)kMaxTrips) { set_error("gfpp_head_frame_march: max_steps must be in 1..%d", kMaxTrips); return GFPP_EUNSUPPORTED; }
    if (!grid_ok(model->pos_grid, 3) || !(grid_ok(model->amb_grid, 2) || grid_ok(model->amb_grid, 3))) {
        set_error("gfpp_head_frame_march: grids must be 16-level fp32, position D=3, ambient D in {2,3}");
        return GFPP_EUNSUPPORTED;
    }
    if (model->cascade < 1 || model->cascade > 8 || !model->density_bitfield || !ws->alive[0] || !ws->alive[1]) {
        set_error("gfpp_head_frame_march: bad model/workspace");
        return GFPP_EINVAL;
    }
    TripArgs a;
    a.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    a.pos = make_grid_dev(model->pos_grid);
    a.amb = make_grid_dev(model->amb_grid);
    a.w = HeadWeights{(const float4 *)model->amb_w0, (const float4 *)model->amb_w1, (const float4 *)model->sig_w0, (const float4 *)model->sig_w1,
                      (const float4 *)model->sig_w2_geo, (const float4 *)model->col_w0, model->amb_w2, model->sig_w2_sig, model->col_w1};
    a.bitfield = model->density_bitfield;
    a.rays_o = rays_o; a.rays_d = rays_d; a.fars = ws->fars;
    a.state = ws->ray_state;
    a.counters = ws->counters;
    a.gcounters = ws->gcounters ? ws->gcounters : ws->counters;
    a.N_global = ws->gcounters ? ws->N_global : ws->N;
    a.frame_consts = ws->frame_consts;
    a.T_thresh = T_thresh; a.density_scale = model->density_scale;
    a.N = ws->N; a.max_steps = max_steps;
    a.dbg_ambient = nullptr;
    // enough workgroups for the worst trip (ceil(N / floor(128/n_step)) tiles), capped: tiles are taken grid-stride
    uint32_t grid = div_up(ws->N, 120);
    if (grid > 4096u) grid = 4096u;
    const hipStream_t st = (hipStream_t)stream;
    for (uint32_t trip = 0; trip < max_steps; ++trip) {
        a.trip = trip;
        a.alive_in = ws->alive[trip & 1];
        a.alive_out = ws->alive[(trip + 1) & 1];
        if (model->amb_grid.D == 3) hipLaunchKernelGGL(k_head_trip<3>, dim3(grid), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL(k_head_trip<2>, dim3(grid), dim3(kThreads), 0, st, a);
        const int rc = check_launch("gfpp_head_frame_march");
        if (rc) return rc;
    }
    return 0;
}

GFPP_API int gfpp_head_frame_trips(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                   float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    if (!model || !ws || !rays_o || !rays_d) { set_error("gfpp_head_frame_trips: null argument"); return GFPP_EINVAL; }
    if (max_steps == 0 || max_steps > (uint32_t)kMaxTrips) { set_error("gfpp_head_frame_trips: max_steps must be in 1..%d", kMaxTrips); return GFPP_EUNSUPPORTED; }
    if (!grid_ok(model->pos_grid, 3) || !(grid_ok(model->amb_grid, 2) || grid_ok(model->amb_grid, 3))) {
        set_error("gfpp_head_frame_trips: grids must be 16-level fp32, position D=3, ambient D in {2,3}");
        return GFPP_EUNSUPPORTED;
    }
    if (!ws->alive[0] || !ws->alive[1] || !ws->sample_t || !ws->sample_cnt || ws->sample_stride < max_steps + 7u) {
        set_error("gfpp_head_frame_trips: the workspace needs alive[2], sample_t [N, sample_stride >= max_steps + 7] and sample_cnt [N] (gfpp_head_frame_premarch)");
        return GFPP_EINVAL;
    }
    TripArgs a;
    a.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    a.pos = make_grid_dev(model->pos_grid);
    a.amb = make_grid_dev(model->amb_grid);
    a.w = HeadWeights{(const float4 *)model->amb_w0, (const float4 *)model->amb_w1, (const float4 *)model->sig_w0, (const float4 *)model->sig_w1,
                      (const float4 *)model->sig_w2_geo, (const float4 *)model->col_w0, model->amb_w2, model->sig_w2_sig, model->col_w1};
    a.bitfield = model->density_bitfield;
    a.rays_o = rays_o; a.rays_d = rays_d; a.fars = ws->fars;
    a.state = ws->ray_state;
    a.counters = ws->counters;
    a.gcounters = ws->gcounters ? ws->gcounters : ws->counters;
    a.N_global = ws->gcounters ? ws->N_global : ws->N;
    a.frame_consts = ws->frame_consts;
    a.T_thresh = T_thresh; a.density_scale = model->density_scale;
    a.N = ws->N; a.max_steps = max_steps;
    a.sample_t = ws->sample_t; a.sample_cnt = ws->sample_cnt; a.sample_stride = ws->sample_stride;
    a.dbg_ambient = nullptr;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    const uint32_t grid = 2u * (uint32_t)cus;   // two resident workgroups per CU (register-bound), tiles are taken wave-stride
    const hipStream_t st = (hipStream_t)stream;
    const uint32_t first = ws->trip_count ? ws->trip_first : 0u;
    const uint32_t stop = ws->trip_count ? (first + ws->trip_count < max_steps ? first + ws->trip_count : max_steps) : max_steps;
    const uint32_t late_grid = grid < 64u ? grid : 64u;   // trips the caller expects to find nothing left: gfpp_frame_ws.full_grid_trips
    for (uint32_t trip = first; trip < stop; ++trip) {
        a.trip = trip;
        a.alive_in = ws->alive[trip & 1];
        a.alive_out = ws->alive[(trip + 1) & 1];
        const uint32_t g = ws->full_grid_trips && trip >= ws->full_grid_trips ? late_grid : grid;
        if (!tuning().trip_pool) {                          // gfpp_tuning.trip_pool = 0: the tile-per-wavefront kernel (A/B runs)
            if (model->amb_grid.D == 3) hipLaunchKernelGGL(k_head_trip_w<3>, dim3(g), dim3(kThreads), 0, st, a);
            else hipLaunchKernelGGL(k_head_trip_w<2>, dim3(g), dim3(kThreads), 0, st, a);
        } else {
            if (model->amb_grid.D == 3) hipLaunchKernelGGL(k_head_trip_wp<3>, dim3(g), dim3(kThreads), 0, st, a);
            else hipLaunchKernelGGL(k_head_trip_wp<2>, dim3(g), dim3(kThreads), 0, st, a);
        }
        const int rc = check_launch("gfpp_head_frame_trips");
        if (rc) return rc;
    }
    return 0;
}

GFPP_API int gfpp_head_eval_samples(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *positions, const float *directions, uint32_t M,
                                    float *sigma, float *color, float *ambient, gfpp_stream_t stream) {
    if (M == 0) return 0;
    if (!model || !ws || !positions || !directions || !sigma || !color || !ws->frame_consts) { set_error("gfpp_head_eval_samples: null argument"); return GFPP_EINVAL; }
    if (!grid_ok(model->pos_grid, 3) || !(grid_ok(model->amb_grid, 2) || grid_ok(model->amb_grid, 3))) {
        set_error("gfpp_head_eval_samples: grids must be 16-level fp32, position D=3, ambient D in {2,3}");
        return GFPP_EUNSUPPORTED;
    }
    EvalArgs e{};
    TripArgs &a = e.t;
    a.mp = make_march_params(model->bound, 0.0f, 16, model->cascade, model->grid_size);
    a.pos = make_grid_dev(model->pos_grid);
    a.amb = make_grid_dev(model->amb_grid);
    a.w = HeadWeights{(const float4 *)model->amb_w0, (const float4 *)model->amb_w1, (const float4 *)model->sig_w0, (const float4 *)model->sig_w1,
                      (const float4 *)model->sig_w2_geo, (const float4 *)model->col_w0, model->amb_w2, model->sig_w2_sig, model->col_w1};
    a.frame_consts = ws->frame_consts;
    a.density_scale = 1.0f;      // forward() returns the unscaled density; render() applies density_scale (renderer.py:376)
    e.positions = positions; e.directions = directions; e.sigma = sigma; e.color = color; e.ambient = ambient; e.M = M;
    uint32_t grid = div_up(M, 32u * (kThreads / 64));
    if (grid > 1024u) grid = 1024u;
    if (model->amb_grid.D == 3) hipLaunchKernelGGL(k_head_eval<3>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, e);
    else hipLaunchKernelGGL(k_head_eval<2>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, e);
    return check_launch("gfpp_head_eval_samples");
}

GFPP_API int gfpp_head_frame_finish(const gfpp_frame_ws *ws, const float *bg_color, float bg_scalar, float *out_image, float *out_depth,
                                    gfpp_stream_t stream) {
    if (!ws || !out_image || !out_depth || ws->N == 0) { set_error("gfpp_head_frame_finish: null argument"); return GFPP_EINVAL; }
    hipLaunchKernelGGL(k_head_finish, dim3(div_up(ws->N, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, ws->N, ws->nears, ws->fars,
                       ws->ray_state, bg_color, bg_scalar, out_image, out_depth);
    return check_launch("gfpp_head_frame_finish");
}
