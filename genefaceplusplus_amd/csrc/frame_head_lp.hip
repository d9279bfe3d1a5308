// frame_head_lp.hip -- the fused head trip kernel with 16-bit MFMA operands (f16 or bf16 inputs, fp32 accumulation).
//
// Same loop semantics as frame_head.hip (renderer.py:354-384, one launch per loop iteration, loop state on the device), but
// the arithmetic of the five wide layers runs on v_mfma_f32_32x32x16_{f16,bf16} -- 16x the rate of the exact-fp32 MFMA -- which
// moves the bound of the kernel from the matrix pipe to the hash-grid gathers.  This is the mode that corresponds to the
// reference's own inference precision: genefacepp_infer.py renders under torch.autocast(fp16), i.e. nn.Linear in half with
// fp32 accumulation (and half activations BETWEEN layers, which this kernel avoids: accumulators stay fp32 in registers and
// are rounded once, when they become the next layer's operand).
//
// Differences of structure from the fp32 kernel, all consequences of the 16x shorter MFMA phase:
//   * The weights (124 KB in 16 bit) are RESIDENT IN LDS for the whole launch: one 512-thread workgroup per CU copies
//     them once, then every wavefront streams its A operands with ds_read_b128 (1 KB per MFMA; from L2 the matrix pipe
//     would starve at 64 B/clk/CU).  The three skinny output layers (3+1+3 rows, fp32, VALU) and the per-frame folded
//     biases sit in LDS too.
//   * sigma_net's last layer (geo_feat rows, no activation) and color_net's first layer are ONE matrix:
//     relu(C0_sh sh + C0_geo (S2_geo h) + b) = relu(C0_sh sh + (C0_geo S2_geo) h + b); the product is formed on the host in
//     fp64.  One 128x128 layer less per sample (128 768 FLOP instead of 161 536) and one rounding of activations less.
//   * Wavefronts are autonomous: a wavefront marches 64 / 32 / 16 rays (<= 128 sample slots), compacts, evaluates 32 samples
//     per pass, composites and appends its survivors -- no workgroup barrier after the weight copy, so one wavefront's
//     gathers overlap another's MFMAs on the same SIMD.
#include <cstdlib>

#include "head_eval_device.h"
#include "head_eval_f32_device.h"
#include "lp_mfma_device.h"

// Round 4 (measured on the GPU, then adopted; the A/B numbers are in docs/LAB_NOTEBOOK.md): the two hash grids are read as 16-bit CORNER-BLOCK tables
// (grid_device.h: one 16-byte gather per z plane of a level, all eight levels of a lane in flight), the three skinny output layers run as MFMA chains on one
// gathered tile (skinny_mfma_fused), and a block's ray directions are requested before its first grid lookup.  Head pass 0.290 -> 0.240 ms at 512^2.

namespace gfpp {

// 512 threads = 2 wavefronts per SIMD with up to 256 registers each.  Measured alternatives: 256 threads (1 per SIMD, no spills) is 1.49x
// slower; 768 threads (3 per SIMD, 168 registers) spills ~140 registers and is 1.8x slower.
constexpr int kLpThreads = 512;
constexpr int kLpWaves = kLpThreads / 64;
constexpr int kLpSlots = 128;   // sample slots of one wavefront tile
constexpr int kLpRays = 64;     // rays of one wavefront tile (at most)
// K = 16 steps of the five MFMA layers, in LDS order
constexpr int kStepAmb0 = 0;    // 2 steps: 32 position features
constexpr int kStepAmb1 = 2;    // 8 steps: 128 activations
constexpr int kStepSig0 = 10;   // 4 steps: 32 position + 32 ambient features
constexpr int kStepSig1 = 14;   // 8 steps
constexpr int kStepCol = 22;    // 9 steps: 16 SH + 128 activations (merged geo layer)
constexpr int kLpSteps = 31;
constexpr int kLpWeightChunks = kLpSteps * 4 * 64;   // 16-byte chunks: [step][tile m][lane]
constexpr int kSkinnyRows = 7;   // ambient_net.2 (3, padded), sigma_net.2 row 0, color_net.1 (3): rows 0-2, 3, 4-6
constexpr int kSkinnyWords = 2 * kSkinnyRows * 32;   // [half][row][32 pairs of 16-bit weights]
constexpr int kSkinnyPitch = 9;                       // 16-byte chunks per (half, row) in LDS: 8 steps + 1 of padding (rows 128 B apart would all start in the same banks)
constexpr int kSkinnyLds = 2 * kSkinnyRows * kSkinnyPitch * 4;

struct LpWaveTile {
    float px[kLpSlots], py[kLpSlots], pz[kLpSlots];   // sample position; after evaluation: sigma, r, g of the slot
    float cb[kLpSlots], dt[kLpSlots], tend[kLpSlots]; // b of the slot; step length; t after the sample
    uint32_t ray[kLpRays];                            // ray id by local ray (direction is re-read from rays_d for the SH basis)
    uint8_t order[kLpSlots];                          // compact index -> slot
    __device__ __forceinline__ uint32_t ray_of(uint32_t local) const { return ray[local]; }
};

// The sample pool of one workgroup round (k_head_trip_pool): the eight wavefront tiles side by side, compacted TOGETHER, so that the
// 32-sample blocks are full and any wavefront can evaluate any of them.  Same member names as LpWaveTile: evaluate_block_lp works on either.
constexpr int kPoolSlots = kLpWaves * kLpSlots;   // 1024
constexpr uint32_t kNoRay = 0xFFu;
struct LpPool {
    float px[kPoolSlots], py[kPoolSlots], pz[kPoolSlots];
    float cb[kPoolSlots], dt[kPoolSlots], tend[kPoolSlots];
    uint32_t ray[kPoolSlots];        // ray id by local ray (wavefront w owns the local rays [w * rw, (w + 1) * rw))
    uint16_t order[kPoolSlots];      // compact index -> slot, over the whole workgroup
    uint8_t cnt[kPoolSlots];         // samples the local ray takes in this trip; kNoRay: no such ray
    uint32_t wave_valid[kLpWaves];   // occupied samples per wavefront tile
    uint32_t wave_surv[kLpWaves];    // surviving rays per wavefront tile
    uint32_t out_base;               // where the workgroup's survivors go in the next trip's list
    __device__ __forceinline__ uint32_t ray_of(uint32_t local) const { return ray[local]; }
};

// The pool of the persistent launch (k_head_frame_persist): the workgroup's alive list lives here too, dt / t_end are recomputed from t0 when
// compositing (same expressions, same bits) to make room for it.
constexpr uint32_t kPTile = 8;            // rays of one ownership tile (8 consecutive rays: 32-ray tiles left the busiest workgroup 22-44 % above the mean, 8-ray tiles 6-10 %)
constexpr uint32_t kPTilesPerSub = 512u / kPTile;          // candidate tiles of one 512-thread pass of the ingest step
constexpr uint32_t kPTilesPerStep = 2u * kPTilesPerSub;
constexpr uint32_t kPRayBits = 22;        // alive entry: ray id | samples consumed << 22 | samples the ray owns << 27
constexpr uint32_t kPRayMask = (1u << kPRayBits) - 1u;
constexpr uint32_t kPMaxFrames = 4;       // frames one launch can render (gfpp_frame_ws.n_frames): rays are numbered frame * N + ray, (frames * N) <= 2^kPRayBits
struct LpPoolP {
    float bias_more[kPMaxFrames - 1][256];   // folded constants of the frames 1.. of a frame group: FIRST member, so that frame f's set is LpShared::bias + 256 f
    float px[kPoolSlots], py[kPoolSlots], pz[kPoolSlots], cb[kPoolSlots];   // sample position; after evaluation: sigma, r, g, b of the slot
    float t0[kPoolSlots];                                                    // t of the sample
    uint32_t alive[kPoolSlots];      // the workgroup's alive list; entry i is also local ray i of the round (ray id = low kPRayBits bits)
    uint16_t order[kPoolSlots];      // compact index -> slot
    uint8_t cnt[kPoolSlots];         // samples the local ray takes in this round; kNoRay: no such ray
    uint32_t wave_valid[kLpWaves], wave_surv[kLpWaves];
    uint32_t tile_cnt[kPTilesPerStep];   // ingest: occupied rays | empty rays << 16 of each candidate tile
    uint32_t hist[kPMaxFrames][32];  // per frame: rays by the sample index their compositing ends at
    // (the list is compacted only after a round's evaluation and compositing: during both, entry `local` is the round's local ray `local`)
    __device__ __forceinline__ uint32_t ray_of(uint32_t local) const { return alive[local] & kPRayMask; }
};

struct LpShared {
    uint4 w[kLpWeightChunks];      // 126 976 B
    uint32_t skinny[kSkinnyLds];   //   1 792 B  skinny output rows as 16-bit pairs, in the operand order of relu_pack_step
    gfpp_grid_level lv[2][16];     //   1 024 B  level descriptors of the position / ambient grid
    float bias[256];               //   1 024 B
    union {
        LpWaveTile tile[kLpWaves]; //  27 648 B  one tile per wavefront (k_head_trip_lp, k_head_eval_lp)
        LpPool pool;               //  31 776 B  one pool per workgroup (k_head_trip_pool)
        LpPoolP poolp;             //  31 808 B  pool + alive list (+ the further frames' constants) of the persistent launch (k_head_frame_persist)
    };
};
static_assert(sizeof(LpShared) <= 163840, "one workgroup per CU: everything must fit the 160 KiB LDS");
static_assert(offsetof(LpShared, poolp) == offsetof(LpShared, bias) + 256 * sizeof(float) && offsetof(LpPoolP, bias_more) == 0,
              "frame f's constants are addressed as bias + 256 f");

struct LpGrid {
    const gfpp_grid_level *levels;   // [16] device memory, read with scalar loads
    const void *table;               // fast levels only: the 16-bit corner-block copy; otherwise (any_slow, SLOW instantiations) the fp32 table
    uint32_t gridtype, interp, align_corners, any_slow;
};

struct LpTripArgs {
    MarchParams mp;
    LpGrid pos, amb;
    const uint4 *w16;
    const uint32_t *skinny16;
    const float *rays_o, *rays_d;
    const float *sample_t;        // [N, sample_stride]: t of every occupied sample of the ray, in march order (k_premarch)
    const uint32_t *sample_cnt;   // [N]: how many of them exist (capped at max_steps + 7, more can never be consumed)
    uint32_t sample_stride;
    float *state;                 // [N, kRayRec] ray records (march_device.h); word 5 = how many samples the previous trips used up (the role of rays_t)
    int32_t *alive[2];            // ping-pong survivor lists: trip k reads alive[k & 1], writes alive[(k + 1) & 1]
    int32_t *counters;
    const int32_t *gcounters;     // frame-wide alive counts per trip (== counters unless this launch renders one ray tile of a frame shared between GPUs)
    uint32_t N_global;            // rays of the whole frame (== N on one GPU)
    int32_t *sync;                // barrier word of multi-trip launches (counters[127], zeroed by k_frame_begin)
    int32_t *timeouts;            // optional sticky word (gfpp_frame_ws.timeouts): +1 per barrier that timed out, reset by nobody but the host
    const float *frame_consts;
    float T_thresh, density_scale;
    uint32_t N, trip, trip_end, max_steps;   // this launch runs the trips [trip, trip_end)
    // persistent launch (k_head_frame_persist)
    uint32_t consts_stride;             // floats between the frames' folded constants (256 when they are a [n_frames, 256] array)
    uint32_t n_frames, tiles_per_frame; // a frame group: n_frames (<= kPMaxFrames) frames of N rays behind each other in every per-ray array, ceil(N / kPTile) tiles each
    int32_t *budget;                    // counters + 128 of the (first) frame: histogram [32], evaluated samples, rounds; frame f's: + f * kCounterWords
    float *snaps;                       // [N, 7, 5] ray state after max_steps .. max_steps + 6 composited samples
    uint32_t n_tiles, tile_mult;        // ownership tiles of kPTile rays; tile of slot q = (q * tile_mult) % n_tiles
    uint32_t xcd_cols8;                 // != 0: XCD-local ownership (gfpp_frame_ws.row_rays): tile columns per image row / 8; tile_mult then permutes the n_tiles / 8 tiles of ONE XCD
    uint32_t step_caps;                 // 4 bits per round (rounds >= 7 use the last): upper bound of the local n_step
    uint32_t spin_limit;                // multi-trip launches: polls of the barrier word before a workgroup gives up and poisons it (GFPP_BARRIER_SPINS, tests)
    float *dbg_ambient;                 // per-sample evaluation entry only (k_head_eval_lp): tanh(ambient_net) of compact sample c -> [c * AMB_D ...]
    unsigned long long *phase_cycles;   // optional [trips][8] (k_head_trip_pool<PROF>): cycles summed over wavefronts by phase, see there
};

// While a wavefront issues a layer's MFMAs it asks for issue priority (s_setprio): of the two wavefronts of a SIMD the one on the matrix pipe goes first and the
// other fills the remaining issue slots with its vector work, instead of both alternating on every instruction (measured with the fused layers below, same box,
// 512^2 bf16: 4 473-4 480 -> 4 496-4 522 frames/s).
constexpr int kMfmaPrio = 2;
template <typename H, int NS>
__device__ __forceinline__ void mfma_steps(v16f (&acc)[4], const uint4 *w, int step0, const typename LpTraits<H>::vec (&b)[NS], int lane) {
    __builtin_amdgcn_s_setprio(kMfmaPrio);
    mfma_layer_lds<H, NS, 4>(acc, reinterpret_cast<const typename LpTraits<H>::vec *>(w) + step0 * 256, b, lane);
    __builtin_amdgcn_s_setprio(0);
}
// a 128 -> 128 layer on the previous layer's accumulators, followed by the skinny rows on ITS accumulators: every activation packed under the MFMAs that follow
// (lp_mfma_device.h: mfma_layer_lds_fused; skinny_mfma_fused below).  keep: the packed hidden state, for a second consumer.
template <typename H>
__device__ __forceinline__ void hidden_layer_and_rows(const LpShared &sh, int step0, const v16f (&prev)[4], int lane, v16f &rows, typename LpTraits<H>::vec (*keep)[8]);

// Transcendentals of the 16-bit kernels on the hardware's exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm's range-reduced forms and IEEE
// divisions: ~100 of a block's ~1 380 vector instructions.  Their inputs carry 8-11 significant bits (16-bit MFMA operands): the forms differ from libm's by a few
// ulp of fp32, four orders of magnitude below that.  (The exact-fp32 kernels keep expf / tanhf.)
__device__ __forceinline__ float lp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float lp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + lp_exp(-x)); }
__device__ __forceinline__ float lp_tanh(float x) {
    // 1 - 2 / (e^{2x} + 1); |x| clamped where the result is +-1 to the last bit, so that e^{2x} never overflows
    const float xc = fminf(fmaxf(x, -10.0f), 10.0f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(lp_exp(2.0f * xc) + 1.0f);
}

// The operand type of ambient_net inside a precision mode (round 5).  ambient_net's output is not a colour or a density but a COORDINATE of the second hash grid:
// 8-bit significands (bf16) displace it by up to 5e-3 of the unit cube -- five cells of the finest level, whose features then belong to other cells -- and that, not
// the radiance layers, is what held bf16 frames at 43-47 dB on the non-convex scenes (tools/lp_emulate.py renders them on the CPU with the rounding points moved
// layer group by layer group: everything bf16 46-47 dB, ambient_net alone on 11-bit operands 57-59 dB, only the OTHER layers on 11 bits 47 dB).  So the bf16 mode
// multiplies ambient_net's two wide layers and three rows as f16 (same MFMA rate; its inputs are the grid features, read from f16 tables in every 16-bit mode,
// and a tanh follows) and everything behind the ambient grid as bf16.  The weight image carries the steps 0..9 and the skinny rows 0..2 in this type.
template <typename H>
struct LpAmbient {
    typedef H type;
};
template <>
struct LpAmbient<__bf16> {
    typedef _Float16 type;
};

// This lane's half of a 16-level, 2-channel grid encoding, packed as MFMA operands.  Half-wave `hi` takes the levels hi, hi+2, ..; value k (= 8 s + e) of the
// lane is level 2 (k/2) + hi, channel k % 2.
//   SLOW = false (every level's index is linear modulo a power of two or provably in range -- the tiled grids of every shipped model): the table is the 16-bit
//     CORNER-BLOCK copy (grid_device.h: level_block_issue / level_block_finish) and the lookup is straight-line code.  The gathers of ALL eight levels of the
//     lane are issued before the first is interpolated -- two memory round trips per block and grid.  Measured at 512^2 bf16 (head pass, same box): fp32
//     tables with two levels in flight 0.290 ms, block tables 0.263, with eight levels in flight 0.249 (with fp32 tables eight levels spill 7 registers).
//   SLOW = true (hash-addressed or true-modulo levels present): the generic lookup on the fp32 table, level by level.
template <int D, typename H, bool SLOW>
__device__ __forceinline__ void encode_half_lp(const float (&u)[D], const LpGrid &g, const gfpp_grid_level *lvl, int hi, bool valid,
                                               typename LpTraits<H>::vec (&b)[2]) {
    bool ok = valid;
    float uc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ok = ok && !(u[d] < 0.0f || u[d] > 1.0f);
        uc[d] = fminf(fmaxf(u[d], 0.0f), 1.0f);
    }
    const bool smooth = g.interp == 1, ac = g.align_corners != 0;
    float f[16];
    if constexpr (SLOW) {
        // (interpolation type as a compile-time constant of two copies of the loop: with the run-time select the smoothstep constants are kept in registers
        // across the whole block loop and the kernel spills)
        auto levels = [&](auto smooth_tag) {
            constexpr uint32_t INTERP = decltype(smooth_tag)::value ? 1u : 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float o[2];
                const gfpp_grid_level lv = lvl[2 * i + hi];      // (the LDS copy, like the fast path: a 32-bit address per lane instead of a 64-bit global one per level)
                grid_level_lookup<D, 2, float, true>(uc, reinterpret_cast<const float *>(g.table), lv.offset, lv.size, lv.scale, lv.resolution, g.gridtype, ac, INTERP, o);
                f[2 * i] = ok ? o[0] : 0.0f;
                f[2 * i + 1] = ok ? o[1] : 0.0f;
            }
        };
        if (smooth) levels(std::true_type{}); else levels(std::false_type{});
    } else {
        // all eight descriptors of this half-wave first (LDS, two distinct addresses per wavefront), then the straight-line lookups: left to the
        // compiler the descriptor of level i is read right before its use, one exposed LDS round trip per level in front of the gathers
        LevelU lvs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const gfpp_grid_level &d = lvl[2 * i + hi];
            lvs[i] = LevelU{d.scale, d.sy, d.sz, d.mask, d.offset};
        }
        __builtin_amdgcn_sched_barrier(0);
        // the interpolation type is wave-uniform: one branch around two specialised bodies instead of a smoothstep polynomial + select per coordinate
        auto levels = [&](auto smooth_tag) {
            constexpr bool SM = decltype(smooth_tag)::value;
            BlockGathers<D> lg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) level_block_issue<D, SM>(uc, g.table, lvs[k], ac, lg[k]);
            __builtin_amdgcn_sched_barrier(0);             // all of the lane's gathers are in flight before the first one is consumed
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float o[2];
                level_block_finish<D>(lg[k], o);
                f[2 * k] = o[0];
                f[2 * k + 1] = o[1];
            }
        };
        if (smooth) levels(std::true_type{}); else levels(std::false_type{});
    }
    // out-of-range / padding samples get zero features (gridencoder.cu:110-135): selected on the 8 packed operand words, not on the 16 floats
    // (the generic path above already zeroed its own)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) b[s][e] = (H)f[8 * s + e];
        if constexpr (!SLOW) {
            const u32x4 w = __builtin_bit_cast(u32x4, b[s]);
            b[s] = __builtin_bit_cast(typename LpTraits<H>::vec, ok ? w : u32x4{0u, 0u, 0u, 0u});
        }
    }
}

// The skinny layers (ambient_net.2: 3 rows, sigma_net.2 row 0, color_net.1: 3 rows) on the matrix pipe.  A row of the skinny image is already an A operand: its
// chunk s holds the weights of the eight activations lane-half h supplies in step s.  One 32-row tile is gathered from the image with a per-lane row map,
//     tile row  0 1 2 3 | 4 5 6 7 | 8 9 10 11 | 12 13 14 15 | 16..31 = rows 0..15 again (never read)
//     image row 0 1 2 3 | 0 1 2 3 | 4 5  6  6 |  4  5  6  6
// so that after eight steps lane (j, h) holds in acc[0..2] the three ambient rows, in acc[3] the density row and in acc[4..6] the three colour rows of
// sample j, in BOTH half-waves (tile rows r and r + 4 are the same image row: no exchange between the halves).  All three layers use the same tile -- each
// multiplies it with its own activations and reads its own rows; the other rows' products are finite and ignored.  8 MFMAs + 8 LDS reads per layer; as packed
// dot products on the VALU (round 3) a layer was 32 v_dot2c + 8 LDS reads per ROW: 224 v_dot2c per block, head pass 0.290 -> 0.282 ms without them.
// The row map and register indices are pinned by the CPU lane emulation in tests/test_packing_cpu.py.
__device__ __forceinline__ uint32_t skinny_tile_chunk(int lane) {
    const uint32_t i = (uint32_t)lane & 15u, h = (uint32_t)lane >> 5;
    const uint32_t low = i & 3u, row = (i & 8u) ? 4u + (low < 2u ? low : 2u) : low;
    static_assert(kSkinnyRows == 7 && kSkinnyPitch == 9, "shift-and-add forms below");
    const uint32_t q = (h << 3) - h + row;          // (half, row)
    return (q << 3) + q;
}
// skinny_mfma with its operands packed from the preceding layer's accumulators UNDER its own MFMAs (lp_mfma_device.h: mfma_layer_lds_fused): the operand of step
// s + 1 -- eight vector instructions, the duration of one MFMA -- behind the MFMA of step s.  keep: the packed operands for a second consumer (the colour layer).
template <typename H>
__device__ __forceinline__ void skinny_mfma_fused(const uint32_t *__restrict__ image, const v16f (&prev)[4], int lane, v16f &acc, typename LpTraits<H>::vec (*keep)[8]) {
    typedef typename LpTraits<H>::vec vec;
    const vec *p = reinterpret_cast<const vec *>(image) + skinny_tile_chunk(lane);
    vec a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = p[s];
    v16f even, odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) even[r] = odd[r] = 0.0f;
    vec b = relu_pack_step<H>(prev, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (keep) (*keep)[s] = b;
        if (s & 1) odd = LpTraits<H>::mfma(a[s], b, odd);
        else even = LpTraits<H>::mfma(a[s], b, even);
        if (s + 1 < 8) {
            b = relu_pack_step<H>(prev, s + 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = even[r] + odd[r];
}

template <typename H>
__device__ __forceinline__ void hidden_layer_and_rows(const LpShared &sh, int step0, const v16f (&prev)[4], int lane, v16f &rows, typename LpTraits<H>::vec (*keep)[8]) {
    v16f acc[4];
    __builtin_amdgcn_s_setprio(kMfmaPrio);
    mfma_layer_lds_fused<H>(acc, reinterpret_cast<const typename LpTraits<H>::vec *>(sh.w) + step0 * 256, prev, lane, nullptr);
    skinny_mfma_fused<H>(sh.skinny, acc, lane, rows, keep);
    __builtin_amdgcn_s_setprio(0);
}

// ambient_net on one 32-sample block: pos operand -> ambient coordinates (pre-tanh), replicated in both half-waves
template <int AMB_D, typename H>
__device__ __forceinline__ void ambient_block(const LpShared &sh, const float *__restrict__ bias, const typename LpTraits<H>::vec (&bpos)[2], int lane, int hi,
                                              float (&amb)[AMB_D]) {
    v16f acc[4], sk;
    load_bias(acc, bias, hi);
    mfma_steps<H, 2>(acc, sh.w, kStepAmb0, bpos, lane);
    hidden_layer_and_rows<H>(sh, kStepAmb1, acc, lane, sk, nullptr);
#pragma unroll
    for (int d = 0; d < AMB_D; ++d) amb[d] = sk[d];
}

// sigma_net + colour net on one 32-sample block; results go to the slots of the block's samples.  `bias`: the sample's frame constants (LDS);
// `dir`: the direction of the sample's ray (requested by the caller before the block's first grid lookup: in front of the SH basis the load was one exposed L2
// round trip per block, the MFMA layers' sched_barriers keep it wherever it is put)
template <typename H, typename Tile>
__device__ __forceinline__ void radiance_block(const LpTripArgs &a, const LpShared &sh, Tile &wt, const float *__restrict__ bias,
                                               const typename LpTraits<H>::vec (&bpos)[2], const typename LpTraits<H>::vec (&bamb)[2], uint32_t c, uint32_t n_valid,
                                               int lane, int hi, const float (&dir)[3]) {
    typedef typename LpTraits<H>::vec vec;
    const bool valid = c < n_valid;
    const uint32_t slot = valid ? wt.order[c] : 0u;
    v16f acc[4];
    vec bh[8];
    zero_acc(acc);
    mfma_steps<H, 2>(acc, sh.w, kStepSig0, bpos, lane);
    mfma_steps<H, 2>(acc, sh.w, kStepSig0 + 2, bamb, lane);
    float logit;
    {
        v16f sk;
        hidden_layer_and_rows<H>(sh, kStepSig1, acc, lane, sk, &bh);   // the hidden state feeds both the density row and the (merged) colour layer
        logit = sk[3];
    }
    const float sigma = a.density_scale * lp_exp(logit);
    {
        vec bcol[9];
        float shv[16];
        sh_basis4(dir[0], dir[1], dir[2], shv);
#pragma unroll
        for (int e = 0; e < 8; ++e) bcol[0][e] = (H)(hi ? shv[8 + e] : shv[e]);
#pragma unroll
        for (int s = 0; s < 8; ++s) bcol[1 + s] = bh[s];
        load_bias(acc, bias + 128, hi);
        mfma_steps<H, 9>(acc, sh.w, kStepCol, bcol, lane);
    }
    float rgb[3];
    {
        v16f sk;
        __builtin_amdgcn_s_setprio(kMfmaPrio);
        skinny_mfma_fused<H>(sh.skinny, acc, lane, sk, nullptr);
        __builtin_amdgcn_s_setprio(0);
        rgb[0] = sk[4]; rgb[1] = sk[5]; rgb[2] = sk[6];
    }
    if (valid && hi == 0) {
        wt.px[slot] = sigma;
        wt.py[slot] = lp_sigmoid(rgb[0]);
        wt.pz[slot] = lp_sigmoid(rgb[1]);
        wt.cb[slot] = lp_sigmoid(rgb[2]);
    }
}

// RADNeRF.forward for the 32 occupied samples [first, first+32) of a tile (one wavefront's LpWaveTile or a workgroup's pool).
// MF (frame groups of the persistent launch): the block's samples may belong to different frames; each takes the folded constants of ITS frame -- the
// biases enter as the accumulators' initial values, per column (= sample) of the block, so nothing else in the block depends on the frame.
template <int AMB_D, typename H, bool SLOW, bool DBG = false, bool MF = false, typename Tile>
__device__ __forceinline__ void evaluate_block_lp(const LpTripArgs &a, const LpShared &sh, Tile &wt, uint32_t first, uint32_t n_valid,
                                                  uint32_t n_step, int lane_in) {
    typedef typename LpTraits<H>::vec vec;
    int lane = lane_in;
    // launder the lane id: keeps tile-loop-invariant per-lane LDS addresses from being hoisted out of the tile loop and spilled
    asm volatile("" : "+v"(lane));
    const int j = lane & 31, hi = lane >> 5;
    const uint32_t c = first + (uint32_t)j;
    const bool valid = c < n_valid;
    const uint32_t slot = valid ? wt.order[c] : 0u;
    // the descriptor tables are addressed through the laundered lane id too (hoisting 32 descriptors out of the tile loop would spill)
    const gfpp_grid_level *lv_pos = &sh.lv[0][0] + (lane & 0), *lv_amb = &sh.lv[1][0] + (lane & 0);
    const uint32_t ray = wt.ray_of(slot / n_step);
    const float *bias = sh.bias;        // one frame per launch: the frame's constants; a frame group: the set of the sample's frame (rays are numbered frame * N + ray)
    if constexpr (MF) bias += 256u * ((ray >= a.N ? 1u : 0u) + (ray >= 2u * a.N ? 1u : 0u) + (ray >= 3u * a.N ? 1u : 0u));

    typedef typename LpAmbient<H>::type HA;        // ambient_net's operand type (f16 in the bf16 mode, see LpAmbient)
    vec bpos[2], bamb[2];
    float dir[3];
    {
        const float *dp = a.rays_d + 3ull * ray;
        dir[0] = dp[0]; dir[1] = dp[1]; dir[2] = dp[2];
    }
    {
        float amb[AMB_D], ua[AMB_D], u3[3];
        const float b2 = 2.0f * a.mp.bound;
        u3[0] = (wt.px[slot] + a.mp.bound) / b2;
        u3[1] = (wt.py[slot] + a.mp.bound) / b2;
        u3[2] = (wt.pz[slot] + a.mp.bound) / b2;
        if constexpr (std::is_same<HA, H>::value) {
            encode_half_lp<3, H, SLOW>(u3, a.pos, lv_pos, hi, valid, bpos);
            ambient_block<AMB_D, H>(sh, bias, bpos, lane, hi, amb);
        } else {
            // the position features as ambient_net's operands (HA) first; sigma_net's copy (H) is made from them behind ambient_net: both copies alive through its
            // layers cost the 8 registers that made the bf16 instantiation spill (the features come from 16-bit tables: HA = f16 holds them to 11 bits, H = bf16 keeps 8)
            typename LpTraits<HA>::vec bpos_a[2];
            encode_half_lp<3, HA, SLOW>(u3, a.pos, lv_pos, hi, valid, bpos_a);
            ambient_block<AMB_D, HA>(sh, bias, bpos_a, lane, hi, amb);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) bpos[s][e] = (H)(float)bpos_a[s][e];
        }
#pragma unroll
        for (int d = 0; d < AMB_D; ++d) {
            const float th = lp_tanh(amb[d]);
            if constexpr (DBG) { if (a.dbg_ambient && valid && hi == 0) a.dbg_ambient[(size_t)c * AMB_D + d] = th; }
            ua[d] = (th + 1.0f) / 2.0f;
        }
        encode_half_lp<AMB_D, H, SLOW>(ua, a.amb, lv_amb, hi, valid, bamb);
    }
    radiance_block<H>(a, sh, wt, bias, bpos, bamb, c, n_valid, lane, hi, dir);
}

// Counters are written by other workgroups (atomics at the device's coherence point) while this kernel runs when it covers several
// trips: read them there, not through the per-XCD caches.
__device__ __forceinline__ uint32_t counter_load(const int32_t *p) {
    return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Device-wide barrier between two trips of one launch.  Every workgroup of the launch is resident (one per CU, the launch never has
// more workgroups than the device has CUs), so spinning cannot starve a workgroup that has not started; kernels of other streams only
// delay it.  Release/acquire at agent scope write back and invalidate the per-XCD L2s, which is what makes one trip's ray state and
// survivor list visible to whichever workgroup picks the ray up in the next trip.  The spin is bounded: on a timeout the barrier word
// is poisoned (negative, so later barriers fall through and the host can see it, FramePipeline.trip_counters) instead of hanging the GPU.
__device__ __forceinline__ void grid_barrier(int32_t *bar, uint32_t target, uint32_t spin_limit, int32_t *timeouts) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (counter_load(bar) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_limit) {
                __hip_atomic_store(bar, (int32_t)0x80000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (timeouts) __hip_atomic_fetch_add(timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // survives the next frame's counter reset
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// Weights, skinny rows, folded biases and level descriptors -> LDS (once per launch; the caller synchronises the workgroup afterwards).
__device__ __forceinline__ void lp_fill_shared(LpShared &sh, const LpTripArgs &a, int tid, int lane) {
    // 124 KB of fragments: direct global -> LDS transfers (no register round trip, all 16 requests of a thread in flight at once); the
    // LDS address of such a load is wave-uniform base + lane x 16, which is exactly the fragment layout
    static_assert(kLpWeightChunks % 64 == 0, "whole wavefront rows");
    for (int i = tid; i < kLpWeightChunks; i += kLpThreads)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.w16 + i),
                                         (__attribute__((address_space(3))) void *)(&sh.w[i - lane]), 16, 0, 0);
    for (int i = tid; i < kSkinnyWords; i += kLpThreads) sh.skinny[(i >> 5) * (4 * kSkinnyPitch) + (i & 31)] = a.skinny16[i];   // rows padded to kSkinnyPitch chunks
    for (int i = tid; i < 256; i += kLpThreads) sh.bias[i] = a.frame_consts[i];
    for (int k = tid; k < 256; k += kLpThreads) {   // 2 x 16 descriptors x 8 dwords
        const int which = k >> 7, w = k & 127;
        reinterpret_cast<uint32_t *>(&sh.lv[which][0])[w] = reinterpret_cast<const uint32_t *>(which ? a.amb.levels : a.pos.levels)[w];
    }
}

// One launch runs the trips [a.trip, a.trip_end) (renderer.py:352-384: one loop iteration each).  The host issues the first few trips as
// separate launches (no barrier needed: the stream orders them) and the rest -- which most frames never reach -- as ONE launch that
// finds its first counter at zero and returns, instead of ten empty launches.
// The samples of a whole workgroup are pooled (rounds 1-3 also kept a kernel with one tile per wavefront, k_head_trip_lp: a tile holds 0..4 blocks and the
// workgroup keeps its CU until its slowest wavefront is done -- at 512x512 a trip carried ~2.4 blocks per wavefront on average but lasted 4 block
// times; removed in round 4).  Every workgroup takes an equal share of the alive rays (wavefront tiles dealt out through a
// multiplicative permutation, so that in trip 0 no workgroup gets only the empty image border), compacts the occupied samples of all
// eight tiles into one list -- every block but the last is full -- and deals the blocks out to its wavefronts round-robin: a round lasts
// ceil(blocks / 8) block times.  Per sample nothing changes (same evaluate_block_lp, same compositing order along a ray).
// PROF: phase_cycles[trip][0..4] = shader cycles summed over wavefronts: gather samples | the two compaction barriers | evaluate | wait for the
// workgroup's last block | composite;  [5] = wavefront rounds, [6] = blocks evaluated, [7] = the longest workgroup round of the trip (cycles)
template <int AMB_D, typename H, bool SLOW, bool PROF = false>
__global__ __launch_bounds__(kLpThreads, kLpThreads / 256) void k_head_trip_pool(LpTripArgs a) {
    __shared__ LpShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool prof = PROF;
    bool weights_resident = false;
    uint32_t barriers = 0;
    uint32_t step_before = 0;
    // the alive counts of every trip up to this launch's first are final: ONE parallel fetch (lane k takes trip k; trips < 64), not a loop of
    // uncached loads that costs a memory latency per earlier trip
    const uint32_t fetched = (uint32_t)lane <= a.trip ? counter_load(a.gcounters + lane) : 0u;
    for (uint32_t k = 0; k < a.trip; ++k) {
        const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)fetched, (int)k);
        if (na == 0) return;
        uint32_t ns = a.N_global / na;
        ns = ns < 1u ? 1u : (ns > 8u ? 8u : ns);
        step_before += ns;
    }
    for (uint32_t trip = a.trip; trip < a.trip_end; ++trip) {
        const uint32_t n_alive_frame = trip == a.trip ? (uint32_t)__builtin_amdgcn_readlane((int)fetched, (int)a.trip) : counter_load(a.gcounters + trip);
        if (n_alive_frame == 0 || step_before >= a.max_steps) return;
        uint32_t n_step = a.N_global / n_alive_frame;
        n_step = n_step < 1u ? 1u : (n_step > 8u ? 8u : n_step);
        step_before += n_step;
        const uint32_t used = step_before - n_step;
        const uint32_t n_alive = a.gcounters == a.counters ? n_alive_frame : counter_load(a.counters + trip);
        if (n_alive == 0) return;
        const int32_t *alive_in = a.alive[trip & 1];
        int32_t *alive_out = a.alive[(trip + 1) & 1];

        // equal shares: `rounds` rounds of gridDim.x * 8 wavefront tiles of rw rays (rw * n_step <= 128 slots per wavefront)
        const uint32_t tiles_per_round = gridDim.x * kLpWaves;
        const uint32_t rw_max = (uint32_t)kLpSlots / n_step;
        const uint32_t rounds = (n_alive + tiles_per_round * rw_max - 1) / (tiles_per_round * rw_max);
        const uint32_t rw = (n_alive + tiles_per_round * rounds - 1) / (tiles_per_round * rounds);
        const uint32_t n_tiles = (n_alive + rw - 1) / rw;
        const uint32_t mult = n_tiles % 1237u ? 1237u : 1u;   // q -> q * mult mod n_tiles is a permutation of the tiles (1237 is prime)
        LpPool &pool = sh.pool;
        unsigned long long t_mark = 0ull, cyc[7] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, longest = 0ull;
        auto lap = [&](int phase) {
            if (prof) {
                const unsigned long long now = __builtin_readcyclecounter();
                cyc[phase] += now - t_mark;
                t_mark = now;
            }
        };

        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t q0 = (r * gridDim.x + blockIdx.x) * kLpWaves;
            if (q0 >= n_tiles) break;   // nothing for this workgroup in this round (the same decision in all its wavefronts)
            const unsigned long long t_round = prof ? __builtin_readcyclecounter() : 0ull;
            t_mark = t_round;
            if (!weights_resident) {
                lp_fill_shared(sh, a, tid, lane);
                weights_resident = true;   // (the first __syncthreads below publishes them)
            }
            const uint32_t q = q0 + (uint32_t)wave;
            const bool has_tile = q < n_tiles;
            const uint32_t tile = has_tile ? (uint32_t)(((unsigned long long)q * mult) % n_tiles) : 0u;

            // ---- phase 1: this trip's samples of the wavefront's rw rays (one or two rays per lane) into the pool ---------------
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
                    const uint32_t ray = trip == 0 ? n : (uint32_t)alive_in[n];
                    const uint32_t avail = a.sample_cnt[ray] - used;
                    cnt = avail < n_step ? avail : n_step;
                    pool.ray[local] = ray;
                    if (cnt) {
                        const float *o = a.rays_o + 3ull * ray, *d = a.rays_d + 3ull * ray;
                        const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
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
                }
                if (in_tile) pool.cnt[local] = (uint8_t)(has_ray ? cnt : kNoRay);
                const uint32_t incl = wave_inclusive_scan(cnt, lane);
                cnt_r[sub] = cnt;
                pos_r[sub] = my_valid + incl - cnt;
                my_valid += (uint32_t)__shfl((int)incl, 63);
            }
            if (lane == 0) pool.wave_valid[wave] = my_valid;
            lap(0);
            __syncthreads();
            // compaction over the workgroup: wavefront tiles in order, rays in order inside a tile
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int v = 0; v < kLpWaves; ++v) {
                const uint32_t x = pool.wave_valid[v];
                before += v < wave ? x : 0u;
                total += x;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const uint32_t slot0 = ((uint32_t)wave * rw + (uint32_t)(sub * 64 + lane)) * n_step;
                for (uint32_t s = 0; s < cnt_r[sub]; ++s) pool.order[before + pos_r[sub] + s] = (uint16_t)(slot0 + s);
            }
            __syncthreads();
            lap(1);

            // ---- phase 2: the pooled blocks, dealt out round-robin -----------------------------------------------------------------
            for (uint32_t first = 32u * (uint32_t)wave; first < total; first += 32u * kLpWaves) {
                evaluate_block_lp<AMB_D, H, SLOW>(a, sh, pool, first, total, n_step, lane);
                if (prof) cyc[6] += 1ull;
            }
            lap(2);
            __syncthreads();
            lap(3);

            // ---- phase 3: composite, ray state update (the owner of the ray); survivor compaction over the workgroup ---------------
            // ONE list-append atomic per workgroup round: the 2 x 2048 per-wavefront appends of a trip all hit one address at the same time
            // after the barrier and queue up behind each other (~10 ns each, measured: the phase took as long as the evaluation)
            unsigned long long alive_bits[2] = {0ull, 0ull};
            uint32_t ray_r[2] = {0u, 0u};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const uint32_t i = (uint32_t)(sub * 64 + lane);
                if ((uint32_t)(sub * 64) >= rw) break;   // wavefront-uniform
                const uint32_t local = (uint32_t)wave * rw + i;
                const uint32_t cnt = i < rw ? (uint32_t)pool.cnt[local] : kNoRay;
                bool survives = false;
                if (cnt != kNoRay) {
                    const uint32_t ray = pool.ray[local];
                    ray_r[sub] = ray;
                    RayAccum acc = ray_state_load(a.state, ray);
                    const uint32_t base = local * n_step;
                    uint32_t s = 0;
                    for (; s < cnt; ++s) {
                        const uint32_t k = base + s;
                        if (composite_sample(acc, pool.px[k], pool.dt[k], pool.tend[k], pool.py[k], pool.pz[k], pool.cb[k], a.T_thresh)) break;
                    }
                    // the reference declares the ray dead when it stops before n_step samples (terminated, or ran out of samples)
                    survives = (s == n_step);
                    ray_state_store(a.state, ray, acc, __uint_as_float(survives ? used + n_step : used));
                }
                alive_bits[sub] = __ballot(survives);
            }
            const uint32_t surv0 = (uint32_t)__popcll(alive_bits[0]), surv1 = (uint32_t)__popcll(alive_bits[1]);
            if (lane == 0) pool.wave_surv[wave] = surv0 + surv1;
            __syncthreads();
            if (tid == 0) {
                uint32_t sum = 0;
#pragma unroll
                for (int v = 0; v < kLpWaves; ++v) sum += pool.wave_surv[v];
                pool.out_base = sum ? (uint32_t)atomicAdd(&a.counters[trip + 1], (int)sum) : 0u;
            }
            __syncthreads();
            {
                uint32_t out = pool.out_base;
#pragma unroll
                for (int v = 0; v < kLpWaves; ++v) out += v < wave ? pool.wave_surv[v] : 0u;
                const unsigned long long below = (1ull << lane) - 1ull;
                if ((alive_bits[0] >> lane) & 1ull) alive_out[out + (uint32_t)__popcll(alive_bits[0] & below)] = (int32_t)ray_r[0];
                if ((alive_bits[1] >> lane) & 1ull) alive_out[out + surv0 + (uint32_t)__popcll(alive_bits[1] & below)] = (int32_t)ray_r[1];
            }
            if (tid == 0 && total) atomicAdd(&a.counters[64 + trip], (int)total);   // evaluated samples of this trip
            lap(4);
            if (prof) {
                cyc[5] += 1ull;
                const unsigned long long dur = __builtin_readcyclecounter() - t_round;
                longest = dur > longest ? dur : longest;
            }
            // (no barrier here: the next round's phase 1 writes only the wavefront's own rays and slots, which nobody else reads after phase 2,
            // and its first barrier comes before wave_surv / out_base are written again)
        }
        if (prof && lane == 0) {
            for (int ph = 0; ph < 7; ++ph) atomicAdd(&a.phase_cycles[8 * trip + ph], cyc[ph]);
            atomicMax(&a.phase_cycles[8 * trip + 7], longest);
        }
        if (step_before >= a.max_steps) return;   // the step budget is used up: no later trip runs (every workgroup decides the same), no barrier needed
        if (trip + 1 < a.trip_end) grid_barrier(a.sync, gridDim.x * ++barriers, a.spin_limit, a.timeouts);
    }
}

// ---- the whole frame as ONE launch with workgroup-local trips (gfpp_head_frame_persist_lp) ----------------------------------------------
// Per ray the reference's loop composites the first min(c, e + 1, B) of its occupied samples (c = samples the ray owns, e = first sample whose
// pre-sample transmittance is below T_thresh, B = step budget = sum of n_step over the trips that ran), and a ray is alive after the trip window
// that ends at sample S exactly if m = min(c, e) >= S (raymarching.cu:978-1022: a shorter take, or a break, declares it dead).  Neither depends
// on HOW the loop cut the samples into trips, so a workgroup can own rays for the whole frame and loop on its own: no alive list in global
// memory, no list-append atomics, one weight copy per workgroup and frame, no device-wide dependency (any number of these launches can overlap).
// What the schedule decides -- B and the n_alive sequence -- is reconstructed from the histogram of m (k_head_budget_resolve); rays that go past
// max_steps samples leave a snapshot of their state after each of the samples max_steps .. max_steps + 6, and the resolve step picks the one at B.
// Same evaluate_block_lp, same composite_sample, same order along a ray as k_head_trip_pool: results are bit-identical to the trip launches.
// MF: a frame GROUP (gfpp_frame_ws.n_frames = 2..4 frames of N rays each, numbered frame * N + ray in every per-ray array).  The tiles of all frames go
// through one permutation, so a workgroup owns rays of every frame and pools their samples in its rounds: what a launch costs whatever its size -- the
// 124 KB weight image into LDS, a partly filled last block in each of the ~4 local rounds, ingest, the tail behind the busiest workgroup -- is paid once
// per group.  At 256^2 rays a workgroup's share of ONE frame is ~110 occupied rays = 27 blocks for 8 wavefronts (0.117 ms per frame, 0.49 of the
// yardstick); four frames give it the ~440 rays of a 512^2 frame.  Per frame: its own constants (LDS, by sample), histogram and counters.
// H = float (round 4): the exact-fp32 parity mode on the same structure.  Nothing of the launch's control flow depends on the operand type: the fp32 instantiation
// runs the trip kernels' own evaluate_block (head_eval_f32_device.h: v_mfma_f32_32x32x2_f32, weights streamed from L2) on the workgroup pool through a view that gives
// the pool's arrays the names that function uses, keeps no weight image in LDS, and takes the fp32 kernels' argument record next to this launch's.  Bit-identical
// to gfpp_head_frame_trips (k_head_trip_wp) for the same reasons the 16-bit instantiations are to k_head_trip_pool.
template <typename H>
struct PersistArgs {
    LpTripArgs a;
};
template <>
struct PersistArgs<float> {
    LpTripArgs a;
    TripArgs t;
};
struct PoolViewF32 {
    struct Dir {
        const LpPoolP *pool;
        const float *rays_d;
        __device__ __forceinline__ float operator[](uint32_t ray_local) const { return rays_d[3ull * pool->ray_of(ray_local)]; }
    };
    float *px, *py, *pz;              // sample positions by slot ...
    float *sigma, *cr, *cg, *cb;      // ... and, after the evaluation, density and colour in the same words (as evaluate_block_lp leaves them)
    const uint16_t *order;
    uint32_t n_valid;
    Dir dx, dy, dz;
};

// PROF (gfpp_frame_ws.phase_cycles != null; bench.py's roofline section, never the frame loop): thread 0's shader clock by phase, summed over the workgroups into the
// budget counters [44..49] -- the production instantiation carries no clock reads.
template <int AMB_D, typename H, bool SLOW, bool MF, bool PROF = false>
__global__ __launch_bounds__(kLpThreads, kLpThreads / 256) void k_head_frame_persist(PersistArgs<H> pa) {
    constexpr bool F32 = std::is_same<H, float>::value;
    static_assert(!F32 || (!SLOW && !MF), "the fp32 instantiation: generic lookups are its only ones, one frame per launch");
    const LpTripArgs &a = pa.a;
    __shared__ LpShared sh;
    LpPoolP &pool = sh.poolp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t G = gridDim.x, b = blockIdx.x;
    // this workgroup owns the tile slots [q0, q0 + my_tiles): equal counts of CONSECUTIVE slots, whose tiles the multiplicative permutation spreads
    // over the rows and columns of the image (slots strided by G would all land in one column block: (G * mult) % n_tiles has a large common
    // factor with n_tiles)
    // XCD-local ownership (a.xcd_cols8 != 0; round 5): the dispatcher places workgroup b on XCD b % 8 (observed, not promised: a wrong guess costs speed, never
    // bits), and the 8-ray tile column c of the image belongs to XCD c % 8 -- every XCD renders a comb of 8-pixel columns, the same share of every part of the
    // image (balance as before) but 1/8 of the x range: at the levels of the position grid whose tiled index drops z (res >= 256: half of them) and wherever a
    // cell is narrower than the comb's period, the x-y rows it gathers are 1/8 of the level's, which an XCD's 4 MiB L2 holds (all workgroups spread over the whole
    // image: every XCD's L2 saw every row -- L2 hit rate 80 %, 510 MB of fabric traffic per frame in round 4's counters).  The G / 8 workgroups of an XCD share
    // its n_tiles / 8 tiles exactly as all G shared all tiles before: consecutive slots, spread by the multiplicative permutation.
    const bool xcd = a.xcd_cols8 != 0u;
    const uint32_t nt = xcd ? a.n_tiles >> 3 : a.n_tiles, Gs = xcd ? G >> 3 : G, bs = xcd ? b >> 3 : b;
    const uint32_t per_wg = nt / Gs, extra = nt % Gs;
    const uint32_t my_tiles = per_wg + (bs < extra ? 1u : 0u), q0 = bs * per_wg + (bs < extra ? bs : extra);
    if (my_tiles == 0u) return;                                   // tiny frames: fewer tiles than workgroups
    if constexpr (!F32) lp_fill_shared(sh, a, tid, lane);
    if constexpr (MF) {
        for (uint32_t i = (uint32_t)tid; i < 256u * (a.n_frames - 1u); i += kLpThreads)
            (&pool.bias_more[0][0])[i] = a.frame_consts[(size_t)((i >> 8) + 1u) * a.consts_stride + (i & 255u)];
    }
    if (tid < (int)(32u * kPMaxFrames)) (&pool.hist[0][0])[tid] = 0u;
    const uint32_t cap = a.max_steps + 7u;                        // a ray can never composite more samples than that (k_premarch stores no more)
    uint32_t j_next = 0, A = 0, evaluated = 0, round = 0;
    unsigned long long t_mark = PROF ? __builtin_readcyclecounter() : 0ull, cyc[4] = {0ull, 0ull, 0ull, 0ull};   // ingest + fetch | compaction | evaluate | composite + list
    unsigned long long cyc_ingest = 0ull;                          // the ingest steps alone (part of cyc[0])
    auto lap = [&](int k) {
        if constexpr (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            cyc[k] += now - t_mark;
            t_mark = now;
        }
    };
    __syncthreads();                                              // weights, descriptors, zeroed histogram

    for (;;) {
        // ---- ingest: the next (up to 128) tiles of this workgroup, whole tiles as long as their occupied rays fit the list --------------------
        if (j_next < my_tiles && kPoolSlots - A >= 256u) {
            uint32_t ent[2], rank[2];
            bool occupied[2];
            const uint32_t in_tile = (uint32_t)tid % kPTile, seg = ((uint32_t)lane / kPTile) * kPTile;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const uint32_t ts = (uint32_t)sub * kPTilesPerSub + (uint32_t)tid / kPTile, j = j_next + ts;
                uint32_t ray = 0, c = 0, frame = 0;
                bool in_range = false;
                if (j < my_tiles) {
                    uint32_t tile = ((q0 + j) * a.tile_mult) % nt;
                    // the XCD's tile number (row of the frame group, column / 8) -> tile: column = 8 (column / 8) + XCD, and a row holds 8 xcd_cols8 tiles
                    if (xcd) tile = tile * 8u + (b & 7u);
                    if constexpr (MF) {
                        frame = tile / a.tiles_per_frame;
                        tile -= frame * a.tiles_per_frame;
                    }
                    const uint32_t in_frame = tile * kPTile + in_tile;
                    if (in_frame < a.N) {
                        in_range = true;
                        ray = frame * a.N + in_frame;
                        const uint32_t cc = a.sample_cnt[ray];
                        c = cc < cap ? cc : cap;
                    }
                }
                occupied[sub] = c > 0u;
                const unsigned long long bo = __ballot(occupied[sub]), be = __ballot(in_range && c == 0u);
                const uint32_t so = (uint32_t)(bo >> seg) & ((1u << kPTile) - 1u), se = (uint32_t)(be >> seg) & ((1u << kPTile) - 1u);
                // occupied rays | empty rays << 16 | frame << 24 (a tile never straddles two frames)
                if (in_tile == 0u) pool.tile_cnt[ts] = (uint32_t)__popc(so) | ((uint32_t)__popc(se) << 16) | (frame << 24);
                rank[sub] = (uint32_t)__popc(so & ((1u << in_tile) - 1u));
                ent[sub] = ray | (c << 27);
            }
            __syncthreads();
            // Tiles are taken in order and the first that does not fit ends the step: "tile t exists and the occupied rays of the tiles 0..t fit" is true for a
            // prefix of the 128 candidates (counts are non-negative), so the accepted count is a population count and the offsets are prefix sums -- one wavefront
            // scan per half (lane l holds the tiles l and 64 + l; every wavefront computes it for itself) instead of a 128-iteration loop in every thread
            // (~2 us per ingest step).
            static_assert(kPTilesPerStep == 128u && kPTilesPerSub == 64u, "two candidate tiles per lane");
            const uint32_t room = kPoolSlots - A, ts0 = (uint32_t)tid / kPTile;          // this thread's own tiles: ts0 and 64 + ts0
            const uint32_t n0 = pool.tile_cnt[lane] & 0xFFFFu, n1 = pool.tile_cnt[64 + lane] & 0xFFFFu;
            const uint32_t inc0 = wave_inclusive_scan(n0, lane);
            const uint32_t inc1 = (uint32_t)__shfl((int)inc0, 63) + wave_inclusive_scan(n1, lane);
            const unsigned long long fit0 = __ballot(j_next + (uint32_t)lane < my_tiles && inc0 <= room);
            const unsigned long long fit1 = __ballot(j_next + 64u + (uint32_t)lane < my_tiles && inc1 <= room);
            const uint32_t accepted = (uint32_t)__popcll(fit0) + (uint32_t)__popcll(fit1);
            uint32_t acc = 0;
            if (accepted > 64u) acc = (uint32_t)__shfl((int)inc1, (int)(accepted - 65u));
            else if (accepted > 0u) acc = (uint32_t)__shfl((int)inc0, (int)(accepted - 1u));
            const uint32_t off0 = (uint32_t)__shfl((int)(inc0 - n0), (int)ts0), off1 = (uint32_t)__shfl((int)(inc1 - n1), (int)ts0), ts1 = ts0 + kPTilesPerSub;
            if ((uint32_t)tid < accepted) {                                              // rays without any occupied sample: m = 0
                const uint32_t x = pool.tile_cnt[tid], empty = (x >> 16) & 0xFFu;
                if (empty) atomicAdd(&pool.hist[x >> 24][0], empty);
            }
            if (ts0 < accepted && occupied[0]) pool.alive[A + off0 + rank[0]] = ent[0];
            if (ts1 < accepted && occupied[1]) pool.alive[A + off1 + rank[1]] = ent[1];
            A += acc;
            j_next += accepted;
            __syncthreads();
            if constexpr (PROF) cyc_ingest += __builtin_readcyclecounter() - t_mark;
        }
        if (A == 0u) {
            if (j_next >= my_tiles) break;
            continue;                                             // the tiles taken so far were all empty (image border): take more
        }

        // ---- one local trip over the A alive rays: wavefront w takes the list entries [w * rw, (w + 1) * rw) -----------------------------------
        const uint32_t rw = (A + (uint32_t)kLpWaves - 1u) / (uint32_t)kLpWaves;   // <= 128
        uint32_t n_step = (uint32_t)kLpSlots / rw;
        {
            const uint32_t capr = (a.step_caps >> (4u * (round < 7u ? round : 7u))) & 15u;
            n_step = n_step < capr ? n_step : capr;
            n_step = n_step < 1u ? 1u : (n_step > 8u ? 8u : n_step);
        }
        // phase 1: the next samples of every ray (one or two rays per lane) into the pool
        uint32_t my_valid = 0, cnt_r[2], pos_r[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t i = (uint32_t)(sub * 64 + lane);
            const uint32_t idx = (uint32_t)wave * rw + i;
            const bool in_tile = i < rw;
            const bool has_ray = in_tile && idx < A;
            uint32_t cnt = 0;
            if (has_ray) {
                const uint32_t e = pool.alive[idx];
                const uint32_t ray = e & kPRayMask, used = (e >> kPRayBits) & 31u, c = e >> 27;
                const uint32_t rem = c - used;                    // >= 1: a ray without samples left never stays on the list
                cnt = rem < n_step ? rem : n_step;
                const float *o = a.rays_o + 3ull * ray, *d = a.rays_d + 3ull * ray;
                const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
                const float *ts = a.sample_t + (size_t)ray * a.sample_stride + used;
                const uint32_t base = idx * n_step;
                // all sample times of the take in ONE memory round trip: eight unconditional loads from clamped indices (used + 7 < sample_stride),
                // issued together with the ray's origin / direction -- a loop of load-then-store iterations costs a round trip per sample
                float tv[8];
#pragma unroll
                for (uint32_t s = 0; s < 8u; ++s) tv[s] = ts[s < cnt ? s : 0u];
#pragma unroll
                for (uint32_t s = 0; s < 8u; ++s) {
                    if (s < cnt) {
                        // the same expressions as march_one_ray (raymarching.cu:873-882) evaluated at the stored t
                        const float t0 = tv[s];
                        pool.px[base + s] = clampf(fmaf(t0, dx, ox), -a.mp.bound, a.mp.bound);
                        pool.py[base + s] = clampf(fmaf(t0, dy, oy), -a.mp.bound, a.mp.bound);
                        pool.pz[base + s] = clampf(fmaf(t0, dz, oz), -a.mp.bound, a.mp.bound);
                        pool.t0[base + s] = t0;
                    }
                }
            }
            if (in_tile) pool.cnt[idx] = (uint8_t)(has_ray ? cnt : kNoRay);
            const uint32_t incl = wave_inclusive_scan(cnt, lane);
            cnt_r[sub] = cnt;
            pos_r[sub] = my_valid + incl - cnt;
            my_valid += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) pool.wave_valid[wave] = my_valid;
        lap(0);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int v = 0; v < kLpWaves; ++v) {
            const uint32_t x = pool.wave_valid[v];
            before += v < wave ? x : 0u;
            total += x;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t slot0 = ((uint32_t)wave * rw + (uint32_t)(sub * 64 + lane)) * n_step;
            for (uint32_t s = 0; s < cnt_r[sub]; ++s) pool.order[before + pos_r[sub] + s] = (uint16_t)(slot0 + s);
        }
        __syncthreads();
        lap(1);

        // phase 2: the pooled 32-sample blocks, dealt out round-robin (a start offset between the two wavefronts of a SIMD was measured in round 5 and dropped:
        // docs/LAB_NOTEBOOK.md, "Fine stagger")
        if constexpr (F32) {
            PoolViewF32 view{pool.px, pool.py, pool.pz, pool.px, pool.py, pool.pz, pool.cb, pool.order, total,
                             {&pool, a.rays_d}, {&pool, a.rays_d + 1}, {&pool, a.rays_d + 2}};
            for (uint32_t first = 32u * (uint32_t)wave; first < total; first += 32u * kLpWaves) evaluate_block<AMB_D>(pa.t, view, first, n_step, lane);
        } else {
            for (uint32_t first = 32u * (uint32_t)wave; first < total; first += 32u * kLpWaves)
                evaluate_block_lp<AMB_D, H, SLOW, false, MF>(a, sh, pool, first, total, n_step, lane);
        }
        __syncthreads();
        lap(2);

        // phase 3: composite (the lane that fetched the ray), snapshots past max_steps, histogram of the rays that end here
        unsigned long long alive_bits[2] = {0ull, 0ull};
        uint32_t keep[2] = {0u, 0u};
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const uint32_t i = (uint32_t)(sub * 64 + lane);
            if ((uint32_t)(sub * 64) >= rw) break;                // wavefront-uniform
            const uint32_t idx = (uint32_t)wave * rw + i;
            const uint32_t cnt = i < rw ? (uint32_t)pool.cnt[idx] : kNoRay;
            bool survives = false;
            if (cnt != kNoRay) {
                const uint32_t e = pool.alive[idx];
                const uint32_t ray = e & kPRayMask, used = (e >> kPRayBits) & 31u, c = e >> 27;
                RayAccum acc = ray_state_load(a.state, ray);
                const uint32_t base = idx * n_step;
                uint32_t s = 0;
                bool stop = false;
                for (; s < cnt; ++s) {
                    const uint32_t k = base + s;
                    const float t0 = pool.t0[k];
                    const float dt = clampf(t0 * a.mp.dt_gamma, a.mp.dt_min, a.mp.dt_max);   // (raymarching.cu:905-913)
                    stop = composite_sample(acc, pool.px[k], dt, t0 + dt, pool.py[k], pool.pz[k], pool.cb[k], a.T_thresh);
                    const uint32_t done = used + s + 1u;
                    if (done >= a.max_steps && done < cap) {      // the budget B may end this ray here: keep the state after `done` samples
                        float *sp = a.snaps + ((size_t)ray * 7u + (done - a.max_steps)) * 5u;
                        sp[0] = acc.wsum; sp[1] = acc.depth; sp[2] = acc.r; sp[3] = acc.g; sp[4] = acc.b;
                    }
                    if (stop) break;
                }
                const uint32_t used2 = used + (stop ? s + 1u : cnt);
                survives = !stop && used2 < c;
                uint32_t frame = 0;
                if constexpr (MF) frame = (ray >= a.N ? 1u : 0u) + (ray >= 2u * a.N ? 1u : 0u) + (ray >= 3u * a.N ? 1u : 0u);
                if (!survives) atomicAdd(&pool.hist[frame][stop ? used + s : c], 1u);     // m = min(c, e) <= cap <= 31
                ray_state_store(a.state, ray, acc, __uint_as_float(used2));
                a.state[(size_t)kRayRec * ray + 7] = __uint_as_float(used2);        // word 7 (zero since the frame began): samples composited, for k_head_budget_resolve
                keep[sub] = (e & ~(31u << kPRayBits)) | (used2 << kPRayBits);
            }
            alive_bits[sub] = __ballot(survives);
        }
        const uint32_t surv0 = (uint32_t)__popcll(alive_bits[0]), surv1 = (uint32_t)__popcll(alive_bits[1]);
        if (lane == 0) pool.wave_surv[wave] = surv0 + surv1;
        evaluated += total;
        ++round;
        __syncthreads();
        {   // the survivors, compacted in place (every read of the old list happened before the barrier)
            uint32_t out = 0, all = 0;
#pragma unroll
            for (int v = 0; v < kLpWaves; ++v) {
                const uint32_t x = pool.wave_surv[v];
                out += v < wave ? x : 0u;
                all += x;
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if ((alive_bits[0] >> lane) & 1ull) pool.alive[out + (uint32_t)__popcll(alive_bits[0] & below)] = keep[0];
            if ((alive_bits[1] >> lane) & 1ull) pool.alive[out + surv0 + (uint32_t)__popcll(alive_bits[1] & below)] = keep[1];
            A = all;
        }
        __syncthreads();
        lap(3);
    }
    if (tid < (int)(32u * (MF ? a.n_frames : 1u))) {       // every frame's histogram into the frame's own counters
        const uint32_t f = (uint32_t)tid >> 5, m = (uint32_t)tid & 31u;
        if (pool.hist[f][m]) atomicAdd(&a.budget[f * (uint32_t)kCounterWords + m], (int)pool.hist[f][m]);
    }
    if (tid == 0) {
        if (evaluated) atomicAdd(&a.budget[kBudgetSamples], (int)evaluated);
        atomicAdd(&a.budget[kBudgetRounds], (int)round);
        atomicMax(&a.budget[kBudgetRoundsMax], (int)round);
        atomicMax(&a.budget[kBudgetSamplesMax], (int)evaluated);
        if constexpr (PROF) {
            // where this workgroup's time went (thread 0's clock, units of 1024 shader cycles; sums over the workgroups + the longest workgroup)
            for (int k = 0; k < 4; ++k) atomicAdd(&a.budget[kBudgetCycles + k], (int)(cyc[k] >> 10));
            atomicMax(&a.budget[kBudgetCycles + 4], (int)((cyc[0] + cyc[1] + cyc[2] + cyc[3]) >> 10));
            atomicAdd(&a.budget[kBudgetCycles + 5], (int)(cyc_ingest >> 10));
        }
    }
}

__global__ __launch_bounds__(256) void k_head_budget_resolve(float *__restrict__ state, const float *__restrict__ snaps, const int32_t *__restrict__ hist,
                                                             int32_t *__restrict__ counters, uint32_t N, uint32_t N_global, uint32_t max_steps) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0) {
        (void)budget_from_hist(hist, N_global, max_steps, counters);
        counters[64] = counters[kBudgetBase + kBudgetSamples];          // the launch's evaluated samples, where trip 0's count used to be
    }
    if (n >= N) return;
    const uint32_t done = __float_as_uint(state[(size_t)kRayRec * n + 7]);   // 0: the ray never had a sample (the frame's begin kernel zeroes the word)
    if (done <= max_steps || done > max_steps + 7u) return;            // B >= max_steps whenever a ray got this far
    const uint32_t B = budget_from_hist(hist, N_global, max_steps, nullptr);
    if (done <= B || B < max_steps) return;
    const float *sp = snaps + ((size_t)n * 7u + (B - max_steps)) * 5u;
    *reinterpret_cast<float4 *>(state + (size_t)kRayRec * n) = float4{sp[0], sp[1], sp[2], sp[3]};
    *reinterpret_cast<float2 *>(state + (size_t)kRayRec * n + 4) = float2{sp[4], __uint_as_float(B)};
}

// The sample positions of a ray do not depend on the radiance field (only on the occupancy bitfield), and the reference's marcher
// carries nothing but t from one loop iteration to the next (raymarching.cu:857, renderer.py:366), so the whole per-ray sample
// sequence can be marched ONCE per frame, at full occupancy, instead of piecewise inside the register- and LDS-heavy trip kernel:
// trip k then just takes the next n_step entries.  A ray can consume at most max_steps + 7 samples (cumulative step < max_steps
// before the last trip, n_step <= 8).
struct PremarchArgs {
    MarchParams mp;
    const uint8_t *bitfield;
    const float *rays_o, *rays_d, *nears, *fars;
    float *sample_t;
    uint32_t *sample_cnt;
    uint32_t N, stride, max_samples;
    // fused frame begin (k_begin_premarch): slab test + state / counter reset in the same pass over the rays
    float min_near, aabb[6];
    float *nears_out, *fars_out, *state;
    int32_t *counters;
    // conservative bounds of the occupied cells (gfpp_head_model.occ_aabb; occ_valid = 0: unknown)
    float occ[6];
    uint32_t occ_valid;
    // a frame GROUP (k_group_begin): `frames` frames of N rays behind each other in every array, counters [frames, kCounterWords]; the rays are generated here
    // from each frame's pose (the arithmetic of k_get_rays, raymarch.hip) and stored for the head launch
    uint32_t fixed_step;           // gfpp_tuning.march_fixed_step (1: rays that qualify take march_one_ray_fixed_step; 0: the A/B and parity partner)
    uint32_t frames;
    const float *poses;            // frame f's cam2world [4,4] (ngp convention) at poses + f * pose_stride
    uint32_t pose_stride, W;
    float fx, fy, cx, cy;
    float *rays_o_out, *rays_d_out;
};

// Where the marcher may stop: beyond the point where the ray leaves the bounds of the occupied cells no cell is occupied, so the loop of
// march_one_ray would only skip empty cells until `far` and emit nothing more -- the emitted samples (and their t, a chain of fp32 additions from
// `near` on: the walk BEFORE the bounds cannot be skipped, and skipping only the bitfield reads there measured slower, a divergent branch per
// cell) are the same bits.  A ray that misses the bounds has no sample at all (-1).
__device__ __forceinline__ float march_far_limit(float ox, float oy, float oz, float dx, float dy, float dz, const PremarchArgs &p, float far,
                                                 float *t_enter = nullptr) {
    if (t_enter) *t_enter = -FLT_MAX;           // (where the ray ENTERS the bounds: before it no cell is occupied either -- the fixed-step walk probes from there on)
    if (!p.occ_valid) return far;
    const float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    float tn = -FLT_MAX, tf = FLT_MAX;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (fabsf(d[k]) < 1e-12f) {
            if (o[k] < p.occ[k] || o[k] > p.occ[3 + k]) return -1.0f;
        } else {
            const float rd = 1.0f / d[k];
            float t1 = (p.occ[k] - o[k]) * rd, t2 = (p.occ[3 + k] - o[k]) * rd;
            if (t1 > t2) { const float t = t1; t1 = t2; t2 = t; }
            tn = fmaxf(tn, t1);
            tf = fminf(tf, t2);
        }
    }
    if (tn > tf) return -1.0f;
    if (t_enter) *t_enter = tn;
    return fminf(far, tf);
}

// The pre-march of one ray: the t of every occupied sample into out[], the count returned.  One-cascade models with <= 256 cells per axis (every shipped one:
// wavefront-uniform test) take the marcher's lean probe (march_device.h: ONE_SMALL_SHELL), same bits.
// The fixed-step walk of the pre-march (march_device.h::march_one_ray_fixed_step: ONE chain t_k+1 = t_k + dt_max, every point probed, same bits as the general walk),
// shaped for what bounds this kernel -- not vector instructions but a chain of DEPENDENT bitfield reads per ray (~27 from the box face to the far side of the
// occupied cells, each an L2 round trip): (1) no probe before the ray enters the bounds of the occupied cells (they carry a whole cell of margin, k_occupancy_bounds:
// a chain point outside them lies in an empty cell) -- the chain itself is still walked addition by addition, it is what fixes the later t; (2) the probes of four
// consecutive chain points are independent of each other, so their reads are issued together and the results consumed in order.
__device__ __forceinline__ uint32_t premarch_fixed(float ox, float oy, float oz, float dx, float dy, float dz, float t, float far, float t_enter, const PremarchArgs &p,
                                                   float *__restrict__ out) {
    const MarchParams &mp = p.mp;
    const float dt = mp.dt_max, mip_rbound = mp.bound <= 1.0f ? mp.rbound : 1.0f;
    while (t < far && t + dt < t_enter) t += dt;           // (the next point is still outside: this one cannot be inside either)
    uint32_t step = 0;
    while (t < far && step < p.max_samples) {
        float tk[4];
        uint32_t cell[4];
        tk[0] = t;
#pragma unroll
        for (int i = 1; i < 4; ++i) tk[i] = tk[i - 1] + dt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = clampf(fmaf(tk[i], dx, ox), -mp.bound, mp.bound);
            const float y = clampf(fmaf(tk[i], dy, oy), -mp.bound, mp.bound);
            const float z = clampf(fmaf(tk[i], dz, oz), -mp.bound, mp.bound);
            cell[i] = morton3_8((uint32_t)voxel_of(x, mip_rbound, mp), (uint32_t)voxel_of(y, mip_rbound, mp), (uint32_t)voxel_of(z, mip_rbound, mp));
        }
        uint32_t byte[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) byte[i] = p.bitfield[cell[i] >> 3];      // (positions are clamped into the box: every address is valid, also beyond `far`)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tk[i] < far && step < p.max_samples && ((byte[i] >> (cell[i] & 7u)) & 1u)) out[step++] = tk[i];
        }
        t = tk[3] + dt;
    }
    return step;
}

__device__ __forceinline__ bool tuning_fixed_step(const PremarchArgs &p) { return p.fixed_step != 0u; }
__device__ __forceinline__ uint32_t premarch_ray(float ox, float oy, float oz, float dx, float dy, float dz, float t, float far_limit, const PremarchArgs &p,
                                                 float *__restrict__ out, float t_enter = -FLT_MAX) {
    if (march_fixed_step_ok(dx, dy, dz, p.mp) && tuning_fixed_step(p))          // (per ray; every ray of a camera that does not look along a voxel diagonal)
        return premarch_fixed(ox, oy, oz, dx, dy, dz, t, far_limit, t_enter, p, out);
    if (p.mp.C == 1u && p.mp.H <= 256u)
        return march_one_ray<true>(ox, oy, oz, dx, dy, dz, t, far_limit, p.max_samples, p.bitfield, p.mp, [&](uint32_t s, const Sample &smp) { out[s] = smp.t0; });
    return march_one_ray<false>(ox, oy, oz, dx, dy, dz, t, far_limit, p.max_samples, p.bitfield, p.mp, [&](uint32_t s, const Sample &smp) { out[s] = smp.t0; });
}

__global__ __launch_bounds__(256) void k_premarch(PremarchArgs p) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.N) return;
    const float *o = p.rays_o + 3ull * n, *d = p.rays_d + 3ull * n;
    float t = p.nears[n];
    float *out = p.sample_t + (size_t)n * p.stride;
    float t_enter;
    const float far_limit = march_far_limit(o[0], o[1], o[2], d[0], d[1], d[2], p, p.fars[n], &t_enter);
    p.sample_cnt[n] = premarch_ray(o[0], o[1], o[2], d[0], d[1], d[2], t, far_limit, p, out, t_enter);
}

// k_frame_begin (frame_head.hip) and k_premarch in one pass: the slab test's near is the marcher's start, so the rays are read once and one
// launch (and the gap behind it) leaves the frame's critical path.  Same expressions, same bits as the two kernels.
__global__ __launch_bounds__(256) void k_begin_premarch(PremarchArgs p) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < kCounterWords) p.counters[threadIdx.x] = threadIdx.x == 0 ? (int32_t)p.N : 0;
    if (n >= p.N) return;
    const float *o = p.rays_o + 3ull * n, *d = p.rays_d + 3ull * n;
    const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
    const RayBox rb = ray_box(ox, oy, oz, dx, dy, dz, p.aabb, p.min_near);
    p.nears_out[n] = rb.near;
    p.fars_out[n] = rb.far;
    *reinterpret_cast<float4 *>(p.state + (size_t)kRayRec * n) = float4{0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<float4 *>(p.state + (size_t)kRayRec * n + 4) = float4{0.0f, rb.near, rb.near, 0.0f};
    float t = rb.near;
    float *out = p.sample_t + (size_t)n * p.stride;
    float t_enter;
    const float far_limit = march_far_limit(ox, oy, oz, dx, dy, dz, p, rb.far, &t_enter);
    p.sample_cnt[n] = premarch_ray(ox, oy, oz, dx, dy, dz, t, far_limit, p, out, t_enter);
}

// The prologue of a frame group as ONE launch: ray generation (k_get_rays' expressions: utils.py:352-363) + slab test + state / counter reset + pre-march
// for the K frames' rays.  In the clip loop's kernel trace the four frames' `k_get_rays` + `k_begin_premarch` of a group (8 launches, ~140 us end to end)
// were the one piece of a lane's prologue that nothing overlapped when both lanes' head launches had just ended.  Same expressions, same bits as the
// separate kernels.
__global__ __launch_bounds__(256) void k_group_begin(PremarchArgs p) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    // every frame's counters from block 0 (a launch of tiny frames has fewer blocks than frames: one block per frame left the later frames' histograms stale)
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < p.frames * (uint32_t)kCounterWords; i += 256u) p.counters[i] = i % (uint32_t)kCounterWords == 0u ? (int32_t)p.N : 0;
    if (n >= p.frames * p.N) return;
    const uint32_t f = n / p.N, pix = n - f * p.N;
    const float *pose = p.poses + (size_t)f * p.pose_stride;
    const uint32_t h = pix / p.W, w = pix - h * p.W;
    const float xs = ((float)w + 0.5f - p.cx) / p.fx;
    const float ys = ((float)h + 0.5f - p.cy) / p.fy;
    const float norm = sqrtf(fmaf(xs, xs, fmaf(ys, ys, 1.0f)));
    const float ux = xs / norm, uy = ys / norm, uz = 1.0f / norm;
    float o3[3], d3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d3[r] = fmaf(pose[4 * r + 2], uz, fmaf(pose[4 * r + 1], uy, pose[4 * r] * ux));      // rays_d = R @ dir
        o3[r] = pose[4 * r + 3];
        p.rays_d_out[3ull * n + r] = d3[r];
        p.rays_o_out[3ull * n + r] = o3[r];
    }
    const float ox = o3[0], oy = o3[1], oz = o3[2], dx = d3[0], dy = d3[1], dz = d3[2];
    const RayBox rb = ray_box(ox, oy, oz, dx, dy, dz, p.aabb, p.min_near);
    p.nears_out[n] = rb.near;
    p.fars_out[n] = rb.far;
    *reinterpret_cast<float4 *>(p.state + (size_t)kRayRec * n) = float4{0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<float4 *>(p.state + (size_t)kRayRec * n + 4) = float4{0.0f, rb.near, rb.near, 0.0f};
    float t = rb.near;
    float *out = p.sample_t + (size_t)n * p.stride;
    float t_enter;
    const float far_limit = march_far_limit(ox, oy, oz, dx, dy, dz, p, rb.far, &t_enter);
    p.sample_cnt[n] = premarch_ray(ox, oy, oz, dx, dy, dz, t, far_limit, p, out, t_enter);
}

// k_head_budget_resolve for the K frames of a group in one launch (each frame against its own histogram / counters)
__global__ __launch_bounds__(256) void k_group_budget_resolve(float *__restrict__ state, const float *__restrict__ snaps, int32_t *__restrict__ counters, uint32_t N,
                                                              uint32_t frames, uint32_t max_steps) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n < frames) {
        int32_t *c = counters + (size_t)n * kCounterWords;
        (void)budget_from_hist(c + kBudgetBase, N, max_steps, c);
        if (n == 0) c[64] = c[kBudgetBase + kBudgetSamples];            // the launch's evaluated samples, where trip 0's count used to be (kept in the first frame's counters)
    }
    if (n >= frames * N) return;
    const uint32_t done = __float_as_uint(state[(size_t)kRayRec * n + 7]);
    if (done <= max_steps || done > max_steps + 7u) return;
    const uint32_t f = n / N;
    const uint32_t B = budget_from_hist(counters + (size_t)f * kCounterWords + kBudgetBase, N, max_steps, nullptr);
    if (done <= B || B < max_steps) return;
    const float *sp = snaps + ((size_t)n * 7u + (B - max_steps)) * 5u;
    *reinterpret_cast<float4 *>(state + (size_t)kRayRec * n) = float4{sp[0], sp[1], sp[2], sp[3]};
    *reinterpret_cast<float2 *>(state + (size_t)kRayRec * n + 4) = float2{sp[4], __uint_as_float(B)};
}

// ---- per-sample evaluation (RADNeRF.forward, radnerf.py:108-141) with the 16-bit trip kernel's own arithmetic -----------------------------
// 32 caller-supplied (position, direction) pairs per wavefront pass, laid out as a one-sample-per-ray tile for evaluate_block_lp -- the function
// the trips run.  Pins the fragment layouts (weights, skinny rows, merged geo/colour matrix, folded biases) sample by sample.
struct LpEvalArgs {
    LpTripArgs t;
    const float *positions;
    float *sigma, *color, *ambient;
    uint32_t M;
    uint32_t waves;   // wavefronts per workgroup that take samples (8; fewer only in occupancy experiments, GFPP_EVAL_WAVES)
};

template <int AMB_D, typename H, bool SLOW>
__global__ __launch_bounds__(kLpThreads, kLpThreads / 256) void k_head_eval_lp(LpEvalArgs e) {
    __shared__ LpShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31;
    lp_fill_shared(sh, e.t, tid, lane);
    __syncthreads();
    LpWaveTile &wt = sh.tile[wave];
    if ((uint32_t)wave >= e.waves) return;
    // experiment order (waves < 8): wave w of the workgroup sits on SIMD w % 4, so waves 0..3 are one per SIMD
    for (uint32_t base = (blockIdx.x * e.waves + wave) * 32u; base < e.M; base += gridDim.x * e.waves * 32u) {
        const uint32_t idx = base + (uint32_t)j;
        const bool ok = idx < e.M;
        if (lane < 32) {
            wt.px[j] = ok ? e.positions[3ull * idx] : 0.0f;
            wt.py[j] = ok ? e.positions[3ull * idx + 1] : 0.0f;
            wt.pz[j] = ok ? e.positions[3ull * idx + 2] : 0.0f;
            wt.ray[j] = ok ? idx : 0u;          // the direction of "ray" idx is read from t.rays_d = directions
            wt.order[j] = (uint8_t)j;
        }
        wave_sync();
        LpTripArgs a = e.t;
        a.dbg_ambient = e.ambient ? e.ambient + (size_t)base * AMB_D : nullptr;
        const uint32_t n_valid = e.M - base < 32u ? e.M - base : 32u;
        evaluate_block_lp<AMB_D, H, SLOW, true>(a, sh, wt, 0, n_valid, 1, lane);
        wave_sync();
        if (lane < 32 && ok) {
            e.sigma[idx] = wt.px[j];
            e.color[3ull * idx] = wt.py[j];
            e.color[3ull * idx + 1] = wt.pz[j];
            e.color[3ull * idx + 2] = wt.cb[j];
        }
        wave_sync();
    }
}

template <int AMB_D, typename H, bool SLOW>
static void launch_eval_lp(uint32_t grid, hipStream_t st, const LpEvalArgs &e) {
    hipLaunchKernelGGL((k_head_eval_lp<AMB_D, H, SLOW>), dim3(grid), dim3(kLpThreads), 0, st, e);
}

template <int AMB_D, typename H, bool SLOW>
static void launch_lp(uint32_t grid, hipStream_t st, const LpTripArgs &a) {
    if (a.phase_cycles) hipLaunchKernelGGL((k_head_trip_pool<AMB_D, H, SLOW, true>), dim3(grid), dim3(kLpThreads), 0, st, a);
    else hipLaunchKernelGGL((k_head_trip_pool<AMB_D, H, SLOW, false>), dim3(grid), dim3(kLpThreads), 0, st, a);
}

static bool lp_grid_ok(const gfpp_grid_desc &g, uint32_t D) {
    return g.table && g.levels && g.D == D && g.L == 16 && g.gridtype <= 1 && g.interp <= 1 && g.dtype == GFPP_F32;
}
// the 16-bit corner-block copy of a grid (gfpp_head_model.pos_grid_blk / amb_grid_blk): same level structure, 16-byte rows of 8 halves
static bool lp_block_grid_ok(const gfpp_grid_desc &b, const gfpp_grid_desc &g) {
    return b.table && b.levels && b.D == g.D && b.L == 16 && b.gridtype == g.gridtype && b.interp == g.interp && b.align_corners == g.align_corners &&
           b.dtype == GFPP_F16 && b.row_padded == 2u;
}

// How many trips get a launch of their own before the multi-trip launch takes over (gfpp_tuning.lp_separate_trips overrides, for experiments).
// 5: with the shipped schedule (n_step 1, 2, 2, 2, 4, 8 against max_steps 16) trip 5 uses up the step budget, so the multi-trip launch that
// starts with it runs that trip and returns without a barrier -- no launch is spent on finding nothing left (6 would: +4 us per frame).
// Frames that go on pay a device-wide barrier (~17 us) per further trip instead of a launch (~9 us).
static uint32_t lp_separate_trips() {
    const int n = tuning().lp_separate_trips;
    return n < 0 ? 5u : (uint32_t)n;
}

static int lp_cu_count() { return cu_count(); }

}  // namespace gfpp

using namespace gfpp;

static int lp_check_common(const char *who, const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                           uint32_t max_steps) {
    if (!model || !ws || !rays_o || !rays_d) { set_error("%s: null argument", who); return GFPP_EINVAL; }
    if (max_steps == 0 || max_steps > (uint32_t)kMaxTrips) { set_error("%s: max_steps must be in 1..%d", who, kMaxTrips); return GFPP_EUNSUPPORTED; }
    if (model->cascade < 1 || model->cascade > 8 || !model->density_bitfield) { set_error("%s: bad model", who); return GFPP_EINVAL; }
    if (!ws->sample_t || !ws->sample_cnt || ws->sample_stride < max_steps + 7u || !ws->nears || !ws->fars) {
        set_error("%s: the workspace needs sample_t [N, sample_stride >= max_steps + 7] and sample_cnt [N]", who);
        return GFPP_EINVAL;
    }
    return 0;
}

// The model-dependent part of the kernel arguments (grids, weight images); shared by the trip launches and the per-sample evaluation entry.
static int lp_model_args(const char *who, const gfpp_head_model *model, LpTripArgs &a) {
    if (!model->lp_weights || !model->lp_skinny || (model->lp_dtype != GFPP_F16 && model->lp_dtype != GFPP_BF16)) {
        set_error("%s: the model carries no 16-bit weight image (lp_weights / lp_skinny / lp_dtype)", who);
        return GFPP_EINVAL;
    }
    if (!lp_grid_ok(model->pos_grid, 3) || !(lp_grid_ok(model->amb_grid, 2) || lp_grid_ok(model->amb_grid, 3))) {
        set_error("%s: grids must be 16-level fp32 tables, position D=3, ambient D in {2,3}", who);
        return GFPP_EUNSUPPORTED;
    }
    if (!model->pos_grid.levels_host || !model->amb_grid.levels_host) { set_error("%s: grid descriptors carry no host level table (levels_host)", who); return GFPP_EINVAL; }
    uint32_t any_slow = 0;
    for (int which = 0; which < 2; ++which) {
        const gfpp_grid_desc &gd = which ? model->amb_grid : model->pos_grid;
        for (int l = 0; l < 16; ++l) any_slow |= gd.levels_host[l].flags & GFPP_LEVEL_SLOW;
    }
    // a model WITHOUT the 16-bit corner-block copies (both descriptors empty: the caller's opt-out, GFPP_LP_BLOCK_TABLE=0 in the Python binding) renders through the
    // generic lookup on its fp32 tables, like a hash-grid model
    if (!model->pos_grid_blk.table && !model->amb_grid_blk.table) any_slow |= GFPP_LEVEL_SLOW;
    for (int which = 0; which < 2; ++which) {
        const gfpp_grid_desc &gd = which ? model->amb_grid : model->pos_grid, &gb = which ? model->amb_grid_blk : model->pos_grid_blk;
        LpGrid &g = which ? a.amb : a.pos;
        g.any_slow = any_slow;            // one kernel instantiation serves both grids: a slow level in either sends both through the generic lookup
        if (any_slow) {
            g.levels = gd.levels;
            g.table = gd.table;
        } else {
            if (!lp_block_grid_ok(gb, gd)) {
                set_error("%s: the model carries no 16-bit corner-block copy of its %s table (gfpp_head_model.%s_grid_blk: f16, row_padded 2)", who,
                          which ? "ambient" : "position", which ? "amb" : "pos");
                return GFPP_EINVAL;
            }
            g.levels = gb.levels;
            g.table = gb.table;
        }
        g.gridtype = gd.gridtype; g.interp = gd.interp; g.align_corners = gd.align_corners;
    }
    a.w16 = (const uint4 *)model->lp_weights;
    a.skinny16 = (const uint32_t *)model->lp_skinny;
    a.density_scale = model->density_scale;
    a.dbg_ambient = nullptr;
    a.phase_cycles = nullptr;
    return 0;
}

static void premarch_occupancy(PremarchArgs &p, const gfpp_head_model *model) {
    const uint32_t mode = tuning().occ_clip ? 1u : 0u;          // A/B switch, read at every issue (a captured graph keeps what it was captured with)
    p.fixed_step = tuning().march_fixed_step ? 1u : 0u;
    p.occ_valid = 0u;
    for (int i = 0; i < 6; ++i) p.occ[i] = model->occ_aabb[i];
    if (model->occ_aabb[3] > model->occ_aabb[0] && model->occ_aabb[4] > model->occ_aabb[1] && model->occ_aabb[5] > model->occ_aabb[2]) p.occ_valid = mode;
}

GFPP_API int gfpp_head_frame_premarch(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                      float dt_gamma, uint32_t max_steps, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_frame_premarch", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    PremarchArgs p;
    p.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    p.bitfield = model->density_bitfield;
    p.rays_o = rays_o; p.rays_d = rays_d; p.nears = ws->nears; p.fars = ws->fars;
    p.sample_t = ws->sample_t; p.sample_cnt = ws->sample_cnt;
    p.N = ws->N; p.stride = ws->sample_stride; p.max_samples = max_steps + 7u;
    p.min_near = 0.0f; p.nears_out = nullptr; p.fars_out = nullptr; p.state = nullptr; p.counters = nullptr;
    for (int i = 0; i < 6; ++i) p.aabb[i] = 0.0f;
    premarch_occupancy(p, model);
    p.frames = 1; p.poses = nullptr; p.pose_stride = 0; p.W = 0; p.fx = p.fy = p.cx = p.cy = 0.0f; p.rays_o_out = nullptr; p.rays_d_out = nullptr;
    hipLaunchKernelGGL(k_premarch, dim3(div_up(ws->N, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gfpp_head_frame_premarch");
}

GFPP_API int gfpp_head_frame_begin_premarch(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                            float dt_gamma, uint32_t max_steps, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_frame_begin_premarch", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    if (!ws->ray_state || !ws->counters || ws->N == 0) { set_error("gfpp_head_frame_begin_premarch: incomplete workspace"); return GFPP_EINVAL; }
    PremarchArgs p;
    p.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    p.bitfield = model->density_bitfield;
    p.rays_o = rays_o; p.rays_d = rays_d; p.nears = nullptr; p.fars = nullptr;
    p.sample_t = ws->sample_t; p.sample_cnt = ws->sample_cnt;
    p.N = ws->N; p.stride = ws->sample_stride; p.max_samples = max_steps + 7u;
    p.min_near = model->min_near;
    for (int i = 0; i < 6; ++i) p.aabb[i] = model->aabb[i];
    p.nears_out = ws->nears; p.fars_out = ws->fars; p.state = ws->ray_state; p.counters = ws->counters;
    premarch_occupancy(p, model);
    p.frames = 1; p.poses = nullptr; p.pose_stride = 0; p.W = 0; p.fx = p.fy = p.cx = p.cy = 0.0f; p.rays_o_out = nullptr; p.rays_d_out = nullptr;
    hipLaunchKernelGGL(k_begin_premarch, dim3(div_up(ws->N, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gfpp_head_frame_begin_premarch");
}

GFPP_API int gfpp_head_group_begin(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *poses, uint32_t pose_stride, float fx, float fy, float cx,
                                   float cy, uint32_t H, uint32_t W, float *rays_o, float *rays_d, float dt_gamma, uint32_t max_steps, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_group_begin", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    const uint32_t frames = ws->n_frames > 1u ? ws->n_frames : 1u;
    if (!poses || !ws->ray_state || !ws->counters || ws->N == 0 || H * W != ws->N || frames > kPMaxFrames || pose_stride < 16u) {
        set_error("gfpp_head_group_begin: needs poses (stride >= 16 floats), ray_state, counters [n_frames, 192], H * W == N, n_frames <= %u", kPMaxFrames);
        return GFPP_EINVAL;
    }
    PremarchArgs p;
    p.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    p.bitfield = model->density_bitfield;
    p.rays_o = nullptr; p.rays_d = nullptr; p.nears = nullptr; p.fars = nullptr;
    p.sample_t = ws->sample_t; p.sample_cnt = ws->sample_cnt;
    p.N = ws->N; p.stride = ws->sample_stride; p.max_samples = max_steps + 7u;
    p.min_near = model->min_near;
    for (int i = 0; i < 6; ++i) p.aabb[i] = model->aabb[i];
    p.nears_out = ws->nears; p.fars_out = ws->fars; p.state = ws->ray_state; p.counters = ws->counters;
    premarch_occupancy(p, model);
    p.frames = frames; p.poses = poses; p.pose_stride = pose_stride; p.W = W;
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy;
    p.rays_o_out = rays_o; p.rays_d_out = rays_d;
    hipLaunchKernelGGL(k_group_begin, dim3(div_up(frames * ws->N, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gfpp_head_group_begin");
}

GFPP_API int gfpp_head_group_resolve(const gfpp_frame_ws *ws, uint32_t max_steps, gfpp_stream_t stream) {
    if (!ws || !ws->ray_state || !ws->counters || !ws->snapshots || ws->N == 0 || ws->gcounters) { set_error("gfpp_head_group_resolve: incomplete workspace"); return GFPP_EINVAL; }
    if (max_steps == 0 || max_steps > 24u) { set_error("gfpp_head_group_resolve: max_steps must be in 1..24"); return GFPP_EUNSUPPORTED; }
    const uint32_t frames = ws->n_frames > 1u ? ws->n_frames : 1u;
    hipLaunchKernelGGL(k_group_budget_resolve, dim3(div_up(frames * ws->N, 256)), dim3(256), 0, (hipStream_t)stream, ws->ray_state, ws->snapshots, ws->counters, ws->N, frames,
                       max_steps);
    return check_launch("gfpp_head_group_resolve");
}

GFPP_API int gfpp_head_frame_trips_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                      float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_frame_trips_lp", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    if (!ws->alive[0] || !ws->alive[1] || !ws->ray_state || !ws->counters || !ws->frame_consts) { set_error("gfpp_head_frame_trips_lp: bad workspace"); return GFPP_EINVAL; }
    LpTripArgs a;
    a.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    { const int rc = lp_model_args("gfpp_head_frame_trips_lp", model, a); if (rc) return rc; }
    a.rays_o = rays_o; a.rays_d = rays_d;
    a.sample_t = ws->sample_t; a.sample_cnt = ws->sample_cnt; a.sample_stride = ws->sample_stride;
    a.state = ws->ray_state;
    a.counters = ws->counters;
    a.gcounters = ws->gcounters ? ws->gcounters : ws->counters;
    a.N_global = ws->gcounters ? ws->N_global : ws->N;
    a.frame_consts = ws->frame_consts;
    a.T_thresh = T_thresh; a.density_scale = model->density_scale;
    a.N = ws->N; a.max_steps = max_steps;
    a.phase_cycles = (unsigned long long *)ws->phase_cycles;
    const uint32_t grid = (uint32_t)lp_cu_count();   // one resident workgroup per CU (LDS-bound), tiles are taken wave-stride
    const hipStream_t st = (hipStream_t)stream;
    const bool bf = model->lp_dtype == GFPP_BF16, slow = (a.pos.any_slow | a.amb.any_slow) != 0, amb3 = model->amb_grid.D == 3;
    void (*launch)(uint32_t, hipStream_t, const LpTripArgs &) =
        amb3 ? (bf ? (slow ? launch_lp<3, __bf16, true> : launch_lp<3, __bf16, false>) : (slow ? launch_lp<3, _Float16, true> : launch_lp<3, _Float16, false>))
             : (bf ? (slow ? launch_lp<2, __bf16, true> : launch_lp<2, __bf16, false>) : (slow ? launch_lp<2, _Float16, true> : launch_lp<2, _Float16, false>));
    a.alive[0] = ws->alive[0]; a.alive[1] = ws->alive[1];
    a.sync = ws->counters + 127;
    a.timeouts = ws->timeouts;
    a.spin_limit = 1u << 22;
    if (tuning().barrier_spins) a.spin_limit = tuning().barrier_spins;   // (tests force a timeout)
    // the first trips one launch each; everything after (rarely reached: the frame-wide n_step doubles as rays die) as one multi-trip launch
    const uint32_t want_separate = ws->separate_trips ? ws->separate_trips : lp_separate_trips();
    // a ray tile of a shared frame: the caller all-reduces the alive counts between trips, so every trip is a launch of its own
    const uint32_t separate = ws->gcounters ? max_steps : (want_separate < max_steps ? want_separate : max_steps);
    const uint32_t first = ws->trip_count ? ws->trip_first : 0u;
    const uint32_t stop = ws->trip_count ? (first + ws->trip_count < max_steps ? first + ws->trip_count : max_steps) : max_steps;
    // trips the caller expects to find nothing left (gfpp_frame_ws.full_grid_trips) get a small grid: such a launch then needs 32 CUs for a
    // moment instead of every CU of the device (a trip workgroup takes a CU's whole LDS); the kernels partition by gridDim, so a late trip
    // that does have work is still rendered, by 32 workgroups
    const uint32_t late_grid = grid < 32u ? grid : 32u;
    for (uint32_t trip = first; trip <= separate && trip < stop; ++trip) {
        a.trip = trip;
        a.trip_end = trip < separate ? trip + 1 : stop;
        launch(ws->full_grid_trips && trip >= ws->full_grid_trips ? late_grid : grid, st, a);
        const int rc = check_launch("gfpp_head_frame_trips_lp");
        if (rc) return rc;
    }
    return 0;
}

template <int AMB_D, typename H, bool SLOW>
static void launch_persist(uint32_t grid, hipStream_t st, const LpTripArgs &a) {
    const PersistArgs<H> pa{a};
    if constexpr (!SLOW) {
        if (a.phase_cycles) {          // the profiling instantiation (tiled-grid models only)
            if (a.n_frames > 1u) hipLaunchKernelGGL((k_head_frame_persist<AMB_D, H, SLOW, true, true>), dim3(grid), dim3(kLpThreads), 0, st, pa);
            else hipLaunchKernelGGL((k_head_frame_persist<AMB_D, H, SLOW, false, true>), dim3(grid), dim3(kLpThreads), 0, st, pa);
            return;
        }
    }
    if (a.n_frames > 1u) hipLaunchKernelGGL((k_head_frame_persist<AMB_D, H, SLOW, true>), dim3(grid), dim3(kLpThreads), 0, st, pa);
    else hipLaunchKernelGGL((k_head_frame_persist<AMB_D, H, SLOW, false>), dim3(grid), dim3(kLpThreads), 0, st, pa);
}

// upper bounds of the local n_step by workgroup round, 4 bits each (gfpp_tuning.persist_caps overrides, experiments).  The take is
// min(128 / rays per wavefront, cap): at 512^2 a workgroup holds ~440 occupied rays and the pool limits it to 2 whatever the cap; the cap matters for
// small shares (256^2: ~110 rays per workgroup), where every round costs a whole block time however few blocks it has -- 4,4,4,8 needs 4 rounds
// instead of 5 there (head pass 0.133 -> 0.122 ms, 1 % more samples evaluated behind rays' ends); 8,8 evaluates 11 % more for nothing
static uint32_t persist_step_caps() {
    if (tuning().persist_caps) return tuning().persist_caps;
    const uint32_t v[8] = {4, 4, 4, 8, 8, 8, 8, 8};
    uint32_t caps = 0;
    for (int k = 0; k < 8; ++k) caps |= v[k] << (4 * k);
    return caps;
}

GFPP_API int gfpp_head_frame_resolve(const gfpp_frame_ws *ws, uint32_t max_steps, gfpp_stream_t stream) {
    if (!ws || !ws->ray_state || !ws->counters || !ws->snapshots || ws->N == 0) { set_error("gfpp_head_frame_resolve: incomplete workspace (ray_state, counters [192], snapshots)"); return GFPP_EINVAL; }
    if (max_steps == 0 || max_steps > 24u) { set_error("gfpp_head_frame_resolve: max_steps must be in 1..24"); return GFPP_EUNSUPPORTED; }
    const int32_t *hist = ws->gcounters ? ws->gcounters : ws->counters + kBudgetBase;
    const uint32_t n_global = ws->gcounters ? ws->N_global : ws->N;
    hipLaunchKernelGGL(k_head_budget_resolve, dim3(div_up(ws->N, 256)), dim3(256), 0, (hipStream_t)stream, ws->ray_state, ws->snapshots, hist, ws->counters, ws->N,
                       n_global, max_steps);
    return check_launch("gfpp_head_frame_resolve");
}

// The launch-control part of the argument record: what the persistent kernel needs whatever the operand type (shared by the 16-bit and the fp32 entry).
static uint32_t persist_control_args(LpTripArgs &a, const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d, float dt_gamma,
                                     uint32_t max_steps, float T_thresh, uint32_t frames) {
    a.mp = make_march_params(model->bound, dt_gamma, max_steps, model->cascade, model->grid_size);
    a.rays_o = rays_o; a.rays_d = rays_d;
    a.sample_t = ws->sample_t; a.sample_cnt = ws->sample_cnt; a.sample_stride = ws->sample_stride;
    a.state = ws->ray_state;
    a.alive[0] = a.alive[1] = nullptr;
    a.counters = ws->counters; a.gcounters = ws->counters; a.N_global = ws->N; a.sync = nullptr; a.timeouts = nullptr;
    a.frame_consts = ws->frame_consts;
    a.T_thresh = T_thresh; a.density_scale = model->density_scale;
    a.N = ws->N; a.max_steps = max_steps; a.trip = 0; a.trip_end = 0;
    a.budget = ws->counters + kBudgetBase;
    a.snaps = ws->snapshots;
    a.n_frames = frames;
    a.consts_stride = ws->frame_consts_stride ? ws->frame_consts_stride : 256u;
    a.tiles_per_frame = div_up(ws->N, kPTile);
    a.n_tiles = frames * a.tiles_per_frame;
    // q -> (q * mult) % n_tiles is a permutation of the tiles when gcd(mult, n_tiles) = 1: consecutive slots land ~1237 tiles apart, so that every
    // workgroup's share (slots b, b + G, ...) is spread over the whole image (equal work without any exchange between workgroups)
    uint32_t grid = (uint32_t)lp_cu_count();
    if (grid > a.n_tiles) grid = a.n_tiles;
    // XCD-local ownership (k_head_frame_persist): needs the pixel order of the rays (row_rays), whole tile columns in eights, one workgroup per CU on whole XCDs,
    // and enough columns per XCD for the comb to balance (8 at 512^2: busiest workgroup 2 % above the mean like before; 4 at 256^2: 12 % instead of 5 %, CPU model in
    // tests/test_persist_budget_cpu.py -- not taken there)
    a.xcd_cols8 = 0u;
    {
        const int mode = tuning().persist_xcd;          // (read at every issue: a captured graph keeps what it was captured with)
        // measured (round 5, 512^2 bf16, four frames per launch, same box): L2 hit rate 80.0 -> 84.4 %, fabric traffic 2.03 -> 1.61 GB per launch, the launch itself
        // +-0 (808 vs 810-821 us) and the clip loop 1 % slower (4 511-4 515 vs 4 554-4 557 frames/s: the comb's workgroup shares are a little less even) -- the kernel is
        // bound by issue and gather LATENCY, and the misses that remain are the ambient grid's, whose coordinates are an MLP output (no image locality to keep): off by
        // default, gfpp_tuning.persist_xcd = 1 turns it on (2: also with four tile columns per XCD, the 256^2 frames)
        const uint32_t W = ws->row_rays;
        if (mode != 0 && W != 0u && W % 64u == 0u && W / 64u >= (mode >= 2 ? 1u : 8u) && ws->N % W == 0u && grid % 8u == 0u && grid == (uint32_t)lp_cu_count() &&
            a.n_tiles / 8u >= grid)
            a.xcd_cols8 = W / 64u;
    }
    const uint32_t nt = a.xcd_cols8 ? a.n_tiles / 8u : a.n_tiles;
    a.tile_mult = 1u;
    for (const uint32_t m : {1237u, 251u, 61u, 7u})
        if (nt % m != 0u && (unsigned long long)nt * m < (1ull << 32)) { a.tile_mult = m; break; }
    a.step_caps = persist_step_caps();
    a.spin_limit = 0;
    return grid;
}

GFPP_API int gfpp_head_frame_persist(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                     float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_frame_persist", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    if (!ws->ray_state || !ws->counters || !ws->frame_consts || !ws->snapshots) {
        set_error("gfpp_head_frame_persist: the workspace needs ray_state, counters [192], frame_consts and snapshots [N, 7, 5]");
        return GFPP_EINVAL;
    }
    if (ws->n_frames > 1u) { set_error("gfpp_head_frame_persist: frame groups are rendered by the 16-bit entry (gfpp_head_frame_persist_lp)"); return GFPP_EUNSUPPORTED; }
    if (max_steps > 24u || (unsigned long long)ws->N > (1ull << kPRayBits)) {
        set_error("gfpp_head_frame_persist: max_steps <= 24 and N <= 2^22 (use gfpp_head_frame_trips beyond)");
        return GFPP_EUNSUPPORTED;
    }
    if (!lp_grid_ok(model->pos_grid, 3) || !(lp_grid_ok(model->amb_grid, 2) || lp_grid_ok(model->amb_grid, 3))) {
        set_error("gfpp_head_frame_persist: grids must be 16-level fp32, position D=3, ambient D in {2,3}");
        return GFPP_EUNSUPPORTED;
    }
    PersistArgs<float> pa{};
    LpTripArgs &a = pa.a;
    const uint32_t grid = persist_control_args(a, model, ws, rays_o, rays_d, dt_gamma, max_steps, T_thresh, 1u);
    a.dbg_ambient = nullptr;
    a.phase_cycles = nullptr;
    TripArgs &t = pa.t;
    t.mp = a.mp;
    t.pos = make_grid_dev(model->pos_grid);
    t.amb = make_grid_dev(model->amb_grid);
    t.w = HeadWeights{(const float4 *)model->amb_w0, (const float4 *)model->amb_w1, (const float4 *)model->sig_w0, (const float4 *)model->sig_w1,
                      (const float4 *)model->sig_w2_geo, (const float4 *)model->col_w0, model->amb_w2, model->sig_w2_sig, model->col_w1};
    t.frame_consts = ws->frame_consts;
    t.density_scale = model->density_scale;
    t.T_thresh = T_thresh;
    t.dbg_ambient = nullptr;
    if (model->amb_grid.D == 3) hipLaunchKernelGGL((k_head_frame_persist<3, float, false, false>), dim3(grid), dim3(kLpThreads), 0, (hipStream_t)stream, pa);
    else hipLaunchKernelGGL((k_head_frame_persist<2, float, false, false>), dim3(grid), dim3(kLpThreads), 0, (hipStream_t)stream, pa);
    const int rc = check_launch("gfpp_head_frame_persist");
    if (rc || ws->gcounters || ws->defer_resolve) return rc;      // (a ray tile of a shared frame: the caller sums the tiles' histograms first)
    return gfpp_head_frame_resolve(ws, max_steps, stream);
}

GFPP_API int gfpp_head_frame_persist_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                        float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    const int bad = lp_check_common("gfpp_head_frame_persist_lp", model, ws, rays_o, rays_d, max_steps);
    if (bad) return bad;
    if (!ws->ray_state || !ws->counters || !ws->frame_consts || !ws->snapshots) {
        set_error("gfpp_head_frame_persist_lp: the workspace needs ray_state, counters [192], frame_consts and snapshots [N, 7, 5]");
        return GFPP_EINVAL;
    }
    const uint32_t frames = ws->n_frames > 1u ? ws->n_frames : 1u;
    if (frames > kPMaxFrames) { set_error("gfpp_head_frame_persist_lp: n_frames must be <= %u", kPMaxFrames); return GFPP_EUNSUPPORTED; }
    if (max_steps > 24u || (unsigned long long)ws->N * frames > (1ull << kPRayBits)) {
        set_error("gfpp_head_frame_persist_lp: max_steps <= 24 and n_frames * N <= 2^22 (use gfpp_head_frame_trips_lp beyond)");
        return GFPP_EUNSUPPORTED;
    }
    if (frames > 1u && ws->gcounters) { set_error("gfpp_head_frame_persist_lp: a frame group cannot be a ray tile of a frame shared between GPUs"); return GFPP_EUNSUPPORTED; }
    LpTripArgs a;
    const uint32_t grid = persist_control_args(a, model, ws, rays_o, rays_d, dt_gamma, max_steps, T_thresh, frames);
    { const int rc = lp_model_args("gfpp_head_frame_persist_lp", model, a); if (rc) return rc; }
    a.phase_cycles = (unsigned long long *)ws->phase_cycles;          // non-null: the instantiation with the phase clocks (k_head_frame_persist<.., PROF>)
    const bool bf = model->lp_dtype == GFPP_BF16, slow = (a.pos.any_slow | a.amb.any_slow) != 0, amb3 = model->amb_grid.D == 3;
    void (*launch)(uint32_t, hipStream_t, const LpTripArgs &) =
        amb3 ? (bf ? (slow ? launch_persist<3, __bf16, true> : launch_persist<3, __bf16, false>) : (slow ? launch_persist<3, _Float16, true> : launch_persist<3, _Float16, false>))
             : (bf ? (slow ? launch_persist<2, __bf16, true> : launch_persist<2, __bf16, false>) : (slow ? launch_persist<2, _Float16, true> : launch_persist<2, _Float16, false>));
    launch(grid, (hipStream_t)stream, a);
    const int rc = check_launch("gfpp_head_frame_persist_lp");
    // a ray tile of a shared frame: the caller sums the histograms of all tiles first; defer_resolve: the consumer kernel resolves; a frame group: the
    // caller resolves (or lets the consumer resolve) every frame with that frame's own workspace record
    if (rc || ws->gcounters || ws->defer_resolve || frames > 1u) return rc;
    return gfpp_head_frame_resolve(ws, max_steps, stream);
}

GFPP_API int gfpp_head_eval_samples_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *positions, const float *directions, uint32_t M,
                                       float *sigma, float *color, float *ambient, gfpp_stream_t stream) {
    if (M == 0) return 0;
    if (!model || !ws || !positions || !directions || !sigma || !color || !ws->frame_consts) { set_error("gfpp_head_eval_samples_lp: null argument"); return GFPP_EINVAL; }
    LpEvalArgs e{};
    LpTripArgs &a = e.t;
    a.mp = make_march_params(model->bound, 0.0f, 16, model->cascade, model->grid_size);
    { const int rc = lp_model_args("gfpp_head_eval_samples_lp", model, a); if (rc) return rc; }
    a.rays_d = directions;
    a.frame_consts = ws->frame_consts;
    a.density_scale = 1.0f;      // forward() returns the unscaled density; render() applies density_scale (renderer.py:376)
    e.positions = positions; e.sigma = sigma; e.color = color; e.ambient = ambient; e.M = M;
    uint32_t grid = div_up(M, 32u * kLpWaves);
    const uint32_t cus = (uint32_t)lp_cu_count();
    if (grid > cus) grid = cus;
    e.waves = kLpWaves;
    const bool bf = model->lp_dtype == GFPP_BF16, slow = (a.pos.any_slow | a.amb.any_slow) != 0, amb3 = model->amb_grid.D == 3;
    void (*launch)(uint32_t, hipStream_t, const LpEvalArgs &) =
        amb3 ? (bf ? (slow ? launch_eval_lp<3, __bf16, true> : launch_eval_lp<3, __bf16, false>) : (slow ? launch_eval_lp<3, _Float16, true> : launch_eval_lp<3, _Float16, false>))
             : (bf ? (slow ? launch_eval_lp<2, __bf16, true> : launch_eval_lp<2, __bf16, false>) : (slow ? launch_eval_lp<2, _Float16, true> : launch_eval_lp<2, _Float16, false>));
    launch(grid, (hipStream_t)stream, e);
    return check_launch("gfpp_head_eval_samples_lp");
}

GFPP_API int gfpp_head_frame_march_lp(const gfpp_head_model *model, const gfpp_frame_ws *ws, const float *rays_o, const float *rays_d,
                                      float dt_gamma, uint32_t max_steps, float T_thresh, gfpp_stream_t stream) {
    const int rc = gfpp_head_frame_premarch(model, ws, rays_o, rays_d, dt_gamma, max_steps, stream);
    if (rc) return rc;
    return gfpp_head_frame_trips_lp(model, ws, rays_o, rays_d, dt_gamma, max_steps, T_thresh, stream);
}
