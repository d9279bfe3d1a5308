// superres.hip -- the StyleGAN2 super-resolution stage of the *_sr models (radnerf_sr.py:14-43: SynthesisBlockNoUp 3 -> 128 @ 256^2,
// SynthesisBlock 128 -> 64 @ 512^2, networks_stylegan2.py:286-478) as three launches of implicit-GEMM convolutions on 16-bit MFMA.
//
// Superresolution feeds ws = ones (radnerf_sr.py:32-33), so every style vector is a constant of the checkpoint: modulation and
// demodulation (networks_stylegan2.py:37-94) are folded into the convolution weights ONCE on the host, in fp64.  What remains per frame:
//   k_sr_conv3<128, FIRST>   block 0: conv 3x3 3 -> 128 (+ noise + bias, lrelu * sqrt 2, clamp; K = 27 padded to 32) computed for the 18 x 18 halo of the patch,
//                            straight into LDS, then conv 3x3 128 -> 128 @ 256^2 + ToRGB 128 -> 3 fused: img256 = rgb_in + clamp(torgb)      K = 1152
//                            (k_sr_first + k_sr_conv3<128> as two launches: the parity partner, GFPP_SR_FUSE_FIRST=0)
//   k_sr_conv3<128, up>      up-conv 128 -> 64, 256^2 -> 512^2: the transposed stride-2 convolution AND the [1,3,3,1] FIR of conv2d_resample.py:
//                            117-133 composed on the host into one 3x3 convolution with 4 x 64 output channels (one set per output phase),
//                            written depth-to-space                                                                              K = 1152, N = 256
//                            (k_sr_up_poly: the same layer un-composed -- polyphase transposed convolution + the FIR as a second GEMM -- opt-in, see below)
//   k_sr_final_resident      conv 3x3 64 -> 64 @ 512^2 + ToRGB 64 -> 3 + upsample2d(img256) fused -> rgb 512^2, one workgroup per CU with the layer's
//                            72 KB of weights resident in LDS (k_sr_conv3<64, final> per patch: the parity partner, GFPP_SR_FINAL_RESIDENT=0)  K = 576
// Activations travel as f16 NHWC (the reference runs both blocks in fp16 on the GPU, use_fp16=True), images as fp32; accumulation fp32.
//
// Kernel shape (k_sr_conv3): a 512-thread workgroup owns a 16x16 output patch; its 18x18 input halo sits in LDS (pixel stride padded by 16 B so
// that the 16 pixels of a row hit distinct banks); each of the 8 wavefronts owns 2 rows = 32 pixels = one 32-column MFMA tile x NT row tiles
// of output channels.  The weights of one (tap, K slice) chunk (pre-packed in fragment order) are double-buffered through LDS by direct
// global -> LDS loads: the next chunk lands while the current one's MFMAs run.
// Where a forward's ~115 us go (tools/sr_phase.py) and why the tap loops stop where they are -- 85-91 % MFMA-pipe occupancy in cycles at the ~1.4 GHz the part
// sustains under dense MFMA (tools/clock_probe_sr.py) -- : DESIGN.md 2.2, docs/LAB_NOTEBOOK.md.  Measured and dropped: 8 x 16 patches for co-residency, the
// 4 x 2 register tile, a three-chunk weight ring with deeper operand prefetch, the last layer's wavefronts in two opposite-phase groups.
#include <cstdlib>
#include <type_traits>

#include <hip/hip_runtime.h>

#include "gfpp_common.h"
#include "lp_mfma_device.h"


namespace gfpp {

constexpr int kSrThreads = 256;
constexpr int kSrPatch = 16;            // output patch side
constexpr int kSrHalo = kSrPatch + 2;   // 3x3 convolution

// ---- noise_mode 'random' inside the kernels: counter-based Philox4x32-10 (Salmon et al., the generator family torch.randn uses on the GPU) -------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

struct SrRng {
    const unsigned long long *state;   // [0] = frame counter, [1] = ticket, [2] = seed word XOR-ed into the key (null: no in-kernel noise)
    unsigned long long seed;
    uint32_t layer;
};

// one unit normal per (pixel, layer, frame): Box-Muller on two of the four Philox words
__device__ __forceinline__ float sr_randn(const SrRng &g, unsigned long long frame, uint32_t pixel) {
    // key = the launch's seed argument XOR the workspace's own seed word (state[2], device memory): a captured graph bakes the argument, the word stays settable
    const unsigned long long seed = g.seed ^ g.state[2];
    const uint4 r = philox4x32_10(make_uint4(pixel, g.layer, (uint32_t)frame, (uint32_t)(frame >> 32)), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853071795864f * u2);
}

enum SrEpilogue { kSrPlain = 0, kSrRgbAdd = 1, kSrUpPhases = 2, kSrFinal = 3 };

struct SrConvArgs {
    const _Float16 *x;        // [H][W][CIN] f16
    const uint4 *w;           // [pass][9 taps][CIN/16][NT][64] fragments
    const float *noise;       // [Hout][Wout] or null
    float noise_strength;
    const float *bias;        // [Cout]
    float act_gain, clamp;    // sqrt(2), 256
    _Float16 *y;              // [Hout][Wout][Cout] f16 (null for kSrFinal)
    uint32_t H, W;            // input = patch-grid resolution
    // ToRGB fused (kSrRgbAdd, kSrFinal)
    const float *w_rgb;       // [Cout][3] modulated 1x1 weights
    const float *b_rgb;       // [3]
    const float *img_in;      // kSrRgbAdd: [H][W][3] fp32 (the NeRF image);  kSrFinal: [H/2][W/2][3] fp32 (img256)
    float *img_out;           // kSrRgbAdd: [H][W][3];  kSrFinal: [H][W][3] fp32 (the 512^2 result)
    float fir[4];             // kSrFinal: 1-D taps of the separable resample filter x up (= [1,3,3,1]/8 * 2)
    SrRng rng;                // noise == null and rng.state != null: unit normals drawn here
    unsigned long long *rng_tick;   // kSrFinal: the last launch of a frame advances the frame counter ([0] counter, [1] ticket)
    uint32_t clamp01;         // kSrFinal: clamp the image to [0, 1]
    uint8_t *u8_out;          // kSrFinal, set inside the kernel from the clip job (k_sr_final_resident): the frame's uint8 slot INSTEAD of img_out, or null
    gfpp_clip_job *job;       // kSrFinal: the clip job (null: none); job_lane's cursor + job_sub = the frame's position, the last workgroup advances the cursor
    uint32_t job_lane, job_sub, job_advance;
    // FIRST (block 0's first convolution computed into the halo patch instead of being read from x): its operands
    const float *first_rgb;   // [H][W][3] fp32, the NeRF image
    const uint4 *first_w;     // [2 steps][4 tiles][64] fragments (SrFirstArgs.w)
    const float *first_bias;  // [128]
    const float *first_noise; // [H][W] or null
    float first_noise_strength;
    SrRng first_rng;
};

__device__ __forceinline__ float sr_act(float v, float gain, float clamp) {
    v = (v >= 0.0f ? v : 0.2f * v) * gain;
    return fminf(fmaxf(v, -clamp), clamp);
}
// The same function on two values in packed fp32 instructions, bit for bit: v g for v >= 0 and (0.2 v) g for v < 0 are both computed (the same roundings as the
// select) and the larger one IS the selected one (v >= 0: v g >= 0.2 v g; v < 0: the other way round); the clamp is one median-of-three.  4.5 vector
// instructions per value instead of 8.4 -- the epilogues of these kernels are vector-ALU bound (tools/sr_phase.py), not matrix bound.
typedef float sr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sr_f32x2 sr_act2(sr_f32x2 v, float gain, float clamp) {
    const sr_f32x2 pos = v * gain, neg = (v * 0.2f) * gain;
    sr_f32x2 o;
    o[0] = __builtin_amdgcn_fmed3f(fmaxf(pos[0], neg[0]), -clamp, clamp);
    o[1] = __builtin_amdgcn_fmed3f(fmaxf(pos[1], neg[1]), -clamp, clamp);
    return o;
}
// four accumulators + the pixel's noise + four channel biases -> activation ((acc + noise) + bias, as everywhere)
__device__ __forceinline__ void sr_act4(const float (&acc)[4], float nz, const float4 &b, float gain, float clamp, float (&out)[4]) {
    const sr_f32x2 lo = sr_act2((sr_f32x2{acc[0], acc[1]} + nz) + sr_f32x2{b.x, b.y}, gain, clamp);
    const sr_f32x2 hi = sr_act2((sr_f32x2{acc[2], acc[3]} + nz) + sr_f32x2{b.z, b.w}, gain, clamp);
    out[0] = lo[0]; out[1] = lo[1]; out[2] = hi[0]; out[3] = hi[1];
}

// One tap's weight fragments (PER_THREAD x 256 x 16 B, already in [step][tile][lane] order) from global memory straight into LDS: the LDS address
// of a direct load is wave-uniform base + lane x 16, which is exactly the fragment layout.
template <int PER_THREAD, int THREADS>
__device__ __forceinline__ void sr_stage_tap(const uint4 *__restrict__ src, uint4 *dst, int tid, int lane) {
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
        const int i = q * THREADS + tid;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i), (__attribute__((address_space(3))) void *)(dst + (i - lane)), 16, 0, 0);
    }
}

// Operand reads of the tap loop as inline assembly.  The loop streams the next weight chunk into LDS with global_load_lds while it multiplies the current
// one; the compiler cannot tell that the ds_read of the CURRENT buffer does not alias the DMA into the OTHER buffer and puts `s_waitcnt vmcnt(0)` in front
// of every chunk's first operand read (visible in the ISA as `stage, wait, MFMAs`): the request for the next chunk was issued and then waited for before
// any multiplying started -- no overlap at all, one exposed L2 -> LDS round trip per chunk (9-18 per workgroup).  A read the compiler does not see as an
// LDS access cannot be given that wait; the counters are then ours to keep: sr_lds_wait ties the freshly read registers to an explicit lgkmcnt(0).
__device__ __forceinline__ uint32_t sr_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ void sr_lds_read128(f16x8 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int NA, int NB2>
__device__ __forceinline__ void sr_lds_wait(f16x8 (&a)[NA], f16x8 (&b)[NB2]) {
    static_assert(NA == 4 || NA == 2, "row tiles per wavefront");
    if constexpr (NA == 4 && NB2 == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0])::"memory");
    else if constexpr (NA == 4 && NB2 == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])::"memory");
    else if constexpr (NA == 2 && NB2 == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1])::"memory");
}

// ---- the image side of a ToRGB epilogue: skip image (block 0: the NeRF image; block 1: upsample2d(img256)) + clamp(torgb + bias) -> image ----------------------
template <int EPI, typename Args>
__device__ __forceinline__ void sr_image_out(const Args &a, const float (&rgb)[3], const float *b_rgb, int Y, int X) {
    float base[3];
    if constexpr (EPI == kSrRgbAdd) {
#pragma unroll
        for (int k = 0; k < 3; ++k) base[k] = a.img_in[((size_t)Y * a.W + X) * 3 + k];
    } else {
        // upsample2d(img256) at (Y, X): zero insertion, [1,3,3,1] FIR, gain 4 (upfirdn2d.py:330-355) = two taps per axis.  The four source
        // pixels are loaded unconditionally from clamped coordinates and zeroed by a select (no branch around a load, see the halo)
        const int h2 = (int)a.H / 2, w2 = (int)a.W / 2;
        const int ya = (Y & 1) ? (Y - 1) / 2 : Y / 2 - 1, xa = (X & 1) ? (X - 1) / 2 : X / 2 - 1;
        const float wy[2] = {(Y & 1) ? a.fir[1] : a.fir[0], (Y & 1) ? a.fir[3] : a.fir[2]};
        const float wx[2] = {(X & 1) ? a.fir[1] : a.fir[0], (X & 1) ? a.fir[3] : a.fir[2]};
        float src[2][2][3];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const int yy = ya + iy, xx = xa + ix;
                const int yc = yy < 0 ? 0 : (yy >= h2 ? h2 - 1 : yy), xc = xx < 0 ? 0 : (xx >= w2 ? w2 - 1 : xx);
                const bool in = yy >= 0 && yy < h2 && xx >= 0 && xx < w2;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = a.img_in[((size_t)yc * w2 + xc) * 3 + k];
                    src[iy][ix][k] = in ? v : 0.0f;
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            base[k] = wy[0] * (wx[0] * src[0][0][k] + wx[1] * src[0][1][k]) + wy[1] * (wx[0] * src[1][0][k] + wx[1] * src[1][1][k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t = fminf(fmaxf(rgb[k] + b_rgb[k], -a.clamp), a.clamp);
        float o = base[k] + t;
        if (EPI == kSrFinal && a.clamp01) o = fminf(fmaxf(o, 0.0f), 1.0f);
        if (EPI == kSrFinal && a.u8_out) a.u8_out[((size_t)Y * a.W + X) * 3 + k] = (uint8_t)fminf(fmaxf(o * 255.0f, 0.0f), 255.0f);   // k_clip_store_u8's expression
        else a.img_out[((size_t)Y * a.W + X) * 3 + k] = o;
    }
}

// the same with `ahead` younger LDS reads allowed to stay in flight (LDS reads return in order; `ahead` must fold to a constant: the callers' loops are unrolled)
template <int NA, int NB2>
__device__ __forceinline__ void sr_lds_wait_n(f16x8 (&a)[NA], f16x8 (&b)[NB2], const int ahead) {
    static_assert(NB2 == 1 && NA == 2, "the operand set of the walk that uses it");
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(ahead) : "memory");
}

// NU = 32-column MFMA tiles per wavefront: 2 (round 1-2) = 4 wavefronts of 64 pixels, one per SIMD; 1 = 8 wavefronts of 32 pixels, two per SIMD --
// a weight fragment then feeds one MFMA instead of two (1.25 KB of LDS operands per MFMA instead of 0.75: still below the LDS's 128 B/clk), but the
// second wavefront of a SIMD runs under the first one's LDS latency, tap barriers and epilogue stores.
// KS = how many slices the input channels are cut into (1 or 2).  With KS = 2 a 128-channel layer keeps only a 64-channel half of its halo patch
// (46 KB) and 16 KB weight chunks in LDS: 80 KB per workgroup, so TWO workgroups share a CU -- one's halo load, tap barriers and epilogue stores run
// under the other's MFMAs (with one 154 KB workgroup per CU those phases were exposed on every CU at the same time).  The accumulators run over both
// halves (18 chunk iterations instead of 9 taps); nothing else changes.
// FIRST: the layer's input is not read from memory but computed in place -- block 0's first convolution (3 -> 128, K = 27; k_sr_first) evaluated
// for the 18 x 18 halo pixels of the patch, straight into the LDS patch: one launch, a 16.8 MB activation write and its 1.27x re-read less per frame.
// Same fragments, same MFMA order, same epilogue as k_sr_first: the values in the patch are the bits k_sr_first would have stored.
template <int CIN, int NT, int EPI, int NU, int KS, bool FIRST = false>
__global__ __launch_bounds__(512 / NU, (NU == 1 ? 2 : 1) * (FIRST ? 1 : KS)) void k_sr_conv3(SrConvArgs a) {   // NU = 2, KS = 2: two 4-wavefront workgroups per CU   // (HIP: second argument = wavefronts per SIMD the register budget must allow)
    constexpr int kSrThreads = 512 / NU;           // (shadows the namespace constant: this kernel's workgroup size)
    typedef LpTraits<_Float16>::vec vec;
    constexpr int CINH = CIN / KS;               // channels of one K slice
    // FIRST computes its input itself, once, for all channels: the whole patch stays in LDS (the launch has one workgroup per CU anyway); the weight chunks
    // and the accumulation order stay those of the K-sliced walk
    constexpr int PCH = FIRST ? CIN : CINH;      // channels of the LDS patch
    constexpr int PS = PCH + 8;                  // pixel stride in halves (16 B of padding: conflict-free ds_read_b128 across a row)
    constexpr int STEPS = CIN / 16, STEPS_H = CINH / 16;
    constexpr int TAPFRAGS = STEPS * NT * 64;    // 16-byte fragments of one tap (all channels)
    constexpr int CHUNKFRAGS = STEPS_H * NT * 64;   // ... of one (tap, K slice) chunk: what is staged through LDS at a time
    constexpr int PER_THREAD = CHUNKFRAGS / kSrThreads;
    static_assert(CHUNKFRAGS % kSrThreads == 0, "chunk weights must split evenly over the workgroup");
    // halo patch | two weight-chunk buffers; after the last chunk the same memory stages the f16 output of the workgroup (a row of 128 halves + 16 B
    // of padding per input-grid pixel), so that it leaves as whole 256-byte rows instead of 8-byte pieces
    // (Measured and dropped for the FIRST instantiation, which has registers and LDS to spare: a ring of three weight chunks with the operand reads three steps
    // ahead of the MFMAs across the chunk barrier -- wavefront 0's walk 11.2 -> 9.4 k-cycles, the launch unchanged: the tap loops run at 85-90 % MFMA-pipe
    // occupancy IN CYCLES already, at the ~1.4 GHz the chip sustains under dense MFMA on every CU (tools/sr_phase.py, tools/clock_probe_sr.py).)
    constexpr int PATCH_BYTES = kSrHalo * kSrHalo * PS * 2, WBUF_BYTES = 2 * CHUNKFRAGS * 16;
    constexpr int SROW = NT * 32 + 8;            // staging row in halves
    constexpr int STAGE_BYTES = (EPI != kSrFinal) ? kSrPatch * kSrPatch * SROW * 2 : 0;
    constexpr int LDS_BYTES = PATCH_BYTES + WBUF_BYTES > STAGE_BYTES ? PATCH_BYTES + WBUF_BYTES : STAGE_BYTES;
    static_assert(PATCH_BYTES % 16 == 0, "weight buffers must stay 16-byte aligned");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    _Float16 *patch = reinterpret_cast<_Float16 *>(lds_raw);
    uint4(*wbuf)[CHUNKFRAGS] = reinterpret_cast<uint4(*)[CHUNKFRAGS]>(lds_raw + PATCH_BYTES);
    _Float16 *stage = reinterpret_cast<_Float16 *>(lds_raw);
    __shared__ float s_rgb[(EPI == kSrRgbAdd || EPI == kSrFinal) ? NT * 32 * 3 + 4 : 1];
    __shared__ __attribute__((aligned(16))) float s_bias[NT * 32];   // this pass's output-channel biases (UpPhases: the 64 channels, twice)
    constexpr int kInSide = kSrHalo + 2;                             // FIRST: the image patch under the halo's own 3 x 3 taps
    __shared__ float s_in[FIRST ? kInSide * kInSide * 3 : 1];
    __shared__ uint4 s_wf[FIRST ? 2 * 4 * 64 : 1];
    __shared__ __attribute__((aligned(16))) float s_fb[FIRST ? 128 : 1];
    __shared__ float s_nz[FIRST ? kSrHalo * kSrHalo : 1];           // FIRST: the first layer's noise term per halo pixel (drawn once, not once per tile and slice)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int x0 = blockIdx.x * kSrPatch, y0 = blockIdx.y * kSrPatch;
    const uint32_t pass = blockIdx.z;
    const uint4 *wg = a.w + (size_t)pass * 9 * TAPFRAGS;
    auto chunk_src = [&](int it) { return wg + (size_t)((it % 9) * STEPS + (it / 9) * STEPS_H) * NT * 64; };   // iteration it = slice * 9 + tap

    // ---- the input halo of one K slice -> LDS (zero padding outside the image) ------------------------------------------------------------------
    // All loads are issued before the first LDS store: the addresses are clamped into the image and the zero padding is a select on the loaded
    // value, so there is no branch around a load (a conditional load makes the compiler wait vmcnt(0) per element: 20 serial memory round
    // trips per thread, ~20 us of the 44 us this kernel took in round 1).
    auto load_patch = [&](int kh) {
        constexpr int HALO_CHUNKS = kSrHalo * kSrHalo * (CINH / 8);
        constexpr int HALO_ITERS = (HALO_CHUNKS + kSrThreads - 1) / kSrThreads;
        uint4 hv[HALO_ITERS];
#pragma unroll
        for (int q = 0; q < HALO_ITERS; ++q) {
            const int i = q * kSrThreads + tid;
            const int ic = i < HALO_CHUNKS ? i : HALO_CHUNKS - 1;
            const int pp = ic / (CINH / 8), c8 = ic % (CINH / 8);
            int py = y0 - 1 + pp / kSrHalo, px = x0 - 1 + pp % kSrHalo;
            py = py < 0 ? 0 : (py >= (int)a.H ? (int)a.H - 1 : py);
            px = px < 0 ? 0 : (px >= (int)a.W ? (int)a.W - 1 : px);
            hv[q] = *reinterpret_cast<const uint4 *>(a.x + ((size_t)py * a.W + px) * CIN + kh * CINH + c8 * 8);
        }
#pragma unroll
        for (int q = 0; q < HALO_ITERS; ++q) {
            const int i = q * kSrThreads + tid;
            if (i < HALO_CHUNKS) {
                const int pp = i / (CINH / 8), c8 = i % (CINH / 8);
                const int py = y0 - 1 + pp / kSrHalo, px = x0 - 1 + pp % kSrHalo;
                const bool in = py >= 0 && py < (int)a.H && px >= 0 && px < (int)a.W;
                *reinterpret_cast<uint4 *>(&patch[pp * PS + c8 * 8]) = in ? hv[q] : make_uint4(0, 0, 0, 0);
            }
        }
    };
    // FIRST: the first convolution at the 324 halo pixels, 32 pixels x 64 channels per unit (11 pixel tiles x 2 channel halves), units dealt out to the wavefronts
    [[maybe_unused]] auto first_patch = [&]() {
        constexpr int HP = kSrHalo * kSrHalo, TILES = (HP + 31) / 32, TL = CINH / 32;
        for (int unit = wave; unit < TILES * KS; unit += kSrThreads / 64) {
            const int tile = unit % TILES, kh = unit / TILES;
            const int pp = tile * 32 + j, ppc = pp < HP ? pp : HP - 1;
            const int hy = ppc / kSrHalo, hx = ppc % kSrHalo;
            v16f facc[TL];
#pragma unroll
            for (int t = 0; t < TL; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) facc[t][r] = 0.0f;
            const vec *Wf = reinterpret_cast<const vec *>(s_wf) + lane;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                vec B;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * s2 + 8 * hi + e;
                    float v = 0.0f;
                    if (k < 27) {
                        const int tap = k / 3, c = k % 3;
                        v = s_in[((hy + tap / 3) * kInSide + hx + tap % 3) * 3 + c];
                    }
                    B[e] = (_Float16)v;
                }
#pragma unroll
                for (int t = 0; t < TL; ++t) facc[t] = LpTraits<_Float16>::mfma(Wf[(s2 * 4 + kh * TL + t) * 64], B, facc[t]);
            }
            const int Y = y0 - 1 + hy, X = x0 - 1 + hx;
            const bool in = Y >= 0 && Y < (int)a.H && X >= 0 && X < (int)a.W;
            const float nz = s_nz[ppc];
            if (pp < HP) {
#pragma unroll
                for (int t = 0; t < TL; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = 32 * t + 8 * q + 4 * hi, ng = kh * CINH + nl;
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        const float av[4] = {facc[t][4 * q], facc[t][4 * q + 1], facc[t][4 * q + 2], facc[t][4 * q + 3]};
                        float v[4];
                        sr_act4(av, nz, *reinterpret_cast<const float4 *>(&s_fb[ng]), a.act_gain, a.clamp, v);
                        h4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = in ? (_Float16)v[e] : (_Float16)0.0f;   // zero padding of THIS layer's input
                        *reinterpret_cast<h4 *>(&patch[pp * PS + ng]) = o;
                    }
            }
        }
    };
    sr_stage_tap<PER_THREAD, kSrThreads>(chunk_src(0), wbuf[0], tid, lane);
    if constexpr (FIRST) {
        for (int i = tid; i < 2 * 4 * 64; i += kSrThreads) s_wf[i] = a.first_w[i];
        if (tid < 128) s_fb[tid] = a.first_bias[tid];
        constexpr int N = kInSide * kInSide * 3, IT = (N + kSrThreads - 1) / kSrThreads;
        float hv[IT];
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * kSrThreads + tid, ic = i < N ? i : N - 1;
            const int p = ic / 3, c = ic % 3;
            int py = y0 - 2 + p / kInSide, px = x0 - 2 + p % kInSide;
            py = py < 0 ? 0 : (py >= (int)a.H ? (int)a.H - 1 : py);
            px = px < 0 ? 0 : (px >= (int)a.W ? (int)a.W - 1 : px);
            hv[q] = a.first_rgb[((size_t)py * a.W + px) * 3 + c];
        }
        // (the noise of the halo pixels is drawn while the image loads are in flight)
        if (tid < kSrHalo * kSrHalo) {
            const int Y = y0 - 1 + tid / kSrHalo, X = x0 - 1 + tid % kSrHalo;
            const bool in = Y >= 0 && Y < (int)a.H && X >= 0 && X < (int)a.W;
            const size_t at = in ? (size_t)Y * a.W + X : 0;
            const unsigned long long fctr = a.first_rng.state ? a.first_rng.state[0] : 0ull;
            s_nz[tid] = a.first_noise ? a.first_noise[at] * a.first_noise_strength
                                      : (a.first_rng.state ? sr_randn(a.first_rng, fctr, (uint32_t)at) * a.first_noise_strength : 0.0f);
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * kSrThreads + tid;
            if (i < N) {
                const int p = i / 3;
                const int py = y0 - 2 + p / kInSide, px = x0 - 2 + p % kInSide;
                // the reference casts the block input to fp16 before the first convolution (superresolution.py:216)
                s_in[i] = (py >= 0 && py < (int)a.H && px >= 0 && px < (int)a.W) ? (float)(_Float16)hv[q] : 0.0f;
            }
        }
        __syncthreads();
        first_patch();
    } else {
        load_patch(0);
    }
    for (int i = tid; i < NT * 32; i += kSrThreads) {
        const int ng = (int)blockIdx.z * NT * 32 + i;
        s_bias[i] = a.bias[EPI == kSrUpPhases ? (ng & 63) : ng];
    }
    if constexpr (EPI == kSrRgbAdd || EPI == kSrFinal) {
        for (int i = tid; i < NT * 32 * 3; i += kSrThreads) s_rgb[i] = a.w_rgb[i];
        if (tid < 3) s_rgb[NT * 32 * 3 + tid] = a.b_rgb[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    v16f acc[NU][NT];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.0f;

    // this lane's NU pixels (one per column tile): rows 2 NU wave + 2 u + (j >> 4), column j & 15 of the patch
    const int prow = 2 * NU * wave + (j >> 4), pcol = j & 15;
    int cur = 0;
    constexpr int ITERS = 9 * KS;
    auto b_base = [&](int it2) {
        const int tap = it2 % 9;
        return sr_lds_addr(&patch[((prow + tap / 3) * kSrHalo + pcol + tap % 3) * PS + (FIRST ? (it2 / 9) * CINH : 0) + 8 * hi]);
    };
    vec Bq[2][NU], Aq[2][NT];
    auto read_ops = [&](uint32_t b0, uint32_t wl, int s2, vec (&B)[NU], vec (&A)[NT]) {   // (sr_lds_read128: see above)
        sr_lds_read128(B[0], b0 + 32u * (uint32_t)s2);
        if constexpr (NU == 2) sr_lds_read128(B[1], b0 + 2u * kSrHalo * PS * 2u + 32u * (uint32_t)s2);
#pragma unroll
        for (int t = 0; t < NT; ++t) sr_lds_read128(A[t], wl + (uint32_t)(s2 * NT + t) * 1024u);
    };
    for (int it = 0; it < ITERS; ++it) {
        const int tap = it % 9;
        if (KS > 1 && !FIRST && it > 0 && tap == 0) {
            // next K slice: every wavefront is done with the old half patch (barrier at the end of the last iteration); its first weight chunk is
            // already in wbuf[cur]
            load_patch(it / 9);
            __syncthreads();
        }
        // the next chunk's fragments go global -> LDS directly (no registers, no ds_write: staged through registers they were spilled to scratch and
        // cost half of the tap loop), into the buffer the previous chunk's MFMAs released at the last barrier; they land while this one computes
        const int nxt = cur ^ 1;
        if (it + 1 < ITERS) sr_stage_tap<PER_THREAD, kSrThreads>(chunk_src(it + 1), wbuf[nxt], tid, lane);
        const uint32_t b0 = b_base(it), wl = sr_lds_addr(wbuf[cur]) + (uint32_t)lane * 16u;
        // operands of step s + 1 are read while the NU x NT MFMAs of step s run
        read_ops(b0, wl, 0, Bq[0], Aq[0]);
        sr_lds_wait<NT, NU>(Aq[0], Bq[0]);
#pragma unroll
        for (int s = 0; s < STEPS_H; ++s) {
            if (s + 1 < STEPS_H) read_ops(b0, wl, s + 1, Bq[(s + 1) & 1], Aq[(s + 1) & 1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = LpTraits<_Float16>::mfma(Aq[s & 1][t], Bq[s & 1][0], acc[0][t]);
                if constexpr (NU == 2) acc[1][t] = LpTraits<_Float16>::mfma(Aq[s & 1][t], Bq[s & 1][1], acc[1][t]);
            }
            if (s + 1 < STEPS_H) sr_lds_wait<NT, NU>(Aq[(s + 1) & 1], Bq[(s + 1) & 1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's part of the chunk being staged has landed in LDS
        __syncthreads();
        cur = nxt;
    }

    // ---- epilogue: noise + bias, leaky relu * gain, clamp; store / ToRGB ----------------------------------------------------------
    // The noise value depends on the output pixel only (for the up-sampling layer: on the pixel and the phase = pair of row tiles), the bias on
    // the channel: both are fetched BEFORE the 2 x NT x 4 store loop (noise: at most 4 loads per lane; bias: LDS).  Inside the loop a
    // conditional global load costs a vmcnt(0) round trip per iteration -- 32 of them were two thirds of this kernel's time in round 1.
    constexpr int NPH = EPI == kSrUpPhases ? (NT + 1) / 2 : 1;
    const unsigned long long frame_ctr = a.rng.state ? a.rng.state[0] : 0ull;
    float nzv[NU][NPH];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            const int Y = y0 + prow + 2 * u, X = x0 + pcol;
            size_t at = (size_t)Y * a.W + X;
            if constexpr (EPI == kSrUpPhases) {
                const int phase = ((int)pass * NT * 32 + 64 * ph) >> 6;
                at = (size_t)(2 * Y + (phase >> 1)) * (2 * a.W) + 2 * X + (phase & 1);
            }
            nzv[u][ph] = a.noise ? a.noise[at] * a.noise_strength : (a.rng.state ? sr_randn(a.rng, frame_ctr, (uint32_t)at) * a.noise_strength : 0.0f);
        }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int Y = y0 + prow + 2 * u, X = x0 + pcol;          // position at the patch-grid (input) resolution
        float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 32 * t + 8 * q + 4 * hi;           // channels n0 .. n0+3 are registers 4 q .. 4 q + 3
                const int ng = (int)pass * NT * 32 + n0;
                float v[4];
                int oy = Y, ox = X, oc = ng;
                uint32_t OW = a.W, OC = NT * 32;
                if constexpr (EPI == kSrUpPhases) {               // depth to space: 4 phases x 64 channels -> (2Y + py, 2X + px)
                    const int phase = ng >> 6;
                    oy = 2 * Y + (phase >> 1); ox = 2 * X + (phase & 1); oc = ng & 63;
                    OW = 2 * a.W; OC = 64;
                }
                const float nz = nzv[u][EPI == kSrUpPhases ? t / 2 : 0];
                const float av[4] = {acc[u][t][4 * q], acc[u][t][4 * q + 1], acc[u][t][4 * q + 2], acc[u][t][4 * q + 3]};
                sr_act4(av, nz, *reinterpret_cast<const float4 *>(&s_bias[n0]), a.act_gain, a.clamp, v);

                if constexpr (EPI != kSrFinal) {
                    // into the wavefront's own rows of the staging area (every wavefront is past the last chunk's barrier: patch and weight buffers are free)
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                    *reinterpret_cast<h4 *>(stage + ((wave * NU + u) * 32 + j) * SROW + n0) = o;
                    (void)oy; (void)ox; (void)oc; (void)OW; (void)OC;
                }
                if constexpr (EPI == kSrRgbAdd || EPI == kSrFinal) {
                    // the next stage sees the f16-rounded activation in the reference too (x stays fp16 through ToRGB)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xv = (float)(_Float16)v[e];
#pragma unroll
                        for (int k = 0; k < 3; ++k) rgb[k] = fmaf(xv, s_rgb[(n0 + e) * 3 + k], rgb[k]);
                    }
                }
            }
        }
        if constexpr (EPI == kSrRgbAdd || EPI == kSrFinal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb[k] += __shfl_xor(rgb[k], 32);
            if (hi == 0) sr_image_out<EPI>(a, rgb, &s_rgb[NT * 32 * 3], Y, X);
        }
    }
    if constexpr (EPI != kSrFinal) {
        // the wavefront's 32 NU pixels leave as rows: 16 lanes x 16 B = the 256 contiguous bytes of one input-grid pixel (plain layers: its 128 channels;
        // up-sampling layer: the two output pixels (2Y + pass, 2X) and (2Y + pass, 2X + 1) x 64 channels, adjacent in memory), four pixels per store
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        static_assert(NT == 4, "a staged row is 128 halves = 16 lanes x 16 B");
#pragma unroll
        for (int r = 0; r < 8 * NU; ++r) {
            const int pl = r * 4 + (lane >> 4), chunk = lane & 15;          // pixel of the wavefront, 16-byte chunk of its row
            const int uu = pl >> 5, jj = pl & 31;
            const int Y = y0 + 2 * NU * wave + 2 * uu + (jj >> 4), X = x0 + (jj & 15);
            const uint4 row = *reinterpret_cast<const uint4 *>(stage + (wave * NU * 32 + pl) * SROW + chunk * 8);
            const size_t base = EPI == kSrUpPhases ? ((size_t)(2 * Y + (int)pass) * (2 * a.W) + 2 * X) * 64 : ((size_t)Y * a.W + X) * 128;
            *reinterpret_cast<uint4 *>(a.y + base + chunk * 8) = row;
        }
    }
    if constexpr (EPI == kSrFinal) {
        // the frame's last launch: when its last workgroup is done -- every workgroup of the frame has read the counter by then -- the next frame begins
        if (a.rng_tick) {
            __syncthreads();
            if (tid == 0) {
                const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y * gridDim.z;
                if (atomicAdd(&a.rng_tick[1], 1ull) == total - 1ull) {
                    a.rng_tick[1] = 0ull;
                    __threadfence();
                    atomicAdd(&a.rng_tick[0], 1ull);
                }
            }
        }
    }
}

// ---- block 1's last layer (64 -> 64 @ 512^2 + ToRGB + image up-sampling) with its weights RESIDENT -------------------------------------------------------------
// The layer's folded weights are 72 KB: they fit in LDS next to one 46 KB halo patch.  One workgroup per CU loads them once and walks over its share of the
// 1 024 patches: no weight chunk and no barrier inside a patch's 72 MFMAs per wavefront (the per-patch launch had nine barrier-separated 8 KB chunks: 5.6 us of
// tap loop for 1.9 us of matrix work, tools/sr_phase.py), the index arithmetic of the halo load done once, and the NEXT patch's halo in flight (registers) under
// the current patch's taps and epilogue.  The 36-step walk keeps its operands DEPTH steps ahead of the MFMAs: a step is only two MFMAs (64 cycles) and an LDS
// read under eight wavefronts' traffic takes longer than that.
// (Measured and dropped: the eight wavefronts as two groups of four on 16 x 8 half patches in opposite phases -- one multiplies while the other runs its epilogue
// -- 34.5 us per launch against 31.5: a wavefront's walk and its epilogue are latency chains, not throughput, and halving the wavefronts per phase halves what
// hides them.)
// Same fragments, same tap / step order, same epilogue arithmetic per pixel as k_sr_conv3<64, 2, kSrFinal>: the same bits (that launch stays as the A/B partner,
// GFPP_SR_FINAL_RESIDENT=0; tests/test_kernels_gpu.py compares the two).
__global__ __launch_bounds__(512, 2) void k_sr_final_resident(SrConvArgs a) {
    typedef LpTraits<_Float16>::vec vec;
    if (a.job) {
        // the frame's slot in the clip job's output ring (every workgroup reads the cursor before the launch's LAST workgroup moves it, below); a position beyond
        // the job (the dry runs of a graph capture, the padding frames of a last group) renders for nothing: its stores go to the fp32 image as without a job
        const uint32_t pos = a.job->cursor[a.job_lane] + a.job_sub;
        a.u8_out = pos < a.job->n ? a.job->out + (size_t)(pos % a.job->ring_frames) * a.job->frame_bytes : nullptr;
    }
    constexpr int CIN = 64, NT = 2, STEPS = CIN / 16, PS = CIN + 8, TAPFRAGS = STEPS * NT * 64, THREADS = 512;
    constexpr int HALO_CHUNKS = kSrHalo * kSrHalo * (CIN / 8), HALO_ITERS = (HALO_CHUNKS + THREADS - 1) / THREADS;
    __shared__ __attribute__((aligned(16))) uint4 wall[9 * TAPFRAGS];
    __shared__ __attribute__((aligned(16))) _Float16 patch[kSrHalo * kSrHalo * PS];
    __shared__ float s_rgb[NT * 32 * 3 + 4];
    __shared__ __attribute__((aligned(16))) float s_bias[NT * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int n_px = (int)a.W / kSrPatch, n_patches = n_px * ((int)a.H / kSrPatch);

    sr_stage_tap<9 * TAPFRAGS / THREADS, THREADS>(a.w, wall, tid, lane);
    for (int i = tid; i < NT * 32; i += THREADS) s_bias[i] = a.bias[i];
    for (int i = tid; i < NT * 32 * 3; i += THREADS) s_rgb[i] = a.w_rgb[i];
    if (tid < 3) s_rgb[NT * 32 * 3 + tid] = a.b_rgb[tid];

    // this thread's chunks of a halo: (row, column) in the 18 x 18 patch and the 8-channel group, the same for every patch
    // (one packed word per chunk: halo pixel | 8-channel group << 9 | row << 12 | column << 17 -- four separate arrays cost the walk its registers)
    int hpk[HALO_ITERS];
#pragma unroll
    for (int q = 0; q < HALO_ITERS; ++q) {
        const int i = q * THREADS + tid, ic = i < HALO_CHUNKS ? i : HALO_CHUNKS - 1;
        const int pp = ic / (CIN / 8), c8 = ic % (CIN / 8);
        hpk[q] = pp | (c8 << 9) | ((pp / kSrHalo) << 12) | ((pp % kSrHalo) << 17);
    }
    uint4 hv[HALO_ITERS];
    auto issue = [&](int p) {        // unconditional loads from clamped coordinates (no branch around a load, see k_sr_conv3)
        const int y0 = (p / n_px) * kSrPatch - 1, x0 = (p % n_px) * kSrPatch - 1;
#pragma unroll
        for (int q = 0; q < HALO_ITERS; ++q) {
            int py = y0 + ((hpk[q] >> 12) & 31), px = x0 + ((hpk[q] >> 17) & 31);
            py = py < 0 ? 0 : (py >= (int)a.H ? (int)a.H - 1 : py);
            px = px < 0 ? 0 : (px >= (int)a.W ? (int)a.W - 1 : px);
            hv[q] = *reinterpret_cast<const uint4 *>(a.x + ((size_t)py * a.W + px) * CIN + ((hpk[q] >> 9) & 7) * 8);
        }
    };
    auto commit = [&](int p) {       // zero padding outside the image by select
        const int y0 = (p / n_px) * kSrPatch - 1, x0 = (p % n_px) * kSrPatch - 1;
#pragma unroll
        for (int q = 0; q < HALO_ITERS; ++q) {
            if (q * THREADS + tid < HALO_CHUNKS) {
                const int py = y0 + ((hpk[q] >> 12) & 31), px = x0 + ((hpk[q] >> 17) & 31);
                const bool in = py >= 0 && py < (int)a.H && px >= 0 && px < (int)a.W;
                *reinterpret_cast<uint4 *>(&patch[(hpk[q] & 511) * PS + ((hpk[q] >> 9) & 7) * 8]) = in ? hv[q] : make_uint4(0, 0, 0, 0);
            }
        }
    };

    const unsigned long long frame_ctr = a.rng.state ? a.rng.state[0] : 0ull;
    const int prow = 2 * wave + (j >> 4), pcol = j & 15;
    const uint32_t wl = sr_lds_addr(wall) + (uint32_t)lane * 16u;
    int p = (int)blockIdx.x;
    if (p < n_patches) issue(p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the weights have landed (and the first halo)
    for (; p < n_patches; p += (int)gridDim.x) {
        commit(p);
        if (p + (int)gridDim.x < n_patches) issue(p + (int)gridDim.x);
        __syncthreads();
        const int y0 = (p / n_px) * kSrPatch, x0 = (p % n_px) * kSrPatch;

        v16f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        constexpr int DEPTH = 3, RING = DEPTH + 1, TOTAL = 9 * STEPS;
        vec Bq[RING][1], Aq[RING][NT];
        auto read_step = [&](int gs, vec (&B)[1], vec (&A)[NT]) {       // gs = tap * STEPS + s
            const int tap = gs / STEPS, s2 = gs % STEPS;
            const uint32_t b0 = sr_lds_addr(&patch[((prow + tap / 3) * kSrHalo + pcol + tap % 3) * PS + 8 * hi]) + 32u * (uint32_t)s2;
            sr_lds_read128(B[0], b0);
#pragma unroll
            for (int t = 0; t < NT; ++t) sr_lds_read128(A[t], wl + (uint32_t)(gs * NT + t) * 1024u);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) read_step(d, Bq[d], Aq[d]);
#pragma unroll
        for (int gs = 0; gs < TOTAL; ++gs) {
            if (gs + DEPTH < TOTAL) read_step(gs + DEPTH, Bq[(gs + DEPTH) % RING], Aq[(gs + DEPTH) % RING]);
            // the reads of the steps after this one may stay in flight: (NT + 1) each
            const int ahead = (TOTAL - 1 - gs < DEPTH ? TOTAL - 1 - gs : DEPTH) * (NT + 1);
            sr_lds_wait_n<NT, 1>(Aq[gs % RING], Bq[gs % RING], ahead);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = LpTraits<_Float16>::mfma(Aq[gs % RING][t], Bq[gs % RING][0], acc[t]);
        }

        // epilogue of k_sr_conv3<64, 2, kSrFinal>: noise + bias, leaky relu * gain, clamp; ToRGB on the f16-rounded activation; image
        const int Y = y0 + prow, X = x0 + pcol;
        const size_t at = (size_t)Y * a.W + X;
        const float nz = a.noise ? a.noise[at] * a.noise_strength : (a.rng.state ? sr_randn(a.rng, frame_ctr, (uint32_t)at) * a.noise_strength : 0.0f);
        float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 32 * t + 8 * q + 4 * hi;
                const float av[4] = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                float v[4];
                sr_act4(av, nz, *reinterpret_cast<const float4 *>(&s_bias[n0]), a.act_gain, a.clamp, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xv = (float)(_Float16)v[e];
#pragma unroll
                    for (int k = 0; k < 3; ++k) rgb[k] = fmaf(xv, s_rgb[(n0 + e) * 3 + k], rgb[k]);
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] += __shfl_xor(rgb[k], 32);
        if (hi == 0) sr_image_out<kSrFinal>(a, rgb, &s_rgb[NT * 32 * 3], Y, X);
        __syncthreads();                 // every wavefront is done with the patch
    }
    // the frame's last launch: when its last workgroup is done -- every workgroup of the frame has read the counter by then -- the next frame begins
    if (a.rng_tick && tid == 0) {
        if (atomicAdd(&a.rng_tick[1], 1ull) == (unsigned long long)gridDim.x - 1ull) {
            a.rng_tick[1] = 0ull;
            __threadfence();
            atomicAdd(&a.rng_tick[0], 1ull);
        }
    }
    // ... and the clip job's cursor of this lane moves on (k_clip_store_u8's hand-over, raymarch.hip)
    if (a.job && a.job_advance != 0xFFFFFFFFu && tid == 0 && atomicAdd(&a.job->ticket[a.job_lane], 1u) == gridDim.x - 1u) {
        a.job->ticket[a.job_lane] = 0u;
        a.job->cursor[a.job_lane] = a.job->cursor[a.job_lane] + (a.job_advance ? a.job_advance : a.job->lanes);
    }
}

// ---- block 1's up-sampling layer (128 -> 64, 256^2 -> 512^2) in POLYPHASE form (round 6) -----------------------------------------------------------------------
// conv2d_resample(up = 2) is a stride-2 transposed 3 x 3 convolution followed by the [1,3,3,1] x [1,3,3,1] FIR (conv2d_resample.py:117-133).  k_sr_conv3<128, up>
// composes both into ONE 3 x 3 convolution with 4 x 64 output channels: one implicit GEMM, but 36 tap matrices per low-resolution pixel where the transposed
// convolution has 9 (38.6 of the stage's 77.3 GFLOP).  Here the two operations stay apart and BOTH run on the matrix pipe:
//   (1) T = the transposed convolution in polyphase form on the (PW + 2) x (PH + 2) grid of low-resolution positions m of a PW x PH patch (T_hi[2 m + p] = T_p[m] per
//       axis; T_e[m] = x[m] w[0] + x[m - 1] w[2], T_o[m] = x[m] w[1]): four accumulators per position tile -- ee (4 taps), eo (2), oe (2), oo (1) -- fed from four
//       SHIFTED reads of the halo patch, 9 tap matrices in all.  Operands swapped with respect to the other layers (A = positions x channels from the patch, B = the
//       weight fragments, same packing): D = positions x output channels, i.e. a lane holds ONE output channel at 16 positions and can write T channel-major;
//   (2) T goes to LDS as f16 -- the reference's own intermediate is an fp16 tensor (x stays fp16 through conv_transpose2d and upfirdn2d) -- laid out
//       [channel][py][my][px][mx] over the memory the halo patch and the weight chunks no longer need;
//   (3) the FIR is a second GEMM: for one high-resolution output row, out[channel][column] = sum_k T[channel][k] G[k][column] over the two runs of two T rows
//       (2 x (72 + 8 pad) entries) the row's four y taps touch; G holds the products of the taps {1/4, 3/4}^2 (exact in f16), a [2][5][64][8] table from the host
//       (gfpp_sr_model.up_fir_g, radnerfs/superres.py::_fir_gemm_table) that a lane loads behind the products.
//       A = T (a lane reads 8 consecutive entries of its channel: two ds_read_b64), B = G: D = channels x 32 output columns -- a lane holds one output pixel's 16
//       channels, which is the layout the activation epilogue of the other layers works on.
// MFMAs per low-resolution pixel: (9 x 8 x 8 tiles + 24 rows x 10) x 2 channel halves / 192 = 8.5 instead of 18; one workgroup = one 16 x 12 patch x 32 output
// channels, 73.9 KB of LDS (the halo patch of one K slice + all nine tap matrices of the slice; T overlays both): two per CU.  Same noise / bias / activation / clamp
// expressions as k_sr_conv3's epilogue; not the same bits as the composed layer (T is rounded to f16 here, and the sums associate differently): compared with it and with
// the oracle by tolerance (tests/test_kernels_gpu.py); the host tables through a numpy restatement of this data flow: tests/test_sr_polyphase_cpu.py.
// STATUS: opt-in (gfpp_tuning.sr_up_poly, default 0).  37-39 us against the composed launch's 38.6-43.7 -- but kernels of OTHER streams that share a CU with this launch's
// MFMA phase were measured to return different bits now and then (include/gfpp_radnerf.h at the field, docs/LAB_NOTEBOOK.md round 6, tools/clip_interference.py).
constexpr int kUpPW = 16, kUpPH = 12, kUpGW = kUpPW + 2, kUpGH = kUpPH + 2, kUpPos = kUpGW * kUpGH;      // 18 x 14 = 252 positions = 8 tiles of 32 (4 idle rows)
constexpr int kUpRow = 2 * kUpGW;                        // one T row of a channel: [px][mx] = 36 entries
constexpr int kUpChStride = 2056;                        // bytes between channels of T: 2 x 14 x 36 x 2 = 2016, padded so that 32 lanes' 8-byte reads hit 64 distinct banks

struct SrUpArgs {
    const _Float16 *x;        // [H][W][128] f16
    const uint4 *w;           // [2 channel halves][2 K slices][9 taps in chunk order][4 steps][64] fragments
    const float *noise;       // [2H][2W] or null
    float noise_strength;
    const float *bias;        // [64]
    float act_gain, clamp;
    _Float16 *y;              // [2H][2W][64] f16
    uint32_t H, W;
    const _Float16 *g;        // the FIR as a GEMM operand (gfpp_sr_model.up_fir_g)
    SrRng rng;
    unsigned long long *prof; // PROF instantiation only (tools/sr_up_phases.py): [workgroup][8] shader-clock stamps of thread 0 at the phase boundaries
};

template <bool PROF>
__global__ __launch_bounds__(512, 4) void k_sr_up_poly(SrUpArgs a) {
    auto stamp = [&](int k) {
        if constexpr (PROF) {
            if (threadIdx.x == 0) a.prof[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + k] = __builtin_readcyclecounter();
        }
    };
    stamp(0);
    typedef LpTraits<_Float16>::vec vec;
    constexpr int PS = 64 + 8;                                           // halves per position in the patch (one 64-channel K slice + 16 B of padding)
    constexpr int PATCH_BYTES = 256 * PS * 2, WBUF_BYTES = 9 * 4 * 64 * 16, T_BYTES = 32 * kUpChStride + 32;      // all nine taps of one K slice resident (36 KB)
    constexpr int LDS_BYTES = PATCH_BYTES + WBUF_BYTES > T_BYTES ? PATCH_BYTES + WBUF_BYTES : T_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    __shared__ __attribute__((aligned(16))) float s_bias[32];
    _Float16 *patch = reinterpret_cast<_Float16 *>(lds_raw);
    uint4 *wall = reinterpret_cast<uint4 *>(lds_raw + PATCH_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
    const int x0 = (int)blockIdx.x * kUpPW, y0 = (int)blockIdx.y * kUpPH;
    const int nt = (int)blockIdx.z;                                      // which 32 of the 64 output channels
    // The weights of a K slice are its nine tap matrices (4 steps x 64 fragments each), grouped by the input shift they multiply: taps 0-3 (shift 0,0), 4-5 (x - 1),
    // 6-7 (y - 1), 8 (both).  ALL nine of the current slice are resident; the group a wavefront has finished with is overwritten by the same group of the NEXT
    // slice right behind the barrier that ends it, so that nobody ever waits for a weight transfer in flight: the first cut of this kernel streamed one group
    // at a time through a ring of two and spent its time there -- a group is 4-16 MFMAs per wavefront, an L2 -> LDS transfer 1-2 us (k_sr_up_poly 37 us against the
    // composed layer's 39-41 with HALF the MFMAs).
    auto stage_taps = [&](int ks, int tap0, int ntap) {
        const uint4 *src = a.w + ((size_t)(nt * 2 + ks) * 9 + tap0) * 4 * 64;
        uint4 *dst = wall + tap0 * 4 * 64;
        for (int i = tid; i < ntap * 256; i += 512)                      // (whole wavefronts: 256 fragments per tap)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i), (__attribute__((address_space(3))) void *)(dst + (i - lane)), 16, 0, 0);
    };
    auto load_patch = [&](int ks, int tid) {                             // the halo of one K slice: unconditional loads from clamped coordinates, zero padding by select
        constexpr int CH = kUpPos * 8, IT = (CH + 511) / 512;
        uint4 hv[IT];
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * 512 + tid, ic = i < CH ? i : CH - 1;
            const int pp = ic >> 3, c8 = ic & 7;
            int py = y0 - 1 + pp / kUpGW, px = x0 - 1 + pp % kUpGW;
            py = py < 0 ? 0 : (py >= (int)a.H ? (int)a.H - 1 : py);
            px = px < 0 ? 0 : (px >= (int)a.W ? (int)a.W - 1 : px);
            hv[q] = *reinterpret_cast<const uint4 *>(a.x + ((size_t)py * a.W + px) * 128 + ks * 64 + c8 * 8);
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * 512 + tid;
            if (i < CH) {
                const int pp = i >> 3, c8 = i & 7;
                const int py = y0 - 1 + pp / kUpGW, px = x0 - 1 + pp % kUpGW;
                const bool in = py >= 0 && py < (int)a.H && px >= 0 && px < (int)a.W;
                *reinterpret_cast<uint4 *>(&patch[pp * PS + c8 * 8]) = in ? hv[q] : make_uint4(0, 0, 0, 0);
            }
        }
    };

    stage_taps(0, 0, 9);
    load_patch(0, tid);
    if (tid < 32) s_bias[tid] = a.bias[nt * 32 + tid];
    if (tid < 4 * PS * 2 / 16) reinterpret_cast<uint4 *>(&patch[kUpPos * PS])[tid] = make_uint4(0, 0, 0, 0);      // the four idle rows of the last tile read something finite
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(1);                                                            // first patch slice + all nine taps of slice 0 have arrived

    // ---- (1) the polyphase products: wavefront w owns the positions [32 w, 32 w + 32) ------------------------------------------------------------------------
    v16f acc[4];                                                         // ee, eo, oe, oo
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    const int pos = 32 * wave + j;                                       // this lane's row of the A operand
    // which accumulator a chunk's tap feeds: shift (0,0): w[0][0] ee, w[0][1] eo, w[1][0] oe, w[1][1] oo; shift x-1: w[0][2] ee, w[1][2] oe; shift y-1: w[2][0] ee, w[2][1] eo;
    // shift (y-1, x-1): w[2][2] ee
    constexpr int kShift[4] = {0, -1, -kUpGW, -kUpGW - 1};
    // one group = one shift of the patch (compile-time: tap count and accumulators), K steps of 16 channels
    auto run_group = [&](auto shc) {
        constexpr int SH = decltype(shc)::value, NTAP = SH == 0 ? 4 : (SH == 3 ? 1 : 2), TB = SH == 0 ? 0 : (SH == 1 ? 4 : (SH == 2 ? 6 : 8));
        int ps = pos + kShift[SH];
        ps = ps < 0 ? 0 : ps;                                            // (only positions whose T entries nobody reads)
        const uint32_t a0 = sr_lds_addr(&patch[ps * PS + 8 * hi]), wl = sr_lds_addr(wall + TB * 4 * 64) + (uint32_t)lane * 16u;
        // operands of step s + 1 are read while the MFMAs of step s run (the reads are inline assembly -- see sr_lds_read128 --: a wait names the registers it guards)
        vec A[2], B[2][NTAP];
        auto read_ops = [&](int s, vec &Aq, vec (&Bq)[NTAP]) {
            sr_lds_read128(Aq, a0 + 32u * (uint32_t)s);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) sr_lds_read128(Bq[t], wl + (uint32_t)(t * 4 + s) * 1024u);
        };
        auto wait_ops = [&](vec &Aq, vec (&Bq)[NTAP]) {
            if constexpr (NTAP == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq), "+v"(Bq[0]), "+v"(Bq[1]), "+v"(Bq[2]), "+v"(Bq[3])::"memory");
            else if constexpr (NTAP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq), "+v"(Bq[0]), "+v"(Bq[1])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Aq), "+v"(Bq[0])::"memory");
        };
        // (the four-tap group reads a step's operands and multiplies them: with its 20 operand registers double-buffered next to the 64 accumulators the kernel
        // spilled at the 128 registers that two workgroups per CU allow; the other wavefronts of the SIMD cover the LDS latency there)
        constexpr bool AHEAD = NTAP < 4;
        read_ops(0, A[0], B[0]);
        wait_ops(A[0], B[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = AHEAD ? (s & 1) : 0, nxt = AHEAD ? ((s + 1) & 1) : 0;
            if (AHEAD && s + 1 < 4) read_ops(s + 1, A[nxt], B[nxt]);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                constexpr int kPh[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0};
                const int ph = kPh[TB + t];                              // (folds: TB and t are constants after unrolling)
                if (ph == 0) acc[0] = LpTraits<_Float16>::mfma(A[cur], B[cur][t], acc[0]);
                else if (ph == 1) acc[1] = LpTraits<_Float16>::mfma(A[cur], B[cur][t], acc[1]);
                else if (ph == 2) acc[2] = LpTraits<_Float16>::mfma(A[cur], B[cur][t], acc[2]);
                else acc[3] = LpTraits<_Float16>::mfma(A[cur], B[cur][t], acc[3]);
            }
            if (s + 1 < 4) {
                if (!AHEAD) read_ops(s + 1, A[0], B[0]);
                wait_ops(A[nxt], B[nxt]);
            }
        }
    };
    // K slice 0; behind every group the same group of slice 1 starts to arrive in its place
    run_group(std::integral_constant<int, 0>{});
    __syncthreads();
    stage_taps(1, 0, 4);
    run_group(std::integral_constant<int, 1>{});
    __syncthreads();
    stage_taps(1, 4, 2);
    run_group(std::integral_constant<int, 2>{});
    __syncthreads();
    stage_taps(1, 6, 2);
    run_group(std::integral_constant<int, 3>{});
    __syncthreads();                                                     // everybody is done with slice 0's patch too
    stamp(2);
    stage_taps(1, 8, 1);
    load_patch(1, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(3);
    run_group(std::integral_constant<int, 0>{});
    run_group(std::integral_constant<int, 1>{});
    run_group(std::integral_constant<int, 2>{});
    run_group(std::integral_constant<int, 3>{});
    __syncthreads();                                                     // patch and weights are dead from here on
    stamp(4);

    // ---- (2) T -> LDS, channel-major f16 (the patch and the weight buffers are dead: every wavefront is past the last barrier) -----------------------------------
    // lane (channel j, hi) holds, in register 4 q + e of accumulator (py, px), position 32 w + 8 q + 4 hi + e
    {
        unsigned char *tch = lds_raw + j * kUpChStride;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 32 * wave + 8 * q + 4 * hi + e;
                if (p < kUpPos) {
                    const int my = p / kUpGW, mx = p - my * kUpGW;
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        const int py = ph >> 1, px = ph & 1;
                        *reinterpret_cast<_Float16 *>(tch + ((py * kUpGH + my) * kUpRow + px * kUpGW + mx) * 2) = (_Float16)acc[ph][4 * q + e];
                    }
                }
            }
        // a run's eight padding entries are the next row's first eight -- or, behind a channel's last row, the channel's own padding: zero, so that 0 x it is 0
        if (wave == 0 && hi == 0)
#pragma unroll
            for (int k = 0; k < (kUpChStride - 2 * kUpGH * kUpRow * 2) / 4; ++k) reinterpret_cast<uint32_t *>(tch + 2 * kUpGH * kUpRow * 2)[k] = 0u;
    }
    __syncthreads();
    stamp(5);

    // ---- (3) FIR as a GEMM + the activation epilogue: wavefront w owns the high-resolution rows 3 w .. 3 w + 2 of the patch's 24 ---------------------------------
    // G for output column j of the patch (cell X = j / 2, parity b): x taps at T columns 2 X + b - 1 .. + 2, i.e. [px][mx] entries
    //   b = 0: (1, X) g0, (0, X + 1) g1, (1, X + 1) g2, (0, X + 2) g3        b = 1: (0, X + 1) g0, (1, X + 1) g1, (0, X + 2) g2, (1, X + 2) g3
    // (mx counts from the patch's halo column: cell X is mx = X + 1).  A run = two T rows (72 entries) + 8 entries of padding with coefficient 0.
    // G comes from the host (gfpp_sr_model.up_fir_g: [2 pairs][5 steps][64 lanes][8] f16): entry k = 16 s + 8 hi + e of a run is T row r = k / 36 (2 = padding,
    // coefficient 0), [px][mx] = k % 36; its coefficient is (row tap of the pair: pair 0 = (g1, g3), pair 1 = (g0, g2)) x (column tap of output column j), see
    // radnerfs/superres.py::_fir_gemm_table.  Loaded here, behind the products, so that its 40 registers are not live under the 64 accumulators.
    {
    vec G[2][5];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int s = 0; s < 5; ++s) G[pr][s] = *reinterpret_cast<const vec *>(a.g + ((size_t)(pr * 5 + s) * 64 + lane) * 8);
    stamp(6);
    const unsigned long long frame_ctr = a.rng.state ? a.rng.state[0] : 0ull;
    const unsigned char *tA = lds_raw + j * kUpChStride + 16 * hi;       // this lane's row of the A operand: channel j, entries 8 hi .. of a step
    // the noise of this lane's three output pixels (rows 3 w .. 3 w + 2, column j): a pixel's value is needed by both lane halves, so each half draws what the other
    // does not -- half 0 rows 0 and 2, half 1 row 1 and (again) 2 -- and they swap: two Philox + Box-Muller evaluations per lane instead of three (a third of this
    // phase's vector instructions)
    float nzr[3];
    {
        const int Xo = 2 * x0 + j;
        auto noise_at = [&](int il) -> float {
            const int Yo = 2 * y0 + il;
            const bool in = Yo < 2 * (int)a.H && Xo < 2 * (int)a.W;
            const size_t at = in ? (size_t)Yo * (2 * a.W) + Xo : 0;
            return a.noise ? a.noise[at] * a.noise_strength : (a.rng.state ? sr_randn(a.rng, frame_ctr, (uint32_t)at) * a.noise_strength : 0.0f);
        };
        const float mine = noise_at(3 * wave + hi), last = noise_at(3 * wave + 2);
        const float other = __shfl_xor(mine, 32);
        nzr[0] = hi ? other : mine;
        nzr[1] = hi ? mine : other;
        nzr[2] = last;
    }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int il = 3 * wave + rr, Y = il >> 1, av = il & 1, my = Y + 1;
        // rows of the two runs: parity 0: py = 0 rows (my, my + 1) with (g1, g3), py = 1 rows (my - 1, my) with (g0, g2); parity 1: py = 0 rows (my, my + 1) with (g0, g2),
        // py = 1 rows (my, my + 1) with (g1, g3)
        const int run0 = (0 * kUpGH + my) * kUpRow * 2, run1 = (1 * kUpGH + (av ? my : my - 1)) * kUpRow * 2;
        v16f d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 lo0 = *reinterpret_cast<const h4 *>(tA + run0 + 32 * s), hi0 = *reinterpret_cast<const h4 *>(tA + run0 + 32 * s + 8);
            const h4 lo1 = *reinterpret_cast<const h4 *>(tA + run1 + 32 * s), hi1 = *reinterpret_cast<const h4 *>(tA + run1 + 32 * s + 8);
            const vec A0 = __builtin_shufflevector(lo0, hi0, 0, 1, 2, 3, 4, 5, 6, 7), A1 = __builtin_shufflevector(lo1, hi1, 0, 1, 2, 3, 4, 5, 6, 7);
            d = LpTraits<_Float16>::mfma(A0, av ? G[1][s] : G[0][s], d);
            d = LpTraits<_Float16>::mfma(A1, av ? G[0][s] : G[1][s], d);
        }
        // epilogue: lane (output column j, hi) holds channels 8 q + 4 hi + e in register 4 q + e
        const int Yo = 2 * y0 + il, Xo = 2 * x0 + j;
        if (Yo < 2 * (int)a.H && Xo < 2 * (int)a.W) {
            const size_t at = (size_t)Yo * (2 * a.W) + Xo;
            const float nz = nzr[rr];
            _Float16 *dst = a.y + at * 64 + nt * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 8 * q + 4 * hi;
                const float av4[4] = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
                float v[4];
                sr_act4(av4, nz, *reinterpret_cast<const float4 *>(&s_bias[n0]), a.act_gain, a.clamp, v);
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                *reinterpret_cast<h4 *>(dst + n0) = o;
            }
        }
    }
    }
    stamp(7);
}

// ---- first layer: 3 -> 128, K = 27 padded to 32 --------------------------------------------------------------------------------------
struct SrFirstArgs {
    const float *rgb;      // [H][W][3] fp32
    const uint4 *w;        // [2 steps][4 tiles][64] fragments; k = 16 s + 8 h + e -> (tap = k / 3, channel = k % 3), k >= 27 zero
    const float *noise;
    float noise_strength;
    const float *bias;     // [128]
    float act_gain, clamp;
    _Float16 *y;           // [H][W][128]
    uint32_t H, W;
    SrRng rng;
};

__global__ __launch_bounds__(kSrThreads) void k_sr_first(SrFirstArgs a) {
    typedef LpTraits<_Float16>::vec vec;
    __shared__ float patch[kSrHalo * kSrHalo * 3];
    __shared__ uint4 wl[2 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float s_bias[128];
    constexpr int SROW = 128 + 8;                                   // staging row in halves (see k_sr_conv3)
    __shared__ __attribute__((aligned(16))) _Float16 stage[kSrPatch * kSrPatch * SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int x0 = blockIdx.x * kSrPatch, y0 = blockIdx.y * kSrPatch;
    for (int i = tid; i < 2 * 4 * 64; i += kSrThreads) wl[i] = a.w[i];
    if (tid < 128) s_bias[tid] = a.bias[tid];
    {
        // unconditional loads from clamped coordinates, zero padding by select (no branch around a load)
        constexpr int N = kSrHalo * kSrHalo * 3, IT = (N + kSrThreads - 1) / kSrThreads;
        float hv[IT];
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * kSrThreads + tid, ic = i < N ? i : N - 1;
            const int p = ic / 3, c = ic % 3;
            int py = y0 - 1 + p / kSrHalo, px = x0 - 1 + p % kSrHalo;
            py = py < 0 ? 0 : (py >= (int)a.H ? (int)a.H - 1 : py);
            px = px < 0 ? 0 : (px >= (int)a.W ? (int)a.W - 1 : px);
            hv[q] = a.rgb[((size_t)py * a.W + px) * 3 + c];
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int i = q * kSrThreads + tid;
            if (i < N) {
                const int p = i / 3;
                const int py = y0 - 1 + p / kSrHalo, px = x0 - 1 + p % kSrHalo;
                // the reference casts the block input to fp16 before the first convolution (superresolution.py:216)
                patch[i] = (py >= 0 && py < (int)a.H && px >= 0 && px < (int)a.W) ? (float)(_Float16)hv[q] : 0.0f;
            }
        }
    }
    __syncthreads();
    const int prow = 4 * wave + (j >> 4), pcol = j & 15;
    v16f acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.0f;
    const vec *W = reinterpret_cast<const vec *>(wl) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        vec B[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s + 8 * hi + e;
                float v = 0.0f;
                if (k < 27) {
                    const int tap = k / 3, c = k % 3;
                    v = patch[((prow + 2 * u + tap / 3) * kSrHalo + pcol + tap % 3) * 3 + c];
                }
                B[u][e] = (_Float16)v;
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const vec A = W[(s * 4 + t) * 64];
            acc[0][t] = LpTraits<_Float16>::mfma(A, B[0], acc[0][t]);
            acc[1][t] = LpTraits<_Float16>::mfma(A, B[1], acc[1][t]);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int Y = y0 + prow + 2 * u, X = x0 + pcol;
        const float nz = a.noise ? a.noise[(size_t)Y * a.W + X] * a.noise_strength
                                 : (a.rng.state ? sr_randn(a.rng, a.rng.state[0], (uint32_t)(Y * a.W + X)) * a.noise_strength : 0.0f);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 32 * t + 8 * q + 4 * hi;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)sr_act(acc[u][t][4 * q + e] + nz + s_bias[n0 + e], a.act_gain, a.clamp);
                *reinterpret_cast<h4 *>(stage + ((wave * 2 + u) * 32 + j) * SROW + n0) = o;
            }
    }
    // whole 256-byte pixel rows leave the workgroup (16 lanes x 16 B per pixel, four pixels per store) instead of 8-byte pieces
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pl = r * 4 + (lane >> 4), chunk = lane & 15;
        const int uu = pl >> 5, jj = pl & 31;
        const int Y = y0 + 4 * wave + 2 * uu + (jj >> 4), X = x0 + (jj & 15);
        const uint4 row = *reinterpret_cast<const uint4 *>(stage + (wave * 64 + pl) * SROW + chunk * 8);
        *reinterpret_cast<uint4 *>(a.y + ((size_t)Y * a.W + X) * 128 + chunk * 8) = row;
    }
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_sr_forward(const gfpp_sr_model *m, const gfpp_sr_ws *ws, const float *rgb_in, const float *const noise[4], float *rgb_out,
                             gfpp_stream_t stream) {
    if (!m || !ws || !rgb_in || !rgb_out) { set_error("gfpp_sr_forward: null argument"); return GFPP_EINVAL; }
    if (!m->w_first || !m->w_b0c1 || !m->w_up || !m->w_b1c1 || !m->rgb0_w || !m->rgb1_w || !ws->x0 || !ws->x1 || !ws->x2 || !ws->img256) {
        set_error("gfpp_sr_forward: incomplete model / workspace");
        return GFPP_EINVAL;
    }
    const hipStream_t st = (hipStream_t)stream;
    const uint32_t R = 256;
    const float gain = 1.4142135623730951f, clamp = m->conv_clamp;
    const bool draw = !noise && ws->rng_state;                      // noise_mode 'random' inside the kernels
    // Shapes: 8 wavefronts of 32 pixels (NU = 1), the 128-channel layers in two K slices (KS = 2).  The other shapes the kernel template describes -- 4 wavefronts
    // of 64 pixels, the whole 128-channel patch in LDS -- were the round-2 / round-3 A/B partners (GFPP_SR_TILES, GFPP_SR_KSLICES: measured, docs/LAB_NOTEBOOK.md)
    // and are no longer instantiated.
    auto rng_of = [&](uint32_t layer) { return SrRng{draw ? (const unsigned long long *)ws->rng_state : nullptr, (unsigned long long)ws->rng_seed, layer}; };
    const bool fuse_first = tuning().sr_fuse_first != 0;            // block 0's first convolution inside the second one's halo load (gfpp_tuning.sr_fuse_first = 0: its own launch, the parity partner)
    if (!fuse_first) {
        SrFirstArgs a{rgb_in, (const uint4 *)m->w_first, noise ? noise[0] : nullptr, m->noise_strength[0], m->bias[0], gain, clamp, (_Float16 *)ws->x0, R, R, rng_of(0)};
        hipLaunchKernelGGL(k_sr_first, dim3(R / kSrPatch, R / kSrPatch), dim3(kSrThreads), 0, st, a);
        const int rc = check_launch("gfpp_sr_forward(block0.conv0)");
        if (rc) return rc;
    }
    {
        SrConvArgs a{};
        a.x = (const _Float16 *)ws->x0; a.w = (const uint4 *)m->w_b0c1; a.noise = noise ? noise[1] : nullptr; a.noise_strength = m->noise_strength[1];
        a.bias = m->bias[1]; a.act_gain = gain; a.clamp = clamp; a.y = (_Float16 *)ws->x1; a.H = R; a.W = R;
        a.w_rgb = m->rgb0_w; a.b_rgb = m->rgb0_b; a.img_in = rgb_in; a.img_out = ws->img256;
        a.rng = rng_of(1);
        if (fuse_first) {
            a.first_rgb = rgb_in; a.first_w = (const uint4 *)m->w_first; a.first_bias = m->bias[0];
            a.first_noise = noise ? noise[0] : nullptr; a.first_noise_strength = m->noise_strength[0]; a.first_rng = rng_of(0);
            hipLaunchKernelGGL((k_sr_conv3<128, 4, kSrRgbAdd, 1, 2, true>), dim3(R / kSrPatch, R / kSrPatch, 1), dim3(512), 0, st, a);
        } else hipLaunchKernelGGL((k_sr_conv3<128, 4, kSrRgbAdd, 1, 2>), dim3(R / kSrPatch, R / kSrPatch, 1), dim3(512), 0, st, a);
        const int rc = check_launch("gfpp_sr_forward(block0.conv1 + torgb)");
        if (rc) return rc;
    }
    if (m->w_up_poly && m->up_fir_g && tuning().sr_up_poly) {
        SrUpArgs a{};
        a.x = (const _Float16 *)ws->x1; a.w = (const uint4 *)m->w_up_poly; a.noise = noise ? noise[2] : nullptr; a.noise_strength = m->noise_strength[2];
        a.bias = m->bias[2]; a.act_gain = gain; a.clamp = clamp; a.y = (_Float16 *)ws->x2; a.H = R; a.W = R;
        a.g = (const _Float16 *)m->up_fir_g;
        a.rng = rng_of(2);
        a.prof = (unsigned long long *)ws->up_prof;
        if (a.prof) hipLaunchKernelGGL(k_sr_up_poly<true>, dim3(R / kUpPW, (R + kUpPH - 1) / kUpPH, 2), dim3(512), 0, st, a);
        else hipLaunchKernelGGL(k_sr_up_poly<false>, dim3(R / kUpPW, (R + kUpPH - 1) / kUpPH, 2), dim3(512), 0, st, a);
        const int rc = check_launch("gfpp_sr_forward(block1.conv0 up, polyphase)");
        if (rc) return rc;
    } else {
        SrConvArgs a{};
        a.x = (const _Float16 *)ws->x1; a.w = (const uint4 *)m->w_up; a.noise = noise ? noise[2] : nullptr; a.noise_strength = m->noise_strength[2];
        a.bias = m->bias[2]; a.act_gain = gain; a.clamp = clamp; a.y = (_Float16 *)ws->x2; a.H = R; a.W = R;
        a.rng = rng_of(2);
        hipLaunchKernelGGL((k_sr_conv3<128, 4, kSrUpPhases, 1, 2>), dim3(R / kSrPatch, R / kSrPatch, 2), dim3(512), 0, st, a);
        const int rc = check_launch("gfpp_sr_forward(block1.conv0 up)");
        if (rc) return rc;
    }
    {
        SrConvArgs a{};
        a.x = (const _Float16 *)ws->x2; a.w = (const uint4 *)m->w_b1c1; a.noise = noise ? noise[3] : nullptr; a.noise_strength = m->noise_strength[3];
        a.bias = m->bias[3]; a.act_gain = gain; a.clamp = clamp; a.y = nullptr; a.H = 2 * R; a.W = 2 * R;
        a.w_rgb = m->rgb1_w; a.b_rgb = m->rgb1_b; a.img_in = ws->img256; a.img_out = rgb_out;
        for (int k = 0; k < 4; ++k) a.fir[k] = m->fir[k];
        a.rng = rng_of(3);
        a.rng_tick = draw ? (unsigned long long *)ws->rng_state : nullptr;
        a.clamp01 = ws->clamp01;
        const bool resident = tuning().sr_final_resident != 0;      // gfpp_tuning.sr_final_resident = 0: one workgroup per patch, weights streamed (A/B runs, parity partner)
        if (ws->clip_job && !resident) { set_error("gfpp_sr_forward: the uint8 store into a clip job needs the resident last layer (gfpp_tuning.sr_final_resident)"); return GFPP_EUNSUPPORTED; }
        a.job = ws->clip_job; a.job_lane = ws->clip_lane; a.job_sub = ws->clip_sub; a.job_advance = ws->clip_advance;
        if (resident) {
            const int cus = cu_count();
            const int patches = (int)(2 * R / kSrPatch) * (int)(2 * R / kSrPatch);
            hipLaunchKernelGGL(k_sr_final_resident, dim3(cus < patches ? cus : patches), dim3(512), 0, st, a);
        } else hipLaunchKernelGGL((k_sr_conv3<64, 2, kSrFinal, 1, 1>), dim3(2 * R / kSrPatch, 2 * R / kSrPatch, 1), dim3(512), 0, st, a);
        const int rc = check_launch("gfpp_sr_forward(block1.conv1 + torgb)");
        if (rc) return rc;
    }
    return 0;
}
