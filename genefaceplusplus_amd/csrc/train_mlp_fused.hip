// train_mlp_fused.hip -- a whole bias-free ReLU MLP of the radiance field (cond_encoder.py:183-202 `MLP`: ambient_net / sigma_net / color_net,
// radnerf.py:60-100) over one training batch as ONE forward launch and ONE backward launch under `amp: true` (half operands, fp32 accumulation:
// what autocast's F.linear + F.relu computes layer by layer, utils/commons/trainer.py wraps the step in torch.autocast).
//
// The per-layer path was, per layer and direction, one BLAS GEMM plus the activation / cast kernels around it: ~10 launches per layer, every
// intermediate [M, 128] matrix written and read again by each of them (M = the step's ~3 x 10^5 samples).  Here a wavefront owns 32 rows (samples)
// and walks all layers with the accumulators in registers, the weights of ALL layers resident in LDS (<= 112 KB), and memory sees
//
//   forward : X [M, 32 TIN] read once; relu(hidden) [NL - 1][M, 128] (saved for the backward pass) and out [M, 32 TOUT] written once
//   backward: dOut and the saved activations read once; the masked hidden gradients G_l = (W_{l+1}^T G_{l+1}) * [act_l > 0] [NL - 1][M, 128] and dX written once
//
// The weight gradients dW_l = G_l^T X_l stay with the split-M kernel (train_mlp.hip), which streams exactly these matrices.
//
// Layout: lp_mfma_device.h's 32x32x16 fragments.  The weight image puts the MFMA row i of tile t on feature 32 t + swap23(i) (bits 2 and 3 of i
// exchanged): with that, the eight accumulator values that lane (j, h) packs into operand (step s, element e) of the next layer are the features
// 16 s + 8 h + e -- the natural order -- so an operand register quad IS 16 contiguous bytes of a row-major [M, features] matrix: every load and store of this
// file is one 16-byte access per lane, and the split-M kernel, torch and the caller all see plain row-major matrices.
#include <hip/hip_runtime.h>

#include "gfpp_common.h"
#include "lp_mfma_device.h"

namespace gfpp {

constexpr int kFmThreads = 512;
constexpr int kFmHidden = 128;

struct FmArgs {
    const f16x8 *in;       // forward: X [M][4 TIN] 16-byte vectors; backward: dOut [M][4 TOUT]
    const f16x8 *image;    // the direction's weight image (k_mlp_train_pack)
    f16x8 *hidden;         // forward: relu(hidden) out, [NL - 1][M][16]; backward: G out, same shape
    const f16x8 *acts;     // backward: the forward pass's relu(hidden)
    f16x8 *out;            // forward: out [M][4 TOUT]; backward: dX [M][4 TIN] or null
    uint32_t M, n_blocks;
};

__device__ __forceinline__ uint32_t swap23(uint32_t i) { return (i & ~12u) | ((i & 4u) << 1) | ((i & 8u) >> 1); }

// ---- the two weight images of one MLP from its fp32 parameters, one launch --------------------------------------------------------------------------------
struct FmPackArgs {
    const float *w[4];     // layer l: [out_l, in_l] row-major fp32 (nn.Linear.weight)
    uint32_t n_layers, in_features, out_features, tin, tout;
    _Float16 *fwd, *bwd;
};

// element (fragment f, lane, e) of an image whose layers are listed as (matrix, transposed?, steps, tiles): value = A[row(t, i)][16 s + 8 h + e]
__global__ __launch_bounds__(256) void k_mlp_train_pack(FmPackArgs a) {
    const uint32_t NL = a.n_layers;
    const uint32_t frag_fwd = 2 * a.tin * 4 + (NL - 2) * 32 + 8 * a.tout, frag_bwd = 2 * a.tout * 4 + (NL - 2) * 32 + 8 * a.tin;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n_fwd = frag_fwd * 512, n_bwd = frag_bwd * 512;
    if (q >= n_fwd + n_bwd) return;
    const bool bw = q >= n_fwd;
    uint32_t p = bw ? q - n_fwd : q;
    const uint32_t e = p & 7, lane = (p >> 3) & 63;
    uint32_t f = p >> 9;
    const uint32_t i = lane & 31, h = lane >> 5;
    // which layer of the walk, its steps x tiles
    uint32_t layer = 0, T = 4, fl = f;
    for (uint32_t k = 0; k < NL; ++k) {
        // walk order: forward 0 .. NL-1; backward NL-1 .. 0
        const uint32_t l = bw ? NL - 1 - k : k;
        uint32_t steps, tiles;
        if (!bw) { steps = l == 0 ? 2 * a.tin : 8; tiles = l == NL - 1 ? a.tout : 4; }
        else { steps = l == NL - 1 ? 2 * a.tout : 8; tiles = l == 0 ? a.tin : 4; }
        if (fl < steps * tiles) { layer = l; T = tiles; break; }
        fl -= steps * tiles;
    }
    const uint32_t s = fl / T, t = fl % T;
    const uint32_t row = 32 * t + swap23(i), col = 16 * s + 8 * h + e;
    const uint32_t in_l = layer == 0 ? a.in_features : kFmHidden, out_l = layer == NL - 1 ? a.out_features : kFmHidden;
    float v = 0.0f;
    if (!bw) { if (row < out_l && col < in_l) v = a.w[layer][(size_t)row * in_l + col]; }
    else { if (row < in_l && col < out_l) v = a.w[layer][(size_t)col * in_l + row]; }          // the transposed layer: rows = its inputs, K = its outputs
    (bw ? a.bwd : a.fwd)[p] = (_Float16)v;
}

// no activation: fp32 accumulators -> the operand / row-major order (act_pack<_, T, 0> spelled with paired conversions)
template <int T>
__device__ __forceinline__ void plain_pack(const v16f (&acc)[T], f16x8 (&b)[2 * T]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef LpTraits<_Float16>::pair pair;
#pragma unroll
    for (int s = 0; s < 2 * T; ++s)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const pair p = __builtin_convertvector((f32x2){acc[s >> 1][8 * (s & 1) + e], acc[s >> 1][8 * (s & 1) + e + 1]}, pair);
            b[s][e] = p[0];
            b[s][e + 1] = p[1];
        }
}

template <int T>
__device__ __forceinline__ void zero_acc(v16f (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

template <int TIN, int NL, int TOUT>
struct FmShape {
    static constexpr int frag_fwd = 2 * TIN * 4 + (NL - 2) * 32 + 8 * TOUT;
    static constexpr int frag_bwd = 2 * TOUT * 4 + (NL - 2) * 32 + 8 * TIN;
};

// ---- forward ---------------------------------------------------------------------------------------------------------------------------------------------------
template <int TIN, int NL, int TOUT>
__global__ __launch_bounds__(kFmThreads, kFmThreads / 256) void k_mlp_train_fwd(FmArgs a) {
    typedef _Float16 H;
    constexpr int NS0 = 2 * TIN, FR = FmShape<TIN, NL, TOUT>::frag_fwd;
    __shared__ __attribute__((aligned(16))) f16x8 img[FR * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    for (int q = tid; q < FR * 64; q += kFmThreads) img[q] = a.image[q];
    __syncthreads();
    const uint32_t stride = gridDim.x * (kFmThreads / 64);
    uint32_t blk = blockIdx.x * (kFmThreads / 64) + wave;
    if (blk >= a.n_blocks) return;
    const uint32_t last_row = a.M - 1;
    f16x8 b0[NS0];
    {
        const uint32_t m = blk * 32 + j, mr = m < last_row ? m : last_row;
        const f16x8 *xr = a.in + (size_t)mr * (2 * NS0) + h;
#pragma unroll
        for (int s = 0; s < NS0; ++s) b0[s] = xr[2 * s];
    }
    for (; blk < a.n_blocks; blk += stride) {
        const uint32_t m = blk * 32 + j;
        const bool live = m < a.M;
        v16f acc[4];
        zero_acc<4>(acc);
        mfma_layer_lds<H, NS0, 4>(acc, img, b0, lane);
        // the next block's rows: in flight under this block's remaining layers
        if (blk + stride < a.n_blocks) {
            const uint32_t mn = (blk + stride) * 32 + j, mr = mn < last_row ? mn : last_row;
            const f16x8 *xr = a.in + (size_t)mr * (2 * NS0) + h;
#pragma unroll
            for (int s = 0; s < NS0; ++s) b0[s] = xr[2 * s];
        }
        const f16x8 *wl = img + NS0 * 4 * 64;
#pragma unroll
        for (int l = 1; l + 1 < NL; ++l) {
            f16x8 keep[8];
            v16f nxt[4];
            mfma_layer_lds_fused<H>(nxt, wl, acc, lane, &keep);
            if (live) {
                f16x8 *dst = a.hidden + ((size_t)(l - 1) * a.M + m) * 16 + h;
#pragma unroll
                for (int s = 0; s < 8; ++s) dst[2 * s] = keep[s];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = nxt[t];
            wl += 32 * 64;
        }
        f16x8 b[8];
        act_pack<H, 4, 1>(acc, b);
        if (live) {
            f16x8 *dst = a.hidden + ((size_t)(NL - 2) * a.M + m) * 16 + h;
#pragma unroll
            for (int s = 0; s < 8; ++s) dst[2 * s] = b[s];
        }
        v16f oacc[TOUT];
        zero_acc<TOUT>(oacc);
        mfma_layer_lds<H, 8, TOUT>(oacc, wl, b, lane);
        f16x8 ob[2 * TOUT];
        plain_pack<TOUT>(oacc, ob);
        if (live) {
            f16x8 *dst = a.out + (size_t)m * (4 * TOUT) + h;
#pragma unroll
            for (int s = 0; s < 2 * TOUT; ++s) dst[2 * s] = ob[s];
        }
    }
}

// ---- backward: the input-gradient chain ------------------------------------------------------------------------------------------------------------------------
// G = round_f16(acc) where the saved activation is positive, else 0 (relu's backward on the rounded half gradient, as autocast's layer-by-layer graph computes it)
__device__ __forceinline__ void masked_pack(const v16f (&acc)[4], const f16x8 (&act)[8], f16x8 (&g)[8]) {
    f16x8 raw[8];
    plain_pack<4>(acc, raw);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const s16x8 keep = act[s] > (f16x8)(_Float16)0.0f;          // all-ones lanes where the activation passed
        g[s] = __builtin_bit_cast(f16x8, (s16x8)(__builtin_bit_cast(s16x8, raw[s]) & keep));
    }
}

template <int TIN, int NL, int TOUT>
__global__ __launch_bounds__(kFmThreads, kFmThreads / 256) void k_mlp_train_bwd(FmArgs a) {
    typedef _Float16 H;
    constexpr int NSO = 2 * TOUT, FR = FmShape<TIN, NL, TOUT>::frag_bwd;
    __shared__ __attribute__((aligned(16))) f16x8 img[FR * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    for (int q = tid; q < FR * 64; q += kFmThreads) img[q] = a.image[q];
    __syncthreads();
    const uint32_t stride = gridDim.x * (kFmThreads / 64);
    const uint32_t last_row = a.M - 1;
    for (uint32_t blk = blockIdx.x * (kFmThreads / 64) + wave; blk < a.n_blocks; blk += stride) {
        const uint32_t m = blk * 32 + j, mr = m < last_row ? m : last_row;
        const bool live = m < a.M;
        f16x8 bo[NSO];
        {
            const f16x8 *gr = a.in + (size_t)mr * (2 * NSO) + h;
#pragma unroll
            for (int s = 0; s < NSO; ++s) bo[s] = gr[2 * s];
        }
        // the saved activation of the last hidden layer: requested before the matrix work that needs it
        f16x8 act[8];
        {
            const f16x8 *ar = a.acts + ((size_t)(NL - 2) * a.M + mr) * 16 + h;
#pragma unroll
            for (int s = 0; s < 8; ++s) act[s] = ar[2 * s];
        }
        v16f acc[4];
        zero_acc<4>(acc);
        mfma_layer_lds<H, NSO, 4>(acc, img, bo, lane);
        const f16x8 *wl = img + NSO * 4 * 64;
        f16x8 g[8];
        masked_pack(acc, act, g);
#pragma unroll
        for (int l = NL - 2; l >= 1; --l) {
            // G_l is final: store it, then G_{l-1} = (W_l^T G_l) * [act_{l-1} > 0]
            const f16x8 *ar = a.acts + ((size_t)(l - 1) * a.M + mr) * 16 + h;
#pragma unroll
            for (int s = 0; s < 8; ++s) act[s] = ar[2 * s];
            if (live) {
                f16x8 *dst = a.hidden + ((size_t)l * a.M + m) * 16 + h;
#pragma unroll
                for (int s = 0; s < 8; ++s) dst[2 * s] = g[s];
            }
            zero_acc<4>(acc);
            mfma_layer_lds<H, 8, 4>(acc, wl, g, lane);
            masked_pack(acc, act, g);
            wl += 32 * 64;
        }
        if (live) {
            f16x8 *dst = a.hidden + (size_t)m * 16 + h;
#pragma unroll
            for (int s = 0; s < 8; ++s) dst[2 * s] = g[s];
        }
        if (a.out) {                                                      // workgroup-uniform
            v16f xacc[TIN];
            zero_acc<TIN>(xacc);
            mfma_layer_lds<H, 8, TIN>(xacc, wl, g, lane);
            f16x8 xb[2 * TIN];
            plain_pack<TIN>(xacc, xb);
            if (live) {
                f16x8 *dst = a.out + (size_t)m * (4 * TIN) + h;
#pragma unroll
                for (int s = 0; s < 2 * TIN; ++s) dst[2 * s] = xb[s];
            }
        }
    }
}

template <int TIN, int NL, int TOUT>
static void fm_launch(bool backward, const FmArgs &a, uint32_t grid, hipStream_t st) {
    if (backward) hipLaunchKernelGGL((k_mlp_train_bwd<TIN, NL, TOUT>), dim3(grid), dim3(kFmThreads), 0, st, a);
    else hipLaunchKernelGGL((k_mlp_train_fwd<TIN, NL, TOUT>), dim3(grid), dim3(kFmThreads), 0, st, a);
}

template <int TIN, int NL>
static bool fm_pick_out(bool backward, uint32_t tout, const FmArgs &a, uint32_t grid, hipStream_t st) {
    if (tout == 1) fm_launch<TIN, NL, 1>(backward, a, grid, st);
    else if (tout == 5) fm_launch<TIN, NL, 5>(backward, a, grid, st);
    else return false;
    return true;
}

template <int TIN>
static bool fm_pick_layers(bool backward, uint32_t nl, uint32_t tout, const FmArgs &a, uint32_t grid, hipStream_t st) {
    if (nl == 2) return fm_pick_out<TIN, 2>(backward, tout, a, grid, st);
    if (nl == 3) return fm_pick_out<TIN, 3>(backward, tout, a, grid, st);
    return false;
}

static bool fm_pick(bool backward, uint32_t tin, uint32_t nl, uint32_t tout, const FmArgs &a, uint32_t grid, hipStream_t st) {
    if (tin == 2) return fm_pick_layers<2>(backward, nl, tout, a, grid, st);
    if (tin == 3) return fm_pick_layers<3>(backward, nl, tout, a, grid, st);
    if (tin == 5) return fm_pick_layers<5>(backward, nl, tout, a, grid, st);
    return false;
}

static int fm_shape_ok(const char *who, uint32_t in_pad, uint32_t hidden, uint32_t n_layers, uint32_t out_pad) {
    const bool ok = hidden == (uint32_t)kFmHidden && (n_layers == 2 || n_layers == 3) && (in_pad == 64 || in_pad == 96 || in_pad == 160) && (out_pad == 32 || out_pad == 160);
    if (!ok) {
        set_error("%s: built for hidden 128, 2 or 3 layers, padded input width 64 / 96 / 160, padded output width 32 / 160 (got hidden %u, %u layers, %u -> %u)", who, hidden,
                  n_layers, in_pad, out_pad);
        return GFPP_EUNSUPPORTED;
    }
    return 0;
}

static uint32_t fm_grid(uint32_t n_blocks) {
    const int cus = cu_count();
    const uint32_t want = div_up(n_blocks, kFmThreads / 64);
    return want < (uint32_t)cus ? want : (uint32_t)cus;
}

}  // namespace gfpp

using namespace gfpp;

GFPP_API int gfpp_mlp_train_pack(const float *const *weights, uint32_t n_layers, uint32_t in_features, uint32_t hidden, uint32_t out_features, uint32_t in_pad,
                                 uint32_t out_pad, void *fwd_image, void *bwd_image, gfpp_stream_t stream) {
    const char *who = "gfpp_mlp_train_pack";
    if (!weights || !fwd_image || !bwd_image) { set_error("%s: null argument", who); return GFPP_EINVAL; }
    int rc = fm_shape_ok(who, in_pad, hidden, n_layers, out_pad);
    if (rc) return rc;
    if (in_features == 0 || in_features > in_pad || out_features == 0 || out_features > out_pad) { set_error("%s: in / out features must fit their padded widths", who); return GFPP_EINVAL; }
    FmPackArgs a;
    for (uint32_t l = 0; l < 4; ++l) a.w[l] = l < n_layers ? weights[l] : nullptr;
    for (uint32_t l = 0; l < n_layers; ++l)
        if (!a.w[l]) { set_error("%s: null weight matrix", who); return GFPP_EINVAL; }
    a.n_layers = n_layers; a.in_features = in_features; a.out_features = out_features; a.tin = in_pad / 32; a.tout = out_pad / 32;
    a.fwd = static_cast<_Float16 *>(fwd_image); a.bwd = static_cast<_Float16 *>(bwd_image);
    const uint32_t frags = (2 * a.tin * 4 + (n_layers - 2) * 32 + 8 * a.tout) + (2 * a.tout * 4 + (n_layers - 2) * 32 + 8 * a.tin);
    hipLaunchKernelGGL(k_mlp_train_pack, dim3(div_up(frags * 512, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch(who);
}

GFPP_API uint32_t gfpp_mlp_train_image_bytes(uint32_t n_layers, uint32_t in_pad, uint32_t out_pad, int backward) {
    const uint32_t tin = in_pad / 32, tout = out_pad / 32;
    const uint32_t frags = backward ? 2 * tout * 4 + (n_layers - 2) * 32 + 8 * tin : 2 * tin * 4 + (n_layers - 2) * 32 + 8 * tout;
    return frags * 1024u;
}

GFPP_API int gfpp_mlp_train_forward(const void *x, const void *fwd_image, uint32_t M, uint32_t in_pad, uint32_t hidden, uint32_t n_layers, uint32_t out_pad,
                                    void *hidden_acts, void *out, gfpp_stream_t stream) {
    const char *who = "gfpp_mlp_train_forward";
    if (!x || !fwd_image || !hidden_acts || !out || M == 0) { set_error("%s: null argument or no rows", who); return GFPP_EINVAL; }
    int rc = fm_shape_ok(who, in_pad, hidden, n_layers, out_pad);
    if (rc) return rc;
    if (((uintptr_t)x | (uintptr_t)fwd_image | (uintptr_t)hidden_acts | (uintptr_t)out) & 15u) { set_error("%s: buffers must be 16-byte aligned", who); return GFPP_EINVAL; }
    FmArgs a;
    a.in = static_cast<const f16x8 *>(x); a.image = static_cast<const f16x8 *>(fwd_image); a.hidden = static_cast<f16x8 *>(hidden_acts); a.acts = nullptr;
    a.out = static_cast<f16x8 *>(out); a.M = M; a.n_blocks = div_up(M, 32);
    if (!fm_pick(false, in_pad / 32, n_layers, out_pad / 32, a, fm_grid(a.n_blocks), (hipStream_t)stream)) { set_error("%s: no kernel for this shape", who); return GFPP_EUNSUPPORTED; }
    return check_launch(who);
}

GFPP_API int gfpp_mlp_train_backward(const void *grad_out, const void *hidden_acts, const void *bwd_image, uint32_t M, uint32_t in_pad, uint32_t hidden, uint32_t n_layers,
                                     uint32_t out_pad, void *grad_hidden, void *grad_x, gfpp_stream_t stream) {
    const char *who = "gfpp_mlp_train_backward";
    if (!grad_out || !hidden_acts || !bwd_image || !grad_hidden || M == 0) { set_error("%s: null argument or no rows", who); return GFPP_EINVAL; }
    int rc = fm_shape_ok(who, in_pad, hidden, n_layers, out_pad);
    if (rc) return rc;
    if (((uintptr_t)grad_out | (uintptr_t)hidden_acts | (uintptr_t)bwd_image | (uintptr_t)grad_hidden | (uintptr_t)grad_x) & 15u) { set_error("%s: buffers must be 16-byte aligned", who); return GFPP_EINVAL; }
    FmArgs a;
    a.in = static_cast<const f16x8 *>(grad_out); a.image = static_cast<const f16x8 *>(bwd_image); a.hidden = static_cast<f16x8 *>(grad_hidden);
    a.acts = static_cast<const f16x8 *>(hidden_acts); a.out = static_cast<f16x8 *>(grad_x); a.M = M; a.n_blocks = div_up(M, 32);
    if (!fm_pick(true, in_pad / 32, n_layers, out_pad / 32, a, fm_grid(a.n_blocks), (hipStream_t)stream)) { set_error("%s: no kernel for this shape", who); return GFPP_EUNSUPPORTED; }
    return check_launch(who);
}
