// A canary for cross-kernel interference (round 6): workgroups of four wavefronts repeat a handful of self-checking computations -- an accumulate chain of MFMAs on
// register operands, the same chain with its A operands read from LDS, a vector-ALU chain, an LDS write / read-back, 16-byte global loads, lane shuffles, packed dot
// products + v_permlane32_swap -- and compare every repetition with the FIRST one (same wavefront, same registers, same inputs: any difference is a transient fault).
// tools/canary.py runs it on one stream while another stream keeps the SR stage in flight.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o build/probe/libcanary.so tools/probe/canary.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kTests = 8;

struct CanaryArgs {
    unsigned long long *res;      // [kTests][4]: mismatching wavefront-repetitions, OR of the lanes' ballots, last repetition seen, register / detail
    const float *tab;             // [n_tab] = tab_value(i)
    uint32_t n_tab, iters;
};

__host__ __device__ inline float tab_value(uint32_t i) { return (float)((i * 2654435761u) >> 20) * (1.0f / 4096.0f); }

__device__ __forceinline__ void report(const CanaryArgs &a, int test, bool bad, uint32_t it, uint32_t detail) {
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) {
        atomicAdd(&a.res[test * 4 + 0], 1ull);
        atomicOr(&a.res[test * 4 + 1], m);
        a.res[test * 4 + 2] = it;
        atomicOr(&a.res[test * 4 + 3], (unsigned long long)detail);
    }
}

__device__ __forceinline__ f16x8 operand(int lane, int s, int mul, int mod, int z) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)(((lane + z) * mul + s * 13 + e * 3) % mod - mod / 2) * 0.125f);
    return v;
}

__global__ __launch_bounds__(256) void k_canary(CanaryArgs a) {
    __shared__ __attribute__((aligned(16))) f16x8 s_w[6 * 64];            // the A operands of test 1, as a weight image would sit in LDS
    __shared__ __attribute__((aligned(16))) float s_rw[4][64 * 4];        // test 3: one region per wavefront
    __shared__ __attribute__((aligned(16))) float s_b[32];                // test 7: a layer's bias vector
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 6 * 64; i += 256) s_w[i] = operand(i & 63, i >> 6, 7, 17, 0);
    if (tid < 32) s_b[tid] = (float)(tid * 3 - 40) * 0.0625f;
    __syncthreads();

    v16f ref0, ref1, ref7;
    float ref2 = 0.0f, ref6 = 0.0f;
    for (uint32_t it = 0; it < a.iters; ++it) {
        int z = 0;
        asm volatile("" : "+v"(z));                                       // (opaque zero: every repetition recomputes everything)
        // ---- 0: six chained MFMAs on one accumulator, operands made in registers ---------------------------------------------------------------------------
        {
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 6; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(operand(lane, s, 7, 17, z), operand(lane, s, 5, 13, z), acc, 0, 0, 0);
            if (it == 0) ref0 = acc;
            bool bad = false;
            uint32_t regs = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (__float_as_uint(acc[r]) != __float_as_uint(ref0[r])) { bad = true; regs |= 1u << r; }
            report(a, 0, bad, it, regs);
        }
        // ---- 1: the same chain, A operands read from LDS two steps ahead (mfma_layer_lds of the torso / head kernels) -----------------------------------------------
        {
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const f16x8 *p = s_w + lane + z;
            f16x8 ring[3];
            ring[0] = p[0]; ring[1] = p[64];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (s + 2 < 6) ring[(s + 2) % 3] = p[(s + 2) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[s % 3], operand(lane, s, 5, 13, z), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == 0) ref1 = acc;
            bool bad = false;
            uint32_t regs = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (__float_as_uint(acc[r]) != __float_as_uint(ref1[r])) { bad = true; regs |= 1u << r; }
            report(a, 1, bad, it, regs);
        }
        // ---- 2: vector ALU chain ------------------------------------------------------------------------------------------------------------------------
        {
            float x = (float)(lane + z) * 0.03125f + 1.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) x = fmaf(x, 0.99f + 0.0001f * (float)k, 0.01f * (float)(k & 3));
            if (it == 0) ref2 = x;
            report(a, 2, __float_as_uint(x) != __float_as_uint(ref2), it, 0);
        }
        // ---- 3: LDS write, read back through another lane mapping -------------------------------------------------------------------------------------------
        {
            float4 v = make_float4((float)(lane + z), (float)(lane * 3 + 1), (float)(it & 255u), (float)(wave + 7));
            *reinterpret_cast<float4 *>(&s_rw[wave][lane * 4]) = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int src = (lane * 5 + 3) & 63;
            const float4 g = *reinterpret_cast<const float4 *>(&s_rw[wave][src * 4]);
            const bool bad = g.x != (float)src || g.y != (float)(src * 3 + 1) || g.z != (float)(it & 255u) || g.w != (float)(wave + 7);
            report(a, 3, bad, it, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 4: 16-byte global loads at 8-byte alignment (the grid tables' rows) ---------------------------------------------------------------------------------
        {
            typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = (2u * (uint32_t)((lane + z) * 37 + (int)(it % 97u) * 101 + k * 7919 + (int)blockIdx.x * 13)) % (a.n_tab - 4u);
                const f32x4_a8 g = *reinterpret_cast<const f32x4_a8 *>(a.tab + i);
                bad = bad || g[0] != tab_value(i) || g[1] != tab_value(i + 1) || g[2] != tab_value(i + 2) || g[3] != tab_value(i + 3);
            }
            // ... and dependent single-byte loads (the pre-march's bitfield probes)
            uint32_t at = (uint32_t)((lane + z) * 53 + (int)(it % 89u) * 211 + (int)blockIdx.x * 17) % (4u * a.n_tab - 4u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint8_t g = reinterpret_cast<const uint8_t *>(a.tab)[at];
                const uint8_t want = (uint8_t)(__float_as_uint(tab_value(at >> 2)) >> (8u * (at & 3u)));
                bad = bad || g != want;
                at = (at * 5u + (uint32_t)g + 1u) % (4u * a.n_tab - 4u);          // the next address depends on the loaded byte
            }
            report(a, 4, bad, it, 0);
        }
        // ---- 5: lane shuffles (ds_bpermute) -----------------------------------------------------------------------------------------------------------------
        {
            const int src = ((lane + z) * 5 + 3) & 63;
            const float g = __shfl((float)(lane * 9 + 1), src);
            report(a, 5, g != (float)(src * 9 + 1), it, 0);
        }
        // ---- 6: packed dot products + the half-wave sum through v_permlane32_swap (skinny_dot) ----------------------------------------------------------------------
        {
            const f16x8 w = operand(lane, 1, 7, 17, z), x = operand(lane, 2, 5, 13, z);
            float s0 = 0.0f;
            s0 = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 0, 1), __builtin_shufflevector(x, x, 0, 1), s0, false);
            s0 = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 2, 3), __builtin_shufflevector(x, x, 2, 3), s0, false);
            s0 = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 4, 5), __builtin_shufflevector(x, x, 4, 5), s0, false);
            s0 = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 6, 7), __builtin_shufflevector(x, x, 6, 7), s0, false);
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s0), false, false);
            const float sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            if (it == 0) ref6 = sum;
            report(a, 6, __float_as_uint(sum) != __float_as_uint(ref6), it, 0);
        }
        // ---- 7: the accumulator STARTS as a bias vector read from LDS (tl_load_bias of the torso kernels: the compiler loads it straight into the accumulator registers,
        //         `ds_read_b128 a[0:3], ...; s_waitcnt lgkmcnt(0); v_mfma a[0:15], ..., a[0:15]`), then the chain of test 1 ------------------------------------------------
        {
            const int hi = lane >> 5;
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = s_b[(r & 3) + 8 * (r >> 2) + 4 * hi + z];
            const f16x8 *p = s_w + lane + z;
            f16x8 ring[3];
            ring[0] = p[0]; ring[1] = p[64];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (s + 2 < 6) ring[(s + 2) % 3] = p[(s + 2) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[s % 3], operand(lane, s, 5, 13, z), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == 0) ref7 = acc;
            bool bad = false;
            uint32_t regs = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (__float_as_uint(acc[r]) != __float_as_uint(ref7[r])) { bad = true; regs |= 1u << r; }
            report(a, 7, bad, it, regs);
        }
    }
}

extern "C" __attribute__((visibility("default"))) int canary_launch(void *stream, unsigned long long *res, const float *tab, uint32_t n_tab, uint32_t iters, uint32_t workgroups) {
    CanaryArgs a{res, tab, n_tab, iters};
    hipLaunchKernelGGL(k_canary, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) float canary_tab_value(uint32_t i) { return tab_value(i); }
