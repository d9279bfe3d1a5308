// mfma_probe.hip -- issue-rate microbenchmark of v_mfma_f32_32x32x16_bf16 (measurement tool, not product): N MFMAs per wavefront with NACC independent
// accumulators, operands in registers (MODE 0) or the A operand re-read from LDS before every MFMA group (MODE 1), 1 or 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void k_probe(float *out, unsigned long long *cyc, int iters) {
    __shared__ bf16x8 lds[4 * 64 * 8];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * 64 * 8; i += blockDim.x) lds[i] = bf16x8{(__bf16)(0.001f * (i & 7)), (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
    __syncthreads();
    v16f acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    bf16x8 A = lds[lane], B = lds[64 + lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const bf16x8 Aa = lds[((it * NACC + a) & 31) * 64 + lane];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa, B, acc[a], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
static void run(const char *label, int threads, int blocks, int iters) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_wave = (double)iters * NACC, waves_per_simd = threads / 256.0;
    printf("%-44s %4d thr x %4d WG: %.1f memtime-cycles per MFMA per wave, wall %.3f ms -> %.1f cycles(2.4GHz) per MFMA per SIMD, %.0f TFLOP/s\n", label, threads, blocks,
           (double)h / mfma_per_wave, ms, ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd), blocks * (threads / 64.0) * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 4000;
    run<8, 0>("8 accumulators, register operands", 256, 256, it);
    run<4, 0>("4 accumulators, register operands", 256, 256, it);
    run<2, 0>("2 accumulators, register operands", 256, 256, it);
    run<1, 0>("1 accumulator (dependent chain)", 256, 256, it);
    run<8, 0>("8 accumulators, 2 waves/SIMD", 512, 256, it);
    run<4, 0>("4 accumulators, 2 waves/SIMD", 512, 256, it);
    run<8, 1>("8 accumulators, A from LDS each MFMA", 256, 256, it);
    run<4, 1>("4 accumulators, A from LDS, 2 waves/SIMD", 512, 256, it);
    run<8, 0>("8 accumulators, ONE workgroup", 256, 1, it);
    return 0;
}
