// tr_read_probe.hip -- which LDS halves does ds_read_b64_tr_b16 hand to which lane?  (measurement tool, not product)
// LDS holds the element index as a half (0 .. 2047 are exact); lane l passes the address of halves [4 l, 4 l + 4); the output says, per lane and element,
// which index arrived -- i.e. (source lane, source element) = (value / 4, value % 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(float *out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (_Float16)(float)i;
    __syncthreads();
    const int lane = threadIdx.x;
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4;
    const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4 *)(lds + 4 * lane));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}

int main() {
    float *out, host[256];
    if (hipMalloc(&out, sizeof(host)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out);
    if (hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int v = (int)host[l * 4 + j];
            printf("  (lane %2d, e %d)", v / 4, v % 4);
            const int want_lane = 16 * (l / 16) + 4 * j + (l % 16) / 4, want_e = (l % 16) % 4;
            if (v / 4 != want_lane || v % 4 != want_e) ok = 0;
        }
        printf("\n");
    }
    printf("prediction out[l][j] = in[16 (l / 16) + 4 j + (l %% 16) / 4][l %% 4]: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
