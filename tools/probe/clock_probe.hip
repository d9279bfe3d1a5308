// clock_probe.hip -- one wavefront sleeps a fixed number of SHADER cycles per sample (s_sleep counts core clocks) and reads the constant 100 MHz
// s_memrealtime: the real time per sample, relative to an idle GPU, is the inverse of the shader clock under load (s_memtime itself turned out to
// tick at a constant 2.4 GHz on gfx950, so it cannot see throttling).  Built and driven by tools/clock_probe.py (measurement tool, not product).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void k_clock_probe(unsigned long long *out, uint32_t n_samples, uint32_t spin) {
    if (threadIdx.x != 0) return;
    for (uint32_t i = 0; i < n_samples; ++i) {
        out[2 * i] = __builtin_readcyclecounter();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (uint32_t k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(127);   // 127 x 64 shader cycles each, whatever the other waves do
    }
}

extern "C" int clock_probe_launch(unsigned long long *out, uint32_t n_samples, uint32_t spin, void *stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n_samples, spin);
    return (int)hipGetLastError();
}
