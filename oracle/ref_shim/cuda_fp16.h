#pragma once
// oracle/_ref build glue (see oracle/build_ref.py): <cuda_fp16.h> -> HIP's half types, plus the one CUDA overload HIP
// does not declare: atomicAdd(__half2*, __half2) (gridencoder.cu:329, half-table backward).  Same semantics
// (an atomic packed add of both halves), spelled as a 32-bit compare-and-swap loop.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

__device__ inline __half2 atomicAdd(__half2* address, __half2 val) {
    unsigned int* p = reinterpret_cast<unsigned int*>(address);
    unsigned int old = *p, assumed;
    do {
        assumed = old;
        __half2 cur = *reinterpret_cast<__half2*>(&assumed);
        __half2 sum = __hadd2(cur, val);
        old = atomicCAS(p, assumed, *reinterpret_cast<unsigned int*>(&sum));
    } while (assumed != old);
    return *reinterpret_cast<__half2*>(&old);
}
