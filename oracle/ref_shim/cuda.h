#pragma once
// oracle/_ref build glue: the reference includes <cuda.h>; on ROCm the same declarations come from HIP.
#include <hip/hip_runtime.h>
