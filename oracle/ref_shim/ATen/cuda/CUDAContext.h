#pragma once
// oracle/_ref build glue: PyTorch-ROCm ships this header under ATen/hip/.
#include <ATen/hip/HIPContext.h>
