#pragma once
// oracle/_ref build glue (see oracle/build_ref.py)
#include <hip/hip_runtime.h>
