#!/bin/bash
# same-box A/B of two library builds on the SR stage (GPU box): tools/sr_ab.sh <lib A> <lib B>   (paths relative to genefaceplusplus_amd/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "superresolution" 2>&1 | tail -2
for rep in 1 2; do for lib in "$1" "$2"; do for m in const random; do echo -n "$lib "; GFPP_LIB_PATH=$GRAFT_REPO_ROOT/genefaceplusplus_amd/$lib timeout 120 python tools/sr_bench.py 200 $m 2>&1 | tail -1; done; done; done
bash tools/ab_lib.sh r03_srasm_ab "$1" "$2" --variant may_torso_sr --hw 256 --precision fp16
cat gpurun_out/r03_srasm_ab.log | cut -c 1-120
