#!/bin/bash
# rocprofv3 kernel-trace + stats of the super-resolution variant (256^2 rays + StyleGAN2 SR -> 512^2), run on the GPU box: tools/profile_sr.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
rm -rf gpurun_out/${tag}_stats
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o bench -- python bench.py --variant may_torso_sr --hw 256 --steps 40 --warmup 4 --precision fp16 --no-cpu-baseline --no-modes --no-grid-stage --no-configs > gpurun_out/${tag}_bench.log 2>&1
tail -1 gpurun_out/${tag}_bench.log | cut -c 1-300
