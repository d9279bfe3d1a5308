#!/bin/bash
# same-box A/B of the SR kernels' shape knobs (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GFPP_SR_TILES_UP=2 GFPP_SR_TILES_FINAL=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "superresolution" 2>&1 | tail -2
for rep in 1 2; do for up in 1 2; do for fin in 1 2; do echo -n "up=$up final=$fin "; GFPP_SR_TILES_UP=$up GFPP_SR_TILES_FINAL=$fin timeout 120 python tools/sr_bench.py 200 random 2>&1 | tail -1; done; done; done
