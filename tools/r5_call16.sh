#!/bin/bash
# round 5, GPU call 16: the training tests with the fused conditioning networks (graph fallback removed), fp32 + amp step times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
L=gpurun_out/r5c16.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_train_gpu.py -q -m gpu > gpurun_out/r5c16_pytest.log 2>&1
echo "pytest rc $?" >> $L
grep -v "^  File \"/usr" gpurun_out/r5c16_pytest.log | tail -40 >> $L
( GFPP_TRAIN_COND=eager timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/eager cond: /' ) >> $L
( timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/fused cond: /' ) >> $L
( timeout 300 python tools/profile_train.py 65536 6 2>&1 | tail -1 | sed 's/^/fp32:       /' ) >> $L
echo done >> $L
