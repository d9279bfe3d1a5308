#!/bin/bash
# round 5, GPU call 15: which conditioning-network training case crashed call 14's pytest -- every case in its own process, no core files, short timeouts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
L=gpurun_out/r5c15.log
: > $L
for k in may_head-fused may_head_sr-fused audio_head-fused may_head-graph may_head_sr-graph audio_head-graph; do
  echo "=== $k" >> $L
  timeout 120 python -X faulthandler -m pytest "tests/test_train_gpu.py::test_conditioning_networks_in_a_training_step[$k]" -q -x -m gpu > gpurun_out/r5c15_one.log 2>&1
  echo "rc $?" >> $L
  grep -v "^  File \"/usr" gpurun_out/r5c15_one.log | head -60 >> $L
done
echo done >> $L
