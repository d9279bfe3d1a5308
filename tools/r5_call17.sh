#!/bin/bash
# round 5, GPU call 17: frame groups of the head-only models (RADNeRF / RADNeRFwithSR): bytes equal to single frames; clip throughput with and without groups
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
L=gpurun_out/r5c17.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_clip_gpu.py -q -m gpu -k "frame_groups" > gpurun_out/r5c17_pytest.log 2>&1
echo "pytest rc $?" >> $L
grep -v "^  File \"/usr" gpurun_out/r5c17_pytest.log | tail -30 >> $L
for g in 1 4; do
  ( GFPP_CLIP_GROUP=$g timeout 300 python bench.py --variant may_head --steps 200 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 2>&1 | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('may_head 512 group $g:', d['value'], d['ms_per_step'], d['config'].get('timed_frames_check',{}).get('ok'))
except Exception as e:
    print('PARSE FAIL', e, l[-600:])" ) >> $L 2>&1
done
echo done >> $L
