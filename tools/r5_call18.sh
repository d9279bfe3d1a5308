#!/bin/bash
# round 5, GPU call 18: the 16-bit modes on the fp32 tables (GFPP_LP_BLOCK_TABLE=0) against the oracle and the default path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
L=gpurun_out/r5c18.log
: > $L
timeout 600 python -X faulthandler -m pytest tests/test_render_gpu.py -q -m gpu -k "16bit" -s > gpurun_out/r5c18_pytest.log 2>&1
echo "pytest rc $?" >> $L
grep -v "^  File \"/usr" gpurun_out/r5c18_pytest.log | tail -30 >> $L
for sw in 1 0; do
  ( GFPP_LP_BLOCK_TABLE=$sw timeout 300 python bench.py --steps 100 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 2>&1 | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('block tables $sw:', d['value'], d['ms_per_step'], d['config'].get('timed_frames_check',{}))
except Exception as e:
    print('PARSE FAIL', e, l[-600:])" ) >> $L 2>&1
done
echo done >> $L
