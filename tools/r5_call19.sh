#!/bin/bash
# round 5, GPU call 19: the head kernel under a VGPR cap (amdgpu_num_vgpr: 232 / 240 of 256) so that the next group's 48-register prologue can run beside it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r5c19.log
: > $L
Q="--steps 400 --warmup 8 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
for rep in 1 2; do for lib in prod v116 v120; do
  if [ $lib = prod ]; then unset GFPP_LIB_PATH; else export GFPP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/lib_$lib.so; fi
  ( timeout 300 python bench.py $Q 2>&1 | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d['roofline']
    print('$lib', d['value'], d['ms_per_step'], 'ok', d['config'].get('timed_frames_check',{}).get('ok'), 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'), r.get('workgroup_kcycles'))
except Exception as e:
    print('$lib PARSE FAIL', e, l[-800:])" ) >> $L 2>&1
done; done
unset GFPP_LIB_PATH
echo done >> $L
