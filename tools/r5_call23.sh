#!/bin/bash
# round 5, GPU call 23: what the end-of-launch histogram atomics of the persistent head kernel cost (timing-only variant without them; its frames are wrong)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r5c23.log
: > $L
Q="--steps 100 --warmup 8 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
for rep in 1 2 3; do for lib in prod nohist; do
  if [ $lib = prod ]; then unset GFPP_LIB_PATH; else export GFPP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/lib_$lib.so; fi
  ( timeout 300 python bench.py $Q 2>&1 | python -c "
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d['roofline']
    print('$lib', d.get('value'), d.get('value_unchecked'), d['ms_per_step'], 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'))
except Exception as e:
    print('$lib PARSE FAIL', e, l[-800:])" ) >> $L 2>&1
done; done
unset GFPP_LIB_PATH
echo done >> $L
