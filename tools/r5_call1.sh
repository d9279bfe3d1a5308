#!/bin/bash
# round 5, GPU call 1: new-path tests + A/B of the group torso launch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_samples_gpu.py -m gpu -q --maxfail=20 -x -s 2>&1 | tail -80 ) > gpurun_out/r5c1_tests_a.log 2>&1
Q="--steps 400 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
for v in "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=3" "GFPP_GROUP_TORSO=0" "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=2" "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=4" "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=3" "GFPP_GROUP_TORSO=0"; do
  echo "== $v" >> gpurun_out/r5c1_ab.log
  ( env $v timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); r=d.get('roofline',{})
    print(json.dumps({'value':d['value'],'unchecked':d.get('value_unchecked'),'ms':d['ms_per_step'],'check':d['config'].get('timed_frames_check'),'frac':r.get('frac'),'launch_ms':r.get('avg_launch_ms'),'mfma':r.get('mfma',{}).get('frac')}))
except Exception as e:
    print('PARSE FAIL',e,l[-2000:])
" ) >> gpurun_out/r5c1_ab.log 2>&1
done
( env timeout 300 python bench.py $Q --precision fp16 2>&1 | tail -1 | cut -c1-600 ) >> gpurun_out/r5c1_ab.log 2>&1
( timeout 1200 python -m pytest tests/test_render_scenes_gpu.py tests/test_render_gpu.py -m gpu -q --maxfail=30 2>&1 | tail -60 ) > gpurun_out/r5c1_tests_b.log 2>&1
echo done
