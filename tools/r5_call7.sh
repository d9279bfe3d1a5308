#!/bin/bash
# round 5, GPU call 7: new clip tests; torso MLP launch at 4 wavefronts per SIMD (lib_tmlp4) vs 3; three lanes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "empty_torso or xcd_local or headline or group_torso" 2>&1 | tail -5 ) > gpurun_out/r5c7_tests.log 2>&1
Q="--steps 400 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
digest='
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d.get("roofline",{})
    print(json.dumps({"value":d["value"],"ms":d["ms_per_step"],"ok":d["config"].get("timed_frames_check",{}).get("ok"),"psnr":d["config"].get("timed_frames_check",{}).get("psnr_vs_fp32_mode_db"),"frac":r.get("frac"),"launch_ms":r.get("avg_launch_ms")}))
except Exception as e:
    print("PARSE FAIL",e,l[-1500:])
'
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2; do
for v in "GFPP_TORSO_GROUP_WGS=3" "GFPP_TORSO_GROUP_WGS=4 GFPP_LIB_PATH=$V/lib_tmlp4.so" "GFPP_TORSO_GROUP_WGS=3 GFPP_LIB_PATH=$V/lib_tmlp4.so"; do
  echo "== $v" >> gpurun_out/r5c7_ab.log
  ( env $v timeout 300 python bench.py $Q 2>&1 | python -c "$digest" ) >> gpurun_out/r5c7_ab.log 2>&1
done; done
echo "== lanes 3" >> gpurun_out/r5c7_ab.log
( timeout 300 python bench.py $Q --lanes 3 2>&1 | python -c "$digest" ) >> gpurun_out/r5c7_ab.log 2>&1
echo "== fp32 60 steps" >> gpurun_out/r5c7_ab.log
( timeout 300 python bench.py --precision fp32 --steps 60 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c7_ab.log 2>&1
echo "== sr256 fp16 tmlp4 wgs4 / production" >> gpurun_out/r5c7_ab.log
for v in "GFPP_TORSO_GROUP_WGS=4 GFPP_LIB_PATH=$V/lib_tmlp4.so" "GFPP_TORSO_GROUP_WGS=3"; do
  ( env $v timeout 300 python bench.py --steps 400 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c7_ab.log 2>&1
done
echo done
