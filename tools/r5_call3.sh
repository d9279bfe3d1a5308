#!/bin/bash
# round 5, GPU call 3: activation packing under the MFMAs (fused, production build) vs the round-4 layer sequence (lib_nofuse) vs fused + s_setprio (lib_prio)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_samples_gpu.py tests/test_clip_gpu.py tests/test_render_gpu.py -m gpu -q -x -k "samples or headline or persistent_launch_equals or full_size or 16bit_mfma or forward or oracle" 2>&1 | tail -6 ) > gpurun_out/r5c3_tests.log 2>&1
Q="--steps 400 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
digest='
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d.get("roofline",{})
    print(json.dumps({"value":d["value"],"ms":d["ms_per_step"],"ok":d["config"].get("timed_frames_check",{}).get("ok"),"frac":r.get("frac"),"launch_ms":r.get("avg_launch_ms"),"mfma":r.get("mfma",{}).get("frac"),"kc":r.get("workgroup_kcycles")}))
except Exception as e:
    print("PARSE FAIL",e,l[-1500:])
'
V=$GRAFT_REPO_ROOT/build/variants
for rep in 1 2 3; do
for lib in "" "$V/lib_nofuse.so" "$V/lib_prio.so"; do
  echo "== lib=${lib:-production(fused)} bf16" >> gpurun_out/r5c3_ab.log
  ( GFPP_LIB_PATH=$lib timeout 300 python bench.py $Q 2>&1 | python -c "$digest" ) >> gpurun_out/r5c3_ab.log 2>&1
done; done
for lib in "" "$V/lib_nofuse.so"; do
  echo "== lib=${lib:-production(fused)} fp16" >> gpurun_out/r5c3_ab.log
  ( GFPP_LIB_PATH=$lib timeout 300 python bench.py $Q --precision fp16 2>&1 | python -c "$digest" ) >> gpurun_out/r5c3_ab.log 2>&1
  echo "== lib=${lib:-production(fused)} sr256 fp16" >> gpurun_out/r5c3_ab.log
  ( GFPP_LIB_PATH=$lib timeout 300 python bench.py --steps 400 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c3_ab.log 2>&1
done
echo done
