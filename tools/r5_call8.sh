#!/bin/bash
# round 5, GPU call 8: fine stagger of a SIMD's second wavefront (units of 1 024 cycles); the 20-frame line with the fingerprint memo
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
Q="--steps 400 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
digest='
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d.get("roofline",{})
    print(json.dumps({"value":d["value"],"ms":d["ms_per_step"],"ok":d["config"].get("timed_frames_check",{}).get("ok"),"frac":r.get("frac"),"launch_ms":r.get("avg_launch_ms"),"kc":r.get("workgroup_kcycles")}))
except Exception as e:
    print("PARSE FAIL",e,l[-1500:])
'
for s in 0 1 2 3 4 6 0 2; do
  echo "== stagger $s" >> gpurun_out/r5c8_ab.log
  ( GFPP_PERSIST_STAGGER=$s timeout 300 python bench.py $Q 2>&1 | python -c "$digest" ) >> gpurun_out/r5c8_ab.log 2>&1
done
for rep in 1 2 3; do
  ( timeout 300 python bench.py --steps 20 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20-frame line', d['value'], d['ms_per_step'], d['config']['timed_frames_check']['ok'])" ) >> gpurun_out/r5c8_ab.log 2>&1
done
timeout 120 python tools/clip_start_profile.py may_torso 512 bf16 20 >> gpurun_out/r5c8_ab.log 2>&1
echo done
