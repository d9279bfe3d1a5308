#!/bin/bash
# round 5, GPU call 2: XCD-local ownership A/B (bytes, frames/s, counters) + kernel trace of the clip loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "headline or group_torso or frame_groups_deliver" 2>&1 | tail -8 ) > gpurun_out/r5c2_tests.log 2>&1
Q="--steps 400 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0"
digest='
import sys,json
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d.get("roofline",{})
    print(json.dumps({"value":d["value"],"ms":d["ms_per_step"],"ok":d["config"].get("timed_frames_check",{}).get("ok"),"frac":r.get("frac"),"launch_ms":r.get("avg_launch_ms"),"mfma":r.get("mfma",{}).get("frac"),"kc":r.get("workgroup_kcycles"),"bal":r.get("workgroup_balance")}))
except Exception as e:
    print("PARSE FAIL",e,l[-1500:])
'
for v in "GFPP_PERSIST_XCD=0" "GFPP_PERSIST_XCD=1" "GFPP_PERSIST_XCD=0" "GFPP_PERSIST_XCD=1" "GFPP_PERSIST_XCD=1 GFPP_GROUP_TORSO=0"; do
  echo "== $v" >> gpurun_out/r5c2_ab.log
  ( env $v timeout 300 python bench.py $Q 2>&1 | python -c "$digest" ) >> gpurun_out/r5c2_ab.log 2>&1
done
echo "== fp16 XCD=1" >> gpurun_out/r5c2_ab.log
( timeout 300 python bench.py $Q --precision fp16 2>&1 | python -c "$digest" ) >> gpurun_out/r5c2_ab.log 2>&1
echo "== sr 256 fp16 XCD=2 (forced)" >> gpurun_out/r5c2_ab.log
( GFPP_PERSIST_XCD=2 timeout 300 python bench.py --steps 400 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c2_ab.log 2>&1
echo "== sr 256 fp16 XCD=0" >> gpurun_out/r5c2_ab.log
( GFPP_PERSIST_XCD=0 timeout 300 python bench.py --steps 400 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c2_ab.log 2>&1
# kernel trace of the clip loop as the bench runs it (graph replay, two lanes)
rm -rf gpurun_out/r5c2_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c2_trace -o bench -- python bench.py --steps 200 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 > gpurun_out/r5c2_trace.log 2>&1
python - <<'PY' > gpurun_out/r5c2_trace_digest.txt 2>&1
import csv, glob, collections
fs = glob.glob("gpurun_out/r5c2_trace/**/*kernel_trace.csv", recursive=True)
rows = [r for f in fs for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatches", len(rows))
# the last 12 groups' worth of kernels: names, start offsets, durations
tail = rows[-90:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print("%9.1f %8.1f  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
agg = collections.defaultdict(list)
for r in rows[len(rows) // 2:]:
    agg[r["Kernel_Name"][:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print()
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%10.1f us total %6d calls %8.1f avg %8.1f max  %s" % (sum(v), len(v), sum(v) / len(v), max(v), k))
PY
rm -rf gpurun_out/r5c2_trace/*/*.db
GFPP_PERSIST_XCD=1 bash tools/pmc_workload.sh r5c2_pmc_xcd1 may_torso 512 bf16 4 > gpurun_out/r5c2_pmc_xcd1.log 2>&1
GFPP_PERSIST_XCD=0 bash tools/pmc_workload.sh r5c2_pmc_xcd0 may_torso 512 bf16 4 > gpurun_out/r5c2_pmc_xcd0.log 2>&1
rm -rf gpurun_out/r5c2_pmc_xcd?_p*/ 
echo done
