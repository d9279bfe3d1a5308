#!/bin/bash
# round 5, GPU call 5 (compose launch without per-workgroup chains): torso group v2 (masked-pixel list + MLP launch + compose launch) and the lean pre-march probe: parity, A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_clip_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r5c5_tests.log 2>&1
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
for v in "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=3" "GFPP_GROUP_TORSO=0" "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=2" "GFPP_GROUP_TORSO=1 GFPP_TORSO_GROUP_WGS=3" "GFPP_GROUP_TORSO=0"; do
  echo "== $v" >> gpurun_out/r5c5_ab.log
  ( env $v timeout 300 python bench.py $Q 2>&1 | python -c "$digest" ) >> gpurun_out/r5c5_ab.log 2>&1
done
for v in "GFPP_GROUP_TORSO=1" "GFPP_GROUP_TORSO=0"; do
  echo "== sr256 fp16 $v" >> gpurun_out/r5c5_ab.log
  ( env $v timeout 300 python bench.py --steps 400 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 0 2>&1 | python -c "$digest" ) >> gpurun_out/r5c5_ab.log 2>&1
done
rm -rf gpurun_out/r5c5_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c5_trace -o bench -- python bench.py --steps 200 --warmup 5 --no-modes --no-configs --no-grid-stage --no-cpu-baseline --long-run-frames 0 > gpurun_out/r5c5_trace.log 2>&1
python - <<'PY' > gpurun_out/r5c5_trace_digest.txt 2>&1
import csv, glob, collections
import numpy as np
fs = glob.glob("gpurun_out/r5c5_trace/**/*kernel_trace.csv", recursive=True)
rows = [r for f in fs for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = np.array(v); print("%6d calls min %8.1f p10 %8.1f median %8.1f mean %8.1f  %s" % (len(v), v.min(), np.percentile(v, 10), np.median(v), v.mean(), n))
heads = [r for r in rows if "k_head_frame_persist" in r["Kernel_Name"] and "Lb0ELb1ELb0" in r["Kernel_Name"]]
starts = [int(r["Start_Timestamp"]) for r in heads]
best = (0, 0); i = 0
while i < len(starts):
    j = i
    while j + 1 < len(starts) and starts[j + 1] - starts[j] < 1_200_000: j += 1
    if j - i > best[1] - best[0]: best = (i, j)
    i = j + 1
print("longest run of head launches", best, "period us", (starts[best[1]] - starts[best[0]]) / 1e3 / max(best[1] - best[0], 1))
t0 = starts[min(best[0] + 20, best[1] - 3)]
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 <= s < t0 + 2_700_000:
        print("%9.1f -> %9.1f (%7.1f) q%s %s" % ((s - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][10:44]))
PY
rm -rf gpurun_out/r5c5_trace/*/*.db
echo done
