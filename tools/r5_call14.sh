#!/bin/bash
# round 5, GPU call 14: the conditioning networks of a training step as graph launches (torch.cuda.make_graphed_callables): tests, step time A/B, trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r5c14.log
: > $L
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu 2>&1 | tail -12 >> $L
for rep in 1 2; do
  ( GFPP_TRAIN_COND=graph timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -2 | sed 's/^/graph:   /' ) >> $L
  ( timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -2 | sed 's/^/fused:   /' ) >> $L
done
tag=r05_train_amp
rm -rf gpurun_out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/${tag}.log 2>&1
tail -1 gpurun_out/${tag}.log >> $L
python - >> $L <<PY
import csv, glob
rows=list(csv.DictReader(open("gpurun_out/${tag}_stats/t_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("${tag}: total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows), "(14 steps)")
f = glob.glob("gpurun_out/${tag}_stats/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "k_march_rays_train" in n]
step = rows[marks[-2]:marks[-1]]
t0 = int(step[0]["Start_Timestamp"])
print("one back-to-back step (march to march):", len(step), "launches, span", (int(rows[marks[-1]]["Start_Timestamp"]) - t0) / 1e3, "us, kernel sum", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3, "us")
prev_end = t0
for r in step:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if st - prev_end > 40000:
        print(f'{(prev_end - t0) / 1e3:9.1f}   -- idle {(st - prev_end) / 1e3:.0f} us --')
    if en - st >= 20000:
        print(f'{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} us  {r["Kernel_Name"][:120]}')
    prev_end = max(prev_end, en)
PY
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv
echo done >> $L
