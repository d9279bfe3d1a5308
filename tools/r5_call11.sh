#!/bin/bash
# round 5, GPU call 11: ds_read_b64_tr_b16 lane mapping (probe), the transposing weight-gradient kernel: tests, step time A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r5c11.log
: > $L
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -k "mlp_weight_gradients" 2>&1 | tail -25 >> $L
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu 2>&1 | tail -8 >> $L
for rep in 1 2; do
  ( GFPP_TRAIN_FUSED_MLP=0 timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/layers: /' ) >> $L
  ( timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/fused:  /' ) >> $L
done
tag=r05_train_amp
rm -rf gpurun_out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/${tag}.log 2>&1
tail -1 gpurun_out/${tag}.log >> $L
python - >> $L <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_stats/t_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("${tag}: total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows), "(14 steps)")
for r in rows[:60]:
    print(f'{int(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["Percentage"]):5.1f}% {r["Calls"]:>5} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:150]}')
PY
python - >> $L <<PY
# the kernels of ONE back-to-back step (the last), >= 6 us, in launch order: which copies / fills are left
import csv, glob
f = glob.glob("gpurun_out/${tag}_stats/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "k_march_rays_train" in n)
step = rows[last:]
t0 = int(step[0]["Start_Timestamp"])
print("one step:", len(step), "launches, span", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "us, kernel sum", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3, "us")
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 6:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {d:8.1f} us  grid {r.get("Grid_Size", "?"):>10}  {r["Kernel_Name"][:130]}')
PY
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv
echo done >> $L
