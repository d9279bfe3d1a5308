#!/bin/bash
# round 5, GPU call 10: ds_read_b64_tr_b16 lane mapping (probe), the transposing weight-gradient kernel: tests, step time A/B, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r5c10.log
: > $L
timeout 60 ./build/probe/tr_read_probe >> $L 2>&1
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -k "mlp_weight_gradients" 2>&1 | tail -25 >> $L
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu 2>&1 | tail -8 >> $L
for rep in 1 2; do
  ( GFPP_WGRAD_TR=0 timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/generic wgrad: /' ) >> $L
  ( timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 | sed 's/^/tr wgrad:      /' ) >> $L
done
tag=r05_train_amp
rm -rf gpurun_out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/${tag}.log 2>&1
tail -1 gpurun_out/${tag}.log >> $L
python - >> $L <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_stats/t_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("${tag}: total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows), "(8 steps)")
for r in rows[:60]:
    print(f'{int(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["Percentage"]):5.1f}% {r["Calls"]:>5} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:150]}')
PY
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv
echo done >> $L
