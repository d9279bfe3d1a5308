#!/bin/bash
# round 5, GPU call 22: the table-gradient range kernel with the level's index arithmetic resolved once per workgroup: tests, kernel time, step time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ulimit -c 0
L=gpurun_out/r5c22.log
: > $L
timeout 900 python -X faulthandler -m pytest tests/test_train_gpu.py tests/test_ref_kernels_gpu.py -q -m gpu > gpurun_out/r5c22_pytest.log 2>&1
echo "pytest rc $?" >> $L
grep -v "^  File \"/usr" gpurun_out/r5c22_pytest.log | tail -12 >> $L
for rep in 1 2; do ( timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1 ) >> $L; done
( timeout 300 python tools/profile_train.py 65536 6 2>&1 | tail -1 ) >> $L
tag=r05_train_amp
rm -rf gpurun_out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/${tag}.log 2>&1
tail -1 gpurun_out/${tag}.log >> $L
python - >> $L <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_stats/t_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("${tag}: total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows), "(14 steps)")
for r in rows[:14]:
    print(f'{int(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["Percentage"]):5.1f}% {r["Calls"]:>5} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv
echo done >> $L
