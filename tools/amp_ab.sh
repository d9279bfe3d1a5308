#!/bin/bash
# training step, fp32 vs autocast: tests + rocprofv3 kernel stats (GPU box): tools/amp_ab.sh <tag> [amp]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r03_train}; mode=$2
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_ref_kernels_gpu.py -q -x -m gpu 2>&1 | tail -15
for rep in 1 2; do timeout 300 python tools/profile_train.py 65536 6 2>&1 | tail -1; timeout 300 python tools/profile_train.py 65536 6 amp 2>&1 | tail -1; done
rm -rf gpurun_out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o t -- python tools/profile_train.py 65536 6 $mode > gpurun_out/${tag}.log 2>&1
tail -1 gpurun_out/${tag}.log
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_stats/t_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:16]:
    print(f'{int(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["Percentage"]):5.1f}% {r["Calls"]:>5} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
