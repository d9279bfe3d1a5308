#!/bin/bash
# k_group_begin with and without the fixed-step pre-march: kernel trace of the clip loop (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  rm -rf gpurun_out/ab_premarch_$v
  GFPP_MARCH_FIXED_STEP=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_premarch_$v -o t -- python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-modes --no-configs --no-grid-stage --long-run-frames 0 > gpurun_out/ab_premarch_$v.log 2>&1
  echo "== fixed_step=$v"; tail -1 gpurun_out/ab_premarch_$v.log | cut -c1-120
  f=$(find gpurun_out/ab_premarch_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
  rm -rf gpurun_out/ab_premarch_$v
done
