#!/bin/bash
# per-kernel times of the SR stage alone (tools/sr_bench.py), polyphase and composed up-sampling layer (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${1:-1 0}; do
  rm -rf gpurun_out/sr_trace_$v
  GFPP_SR_UP_POLY=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sr_trace_$v -o t -- python tools/sr_bench.py 200 random > gpurun_out/sr_trace_$v.log 2>&1
  echo "== sr_up_poly=$v"; tail -1 gpurun_out/sr_trace_$v.log | cut -c1-160
  f=$(find gpurun_out/sr_trace_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f}")
PY
  rm -rf gpurun_out/sr_trace_$v
done
