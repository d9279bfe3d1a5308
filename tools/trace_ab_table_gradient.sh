#!/bin/bash
# table-gradient kernels of a training step with and without the per-range point lists (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  rm -rf gpurun_out/ab_tablegrad_$v
  GFPP_GRID_BWD_BINS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_tablegrad_$v -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/ab_tablegrad_$v.log 2>&1
  echo "== bins=$v"; tail -1 gpurun_out/ab_tablegrad_$v.log | cut -c1-160
  f=$(find gpurun_out/ab_tablegrad_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time total ms", tot/1e6, "launches", sum(int(r['Calls']) for r in rows))
for r in rows:
    if 'grid' in r['Name'] or rows.index(r) < 6:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
  rm -rf gpurun_out/ab_tablegrad_$v
done
