#!/bin/bash
# same-box A/B of one environment switch (GPU box): tools/ab_env.sh <out tag> <VAR> <value A> <value B> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1.log; var=$2; va=$3; vb=$4; shift 4
for rep in 1 2; do for v in "$va" "$vb"; do
env $var=$v timeout 300 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-modes --no-configs --no-grid-stage "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$var=$v', '$*', d['value'], d['ms_per_step'], 'head pass', r['avg_launch_ms'], 'frac', r['frac'])" >> $out
done; done
