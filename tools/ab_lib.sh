#!/bin/bash
# same-box A/B of two builds of the library (GPU box): tools/ab_lib.sh <out tag> <lib A> <lib B> [bench args...]   (paths relative to genefaceplusplus_amd/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1.log; la=$2; lb=$3; shift 3
for rep in 1 2 3; do for lib in "$la" "$lb"; do
GFPP_LIB_PATH=$GRAFT_REPO_ROOT/genefaceplusplus_amd/$lib timeout 300 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-modes --no-configs --no-grid-stage "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$lib', '$*', d['value'], d['ms_per_step'], 'head pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('workgroup_kcycles'))" >> $out
done; done
