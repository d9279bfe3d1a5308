#!/bin/bash
# Counter passes of one frame-render process under rocprofv3 (PMC only, each group in its own run, as gpurun requires).
# usage: tools/pmc_passes.sh <tag> <precision> "<group1 counters>" "<group2 counters>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; prec=$2; shift 2
i=0
for grp in "$@"; do
  out=gpurun_out/pmc_${tag}_$i
  rm -rf "$out"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out" -o p -- python tools/profile_frame.py may_torso 512 3 "$prec" > "$out.log" 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}.txt 2>&1
cat gpurun_out/pmc_${tag}.txt
