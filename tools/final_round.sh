#!/bin/bash
# End-of-round records (GPU box): the whole -m gpu suite, the driver-shaped bench lines, kernel traces + counter passes: tools/final_round.sh <round tag, e.g. r03>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
r=$1
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > gpurun_out/${r}_gpu_suite.log
timeout 600 python bench.py > gpurun_out/${r}_bench_final.json 2> gpurun_out/${r}_bench_final.err
timeout 300 python bench.py --variant may_torso_sr --hw 256 --precision fp16 --no-modes --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_sr.json 2>/dev/null
timeout 300 python bench.py --precision fp32 --steps 60 --no-modes --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_fp32.json 2>/dev/null
bash tools/profile_round.sh ${r}_bf16 bf16 > gpurun_out/${r}_profile_round.log 2>&1
bash tools/profile_sr.sh ${r}_sr > gpurun_out/${r}_profile_sr.log 2>&1
for mode in amp fp32; do
  rm -rf gpurun_out/${r}_train_${mode}_stats
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${r}_train_${mode}_stats -o t -- python tools/profile_train.py 65536 6 $([ $mode = amp ] && echo amp) > gpurun_out/${r}_train_${mode}.log 2>&1
  timeout 120 python tools/profile_train.py 65536 6 $([ $mode = amp ] && echo amp) 2>/dev/null | tail -1
done
GFPP_SR_TILES=2 GFPP_SR_KSLICES=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k superresolution 2>&1 | tail -1
cat gpurun_out/${r}_gpu_suite.log; for f in final sr fp32; do python -c "
import json; d=json.load(open('gpurun_out/${r}_bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done
