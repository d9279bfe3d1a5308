#!/bin/bash
# End-of-round records (GPU box) in one gpurun call: the whole -m gpu suite, the driver-shaped bench lines (headline, SR variant, fp32, the trained fields), kernel traces
# of the headline, the SR configuration and the trained 512^2 field, the counter passes of the workloads as the clip loop runs them:  tools/final_round.sh <round tag> [nopmc]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
r=$1
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -X faulthandler -m pytest tests -q -x -m gpu > gpurun_out/${r}_gpu_suite_full.log 2>&1; grep -v '^  File "/usr' gpurun_out/${r}_gpu_suite_full.log | tail -25 > gpurun_out/${r}_gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${r}_bench_final.json 2> gpurun_out/${r}_bench_final.err
timeout 300 python bench.py --variant may_torso_sr --hw 256 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_sr.json 2>/dev/null
timeout 300 python bench.py --precision fp32 --steps 60 --no-modes --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_fp32.json 2>/dev/null
timeout 400 python bench.py --trained plain --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_trained.json 2>/dev/null
timeout 400 python bench.py --trained sr --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage > gpurun_out/${r}_bench_trained_sr.json 2>/dev/null
rm -rf gpurun_out/${r}_bf16_stats gpurun_out/${r}_sr_stats gpurun_out/${r}_trained_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${r}_bf16_stats -o bench -- python bench.py --steps 40 --warmup 5 --precision bf16 --no-cpu-baseline --no-modes --no-configs > gpurun_out/${r}_bf16_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${r}_sr_stats -o bench -- python bench.py --steps 40 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-cpu-baseline --no-modes --no-configs --no-grid-stage > gpurun_out/${r}_sr_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${r}_trained_stats -o bench -- python bench.py --trained plain --steps 40 --warmup 5 --no-cpu-baseline --no-modes --no-configs --no-grid-stage --ckpt-parity 0 > gpurun_out/${r}_trained_bench.log 2>&1
if [ "$2" != "nopmc" ]; then
  bash tools/pmc_workload.sh ${r}_pmc_may_torso_512_bf16 may_torso 512 bf16 4
  bash tools/pmc_workload.sh ${r}_pmc_may_torso_sr_256_bf16 may_torso_sr 256 bf16 4
  bash tools/pmc_workload.sh ${r}_pmc_trained_may_torso_512_bf16 trained_may_torso 512 bf16 4
  bash tools/pmc_workload.sh ${r}_pmc_trained_may_torso_sr_256_bf16 trained_may_torso_sr 256 bf16 4
  rm -rf gpurun_out/${r}_pmc_*_p[0-9] gpurun_out/${r}_pmc_*_ptrace
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${r}_smoke.log 2>&1; tail -1 gpurun_out/${r}_smoke.log
for mode in amp ""; do timeout 300 python tools/profile_train.py 65536 6 $mode 2>&1 | tail -1; done > gpurun_out/${r}_train_steps.log
rm -rf gpurun_out/${r}_train_amp_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${r}_train_amp_stats -o t -- python tools/profile_train.py 65536 6 amp > gpurun_out/${r}_train_amp_prof.log 2>&1
timeout 120 python tools/sr_bench.py 200 random > gpurun_out/${r}_sr_bench_stage.log 2>&1
timeout 120 python tools/clock_probe_sr.py > gpurun_out/${r}_clock_probe_sr.log 2>&1
cat gpurun_out/${r}_gpu_suite.log | tail -4; for f in final sr fp32 trained trained_sr; do echo "== $f"; python tools/bench_digest.py gpurun_out/${r}_bench_$f.json 2>&1 | head -14; done
cat gpurun_out/${r}_train_steps.log; tail -1 gpurun_out/${r}_sr_bench_stage.log; tail -3 gpurun_out/${r}_clock_probe_sr.log
