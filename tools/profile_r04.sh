#!/bin/bash
# Round-4 profiles in one call (GPU box): rocprofv3 --kernel-trace --stats of the bench command for the headline and the SR configuration, then the counter passes of
# both workloads as the clip loop runs them (4 frames per head launch).  Outputs under gpurun_out/r04_*; tools/profile_digest.py + pmc_workload.py write the summaries.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r04_bf16_stats gpurun_out/r04_sr_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_bf16_stats -o bench -- python bench.py --steps 40 --warmup 5 --precision bf16 --no-cpu-baseline --no-modes --no-configs > gpurun_out/r04_bf16_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_sr_stats -o bench -- python bench.py --steps 40 --warmup 5 --variant may_torso_sr --hw 256 --precision fp16 --no-cpu-baseline --no-modes --no-configs --no-grid-stage > gpurun_out/r04_sr_bench.log 2>&1
bash tools/pmc_workload.sh r04_pmc_may_torso_512_bf16 may_torso 512 bf16 4
bash tools/pmc_workload.sh r04_pmc_may_torso_sr_256_bf16 may_torso_sr 256 bf16 4
bash tools/pmc_workload.sh r04_pmc_may_torso_sr_256_fp16 may_torso_sr 256 fp16 4
tail -1 gpurun_out/r04_bf16_bench.log | cut -c1-200
tail -1 gpurun_out/r04_sr_bench.log | cut -c1-200
