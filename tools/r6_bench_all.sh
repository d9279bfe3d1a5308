#!/bin/bash
# Round 6: the driver-shaped line, the trained-field lines, the SR line (gpurun: outputs under gpurun_out/)
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
python bench.py --trained plain --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 1000 > gpurun_out/r6_bench_trained.json 2> gpurun_out/r6_bench_trained.err
python bench.py --trained sr --steps 100 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 1000 > gpurun_out/r6_bench_trained_sr.json 2> gpurun_out/r6_bench_trained_sr.err
python bench.py --variant may_torso_sr --hw 256 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-grid-stage --long-run-frames 1000 > gpurun_out/r6_bench_sr.json 2> gpurun_out/r6_bench_sr.err
for f in default trained trained_sr sr; do echo "== $f"; tail -3 gpurun_out/r6_bench_$f.err; python tools/bench_digest.py gpurun_out/r6_bench_$f.json; done
