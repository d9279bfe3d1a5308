cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "superresolution" 2>&1 | tail -3
for rep in 1 2; do for f in 0 1; do for m in const random; do echo -n "fuse=$f "; GFPP_SR_FUSE_FIRST=$f timeout 120 python tools/sr_bench.py 200 $m 2>&1 | tail -1; done; done; done
bash tools/ab_env.sh r03_srfuse_ab GFPP_SR_FUSE_FIRST 0 1 --variant may_torso_sr --hw 256 --precision fp16
cat gpurun_out/r03_srfuse_ab.log
