cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -f gpurun_out/c8_*
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_ref_caller_gpu.py tests/test_clip_gpu.py -q -x -m gpu 2>&1 | tail -4 > gpurun_out/c8_tests.log
bash tools/ab_lib.sh c8_ab ../build/variants/lib_prev.so libgfpp_radnerf.so
bash tools/ab_lib.sh c8_ab_sr ../build/variants/lib_prev.so libgfpp_radnerf.so --variant may_torso_sr --hw 256
cat gpurun_out/c8_tests.log gpurun_out/c8_ab.log gpurun_out/c8_ab_sr.log
