#!/bin/bash
# One gpurun call for the experiment builds of the head pass that were written without GPU time at the end of round 3:
#   (here)   tools/build_variant.sh earlydir frame_head_lp.hip -DGFPP_LP_EARLY_DIR=2
#            tools/build_variant.sh skmfma frame_head_lp.hip -DGFPP_LP_SKINNY_MFMA=1
#            tools/build_variant.sh lean frame_head_lp.hip -DGFPP_MARCH_LEAN=1
#            tools/build_variant.sh all3 frame_head_lp.hip -DGFPP_MARCH_LEAN=1 -DGFPP_LP_EARLY_DIR=2 -DGFPP_LP_SKINNY_MFMA=1
#            tools/build_variant.sh blk frame_head_lp.hip -DGFPP_LP_BLOCK_TABLE=1          (names with "blk" / "all4" run with GFPP_LP_BLOCK_TABLE=1: corner-block tables)
#            tools/build_variant.sh blk2 frame_head_lp.hip -DGFPP_LP_BLOCK_TABLE=2         (x-y-z blocks: one cache line per level; runs with GFPP_LP_BLOCK_TABLE=2)
#            tools/build_variant.sh blk_g8 frame_head_lp.hip -DGFPP_LP_BLOCK_TABLE=1 -DGFPP_LP_LEVEL_GROUP=8     (all eight levels of a lane in flight: 237 VGPRs, no scratch)
#            tools/build_variant.sh all4 frame_head_lp.hip -DGFPP_LP_BLOCK_TABLE=1 -DGFPP_LP_LEVEL_GROUP=8 -DGFPP_MARCH_LEAN=1 -DGFPP_LP_EARLY_DIR=2 -DGFPP_LP_SKINNY_MFMA=1
#   (box)    tools/variants_ab.sh <tag> bench|parity [lib names ...]        default: the eight above
#            bench: ~25 s per library and repetition (2 x 512^2 + 1 x 256^2 SR); parity: ~4 min per library -- run it for the winners only
# Per variant: the parity tests that exercise the 16-bit head kernels (per-sample outputs vs the reference's forward, frames vs the oracle, persistent launch vs
# trip launches), then the same-box A/B against the production library on the headline bench and on the 256^2 SR variant.  Results: gpurun_out/<tag>.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; what=$2; shift 2
libs=${*:-"earlydir skmfma lean all3 blk blk2 blk_g8 all4"}
block_env() { case "$1" in *blk2*) echo 2;; *blk*|*all4*) echo 1;; *) echo 0;; esac; }
out=gpurun_out/$tag.log
if [ "$what" = parity ]; then
  for l in $libs; do
    echo "== parity on lib_$l.so" >> $out
    GFPP_LP_BLOCK_TABLE=$(block_env $l) GFPP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/lib_$l.so timeout 900 python -m pytest tests/test_samples_gpu.py tests/test_render_gpu.py -m gpu -q 2>&1 | tail -12 >> $out
  done
  exit 0
fi
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$1', '$2', d['value'], d['ms_per_step'], 'head pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('workgroup_kcycles'))"
}
libpath() { case "$1" in libgfpp_radnerf.so) echo $GRAFT_REPO_ROOT/genefaceplusplus_amd/$1;; *) echo $GRAFT_REPO_ROOT/build/variants/$1;; esac; }
all="libgfpp_radnerf.so $(for x in $libs; do echo lib_$x.so; done)"
for rep in $(seq 1 ${REPS:-2}); do for l in $all; do
  GFPP_LP_BLOCK_TABLE=$(block_env $l) GFPP_LIB_PATH=$(libpath $l) timeout 300 python bench.py --steps 200 --warmup 8 --no-cpu-baseline --no-modes --no-configs --no-grid-stage 2>/dev/null | line $l 512 >> $out
done; done
for l in $all; do
  GFPP_LP_BLOCK_TABLE=$(block_env $l) GFPP_LIB_PATH=$(libpath $l) timeout 300 python bench.py --variant may_torso_sr --hw 256 --precision fp16 --steps 200 --warmup 8 --no-cpu-baseline --no-modes --no-configs --no-grid-stage 2>/dev/null | line $l sr256 >> $out
done
