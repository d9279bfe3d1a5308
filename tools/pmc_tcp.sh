#!/bin/bash
# Vector-L1 (TCP) counters of the head pass, at most three per pass (the five-counter groups of pmc_ta.sh timed out on this pool): how busy the L1 is with the
# table gathers -- cache accesses (one per distinct line of a gather), requests passed on to the L2, cycles the tag pipeline was gated on / stalled.
#   tools/pmc_tcp.sh <tag> <precision> [lib]     (lib: an experiment build, e.g. lib_blk.so -- names with "blk" run with GFPP_LP_BLOCK_TABLE set)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; prec=$2; lib=${3:-libgfpp_radnerf.so}
case "$lib" in *blk2*) export GFPP_LP_BLOCK_TABLE=2;; *blk*|*all4*) export GFPP_LP_BLOCK_TABLE=1;; esac
export GFPP_LIB_PATH=$GRAFT_REPO_ROOT/genefaceplusplus_amd/$lib
out=gpurun_out/${tag}
rm -rf ${out}_tcp*
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES"; do
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d ${out}_tcp$i -o p -- python tools/profile_frame.py may_torso 512 3 $prec > ${out}_tcp$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py ${out}_tcp? > ${out}_tcp.txt 2>&1
cat ${out}_tcp.txt | cut -c1-400
