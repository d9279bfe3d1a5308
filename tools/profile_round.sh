#!/bin/bash
# Round profile: rocprofv3 kernel-trace + stats of the bench command, and PMC passes (each counter group in its own run, PMC only, as
# gpurun requires) of a 3-frame render, for one precision mode.  Run on the GPU box: tools/profile_round.sh <tag> <precision>
# Outputs under gpurun_out/<tag>_*; tools/profile_digest.py turns them into the files committed under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; prec=$2
out=gpurun_out/${tag}
rm -rf ${out}_stats ${out}_pmc*
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o bench -- python bench.py --steps 40 --warmup 4 --precision $prec --no-cpu-baseline --no-modes --no-configs > ${out}_bench.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d ${out}_pmc$i -o p -- python tools/profile_frame.py may_torso 512 3 $prec > ${out}_pmc$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py ${out}_pmc? > ${out}_pmc.txt 2>&1
tail -1 ${out}_bench.log | cut -c 1-300
ls ${out}_stats
