#!/bin/bash
# Counter passes of ONE workload (variant, frame side, precision, frames per head launch): each counter group in its own rocprofv3 run (PMC + kernel trace only, as
# gpurun requires), target = tools/profile_clip.py; then tools/pmc_workload.py digests the last head launch of every pass into profiles/<tag>.json / .md.
#   tools/pmc_workload.sh r04_pmc_may_torso_512_bf16 may_torso 512 bf16 1
#   tools/pmc_workload.sh r04_pmc_may_torso_sr_256_fp16 may_torso_sr 256 fp16 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; variant=$2; hw=$3; prec=$4; group=${5:-1}
out=gpurun_out/$tag
rm -rf ${out}_p*
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d ${out}_p$i -o p -- python tools/profile_clip.py $variant $hw $prec $group > ${out}_p$i.log 2>&1
  i=$((i+1))
done
# the same target without counters: the launch duration the request RATES are computed with
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_ptrace -o p -- python tools/profile_clip.py $variant $hw $prec $group > ${out}_ptrace.log 2>&1
python tools/pmc_workload.py $tag $variant $hw $prec $group
