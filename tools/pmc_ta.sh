#!/bin/bash
# PMC passes for the texture-address / L1 / LDS path of the trip kernel (each group its own run, PMC only).  tools/pmc_ta.sh <tag> <precision>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; prec=$2
out=gpurun_out/${tag}
rm -rf ${out}_ta*
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d ${out}_ta$i -o p -- python tools/profile_frame.py may_torso 512 3 $prec > ${out}_ta$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py ${out}_ta? > ${out}_ta.txt 2>&1
cat ${out}_ta.txt | cut -c1-400
