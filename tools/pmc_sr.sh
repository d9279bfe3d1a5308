#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/r04_srpmc$i
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/r04_srpmc$i -o p -- python tools/profile_frame.py may_torso_sr 256 3 fp16 > gpurun_out/r04_srpmc$i.log 2>&1
  i=$((i+1))
done
python - <<'EOF'
import csv,glob,collections
for d in ("gpurun_out/r04_srpmc0","gpurun_out/r04_srpmc1"):
    f=sorted(glob.glob(d+"/**/*counter_collection.csv",recursive=True))[-1]
    rows=[r for r in csv.DictReader(open(f)) if "k_sr_" in r["Kernel_Name"]]
    last=collections.OrderedDict()
    for r in rows: last[(r["Kernel_Name"][:40],r["Counter_Name"])]=float(r["Counter_Value"])
    for k,v in last.items(): print(k,v)
EOF
