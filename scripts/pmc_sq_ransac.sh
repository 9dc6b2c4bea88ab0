#!/bin/bash
# GPU box: SQ counter passes (counters only, one rocprofv3 run per group) over one RANSAC leg at the C5 shape
# usage: pmc_sq_ransac.sh <leg> <pairs>   -> gpurun_out/sq_<leg>/pass*.txt
set -u
LEG="${1:-dls}"; NP="${2:-250}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/sq_$LEG"; mkdir -p "$OUT"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64"; do
  i=$((i+1))
  bash "$R/scripts/pmc_kernel.sh" "$grp" scripts/gpu_time_ransac.py "$LEG" "$NP" > "$OUT/pass$i.txt" 2>&1
  grep -v "^$" "$OUT/pass$i.txt" | tail -n 6
done
