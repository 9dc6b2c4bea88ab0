#!/bin/bash
# GPU box: SQ counter passes (counters only, one rocprofv3 run per group) over the C4 BA bench without the CPU / RANSAC
# legs; per-kernel means -> gpurun_out/sq/*.txt.  Issue vs stall picture of k_lin_schur / k_backsub / K3.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/sq"; mkdir -p "$OUT"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  bash "$R/scripts/pmc_kernel.sh" "$grp" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/pass$i.txt" 2>&1
  tail -n 12 "$OUT/pass$i.txt"
done
