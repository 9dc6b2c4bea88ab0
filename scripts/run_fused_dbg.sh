R="${GRAFT_REPO_ROOT:-/root/repo}"
for d in 1 3; do
  export THEIA_HIP_FUSED_DBG=$d
  echo "== dbg $d"
  bash "$R/scripts/pmc_kernel.sh" "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 | grep -E "k_lin_schur"
  bash "$R/scripts/pmc_kernel.sh" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 | grep -E "k_lin_schur"
  python "$R/bench.py" --steps 16 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['phase_ms_per_iteration'], b['roofline']['avg_launch_ms'])"
done
