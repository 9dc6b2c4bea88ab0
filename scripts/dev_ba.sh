#!/bin/bash
# GPU box (development): BA parity tests on the fused path, then the linearise + Schur launch group at C4 for the
# kernel variants named in $VARIANTS (environment switches), and the plan shape.  Output -> gpurun_out/dev_*.log
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
TESTS="${TESTS:-tests/test_ba_gpu.py tests/test_parity_gpu.py tests/test_fountain_gpu.py tests/test_create_fuzz_gpu.py}"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest $TESTS -x -q -m gpu > gpurun_out/dev_tests.log 2>&1
  tail -15 gpurun_out/dev_tests.log
fi
for u in rcp_acc lds_atomic; do [ -x scripts/ubench/$u ] && timeout 120 scripts/ubench/$u > gpurun_out/dev_$u.log 2>&1; done
THEIA_HIP_CREATE_TIMING=1 timeout 600 python scripts/gpu_time_lin_kernel.py > gpurun_out/dev_time_default.log 2>&1
tail -3 gpurun_out/dev_time_default.log
for v in ${VARIANTS:-THEIA_HIP_FUSED_DBG=1 THEIA_HIP_FUSED_DBG=2 THEIA_HIP_FUSED_DBG=3}; do
  env $v timeout 600 python scripts/gpu_time_lin_kernel.py > "gpurun_out/dev_time_$v.log" 2>&1
  echo "$v: $(tail -1 "gpurun_out/dev_time_$v.log")"
done
