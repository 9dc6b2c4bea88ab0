#!/bin/bash
# GPU box (development): inner-iteration parity tests, LM iteration time at C4 with inner iterations (and with the
# pipelines' default intrinsics mask), kernel trace of that run.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests/test_inner_gpu.py -x -q -m gpu > gpurun_out/dev_inner_tests.log 2>&1
  tail -8 gpurun_out/dev_inner_tests.log
fi
timeout 600 python scripts/gpu_time_inner_c4.py 2>&1 | tail -1
timeout 600 python scripts/gpu_time_inner_c4.py 0x11 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_in
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_in -o ks -- python "$R/scripts/gpu_time_inner_c4.py" ${INNER_MASK:-0x11} > /tmp/prof_in.log 2>&1
f=$(find /tmp/prof_in -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$R/gpurun_out/dev_inner_kernel_stats.csv" && python "$R/scripts/kernel_stats_summary.py" "$f" 22
