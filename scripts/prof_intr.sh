#!/bin/bash
# GPU box: C4 with FOCAL | RADIAL free under rocprofv3 kernel trace; per-kernel summary -> gpurun_out/$1/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-intr}"; GROUPS_ARG="${2:-8}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_intr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_intr -o ks -- python "$R/scripts/gpu_time_intr_c4.py" "$GROUPS_ARG" > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_intr -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; fi
grep "^C4" "$OUT/run.log"
python "$R/scripts/kernel_stats_summary.py" "$OUT/kernel_stats.csv" 16
