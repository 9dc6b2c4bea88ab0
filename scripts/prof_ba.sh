#!/bin/bash
# GPU box: the C4 BA headline under rocprofv3 kernel trace; per-kernel summary -> gpurun_out/$1/.  Extra environment
# (e.g. THEIA_HIP_SPLIT=3) is taken from the caller.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-ba}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ba
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -o ks -- python "$R/bench.py" --steps 16 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; fi
python "$R/scripts/kernel_stats_summary.py" "$OUT/kernel_stats.csv" 12
