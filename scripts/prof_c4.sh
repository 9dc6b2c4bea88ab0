#!/bin/bash
# GPU box: C4-size BA (1k views / 500k tracks) under rocprofv3 kernel trace; summary -> gpurun_out/c4/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-c4}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o ks -- python "$R/scripts/gpu_check_c4.py" > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; fi
tail -n 12 "$OUT/run.log"
head -n 30 "$OUT/kernel_stats.csv"
