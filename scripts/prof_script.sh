#!/bin/bash
# kernel-trace summary of any script: prof_script.sh <outdir under gpurun_out> <script.py> [args...]
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/$1"; shift; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_any
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_any -o ks -- python "$R/$@" > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_any -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
grep -v "rocprofv3\|^$" "$OUT/run.log" | tail -3
