#!/bin/bash
# kernel-trace summary of the DLS RANSAC path (gpu_check_dls.py), written to gpurun_out/dls_prof/
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/dls_prof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dls
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dls -o ks -- python "$R/scripts/gpu_check_dls.py" > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_dls -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
tail -5 "$OUT/run.log"
cut -c1-160 "$OUT/kernel_stats.csv" | head -8
