#!/bin/bash
# kernel-trace summary of one RANSAC leg at the C5 shape (scripts/gpu_time_ransac.py <leg> <pairs>), to gpurun_out/<leg>_prof/
LEG="${1:-dls}"; NP="${2:-1000}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/${LEG}_prof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_leg
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_leg -o ks -- python "$R/scripts/gpu_time_ransac.py" "$LEG" "$NP" > "$OUT/run.log" 2>&1
f=$(find /tmp/prof_leg -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
tail -3 "$OUT/run.log"
cut -c1-150 "$OUT/kernel_stats.csv" | head -8
