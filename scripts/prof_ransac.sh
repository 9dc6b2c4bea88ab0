#!/bin/bash
# kernel-trace summary of one RANSAC estimator (scripts/gpu_time_ransac.py <est>) -> stdout
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -o ks -- python "$R/scripts/gpu_time_ransac.py" $1 > /tmp/prof_rs.log 2>&1
f=$(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"]); nm = m.group(1) if m else r["Name"][:40]
    print("%-28s calls %4s total %9.2f ms  max %9.2f ms" % (nm[:28], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
