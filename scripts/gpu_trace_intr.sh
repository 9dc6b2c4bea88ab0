#!/bin/bash
# GPU box: kernel trace of C4 with FOCAL_LENGTH | RADIAL_DISTORTION (the pipelines' default subset)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_intr
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_intr -o ks -- python "$R/scripts/${INTR_SCRIPT:-gpu_time_intr.py}" > /tmp/intr_trace.log 2>&1
tail -4 /tmp/intr_trace.log
f=$(find /tmp/prof_intr -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    n = r["Name"].replace("thip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-44s calls %6s avg %9.1f us  %5.1f%%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
