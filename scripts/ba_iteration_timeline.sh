#!/bin/bash
# GPU box: start/end timestamps of the kernels of ONE LM iteration of the C4 bench (rocprofv3 kernel trace)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python "$R/bench.py" --steps 16 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > /tmp/tl.log 2>&1
f=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("thip::(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
# find the last-but-one k_lin_schur and print until the next one
idx = [i for i, n in enumerate(names) if n.startswith("k_lin_schur")]
a, b = idx[40], idx[41]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for i in range(a - 3, b + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%-40s start %9.1f us  dur %7.1f us  gap %6.1f us  grid %s" % (names[i][:40], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, rows[i].get("Grid_Size", "")))
    prev_end = e
PY
