#!/bin/bash
# GPU box: rocprofv3 kernel trace of the C4 BA bench alone; per-kernel calls / mean / share -> stdout
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ba
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -o ks -- python "$R/bench.py" --steps 40 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > /tmp/ba_trace.log 2>&1
f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1)
mkdir -p "$R/gpurun_out"; cp "$f" "$R/gpurun_out/ba_only_kernel_stats.csv"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    n = r["Name"].replace("thip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-44s calls %6s avg %9.1f us  %5.1f%%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
