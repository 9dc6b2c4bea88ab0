#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and the
# two PMC passes (separate runs, counters only), all written under gpurun_out/refresh/ for copying into profiles/.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/refresh"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python "$R/bench.py" --steps 100 --warmup 10 > "$OUT/ks.log" 2>&1
f=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/bench_kernel_stats.csv"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o pmc -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-ransac > "$OUT/pmc_$c.log" 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python "$R/scripts/pmc_summary.py" $c /tmp/prof_$c "$OUT/pmc_$lc.csv" > /dev/null 2>&1
done
tail -c 1500 "$OUT/bench.json"
ls -la "$OUT"
