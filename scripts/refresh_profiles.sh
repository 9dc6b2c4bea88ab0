#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and the
# two PMC passes (separate runs, counters only: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), all written under
# gpurun_out/refresh/ for copying into profiles/<tag>_*.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/refresh"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 python "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/prof_ks
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python "$R/bench.py" --no-cpu-baseline > "$OUT/ks.log" 2>&1
f=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/bench_kernel_stats.csv"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o pmc -- python "$R/bench.py" --steps 16 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/pmc_$c.log" 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python "$R/scripts/pmc_summary.py" $c /tmp/prof_$c "$OUT/pmc_$lc.csv" > /dev/null 2>&1
done
tail -c 300 "$OUT/bench.err"
ls -la "$OUT"
head -5 "$OUT/pmc_fetch_size.csv" "$OUT/pmc_write_size.csv"
