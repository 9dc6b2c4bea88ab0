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
# C4-only kernel trace (the BA headline alone: no C2, no RANSAC)
rm -rf /tmp/prof_ba
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -o ks -- python "$R/bench.py" --steps 40 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/ks_ba.log" 2>&1
f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/ba_only_kernel_stats.csv"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o pmc -- python "$R/bench.py" --steps 16 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/pmc_$c.log" 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python "$R/scripts/pmc_summary.py" $c /tmp/prof_$c "$OUT/pmc_$lc.csv" > /dev/null 2>&1
done
# MFMA busy cycles of the K3 kernels (counters only)
bash "$R/scripts/pmc_kernel.sh" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 > "$OUT/sq_counters.txt" 2>&1
bash "$R/scripts/pmc_kernel.sh" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 >> "$OUT/sq_counters.txt" 2>&1
bash "$R/scripts/pmc_kernel.sh" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 >> "$OUT/sq_counters.txt" 2>&1
# the fused path with intrinsics (FOCAL | RADIAL over the 8 groups of C4): kernel trace + the two traffic counters
bash "$R/scripts/prof_intr.sh" refresh/intr 8 > "$OUT/intr_summary.txt" 2>&1
cp "$R/gpurun_out/refresh/intr/kernel_stats.csv" "$OUT/intr_kernel_stats.csv" 2>/dev/null
bash "$R/scripts/pmc_kernel.sh" "FETCH_SIZE" scripts/gpu_time_intr_c4.py 8 > "$OUT/intr_pmc_fetch_size.txt" 2>&1
bash "$R/scripts/pmc_kernel.sh" "WRITE_SIZE" scripts/gpu_time_intr_c4.py 8 > "$OUT/intr_pmc_write_size.txt" 2>&1
# DLS leg at the C5 shape: kernel trace + the SQ counter passes of stage A / stage B (250 pairs)
bash "$R/scripts/prof_ransac_leg.sh" dls 1000 > "$OUT/dls_summary.txt" 2>&1
cp "$R/gpurun_out/dls_prof/kernel_stats.csv" "$OUT/dls_kernel_stats.csv" 2>/dev/null
bash "$R/scripts/pmc_sq_ransac.sh" dls 250 > "$OUT/sq_dls.txt" 2>&1
# P4Pfr leg at the C5 shape (1000 pairs x 2000 x 4096): kernel trace
bash "$R/scripts/prof_ransac_leg.sh" p4pfr 1000 > "$OUT/p4pfr_summary.txt" 2>&1
cp "$R/gpurun_out/p4pfr_prof/kernel_stats.csv" "$OUT/p4pfr_kernel_stats.csv" 2>/dev/null
# the pipelines' default configuration (FOCAL | RADIAL free + inner iterations) at C4: kernel trace
bash "$R/scripts/prof_script.sh" default_prof scripts/gpu_time_inner_c4.py 0x11 > "$OUT/default_summary.txt" 2>&1
cp "$R/gpurun_out/default_prof/kernel_stats.csv" "$OUT/default_kernel_stats.csv" 2>/dev/null
# what the counters were collected on: hashes of the kernel sources (bench.py compares them with the tree it runs in)
python - "$R" > "$OUT/collected_at.json" <<'PY'
import hashlib, json, os, sys
R = sys.argv[1]
files = ["pytheiasfm_amd/csrc/ba_fused.hip", "pytheiasfm_amd/csrc/ba_fused_intr.hip", "pytheiasfm_amd/csrc/ba_lane.h", "pytheiasfm_amd/csrc/ba_device.h",
         "pytheiasfm_amd/csrc/sparse_cholesky.hip", "pytheiasfm_amd/csrc/cholesky_device.h", "pytheiasfm_amd/csrc/ba_kernels.h",
         "pytheiasfm_amd/csrc/ba_solver.hip", "pytheiasfm_amd/csrc/ba_fused_lin.h", "pytheiasfm_amd/csrc/ba_kernels.hip",
         "pytheiasfm_amd/csrc/ba_priors.h", "pytheiasfm_amd/csrc/ba_inner.hip"]
print(json.dumps({f: hashlib.sha256(open(os.path.join(R, f), "rb").read()).hexdigest()[:16] for f in files}))
PY
tail -c 300 "$OUT/bench.err"
ls -la "$OUT"
head -5 "$OUT/pmc_fetch_size.csv" "$OUT/pmc_write_size.csv"
