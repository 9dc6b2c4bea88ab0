#!/bin/bash
# GPU box: one rocprofv3 PMC pass (counters only) over a python script; per-kernel mean counter values -> stdout
# usage: pmc_kernel.sh "<COUNTERS space separated>" <script.py> [args...]
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
C="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pmc
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -o pmc -- python "$R/$@" > /tmp/pmc_run.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for fn in glob.glob("/tmp/prof_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("thip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    if not any(x in k for x in ("k_lin_schur", "k_backsub", "k_schur_sum", "k_sum_items", "k_sp_", "k_fit", "k_dls", "k_score", "k_sqp", "k_p4pf", "k_id_")): continue
    print(k, {c: round(v[1] / v[0], 1) for c, v in d.items()})
PY
tail -n 2 /tmp/pmc_run.log
