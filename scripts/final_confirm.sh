#!/bin/bash
# GPU box: the round's last check on the committed tree -- clean rebuild by build(), smoke(), the default bench line,
# pytest -m gpu -- summarised as gpurun_out/final_confirm.json (copied to profiles/<tag>_final_confirm.json).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
rm -rf pytheiasfm_amd/csrc/_obj pytheiasfm_amd/libtheia_hip.so
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/confirm_build.log 2>&1; brc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/confirm_smoke.log 2>&1; src=$?
timeout 900 python bench.py > gpurun_out/confirm_bench.json 2> gpurun_out/confirm_bench.err
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/confirm_tests.log 2>&1
python - "$brc" "$src" <<'PY'
import json, sys
line = open("gpurun_out/confirm_bench.json").read().strip().splitlines()[-1]
tests = [l for l in open("gpurun_out/confirm_tests.log").read().splitlines() if " passed" in l or " failed" in l]
out = {"what": "clean rebuild by build(), smoke(), default bench.py, pytest -m gpu on a fresh MI355X box", "build_rc": int(sys.argv[1]),
       "smoke_rc": int(sys.argv[2]), "gpu_tests": tests[-1] if tests else "no summary line", "bench_line": json.loads(line)}
json.dump(out, open("gpurun_out/final_confirm.json", "w"), indent=1)
b = out["bench_line"]
print(out["build_rc"], out["smoke_rc"], out["gpu_tests"], b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline"]["traffic"], b["roofline_k3"]["avg_solve_ms"])
PY
