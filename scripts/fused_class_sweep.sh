R="${GRAFT_REPO_ROOT:-/root/repo}"
run() { python "$R/bench.py" --steps 24 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['roofline']['avg_launch_ms'], b['config']['fused_kernel_runs'])"; }
export THEIA_HIP_FUSED_RUN_OBS=1280
echo "1280 cut7"; run; run
echo "1280 cut6"; THEIA_HIP_FUSED_CUT0=6 run; THEIA_HIP_FUSED_CUT0=6 run
echo "1280 3cls"; THEIA_HIP_FUSED_CLASSES=3 run; THEIA_HIP_FUSED_CLASSES=3 run
export THEIA_HIP_FUSED_RUN_OBS=1152
echo "1152 cut7"; run; echo "1152 cut6"; THEIA_HIP_FUSED_CUT0=6 run
export THEIA_HIP_FUSED_RUN_OBS=1408
echo "1408 cut7"; run; echo "1408 cut6"; THEIA_HIP_FUSED_CUT0=6 run
