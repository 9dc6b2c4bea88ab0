R="${GRAFT_REPO_ROOT:-/root/repo}"
run() { python "$R/bench.py" --steps 24 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['roofline']['avg_launch_ms'], b['config']['fused_kernel_runs'])"; }
echo "default"; run
for r in 1280 1536 1792 2048 2048 2304 2560 2816; do echo "run_obs=$r"; THEIA_HIP_FUSED_RUN_OBS=$r run; done
