R="${GRAFT_REPO_ROOT:-/root/repo}"
run() { python "$R/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-ransac 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['c2']['ms_per_step'], b['c2']['phase_ms_per_iteration']['linearize_schur'], b['c2']['roofline']['avg_launch_ms'])"; }
echo default; run
for r in 256 320 512; do echo "run_obs=$r"; THEIA_HIP_FUSED_RUN_OBS=$r run; done
