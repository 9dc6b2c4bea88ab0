#!/bin/bash
# Development: build ba_fused.hip / ba_kernels.hip with alternative AMDGPU scheduler strategies and time the C4 bench
# with each (run on the GPU box; the variant .so replaces libtheia_hip.so inside the box's copy only).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R/pytheiasfm_amd/csrc"
cp ../libtheia_hip.so /tmp/libtheia_hip.orig.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -freciprocal-math -fno-math-errno -fapprox-func -munsafe-fp-atomics -I../../include -I."
run() { python "$R/bench.py" --steps 24 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['phase_ms_per_iteration'], b['roofline']['avg_launch_ms'])"; }
echo "== default"; run
for strat in max-ilp max-memory-clause; do
  for f in ba_fused ba_kernels; do /opt/rocm/bin/hipcc $FL -mllvm -amdgpu-sched-strategy=$strat -c $f.hip -o /tmp/${f}_$strat.o & done; wait
  OBJS=""; for o in _obj/*.o; do b=$(basename $o .o); if [ -f /tmp/${b}_$strat.o ]; then OBJS="$OBJS /tmp/${b}_$strat.o"; else OBJS="$OBJS $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtheia_hip.so $OBJS
  echo "== $strat"; run
done
cp /tmp/libtheia_hip.orig.so ../libtheia_hip.so
cd "$R/scripts/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 fp64_ilp.hip -o /tmp/fp64_ilp && /tmp/fp64_ilp
