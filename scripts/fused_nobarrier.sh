#!/bin/bash
# Development (GPU box, timing only -- wrong results): k_lin_schur without the two workgroup barriers per sub-chunk,
# to see how much of its time is waves waiting for each other at the phase boundaries.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R/pytheiasfm_amd/csrc"
cp ../libtheia_hip.so /tmp/libtheia_hip.orig.so
run() { python "$R/bench.py" --steps 24 --warmup 8 --no-cpu-baseline --no-ransac --no-c2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['roofline']['avg_launch_ms'])"; }
echo "== with barriers"; python "$R/scripts/gpu_time_lin_kernel.py" | tail -1
sed -e '266s/__syncthreads();/__builtin_amdgcn_wave_barrier();/' -e '328s/__syncthreads();/__builtin_amdgcn_wave_barrier();/' ba_fused.hip > /tmp/ba_fused_nb.hip
cp /tmp/ba_fused_nb.hip ./ba_fused_nb_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -freciprocal-math -fno-math-errno -fapprox-func -munsafe-fp-atomics -I../../include -I. -c ba_fused_nb_tmp.hip -o /tmp/ba_fused_nb.o
rm -f ba_fused_nb_tmp.hip
OBJS=""; for o in _obj/*.o; do b=$(basename $o .o); if [ "$b" = "ba_fused" ]; then OBJS="$OBJS /tmp/ba_fused_nb.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtheia_hip.so $OBJS
echo "== without the sub-chunk barriers (wrong results)"; timeout 200 python "$R/scripts/gpu_time_lin_kernel.py" | tail -3
cp /tmp/libtheia_hip.orig.so ../libtheia_hip.so
