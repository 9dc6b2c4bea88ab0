#!/bin/bash
# Development (GPU box): time the SQPnP RANSAC fit kernel with and without its 9 x 9 SVD (the variant returns wrong results)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R/pytheiasfm_amd/csrc"
cp ../libtheia_hip.so /tmp/libtheia_hip.orig.so
echo "== full"; python "$R/scripts/gpu_time_ransac.py" 4 | tail -1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -DTHIP_SQPNP_SKIP_SVD -I../../include -I. -c ransac.hip -o /tmp/ransac_nosvd.o
OBJS=""; for o in _obj/*.o; do b=$(basename $o .o); if [ "$b" = "ransac" ]; then OBJS="$OBJS /tmp/ransac_nosvd.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtheia_hip.so $OBJS
echo "== without the SVD"; python "$R/scripts/gpu_time_ransac.py" 4 | tail -1
cp /tmp/libtheia_hip.orig.so ../libtheia_hip.so
