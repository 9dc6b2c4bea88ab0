#!/bin/bash
# GPU box (development): rebuilds ONE translation unit with extra -D switches into a scratch copy of the library and runs a command
# against it (THEIA_HIP_LIBRARY).  usage: dev_variant.sh <file.hip> "<-D switches>" <command...>
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
F="$1"; DEFS="$2"; shift 2
cd "$R/pytheiasfm_amd/csrc"
D=/tmp/variant_obj; mkdir -p $D
CONTRACT=off
case "$F" in ba_fused.hip|ba_fused_intr.hip|ba_kernels.hip) CONTRACT="fast -freciprocal-math -fno-math-errno -fapprox-func" ;; esac
case "$F" in dls_kernels.hip|upnp_kernels.hip|ba_inner.hip) CONTRACT="off -mllvm -simplifycfg-sink-common=false" ;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=$CONTRACT -munsafe-fp-atomics -I../../include -I. $DEFS -c "$F" -o "$D/variant.o" 2>/dev/null || { echo "variant build failed"; exit 1; }
OBJS=""
for f in *.hip; do o=_obj/${f%.hip}.o; [ "$f" = "$F" ] && o="$D/variant.o"; OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libtheia_hip.so $OBJS || exit 1
cd "$R"
THEIA_HIP_LIBRARY=$D/libtheia_hip.so "$@"
