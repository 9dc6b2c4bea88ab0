#!/bin/bash
# GPU box (development): rebuilds ransac.hip with -DTHIP_EIG_STAMPS into a scratch copy of the library and prints the phase split of
# eig_team (eig_team.h) for the DLS and five-point legs.  The committed library is not touched.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R/pytheiasfm_amd/csrc"
mkdir -p /tmp/stamps_obj
OBJS=""
for f in *.hip; do
  o=_obj/${f%.hip}.o
  if [ "$f" = "ransac.hip" ]; then
    o=/tmp/stamps_obj/ransac.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I../../include -I. -DTHIP_EIG_STAMPS -c "$f" -o "$o" || exit 1
  fi
  if [ "$f" = "dls_kernels.hip" ]; then
    o=/tmp/stamps_obj/dls_kernels.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -simplifycfg-sink-common=false -munsafe-fp-atomics -I../../include -I. -DTHIP_DLS_STAMPS -c "$f" -o "$o" || exit 1
  fi
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stamps_obj/libtheia_hip.so $OBJS || exit 1
cd "$R"
python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pytheiasfm_amd import _capi as capi, ransac, synth
capi.LIB_PATH = "/tmp/stamps_obj/libtheia_hip.so"   # the instrumented copy
L = capi.lib()
names = ["orthes", "accumulate", "deflation + roots", "shift + start search", "chase steps", "back-substitution", "back-transformation"]
for leg, est, kind, thr in (("dls", ransac.EST_ABS_DLS, "absolute", (4 / 1000.0) ** 2), ("five_point", ransac.EST_RELATIVE_POSE, "relative", (2 / 1000.0) ** 2)):
    data, offsets, _ = synth.synth_ransac_v1(64, 2000, kind, seed=0x5AC50005)
    p = ransac.RansacParameters(); p.error_thresh = thr; p.min_iterations = 4096; p.max_iterations = 4096; p.seed = 1
    out = (C.c_ulonglong * 8)()
    ransac.estimate_batch(est, data, offsets, p)
    L.theia_hip_debug_eig_stamps(out)
    if leg == "dls":
        L.theia_hip_debug_dls_stamps((C.c_ulonglong * 16)())
    ransac.estimate_batch(est, data, offsets, p)
    L.theia_hip_debug_eig_stamps(out)
    if leg == "dls":
        d = (C.c_ulonglong * 16)()
        L.theia_hip_debug_dls_stamps(d)
        dn = ["front end", "register load", "elimination", "back-substitution", "M00 - M01 X + stores"]
        dt = float(sum(d[:5]))
        print("dls stage A: workgroups", d[5], "ticks per workgroup", dt / max(1, d[5]))
        for k in range(5):
            print("  %-22s %5.1f %%  %10.0f ticks" % (dn[k], 100.0 * d[k] / dt, d[k] / max(1, d[5])))
        sn = ["pivot block (thread 0's wave holds the pivot column in 2 of 6 steps)", "barrier", "pivot row out", "update"]
        for k in range(4):
            print("  step: %-70s %8.0f ticks per step" % (sn[k], d[8 + k] / max(1, d[12])))
    tot = float(sum(out[:7])); n = max(1, out[7])
    print(leg, "matrices", n, "ticks per matrix", tot / n)
    for k in range(7):
        print("  %-22s %5.1f %%  %10.0f ticks per matrix" % (names[k], 100.0 * out[k] / tot, out[k] / n))
PY
