"""GPU box (development, through scripts/dev_variant.sh upnp_kernels.hip -DTHIP_UPNP_STAMPS): the section split of a Gauss-Jordan
step of k_upnp_a (thread 0 of every workgroup, s_memtime)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytheiasfm_amd import _capi as capi, ransac, synth
data, offsets, _ = synth.synth_ransac_v1(64, 2000, "absolute", seed=0x5AC50005)
data = ransac.central_correspondence_rows(data)
p = ransac.RansacParameters(); p.error_thresh = (4 / 1000.0) ** 2; p.min_iterations = 4096; p.max_iterations = 4096; p.seed = 1
ransac.estimate_batch(ransac.EST_RIGID_TRANSFORMATION_2D3D, data, offsets, p)
out = (C.c_ulonglong * 8)()
capi.lib().theia_hip_debug_upnp_stamps(out)
ransac.estimate_batch(ransac.EST_RIGID_TRANSFORMATION_2D3D, data, offsets, p)
capi.lib().theia_hip_debug_upnp_stamps(out)
names = ["column out + barrier", "pivot search", "pivot row out + barrier", "row / pivot + barrier", "update"]
steps = max(1, out[5]); tot = sum(out[k] for k in range(5))
print(f"k_upnp_a: {out[6]} workgroups, {out[5] / max(1, out[6]):.0f} steps each, {tot / steps:.0f} ticks per step")
for k in range(5): print(f"  {names[k]:26s} {out[k] / steps:8.0f} ticks  {100.0 * out[k] / tot:5.1f} %")
