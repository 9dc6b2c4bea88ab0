import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pytheiasfm_amd import synth, ba
p = synth.ba_config("C2")
o = ba.default_options()
h = ba.BaHandle(p.copy(), o); del h
for rep in range(3):
    t = time.time(); h = ba.BaHandle(p.copy(), o); t1 = time.time() - t
    t = time.time(); s, tr = h.run(); t2 = time.time() - t
    t = time.time(); h.download(); t3 = time.time() - t
    print("create %.1f ms  run %.1f ms (%d it)  download %.1f ms" % (1e3 * t1, 1e3 * t2, s.num_iterations, 1e3 * t3), flush=True)
    del h
o.intrinsics_to_optimize = 1 | 8  # focal | radial
t = time.time(); h = ba.BaHandle(p.copy(), o); print("INTR create %.1f ms" % (1e3 * (time.time() - t)))
