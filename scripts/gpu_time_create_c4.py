"""Wall time of create() at BASELINE's C4, with the stage ticks (THEIA_HIP_CREATE_TIMING=1) on stderr."""
import sys, time, os
os.environ["THEIA_HIP_CREATE_TIMING"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pytheiasfm_amd import synth, ba
p = synth.ba_config("C4")
o = ba.default_options()
h = ba.BaHandle(p.copy(), o); del h
for rep in range(3):
    q = p.copy()
    t = time.time(); h = ba.BaHandle(q, o); t1 = time.time() - t
    t = time.time(); s, tr = h.run(); t2 = time.time() - t
    t = time.time(); h.download(); t3 = time.time() - t
    t = time.time(); del h; t4 = time.time() - t
    print("create %.1f ms  run %.1f ms (%d it)  download %.1f ms  destroy %.1f ms"
          % (1e3 * t1, 1e3 * t2, s.num_iterations, 1e3 * t3, 1e3 * t4), flush=True)
