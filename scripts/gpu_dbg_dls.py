import sys
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytheiasfm_amd import ransac
from tests import oracle_lib as ol
rng = np.random.default_rng(5)
def q2r(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y)],[2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x)],[2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)]])
def rand_rot():
    q = rng.normal(size=4); q /= np.linalg.norm(q); return q2r(q)
fl, wl, gt = [], [], []
for k in range(64):
    n = [3, 4, 5, 10, 50, 200][k % 6]
    R = rand_rot(); t = rng.normal(size=3)
    Xc = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(2, 6, n)]
    fl.append(Xc[:, :2] / Xc[:, 2:3]); wl.append((Xc - t) @ R); gt.append((R, t))
ns, q, t = ransac.DlsPnp(fl, wl)
for k in range(64):
    qo, to = ol.dls_pnp(fl[k], wl[k], call_index=k)
    R, tt = gt[k]
    eg = min([np.abs(q2r(q[k, i]) - R).max() + np.abs(t[k, i] - tt).max() for i in range(ns[k])] or [9])
    eo = min([np.abs(q2r(qo[i]) - R).max() + np.abs(to[i] - tt).max() for i in range(len(qo))] or [9])
    d = max([min(np.abs(q[k, i] - qo[j]).max() + np.abs(t[k, i] - to[j]).max() for j in range(len(qo))) for i in range(ns[k])] or [0])
    if max(eg, eo, d) > 1e-7: print(k, "n", fl[k].shape[0], "nsol", ns[k], len(qo), "gt err gpu %.2e oracle %.2e, gpu-oracle %.2e" % (eg, eo, d))
