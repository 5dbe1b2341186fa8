import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C
import vplines_slam_amd as v, oracle_api as o
import test_gpu_chain as t
from test_gpu_solve import pose_err
ws, opt = t._windows()
ctx = v.Context(device=0, max_windows=4)
dev, prior = [], None
for w in ws:
    c = w.copy(); c.prior = prior
    pri, rep = ctx.solve_windows([c], opt)
    prior = t._copy_prior(pri[0]); dev.append((c, rep[0], prior))
ref = t._oracle_chain(ws, opt, 0.0)
rel = t._oracle_chain(ws, opt, float(os.environ.get("REL", "1e-9")))
for k in range(t.N_CHAIN):
    (wd, rd, pd), (wr, rr, pr), (wq, rq, pq) = dev[k], ref[k], rel[k]
    a = pose_err(wd, wr); b = pose_err(wq, wr); c = pose_err(wd, wq)
    Jd, Jr, Jq = pd.J(), pr.J(), pq.J()
    Ar = Jr.T @ Jr
    lam = np.linalg.eigvalsh(0.5*(Ar+Ar.T))
    print(k, "it %d/%d/%d acc %d/%d/%d" % (rd.iterations, rr.iterations, rq.iterations, rd.num_successful_steps, rr.num_successful_steps, rq.num_successful_steps),
          "dev-ref %.2e %.2e | rel-ref %.2e %.2e | dev-rel %.2e %.2e" % (a + b + c),
          "| dA dev %.1e rel %.1e | lam min %.1e max %.1e n<1e-8: %d n<1e-9max: %d" % (np.abs(Jd.T@Jd-Ar).max()/np.abs(Ar).max(), np.abs(Jq.T@Jq-Ar).max()/np.abs(Ar).max(), lam[0], lam[-1], (lam<1e-8).sum(), (lam < 1e-9*np.diag(Ar).max()).sum()))
