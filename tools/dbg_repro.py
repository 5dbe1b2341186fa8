"""Diagnostic: run-to-run reproducibility of the batched solve on identical inputs (atomics make the summation order
vary; anything beyond rounding-level differences points at a race or a borderline accept/reject decision).
usage: dbg_repro.py P L vp windows reps marg_flag iterations"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as o
import vplines_slam_amd as v

def main():
    P, L, vp, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    flag = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    opt = v.default_options()
    opt.marginalization_flag = flag
    opt.num_iterations = iters
    cfg = v.workload.config(P, L, bool(vp))
    ws = [v.workload.generate(v.workload.seed_for(3, 60 + i), cfg, 0.61 * (60 + i)) for i in range(n)]
    o.preintegrate_windows(ws, opt)
    ctx = v.Context(device=0, max_windows=n)
    wg = [w.copy() for w in ws]
    ctx.upload(wg, opt)
    res = []
    for r in range(reps):
        ctx.reset_state()
        ctx.solve()
        ctx.synchronize()
        _, rp = ctx.download()
        res.append(dict(pose=np.stack([w.pose.copy() for w in wg]), invd=np.stack([w.inv_depth.copy() for w in wg]),
                        plk=np.stack([w.line_plk.copy() for w in wg]),
                        ic=np.array([x.initial_cost for x in rp]), fc=np.array([x.final_cost for x in rp]),
                        it=np.array([x.iterations * 100 + x.num_successful_steps for x in rp])))
    for key in ("ic", "fc", "pose", "invd", "plk"):
        a = np.stack([r[key] for r in res]).reshape(reps, n, -1)
        spread = (a.max(axis=0) - a.min(axis=0)).max(axis=1)
        scale = np.abs(a).max(axis=(0, 2)) if key in ("ic", "fc") else 1.0
        rel = spread / scale
        w = int(np.argmax(rel))
        print("%-5s worst window %3d spread %.3e  (median over windows %.3e)" % (key, w, rel[w], np.median(rel)))
    it = np.stack([r["it"] for r in res])
    print("iteration/accept pattern differs between runs in windows:", np.nonzero((it != it[0]).any(axis=0))[0])

main()
