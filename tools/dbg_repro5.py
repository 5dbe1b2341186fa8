"""Diagnostic (GPU box): reproducibility of the 512-window benchmark batch -- the same batch solved twice in one context,
and its second half solved on its own; lists the windows whose states differ by more than 1e-9 and their accept patterns."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 512
opt = v.default_options()
cfg = v.workload.config(200, 80, True)
ctx = v.Context(device=0, max_windows=NW, max_points=200, max_point_obs=1200, max_lines=80, max_line_obs=480)
B, keep = v.workload.primed_batch(ctx, range(NW), cfg, opt, 3)
pristine = [b.copy() for b in B]
ctx.upload(B, opt)
runs = []
for r in range(3):
    ctx.reset_state(); ctx.solve(); ctx.synchronize()
    _, rep = ctx.download()
    runs.append((v.shard.pack_states(B), [(rep[i].iterations, rep[i].num_successful_steps, rep[i].final_cost) for i in range(NW)]))
for r in (1, 2):
    d = np.abs(runs[r][0] - runs[0][0]).max(axis=1)
    bad = np.nonzero(d > 1e-9)[0]
    print("run %d vs 0: max diff %.3e, windows > 1e-9: %s" % (r, d.max(), list(bad[:20])))
    for w in bad[:10]:
        print("   window %d diff %.3e  run0 %s  run%d %s" % (w, d[w], runs[0][1][w], r, runs[r][1][w]))
half = [pristine[g].copy() for g in range(NW // 2, NW)]
c2 = v.Context(device=0, max_windows=NW // 2, max_points=200, max_point_obs=1200, max_lines=80, max_line_obs=480)
_, rep2 = c2.solve_windows(half, opt)
st = v.shard.pack_states(half)
d = np.abs(st - runs[0][0][NW // 2:]).max(axis=1)
bad = np.nonzero(d > 1e-9)[0]
print("second half on its own vs batch: max diff %.3e, windows > 1e-9: %s" % (d.max(), list(bad[:20] + NW // 2)))
for w in bad[:10]:
    print("   window %d diff %.3e  batch %s  alone %s" % (w + NW // 2, d[w], runs[0][1][w + NW // 2], (rep2[w].iterations, rep2[w].num_successful_steps, rep2[w].final_cost)))
