"""k_lin2 with one / two work-groups per window: the states after a solve must be the same bits."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vplines_slam_amd as v
from test_gpu_solve import make_windows
nw = int(os.environ.get("NW", "16"))
for nit in (1, 2, 5):
    res = []
    for split in ("0", "1"):
        os.environ["VPL_BA_LIN_SPLIT"] = split
        ctx = v.Context(device=0, max_windows=nw, max_points=200, max_point_obs=1200, max_lines=80, max_line_obs=480)
        ws, opt = make_windows(nw, 200, 80, True)
        opt.num_iterations = nit
        pri, rep = ctx.solve_windows(ws, opt)
        res.append((v.shard.pack_states(ws), [r.final_cost for r in rep], [r.num_successful_steps for r in rep]))
        ctx.close()
    d = np.abs(res[0][0] - res[1][0]).max()
    print("iterations", nit, "max state diff", d, "cost diff", max(abs(a - b) for a, b in zip(res[0][1], res[1][1])), "steps equal", res[0][2] == res[1][2])
