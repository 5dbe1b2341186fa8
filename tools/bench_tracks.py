"""k_schur on batches with long tracks: per-kernel ms per step (512 windows, 200 points + 80 lines) for
  short   every track 6 frames (the benchmark shape: k_schur<3>)
  graft   a tenth of the point tracks over all 11 frames (bench.py's steady state: k_schur_mixed, few wide entries)
  long    every track 11 frames (every entry wide)
Run twice: default, and VPL_BA_SCHUR_WIDE=1 (round 3's k_schur<5> for every batch with rows wider than 6 frames)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vplines_slam_amd as v

dev = torch.device("cuda", 0)
opt = v.default_options()
nW = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for name in ("short", "graft", "long"):
    cfg = v.workload.config(200, 80, True)
    if name == "long":
        cfg.track_len = 11
    gen = (lambda s, t: v.workload.graft_long_tracks(s, cfg, t)) if name == "graft" else (lambda s, t: v.workload.generate(s, cfg, t))
    ws = [gen(v.workload.seed_for(3, 50000 + i), 0.37 * i) for i in range(nW)]
    pobs = max(int(w.point_nobs.sum()) for w in ws)
    lobs = max(int(w.line_nobs.sum()) for w in ws)
    ctx = v.Context(device=0, max_windows=nW, max_points=200, max_point_obs=pobs, max_lines=80, max_line_obs=lobs)
    v.workload.set_preintegrations(ws, ctx.preintegrate(*v.workload.imu_batch_arrays(ws), opt))
    ctx.upload(ws, opt)
    for _ in range(2):
        ctx.reset_state(); ctx.solve()
    ctx.synchronize()
    ctx.enable_kernel_timing(True)
    reps = 5
    for _ in range(reps):
        ctx.reset_state(); ctx.solve(); ctx.synchronize()
    kt = ctx.kernel_times()
    ctx.enable_kernel_timing(False)
    print(name, "SCHUR_WIDE=%s" % os.environ.get("VPL_BA_SCHUR_WIDE", "0"),
          {k: round(ms / reps, 4) for k, (ms, _) in kt.items() if k in ("k_schur", "k_back", "k_lin", "k_marg", "k_chol")})
    ctx.close()
