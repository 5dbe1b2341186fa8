"""BASELINE config 5 / the benchmark batch itself under pytest: 512 independent config-3 windows (200 points + 80 lines +
VP observations, prior from the warm-up solve of the preceding window, built exactly as bench.py builds them --
workload.primed_batch) through vpl_ba_upload / vpl_ba_solve / vpl_ba_download in ONE batch.
 * oracle parity on 32 windows sampled across the whole batch: pose <= 1e-4 m / 1e-6 rad on every frame, the same
   iteration and accepted-step counts, the same prior through J0^T J0;
 * properties on all 512: finite states, termination, iteration bounds, prior size and block table, J0^T J0 = A of the
   marginalisation (the reference's own commented check, marginalization_factor.cpp:361-362) and A positive
   semi-definite up to rounding;
 * the block partition of the batch over 2 / 8 ranks reproduces the single-batch results window for window, BIT FOR BIT
   (what the multi-GPU run shards is the batch index and nothing else; the solve is deterministic)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as o
import vplines_slam_amd as v
from test_gpu_solve import POS_TOL, ROT_TOL, pose_err

pytestmark = pytest.mark.gpu

NW = 512


@pytest.fixture(scope="module")
def batch():
    opt = v.default_options()
    cfg = v.workload.config(200, 80, True)
    ctx = v.Context(device=0, max_windows=NW, max_points=200, max_point_obs=200 * cfg.track_len, max_lines=80,
                    max_line_obs=80 * cfg.track_len)
    B, keep = v.workload.primed_batch(ctx, range(NW), cfg, opt, 3)
    pristine = [b.copy() for b in B]
    ctx.upload(B, opt)
    ctx.solve()
    ctx.synchronize()
    priors, reports = ctx.download()
    yield dict(ctx=ctx, opt=opt, cfg=cfg, B=B, keep=keep, pristine=pristine, priors=priors, reports=reports)
    ctx.close()


def test_config5_oracle_parity_on_32_sampled_windows(batch):
    ids = [int(round(k * (NW - 1) / 31)) for k in range(32)]
    sample = [batch["pristine"][g].copy() for g in ids]
    pc, rc = o.solve_windows(sample, batch["opt"], threads=8)
    worst = (0.0, 0.0)
    for k, g in enumerate(ids):
        rg = batch["reports"][g]
        assert rg.iterations == rc[k].iterations, g
        assert rg.num_successful_steps == rc[k].num_successful_steps, g
        assert rg.termination == rc[k].termination, g
        assert abs(rg.initial_cost - rc[k].initial_cost) <= 1e-9 * rc[k].initial_cost
        dp, dr = pose_err(batch["B"][g], sample[k])
        assert dp <= POS_TOL and dr <= ROT_TOL, (g, dp, dr)
        worst = (max(worst[0], dp), max(worst[1], dr))
        assert rg.prior_n == rc[k].prior_n and rg.prior_m == rc[k].prior_m
        Jg, Jc = batch["priors"][g].J(), pc[k].J()
        Ac = Jc.T @ Jc
        assert np.abs(Jg.T @ Jg - Ac).max() <= 1e-5 * np.abs(Ac).max(), g
    print("config 5: worst of 32 sampled windows dp %.3g m, dr %.3g rad" % worst)


def test_config5_properties_on_all_512_windows(batch):
    ctx, reports, priors = batch["ctx"], batch["reports"], batch["priors"]
    lib = v.load_hip_library()
    lib.vpl_ba_debug_marg_Ab.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    n_it = np.array([reports[i].iterations for i in range(NW)])
    n_ok = np.array([reports[i].num_successful_steps for i in range(NW)])
    assert n_it.min() >= 1 and n_it.max() <= batch["opt"].num_iterations
    assert n_ok.min() >= 1 and np.all(n_ok <= n_it + 1)
    A = np.zeros(80 * 80)
    b = np.zeros(80)
    for g in range(NW):
        w, r, p = batch["B"][g], reports[g], priors[g]
        assert r.termination in (0, 1), g
        assert np.isfinite(w.pose).all() and np.isfinite(w.speed_bias).all() and np.isfinite(w.ex_pose).all()
        assert np.isfinite(w.inv_depth).all() and np.isfinite(w.line_plk).all()
        assert np.abs(np.linalg.norm(w.pose[:, 3:], axis=1) - 1).max() < 1e-12
        assert np.isfinite(r.final_cost) and r.final_cost <= r.initial_cost * (1 + 1e-12)
        # the prior of the NEXT window.  Window B's own prior (from the un-primed window A) spans the poses 0..4, the
        # tracks that start in frame 0 reach frame 5, IMU(0,1) brings speed/bias 1: after dropping frame 0 the kept
        # blocks are the poses 1..5, speed/bias 1 and the extrinsic = 5 * 6 + 9 + 6 = 45 dims, renumbered 0..4 / 0
        assert p.n == 45 and p.n_blocks == 7 and r.prior_n == 45 and r.prior_m == 15 + 34 + 4 * 14
        assert list(p.block_kind[:7]) == [0] * 5 + [1, 2]
        assert list(p.block_frame[:7]) == [0, 1, 2, 3, 4, 0, 0]
        assert list(p.block_idx[:7]) == [0, 6, 12, 18, 24, 30, 39]
        J, r0 = p.J(), p.r()
        assert np.isfinite(J).all() and np.isfinite(r0).all()
        n = lib.vpl_ba_debug_marg_Ab(ctx.h, g, A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)))
        assert n == 45
        Am = A[: n * n].reshape(n, n)
        assert np.abs(Am - Am.T).max() <= 1e-9 * np.abs(Am).max()
        assert np.abs(J.T @ J - Am).max() <= 1e-6 * np.abs(Am).max(), g
        lam = np.linalg.eigvalsh(0.5 * (Am + Am.T))
        assert lam[0] >= -1e-7 * lam[-1], (g, lam[0], lam[-1])


@pytest.mark.parametrize("world", [2, 8])
def test_config5_block_partition_reproduces_the_batch(batch, world):
    """every rank's block solved on its own (same context type, same inputs) gives the states of the single batch"""
    opt, cfg = batch["opt"], batch["cfg"]
    full = v.shard.pack_states(batch["B"])
    ctx = v.Context(device=0, max_windows=NW // world, max_points=200, max_point_obs=200 * cfg.track_len, max_lines=80,
                    max_line_obs=80 * cfg.track_len)
    parts = []
    for r in (0, world - 1):
        lo, hi = v.shard.split_batch(NW, r, world)
        blk = [batch["pristine"][g].copy() for g in range(lo, hi)]
        ctx.solve_windows(blk, opt)
        st = v.shard.pack_states(blk)
        # A window's arithmetic depends neither on the batch it sits in nor on the run: every accumulation of the solve
        # happens in a fixed order, so the block solved on its own gives the states of the single batch bit for bit
        assert np.array_equal(st, full[lo:hi]), (r, np.abs(st - full[lo:hi]).max())
        parts.append((lo, hi))
    assert parts[0][0] == 0 and parts[-1][1] == NW
    ctx.close()


_PACK_SCRIPT = r"""
import sys, os
import numpy as np
import torch
torch.cuda.init()                      # torch first: it brings its own HIP runtime, as in bench.py
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import vplines_slam_amd as v
from test_gpu_solve import make_windows
dev = torch.device("cuda", 0)
ws, opt = make_windows(12, 60, 20, True, seed0=300)
ctx = v.Context(device=0, max_windows=12, max_points=60, max_point_obs=360, max_lines=20, max_line_obs=120)
ctx.solve_windows(ws, opt)
t = v.shard.gather_states_device(None, ctx, 12, 12, dev)
assert t.is_cuda and tuple(t.shape) == (12, v.shard.STATE_DOUBLES)
assert np.array_equal(t.cpu().numpy(), v.shard.pack_states(ws))
print("PACK_OK")
"""


def test_states_packed_on_the_device_equal_the_downloaded_ones():
    """vpl_ba_pack_states_device + shard.gather_states_device (what bench.py hands to the RCCL all-gather): bit for bit the
    [nW][183] table shard.pack_states builds from the downloaded windows.  Own process: torch's HIP runtime has to be
    initialised before the library's (bench.py's order)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _PACK_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "PACK_OK" in out.stdout, out.stderr[-2000:]
