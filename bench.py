#!/usr/bin/env python3
"""bench.py -- sliding-window BA solves/sec on MI355X (BASELINE.json metric).

One step = one batched execution of the optimizationwithLine() body (<=5 trust-region
iterations -> gauge fix -> MARGIN_OLD marginalisation -> new prior) over a batch of
independent synthetic windows of the named shape (10 KF window = 11 frames, 200 points +
80 lines + VP observations, prior from a warm-up solve of the preceding window), inputs
already resident in HBM.  Multi-GPU: one process per GPU, the batch dimension is sharded,
no collective in the data path (weak scaling: every rank solves `--windows` windows).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import vplines_slam_amd as v

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(P, L, n_prior, obs_p, obs_l):
    """SURVEY.md 8d compulsory FP64 traffic of one trust-region iteration of one window, and the
    share of it that each kernel of this implementation must move (DESIGN.md 'bytes')."""
    F_p = obs_p - P          # point factors
    F_l = obs_l              # line factors (start frame included)
    F_v = obs_l
    reads = 8 * ((11 * 16 + 7 + P + 4 * L) + (6 * F_p + 4 * F_l + 3 * F_v) + 10 * 287 + (n_prior ** 2 + n_prior + 86))
    writes = 8 * ((171 * 171 + 171) + (2 * P + 20 * L) + (42 * P + 168 * L))
    k_lin = reads + 8 * ((171 * 172 // 2 + 171) + (2 * P + 20 * L) + (42 * P + 168 * L))
    k_solve = 8 * ((171 * 172 // 2 + 171) + (2 * P + 20 * L) + (42 * P + 168 * L)) + 8 * 2 * (171 + P + 4 * L)
    k_cost = 8 * ((11 * 16 + 7 + P + 4 * L) + (3 * obs_p + 8 * obs_l) + 10 * 62 + (n_prior ** 2 + n_prior + 86))
    return dict(iteration=reads + writes, k_lin=k_lin, k_solve=k_solve, k_cost=k_cost)


def pmc_traffic(kernel, nW, P, L):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*/pmc_traffic.json, produced
    by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this very command at the default workload,
    corrected as MI355X_MICROARCH.md prescribes).  None when the workload differs from the profiled one."""
    if (nW, P, L) != (512, 200, 80):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1])).get("kernels", {})
    key = {"k_lin": "k_lin<0>"}.get(kernel, kernel)
    return k[key]["total"] if key in k else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=512, help="windows per GPU per step")
    ap.add_argument("--points", type=int, default=200)
    ap.add_argument("--lines", type=int, default=80)
    ap.add_argument("--cpu-windows", type=int, default=0, help="oracle sample size (0 = auto, ~10-30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("VPL_FORCE_DIST"):   # VPL_FORCE_DIST: exercise the RCCL path with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    nW, P, L = args.windows, args.points, args.lines
    opt = v.default_options()
    cfg = v.workload.config(P, L, True)
    TL = cfg.track_len
    ctx = v.Context(device=local_rank, max_windows=nW, max_points=max(P, 1), max_point_obs=max(P * TL, 1),
                    max_lines=max(L, 1), max_line_obs=max(L * TL, 1))
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    # ---- setup (untimed): windows A (preceding window, no prior) and B (the timed batch) ----
    t_setup = time.time()
    lo, hi = v.shard.window_range(rank, world, nW)
    seeds = [v.shard.window_seeds(3, g) for g in range(lo, hi)]
    A = [v.workload.generate(sa, cfg, t) for (sa, sb, t) in seeds]
    B = [v.workload.generate(sb, cfg, t + cfg.kf_dt) for (sa, sb, t) in seeds]
    pre = ctx.preintegrate(*v.workload.imu_batch_arrays(A + B), opt)     # IntegrationBase on device
    v.workload.set_preintegrations(A + B, pre)
    priors, _ = ctx.solve_windows(A, opt)                                # warm-up solve -> priors for B
    keep = (v.Prior * nW)()
    C.memmove(keep, priors, C.sizeof(keep))
    for i in range(nW):
        B[i].prior = keep[i]
    pristine = [b.copy() for b in B]                                     # download() overwrites B in place
    ctx.upload(B, opt)                                                   # inputs now resident in HBM
    n_prior = int(np.mean([keep[i].n for i in range(nW)]))
    t_setup = time.time() - t_setup

    def step():
        ctx.reset_state()
        ctx.solve()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = v.shard.reduce_max(dist, elapsed, dev)
    value = world * nW * args.steps / elapsed

    # ---- per-kernel device time with HIP events on the launch stream (separate, un-timed pass) ----
    ctx.enable_kernel_timing(True)
    ksteps = max(3, min(args.steps, 10))
    for _ in range(ksteps):
        step()
    torch.cuda.synchronize(dev)
    kt = ctx.kernel_times()
    ctx.enable_kernel_timing(False)
    _, reports = ctx.download()
    iters = float(np.mean([reports[i].iterations for i in range(nW)]))
    succ = float(np.mean([reports[i].num_successful_steps for i in range(nW)]))

    out = None
    if rank == 0:
        ab = algorithmic_bytes(P, L, n_prior, P * TL, L * TL)
        kavg = {k: (ms / max(1, cnt)) for k, (ms, cnt) in kt.items()}
        total_ms = sum(ms for ms, _ in kt.values()) / ksteps
        dom = max(("k_lin", "k_solve", "k_cost"), key=lambda k: kt.get(k, (0, 1))[0])
        # windows that actually run the dominant kernel differ per launch (rejected steps skip the
        # re-linearisation): the launch processes the whole batch, priced at nW windows per launch
        achieved = ab[dom] * nW / (kavg[dom] * 1e-3) / 1e9
        iter_ms = sum(kt.get(k, (0, 0))[0] for k in ("k_lin", "k_solve", "k_cost")) / ksteps
        pipeline = ab["iteration"] * nW * iters / (iter_ms * 1e-3) / 1e9
        out = {
            "metric": "sliding-window BA solves/sec (10 KF, 200 pts + 80 lines, 5 iters)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch of %d independent synthetic sliding windows per GPU: 11 frames, %d points + "
                                   "%d lines (x%d obs) + VP obs, 10 IMU factors, prior n=%d from a warm-up solve; "
                                   "max 5 TR iterations + gauge fix + MARGIN_OLD marginalisation" % (nW, P, L, TL, n_prior),
                       "windows_per_gpu": nW, "points": P, "lines": L, "track_len": TL, "prior_dim": n_prior,
                       "mean_tr_iterations": iters, "mean_successful_steps": succ, "parallelism": "batch-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, nW, P, L),
                         "avg_launch_ms": kavg[dom], "algorithmic_bytes_per_launch": ab[dom] * nW,
                         "pipeline_GBps": pipeline, "pipeline_frac": pipeline / HBM_PEAK_GBS},
            "kernels_ms_per_step": {k: ms / ksteps for k, (ms, _) in kt.items()},
            "device_ms_per_step": total_ms,
            "setup_s": t_setup,
        }

    # ---- CPU baseline + parity on a bounded sample: rank 0, single-GPU run only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_api as o       # the oracle is the CPU baseline / checker here, never the product path
        cores = min(os.cpu_count() or 1, 16)   # the GPU box's CPU share for one GPU
        ns = args.cpu_windows or nW
        # single-thread figure first, on a small sample, to size the threaded run to ~10-20 s of work
        ns1 = max(2, min(ns, 8))
        s1 = [pristine[i].copy() for i in range(ns1)]
        t1 = time.perf_counter()
        o.solve_windows(s1, opt, threads=1)
        t1 = time.perf_counter() - t1
        reps = max(1, int(round(12.0 * cores * (ns1 / t1) / ns)))
        tc = 0.0
        for _ in range(reps):
            sample = [pristine[i].copy() for i in range(ns)]
            t = time.perf_counter()
            o.solve_windows(sample, opt, threads=cores)
            tc += time.perf_counter() - t
        dpm, drm = 0.0, 0.0
        from test_gpu_solve import pose_err
        for i in range(ns):
            dp, dr = pose_err(B[i], sample[i])   # B[i] holds the downloaded GPU result
            dpm, drm = max(dpm, dp), max(drm, dr)
        out["cpu_baseline"] = {"value": ns * reps / tc, "unit": "solves/s", "cores": cores, "kind": "port",
                               "sample": "%d of the %d timed windows x %d repeats, oracle (CPU restatement of the "
                                         "reference path) fanned over %d host threads, %.1f s" % (ns, nW, reps, cores, tc),
                               "single_thread_solves_per_s": ns1 / t1}
        out["parity"] = {"windows": ns, "max_dp_m": dpm, "max_dr_rad": drm}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
