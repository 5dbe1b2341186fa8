#!/usr/bin/env python3
"""bench.py -- sliding-window BA solves/sec on MI355X (BASELINE.json metric).

One step = one batched execution of the optimizationwithLine() body (<= 5 trust-region iterations -> gauge fix ->
MARGIN_OLD marginalisation -> new prior) over a batch of independent synthetic windows of the named shape (10 KF window
= 11 frames, 200 points + 80 lines + VP observations, prior from a warm-up solve of the preceding window), inputs
already resident in HBM.

Multi-GPU (BASELINE config 5): ONE batch of `--windows` (512) windows is block-partitioned over the
ranks (512 / 256 / 128 / 64 per GPU at 1 / 2 / 4 / 8), one process per GPU, every rank solves its block, then one RCCL
all-gather of the per-window states (183 doubles per window) and an all-reduce(MAX) of the timing / parity summary.
`value` is that strong-scaling figure; `weak_scaling` in the same JSON line is the same measurement with `--windows`
windows on EVERY rank.  `python bench.py --gpus N` without a launcher starts the N ranks itself (before any GPU call);
under torchrun, WORLD_SIZE must equal --gpus.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(P, L, n_prior, obs_p, obs_l):
    """SURVEY.md 8d compulsory FP64 traffic of one trust-region iteration of ONE window, and the share of it that each
    kernel of this implementation must move (DESIGN.md 'bytes')."""
    F_p = obs_p - P          # point factors
    F_l = obs_l              # line factors (start frame included)
    F_v = obs_l
    nfull = 171 + P + 4 * L
    reads = 8 * ((11 * 16 + 7 + P + 4 * L) + (6 * F_p + 4 * F_l + 3 * F_v) + 10 * 287 + (n_prior ** 2 + n_prior + 86))
    writes = 8 * ((171 * 171 + 171) + (2 * P + 20 * L) + (42 * P + 168 * L))
    lin_out = 8 * ((171 * 172 // 2 + 171) + (2 * P + 20 * L) + (42 * P + 168 * L))
    # SURVEY 8(d)'s bytes and nothing else: what the implementation moves on top of them (k_lin2's hand-over of the point
    # work-group's part of the visual Hessian through HBM, 47 KB per split window; the lines' world Pluecker cache) is
    # implementation traffic, not compulsory traffic, and shows up in `traffic`, not here
    k_lin = reads + lin_out
    # the trust-region step (k_schur + k_chol + k_back, or k_solve on the general path) of a window that computes a new
    # Gauss-Newton step reads the linearisation once; one that re-uses the step of a rejected iteration moves vectors only
    k_solve = lin_out + 8 * 2 * nfull
    k_solve_reuse = 8 * 6 * nfull
    k_cost = 8 * ((11 * 16 + 7 + P + 4 * L) + (3 * obs_p + 8 * obs_l) + 10 * 62 + (n_prior ** 2 + n_prior + 86))
    return dict(iteration=reads + writes, k_lin=k_lin, k_solve=k_solve, k_solve_reuse=k_solve_reuse, k_cost=k_cost)


FP64_PEAK_TFLOPS = 78.6  # MI355X dense FP64 (vector = matrix) peak (MI355X_MICROARCH.md)
STEP_KERNELS = ("k_schur", "k_chol", "k_solve", "k_back")    # the launches of one trust-region step


def _pmc_file(name):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    return files[-1] if files else None


def _pmc_keys(kernel):
    if kernel == "k_step":
        return ["k_schur<3>", "k_chol", "k_solve", "k_back"]
    return [{"k_lin": "k_lin2"}.get(kernel, kernel)]      # (k_lin2: the solve pass, two work-groups per window)


def pmc_traffic(kernel, nW, P, L, which="total"):
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC passes (profiles/r*/pmc_traffic.json, produced by
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this very command at the default workload,
    corrected as MI355X_MICROARCH.md prescribes) -- not measured in this run.  None when the workload differs from the
    profiled one or the profile does not hold the kernel."""
    if (nW, P, L) != (512, 200, 80):
        return None
    f = _pmc_file("pmc_traffic.json")
    if not f:
        return None
    k = json.load(open(f)).get("kernels", {})
    keys = _pmc_keys(kernel)
    if not all(q in k for q in keys):
        return None
    return sum(k[q].get(which, 0.0) for q in keys)


def pmc_flops(kernel, nW, P, L):
    """FP64 operations per launch of `kernel` (mean over the launches of the profiled run) counted by the SQ instruction
    counters (profiles/r*/pmc_flops.json: 64 x (ADD + MUL + TRANS + 2 FMA) + 512 x MFMA_MOPS), also from the committed
    profile.  Returns (flops, n_launches_sampled) or None."""
    if (nW, P, L) != (512, 200, 80):
        return None
    f = _pmc_file("pmc_flops.json")
    if not f:
        return None
    k = json.load(open(f)).get("kernels", {})
    keys = _pmc_keys(kernel)
    if not all(q in k for q in keys):
        return None
    return sum(k[q]["flops"] for q in keys)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (torch.distributed.run) BEFORE this process
    touches the GPU, pass their output through and exit with their code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_share():
    """host cores this process may really use: the affinity mask, cut by the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + per // 2) // per))
        except (OSError, ValueError):
            pass
    return n


class Batch:
    """One resident batch of primed windows on a context + the timing / profile helpers."""

    def __init__(self, v, ctx, ids, cfg, opt, config_id, steady_chain=0):
        self.v, self.ctx, self.ids = v, ctx, list(ids)
        t0 = time.time()
        if steady_chain:   # the same windows behind a chain of solves: the reference's steady-state prior (n = 75)
            self.B, self.n_prior = v.workload.steady_batch(ctx, self.ids, cfg, opt, config_id, steady_chain)
            self.setup_s = time.time() - t0
            return
        self.B, self.keep = v.workload.primed_batch(ctx, self.ids, cfg, opt, config_id)
        self.pristine = [b.copy() for b in self.B]      # download() overwrites B in place
        ctx.upload(self.B, opt)                          # inputs now resident in HBM
        self.n_prior = int(round(sum(self.keep[i].n for i in range(len(self.B))) / max(1, len(self.B))))
        self.setup_s = time.time() - t0

    def step(self):
        self.ctx.reset_state()
        self.ctx.solve()


class MultiBatch:
    """The same batch cut into `parts` blocks, each on a context and a HIP stream of its own: the kernels of the blocks run side
    by side, so that the low-occupancy launches of one block (iterations in which few of its windows linearise and factor)
    overlap with the other blocks' work."""

    def __init__(self, torch, v, dev, ids, cfg, opt, config_id, parts, new_ctx_on):
        ids = list(ids)
        self.parts = []
        n = len(ids)
        for p in range(parts):
            lo, hi = n * p // parts, n * (p + 1) // parts
            st = torch.cuda.Stream(dev)
            ctx = new_ctx_on(hi - lo, st)
            self.parts.append((Batch(v, ctx, ids[lo:hi], cfg, opt, config_id), st))

    def step(self):
        for b, _ in self.parts:
            b.step()

    def close(self):
        for b, _ in self.parts:
            b.ctx.close()


def timed(torch, dist, dev, batch, steps, warmup, v):
    for _ in range(warmup):
        batch.step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    return v.shard.reduce_max_vec(dist, [elapsed], dev)[0]


def launch_profile(torch, dev, batch, reps):
    """Per-launch device time (HIP events on the launch stream) and active-window counts, averaged over `reps` solves."""
    ctx = batch.ctx
    ctx.enable_kernel_timing(True)
    acc = None
    for _ in range(reps):
        batch.step()
        torch.cuda.synchronize(dev)
        prof = ctx.launch_profile()
        if acc is None:
            acc = [[name, 0.0, [0, 0, 0, 0]] for (name, _, _) in prof]
        for a, (name, ms, act) in zip(acc, prof):
            a[1] += ms / reps
            for k in range(4):
                a[2][k] += act[k] / reps
    kt = ctx.kernel_times()
    ctx.enable_kernel_timing(False)
    return acc, {k: ms / reps for k, (ms, _) in kt.items()}


def roofline_from_profile(prof, ab, nW, P, L):
    """Prices every launch of k_lin / the step / k_cost by the windows that DID WORK in it (device-side counters), not by
    the batch size.  The step is k_schur + k_chol + k_solve (general path, normally idle) + k_back: priced and timed as one
    unit `k_step` (sum of the four launches).  Returns the roofline object of the dominant kernel plus the per-kernel table."""
    per = {}
    units = []                   # (name, ms, bytes, full-work windows)
    i = 0
    while i < len(prof):
        name, ms, act = prof[i]
        if name == "k_lin":
            units.append(("k_lin", ms, ab["k_lin"] * act[0], act[0]))
        elif name == "k_schur":
            grp = prof[i:i + 4]
            assert [g[0] for g in grp] == list(STEP_KERNELS), [g[0] for g in grp]
            tms = sum(g[1] for g in grp)
            new = grp[0][2][1] + grp[2][2][1]                 # three-kernel path + general path
            reuse = grp[3][2][2] + grp[2][2][2]
            units.append(("k_step", tms, ab["k_solve"] * new + ab["k_solve_reuse"] * reuse, new))
            i += 3
        elif name == "k_cost":
            units.append(("k_cost", ms, ab["k_cost"] * act[3], act[3]))
        i += 1
    for name, ms, b, full in units:
        e = per.setdefault(name, dict(ms=0.0, bytes=0.0, launches=0, heavy=None))
        e["ms"] += ms
        e["bytes"] += b
        e["launches"] += 1
        if full >= 0.999 * nW and (e["heavy"] is None or ms > e["heavy"]["ms"]):
            e["heavy"] = dict(ms=ms, bytes=b, windows=full)          # slowest launch in which every window did the full work
    # the dominant KERNEL (one __global__ function): the step is three kernels priced as one unit, so it competes with its
    # longest member, not with its sum (VERDICT r3 reads the rocprof table the same way: k_lin2 30 % of GPU time, k_chol 19 %)
    single = {}
    for name, ms, act in prof:
        single[name] = single.get(name, 0.0) + ms
    longest = max((k for k in single if k in ("k_lin",) + STEP_KERNELS + ("k_cost",)), key=lambda k: single[k])
    dom = "k_step" if longest in STEP_KERNELS else longest
    table = {}
    for k, e in per.items():
        gbs = e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["ms"] > 0 else 0.0
        table[k] = {"ms_per_solve": e["ms"], "launches": e["launches"], "avg_launch_ms": e["ms"] / e["launches"],
                    "algorithmic_bytes_per_solve": e["bytes"], "GBps": gbs, "frac": gbs / HBM_PEAK_GBS}
        fl = pmc_flops(k, nW, P, L)
        if fl is not None:
            tf = fl * e["launches"] / (e["ms"] * 1e-3) / 1e12
            table[k]["fp64"] = {"flops_per_launch": fl, "TFLOPs": tf, "frac": tf / FP64_PEAK_TFLOPS}
        if e["heavy"]:
            h = e["heavy"]
            hg = h["bytes"] / (h["ms"] * 1e-3) / 1e9
            table[k]["heavy_launch"] = {"ms": h["ms"], "windows": h["windows"], "algorithmic_bytes": h["bytes"], "GBps": hg,
                                        "frac": hg / HBM_PEAK_GBS, "traffic": pmc_traffic(k, nW, P, L, "total_max")}
    d = table[dom]
    roof = {"bound": "hbm", "kernel": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
            "traffic": pmc_traffic(dom, nW, P, L),
            "traffic_source": "committed PMC profile (%s), not collected in this run" % (os.path.relpath(_pmc_file("pmc_traffic.json"), ROOT) if _pmc_file("pmc_traffic.json") else "none"),
            "avg_launch_ms": d["avg_launch_ms"],
            "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_solve"] / d["launches"],
            "pricing": "algorithmic bytes of the windows that did work in each launch (device counters) / launch time "
                       "(HIP events on the launch stream), summed over the %d launches of one solve; k_step = k_schur + k_chol + "
                       "k_solve (general path) + k_back" % d["launches"],
            "heavy_launch": d.get("heavy_launch"), "per_kernel": table,
            "dominant_by": "largest total time of a single kernel over the solve's launches: %s"
                           % ", ".join("%s %.3f ms" % (k, single[k]) for k in sorted(single, key=lambda q: -single[q])[:4])}
    it_ms = sum(e["ms"] for e in per.values())
    it_bytes = sum(e["bytes"] for e in per.values())
    roof["pipeline_GBps"] = it_bytes / (it_ms * 1e-3) / 1e9
    roof["pipeline_frac"] = roof["pipeline_GBps"] / HBM_PEAK_GBS
    # second roofline: FP64.  With ~10 MFLOP and ~0.5 MB per window-iteration the kernels sit right of the FP64 ridge
    # (78.6 TF / 8 TB/s = 9.8 FLOP/B): `binding` names the roof the dominant kernel is closer to.
    if "fp64" in d:
        roof["fp64"] = {"bound": "fp64", "achieved": d["fp64"]["TFLOPs"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": d["fp64"]["frac"], "flops_per_launch": d["fp64"]["flops_per_launch"],
                        "source": "SQ FP64 instruction counters of the committed PMC profile (%s)" % os.path.relpath(_pmc_file("pmc_flops.json"), ROOT)}
        roof["binding"] = "fp64" if roof["fp64"]["frac"] > roof["frac"] else "hbm"
    else:
        roof["fp64"] = None
        roof["binding"] = "unknown (no FP64 counter profile for this workload)"
    return roof


def step_stats(reports, n):
    import numpy as np
    it = np.array([reports[i].iterations for i in range(n)])
    # accepted steps only: the device counter starts at 0 (csrc/ba_lin.h k_prep) and counts accepted steps, as the oracle's
    # report does (oracle/window.cpp) -- iteration 0 is not in it
    acc = np.array([reports[i].num_successful_steps for i in range(n)])
    term = np.array([reports[i].termination for i in range(n)])
    return {"mean_tr_iterations": float(it.mean()), "mean_accepted_steps": float(acc.mean()),
            "mean_rejected_steps": float((it - acc).mean()),
            "accepted_histogram": [int((acc == k).sum()) for k in range(int(it.max()) + 1)],
            "terminated_by_tolerance": int((term == 1).sum()), "failed": int((term == 2).sum())}


def frontend_config4(torch, v, dev, steps=5, warm=2, n=64, imgs=None):
    """BASELINE config 4 as an extra key: EDLines + KLT matching of the 64-frame 752x480 stream, frames resident in HBM.
    n = 256 (the stream four times over) is the second operating point: the routing stage runs one wave per frame, so a
    batch of 64 occupies 64 of the 1024 SIMDs and four batches in flight cost the same wall clock."""
    import numpy as np
    if imgs is None:
        imgs = v.workload.frame_stream(64)
    if n > len(imgs):
        imgs = np.concatenate([imgs] * (n // len(imgs)))
    fe = v.frontend.FrontendContext(device=dev.index or 0, max_images=n, width=752, height=480, max_lines=256,
                                    stream=torch.cuda.current_stream(dev).cuda_stream)
    fe.match_reserve(n - 1, 8192)
    fe.upload(imgs)
    pairs = [(i, i + 1) for i in range(n - 1)]

    def step():
        # detector -> matcher on the device (vpl_match_from_detected); lines and matches come back at the end of the batch
        fe.detect()
        fe.match_from_detected(pairs, 256)
        fe.match_run()
        fe.synchronize()
        lines = [l[:256] for l in fe.download()]
        return lines, fe.match_download()

    for _ in range(warm):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        lines, (r2c, ok) = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); fe.detect(); e1.record(); fe.match_run(); e2.record()
    torch.cuda.synchronize(dev)
    fe.enable_kernel_timing(True)      # per-kernel device time of one more batch (hipEvents on the launch stream)
    fe.detect(); fe.match_run()
    fe.synchronize()
    kt = fe.kernel_times()
    fe.enable_kernel_timing(False)
    out = {"metric": "line front-end frames/s (EDLines + KLT matching, 752x480, batch 64)", "value": n / dt,
           "unit": "frames/s", "ms_per_batch": 1e3 * dt, "device_ms_detect": e0.elapsed_time(e1),
           "device_ms_match": e1.elapsed_time(e2), "mean_lines_per_frame": float(np.mean([len(l) for l in lines])),
           "mean_matches_per_pair": float(np.mean([(r >= 0).sum() for r in r2c])),
           "kernels_ms_per_batch": {k: round(x, 4) for k, x in kt.items()}}
    if n != 64:
        out["metric"] = "line front-end frames/s (EDLines + KLT matching, 752x480, %d frames in flight)" % n
    if "k_ed_grad" in kt:
        ms = kt["k_ed_grad"]
        out["k_ed_grad"] = {"ms": ms, "algorithmic_bytes": n * 752 * 480 * 8, "GBps": n * 752 * 480 * 8 / (ms * 1e-3) / 1e9,
                            "frac": n * 752 * 480 * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    fe.close()
    return out


def frontend_cpu_baseline(imgs, cores, budget_s=8.0):
    """CPU baseline of config 4: the oracle's EDLines + line matching (CPU restatement of the reference path) on a bounded
    sample of the same frame stream, fanned over the host threads this process may use (ctypes releases the GIL)."""
    import numpy as np
    import oracle_api as o       # checker / CPU baseline only
    from concurrent.futures import ThreadPoolExecutor
    o.load()
    n = min(len(imgs), max(8, 2 * cores))
    sample = imgs[:n]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        lines = list(ex.map(lambda im: o.edlines(im)[:256], sample))
        list(ex.map(lambda i: o.line_match(sample[i], sample[i + 1], lines[i], lines[i + 1])[0], range(n - 1)))
    one = time.perf_counter() - t0
    reps = max(1, min(50, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        with ThreadPoolExecutor(cores) as ex:
            lines = list(ex.map(lambda im: o.edlines(im)[:256], sample))
            list(ex.map(lambda i: o.line_match(sample[i], sample[i + 1], lines[i], lines[i + 1])[0], range(n - 1)))
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": o.cpu_model(),
            "sample": "first %d frames of the stream (detect + %d consecutive-pair matches) x %d repeats over %d host threads, %.1f s"
                      % (n, n - 1, reps, cores, dt)}


def c_abi_latency():
    """tests/native/latency_check.cpp: wall-clock latency of upload / solve / download through the C ABI from a C++ caller
    (1, 8, 64 windows).  None when the program cannot be built or run here."""
    import tempfile
    try:
        exe = os.path.join(tempfile.mkdtemp(prefix="vpl_lat_"), "latency_check")
        libdir = os.path.join(ROOT, "vplines-slam_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "native", "latency_check.cpp"),
                               "-L", libdir, "-lvplines_hip", "-Wl,-rpath," + libdir, "-o", exe], stderr=subprocess.DEVNULL)
        return json.loads(subprocess.check_output([exe], text=True, timeout=300))
    except Exception as e:      # reported, not hidden
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=512, help="the batch: windows in total (strong) / per GPU (weak)")
    ap.add_argument("--points", type=int, default=200)
    ap.add_argument("--lines", type=int, default=80)
    ap.add_argument("--cpu-windows", type=int, default=64, help="oracle sample size for parity + CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-2 / config-4 / weak-scaling extra keys")
    ap.add_argument("--streams", type=int, default=1, help="experiment: cut the rank's block into this many blocks on streams of their own")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torchrun --nproc-per-node %d) "
                         "or run without a launcher" % (args.gpus, world, args.gpus))

    import numpy as np
    import torch
    import vplines_slam_amd as v

    dist = None
    if world > 1 or os.environ.get("VPL_FORCE_DIST"):   # VPL_FORCE_DIST: exercise the RCCL path with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    total, P, L = args.windows, args.points, args.lines
    opt = v.default_options()
    cfg = v.workload.config(P, L, True)
    TL = cfg.track_len
    config_id = 3 if L else 2

    # a stream of its own (not the legacy default stream): the library replays the solve as a hipGraph captured on it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    def new_ctx(nw, p, l):
        c = v.Context(device=local_rank, max_windows=max(nw, 1), max_points=max(p, 1), max_point_obs=max(p * TL, 1),
                      max_lines=max(l, 1), max_line_obs=max(l * TL, 1))
        c.set_stream(stream.cuda_stream)
        return c

    # ---- strong scaling: the ONE batch of `total` windows, block-partitioned ----
    lo, hi = v.shard.split_batch(total, rank, world)
    ctx = new_ctx(hi - lo, P, L)
    strong = Batch(v, ctx, range(lo, hi), cfg, opt, config_id)
    elapsed = timed(torch, dist, dev, strong, args.steps, args.warmup, v)
    value = total * args.steps / elapsed
    multi = None
    if args.streams > 1:
        def new_ctx_on(nw, st):
            c = v.Context(device=local_rank, max_windows=max(nw, 1), max_points=max(P, 1), max_point_obs=max(P * TL, 1),
                          max_lines=max(L, 1), max_line_obs=max(L * TL, 1))
            c.set_stream(st.cuda_stream)
            return c
        mb = MultiBatch(torch, v, dev, range(lo, hi), cfg, opt, config_id, args.streams, new_ctx_on)
        em = timed(torch, dist, dev, mb, args.steps, args.warmup, v)
        multi = {"streams": args.streams, "value": total * args.steps / em, "ms_per_step": 1e3 * em / args.steps}
        mb.close()

    # results: one all-gather of the per-window states, parity of the gathered set against the oracle on rank 0
    _, reports = ctx.download()
    n_local = hi - lo
    # the block's states are packed on the device and gathered device to device (RCCL); only the sampled rows come to the host
    gathered_dev = v.shard.gather_states_device(dist, ctx, n_local, total, dev)
    gathered = gathered_dev.cpu().numpy()
    host_pack = v.shard.pack_states(strong.B)
    assert np.array_equal(gathered[lo:hi], host_pack), "device-packed states differ from the downloaded ones"
    stats = step_stats(reports, n_local)

    # ---- the same batch with the host round trip (upload of the caller's windows + solve + download): never `value` ----
    pcie = None
    if rank == 0 and not args.no_extras:
        hostw = [[b.copy() for b in strong.pristine] for _ in range(4)]    # fresh inputs per repetition (results come back in place)
        ctx.solve_windows(hostw[0], opt)
        tp = time.perf_counter()
        for r in range(1, 4):
            ctx.solve_windows(hostw[r], opt)
        pcie = n_local * 3 / (time.perf_counter() - tp)
        strong.B = [b.copy() for b in strong.pristine]     # back to the resident batch of the timed region
        ctx.upload(strong.B, opt)

    # ---- per-launch profile (separate, un-timed pass) ----
    prof, kms = launch_profile(torch, dev, strong, max(3, min(args.steps, 10)))
    ab = algorithmic_bytes(P, L, strong.n_prior, P * TL, L * TL)
    roof = roofline_from_profile(prof, ab, n_local, P, L)

    out = None
    if rank == 0:
        out = {
            "metric": "sliding-window BA solves/sec (10 KF, 200 pts + 80 lines, 5 iters)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "one batch of %d independent synthetic sliding windows (BASELINE config %d shape) block-"
                                   "partitioned over %d GPU(s): 11 frames at 10 Hz, %d points + %d lines (x%d obs) + VP obs, "
                                   "10 IMU factors (20 samples each), prior n=%d from a warm-up solve; max %d TR iterations + "
                                   "gauge fix + MARGIN_OLD marginalisation; generator as DESIGN.md 2b (drift-style initial error, "
                                   "triad about the direction of travel, d_z < 0 line directions: deviations from SURVEY 8d)"
                                   % (total, config_id if world == 1 else 5, world, P, L, TL, strong.n_prior, opt.num_iterations),
                       "windows_total": total, "windows_per_gpu": n_local, "points": P, "lines": L, "track_len": TL,
                       "prior_dim": strong.n_prior, "parallelism": "batch block-partitioned x%d, all-gather of results" % world,
                       **stats},
            "roofline": roof,
            "linearisations_per_solve": sum(act[0] for n_, _, act in prof if n_ == "k_lin") / max(1, n_local),
            "new_steps_per_solve": sum(act[1] for n_, _, act in prof if n_ in ("k_schur", "k_solve")) / max(1, n_local),
            "work_note": "'5 iters' = max_num_iterations; a window linearises and factors only after an accepted step "
                         "(linearisations_per_solve), a rejected iteration re-uses the Gauss-Newton step (DESIGN.md 2b)",
            "kernels_ms_per_step": kms,
            "device_ms_per_step": sum(kms.values()),
            "launches": [{"kernel": n, "ms": round(ms, 5), "active": [round(a, 1) for a in act]} for n, ms, act in prof],
            "setup_s": strong.setup_s,
            "multi_stream": multi,
            "host_round_trip": None if pcie is None else {
                "value": pcie, "unit": "solves/s",
                "what": "vpl_ba_solve_windows on host windows: pack + PCIe upload + solve + download + unpack, single host thread"},
        }

    # ---- parity of the gathered result set + CPU baseline on a bounded sample (rank 0) ----
    if rank == 0 and not args.no_cpu_baseline:
        import oracle_api as o       # the oracle is the CPU baseline / checker here, never the product path
        from test_gpu_solve import rot_angle
        ns = min(args.cpu_windows, total)
        sample_ids = [int(round(k * (total - 1) / max(1, ns - 1))) for k in range(ns)] if ns > 1 else [0]
        sample_ids = sorted(set(sample_ids))
        # the sampled windows span every rank's block; rank 0 rebuilds their inputs from the window ids alone
        if world == 1:
            sample = [strong.pristine[g].copy() for g in sample_ids]
        else:
            sctx = new_ctx(len(sample_ids), P, L)
            sb, _keep = v.workload.primed_batch(sctx, sample_ids, cfg, opt, config_id)
            sample = [b.copy() for b in sb]
            sctx.close()
        cores = cpu_share()
        native = o.build_native()
        if native:
            o.load(native)
        lib = o.load()
        # (i) the reference's own threading: one solve thread + NUM_THREADS = 4 for the marginalisation assembly
        lib.orc_set_marg_threads(4)
        s1 = [w.copy() for w in sample[: max(2, min(len(sample), 8))]]
        t1 = time.perf_counter()
        o.solve_windows(s1, opt, threads=1)
        t1 = time.perf_counter() - t1
        lib.orc_set_marg_threads(1)
        # (ii) windows fanned over all host cores: the sample repeated until every core has work, one pass timed to size
        #      the run to ~10-15 s of wall clock
        work = [w for _ in range(max(1, -(-2 * cores // len(sample)))) for w in sample]
        res = [w.copy() for w in work]
        t = time.perf_counter()
        o.solve_windows(res, opt, threads=cores)
        tp = time.perf_counter() - t
        reps = max(1, min(400, int(12.0 / max(tp, 1e-3))))
        tc = 0.0
        for _ in range(reps):
            res = [w.copy() for w in work]
            t = time.perf_counter()
            o.solve_windows(res, opt, threads=cores)
            tc += time.perf_counter() - t
        dpm = drm = 0.0
        for k, g in enumerate(sample_ids):
            pose, _, _ = v.shard.unpack_state(gathered[g])
            dp = np.linalg.norm(pose[:, :3] - res[k].pose[:, :3], axis=1).max()
            dr = max(rot_angle(pose[i, 3:], res[k].pose[i, 3:]) for i in range(11))
            dpm, drm = max(dpm, dp), max(drm, dr)
        out["cpu_baseline"] = {
            "value": len(work) * reps / tc, "unit": "solves/s", "cores": cores, "kind": "port",
            "cpu_model": o.cpu_model(), "nproc": os.cpu_count(),
            "build": "-O3 -march=native (built on this host)" if native else "-O2 portable build (native compile failed)",
            "sample": "%d of the %d timed windows (x%d so that every core has work) x %d repeats, oracle (CPU restatement of "
                      "the reference path) fanned over %d host threads, %.1f s" % (len(sample), total, len(work) // len(sample), reps, cores, tc),
            "reference_threading_solves_per_s": len(s1) / t1,
            "reference_threading": "1 solve thread + 4 marginalisation threads (marginalization_factor.h:13), %d windows" % len(s1)}
        out["parity"] = {"windows": len(sample), "of_gathered": total, "max_dp_m": dpm, "max_dr_rad": drm,
                         "bar": "1e-4 m / 1e-6 rad"}
    # all-reduce(MAX) of the parity summary: every rank ends up with the figures rank 0 prints
    par = v.shard.reduce_max_vec(dist, [out["parity"]["max_dp_m"], out["parity"]["max_dr_rad"]] if (out and "parity" in out) else [0.0, 0.0], dev)
    if rank == 0 and out and "parity" in out:
        out["parity"]["allreduce_max"] = par

    # ---- extra keys: weak scaling, config 2 (points only), config 4 (line front-end) ----
    if not args.no_extras:
        if world > 1:
            ctx.close()
            wl, wh = v.shard.window_range(rank, world, total)
            wctx = new_ctx(total, P, L)
            weak = Batch(v, wctx, range(wl, wh), cfg, opt, config_id)
            we = timed(torch, dist, dev, weak, args.steps, args.warmup, v)
            if rank == 0:
                out["weak_scaling"] = {"value": world * total * args.steps / we, "unit": "solves/s",
                                       "windows_per_gpu": total, "ms_per_step": 1e3 * we / args.steps}
            wctx.close()
        elif rank == 0:
            out["weak_scaling"] = {"value": value, "unit": "solves/s", "windows_per_gpu": total,
                                   "ms_per_step": 1e3 * elapsed / args.steps}
        if rank == 0 and world == 1:
            # ---- steady state: the same windows with the prior of a 7-window chain (n = 75, the reference's size once every
            # frame of the window is tied to the one that leaves) instead of the 45 dims of one warm-up window ----
            try:
                pobs = v.workload.steady_point_obs(cfg)
                cs = v.Context(device=local_rank, max_windows=total, max_points=max(P, 1), max_point_obs=max(pobs, 1),
                               max_lines=max(L, 1), max_line_obs=max(L * TL, 1))
                cs.set_stream(stream.cuda_stream)
                sb = Batch(v, cs, range(total), cfg, opt, config_id, steady_chain=7)
                es = timed(torch, None, dev, sb, args.steps, args.warmup, v)
                _, rs = cs.download()
                ps, ks = launch_profile(torch, dev, sb, 3)
                abs_ = algorithmic_bytes(P, L, sb.n_prior, pobs, L * TL)
                rfs = roofline_from_profile(ps, abs_, total, P, L)
                out["steady_state"] = {
                    "what": "the %d windows of the headline (same seeds and times) with a tenth of their point tracks living "
                            "through all 11 frames, behind a chain of 7 such windows one keyframe apart whose priors are handed "
                            "over on the device (vpl_ba_upload_chained): prior in and prior out have the reference's steady-state "
                            "size (10 poses + speed/bias 1 + extrinsic); with 6-frame tracks only the prior stays at 45 dims "
                            "however long the chain" % total,
                    "value": total * args.steps / es, "unit": "solves/s", "ms_per_step": 1e3 * es / args.steps,
                    "ratio_to_headline": (total * args.steps / es) / value, "prior_dim": sb.n_prior,
                    "prior_dim_out": int(round(sum(rs[i].prior_n for i in range(total)) / total)), "chain": 7,
                    "point_observations": pobs, "kernels_ms_per_step": ks, "roofline_kernel": rfs["kernel"],
                    "roofline_frac": rfs["frac"], "heavy_launch": rfs["heavy_launch"], "setup_s": sb.setup_s,
                    **step_stats(rs, total)}
                cs.close()
            except Exception as e:   # reported, not hidden
                out["steady_state"] = {"error": repr(e)}
        if rank == 0 and world == 1 and L > 0:
            ctx.close()
            c2 = new_ctx(total, P, 0)
            cfg2 = v.workload.config(P, 0, True)
            b2 = Batch(v, c2, range(total), cfg2, opt, 2)
            st = max(3, args.steps // 2)
            e2 = timed(torch, None, dev, b2, st, 2, v)
            _, r2 = c2.download()
            p2, k2 = launch_profile(torch, dev, b2, 3)
            ab2 = algorithmic_bytes(P, 0, b2.n_prior, P * TL, 0)
            rf2 = roofline_from_profile(p2, ab2, total, P, 0)
            out["config2"] = {"metric": "sliding-window BA solves/sec (10 KF, 200 pts, no lines, 5 iters)",
                              "value": total * st / e2, "unit": "solves/s", "ms_per_step": 1e3 * e2 / st,
                              "roofline_kernel": rf2["kernel"], "roofline_frac": rf2["frac"],
                              "heavy_launch": rf2["heavy_launch"], **step_stats(r2, total)}
            c2.close()
            try:
                frames = v.workload.frame_stream(64)
                out["config4"] = frontend_config4(torch, v, dev, imgs=frames)
                b256 = frontend_config4(torch, v, dev, steps=3, warm=1, n=256, imgs=frames)
                out["config4"]["frames_in_flight_256"] = {k: b256[k] for k in ("value", "unit", "ms_per_batch", "device_ms_detect",
                                                                               "device_ms_match", "k_ed_grad")}
                if not args.no_cpu_baseline:
                    out["config4"]["cpu_baseline"] = frontend_cpu_baseline(frames, cpu_share())
            except Exception as e:   # the headline line must survive a front-end failure; it is reported, not hidden
                out["config4"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extras:
        lat = c_abi_latency()
        out["c_abi_latency"] = lat
        out["latency_1w_ms"] = lat.get("nW1", {}).get("total_ms") if isinstance(lat, dict) else None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
