"""Randomised CALL SEQUENCES on one long-lived context (GPU box): whatever was called before -- batches of other sizes and
shapes, enqueued calls still in flight, refused calls, a solve left without its download -- every call must give, BIT FOR
BIT, what the same call gives on a context created for it alone.  (What the calls compute is held to the oracle by
fuzz_parity.py / fuzz_map.py; this sweep is about the state a context carries from call to call: staging buffers sized by an
earlier batch, the device-resident priors, the packed layouts, the pending-call slot.)

Calls drawn: solve_windows; upload / solve / download, reset_state and the solve again (same bits); a solve followed by a
chained upload one keyframe on; marginalize, triangulate_lines, triangulate_points, only_line_opt, slide_window (synchronous,
or enqueued and collected at once, or enqueued and left for the next call to settle); solve_odometry; an upload that is
refused half-way (then solve and download must refuse too); a window with a NaN among its inputs; a solve whose priors
are handed by the caller to the windows one keyframe on, marginalised and solved; the single-factor evaluators and manifold
operations (which leave a call in flight as it is).  The kernel / leg timers are switched on and off and the
context is moved between four streams along the way.

    python tools/fuzz_sequence.py [calls=120] [windows per context=6] [seed=1]
Exit status 1 on the first difference.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C

import numpy as np

import oracle_api as o
import vplines_slam_amd as v
from fuzz_parity import draw_window, make_ctx
from vplines_slam_amd.capi import Prior

CALLS = ["solve", "solve", "twice", "chain", "marg", "tri_lines", "tri_points", "line_opt", "slide", "odometry", "refused", "nan", "prior", "helpers"]


def state(w):
    return tuple(a.tobytes() for a in (w.pose, w.speed_bias, w.ex_pose, w.inv_depth, w.line_plk, w.line_removed, w.line_triangulated))


def rep(r):
    return (r.iterations, r.num_successful_steps, r.termination, r.initial_cost, r.final_cost, r.n_lines_removed, r.prior_m, r.prior_n)


def solved(ws, priors, reports):
    return [state(w) for w in ws] + [bytes(p) for p in priors] + [rep(r) for r in reports]


def where(a, b):
    """which of the compared items differ (index and, for byte strings, the first differing double)"""
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        if x != y:
            if isinstance(x, tuple) and x and isinstance(x[0], bytes):
                out.append("item %d: fields %s" % (k, [j for j, (p, q) in enumerate(zip(x, y)) if p != q]))
            elif isinstance(x, bytes) and len(x) == len(y) == C.sizeof(Prior):
                pa, pb = Prior.from_buffer_copy(x), Prior.from_buffer_copy(y)
                if pa.n != pb.n:
                    out.append("item %d: prior of %d / %d rows" % (k, pa.n, pb.n))
                else:
                    Ja, Jb = pa.J(), pb.J()
                    d = np.abs(Ja - Jb)
                    print("   ranks (rows of J0 that are not zero): %d / %d; first pivots (column of the first nonzero of each row): %s / %s" % (
                        int((np.abs(Ja).max(1) > 0).sum()), int((np.abs(Jb).max(1) > 0).sum()),
                        [int(np.nonzero(r)[0][0]) if np.any(r) else -1 for r in Ja], [int(np.nonzero(r)[0][0]) if np.any(r) else -1 for r in Jb]))
                    print("   diag J^T J long-lived", np.array2string(np.diag(Ja.T @ Ja), precision=4))
                    print("   diag J^T J fresh     ", np.array2string(np.diag(Jb.T @ Jb), precision=4))
                    out.append("item %d: prior n=%d, J0 differs in %d entries (rows %s), max |dJ| %.3e of max |J| %.3e, max |dr| %.3e; J^T J rel diff %.2e" % (
                        k, pa.n, int((d > 0).sum()), sorted(set(np.nonzero(d > 0)[0].tolist()))[:12], d.max(), np.abs(Jb).max(), np.abs(pa.r() - pb.r()).max(),
                        np.abs(Ja.T @ Ja - Jb.T @ Jb).max() / max(1e-300, np.abs(Jb.T @ Jb).max())))
            elif isinstance(x, bytes) and len(x) == len(y) and len(x) % 4 == 0:
                xa, ya = np.frombuffer(x, np.int32), np.frombuffer(y, np.int32)
                d = np.nonzero(xa != ya)[0]
                out.append("item %d: %d of %d words differ, first at byte %d" % (k, len(d), len(xa), 4 * int(d[0])))
            else:
                out.append("item %d: %s vs %s" % (k, x, y))
    return "; ".join(out[:6]) + (" (lengths %d / %d)" % (len(a), len(b)) if len(a) != len(b) else "")


def run_call(ctx, kind, ws, w2, opt, prm, pend):
    """one call on `ctx`; -> a function that returns the call's results as comparable bytes (evaluated by the caller AFTER the
    next call when `pend` asks to leave an enqueued call in flight)"""
    ws = [w.copy() for w in ws]
    a = pend != "sync"
    def done(f):
        if pend == "collect":
            ctx.collect()
        return f
    if kind in ("solve", "nan"):
        p, r = ctx.solve_windows(ws, opt)
        return lambda: solved(ws, p, r)
    if kind == "twice":
        ctx.upload(ws, opt); ctx.solve(); ctx.synchronize()
        first = [w.copy() for w in ws]
        p, r = ctx.download()
        s1 = solved(ws, p, r)
        for w, f in zip(ws, first):                      # download wrote the results into the Windows: back to the inputs
            w.pose[:], w.speed_bias[:], w.ex_pose[:] = f.pose, f.speed_bias, f.ex_pose
        ctx.reset_state(); ctx.solve(); ctx.synchronize()
        p, r = ctx.download()
        s2 = solved(ws, p, r)
        if s1 != s2:
            raise AssertionError("reset_state + solve does not repeat the solve")
        return lambda: s1
    if kind == "chain":
        p, r = ctx.solve_windows(ws, opt)
        s1 = solved(ws, p, r)
        w2 = [w.copy() for w in w2]
        ctx.upload(w2, opt, chained=True); ctx.solve(); ctx.synchronize()
        p2, r2 = ctx.download()
        return lambda: s1 + solved(w2, p2, r2)
    if kind == "helpers":
        # the single-factor evaluators and the two manifold operations: they share the context's stream with a call that is
        # still in flight, and must neither complete it nor disturb it
        k = prm["helpers"]
        out = [ctx.pose_plus(k["x"], k["dx"]).tobytes(), ctx.line_orth_plus(k["orth"], k["dorth"]).tobytes()]
        for r, j in (ctx.projection_factor(k["proj"], k["pts"]), ctx.line_factor(k["linep"], k["lobs"]), ctx.vp_factor(k["linep"], k["vp"])):
            out += [r.tobytes(), j.tobytes()]
        return lambda: out
    if kind == "prior":
        # the host hand-over: the priors of one solve given to the windows one keyframe on, which are marginalised as they
        # stand and then solved (the calls after this one come without a prior again)
        p, r = ctx.solve_windows(ws, opt)
        w2 = [w.copy() for w in w2]
        for w, q in zip(w2, p):
            w.prior = q
        pm, m, n = ctx.marginalize(w2, opt, prm["flag"])
        p2, r2 = ctx.solve_windows(w2, opt)
        return lambda: solved(ws, p, r) + [bytes(x) for x in pm] + [m.tobytes(), n.tobytes()] + solved(w2, p2, r2)
    if kind == "marg":
        p, m, n = ctx.marginalize(ws, opt, prm["flag"], async_=a)
        return done(lambda: [bytes(x) for x in p] + [m.tobytes(), n.tobytes()])
    if kind == "tri_lines":
        for w in ws:
            nl = len(w.line_start)
            if nl:
                w.line_triangulated[:nl] = prm["untri"][:nl]
                w.line_plk[prm["untri"][:nl] == 0] = 0
        ctx.triangulate_lines(ws, async_=a)
        return done(lambda: [state(w) for w in ws])
    if kind == "tri_points":
        for w in ws:
            if len(w.inv_depth):
                w.inv_depth[prm["unset"][:len(w.inv_depth)]] = -1.0
        ctx.triangulate_points(ws, 5.0, async_=a)
        return done(lambda: [state(w) for w in ws])
    if kind == "line_opt":
        reps = ctx.only_line_opt(ws, opt, async_=a)
        return done(lambda: [state(w) for w in ws] + [rep(r) for r in reps])
    if kind == "slide":
        st = ctx.slide_window(ws, prm["flag"], 5.0, async_=a)
        return done(lambda: [state(w) for w in ws] + [tuple(x.tobytes() for x in (s.point_start, s.point_nobs, s.point_drop, s.line_start,
                                                                                   s.line_nobs, s.line_drop)) for s in st])
    if kind == "odometry":
        p, lr, r = ctx.solve_odometry(ws, opt, 5.0)
        return lambda: solved(ws, p, r) + [rep(x) for x in lr]
    if kind == "refused":
        have = [w for w in ws if len(w.point_nobs)]
        if not have:
            ctx.synchronize()                            # (completes a call in flight, as every other kind does)
            return lambda: ["no point track to spoil"]
        bad = have[-1]
        bad.point_nobs[len(bad.point_nobs) // 2] = 1     # a track the reference's filter would not have passed: CONTRACT
        said = []
        for what, f in (("upload", lambda: ctx.upload(ws, opt)), ("solve", ctx.solve), ("download", ctx.download)):
            try:
                f()
                said.append(what + " accepted")
            except RuntimeError as e:
                said.append(what + " refused: " + str(e).split(":")[0])
        if any("accepted" in s for s in said):
            raise AssertionError("after a refused upload: %s" % said)
        return lambda: said
    raise ValueError(kind)


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    import torch                                         # (initialised before the library's first HIP call)
    keep = [torch.cuda.Stream(device=0) for _ in range(3)]
    streams = [0] + [k.cuda_stream for k in keep]
    ctx = make_ctx(per)
    # replay aids: FUZZ_SEQ_ONLY="30,31,32" runs these calls alone, FUZZ_SEQ_PEND="31:sync" overrides how call 31 is issued
    only = set(int(x) for x in os.environ.get("FUZZ_SEQ_ONLY", "").split(",") if x)
    force = dict((int(x.split(":")[0]), x.split(":")[1]) for x in os.environ.get("FUZZ_SEQ_PEND", "").split(",") if x)
    waiting = None                                       # (tag, results of the long-lived context, results of the fresh one)
    count = {}
    timing = False
    for c in range(calls):
        kind = CALLS[int(rng.integers(0, len(CALLS)))]
        n = int(rng.integers(1, per + 1))
        opt = v.default_options()
        opt.num_iterations = int(rng.choice([1, 2, 5]))
        opt.estimate_extrinsic = int(rng.integers(0, 2))
        opt.marginalization_flag = int(rng.choice([v.MARGIN_OLD, v.MARGIN_SECOND_NEW] + ([] if kind in ("chain", "prior") else [v.MARGIN_NONE])))
        opt.remove_line_outliers = int(rng.integers(0, 2))
        ws, w2, shapes = [], [], []
        for i in range(n):
            w, sh = draw_window(rng, 1000 * c + i, 0.41 * (c * per + i))
            ws.append(w); shapes.append(sh[:3])
            w2.append(draw_window(np.random.default_rng(9000 + 100 * c + i), 1000 * c + 500 + i, 0.41 * (c * per + i) + 0.1)[0])
        o.preintegrate_windows(ws + w2, opt)
        if kind == "nan":
            k = int(rng.integers(0, n))
            ws[k].pose[int(rng.integers(0, 11)), int(rng.integers(0, 3))] = np.nan
        prm = dict(flag=int(rng.choice([v.MARGIN_OLD, v.MARGIN_SECOND_NEW])), untri=(rng.random(512) < 0.4).astype(np.int32),
                   unset=rng.random(512) < 0.5)
        if kind == "helpers":
            m = 37
            q = rng.normal(size=(m, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
            pose = lambda: np.concatenate([rng.normal(0, 1, (m, 3)), q[rng.permutation(m)]], 1)
            prm["helpers"] = dict(x=pose(), dx=rng.normal(0, 0.1, (m, 6)), orth=rng.normal(0, 1, (m, 4)), dorth=rng.normal(0, 0.1, (m, 4)),
                                  proj=np.concatenate([pose(), pose(), pose(), rng.uniform(0.1, 1, (m, 1))], 1),
                                  pts=np.concatenate([rng.normal(0, 0.3, (m, 2)), np.ones((m, 1)), rng.normal(0, 0.3, (m, 2)), np.ones((m, 1))], 1),
                                  linep=np.concatenate([pose(), pose(), rng.normal(0, 1, (m, 4))], 1), lobs=rng.normal(0, 0.3, (m, 4)),
                                  vp=rng.normal(0, 1, (m, 3)))
        enq = kind in ("marg", "tri_lines", "tri_points", "line_opt", "slide")
        pend = str(rng.choice(["sync", "collect", "leave"])) if enq else "sync"
        toggle = rng.random() < 0.15
        move = int(rng.integers(0, len(streams))) if rng.random() < 0.12 else -1
        if only and c not in only:                       # (replay of a subset: the draws above keep the inputs the same)
            continue
        pend = force.get(c, pend)
        tag = "call %d %s%s n=%d %s it%d ex%d flag%d" % (c, kind, {"sync": "", "collect": " (enqueued, collected)", "leave": " (enqueued, left in flight)"}[pend],
                                                      n, shapes, opt.num_iterations, opt.estimate_extrinsic, opt.marginalization_flag)
        if toggle and waiting is None:                   # the instrumentation switched on or off: same results
            timing = not timing
            ctx.enable_kernel_timing(timing)
            ctx.lib.vpl_ctx_enable_leg_timing(ctx.h, 1 if timing else 0)
            if not timing:
                ctx.kernel_times()
        if move >= 0:                                    # another stream, whatever is in flight (vpl_ctx_set_stream completes it)
            ctx.set_stream(streams[move])
        try:
            mine = run_call(ctx, kind, ws, w2, opt, prm, pend)
            fresh_ctx = make_ctx(per)
            theirs = run_call(fresh_ctx, kind, ws, w2, opt, prm, "sync")()
            if os.environ.get("FUZZ_SEQ_DUMP", "") == str(c):    # replay aid: the kept block (A, b) of window 0 before it is factored
                got = []
                for x in (ctx, fresh_ctx):
                    A, bb = np.zeros(171 * 171), np.zeros(171)
                    nn = x.lib.vpl_ba_debug_marg_Ab(x.h, 0, A.ctypes.data_as(C.POINTER(C.c_double)), bb.ctypes.data_as(C.POINTER(C.c_double)))
                    got.append((nn, A[:nn * nn].reshape(nn, nn).copy(), bb[:nn].copy()))
                (n1, A1, b1), (n2, A2, b2) = got
                pc, rc = o.solve_window(ws[0].copy(), opt)
                Jc = pc.J()
                print("   oracle: prior n=%d, diag J^T J %s" % (pc.n, np.array2string(np.diag(Jc.T @ Jc), precision=4)))
                print("   A row 0", np.array2string(A2[0], precision=6))
                print("   A row 1", np.array2string(A2[1], precision=6))
                print("   oracle J^T J row 1", np.array2string((Jc.T @ Jc)[1], precision=6))
                print("   kept block: n %d / %d" % (n1, n2))
                if n1 == n2:
                    d = np.abs(A1 - A2)
                    print("   A differs in %d entries, max %.3e of %.3e, rows %s; b differs max %.3e" % (
                        int((d > 0).sum()), d.max(), np.abs(A2).max(), sorted(set(np.nonzero(d > 0)[0].tolist())), np.abs(b1 - b2).max()))
                    print("   diag long-lived", np.array2string(np.diag(A1), precision=3))
                    print("   diag fresh     ", np.array2string(np.diag(A2), precision=3))
            fresh_ctx.close()
        except (RuntimeError, AssertionError) as e:
            print(tag, "FAILED:", e)
            return 1
        if waiting is not None and kind != "helpers":    # the call before was left in flight: this one has settled it
            if waiting[1]() != waiting[2]:
                print(waiting[0], "DIFFERS from the same call on a fresh context (collected by the next call, %s):" % kind, where(waiting[1](), waiting[2]))
                return 1
            waiting = None
        if pend == "leave":
            waiting = (tag, mine, theirs)
        elif mine() != theirs:
            print(tag, "DIFFERS from the same call on a fresh context:", where(mine(), theirs))
            return 1
        count[kind] = count.get(kind, 0) + 1
        print(tag, "ok")
    ctx.synchronize()
    if waiting is not None and waiting[1]() != waiting[2]:
        print(waiting[0], "DIFFERS from the same call on a fresh context (collected by synchronize)")
        return 1
    ctx.close()
    print("fuzz_sequence: %d calls on one context %s: each identical, bit for bit, to the same call on a fresh context" % (calls, count))
    return 0


if __name__ == "__main__":
    sys.exit(main())
