"""Host-only checks of the point work-unit tables of k_lin (csrc/ba_pack.h) -- runs without a GPU.

Every unit's Hessian tile is committed in a ticket order the host assigns; a wave spins until its unit's ticket comes up.
The MARGIN_OLD pass runs only the first `rounds0` rounds, so it needs a ticket sequence of its own that numbers exactly the
units of those rounds: with the solve pass's numbers a round-0 unit can hold a ticket larger than that of a unit in a round
the pass never executes -- an endless spin on the device (ragged windows with most tracks starting in frame 0).
"""
import ctypes as C

import numpy as np
import pytest

import vplines_slam_amd as v

NF = 11
MAXR = 24


def _tables(start, nobs):
    lib = v.load_hip_library()
    n = len(start)
    lt = np.full((MAXR, 512, 2), -7, np.int32)
    st = np.full((MAXR, 32, 8, 2), -7, np.int32)
    r, r0 = C.c_int(), C.c_int()
    ip = C.POINTER(C.c_int)
    s = np.ascontiguousarray(start, np.int32)
    o = np.ascontiguousarray(nobs, np.int32)
    rc = lib.vpl_ba_debug_point_units(n, s.ctypes.data_as(ip), o.ctypes.data_as(ip), MAXR, lt.ctypes.data_as(ip),
                                      st.ctypes.data_as(ip), C.byref(r), C.byref(r0))
    assert rc == 0, rc
    return lt[:r.value], st[:r.value], r.value, r0.value


def _replay(st, nrounds, column):
    """Wave-by-wave replay of the two commit chains in Python (independent of the C++ replay): True = all waves finish."""
    for hf in range(2):
        items = []
        for wv in range(4):
            seq = []
            for rnd in range(nrounds):
                for qq in range(4):
                    for i in range(8):
                        d, t = st[rnd, (4 * hf + wv) * 4 + qq, i]
                        if d == 0:
                            break
                        seq.append((t >> 16) & 0xffff if column else t & 0xffff)
            items.append(seq)
        pos = [0] * 4
        tick = 0
        while any(pos[w] < len(items[w]) for w in range(4)):
            moved = False
            for w in range(4):
                if pos[w] < len(items[w]) and items[w][pos[w]] == tick:
                    tick += 1
                    pos[w] += 1
                    moved = True
            if not moved:
                return False
    return True


def _random_window(rng, P, frac0, lo=2, hi=NF):
    start = np.where(rng.random(P) < frac0, 0, rng.integers(0, NF - 1, P)).astype(np.int32)
    nobs = np.minimum(rng.integers(lo, hi + 1, P), NF - start).astype(np.int32)
    nobs = np.maximum(nobs, 2)
    start = np.minimum(start, NF - nobs).astype(np.int32)
    return start, nobs


def test_every_factor_has_one_lane_and_tickets_are_permutations():
    rng = np.random.default_rng(5)
    for trial in range(60):
        P = int(rng.integers(1, 200))
        start, nobs = _random_window(rng, P, rng.choice([0.1, 0.6, 0.9]))
        lt, st, R, R0 = _tables(start, nobs)
        assert 0 <= R0 <= R <= MAXR
        seen = set()
        for rnd in range(R):
            for lane in range(512):
                rec, off = lt[rnd, lane]
                if rec < 0:
                    continue
                p, k, s = rec & 0xffff, (rec >> 16) & 15, rec >> 20
                assert s == start[p] and 1 <= k < nobs[p]
                assert off == int(nobs[:p].sum())
                assert (p, k) not in seen
                seen.add((p, k))
                if s == 0:
                    assert rnd < R0, "a start-frame-0 factor sits in a round the marginalisation pass does not run"
        assert len(seen) == int((nobs - 1).sum())
        for hf in range(2):
            sq, sq0 = [], []
            for rnd in range(R):
                for slot in range(16 * hf, 16 * hf + 16):
                    for i in range(8):
                        d, t = st[rnd, slot, i]
                        if d == 0:
                            assert all(st[rnd, slot, j, 0] == 0 for j in range(i, 8))
                            break
                        sq.append(t & 0xffff)
                        if rnd < R0:
                            sq0.append((t >> 16) & 0xffff)
                        else:
                            assert (t >> 16) & 0xffff == 0xffff
            assert sorted(sq) == list(range(len(sq)))
            assert sorted(sq0) == list(range(len(sq0)))


def test_commit_chains_of_both_passes_finish_on_skewed_ragged_windows():
    """The shape ADVICE r2 flagged: ~100 points, 60 % of the tracks start in frame 0, lengths 2..11."""
    lib = v.load_hip_library()
    rng = np.random.default_rng(11)
    ip = C.POINTER(C.c_int)
    old_rule_hangs = 0
    multi = 0
    for trial in range(1500):
        start, nobs = _random_window(rng, 100, 0.6)
        lt, st, R, R0 = _tables(start, nobs)
        stc = np.ascontiguousarray(st)
        assert lib.vpl_ba_debug_point_chains(stc.ctypes.data_as(ip), R, R0, 0) == 1
        assert lib.vpl_ba_debug_point_chains(stc.ctypes.data_as(ip), R, R0, 1) == 1
        if trial < 300:   # the independent Python replay agrees with the library's
            assert _replay(st, R, 0)
            assert _replay(st, R0, 1)
        if R > R0 >= 1:
            multi += 1
            # what round 2 did: the marginalisation pass waiting on the solve pass's tickets
            if not _replay(st, R0, 0):
                old_rule_hangs += 1
    assert multi > 100, "the generator no longer produces windows whose marginalisation pass runs fewer rounds"
    assert old_rule_hangs > 0, "the test lost its teeth: the single-sequence rule no longer hangs on any window"


@pytest.mark.parametrize("P,frac0", [(200, 1 / 6.0), (200, 1.0), (16, 0.5), (1, 1.0)])
def test_uniform_and_degenerate_layouts(P, frac0):
    rng = np.random.default_rng(3)
    start, nobs = _random_window(rng, P, frac0, lo=6, hi=6)
    lt, st, R, R0 = _tables(start, nobs)
    assert _replay(st, R, 0) and _replay(st, R0, 1)


def test_layout_packers_under_the_address_sanitizer(tmp_path):
    """tests/native/pack_fuzz.cpp: 1 500 random track layouts through pack_point_units / point_unit_chains_finish /
    pack_schur_ksteps (csrc/ba_pack.h, pure C++), built with -fsanitize=address,undefined, every output table allocated at
    exactly the size the library gives it: no overrun, every factor packed exactly once, every commit chain finishes."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pack_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-D__host__=", "-D__device__=",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "vplines-slam_amd", "csrc"),
                           os.path.join(root, "tests", "native", "pack_fuzz.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, "1500", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "no table overrun" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
