"""NumPy prototype of k_chol's elimination order (csrc/ba_chol.h): speed/bias blocks 1..4 and 10..6 as two block chains
(row strips with the columns in lanes), then the dense system [vis 72 | sb_0 | sb_5 | rhs].  Checks the front layouts
(no fill outside the fronts) and the back-substitution against a dense solve of a matrix with the reduced system's
structure: IMU factors couple frames (f-1, f), the prior couples sb_0 with poses / extrinsic, visual terms are dense
on the 72 vis dims."""
import numpy as np

NF, NC = 11, 171
rng = np.random.default_rng(0)


def pose(f): return list(range(15 * f, 15 * f + 6))
def sb(f): return list(range(15 * f + 6, 15 * f + 15))
EX = list(range(165, 171))
VIS = [c for f in range(NF) for c in pose(f)] + EX


def make_system():
    H = np.zeros((NC, NC))
    g = rng.normal(size=NC)
    for j in range(1, NF):
        J = rng.normal(size=(15, 30))
        idx = list(range(15 * (j - 1), 15 * j + 15))
        H[np.ix_(idx, idx)] += J.T @ J
    pidx = [c for f in range(5) for c in pose(f)] + sb(0) + EX
    J0 = rng.normal(size=(len(pidx), len(pidx)))
    H[np.ix_(pidx, pidx)] += J0.T @ J0
    Jv = rng.normal(size=(200, 72))
    H[np.ix_(VIS, VIS)] += Jv.T @ Jv
    H += 1e-3 * np.eye(NC)
    return H, g


# lane -> cam column of chain A (ascending 1..4) and chain B (descending 10..6); slots a / b alternate between the pivot
# block and the next block of the chain
def front_A(f):
    lanes = [-1] * 64
    D, nxt = sb(f), sb(f + 1)
    a, b = (D, nxt) if f % 2 == 1 else (nxt, D)
    lanes[0:9], lanes[9:18] = a, b
    lanes[18:27] = sb(0)
    for p in range(6):
        lanes[27 + 6 * p:33 + 6 * p] = pose(p)
    lanes[63] = "rhs"
    return lanes, (0 if f % 2 == 1 else 9), (9 if f % 2 == 1 else 0)


def front_B(f):
    lanes = [-1] * 64
    D, nxt = sb(f), sb(f - 1)
    a, b = (D, nxt) if f % 2 == 0 else (nxt, D)
    lanes[0:9], lanes[9:18] = a, b
    for p in range(6):
        lanes[18 + 6 * p:24 + 6 * p] = pose(5 + p)
    lanes[54] = "rhs"
    return lanes, (0 if f % 2 == 0 else 9), (9 if f % 2 == 0 else 0)


def run_chain(H, g, blocks, front):
    """returns per block (lanes, X[9][64], dslot, nslot)"""
    out = []
    prevX, prev_n = None, None
    for f in blocks:
        lanes, ds, ns = front(f)
        rows = lanes[ds:ds + 9]
        R = np.zeros((9, 64))
        for j, c in enumerate(lanes):
            if c == -1:
                continue
            R[:, j] = g[rows] if c == "rhs" else H[np.ix_(rows, [c])][:, 0]
        if prevX is not None:
            Xn = prevX[:, prev_n:prev_n + 9]            # X[k][next dim i] of the previous block: the new pivot rows
            fill = Xn.T @ prevX                          # (9 x 64)
            fill[:, ns:ns + 9] = 0.0                      # lanes of the old pivot slot now hold the NEW next block: no fill
            R -= fill
        piv = np.zeros(9)
        for k in range(9):
            piv[k] = R[k, ds + k]
            assert piv[k] > 0
            fj = R[k, :] / piv[k]
            for i in range(k + 1, 9):
                R[i, :] -= R[i, ds + k] * fj
        X = R / np.sqrt(piv)[:, None]
        out.append((lanes, X, ds, ns))
        prevX, prev_n = X, ns
    return out


def dense_index():
    d = {}
    for i, c in enumerate(VIS):
        d[c] = i
    for i, c in enumerate(sb(0)):
        d[c] = 72 + i
    for i, c in enumerate(sb(5)):
        d[c] = 81 + i
    d["rhs"] = 90
    return d


def solve_proto(H, g):
    dmap = dense_index()
    Sd = np.zeros((91, 91))
    dcols = VIS + sb(0) + sb(5)
    Sd[:90, :90] = H[np.ix_(dcols, dcols)]
    Sd[90, :90] = Sd[:90, 90] = g[dcols]
    chains = [run_chain(H, g, [1, 2, 3, 4], front_A), run_chain(H, g, [10, 9, 8, 7, 6], front_B)]
    for ch in chains:
        nblk = len(ch)
        for bi, (lanes, X, ds, ns) in enumerate(ch):
            Xd = X.copy()
            Xd[:, ds:ds + 9] = 0.0
            if bi != nblk - 1:
                Xd[:, ns:ns + 9] = 0.0      # the next block is a chain block, not dense (only the last one's next is sb_5)
            A = Xd.T @ Xd
            for i, ci in enumerate(lanes):
                for j, cj in enumerate(lanes):
                    if ci == -1 or cj == -1 or A[i, j] == 0.0:
                        continue
                    Sd[dmap[ci], dmap[cj]] -= A[i, j]
    yd = np.linalg.solve(Sd[:90, :90], Sd[:90, 90])
    y = np.zeros(NC)
    for c in dcols:
        y[c] = yd[dmap[c]]
    for ch in chains:
        for (lanes, X, ds, ns) in reversed(ch):
            t = X[:, lanes.index("rhs")].copy()
            for j, c in enumerate(lanes):
                if c == -1 or c == "rhs" or ds <= j < ds + 9:
                    continue
                t -= X[:, j] * y[c]
            LT = X[:, ds:ds + 9]             # upper triangular L^T
            yf = np.zeros(9)
            for k in range(8, -1, -1):
                yf[k] = (t[k] - LT[k, k + 1:] @ yf[k + 1:]) / LT[k, k]
            y[lanes[ds:ds + 9]] = yf
    return y


if __name__ == "__main__":
    H, g = make_system()
    y = solve_proto(H, g)
    ref = np.linalg.solve(H, g)
    print("max |y - ref| / max|ref| =", np.abs(y - ref).max() / np.abs(ref).max())
    assert np.abs(y - ref).max() <= 1e-9 * np.abs(ref).max()
