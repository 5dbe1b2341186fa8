"""Generates tests/golden/ba_factors.npz: inputs and expected outputs of the four factor
residuals, the two Plus operators and the Pluecker/orthonormal maps, computed by an INDEPENDENT
NumPy float64 restatement of the formulas (derived from the cited reference
lines).  The reference itself holds no golden vectors for this path and cannot be built here
(no Eigen/Ceres/OpenCV/ROS), so these fixtures pin the oracle against a second implementation,
not against the reference binary: DESIGN.md states 'parity unpinned'.

Run:  python tests/golden/make_golden.py     (writes ba_factors.npz next to this file)
"""
import os

import numpy as np


def quat_R(q):           # q = (x, y, z, w), Eigen toRotationMatrix
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):          # Hamilton product, (x,y,z,w) storage
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    return np.array([-q[0], -q[1], -q[2], q[3]]) / np.dot(q, q)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def pose_plus(x, d):     # A(8): p + dp, normalise(q (x) [1, dtheta/2])
    dq = np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0])
    q = qmul(x[3:], dq)
    return np.concatenate([x[:3] + d[:3], q / np.linalg.norm(q)])


def orth_R(t):
    s1, c1, s2, c2, s3, c3 = np.sin(t[0]), np.cos(t[0]), np.sin(t[1]), np.cos(t[1]), np.sin(t[2]), np.cos(t[2])
    return np.array([[c2 * c3, s1 * s2 * c3 - c1 * s3, c1 * s2 * c3 + s1 * s3],
                     [c2 * s3, s1 * s2 * s3 + c1 * c3, c1 * s2 * s3 - s1 * c3],
                     [-s2, s1 * c2, c1 * c2]])


def orth_to_plk(o):      # A2
    R = orth_R(o[:3])
    return np.concatenate([np.cos(o[3]) * R[:, 0], np.sin(o[3]) * R[:, 1]])


def plk_to_orth(L):
    n, v = L[:3], L[3:]
    u1, u2 = n / np.linalg.norm(n), v / np.linalg.norm(v)
    u3 = np.cross(u1, u2)
    return np.array([np.arctan2(u2[2], u3[2]), np.arcsin(-u1[2]), np.arctan2(u1[1], u1[0]),
                     np.arcsin(np.linalg.norm(v) / np.hypot(np.linalg.norm(n), np.linalg.norm(v)))])


def orth_plus(o, d):
    def rx(a): return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    def ry(a): return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    def rz(a): return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    R = orth_R(o[:3]) @ rx(d[0]) @ ry(d[1]) @ rz(d[2])
    W = np.array([[np.cos(o[3]), -np.sin(o[3])], [np.sin(o[3]), np.cos(o[3])]]) @ \
        np.array([[np.cos(d[3]), -np.sin(d[3])], [np.sin(d[3]), np.cos(d[3])]])
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arcsin(-R[2, 0]), np.arctan2(R[1, 0], R[0, 0]), np.arcsin(W[1, 0])])


def plk_from_pose(L, R, t):   # L_c = [R^T (n - t x v); R^T v]
    n, v = L[:3], L[3:]
    return np.concatenate([R.T @ (n - np.cross(t, v)), R.T @ v])


def projection_residual(pi, pj, ex, lam, pts_i, pts_j, sqrt_info):   # A1
    Ri, Rj, ric = quat_R(pi[3:]), quat_R(pj[3:]), quat_R(ex[3:])
    pc_i = pts_i / lam
    pb_i = ric @ pc_i + ex[:3]
    pw = Ri @ pb_i + pi[:3]
    pb_j = Rj.T @ (pw - pj[:3])
    pc_j = ric.T @ (pb_j - ex[:3])
    a = pts_j / np.linalg.norm(pts_j)
    t = np.array([0, 0, 1.0]) if not np.array_equal(a, [0, 0, 1.0]) else np.array([1.0, 0, 0])
    b1 = t - a * (a @ t)
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(a, b1)
    return sqrt_info * np.array([b1, b2]) @ (pc_j / np.linalg.norm(pc_j) - a)


def line_residual(pose, ex, orth, obs, sqrt_info):                     # A3
    Lb = plk_from_pose(orth_to_plk(orth), quat_R(pose[3:]), pose[:3])
    Lc = plk_from_pose(Lb, quat_R(ex[3:]), ex[:3])
    nc = Lc[:3]
    ln = np.hypot(nc[0], nc[1])
    return sqrt_info * np.array([obs[0] * nc[0] + obs[1] * nc[1] + nc[2], obs[2] * nc[0] + obs[3] * nc[1] + nc[2]]) / ln


def vp_residual(pose, ex, orth, vp, sqrt_info):                         # A4
    Lb = plk_from_pose(orth_to_plk(orth), quat_R(pose[3:]), pose[:3])
    dc = plk_from_pose(Lb, quat_R(ex[3:]), ex[:3])[3:]
    return sqrt_info * np.array([dc[0] / dc[2] - vp[0] / vp[2], dc[1] / dc[2] - vp[1] / vp[2]])


def imu_residual_raw(pi, sbi, pj, sbj, pre, g):                         # A6 (un-whitened)
    Ri = quat_R(pi[3:])
    G = np.array([0, 0, g])
    dt = pre["sum_dt"]
    J = pre["jacobian"].reshape(15, 15)
    dba, dbg = sbi[3:6] - pre["lba"], sbi[6:9] - pre["lbg"]
    th = J[3:6, 12:15] @ dbg
    cq = qmul(pre["dq"], np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0]))
    cv = pre["dv"] + J[6:9, 9:12] @ dba + J[6:9, 12:15] @ dbg
    cp = pre["dp"] + J[0:3, 9:12] @ dba + J[0:3, 12:15] @ dbg
    rp = Ri.T @ (0.5 * G * dt * dt + pj[:3] - pi[:3] - sbi[:3] * dt) - cp
    rq = 2 * qmul(qinv(cq), qmul(qinv(pi[3:]), pj[3:]))[:3]
    rv = Ri.T @ (G * dt + sbj[:3] - sbi[:3]) - cv
    return np.concatenate([rp, rq, rv, sbj[3:6] - sbi[3:6], sbj[6:9] - sbi[6:9]])


def rand_pose(rng, scale=1.0):
    q = rng.normal(size=4)
    return np.concatenate([rng.normal(size=3) * scale, q / np.linalg.norm(q)])


def main():
    rng = np.random.default_rng(20261003)
    out = {}
    n = 24
    P = np.zeros((n, 22)); pts = np.zeros((n, 6)); R = np.zeros((n, 2))
    for i in range(n):
        pi, pj, ex = rand_pose(rng, .5), rand_pose(rng, .5), rand_pose(rng, .05)
        pj[3:] = pi[3:] + .1 * rng.normal(size=4); pj[3:] /= np.linalg.norm(pj[3:])
        lam = rng.uniform(.1, .5)
        pts[i] = np.concatenate([rng.uniform(-.5, .5, 2), [1.], rng.uniform(-.5, .5, 2), [1.]])
        P[i] = np.concatenate([pi, pj, ex, [lam]])
        R[i] = projection_residual(pi, pj, ex, lam, pts[i, :3], pts[i, 3:], 460 / 1.5)
    out.update(proj_params=P, proj_pts=pts, proj_res=R)
    L = np.zeros((n, 18)); obs = rng.uniform(-.5, .5, (n, 4)); vp = rng.normal(size=(n, 3)); vp[:, 2] = np.abs(vp[:, 2]) + .5
    RL = np.zeros((n, 2)); RV = np.zeros((n, 2)); plk = np.zeros((n, 6)); oplus = np.zeros((n, 4)); od = rng.normal(size=(n, 4)) * .1
    for i in range(n):
        pose, ex = rand_pose(rng, 1.), rand_pose(rng, .05)
        p1 = rng.normal(size=3) * 2 + [0, 0, 5]; d = rng.normal(size=3); d /= np.linalg.norm(d)
        orth = plk_to_orth(np.concatenate([np.cross(p1, d), d]))
        L[i] = np.concatenate([pose, ex, orth])
        RL[i] = line_residual(pose, ex, orth, obs[i], 306.666666667)
        RV[i] = vp_residual(pose, ex, orth, vp[i], 10.0)
        plk[i] = orth_to_plk(orth)
        oplus[i] = orth_plus(orth, od[i])
    out.update(line_params=L, line_obs=obs, line_res=RL, vp_obs=vp, vp_res=RV, orth_plk=plk, orth_delta=od, orth_plus=oplus)
    X = np.array([rand_pose(rng) for _ in range(n)]); D = rng.normal(size=(n, 6)) * .1
    out.update(pose_x=X, pose_delta=D, pose_plus=np.array([pose_plus(X[i], D[i]) for i in range(n)]))
    # IMU: synthetic pre-integration quantities (not physically consistent -- only the algebra is pinned)
    m = 8
    Ip = np.zeros((m, 32)); Ir = np.zeros((m, 15)); pre_pack = np.zeros((m, 1 + 3 + 4 + 3 + 3 + 3 + 225))
    for i in range(m):
        pi, pj = rand_pose(rng, .5), rand_pose(rng, .5)
        pj[3:] = pi[3:] + .05 * rng.normal(size=4); pj[3:] /= np.linalg.norm(pj[3:])
        sbi, sbj = rng.normal(size=9) * .1, rng.normal(size=9) * .1
        dq = rng.normal(size=4) * .05 + [0, 0, 0, 1]; dq /= np.linalg.norm(dq)
        pre = dict(sum_dt=rng.uniform(.05, .3), dp=rng.normal(size=3) * .1, dq=dq, dv=rng.normal(size=3) * .1,
                   lba=rng.normal(size=3) * .01, lbg=rng.normal(size=3) * .001, jacobian=rng.normal(size=225) * .1)
        Ip[i] = np.concatenate([pi, sbi, pj, sbj])
        Ir[i] = imu_residual_raw(pi, sbi, pj, sbj, pre, 9.81007)
        pre_pack[i] = np.concatenate([[pre["sum_dt"]], pre["dp"], pre["dq"], pre["dv"], pre["lba"], pre["lbg"], pre["jacobian"]])
    out.update(imu_params=Ip, imu_pre=pre_pack, imu_res_raw=Ir)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ba_factors.npz"), **out)
    print("wrote ba_factors.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
