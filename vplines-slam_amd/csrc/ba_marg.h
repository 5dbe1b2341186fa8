// k_gauge : Estimator::double2vector2 (estimator.cpp:810-900) -- yaw/position gauge fix of the
//           whole window, lines carried through setLineOrth (feature_manager.cpp:367-388), then
//           the vector2double() round trip the reference performs before marginalising (:1233).
// k_marg  : MarginalizationInfo::marginalize (marginalization_factor.cpp:177-363) on the normal
//           equations k_lin<MARG> assembled: landmark elimination, eigen pseudo-inverse of the
//           oldest frame's 15 dims, eigen decomposition of the kept block -> J0, r0.
#pragma once
#include "ba_common.h"

namespace vpl {

__global__ __launch_bounds__(128) void k_gauge(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x;
  __shared__ double rd[9], p0a[3], p0b[3], newpose[NF * 7], newex[7];
  const double* gz = B.gauge + (size_t)w * 4;
  double* pose = B.pose + (size_t)w * 77;
  double* sb = B.sb + (size_t)w * 99;
  double* ex = B.ex + (size_t)w * 7;
  if (tid == 0) {
    V3 ypr00 = R2ypr(qmat(qpose(pose)));
    double y_diff = gz[0] - ypr00.x;
    M3 R = ypr2R(V3{y_diff, 0, 0});
    for (int k = 0; k < 9; ++k) rd[k] = R.m[k];
    for (int k = 0; k < 3; ++k) { p0a[k] = pose[k]; p0b[k] = gz[1 + k]; }
  }
  __syncthreads();
  M3 rot_diff;
  for (int k = 0; k < 9; ++k) rot_diff.m[k] = rd[k];
  V3 P0a{p0a[0], p0a[1], p0a[2]}, P0b{p0b[0], p0b[1], p0b[2]};
  if (tid < NF) {
    const double* x = pose + 7 * tid;
    M3 Rs = mul(rot_diff, qmat(qnormalized(qpose(x))));
    V3 Ps = mul(rot_diff, V3{x[0] - P0a.x, x[1] - P0a.y, x[2] - P0a.z}) + P0b;
    Q4 q = mat2q(Rs);   // what the next vector2double() writes
    double* o = newpose + 7 * tid;
    o[0] = Ps.x; o[1] = Ps.y; o[2] = Ps.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    double* s = sb + 9 * tid;
    V3 V = mul(rot_diff, V3{s[0], s[1], s[2]});
    s[0] = V.x; s[1] = V.y; s[2] = V.z;
  }
  if (tid == NF) {
    Q4 q = mat2q(qmat(qpose(ex)));   // ric = q.toRotationMatrix(); para = Quaterniond(ric)
    for (int k = 0; k < 3; ++k) newex[k] = ex[k];
    newex[3] = q.x; newex[4] = q.y; newex[5] = q.z; newex[6] = q.w;
  }
  __syncthreads();
  // lines: orth (world, optimised gauge) -> rotate back -> start camera frame -> orth again
  const int nL = B.nL[w];
  V3 tw1b = P0a;
  V3 twow1 = -mul(rot_diff, tw1b) + P0b;
  for (int l = tid; l < nL; l += blockDim.x) {
    const size_t li = (size_t)w * B.maxL + l;
    Plk Lw1 = orth_to_plk(B.orth + li * 4);
    Plk Lwo = plk_to_pose(Lw1, rot_diff, twow1);
    double o4[4];
    plk_to_orth(Lwo, o4);
    Plk Lw = orth_to_plk(o4);
    const int s = B.ln_start[li];
    // Rs[s] as stored by double2vector2 is rot_diff * R(q): rebuild it the same way from the OLD pose
    const double* xo = pose + 7 * s;
    M3 Rs = mul(rot_diff, qmat(qnormalized(qpose(xo))));
    V3 Ps{newpose[7 * s], newpose[7 * s + 1], newpose[7 * s + 2]};
    M3 ric = qmat(qpose(ex));
    V3 tic{ex[0], ex[1], ex[2]};
    V3 twc = Ps + mul(Rs, tic);
    M3 Rwc = mul(Rs, ric);
    Plk Lc = plk_from_pose(Lw, Rwc, twc);
    double* pl = B.plk + li * 6;
    pl[0] = Lc.n.x; pl[1] = Lc.n.y; pl[2] = Lc.n.z; pl[3] = Lc.v.x; pl[4] = Lc.v.y; pl[5] = Lc.v.z;
    // vector2double() before the marginalisation: getLineOrthVector on the updated state
    Plk Lw2 = plk_to_pose(Lc, Rwc, twc);
    plk_to_orth(Lw2, B.orth + li * 4);
  }
  __syncthreads();
  for (int i = tid; i < 77; i += blockDim.x) pose[i] = newpose[i];
  for (int i = tid; i < 7; i += blockDim.x) ex[i] = newex[i];
}

// ---------------------------------------------------------------------------------------------------
// Block-cooperative cyclic Jacobi eigen-solver on a symmetric n x n matrix in LDS (row stride ld).
// On exit A holds the eigenvalues on its diagonal and V (n x n, row stride ld) the eigenvectors
// in its columns.  Parallel round-robin ordering: n/2 disjoint rotations per step.
__device__ void jacobi_eig(double* A, double* V, int n, int ld, double* cs, int* prm, double* red) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int m = (n + 1) & ~1;       // even number of players (one dummy when n is odd)
  const int half = m / 2;
  for (int i = tid; i < n * n; i += T) V[(i / n) * ld + (i % n)] = (i / n == i % n) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, dia = 0.0;
    __syncthreads();
    for (int i = tid; i < n * n; i += T) {
      const int r = i / n, c = i % n;
      const double v = A[r * ld + c];
      if (r == c) dia += v * v; else off += v * v;
    }
    off = block_sum(off, red);
    dia = block_sum(dia, red);
    if (off <= 1e-60 || off <= 1e-28 * dia) break;
    for (int step = 0; step < m - 1; ++step) {
      // round-robin pairing: player 0 fixed, the others rotate
      if (tid < half) {
        int a = tid == 0 ? 0 : 1 + (tid - 1 + step) % (m - 1);
        int b = 1 + (m - 1 - tid - 1 + step) % (m - 1);
        if (tid == 0) b = 1 + (m - 2 + step) % (m - 1);
        int p = min(a, b), q = max(a, b);
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double apq = A[p * ld + q];
          if (apq != 0.0) {
            const double theta = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
          }
        }
        prm[2 * tid] = p; prm[2 * tid + 1] = q;
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
      __syncthreads();
      // columns of A and V
      for (int it = tid; it < half * n; it += T) {
        const int pr = it / n, k = it % n;
        const int p = prm[2 * pr], q = prm[2 * pr + 1];
        if (q >= n) continue;
        const double c = cs[2 * pr], s = cs[2 * pr + 1];
        const double akp = A[k * ld + p], akq = A[k * ld + q];
        A[k * ld + p] = c * akp - s * akq;
        A[k * ld + q] = s * akp + c * akq;
        const double vkp = V[k * ld + p], vkq = V[k * ld + q];
        V[k * ld + p] = c * vkp - s * vkq;
        V[k * ld + q] = s * vkp + c * vkq;
      }
      __syncthreads();
      // rows of A
      for (int it = tid; it < half * n; it += T) {
        const int pr = it / n, k = it % n;
        const int p = prm[2 * pr], q = prm[2 * pr + 1];
        if (q >= n) continue;
        const double c = cs[2 * pr], s = cs[2 * pr + 1];
        const double apk = A[p * ld + k], aqk = A[q * ld + k];
        A[p * ld + k] = c * apk - s * aqk;
        A[q * ld + k] = s * apk + c * aqk;
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

constexpr int MARG_THREADS = 256;
constexpr int MD = 96;            // dense working dimension: 15 (frame 0) + kept (<= 75), padded
constexpr double kMargEps = 1e-8; // marginalization_factor.h:67

// LDS: A (MD*MD) + V (MD*MD) + small.  The Schur step runs on the packed Hessian in HBM/L2
// (written by k_lin<MARG>), output-stationary, before the dense block is pulled into LDS.
__global__ __launch_bounds__(MARG_THREADS) void k_marg(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x, T = MARG_THREADS;
  extern __shared__ double sm[];
  double* A = sm;                 // MD x MD
  double* V = A + MD * MD;        // MD x MD
  double* bv = V + MD * MD;       // MD
  double* tmp = bv + MD;          // MD * 16  (Arm * Amm_inv)
  double* cs = tmp + MD * 16;     // MD
  double* red = cs + MD;          // 20
  int* prm = (int*)(red + 20);    // MD
  int* dmap = prm + MD;           // MD : dense index -> cam index

  const int nP = B.nP[w], nL = B.nL[w];
  const int nb = B.mg_nb[w];
  const int n = B.mg_n[w];
  // dense order: [sb_0 (9), pose_0 (6) | kept blocks in canonical order]  (the reference moves the
  // pose-like marginalised blocks behind the landmarks in descending index order, :291-309)
  if (tid < 15) dmap[tid] = tid < 9 ? 6 + tid : tid - 9;
  if (tid < nb) {
    const int kind = B.mg_kind[(size_t)w * MAXPB + tid];
    const int base = B.mg_cam[(size_t)w * MAXPB + tid], idx = B.mg_idx[(size_t)w * MAXPB + tid];
    const int ls = kind == 1 ? 9 : 6;
    for (int k = 0; k < ls; ++k) dmap[15 + idx + k] = base + k;
  }
  __syncthreads();
  const int nd = 15 + n;
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  // dense A, b over the involved dims, then subtract the landmark Schur terms (plain inverse, :316-326)
  for (int it = tid; it < nd * nd; it += T) {
    const int i = it / nd, j = it % nd;
    const int ci = dmap[i], cj = dmap[j];
    double v = ci >= cj ? Hcc[tri(ci, cj)] : Hcc[tri(cj, ci)];
    const int vi = cam2vis(ci), vj = cam2vis(cj);
    if (vi >= 0 && vj >= 0) {
      double s = 0.0;
      for (int p = 0; p < nP; ++p) {
        const size_t pi = (size_t)w * B.maxP + p;
        if (B.pt_start[pi] != 0) continue;
        const double h = B.Hpp[pi];
        if (h == 0.0) continue;
        s += B.Wp[pi * NV + vi] * B.Wp[pi * NV + vj] / h;
      }
      for (int l = 0; l < nL; ++l) {
        const size_t li = (size_t)w * B.maxL + l;
        if (B.ln_start[li] != 0) continue;
        // x = H^-1 w_j by Cholesky; s += w_i . x
        const double* Hl = B.Hll + li * 16;
        double C[10], x[4];
        int t = 0;
        for (int a = 0; a < 4; ++a)
          for (int c = 0; c <= a; ++c, ++t) C[t] = Hl[4 * a + c];
        for (int jj = 0; jj < 4; ++jj) {
          double d = C[tri(jj, jj)];
          for (int k = 0; k < jj; ++k) d -= C[tri(jj, k)] * C[tri(jj, k)];
          d = sqrt(d);
          C[tri(jj, jj)] = d;
          for (int ii = jj + 1; ii < 4; ++ii) {
            double s2 = C[tri(ii, jj)];
            for (int k = 0; k < jj; ++k) s2 -= C[tri(ii, k)] * C[tri(jj, k)];
            C[tri(ii, jj)] = s2 / d;
          }
        }
        for (int a = 0; a < 4; ++a) {
          double s2 = B.Wl[(li * 4 + a) * NV + vj];
          for (int k = 0; k < a; ++k) s2 -= C[tri(a, k)] * x[k];
          x[a] = s2 / C[tri(a, a)];
        }
        for (int a = 3; a >= 0; --a) {
          double s2 = x[a];
          for (int k = a + 1; k < 4; ++k) s2 -= C[tri(k, a)] * x[k];
          x[a] = s2 / C[tri(a, a)];
        }
        for (int a = 0; a < 4; ++a) s += B.Wl[(li * 4 + a) * NV + vi] * x[a];
      }
      v -= s;
    }
    A[i * MD + j] = v;
  }
  for (int i = tid; i < nd; i += T) {
    const int ci = dmap[i];
    double v = gc[ci];
    const int vi = cam2vis(ci);
    if (vi >= 0) {
      double s = 0.0;
      for (int p = 0; p < nP; ++p) {
        const size_t pi = (size_t)w * B.maxP + p;
        if (B.pt_start[pi] != 0) continue;
        const double h = B.Hpp[pi];
        if (h == 0.0) continue;
        s += B.Wp[pi * NV + vi] * B.gp[pi] / h;
      }
      for (int l = 0; l < nL; ++l) {
        const size_t li = (size_t)w * B.maxL + l;
        if (B.ln_start[li] != 0) continue;
        const double* Hl = B.Hll + li * 16;
        double C[10], x[4];
        int t = 0;
        for (int a = 0; a < 4; ++a)
          for (int c = 0; c <= a; ++c, ++t) C[t] = Hl[4 * a + c];
        for (int jj = 0; jj < 4; ++jj) {
          double d = C[tri(jj, jj)];
          for (int k = 0; k < jj; ++k) d -= C[tri(jj, k)] * C[tri(jj, k)];
          d = sqrt(d);
          C[tri(jj, jj)] = d;
          for (int ii = jj + 1; ii < 4; ++ii) {
            double s2 = C[tri(ii, jj)];
            for (int k = 0; k < jj; ++k) s2 -= C[tri(ii, k)] * C[tri(jj, k)];
            C[tri(ii, jj)] = s2 / d;
          }
        }
        for (int a = 0; a < 4; ++a) {
          double s2 = B.gl[li * 4 + a];
          for (int k = 0; k < a; ++k) s2 -= C[tri(a, k)] * x[k];
          x[a] = s2 / C[tri(a, a)];
        }
        for (int a = 3; a >= 0; --a) {
          double s2 = x[a];
          for (int k = a + 1; k < 4; ++k) s2 -= C[tri(k, a)] * x[k];
          x[a] = s2 / C[tri(a, a)];
        }
        for (int a = 0; a < 4; ++a) s += B.Wl[(li * 4 + a) * NV + vi] * x[a];
      }
      v -= s;
    }
    bv[i] = v;
  }
  __syncthreads();

  // ---- marginalise the 15 dims of frame 0 through the eigen pseudo-inverse (:329-346) ----------
  // Amm = 0.5 (A + A^T) is symmetric by construction here.  Work on a 16-stride copy in V's space.
  double* Amm = V;            // 15 x 15, stride 16
  double* Vmm = V + 16 * 16;  // eigenvectors
  for (int it = tid; it < 225; it += T) Amm[(it / 15) * 16 + it % 15] = 0.5 * (A[(it / 15) * MD + it % 15] + A[(it % 15) * MD + it / 15]);
  __syncthreads();
  jacobi_eig(Amm, Vmm, 15, 16, cs, prm, red);
  // Amm_inv = V diag(1/l if l > eps) V^T  -> stored over Amm's off-diagonal-free space: reuse tmp
  double* Ainv = V + 2 * 16 * 16;   // 15 x 15 stride 16
  for (int it = tid; it < 225; it += T) {
    const int i = it / 15, j = it % 15;
    double s = 0;
    for (int k = 0; k < 15; ++k) {
      const double lam = Amm[k * 16 + k];
      if (lam > kMargEps) s += Vmm[i * 16 + k] * Vmm[j * 16 + k] / lam;
    }
    Ainv[i * 16 + j] = s;
  }
  __syncthreads();
  // tmp = Arm * Amm_inv  (n x 15)
  for (int it = tid; it < n * 15; it += T) {
    const int i = it / 15, j = it % 15;
    double s = 0;
    for (int k = 0; k < 15; ++k) s += A[(15 + i) * MD + k] * Ainv[k * 16 + j];
    tmp[i * 16 + j] = s;
  }
  __syncthreads();
  // A <- Arr - tmp * Amr ; b <- brr - tmp * bmm   (results moved to the top-left n x n of V's upper area later)
  double* An = V + 3 * 16 * 16;   // n x n, stride MD  (fits: 768 + 75*96 < MD*MD)
  for (int it = tid; it < n * n; it += T) {
    const int i = it / n, j = it % n;
    double s = 0;
    for (int k = 0; k < 15; ++k) s += tmp[i * 16 + k] * A[k * MD + 15 + j];
    An[i * MD + j] = A[(15 + i) * MD + 15 + j] - s;
  }
  for (int i = tid; i < n; i += T) {
    double s = 0;
    for (int k = 0; k < 15; ++k) s += tmp[i * 16 + k] * bv[k];
    cs[i] = bv[15 + i] - s;   // cs doubles as b_n until the eig below (copied out first)
  }
  __syncthreads();
  double* Aout = B.mg_A + (size_t)w * MAXKEEP * MAXKEEP;
  double* bout = B.mg_b + (size_t)w * MAXKEEP;
  for (int it = tid; it < n * n; it += T) Aout[it] = An[(it / n) * MD + it % n];
  for (int i = tid; i < n; i += T) { bout[i] = cs[i]; bv[i] = cs[i]; }
  __syncthreads();
  // ---- eigen decomposition of the kept block (:349-357) ------------------------------------------
  // move An to A (stride MD), eigenvectors into V
  for (int it = tid; it < n * n; it += T) A[(it / n) * MD + it % n] = Aout[it];
  __syncthreads();
  jacobi_eig(A, V, n, MD, cs, prm, red);
  double* J0 = B.mg_J0 + (size_t)w * MAXKEEP * MAXKEEP;
  double* r0 = B.mg_r0 + (size_t)w * MAXKEEP;
  // the reference's SelfAdjointEigenSolver returns eigenvalues ascending; row order of J0 is
  // irrelevant to the prior it defines (rows of an orthogonal transform), kept in Jacobi order.
  for (int it = tid; it < n * n; it += T) {
    const int k = it / n, i = it % n;
    const double lam = A[k * MD + k];
    const double S = lam > kMargEps ? lam : 0.0;
    J0[it] = sqrt(S) * V[i * MD + k];
  }
  for (int k = tid; k < n; k += T) {
    const double lam = A[k * MD + k];
    const double Sinv = lam > kMargEps ? 1.0 / lam : 0.0;
    double vb = 0;
    for (int i = 0; i < n; ++i) vb += V[i * MD + k] * bv[i];
    r0[k] = sqrt(Sinv) * vb;
  }
  // x0 of the kept blocks: the linearisation point (preMarginalize copies, :110-129)
  if (tid < nb) {
    const int kind = B.mg_kind[(size_t)w * MAXPB + tid];
    const int base = B.mg_cam[(size_t)w * MAXPB + tid];
    const double* x = kind == 0 ? B.pose + ((size_t)w * NF + base / 15) * 7
                      : kind == 1 ? B.sb + ((size_t)w * NF + base / 15) * 9 : B.ex + (size_t)w * 7;
    const int gs = kind == 1 ? 9 : 7;
    for (int k = 0; k < 9; ++k) B.mg_x0[((size_t)w * MAXPB + tid) * 9 + k] = k < gs ? x[k] : 0.0;
  }
}

constexpr size_t MARG_SMEM = (size_t)(2 * MD * MD + MD + MD * 16 + MD + 20) * sizeof(double) + 2 * MD * sizeof(int);

}  // namespace vpl
