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
// Block-cooperative cyclic Jacobi eigen-solver for a symmetric matrix held in LDS.
// The matrix is padded to an even dimension m (dummy row/column of zeros when n is odd) with row
// stride ld.  Parallel round-robin ordering: m/2 disjoint rotations per step; every step is ONE pass
// over 2x2 blocks B' = R_P^T B R_Q (double buffered between A0/A1) plus the column update of V, so a
// step costs two barriers.  On exit *Aout points at the buffer whose diagonal holds the eigenvalues;
// V's columns are the eigenvectors.
__device__ double* jacobi_eig(double* A0, double* A1, double* V, int m, int ld, double* cs, int* prm, double* red, int* nsweeps) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int half = m / 2;
  for (int i = tid; i < m * m; i += T) V[(i / m) * ld + (i % m)] = (i / m == i % m) ? 1.0 : 0.0;
  // the task list of a step does not depend on the step: decode it once (no integer divisions inside)
  constexpr int MAXT = 8;
  const int nblk = half * (half + 1) / 2;        // 2x2 blocks with P >= Q (the mirror is written too)
  const int ntask = nblk + half * m;
  int tP[MAXT], tQ[MAXT];                        // A task: (P, Q) ; V task: (P, -1 - k)
  int nt = 0;
  for (int it = tid; it < ntask && nt < MAXT; it += T, ++nt) {
    if (it < nblk) {
      int P, Q;
      tri_decode(it, P, Q);
      tP[nt] = P; tQ[nt] = Q;
    } else {
      const int it2 = it - nblk;
      tP[nt] = it2 / m;
      tQ[nt] = -1 - (it2 % m);
    }
  }
  double* cur = A0;
  double* nxt = A1;
  // Convergence: the classical relative criterion -- a sweep in which no pair needed a rotation
  // (|a_pq| <= 1e-15 sqrt(|a_pp a_qq|)).  The matrices here are strongly graded (eigenvalues from
  // 1e-8 to 1e10) and the pseudo-inverse / 1/sqrt(lambda) of the small ones is what is consumed,
  // so relative (not Frobenius) accuracy is required; graded inputs take 12-20 sweeps.
  int* rotflag = prm + 2 * half;   // prm holds 2*half ints; one spare slot follows (m + 1 allocated)
  for (int sweep = 0; sweep < 40; ++sweep) {
    __syncthreads();
    const int had = (sweep == 0) ? 1 : *rotflag;
    __syncthreads();
    if (nsweeps && threadIdx.x == 0) *nsweeps = sweep;
    if (!had) break;
    if (tid == 0) *rotflag = 0;
    __syncthreads();
    for (int step = 0; step < m - 1; ++step) {
      if (tid < half) {   // round-robin pairing: player 0 fixed, the others rotate
        int a = tid == 0 ? 0 : 1 + (tid - 1 + step) % (m - 1);
        int b = tid == 0 ? 1 + (m - 2 + step) % (m - 1) : 1 + (m - 2 - tid + step) % (m - 1);
        const int p = min(a, b), q = max(a, b);
        double c = 1.0, s = 0.0;
        const double apq = cur[p * ld + q], app = cur[p * ld + p], aqq = cur[q * ld + q];
        if (fabs(apq) > 1e-15 * sqrt(fabs(app * aqq)) && apq != 0.0) {
          *rotflag = 1;
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
        prm[2 * tid] = p; prm[2 * tid + 1] = q;
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
      __syncthreads();
#pragma unroll
      for (int k2 = 0; k2 < MAXT; ++k2) {
        if (k2 >= nt) break;
        const int P = tP[k2], Q = tQ[k2];
        const int p = prm[2 * P], q = prm[2 * P + 1];
        const double c1 = cs[2 * P], s1 = cs[2 * P + 1];
        if (Q >= 0) {
          const int r = prm[2 * Q], u = prm[2 * Q + 1];
          const double c2 = cs[2 * Q], s2 = cs[2 * Q + 1];
          const double bpr = cur[p * ld + r], bpu = cur[p * ld + u], bqr = cur[q * ld + r], bqu = cur[q * ld + u];
          // columns: [x_r, x_u] <- [c2 x_r - s2 x_u, s2 x_r + c2 x_u] ; rows likewise with (c1, s1)
          const double tpr = c2 * bpr - s2 * bpu, tpu = s2 * bpr + c2 * bpu;
          const double tqr = c2 * bqr - s2 * bqu, tqu = s2 * bqr + c2 * bqu;
          const double npr = c1 * tpr - s1 * tqr, npu = c1 * tpu - s1 * tqu;
          const double nqr = s1 * tpr + c1 * tqr, nqu = s1 * tpu + c1 * tqu;
          nxt[p * ld + r] = npr; nxt[p * ld + u] = npu; nxt[q * ld + r] = nqr; nxt[q * ld + u] = nqu;
          nxt[r * ld + p] = npr; nxt[u * ld + p] = npu; nxt[r * ld + q] = nqr; nxt[u * ld + q] = nqu;
        } else {
          const int k = -1 - Q;
          const double vkp = V[k * ld + p], vkq = V[k * ld + q];
          V[k * ld + p] = c1 * vkp - s1 * vkq;
          V[k * ld + q] = s1 * vkp + c1 * vkq;
        }
      }
      __syncthreads();
      double* t = cur; cur = nxt; nxt = t;
    }
  }
  __syncthreads();
  return cur;
}

constexpr int MARG_THREADS = 512;
constexpr double kMargEps = 1e-8;   // marginalization_factor.h:67
constexpr int MTROWS = 32;          // landmark rows staged per elimination pass

// LDS layout of k_marg (doubles), shared by host (size) and device (offsets)
struct MargLayout {
  int m, ldm, EB, nd, ldd, WS, total;
};
__host__ __device__ inline MargLayout marg_layout(int n) {
  MargLayout L;
  L.m = (n + 1) & ~1;
  if (L.m < 2) L.m = 2;
  L.ldm = L.m + 1;
  L.EB = L.m * L.ldm;
  L.nd = 15 + n;
  L.ldd = L.nd + 1;
  const int need = L.nd * L.ldd + MTROWS * 74 + 3 * 16 * 17;
  L.WS = need > 2 * L.EB ? need : 2 * L.EB;
  // A0 | workspace | bv(nd) | tmp(nd*16) | cs(m) | red(20) | ints: prm(m) dmap(nd) lst(2*1024)
  L.total = L.EB + L.WS + L.nd + L.nd * 16 + L.m + 20 + (L.m + 2 + L.nd + 2048 + 8) / 2 + 4;
  return L;
}

__global__ __launch_bounds__(MARG_THREADS) void k_marg(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x, T = MARG_THREADS;
  extern __shared__ double sm[];
  const int nP = B.nP[w], nL = B.nL[w];
  const int nb = B.mg_nb[w];
  const int n = B.mg_n[w];
  const MargLayout L = marg_layout(n);
  const int nd = L.nd, ldd = L.ldd, m = L.m, ldm = L.ldm;
  double* A0 = sm;                       // m x ldm : kept block / eigen buffer 0
  double* WSP = A0 + L.EB;               // workspace
  double* Ad = WSP;                      // nd x ldd dense pre-marginalisation matrix
  double* tile = Ad + nd * ldd;          // MTROWS x 74
  double* E0 = tile + MTROWS * 74;       // 3 x (16 x 17) buffers for the 15x15 eigen problem
  double* bv = WSP + L.WS;               // nd
  double* tmp = bv + nd;                 // nd * 16
  double* cs = tmp + nd * 16;            // m
  double* red = cs + m;                  // 20
  int* prm = (int*)(red + 20);           // m
  int* dmap = prm + m + 2;               // nd
  int* lst = dmap + nd;                  // 2 x 1024
  constexpr int LOFF = 1024;
  __shared__ int s_np0, s_nl0;

  // dense order: [sb_0 (9), pose_0 (6) | kept blocks in canonical order]  (the reference moves the
  // pose-like marginalised blocks behind the landmarks in descending index order, :291-309)
  if (tid < 15) dmap[tid] = tid < 9 ? 6 + tid : tid - 9;
  if (tid < nb) {
    const int kind = B.mg_kind[(size_t)w * MAXPB + tid];
    const int base = B.mg_cam[(size_t)w * MAXPB + tid], idx = B.mg_idx[(size_t)w * MAXPB + tid];
    const int ls = kind == 1 ? 9 : 6;
    for (int k = 0; k < ls; ++k) dmap[15 + idx + k] = base + k;
  }
  if (tid == 32) {
    int a = 0;
    for (int p = 0; p < nP && a < LOFF; ++p)
      if (B.pt_start[(size_t)w * B.maxP + p] == 0 && B.Hpp[(size_t)w * B.maxP + p] != 0.0) lst[a++] = p;
    s_np0 = a;
    int c = 0;
    for (int l = 0; l < nL && c < LOFF; ++l)
      if (B.ln_start[(size_t)w * B.maxL + l] == 0 && B.ln_nobs[(size_t)w * B.maxL + l] >= 2) lst[LOFF + c++] = l;
    s_nl0 = c;
  }
  __syncthreads();
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  // ---- landmark elimination (plain inverse of the block-diagonal landmark part, :316-326) --------
  // rows X = C^-1 [W | g] per start-frame-0 landmark (C C^T = H_ll); A = Hcc - X^T X, b = gc - X^T z
  for (int it = tid; it < nd * nd; it += T) {
    const int i = it / nd, j = it % nd;
    const int ci = dmap[i], cj = dmap[j];
    Ad[i * ldd + j] = ci >= cj ? Hcc[tri(ci, cj)] : Hcc[tri(cj, ci)];
  }
  for (int i = tid; i < nd; i += T) bv[i] = gc[dmap[i]];
  __syncthreads();
  const int np0 = s_np0, nl0 = s_nl0;
  for (int base = 0; base < np0 + nl0; ) {
    int nrows;
    if (base < np0) {
      const int cnt = min(MTROWS, np0 - base);
      nrows = cnt;
      for (int it = tid; it < cnt * 73; it += T) {
        const int rr = it / 73, c = it % 73;
        const size_t pi = (size_t)w * B.maxP + lst[base + rr];
        const double isq = 1.0 / sqrt(B.Hpp[pi]);
        tile[rr * 74 + c] = isq * (c < NV ? B.Wp[pi * NV + c] : B.gp[pi]);
      }
      base += cnt;
    } else {
      const int l0 = base - np0;
      const int cnt = min(MTROWS / 4, nl0 - l0);
      nrows = 4 * cnt;
      for (int it = tid; it < cnt * 73; it += T) {
        const int ll = it / 73, c = it % 73;
        const size_t li = (size_t)w * B.maxL + lst[LOFF + l0 + ll];
        const double* Hl = B.Hll + li * 16;
        double Cc[10], x[4];
        int t = 0;
        for (int a = 0; a < 4; ++a)
          for (int cc = 0; cc <= a; ++cc, ++t) Cc[t] = Hl[4 * a + cc];
        for (int jj = 0; jj < 4; ++jj) {
          double d = Cc[tri(jj, jj)];
          for (int k = 0; k < jj; ++k) d -= Cc[tri(jj, k)] * Cc[tri(jj, k)];
          d = sqrt(d);
          Cc[tri(jj, jj)] = d;
          for (int ii = jj + 1; ii < 4; ++ii) {
            double s2 = Cc[tri(ii, jj)];
            for (int k = 0; k < jj; ++k) s2 -= Cc[tri(ii, k)] * Cc[tri(jj, k)];
            Cc[tri(ii, jj)] = s2 / d;
          }
        }
        for (int a = 0; a < 4; ++a) {
          double s2 = c < NV ? B.Wl[(li * 4 + a) * NV + c] : B.gl[li * 4 + a];
          for (int k = 0; k < a; ++k) s2 -= Cc[tri(a, k)] * x[k];
          x[a] = s2 / Cc[tri(a, a)];
          tile[(4 * ll + a) * 74 + c] = x[a];
        }
      }
      base += cnt;
    }
    __syncthreads();
    for (int it = tid; it < nd * nd; it += T) {
      const int i = it / nd, j = it % nd;
      const int vi = cam2vis(dmap[i]), vj = cam2vis(dmap[j]);
      if (vi >= 0 && vj >= 0) {
        double s = 0.0;
        for (int r = 0; r < nrows; ++r) s += tile[r * 74 + vi] * tile[r * 74 + vj];
        Ad[i * ldd + j] -= s;
      }
    }
    for (int i = tid; i < nd; i += T) {
      const int vi = cam2vis(dmap[i]);
      if (vi >= 0) {
        double s = 0.0;
        for (int r = 0; r < nrows; ++r) s += tile[r * 74 + vi] * tile[r * 74 + 72];
        bv[i] -= s;
      }
    }
    __syncthreads();
  }

  // ---- marginalise the 15 dims of frame 0 through the eigen pseudo-inverse (:329-346) ----------
  double* Em0 = E0;
  double* Em1 = E0 + 16 * 17;
  double* Vmm = E0 + 2 * 16 * 17;
  for (int it = tid; it < 16 * 16; it += T) {
    const int i = it / 16, j = it % 16;
    Em0[i * 17 + j] = (i < 15 && j < 15) ? 0.5 * (Ad[i * ldd + j] + Ad[j * ldd + i]) : 0.0;
  }
  __syncthreads();
  double* Emm = jacobi_eig(Em0, Em1, Vmm, 16, 17, cs, prm, red, nullptr);
  double* Ainv = (Emm == Em0) ? Em1 : Em0;   // the other buffer is free now
  for (int it = tid; it < 225; it += T) {
    const int i = it / 15, j = it % 15;
    double s = 0;
    for (int k = 0; k < 15; ++k) {
      const double lam = Emm[k * 17 + k];
      if (lam > kMargEps) s += Vmm[i * 17 + k] * Vmm[j * 17 + k] / lam;
    }
    Ainv[i * 17 + j] = s;
  }
  __syncthreads();
  // tmp = Arm * Amm_inv  (n x 15)
  for (int it = tid; it < n * 15; it += T) {
    const int i = it / 15, j = it % 15;
    double s = 0;
    for (int k = 0; k < 15; ++k) s += Ad[(15 + i) * ldd + k] * Ainv[k * 17 + j];
    tmp[i * 16 + j] = s;
  }
  __syncthreads();
  // A0 <- Arr - tmp * Amr (padded to m x m with zeros) ; b <- brr - tmp * bmm
  double* Aout = B.mg_A + (size_t)w * MAXKEEP * MAXKEEP;
  double* bout = B.mg_b + (size_t)w * MAXKEEP;
  for (int it = tid; it < m * m; it += T) {
    const int i = it / m, j = it % m;
    double v = 0.0;
    if (i < n && j < n) {
      double s = 0;
      for (int k = 0; k < 15; ++k) s += tmp[i * 16 + k] * Ad[k * ldd + 15 + j];
      v = Ad[(15 + i) * ldd + 15 + j] - s;
      Aout[i * n + j] = v;
    }
    A0[i * ldm + j] = v;
  }
  for (int i = tid; i < n; i += T) {
    double s = 0;
    for (int k = 0; k < 15; ++k) s += tmp[i * 16 + k] * bv[k];
    const double v = bv[15 + i] - s;
    cs[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) { bv[i] = cs[i]; bout[i] = cs[i]; }
  __syncthreads();
  // ---- eigen decomposition of the kept block (:349-357): workspace now holds A1 and V ------------
  double* A1 = WSP;
  double* V = WSP + L.EB;
  double* Af = jacobi_eig(A0, A1, V, m, ldm, cs, prm, red, B.mg_m + w);
  double* J0 = B.mg_J0 + (size_t)w * MAXKEEP * MAXKEEP;
  double* r0 = B.mg_r0 + (size_t)w * MAXKEEP;
  // the reference's SelfAdjointEigenSolver returns eigenvalues ascending; the row order of J0 is
  // irrelevant to the prior it defines (an orthogonal transform of the residual), kept in Jacobi order.
  for (int it = tid; it < n * n; it += T) {
    const int k = it / n, i = it % n;
    const double lam = Af[k * ldm + k];
    const double S = lam > kMargEps ? lam : 0.0;
    J0[it] = sqrt(S) * V[i * ldm + k];
  }
  for (int k = tid; k < n; k += T) {
    const double lam = Af[k * ldm + k];
    const double Sinv = lam > kMargEps ? 1.0 / lam : 0.0;
    double vb = 0;
    for (int i = 0; i < n; ++i) vb += V[i * ldm + k] * bv[i];
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

}  // namespace vpl
