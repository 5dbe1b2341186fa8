// k_prep : once per solve -- IMU whitening matrices, vector2double of the lines, prior
//          normal equations, trust-region state, gauge reference.
// k_lin  : fused linearise + robustify + normal-equation accumulation of every factor of a
//          window at the current x (replaces all CostFunction::Evaluate calls of one ceres
//          evaluator pass + the JtJ accumulation of the Schur eliminator; in MARG mode the
//          factor subset and ThreadsConstructA of marginalization_factor.cpp:144-175).
//          One workgroup per window; the Jacobian is never materialised in HBM.
#pragma once
#include "ba_common.h"

namespace vpl {

constexpr int LIN_THREADS = 512;

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x;
  __shared__ double red[17];
  // (a) IMU: sqrt_info = LLT(cov^-1).matrixL().transpose()  (imu_factor.h:68)
  if (tid >= 1 && tid < NF) {
    DevPreint& P = B.pre[(size_t)w * NF + tid];
    double a[225], inv[225];
    for (int k = 0; k < 225; ++k) { a[k] = P.cov[k]; inv[k] = (k / 15 == k % 15) ? 1.0 : 0.0; }
    // Gauss-Jordan with partial pivoting
    for (int c = 0; c < 15; ++c) {
      int piv = c;
      double best = fabs(a[c * 15 + c]);
      for (int r = c + 1; r < 15; ++r)
        if (fabs(a[r * 15 + c]) > best) { best = fabs(a[r * 15 + c]); piv = r; }
      if (piv != c)
        for (int k = 0; k < 15; ++k) {
          double t = a[c * 15 + k]; a[c * 15 + k] = a[piv * 15 + k]; a[piv * 15 + k] = t;
          t = inv[c * 15 + k]; inv[c * 15 + k] = inv[piv * 15 + k]; inv[piv * 15 + k] = t;
        }
      double d = 1.0 / a[c * 15 + c];
      for (int k = 0; k < 15; ++k) { a[c * 15 + k] *= d; inv[c * 15 + k] *= d; }
      for (int r = 0; r < 15; ++r) {
        if (r == c) continue;
        double f = a[r * 15 + c];
        if (f == 0.0) continue;
        for (int k = 0; k < 15; ++k) { a[r * 15 + k] -= f * a[c * 15 + k]; inv[r * 15 + k] -= f * inv[c * 15 + k]; }
      }
    }
    // Cholesky (lower) of inv, reading its lower triangle
    for (int j = 0; j < 15; ++j) {
      double d = inv[j * 15 + j];
      for (int k = 0; k < j; ++k) d -= a[j * 15 + k] * a[j * 15 + k];
      d = sqrt(d);
      a[j * 15 + j] = d;
      for (int i = j + 1; i < 15; ++i) {
        double s = inv[i * 15 + j];
        for (int k = 0; k < j; ++k) s -= a[i * 15 + k] * a[j * 15 + k];
        a[i * 15 + j] = s / d;
      }
    }
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) P.sqrt_info[i * 15 + j] = (j >= i) ? a[j * 15 + i] : 0.0;  // L^T
  }
  // (b) states: Rs = normalized(q).toRotationMatrix(); para = Quaterniond(Rs)   (vector2double, estimator.cpp:650-705)
  if (tid < NF + 1) {
    double* x = tid < NF ? B.pose + ((size_t)w * NF + tid) * 7 : B.ex + (size_t)w * 7;
    Q4 q = mat2q(qmat(qnormalized(qpose(x))));
    if (tid == 0) {
      M3 R0 = qmat(qnormalized(qpose(x)));
      V3 ypr = R2ypr(R0);
      double* g = B.gauge + (size_t)w * 4;
      g[0] = ypr.x; g[1] = x[0]; g[2] = x[1]; g[3] = x[2];
    }
    x[3] = q.x; x[4] = q.y; x[5] = q.z; x[6] = q.w;
  }
  __syncthreads();
  // (c) lines: world orth from the start-camera-frame Pluecker (getLineOrthVector, feature_manager.cpp:341-365)
  const int nL = B.nL[w];
  for (int l = tid; l < nL; l += blockDim.x) {
    const int s = B.ln_start[(size_t)w * B.maxL + l];
    const double* ps = B.pose + ((size_t)w * NF + s) * 7;
    const double* ex = B.ex + (size_t)w * 7;
    M3 Rs = qmat(qpose(ps)), ric = qmat(qpose(ex));
    V3 P{ps[0], ps[1], ps[2]}, tic{ex[0], ex[1], ex[2]};
    V3 twc = P + mul(Rs, tic);
    M3 Rwc = mul(Rs, ric);
    const double* pl = B.plk + ((size_t)w * B.maxL + l) * 6;
    Plk Lc{V3{pl[0], pl[1], pl[2]}, V3{pl[3], pl[4], pl[5]}};
    Plk Lw = plk_to_pose(Lc, Rwc, twc);
    plk_to_orth(Lw, B.orth + ((size_t)w * B.maxL + l) * 4);
  }
  // (d) prior: H = J0^T J0 and the column map
  const int n = B.pr_n[w];
  if (n > 0) {
    const double* J0 = B.pr_J0 + (size_t)w * MAXPN * MAXPN;
    double* H = B.pr_H + (size_t)w * MAXPN * MAXPN;
    for (int idx = tid; idx < n * n; idx += blockDim.x) {
      int a = idx / n, b = idx % n;
      double s = 0;
      for (int k = 0; k < n; ++k) s += J0[(size_t)k * n + a] * J0[(size_t)k * n + b];
      H[idx] = s;
    }
    if (tid < B.pr_nb[w]) {
      int kind = B.pr_kind[(size_t)w * MAXPB + tid], fr = B.pr_frame[(size_t)w * MAXPB + tid];
      int idx = B.pr_idx[(size_t)w * MAXPB + tid];
      int base = kind == 0 ? 15 * fr : kind == 1 ? 15 * fr + 6 : 165;
      int ls = kind == 1 ? 9 : 6;
      for (int k = 0; k < ls; ++k) B.pr_map[(size_t)w * MAXPN + idx + k] = base + k;
    }
  }
  // (e) trust-region state (ceres defaults: initial radius 1e4, DoglegStrategy mu = 1e-8)
  if (tid == 0) {
    TrState t;
    t.radius = 1e4; t.mu = 1e-8; t.x_cost = 0; t.cand_cost = 0; t.model_cost_change = 0; t.x_norm = 0;
    t.step_norm = 0; t.dogleg_step_norm = 0; t.alpha = 0; t.a1 = t.a2 = t.a3 = 0; t.initial_cost = 0;
    t.iter = 0; t.status = 0; t.reuse = 0; t.step_valid = 0; t.fresh_lin = 0; t.num_successful = 0; t.num_invalid = 0;
    t.pad = 0;
    B.tr[w] = t;
  }
}

// ---------------------------------------------------------------------------------------
// helpers accumulating J_a^T J_b (2 residual rows, 6-wide blocks) into the LDS vis Hessian
__device__ __forceinline__ void acc_off(double* Hv, int ba, int bb, const double* Ja, const double* Jb) {
  // block row ba > block col bb
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) lds_add(&Hv[(6 * ba + r) * NV + 6 * bb + c], Ja[r] * Jb[c] + Ja[6 + r] * Jb[6 + c]);
}
__device__ __forceinline__ void acc_diag(double* Hv, int ba, const double* Ja) {
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) lds_add(&Hv[(6 * ba + r) * NV + 6 * ba + c], Ja[r] * Ja[c] + Ja[6 + r] * Ja[6 + c]);
}
__device__ __forceinline__ void acc_g(double* gv, int ba, const double* Ja, const double* r) {
#pragma unroll
  for (int c = 0; c < 6; ++c) lds_add(&gv[6 * ba + c], Ja[c] * r[0] + Ja[6 + c] * r[1]);
}

template <bool MARG>
__global__ __launch_bounds__(LIN_THREADS) void k_lin(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x, T = LIN_THREADS;
  TrState* tr = &B.tr[w];
  if (!MARG) {
    if (tr->status != 0 || tr->fresh_lin) return;
  }
  extern __shared__ double sm[];
  double* Hv = sm;                 // NV*NV, lower triangle used
  double* gv = Hv + NV * NV;       // NV
  double* xp = gv + NV;            // 12*7 poses + ex
  double* xs = xp + 84;            // 11*9
  double* imuJ = xs + 99;          // 10*450 whitened Jacobians
  double* imur = imuJ + 4500;      // 10*15 whitened residuals
  double* prr = imur + 150;        // MAXPN prior residual
  double* prdx = prr + MAXPN;      // MAXPN
  double* prg = prdx + MAXPN;      // MAXPN  J0^T r
  double* red = prg + MAXPN;       // 17
  int* invmap = (int*)(red + 18);  // NC
  int* imuact = invmap + NC;       // 10 (+2 pad)
  double* lacc = (double*)(imuact + 13);  // maxL * 38 per-line accumulators (NC + 13 ints keeps it 8-byte aligned)

  const int nP = B.nP[w], nL = B.nL[w];
  // the marginalisation evaluates every block, constant or not (marginalization_factor.cpp:3-69)
  const bool ex_free = MARG || B.opt.estimate_extrinsic != 0;
  for (int i = tid; i < NV * NV + NV; i += T) sm[i] = 0.0;
  for (int i = tid; i < 84; i += T) xp[i] = i < 77 ? B.pose[(size_t)w * 77 + i] : B.ex[(size_t)w * 7 + (i - 77)];
  for (int i = tid; i < 99; i += T) xs[i] = B.sb[(size_t)w * 99 + i];
  for (int i = tid; i < NC; i += T) invmap[i] = -1;
  __syncthreads();
  double cost = 0.0;
  VPL_STAMP(B, w, 16);

  // ---- prior: r = r0 + J0 dx ; g = J0^T r -------------------------------------------
  const int n = B.pr_n[w];
  if (n > 0) {
    const int nb = B.pr_nb[w];
    if (tid < nb) {
      int kind = B.pr_kind[(size_t)w * MAXPB + tid], fr = B.pr_frame[(size_t)w * MAXPB + tid];
      int idx = B.pr_idx[(size_t)w * MAXPB + tid];
      const double* x = kind == 0 ? xp + 7 * fr : kind == 1 ? xs + 9 * fr : xp + 77;
      double dx[9];
      prior_block_dx(kind, x, B.pr_x0 + ((size_t)w * MAXPB + tid) * 9, dx);
      int ls = kind == 1 ? 9 : 6;
      for (int k = 0; k < ls; ++k) prdx[idx + k] = dx[k];
    }
    for (int i = tid; i < n; i += T) invmap[B.pr_map[(size_t)w * MAXPN + i]] = i;
    __syncthreads();
    const double* J0 = B.pr_J0 + (size_t)w * MAXPN * MAXPN;
    for (int r = tid; r < n; r += T) {
      double s = B.pr_r0[(size_t)w * MAXPN + r];
      for (int c = 0; c < n; ++c) s += J0[(size_t)r * n + c] * prdx[c];
      prr[r] = s;
      cost += 0.5 * s * s;
    }
    __syncthreads();
    for (int c = tid; c < n; c += T) {
      double s = 0;
      for (int r = 0; r < n; ++r) s += J0[(size_t)r * n + c] * prr[r];
      prg[c] = s;
    }
  }

  VPL_STAMP(B, w, 17);
  // ---- IMU factors: raw residual / Jacobian per factor, then cooperative whitening ------
  if (tid < 10) {
    const int j = tid + 1;
    const DevPreint& dp = B.pre[(size_t)w * NF + j];
    bool act = MARG ? (j == 1 && dp.sum_dt < 10.0) : !(dp.sum_dt > 10.0);   // estimator.cpp:1088, :1261
    imuact[tid] = act ? 1 : 0;
    double* J = imuJ + 450 * tid;
    for (int k = 0; k < 450; ++k) J[k] = 0.0;
    if (act) {
      PreInt p = load_preint(dp);
      imu_residual_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm, imur + 15 * tid);
      ImuJac JB = imu_jacobian_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm);
      imu_jac_dense(JB, J);
    } else {
      for (int k = 0; k < 15; ++k) imur[15 * tid + k] = 0.0;
    }
  }
  __syncthreads();
  // whiten in place: column-wise, rows ascending (S upper triangular)
  for (int it = tid; it < 10 * 31; it += T) {
    const int f = it / 31, c = it % 31;
    if (!imuact[f]) continue;
    const double* S = B.pre[(size_t)w * NF + f + 1].sqrt_info;
    if (c < 30) {
      double* J = imuJ + 450 * f;
      for (int r = 0; r < 15; ++r) {
        double s = 0;
        for (int k = r; k < 15; ++k) s += S[r * 15 + k] * J[k * 30 + c];
        J[r * 30 + c] = s;
      }
    } else {
      double* rr = imur + 15 * f;
      for (int r = 0; r < 15; ++r) {
        double s = 0;
        for (int k = r; k < 15; ++k) s += S[r * 15 + k] * rr[k];
        rr[r] = s;
      }
    }
  }
  __syncthreads();
  if (tid < 10 && imuact[tid]) {
    double s = 0;
    for (int k = 0; k < 15; ++k) s += imur[15 * tid + k] * imur[15 * tid + k];
    cost += 0.5 * s;
  }

  VPL_STAMP(B, w, 18);
  // ---- visual factors ----------------------------------------------------------------------
  // Wave-uniform rounds: in every round each lane linearises (at most) one factor; the extrinsic
  // block, which every factor touches, is reduced across the wave with DPP/shuffles and added once
  // per wave, the pose blocks go to the LDS Hessian with ds_add_f64, landmark-local sums stay in
  // registers (points: one lane per track) or in LDS accumulators (lines: one lane per observation).
  const double* xe = xp + 77;
  const double hub = B.opt.huber_delta;
  const int lane = tid & 63;
  // zero the dense W rows first (frames a track does not observe must read as zero)
  {
    double* Wp0 = B.Wp + (size_t)w * B.maxP * NV;
    for (int i = tid; i < nP * NV; i += T) Wp0[i] = 0.0;
    double* Wl0 = B.Wl + (size_t)w * B.maxL * 4 * NV;
    for (int i = tid; i < nL * 4 * NV; i += T) Wl0[i] = 0.0;
    for (int i = tid; i < nL * 38; i += T) lacc[i] = 0.0;
  }
  __syncthreads();

  // points: one lane per track; rounds over the observation index k (wave-uniform trip count)
  for (int p0 = 0; p0 < nP; p0 += T) {
    const int p = p0 + tid;
    const bool live = p < nP;
    const size_t pi = (size_t)w * B.maxP + (live ? p : 0);
    const int s = live ? B.pt_start[pi] : 0, no = live ? B.pt_nobs[pi] : 0, off = live ? B.pt_off[pi] : 0;
    const bool use = live && (!MARG || s == 0);
    int nomax = use ? no : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nomax = max(nomax, __shfl_xor(nomax, o, 64));
    double* Wrow = B.Wp + pi * NV;
    const double lam = live ? B.invd[pi] : 1.0;
    const double* o0 = B.pt_obs + ((size_t)w * B.maxPO + off) * 3;
    double hll = 0, gll = 0, Wi[6], We[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { Wi[k] = 0.0; We[k] = 0.0; }
    for (int k = 1; k < nomax; ++k) {
      const bool act = use && k < no;
      double r[2] = {0, 0}, Ji[12], Jj[12], Je[12], Jl[2] = {0, 0};
#pragma unroll
      for (int q = 0; q < 12; ++q) { Ji[q] = 0.0; Jj[q] = 0.0; Je[q] = 0.0; }
      const int j = s + k;
      if (act) {
        const double* oj = o0 + 3 * k;
        projection_factor(xp + 7 * s, xp + 7 * j, xe, lam, V3{o0[0], o0[1], o0[2]}, V3{oj[0], oj[1], oj[2]},
                          B.opt.sqrt_info_point, r, true, Ji, Jj, Je, Jl);
        double sc;
        cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
        r[0] *= sc; r[1] *= sc; Jl[0] *= sc; Jl[1] *= sc;
#pragma unroll
        for (int q = 0; q < 12; ++q) { Ji[q] *= sc; Jj[q] *= sc; Je[q] *= sc; }
        if (!ex_free) {
#pragma unroll
          for (int q = 0; q < 12; ++q) Je[q] = 0.0;
        }
        hll += Jl[0] * Jl[0] + Jl[1] * Jl[1];
        gll += Jl[0] * r[0] + Jl[1] * r[1];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          Wi[a] += Jl[0] * Ji[a] + Jl[1] * Ji[6 + a];
          We[a] += Jl[0] * Je[a] + Jl[1] * Je[6 + a];
          Wrow[6 * j + a] = Jl[0] * Jj[a] + Jl[1] * Jj[6 + a];
        }
        acc_diag(Hv, j, Jj);
        acc_off(Hv, j, s, Jj, Ji);
        acc_off(Hv, 11, j, Je, Jj);
        acc_g(gv, j, Jj, r);
        acc_diag(Hv, s, Ji);
        acc_off(Hv, 11, s, Je, Ji);
        acc_g(gv, s, Ji, r);
      }
      // extrinsic block: reduce over the wave, one LDS add per wave
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double gev = wave_sum(Je[a] * r[0] + Je[6 + a] * r[1]);
        if (lane == 0) lds_add(&gv[66 + a], gev);
#pragma unroll
        for (int c = 0; c <= a; ++c, ++t) {
          const double v = wave_sum(Je[a] * Je[c] + Je[6 + a] * Je[6 + c]);
          if (lane == 0) lds_add(&Hv[(66 + a) * NV + 66 + c], v);
        }
      }
    }
    if (live) {
      if (use) {
#pragma unroll
        for (int a = 0; a < 6; ++a) { Wrow[6 * s + a] = Wi[a]; Wrow[66 + a] = We[a]; }
      }
      B.Hpp[pi] = hll;
      B.gp[pi] = gll;
    }
  }

  // lines: one lane per (track, observation); per-track sums in the LDS accumulators lacc[l][38]
  //        = H4 (10, packed lower) | g4 (4) | W_ext (4 x 6)
  {
    const int nLO = B.nLO[w];
    const int* lo_ln = B.lo_ln + (size_t)w * B.maxLO;
    for (int o0 = 0; o0 < nLO; o0 += T) {
      const int o = o0 + tid;
      const bool inb = o < nLO;
      const int l = inb ? lo_ln[o] : 0;
      const size_t li = (size_t)w * B.maxL + l;
      const int s = B.ln_start[li], off = B.ln_off[li];
      const int k = o - off, j = s + k;
      const bool act = inb && (!MARG || (s == 0 && k >= 1));   // MARG: start-frame obs skipped (estimator.cpp:1322-1326)
      const double* ob = B.ln_obs + ((size_t)w * B.maxLO + (inb ? o : 0)) * 8;
      LineCtx c;
      if (act) c = line_ctx(xp + 7 * j, xe, B.orth + li * 4);
      double Wj[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) Wj[q] = 0.0;
      double* la = lacc + l * 38;
#pragma unroll 1
      for (int fct = 0; fct < 2; ++fct) {
        // VP factor only in the solve and only when flagged (estimator.cpp:1153, :1341-1351)
        const bool fa = act && (fct == 0 || (!MARG && ob[7] == 1.0));
        double r[2] = {0, 0}, Je[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) Je[q] = 0.0;
        if (fa) {
          double jel[6], Jp[12], Jo[8];
          if (fct == 0) line_factor_res(c, ob, B.opt.sqrt_info_line, r, jel);
          else vp_factor_res(c, ob + 4, B.opt.sqrt_info_vp, r, jel);
          line_chain_jac(c, jel, fct, Jp, Je, Jo);
          double sc;
          cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
          r[0] *= sc; r[1] *= sc;
#pragma unroll
          for (int q = 0; q < 12; ++q) { Jp[q] *= sc; Je[q] *= sc; }
#pragma unroll
          for (int q = 0; q < 8; ++q) Jo[q] *= sc;
          if (!ex_free) {
#pragma unroll
            for (int q = 0; q < 12; ++q) Je[q] = 0.0;
          }
          int t = 0;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            lds_add(&la[10 + a], Jo[a] * r[0] + Jo[4 + a] * r[1]);
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2, ++t) lds_add(&la[t], Jo[a] * Jo[c2] + Jo[4 + a] * Jo[4 + c2]);
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
              Wj[6 * a + c2] += Jo[a] * Jp[c2] + Jo[4 + a] * Jp[6 + c2];
              lds_add(&la[14 + 6 * a + c2], Jo[a] * Je[c2] + Jo[4 + a] * Je[6 + c2]);
            }
          }
          acc_diag(Hv, j, Jp);
          acc_off(Hv, 11, j, Je, Jp);
          acc_g(gv, j, Jp, r);
        }
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double gev = wave_sum(Je[a] * r[0] + Je[6 + a] * r[1]);
          if (lane == 0) lds_add(&gv[66 + a], gev);
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2, ++t) {
            const double v = wave_sum(Je[a] * Je[c2] + Je[6 + a] * Je[6 + c2]);
            if (lane == 0) lds_add(&Hv[(66 + a) * NV + 66 + c2], v);
          }
        }
      }
      if (act) {
        double* Wl = B.Wl + li * 4 * NV;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c2 = 0; c2 < 6; ++c2) Wl[a * NV + 6 * j + c2] = Wj[6 * a + c2];
      }
    }
  }
  __syncthreads();
  // per-line results out of the LDS accumulators
  for (int l = tid; l < nL; l += T) {
    const size_t li = (size_t)w * B.maxL + l;
    const double* la = lacc + l * 38;
    double* Hl = B.Hll + li * 16;
    int t = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      B.gl[li * 4 + a] = la[10 + a];
#pragma unroll
      for (int c2 = 0; c2 <= a; ++c2, ++t) { Hl[4 * a + c2] = la[t]; Hl[4 * c2 + a] = la[t]; }
#pragma unroll
      for (int c2 = 0; c2 < 6; ++c2) B.Wl[(li * 4 + a) * NV + 66 + c2] = la[14 + 6 * a + c2];
    }
  }
  __syncthreads();

  VPL_STAMP(B, w, 20);
  // ---- assemble the packed cam Hessian and gradient in HBM -----------------------------------
  double* Hout = B.Hcc + (size_t)w * NCP;
  const double* pH = B.pr_H + (size_t)w * MAXPN * MAXPN;
  for (int idx = tid; idx < NCP; idx += T) {
    int r, c;
    tri_decode(idx, r, c);
    double v = 0.0;
    const int vr = cam2vis(r), vc = cam2vis(c);
    if (vr >= 0 && vc >= 0) v += Hv[vr * NV + vc];
    if (r < 165) {
      const int fr = r / 15, fc = c / 15;
      if (fc == fr || fc == fr - 1) {
        const int t0 = fr - 1;
        if (t0 >= 0 && imuact[t0]) {
          const double* J = imuJ + 450 * t0;
          const int a = r - 15 * t0, b = c - 15 * t0;
          double s = 0;
#pragma unroll
          for (int k = 0; k < 15; ++k) s += J[k * 30 + a] * J[k * 30 + b];
          v += s;
        }
        if (fc == fr && fr < 10 && imuact[fr]) {
          const double* J = imuJ + 450 * fr;
          const int a = r - 15 * fr, b = c - 15 * fr;
          double s = 0;
#pragma unroll
          for (int k = 0; k < 15; ++k) s += J[k * 30 + a] * J[k * 30 + b];
          v += s;
        }
      }
    }
    if (n > 0) {
      const int pr = invmap[r], pc = invmap[c];
      if (pr >= 0 && pc >= 0) v += pH[(size_t)pr * n + pc];
    }
    if (!ex_free && r >= 165) v = 0.0;
    Hout[idx] = v;
  }
  for (int r = tid; r < NC; r += T) {
    double v = 0.0;
    const int vr = cam2vis(r);
    if (vr >= 0) v += gv[vr];
    if (r < 165) {
      const int fr = r / 15;
      if (fr >= 1 && imuact[fr - 1]) {
        const double* J = imuJ + 450 * (fr - 1);
        const double* rr = imur + 15 * (fr - 1);
        const int a = r - 15 * (fr - 1);
        double s = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s += J[k * 30 + a] * rr[k];
        v += s;
      }
      if (fr < 10 && imuact[fr]) {
        const double* J = imuJ + 450 * fr;
        const double* rr = imur + 15 * fr;
        const int a = r - 15 * fr;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s += J[k * 30 + a] * rr[k];
        v += s;
      }
    }
    if (n > 0 && invmap[r] >= 0) v += prg[invmap[r]];
    if (!ex_free && r >= 165) v = 0.0;
    B.gc[(size_t)w * NC + r] = v;
  }
  VPL_STAMP(B, w, 21);
  cost = block_sum(cost, red);
  if (tid == 0 && !MARG) {
    tr->x_cost = cost;
    if (tr->iter == 0) tr->initial_cost = cost;
    tr->fresh_lin = 1;
  }
}

constexpr size_t LIN_SMEM_BASE = (size_t)(NV * NV + NV + 84 + 99 + 4500 + 150 + 3 * MAXPN + 18) * sizeof(double) +
                                 (size_t)(NC + 12 + 1) * sizeof(int);
inline size_t lin_smem(int maxL) { return LIN_SMEM_BASE + (size_t)maxL * 38 * sizeof(double); }

}  // namespace vpl
