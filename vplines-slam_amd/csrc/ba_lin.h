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

typedef double v4d_lin __attribute__((ext_vector_type(4)));
constexpr int LIN_THREADS = 512;
constexpr int STG_LD = 17;                // row stride of the MFMA staging tiles (16 columns; odd: the 16 writer lanes spread over the banks)
constexpr int LIN_STAGE = 8 * 32 * STG_LD;   // MFMA staging of the point phase (doubles)
constexpr int PRE_LDS = 62;               // leading doubles of DevPreint: sum_dt, dp, dq, dv, lba, lbg, the five 3x3 Jacobian blocks
constexpr int LIN_HW = 720;               // wave-private partial sums of the line phase: 11 x (21 + 36 + 6) + 21 + 6 doubles
constexpr int PREP_NMAX = 112;            // prior dims staged in LDS by k_prep (larger priors read HBM/L2); 10 x 675 + 112^2 doubles = 151 KB
__host__ __device__ constexpr int lin_stage_doubles(int maxL) {   // MFMA staging | line partial sums + per-track sums | IMU
  const int ln = 8 * LIN_HW + 38 * maxL;                 // line phase
  const int pt = LIN_STAGE + 78 * 36 + NV;               // point phase: MFMA tiles + the second commit chain's Hessian copy
  const int m = ln > pt ? ln : pt;
  return m > 4650 + 3570 ? m : 4650 + 3570;              // IMU phase: whitened [J | r] of the ten factors + their J^T J blocks (imuH)
}
constexpr int PREP_THREADS = 640;         // ten waves: one IMU factor each, all ten in one round
// per-wave scratch of the whitening (3 x 225) + the staged prior J0 of the batch's largest prior (capped at PREP_NMAX)
inline size_t prep_smem(int max_prior_n) {
  const int ns = max_prior_n < PREP_NMAX ? max_prior_n : PREP_NMAX;
  return (size_t)((PREP_THREADS / 64) * 675 + ns * ns) * sizeof(double);
}
constexpr size_t PREP_SMEM = (size_t)((PREP_THREADS / 64) * 675 + PREP_NMAX * PREP_NMAX) * sizeof(double);

// Every kernel of the solve is a `*_body` device function of (batch, window, dynamic LDS) plus a thin __global__ wrapper.
__device__ __forceinline__ void prep_body(const DevBatch& B, const int w, double* psm, int nstage) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int NWV = blockDim.x >> 6;       // 10 waves: one IMU factor each
  double* wsc = psm + wv * 675;          // per-wave scratch: G (225) | X = G^-1 (225) | P = cov^-1 (225)
  double* Jl = psm + NWV * 675;          // prior J0 staged (n * n), room for nstage x nstage
  // the prior's J0 is requested first: its round trip runs under the whitening below
  const int n = B.pr_n[w];
  const bool fits = n > 0 && n <= nstage;
  if (fits) {
    const double* J0g = B.pr_J0 + (size_t)w * B.prS;
    for (int idx = tid; idx < n * n; idx += blockDim.x) Jl[idx] = J0g[idx];
  }
  // (a) IMU: sqrt_info = LLT(cov^-1).matrixL().transpose()  (imu_factor.h:68).  cov is SPD: its inverse
  //     is formed from its Cholesky factor (cov = G G^T, cov^-1 = G^-T G^-1), then factored again.
  for (int j = 1 + wv; j < NF; j += NWV) {
    DevPreint& P = B.pre[(size_t)w * NF + j];
    double* G = wsc;
    double* X = wsc + 225;
    double* Pm = wsc + 450;
    for (int k = lane; k < 225; k += 64) G[k] = P.cov[k];
    __builtin_amdgcn_wave_barrier();
    // cov = G G^T and X = G^-1 in one pass: the register tile factorisation carries an identity along (ba_common.h)
    wave_chol16(G, 15, 15, X, lane);
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 225; e += 64) {   // P = X^T X
      const int a = e / 15, b2 = e % 15;
      double s2 = 0;
      for (int k = (a > b2 ? a : b2); k < 15; ++k) s2 += X[k * 15 + a] * X[k * 15 + b2];
      Pm[e] = s2;
    }
    __builtin_amdgcn_wave_barrier();
    wave_chol16(Pm, 15, 15, nullptr, lane);
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 225; e += 64) {
      const int i = e / 15, jj = e % 15;
      P.sqrt_info[e] = (jj >= i) ? Pm[jj * 15 + i] : 0.0;   // L^T
    }
    __builtin_amdgcn_wave_barrier();
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
      const double* fr = B.fail_ref + (size_t)w * 13;
      if (fr[0] != 0.0) {   // failure_occur: origin_R0 = R2ypr(last_R0), origin_P0 = last_P0 (estimator.cpp:818-823)
        M3 Rl;
        for (int k = 0; k < 9; ++k) Rl.m[k] = fr[4 + k];
        g[0] = R2ypr(Rl).x; g[1] = fr[1]; g[2] = fr[2]; g[3] = fr[3];
      }
    }
    x[3] = q.x; x[4] = q.y; x[5] = q.z; x[6] = q.w;
    // the lines below read the states from LDS (wave 0's scratch is free now), not back from HBM behind the stores
    double* xq = psm + 7 * tid;
    xq[0] = x[0]; xq[1] = x[1]; xq[2] = x[2]; xq[3] = q.x; xq[4] = q.y; xq[5] = q.z; xq[6] = q.w;
  }
  __syncthreads();
  // (c) lines: world orth from the start-camera-frame Pluecker (getLineOrthVector, feature_manager.cpp:341-365)
  const int nL = B.orth_in[w] ? 0 : B.nL[w];
  for (int l = tid; l < nL; l += blockDim.x) {
    const int s = B.ln_start[(size_t)w * B.maxL + l];
    const double* ps = psm + 7 * s;
    const double* ex = psm + 7 * NF;
    M3 Rs = qmat(qpose(ps)), ric = qmat(qpose(ex));
    V3 P{ps[0], ps[1], ps[2]}, tic{ex[0], ex[1], ex[2]};
    V3 twc = P + mul(Rs, tic);
    M3 Rwc = mul(Rs, ric);
    const double* pl = B.plk + ((size_t)w * B.maxL + l) * 6;
    Plk Lc{V3{pl[0], pl[1], pl[2]}, V3{pl[3], pl[4], pl[5]}};
    Plk Lw = plk_to_pose(Lc, Rwc, twc);
    plk_to_orth(Lw, B.orth + ((size_t)w * B.maxL + l) * 4);
  }
  // world Pluecker coordinates of every line's orthonormal parameters (what the factors evaluate, line_parameterization /
  // utility orth_to_plk), once per line: a lane writes the parameters above and reads them back itself
  for (int l = tid; l < B.nL[w]; l += blockDim.x) {
    const size_t li = (size_t)w * B.maxL + l;
    const Plk Lw = orth_to_plk(B.orth + li * 4);
    double* o = B.lw + li * 6;
    o[0] = Lw.n.x; o[1] = Lw.n.y; o[2] = Lw.n.z; o[3] = Lw.v.x; o[4] = Lw.v.y; o[5] = Lw.v.z;
  }
  // (d) prior: H = J0^T J0 (J0 staged in LDS) and the column map
  if (n > 0) {
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
    double* H = B.pr_H + (size_t)w * B.prS;
    const double* Js = fits ? Jl : J0;   // (staged at the top; the barrier after the states made it visible)
    if (fits) {
      // on the FP64 matrix cores (round 4): the lower 16 x 16 tiles of J0^T J0 dealt to the waves, K-steps of four rows of J0;
      // lane (kk, m) supplies J0[4 ks + kk][16 t + m] for both operands and holds C[kk + 4 v][m].  One scalar dot product of
      // length n per entry cost 0.25 ms per 512 windows at the reference's steady-state size n = 75 (0.10 at n = 45).
      const int NT = (n + 15) >> 4, NLT = NT * (NT + 1) / 2, KS = (n + 3) >> 2;
      const int m = lane & 15, kk = lane >> 4;
      for (int t = wv; t < NLT; t += NWV) {
        int ta, tb;
        tri_decode(t, ta, tb);
        const int ca = 16 * ta + m, cb = 16 * tb + m;
        v4d_lin acc = {0, 0, 0, 0};
        for (int ks = 0; ks < KS; ++ks) {
          const int k = 4 * ks + kk;
          const double av = (k < n && ca < n) ? Js[k * n + ca] : 0.0;
          const double bv = (k < n && cb < n) ? Js[k * n + cb] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        const double vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * ta + kk + 4 * v, c2 = cb;
          if (r < n && c2 < n && (ta != tb || r >= c2)) {
            H[(size_t)r * n + c2] = vals[v];
            H[(size_t)c2 * n + r] = vals[v];
          }
        }
      }
    } else
    for (int idx = tid; idx < n * n; idx += blockDim.x) {
      int a = idx / n, b = idx % n;
      double s = 0;
      for (int k = 0; k < n; ++k) s += Js[(size_t)k * n + a] * Js[(size_t)k * n + b];
      H[idx] = s;
    }
    // g0 = J0^T r0: with H it turns the gradient of the prior, J0^T (r0 + J0 dx), into g0 + H dx -- a mat-vec that does not
    // wait for the residual
    for (int c2 = tid; c2 < n; c2 += blockDim.x) {
      double s = 0;
      const double* r0g = B.pr_r0 + (size_t)w * MAXPN;
#pragma unroll 8
      for (int k = 0; k < n; ++k) s += Js[(size_t)k * n + c2] * r0g[k];
      B.pr_g0[(size_t)w * MAXPN + c2] = s;
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
__global__ __launch_bounds__(PREP_THREADS) void k_prep(DevBatch B, int nstage) {
  extern __shared__ double psm[];
  prep_body(B, blockIdx.x, psm, nstage);
}

// ---------------------------------------------------------------------------------------
// The vis Hessian in LDS: lower block triangle of 12 x 12 blocks of 6 x 6 (frames 0..10, extrinsic), every block stored
// full -- 78 blocks = 2808 doubles instead of the 72 x 72 square.  Entry (r, c), r >= c in vis indices:
constexpr int HV_DOUBLES = 78 * 36;
constexpr int LIN_PART = 78 * 36 + 72 + 8;      // B.lin_part per window: visual Hessian | gradient | cost of the point factors (k_lin2)
__host__ __device__ __forceinline__ int hvi(int r, int c) {
  const int br = r / 6, bc = c / 6;
  return 36 * (br * (br + 1) / 2 + bc) + 6 * (r - 6 * br) + (c - 6 * bc);
}

// Static description of packed cam-Hessian entry (r, c), r >= c, for the assembly pass of k_lin (the same for every
// window; built once per context):  x = (index into the LDS vis Hessian + 1) | (index into the LDS IMU blocks + 1) << 16,
// 0 meaning "no such source";  y = r | c << 8.  IMU blocks: 11 diagonal 15x15 lower triangles (120 each), then 10
// sub-diagonal full blocks (225 each).
inline void lin_asm_entry(int r, int c, int* out) {
  int hv = 0, im = 0;
  const int vr = cam2vis(r), vc = cam2vis(c);
  if (vr >= 0 && vc >= 0) hv = hvi(vr, vc) + 1;
  if (r < 165) {
    const int fr = r / 15, ar = r - 15 * fr, fc = c / 15, bc = c - 15 * fc;
    if (fc == fr) im = 120 * fr + ar * (ar + 1) / 2 + bc + 1;
    else if (fc == fr - 1) im = 11 * 120 + 225 * (fr - 1) + 15 * ar + bc + 1;
  }
  out[0] = hv | im << 16;
  out[1] = r | c << 8;
}

// MODE 0: solve linearisation; 1: MARGIN_OLD assembly (prior + IMU(0,1) + landmarks that start in frame 0);
// 2: MARGIN_SECOND_NEW assembly (the prior alone, estimator.cpp:1387-1405)
// ROLE 0: the whole linearisation in one work-group (the marginalisation passes).  The solve pass runs as TWO work-groups per
// window, side by side on two CUs: ROLE 1 = the point factors (their per-track sums go to HBM as before, their part of the
// visual Hessian / gradient to B.lin_part), ROLE 2 = prior, line / VP factors, IMU factors and the assembly, which adds ROLE
// 1's part in before it reads the visual Hessian.  In the launches where a few windows linearise (most of a solve: the step
// of the others was rejected) more than half of the CUs were idle while one work-group per window walked the phases one
// after the other; with every CU busy the split costs nothing (the same work in twice as many work-groups).  ROLE 2's
// work-group follows ROLE 1's in dispatch order, so the one it waits for is always resident or done.
template <int MODE, int ROLE>
__device__ __forceinline__ void lin_body(const DevBatch& B, const int w, double* sm) {
  constexpr bool MARG = MODE != 0;
  constexpr bool PRIOR_ONLY = MODE == 2;
  constexpr bool DO_PTS = ROLE != 2, DO_REST = ROLE != 1;
  static_assert(ROLE == 0 || MODE == 0, "the marginalisation passes are one work-group");
  const int tid = threadIdx.x, T = LIN_THREADS;
  TrState* tr = &B.tr[w];
  if (!MARG) {
    if (tr->status != 0 || tr->fresh_lin) return;
    if (DO_REST) count_active(B, 0);
  }
  if (PRIOR_ONLY && B.mg_n[w] == 0) return;
  double* Hv = sm;                 // HV_DOUBLES, see hvi()
  double* gv = Hv + HV_DOUBLES;    // NV
  double* xp = gv + NV;            // 12*7 poses + ex
  double* xs = xp + 84;            // 11*9
  // [imuJ | imur] is one region with three tenants in turn: the MFMA staging of the point phase (8 waves x 32 rows x 20
  // doubles), the wave-private partial sums + per-track accumulators of the line phase (8 x LIN_HW + 38 maxL), the whitened
  // IMU Jacobians (the assembly reads them)
  double* imuJ = xs + 99;          // 10*450 whitened Jacobians
  double* imur = imuJ + 4500;      // 10*15 whitened residuals
  const int stg_doubles = lin_stage_doubles(B.maxL);
  double* prr = imuJ + stg_doubles;  // MAXPN prior residual
  double* prdx = prr + MAXPN;      // MAXPN
  double* prg = prdx + MAXPN;      // MAXPN  J0^T r
  double* red = prg + MAXPN;       // 18
  const int PST = B.maxP | 1;      // pacc is [14][PST]: lanes with different tracks hit different banks
  double* pacc = red + 18;         // per-track sums over the factors: H_ll | g_l | W_s (6) | W_ext (6)
  double* prH = pacc + 14 * PST;   // packed lower triangle of the prior's J0^T J0 (priors of up to PRH_N dims)
  const int PRH_N = B.prhN;        // priors up to this size keep J0^T J0 in LDS for the assembly (what the context's LDS budget leaves)
  double* plds = prH + PRH_N * (PRH_N + 1) / 2;   // 10 x PRE_LDS: the part of the pre-integrations the IMU factors read, staged at the start
  int* invmap = (int*)(plds + 10 * PRE_LDS);  // NC
  int* imuact = invmap + NC;       // 10
  int* tick = imuact + 10;         // ticket counters of the point phase (1, 2: Hessian tiles of the two halves; 3, 4: per-track sums)

  const int nP = B.nP[w], nL = B.nL[w];
  // the marginalisation evaluates every block, constant or not (marginalization_factor.cpp:3-69)
  const bool ex_free = MARG || B.opt.estimate_extrinsic != 0;
  for (int i = tid; i < HV_DOUBLES + NV; i += T) sm[i] = 0.0;
  if (DO_PTS) for (int i = tid; i < 14 * PST; i += T) pacc[i] = 0.0;
  for (int i = tid; i < 84; i += T) xp[i] = i < 77 ? B.pose[(size_t)w * 77 + i] : B.ex[(size_t)w * 7 + (i - 77)];
  if (DO_REST) {
    for (int i = tid; i < 99; i += T) xs[i] = B.sb[(size_t)w * 99 + i];
    for (int i = tid; i < 10 * PRE_LDS; i += T) {   // (the IMU phase is then free of global round trips)
      const int f = i / PRE_LDS;
      plds[i] = ((const double*)&B.pre[(size_t)w * NF + f + 1])[i - PRE_LDS * f];
    }
  }
  for (int i = tid; i < NC; i += T) invmap[i] = -1;
  if (tid < 5) tick[tid] = 0;
  // The prior's inputs are requested HERE, with the states: its block table and linearisation point (one lane per block), the
  // index map, and this lane's part of the rows of J0 and H = J0^T J0 (eight lanes per row, the first LIN_PJ columns of
  // each) -- the phase was three dependent round trips (table -> barrier -> rows), each ~3 k cycles with every CU in it.
  constexpr int LIN_PJ = 6;
  const int n = DO_REST ? B.pr_n[w] : 0, nb = n > 0 ? B.pr_nb[w] : 0;
  const int prow = tid >> 3, psub = tid & 7;
  int pkind = 0, pfr = 0, pidx = 0, pmap = -1;
  double px0[9], pj0[LIN_PJ], phv[LIN_PJ], r0r = 0.0, g0r = 0.0;
  if (tid < nb) {
    pkind = B.pr_kind[(size_t)w * MAXPB + tid]; pfr = B.pr_frame[(size_t)w * MAXPB + tid]; pidx = B.pr_idx[(size_t)w * MAXPB + tid];
#pragma unroll
    for (int k = 0; k < 9; ++k) px0[k] = B.pr_x0[((size_t)w * MAXPB + tid) * 9 + k];
  }
  if (tid < n) pmap = B.pr_map[(size_t)w * MAXPN + tid];
  {
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
    const double* Hp = B.pr_H + (size_t)w * B.prS;
#pragma unroll
    for (int q = 0; q < LIN_PJ; ++q) {
      const int c = psub + 8 * q;
      const bool in = prow < n && c < n;
      pj0[q] = in ? J0[(size_t)prow * n + c] : 0.0;
      phv[q] = in ? Hp[(size_t)prow * n + c] : 0.0;
    }
    if (prow < n && psub == 0) { r0r = B.pr_r0[(size_t)w * MAXPN + prow]; g0r = B.pr_g0[(size_t)w * MAXPN + prow]; }
  }
  __syncthreads();
  double cost = 0.0;
  VPL_STAMP(B, w, 16);

  // ---- prior: r = r0 + J0 dx ; g = J0^T r -------------------------------------------
  if (n > 0) {
    if (tid < nb) {
      const double* x = pkind == 0 ? xp + 7 * pfr : pkind == 1 ? xs + 9 * pfr : xp + 77;
      double dx[9];
      prior_block_dx(pkind, x, px0, dx);
      int ls = pkind == 1 ? 9 : 6;
      for (int k = 0; k < ls; ++k) prdx[pidx + k] = dx[k];
    }
    if (tid < n) invmap[pmap] = tid;
    for (int i = tid + T; i < n; i += T) invmap[B.pr_map[(size_t)w * MAXPN + i]] = i;
    __syncthreads();
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
    // r = r0 + J0 dx and g = J0^T r = g0 + H dx (H = J0^T J0, g0 = J0^T r0 from k_prep) in ONE pass: eight lanes per row
    const double* Hp = B.pr_H + (size_t)w * B.prS;
    for (int r = prow; r < n; r += T >> 3) {
      const bool first = r == prow;
      double s = 0, sg = 0;
      if (first) {
#pragma unroll
        for (int q = 0; q < LIN_PJ; ++q) {
          const int c = psub + 8 * q;
          if (c < n) {
            const double dxc = prdx[c];
            s += pj0[q] * dxc;
            sg += phv[q] * dxc;
            if (n <= PRH_N && c <= r) prH[r * (r + 1) / 2 + c] = phv[q];   // kept for the assembly
          }
        }
      }
      for (int c = psub + (first ? 8 * LIN_PJ : 0); c < n; c += 8) {
        const double dxc = prdx[c];
        const double hrc = Hp[(size_t)r * n + c];
        s += J0[(size_t)r * n + c] * dxc;
        sg += hrc * dxc;
        if (n <= PRH_N && c <= r) prH[r * (r + 1) / 2 + c] = hrc;
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      sg += __shfl_xor(sg, 1, 64); sg += __shfl_xor(sg, 2, 64); sg += __shfl_xor(sg, 4, 64);
      if (psub == 0) {
        if (!first) { r0r = B.pr_r0[(size_t)w * MAXPN + r]; g0r = B.pr_g0[(size_t)w * MAXPN + r]; }
        s += r0r;
        prr[r] = s;
        prg[r] = sg + g0r;
        cost += 0.5 * s * s;
      }
    }
  }

  double costp = 0.0;      // this thread's share of the point factors' cost, kept apart from the rest (see below)
  VPL_STAMP(B, w, 18);
  // ---- visual factors ----------------------------------------------------------------------
  // Wave-uniform rounds: in every round each lane linearises (at most) one factor; the extrinsic
  // block, which every factor touches, is reduced across the wave with DPP/shuffles and added once
  // per wave, the pose blocks go to the LDS Hessian with ds_add_f64, landmark-local sums stay in
  // registers (points: one lane per track) or in LDS accumulators (lines: one lane per observation).
  const double* xe = xp + 77;
  const double hub = B.opt.huber_delta;
  const int lane = tid & 63;
  // zero the W rows first where some slot is not written by a factor (shorter tracks, erased lines, the start-frame
  // slot of the lines in the marginalisation pass); uniform track lengths need no fill in the solve
  const int WS = B.WS;
  if (B.wfill) {
    // (the marginalisation pass only ever reads the rows of the tracks that start in frame 0)
    double* Wp0 = B.Wp + (size_t)w * B.maxP * WS;
    if (DO_PTS)
      for (int i = tid; i < nP * WS; i += T)
        if (!MARG || B.pt_start[(size_t)w * B.maxP + i / WS] == 0) Wp0[i] = 0.0;
    double* Wl0 = B.Wl + (size_t)w * B.maxL * 4 * WS;
    if (DO_REST)
      for (int i = tid; i < nL * 4 * WS; i += T)
        if (!MARG || B.ln_start[(size_t)w * B.maxL + i / (4 * WS)] == 0) Wl0[i] = 0.0;
  } else if (MARG) {
    // uniform tracks: the one slot no factor writes is the start-frame block of the lines (their start observation is
    // skipped in this pass) -- 24 entries per line, not the 13 k entries of all rows
    double* Wl0 = B.Wl + (size_t)w * B.maxL * 4 * WS;
    for (int i = tid; i < nL * 24; i += T) {
      const int l = i / 24, e = i - 24 * l;
      if (B.ln_start[(size_t)w * B.maxL + l] == 0) Wl0[(l * 4 + e / 6) * WS + e % 6] = 0.0;
    }
  }
  __syncthreads();

  VPL_STAMP(B, w, 23);
  // points: the work unit is (start frame s, observation index k, <= 16 of the tracks that start in s and are seen at k).
  // Inside a unit every lane works on the same pair of frames (s, j = s + k), so the 6x6 blocks a point factor touches are
  // uniform over its lanes and the sum of J^T [J | r] over them is a rank-2n update: the lanes stage their two Jacobian rows
  // in LDS and the FP64 matrix cores reduce them.  The position columns of frame j are MINUS those of frame s
  // (projection_factor.cpp:84, :96: jaco_i.leftCols<3>() = ric^T Rj^T, jaco_j.leftCols<3>() = -ric^T Rj^T), bit for bit, so a row
  // has 16 distinct columns -- s position, s rotation, j rotation, extrinsic, residual -- and ONE 16x16 tile (K = 4 per
  // instruction) holds the whole 19x19 sum; the blocks of j's position columns are committed from the same accumulator
  // entries with the sign flipped.  The host packs the units into quarter-wave slots (pu_lane / pu_sub): a full unit takes
  // a slot, small ones share one -- 200 tracks x 5 factors are 60 full units + 30 of two tracks = 64 slots = 2 rounds of
  // the 8 waves.
  if (DO_PTS) {
    const int wvi = tid >> 6, nwv = T >> 6;
    const int nRounds = PRIOR_ONLY ? 0 : (MARG ? B.pu_cnt0[w] : B.pu_cnt[w]);
    const int2* plane = (const int2*)B.pu_lane + (size_t)w * B.maxPR * 512;
    const int* psub = B.pu_sub + (size_t)w * B.maxPR * 512;
    double* stg = imuJ + wvi * (32 * STG_LD);              // this wave's staging tile: 32 rows x 20 (stride STG_LD)
    const int m16 = lane & 15, kk = lane >> 4;
    const int b_cls = m16 < 6 ? 0 : (m16 < 9 ? 1 : (m16 < 15 ? 2 : 3));       // tile column of this lane: block (s | j | extrinsic | residual)
    const int b_off = m16 < 6 ? m16 : (m16 < 9 ? m16 - 3 : m16 - 9);        // and offset inside the 6-wide block
    // Every accumulator of this phase is shared by all waves, and floating-point addition does not associate: the adds are
    // therefore committed IN A FIXED ORDER: every unit carries a ticket number from the host; its wave waits for the
    // ticket, adds its Hessian tile, passes the ticket on.  The factor math and the matrix-core reductions of the other
    // waves go on meanwhile; what is serialised is ~20 LDS instructions per unit.
    // Two commit chains run side by side: waves 0..3 add into Hv / gv, waves 4..7 into a second copy that lives in the
    // part of the staging region the MFMA tiles leave free; the copy is folded into Hv after the phase.
    const int half = wvi >= nwv / 2 ? 1 : 0, wih = wvi - half * (nwv / 2);
    double* HvC = half ? imuJ + LIN_STAGE : Hv;
    double* gvC = half ? imuJ + LIN_STAGE + HV_DOUBLES : gv;
    if (half) for (int i = tid - T / 2; i < HV_DOUBLES + NV; i += T / 2) imuJ[LIN_STAGE + i] = 0.0;
    __syncthreads();
    auto ticket_wait = [&](int which, int seq) {
      while (__hip_atomic_load(&tick[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) {}
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // (what the ticket orders is LDS; global loads in flight stay in flight)
    };
    auto ticket_pass = [&](int which, int seq) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if (lane == 0) __hip_atomic_store(&tick[which], seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
#ifdef VPL_STAMPS
    long long pt_t[6] = {0, 0, 0, 0, 0, 0};     // factor math | per-track chain | staging | MFMA | ticket wait | commit
    long long pt_c = __builtin_readcyclecounter();
#define PT_LAP(i) do { const long long t_ = __builtin_readcyclecounter(); pt_t[i] += t_ - pt_c; pt_c = t_; } while (0)
#else
#define PT_LAP(i) do {} while (0)
#endif
    // A global round trip costs ~3 k cycles (1.3 us) when all CUs are in this phase together: the lane records (track, k, start frame,
    // observation offset -- everything the loads of the factor need) and the unit table of round r + 1 are fetched while
    // round r is worked on, so that a round starts one round trip deep, not three.
    // Commit of one unit's tile.  Accumulator entry (row a = kk + 4 v, col b = m16; the tile is symmetric and every (a, b) sits
    // in exactly one lane) -> LDS Hessian / gradient, in ticket order.  Tile columns: 0..2 s position, 3..5 s rotation, 6..8 j
    // rotation, 9..14 extrinsic, 15 residual.  Four adds of the lower triangle (the residual row goes to the gradient, which
    // follows the Hessian in LDS); the lanes of rows 0..2 (kk < 3, v = 0) hold T[q][c] for every column c and add it a second
    // time, negated, as the entry of j's position row / column q (and a third time for the (j pos, j pos) block).  All indices
    // are computed before the ticket is waited for: the critical section is six LDS adds.
    auto commit = [&](const v4d_lin& acc, const int desc, const int seq) {
      const int sq = desc & 15, jq = (desc >> 4) & 15;
      const int tS = 36 * (sq * (sq + 1) / 2), tJ = 36 * (jq * (jq + 1) / 2), tE = 36 * 66, cS = 36 * sq, cJ = 36 * jq, cE = 36 * 11;
      const int colbase = (b_cls == 0 ? cS : (b_cls == 1 ? cJ : cE)) + b_off;
      int idx[4];
      bool on[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = kk + 4 * v;
        const int a_cls = a < 6 ? 0 : (a < 9 ? 1 : (a < 15 ? 2 : 3));
        const int a_off = a < 6 ? a : (a < 9 ? a - 3 : a - 9);
        idx[v] = (a_cls == 0 ? tS : (a_cls == 1 ? tJ : tE)) + 6 * a_off + colbase;
        on[v] = a >= m16;
        if (a_cls == 3) {
          idx[v] = HV_DOUBLES + (b_cls == 0 ? 6 * sq : (b_cls == 1 ? 6 * jq : 66)) + b_off;
          on[v] = m16 < 15;
        }
      }
      const int didx = m16 < 6 ? tJ + cS + 6 * kk + m16
                     : (m16 < 9 ? tJ + cJ + 6 * (m16 - 3) + kk : (m16 < 15 ? tE + cJ + 6 * (m16 - 9) + kk : HV_DOUBLES + 6 * jq + kk));
      const int jidx = tJ + cJ + 6 * kk + m16;
      const bool don = kk < 3, jon = kk < 3 && m16 <= kk;
      const double v00[4] = {acc.x, acc.y, acc.z, acc.w};
      ticket_wait(1 + half, seq);
      PT_LAP(4);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (on[v]) lds_add(&HvC[idx[v]], v00[v]);
      if (don) lds_add(&HvC[didx], -v00[0]);
      if (jon) lds_add(&HvC[jidx], v00[0]);
      ticket_pass(1 + half, seq);
      PT_LAP(5);
    };
    v4d_lin hacc = {0, 0, 0, 0};
    int pdesc = 0, pseq = 0;
    bool have = false;
    int2 rec_n = nRounds > 0 ? plane[tid] : int2{-1, 0};
    int subv_n = nRounds > 0 ? psub[(wvi * 4) * 16 + lane] : 0;
    for (int round = 0; round < nRounds; ++round) {
      const int2 rec = rec_n;
      // the wave's unit table of this round is 4 slots x 8 x (descriptor, ticket) = one int per lane, read with v_readlane
      const int subv = subv_n;
      if (round + 1 < nRounds) {
        rec_n = plane[(round + 1) * 512 + tid];
        subv_n = psub[((round + 1) * 32 + wvi * 4) * 16 + lane];
      }
      const bool has = rec.x >= 0;
      const int p = has ? rec.x & 0xffff : 0, k = has ? (rec.x >> 16) & 15 : 1, s = has ? rec.x >> 20 : 0;
      const size_t pi = (size_t)w * B.maxP + p;
      const bool act = has && !(MARG && s != 0);     // (the table lists observed factors only: k < pt_nobs)
      const int off = rec.y;
      const int j = s + k;
      double* Wrow = B.Wp + pi * WS;
      double r[2] = {0, 0}, Ji[12], Jj[12], Je[12], pv[14];
#pragma unroll
      for (int q = 0; q < 12; ++q) { Ji[q] = 0.0; Jj[q] = 0.0; Je[q] = 0.0; }
      if (act) {
        const double lam = B.invd[pi];
        const double* o0 = B.pt_obs + ((size_t)w * B.maxPO + off) * 3;
        const double* oj = o0 + 3 * k;
        double Jl[2] = {0, 0};
        projection_factor(xp + 7 * s, xp + 7 * j, xe, lam, V3{o0[0], o0[1], o0[2]}, V3{oj[0], oj[1], oj[2]},
                          B.opt.sqrt_info_point, r, true, Ji, Jj, Je, Jl);
        double sc;
        costp += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
        r[0] *= sc; r[1] *= sc; Jl[0] *= sc; Jl[1] *= sc;
#pragma unroll
        for (int q = 0; q < 12; ++q) { Ji[q] *= sc; Jj[q] *= sc; Je[q] *= sc; }
        if (!ex_free) {
#pragma unroll
          for (int q = 0; q < 12; ++q) Je[q] = 0.0;
        }
        pv[0] = Jl[0] * Jl[0] + Jl[1] * Jl[1];
        pv[1] = Jl[0] * r[0] + Jl[1] * r[1];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          pv[2 + a] = Jl[0] * Ji[a] + Jl[1] * Ji[6 + a];
          pv[8 + a] = Jl[0] * Je[a] + Jl[1] * Je[6 + a];
          // this (track, frame) entry has one writer; j's position columns are minus s's (the same bits, negated): they are
          // never formed
          Wrow[6 * k + a] = a < 3 ? -pv[2 + a] : Jl[0] * Jj[a] + Jl[1] * Jj[6 + a];
        }
      }
      const unsigned long long actmask = __ballot(act);
      // per-track sums over k: a track's factors sit in different units, i.e. in different waves -- one chain over all
      // waves; two lanes of a wave may hold the same track with different k: atomic adds, lane order
      PT_LAP(0);
      // Per-track sums over k (H_ll, g_l, W_s, W_ext): a track's factors sit in different units, but (host packing) all in
      // the same half of the work-group: one chain over the half's four waves orders the LDS adds (tick[3 + half]); two
      // lanes of a wave may hold the same track with different k -- atomic adds, lane order.
      ticket_wait(3 + half, round * (nwv / 2) + wih);
      if (act) {
        double* pa = pacc + p;
#pragma unroll
        for (int a = 0; a < 14; ++a) lds_add(&pa[a * PST], pv[a]);
      }
      ticket_pass(3 + half, round * (nwv / 2) + wih);
      PT_LAP(1);
#pragma unroll 1
      for (int qq = 0; qq < 4; ++qq) {
        if (__builtin_amdgcn_readlane(subv, 16 * qq) == 0) continue;      // empty slot (uniform)
        const unsigned qmask = (unsigned)((actmask >> (16 * qq)) & 0xffffull);
        __builtin_amdgcn_wave_barrier();
        if (kk == qq && qmask != 0) {
          double* d0 = stg + (2 * m16) * STG_LD;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              d0[rr * STG_LD + a] = Ji[6 * rr + a];
              d0[rr * STG_LD + 9 + a] = Je[6 * rr + a];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) d0[rr * STG_LD + 6 + a] = Jj[6 * rr + 3 + a];   // (Jj[0..2] = -Ji[0..2])
            d0[rr * STG_LD + 15] = r[rr];
          }
        }
        __builtin_amdgcn_wave_barrier();
        PT_LAP(2);
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
          const int desc = __builtin_amdgcn_readlane(subv, 16 * qq + 2 * i);
          if (desc == 0) break;
          // tickets of the two passes: the marginalisation pass numbers only the units of the rounds it runs (ba_pack.h)
          const int tk = __builtin_amdgcn_readlane(subv, 16 * qq + 2 * i + 1);
          const int seq = MARG ? (tk >> 16) & 0xffff : tk & 0xffff;
          const int ks0 = (desc >> 8) & 15, ks1 = (desc >> 12) & 15;
          const unsigned smask = ((1u << (2 * (ks1 - ks0))) - 1u) << (2 * ks0);
          if ((qmask & smask) == 0) {       // uniform: nothing to add (units of later start frames in the marginalisation
            if (have) { commit(hacc, pdesc, pseq); have = false; }
            ticket_wait(1 + half, seq);     // pass), the ticket still goes round
            ticket_pass(1 + half, seq);
            continue;
          }
          v4d_lin c00 = {0, 0, 0, 0};
          // operands of four steps are read before the first of their MFMAs issues (one LDS latency per group)
#pragma unroll 1
          for (int kb = ks0; kb < ks1; kb += 4) {
            double lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (kb + q < ks1) lo[q] = stg[(4 * (kb + q) + kk) * STG_LD + m16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (kb + q < ks1) c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo[q], lo[q], c00, 0, 0, 0);
          }
          PT_LAP(3);
          // the PREVIOUS unit's tile is committed now, behind this unit's matrix-core instructions: the wait for its ticket
          // (a third of the phase when every unit stopped for its own) overlaps with work that does not need the ticket
          if (have) commit(hacc, pdesc, pseq);
          hacc = c00; pdesc = desc; pseq = seq; have = true;
        }
      }
      // nothing stays pending across a round: the next thing this wave waits for is the per-track chain of the next round,
      // and the waves ahead of it in that chain may be waiting for this very ticket
      if (have) { commit(hacc, pdesc, pseq); have = false; }
    }
#ifdef VPL_STAMPS
    if (lane == 0) B.dbg[(size_t)w * 64 + 56 + wvi] = __builtin_readcyclecounter() - B.dbg[(size_t)w * 64 + 23];
    if (lane == 0 && wvi == 0)
      for (int a = 0; a < 6; ++a) B.dbg[(size_t)w * 64 + 44 + a] = pt_t[a];
#endif
  }
  __syncthreads();
  VPL_STAMP(B, w, 51);
  if (DO_PTS) for (int i = tid; i < HV_DOUBLES + NV; i += T) sm[i] += imuJ[LIN_STAGE + i];   // second commit chain's copy (Hv | gv are contiguous)
  VPL_STAMP(B, w, 50);
  if (DO_PTS) for (int p = tid; p < nP; p += T) {   // per-track sums out of LDS
    const size_t pi = (size_t)w * B.maxP + p;
    const double* pa = pacc + p;
    double* Wrow = B.Wp + pi * WS;
    B.Hpp[pi] = pa[0];
    B.gp[pi] = pa[PST];
#pragma unroll
    for (int a = 0; a < 6; ++a) { Wrow[a] = pa[(2 + a) * PST]; Wrow[WS - 6 + a] = pa[(8 + a) * PST]; }
  }
  VPL_STAMP(B, w, 55);
  __syncthreads();   // staging space is handed over to the IMU / line phases
  VPL_STAMP(B, w, 24);
  // The point factors' cost is summed on its own and joins the rest in thread 0 before the final sum, in EVERY role: one
  // work-group or two, the window's cost (and with it every accept / reject decision) is the same bits.
  double cost_pts = 0.0;
  if (!MARG && DO_PTS) {
    cost_pts = block_sum(costp, red);
  }
  if (ROLE == 1) {
    // this work-group's part of the visual Hessian / gradient and of the cost go to the window's other work-group with
    // device-coherent stores (agent scope: global_store ... sc1, written through to the L2 both work-groups share -- they sit
    // on the same XCD).  RELEASE, spelled out: every thread waits until ITS stores have been acknowledged (s_waitcnt
    // vmcnt(0): on gfx9 stores count in vmcnt, and a write-through store is acknowledged by the L2), THEN the barrier, THEN
    // the flag.  s_barrier alone does not wait for outstanding stores (the payload of another wave could still
    // be on its way to a different L2 channel when the flag landed).  __builtin_amdgcn_fence(release, "agent") would do the
    // wait too but adds buffer_wbl2 sc1, a write-back of the whole L2 per wave that every work-group of the XCD pays for
    // (measured in round 3: 490 us against 270 for the 512-window launch); with write-through stores there is nothing
    // to write back.
    double* part = B.lin_part + (size_t)w * LIN_PART;
    for (int i = tid; i < HV_DOUBLES + NV; i += T) __hip_atomic_store(&part[i], sm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(&part[HV_DOUBLES + NV], cost_pts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt 7, lgkmcnt 15: not waited for)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&B.lin_flag[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  VPL_STAMP(B, w, 22);
  // ---- lines -----------------------------------------------------------------------------------------------------
  // One lane per (track, observation), laid out by the host table ll_tab: every wave holds WHOLE tracks, k-major --
  // lane = k * NLW + i for observation k of the wave's i-th line.  Every LDS accumulator of this phase is touched by ONE
  // wave only: the per-track sums lacc[l] (H4 | g4 | W_ext) because a track lives in one wave, the pose / extrinsic
  // blocks because each wave adds into its PRIVATE partial sums Hw (LIN_HW doubles per wave), folded into the LDS
  // Hessian in fixed order after the phase.  The adds of one wave reach the LDS in program order (and the lanes of one
  // instruction in lane order), so the sums -- and with them the whole solve -- are the same bits on every run; the
  // cross-wave ds_add_f64 of round 1 made two solves of identical inputs differ by 1e-9 .. 1e-6 m.
  {
    const int NLW = B.llNLW, KL = B.llK;
    const int wvi = tid >> 6;
    double* Hw = imuJ + wvi * LIN_HW;
    double* lacc = imuJ + 8 * LIN_HW;                    // maxL * 38 per-track accumulators
    for (int i = lane; i < LIN_HW; i += 64) Hw[i] = 0.0;
    for (int i = tid; i < nL * 38; i += T) lacc[i] = 0.0;
    __syncthreads();
    const int2* ltab = (const int2*)B.ll_tab + (size_t)w * B.llSlots;
    const int npass = B.ll_np[w];
    const int krow = lane / NLW;                         // observation index of this lane's slot
    for (int pass = 0; pass < npass; ++pass) {
      // the lane's record: observation, line | k << 16 | start frame << 20 (one load; the factor's data is the next trip)
      const int2 lr = (!PRIOR_ONLY && krow < KL) ? ltab[pass * T + tid] : int2{-1, 0};
      const int o = lr.x;
      const bool inb = o >= 0;
      const int l = inb ? lr.y & 0xffff : 0;
      const size_t li = (size_t)w * B.maxL + l;
      const int s = inb ? lr.y >> 20 : 0;
      const int k = inb ? (lr.y >> 16) & 15 : 0, j = s + k;
      // MARG: start-frame obs skipped (estimator.cpp:1322-1326), erased lines are no longer in f_manager.linefeature
      const bool act = inb && (!MARG || (s == 0 && k >= 1 && !B.ln_removed[li]));
      const double* ob = B.ln_obs + ((size_t)w * B.maxLO + (inb ? o : 0)) * 8;
      LineCtx c;
      if (act) {
        const double* lw = B.lw + li * 6;     // orth_to_plk(B.orth) of this line (k_prep / k_cost keep it current)
        c = line_ctx_plk(xp + 7 * j, xe, Plk{V3{lw[0], lw[1], lw[2]}, V3{lw[3], lw[4], lw[5]}});
      }
      double Wj[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) Wj[q] = 0.0;
      double* la = lacc + l * 38;
      double* Hf = Hw + 63 * (act ? j : 0);
#pragma unroll
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
          // the VP factor has no translation columns (Jp[0..2] = Je[0..2] = 0 in both rows): its zero products are skipped
          const int c0 = fct == 0 ? 0 : 3;
          int t = 0;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            lds_add(&la[10 + a], Jo[a] * r[0] + Jo[4 + a] * r[1]);
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2, ++t) lds_add(&la[t], Jo[a] * Jo[c2] + Jo[4 + a] * Jo[4 + c2]);
#pragma unroll
            for (int c2 = c0; c2 < 6; ++c2) {
              Wj[6 * a + c2] += Jo[a] * Jp[c2] + Jo[4 + a] * Jp[6 + c2];
              lds_add(&la[14 + 6 * a + c2], Jo[a] * Je[c2] + Jo[4 + a] * Je[6 + c2]);
            }
          }
          // blocks (j, j), (ext, j), g_j of this wave's partial sums
          t = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2, ++t)
              if (a >= c0 && c2 >= c0) lds_add(&Hf[t], Jp[a] * Jp[c2] + Jp[6 + a] * Jp[6 + c2]);
            if (a < c0) continue;
#pragma unroll
            for (int c2 = c0; c2 < 6; ++c2) lds_add(&Hf[21 + 6 * a + c2], Je[a] * Jp[c2] + Je[6 + a] * Jp[6 + c2]);
            lds_add(&Hf[57 + a], Jp[a] * r[0] + Jp[6 + a] * r[1]);
          }
        }
        // extrinsic block: every factor touches it.  DPP row_shr sums inside each row of 16 lanes (VALU only), then the four
        // row leaders add to the wave's partial sums
        {
          const bool leader = (lane & 15) == 15;
          const int c0 = fct == 0 ? 0 : 3;
          int t = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            if (a >= c0) {
              const double gev = row_sum16(Je[a] * r[0] + Je[6 + a] * r[1]);
              if (leader) lds_add(&Hw[714 + a], gev);
            }
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2, ++t) {
              if (a < c0 || c2 < c0) continue;
              const double v = row_sum16(Je[a] * Je[c2] + Je[6 + a] * Je[6 + c2]);
              if (leader) lds_add(&Hw[693 + t], v);
            }
          }
        }
      }
      if (act) {
        double* Wl = B.Wl + li * 4 * WS;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c2 = 0; c2 < 6; ++c2) Wl[a * WS + 6 * k + c2] = Wj[6 * a + c2];
      }
    }
  }
  __syncthreads();
  // per-line results out of the LDS accumulators
  {
    const double* lacc = imuJ + 8 * LIN_HW;
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
        for (int c2 = 0; c2 < 6; ++c2) B.Wl[(li * 4 + a) * WS + WS - 6 + c2] = la[14 + 6 * a + c2];
      }
    }
  }
  // fold the eight partial sums into the LDS Hessian / gradient, one thread per entry, waves in order
  for (int e = tid; e < LIN_HW; e += T) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += imuJ[q * LIN_HW + e];
    if (e < 693) {
      const int f = e / 63, q = e - 63 * f;
      if (q < 21) {
        int a, c2;
        tri_decode(q, a, c2);
        Hv[hvi(6 * f + a, 6 * f + c2)] += v;
      } else if (q < 57) {
        const int a = (q - 21) / 6, c2 = (q - 21) % 6;
        Hv[hvi(66 + a, 6 * f + c2)] += v;
      } else {
        gv[6 * f + (q - 57)] += v;
      }
    } else if (e < 714) {
      int a, c2;
      tri_decode(e - 693, a, c2);
      Hv[hvi(66 + a, 66 + c2)] += v;
    } else {
      gv[66 + (e - 714)] += v;
    }
  }
  __syncthreads();   // the staging region goes to the IMU phase
  VPL_STAMP(B, w, 25);
  // ---- IMU factors: raw residual / Jacobian per factor, then cooperative whitening ------
  for (int i = tid; i < 4650; i += T) imuJ[i] = 0.0;   // [imuJ | imur]: by everybody, not 450 stores by each of the ten lanes below
  // S^T of this wave's factors as the A operand of the whitening (A[k][i] = S[i][k]), requested before the raw evaluation
  double sa[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f = (tid >> 6) + 8 * q;
    const double* S = B.pre[(size_t)w * NF + (f < 10 ? f : 0) + 1].sqrt_info;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = 4 * ks + (lane >> 4), i = lane & 15;
      sa[q][ks] = (f < 10 && i < 15 && k < 15 && k >= i) ? S[i * 15 + k] : 0.0;
    }
  }
  __syncthreads();
  // raw residual by ten lanes of wave 0, raw Jacobian by ten lanes of wave 1 (each is one long dependent chain)
  if ((tid & 63) < 10 && tid < 128) {
    const int f = tid & 63, j = f + 1;
    const DevPreint& dp = *(const DevPreint*)(plds + PRE_LDS * f);   // (only the staged leading part is read)
    bool act = MARG ? (j == 1 && dp.sum_dt < 10.0) : !(dp.sum_dt > 10.0);   // estimator.cpp:1088, :1261
    if (PRIOR_ONLY) act = false;
    if (tid < 64) imuact[f] = act ? 1 : 0;
    if (act) {
      PreInt p = load_preint(dp);
      if (tid < 64) {
        imu_residual_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm, imur + 15 * f);
      } else {
        ImuJac JB = imu_jacobian_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm);
        imu_jac_dense(JB, imuJ + 450 * f);
      }
    }
  }
  __syncthreads();
  VPL_STAMP(B, w, 28);
  // the static table entries of this thread's packed entries (e = u T + tid): one batch of loads, requested before the
  // whitening
  constexpr int ASM_PER_THREAD = (NCP + LIN_THREADS - 1) / LIN_THREADS;
  int2 asmd[ASM_PER_THREAD];
#pragma unroll
  for (int u = 0; u < ASM_PER_THREAD; ++u) {
    const int e = u * T + tid;
    asmd[u] = ((const int2*)B.asm_tab)[e < NCP ? e : 0];
  }
  // Whitening J <- S J (S upper triangular 15 x 15, [J | r] 15 x 31) and the factor's J^T J on the FP64 matrix cores: one
  // wave per factor (waves 0 and 1 take two).  C[i][j] = sum_k A[k][i] B[k][j]: the A lane (kk, m) supplies A[4 ks + kk][m],
  // the B lane B[4 ks + kk][m], the accumulator lane (kk, m) holds C[kk + 4 v][m].  The 3570 entries of the 11 diagonal
  // 15 x 15 blocks (lower triangles) and 10 sub-diagonal blocks go to imuH for the assembly: a frame's diagonal block is
  // (term of the factor that ends in it) + (term of the factor that starts in it), in that order -- stored, barrier, added.
  double* imuH = imuJ + 4650;
  {
    const int wvi = tid >> 6, m16 = lane & 15, kk = lane >> 4;
    v4d_lin cd[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};        // (0..14)^2 part of this wave's factors, added after the barrier
    for (int i = tid; i < 120; i += T) imuH[i] = 0.0;    // frame 0 has no factor ending in it
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = wvi + 8 * q;
      if (f >= 10) break;
      double* J = imuJ + 450 * f;
      double* rr = imur + 15 * f;
      const bool on = imuact[f] != 0;
      v4d_lin jw[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int c = 16 * ct + m16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int k = 4 * ks + kk;
          const double bv = k < 15 ? (c < 30 ? J[k * 30 + c] : (c == 30 ? rr[k] : 0.0)) : 0.0;
          jw[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(on ? sa[q][ks] : 0.0, bv, jw[ct], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (on) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const double vals[4] = {jw[ct].x, jw[ct].y, jw[ct].z, jw[ct].w};
          const int c = 16 * ct + m16;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = kk + 4 * v;
            if (r < 15) {
              if (c < 30) J[r * 30 + c] = vals[v];
              else if (c == 30) rr[r] = vals[v];
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // J^T J: tiles (0,0), (1,0), (1,1) of the 30 x 30 product
      v4d_lin c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = 4 * ks + kk;
        const double lo = k < 15 ? J[k * 30 + m16] : 0.0;
        const double hi = (k < 15 && m16 < 14) ? J[k * 30 + 16 + m16] : 0.0;
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, lo, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, lo, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, hi, c11, 0, 0, 0);
      }
      // entry (i, j) of the product, i >= j: i, j < 15 -> diagonal block of frame f (held back); i >= 15 > j -> the
      // sub-diagonal block (f + 1, f); both >= 15 -> diagonal block of frame f + 1 (stored now)
      {
        const double v00[4] = {c00.x, c00.y, c00.z, c00.w}, v10[4] = {c10.x, c10.y, c10.z, c10.w}, v11[4] = {c11.x, c11.y, c11.z, c11.w};
        double keep[4] = {0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i0 = kk + 4 * v, j0 = m16;                  // tile (0,0)
          if (i0 < 15 && j0 < 15) keep[v] = v00[v];             // (all of it is block (f, f); the row-15 entries go with tile (1,0))
          const int i1 = 16 + kk + 4 * v;                       // tiles (1,0) and (1,1): product rows 16..31
          if (i1 < 30) {
            if (j0 < 15) imuH[11 * 120 + 225 * f + 15 * (i1 - 15) + j0] = v10[v];
            else imuH[120 * (f + 1) + (i1 - 15) * (i1 - 14) / 2 + 0] = v10[v];                     // (i1, 15): column 0 of block (f+1, f+1)
            const int j1 = 16 + m16;
            if (j1 < 30 && j1 <= i1) imuH[120 * (f + 1) + (i1 - 15) * (i1 - 14) / 2 + (j1 - 15)] = v11[v];
          }
          if (i0 == 15) {                                       // product row 15 sits in tile (0,0) / its column 15
            if (j0 < 15) imuH[11 * 120 + 225 * f + j0] = v00[v];
            else imuH[120 * (f + 1)] = v00[v];                  // (15, 15)
          }
        }
        cd[q] = v4d_lin{keep[0], keep[1], keep[2], keep[3]};
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = wvi + 8 * q;
      if (f >= 10) break;
      const double keep[4] = {cd[q].x, cd[q].y, cd[q].z, cd[q].w};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i0 = kk + 4 * v, j0 = m16;
        if (i0 < 15 && j0 <= i0) imuH[120 * f + i0 * (i0 + 1) / 2 + j0] += keep[v];
      }
    }
  }
  __syncthreads();
  if (tid < 10 && imuact[tid]) {
    double s = 0;
    for (int k = 0; k < 15; ++k) s += imur[15 * tid + k] * imur[15 * tid + k];
    cost += 0.5 * s;
  }


  if (ROLE == 2) {
    // the point factors' part (the other work-group of this window): Hv = (lines) + (points), gv likewise -- the sum the
    // single work-group forms, in the other order of its two terms
    // ACQUIRE, spelled out: the flag is polled with agent-scope loads (sc1: served by the L2, never by this CU's vector
    // cache); the payload loads below are agent-scope loads too, issued after the barrier that follows the successful
    // poll (a wave issues its memory instructions in order and none of them before s_barrier completes), so they reach the
    // L2 after the flag store did -- and the writer's stores were acknowledged by that L2 before the flag was written.
    // No buffer_inv is needed because no load of this payload may hit the vector cache.
    if (tid == 0) {
      while (__hip_atomic_load(&B.lin_flag[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) __builtin_amdgcn_s_sleep(32);
      __hip_atomic_store(&B.lin_flag[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double* part = B.lin_part + (size_t)w * LIN_PART;
    for (int i = tid; i < HV_DOUBLES + NV; i += T) sm[i] += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) cost += __hip_atomic_load(&part[HV_DOUBLES + NV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  } else if (!MARG) {
    if (tid == 0) cost += cost_pts;
  }
  VPL_STAMP(B, w, 20);
  // ---- assemble the packed cam Hessian and gradient in HBM -----------------------------------
  double* Hout = B.Hcc + (size_t)w * NCP;
  const double* pH = B.pr_H + (size_t)w * B.prS;
  // Every entry is stored ONCE, with its three sources summed in registers: adding to what an earlier pass had stored
  // is a global read-modify-write, i.e. a full round trip (~3 k cycles with all CUs in this phase) behind every store.
  // (1) the 3570 entries the IMU factors touch (11 diagonal 15x15 blocks, 10 sub-diagonal ones), each summing its one or
  // two J^T J terms, go to LDS; (2) rows to waves, columns to lanes: visual block + IMU block + prior (J0^T J0 staged in LDS
  // by the prior phase), coalesced stores, no index decoding beyond c / 15.  Priors larger than PRH_N (none in the
  // reference's windows) take a third pass that adds their entries in HBM.
  VPL_STAMP(B, w, 26);
  const bool prior_lds = n > 0 && n <= PRH_N;
  {
    // flat over the packed triangle: fully coalesced stores; the table entries were requested before the IMU blocks
#pragma unroll
    for (int u = 0; u < ASM_PER_THREAD; ++u) {
      const int e = u * T + tid;
      if (e >= NCP) continue;
      const int hv = asmd[u].x & 0xffff, im = asmd[u].x >> 16, r = asmd[u].y & 255, c = asmd[u].y >> 8;
      // branch-free: every source is read (clamped index) and selected, so that the LDS reads of many entries overlap
      const double vh = Hv[hv ? hv - 1 : 0], vi = imuH[im ? im - 1 : 0];
      const int ir = prior_lds ? invmap[r] : -1, ic = prior_lds ? invmap[c] : -1;
      const bool hp = ir >= 0 && ic >= 0;
      const int hi_ = ic > ir ? ic : ir, lo_ = ic > ir ? ir : ic;
      const double vp = prH[hp ? hi_ * (hi_ + 1) / 2 + lo_ : 0];
      double v = hv ? vh : 0.0;
      v += im ? vi : 0.0;
      v += hp ? vp : 0.0;
      if (!ex_free && r >= 165) v = 0.0;
      Hout[e] = v;
    }
  }
  VPL_STAMP(B, w, 27);
  if (n > PRH_N) {
    __syncthreads();
    const int np = n * (n + 1) / 2;
    for (int base = 0; base < np; base += 4 * T) {
      double ph[4];
      int tr_[4], tc_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = base + u * T + tid;
        ph[u] = 0.0; tr_[u] = -1; tc_[u] = 0;
        if (e < np) {
          int i, j;
          tri_decode(e, i, j);
          ph[u] = pH[(size_t)i * n + j];
          const int ri = B.pr_map[(size_t)w * MAXPN + i], rj = B.pr_map[(size_t)w * MAXPN + j];
          tr_[u] = ri > rj ? ri : rj;
          tc_[u] = ri > rj ? rj : ri;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tr_[u] >= 0 && (ex_free || tr_[u] < 165)) Hout[(size_t)tr_[u] * (tr_[u] + 1) / 2 + tc_[u]] += ph[u];
    }
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
    // A NaN / inf among the inputs: ceres' evaluator rejects the initial point ("Initial residual and Jacobian evaluation
    // failed", trust_region_minimizer.cc IterationZero) and Solve returns FAILURE with an empty iteration list -- the
    // reference goes on to double2vector2 and the marginalisation with the states as they were.  Reported the same way:
    // termination 2, iterations = successful steps = -1 (size of the list - 1), costs 0; every later kernel of the solve
    // leaves a window whose status is set.
    if (tr->iter == 0 && tr->num_successful == 0 && !isfinite(cost)) {
      tr->status = 2; tr->iter = -1; tr->num_successful = -1; tr->x_cost = 0.0; tr->initial_cost = 0.0;
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(LIN_THREADS) void k_lin(DevBatch B) {
  extern __shared__ double sm[];
  lin_body<MODE, 0>(B, MODE == 0 ? ordered_window(B) : (int)blockIdx.x, sm);
}
// the solve pass: two work-groups per window (grid 2 nW), see lin_body
__global__ __launch_bounds__(LIN_THREADS) void k_lin2(DevBatch B) {
  extern __shared__ double sm[];
  // The split pays while the two work-groups of every window that linearises find a CU each (the list k_cost made of those
  // windows holds their number); beyond that it only adds a second round of work-groups and the hand-over: then the first
  // nW work-groups do one whole window each, as k_lin<0> did, and the others leave (they sit at the END of the grid:
  // interleaved with the working ones they cost a third of the launch).
  const int nlin = B.ord_it == 0 ? B.nW : B.ord_cnt[2 * (B.ord_it & 1)];
  if (2 * nlin > B.ncu) {
    if ((int)blockIdx.x < B.nW) lin_body<0, 0>(B, ordered_window(B), sm);
    return;
  }
  // work-groups go to the XCDs round-robin by index: blocks 16 q + x and 16 q + 8 + x (x < 8) are the two work-groups of
  // window 8 q + x -- same XCD (one L2 between them), the point work-group first in dispatch order
  const int b = (int)(blockIdx.x >> 4) * 8 + (int)(blockIdx.x & 7);
  if (b >= B.nW) return;
  const int w = __builtin_amdgcn_readfirstlane(B.ord_it == 0 ? b : B.order[(size_t)(B.ord_it & 1) * B.nW + b]);
  if (blockIdx.x & 8) lin_body<0, 2>(B, w, sm);
  else lin_body<0, 1>(B, w, sm);
}

// LDS of k_lin: the fixed part, then whatever is left of the budget holds the prior's J0^T J0 for the assembly (priors of up
// to lin_prh_n dims; the reference's are 6 + 9 + 6 x frames <= 75; larger ones are added in HBM by a third pass)
constexpr size_t LIN_LDS_BUDGET = 158 * 1024;
inline size_t lin_smem_base(int maxP, int maxL) {
  return (size_t)(HV_DOUBLES + NV + 84 + 99 + lin_stage_doubles(maxL) + 3 * MAXPN + 18 + 14 * (maxP | 1) + 10 * PRE_LDS) * sizeof(double) + (size_t)(NC + 16) * sizeof(int);
}
inline int lin_prh_n(int maxP, int maxL) {
  const size_t base = lin_smem_base(maxP, maxL);
  if (base >= LIN_LDS_BUDGET) return 0;
  const size_t cap = (LIN_LDS_BUDGET - base) / sizeof(double);
  int n = 0;
  while (n < 76 && (size_t)(n + 1) * (n + 2) / 2 <= cap) ++n;   // (75 is the largest prior of the reference)
  return n;
}
inline size_t lin_smem(int maxP, int maxL) {
  const int n = lin_prh_n(maxP, maxL);
  return lin_smem_base(maxP, maxL) + (size_t)(n * (n + 1) / 2) * sizeof(double);
}

}  // namespace vpl
