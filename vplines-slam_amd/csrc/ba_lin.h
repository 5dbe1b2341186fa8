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
constexpr int STG_LD = 21;                // row stride of the MFMA staging tiles (odd: the 16 writer lanes spread over the banks)
constexpr int LIN_STAGE = 8 * 32 * STG_LD;   // MFMA staging of the point phase (doubles)
constexpr int LIN_HW = 720;               // wave-private partial sums of the line phase: 11 x (21 + 36 + 6) + 21 + 6 doubles
constexpr int PREP_NMAX = 112;            // prior dims staged in LDS by k_prep (larger priors read HBM/L2); 10 x 675 + 112^2 doubles = 151 KB
__host__ __device__ constexpr int lin_stage_doubles(int maxL) {   // MFMA staging | line partial sums + per-track sums | IMU
  const int ln = 8 * LIN_HW + 38 * maxL;                 // line phase
  const int pt = LIN_STAGE + 78 * 36 + NV;               // point phase: MFMA tiles + the second commit chain's Hessian copy
  const int m = ln > pt ? ln : pt;
  return m > 4650 ? m : 4650;
}
constexpr int PREP_THREADS = 640;         // ten waves: one IMU factor each, all ten in one round
// per-wave scratch of the whitening (3 x 225) + the staged prior J0 of the batch's largest prior (capped at PREP_NMAX)
inline size_t prep_smem(int max_prior_n) {
  const int ns = max_prior_n < PREP_NMAX ? max_prior_n : PREP_NMAX;
  return (size_t)((PREP_THREADS / 64) * 675 + ns * ns) * sizeof(double);
}
constexpr size_t PREP_SMEM = (size_t)((PREP_THREADS / 64) * 675 + PREP_NMAX * PREP_NMAX) * sizeof(double);

// Every kernel of the solve is a `*_body` device function of (batch, window, dynamic LDS) plus a thin __global__ wrapper:
// the bodies are also called back to back by the one-kernel-per-solve k_window (ba_window.h).
__device__ __forceinline__ void prep_body(const DevBatch& B, const int w, double* psm, int nstage) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int NWV = blockDim.x >> 6;       // 10 waves stand-alone (one IMU factor each), 8 inside k_window
  double* wsc = psm + wv * 675;          // per-wave scratch: G (225) | X = G^-1 (225) | P = cov^-1 (225)
  double* Jl = psm + NWV * 675;          // prior J0 staged (n * n), room for nstage x nstage
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
  }
  __syncthreads();
  // (c) lines: world orth from the start-camera-frame Pluecker (getLineOrthVector, feature_manager.cpp:341-365)
  const int nL = B.orth_in[w] ? 0 : B.nL[w];
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
  // (d) prior: H = J0^T J0 (J0 staged in LDS) and the column map
  const int n = B.pr_n[w];
  if (n > 0) {
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
    double* H = B.pr_H + (size_t)w * B.prS;
    const bool fits = n <= nstage;
    if (fits) {
      for (int idx = tid; idx < n * n; idx += blockDim.x) Jl[idx] = J0[idx];
      __syncthreads();
    }
    const double* Js = fits ? Jl : J0;
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
      for (int k = 0; k < n; ++k) s += Js[(size_t)k * n + c2] * B.pr_r0[(size_t)w * MAXPN + k];
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
__device__ __forceinline__ int hvi(int r, int c) {
  const int br = r / 6, bc = c / 6;
  return 36 * (br * (br + 1) / 2 + bc) + 6 * (r - 6 * br) + (c - 6 * bc);
}

// MODE 0: solve linearisation; 1: MARGIN_OLD assembly (prior + IMU(0,1) + landmarks that start in frame 0);
// 2: MARGIN_SECOND_NEW assembly (the prior alone, estimator.cpp:1387-1405)
template <int MODE>
__device__ __forceinline__ void lin_body(const DevBatch& B, const int w, double* sm) {
  constexpr bool MARG = MODE != 0;
  constexpr bool PRIOR_ONLY = MODE == 2;
  const int tid = threadIdx.x, T = LIN_THREADS;
  TrState* tr = &B.tr[w];
  if (!MARG) {
    if (tr->status != 0 || tr->fresh_lin) return;
    count_active(B, 0);
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
  double* pacc = red + 18;         // maxP x 14: per-track sums over its factors  H_ll | g_l | W_s (6) | W_ext (6)
  int* invmap = (int*)(pacc + 14 * B.maxP);  // NC
  int* imuact = invmap + NC;       // 10
  int* tick = imuact + 10;         // 3 ticket counters of the point phase's ordered LDS commits

  const int nP = B.nP[w], nL = B.nL[w];
  // the marginalisation evaluates every block, constant or not (marginalization_factor.cpp:3-69)
  const bool ex_free = MARG || B.opt.estimate_extrinsic != 0;
  for (int i = tid; i < HV_DOUBLES + NV; i += T) sm[i] = 0.0;
  for (int i = tid; i < 14 * nP; i += T) pacc[i] = 0.0;
  for (int i = tid; i < 84; i += T) xp[i] = i < 77 ? B.pose[(size_t)w * 77 + i] : B.ex[(size_t)w * 7 + (i - 77)];
  for (int i = tid; i < 99; i += T) xs[i] = B.sb[(size_t)w * 99 + i];
  for (int i = tid; i < NC; i += T) invmap[i] = -1;
  if (tid < 3) tick[tid] = 0;
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
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
    // r = r0 + J0 dx and g = J0^T r = g0 + H dx (H = J0^T J0, g0 = J0^T r0 from k_prep) in ONE pass: eight lanes per row,
    // both rows' loads in flight together -- the second mat-vec used to start after the first (and a barrier) and paid
    // the global latency of its column reads all over again
    const double* Hp = B.pr_H + (size_t)w * B.prS;
    for (int r = tid >> 3; r < n; r += T >> 3) {
      const int sub = tid & 7;
      double s = 0, sg = 0;
#pragma unroll 4
      for (int c = sub; c < n; c += 8) {
        const double dxc = prdx[c];
        s += J0[(size_t)r * n + c] * dxc;
        sg += Hp[(size_t)r * n + c] * dxc;
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      sg += __shfl_xor(sg, 1, 64); sg += __shfl_xor(sg, 2, 64); sg += __shfl_xor(sg, 4, 64);
      if (sub == 0) {
        s += B.pr_r0[(size_t)w * MAXPN + r];
        prr[r] = s;
        prg[r] = sg + B.pr_g0[(size_t)w * MAXPN + r];
        cost += 0.5 * s * s;
      }
    }
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
  // zero the W rows first where some slot is not written by a factor (shorter tracks, erased lines, the start-frame
  // slot of the lines in the marginalisation pass); uniform track lengths need no fill in the solve
  const int WS = B.WS;
  if (MARG || B.wfill) {
    // (the marginalisation pass only ever reads the rows of the tracks that start in frame 0)
    double* Wp0 = B.Wp + (size_t)w * B.maxP * WS;
    for (int i = tid; i < nP * WS; i += T)
      if (!MARG || B.pt_start[(size_t)w * B.maxP + i / WS] == 0) Wp0[i] = 0.0;
    double* Wl0 = B.Wl + (size_t)w * B.maxL * 4 * WS;
    for (int i = tid; i < nL * 4 * WS; i += T)
      if (!MARG || B.ln_start[(size_t)w * B.maxL + i / (4 * WS)] == 0) Wl0[i] = 0.0;
  }
  __syncthreads();

  VPL_STAMP(B, w, 23);
  // points: the work unit is (start frame s, chunk of <= 16 tracks that start there, observation index k) -- one QUARTER
  // of a wave.  Inside a quarter every lane works on the same pair of frames (s, j = s + k), so the six 6x6 blocks a
  // point factor touches are uniform over its 16 lanes and the sum of J^T [J | r] (19 x 19, J = [J_s J_j J_e]) over them
  // is a rank-32 update: the lanes stage their two Jacobian rows in LDS and the FP64 matrix cores reduce them (3 tiles of
  // 16x16, K = 4 per instruction), one pass per quarter; each lane then adds its accumulator entries to the LDS Hessian.
  // The factor math of the four quarters of a wave runs concurrently (different (s, j) per quarter), so a window of 200
  // tracks x 5 factors takes 3 rounds of the 8 waves.  Per-track sums over k (H_ll, g_l, W_s, W_ext) are LDS atomics into
  // pacc, written out once at the end of the phase.
  {
    const int wvi = tid >> 6, nwv = T >> 6;
    const int* plist = B.ps_list + (size_t)w * B.maxP;     // track ids sorted by start frame
    // the marginalisation pass linearises the tracks of start frame 0 only: their units come first in the table
    const int nU = PRIOR_ONLY ? 0 : (MARG ? B.pu_cnt0[w] : B.pu_cnt[w]);
    const int* pu = B.pu_tab + (size_t)w * B.maxPU * 4;    // (s, first index in plist, tracks, k) per unit
    double* stg = imuJ + wvi * (32 * STG_LD);              // this wave's staging tile: 32 rows x 20 (stride STG_LD)
    const int m16 = lane & 15, kk = lane >> 4;
    // Every accumulator of this phase is shared by all waves, and floating-point addition does not associate: the adds are
    // therefore committed IN WAVE ORDER: wave v of (round, quarter) waits for its ticket (tick[1]), adds its Hessian tile
    // (and, in the first quarter, the per-track sums of its four units), passes the ticket on.  The factor math and the
    // matrix-core reductions of the other waves go on meanwhile; what is serialised is ~25 LDS instructions per quarter.
    // All waves run the same number of rounds (empty units included) so that the ticket sequence is complete.
    const int nRounds = (nU + 4 * nwv - 1) / (4 * nwv);
    // Two commit chains run side by side: waves 0..3 add into Hv / gv, waves 4..7 into a second copy that lives in the
    // part of the staging region the MFMA tiles leave free; the copy is folded into Hv after the phase.  The per-track
    // sums have one chain over all eight waves (tick[0]).
    const int half = wvi >= nwv / 2 ? 1 : 0, wih = wvi - half * (nwv / 2);
    double* HvC = half ? imuJ + LIN_STAGE : Hv;
    double* gvC = half ? imuJ + LIN_STAGE + HV_DOUBLES : gv;
    if (half) for (int i = tid - T / 2; i < HV_DOUBLES + NV; i += T / 2) imuJ[LIN_STAGE + i] = 0.0;
    __syncthreads();
    auto ticket_wait = [&](int which, int seq) {
      while (__hip_atomic_load(&tick[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) {}
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto ticket_pass = [&](int which, int seq) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&tick[which], seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int round = 0; round < nRounds; ++round) {
      const int u0 = 4 * nwv * round + 4 * wvi;
      const int u = u0 + kk;
      const bool has = u < nU;
      const int s = has ? pu[4 * u] : 0, k = has ? pu[4 * u + 3] : 1, ucnt = has ? pu[4 * u + 2] : 0;
      const bool live = has && m16 < ucnt && !(MARG && s != 0) && !PRIOR_ONLY;
      const int p = live ? plist[pu[4 * u + 1] + m16] : 0;
      const size_t pi = (size_t)w * B.maxP + p;
      const int no = live ? B.pt_nobs[pi] : 0, off = live ? B.pt_off[pi] : 0;
      const bool act = live && k < no;
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
        cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
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
          Wrow[6 * k + a] = Jl[0] * Jj[a] + Jl[1] * Jj[6 + a];   // this (track, frame) entry has one writer
        }
      }
      // ---- per quarter: reduction of [Js Jj Je r]^T [Js Jj Je r] over its 16 lanes on the matrix cores ----
      const unsigned long long actmask = __ballot(act);
#pragma unroll 1
      for (int qq = 0; qq < 4; ++qq) {
        const unsigned qmask = (unsigned)((actmask >> (16 * qq)) & 0xffffull);
        const int seq = (round * 4 + qq) * (nwv / 2) + wih;
        if (qq == 0) {              // per-track sums over k of the wave's four units (a track's factors sit in different
          ticket_wait(0, round * nwv + wvi);   // units, i.e. in different waves): one chain over all waves
          if (act) {
            double* pa = pacc + 14 * p;        // (two quarters of a wave may hold the same track with different k: atomic adds)
#pragma unroll
            for (int a = 0; a < 14; ++a) lds_add(&pa[a], pv[a]);
          }
          ticket_pass(0, round * nwv + wvi);
        }
        if (qmask == 0) {           // uniform: no tile to add, the ticket still goes round
          ticket_wait(1 + half, seq);
          ticket_pass(1 + half, seq);
          continue;
        }
        const int sq = __builtin_amdgcn_readlane(s, 16 * qq), jq = __builtin_amdgcn_readlane(j, 16 * qq);
        const int rows = 2 * (32 - __builtin_clz(qmask));   // staged rows that can be non-zero
        __builtin_amdgcn_wave_barrier();
        if (kk == qq) {
          double* d0 = stg + (2 * m16) * STG_LD;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              d0[rr * STG_LD + a] = Ji[6 * rr + a];
              d0[rr * STG_LD + 6 + a] = Jj[6 * rr + a];
              d0[rr * STG_LD + 12 + a] = Je[6 * rr + a];
            }
            d0[rr * STG_LD + 18] = r[rr];
            d0[rr * STG_LD + 19] = 0.0;
          }
        }
        __builtin_amdgcn_wave_barrier();
        v4d_lin c00 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
        const int ksmax = (rows + 3) >> 2;
#pragma unroll 2
        for (int ks = 0; ks < ksmax; ++ks) {
          const double* row = stg + (4 * ks + kk) * STG_LD;
          const double lo = row[m16];
          const double hi = m16 < 4 ? row[16 + m16] : 0.0;
          c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, lo, c00, 0, 0, 0);
          c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, lo, c10, 0, 0, 0);
          c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, hi, c11, 0, 0, 0);
        }
        // accumulator entry (row a = kk + 4 v (+16), col b = m16 (+16)) -> LDS Hessian / gradient, in wave order
        ticket_wait(1 + half, seq);
        {
          const double v00[4] = {c00.x, c00.y, c00.z, c00.w}, v10[4] = {c10.x, c10.y, c10.z, c10.w},
                       v11[4] = {c11.x, c11.y, c11.z, c11.w};
          auto visof = [&](int a) { return a < 6 ? 6 * sq + a : (a < 12 ? 6 * jq + (a - 6) : 66 + (a - 12)); };
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int a0 = kk + 4 * v, bcol = m16;
            if (a0 >= bcol) lds_add(&HvC[hvi(visof(a0), visof(bcol))], v00[v]);
            const int a1 = 16 + kk + 4 * v;
            if (a1 < 18) lds_add(&HvC[hvi(visof(a1), visof(bcol))], v10[v]);
            else if (a1 == 18) lds_add(&gvC[visof(bcol)], v10[v]);
            const int b1c = 16 + m16;
            if (b1c < 18) {
              if (a1 < 18 && a1 >= b1c) lds_add(&HvC[hvi(visof(a1), visof(b1c))], v11[v]);
              else if (a1 == 18) lds_add(&gvC[visof(b1c)], v11[v]);
            }
          }
        }
        ticket_pass(1 + half, seq);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < HV_DOUBLES + NV; i += T) sm[i] += imuJ[LIN_STAGE + i];   // second commit chain's copy (Hv | gv are contiguous)
  for (int p = tid; p < nP; p += T) {   // per-track sums out of LDS
    const size_t pi = (size_t)w * B.maxP + p;
    const double* pa = pacc + 14 * p;
    double* Wrow = B.Wp + pi * WS;
    B.Hpp[pi] = pa[0];
    B.gp[pi] = pa[1];
#pragma unroll
    for (int a = 0; a < 6; ++a) { Wrow[a] = pa[2 + a]; Wrow[WS - 6 + a] = pa[8 + a]; }
  }
  __syncthreads();   // staging space is handed over to the IMU / line phases
  VPL_STAMP(B, w, 24);
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
    const int* ltab = B.ll_tab + (size_t)w * B.llSlots;
    const int npass = B.ll_np[w];
    const int krow = lane / NLW;                         // observation index of this lane's slot
    for (int pass = 0; pass < npass; ++pass) {
      const int o = (!PRIOR_ONLY && krow < KL) ? ltab[pass * T + tid] : -1;
      const bool inb = o >= 0;
      const int l = inb ? B.lo_ln[(size_t)w * B.maxLO + o] : 0;
      const size_t li = (size_t)w * B.maxL + l;
      const int s = B.ln_start[li], off = B.ln_off[li];
      const int k = inb ? o - off : 0, j = s + k;
      // MARG: start-frame obs skipped (estimator.cpp:1322-1326), erased lines are no longer in f_manager.linefeature
      const bool act = inb && (!MARG || (s == 0 && k >= 1 && !B.ln_removed[li]));
      const double* ob = B.ln_obs + ((size_t)w * B.maxLO + (inb ? o : 0)) * 8;
      LineCtx c;
      if (act) c = line_ctx(xp + 7 * j, xe, B.orth + li * 4);
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
  __syncthreads();
  if (tid < 10) {
    const int j = tid + 1;
    const DevPreint& dp = B.pre[(size_t)w * NF + j];
    bool act = MARG ? (j == 1 && dp.sum_dt < 10.0) : !(dp.sum_dt > 10.0);   // estimator.cpp:1088, :1261
    if (PRIOR_ONLY) act = false;
    imuact[tid] = act ? 1 : 0;
    double* J = imuJ + 450 * tid;
    if (act) {
      PreInt p = load_preint(dp);
      imu_residual_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm, imur + 15 * tid);
      ImuJac JB = imu_jacobian_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm);
      imu_jac_dense(JB, J);
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


  VPL_STAMP(B, w, 20);
  // ---- assemble the packed cam Hessian and gradient in HBM -----------------------------------
  double* Hout = B.Hcc + (size_t)w * NCP;
  const double* pH = B.pr_H + (size_t)w * B.prS;
  // Three passes over the packed lower triangle instead of one that decodes every index and asks every entry for all
  // three sources: (A) rows to waves, columns to lanes -- the visual block, coalesced stores, no index decoding;
  // (B) the 3570 entries the IMU factors touch (11 diagonal 15x15 blocks, 10 sub-diagonal ones), each summing its one or
  // two J^T J terms; (C) the n (n + 1) / 2 entries of the prior, all of its loads in one batch.  (B) and (C) add to what
  // (A) stored: same workgroup, a barrier in between.
  {
    const int lane2 = tid & 63, wv2 = tid >> 6;
    for (int r = wv2; r < NC; r += T >> 6) {
      const int vr = cam2vis(r);
      const bool dead = !ex_free && r >= 165;
      double* row = Hout + (size_t)r * (r + 1) / 2;
      for (int c = lane2; c <= r; c += 64) {
        double v = 0.0;
        if (vr >= 0 && !dead) {
          const int vc = cam2vis(c);
          if (vc >= 0) v = Hv[hvi(vr, vc)];
        }
        row[c] = v;
      }
    }
  }
  __syncthreads();
  for (int item = tid; item < 11 * 120 + 10 * 225; item += T) {
    int fr, a, b;   // entry (15 fr + a, 15 fc + b)
    bool diag;
    if (item < 11 * 120) {
      fr = item / 120;
      tri_decode(item - 120 * fr, a, b);
      diag = true;
    } else {
      const int e = item - 11 * 120;
      fr = 1 + e / 225;
      const int q = e - 225 * (fr - 1);
      a = q / 15; b = q - 15 * a;
      diag = false;
    }
    double v = 0.0;
    const int t0 = fr - 1;
    if (t0 >= 0 && imuact[t0]) {           // factor (fr-1, fr): rows 15..29 of its 30 columns are frame fr
      const double* J = imuJ + 450 * t0;
      const int ja = 15 + a, jb = diag ? 15 + b : b;
      double s2 = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s2 += J[k * 30 + ja] * J[k * 30 + jb];
      v += s2;
    }
    if (diag && fr < 10 && imuact[fr]) {   // factor (fr, fr+1): columns 0..14 are frame fr
      const double* J = imuJ + 450 * fr;
      double s2 = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s2 += J[k * 30 + a] * J[k * 30 + b];
      v += s2;
    }
    const int r = 15 * fr + a, c = diag ? 15 * fr + b : 15 * (fr - 1) + b;
    if (v != 0.0) Hout[(size_t)r * (r + 1) / 2 + c] += v;
  }
  if (n > 0) {
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
      __syncthreads();   // (uniform trip count) the IMU pass has finished with the entries the prior shares
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
  }
}
template <int MODE>
__global__ __launch_bounds__(LIN_THREADS) void k_lin(DevBatch B) {
  extern __shared__ double sm[];
  lin_body<MODE>(B, MODE == 0 ? ordered_window(B) : (int)blockIdx.x, sm);
}

inline size_t lin_smem(int maxP, int maxL) {
  return (size_t)(HV_DOUBLES + NV + 84 + 99 + lin_stage_doubles(maxL) + 3 * MAXPN + 18 + 14 * maxP) * sizeof(double) + (size_t)(NC + 16) * sizeof(int);
}

}  // namespace vpl
