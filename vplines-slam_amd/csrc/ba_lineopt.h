// Line-map maintenance that runs before every main solve (estimator.cpp:635-638):
//   k_triangulate : FeatureManager::triangulateLine (feature_manager.cpp:413-563), one lane per line
//   k_line_opt    : Estimator::onlyLineOpt (estimator.cpp:950-1039) -- poses and extrinsic constant, line factors only,
//                   CauchyLoss(1.0), ceres LEVENBERG_MARQUARDT + DENSE_SCHUR.  With the cameras constant every line is its own
//                   4x4 block of the normal equations; what couples the lines is the minimizer itself (one radius, one
//                   accept/reject decision on the total cost).  One workgroup per window, one lane per line; the
//                   TrustRegionMinimizer loop runs inside the kernel (all iterations, block reductions for the scalars).
// The world orth vectors come from k_prep, setLineOrth + removeLineOutlier are k_gauge's (gauge transform = identity
// here: the poses did not move).
#pragma once
#include "ba_common.h"

namespace vpl {

// pi_from_ppp (line_geometry.cpp:134-139)
VPL_HD void pi_from_ppp(V3 x1, V3 x2, V3 x3, double* pi) {
  const V3 n = cross(x1 - x3, x2 - x3);
  pi[0] = n.x; pi[1] = n.y; pi[2] = n.z;
  pi[3] = -dot(x3, cross(x1, x2));
}

__global__ __launch_bounds__(128) void k_triangulate(DevBatch B) {
  const int w = blockIdx.x;
  const int nL = B.nL[w];
  const double* pose = B.pose + (size_t)w * 77;
  const double* ex = B.ex + (size_t)w * 7;
  const M3 ric = qmat(qnormalized(qpose(ex)));
  const V3 tic{ex[0], ex[1], ex[2]};
  for (int l = threadIdx.x; l < nL; l += blockDim.x) {
    const size_t li = (size_t)w * B.maxL + l;
    if (B.ln_tri[li]) continue;
    const int s = B.ln_start[li], no = B.ln_nobs[li];
    const double* ob0 = B.ln_obs + ((size_t)w * B.maxLO + B.ln_off[li]) * 8;
    const double* x0 = pose + 7 * s;
    const M3 Rs0 = qmat(qnormalized(qpose(x0)));
    const V3 t0 = V3{x0[0], x0[1], x0[2]} + mul(Rs0, tic);
    const M3 R0 = mul(Rs0, ric);
    double pii[4];
    pi_from_ppp(V3{ob0[0], ob0[1], 1.0}, V3{ob0[2], ob0[3], 1.0}, V3{0, 0, 0}, pii);
    V3 ni{pii[0], pii[1], pii[2]};
    ni = ni * (1.0 / norm(ni));
    double min_cos = 1.0;
    V3 tij{0, 0, 0};
    M3 Rij = R0;
    int kj = -1;
    for (int k = 1; k < no; ++k) {
      const double* xj = pose + 7 * (s + k);
      const M3 Rsj = qmat(qnormalized(qpose(xj)));
      const V3 t1 = V3{xj[0], xj[1], xj[2]} + mul(Rsj, tic);
      const M3 R1 = mul(Rsj, ric);
      const V3 t = mulT(R0, t1 - t0);
      const M3 R = mulTA(R0, R1);
      const double* ob = ob0 + 8 * k;
      const V3 p3 = mul(R, V3{ob[0], ob[1], 1.0}) + t, p4 = mul(R, V3{ob[2], ob[3], 1.0}) + t;
      double pij[4];
      pi_from_ppp(p3, p4, t, pij);
      V3 nj{pij[0], pij[1], pij[2]};
      nj = nj * (1.0 / norm(nj));
      const double c = dot(ni, nj);
      if (c < min_cos) { min_cos = c; tij = t; Rij = R; kj = k; }
    }
    if (min_cos > 0.998 || kj < 0) continue;
    const double* ob = ob0 + 8 * kj;
    const V3 p3 = mul(Rij, V3{ob[0], ob[1], 1.0}) + tij, p4 = mul(Rij, V3{ob[2], ob[3], 1.0}) + tij;
    double pij[4];
    pi_from_ppp(p3, p4, tij, pij);
    // pipi_plk (line_geometry.cpp:142-148): dp = pi1 pi2^T - pi2 pi1^T
    auto dp = [&](int i, int j) { return pii[i] * pij[j] - pij[i] * pii[j]; };
    double* pl = B.plk + li * 6;
    pl[0] = dp(0, 3); pl[1] = dp(1, 3); pl[2] = dp(2, 3);
    pl[3] = -dp(1, 2); pl[4] = dp(0, 2); pl[5] = -dp(0, 1);
    B.ln_tri[li] = 1;
  }
}

// FeatureManager::triangulate (feature_manager.cpp:565-621), one lane per track whose inverse depth is negative (the
// reference's estimated_depth = -1): DLT rows of every observation in the start camera frame, one-sided Jacobi SVD of
// the 2m x 4 matrix (kept in per-lane scratch: this runs once per new track, not in the solve loop).
__global__ __launch_bounds__(128) void k_triangulate_points(DevBatch B, double init_depth) {
  const int w = blockIdx.x;
  const int nP = B.nP[w];
  const double* pose = B.pose + (size_t)w * 77;
  const double* ex = B.ex + (size_t)w * 7;
  const M3 ric = qmat(qnormalized(qpose(ex)));
  const V3 tic{ex[0], ex[1], ex[2]};
  for (int p = threadIdx.x; p < nP; p += blockDim.x) {
    const size_t pi = (size_t)w * B.maxP + p;
    if (!(B.invd[pi] < 0.0)) continue;
    const int s = B.pt_start[pi], no = B.pt_nobs[pi];
    const double* o0 = B.pt_obs + ((size_t)w * B.maxPO + B.pt_off[pi]) * 3;
    const double* x0 = pose + 7 * s;
    const M3 Rs0 = qmat(qnormalized(qpose(x0)));
    const V3 t0 = V3{x0[0], x0[1], x0[2]} + mul(Rs0, tic);
    const M3 R0 = mul(Rs0, ric);
    double A[2 * NF * 4];
    const int m = 2 * no;
    for (int k = 0; k < no; ++k) {
      const double* xj = pose + 7 * (s + k);
      const M3 Rsj = qmat(qnormalized(qpose(xj)));
      const V3 t1 = V3{xj[0], xj[1], xj[2]} + mul(Rsj, tic);
      const M3 R1 = mul(Rsj, ric);
      const V3 t = mulT(R0, t1 - t0);
      const M3 R = mulTA(R0, R1);
      const M3 Rt = transpose(R);
      const V3 mt = -mul(Rt, t);
      const double* ob = o0 + 3 * k;
      const double rn = sqrt(ob[0] * ob[0] + ob[1] * ob[1] + ob[2] * ob[2]);
      const double f0 = ob[0] / rn, f1 = ob[1] / rn, f2 = ob[2] / rn;
      for (int c = 0; c < 4; ++c) {
        const double P0 = c < 3 ? Rt.m[c] : mt.x, P1 = c < 3 ? Rt.m[3 + c] : mt.y, P2 = c < 3 ? Rt.m[6 + c] : mt.z;
        A[(2 * k) * 4 + c] = f0 * P2 - f2 * P0;
        A[(2 * k + 1) * 4 + c] = f1 * P2 - f2 * P1;
      }
    }
    double V[16];
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
      bool rotated = false;
      for (int pp = 0; pp < 3; ++pp)
        for (int q = pp + 1; q < 4; ++q) {
          double al = 0, be = 0, ga = 0;
          for (int r = 0; r < m; ++r) { al += A[r * 4 + pp] * A[r * 4 + pp]; be += A[r * 4 + q] * A[r * 4 + q]; ga += A[r * 4 + pp] * A[r * 4 + q]; }
          if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
          rotated = true;
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
          for (int r = 0; r < m; ++r) {
            const double a = A[r * 4 + pp], b = A[r * 4 + q];
            A[r * 4 + pp] = c * a - sn * b;
            A[r * 4 + q] = sn * a + c * b;
          }
          for (int r = 0; r < 4; ++r) {
            const double a = V[r * 4 + pp], b = V[r * 4 + q];
            V[r * 4 + pp] = c * a - sn * b;
            V[r * 4 + q] = sn * a + c * b;
          }
        }
      if (!rotated) break;
    }
    int cmin = 0;
    double smin = 0;
    for (int c = 0; c < 4; ++c) {
      double s2 = 0;
      for (int r = 0; r < m; ++r) s2 += A[r * 4 + c] * A[r * 4 + c];
      if (c == 0 || s2 < smin) { smin = s2; cmin = c; }   // (ties resolved to the first column, as the oracle's sort)
    }
    double depth = V[2 * 4 + cmin] / V[3 * 4 + cmin];
    if (depth < 0.1) depth = init_depth;
    B.invd[pi] = 1.0 / depth;
  }
}

// FeatureManager::removeBackShiftDepth (feature_manager.cpp:800-874), the arithmetic: one lane per track that started in
// the marginalised frame and survives.  fr: per window pose[0], pose[1] (before the shift) and ex_pose, 21 doubles.
// Points pd[i] = {u, v, 1, inv_depth} -> inv_depth in the next frame; lines ld[i] = plk -> plk_to_pose(plk, Rji, tji).
__global__ __launch_bounds__(256) void k_slide_shift(const double* __restrict__ fr, int nPts, const int* __restrict__ pw,
                                                     double* __restrict__ pd, int nLns, const int* __restrict__ lw,
                                                     double* __restrict__ ld, double init_depth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPts + nLns) return;
  const bool is_pt = i < nPts;
  const int k = is_pt ? i : i - nPts;
  const double* f = fr + (size_t)(is_pt ? pw[k] : lw[k]) * 21;
  const M3 ric = qmat(qnormalized(qpose(f + 14)));
  const V3 tic{f[14], f[15], f[16]};
  const M3 Rs0 = qmat(qnormalized(qpose(f))), Rs1 = qmat(qnormalized(qpose(f + 7)));
  const M3 marg_R = mul(Rs0, ric), new_R = mul(Rs1, ric);
  const V3 marg_P = V3{f[0], f[1], f[2]} + mul(Rs0, tic), new_P = V3{f[7], f[8], f[9]} + mul(Rs1, tic);
  if (is_pt) {
    double* d = pd + (size_t)k * 4;
    const double depth = 1.0 / d[3];
    const V3 pts_i = V3{d[0], d[1], d[2]} * depth;
    const V3 w_pts_i = mul(marg_R, pts_i) + marg_P;
    const V3 pts_j = mulT(new_R, w_pts_i - new_P);
    d[3] = 1.0 / (pts_j.z > 0 ? pts_j.z : init_depth);
  } else {
    double* d = ld + (size_t)k * 6;
    const M3 Rji = mulTA(new_R, marg_R);
    const V3 tji = mulT(new_R, marg_P - new_P);
    const Plk L = plk_to_pose(Plk{V3{d[0], d[1], d[2]}, V3{d[3], d[4], d[5]}}, Rji, tji);
    d[0] = L.n.x; d[1] = L.n.y; d[2] = L.n.z; d[3] = L.v.x; d[4] = L.v.y; d[5] = L.v.z;
  }
}

// ---- onlyLineOpt ------------------------------------------------------------------------------------------
constexpr int LOPT_THREADS = 256;

// cost (with the Cauchy loss) and, if H, the corrected normal equations of one line at orth `o`:
// H (packed lower 4x4, 10), g (4).  ceres Corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho').
// G lanes share a line: lane `sub` takes the observations sub, sub + G, ... and the sums are folded over the lanes with xor
// shuffles, so that every lane of the line holds the same cost, H and g (a + b == b + a bit for bit).
template <int G>
__device__ inline double lopt_eval(const DevBatch& B, int w, size_t li, const double* xp, const double* xe, const double* o,
                                   double* H, double* g, int sub) {
  const int s = B.ln_start[li], no = B.ln_nobs[li];
  const double* ob0 = B.ln_obs + ((size_t)w * B.maxLO + B.ln_off[li]) * 8;
  double cost = 0.0;
  if (H) {
#pragma unroll
    for (int k = 0; k < 10; ++k) H[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = 0.0;
  }
  for (int k = sub; k < no; k += G) {
    const LineCtx c = line_ctx(xp + 7 * (s + k), xe, o);
    double r[2], jel[6];
    line_factor_res(c, ob0 + 8 * k, B.opt.sqrt_info_line, r, H ? jel : nullptr);
    const double sq = r[0] * r[0] + r[1] * r[1];
    const double sum = 1.0 + sq;              // CauchyLoss(1.0): b = c = 1
    cost += 0.5 * log(sum);
    if (H) {
      double rho1 = 1.0 / sum;
      if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
      const double sc = sqrt(rho1);
      double Jp[12], Je[12], Jo[8];
      line_chain_jac(c, jel, 0, Jp, Je, Jo);
      r[0] *= sc; r[1] *= sc;
#pragma unroll
      for (int q = 0; q < 8; ++q) Jo[q] *= sc;
      int t = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        g[a] += Jo[a] * r[0] + Jo[4 + a] * r[1];
#pragma unroll
        for (int b = 0; b <= a; ++b, ++t) H[t] += Jo[a] * Jo[b] + Jo[4 + a] * Jo[4 + b];
      }
    }
  }
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    cost += __shfl_xor(cost, m, 64);
    if (H) {
#pragma unroll
      for (int k = 0; k < 10; ++k) H[k] += __shfl_xor(H[k], m, 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] += __shfl_xor(g[k], m, 64);
    }
  }
  return cost;
}

__device__ inline double block_max(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = red[0];
    for (int i = 1; i < nw; ++i) s = fmax(s, red[i]);
    red[16] = s;
  }
  __syncthreads();
  return red[16];
}

// G = lanes per line (1, 2 or 4; G * max_lines <= LOPT_THREADS): the latency of a call is the chain of factor evaluations of
// one line, 2 x up to 11 per iteration on one lane with G = 1 (180 us for a window of the sequence test; round 4).
template <int G>
__global__ __launch_bounds__(LOPT_THREADS) void k_line_opt(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x;
  const int ln = tid / G, sub = tid % G;
  const int nL = B.nL[w];
  __shared__ double xp[84], red[18];
  TrState* tr = &B.tr[w];
  if (nL < 4) {   // "if (feature_index < 3) return;" estimator.cpp:1019-1022
    if (tid == 0) { tr->iter = 0; tr->num_successful = 0; tr->status = 3; tr->initial_cost = 0.0; tr->x_cost = 0.0; }
    return;
  }
  for (int i = tid; i < 84; i += LOPT_THREADS) xp[i] = i < 77 ? B.pose[(size_t)w * 77 + i] : B.ex[(size_t)w * 7 + (i - 77)];
  __syncthreads();
  const double* xe = xp + 77;
  const bool live = ln < nL;
  const bool lead = live && sub == 0;     // the lane whose copy of the line's scalars enters the block sums
  const size_t li = (size_t)w * B.maxL + (live ? ln : 0);
  double x[4] = {0, 0, 0, 0}, H[10], g[4], sc[4] = {1, 1, 1, 1}, diag[4] = {1, 1, 1, 1};
  if (live)
    for (int k = 0; k < 4; ++k) x[k] = B.orth[li * 4 + k];
  for (int k = 0; k < 10; ++k) H[k] = 0.0;
  for (int k = 0; k < 4; ++k) g[k] = 0.0;

  // ---- iteration 0: evaluate, Jacobi scaling, gradient max norm ----
  double c0 = live ? lopt_eval<G>(B, w, li, xp, xe, x, H, g, sub) : 0.0;
  double x_cost = block_sum(lead ? c0 : 0.0, red);
  const double initial_cost = x_cost;
  if (live)
    for (int a = 0; a < 4; ++a) sc[a] = 1.0 / (1.0 + sqrt(H[tri(a, a)]));
  auto grad_max = [&]() {
    double m = 0.0;
    if (live) {
      double ng[4] = {-g[0], -g[1], -g[2], -g[3]}, px[4];
      line_orth_plus(x, ng, px);
      for (int a = 0; a < 4; ++a) m = fmax(m, fabs(x[a] - px[a]));
    }
    return block_max(m, red);
  };
  double gmax = grad_max();
  double xn2 = 0.0;
  if (lead) for (int a = 0; a < 4; ++a) xn2 += x[a] * x[a];
  double x_norm = sqrt(block_sum(xn2, red));

  // ceres LevenbergMarquardtStrategy state (levenberg_marquardt_strategy.cc)
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int iter = 0, num_successful = 0, num_invalid = 0, status = 0;   // 1 convergence, 2 failure, 3 max iterations
  bool last_successful = true;
  for (;;) {
    if (iter >= B.opt.num_iterations) { status = 3; break; }
    if (last_successful && gmax <= 1e-10) { status = 1; break; }
    if (radius <= 1e-32) { status = 1; break; }
    ++iter;
    // ---- ComputeStep: (S H S + diag / radius) y = S g per line ----
    double step[4] = {0, 0, 0, 0};
    double mcc = 0.0;
    int fail = 0;
    if (live) {
      double Hs[10], gs[4], A[10];
      int t = 0;
      for (int a = 0; a < 4; ++a) {
        gs[a] = sc[a] * g[a];
        for (int b = 0; b <= a; ++b, ++t) Hs[t] = sc[a] * H[t] * sc[b];
      }
      if (!reuse_diagonal)
        for (int a = 0; a < 4; ++a) diag[a] = fmin(fmax(Hs[tri(a, a)], 1e-6), 1e32);
      for (int k = 0; k < 10; ++k) A[k] = Hs[k];
      for (int a = 0; a < 4; ++a) {
        const double lm = sqrt(diag[a] / radius);
        A[tri(a, a)] += lm * lm;
      }
      // Cholesky of the 4x4 block (ceres InvertPSDMatrix / LLT)
      for (int j = 0; j < 4 && !fail; ++j) {
        double d = A[tri(j, j)];
        for (int k = 0; k < j; ++k) d -= A[tri(j, k)] * A[tri(j, k)];
        if (!(d > 0.0)) { fail = 1; break; }
        d = sqrt(d);
        A[tri(j, j)] = d;
        for (int i = j + 1; i < 4; ++i) {
          double s2 = A[tri(i, j)];
          for (int k = 0; k < j; ++k) s2 -= A[tri(i, k)] * A[tri(j, k)];
          A[tri(i, j)] = s2 / d;
        }
      }
      if (!fail) {
        double y[4];
        for (int i = 0; i < 4; ++i) {
          double s2 = gs[i];
          for (int k = 0; k < i; ++k) s2 -= A[tri(i, k)] * y[k];
          y[i] = s2 / A[tri(i, i)];
        }
        for (int i = 3; i >= 0; --i) {
          double s2 = y[i];
          for (int k = i + 1; k < 4; ++k) s2 -= A[tri(k, i)] * y[k];
          y[i] = s2 / A[tri(i, i)];
        }
        for (int a = 0; a < 4; ++a) {
          step[a] = -y[a];
          if (!isfinite(step[a])) fail = 1;
        }
        // model_cost_change = -(step^T gs + 1/2 step^T Hs step)   (= -model_residuals.(residuals + model_residuals/2))
        double sHs = 0.0, sg = 0.0;
        for (int a = 0; a < 4; ++a) {
          sg += step[a] * gs[a];
          for (int b = 0; b < 4; ++b) sHs += step[a] * Hs[a >= b ? tri(a, b) : tri(b, a)] * step[b];
        }
        mcc = -(sg + 0.5 * sHs);
      }
    }
    const double anyfail = block_sum(lead ? (double)fail : 0.0, red);
    const double model_cost_change = block_sum(lead ? mcc : 0.0, red);
    reuse_diagonal = true;   // ComputeStep (a failed solve leaves it as it was: the next attempt recomputes nothing else)
    if (anyfail > 0.0 || !(model_cost_change > 0.0)) {
      if (anyfail > 0.0) reuse_diagonal = false;
      last_successful = false;
      if (++num_invalid >= 5) { status = 2; --iter; break; }
      continue;   // StepIsInvalid() is empty for this strategy
    }
    num_invalid = 0;
    // ---- candidate, its cost, tolerances ----
    double cand[4] = {0, 0, 0, 0}, cc = 0.0, sn2 = 0.0;
    if (live) {
      double delta[4];
      for (int a = 0; a < 4; ++a) delta[a] = step[a] * sc[a];
      line_orth_plus(x, delta, cand);
      cc = lopt_eval<G>(B, w, li, xp, xe, cand, nullptr, nullptr, sub);
      for (int a = 0; a < 4; ++a) sn2 += (x[a] - cand[a]) * (x[a] - cand[a]);
    }
    const double cand_cost = block_sum(lead ? cc : 0.0, red);
    const double step_norm = sqrt(block_sum(lead ? sn2 : 0.0, red));
    // a tolerance fires before the iteration is recorded (ceres: the summary of the terminating iteration is never pushed)
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { status = 1; --iter; break; }
    if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { status = 1; --iter; break; }
    const double rho = (x_cost - cand_cost) / model_cost_change;
    if (rho > 1e-3) {
      if (live) {
        for (int a = 0; a < 4; ++a) x[a] = cand[a];
        c0 = lopt_eval<G>(B, w, li, xp, xe, x, H, g, sub);
      } else {
        c0 = 0.0;
      }
      x_cost = block_sum(lead ? c0 : 0.0, red);
      xn2 = 0.0;
      if (lead) for (int a = 0; a < 4; ++a) xn2 += x[a] * x[a];
      x_norm = sqrt(block_sum(xn2, red));
      gmax = grad_max();
      last_successful = true;
      ++num_successful;
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0));
      radius = fmin(1e16, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      last_successful = false;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  if (lead)
    for (int k = 0; k < 4; ++k) B.orth[li * 4 + k] = x[k];
  if (tid == 0) {
    tr->iter = iter;
    tr->num_successful = num_successful;
    tr->status = status;
    tr->initial_cost = initial_cost;
    tr->x_cost = x_cost;
  }
}

}  // namespace vpl
