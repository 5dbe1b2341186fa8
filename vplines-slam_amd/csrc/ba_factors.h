// Batched single-factor evaluators behind the Ceres-style CostFunction::Evaluate ABI
// (include/vplines_ba.h).  One lane per factor; parameters arrive packed per factor in the
// reference's block order; Jacobians leave in the reference's row-major global layout with
// the 7th pose column written as zero (projection_factor.cpp:81-88).
#pragma once
#include "ba_common.h"
#include "vpl_preint.h"

namespace vpl {

__device__ __forceinline__ void store_pose_jac(double* J7, const double* J6, int rows) {
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < 6; ++c) J7[7 * r + c] = J6[6 * r + c];
    J7[7 * r + 6] = 0.0;
  }
}

__global__ void k_eval_projection(int n, const double* params, const double* pts, double sqrt_info, double* res,
                                  double* jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = params + (size_t)i * 22;
  const double* o = pts + (size_t)i * 6;
  double r[2], Ji[12], Jj[12], Je[12], Jl[2];
  projection_factor(p, p + 7, p + 14, p[21], V3{o[0], o[1], o[2]}, V3{o[3], o[4], o[5]}, sqrt_info, r, jac != nullptr,
                    Ji, Jj, Je, Jl);
  res[2 * i] = r[0];
  res[2 * i + 1] = r[1];
  if (jac) {
    double* J = jac + (size_t)i * 44;
    store_pose_jac(J, Ji, 2);
    store_pose_jac(J + 14, Jj, 2);
    store_pose_jac(J + 28, Je, 2);
    J[42] = Jl[0];
    J[43] = Jl[1];
  }
}

template <int SEL>
__global__ void k_eval_line(int n, const double* params, const double* obs, double sqrt_info, double* res, double* jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = params + (size_t)i * 18;
  LineCtx c = line_ctx(p, p + 7, p + 14);
  double r[2], jel[6];
  if (SEL == 0) line_factor_res(c, obs + (size_t)i * 4, sqrt_info, r, jel);
  else vp_factor_res(c, obs + (size_t)i * 3, sqrt_info, r, jel);
  res[2 * i] = r[0];
  res[2 * i + 1] = r[1];
  if (jac) {
    double Jp[12], Je[12], Jo[8];
    line_chain_jac(c, jel, SEL, Jp, Je, Jo);
    double* J = jac + (size_t)i * 36;
    store_pose_jac(J, Jp, 2);
    store_pose_jac(J + 14, Je, 2);
    for (int k = 0; k < 8; ++k) J[28 + k] = Jo[k];
  }
}

// sqrt_info from the covariance, as k_prep does (thread-local Gauss-Jordan + Cholesky)
__device__ void imu_sqrt_info(const double* cov, double* S /*225 out, upper*/) {
  double a[225], inv[225];
  for (int k = 0; k < 225; ++k) { a[k] = cov[k]; inv[k] = (k / 15 == k % 15) ? 1.0 : 0.0; }
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
    for (int j = 0; j < 15; ++j) S[i * 15 + j] = (j >= i) ? a[j * 15 + i] : 0.0;
}

__global__ void k_eval_imu(int n, const double* params, const DevPreint* pre, double g_norm, double* res, double* jac,
                           double* scratch /* n * (225 + 450) */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = params + (size_t)i * 32;
  double* S = scratch + (size_t)i * 675;
  double* J = S + 225;
  imu_sqrt_info(pre[i].cov, S);
  PreInt P = load_preint(pre[i]);
  double r[15];
  imu_residual_raw(P, p, p + 7, p + 16, p + 23, g_norm, r);
  for (int a = 0; a < 15; ++a) {
    double s = 0;
    for (int k = a; k < 15; ++k) s += S[a * 15 + k] * r[k];
    res[(size_t)i * 15 + a] = s;
  }
  if (jac) {
    for (int k = 0; k < 450; ++k) J[k] = 0.0;
    ImuJac JB = imu_jacobian_raw(P, p, p + 7, p + 16, p + 23, g_norm);
    imu_jac_dense(JB, J);
    // whitened, scattered into 15x7, 15x9, 15x7, 15x9 row-major blocks
    double* o = jac + (size_t)i * 480;
    const int off[4] = {0, 105, 240, 345}, gs[4] = {7, 9, 7, 9}, c0[4] = {0, 6, 15, 21}, ls[4] = {6, 9, 6, 9};
    for (int b = 0; b < 4; ++b)
      for (int a = 0; a < 15; ++a) {
        for (int c = 0; c < gs[b]; ++c) {
          double s = 0;
          if (c < ls[b])
            for (int k = a; k < 15; ++k) s += S[a * 15 + k] * J[k * 30 + c0[b] + c];
          o[off[b] + a * gs[b] + c] = s;
        }
      }
  }
}

__global__ void k_eval_pose_plus(int n, const double* x, const double* d, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pose_plus(x + 7 * (size_t)i, d + 6 * (size_t)i, out + 7 * (size_t)i);
}
__global__ void k_eval_orth_plus(int n, const double* x, const double* d, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) line_orth_plus(x + 4 * (size_t)i, d + 4 * (size_t)i, out + 4 * (size_t)i);
}

// MarginalizationFactor::Evaluate for one prior (block-cooperative): residuals n, Jacobian blocks
__global__ __launch_bounds__(256) void k_eval_prior(int n, int nb, const int* kind, const int* idx, const double* x0, const double* J0,
                             const double* r0, const double* params, double* res, double* jac) {
  __shared__ double dx[MAXPN];
  __shared__ int poff[MAXPB + 1];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int o = 0;
    for (int b = 0; b < nb; ++b) { poff[b] = o; o += kind[b] == 1 ? 9 : 7; }
    poff[nb] = o;
  }
  __syncthreads();
  if (tid < nb) {
    double d[9];
    prior_block_dx(kind[tid], params + poff[tid], x0 + 9 * tid, d);
    const int ls = kind[tid] == 1 ? 9 : 6;
    for (int k = 0; k < ls; ++k) dx[idx[tid] + k] = d[k];
  }
  __syncthreads();
  for (int r = tid; r < n; r += blockDim.x) {
    double s = r0[r];
    for (int c = 0; c < n; ++c) s += J0[(size_t)r * n + c] * dx[c];
    res[r] = s;
  }
  if (jac) {
    // block b: n x gs row-major at offset n * poff[b]
    for (int b = 0; b < nb; ++b) {
      const int gs = kind[b] == 1 ? 9 : 7, ls = kind[b] == 1 ? 9 : 6;
      double* o = jac + (size_t)n * poff[b];
      for (int it = tid; it < n * gs; it += blockDim.x) {
        const int r = it / gs, c = it % gs;
        o[it] = c < ls ? J0[(size_t)r * n + idx[b] + c] : 0.0;
      }
    }
  }
}

// IntegrationBase over raw samples: one lane per keyframe interval, big matrices in scratch
__global__ void k_preintegrate(int n, const int* offset, const int* nsamples, const double* samples, const double* acc0,
                               const double* gyr0, const double* lba, const double* lbg, double an2, double gn2,
                               double aw2, double gw2, DevPreint* out, double* scratch /* n * (225*2 + 225+270+225) */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double* base = scratch + (size_t)i * 1170;
  PreintState* st = nullptr;
  (void)st;
  // state lives in scratch to keep the register file small
  double* Jm = base;
  double* Pm = base + 225;
  double* F = base + 450;
  double* Vv = base + 675;
  double* Tm = base + 945;
  PreintState s;
  s.dp = V3{0, 0, 0}; s.dv = V3{0, 0, 0}; s.dq = Q4{1, 0, 0, 0}; s.sum_dt = 0;
  for (int k = 0; k < 225; ++k) { s.J[k] = (k / 15 == k % 15) ? 1.0 : 0.0; s.P[k] = 0.0; }
  const double nz2[4] = {an2, gn2, aw2, gw2};
  V3 a0{acc0[3 * i], acc0[3 * i + 1], acc0[3 * i + 2]}, g0{gyr0[3 * i], gyr0[3 * i + 1], gyr0[3 * i + 2]};
  V3 ba{lba[3 * i], lba[3 * i + 1], lba[3 * i + 2]}, bg{lbg[3 * i], lbg[3 * i + 1], lbg[3 * i + 2]};
  const double* sp = samples + (size_t)offset[i] * 7;
  for (int k = 0; k < nsamples[i]; ++k) {
    const double* p = sp + 7 * k;
    V3 a1{p[1], p[2], p[3]}, g1{p[4], p[5], p[6]};
    preint_step(s, p[0], a0, g0, a1, g1, ba, bg, nz2, F, Vv, Tm);
    a0 = a1; g0 = g1;
  }
  (void)Jm; (void)Pm;
  DevPreint& o = out[i];
  o.sum_dt = s.sum_dt;
  o.dp[0] = s.dp.x; o.dp[1] = s.dp.y; o.dp[2] = s.dp.z;
  o.dv[0] = s.dv.x; o.dv[1] = s.dv.y; o.dv[2] = s.dv.z;
  o.dq[0] = s.dq.x; o.dq[1] = s.dq.y; o.dq[2] = s.dq.z; o.dq[3] = s.dq.w;
  for (int k = 0; k < 3; ++k) { o.lba[k] = lba[3 * i + k]; o.lbg[k] = lbg[3 * i + k]; }
  for (int k = 0; k < 225; ++k) { o.cov[k] = s.P[k]; o.sqrt_info[k] = s.J[k]; }  // sqrt_info slot carries the full 15x15 jacobian out
}

}  // namespace vpl
