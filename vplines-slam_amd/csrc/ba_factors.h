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

// IntegrationBase over raw samples (integration_base.h:30-36,54-198): 16 lanes per keyframe interval, 4 intervals per wave.
// Lane j < 15 owns COLUMN j of the 15 x 15 jacobian and of the covariance in registers.  F is block sparse (3 x 3 blocks
// built from four matrices: Rq + Rr, Rr [a1]x, Rq [a0]x + Rr [a1]x (I - [w]x dt), I - [w]x dt), so J <- F J and T = F P are
// ~100 FMAs per lane without any matrix in memory; P <- T F^T is F applied to the ROWS of T (P is symmetric), which is the
// one cross-lane step: T passes through a 15 x 16 LDS tile per interval.  V Q V^T has closed-form blocks.  The old kernel
// kept F, V, T, J and P of every lane in scratch: 50 ms and 16 GB of traffic for 10 240 intervals.
constexpr int PREINT_GROUP = 16;

struct PreintF {     // the four matrices F and V are made of, and the step
  M3 S1, S2, S3, ImW;
  double dt;
};
// x <- F x for one column x (15 doubles, blocks of 3: p, theta, v, ba, bg), in place
__device__ __forceinline__ void preint_apply_F(const PreintF& f, double* x) {
  const double dt = f.dt;
  const V3 x1{x[3], x[4], x[5]}, x2{x[6], x[7], x[8]}, x3{x[9], x[10], x[11]}, x4{x[12], x[13], x[14]};
  const V3 u = mul(f.S3, x1), sv = mul(f.S1, x3), r = mul(f.S2, x4);
  const V3 y0 = V3{x[0], x[1], x[2]} + u * (-0.25 * dt * dt) + x2 * dt + sv * (-0.25 * dt * dt) + r * (-0.25 * dt * dt * -dt);
  const V3 y1 = mul(f.ImW, x1) + x4 * (-1.0 * dt);
  const V3 y2 = u * (-0.5 * dt) + x2 + sv * (-0.5 * dt) + r * (-0.5 * dt * -dt);
  x[0] = y0.x; x[1] = y0.y; x[2] = y0.z; x[3] = y1.x; x[4] = y1.y; x[5] = y1.z; x[6] = y2.x; x[7] = y2.y; x[8] = y2.z;
}
// column / row c (lane dependent) of a 3 x 3 matrix as a 0/1-weighted sum: exact, and no dynamic index or select on the
// matrix elements (either one makes the compiler keep the matrix on the stack)
__device__ __forceinline__ V3 m3col(const M3& A, int c) {
  const double w0 = c == 0 ? 1.0 : 0.0, w1 = c == 1 ? 1.0 : 0.0, w2 = c == 2 ? 1.0 : 0.0;
  return V3{A.m[0] * w0 + A.m[1] * w1 + A.m[2] * w2, A.m[3] * w0 + A.m[4] * w1 + A.m[5] * w2,
            A.m[6] * w0 + A.m[7] * w1 + A.m[8] * w2};
}
__device__ __forceinline__ V3 m3row(const M3& A, int r) {
  const double w0 = r == 0 ? 1.0 : 0.0, w1 = r == 1 ? 1.0 : 0.0, w2 = r == 2 ? 1.0 : 0.0;
  return V3{A.m[0] * w0 + A.m[3] * w1 + A.m[6] * w2, A.m[1] * w0 + A.m[4] * w1 + A.m[7] * w2,
            A.m[2] * w0 + A.m[5] * w1 + A.m[8] * w2};
}

__global__ __launch_bounds__(64) void k_preintegrate(int n, const int* offset, const int* nsamples, const double* samples,
                                                     const double* acc0, const double* gyr0, const double* lba,
                                                     const double* lbg, double an2, double gn2, double aw2, double gw2,
                                                     DevPreint* out) {
  __shared__ double tile[4][15 * PREINT_GROUP];
  const int grp = threadIdx.x / PREINT_GROUP, j = threadIdx.x % PREINT_GROUP;
  const int iv = blockIdx.x * 4 + grp;
  const bool live = iv < n;
  const int i = live ? iv : n - 1;            // idle groups shadow the last interval (no stores) so that barriers stay uniform
  const int jc = j < 15 ? j : 14, jb = jc / 3, jj = jc % 3;
  double Jc[15], Pc[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) { Jc[k] = (k == jc) ? 1.0 : 0.0; Pc[k] = 0.0; }
  V3 dp{0, 0, 0}, dv{0, 0, 0};
  Q4 dq{1, 0, 0, 0};
  double sum_dt = 0.0;
  V3 a0{acc0[3 * i], acc0[3 * i + 1], acc0[3 * i + 2]}, g0{gyr0[3 * i], gyr0[3 * i + 1], gyr0[3 * i + 2]};
  const V3 ba{lba[3 * i], lba[3 * i + 1], lba[3 * i + 2]}, bg{lbg[3 * i], lbg[3 * i + 1], lbg[3 * i + 2]};
  const double* sp = samples + (size_t)offset[i] * 7;
  // every group of the block runs the same number of steps (barriers inside); a group past its own count idles
  int ns = nsamples[i], nmax = ns;
  nmax = max(nmax, __shfl_xor(nmax, 16, 64));
  nmax = max(nmax, __shfl_xor(nmax, 32, 64));
  double* tl = tile[grp];
  for (int k = 0; k < nmax; ++k) {
    const bool on = k < ns;
    const double* p = sp + 7 * (on ? k : 0);
    const double dt = on ? p[0] : 0.0;   // a group past its own sample count applies F = I, N = 0 (exact: every term carries dt)
    const V3 a1{p[1], p[2], p[3]}, g1{p[4], p[5], p[6]};
    // midPointIntegration (:54-198)
    const V3 un_acc_0 = qrot(dq, a0 - ba);
    const V3 un_gyr = (g0 + g1) * 0.5 - bg;
    const Q4 rq = qmul(dq, Q4{1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});
    const V3 un_acc_1 = qrot(rq, a1 - ba);
    const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    const V3 rp = dp + dv * dt + un_acc * (0.5 * dt * dt);
    const V3 rv = dv + un_acc * dt;
    const M3 Rq = qmat(dq), Rr = qmat(rq);
    PreintF f;
    f.dt = dt;
    f.ImW = add(ident(), scale(skew(un_gyr), -dt));
    f.S2 = mul(Rr, skew(a1 - ba));                       // Rr [a1 - ba]x
    f.S3 = add(mul(Rq, skew(a0 - ba)), mul(f.S2, f.ImW));
    f.S1 = add(Rq, Rr);
    // J <- F J ; T = F P (both in place; an idle group keeps working on its registers and never stores them)
    preint_apply_F(f, Jc);
    preint_apply_F(f, Pc);
    // rows of T through LDS: lane k wrote column k, lane j reads row j
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 15; ++r) tl[r * PREINT_GROUP + j] = Pc[r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 15; ++c) Pc[c] = on ? tl[jc * PREINT_GROUP + c] : Pc[c];
    preint_apply_F(f, Pc);                               // column j of F T^T = column j of (T F^T)^T = column j of P'
    // + column j of N = V Q V^T (V of :107-125; Q = diag(an2, gn2, an2, gn2, aw2, gw2) x I3).  With A = Rq, B = Rr,
    // C = Rr [a1]x: G = A A^T + B B^T and C C^T are symmetric, so their column jj is the matrix times its own row jj
    {
      const double a = 0.25 * dt * dt, b = -(0.25 * dt * dt * 0.5 * dt), c = 0.5 * dt, d = 0.5 * dt, e = -(0.5 * dt * 0.5 * dt);
      V3 n0{0, 0, 0}, n1{0, 0, 0}, n2{0, 0, 0};
      if (jb == 0 || jb == 2) {
        const V3 gcol = mul(Rq, m3row(Rq, jj)) + mul(Rr, m3row(Rr, jj));
        const V3 ccol = mul(f.S2, m3row(f.S2, jj));
        if (jb == 0) {        // blocks N00, N10 = N01^T, N20 = N02^T
          n0 = gcol * (an2 * a * a) + ccol * (2.0 * gn2 * b * b);
          n1 = m3row(f.S2, jj) * (2.0 * gn2 * b * c);
          n2 = gcol * (an2 * a * d) + ccol * (2.0 * gn2 * b * e);
        } else {              // N02, N12, N22
          n0 = gcol * (an2 * a * d) + ccol * (2.0 * gn2 * b * e);
          n1 = m3row(f.S2, jj) * (2.0 * gn2 * c * e);
          n2 = gcol * (an2 * d * d) + ccol * (2.0 * gn2 * e * e);
        }
      } else if (jb == 1) {   // N01, N11, N21 = N12^T
        n0 = m3col(f.S2, jj) * (2.0 * gn2 * b * c);
        n1 = V3{jj == 0 ? 1.0 : 0.0, jj == 1 ? 1.0 : 0.0, jj == 2 ? 1.0 : 0.0} * (2.0 * gn2 * c * c);
        n2 = m3col(f.S2, jj) * (2.0 * gn2 * c * e);
      }
      Pc[0] += n0.x; Pc[1] += n0.y; Pc[2] += n0.z; Pc[3] += n1.x; Pc[4] += n1.y; Pc[5] += n1.z;
      Pc[6] += n2.x; Pc[7] += n2.y; Pc[8] += n2.z;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        Pc[9 + q] += (jb == 3 && jj == q) ? aw2 * dt * dt : 0.0;
        Pc[12 + q] += (jb == 4 && jj == q) ? gw2 * dt * dt : 0.0;
      }
    }
    if (on) {
      dp = rp; dv = rv; dq = qnormalized(rq);
      sum_dt += dt;
      a0 = a1; g0 = g1;
    }
  }
  if (!live || j >= 15) return;
  DevPreint& o = out[iv];
  if (j == 0) {
    o.sum_dt = sum_dt;
    o.dp[0] = dp.x; o.dp[1] = dp.y; o.dp[2] = dp.z;
    o.dv[0] = dv.x; o.dv[1] = dv.y; o.dv[2] = dv.z;
    o.dq[0] = dq.x; o.dq[1] = dq.y; o.dq[2] = dq.z; o.dq[3] = dq.w;
    for (int k = 0; k < 3; ++k) { o.lba[k] = lba[3 * iv + k]; o.lbg[k] = lbg[3 * iv + k]; }
  }
#pragma unroll
  for (int r = 0; r < 15; ++r) { o.cov[r * 15 + j] = Pc[r]; o.sqrt_info[r * 15 + j] = Jc[r]; }  // sqrt_info slot carries the 15x15 jacobian out
}

}  // namespace vpl
