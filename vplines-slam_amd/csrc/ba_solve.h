// k_solve : one trust-region step per window, whole step on device:
//   Jacobi scaling -> dogleg diagonal/gradient/Cauchy point -> Schur complement of the
//   landmark blocks (inverse depth 1x1, line orthonormal 4x4) into the 171x171 reduced camera
//   system held in LDS -> Cholesky + triangular solves -> back-substitution -> dogleg step ->
//   model cost change -> candidate x (Plus).
// Restates, for the configuration at vins_estimator/src/estimator.cpp:1207-1215
// (DENSE_SCHUR + DOGLEG), ceres-solver 1.12 DoglegStrategy::ComputeStep /
// SchurEliminator / dense Cholesky / TrustRegionMinimizer::ComputeTrustRegionStep
// (third party, absent from the reference tree).
// k_cost  : cost-only evaluation of the candidate + step acceptance
//   (TrustRegionMinimizer::{ComputeCandidatePointAndEvaluateCost, ParameterToleranceReached,
//    FunctionToleranceReached, IsStepSuccessful, HandleSuccessfulStep, HandleUnsuccessfulStep}).
#pragma once
#include "ba_common.h"

namespace vpl {

constexpr int SOLVE_THREADS = 512;
constexpr int NA = NC + 1;                   // reduced system augmented with the rhs as last row
constexpr int NAP = NA * (NA + 1) / 2;       // 14878
constexpr int TK = 36;                       // rows of the landmark tile staged per pass
constexpr int TW = NV + 1;                   // tile width: 72 vis dims + rhs column
constexpr int NB3 = 25;                      // 3-wide output blocks over the 73(+2 pad) tile columns

// ceres defaults (solver.h, 1.12)
constexpr double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxMu = 1.0, kMuIncrease = 10.0;
constexpr double kMinRelDecrease = 1e-3, kFuncTol = 1e-6, kParamTol = 1e-8, kMinRadius = 1e-32;
constexpr int kMaxInvalid = 5;

__device__ __forceinline__ int tcol2row(int a) { return a < NV ? vis2cam(a) : NC; }

__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x, T = SOLVE_THREADS;
  TrState* tr = &B.tr[w];
  if (tr->status != 0) return;
  extern __shared__ double sm[];
  double* S = sm;                 // NAP packed lower (row NC = rhs)
  double* sc = S + NAP;           // NC  jacobi scale of cam dims
  double* dg = sc + NC;           // NC  dogleg diagonal of cam dims
  double* uc = dg + NC;           // NC  work vector (u for the Cauchy point, later S_c y_c)
  double* yv = uc + NC;           // NC  solution of the reduced system / z
  double* tile = yv + NC;         // TK * (TW+2)
  double* Cl = tile + TK * (TW + 2);  // (TK/4) * 10 line Cholesky factors
  double* red = Cl + (TK / 4) * 10;   // 20
  int* flag = (int*)(red + 20);       // 4

  const int nP = B.nP[w], nL = B.nL[w];
  const size_t fb = (size_t)w * B.nfull;
  double* gscale = B.scale + fb;
  double* gdiag = B.diag + fb;
  double* ggrad = B.grad + fb;
  double* ggn = B.gn + fb;
  double* gdelta = B.delta + fb;
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  const int LP = NC, LL = NC + B.maxP;   // offsets of the landmark sections in the full index
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }
  __syncthreads();

  if (!tr->reuse) {
    // ---- jacobi scaling (iteration 0 only), diagonal_, gradient_ --------------------------
    const bool first = (tr->iter == 0);
    double a1 = 0.0;
    for (int c = tid; c < NC; c += T) {
      const double h = Hcc[tri(c, c)];
      double s = first ? 1.0 / (1.0 + sqrt(h)) : gscale[c];
      if (first) gscale[c] = s;
      double d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
      double g = s * gc[c] / d;
      sc[c] = s; dg[c] = d;
      gdiag[c] = d; ggrad[c] = g;
      uc[c] = s * g / d;   // unscaled-space vector of gradient_/diagonal_
      a1 += g * g;
    }
    double q = 0.0;
    __syncthreads();
    for (int p = tid; p < nP; p += T) {
      const size_t pi = (size_t)w * B.maxP + p;
      const double h = B.Hpp[pi];
      double s = first ? 1.0 / (1.0 + sqrt(h)) : gscale[LP + p];
      if (first) gscale[LP + p] = s;
      double d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
      double g = s * B.gp[pi] / d;
      gdiag[LP + p] = d; ggrad[LP + p] = g;
      a1 += g * g;
      const double u = s * g / d;
      const double* Wr = B.Wp + pi * NV;
      double wu = 0;
      for (int k = 0; k < NV; ++k) wu += Wr[k] * uc[vis2cam(k)];
      q += u * (h * u + 2.0 * wu);
    }
    for (int l = tid; l < nL; l += T) {
      const size_t li = (size_t)w * B.maxL + l;
      const double* Hl = B.Hll + li * 16;
      double u[4];
      for (int a = 0; a < 4; ++a) {
        const double h = Hl[5 * a];
        double s = first ? 1.0 / (1.0 + sqrt(h)) : gscale[LL + 4 * l + a];
        if (first) gscale[LL + 4 * l + a] = s;
        double d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
        double g = s * B.gl[li * 4 + a] / d;
        gdiag[LL + 4 * l + a] = d; ggrad[LL + 4 * l + a] = g;
        a1 += g * g;
        u[a] = s * g / d;
      }
      const double* Wl = B.Wl + li * 4 * NV;
      for (int a = 0; a < 4; ++a) {
        double wu = 0;
        for (int k = 0; k < NV; ++k) wu += Wl[a * NV + k] * uc[vis2cam(k)];
        double hu = 0;
        for (int b = 0; b < 4; ++b) hu += Hl[4 * a + b] * u[b];
        q += u[a] * (hu + 2.0 * wu);
      }
    }
    // u_c^T Hcc u_c while loading the packed Hessian into LDS
    for (int idx = tid; idx < NCP; idx += T) {
      int r, c;
      tri_decode(idx, r, c);
      const double h = Hcc[idx];
      S[idx] = h;
      q += (r == c ? 1.0 : 2.0) * uc[r] * h * uc[c];
    }
    for (int c = tid; c < NC; c += T) S[tri(NC, c)] = gc[c];
    a1 = block_sum(a1, red);
    q = block_sum(q, red);
    const double alpha = a1 / q;   // DoglegStrategy::ComputeCauchyPoint

    // ---- Gauss-Newton step: (J^T J + mu D^2) y = J^T r via the Schur complement -------------
    double mu = tr->mu;
    bool solved = false;
    while (mu < kMaxMu) {
      // 3x3 register blocks of T -= A''^T A''
      const bool own = tid < NB3 * (NB3 + 1) / 2;
      int ba = 0, bb = 0;
      if (own) tri_decode(tid, ba, bb);
      double acc[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[k] = 0.0;
      const int rowsP = nP, rowsL = 4 * nL;
      for (int base = 0; base < rowsP + rowsL; ) {
        const bool isP = base < rowsP;
        int cnt;
        __syncthreads();
        if (isP) {
          cnt = min(TK, rowsP - base);
          // points: row = sqrt(m) [W | g],  m = s^2 / (s^2 H + mu d^2)
          for (int it = tid; it < cnt * TW; it += T) {
            const int rr = it / TW, c = it % TW;
            const int p = base + rr;
            const size_t pi = (size_t)w * B.maxP + p;
            const double s = gscale[LP + p], d = gdiag[LP + p];
            const double Al = s * s * B.Hpp[pi] + mu * d * d;
            if (!(Al > 0.0)) flag[0] = 1;
            const double sm_ = s / sqrt(Al);
            tile[rr * (TW + 2) + c] = sm_ * (c < NV ? B.Wp[pi * NV + c] : B.gp[pi]);
          }
        } else {
          const int l0 = (base - rowsP) / 4;
          const int nl = min(TK / 4, nL - l0);
          cnt = 4 * nl;
          if (tid < nl) {   // Cholesky of A_l = S H S + mu D^2
            const int l = l0 + tid;
            const size_t li = (size_t)w * B.maxL + l;
            const double* Hl = B.Hll + li * 16;
            double A[10];
            int t = 0;
            for (int a = 0; a < 4; ++a)
              for (int b = 0; b <= a; ++b, ++t) {
                const double sa = gscale[LL + 4 * l + a], sb_ = gscale[LL + 4 * l + b];
                A[t] = sa * sb_ * Hl[4 * a + b];
                if (a == b) { const double d = gdiag[LL + 4 * l + a]; A[t] += mu * d * d; }
              }
            // packed lower Cholesky 4x4
            bool ok = true;
            for (int j = 0; j < 4; ++j) {
              double d = A[tri(j, j)];
              for (int k = 0; k < j; ++k) d -= A[tri(j, k)] * A[tri(j, k)];
              if (!(d > 0.0)) { ok = false; d = 1.0; }
              d = sqrt(d);
              A[tri(j, j)] = d;
              for (int i = j + 1; i < 4; ++i) {
                double s2 = A[tri(i, j)];
                for (int k = 0; k < j; ++k) s2 -= A[tri(i, k)] * A[tri(j, k)];
                A[tri(i, j)] = s2 / d;
              }
            }
            if (!ok) flag[0] = 1;
            for (int k = 0; k < 10; ++k) Cl[tid * 10 + k] = A[k];
          }
          __syncthreads();
          // rows X = C^-1 (S_l [W_l | g_l])  column by column
          for (int it = tid; it < nl * TW; it += T) {
            const int ll = it / TW, c = it % TW;
            const int l = l0 + ll;
            const size_t li = (size_t)w * B.maxL + l;
            const double* C = Cl + ll * 10;
            double x[4];
            for (int a = 0; a < 4; ++a) {
              const double v = c < NV ? B.Wl[(li * 4 + a) * NV + c] : B.gl[li * 4 + a];
              double s2 = gscale[LL + 4 * l + a] * v;
              for (int k = 0; k < a; ++k) s2 -= C[tri(a, k)] * x[k];
              x[a] = s2 / C[tri(a, a)];
              tile[(4 * ll + a) * (TW + 2) + c] = x[a];
            }
          }
        }
        // pad columns 73,74 of the tile are never read as outputs of interest but must be finite
        for (int it = tid; it < cnt * 2; it += T) tile[(it >> 1) * (TW + 2) + TW + (it & 1)] = 0.0;
        __syncthreads();
        if (own) {
          const double* ta = tile + 3 * ba;
          const double* tb = tile + 3 * bb;
          for (int rr = 0; rr < cnt; ++rr) {
            const double a0 = ta[rr * (TW + 2)], a1_ = ta[rr * (TW + 2) + 1], a2 = ta[rr * (TW + 2) + 2];
            const double b0 = tb[rr * (TW + 2)], b1 = tb[rr * (TW + 2) + 1], b2 = tb[rr * (TW + 2) + 2];
            acc[0] += a0 * b0; acc[1] += a0 * b1; acc[2] += a0 * b2;
            acc[3] += a1_ * b0; acc[4] += a1_ * b1; acc[5] += a1_ * b2;
            acc[6] += a2 * b0; acc[7] += a2 * b1; acc[8] += a2 * b2;
          }
        }
        base += cnt;
      }
      __syncthreads();
      if (own) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int a = 3 * ba + i, b = 3 * bb + j;
            if (a < TW && b < TW && a >= b && !(a == NV && b == NV)) S[tri(tcol2row(a), tcol2row(b))] -= acc[3 * i + j];
          }
      }
      __syncthreads();
      // scale to the Jacobi-scaled space and add the LM diagonal
      for (int idx = tid; idx < NAP - 1; idx += T) {
        int r, c;
        tri_decode(idx, r, c);
        double v = S[idx];
        if (r < NC) {
          v *= sc[r] * sc[c];
          if (r == c) v += mu * dg[r] * dg[r];
        } else {
          v *= sc[c];
        }
        S[idx] = v;
      }
      __syncthreads();
      // Cholesky, right-looking, one barrier per column; the rhs row rides along (forward solve for free)
      {
        const int tx = tid & 31, ty = tid >> 5;
        for (int j = 0; j < NC; ++j) {
          const double ajj = S[tri(j, j)];
          if (!(ajj > 0.0)) { if (tid == 0) flag[0] = 1; break; }   // uniform: every lane reads the same value
          const double inv = 1.0 / ajj;
          for (int r = j + 1 + ty; r <= NC; r += 16) {
            const double f = S[tri(r, j)] * inv;
            const int cmax = r < NC ? r : NC - 1;
            for (int c = j + 1 + tx; c <= cmax; c += 32) S[tri(r, c)] -= f * S[tri(c, j)];
          }
          __syncthreads();
        }
      }
      __syncthreads();
      if (flag[0]) {   // LINEAR_SOLVER_FAILURE: raise mu and retry from the stored linearisation
        mu *= kMuIncrease;
        __syncthreads();
        if (tid == 0) flag[0] = 0;
        for (int idx = tid; idx < NCP; idx += T) S[idx] = Hcc[idx];
        for (int c = tid; c < NC; c += T) S[tri(NC, c)] = gc[c];
        __syncthreads();
        continue;
      }
      solved = true;
      break;
    }
    if (!solved) {
      // every mu < 1 failed: invalid step (TrustRegionMinimizer::HandleInvalidStep)
      if (tid == 0) {
        tr->mu = mu;
        tr->step_valid = 0;
        tr->iter += 1;
        tr->num_invalid += 1;
        if (tr->num_invalid >= kMaxInvalid) tr->status = 2;
        else if (tr->iter >= B.opt.num_iterations) tr->status = 3;
        tr->mu *= kMuIncrease;   // StepIsInvalid
        tr->reuse = 0;
      }
      return;
    }
    // L = S with columns divided by sqrt(pivot); z_j = S[NC][j] / sqrt(pivot_j)
    // back substitution L^T y = z on one wave (column oriented, rows of L are contiguous)
    double* isd = tile;   // 1/sqrt(pivot)
    for (int c = tid; c < NC; c += T) {
      const double v = 1.0 / sqrt(S[tri(c, c)]);
      isd[c] = v;
      yv[c] = S[tri(NC, c)] * v;
    }
    __syncthreads();
    if (tid < 64) {
      volatile double* yy = yv;
      for (int j = NC - 1; j >= 0; --j) {
        const double yj = yy[j] * isd[j];
        for (int i = tid; i < j; i += 64) yy[i] -= (S[tri(j, i)] * isd[i]) * yj;
        if (tid == 0) yy[j] = yj;
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    // y_c (scaled space) in yv; uc <- S_c y_c ; gn_c = -diag y
    double a2 = 0.0, a3 = 0.0;
    for (int c = tid; c < NC; c += T) {
      const double y = yv[c];
      uc[c] = sc[c] * y;
      const double gnv = -dg[c] * y;
      ggn[c] = gnv;
      a2 += gnv * gnv;
      a3 += ggrad[c] * gnv;
    }
    __syncthreads();
    // landmark back substitution: y_l = A_l^-1 S_l (g_l - W_l S_c y_c)
    for (int p = tid; p < nP; p += T) {
      const size_t pi = (size_t)w * B.maxP + p;
      const double s = gscale[LP + p], d = gdiag[LP + p];
      const double Al = s * s * B.Hpp[pi] + mu * d * d;
      const double* Wr = B.Wp + pi * NV;
      double wy = 0;
      for (int k = 0; k < NV; ++k) wy += Wr[k] * uc[vis2cam(k)];
      const double y = s * (B.gp[pi] - wy) / Al;
      const double gnv = -d * y;
      ggn[LP + p] = gnv;
      a2 += gnv * gnv;
      a3 += ggrad[LP + p] * gnv;
    }
    for (int l = tid; l < nL; l += T) {
      const size_t li = (size_t)w * B.maxL + l;
      const double* Hl = B.Hll + li * 16;
      double A[10], t4[4];
      int t = 0;
      for (int a = 0; a < 4; ++a) {
        const double sa = gscale[LL + 4 * l + a];
        for (int b = 0; b <= a; ++b, ++t) {
          A[t] = sa * gscale[LL + 4 * l + b] * Hl[4 * a + b];
          if (a == b) { const double d = gdiag[LL + 4 * l + a]; A[t] += mu * d * d; }
        }
        const double* Wl = B.Wl + (li * 4 + a) * NV;
        double wy = 0;
        for (int k = 0; k < NV; ++k) wy += Wl[k] * uc[vis2cam(k)];
        t4[a] = sa * (B.gl[li * 4 + a] - wy);
      }
      for (int j = 0; j < 4; ++j) {
        double d = A[tri(j, j)];
        for (int k = 0; k < j; ++k) d -= A[tri(j, k)] * A[tri(j, k)];
        d = sqrt(d);
        A[tri(j, j)] = d;
        for (int i = j + 1; i < 4; ++i) {
          double s2 = A[tri(i, j)];
          for (int k = 0; k < j; ++k) s2 -= A[tri(i, k)] * A[tri(j, k)];
          A[tri(i, j)] = s2 / d;
        }
      }
      for (int a = 0; a < 4; ++a) {
        double s2 = t4[a];
        for (int k = 0; k < a; ++k) s2 -= A[tri(a, k)] * t4[k];
        t4[a] = s2 / A[tri(a, a)];
      }
      for (int a = 3; a >= 0; --a) {
        double s2 = t4[a];
        for (int k = a + 1; k < 4; ++k) s2 -= A[tri(k, a)] * t4[k];
        t4[a] = s2 / A[tri(a, a)];
      }
      for (int a = 0; a < 4; ++a) {
        const double gnv = -gdiag[LL + 4 * l + a] * t4[a];
        ggn[LL + 4 * l + a] = gnv;
        a2 += gnv * gnv;
        a3 += ggrad[LL + 4 * l + a] * gnv;
      }
    }
    a2 = block_sum(a2, red);
    a3 = block_sum(a3, red);
    if (tid == 0) {
      tr->mu = mu;
      tr->alpha = alpha;
      tr->a1 = a1; tr->a2 = a2; tr->a3 = a3;
      tr->reuse = 1;   // DoglegStrategy::ComputeStep sets reuse_ = true
    }
    __syncthreads();
  }

  // ---- DoglegStrategy::ComputeTraditionalDoglegStep ---------------------------------------------
  const double radius = tr->radius, alpha = tr->alpha, a1 = tr->a1, a2 = tr->a2, a3 = tr->a3, mu = tr->mu;
  const double gradient_norm = sqrt(a1), gauss_newton_norm = sqrt(a2);
  double c1, c2, dnorm;   // step (scaled space, before /diag) = -c1 * gradient_ + c2 * gauss_newton_step
  if (gauss_newton_norm <= radius) {
    c1 = 0.0; c2 = 1.0; dnorm = gauss_newton_norm;
  } else if (gradient_norm * alpha >= radius) {
    c1 = radius / gradient_norm; c2 = 0.0; dnorm = radius;
  } else {
    const double b_dot_a = -alpha * a3;
    const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
    const double bma = a_sq - 2 * b_dot_a + a2;
    const double c = b_dot_a - a_sq;
    const double d = sqrt(c * c + bma * (radius * radius - a_sq));
    const double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
    c1 = alpha * (1.0 - beta); c2 = beta;
    dnorm = sqrt(c1 * c1 * a1 - 2.0 * c1 * c2 * a3 + c2 * c2 * a2);
  }
  // model_cost_change = -(step^T gs + 1/2 step^T Hs step) with Hs y = gs - mu D^2 y folded in
  //   step = -c1 v - c2 y,  v = gradient_/diag, y = -gn/diag
  const double q_cauchy = a1 / alpha;                 // v^T Hs v
  const double sg = -c1 * a1 + c2 * a3;               // step^T gs
  const double vHy = a1 + mu * a3;
  const double yHy = -a3 - mu * a2;
  const double sHs = c1 * c1 * q_cauchy + 2.0 * c1 * c2 * vHy + c2 * c2 * yHy;
  const double model_cost_change = -(sg + 0.5 * sHs);
  const bool valid = model_cost_change > 0.0;

  if (!valid) {
    if (tid == 0) {
      tr->step_valid = 0;
      tr->iter += 1;
      tr->num_invalid += 1;
      if (tr->num_invalid >= kMaxInvalid) tr->status = 2;
      else if (tr->iter >= B.opt.num_iterations) tr->status = 3;
      tr->mu *= kMuIncrease;
      tr->reuse = 0;
    }
    return;
  }

  // ---- delta = step * jacobi scale; candidate = Plus(x, delta) -----------------------------------
  const int nfull_used = NC + B.maxP + 4 * nL;
  for (int k = tid; k < nfull_used; k += T) {
    const bool live = k < NC || (k >= LP && k < LP + nP) || k >= LL;
    if (live) gdelta[k] = gscale[k] * (-c1 * ggrad[k] + c2 * ggn[k]) / gdiag[k];
  }
  __syncthreads();
  double sn = 0.0, xn = 0.0;
  const bool ex_free = B.opt.estimate_extrinsic != 0;
  if (tid < NF + 1) {
    const bool isex = tid == NF;
    const double* x = isex ? B.ex + (size_t)w * 7 : B.pose + ((size_t)w * NF + tid) * 7;
    double* xc = isex ? B.ex_c + (size_t)w * 7 : B.pose_c + ((size_t)w * NF + tid) * 7;
    if (isex && !ex_free) {
      for (int k = 0; k < 7; ++k) xc[k] = x[k];
    } else {
      double out[7];
      pose_plus(x, gdelta + (isex ? 165 : 15 * tid), out);
      for (int k = 0; k < 7; ++k) { xc[k] = out[k]; sn += (x[k] - out[k]) * (x[k] - out[k]); xn += x[k] * x[k]; }
    }
  } else if (tid >= 64 && tid < 64 + NF) {
    const int f = tid - 64;
    const double* x = B.sb + ((size_t)w * NF + f) * 9;
    double* xc = B.sb_c + ((size_t)w * NF + f) * 9;
    for (int k = 0; k < 9; ++k) {
      const double d = gdelta[15 * f + 6 + k];
      xc[k] = x[k] + d;
      sn += d * d; xn += x[k] * x[k];
    }
  }
  for (int p = tid; p < nP; p += T) {
    const size_t pi = (size_t)w * B.maxP + p;
    const double d = gdelta[LP + p];
    B.invd_c[pi] = B.invd[pi] + d;
    sn += d * d; xn += B.invd[pi] * B.invd[pi];
  }
  for (int l = tid; l < nL; l += T) {
    const size_t li = (size_t)w * B.maxL + l;
    double out[4];
    line_orth_plus(B.orth + li * 4, gdelta + LL + 4 * l, out);
    for (int k = 0; k < 4; ++k) {
      const double x = B.orth[li * 4 + k];
      B.orth_c[li * 4 + k] = out[k];
      sn += (x - out[k]) * (x - out[k]); xn += x * x;
    }
  }
  sn = block_sum(sn, red);
  xn = block_sum(xn, red);
  if (tid == 0) {
    tr->dogleg_step_norm = dnorm;
    tr->model_cost_change = model_cost_change;
    tr->step_norm = sqrt(sn);
    tr->x_norm = sqrt(xn);
    tr->step_valid = 1;
    tr->num_invalid = 0;
  }
}

constexpr size_t SOLVE_SMEM = (size_t)(NAP + 4 * NC + TK * (TW + 2) + (TK / 4) * 10 + 20) * sizeof(double) + 4 * sizeof(int);

// ---------------------------------------------------------------------------------------------------
constexpr int COST_THREADS = 512;

__global__ __launch_bounds__(COST_THREADS) void k_cost(DevBatch B) {
  const int w = blockIdx.x, tid = threadIdx.x, T = COST_THREADS;
  TrState* tr = &B.tr[w];
  if (tr->status != 0 || !tr->step_valid) return;
  __shared__ double xp[84], xs[99], prdx[MAXPN], red[20];
  __shared__ int accept;
  const int nP = B.nP[w], nL = B.nL[w];
  for (int i = tid; i < 84; i += T) xp[i] = i < 77 ? B.pose_c[(size_t)w * 77 + i] : B.ex_c[(size_t)w * 7 + (i - 77)];
  for (int i = tid; i < 99; i += T) xs[i] = B.sb_c[(size_t)w * 99 + i];
  __syncthreads();
  double cost = 0.0;
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
    __syncthreads();
    const double* J0 = B.pr_J0 + (size_t)w * MAXPN * MAXPN;
    for (int r = tid; r < n; r += T) {
      double s = B.pr_r0[(size_t)w * MAXPN + r];
      for (int c = 0; c < n; ++c) s += J0[(size_t)r * n + c] * prdx[c];
      cost += 0.5 * s * s;
    }
  }
  // IMU (one lane per factor; cost only)
  if (tid >= 64 && tid < 74) {
    const int j = tid - 64 + 1;
    const DevPreint& dp = B.pre[(size_t)w * NF + j];
    if (!(dp.sum_dt > 10.0)) {
      PreInt p = load_preint(dp);
      double r[15];
      imu_residual_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm, r);
      double s = 0;
      for (int a = 0; a < 15; ++a) {
        double v = 0;
        for (int k = a; k < 15; ++k) v += dp.sqrt_info[a * 15 + k] * r[k];
        s += v * v;
      }
      cost += 0.5 * s;
    }
  }
  const double hub = B.opt.huber_delta;
  const double* xe = xp + 77;
  for (int p = tid; p < nP; p += T) {
    const size_t pi = (size_t)w * B.maxP + p;
    const int s = B.pt_start[pi], no = B.pt_nobs[pi], off = B.pt_off[pi];
    const double lam = B.invd_c[pi];
    const double* o0 = B.pt_obs + ((size_t)w * B.maxPO + off) * 3;
    V3 pts_i{o0[0], o0[1], o0[2]};
    for (int k = 1; k < no; ++k) {
      const double* oj = o0 + 3 * k;
      double r[2], sc;
      projection_factor(xp + 7 * s, xp + 7 * (s + k), xe, lam, pts_i, V3{oj[0], oj[1], oj[2]}, B.opt.sqrt_info_point, r,
                        false, nullptr, nullptr, nullptr, nullptr);
      cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
    }
  }
  for (int l = T - 1 - tid; l < nL; l += T) {
    const size_t li = (size_t)w * B.maxL + l;
    const int s = B.ln_start[li], no = B.ln_nobs[li], off = B.ln_off[li];
    const double* orth = B.orth_c + li * 4;
    for (int k = 0; k < no; ++k) {
      const double* ob = B.ln_obs + ((size_t)w * B.maxLO + off + k) * 8;
      LineCtx c = line_ctx(xp + 7 * (s + k), xe, orth);
      double r[2], sc;
      line_factor_res(c, ob, B.opt.sqrt_info_line, r, nullptr);
      cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
      if (ob[7] == 1.0) {
        vp_factor_res(c, ob + 4, B.opt.sqrt_info_vp, r, nullptr);
        cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
      }
    }
  }
  cost = block_sum(cost, red);
  if (tid == 0) {
    int acc = 0;
    double cand = cost;
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    tr->cand_cost = cand;
    const double x_cost = tr->x_cost;
    if (tr->step_norm <= kParamTol * (tr->x_norm + kParamTol)) {
      tr->status = 1;   // parameter tolerance: terminate without taking the step
    } else if (fabs(x_cost - cand) <= kFuncTol * x_cost) {
      tr->status = 1;   // function tolerance: terminate without taking the step
    } else {
      const double rel = (x_cost - cand) / tr->model_cost_change;
      tr->iter += 1;
      if (rel > kMinRelDecrease) {
        acc = 1;
        // DoglegStrategy::StepAccepted
        if (rel < 0.25) tr->radius *= 0.5;
        if (rel > 0.75) tr->radius = fmax(tr->radius, 3.0 * tr->dogleg_step_norm);
        tr->mu = fmax(1e-8, 2.0 * tr->mu / kMuIncrease);
        tr->reuse = 0;
        tr->fresh_lin = 0;
        tr->x_cost = cand;
        tr->num_successful += 1;
      } else {
        tr->radius *= 0.5;   // StepRejected
        tr->reuse = 1;
      }
      if (tr->iter >= B.opt.num_iterations) tr->status = 3;
      else if (tr->radius <= kMinRadius) tr->status = 1;
    }
    tr->step_valid = 0;
    accept = acc;
  }
  __syncthreads();
  if (accept) {
    for (int i = tid; i < 77; i += T) B.pose[(size_t)w * 77 + i] = B.pose_c[(size_t)w * 77 + i];
    for (int i = tid; i < 99; i += T) B.sb[(size_t)w * 99 + i] = B.sb_c[(size_t)w * 99 + i];
    for (int i = tid; i < 7; i += T) B.ex[(size_t)w * 7 + i] = B.ex_c[(size_t)w * 7 + i];
    for (int p = tid; p < nP; p += T) B.invd[(size_t)w * B.maxP + p] = B.invd_c[(size_t)w * B.maxP + p];
    for (int i = tid; i < 4 * nL; i += T) B.orth[(size_t)w * B.maxL * 4 + i] = B.orth_c[(size_t)w * B.maxL * 4 + i];
  }
}

}  // namespace vpl
