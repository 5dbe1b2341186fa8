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
constexpr int NT16 = 11;                     // 16x16 tiles per dimension (176 >= 172)
constexpr int NTILES = NT16 * (NT16 + 1) / 2;  // lower-triangular tile count (66)
constexpr int NAP = NTILES * 256;            // tile-major lower storage of the reduced system (16896 doubles)
constexpr int TK = 16;                       // rows of the landmark tile staged per pass
constexpr int TW = NV + 1;                   // tile width: 72 vis dims + rhs column
constexpr int NB3 = 25;                      // 3-wide output blocks over the 73(+2 pad) tile columns

// ceres defaults (solver.h, 1.12)
constexpr double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxMu = 1.0, kMuIncrease = 10.0;
constexpr double kMinRelDecrease = 1e-3, kFuncTol = 1e-6, kParamTol = 1e-8, kMinRadius = 1e-32;
constexpr int kMaxInvalid = 5;

__device__ __forceinline__ int tcol2row(int a) { return a < NV ? vis2cam(a) : NC; }
// Offset of element (r, c) inside a 16x16 tile.  Rows are XOR-swizzled: a column read by 16 lanes (the MFMA operands,
// one lane per row in the triangular solves) would otherwise land in one LDS bank pair.
__device__ __forceinline__ int tsw(int r, int c) { return (r << 4) + (c ^ r); }
// tile-major index of element (r, c), r >= c (or both inside a diagonal tile)
__device__ __forceinline__ int tix(int r, int c) {
  const int I = r >> 4, J = c >> 4;
  return ((I * (I + 1) / 2 + J) << 8) + tsw(r & 15, c & 15);
}
typedef double v4d __attribute__((ext_vector_type(4)));

// Compact "vis" system used by the Schur accumulation: 72 vis dims + rhs column (72) + Cauchy
// column (73), padded to 80 = 5 tiles of 16.  15 lower tiles live in MFMA accumulators.
constexpr int CW = 80;             // staged row width (doubles)
constexpr int CROWS = 64;          // landmark rows per staged chunk
constexpr int NCT = 15;            // lower tiles of the 5x5 compact tile grid

__device__ __forceinline__ void chol4(double* A, bool& ok) {   // packed lower 4x4, in place
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double d = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < j) d -= A[tri(j, k)] * A[tri(j, k)];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    A[tri(j, j)] = d;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i > j) {
      double s2 = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < j) s2 -= A[tri(i, k)] * A[tri(j, k)];
      A[tri(i, j)] = s2 / d;
    }
  }
}

__device__ __forceinline__ void solve_body(const DevBatch& B, const int w, double* sm) {
  const int tid = threadIdx.x, T = SOLVE_THREADS;
  const int lane = tid & 63, wv = tid >> 6;
  TrState* tr = &B.tr[w];
  if (tr->status != 0) return;
  if (B.path[w] == 0) return;     // the window takes the three-kernel path (ba_step.h)
  count_active(B, tr->reuse ? 2 : 1);
  double* S = sm;                 // NAP tile-major lower (row NC = rhs); first used as 2 staging buffers
  double* sc = S + NAP;           // 176 jacobi scale of cam dims
  double* dg = sc + 176;          // 176 dogleg diagonal of cam dims
  double* uc = dg + 176;          // 176 work vector (u for the Cauchy point, later S_c y_c)
  double* yv = uc + 176;          // 176 solution of the reduced system / z
  double* isd = yv + 176;         // 176 1/L_jj
  double* Linv = isd + 176;       // 256  inverse of the current diagonal tile's factor (swizzled tile)
  double* red = Linv + 256;       // 24
  int* flag = (int*)(red + 24);   // 4
  int* pSt = flag + 4;            // maxP  start frame of every point track (column map of the compact W rows)
  int* lSt = pSt + B.maxP;        // maxL  same for the lines

  const int nP = B.nP[w], nL = B.nL[w];
  const int WS = B.WS;
  for (int p = tid; p < nP; p += T) pSt[p] = B.pt_start[(size_t)w * B.maxP + p];
  for (int l = tid; l < nL; l += T) lSt[l] = B.ln_start[(size_t)w * B.maxL + l];
  const size_t fb = (size_t)w * B.nfull;
  double* gscale = B.scale + fb;
  double* gdiag = B.diag + fb;
  double* ggrad = B.grad + fb;
  double* ggn = B.gn + fb;
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  double* lch = B.lchol + (size_t)w * B.maxL * 10;
  const int LP = NC, LL = NC + B.maxP;   // offsets of the landmark sections in the full index
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }
  __syncthreads();

  // The Gauss-Newton step and the dogleg step also live in LDS (the reduced system's space is free when they are made): read
  // back from HBM right after being stored, each cost the store's completion plus a load round trip (~3 k cycles apiece)
  double* lgn = S + 4 * B.maxL;    // nfull  Gauss-Newton step (scaled space), copy of ggn
  double* gdelta = lgn + B.nfull;  // nfull  step * jacobi scale
  const bool reuse0 = tr->reuse != 0;
  VPL_STAMP(B, w, 0);
  if (!reuse0) {
    // ---- jacobi scaling (iteration 0 only), diagonal_, gradient_ --------------------------
    const bool first = (tr->iter == 0);
    // What the scaling works out for the landmarks (scale, diagonal, gradient, and their H blocks) is also left in LDS, in the
    // space of the staging buffers (free until the first chunk is staged): the landmark constants of the first factorisation
    // attempt read it there instead of loading back from HBM what was stored a moment ago.  A retry with a larger mu reloads.
    double* kP = S;                          // nP x 4: s, d, g, H_pp
    double* kL = S + 4 * B.maxP;             // nL x 28: s(4), d(4), g(4), H_ll(16)
    double a1 = 0.0, q = 0.0;
    // The inputs of this thread's first point and first line are requested BEFORE the camera entries are worked out and
    // stored: the three loops below otherwise pay three global round trips back to back (the compiler may not move the
    // later loops' loads above the earlier loops' stores).  Same arithmetic, same order.
    double pre_hp = 0.0, pre_sp = 0.0, pre_gp = 0.0, pre_Hl[16], pre_sl[4], pre_gl[4];
    if (tid < nP) {
      const size_t pi = (size_t)w * B.maxP + tid;
      pre_hp = B.Hpp[pi]; pre_gp = B.gp[pi];
      if (!first) pre_sp = gscale[LP + tid];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) pre_Hl[k] = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) { pre_sl[a] = 0.0; pre_gl[a] = 0.0; }
    if (tid < nL) {
      const size_t li = (size_t)w * B.maxL + tid;
#pragma unroll
      for (int k = 0; k < 16; ++k) pre_Hl[k] = B.Hll[li * 16 + k];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        pre_gl[a] = B.gl[li * 4 + a];
        if (!first) pre_sl[a] = gscale[LL + 4 * tid + a];
      }
    }
    for (int c = tid; c < 176; c += T) {
      double s = 0.0, d = 1.0, g = 0.0;
      if (c < NC) {
        const double h = Hcc[tri(c, c)];
        s = first ? 1.0 / (1.0 + sqrt(h)) : gscale[c];
        if (first) gscale[c] = s;
        d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
        g = s * gc[c] / d;
        gdiag[c] = d; ggrad[c] = g;
        a1 += g * g;
      }
      sc[c] = s; dg[c] = d;
      uc[c] = s * g / d;   // unscaled-space vector of gradient_/diagonal_
    }
    for (int p = tid; p < nP; p += T) {
      const size_t pi = (size_t)w * B.maxP + p;
      const bool pre = p == tid;   // this thread's first point was requested together with its camera entry (above)
      const double h = pre ? pre_hp : B.Hpp[pi];
      const double s = first ? 1.0 / (1.0 + sqrt(h)) : (pre ? pre_sp : gscale[LP + p]);
      if (first) gscale[LP + p] = s;
      const double d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
      const double g = s * (pre ? pre_gp : B.gp[pi]) / d;
      gdiag[LP + p] = d; ggrad[LP + p] = g;
      kP[4 * p] = s; kP[4 * p + 1] = d; kP[4 * p + 2] = g; kP[4 * p + 3] = h;
      a1 += g * g;
      const double u = s * g / d;
      q += u * h * u;
    }
    for (int l = tid; l < nL; l += T) {
      const size_t li = (size_t)w * B.maxL + l;
      const bool pre = l == tid;
      double Hl[16], sl4[4], gl4[4];
#pragma unroll
      for (int k = 0; k < 16; ++k) Hl[k] = pre ? pre_Hl[k] : B.Hll[li * 16 + k];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        sl4[a] = first ? 0.0 : (pre ? pre_sl[a] : gscale[LL + 4 * l + a]);
        gl4[a] = pre ? pre_gl[a] : B.gl[li * 4 + a];
      }
      double u[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double h = Hl[5 * a];
        const double s = first ? 1.0 / (1.0 + sqrt(h)) : sl4[a];
        if (first) gscale[LL + 4 * l + a] = s;
        const double d = sqrt(fmin(fmax(s * s * h, kMinDiag), kMaxDiag));
        const double g = s * gl4[a] / d;
        gdiag[LL + 4 * l + a] = d; ggrad[LL + 4 * l + a] = g;
        kL[28 * l + a] = s; kL[28 * l + 4 + a] = d; kL[28 * l + 8 + a] = g;
        a1 += g * g;
        u[a] = s * g / d;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) kL[28 * l + 12 + k] = Hl[k];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double hu = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) hu += Hl[4 * a + b] * u[b];
        q += u[a] * hu;
      }
    }
    a1 = block_sum(a1, red);
    q = block_sum(q, red);
    VPL_STAMP(B, w, 1);

    // ---- Gauss-Newton step: (J^T J + mu D^2) y = J^T r via the Schur complement -------------
    double mu = tr->mu;
    bool solved = false;
    double alpha = 0.0;
    bool first_attempt = true;
    while (mu < kMaxMu) {
      // Schur accumulation on the matrix cores: Acc = X^T X over all landmark rows, where a row of X is
      // C^-1 S_l [W | g | e] (e chosen so that X^T e = W^T u: the Cauchy cross term).  Rows are streamed through two LDS
      // buffers of CROWS x CW; 15 lower 16x16 tiles of the compact 80x80 result stay in accumulator registers (2 tiles
      // per wave).
      v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      int ta0 = 0, tb0 = 0, ta1 = 0, tb1 = 0;
      tri_decode(wv, ta0, tb0);
      const bool has1 = wv + 8 < NCT;
      if (has1) tri_decode(wv + 8, ta1, tb1);
      const int nchP = (nP + CROWS - 1) / CROWS, nchL = (4 * nL + CROWS - 1) / CROWS;
      const int nch = nchP + nchL;
      double* buf0 = S;
      double* buf1 = S + CROWS * CW;
      // Per-landmark constants of the row transform, in LDS behind the two staging buffers: row scale and e-value of the
      // points, Cholesky factor / scale / e-vector of the lines.
      double* pS = S + 2 * CROWS * CW;        // nP   s / sqrt(A)
      double* pE = pS + B.maxP;               // nP   e = u / (s / sqrt(A))
      double* lC = pE + B.maxP;               // nL x 10
      double* lS = lC + 10 * B.maxL;          // nL x 4   jacobi scale / diagonal of C: the row solve multiplies, never divides
      double* lE = lS + 4 * B.maxL;           // nL x 4   e = C^T (u ./ s),  u = s g~ / d  =>  u/s = g~/d
      // per-landmark regularised blocks: points A = s^2 H + mu d^2 -> row scale s/sqrt(A); lines A_l = S H S + mu D^2 =
      // C C^T -> C to lch (the back substitution reads it) and, pre-divided, to LDS.  ONE pass per landmark kind, every
      // HBM operand requested before the arithmetic: as two passes with the row scale / factor handed over through HBM the
      // second one waited for the first one's stores and then for its own loads.
      const double* kP = S;                  // (see the scaling phase)
      const double* kL = S + 4 * B.maxP;
      for (int p = tid; p < nP; p += T) {
        const size_t pi = (size_t)w * B.maxP + p;
        double s, d, g, h;
        if (first_attempt) { s = kP[4 * p]; d = kP[4 * p + 1]; g = kP[4 * p + 2]; h = kP[4 * p + 3]; }
        else { s = gscale[LP + p]; d = gdiag[LP + p]; g = ggrad[LP + p]; h = B.Hpp[pi]; }
        const double Al = s * s * h + mu * d * d;
        if (!(Al > 0.0)) flag[0] = 1;
        const double smv = s / sqrt(Al);
        pS[p] = smv;
        pE[p] = (s * g / d) / smv;
      }
      for (int l = tid; l < nL; l += T) {
        const size_t li = (size_t)w * B.maxL + l;
        double Hl[16], s4[4], d4[4], g4[4];
        if (first_attempt) {
#pragma unroll
          for (int a = 0; a < 4; ++a) { s4[a] = kL[28 * l + a]; d4[a] = kL[28 * l + 4 + a]; g4[a] = kL[28 * l + 8 + a]; }
#pragma unroll
          for (int k = 0; k < 16; ++k) Hl[k] = kL[28 * l + 12 + k];
        } else {
#pragma unroll
          for (int k = 0; k < 16; ++k) Hl[k] = B.Hll[li * 16 + k];
#pragma unroll
          for (int a = 0; a < 4; ++a) { s4[a] = gscale[LL + 4 * l + a]; d4[a] = gdiag[LL + 4 * l + a]; g4[a] = ggrad[LL + 4 * l + a]; }
        }
        double A[10];
        int t = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) if (b <= a) {
            A[t] = s4[a] * s4[b] * Hl[4 * a + b];
            if (a == b) A[t] += mu * d4[a] * d4[a];
            ++t;
          }
        bool ok = true;
        chol4(A, ok);
        if (!ok) flag[0] = 1;
        double us[4];
#pragma unroll
        for (int k = 0; k < 10; ++k) { lch[l * 10 + k] = A[k]; lC[l * 10 + k] = A[k]; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const double rd = 1.0 / A[tri(a, a)];
          lS[4 * l + a] = s4[a] * rd;   // x_a = (s_a w_a - sum_q C_aq x_q) / C_aa = lS_a w_a - sum_q (C_aq / C_aa) x_q
          us[a] = g4[a] / d4[a];
#pragma unroll
          for (int q = 0; q < 4; ++q) if (q < a) lC[l * 10 + tri(a, q)] = A[tri(a, q)] * rd;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          double s2 = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) if (q >= a) s2 += A[tri(q, a)] * us[q];
          lE[4 * l + a] = s2;
        }
      }
      __syncthreads();
      // Loads first, arithmetic and LDS stores after: a loop that stores each element before loading the next one
      // pays the L2 / HBM latency once per element (10 times per chunk); batched, twice per chunk.
      auto stage = [&](int ch, double* buf) {
        if (ch < nchP) {
          const int r0 = ch * CROWS, cnt = min(CROWS, nP - r0);
          constexpr int NIT = CROWS * CW / SOLVE_THREADS / 2;   // 2 x 5 elements per thread
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {
            double raw[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
              const int it = tid + (half * NIT + k) * T;
              const int rr = it / CW, c = it - rr * CW;
              raw[k] = 0.0;
              if (rr < cnt && c <= NV) {
                const size_t pi = (size_t)w * B.maxP + r0 + rr;
                const int cc = c < NV ? wcol(c, pSt[r0 + rr], WS) : 0;
                if (cc >= 0) raw[k] = c < NV ? B.Wp[pi * WS + cc] : B.gp[pi];
              }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
              const int it = tid + (half * NIT + k) * T;
              const int rr = it / CW, c = it - rr * CW;
              double v = 0.0;
              if (rr < cnt && c < 74) v = c <= NV ? pS[r0 + rr] * raw[k] : pE[r0 + rr];
              buf[it] = v;
            }
          }
        } else {
          const int l0 = (ch - nchP) * (CROWS / 4), cnt = min(CROWS / 4, nL - l0);
          constexpr int NIT = ((CROWS / 4) * CW + SOLVE_THREADS - 1) / SOLVE_THREADS;   // 3 items per thread
          double wv4[NIT][4];
#pragma unroll
          for (int k = 0; k < NIT; ++k) {
            const int it = tid + k * T;
            const int ll = it / CW, c = it - ll * CW;
            const bool on = it < (CROWS / 4) * CW && ll < cnt && c <= NV;
            const size_t li = (size_t)w * B.maxL + l0 + (on ? ll : 0);
            const int cc = (on && c < NV) ? wcol(c, lSt[l0 + ll], WS) : 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              wv4[k][a] = 0.0;
              if (on && cc >= 0) wv4[k][a] = c < NV ? B.Wl[(li * 4 + a) * WS + cc] : B.gl[li * 4 + a];
            }
          }
#pragma unroll
          for (int k = 0; k < NIT; ++k) {
            const int it = tid + k * T;
            if (it >= (CROWS / 4) * CW) break;
            const int ll = it / CW, c = it - ll * CW;
            double x[4] = {0, 0, 0, 0};
            if (ll < cnt && c < 74) {
              const int l = l0 + ll;
              if (c <= NV) {
                const double* C = lC + l * 10;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                  double s2 = lS[4 * l + a] * wv4[k][a];
#pragma unroll
                  for (int q = 0; q < 4; ++q) if (q < a) s2 -= C[tri(a, q)] * x[q];   // off-diagonals pre-divided by C_aa
                  x[a] = s2;
                }
              } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) x[a] = lE[4 * l + a];
              }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) buf[(4 * ll + a) * CW + c] = x[a];
          }
        }
      };
#ifdef VPL_STAMPS
      long long st_stage = 0, st_mfma = 0, st_bar = 0;
#endif
      if (nch > 0) stage(0, buf0);
      __syncthreads();
      for (int ch = 0; ch < nch; ++ch) {
        double* cur = (ch & 1) ? buf1 : buf0;
#ifdef VPL_STAMPS
        long long t1 = __builtin_readcyclecounter();
#endif
        if (ch + 1 < nch) stage(ch + 1, (ch & 1) ? buf0 : buf1);
#ifdef VPL_STAMPS
        { const long long tt = __builtin_readcyclecounter(); st_stage += tt - t1; t1 = tt; }
#endif
        const int m = lane & 15, kk = lane >> 4;
        // operands of step ks + 2 are read while the matrix cores work on step ks: with the reads issued right in front of
        // their MFMA a step took ~280 cycles (LDS latency) for 64 cycles of matrix-core time
        const double* rowp = cur + kk * CW;
        const int oa0 = 16 * ta0 + m, ob0 = 16 * tb0 + m, oa1 = 16 * ta1 + m, ob1 = 16 * tb1 + m;
        double pa0[2], pb0[2], pa1[2], pb1[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double* row = rowp + 4 * q * CW;
          pa0[q] = row[oa0]; pb0[q] = row[ob0];
          pa1[q] = has1 ? row[oa1] : 0.0; pb1[q] = has1 ? row[ob1] : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < CROWS / 4; ++ks) {
          const double a0 = pa0[ks & 1], b0 = pb0[ks & 1], a1v = pa1[ks & 1], b1v = pb1[ks & 1];
          if (ks + 2 < CROWS / 4) {
            const double* row = rowp + 4 * (ks + 2) * CW;
            pa0[ks & 1] = row[oa0]; pb0[ks & 1] = row[ob0];
            if (has1) { pa1[ks & 1] = row[oa1]; pb1[ks & 1] = row[ob1]; }
          }
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
          if (has1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1v, b1v, acc1, 0, 0, 0);
        }
#ifdef VPL_STAMPS
        { const long long tt = __builtin_readcyclecounter(); st_mfma += tt - t1; t1 = tt; }
#endif
        __syncthreads();
#ifdef VPL_STAMPS
        { const long long tt = __builtin_readcyclecounter(); st_bar += tt - t1; t1 = tt; }
        if (tid == 0 && ch + 1 == nch) { long long* dg_ = B.dbg + (size_t)w * 64; dg_[52] = st_stage; dg_[53] = st_mfma; dg_[54] = st_bar; }
#endif
      }
      VPL_STAMP(B, w, 2);
      // ---- reduced system: S = Hcc - X^T X (tile-major), rhs row = gc - X^T z; Cauchy denominator ----
      // One pass: the entries go to their tile slots already in the Jacobi-scaled space with the LM diagonal added
      // (every slot of tile rows 0..9 is written; only the last tile row, which holds padding, is zeroed first).
      for (int idx = tid; idx < NT16 * 256; idx += T) S[(((NT16 - 1) * NT16 / 2) << 8) + idx] = 0.0;
      __syncthreads();
      double qq = 0.0;
      for (int base = 0; base < NCP; base += 8 * T) {   // eight loads in flight per thread, then the LDS scatter
        double hh[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * T + tid;
          hh[u] = idx < NCP ? Hcc[idx] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * T + tid;
          if (idx < NCP) {
            int r, c;
            tri_decode(idx, r, c);
            const double h = hh[u];
            double hs = h * (sc[r] * sc[c]);
            if (r == c) hs += mu * dg[r] * dg[r];
            S[tix(r, c)] = hs;
            if ((r >> 4) == (c >> 4)) S[tix(c, r)] = hs;   // diagonal tiles are kept as full squares
            qq += (r == c ? 1.0 : 2.0) * uc[r] * h * uc[c];
          }
        }
      }
      for (int c = tid; c < NC; c += T) S[tix(NC, c)] = gc[c] * sc[c];
      if (tid < 5) S[tix(NC + tid, NC + tid)] = 1.0;   // padding rows 171..175: unit pivots, never used
      __syncthreads();
      {
        const int m = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (half == 1 && !has1) break;
          const v4d acc = half ? acc1 : acc0;
          const int ta = half ? ta1 : ta0, tb = half ? tb1 : tb0;
          const double vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int a = 16 * ta + kk + 4 * v, b = 16 * tb + m;   // compact row / col
            if (a < b || b >= NV || a > NV + 1) continue;
            if (a <= NV) {
              const int ra = tcol2row(a), rb = vis2cam(b);
              const double vs = vals[v] * (ra < NC ? sc[ra] * sc[rb] : sc[rb]);
              S[tix(ra, rb)] -= vs;
              if ((ra >> 4) == (rb >> 4) && ra != rb) S[tix(rb, ra)] -= vs;
            } else {
              qq += 2.0 * vals[v] * uc[vis2cam(b)];   // (W^T u)_b u_c,b
            }
          }
        }
      }
      qq = block_sum(qq, red);
      alpha = a1 / (q + qq);   // DoglegStrategy::ComputeCauchyPoint
      __syncthreads();
      VPL_STAMP(B, w, 3);
      // ---- left-looking tile Cholesky (16x16 tiles).  Per tile column K:
      //   (a) C(I,K) -= sum_{J<K} L(I,J) L(K,J)^T  on the FP64 matrix cores, accumulator in registers
      //   (b) factor the diagonal tile (one wave, 16 column steps)
      //   (c) rows below: x = a L(K,K)^-T, one lane per row
      // The rhs row (row NC, inside tile row 10) rides along: forward substitution for free.
#ifdef VPL_STAMPS
      long long ta = 0, tb = 0, tc = 0, td = 0, t0 = __builtin_readcyclecounter();
#endif
      for (int K = 0; K < NT16; ++K) {
        // C(I,K) -= sum_{J<K} L(I,J) L(K,J)^T is applied in parts: the terms J < K-1 by the idle waves while wave 0 factored
        // the previous diagonal tile (look-ahead); the term J = K-1 of the diagonal tile by wave 0 at the end of the previous
        // solve phase, of the tiles below it by the idle waves while wave 0 factors this diagonal tile.  Two barriers per
        // tile column.  The order of the terms, and therefore every bit of the result, is the one of a single pass.
        auto rank_update = [&](int col, int I, int J0, int J1) {
          const int m = lane & 15, kk = lane >> 4;
          double* Ct = S + ((I * (I + 1) / 2 + col) << 8);
          v4d c;
          c.x = Ct[tsw(kk, m)]; c.y = Ct[tsw(kk + 4, m)]; c.z = Ct[tsw(kk + 8, m)]; c.w = Ct[tsw(kk + 12, m)];
          for (int J = J0; J < J1; ++J) {
            const double* Ai = S + ((I * (I + 1) / 2 + J) << 8);
            const double* Bk = S + ((col * (col + 1) / 2 + J) << 8);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double av = -Ai[tsw(m, 4 * ks + kk)];
              const double bv = Bk[tsw(m, 4 * ks + kk)];
              c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
            }
          }
          Ct[tsw(kk, m)] = c.x; Ct[tsw(kk + 4, m)] = c.y; Ct[tsw(kk + 8, m)] = c.z; Ct[tsw(kk + 12, m)] = c.w;
        };
#ifdef VPL_STAMPS
        { const long long t1 = __builtin_readcyclecounter(); ta += t1 - t0; t0 = t1; }
#endif
        if (wv == 0) {
          // Diagonal tile in registers: lane (r4, cc) holds rows r4, r4+4, r4+8, r4+12 of column cc of the (symmetric)
          // tile.  Per column step the pivot comes by v_readlane, the pivot column of the lane's rows by ds_swizzle inside
          // its 16-lane row group, the pivot row entry D[j][cc] (= D[cc][j]) by one ds_bpermute -- no LDS round trips.
          double* D = S + ((K * (K + 1) / 2 + K) << 8);
          const int r4 = lane >> 4, cc = lane & 15;
          double d[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) d[v] = D[tsw(r4 + 4 * v, cc)];
          const int ncol = min(16, NC - 16 * K);
          double pivc = 1.0;
          bool bad = false;
          double m[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) m[v] = (r4 + 4 * v == cc) ? 1.0 : 0.0;
#ifdef VPL_STAMPS
          const long long td0 = __builtin_readcyclecounter();
#endif
#define VPL_DSTEP(J) diag_tile_step<J>(d, m, r4, cc, ncol, pivc, bad);
          VPL_DSTEP(0) VPL_DSTEP(1) VPL_DSTEP(2) VPL_DSTEP(3) VPL_DSTEP(4) VPL_DSTEP(5) VPL_DSTEP(6) VPL_DSTEP(7)
          VPL_DSTEP(8) VPL_DSTEP(9) VPL_DSTEP(10) VPL_DSTEP(11) VPL_DSTEP(12) VPL_DSTEP(13) VPL_DSTEP(14) VPL_DSTEP(15)
#undef VPL_DSTEP
#ifdef VPL_STAMPS
          td += __builtin_readcyclecounter() - td0;
#endif
          if (bad) {
            if (lane == 0) flag[0] = 1;
          } else {
            // scale columns: L_ij = a_ij / sqrt(a_jj); publish 1/L_jj
            const double sq = sqrt(pivc);
            if (cc < ncol) {
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const int r = r4 + 4 * v;
                if (r > cc) D[tsw(r, cc)] = d[v] / sq;
                else if (r == cc) { const double id = 1.0 / sq; isd[16 * K + cc] = id; D[tsw(r, cc)] = 1.0 / id; }
              }
            }
            if (K < NT16 - 1) {
              // L^-1 = diag(1 / sqrt(pivot)) m for the tiles below (ncol == 16 here); 1/sqrt(pivot of row r) sits in lane r
              const double isq = 1.0 / sq;
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const int r = r4 + 4 * v;
                const double ir = __shfl(isq, r, 64);
                Linv[tsw(r, cc)] = cc <= r ? m[v] * ir : 0.0;
              }
            }
          }
        } else if (K >= 1) {
          // Meanwhile the other waves (i) finish tile column K below the diagonal tile with its last term J = K-1 (the
          // diagonal tile itself got that term from wave 0 in the previous solve phase, below) and (ii) look ahead: the
          // finished columns J < K are subtracted from tile column K+1.
          for (int I = K + 1 + (wv - 1); I < NT16; I += SOLVE_THREADS / 64 - 1) rank_update(K, I, K - 1, K);
          if (K + 1 < NT16)
            for (int I = K + 1 + (wv - 1); I < NT16; I += SOLVE_THREADS / 64 - 1) rank_update(K + 1, I, 0, K);
        }
        __syncthreads();
#ifdef VPL_STAMPS
        { const long long t1 = __builtin_readcyclecounter(); tb += t1 - t0; t0 = t1; }
#endif
        if (flag[0]) break;
        // tiles below the diagonal tile (the rhs row is in tile row 10): X = A(I,K) L(K,K)^-T = A(I,K) Linv^T on the
        // matrix cores, X[m][n] = sum_k A[m][k] Linv[n][k]
        for (int I = K + 1 + wv; I < NT16; I += SOLVE_THREADS / 64) {
          const int m = lane & 15, kk = lane >> 4;
          double* At = S + ((I * (I + 1) / 2 + K) << 8);
          v4d c = {0.0, 0.0, 0.0, 0.0};
          double av[4], bv[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) { av[ks] = At[tsw(m, 4 * ks + kk)]; bv[ks] = Linv[tsw(m, 4 * ks + kk)]; }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], c, 0, 0, 0);
          At[tsw(kk, m)] = c.x; At[tsw(kk + 4, m)] = c.y; At[tsw(kk + 8, m)] = c.z; At[tsw(kk + 12, m)] = c.w;
          if (I == K + 1) {
            // wave 0 has just made L(K+1, K): it gives the next diagonal tile its last term right away, so that the
            // factorisation of D(K+1) starts after this phase's barrier instead of after an update phase of its own
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            rank_update(K + 1, K + 1, K, K + 1);
          }
        }
        __syncthreads();
#ifdef VPL_STAMPS
        { const long long t1 = __builtin_readcyclecounter(); tc += t1 - t0; t0 = t1; }
#endif
      }
#ifdef VPL_STAMPS
      if (tid == 0) { B.dbg[(size_t)w * 64 + 40] = ta; B.dbg[(size_t)w * 64 + 41] = tb; B.dbg[(size_t)w * 64 + 42] = tc; B.dbg[(size_t)w * 64 + 43] = td; }
#endif
      __syncthreads();
      if (flag[0]) {   // LINEAR_SOLVER_FAILURE: raise mu and retry from the stored linearisation
        mu *= kMuIncrease;
        first_attempt = false;
        __syncthreads();
        if (tid == 0) flag[0] = 0;
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
        if (tr->num_invalid >= kMaxInvalid) { tr->status = 2; tr->iter -= 1; }   // FAILURE breaks before the iteration is recorded
        else if (tr->iter >= B.opt.num_iterations) tr->status = 3;
        tr->mu *= kMuIncrease;   // StepIsInvalid
        tr->reuse = 0;
      }
      return;
    }
    VPL_STAMP(B, w, 4);
    // ---- back substitution L^T y = z, tile by tile from the bottom: the diagonal tile is solved by one
    //      wave (16 dependent register steps), the update z_J -= L(K,J)^T y_K is spread over the workgroup.
    for (int c = tid; c < 176; c += T) yv[c] = c < NC ? S[tix(NC, c)] : 0.0;
    __syncthreads();
    for (int K = NT16 - 1; K >= 0; --K) {
      const double* D = S + ((K * (K + 1) / 2 + K) << 8);
      if (wv == 0) {
        // diagonal tile: lane i keeps y_i in a register and has its column L[j][i] preloaded; the 16 dependent steps are
        // a v_readlane + two multiplies each (no LDS round trip inside the chain)
        const int ncol = min(16, NC - 16 * K);
        double yl = lane < 16 ? yv[16 * K + lane] : 0.0;
        const double isdl = lane < ncol ? isd[16 * K + lane] : 0.0;
        double Lcol[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) Lcol[j] = (lane < j && j < ncol) ? D[tsw(j, lane & 15)] : 0.0;
#pragma unroll
        for (int j = 15; j >= 0; --j) {
          if (j < ncol) {
            const double yj = readlane_f64(yl, j) * readlane_f64(isdl, j);
            if (lane < j) yl -= Lcol[j] * yj;
            if (lane == j) yl = yj;
          }
        }
        if (lane < ncol) yv[16 * K + lane] = yl;
      }
      __syncthreads();
      for (int c = tid; c < 16 * K; c += T) {
        const double* Lk = S + ((K * (K + 1) / 2 + (c >> 4)) << 8);
        double s2 = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s2 += Lk[tsw(r, c & 15)] * yv[16 * K + r];
        yv[c] -= s2;
      }
      __syncthreads();
    }
    VPL_STAMP(B, w, 5);
    // y_c (scaled space) in yv; uc <- S_c y_c ; gn_c = -diag y
    double a2 = 0.0, a3 = 0.0;
    for (int c = tid; c < NC; c += T) {
      const double y = yv[c];
      uc[c] = sc[c] * y;
      const double gnv = -dg[c] * y;
      ggn[c] = gnv; lgn[c] = gnv;
      a2 += gnv * gnv;
      a3 += ggrad[c] * gnv;
    }
    __syncthreads();
    VPL_STAMP(B, w, 8);
    // ---- landmark back substitution y_l = A_l^-1 S_l (g_l - W_l S_c y_c): 8 lanes per landmark row so
    //      that every 72-wide row of W is read as one contiguous 576-byte segment
    {
      const int sub = lane & 7, grp = tid >> 3;   // 64 row groups per pass
      const int nblk = WS / 6;
      for (int p0 = 0; p0 < nP; p0 += 2 * (T / 8)) {   // two row groups per trip: their loads are in flight together
        double wyv[2] = {0.0, 0.0}, sv[2], dv[2], hv[2], gv2[2], grv[2];
        size_t piv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = p0 + h * (T / 8) + grp;
          piv[h] = (size_t)w * B.maxP + (p < nP ? p : 0);
          sv[h] = dv[h] = 1.0; hv[h] = gv2[h] = grv[h] = 0.0;
          if (p < nP) {
            // compact row: 6-blocks of the frames start .. start + maxTrack - 1, then the extrinsic block
            const int s0 = pSt[p];
            for (int blk = sub; blk < nblk; blk += 8) {
              const bool exb = blk == nblk - 1;
              const int vb = exb ? 66 : 6 * (s0 + blk);
              if (!exb && vb >= 66) continue;               // slot of a frame past the window
              const double* Wr = B.Wp + piv[h] * WS + 6 * blk;
#pragma unroll
              for (int k = 0; k < 6; ++k) wyv[h] += Wr[k] * uc[vis2cam(vb + k)];
            }
            if (sub == 0) {
              sv[h] = gscale[LP + p]; dv[h] = gdiag[LP + p]; hv[h] = B.Hpp[piv[h]]; gv2[h] = B.gp[piv[h]];
              grv[h] = ggrad[LP + p];
            }
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = p0 + h * (T / 8) + grp;
          double wy = wyv[h];
          wy += __shfl_xor(wy, 1, 64); wy += __shfl_xor(wy, 2, 64); wy += __shfl_xor(wy, 4, 64);
          if (p < nP && sub == 0) {
            const double s = sv[h], d = dv[h];
            const double Al = s * s * hv[h] + mu * d * d;
            const double y = s * (gv2[h] - wy) / Al;
            const double gnv = -d * y;
            ggn[LP + p] = gnv; lgn[LP + p] = gnv;
            a2 += gnv * gnv;
            a3 += grv[h] * gnv;
          }
        }
      }
      VPL_STAMP(B, w, 9);
      // lines: NLH row groups per trip, each row's W entries and g_l requested together; the rows' right-hand sides
      // s (g_l - W_l u_c) go to LDS (the reduced system's space is free by now) and ONE pass with a lane per line does the
      // 4x4 triangular solves -- done by the leader lane of each group they were a chain of 8 divisions per 64 rows
      double* lrhs = S;
      constexpr int NLH = 3;
      for (int r0 = 0; r0 < 4 * nL; r0 += NLH * (T / 8)) {
        double wyv[NLH], glv[NLH];
#pragma unroll
        for (int h = 0; h < NLH; ++h) {
          const int r = r0 + h * (T / 8) + grp, l = r >> 2, a = r & 3;
          wyv[h] = 0.0; glv[h] = 0.0;
          if (l < nL) {
            const size_t li = (size_t)w * B.maxL + l;
            const int s0 = lSt[l];
            for (int blk = sub; blk < nblk; blk += 8) {
              const bool exb = blk == nblk - 1;
              const int vb = exb ? 66 : 6 * (s0 + blk);
              if (!exb && vb >= 66) continue;
              const double* Wr = B.Wl + (li * 4 + a) * WS + 6 * blk;
#pragma unroll
              for (int k = 0; k < 6; ++k) wyv[h] += Wr[k] * uc[vis2cam(vb + k)];
            }
            if (sub == 0) glv[h] = B.gl[li * 4 + a];
          }
        }
#pragma unroll
        for (int h = 0; h < NLH; ++h) {
          const int r = r0 + h * (T / 8) + grp;
          double wy = wyv[h];
          wy += __shfl_xor(wy, 1, 64); wy += __shfl_xor(wy, 2, 64); wy += __shfl_xor(wy, 4, 64);
          if (r < 4 * nL && sub == 0) lrhs[r] = gscale[LL + r] * (glv[h] - wy);
        }
      }
      __syncthreads();
      for (int l = tid; l < nL; l += T) {
        double C[10], t4[4], gd4[4], gr4[4];   // the factor, diagonal and gradient come from HBM: one batch, before the chain
#pragma unroll
        for (int k = 0; k < 10; ++k) C[k] = lch[l * 10 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) { gd4[k] = gdiag[LL + 4 * l + k]; gr4[k] = ggrad[LL + 4 * l + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] = lrhs[4 * l + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double s2 = t4[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) if (j < k) s2 -= C[tri(k, j)] * t4[j];
          t4[k] = s2 / C[tri(k, k)];
        }
#pragma unroll
        for (int k = 3; k >= 0; --k) {
          double s2 = t4[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) if (j > k) s2 -= C[tri(j, k)] * t4[j];
          t4[k] = s2 / C[tri(k, k)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double gnv = -gd4[k] * t4[k];
          ggn[LL + 4 * l + k] = gnv; lgn[LL + 4 * l + k] = gnv;
          a2 += gnv * gnv;
          a3 += gr4[k] * gnv;
        }
      }
    }
    VPL_STAMP(B, w, 10);
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

  VPL_STAMP(B, w, 6);
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
      if (tr->num_invalid >= kMaxInvalid) { tr->status = 2; tr->iter -= 1; }   // FAILURE breaks before the iteration is recorded
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
    if (live) gdelta[k] = gscale[k] * (-c1 * ggrad[k] + c2 * (reuse0 ? ggn[k] : lgn[k])) / gdiag[k];   // (a re-used step comes from HBM)
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
    const Plk Lc_ = orth_to_plk(out);     // the candidate's world Pluecker line, once per line (B.lw_c)
    double* lwc = B.lw_c + li * 6;
    lwc[0] = Lc_.n.x; lwc[1] = Lc_.n.y; lwc[2] = Lc_.n.z; lwc[3] = Lc_.v.x; lwc[4] = Lc_.v.y; lwc[5] = Lc_.v.z;
  }
  sn = block_sum(sn, red);
  xn = block_sum(xn, red);
  VPL_STAMP(B, w, 7);
  if (tid == 0) {
    tr->dogleg_step_norm = dnorm;
    tr->model_cost_change = model_cost_change;
    tr->step_norm = sqrt(sn);
    tr->x_norm = sqrt(xn);
    tr->step_valid = 1;
    tr->num_invalid = 0;
  }
}
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(DevBatch B) {
  extern __shared__ double sm[];
  // the list k_cost of THIS iteration fills is emptied here (k_cost runs after this whole kernel)
  if (blockIdx.x == 0 && threadIdx.x == 0) { B.ord_cnt[2 * ((B.ord_it + 1) & 1)] = 0; B.ord_cnt[2 * ((B.ord_it + 1) & 1) + 1] = 0; }
  solve_body(B, ordered_window(B), sm);
}

constexpr size_t SOLVE_SMEM = (size_t)(NAP + 5 * 176 + 256 + 24) * sizeof(double) + 4 * sizeof(int);   // + (maxP + maxL) ints, see solve_smem
inline size_t solve_smem(int maxP, int maxL) { return SOLVE_SMEM + (size_t)(maxP + maxL) * sizeof(int); }
static_assert(2 * CROWS * CW <= NAP, "the two staging buffers alias the reduced-system storage");

// ---------------------------------------------------------------------------------------------------
constexpr int COST_THREADS = 512;
constexpr int COST_PRE_LDS = 62 + 225;   // staged part of a pre-integration in k_cost: the PRE_LDS leading doubles + sqrt_info

// two workgroups per CU (4 waves per SIMD, <= 128 VGPRs): the kernel is a latency chain per lane, occupancy is what helps
// (0.327 -> 0.283 ms per step against one workgroup per CU at 136 VGPRs)
__device__ __forceinline__ void cost_body(const DevBatch& B, const int w) {
  const int tid = threadIdx.x, T = COST_THREADS;
  TrState* tr = &B.tr[w];
  if (tr->status != 0 || !tr->step_valid) return;
  count_active(B, 3);
  // The kernel is a chain of global round trips (~2-3 k cycles each with every CU in it), not arithmetic: everything whose
  // address depends on the window alone is requested in ONE batch at the top -- the candidate state, the leading part and
  // the information factor of the ten pre-integrations (to LDS, coalesced), the prior's block table and linearisation
  // point, its Jacobian rows, and the lane records of the point / line factors (the host tables of k_lin: track, k, start
  // frame and observation offset in one word pair, so that the factor's own data is the second and last trip).
  __shared__ double xp[84], xs[99], prdx[MAXPN], red[20];
  __shared__ double plds[10 * COST_PRE_LDS], imu_raw[10 * 15];
  __shared__ int imu_on[10];
  __shared__ int accept;
  const int nL = B.nL[w], nP = B.nP[w];
  const int n = B.pr_n[w], nb = n > 0 ? B.pr_nb[w] : 0;
  for (int i = tid; i < 84; i += T) xp[i] = i < 77 ? B.pose_c[(size_t)w * 77 + i] : B.ex_c[(size_t)w * 7 + (i - 77)];
  for (int i = tid; i < 99; i += T) xs[i] = B.sb_c[(size_t)w * 99 + i];
  for (int i = tid; i < 10 * COST_PRE_LDS; i += T) {      // per factor: 62 leading doubles, then the 225 of sqrt_info
    const int f = i / COST_PRE_LDS, e = i - COST_PRE_LDS * f;
    plds[i] = ((const double*)&B.pre[(size_t)w * NF + f + 1])[e < PRE_LDS ? e : e + 225];
  }
  int pkind = 0, pfr = 0, pidx = 0;
  double px0[9];
  if (tid < nb) {
    pkind = B.pr_kind[(size_t)w * MAXPB + tid]; pfr = B.pr_frame[(size_t)w * MAXPB + tid]; pidx = B.pr_idx[(size_t)w * MAXPB + tid];
#pragma unroll
    for (int k = 0; k < 9; ++k) px0[k] = B.pr_x0[((size_t)w * MAXPB + tid) * 9 + k];
  }
  // eight lanes per row of J0 (coalesced): the first COST_J0 columns of this lane's row wait in registers
  constexpr int COST_J0 = 6;
  const int prow = tid >> 3, psub = tid & 7;
  double j0v[COST_J0], pr0 = 0.0;
  {
    const double* J0 = B.pr_J0 + (size_t)w * B.prS;
#pragma unroll
    for (int q = 0; q < COST_J0; ++q) {
      const int c = psub + 8 * q;
      j0v[q] = (prow < n && c < n) ? J0[(size_t)prow * n + c] : 0.0;
    }
    if (prow < n && psub == 0) pr0 = B.pr_r0[(size_t)w * MAXPN + prow];
  }
  const int nRounds = B.pu_cnt[w];
  const int2* plane = (const int2*)B.pu_lane + (size_t)w * B.maxPR * 512;
  int2 rec0 = nRounds > 0 ? plane[tid] : int2{-1, 0}, rec1 = nRounds > 1 ? plane[512 + tid] : int2{-1, 0};
  const int2* ltab = (const int2*)B.ll_tab + (size_t)w * B.llSlots;
  const int npass = B.ll_np[w];
  const int2 lrec0 = npass > 0 ? ltab[tid] : int2{-1, 0};
  __syncthreads();
  double cost = 0.0;
  if (n > 0) {
    if (tid < nb) {
      const double* x = pkind == 0 ? xp + 7 * pfr : pkind == 1 ? xs + 9 * pfr : xp + 77;
      double dx[9];
      prior_block_dx(pkind, x, px0, dx);
      const int ls = pkind == 1 ? 9 : 6;
      for (int k = 0; k < ls; ++k) prdx[pidx + k] = dx[k];
    }
  }
  // IMU, cost only: the raw residuals by one lane per factor, to LDS; their whitening S r by one lane per (factor, row) after
  // the visual factors
  if (tid >= 64 && tid < 74) {
    const int j = tid - 64 + 1;
    const DevPreint& dp = *(const DevPreint*)(plds + COST_PRE_LDS * (j - 1));   // (only the staged leading part is read)
    const bool on = !(dp.sum_dt > 10.0);
    imu_on[j - 1] = on ? 1 : 0;
    if (on) {
      PreInt p = load_preint(dp);
      double r[15];
      imu_residual_raw(p, xp + 7 * (j - 1), xs + 9 * (j - 1), xp + 7 * j, xs + 9 * j, B.opt.g_norm, r);
#pragma unroll
      for (int a = 0; a < 15; ++a) imu_raw[15 * (j - 1) + a] = r[a];
    }
  }
  const double hub = B.opt.huber_delta;
  const double* xe = xp + 77;
  // one lane per factor.  Data of the first two point rounds and the first line pass: one batch of loads
  auto pt_load = [&](const int2 rec, double* d) {
    const bool has = rec.x >= 0;
    const int p = has ? rec.x & 0xffff : 0, k = has ? (rec.x >> 16) & 15 : 0;
    const double* o0 = B.pt_obs + ((size_t)w * B.maxPO + (has ? rec.y : 0)) * 3;
    d[0] = B.invd_c[(size_t)w * B.maxP + p];
#pragma unroll
    for (int q = 0; q < 3; ++q) { d[1 + q] = o0[q]; d[4 + q] = o0[3 * k + q]; }
  };
  auto pt_eval = [&](const int2 rec, const double* d) {
    if (rec.x < 0) return;
    const int k = (rec.x >> 16) & 15, s = rec.x >> 20;
    double r[2], sc;
    projection_factor(xp + 7 * s, xp + 7 * (s + k), xe, d[0], V3{d[1], d[2], d[3]}, V3{d[4], d[5], d[6]},
                      B.opt.sqrt_info_point, r, false, nullptr, nullptr, nullptr, nullptr);
    cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
  };
  auto ln_load = [&](const int2 lr, double* ob, double* orth) {
    const int o = lr.x >= 0 ? lr.x : 0, l = lr.x >= 0 ? lr.y & 0xffff : 0;
    const double* po = B.ln_obs + ((size_t)w * B.maxLO + o) * 8;
    const double* pq = B.lw_c + ((size_t)w * B.maxL + l) * 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) ob[q] = po[q];
#pragma unroll
    for (int q = 0; q < 6; ++q) orth[q] = pq[q];
  };
  auto ln_eval = [&](const int2 lr, const double* ob, const double* orth) {
    if (lr.x < 0) return;
    const int k = (lr.y >> 16) & 15, s = lr.y >> 20;
    LineCtx c = line_ctx_plk(xp + 7 * (s + k), xe, Plk{V3{orth[0], orth[1], orth[2]}, V3{orth[3], orth[4], orth[5]}});
    double r[2], sc;
    line_factor_res(c, ob, B.opt.sqrt_info_line, r, nullptr);
    cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
    if (ob[7] == 1.0) {
      vp_factor_res(c, ob + 4, B.opt.sqrt_info_vp, r, nullptr);
      cost += 0.5 * huber(r[0] * r[0] + r[1] * r[1], hub, &sc);
    }
  };
  double pd0[7], pd1[7], lob[8], lor[6];
  pt_load(rec0, pd0);
  pt_load(rec1, pd1);
  __syncthreads();      // prdx
  if (n > 0) {
    // r = r0 + J0 dx
    if (prow < n) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < COST_J0; ++q) {
        const int c = psub + 8 * q;
        s += j0v[q] * prdx[c < n ? c : 0];       // (j0v is zero beyond the row)
      }
      if (n > 8 * COST_J0) {
        const double* J0 = B.pr_J0 + (size_t)w * B.prS;
        for (int c = psub + 8 * COST_J0; c < n; c += 8) s += J0[(size_t)prow * n + c] * prdx[c];
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (psub == 0) {
        s += pr0;
        cost += 0.5 * s * s;
      }
    }
    for (int r = prow + (T >> 3); r < n; r += T >> 3) {      // priors of more than 64 rows (none in the reference's windows)
      const double* J0 = B.pr_J0 + (size_t)w * B.prS;
      double s = 0;
      for (int c = psub; c < n; c += 8) s += J0[(size_t)r * n + c] * prdx[c];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (psub == 0) {
        s += B.pr_r0[(size_t)w * MAXPN + r];
        cost += 0.5 * s * s;
      }
    }
  }
  ln_load(lrec0, lob, lor);      // (in flight behind the point factors)
  pt_eval(rec0, pd0);
  pt_eval(rec1, pd1);
  ln_eval(lrec0, lob, lor);
  for (int round = 2; round < nRounds; ++round) {
    const int2 rec = plane[round * 512 + tid];
    pt_load(rec, pd0);
    pt_eval(rec, pd0);
  }
  for (int pass = 1; pass < npass; ++pass) {
    const int2 lr = ltab[pass * T + tid];
    ln_load(lr, lob, lor);
    ln_eval(lr, lob, lor);
  }
  (void)nP; (void)nL;
  __syncthreads();
  if (tid < 150) {
    const int f = tid / 15, a = tid - 15 * f;
    if (imu_on[f]) {
      const double* S = plds + COST_PRE_LDS * f + PRE_LDS + a * 15;
      double v = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) if (k >= a) v += S[k] * imu_raw[15 * f + k];
      cost += 0.5 * v * v;
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
    for (int i = tid; i < 6 * nL; i += T) B.lw[(size_t)w * B.maxL * 6 + i] = B.lw_c[(size_t)w * B.maxL * 6 + i];
  }
}
// two workgroups per CU (4 waves per SIMD, <= 128 VGPRs): the kernel is a latency chain per lane, occupancy is what helps
// (0.327 -> 0.283 ms per step against one workgroup per CU at 136 VGPRs)
#ifndef VPL_COST_WAVES
#define VPL_COST_WAVES 4          // A/B switch, as VPL_BACK_WAVES
#endif
__global__ __launch_bounds__(COST_THREADS, VPL_COST_WAVES) void k_cost(DevBatch B) {
  const int w = ordered_window(B);
  cost_body(B, w);
  // order of the next iteration: windows that will linearise / factor again go to the front of the list, the others (step
  // rejected: the Gauss-Newton step is re-used; terminated) fill it from the back
  if (threadIdx.x == 0) {
    const TrState* tr = &B.tr[w];
    const bool heavy = tr->status == 0 && (!tr->fresh_lin || !tr->reuse);
    const int nx = (B.ord_it + 1) & 1;
    const int pos = atomicAdd(&B.ord_cnt[2 * nx + (heavy ? 0 : 1)], 1);
    B.order[(size_t)nx * B.nW + (heavy ? pos : B.nW - 1 - pos)] = w;
  }
}

}  // namespace vpl
