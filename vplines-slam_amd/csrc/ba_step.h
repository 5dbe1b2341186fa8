// The trust-region step of a window as three kernels, each shaped for what it does (round 3):
//
//   k_schur  (512 threads, ~45 KB LDS, two work-groups per CU)  Jacobi scaling, dogleg diagonal / gradient, Cauchy point,
//            regularised landmark blocks, and the landmark elimination: the rows of X = C^-1 S [W | g | e] go from HBM
//            straight into the FP64 matrix cores -- the lane that supplies operand element (row, column) loads exactly that
//            element -- in the COMPACT coordinates of the rows' start frame (WS + 2 columns instead of 80), every wave on its
//            own span of rows.  No staging buffers, no work-group barriers in the product.
//   k_chol   (256 threads, ~70 KB LDS, two work-groups per CU)  the reduced camera system.  The 99 speed/bias dims touch IMU
//            factors and the prior only: the blocks of frames 1..4 and 10..6 are eliminated first as two block chains (one
//            wave each, 9-row strips held in registers with the columns in the lanes), which leaves the DENSE system
//            [72 pose / extrinsic dims | speed/bias 0 | speed/bias 5 | rhs] = 91 rows = 6 tile columns instead of 11.
//   k_back   (512 threads, ~16 KB LDS)  landmark back-substitution, dogleg step, model cost change, candidate x (Plus).
//
// Any elimination order gives the same Gauss-Newton step up to rounding; the arithmetic per entry is
// the one of ba_solve.h.  Windows the fast path does not cover -- a prior that holds a speed/bias block of another frame
// than 0, a factorisation that failed and is retried with a larger mu (ceres' LINEAR_SOLVER_FAILURE loop) -- are flagged in
// B.path and take k_solve (ba_solve.h), which is launched between k_chol and k_back and leaves at once for everybody else.
// Restates ceres-solver 1.12 DoglegStrategy::ComputeStep / SchurEliminator / TrustRegionMinimizer (third party, absent
// from the reference tree) for the configuration at vins_estimator/src/estimator.cpp:1207-1215.
#pragma once
#include "ba_common.h"
#include "ba_solve.h"

namespace vpl {

constexpr int SCHUR_THREADS = 256;
constexpr int CHOL_THREADS = 256;
constexpr int BACK_THREADS = 256;

__device__ __forceinline__ void lds_ticket_wait(int* t, int seq) {
  while (__hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   // (LDS only: global loads in flight stay in flight)
}
__device__ __forceinline__ void lds_ticket_pass(int* t, int seq, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if (lane == 0) __hip_atomic_store(t, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// vis index (0..73: 72 = rhs row, 73 = Cauchy row) of compact column c of a row whose track starts in frame s; -1: the slot
// of a frame past the window (always zero) or padding
__device__ __forceinline__ int compact2vis(int c, int s, int WS) {
  if (c < WS - 6) { const int v = 6 * s + c; return v < 66 ? v : -1; }
  if (c < WS) return 66 + (c - (WS - 6));
  return c < WS + 2 ? 72 + (c - WS) : -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_schur
// ---------------------------------------------------------------------------------------------------------------------
// NTC: 16-column tiles of the compact row (WS + 2 <= 16 NTC).  NTC = 3 covers tracks of up to 6 frames (the benchmark
// shape): 6 lower tiles = 48 accumulator registers per lane; NTC = 5 covers every track length (15 tiles).
constexpr int CSQ_LD = 80;                   // row stride of the compact system in k_schur's LDS
constexpr int CSQ_N = 74 * CSQ_LD + 32;      // (+ slack: adds of exact zeros for frame slots past the window land behind a row's end)
// A WIDE chunk of k_schur_mixed (entries that hold a track of more than SCHUR_NARROW_FRAMES frames), out of line: its
// register allocation is its own, the narrow product loop of the caller keeps its zero scratch.
struct SchurWide {
  const int4* etab;
  const double *Wp, *Wl;
  int WS;
  const double *lC, *lS, *lE, *lG, *pS, *pE, *pG;
  double* Cacc;
  int* tick;
};
__device__ __noinline__ void schur_wide_chunk(const SchurWide& A, const int k0, const int k1, const int seq) {
  const int lane = threadIdx.x & 63, m = lane & 15, kk = lane >> 4;
  const int4* etab = A.etab;
  const double *Wp = A.Wp, *Wl = A.Wl, *lC = A.lC, *lS = A.lS, *lE = A.lE, *lG = A.lG, *pS = A.pS, *pE = A.pE, *pG = A.pG;
  double* Cacc = A.Cacc;
  int* tick = A.tick;
  const int WS = A.WS;
  auto fetch = [&](int k) {
    int4 e = etab[k];
    e.x = __builtin_amdgcn_readfirstlane(e.x); e.y = __builtin_amdgcn_readfirstlane(e.y); e.z = __builtin_amdgcn_readfirstlane(e.z);
    return e;
  };
  auto lm_id = [&](const int4 e) { return ((kk & 2 ? e.z : e.y) >> (16 * (kk & 1))) & 0xffff; };
  {
        // ---- a WIDE chunk: all NTW = 5 column tiles of the WS + 2 columns, ONE ROW OF TILES PER PASS over the chunk's entries
        // (pass ta: tiles (ta, 0 .. 4), the A operand is column tile ta, loaded as a sixth column): 5 accumulator tiles, 6
        // raw and 6 transformed columns = fewer registers than the narrow product loop, which therefore keeps its zero
        // scratch (all 15 tiles at once: 1.2 KB of scratch per lane for the whole kernel).  Wide chunks are the few entries
        // with long tracks; their rows are read five times.  Flush targets by arithmetic (no second 15 KB table in LDS).
        // The ticket is taken at the first pass's adds and passed on after the last's.
        constexpr int NTW = 5;
        const int sfr = fetch(k0).x & 15, s6 = 6 * sfr;
#pragma unroll 1
        for (int ta = 0; ta < NTW && 16 * ta < WS + 2; ++ta) {
          v4d accw[NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) accw[t] = v4d{0, 0, 0, 0};
          auto col_of = [&](int t) { return 16 * (t < NTW ? t : ta) + m; };
          auto load_w = [&](const int4 e, double (&raw)[NTW + 1][4]) {
            const bool isl = (e.x & 32) != 0;
            const int idr = lm_id(e);
            const int id = idr != 0xffff ? idr : 0;
            const double* row = isl ? Wl + (unsigned)(id * 4 * WS) : Wp + (unsigned)(id * WS);
#pragma unroll
            for (int t = 0; t <= NTW; ++t) {
              const int c = min(col_of(t), WS - 1);
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) raw[t][qd] = (qd == 0 || isl) ? row[(unsigned)((isl ? qd * WS : 0) + c)] : 0.0;
            }
          };
          auto xform_w = [&](const int4 e, const double (&raw)[NTW + 1][4], double (&x)[NTW + 1][4]) {
            const bool isl = (e.x & 32) != 0;
            const int idr = lm_id(e);
            const int id = idr != 0xffff ? idr : 0;
            const double pm = idr != 0xffff ? 1.0 : 0.0;
            double s4[4] = {0, 0, 0, 0}, ev[4] = {0, 0, 0, 0}, gv[4] = {0, 0, 0, 0};
            double c10 = 0, c20 = 0, c21 = 0, c30 = 0, c31 = 0, c32 = 0;
            if (isl) {
              const double* C = lC + id * 10;
#pragma unroll
              for (int a = 0; a < 4; ++a) { s4[a] = lS[4 * id + a] * pm; ev[a] = lE[4 * id + a] * pm; gv[a] = lG[4 * id + a] * pm; }
              c10 = C[1]; c20 = C[3]; c21 = C[4]; c30 = C[6]; c31 = C[7]; c32 = C[8];
            } else {
              s4[0] = pS[id] * pm; ev[0] = pE[id] * pm; gv[0] = pG[id] * pm;
            }
#pragma unroll
            for (int t = 0; t <= NTW; ++t) {
              const int c = col_of(t);
              const double wmv = c < WS ? 1.0 : 0.0, gmv = c == WS ? 1.0 : 0.0, emv = c == WS + 1 ? 1.0 : 0.0;
              // (same operations per element as the table-driven path: s * raw first, the triangular solve for the lines)
              const double x0 = s4[0] * raw[t][0];
              const double x1 = s4[1] * raw[t][1] - c10 * x0;
              const double x2 = s4[2] * raw[t][2] - c20 * x0 - c21 * x1;
              const double x3 = s4[3] * raw[t][3] - c30 * x0 - c31 * x1 - c32 * x2;
              x[t][0] = x0 * wmv + gv[0] * gmv + ev[0] * emv;
              x[t][1] = isl ? x1 * wmv + gv[1] * gmv + ev[1] * emv : 0.0;
              x[t][2] = isl ? x2 * wmv + gv[2] * gmv + ev[2] * emv : 0.0;
              x[t][3] = isl ? x3 * wmv + gv[3] * gmv + ev[3] * emv : 0.0;
            }
          };
          double rw[NTW + 1][4];
          load_w(fetch(k0), rw);
#pragma unroll 1
          for (int k = k0; k < k1; ++k) {
            const int4 e = fetch(k);
            double x[NTW + 1][4];
            xform_w(e, rw, x);
            load_w(fetch(min(k + 1, k1 - 1)), rw);
            const bool isl = (e.x & 32) != 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              if (a == 0 || isl) {
#pragma unroll
                for (int tb = 0; tb < NTW; ++tb) accw[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[NTW][a], x[tb][a], accw[tb], 0, 0, 0);
              }
            }
          }
          if (ta == 0) lds_ticket_wait(&tick[0], seq);
#pragma unroll
          for (int tb = 0; tb < NTW; ++tb) {
            const double vals[4] = {accw[tb].x, accw[tb].y, accw[tb].z, accw[tb].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int ca = 16 * ta + kk + 4 * v, cb = 16 * tb + m;
              if (tb <= ta && ca >= cb && ca < WS + 2 && cb < WS) {
                const int a0 = compact2vis(ca, 0, WS), b0 = compact2vis(cb, 0, WS);
                const int stp = (ca < WS - 6 ? CSQ_LD : 0) + (cb < WS - 6 ? 1 : 0);
                // columns of frames past the window (6 s + c >= 66) hold exact zeros: kept out of the square
                const bool in_a = ca >= WS - 6 || s6 + ca < 66, in_b = cb >= WS - 6 || s6 + cb < 66;
                if (in_a && in_b) lds_add(&Cacc[a0 * CSQ_LD + b0 + s6 * stp], vals[v]);
              }
            }
          }
        }
        lds_ticket_pass(&tick[0], seq, lane);
  }
}

// MIXED (round 4; NTC = 3): the batch holds tracks of more than SCHUR_NARROW_FRAMES frames, so the compact rows in HBM are WS > 42
// doubles wide -- but most entries still hold 6-frame tracks only.  Their chunks run the 3-tile product in a NARROW VIEW of the
// row (the 36 pose columns of frames s .. s + 5, then the extrinsic block found at memory column WS - 6, g, e: 44 columns);
// the chunks the host flagged wide (bit 6 of the group) run all 15 tiles of the WS + 2 columns.  Before, one long track in the
// batch sent every entry through k_schur<5>: 120 accumulator + 120 prefetch registers (spills), 86 KB of LDS (one work-group
// per CU), 5 x the time.
constexpr int SCHUR_NARROW_FRAMES = 6;
template <int NTC, bool MIXED = false>
__device__ __forceinline__ void schur_body(const DevBatch& B, const int w, double* sm) {
  static_assert(!MIXED || NTC == 3, "the narrow view is three column tiles");
  constexpr int NLT = NTC * (NTC + 1) / 2;
  const int tid = threadIdx.x, T = SCHUR_THREADS;
  const int lane = tid & 63, wv = tid >> 6;
  TrState* tr = &B.tr[w];
  if (tr->status != 0 || tr->reuse != 0 || B.path[w] != 0) return;
  count_active(B, 1);
  const int nP = B.nP[w], nL = B.nL[w];
  const int WS = B.WS;
  // the view the table-driven product works in: logical row width WSv, memory column of logical column c >= WSv - 6 is c + coff
  const int WSv = MIXED ? 6 * SCHUR_NARROW_FRAMES + 6 : WS;
  const int coff = WS - WSv;
  const int r1 = max(4 * B.maxP + 28 * B.maxL, CSQ_N);
  double* kP = sm;                          // nP x 4: s, d, g, H_pp            (scaling -> landmark constants)
  double* kL = sm + 4 * B.maxP;             // nL x 28: s(4), d(4), g(4), H_ll(16)
  double* Cacc = sm;                        // CSQ_N: the compact system as a 74 x 80 row-major square (lower part used), after the constants are made
  double* pS = sm + r1;                     // nP   s / sqrt(A)
  double* pE = pS + B.maxP;                 // nP   e = u / (s / sqrt(A))
  double* lC = pE + B.maxP;                 // nL x 10  Cholesky factor of the line block, off-diagonals pre-divided
  double* lS = lC + 10 * B.maxL;            // nL x 4   jacobi scale / diagonal of C
  double* lE = lS + 4 * B.maxL;             // nL x 4   e = C^T (u ./ s)
  double* lG = lE + 4 * B.maxL;             // nL x 4   g column of X: C^-1 S g_l
  double* pG = lG + 4 * B.maxL;             // nP       g column of X: (s / sqrt(A)) g_p
  double* uc = pG + B.maxP;                 // 176  unscaled-space vector of gradient_ / diagonal_ (cam dims)
  double* red = uc + 176;                   // 24
  int* tick = (int*)(red + 24);             // 2
  int* flag = tick + 2;                     // 2
  int4* etab = (int4*)(((uintptr_t)(flag + 2) + 15) & ~(uintptr_t)15);   // maxKS entries of the K-step table
  int* ftab = (int*)(etab + B.maxKS);       // NLT x 4 x 64 flush targets, see below

  {
    const int4* ktab = (const int4*)B.sk_tab + (size_t)w * B.maxKS;
    for (int k = tid; k < B.maxKS; k += T) etab[k] = ktab[k];
  }
  const size_t fb = (size_t)w * B.nfull;
  double* gscale = B.scale + fb;
  double* gdiag = B.diag + fb;
  double* ggrad = B.grad + fb;
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  double* lch = B.lchol + (size_t)w * B.maxL * 10;
  const int LP = NC, LL = NC + B.maxP;
  if (tid == 0) { flag[0] = 0; tick[0] = 0; }
  const double mu = tr->mu;
  const bool first = (tr->iter == 0);
  VPL_STAMP(B, w, 0);
  // ---- jacobi scaling (iteration 0 only), diagonal_, gradient_  (same arithmetic as ba_solve.h) ----
  double a1 = 0.0, q = 0.0;
  {
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
      uc[c] = s * g / d;
    }
    for (int p = tid; p < nP; p += T) {
      const size_t pi = (size_t)w * B.maxP + p;
      const bool pre = p == tid;
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
  }
  __syncthreads();   // uc, kP, kL complete
  // camera part of the Cauchy denominator u^T Hcc u over the entries the factors can fill (B.nz_tab: 6147 of the 14706 of the
  // packed triangle; everything else is a structural zero on this path), 12 loads in flight per thread
  double qq = 0.0;
  {
    constexpr int NB = (NZ_N + SCHUR_THREADS - 1) / SCHUR_THREADS;     // 25 at 256 threads
    constexpr int HALF = (NB + 1) / 2;
#pragma unroll 1
    for (int h0 = 0; h0 < NB; h0 += HALF) {
      int code[HALF];
      double hh[HALF];
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        const int e = (h0 + u) * T + tid;
        code[u] = (h0 + u < NB && e < NZ_N) ? B.nz_tab[e] : -1;
      }
#pragma unroll
      for (int u = 0; u < HALF; ++u) hh[u] = code[u] >= 0 ? Hcc[code[u] & 0x3fff] : 0.0;
#pragma unroll
      for (int u = 0; u < HALF; ++u)
        if (code[u] >= 0) {
          const int r = (code[u] >> 14) & 255, c = code[u] >> 22;
          qq += (r == c ? 1.0 : 2.0) * uc[r] * hh[u] * uc[c];
        }
    }
  }
  VPL_STAMP(B, w, 1);
  // ---- regularised landmark blocks: points A = s^2 H + mu d^2 -> row scale s / sqrt(A); lines A_l = S H S + mu D^2 = C C^T ----
  for (int p = tid; p < nP; p += T) {
    const double s = kP[4 * p], d = kP[4 * p + 1], g = kP[4 * p + 2], h = kP[4 * p + 3];
    const double Al = s * s * h + mu * d * d;
    if (!(Al > 0.0)) flag[0] = 1;
    const double smv = s / sqrt(Al);
    pS[p] = smv;
    pE[p] = (s * g / d) / smv;
    pG[p] = smv * (g * d / s);               // g_p = g~ d / s (gradient_ = s g / d)
  }
  for (int l = tid; l < nL; l += T) {
    double Hl[16], s4[4], d4[4], g4[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { s4[a] = kL[28 * l + a]; d4[a] = kL[28 * l + 4 + a]; g4[a] = kL[28 * l + 8 + a]; }
#pragma unroll
    for (int k = 0; k < 16; ++k) Hl[k] = kL[28 * l + 12 + k];
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
      lS[4 * l + a] = s4[a] * rd;
      us[a] = g4[a] / d4[a];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) if (qd < a) lC[l * 10 + tri(a, qd)] = A[tri(a, qd)] * rd;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double s2 = 0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) if (qd >= a) s2 += A[tri(qd, a)] * us[qd];
      lE[4 * l + a] = s2;
    }
    {   // x_g = C^-1 S g_l by the same forward substitution the rows of W go through (g_l = g~ d / s)
      double xg[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double s2 = (s4[a] / A[tri(a, a)]) * (g4[a] * d4[a] / s4[a]);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) if (qd < a) s2 -= (A[tri(a, qd)] / A[tri(a, a)]) * xg[qd];
        xg[a] = s2;
        lG[4 * l + a] = s2;
      }
    }
  }
  __syncthreads();   // constants complete; kP / kL are dead: their space becomes the compact system
  for (int i = tid; i < CSQ_N; i += T) Cacc[i] = 0.0;
  // Flush targets of the accumulator entries (tile t, register v, lane): entry (ca, cb) of the compact product of start
  // frame s goes to row va = map(ca), column vb = map(cb) of the 74 x 80 square, where a pose column maps to 6 s + c and an
  // extrinsic / g / e column to 66.. / 72 / 73: index = idx0 + 6 s * step with step in {0, 1, 80, 81}.  One int per entry in
  // LDS (idx0 | step << 16, -1 = not part of the system) instead of two dozen loop-invariant index registers per lane.
  for (int i = tid; i < NLT * 4 * 64; i += T) {
    const int ln = i & 63, v = (i >> 6) & 3, t = i >> 8;
    int ta = 0, tb = 0;
    tri_decode(t, ta, tb);
    const int ca = 16 * ta + (ln >> 4) + 4 * v, cb = 16 * tb + (ln & 15);
    int code = -1;
    if (ca >= cb && ca < WSv + 2 && cb < WSv) {
      const int a0 = compact2vis(ca, 0, WSv), b0 = compact2vis(cb, 0, WSv);
      const int step = (ca < WSv - 6 ? CSQ_LD : 0) + (cb < WSv - 6 ? 1 : 0);
      code = (a0 * CSQ_LD + b0) | step << 16;
    }
    ftab[i] = code;
  }
  __syncthreads();
  VPL_STAMP(B, w, 2);
  // ---- X^T X on the matrix cores, operands straight from HBM ---------------------------------------------------------
  // v_mfma_f64_16x16x4: the lane (kk, m) supplies A[k = kk][m] and B[kk][m] of the K-step and holds C[kk + 4 v][m].  With
  // x[t] = X[row kk of the K-step][16 t + m] the lower tiles of the compact product are C(ta, tb) += x[ta] (x) x[tb].
  {
    const int m = lane & 15, kk = lane >> 4;
    // the wave's row of the span / ticket table: one int per lane, read with v_readlane
    const int wrow = lane < SK_WSTRIDE ? B.sk_wave[((size_t)w * 8 + wv) * SK_WSTRIDE + lane] : -1;
    const int nchunk = __builtin_amdgcn_readlane(wrow, 0);
    auto fetch = [&](int k) {        // the entry is the same for every lane: scalar registers, uniform branches
      int4 e = etab[k];
      e.x = __builtin_amdgcn_readfirstlane(e.x); e.y = __builtin_amdgcn_readfirstlane(e.y); e.z = __builtin_amdgcn_readfirstlane(e.z);
      return e;
    };
    v4d acc[NLT];
#pragma unroll
    for (int t = 0; t < NLT; ++t) acc[t] = v4d{0, 0, 0, 0};
    const double* Wp = B.Wp + (size_t)w * B.maxP * WS;
    const double* Wl = B.Wl + (size_t)w * B.maxL * 4 * WS;
    const double* gpv = B.gp + (size_t)w * B.maxP;
    const double* glv = B.gl + (size_t)w * B.maxL * 4;
    // A table entry is one K-step: four point rows of one start frame, or the four rows of ONE line.  The lane (kk, m)
    // supplies X[row kk][16 t + m]: a point lane loads one value per column tile; a line lane needs rows 0..kk of the line
    // for the triangular solve with the line block's factor.  The factor of the line is the same for the whole wave: its
    // entries are moved to SCALAR registers (v_readfirstlane) -- as vector registers they, the raw rows and the accumulators
    // do not fit into the 128 registers that two work-groups per CU leave a lane.
    // Branch-free and unconditional: every load is issued with a clamped address, so that the 16 loads of a K-step leave back
    // to back and are waited for once.  The column classes (W column / g column / e column / padding) are applied as 0 / 1
    // MULTIPLIERS held in vector registers: as compare masks they are loop invariants in scalar register pairs, a dozen per
    // inlined copy of this code -- the scalar file overflows, the masks get spilled, and the scheduler, in register-pressure
    // mode, serialises every load behind its select.
    double wm[NTC], gm[NTC], em[NTC];
    int cl[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      const int c = 16 * t + m;
      wm[t] = c < WSv ? 1.0 : 0.0;
      gm[t] = c == WSv ? 1.0 : 0.0;
      em[t] = c == WSv + 1 ? 1.0 : 0.0;
      cl[t] = c < WSv - 6 ? c : (c < WSv ? c + coff : WS - 1);
    }
    // A table entry is four landmarks of one start frame: four points = ONE K-step, four lines = FOUR K-steps (K-step a takes
    // row a of each line).  The lane (kk, m) works on landmark kk of the entry: a point lane loads one value per column tile,
    // a line lane the four rows of ITS line, solves the 4 x 4 triangular system once per column and supplies row a in
    // K-step a -- every loaded value is used once and the solve is not repeated per K-step.
    // (the g column of X is made once per landmark with the constants: the loop loads rows of W and nothing else)
    auto lm_id = [&](const int4 e) { return ((kk & 2 ? e.z : e.y) >> (16 * (kk & 1))) & 0xffff; };
    auto load_raw = [&](const int4 e, double (&raw)[NTC][4]) {
      const bool isl = (e.x & 32) != 0;
      const int idr = lm_id(e);
      const int id = idr != 0xffff ? idr : 0;
      if (isl) {
        const double* Wr = Wl + (unsigned)(id * 4 * WS);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) raw[t][qd] = Wr[(unsigned)(qd * WS + cl[t])];
      } else {
#pragma unroll
        for (int t = 0; t < NTC; ++t) raw[t][0] = Wp[(unsigned)(id * WS + cl[t])];
#pragma unroll
        for (int t = 0; t < NTC; ++t)       // (every element defined on both paths)
#pragma unroll
          for (int qd = 1; qd < 4; ++qd) raw[t][qd] = 0.0;
      }
    };
    // x[t][a] = X[row a of the lane's landmark][16 t + m]  (points: a = 0 only)
    auto transform = [&](const int4 e, const double (&raw)[NTC][4], double (&x)[NTC][4]) {
      const bool isl = (e.x & 32) != 0;
      const int idr = lm_id(e);
      const int id = idr != 0xffff ? idr : 0;
      const double pm = idr != 0xffff ? 1.0 : 0.0;
      if (isl) {
        const double* C = lC + id * 10;
        const double s0 = lS[4 * id] * pm, s1 = lS[4 * id + 1] * pm, s2 = lS[4 * id + 2] * pm, s3 = lS[4 * id + 3] * pm;
        const double c10 = C[1], c20 = C[3], c21 = C[4], c30 = C[6], c31 = C[7], c32 = C[8];
        double ev[4], gv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { ev[a] = lE[4 * id + a] * pm; gv[a] = lG[4 * id + a] * pm; }
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          const double x0 = s0 * raw[t][0];
          const double x1 = s1 * raw[t][1] - c10 * x0;
          const double x2 = s2 * raw[t][2] - c20 * x0 - c21 * x1;
          const double x3 = s3 * raw[t][3] - c30 * x0 - c31 * x1 - c32 * x2;
          x[t][0] = x0 * wm[t] + gv[0] * gm[t] + ev[0] * em[t];
          x[t][1] = x1 * wm[t] + gv[1] * gm[t] + ev[1] * em[t];
          x[t][2] = x2 * wm[t] + gv[2] * gm[t] + ev[2] * em[t];
          x[t][3] = x3 * wm[t] + gv[3] * gm[t] + ev[3] * em[t];
        }
      } else {
        const double sp = pS[id] * pm, ep = pE[id] * pm, gq = pG[id] * pm;
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          x[t][0] = sp * raw[t][0] * wm[t] + gq * gm[t] + ep * em[t];
          x[t][1] = 0.0; x[t][2] = 0.0; x[t][3] = 0.0;
        }
      }
    };
#ifdef VPL_STAMPS
    long long st_flush = 0, st_wait = 0, st_xf = 0, st_mf = 0, st_ld = 0;
#endif
    auto flush = [&](int g, int seq) {
#ifdef VPL_STAMPS
      const long long tf0 = __builtin_readcyclecounter();
#endif
      const int s6 = 6 * (g & 15);
      // targets first (one batch of LDS reads), then the ticket, then 4 NLT hardware adds (ds_add_f64: no read-back round
      // trip).  Only the holder of the ticket adds, one add per address: the order of the terms of every sum is the ticket order.
      int code[NLT * 4];
#pragma unroll
      for (int i = 0; i < NLT * 4; ++i) code[i] = ftab[i * 64 + lane];
      lds_ticket_wait(&tick[0], seq);
#ifdef VPL_STAMPS
      st_wait += __builtin_readcyclecounter() - tf0;
#endif
#pragma unroll
      for (int t = 0; t < NLT; ++t) {
        const double vals[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c2 = code[4 * t + v];
          if (c2 >= 0) lds_add(&Cacc[(c2 & 0xffff) + s6 * (c2 >> 16)], vals[v]);
        }
        acc[t] = v4d{0, 0, 0, 0};
      }
      lds_ticket_pass(&tick[0], seq, lane);
#ifdef VPL_STAMPS
      st_flush += __builtin_readcyclecounter() - tf0;
#endif
    };
    int nent = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ++ch) {
      const int k0 = __builtin_amdgcn_readlane(wrow, __builtin_amdgcn_readfirstlane(1 + 3 * ch));
      const int k1 = __builtin_amdgcn_readlane(wrow, __builtin_amdgcn_readfirstlane(2 + 3 * ch));
      const int seq = __builtin_amdgcn_readlane(wrow, __builtin_amdgcn_readfirstlane(3 + 3 * ch));
      nent += k1 - k0;
      if (MIXED && (fetch(k0).x & 64)) {   // a WIDE chunk: out of line, with a register allocation of its own
        SchurWide A;
        A.etab = etab; A.Wp = Wp; A.Wl = Wl; A.WS = WS; A.lC = lC; A.lS = lS; A.lE = lE; A.lG = lG; A.pS = pS; A.pE = pE; A.pG = pG;
        A.Cacc = Cacc; A.tick = tick;
        schur_wide_chunk(A, k0, k1, seq);
        continue;
      }
      double raw0[NTC][4], raw1[NTC][4], raw2[NTC][4];
      constexpr int NPF = 3;
      load_raw(fetch(k0), raw0);
      load_raw(fetch(min(k0 + 1, k1 - 1)), raw1);
      load_raw(fetch(min(k0 + 2, k1 - 1)), raw2);
      const int gcur = fetch(k0).x;
      auto step = [&](int k, double (&rw)[NTC][4]) {
        const int4 e = fetch(k);
#ifdef VPL_STAMPS
        const long long ts0 = __builtin_readcyclecounter();
#endif
        double x[NTC][4];
        transform(e, rw, x);
#ifdef VPL_STAMPS
        const long long ts1 = __builtin_readcyclecounter();
        st_xf += ts1 - ts0;
#endif
        load_raw(fetch(min(k + NPF, k1 - 1)), rw);      // (past the end: a harmless re-load of the last entry)
#ifdef VPL_STAMPS
        const long long ts2 = __builtin_readcyclecounter();
        st_ld += ts2 - ts1;
#endif
        const bool isl = (e.x & 32) != 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (a == 0 || isl) {
            int t = 0;
#pragma unroll
            for (int ta = 0; ta < NTC; ++ta)
#pragma unroll
              for (int tb = 0; tb <= ta; ++tb, ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ta][a], x[tb][a], acc[t], 0, 0, 0);
          }
        }
#ifdef VPL_STAMPS
        st_mf += __builtin_readcyclecounter() - ts2;
#endif
      };
#pragma unroll 1
      for (int kb = k0; kb < k1; kb += NPF) {
        step(kb, raw0);
        if (kb + 1 < k1) step(kb + 1, raw1);
        if (kb + 2 < k1) step(kb + 2, raw2);
      }
      flush(gcur, seq);
    }
#ifdef VPL_STAMPS
    if (lane == 0 && wv < 2) {
      long long* dg_ = B.dbg + (size_t)w * 64 + 30 + 6 * wv;
      dg_[0] = st_xf; dg_[1] = st_ld; dg_[2] = st_mf; dg_[3] = st_flush; dg_[4] = st_wait; dg_[5] = nent;
    }
#endif
  }
  __syncthreads();
  VPL_STAMP(B, w, 3);
  // Cauchy cross term (W^T u)_b u_c,b from the e-row of the compact system; alpha = |g~|^2 / (u^T H u)
  for (int b = tid; b < NV; b += T) qq += 2.0 * Cacc[(NV + 1) * CSQ_LD + b] * uc[vis2cam(b)];
  double* sacc = B.sacc + (size_t)w * SACC_N;
  for (int i = tid; i < SACC_N; i += T) {
    int a, b;
    tri_decode(i, a, b);
    sacc[i] = Cacc[a * CSQ_LD + b];
  }
  a1 = block_sum(a1, red);
  q = block_sum(q, red);
  qq = block_sum(qq, red);
  if (tid == 0) {
    tr->a1 = a1;
    tr->alpha = a1 / (q + qq);   // DoglegStrategy::ComputeCauchyPoint
    if (flag[0]) B.path[w] = 2;  // a landmark block is not positive definite: the retry loop lives in k_solve
  }
  VPL_STAMP(B, w, 4);
}

template <int NTC>
__global__ __launch_bounds__(SCHUR_THREADS, 2) void k_schur(DevBatch B) {
  extern __shared__ double sm[];
  // the list k_cost of THIS iteration fills is emptied here (k_cost runs after this whole kernel)
  if (blockIdx.x == 0 && threadIdx.x == 0) { B.ord_cnt[2 * ((B.ord_it + 1) & 1)] = 0; B.ord_cnt[2 * ((B.ord_it + 1) & 1) + 1] = 0; }
  schur_body<NTC>(B, ordered_window(B), sm);
}
__global__ __launch_bounds__(SCHUR_THREADS, 2) void k_schur_mixed(DevBatch B) {
  extern __shared__ double sm[];
  if (blockIdx.x == 0 && threadIdx.x == 0) { B.ord_cnt[2 * ((B.ord_it + 1) & 1)] = 0; B.ord_cnt[2 * ((B.ord_it + 1) & 1) + 1] = 0; }
  schur_body<3, true>(B, ordered_window(B), sm);
}
inline size_t schur_smem(int maxP, int maxL, int ntc = 5) {
  const int r1 = std::max(4 * maxP + 28 * maxL, (int)CSQ_N);
  const int maxKS = maxP / 4 + maxL + NF + 2;
  return (size_t)(r1 + 3 * maxP + 22 * maxL + 176 + 24) * sizeof(double) + (size_t)(8 + 4 * maxKS + ntc * (ntc + 1) / 2 * 256) * sizeof(int);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_chol
// ---------------------------------------------------------------------------------------------------------------------
// Elimination order of the reduced camera system H~ = S (Hcc - Schur) S + mu D^2 (Jacobi-scaled space):
//   chain A: speed/bias blocks of the frames 1, 2, 3, 4 (ascending),  chain B: 10, 9, 8, 7, 6 (descending)
//   dense  : [72 pose / extrinsic dims (vis order) | speed/bias 0 (72..80) | speed/bias 5 (81..89) | rhs (90)]
// A chain block s_f touches, when its turn comes, only the next block of its chain, the poses its chain has met so far, the
// rhs and (chain A) speed/bias 0: at most 64 columns, one per lane.  The 9-row strip of the pivot block is held in
// registers (lane j = column j): nine right-looking steps turn it into X = L^-1 [D | R] -- L^T in the pivot lanes, the
// columns of the factor below it in the others.  Lane layouts (the two 9-lane slots alternate between pivot and next block):
//   chain A: slot a 0..8 | slot b 9..17 | speed/bias 0 18..26 | poses 0..5 27..62 | rhs 63
//   chain B: slot a 0..8 | slot b 9..17 | poses 5..10 18..53 | rhs 54
// tools/proto_chain.py is the NumPy statement of the same scheme (fronts, fill, back-substitution).
constexpr int DN = 90;                       // dense dims (rhs row = DN)
constexpr int DNT = 6;                       // 16x16 tiles per dimension of the dense system (96 >= 91)
constexpr int DNAP = DNT * (DNT + 1) / 2 * 256;   // tile-major lower storage (5376 doubles)
#ifdef VPL_CHOL_OCC3                         // A/B switch: three work-groups per CU (53 KB of LDS, 168 registers)
constexpr int XLD = 66;
constexpr int CHOL_MIN_WAVES = 3;
#else
constexpr int XLD = 72;                      // row stride of the chains' X rows in LDS (64 lanes + 8)
constexpr int CHOL_MIN_WAVES = 2;
#endif
constexpr int XROWS_A = 36, XROWS_B = 48;    // 4 x 9 rows ; 5 x 9 rows padded to a multiple of 4

__device__ __forceinline__ int chain_frame(int ch, int b) { return ch == 0 ? 1 + b : 10 - b; }
// cam index of the column of `lane` in block b of chain ch; NC = rhs, -1 = none
__device__ __forceinline__ int chain_col(int ch, int b, int lane) {
  if (lane < 18) {
    const int ds = (b & 1) ? 9 : 0;                       // lanes of the pivot block
    const int f = chain_frame(ch, b);
    const int fr = (lane >= ds && lane < ds + 9) ? f : (ch == 0 ? f + 1 : f - 1);
    return 15 * fr + 6 + (lane < 9 ? lane : lane - 9);
  }
  if (ch == 0) {
    if (lane < 27) return 6 + (lane - 18);
    if (lane < 63) return 15 * ((lane - 27) / 6) + (lane - 27) % 6;
    return NC;
  }
  if (lane < 54) return 15 * (5 + (lane - 18) / 6) + (lane - 18) % 6;
  return lane == 54 ? NC : -1;
}
// dense index of a cam dim (NC = rhs), -1 for the speed/bias blocks the chains eliminate
__device__ __forceinline__ int cam2dense(int c) {
  if (c < 0) return -1;
  if (c == NC) return DN;
  const int v = cam2vis(c);
  if (v >= 0) return v;
  if (c < 15) return 72 + (c - 6);
  return (c >= 81 && c < 90) ? c : -1;
}
__device__ __forceinline__ int dense2cam(int d) { return d < NV ? vis2cam(d) : (d < 81 ? 6 + (d - 72) : d); }
// dense index of lane `lane` of a chain's X rows (the slot lanes hold speed/bias 5 in the chain's last block only; in the
// other blocks their X entries are stored as zeros)
__device__ __forceinline__ int chain_dense(int ch, int lane) {
  if (lane < 18) {
    const bool s5 = ch == 0 ? lane < 9 : lane >= 9;       // chain A ends with its next block in slot a, chain B in slot b
    return s5 ? 81 + (lane < 9 ? lane : lane - 9) : -1;
  }
  return cam2dense(chain_col(ch, 0, lane));
}

__device__ __forceinline__ void chol_body(const DevBatch& B, const int w, double* sm) {
  const int tid = threadIdx.x, T = CHOL_THREADS;
  // Two work-groups share a CU, and the serial work of this kernel sits in "wave 0" and "wave 1" (the chains, the diagonal
  // tiles): with the same roles in both, the two chains of the two windows would share one SIMD while the SIMDs of the
  // partner waves idle.  Work-groups b and b + 256 are the ones that normally meet on a CU (round-robin over 8 XCDs x 32
  // CUs): every second group of 256 rotates its roles by two waves.
  const int lane = tid & 63, wv = ((tid >> 6) + ((blockIdx.x >> 8) & 1) * 2) & 3;
  TrState* tr = &B.tr[w];
  if (tr->status != 0 || tr->reuse != 0 || B.path[w] != 0) return;
  count_active(B, 1);
  constexpr int REG = (XROWS_A + XROWS_B) * XLD > DNAP ? (XROWS_A + XROWS_B) * XLD : DNAP;
  double* XA = sm;                       // chain A's X rows (36 x XLD); chain B's behind them (48 x XLD)
  double* XBm = sm + XROWS_A * XLD;
  double* S = sm;                        // the dense system, tile-major lower, once the chains' products are in registers
  double* scv = sm + REG;                // 176 jacobi scale of the cam dims
  double* dgv = scv + 176;               // 176 dogleg diagonal
  double* ycam = dgv + 176;              // 176 solution, cam-indexed (scaled space)
  double* yv = ycam + 176;               // 96  dense solution
  double* isd = yv + 96;                 // 96  1 / L_jj of the dense factor
  double* Linv = isd + 96;               // 256 inverse of the current diagonal tile's factor
  double* red = Linv + 256;              // 24
  int* flag = (int*)(red + 24);          // [0] failure, [1], [2] blocks finished by chain A / B
  int* cdmap = flag + 8;                 // 2 x 64: dense index of the lanes of the chains' X rows
  double* rsL = (double*)(cdmap + 128);  // 2 x 5 x 9: 1 / L_kk of the chains' pivot blocks (for the back-substitution)

  const size_t fb = (size_t)w * B.nfull;
  const double* gscale = B.scale + fb;
  const double* gdiag = B.diag + fb;
  const double* ggrad = B.grad + fb;
  double* ggn = B.gn + fb;
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  const double* sacc = B.sacc + (size_t)w * SACC_N;
  const double mu = tr->mu;
  for (int c = tid; c < 176; c += T) { scv[c] = c < NC ? gscale[c] : 0.0; dgv[c] = c < NC ? gdiag[c] : 1.0; ycam[c] = 0.0; }
  if (tid < 4) flag[tid] = 0;
  if (tid < 128) cdmap[tid] = chain_dense(tid >> 6, tid & 63);
  for (int i = tid; i < 3 * XLD; i += T) XBm[(XROWS_B - 3) * XLD + i] = 0.0;   // rows 45..47 of chain B: K padding
  __syncthreads();
  VPL_STAMP(B, w, 8);

  // ================= phase 1: the two chains (waves 0, 1) and their X^T X on the matrix cores (waves 2, 3) =============
  // One register array, two tenants: the chain waves keep X of every block for the back-substitution (Xs(b, k) = U[9 b + k]),
  // the partner waves the lower tiles of sum_b X_b^T X_b in lane coordinates (64 x 64: 10 tiles x 4 = U[0..39]).
  double U[45];
#pragma unroll
  for (int i = 0; i < 45; ++i) U[i] = 0.0;
#define Xs(b, k) U[9 * (b) + (k)]
  if (wv < 2) {
    const int ch = wv, nblk = ch == 0 ? 4 : 5;
    double* Xout = ch == 0 ? XA : XBm;
    bool bad = false;
    // original strip of block b: H~[pivot rows][lane's column]; nine independent loads per lane
    auto load_strip = [&](int b, double (&raw)[9]) {
      const int col = chain_col(ch, b, lane);
      const int f = chain_frame(ch, b);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int r = 15 * f + 6 + i;
        raw[i] = 0.0;
        if (col == NC) raw[i] = gc[r];
        else if (col >= 0) raw[i] = Hcc[col > r ? tri(col, r) : tri(r, col)];
      }
    };
    double rawn[9];
    load_strip(0, rawn);
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      if (b < nblk) {
        const int ds = (b & 1) ? 9 : 0, ns = 9 - ds;
        const int col = chain_col(ch, b, lane);
        const int f = chain_frame(ch, b);
        double R[9];
        {
          const double scol = col == NC ? 1.0 : (col >= 0 ? scv[col] : 0.0);
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const int r = 15 * f + 6 + i;
            R[i] = rawn[i] * (scv[r] * scol);
            if (col == r) R[i] += mu * dgv[r] * dgv[r];
          }
        }
        if (b + 1 < nblk) load_strip(b + 1, rawn);     // in flight during this block's elimination
        if (b > 0) {
          // fill from the previous block: R[i][j] -= sum_k X[k][pivot lane i] X[k][j]; the lanes of the previous pivot slot
          // now hold the NEW next block, which the previous block did not touch
          const bool fresh = lane >= ns && lane < ns + 9;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const double xk = fresh ? 0.0 : Xs(b - 1, k);
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] -= readlane_f64(Xs(b - 1, k), ds + i) * xk;
          }
        }
        // nine right-looking steps; the row scaling by 1 / sqrt(pivot) is applied at the end (off the dependent chain)
        double piv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const double pk = readlane_f64(R[k], ds + k);
          piv[k] = pk;
          if (!(pk > 0.0)) bad = true;
          double rinv = __builtin_amdgcn_rcp(pk);
          rinv = fma(rinv, fma(-pk, rinv, 1.0), rinv);
          rinv = fma(rinv, fma(-pk, rinv, 1.0), rinv);
          const double fj = R[k] * rinv;
#pragma unroll
          for (int i = k + 1; i < 9; ++i) R[i] -= readlane_f64(R[i], ds + k) * fj;
        }
        const bool dcol = cam2dense(col) >= 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          // 1 / sqrt(pivot) by v_rsq_f64 + two Newton steps (an IEEE sqrt and divide are ~25 dependent instructions)
          double rs = __builtin_amdgcn_rsq(piv[k]);
          rs = rs * fma(-0.5 * piv[k] * rs, rs, 1.5);
          rs = rs * fma(-0.5 * piv[k] * rs, rs, 1.5);
          if (lane == 0) rsL[(5 * ch + b) * 9 + k] = rs;
          const double xv = R[k] * rs;
          Xs(b, k) = col >= 0 ? xv : 0.0;
          Xout[(9 * b + k) * XLD + lane] = dcol ? xv : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&flag[1 + ch], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (bad && lane == 0) flag[0] = 1;
  } else {
    // partner of chain wv - 2: consumes the X rows K-step by K-step as the blocks are finished
    const int ch = wv - 2, nblk = ch == 0 ? 4 : 5;
    const double* Xin = ch == 0 ? XA : XBm;
    const int m = lane & 15, kk = lane >> 4;
    const int nks = (ch == 0 ? XROWS_A : XROWS_B) / 4;
    int have = 0;
    v4d pacc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) pacc[t] = v4d{0, 0, 0, 0};
    for (int ks = 0; ks < nks; ++ks) {
      const int need = min(nblk, (4 * ks + 3) / 9 + 1);
      while (have < need) {
        have = __hip_atomic_load(&flag[1 + ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (have >= need) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      const double* row = Xin + (4 * ks + kk) * XLD + m;
      const double x0 = row[0], x1 = row[16], x2 = row[32], x3 = row[48];
      pacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, pacc[0], 0, 0, 0);
      pacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, pacc[1], 0, 0, 0);
      pacc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, pacc[2], 0, 0, 0);
      pacc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x0, pacc[3], 0, 0, 0);
      pacc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x1, pacc[4], 0, 0, 0);
      pacc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x2, pacc[5], 0, 0, 0);
      pacc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x0, pacc[6], 0, 0, 0);
      pacc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x1, pacc[7], 0, 0, 0);
      pacc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x2, pacc[8], 0, 0, 0);
      pacc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x3, pacc[9], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) { U[4 * t] = pacc[t].x; U[4 * t + 1] = pacc[t].y; U[4 * t + 2] = pacc[t].z; U[4 * t + 3] = pacc[t].w; }
  }
  __syncthreads();   // the X rows are dead: their space becomes the dense system
  VPL_STAMP(B, w, 9);

  // ================= phase 2: the dense system in LDS ================================================================
  // Every global operand of this thread is requested first (one batch: ~30 loads in flight), the LDS image is zeroed
  // meanwhile, then the entries are scaled and stored.
  auto put = [&](int r, int c, double v) {             // r >= c; diagonal tiles are kept as full squares
    S[tix(r, c)] = v;
    if ((r >> 4) == (c >> 4) && r != c) S[tix(c, r)] = v;
  };
  {
    constexpr int NVP = NV * (NV + 1) / 2;
    constexpr int NB1 = (NVP + CHOL_THREADS - 1) / CHOL_THREADS;          // 11 entries of the vis triangle per thread
    constexpr int NB2 = (19 * 96 + CHOL_THREADS - 1) / CHOL_THREADS;      // 8 entries of the speed/bias rows + rhs row
    double hh[NB1], ss[NB1], h2[NB2], s2[NB2];
    int rc1[NB1], rc2[NB2];
#pragma unroll
    for (int u = 0; u < NB1; ++u) {
      const int e = u * T + tid;
      hh[u] = 0.0; ss[u] = 0.0; rc1[u] = -1;
      if (e < NVP) {
        int vr, vc;
        tri_decode(e, vr, vc);
        rc1[u] = vr | vc << 8;
        hh[u] = Hcc[tri(vis2cam(vr), vis2cam(vc))];
        ss[u] = sacc[e];
      }
    }
#pragma unroll
    for (int u = 0; u < NB2; ++u) {
      const int idx = u * T + tid;
      const int i = idx / 96, dc = idx - 96 * i;
      const int dr = 72 + i;
      h2[u] = 0.0; s2[u] = 0.0; rc2[u] = -1;
      if (idx < 19 * 96 && dc <= dr && dc < DN) {
        rc2[u] = dr | dc << 8;
        const int c = dense2cam(dc);
        if (dr < DN) {
          const int r = dense2cam(dr);
          h2[u] = Hcc[r >= c ? tri(r, c) : tri(c, r)];
        } else {
          h2[u] = gc[c];
          if (dc < NV) s2[u] = sacc[tri(NV, dc)];
        }
      }
    }
    for (int i = tid; i < DNAP; i += T) S[i] = 0.0;
    __syncthreads();
    VPL_STAMP(B, w, 14);
    // (i) pose / extrinsic block: (Hcc - Schur) in the scaled space + mu D^2; the packed index of the 72 x 72 vis triangle is
    //     the index into the compact Schur product
#pragma unroll
    for (int u = 0; u < NB1; ++u)
      if (rc1[u] >= 0) {
        const int vr = rc1[u] & 255, vc = rc1[u] >> 8;
        const int r = vis2cam(vr), c = vis2cam(vc);
        double v = (hh[u] - ss[u]) * (scv[r] * scv[c]);
        if (r == c) v += mu * dgv[r] * dgv[r];
        put(vr, vc, v);
      }
    // (ii) rows of speed/bias 0 and 5, (iii) the rhs row
#pragma unroll
    for (int u = 0; u < NB2; ++u)
      if (rc2[u] >= 0) {
        const int dr = rc2[u] & 255, dc = rc2[u] >> 8;
        const int c = dense2cam(dc);
        if (dr < DN) {
          const int r = dense2cam(dr);
          double v = h2[u] * (scv[r] * scv[c]);
          if (r == c) v += mu * dgv[r] * dgv[r];
          put(dr, dc, v);
        } else {
          put(DN, dc, (h2[u] - s2[u]) * scv[c]);
        }
      }
    if (tid < 6) S[tix(DN + tid, DN + tid)] = 1.0;      // rhs row and padding rows: unit diagonal, never a pivot
  }
  __syncthreads();
  VPL_STAMP(B, w, 15);
  // (iv) minus the chains' products.  An entry gets a term from both chains only when both of its dims are met by both:
  // pose 5 (vis 30..35), speed/bias 5 (81..89), the rhs (90).  Pass 0: chain A's partner subtracts everything it has, chain B's
  // partner everything outside that overlap -- side by side, on different entries; pass 1: chain B's partner subtracts its
  // overlap entries.  Every entry sees its terms in the order A, B.
  auto in_both = [](int d) { return (d >= 30 && d < 36) || d >= 81; };
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if ((wv == 2 && pass == 0) || wv == 3) {
      const int ch = wv - 2;
      const int m = lane & 15, kk = lane >> 4;
      int t = 0;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb, ++t) {
          const int j = 16 * tb + m, dj = cdmap[64 * ch + j];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = 16 * ta + kk + 4 * v;
            if (i < j) continue;
            const int di = cdmap[64 * ch + i];
            if (di < 0 || dj < 0) continue;
            if (ch == 1 && (in_both(di) && in_both(dj)) != (pass == 1)) continue;
            const int r = di > dj ? di : dj, c = di > dj ? dj : di;
            if (c >= DN) continue;                       // (rhs, rhs) is not part of the system
            const double val = U[4 * t + v];
            S[tix(r, c)] -= val;
            if ((r >> 4) == (c >> 4) && r != c) S[tix(c, r)] -= val;
          }
        }
    }
    __syncthreads();
  }
  VPL_STAMP(B, w, 10);

  // ================= phase 3: left-looking tile Cholesky of the dense system (the scheme of ba_solve.h, 6 tile columns,
  // 4 waves); the rhs row rides along ===================================================================================
  {
    constexpr int NWV = CHOL_THREADS / 64;
    for (int K = 0; K < DNT; ++K) {
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
      if (wv == 0) {
        double* D = S + ((K * (K + 1) / 2 + K) << 8);
        const int r4 = lane >> 4, cc = lane & 15;
        double d[4], mm[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { d[v] = D[tsw(r4 + 4 * v, cc)]; mm[v] = (r4 + 4 * v == cc) ? 1.0 : 0.0; }
        const int ncol = min(16, DN - 16 * K);
        double pivc = 1.0;
        bool bad = false;
#define VPL_DSTEP(J) diag_tile_step<J>(d, mm, r4, cc, ncol, pivc, bad);
        VPL_DSTEP(0) VPL_DSTEP(1) VPL_DSTEP(2) VPL_DSTEP(3) VPL_DSTEP(4) VPL_DSTEP(5) VPL_DSTEP(6) VPL_DSTEP(7)
        VPL_DSTEP(8) VPL_DSTEP(9) VPL_DSTEP(10) VPL_DSTEP(11) VPL_DSTEP(12) VPL_DSTEP(13) VPL_DSTEP(14) VPL_DSTEP(15)
#undef VPL_DSTEP
        if (bad) {
          if (lane == 0) flag[0] = 1;
        } else {
          const double sq = sqrt(pivc);
          if (cc < ncol) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int r = r4 + 4 * v;
              if (r > cc) D[tsw(r, cc)] = d[v] / sq;
              else if (r == cc) { const double id = 1.0 / sq; isd[16 * K + cc] = id; D[tsw(r, cc)] = 1.0 / id; }
            }
          }
          if (K < DNT - 1) {
            const double isq = 1.0 / sq;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int r = r4 + 4 * v;
              const double ir = __shfl(isq, r, 64);
              Linv[tsw(r, cc)] = cc <= r ? mm[v] * ir : 0.0;
            }
          }
        }
      } else if (K >= 1) {
        for (int I = K + 1 + (wv - 1); I < DNT; I += NWV - 1) rank_update(K, I, K - 1, K);
        if (K + 1 < DNT)
          for (int I = K + 1 + (wv - 1); I < DNT; I += NWV - 1) rank_update(K + 1, I, 0, K);
      }
      __syncthreads();
      if (flag[0]) break;
      for (int I = K + 1 + wv; I < DNT; I += NWV) {
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
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          rank_update(K + 1, K + 1, K, K + 1);
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (flag[0]) {   // LINEAR_SOLVER_FAILURE: the retry with a larger mu is k_solve's loop
    if (tid == 0) B.path[w] = 2;
    return;
  }
  VPL_STAMP(B, w, 11);
  // ================= phase 4: back substitution L^T y = z of the dense part =============================================
  for (int c = tid; c < 96; c += T) yv[c] = c < DN ? S[tix(DN, c)] : 0.0;
  __syncthreads();
  for (int K = DNT - 1; K >= 0; --K) {
    const double* D = S + ((K * (K + 1) / 2 + K) << 8);
    if (wv == 0) {
      const int ncol = min(16, DN - 16 * K);
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
  for (int d = tid; d < DN; d += T) ycam[dense2cam(d)] = yv[d];
  __syncthreads();
  VPL_STAMP(B, w, 12);
  // ================= phase 5: the chains backwards ===================================================================
  if (wv < 2) {
    const int ch = wv, nblk = ch == 0 ? 4 : 5;
#pragma unroll
    for (int b = 4; b >= 0; --b) {
      if (b < nblk) {
        const int ds = (b & 1) ? 9 : 0;
        const int col = chain_col(ch, b, lane);
        const int f = chain_frame(ch, b);
        const bool pivl = lane >= ds && lane < ds + 9;
        // t = z - sum over the other columns of X[.][j] y_j
        const double cf = col == NC ? 1.0 : ((col < 0 || pivl) ? 0.0 : -ycam[col]);
        double t[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = wave_sum_dpp(Xs(b, k) * cf);
        // L^T y = t with L^T[k][j] = X[k][pivot lane j]
#pragma unroll
        for (int j = 8; j >= 0; --j) {
          const double yj = t[j] * rsL[(5 * ch + b) * 9 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) t[k] -= readlane_f64(Xs(b, k), ds + j) * yj;
          t[j] = yj;
        }
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < 9; ++j) ycam[15 * f + 6 + j] = t[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
    }
  }
  __syncthreads();
  // ================= outputs: S_c y_c for the landmark back-substitution, the Gauss-Newton step of the cam dims ==========
  double a2 = 0.0, a3 = 0.0;
  double* ycs = B.ycs + (size_t)w * 176;
  for (int c = tid; c < 176; c += T) {
    double u = 0.0;
    if (c < NC) {
      const double y = ycam[c];
      u = scv[c] * y;
      const double gnv = -dgv[c] * y;
      ggn[c] = gnv;
      a2 += gnv * gnv;
      a3 += ggrad[c] * gnv;
    }
    ycs[c] = u;
  }
  a2 = block_sum(a2, red);
  a3 = block_sum(a3, red);
  if (tid == 0) { B.sx[(size_t)w * 8] = a2; B.sx[(size_t)w * 8 + 1] = a3; }
  VPL_STAMP(B, w, 13);
}
#undef Xs
__global__ __launch_bounds__(CHOL_THREADS, CHOL_MIN_WAVES) void k_chol(DevBatch B) {
  extern __shared__ double sm[];
  chol_body(B, ordered_window(B), sm);
}
constexpr size_t CHOL_SMEM = (size_t)(((XROWS_A + XROWS_B) * XLD > DNAP ? (XROWS_A + XROWS_B) * XLD : DNAP) + 3 * 176 + 2 * 96 + 256 + 24 + 96) * sizeof(double) + (8 + 128) * sizeof(int);

// ---------------------------------------------------------------------------------------------------------------------
// k_back : landmark back-substitution y_l = A_l^-1 S_l (g_l - W_l S_c y_c), then DoglegStrategy::ComputeTraditionalDoglegStep,
// the model cost change and the candidate x (+) delta -- for a window that re-uses the Gauss-Newton step of a rejected
// iteration only the latter (the arithmetic is the one of ba_solve.h).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void back_body(const DevBatch& B, const int w, double* sm) {
  const int tid = threadIdx.x, T = BACK_THREADS;
  const int lane = tid & 63;
  TrState* tr = &B.tr[w];
  if (tr->status != 0) return;
  const int path = B.path[w];
  if (path != 0) {                       // k_solve did this window's whole step
    __syncthreads();                     // (everybody has read the flag)
    if (tid == 0 && path == 2) B.path[w] = 0;
    return;
  }
  const int nP = B.nP[w], nL = B.nL[w];
  const int WS = B.WS;
  double* uc = sm;                       // 176 S_c y_c
  double* lrhs = uc + 176;               // 4 maxL
  double* lgn = lrhs + 4 * B.maxL;       // nfull Gauss-Newton step (scaled space)
  double* gdelta = lgn + B.nfull;        // nfull
  double* red = gdelta + B.nfull;        // 24
  int* pSt = (int*)(red + 24);
  int* lSt = pSt + B.maxP;
  const size_t fb = (size_t)w * B.nfull;
  const double* gscale = B.scale + fb;
  const double* gdiag = B.diag + fb;
  const double* ggrad = B.grad + fb;
  double* ggn = B.gn + fb;
  const double* lch = B.lchol + (size_t)w * B.maxL * 10;
  const int LP = NC, LL = NC + B.maxP;
  const bool reuse0 = tr->reuse != 0;
  count_active(B, reuse0 ? 2 : 1);
  VPL_STAMP(B, w, 5);
  if (!reuse0) {
    const double mu = tr->mu;
    for (int p = tid; p < nP; p += T) pSt[p] = B.pt_start[(size_t)w * B.maxP + p];
    for (int l = tid; l < nL; l += T) lSt[l] = B.ln_start[(size_t)w * B.maxL + l];
    for (int c = tid; c < 176; c += T) uc[c] = c < NV ? B.ycs[(size_t)w * 176 + vis2cam(c)] : 0.0;   // vis order: W's column order
    for (int c = tid; c < NC; c += T) lgn[c] = ggn[c];
    __syncthreads();
    double a2 = 0.0, a3 = 0.0;
    {
      const int sub = lane & 7, grp = tid >> 3;   // 64 row groups per pass
      const int nblk = WS / 6;
      constexpr int NPH = 4;   // row groups per trip: their loads are in flight together
      for (int p0 = 0; p0 < nP; p0 += NPH * (T / 8)) {
        double wyv[NPH], sv[NPH], dv[NPH], hv[NPH], gv2[NPH], grv[NPH];
        size_t piv[NPH];
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
          const int p = p0 + h * (T / 8) + grp;
          piv[h] = (size_t)w * B.maxP + (p < nP ? p : 0);
          wyv[h] = 0.0; sv[h] = dv[h] = 1.0; hv[h] = gv2[h] = grv[h] = 0.0;
          if (p < nP) {
            const int s0 = pSt[p];
            for (int blk = sub; blk < nblk; blk += 8) {
              const bool exb = blk == nblk - 1;
              const int vb = exb ? 66 : 6 * (s0 + blk);
              if (!exb && vb >= 66) continue;
              const double* Wr = B.Wp + piv[h] * WS + 6 * blk;
#pragma unroll
              for (int k = 0; k < 6; ++k) wyv[h] += Wr[k] * uc[vb + k];
            }
            if (sub == 0) {
              sv[h] = gscale[LP + p]; dv[h] = gdiag[LP + p]; hv[h] = B.Hpp[piv[h]]; gv2[h] = B.gp[piv[h]];
              grv[h] = ggrad[LP + p];
            }
          }
        }
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
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
      constexpr int NLH = 5;
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
              for (int k = 0; k < 6; ++k) wyv[h] += Wr[k] * uc[vb + k];
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
        double C[10], t4[4], gd4[4], gr4[4];
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
    a2 = block_sum(a2, red);
    a3 = block_sum(a3, red);
    if (tid == 0) {
      tr->a2 = a2 + B.sx[(size_t)w * 8];        // camera part from k_chol
      tr->a3 = a3 + B.sx[(size_t)w * 8 + 1];
      tr->reuse = 1;   // DoglegStrategy::ComputeStep sets reuse_ = true
    }
    __syncthreads();
  }
  VPL_STAMP(B, w, 6);
  // ---- DoglegStrategy::ComputeTraditionalDoglegStep ---------------------------------------------
  const double radius = tr->radius, alpha = tr->alpha, a1 = tr->a1, a2 = tr->a2, a3 = tr->a3, mu = tr->mu;
  const double gradient_norm = sqrt(a1), gauss_newton_norm = sqrt(a2);
  double c1, c2, dnorm;
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
  const double q_cauchy = a1 / alpha;
  const double sg = -c1 * a1 + c2 * a3;
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
      if (tr->num_invalid >= kMaxInvalid) { tr->status = 2; tr->iter -= 1; }
      else if (tr->iter >= B.opt.num_iterations) tr->status = 3;
      tr->mu *= kMuIncrease;
      tr->reuse = 0;
    }
    return;
  }
  const int nfull_used = NC + B.maxP + 4 * nL;
  for (int k = tid; k < nfull_used; k += T) {
    const bool live = k < NC || (k >= LP && k < LP + nP) || k >= LL;
    if (live) gdelta[k] = gscale[k] * (-c1 * ggrad[k] + c2 * (reuse0 ? ggn[k] : lgn[k])) / gdiag[k];
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
#ifndef VPL_BACK_WAVES
#define VPL_BACK_WAVES 2          // A/B switch: waves per SIMD the register allocation of k_back is held to
#endif
__global__ __launch_bounds__(BACK_THREADS, VPL_BACK_WAVES) void k_back(DevBatch B) {
  extern __shared__ double sm[];
  back_body(B, ordered_window(B), sm);
}
inline size_t back_smem(int maxP, int maxL) {
  const int nfull = NC + maxP + 4 * maxL;
  return (size_t)(176 + 4 * maxL + 2 * nfull + 24) * sizeof(double) + (size_t)(maxP + maxL) * sizeof(int);
}

}  // namespace vpl
