// Block / wave level helpers shared by the BA kernels (wave64, gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include "ba_types.h"
#include "vpl_math.h"

namespace vpl {

#ifdef VPL_STAMPS
#define VPL_STAMP(B, w, i) do { if (threadIdx.x == 0) (B).dbg[(size_t)(w) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define VPL_STAMP(B, w, i) do {} while (0)
#endif

// one workgroup reports that it did the work of kind k (0 k_lin, 1 k_solve new step, 2 k_solve re-used step, 3 k_cost)
__device__ __forceinline__ void count_active(const DevBatch& B, int k) {
  if (B.act && threadIdx.x == 0 && B.launch < ACT_SLOTS) atomicAdd(&B.act[4 * B.launch + k], 1);
}

// window of this work-group in a launch of trust-region iteration B.ord_it (identity before the first k_cost)
// (readfirstlane: the index is the same for the whole work-group, but it comes out of a load -- without the hint every
// per-window pointer and count derived from it lives in vector registers and all their address arithmetic runs on the VALU)
__device__ __forceinline__ int ordered_window(const DevBatch& B) {
  const int w = B.ord_it == 0 ? (int)blockIdx.x : B.order[(size_t)(B.ord_it & 1) * B.nW + blockIdx.x];
  return __builtin_amdgcn_readfirstlane(w);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over each row of 16 lanes with DPP row_shr (no LDS / bpermute traffic); valid in lane 15 of the row
template <int CTRL>
__device__ __forceinline__ double dpp_shr_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double v) {
  v += dpp_shr_f64<0x111>(v);   // row_shr:1
  v += dpp_shr_f64<0x112>(v);   // row_shr:2
  v += dpp_shr_f64<0x114>(v);   // row_shr:4
  v += dpp_shr_f64<0x118>(v);   // row_shr:8
  return v;
}

// sum over the wave without LDS traffic: DPP row sums, then the four row leaders through the scalar unit (uniform result)
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = row_sum16(v);
  double t = 0.0;
#pragma unroll
  for (int l = 15; l < 64; l += 16)
    t += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  return t;
}

// sum over the whole workgroup; `red` is LDS scratch of >= 17 doubles; result broadcast
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < nw; ++i) s += red[i];
    red[16] = s;
  }
  __syncthreads();
  return red[16];
}

// hardware LDS f64 add (ds_add_f64)
__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// cam index -> vis index or -1
__host__ __device__ __forceinline__ int cam2vis(int c) {
  if (c >= 165) return 66 + (c - 165);
  int o = c % 15;
  return o < 6 ? 6 * (c / 15) + o : -1;
}

// decode a packed lower-triangular index into (r, c), r >= c
__device__ __forceinline__ void tri_decode(int idx, int& r, int& c) {
  // single-precision root (one v_sqrt_f32) as the first guess; the two loops make it exact for any idx < 2^23
  int rr = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
  while ((rr + 1) * (rr + 2) / 2 <= idx) ++rr;
  while (rr * (rr + 1) / 2 > idx) --rr;
  r = rr;
  c = idx - rr * (rr + 1) / 2;
}

__device__ __forceinline__ PreInt load_preint(const DevPreint& d) {
  PreInt p;
  p.sum_dt = d.sum_dt;
  p.dp = V3{d.dp[0], d.dp[1], d.dp[2]};
  p.dv = V3{d.dv[0], d.dv[1], d.dv[2]};
  p.dq = Q4{d.dq[3], d.dq[0], d.dq[1], d.dq[2]};
  p.lba = V3{d.lba[0], d.lba[1], d.lba[2]};
  p.lbg = V3{d.lbg[0], d.lbg[1], d.lbg[2]};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    p.dp_dba.m[k] = d.dp_dba[k];
    p.dp_dbg.m[k] = d.dp_dbg[k];
    p.dq_dbg.m[k] = d.dq_dbg[k];
    p.dv_dba.m[k] = d.dv_dba[k];
    p.dv_dbg.m[k] = d.dv_dbg[k];
  }
  return p;
}

// dx of one kept prior block (MarginalizationFactor::Evaluate, marginalization_factor.cpp:502-520)
__device__ __forceinline__ void prior_block_dx(int kind, const double* x, const double* x0, double* dx) {
  if (kind == 1) {  // speed/bias: plain difference
#pragma unroll
    for (int k = 0; k < 9; ++k) dx[k] = x[k] - x0[k];
  } else {
    dx[0] = x[0] - x0[0]; dx[1] = x[1] - x0[1]; dx[2] = x[2] - x0[2];
    Q4 dq = qmul(qinv(qpose(x0)), qpose(x));
    double sgn = (dq.w >= 0) ? 2.0 : -2.0;
    dx[3] = sgn * dq.x; dx[4] = sgn * dq.y; dx[5] = sgn * dq.z;
  }
}

// cross-lane helpers of the register-resident diagonal-tile factorisation
__device__ __forceinline__ double readlane_f64(double v, int srclane) {   // srclane uniform (compile-time after unrolling)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ double swizzle_row_f64(double v) {   // value of lane (lane & 48) | J: DPP row_newbcast:J (gfx90a+),
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + J, 0xf, 0xf, false);   // VALU speed, no LDS path
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + J, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// one column step of the right-looking factorisation of the 16x16 diagonal tile held in registers.  The same row
// operations are applied to an identity matrix m (same lane layout): after the 16 steps m = L~^-1 with L~ the unit lower
// factor, so L^-1 = diag(1 / sqrt(pivot)) m comes out of the factorisation for the price of one more bpermute per step
// (independent of the first) -- and the rows below the tile become a matrix product instead of 16-step substitutions.
template <int J>
__device__ __forceinline__ void diag_tile_step(double (&d)[4], double (&m)[4], int r4, int cc, int ncol, double& pivc, bool& bad) {
  if (J < ncol && !bad) {
    const double ajj = readlane_f64(d[J >> 2], ((J & 3) << 4) | J);
    if (!(ajj > 0.0)) {
      bad = true;
    } else {
      if (cc == J) pivc = ajj;
      // 1 / a_JJ by v_rcp_f64 + two Newton steps (5 dependent instructions; an IEEE divide is ~10 on this chain).
      // Measured alternatives, all slower on this lone wave: branch-free steps with the products formed ahead of the
      // reciprocal, v_permlane swaps instead of ds_bpermute, a column-per-lane layout with v_readlane / DPP broadcasts.
      double rinv = __builtin_amdgcn_rcp(ajj);
      rinv = fma(rinv, fma(-ajj, rinv, 1.0), rinv);
      rinv = fma(rinv, fma(-ajj, rinv, 1.0), rinv);
      const double f = __shfl(d[J >> 2], ((J & 3) << 4) | cc, 64) * rinv;            // D[J][cc] / a_JJ
      const double f2 = __shfl(m[J >> 2], ((J & 3) << 4) | cc, 64) * rinv;           // M[J][cc] / a_JJ
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double cv = swizzle_row_f64<J>(d[v]);                                 // D[r4 + 4v][J]
        if (r4 + 4 * v > J) {
          if (cc > J) d[v] -= cv * f;
          m[v] -= cv * f2;
        }
      }
    }
  }
}


// Cholesky of a small SPD matrix (n <= 16, row-major, leading dimension ld, in LDS) by ONE wave with the register tile
// algorithm above: A = L L^T.  Writes L (lower, zeros above the diagonal) back into A and, if Xinv != nullptr, L^-1 (lower)
// into Xinv (same shape).  All 64 lanes of the wave must call it.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool wave_chol16(double* A, int n, int ld, double* Xinv, int lane) {
  const int r4 = lane >> 4, cc = lane & 15;
  double d[4], m[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = r4 + 4 * v;
    d[v] = (r < n && cc < n) ? A[r * ld + cc] : (r == cc ? 1.0 : 0.0);     // identity padding
    m[v] = r == cc ? 1.0 : 0.0;
  }
  double pivc = 1.0;
  bool bad = false;
#define VPL_DSTEP(J) diag_tile_step<J>(d, m, r4, cc, 16, pivc, bad);
  VPL_DSTEP(0) VPL_DSTEP(1) VPL_DSTEP(2) VPL_DSTEP(3) VPL_DSTEP(4) VPL_DSTEP(5) VPL_DSTEP(6) VPL_DSTEP(7)
  VPL_DSTEP(8) VPL_DSTEP(9) VPL_DSTEP(10) VPL_DSTEP(11) VPL_DSTEP(12) VPL_DSTEP(13) VPL_DSTEP(14) VPL_DSTEP(15)
#undef VPL_DSTEP
  if (bad) return false;
  const double sq = sqrt(pivc), isq = 1.0 / sq;       // pivot of column cc
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = r4 + 4 * v;
    const double ir = __shfl(isq, r, 64);             // 1 / sqrt(pivot of row r) sits in lane r
    if (r < n && cc < n) {
      A[r * ld + cc] = r > cc ? d[v] / sq : (r == cc ? sq : 0.0);
      if (Xinv) Xinv[r * ld + cc] = r >= cc ? m[v] * ir : 0.0;
    }
  }
  return true;
}

}  // namespace vpl
