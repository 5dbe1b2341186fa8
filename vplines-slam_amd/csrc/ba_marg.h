// k_gauge : Estimator::double2vector2 (estimator.cpp:810-900) -- yaw/position gauge fix of the
//           whole window, lines carried through setLineOrth (feature_manager.cpp:367-388), then
//           the vector2double() round trip the reference performs before marginalising (:1233).
// k_marg  : MarginalizationInfo::marginalize (marginalization_factor.cpp:177-363) on the normal
//           equations k_lin<MARG> assembled: landmark elimination, eigen pseudo-inverse of the
//           oldest frame's 15 dims, eigen decomposition of the kept block -> J0, r0.
#pragma once
#include "ba_common.h"

namespace vpl {

__device__ __forceinline__ void gauge_body(const DevBatch& B, const int w) {
  const int tid = threadIdx.x;
  __shared__ double rd[9], p0a[3], p0b[3], newpose[NF * 7], newex[7];
  const double* gz = B.gauge + (size_t)w * 4;
  double* pose = B.pose + (size_t)w * 77;
  double* sb = B.sb + (size_t)w * 99;
  double* ex = B.ex + (size_t)w * 7;
  if (tid == 0) {
    V3 ypr00 = R2ypr(qmat(qpose(pose)));
    double y_diff = gz[0] - ypr00.x;
    M3 R = ypr2R(V3{y_diff, 0, 0});
    for (int k = 0; k < 9; ++k) rd[k] = R.m[k];
    for (int k = 0; k < 3; ++k) { p0a[k] = pose[k]; p0b[k] = gz[1 + k]; }
  }
  __syncthreads();
  M3 rot_diff;
  for (int k = 0; k < 9; ++k) rot_diff.m[k] = rd[k];
  V3 P0a{p0a[0], p0a[1], p0a[2]}, P0b{p0b[0], p0b[1], p0b[2]};
  if (tid < NF) {
    const double* x = pose + 7 * tid;
    M3 Rs = mul(rot_diff, qmat(qnormalized(qpose(x))));
    V3 Ps = mul(rot_diff, V3{x[0] - P0a.x, x[1] - P0a.y, x[2] - P0a.z}) + P0b;
    Q4 q = mat2q(Rs);   // what the next vector2double() writes
    double* o = newpose + 7 * tid;
    o[0] = Ps.x; o[1] = Ps.y; o[2] = Ps.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    double* s = sb + 9 * tid;
    V3 V = mul(rot_diff, V3{s[0], s[1], s[2]});
    s[0] = V.x; s[1] = V.y; s[2] = V.z;
  }
  if (tid == NF) {
    Q4 q = mat2q(qmat(qpose(ex)));   // ric = q.toRotationMatrix(); para = Quaterniond(ric)
    for (int k = 0; k < 3; ++k) newex[k] = ex[k];
    newex[3] = q.x; newex[4] = q.y; newex[5] = q.z; newex[6] = q.w;
  }
  __syncthreads();
  // lines: orth (world, optimised gauge) -> rotate back -> start camera frame -> orth again
  const int nL = B.nL[w];
  V3 tw1b = P0a;
  V3 twow1 = -mul(rot_diff, tw1b) + P0b;
  for (int l = tid; l < nL; l += blockDim.x) {
    const size_t li = (size_t)w * B.maxL + l;
    Plk Lw1 = orth_to_plk(B.orth + li * 4);
    Plk Lwo = plk_to_pose(Lw1, rot_diff, twow1);
    double o4[4];
    plk_to_orth(Lwo, o4);
    Plk Lw = orth_to_plk(o4);
    const int s = B.ln_start[li];
    // Rs[s] as stored by double2vector2 is rot_diff * R(q): rebuild it the same way from the OLD pose
    const double* xo = pose + 7 * s;
    M3 Rs = mul(rot_diff, qmat(qnormalized(qpose(xo))));
    V3 Ps{newpose[7 * s], newpose[7 * s + 1], newpose[7 * s + 2]};
    M3 ric = qmat(qpose(ex));
    V3 tic{ex[0], ex[1], ex[2]};
    V3 twc = Ps + mul(Rs, tic);
    M3 Rwc = mul(Rs, ric);
    Plk Lc = plk_from_pose(Lw, Rwc, twc);
    double* pl = B.plk + li * 6;
    pl[0] = Lc.n.x; pl[1] = Lc.n.y; pl[2] = Lc.n.z; pl[3] = Lc.v.x; pl[4] = Lc.v.y; pl[5] = Lc.v.z;
    // vector2double() before the marginalisation: getLineOrthVector on the updated state
    Plk Lw2 = plk_to_pose(Lc, Rwc, twc);
    plk_to_orth(Lw2, B.orth + li * 4);
    {   // the marginalisation pass of k_lin reads the Pluecker image of these parameters (B.lw)
      const Plk Lq = orth_to_plk(B.orth + li * 4);
      double* o = B.lw + li * 6;
      o[0] = Lq.n.x; o[1] = Lq.n.y; o[2] = Lq.n.z; o[3] = Lq.v.x; o[4] = Lq.v.y; o[5] = Lq.v.z;
    }

    // FeatureManager::removeLineOutlier (feature_manager.cpp:702-798) on the gauge-fixed state
    int erase = 0;
    if (B.opt.remove_line_outliers) {
      const double* ob0 = B.ln_obs + ((size_t)w * B.maxLO + B.ln_off[li]) * 8;
      const V3 p11{ob0[0], ob0[1], 1.0}, p21{ob0[2], ob0[3], 1.0};
      const V3 cr = cross(p11, p21);
      const double lnn = sqrt(cr.x * cr.x + cr.y * cr.y);
      const double lx = cr.x / lnn, ly = cr.y / lnn;
      const V3 p12{p11.x + lx, p11.y + ly, 1.0}, p22{p21.x + lx, p21.y + ly, 1.0};
      const V3 cam{0, 0, 0};
      // pi_from_ppp (line_geometry.cpp:134-139) ; Lc = [skew(nc) vc; -vc^T 0]
      const V3 n1 = cross(cam - p12, p11 - p12), n2 = cross(cam - p22, p21 - p22);
      const double d1 = -dot(p12, cross(cam, p11)), d2 = -dot(p22, cross(cam, p21));
      const V3 e1x = mul(skew(Lc.n), n1) + Lc.v * d1, e2x = mul(skew(Lc.n), n2) + Lc.v * d2;
      const double e1w = -dot(Lc.v, n1), e2w = -dot(Lc.v, n2);
      const V3 e1{e1x.x / e1w, e1x.y / e1w, e1x.z / e1w}, e2{e2x.x / e2w, e2x.y / e2w, e2x.z / e2w};
      if (e1.z < 0 || e2.z < 0) erase = 1;
      else if (norm(e1 - e2) > 10) erase = 1;
      else {
        const Plk line_w = plk_to_pose(Lc, Rwc, twc);
        double allerr = 0;
        const int no = B.ln_nobs[li];
        for (int k = 0; k < no; ++k) {
          const int j = s + k;
          const double* xj = pose + 7 * j;
          const M3 Rj = mul(rot_diff, qmat(qnormalized(qpose(xj))));
          const V3 Pj{newpose[7 * j], newpose[7 * j + 1], newpose[7 * j + 2]};
          const V3 t1 = Pj + mul(Rj, tic);
          const M3 R1 = mul(Rj, ric);
          const Plk lc = plk_from_pose(line_w, R1, t1);   // feature_manager.cpp:390-411
          const double sql = sqrt(lc.n.x * lc.n.x + lc.n.y * lc.n.y);
          const V3 nn{lc.n.x / sql, lc.n.y / sql, lc.n.z / sql};
          const double* ob = ob0 + 8 * k;
          const double err = (fabs(nn.x * ob[0] + nn.y * ob[1] + nn.z) + fabs(nn.x * ob[2] + nn.y * ob[3] + nn.z)) / 2.0;
          if (allerr < err) allerr = err;
        }
        if (allerr > 3.0 / 500.0) erase = 1;
      }
    }
    B.ln_removed[li] = erase;
  }
  __syncthreads();
  for (int i = tid; i < 77; i += blockDim.x) pose[i] = newpose[i];
  for (int i = tid; i < 7; i += blockDim.x) ex[i] = newex[i];
  // erased lines add no factor to the marginalisation: the kept-block table may shrink
  if (tid == 0 && B.opt.remove_line_outliers && B.opt.marginalization_flag == 0) {
    KeepSrc S;
    S.nP = B.nP[w]; S.pt_start = B.pt_start + (size_t)w * B.maxP; S.pt_nobs = B.pt_nobs + (size_t)w * B.maxP;
    S.nL = nL; S.ln_start = B.ln_start + (size_t)w * B.maxL; S.ln_nobs = B.ln_nobs + (size_t)w * B.maxL;
    S.ln_removed = B.ln_removed + (size_t)w * B.maxL;
    S.pr_nb = B.pr_n[w] > 0 ? B.pr_nb[w] : 0; S.pr_kind = B.pr_kind + (size_t)w * MAXPB; S.pr_frame = B.pr_frame + (size_t)w * MAXPB;
    S.imu01 = B.pre[(size_t)w * NF + 1].sum_dt < 10.0;
    keep_tables_old(S, B.mg_kind + (size_t)w * MAXPB, B.mg_frame + (size_t)w * MAXPB, B.mg_idx + (size_t)w * MAXPB,
                    B.mg_cam + (size_t)w * MAXPB, B.mg_n + w, B.mg_nb + w, B.mg_m + w);
  }
}
__global__ __launch_bounds__(128) void k_gauge(DevBatch B) { gauge_body(B, blockIdx.x); }

// ---------------------------------------------------------------------------------------------------
// Spectral factor of a symmetric positive semi-definite matrix held in LDS (full storage, row stride ld):
//     A = B B^T,  B = P^T G,  columns of G mutually orthogonal,  |g_k|^2 = lambda_k (eigenvalues of A).
// Step 1: Cholesky with diagonal pivoting, P A P^T = L L^T (stops at the first pivot <= n eps max_i a_ii, as dpstrf).
// Step 2: one-sided (Hestenes) Jacobi on the columns of L until they are orthogonal: G = L J.
// The Cholesky factor of a graded PSD matrix is what makes Jacobi converge in a few sweeps and to high
// RELATIVE accuracy of the small eigenvalues (Demmel & Veselic), which the marginalisation needs because it
// consumes 1/lambda (pseudo-inverse) and 1/sqrt(lambda).  Round-robin ordering, 8 lanes per column pair,
// one barrier per step.  On exit: A holds G (n x rank), lam[k] = |g_k|^2, perm[t] = original index of row t.
// Returns the rank.  Needs blockDim.x >= 8 * ((n + 1) / 2).  rel_tol: pivots <= max(n eps, rel_tol) * max_i a_ii stop the
// factorisation (the trailing block is then treated as zero).
// Cholesky with diagonal pivoting of the PSD matrix A (LDS, full storage): P A P^T = L L^T, L = n x rank lower trapezoidal
// left in A (upper part zeroed), perm[t] = original index of row t.  An optional right-hand side c (length n, permuted in
// step) rides along: on exit c[k] = y_k for k < rank, where L(0:rank,0:rank) y = (P c)(0:rank).
// Stops at the first pivot <= max(abs_tol, max(n eps, rel_tol) * max_i a_ii); the trailing block is then treated as zero.
__device__ __forceinline__ int psd_pivoted_cholesky(double* A, int n, int ld, int* perm, double* red, int* iflag, double rel_tol,
                                    double abs_tol, double* c, double* col /* n doubles of scratch */) {
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  for (int i = tid; i < n; i += T) perm[i] = i;
  __syncthreads();
  int rank = n;
  double tol = 0.0;
  for (int k = 0; k < n; ++k) {
    // pivot: largest remaining diagonal (first index on ties), found by wave 0
    if (tid < 64) {
      double best = -1.0;
      int bi = k;
      for (int i = k + lane; i < n; i += 64) {
        const double d = A[i * ld + i];
        if (d > best) { best = d; bi = i; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { iflag[0] = bi; red[0] = best; }
    }
    __syncthreads();
    const int p = iflag[0];
    const double piv = red[0];
    // dpstrf-style stopping rule with a caller-supplied relative tolerance: a pivot at the noise level of the matrix
    // must not be accepted -- dividing a column of noise by the root of a noise pivot fabricates an O(1) direction
    if (k == 0) tol = fmax(abs_tol, fmax((double)n * 2.220446049250313e-16, rel_tol) * piv);
    if (!(piv > tol)) { rank = k; break; }
    if (p != k) {   // symmetric swap k <-> p (rows, then columns), full storage
      for (int c = tid; c < n; c += T) { const double t = A[k * ld + c]; A[k * ld + c] = A[p * ld + c]; A[p * ld + c] = t; }
      __syncthreads();
      for (int r = tid; r < n; r += T) { const double t = A[r * ld + k]; A[r * ld + k] = A[r * ld + p]; A[r * ld + p] = t; }
      if (tid == 0) {
        const int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        if (c) { const double tc = c[k]; c[k] = c[p]; c[p] = tc; }
      }
      __syncthreads();
    }
    const double lkk = sqrt(piv);
    const double inv = 1.0 / lkk;
    // The scaled column goes to a side vector first (one barrier); the trailing update reads only that vector, so the
    // column of L, the zeros of row k and the rhs entry can be written in the same phase: two barriers per pivot step
    // instead of three, and no division inside the update.
    for (int i = k + tid; i < n; i += T) col[i] = (i == k) ? lkk : A[i * ld + k] * inv;
    __syncthreads();
    const int m = n - k - 1;
    for (int it = tid; it < m * m; it += T) {
      const int ii = it / m, i = k + 1 + ii, j = k + 1 + (it - ii * m);
      A[i * ld + j] -= col[i] * col[j];
    }
    for (int i = k + tid; i < n; i += T) {
      A[i * ld + k] = col[i];
      if (i > k) A[k * ld + i] = 0.0;           // upper part of row k is not part of L
    }
    if (c) {   // forward substitution rides along: y_k = c_k / l_kk, c_i -= l_ik y_k
      const double yk = c[k] * inv;
      for (int i = k + 1 + tid; i < n; i += T) c[i] -= col[i] * yk;
      __syncthreads();                          // every thread has read c[k]
      if (tid == 0) c[k] = yk;
    } else {
      __syncthreads();
    }
  }
  __syncthreads();
  // columns rank..n-1 do not exist
  for (int it = tid; it < n * (n - rank); it += T) A[(it / (n - rank)) * ld + rank + it % (n - rank)] = 0.0;
  __syncthreads();
  return rank;
}

// The same factorisation (same arithmetic per entry, same pivots, same bits) by ONE wave for n <= NMAX <= 48, the matrix
// in registers: lane j owns column j (a[i] = A(i, j), static indices only), its diagonal entry and its rhs entry.  A pivot
// step is: arg-max of the live diagonals by shuffles; the pivot lane writes its scaled column to an LDS vector (col); every
// lane reads its own entry l_j and, broadcast, every l_i, and updates its column -- no work-group barrier and no swaps
// (the permutation is applied once at the end).  The work-group version above pays five barriers per pivot step
// (~3.9 k cycles; 175 k for the 45 x 45 kept block).  All threads of the block must call it; Lo is n x ld scratch.
template <int NMAX>
__device__ __forceinline__ int psd_pivoted_cholesky_wave(double* A, int n, int ld, int* perm, int* iflag, double rel_tol, double abs_tol,
                                         double* c, double* Lo /* n x ld + NMAX doubles */) {
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  double* col = Lo + n * ld;             // NMAX: the pivot column, zero beyond n -- the loops over it have no bounds checks
  if (tid < 64) {
    double a[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) a[i] = (i < n && lane < n) ? A[i * ld + lane] : 0.0;
    double diag = lane < n ? A[lane * ld + lane] : -1.0;
    double cj = (c && lane < n) ? c[lane] : 0.0;
    bool alive = lane < n;
    unsigned long long mask = __ballot(alive);
    int rank = n;
    double tol = 0.0;
    for (int k = 0; k < n; ++k) {
      // pivot: the largest live diagonal, first index on ties -- maximum by DPP row steps + four readlanes (VALU only; a
      // shuffle butterfly over (value, index) is six dependent LDS-path round trips), index from the ballot of the lanes
      // that hold it.  Dead lanes and the DPP fill value count as 0: no live diagonal above 0 ends the factorisation.
      const double best = alive ? diag : 0.0;
      double mx = best;
      mx = fmax(mx, dpp_shr_f64<0x111>(mx));
      mx = fmax(mx, dpp_shr_f64<0x112>(mx));
      mx = fmax(mx, dpp_shr_f64<0x114>(mx));
      mx = fmax(mx, dpp_shr_f64<0x118>(mx));
      const double piv = fmax(fmax(readlane_f64(mx, 15), readlane_f64(mx, 31)), fmax(readlane_f64(mx, 47), readlane_f64(mx, 63)));
      const unsigned long long hit = __ballot(alive && diag == piv);
      const int p = hit ? __builtin_ctzll(hit) : 0;
      if (k == 0) tol = fmax(abs_tol, fmax((double)n * 2.220446049250313e-16, rel_tol) * piv);
      if (!(piv > tol)) { rank = k; break; }
      const double lkk = sqrt(piv);
      const double inv = 1.0 / lkk;
      if (lane == p) {
        // (rows that are already eliminated get a meaningless entry: it only ever touches their own dead rows and the
        //  part of Lo above the diagonal, which the final pass never reads)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) col[i] = a[i] * inv;
        col[p] = lkk;
        perm[k] = p;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double lj = lane < n ? col[lane] : 0.0;
      const bool upd = alive && lane != p;
      const double ljm = upd ? lj : 0.0;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) a[i] -= col[i] * ljm;     // (uniform addresses: broadcast reads, all in flight)
      if (upd) diag -= lj * lj;
      if (lane < n) Lo[lane * ld + k] = lj;
      if (c) {
        const double yk = readlane_f64(cj, p) * inv;
        if (upd) cj -= lj * yk;
        if (lane == p) cj = yk;          // lane p keeps y_k: stored at position k below
      }
      if (lane == p) alive = false;
      mask &= ~(1ull << p);
      __builtin_amdgcn_wave_barrier();   // col is rewritten by the next pivot lane
    }
    // positions rank..n-1: the indices that were never pivots, ascending
    {
      const int before = __popcll(mask & ((1ull << lane) - 1ull));
      if (alive) perm[rank + before] = lane;
    }
    if (c && lane < n) col[lane] = cj;   // y_k sits in lane perm[k]
    if (lane == 0) iflag[0] = rank;
  }
  __syncthreads();
  const int rank = iflag[0];
  // L(t, k) = Lo(perm[t], k), t >= k, k < rank; everything else zero
  for (int it = tid; it < n * n; it += T) {
    const int t = it / n, k = it - t * n;
    A[t * ld + k] = (k < rank && t >= k) ? Lo[perm[t] * ld + k] : 0.0;
  }
  if (c)
    for (int k = tid; k < n; k += T) c[k] = k < rank ? col[perm[k]] : 0.0;
  __syncthreads();
  return rank;
}

// The same factorisation for 48 < n <= NMAX = NQ x NH by 2 NQ waves with column SLICES in registers (round 4: the reference's
// steady-state kept block has 75 dims; a whole 76-entry column per lane does not fit the register file next to what the
// compiler needs around it -- DESIGN.md section 8 (2)).  Wave w: column group w & 1 (columns 64 (w & 1) + lane), row slice
// w >> 1 (rows NH (w >> 1) .. + NH - 1); k_marg<512> uses NQ = 4 slices of 19 rows on its eight waves.  Every lane keeps the
// diagonal entry of its column (all slices update it with the same arithmetic), the lanes of slice 0 the rhs entry.  Pivot step k:
//   A  the two waves of slice 0 find the largest live diagonal of their columns (first lane on ties) and write (value, index,
//      rhs entry) into slot k & 1 of `meta`; EVERY wave then arrives at counter A and waits until all have: everybody has
//      finished step k - 1 (its reads of the previous column and candidates), and the candidates of step k are visible.  The
//      winner is the larger value, column group 0 on ties = the lower index (the one-wave rule);
//   B  the NQ lanes that own the winning column write its scaled slices into `col`, their waves arrive at counter B; every
//      wave waits for them;
//   C  every lane reads its own entry l_j and, broadcast, the NH entries of its row slice, and updates.
// The counters only grow (relaxed LDS atomics + work-group fences).  A wait gives up after 2^24 polls and the call returns
// -1 instead of hanging the device.  Same pivots and same arithmetic per entry as the other versions, except 1 / sqrt(pivot)
// (hardware estimate + two Newton steps).  Lo: n x ld + NMAX + 24 doubles of scratch.
template <int NH, int NQ>
__device__ __forceinline__ int psd_pivoted_cholesky_wave4(double* A, int n, int ld, int* perm, int* iflag, double rel_tol, double abs_tol,
                                                          double* c, double* Lo, int* phase /* 4 ints of LDS */,
                                                          long long* stamps = nullptr) {
  constexpr int NMAX = NQ * NH;     // NQ row slices of NH rows: 2 NQ waves
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6;
  double* col = Lo + n * ld;             // NMAX: the winning column, scaled
  double* meta = col + NMAX;             // [2 slots][2 groups][4]: value, global index, rhs entry; [16], [17]: tail counts
  if (tid < 4) phase[tid] = 0;
  if (tid == 4) iflag[1] = 0;
  __syncthreads();
  double cj = 0.0;
  int gcol = 0, hf = 1;
  if (wv < 2 * NQ) {
    const int cg = wv & 1, r0 = NH * (wv >> 1);
    hf = wv >> 1;      // row slice; slice 0 holds the candidates, the rhs and writes the factor
    gcol = 64 * cg + lane;
    double a[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) a[i] = (r0 + i < n && gcol < n) ? A[(r0 + i) * ld + gcol] : 0.0;
    double diag = gcol < n ? A[gcol * ld + gcol] : -1.0;
    cj = (c && hf == 0 && gcol < n) ? c[gcol] : 0.0;
    bool alive = gcol < n;
    int rank = n;
    double tol = 0.0;
    bool dead = false;                   // a wait timed out
    // phase[0]: arrivals at A (every wave, once per step); phase[1]: arrivals at B (the two waves that own the winning column);
    // phase[2]: the tail of the top-half waves.  Counters only grow; one LDS word is polled per wait.
#ifdef VPL_STAMPS
    long long st_wait[2] = {0, 0}, st_b = 0, st_c = 0;
#endif
    auto wait_cnt = [&](int which, int v) {
#ifdef VPL_STAMPS
      const long long t0 = __builtin_readcyclecounter();
#endif
      int polls = 0;
      while (!dead && __hip_atomic_load(&phase[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v)
        if (++polls > (1 << 24)) dead = true;
#ifdef VPL_STAMPS
      if (which < 2) st_wait[which] += __builtin_readcyclecounter() - t0;
#endif
    };
    auto arrive = [&](int which) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_fetch_add(&phase[which], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int k = 0; k < n; ++k) {
      // A: candidates of the two column groups (top-half waves), slot k & 1
      if (hf == 0) {
        const double best = alive ? diag : 0.0;
        double mx = best;
        mx = fmax(mx, dpp_shr_f64<0x111>(mx));
        mx = fmax(mx, dpp_shr_f64<0x112>(mx));
        mx = fmax(mx, dpp_shr_f64<0x114>(mx));
        mx = fmax(mx, dpp_shr_f64<0x118>(mx));
        const double pw = fmax(fmax(readlane_f64(mx, 15), readlane_f64(mx, 31)), fmax(readlane_f64(mx, 47), readlane_f64(mx, 63)));
        const unsigned long long hit = __ballot(alive && diag == pw);
        const int pl = hit ? __builtin_ctzll(hit) : 0;
        if (lane == pl) {
          double* mm = meta + 8 * (k & 1) + 4 * cg;
          mm[0] = (hit && pw > 0.0) ? pw : 0.0;     // (a group without a live positive diagonal offers 0)
          mm[1] = (double)gcol;
          mm[2] = cj;
        }
      }
      arrive(0);
      wait_cnt(0, 2 * NQ * (k + 1));
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (dead) break;
      const double* mk = meta + 8 * (k & 1);
      const double p0 = mk[0], p1 = mk[4];
      const int win = p1 > p0 ? 1 : 0;
      const double piv = win ? p1 : p0;
      const int p = (int)(win ? mk[5] : mk[1]);
      const double cjp = win ? mk[6] : mk[2];
      if (k == 0) tol = fmax(abs_tol, fmax((double)n * 2.220446049250313e-16, rel_tol) * piv);
      if (!(piv > tol)) { rank = k; break; }
      // 1 / sqrt(pivot) by v_rsq_f64 + two Newton steps, sqrt(pivot) = pivot / sqrt(pivot): ~10 dependent instructions on the
      // critical path of every wave where an IEEE sqrt and divide are ~45 (as k_chol's chains do; the results agree with the
      // correctly rounded ones to an ulp -- the other versions of this factorisation keep sqrt() and the division)
      double inv = __builtin_amdgcn_rsq(piv);
      inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
      inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
      const double lkk = piv * inv;
      // B: the two lanes of column p write its halves (everybody is past step k - 1: the wait above)
      if (gcol == p) {
#pragma unroll
        for (int i = 0; i < NH; ++i) col[r0 + i] = a[i] * inv;
        if (p >= r0 && p < r0 + NH) col[p] = lkk;
      }
      if (cg == win) arrive(1);          // (the wave holds column p: wave-uniform)
      wait_cnt(1, NQ * (k + 1));
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (dead) break;
      // C: update
      const double lj = gcol < n ? col[gcol] : 0.0;
      const bool upd = alive && gcol != p;
      const double ljm = upd ? lj : 0.0;
#pragma unroll
      for (int i = 0; i < NH; ++i) a[i] -= col[r0 + i] * ljm;     // (uniform addresses: broadcast reads)
      if (upd) diag -= lj * lj;
      if (hf == 0) {
        if (gcol < n) Lo[gcol * ld + k] = lj;
        if (c) {
          const double yk = cjp * inv;
          if (upd) cj -= lj * yk;
          if (gcol == p) cj = yk;          // column p keeps y_k: stored at position k below
        }
        if (gcol == p) perm[k] = p;
      }
      if (gcol == p) alive = false;
    }
    // positions rank..n-1: the indices that were never pivots, ascending -- column group 0 first (top-half waves)
    if (hf == 0) {
      const unsigned long long mask = __ballot(alive);
      if (lane == 0) meta[16 + cg] = (double)__popcll(mask);
      arrive(2);
      wait_cnt(2, 2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (!dead) {
        const int before = __popcll(mask & ((1ull << lane) - 1ull)) + (cg ? (int)meta[16] : 0);
        if (alive) perm[rank + before] = gcol;
      }
    }
    if (dead && lane == 0) iflag[1] = 1;
    if (tid == 0) iflag[0] = rank;
#ifdef VPL_STAMPS
    if (stamps && lane == 0 && wv < 4) { stamps[2 * wv] = st_wait[0]; stamps[2 * wv + 1] = st_wait[1]; }
#endif
  }
  __syncthreads();                       // every wave is past its last read of col
  if (c && hf == 0 && gcol < n) col[gcol] = cj;   // y_k sits in column perm[k]
  __syncthreads();
  if (iflag[1]) return -1;
  const int rank = iflag[0];
  // L(t, k) = Lo(perm[t], k), t >= k, k < rank; everything else zero
  for (int it = tid; it < n * n; it += T) {
    const int t = it / n, k = it - t * n;
    A[t * ld + k] = (k < rank && t >= k) ? Lo[perm[t] * ld + k] : 0.0;
  }
  if (c)
    for (int k = tid; k < n; k += T) c[k] = k < rank ? col[perm[k]] : 0.0;
  __syncthreads();
  return rank;
}

__device__ __forceinline__ int psd_spectral_factor(double* A, int n, int ld, int* perm, double* lam, double* red, int* iflag,
                                   double rel_tol, double* Lo /* n x ld scratch */) {
  const int tid = threadIdx.x, T = blockDim.x;
  const int rank = n <= 16 ? psd_pivoted_cholesky_wave<16>(A, n, ld, perm, iflag, rel_tol, 0.0, nullptr, Lo)
                           : psd_pivoted_cholesky(A, n, ld, perm, red, iflag, rel_tol, 0.0, nullptr, lam);
  // ---- one-sided Jacobi on the rank columns ----
  const int m = (rank + 1) & ~1, half = m / 2;
  const int P = tid >> 3, sub = tid & 7;
  for (int sweep = 0; sweep < 40 && rank > 1; ++sweep) {
    if (tid == 0) iflag[1] = 0;
    __syncthreads();
    for (int step = 0; step < m - 1; ++step) {
      if (P < half) {
        const int a = P == 0 ? 0 : 1 + (P - 1 + step) % (m - 1);
        const int b = P == 0 ? 1 + (m - 2 + step) % (m - 1) : 1 + (m - 2 - P + step) % (m - 1);
        const int ci = min(a, b), cj = max(a, b);
        double gi[10], gj[10];
        double al = 0, be = 0, ga = 0;
        if (cj < rank) {
#pragma unroll
          for (int q = 0; q < 10; ++q) {
            const int r = sub + 8 * q;
            gi[q] = r < n ? A[r * ld + ci] : 0.0;
            gj[q] = r < n ? A[r * ld + cj] : 0.0;
            al += gi[q] * gi[q]; be += gj[q] * gj[q]; ga += gi[q] * gj[q];
          }
        }
        al += __shfl_xor(al, 1, 64); al += __shfl_xor(al, 2, 64); al += __shfl_xor(al, 4, 64);
        be += __shfl_xor(be, 1, 64); be += __shfl_xor(be, 2, 64); be += __shfl_xor(be, 4, 64);
        ga += __shfl_xor(ga, 1, 64); ga += __shfl_xor(ga, 2, 64); ga += __shfl_xor(ga, 4, 64);
        if (cj < rank && fabs(ga) > 1e-15 * sqrt(al * be) && ga != 0.0) {
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
          for (int q = 0; q < 10; ++q) {
            const int r = sub + 8 * q;
            if (r < n) {
              A[r * ld + ci] = c * gi[q] - sn * gj[q];
              A[r * ld + cj] = sn * gi[q] + c * gj[q];
            }
          }
          if (sub == 0) iflag[1] = 1;
        }
      }
      __syncthreads();
    }
    if (!iflag[1]) break;
    __syncthreads();
  }
  __syncthreads();
  for (int k = tid; k < n; k += T) {
    double s2 = 0;
    if (k < rank)
      for (int r = 0; r < n; ++r) s2 += A[r * ld + k] * A[r * ld + k];
    lam[k] = s2;
  }
  __syncthreads();
  return rank;
}

constexpr int MARG_THREADS = 512;
constexpr double kMargEps = 1e-8;   // marginalization_factor.h:67
#ifndef VPL_MARG_NOISE_REL
#define VPL_MARG_NOISE_REL 0.0
#endif
constexpr double kMargNoiseRel = VPL_MARG_NOISE_REL;   // pivots of the kept block below this fraction of its largest diagonal are noise
constexpr int MTROWS_MAX = 96;      // landmark rows staged per elimination pass: 96 where the prior leaves the LDS for it (every pass
                                    // costs two global round trips and three barriers), 64 or 32 for large priors
static_assert(MAXKEEP <= 80, "psd_spectral_factor keeps 10 rows per lane (8 lanes per column pair)");

// LDS layout of k_marg (doubles), shared by host (size) and device (offsets)
struct MargLayout {
  int ldm, EB, nd, ldd, total, mtrows, loff;
};
// small = the 256-thread form of k_marg: at most 79 KB, so that two work-groups share a CU (most of the kernel is a lone
// wave factoring a 15 x 15 and the kept block while the other waves wait: a second window on the CU uses the idle SIMDs);
// its landmark lists hold 256 entries per kind
constexpr size_t MARG_LDS_BIG = 150 * 1024, MARG_LDS_SMALL = 79 * 1024;
__host__ __device__ inline MargLayout marg_layout(int n, bool small = false) {
  MargLayout L;
  L.loff = small ? 256 : 1024;
  const int nn = n < 2 ? 2 : n;
  L.ldm = nn | 1;                       // odd row stride
  L.EB = nn * L.ldm;
  L.nd = 15 + n;
  L.ldd = L.nd | 1;
  // G (kept block) | Ad (nd x ldd) | tile | E15 (15 x 17) | bv(nd) | tmp(nd*16) | lam(nn) | red(24) | ints
  const int fixed = L.EB + L.nd * L.ldd + 16 * 17 + L.nd + (L.nd * 16 < 640 ? 640 : L.nd * 16) + nn + 16 + 24 + (nn + L.nd + 2 * L.loff + 16) / 2 + 4;
  L.mtrows = MTROWS_MAX;
  while (L.mtrows > 32 && (size_t)(fixed + L.mtrows * 74) * sizeof(double) > (small ? MARG_LDS_SMALL : MARG_LDS_BIG)) L.mtrows -= 32;
  L.total = fixed + L.mtrows * 74;
  return L;
}

template <int T>
__device__ __forceinline__ void marg_body(const DevBatch& B, const int w, double* sm) {
  const int tid = threadIdx.x;
  const int nP = B.nP[w], nL = B.nL[w];
  const int nb = B.mg_nb[w];
  const int n = B.mg_n[w];
  if (n == 0) return;   // MARGIN_SECOND_NEW without pose[WINDOW_SIZE-1] in the prior: nothing to do (estimator.cpp:1385)
  const bool second_new = B.opt.marginalization_flag == 1;
  const int md = second_new ? 6 : 15;    // dims marginalised through the pseudo-inverse
  const MargLayout L = marg_layout(n, T == 256);
  const int nd = md + n, ldd = L.ldd, ldm = L.ldm;
  double* G = sm;                        // n x ldm : kept block -> spectral factor
  double* Ad = G + L.EB;                 // nd x ldd dense pre-marginalisation matrix
  const int MTROWS = L.mtrows;
  double* tile = Ad + nd * ldd;          // MTROWS x 74
  double* E15 = tile + MTROWS * 74;      // 15 x 17: Amm -> its spectral factor
  double* bv = E15 + 16 * 17;            // nd
  double* tmp = bv + nd;                 // max(nd * 16, 640)
  double* lam = tmp + (nd * 16 < 640 ? 640 : nd * 16);   // max(n, 15) + 1
  double* red = lam + (n < 2 ? 2 : n) + 16;   // 24
  int* perm = (int*)(red + 24);          // max(n, 15)
  int* dmap = perm + (n < 16 ? 16 : n) + (n & 1);   // nd
  int* lst = dmap + nd + (nd & 1);       // 2 x LOFF
  const int LOFF = L.loff;
  __shared__ int s_np0, s_nl0, s_flag[4], s_phase[4];

  // dense order: [sb_0 (9), pose_0 (6) | kept blocks in canonical order]  (the reference moves the
  // pose-like marginalised blocks behind the landmarks in descending index order, :291-309)
  // MARGIN_SECOND_NEW: [pose_9 (6) | kept blocks], no landmarks (estimator.cpp:1387-1405)
  if (tid < md) dmap[tid] = second_new ? 15 * (NF - 2) + tid : (tid < 9 ? 6 + tid : tid - 9);
  if (tid < nb) {
    const int kind = B.mg_kind[(size_t)w * MAXPB + tid];
    const int base = B.mg_cam[(size_t)w * MAXPB + tid], idx = B.mg_idx[(size_t)w * MAXPB + tid];
    const int ls = kind == 1 ? 9 : 6;
    for (int k = 0; k < ls; ++k) dmap[md + idx + k] = base + k;
  }
  // landmarks that start in frame 0: the points are the first bucket of the start-frame sort (k_lin leaves H_pp = 0 on a
  // track without factors: its row is staged as zeros), the lines are compacted in order by wave 0
  if (tid >= 64 && !second_new) {
    const int np0 = min(LOFF, B.ps_cnt[(size_t)w * (NF + 1) + 1]);
    for (int a = tid - 64; a < np0; a += T - 64) lst[a] = B.ps_list[(size_t)w * B.maxP + a];
    if (tid == 64) s_np0 = np0;
  }
  if (tid < 64) {
    int c = 0;
    if (!second_new)
      for (int l0 = 0; l0 < nL; l0 += 64) {
        const int l = l0 + tid;
        const bool on = l < nL && B.ln_start[(size_t)w * B.maxL + l] == 0 && B.ln_nobs[(size_t)w * B.maxL + l] >= 2 &&
                        !B.ln_removed[(size_t)w * B.maxL + l];
        const unsigned long long m = __ballot(on);
        if (on && c + __popcll(m & ((1ull << tid) - 1ull)) < LOFF) lst[LOFF + c + __popcll(m & ((1ull << tid) - 1ull))] = l;
        c += __popcll(m);
      }
    if (tid == 0) { s_nl0 = min(c, LOFF); if (second_new) s_np0 = 0; }
  }
  __syncthreads();
  VPL_STAMP(B, w, 32);
  const double* Hcc = B.Hcc + (size_t)w * NCP;
  const double* gc = B.gc + (size_t)w * NC;
  // ---- landmark elimination (plain inverse of the block-diagonal landmark part, :316-326) --------
  // rows X = C^-1 [W | g] per start-frame-0 landmark (C C^T = H_ll); A = Hcc - X^T X, b = gc - X^T z
  for (int base = 0; base < nd * nd; base += 8 * T) {   // eight global loads in flight per lane, then the LDS stores
    double hv[8];
    int dst[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = base + u * T + tid;
      dst[u] = -1;
      hv[u] = 0.0;
      if (it < nd * nd) {
        const int i = it / nd, j = it - i * nd;
        const int ci = dmap[i], cj = dmap[j];
        dst[u] = i * ldd + j;
        hv[u] = ci >= cj ? Hcc[tri(ci, cj)] : Hcc[tri(cj, ci)];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (dst[u] >= 0) Ad[dst[u]] = hv[u];
  }
  VPL_STAMP(B, w, 38);
  for (int i = tid; i < nd; i += T) bv[i] = gc[dmap[i]];
  // list of the dense dims with a visual index (tmp is free until the Schur complement of the marginalised block)
  int* vlist = (int*)tmp;
  int* vinv = vlist + 96;   // 72: visual column -> dense index or -1 (nd <= 95 entries of vlist in front of it)
  if (tid < 64) {   // (wave 0: ballots and prefix counts; one thread walking the nd entries cost 17 k cycles)
    const int lane = tid;
    const bool f0 = lane < nd && cam2vis(dmap[lane]) >= 0;
    const bool f1 = lane + 64 < nd && cam2vis(dmap[lane + 64]) >= 0;
    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (f0) vlist[__popcll(m0 & below)] = lane;
    if (f1) vlist[__popcll(m0) + __popcll(m1 & below)] = lane + 64;
    if (lane == 0) s_flag[2] = __popcll(m0) + __popcll(m1);
  } else if (tid < 64 + 72) {
    vinv[tid - 64] = -1;
  }
  __syncthreads();
  for (int i = tid; i < nd; i += T) {
    const int vi = cam2vis(dmap[i]);
    if (vi >= 0) vinv[vi] = i;
  }
  __syncthreads();
  const int nv = s_flag[2];
  const int np0 = s_np0, nl0 = s_nl0;
  VPL_STAMP(B, w, 39);
  for (int base = 0; base < np0 + nl0; ) {
    int nrows;
    // Per-row constants first (one lane per landmark: its scale or 4x4 factor, its start frame), then the elements with
    // one global load each, four in flight per lane.  Loading them per element put two dependent global latencies into
    // every trip of the element loop and repeated the 4x4 Cholesky of a line 73 times.
    double* rs = tmp + 128;              // MTROWS point scales
    double* lineC = tmp + 128 + MTROWS_MAX;                  // (MTROWS / 4) x 10 line factors
    int* rstart = (int*)(tmp + 128 + MTROWS_MAX + 10 * (MTROWS_MAX / 4));   // MTROWS start frames
    int* rland = rstart + MTROWS;        // MTROWS landmark indices
    if (base < np0) {
      const int cnt = min(MTROWS, np0 - base);
      nrows = cnt;
      if (tid < cnt) {
        const int p = lst[base + tid];
        const size_t pi = (size_t)w * B.maxP + p;
        const double hpp = B.Hpp[pi];
        rs[tid] = hpp != 0.0 ? 1.0 / sqrt(hpp) : 0.0;
        rstart[tid] = B.pt_start[pi];
        rland[tid] = p;
      }
      __syncthreads();
      for (int it0 = tid; it0 < cnt * 73; it0 += 4 * T) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int it = it0 + u * T;
          v[u] = 0.0;
          if (it < cnt * 73) {
            const int rr = it / 73, c = it - rr * 73;
            const size_t pi = (size_t)w * B.maxP + rland[rr];
            const int cc = c < NV ? wcol(c, rstart[rr], B.WS) : 0;
            if (cc >= 0) v[u] = c < NV ? B.Wp[pi * B.WS + cc] : B.gp[pi];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int it = it0 + u * T;
          if (it < cnt * 73) {
            const int rr = it / 73, c = it - rr * 73;
            tile[rr * 74 + c] = rs[rr] * v[u];
          }
        }
      }
      base += cnt;
    } else {
      const int l0 = base - np0;
      const int cnt = min(MTROWS / 4, nl0 - l0);
      nrows = 4 * cnt;
      if (tid < cnt) {
        const int l = lst[LOFF + l0 + tid];
        const size_t li = (size_t)w * B.maxL + l;
        const double* Hl = B.Hll + li * 16;
        double Cc[10];
        int t = 0;
        for (int a = 0; a < 4; ++a)
          for (int cc = 0; cc <= a; ++cc, ++t) Cc[t] = Hl[4 * a + cc];
        for (int jj = 0; jj < 4; ++jj) {
          double d = Cc[tri(jj, jj)];
          for (int k = 0; k < jj; ++k) d -= Cc[tri(jj, k)] * Cc[tri(jj, k)];
          d = sqrt(d);
          Cc[tri(jj, jj)] = d;
          for (int ii = jj + 1; ii < 4; ++ii) {
            double s2 = Cc[tri(ii, jj)];
            for (int k = 0; k < jj; ++k) s2 -= Cc[tri(ii, k)] * Cc[tri(jj, k)];
            Cc[tri(ii, jj)] = s2 / d;
          }
        }
        for (int q = 0; q < 10; ++q) lineC[10 * tid + q] = Cc[q];
        rstart[tid] = B.ln_start[li];
        rland[tid] = l;
      }
      __syncthreads();
      for (int it = tid; it < cnt * 73; it += T) {
        const int ll = it / 73, c = it - ll * 73;
        const size_t li = (size_t)w * B.maxL + rland[ll];
        const double* Cc = lineC + 10 * ll;
        const int cc = c < NV ? wcol(c, rstart[ll], B.WS) : 0;
        double wv4[4], x[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wv4[a] = cc < 0 ? 0.0 : (c < NV ? B.Wl[(li * 4 + a) * B.WS + cc] : B.gl[li * 4 + a]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          double s2 = wv4[a];
#pragma unroll
          for (int k = 0; k < 4; ++k) if (k < a) s2 -= Cc[tri(a, k)] * x[k];
          x[a] = s2 / Cc[tri(a, a)];
          tile[(4 * ll + a) * 74 + c] = x[a];
        }
      }
      base += cnt;
    }
    __syncthreads();
    VPL_STAMP(B, w, base <= np0 ? 29 : 30);
    // X^T X of the staged rows on the FP64 matrix cores: 73 columns (72 visual dims + the rhs) = 5 x 5 tiles of 16, the 15
    // lower ones over the 8 waves; C[a][b] = sum_r X[r][a] X[r][b] with the A lane (kk, m) supplying X[4 ks + kk][16 ta + m].
    // Only the dense dims with a visual index couple to the landmarks: vinv maps a visual column to its dense index.
    {
      const int lane = tid & 63, wv = tid >> 6, m = lane & 15, kk = lane >> 4;
      const int ksteps = (nrows + 3) >> 2;
#pragma unroll 1
      for (int tix2 = wv; tix2 < 15; tix2 += T / 64) {
        int ta, tb;
        tri_decode(tix2, ta, tb);
        const int ca = 16 * ta + m, cb = 16 * tb + m;
        typedef double v4dm __attribute__((ext_vector_type(4)));
        v4dm acc = {0, 0, 0, 0};
        for (int ks = 0; ks < ksteps; ++ks) {
          const int r = 4 * ks + kk;
          const double av = (r < nrows && ca < 73) ? tile[r * 74 + ca] : 0.0;
          const double bw = (r < nrows && cb < 73) ? tile[r * 74 + cb] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bw, acc, 0, 0, 0);
        }
        const double vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int a2 = 16 * ta + kk + 4 * v, b2 = cb;
          if (b2 >= 72 || a2 > 72) continue;
          const int j = vinv[b2];
          if (j < 0) continue;
          if (a2 == 72) { bv[j] -= vals[v]; continue; }
          const int i = vinv[a2];
          if (i < 0) continue;
          Ad[i * ldd + j] -= vals[v];
          if (ta != tb) Ad[j * ldd + i] -= vals[v];   // (inside a diagonal tile the mirror entry has a lane of its own)
        }
      }
    }
    __syncthreads();
  }

  VPL_STAMP(B, w, 33);
  // ---- marginalise the md pose-like dims through the eigen pseudo-inverse (:329-346) ----------
  //      Amm^+ = sum_{lambda_k > eps} b_k b_k^T / lambda_k^2   with Amm = B B^T, b_k orthogonal, |b_k|^2 = lambda_k
  for (int it = tid; it < md * md; it += T) {
    const int i = it / md, j = it % md;
    E15[i * 17 + j] = 0.5 * (Ad[i * ldd + j] + Ad[j * ldd + i]);
  }
  __syncthreads();
  // Positive definite Amm (every eigenvalue far above the reference's 1e-8): the pseudo-inverse is the inverse and two
  // triangular solves per row of Arm replace the eigen-decomposition.  A rank-deficient Amm (a pivot at the rounding
  // level) takes the spectral route, which reproduces the eigenvalue test of :334-346.
  double* Y = tile;   // n x 16 scratch (tile is free now)
  for (int it = tid; it < md * md; it += T) Y[it] = E15[(it / md) * 17 + it % md];   // keep a copy for the fallback
  __syncthreads();
  const int rank_mm = psd_pivoted_cholesky_wave<16>(E15, md, 17, perm, s_flag, 0.0, kMargEps, nullptr, tile);
  if (rank_mm == md) {
    VPL_STAMP(B, w, 34);
    // tmp(i, :) = Arm(i, :) Amm^-1 :  L L^T x = P a, one lane per row.  Compile-time loop bounds (md <= 15 guards) keep x in
    // registers and the reads of L broadcast and pipelined; reciprocal diagonals from LDS: with run-time bounds x lived in
    // scratch and every step paid an LDS round trip and a division (30 k cycles for 45 rows).
    double* rdiag = lam;   // md reciprocals of the diagonal of L (lam is free between the factorisations)
    if (tid < md) rdiag[tid] = 1.0 / E15[tid * 17 + tid];
    __syncthreads();
    for (int i = tid; i < n; i += T) {
      double x[15];
#pragma unroll
      for (int t = 0; t < 15; ++t) {
        x[t] = 0.0;
        if (t < md) {
          double s2 = Ad[(md + i) * ldd + perm[t]];
#pragma unroll
          for (int k = 0; k < t; ++k) s2 -= E15[t * 17 + k] * x[k];
          x[t] = s2 * rdiag[t];
        }
      }
#pragma unroll
      for (int t = 14; t >= 0; --t) {
        if (t < md) {
          double s2 = x[t];
#pragma unroll
          for (int k = t + 1; k < 15; ++k) if (k < md) s2 -= E15[k * 17 + t] * x[k];
          x[t] = s2 * rdiag[t];
        }
      }
#pragma unroll
      for (int t = 0; t < 15; ++t) if (t < md) tmp[i * 16 + perm[t]] = x[t];
    }
    __syncthreads();
  } else {
    for (int it = tid; it < md * md; it += T) E15[(it / md) * 17 + it % md] = Y[it];
    __syncthreads();
    psd_spectral_factor(E15, md, 17, perm, lam, red, s_flag, 0.0, tile);
    VPL_STAMP(B, w, 34);
    // tmp(n x md) = Arm * Amm^+ :  first Y = Arm * B (n x md), then tmp = (Y ./ lambda^2) * B^T
    for (int it = tid; it < n * md; it += T) {
      const int i = it / md, k = it % md;
      double s = 0;
      for (int t = 0; t < md; ++t) s += Ad[(md + i) * ldd + perm[t]] * E15[t * 17 + k];
      Y[i * 16 + k] = lam[k] > kMargEps ? s / (lam[k] * lam[k]) : 0.0;
    }
    __syncthreads();
    for (int it = tid; it < n * md; it += T) {
      const int i = it / md, t = it % md;   // column perm[t] of tmp
      double s = 0;
      for (int k = 0; k < md; ++k) s += Y[i * 16 + k] * E15[t * 17 + k];
      tmp[i * 16 + perm[t]] = s;
    }
    __syncthreads();
  }
  // G <- Arr - tmp * Amr ; b <- brr - tmp * bmm
  double* Aout = B.mg_A + (size_t)w * MAXKEEP * MAXKEEP;
  double* bout = B.mg_b + (size_t)w * MAXKEEP;
  for (int it = tid; it < n * n; it += T) {
    const int i = it / n, j = it % n;
    double s = 0;
    for (int k = 0; k < md; ++k) s += tmp[i * 16 + k] * Ad[k * ldd + md + j];
    const double v = Ad[(md + i) * ldd + md + j] - s;
    Aout[i * n + j] = v;
    G[i * ldm + j] = v;
  }
  double bkeep = 0.0;   // (n <= MAXKEEP < T: one entry per thread; read back from HBM it cost a store completion + a round trip)
  if (tid < n) {
    double s = 0;
    for (int k = 0; k < md; ++k) s += tmp[tid * 16 + k] * bv[k];
    bkeep = bv[md + tid] - s;
    bout[tid] = bkeep;
  }
  __syncthreads();
  if (tid < n) bv[tid] = bkeep;
  // the reference eigen-decomposes A as it is; symmetrise the copy the factorisation works on
  for (int it = tid; it < n * n; it += T) {
    const int i = it / n, j = it % n;
    if (i > j) G[i * ldm + j] = 0.5 * (G[i * ldm + j] + G[j * ldm + i]);
  }
  __syncthreads();
  for (int it = tid; it < n * n; it += T) {
    const int i = it / n, j = it % n;
    if (i < j) G[i * ldm + j] = G[j * ldm + i];
  }
  __syncthreads();
  // ---- factor of the kept block (:349-357) ------------------------------------------------------------------
  // The reference sets J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b from A = V S V^T.  The prior enters every later computation
  // only through |r0 + J0 dx|^2, i.e. through J0^T J0 = A and J0^T r0 = (b projected on range A); any J0' = Q J0,
  // r0' = Q r0 with Q orthogonal is the same prior.  The pivoted Cholesky factor is such a pair and needs no
  // eigen-iteration:  P A P^T = L L^T,  J0 = L^T P^T,  L(0:r,0:r) r0 = (P b)(0:r)  (the forward substitution rides along).
  // The kept block is the difference of numbers five orders larger (A = Arr - Arm Amm^+ Amr after the landmark
  // elimination): what the reference's eigen-solver reports below ~1e-10 lambda_max of it is rounding noise of either
  // sign (it keeps the positive part above 1e-8 with a negligible weight).  Pivots below max(1e-8, 1e-9 max diagonal)
  // end the factorisation; the trailing block is treated as zero.
  VPL_STAMP(B, w, 35);
  // (the dense pre-marginalisation matrix is not needed any more: its space is the one-wave version's scratch)
  // kept blocks of up to 48 dims: one wave, whole columns in registers; up to 76 (the reference's steady state is 75): four
  // waves with half columns (round 4); the work-group version remains for 77..80 and behind VPL_MARG_WG_FACTOR (A/B runs)
  int rank;
#ifdef VPL_MARG_WG_FACTOR
  constexpr bool use_wave4 = false;
#else
  constexpr bool use_wave4 = T >= 512;   // eight waves: two column groups x four row slices
#endif
  if (n <= 48) rank = psd_pivoted_cholesky_wave<48>(G, n, ldm, perm, s_flag, kMargNoiseRel, kMargEps, bv, Ad);
  else if (use_wave4 && n <= 76) {
    #ifdef VPL_STAMPS
    rank = psd_pivoted_cholesky_wave4<19, 4>(G, n, ldm, perm, s_flag, kMargNoiseRel, kMargEps, bv, Ad, s_phase, B.dbg + (size_t)w * 64 + 40);
#else
    rank = psd_pivoted_cholesky_wave4<19, 4>(G, n, ldm, perm, s_flag, kMargNoiseRel, kMargEps, bv, Ad, s_phase);
#endif
    // a hand-shake that timed out (never observed) leaves G and bv as they were -- the columns live in registers until the
    // routine's last pass: the work-group version takes over
    if (rank < 0) rank = psd_pivoted_cholesky(G, n, ldm, perm, red, s_flag, kMargNoiseRel, kMargEps, bv, lam);
  } else rank = psd_pivoted_cholesky(G, n, ldm, perm, red, s_flag, kMargNoiseRel, kMargEps, bv, lam);
  VPL_STAMP(B, w, 36);
  double* J0 = B.mg_J0 + (size_t)w * MAXKEEP * MAXKEEP;
  double* r0 = B.mg_r0 + (size_t)w * MAXKEEP;
  // (the block is cleared first: the loop below covers it only if perm is a permutation, and a column it leaves out must not
  //  keep the prior of the batch the context solved before -- DESIGN.md section 6, the open item of round 4)
  for (int it = tid; it < n * n; it += T) J0[it] = 0.0;
  __syncthreads();
  for (int it = tid; it < n * n; it += T) {
    const int k = it / n, t = it % n;   // J0[k][perm[t]] = L[t][k]
    J0[k * n + perm[t]] = (k < rank && t >= k) ? G[t * ldm + k] : 0.0;
  }
  for (int k = tid; k < n; k += T) r0[k] = k < rank ? bv[k] : 0.0;
  // x0 of the kept blocks: the linearisation point (preMarginalize copies, :110-129)
  if (tid < nb) {
    const int kind = B.mg_kind[(size_t)w * MAXPB + tid];
    const int base = B.mg_cam[(size_t)w * MAXPB + tid];
    const double* x = kind == 0 ? B.pose + ((size_t)w * NF + base / 15) * 7
                      : kind == 1 ? B.sb + ((size_t)w * NF + base / 15) * 9 : B.ex + (size_t)w * 7;
    const int gs = kind == 1 ? 9 : 7;
    for (int k = 0; k < 9; ++k) B.mg_x0[((size_t)w * MAXPB + tid) * 9 + k] = k < gs ? x[k] : 0.0;
  }
  VPL_STAMP(B, w, 37);
}
template <int T>
__global__ __launch_bounds__(T) void k_marg(DevBatch B) {
  extern __shared__ double sm[];
  marg_body<T>(B, blockIdx.x, sm);
}

// Prior handoff on the device (vpl_ba_upload_chained): the prior the previous solve's marginalisation left in mg_* becomes the
// prior of the window that is about to be solved -- J0 (n x n, re-strided from MAXKEEP^2 to the batch's prS), r0, x0.  The
// block tables travel through the host (a few ints per window; upload builds its layout tables from them).
// keep[w] != 0: the window keeps the prior it already has in pr_* (MARGIN_SECOND_NEW left it untouched, estimator.cpp:1385).
__global__ void k_prior_handoff(DevBatch B, const int* keep) {
  const int w = blockIdx.x;
  if (keep[w]) return;
  const int n = B.mg_n[w];
  const double* J = B.mg_J0 + (size_t)w * MAXKEEP * MAXKEEP;
  double* Jo = B.pr_J0 + (size_t)w * B.prS;
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) Jo[i] = J[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) B.pr_r0[(size_t)w * MAXPN + i] = B.mg_r0[(size_t)w * MAXKEEP + i];
  for (int i = threadIdx.x; i < MAXPB * 9; i += blockDim.x) B.pr_x0[(size_t)w * MAXPB * 9 + i] = B.mg_x0[(size_t)w * MAXPB * 9 + i];
}

// Chained upload whose batch stride of pr_J0 changed: the windows that KEEP their prior have J0 at w * old_prS.  phase 0 copies
// their n x n into tmp[w][MAXPN^2], phase 1 (next launch) back to w * B.prS.  pr_n still holds the kept prior's size (the new
// tables are scattered after this).
__global__ void k_prior_restride(DevBatch B, const int* keep, int old_prS, double* tmp, int phase) {
  const int w = blockIdx.x;
  if (!keep[w]) return;
  const int n = B.pr_n[w];
  double* T = tmp + (size_t)w * MAXPN * MAXPN;
  if (phase == 0) {
    const double* J = B.pr_J0 + (size_t)w * old_prS;
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) T[i] = J[i];
  } else {
    double* J = B.pr_J0 + (size_t)w * B.prS;
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) J[i] = T[i];
  }
}

// Per-window states of the solved batch as one [nW][183] device array (pose 77 | speed/bias 99 | extrinsic 7): what the
// multi-GPU run all-gathers over RCCL, packed on the device so that the collective reads HBM, not a host staging copy.
__global__ void k_pack_states(DevBatch B, double* out) {
  const int w = blockIdx.x;
  for (int i = threadIdx.x; i < 183; i += blockDim.x)
    out[(size_t)w * 183 + i] = i < 77 ? B.pose[(size_t)w * 77 + i] : (i < 176 ? B.sb[(size_t)w * 99 + (i - 77)] : B.ex[(size_t)w * 7 + (i - 176)]);
}

}  // namespace vpl
