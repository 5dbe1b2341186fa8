// C ABI of the MI355X-native bundle-adjustment path (include/vplines_ba.h).
// Host side: packs caller-owned windows into the device SoA of ba_types.h, enqueues the
// kernel sequence of one batched solve on the context's stream, unpacks results.
// There is no CPU compute path in this library: every entry point that computes launches
// HIP kernels and reports VPL_E_NODEVICE / VPL_E_HIP when that is impossible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "vplines_ba.h"
#include "ba_types.h"
#include "ba_lin.h"
#include "ba_pack.h"
#include "ba_solve.h"
#include "ba_step.h"
#include "ba_marg.h"
#include "ba_lineopt.h"
#include "ba_factors.h"

using namespace vpl;

// ---- staging ------------------------------------------------------------------------------------------------------------
// Host <-> device traffic of the window entry points goes through ONE pinned host arena and ONE device arena per context:
//   upload   : the host packs every array into the pinned arena, ONE hipMemcpyAsync moves it to the device arena, ONE kernel
//              (k_copy_segments) scatters the pieces into the batch's arrays;
//   download : one kernel gathers the result arrays into the device arena, ONE hipMemcpyAsync brings it to the pinned arena,
//              the host scatters into the caller's structs.
// Round 3 issued ~45 hipMemcpyAsync per upload and ~17 per download straight from / into pageable std::vectors and the
// caller's vpl_prior structs: each pageable copy above the runtime's staging threshold pins and unpins its pages (a kernel
// driver call that can quiesce the process's queues), which showed up as 26 ms instead of 1.4 in the solve leg of a
// 64-window call on the round-3 driver box.  The library now hands pageable memory to the runtime nowhere
// on the upload / solve / download path.
struct CopySeg {
  const char* src;
  char* dst;
  unsigned bytes;
  unsigned u0;   // index of the segment's first 16-byte unit in the launch
};
static_assert(sizeof(CopySeg) == 24, "CopySeg layout");

// one thread per 16-byte unit; the segment of a unit by binary search over the (<= few thousand) prefix entries
__global__ __launch_bounds__(256) void k_copy_segments(const CopySeg* tab, int nseg, unsigned total_units) {
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < total_units; u += gridDim.x * 256u) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].u0 <= u) lo = mid; else hi = mid - 1;
    }
    const CopySeg sg = tab[lo];
    const unsigned off = (u - sg.u0) * 16u;
    const unsigned rem = sg.bytes - off;
    const char* src = sg.src + off;
    char* dst = sg.dst + off;
    if (rem >= 16u && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {   // tails and 4- / 8-byte aligned pieces (every array is made of 4- or 8-byte elements)
      const unsigned nb = rem < 16u ? rem : 16u;
      for (unsigned b = 0; b < nb; b += 4) *reinterpret_cast<uint32_t*>(dst + b) = *reinterpret_cast<const uint32_t*>(src + b);
    }
  }
}

template <typename T>
struct Span {   // a piece of the pinned arena with the few std::vector members the packing code uses
  T* p = nullptr;
  size_t n = 0;
  T& operator[](size_t i) const { return p[i]; }
  T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T* begin() const { return p; }
  T* end() const { return p + n; }
};

struct Stage {
  char* h = nullptr;   // hipHostMalloc
  char* d = nullptr;   // hipMalloc
  size_t cap = 0, used = 0;
  std::vector<CopySeg> segs;
  unsigned units = 0;
  bool overflow = false;

  hipError_t reserve(size_t need) {
    used = 0; segs.clear(); units = 0; overflow = false;
    if (need <= cap) return hipSuccess;
    if (h) hipHostFree(h);
    if (d) hipFree(d);
    h = d = nullptr; cap = 0;
    const size_t want = need + need / 4 + (1u << 16);
    hipError_t e = hipHostMalloc((void**)&h, want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    e = hipMalloc((void**)&d, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (h) hipHostFree(h);
    if (d) hipFree(d);
    h = d = nullptr; cap = 0;
  }
  template <typename T>
  Span<T> take(size_t n) {   // uninitialised
    used = (used + 63) & ~(size_t)63;
    Span<T> sp;
    if (used + n * sizeof(T) > cap) { overflow = true; static T dummy; sp.p = &dummy; sp.n = 0; return sp; }
    sp.p = reinterpret_cast<T*>(h + used);
    sp.n = n;
    used += n * sizeof(T);
    return sp;
  }
  template <typename T>
  Span<T> take(size_t n, T fill) {
    Span<T> sp = take<T>(n);
    std::fill(sp.begin(), sp.end(), fill);
    return sp;
  }
  char* dev_of(const void* host_ptr) const { return d + (reinterpret_cast<const char*>(host_ptr) - h); }
  void seg(const void* src, void* dst, size_t bytes) {
    if (!bytes) return;
    CopySeg c{reinterpret_cast<const char*>(src), reinterpret_cast<char*>(dst), (unsigned)bytes, units};
    units += (unsigned)((bytes + 15) / 16);
    segs.push_back(c);
  }
  // upload: a piece of the arena (host address) to a device array
  template <typename T>
  void to_device(T* dev_dst, const T* host_src, size_t n) { seg(dev_of(host_src), dev_dst, n * sizeof(T)); }
  // download: a device array to a piece of the arena (host address)
  template <typename T>
  void from_device(T* host_dst, const T* dev_src, size_t n) { seg(dev_src, dev_of(host_dst), n * sizeof(T)); }
};

struct vpl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int maxW = 0, maxP = 0, maxPO = 0, maxL = 0, maxLO = 0;
  DevBatch B;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;   // payload of allocs[i]; 64 pad bytes follow (VPL_DEBUG_GUARDS=1: filled with 0xA5, vpl_ba_debug_guards)
  bool guards = false;
  int nW = 0;
  vpl_ba_options opt;
  std::string err;
  bool timing = false;
  std::map<std::string, std::pair<double, int>> ktimes;
  std::vector<std::pair<const char*, double>> ltimes;   // (kernel, ms) of every launch of the last timed solve, in order
  int* d_act = nullptr;                                 // [ACT_SLOTS][4] activity counters of those launches
  // the ~19 launches of one solve as a hipGraph, captured on the first vpl_ba_solve after an upload (the kernel arguments --
  // the batch descriptor by value -- are fixed until the next upload); VPL_BA_GRAPH=0 launches kernel by kernel
  hipGraphExec_t graph_exec = nullptr;
  bool use_graph = true;
  // signature of the track layout (start frames, lengths, selected lines) of the last upload: when the next batch has the same
  // one -- the usual case between two solves of a tracker that lost and gained nothing, and every repetition of a benchmark --
  // the host-built lane / unit / K-step tables and the index arrays already on the device are the right ones and are neither
  // rebuilt nor uploaded again
  unsigned long long layout_sig = 0;
  std::vector<int> layout_key;                   // the integers the signature was made of (exact comparison on a hash match)
  bool layout_valid = false;
  bool force_general = false;                    // VPL_BA_GENERAL=1: every window takes k_solve (A/B runs, tests of the general path)
  bool schur_mostly_wide = false;                // more than 35 % of the landmark elimination's weight sits in wide entries: k_schur<5>
  bool schur_never_wide = false;                 // VPL_BA_SCHUR_WIDE=-1: k_schur_mixed whatever the share of wide entries (A/B runs)
  bool schur_wide_all = false;                   // VPL_BA_SCHUR_WIDE=1: round 3's k_schur<5> for batches with long tracks (A/B runs, tests)
  std::vector<std::string> kname_store;
  // host-side marg structure of the uploaded windows
  std::vector<int> h_mg_m;
  std::vector<int> h_passthrough;              // MARGIN_SECOND_NEW: window keeps its input prior (index into h_pass_priors or -1)
  std::vector<vpl_prior> h_pass_priors;
  bool any_second_new = false;
  std::vector<int> h_nP, h_nL;
  std::vector<std::vector<int>> h_lmap;          // per window: device line index -> index in the vpl_window arrays
  size_t marg_smem = 0;
  bool marg_small = false;                       // k_marg<256> (two work-groups per CU) instead of k_marg<512>
  int maxPriorN = 0;                             // largest prior of the uploaded batch (k_prep stages J0 in LDS)
  // asynchronous variants of the line-map entry points: the host-side completion (wait for the stream, scatter the staged
  // results into the caller's arrays) of the call that was enqueued last; run by vpl_ba_collect or by the next call that
  // touches the batch
  std::function<int()> pending;
  bool upload_open = false;                      // an upload has started to rewrite the host tables of the batch and has not finished
  bool prior_resident = false;                   // the last solve / marginalisation of the uploaded batch left its priors in mg_* (vpl_ba_upload_chained)
  int prior_resident_nW = 0;
  double odo_ms[3] = {0, 0, 0};                  // vpl_ba_debug_odometry_ms
  Stage stage;                                   // pinned + device staging arenas of upload / download
  std::vector<int> h_mg_n;                       // kept dims of the next prior as the host computed them (>= the device's)
  // device time of the last upload / solve / download (hipEvents on the context's stream), vpl_ctx_enable_leg_timing
  bool leg_timing = false;
  hipEvent_t leg_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// appends the segment table to the arena, moves arena (upload: data + table, download: table only) and runs the copy kernel
static hipError_t stage_run(vpl_ctx* c, bool upload) {
  Stage& S = c->stage;
  if (S.overflow) return hipErrorOutOfMemory;
  if (S.segs.empty()) return hipSuccess;
  const size_t data_end = S.used;
  Span<CopySeg> tab = S.take<CopySeg>(S.segs.size());
  if (S.overflow) return hipErrorOutOfMemory;
  std::memcpy(tab.p, S.segs.data(), S.segs.size() * sizeof(CopySeg));
  const size_t tab_off = reinterpret_cast<char*>(tab.p) - S.h;
  hipError_t e;
  if (upload) e = hipMemcpyAsync(S.d, S.h, S.used, hipMemcpyHostToDevice, c->stream);
  else e = hipMemcpyAsync(S.d + tab_off, S.h + tab_off, S.segs.size() * sizeof(CopySeg), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return e;
  const unsigned blocks = std::min<unsigned>((S.units + 255) / 256, 2048u);
  hipLaunchKernelGGL(k_copy_segments, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const CopySeg*>(S.d + tab_off),
                     (int)S.segs.size(), S.units);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (!upload) e = hipMemcpyAsync(S.h, S.d, data_end, hipMemcpyDeviceToHost, c->stream);
  return e;
}

static void drop_graph(vpl_ctx* c) {
  if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
}

static int fail(vpl_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

// completes the asynchronous call that is still pending on this context, if any
static int settle(vpl_ctx* c) {
  if (!c || !c->pending) return VPL_OK;
  std::function<int()> fin;
  fin.swap(c->pending);
  return fin();
}
// the tail of an entry point: now, or (asynchronous variant) when the caller collects
static int finish_or_defer(vpl_ctx* c, bool async, std::function<int()> fin) {
  if (!async) return fin();
  c->pending = std::move(fin);
  return VPL_OK;
}
#define HIPCHK(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) return fail(ctx, VPL_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// One hipMalloc per array.  (Measured alternative, round 2: all ~90 arrays out of one arena, with and without skewed
// offsets -- k_cost, the kernel closest to the bandwidth roof, then runs at 0.30 ms per step in EVERY process, whereas with
// separate allocations it is 0.24 ms in most processes and 0.30 ms in some: the difference is where the driver places the
// buffers, not the code.  DESIGN.md section 8.)
template <typename T>
static hipError_t dalloc(vpl_ctx* c, T** p, size_t n) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, n * sizeof(T) + 64);
  if (e != hipSuccess) return e;
  e = hipMemset(q, 0, n * sizeof(T) + 64);
  if (e == hipSuccess && c->guards) e = hipMemset((char*)q + n * sizeof(T), 0xA5, 64);
  c->allocs.push_back(q);
  c->alloc_bytes.push_back(n * sizeof(T));
  *p = (T*)q;
  return e;
}

// ---- helpers -----------------------------------------------------------------------------------
static void to_dev_preint(const vpl_preintegration& p, DevPreint& d) {
  d.sum_dt = p.sum_dt;
  for (int k = 0; k < 3; ++k) { d.dp[k] = p.delta_p[k]; d.dv[k] = p.delta_v[k]; d.lba[k] = p.linearized_ba[k]; d.lbg[k] = p.linearized_bg[k]; }
  for (int k = 0; k < 4; ++k) d.dq[k] = p.delta_q[k];
  auto blk = [&](double* o, int r0, int c0) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o[3 * i + j] = p.jacobian[(r0 + i) * 15 + c0 + j];
  };
  blk(d.dp_dba, 0, 9); blk(d.dp_dbg, 0, 12); blk(d.dq_dbg, 3, 12); blk(d.dv_dba, 6, 9); blk(d.dv_dbg, 6, 12);
  std::memcpy(d.cov, p.covariance, sizeof(d.cov));
  std::memset(d.sqrt_info, 0, sizeof(d.sqrt_info));
}

template <typename T>
static hipError_t up(vpl_ctx* c, T* dst, const Span<T>& src) {   // a piece of the pinned arena: joins the upload's one copy
  c->stage.to_device(dst, src.p, src.n);
  return hipSuccess;
}
template <typename T>
static hipError_t up(vpl_ctx* c, T* dst, const std::vector<T>& src) {
  if (src.empty()) return hipSuccess;
  return hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c->stream);
}

struct KTimer {
  vpl_ctx* c;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  KTimer(vpl_ctx* c_, const char* n) : c(c_), name(n) {
    if (c->timing) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->stream); }
  }
  ~KTimer() {
    if (c->timing) {
      hipEventRecord(b, c->stream);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      auto& e = c->ktimes[name];
      e.first += ms;
      e.second += 1;
      c->ltimes.emplace_back(name, (double)ms);
      hipEventDestroy(a);
      hipEventDestroy(b);
    }
  }
};

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  double* d() { return (double*)p; }
};

template <typename Launch>
static int eval_generic(vpl_ctx* c, int n, const double* params, int psz, const double* consts, int csz, int nres,
                        int njac, double* residuals, double* jac, Launch launch) {
  if (!c || n < 0 || !params || !consts || !residuals) return VPL_E_INVALID;
  if (n == 0) return VPL_OK;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf dp, dc, dr, dj;
  HIPCHK(c, dp.alloc((size_t)n * psz * 8));
  HIPCHK(c, dc.alloc((size_t)n * csz * 8));
  HIPCHK(c, dr.alloc((size_t)n * nres * 8));
  if (jac) HIPCHK(c, dj.alloc((size_t)n * njac * 8));
  HIPCHK(c, hipMemcpyAsync(dp.p, params, (size_t)n * psz * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dc.p, consts, (size_t)n * csz * 8, hipMemcpyHostToDevice, c->stream));
  launch(dp.d(), dc.d(), dr.d(), jac ? dj.d() : nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(residuals, dr.p, (size_t)n * nres * 8, hipMemcpyDeviceToHost, c->stream));
  if (jac) HIPCHK(c, hipMemcpyAsync(jac, dj.p, (size_t)n * njac * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}

extern "C" {

void vpl_ba_default_options(vpl_ba_options* o) {
  o->num_iterations = 5;
  o->estimate_extrinsic = 1;
  o->marginalization_flag = VPL_MARGIN_OLD;
  o->remove_line_outliers = 0;
  o->focal_length = 460.0;
  o->line_factor = 306.666666667;
  o->vp_factor = 10.0;
  o->g_norm = 9.81007;
  o->acc_n = 0.08; o->gyr_n = 0.004; o->acc_w = 0.00004; o->gyr_w = 2.0e-6;
  o->huber_delta = 1.0;
}

int vpl_ctx_create(vpl_ctx** out, int device, int max_windows, int max_points, int max_point_obs, int max_lines,
                   int max_line_obs) {
  if (!out || max_windows < 1 || max_points < 0 || max_lines < 0) return VPL_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device >= ndev) return VPL_E_NODEVICE;
  vpl_ctx* c = new vpl_ctx();
  { const char* g = getenv("VPL_DEBUG_GUARDS"); c->guards = g && g[0] == '1'; }
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return VPL_E_NODEVICE; }
  c->maxW = max_windows;
  c->maxP = max_points > 0 ? (max_points + 1) & ~1 : 2;   // even: k_solve streams the gradient entries in 16-byte units
  c->maxPO = max_point_obs > 0 ? max_point_obs : 1;
  c->maxL = max_lines > 0 ? max_lines : 1;
  c->maxLO = max_line_obs > 0 ? max_line_obs : 1;
  // k_solve keeps 2 doubles per point and 18 per line in the LDS space behind its two staging buffers
  if (2 * c->maxP + 18 * c->maxL > NAP - 2 * CROWS * CW) { delete c; return VPL_E_CAPACITY; }
  // ... and 4 doubles per point and 28 per line in the staging buffers' space between the scaling and the first chunk
  if (4 * c->maxP + 28 * c->maxL > 2 * CROWS * CW) { delete c; return VPL_E_CAPACITY; }
  if (solve_smem(c->maxP, c->maxL) > 159 * 1024) { delete c; return VPL_E_CAPACITY; }
  DevBatch& B = c->B;
  std::memset(&B, 0, sizeof(B));
  B.maxP = c->maxP; B.maxPO = c->maxPO; B.maxL = c->maxL; B.maxLO = c->maxLO;
  B.nfull = NC + B.maxP + 4 * B.maxL;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    if (const char* sv = std::getenv("VPL_BA_LIN_SPLIT")) ncu = std::atoi(sv) ? 1 << 30 : 0;   // 1: always two work-groups per window, 0: never (A/B runs, tests)
    B.ncu = ncu;
  }
  // point work units of k_lin: (start frame, chunk of <= 16 tracks, observation); a unit needs at most one quarter-wave slot
  // slots <= factor lanes / 16 + one partial unit per (start, k) pair; the halves of the work-group differ by less than one chunk
  B.maxPR = std::min((B.maxP / 16 + NF) * (NF - 1), B.maxPO / 16 + NF * (NF - 1) / 2) / 32 + 2;
  const size_t W = max_windows;
  hipError_t e = hipSuccess;
#define AL(ptr, n) if (e == hipSuccess) e = dalloc(c, &B.ptr, (size_t)(n))
#define AL2(ptr, n) if (e == hipSuccess) e = dalloc(c, &B.ptr, (size_t)(n))
  AL(pose, W * 77); AL(sb, W * 99); AL(ex, W * 7); AL(invd, W * B.maxP); AL(orth, W * B.maxL * 4);
  AL(pose_c, W * 77); AL(sb_c, W * 99); AL(ex_c, W * 7); AL(invd_c, W * B.maxP); AL(orth_c, W * B.maxL * 4); AL(lw, W * B.maxL * 6); AL(lw_c, W * B.maxL * 6);
  AL(lin_part, W * LIN_PART); AL(lin_pcost, W); AL(lin_flag, W);
  AL(pose_0, W * 77); AL(sb_0, W * 99); AL(ex_0, W * 7); AL(invd_0, W * B.maxP); AL(plk_0, W * B.maxL * 6);
  AL(plk, W * B.maxL * 6); AL(gauge, W * 4); AL(fail_ref, W * 13); AL(orth_in, W);
  AL(nP, W); AL(nL, W);
  AL(pt_start, W * B.maxP); AL(pt_nobs, W * B.maxP); AL(pt_off, W * B.maxP); AL(pt_obs, W * B.maxPO * 3);
  AL(ps_list, W * B.maxP); AL(ps_cnt, W * (NF + 1)); AL(pu_lane, W * B.maxPR * 1024); AL(pu_sub, W * B.maxPR * 512); AL(pu_cnt, W); AL(pu_cnt0, W);
  AL(ln_start, W * B.maxL); AL(ln_nobs, W * B.maxL); AL(ln_off, W * B.maxL); AL(ln_obs, W * B.maxLO * 8);
  AL(nLO, W); AL(lo_ln, W * B.maxLO);
  B.llSlots = 512 * ((B.maxL + 8 * (64 / NF) - 1) / (8 * (64 / NF)));   // worst case: 11-frame tracks, 5 lines per wave
  AL(ll_tab, W * B.llSlots * 2); AL(ll_np, W);
  AL(pre, W * NF);
  AL(pr_n, W); AL(pr_nb, W); AL(pr_kind, W * MAXPB); AL(pr_frame, W * MAXPB); AL(pr_idx, W * MAXPB);
  AL(pr_x0, W * MAXPB * 9); AL(pr_J0, W * MAXPN * MAXPN); AL(pr_r0, W * MAXPN); AL(pr_H, W * MAXPN * MAXPN); AL(pr_g0, W * MAXPN);
  AL(pr_map, W * MAXPN);
  AL(Hcc, W * NCP); AL(gc, W * NC); AL(asm_tab, 2 * NCP); AL(Hpp, W * B.maxP); AL(gp, W * B.maxP); AL(Wp, W * B.maxP * NV);
  AL(Hll, W * B.maxL * 16); AL(gl, W * B.maxL * 4); AL(Wl, W * B.maxL * 4 * NV); AL(lchol, W * B.maxL * 10);
  AL(tr, W);
  AL(scale, W * B.nfull); AL(diag, W * B.nfull); AL(grad, W * B.nfull); AL(gn, W * B.nfull);
  AL(mg_n, W); AL(mg_nb, W); AL(mg_kind, W * MAXPB); AL(mg_frame, W * MAXPB); AL(mg_idx, W * MAXPB);
  AL(mg_cam, W * MAXPB); AL(mg_x0, W * MAXPB * 9); AL(mg_J0, W * MAXKEEP * MAXKEEP); AL(mg_r0, W * MAXKEEP);
  AL(mg_A, W * MAXKEEP * MAXKEEP); AL(mg_b, W * MAXKEEP); AL(mg_m, W); AL(dbg, W * 64); AL(ln_removed, W * B.maxL); AL(ln_tri, W * B.maxL);
#undef AL
  if (e == hipSuccess) e = dalloc(c, &c->d_act, (size_t)ACT_SLOTS * 4);
  AL2(order, 2 * W); AL2(ord_cnt, 4);
  B.maxKS = B.maxP / 4 + B.maxL + NF + 2;
  AL2(sk_tab, W * B.maxKS * 4); AL2(sk_wave, W * 8 * SK_WSTRIDE); AL2(sacc, W * SACC_N); AL2(nz_tab, NZ_N); AL2(ycs, W * 176); AL2(sx, W * 8); AL2(path, W);
  if (e != hipSuccess) {
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return VPL_E_HIP;
  }
  if (lin_smem_base(c->maxP, c->maxL) > LIN_LDS_BUDGET) { for (void* p : c->allocs) hipFree(p); delete c; return VPL_E_CAPACITY; }
  B.prhN = lin_prh_n(c->maxP, c->maxL);
  {   // static table of the assembly pass of k_lin
    std::vector<int> tab(2 * NCP);
    for (int r = 0, e2 = 0; r < NC; ++r)
      for (int cc = 0; cc <= r; ++cc, ++e2) lin_asm_entry(r, cc, &tab[2 * e2]);
    if (hipMemcpy(B.asm_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      for (void* p : c->allocs) hipFree(p);
      delete c;
      return VPL_E_HIP;
    }
  }
  {   // non-zero pattern of the packed cam Hessian of a fast-path window (ba_types.h)
    std::vector<int> nz;
    for (int r = 0; r < NC; ++r)
      for (int cc = 0; cc <= r; ++cc) {
        const bool vis = cam2vis(r) >= 0 && cam2vis(cc) >= 0;
        const int fr = r < 165 ? r / 15 : -1, fc = cc < 165 ? cc / 15 : -1;
        const bool band = fr >= 0 && fc >= 0 && (fr == fc || fr == fc + 1);
        const bool sb0 = cc >= 6 && cc < 15 && cam2vis(r) >= 0;
        if (vis || band || sb0) nz.push_back(tri(r, cc) | r << 14 | cc << 22);
      }
    if ((int)nz.size() != NZ_N || hipMemcpy(B.nz_tab, nz.data(), nz.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      for (void* p : c->allocs) hipFree(p);
      delete c;
      return VPL_E_HIP;
    }
  }
  // the attribute is per kernel, not per context: never lower what a larger context of this process has asked for
  static size_t lin_max = 0, solve_max = 0;
  lin_max = std::max(lin_max, lin_smem(c->maxP, c->maxL));
  solve_max = std::max(solve_max, solve_smem(c->maxP, c->maxL));
  hipFuncSetAttribute((const void*)k_lin2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lin_max);
  hipFuncSetAttribute((const void*)k_lin<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lin_max);
  hipFuncSetAttribute((const void*)k_lin<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lin_max);
  hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_max);
  static size_t schur_max = 0, back_max = 0;
  schur_max = std::max(schur_max, schur_smem(c->maxP, c->maxL));
  back_max = std::max(back_max, back_smem(c->maxP, c->maxL));
  if (schur_max > 159 * 1024 || back_max > 159 * 1024) { for (void* p : c->allocs) hipFree(p); delete c; return VPL_E_CAPACITY; }
  hipFuncSetAttribute((const void*)k_schur<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_max);
  hipFuncSetAttribute((const void*)k_schur<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_max);
  hipFuncSetAttribute((const void*)k_schur_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_max);
  hipFuncSetAttribute((const void*)k_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_SMEM);
  hipFuncSetAttribute((const void*)k_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)back_max);
  hipFuncSetAttribute((const void*)k_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_SMEM);
  hipFuncSetAttribute((const void*)k_marg<MARG_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipFuncSetAttribute((const void*)k_marg<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MARG_LDS_SMALL);
  (void)hipGetLastError();
  vpl_ba_default_options(&c->opt);
  if (const char* gv = std::getenv("VPL_BA_GRAPH")) c->use_graph = std::atoi(gv) != 0;
  if (const char* gv = std::getenv("VPL_BA_GENERAL")) c->force_general = std::atoi(gv) != 0;
  if (const char* gv = std::getenv("VPL_BA_SCHUR_WIDE")) { c->schur_wide_all = std::atoi(gv) > 0; c->schur_never_wide = std::atoi(gv) < 0; }
  *out = c;
  return VPL_OK;
}

void vpl_ctx_destroy(vpl_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  drop_graph(c);
  for (void* p : c->allocs) hipFree(p);
  c->stage.release();
  for (hipEvent_t& e : c->leg_ev) if (e) hipEventDestroy(e);
  delete c;
}

int vpl_ctx_set_stream(vpl_ctx* c, void* s) {
  if (!c) return VPL_E_INVALID;
  // what is in flight on the stream so far is completed first (an enqueued call would otherwise be collected behind a
  // synchronisation of the NEW stream, and a later download would not be ordered behind a solve on the old one)
  const int rs = settle(c);
  if (rs) return rs;
  if (c->stream != (hipStream_t)s) {
    HIPCHK(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);   // (an old stream the caller has destroyed already has nothing in flight)
    (void)hipGetLastError();
  }
  drop_graph(c);
  c->stream = (hipStream_t)s;
  return VPL_OK;
}
const char* vpl_last_error(const vpl_ctx* c) { return c ? c->err.c_str() : "null context"; }

// device time of the legs of the last vpl_ba_upload / vpl_ba_solve / vpl_ba_download (hipEvents on the context's stream around
// each call's device work: copy + scatter | the solve's launches | gather + copy), next to the wall clock a caller measures
int vpl_ctx_enable_leg_timing(vpl_ctx* c, int enable) {
  if (!c) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  if (enable)
    for (hipEvent_t& e : c->leg_ev) if (!e) HIPCHK(c, hipEventCreate(&e));
  c->leg_timing = enable != 0;
  return VPL_OK;
}
int vpl_ctx_leg_times(vpl_ctx* c, double* ms3) {
  if (!c || !ms3 || !c->leg_timing) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 3; ++k) {
    float ms = 0.f;
    const hipError_t e = hipEventElapsedTime(&ms, c->leg_ev[2 * k], c->leg_ev[2 * k + 1]);
    ms3[k] = e == hipSuccess ? (double)ms : -1.0;   // -1: that leg has not run with timing on
  }
  return VPL_OK;
}

int vpl_ctx_synchronize(vpl_ctx* c) {
  if (!c) return VPL_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return settle(c);
}
int vpl_ba_collect(vpl_ctx* c) {
  if (!c) return VPL_E_INVALID;
  return settle(c);
}

// ---- IMU pre-integration ---------------------------------------------------------------------------
int vpl_preintegrate_batch(vpl_ctx* c, int n, const int* offset, const int* nsamples, const double* samples,
                           const double* acc0, const double* gyr0, const double* lin_ba, const double* lin_bg,
                           const vpl_ba_options* opt, vpl_preintegration* out) {
  if (!c || n < 0 || !opt) return VPL_E_INVALID;
  if (n == 0) return VPL_OK;
  HIPCHK(c, hipSetDevice(c->device));
  size_t total = 0;
  for (int i = 0; i < n; ++i) total = std::max(total, (size_t)offset[i] + nsamples[i]);
  int *d_off, *d_ns;
  double *d_s, *d_a, *d_g, *d_ba, *d_bg;
  DevPreint* d_out;
  HIPCHK(c, hipMalloc(&d_off, n * sizeof(int)));
  HIPCHK(c, hipMalloc(&d_ns, n * sizeof(int)));
  HIPCHK(c, hipMalloc(&d_s, total * 7 * sizeof(double) + 8));
  HIPCHK(c, hipMalloc(&d_a, n * 3 * sizeof(double)));
  HIPCHK(c, hipMalloc(&d_g, n * 3 * sizeof(double)));
  HIPCHK(c, hipMalloc(&d_ba, n * 3 * sizeof(double)));
  HIPCHK(c, hipMalloc(&d_bg, n * 3 * sizeof(double)));
  HIPCHK(c, hipMalloc(&d_out, n * sizeof(DevPreint)));
  HIPCHK(c, hipMemcpyAsync(d_off, offset, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_ns, nsamples, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_s, samples, total * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_a, acc0, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_g, gyr0, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_ba, lin_ba, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_bg, lin_bg, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  { KTimer t(c, "k_preintegrate");
    hipLaunchKernelGGL(k_preintegrate, dim3((n + 3) / 4), dim3(64), 0, c->stream, n, d_off, d_ns, d_s, d_a, d_g, d_ba,
                       d_bg, opt->acc_n * opt->acc_n, opt->gyr_n * opt->gyr_n, opt->acc_w * opt->acc_w,
                       opt->gyr_w * opt->gyr_w, d_out); }
  HIPCHK(c, hipGetLastError());
  std::vector<DevPreint> h(n);
  HIPCHK(c, hipMemcpyAsync(h.data(), d_out, n * sizeof(DevPreint), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) {
    vpl_preintegration& o = out[i];
    o.sum_dt = h[i].sum_dt;
    for (int k = 0; k < 3; ++k) { o.delta_p[k] = h[i].dp[k]; o.delta_v[k] = h[i].dv[k]; o.linearized_ba[k] = h[i].lba[k]; o.linearized_bg[k] = h[i].lbg[k]; }
    for (int k = 0; k < 4; ++k) o.delta_q[k] = h[i].dq[k];
    std::memcpy(o.jacobian, h[i].sqrt_info, sizeof(o.jacobian));   // k_preintegrate carries J out in this slot
    std::memcpy(o.covariance, h[i].cov, sizeof(o.covariance));
  }
  hipFree(d_off); hipFree(d_ns); hipFree(d_s); hipFree(d_a); hipFree(d_g); hipFree(d_ba); hipFree(d_bg); hipFree(d_out);
  return VPL_OK;
}

// ---- single-factor evaluators ------------------------------------------------------------------------
int vpl_projection_factor_evaluate(vpl_ctx* c, int n, const double* params, const double* pts, double sqrt_info,
                                   double* residuals, double* jac) {
  return eval_generic(c, n, params, 22, pts, 6, 2, 44, residuals, jac, [&](double* p, double* k, double* r, double* j) {
    hipLaunchKernelGGL(k_eval_projection, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, p, k, sqrt_info, r, j);
  });
}
int vpl_line_factor_evaluate(vpl_ctx* c, int n, const double* params, const double* obs, double sqrt_info,
                             double* residuals, double* jac) {
  return eval_generic(c, n, params, 18, obs, 4, 2, 36, residuals, jac, [&](double* p, double* k, double* r, double* j) {
    hipLaunchKernelGGL(k_eval_line<0>, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, p, k, sqrt_info, r, j);
  });
}
int vpl_vp_factor_evaluate(vpl_ctx* c, int n, const double* params, const double* vp, double sqrt_info,
                           double* residuals, double* jac) {
  return eval_generic(c, n, params, 18, vp, 3, 2, 36, residuals, jac, [&](double* p, double* k, double* r, double* j) {
    hipLaunchKernelGGL(k_eval_line<1>, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, p, k, sqrt_info, r, j);
  });
}
int vpl_imu_factor_evaluate(vpl_ctx* c, int n, const double* params, const vpl_preintegration* pre, double g_norm,
                            double* residuals, double* jac) {
  if (!c || n < 0 || !params || !pre || !residuals) return VPL_E_INVALID;
  if (n == 0) return VPL_OK;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<DevPreint> h(n);
  for (int i = 0; i < n; ++i) to_dev_preint(pre[i], h[i]);
  DevBuf dp, dpre, dr, dj, ds;
  HIPCHK(c, dp.alloc((size_t)n * 32 * 8));
  HIPCHK(c, dpre.alloc((size_t)n * sizeof(DevPreint)));
  HIPCHK(c, dr.alloc((size_t)n * 15 * 8));
  HIPCHK(c, ds.alloc((size_t)n * 675 * 8));
  if (jac) HIPCHK(c, dj.alloc((size_t)n * 480 * 8));
  HIPCHK(c, hipMemcpyAsync(dp.p, params, (size_t)n * 32 * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dpre.p, h.data(), (size_t)n * sizeof(DevPreint), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_eval_imu, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, dp.d(), (const DevPreint*)dpre.p,
                     g_norm, dr.d(), jac ? dj.d() : nullptr, ds.d());
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(residuals, dr.p, (size_t)n * 15 * 8, hipMemcpyDeviceToHost, c->stream));
  if (jac) HIPCHK(c, hipMemcpyAsync(jac, dj.p, (size_t)n * 480 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}
int vpl_prior_factor_evaluate(vpl_ctx* c, const vpl_prior* pr, const double* params, double* residuals, double* jac) {
  if (!c || !pr || !params || !residuals || pr->n < 0 || pr->n > MAXPN || pr->n_blocks > MAXPB) return VPL_E_INVALID;
  if (pr->n == 0) return VPL_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const int n = pr->n, nb = pr->n_blocks;
  int psz = 0;
  for (int b = 0; b < nb; ++b) psz += pr->block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7;
  DevBuf dk, di, dx0, dJ, dr0, dp, dr, dj;
  HIPCHK(c, dk.alloc(nb * 4)); HIPCHK(c, di.alloc(nb * 4)); HIPCHK(c, dx0.alloc(nb * 9 * 8));
  HIPCHK(c, dJ.alloc((size_t)n * n * 8)); HIPCHK(c, dr0.alloc(n * 8)); HIPCHK(c, dp.alloc(psz * 8));
  HIPCHK(c, dr.alloc(n * 8));
  if (jac) HIPCHK(c, dj.alloc((size_t)n * psz * 8));
  HIPCHK(c, hipMemcpyAsync(dk.p, pr->block_kind, nb * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(di.p, pr->block_idx, nb * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dx0.p, pr->x0, nb * 9 * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dJ.p, pr->J0, (size_t)n * n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dr0.p, pr->r0, n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dp.p, params, psz * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_eval_prior, dim3(1), dim3(256), 0, c->stream, n, nb, (const int*)dk.p, (const int*)di.p, dx0.d(),
                     dJ.d(), dr0.d(), dp.d(), dr.d(), jac ? dj.d() : nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(residuals, dr.p, n * 8, hipMemcpyDeviceToHost, c->stream));
  if (jac) HIPCHK(c, hipMemcpyAsync(jac, dj.p, (size_t)n * psz * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}
int vpl_pose_plus(vpl_ctx* c, int n, const double* x, const double* delta, double* out) {
  return eval_generic(c, n, x, 7, delta, 6, 7, 0, out, nullptr, [&](double* p, double* k, double* r, double*) {
    hipLaunchKernelGGL(k_eval_pose_plus, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, p, k, r);
  });
}
int vpl_line_orth_plus(vpl_ctx* c, int n, const double* x, const double* delta, double* out) {
  return eval_generic(c, n, x, 4, delta, 4, 4, 0, out, nullptr, [&](double* p, double* k, double* r, double*) {
    hipLaunchKernelGGL(k_eval_orth_plus, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, p, k, r);
  });
}

// ---- window batch: upload / solve / download ------------------------------------------------------------
static int upload_body(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt, bool all_lines, bool chained);
// An upload that is refused half way (a track outside the window in the third window, a prior of impossible size) has already
// rewritten part of the batch's host tables and its size: the context then holds NO batch -- solve / download / reset return
// VPL_E_INVALID until the next successful upload -- instead of running the new size on the old device data.  A refusal before
// anything was touched (too many windows, unknown flag, no resident prior for a chained upload) leaves the previous batch as it was.
static int upload_impl(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt, bool all_lines, bool chained = false) {
  const int rc = upload_body(c, nW, win, opt, all_lines, chained);
  if (c) {
    if (rc != VPL_OK && c->upload_open) { c->nW = 0; c->prior_resident = false; }
    c->upload_open = false;
  }
  return rc;
}
static int upload_body(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt, bool all_lines, bool chained) {
  if (!c || !win || !opt || nW < 1) return VPL_E_INVALID;
  if (nW > c->maxW) return fail(c, VPL_E_CAPACITY, "more windows than max_windows");
  if (opt->marginalization_flag != VPL_MARGIN_OLD && opt->marginalization_flag != VPL_MARGIN_SECOND_NEW &&
      opt->marginalization_flag != VPL_MARGIN_NONE)
    return fail(c, VPL_E_INVALID, "unknown marginalization_flag");
  HIPCHK(c, hipSetDevice(c->device));
  { const int rs = settle(c); if (rs) return rs; }   // an asynchronous call whose results have not been collected yet
  // chained: the priors of THIS batch size must be resident, i.e. the context's last upload was solved (or marginalised) with a
  // marginalisation and nothing was uploaded since (any upload rewrites mg_n / mg_nb / mg_kind, the tables of the NEXT prior)
  if (chained && (!c->prior_resident || c->prior_resident_nW != nW))
    return fail(c, VPL_E_INVALID, "upload_chained: needs a previous solve of the same batch size with a marginalisation, and no upload since");
  const int prev_prS = c->B.prS;
  c->prior_resident = false;
  if (c->leg_timing) HIPCHK(c, hipEventRecord(c->leg_ev[0], c->stream));
  drop_graph(c);
  c->upload_open = true;
  c->opt = *opt;
  c->nW = nW;
  DevBatch& B = c->B;
  B.nW = nW;
  B.opt.num_iterations = opt->num_iterations;
  B.opt.estimate_extrinsic = opt->estimate_extrinsic;
  B.opt.marginalization_flag = opt->marginalization_flag;
  B.opt.remove_line_outliers = opt->remove_line_outliers;
  B.opt.sqrt_info_point = opt->focal_length / 1.5;
  B.opt.sqrt_info_line = opt->line_factor;
  B.opt.sqrt_info_vp = opt->vp_factor;
  B.opt.g_norm = opt->g_norm;
  B.opt.huber_delta = opt->huber_delta;

  const size_t W = nW;
  // Every array that travels is packed straight into the context's pinned arena (Stage); the bound below is the sum of the
  // pieces taken from it (+ 64-byte alignment per piece, + the segment table).
  Stage& SG = c->stage;
  {
    size_t j0 = 0;
    if (!chained)
      for (size_t w = 0; w < W; ++w)
        if (win[w].has_prior && win[w].prior && win[w].prior->n > 0 && win[w].prior->n <= MAXPN) j0 += (size_t)win[w].prior->n * win[w].prior->n * 8 + 64;
    const size_t dbl = 77 + 99 + 7 + B.maxP + 6 * B.maxL + 3 * B.maxPO + 8 * B.maxLO + 13 + 4 * B.maxL + 9 * MAXPB + MAXPN;
    const size_t ints = 2 + 3 * B.maxP + 4 * B.maxL + 2 + 4 * B.maxKS + 8 * SK_WSTRIDE + 1 + 1 + B.maxLO + 2 * B.llSlots + 1 + B.maxP + (NF + 1) +
                        1 + 2 + 3 * MAXPB + 2 + 4 * MAXPB + 1 + 1 + B.maxPR * 1536;
    const size_t perW = dbl * 8 + ints * 4 + NF * sizeof(DevPreint);
    const size_t need = W * perW + j0 + 96 * 64 + (96 + 4 * W) * sizeof(CopySeg) + 4096;
    HIPCHK(c, SG.reserve(need));
  }
  Span<double> pose = SG.take<double>(W * 77), sb = SG.take<double>(W * 99), ex = SG.take<double>(W * 7), invd = SG.take<double>(W * B.maxP, 1.0),
               plk = SG.take<double>(W * B.maxL * 6, 0.0);
  Span<int> nP = SG.take<int>(W), nL = SG.take<int>(W), pt_start = SG.take<int>(W * B.maxP, 0), pt_nobs = SG.take<int>(W * B.maxP, 0),
            pt_off = SG.take<int>(W * B.maxP, 0);
  Span<int> ln_start = SG.take<int>(W * B.maxL, 0), ln_nobs = SG.take<int>(W * B.maxL, 0), ln_off = SG.take<int>(W * B.maxL, 0),
            ln_tri = SG.take<int>(W * B.maxL, 1);
  // (plain heap, no value-initialisation: only the rounds a window uses are filled; the used rounds are packed into the arena
  // after the loop)
  std::unique_ptr<int[]> pu_lane(new int[W * B.maxPR * 1024]), pu_sub(new int[W * B.maxPR * 512]);
  Span<int> pu_cnt = SG.take<int>(W, 0), pu_cnt0 = SG.take<int>(W, 0);
  Span<int> sk_tab = SG.take<int>(W * B.maxKS * 4, 0), sk_wave = SG.take<int>(W * 8 * SK_WSTRIDE, -1), path = SG.take<int>(W, 0);
  Span<int> nLO = SG.take<int>(W, 0), lo_ln = SG.take<int>(W * B.maxLO, 0), ll_tab = SG.take<int>(W * B.llSlots * 2, -1), ll_np = SG.take<int>(W, 0);
  Span<int> ps_list = SG.take<int>(W * B.maxP, 0), ps_cnt = SG.take<int>(W * (NF + 1), 0);
  Span<double> pt_obs = SG.take<double>(W * B.maxPO * 3, 0.0), ln_obs = SG.take<double>(W * B.maxLO * 8, 0.0);
  Span<DevPreint> pre = SG.take<DevPreint>(W * NF);
  Span<double> fail_ref = SG.take<double>(W * 13, 0.0), orth = SG.take<double>(W * B.maxL * 4, 0.0);
  Span<int> orth_in = SG.take<int>(W, 0);
  Span<int> pr_n = SG.take<int>(W, 0), pr_nb = SG.take<int>(W, 0), pr_kind = SG.take<int>(W * MAXPB, 0), pr_frame = SG.take<int>(W * MAXPB, 0),
            pr_idx = SG.take<int>(W * MAXPB, 0);
  Span<double> pr_x0 = SG.take<double>(W * MAXPB * 9, 0.0), pr_r0 = SG.take<double>(W * MAXPN, 0.0);
  Span<int> mg_n = SG.take<int>(W, 0), mg_nb = SG.take<int>(W, 0), mg_kind = SG.take<int>(W * MAXPB, 0), mg_frame = SG.take<int>(W * MAXPB, 0),
            mg_idx = SG.take<int>(W * MAXPB, 0), mg_cam = SG.take<int>(W * MAXPB, 0);
  if (SG.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
  // chained mode: window w takes the prior the context's previous solve left for window w (device resident); only its block
  // table comes through the host
  struct HostTab { int n = 0, nb = 0, kind[MAXPB], frame[MAXPB], idx[MAXPB]; };
  std::vector<HostTab> ctab;
  std::vector<int> keep_prior(W, 0);
  if (chained) {
    ctab.resize(W);
    std::vector<int> t_n(W), t_nb(W), t_kind(W * MAXPB), t_frame(W * MAXPB), t_idx(W * MAXPB), p_n(W), p_nb(W), p_kind(W * MAXPB),
        p_frame(W * MAXPB), p_idx(W * MAXPB);
    HIPCHK(c, hipMemcpyAsync(t_n.data(), B.mg_n, W * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(t_nb.data(), B.mg_nb, W * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(t_kind.data(), B.mg_kind, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(t_frame.data(), B.mg_frame, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(t_idx.data(), B.mg_idx, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(p_n.data(), B.pr_n, W * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(p_nb.data(), B.pr_nb, W * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(p_kind.data(), B.pr_kind, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(p_frame.data(), B.pr_frame, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(p_idx.data(), B.pr_idx, W * MAXPB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t w = 0; w < W; ++w) {
      const bool keep = c->h_passthrough[w] >= 0;      // MARGIN_SECOND_NEW left this window's prior as it was
      keep_prior[w] = keep ? 1 : 0;
      HostTab& T = ctab[w];
      T.n = keep ? p_n[w] : t_n[w];
      T.nb = keep ? p_nb[w] : t_nb[w];
      if (T.n < 0 || T.n > MAXPN || T.nb < 0 || T.nb > MAXPB) return fail(c, VPL_E_INVALID, "upload_chained: resident prior table out of range");
      for (int b = 0; b < T.nb; ++b) {
        T.kind[b] = (keep ? p_kind : t_kind)[w * MAXPB + b];
        T.frame[b] = (keep ? p_frame : t_frame)[w * MAXPB + b];
        T.idx[b] = (keep ? p_idx : t_idx)[w * MAXPB + b];
      }
    }
  }
  c->h_mg_m.assign(W, 0);
  c->maxPriorN = 0;
  c->h_passthrough.assign(W, -1);
  c->h_pass_priors.clear();
  c->any_second_new = false;
  c->h_nP.assign(W, 0);
  c->h_nL.assign(W, 0);
  c->h_lmap.resize(W);

  {   // stride of the per-window prior matrices: the largest prior of the batch
    int nmax = 1;
    for (size_t w = 0; w < W; ++w) {
      if (chained) nmax = std::max(nmax, ctab[w].n);
      else if (win[w].has_prior && win[w].prior) nmax = std::max(nmax, win[w].prior->n);
    }
    if (nmax > MAXPN) return fail(c, VPL_E_CAPACITY, "prior larger than MAXPN");
    B.prS = (nmax * nmax + 7) & ~7;
  }
  // compact W rows: stride from the longest track of the batch; zero fill only where some slot has no writer
  {
    int maxTrack = 2, minTrack = NF;
    for (size_t w = 0; w < W; ++w) {
      for (int p = 0; p < win[w].n_points; ++p) { maxTrack = std::max(maxTrack, win[w].point_nobs[p]); minTrack = std::min(minTrack, win[w].point_nobs[p]); }
      for (int l = 0; l < win[w].n_lines; ++l) { maxTrack = std::max(maxTrack, win[w].line_nobs[l]); minTrack = std::min(minTrack, win[w].line_nobs[l]); }
    }
    maxTrack = std::min(maxTrack, (int)NF);
    B.WS = 6 * maxTrack + 6;
    int maxLineTrack = 1;
    for (size_t w = 0; w < W; ++w)
      for (int l = 0; l < win[w].n_lines; ++l) maxLineTrack = std::max(maxLineTrack, std::min(win[w].line_nobs[l], (int)NF));
    B.llK = maxLineTrack;
    B.llNLW = 64 / maxLineTrack;
    B.wfill = (minTrack != maxTrack || opt->remove_line_outliers) ? 1 : 0;
  }

  long schur_wide_w = 0, schur_total_w = 0;   // matrix-core weight of the wide entries / of all entries of k_schur's table
  bool same_layout = false;
  {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    const int hdr[8] = {nW, B.WS, B.llK, B.llNLW, B.wfill, all_lines ? 1 : 0, B.maxP, B.maxL};
    mix(hdr, sizeof(hdr));
    for (size_t w = 0; w < W; ++w) {
      const vpl_window& v = win[w];
      mix(&v.n_points, 4); mix(&v.n_lines, 4);
      if (v.n_points > 0 && v.point_start && v.point_nobs) { mix(v.point_start, 4 * (size_t)v.n_points); mix(v.point_nobs, 4 * (size_t)v.n_points); }
      if (v.n_lines > 0 && v.line_start && v.line_nobs) { mix(v.line_start, 4 * (size_t)v.n_lines); mix(v.line_nobs, 4 * (size_t)v.n_lines); }
      if (v.n_lines > 0 && v.line_triangulated) mix(v.line_triangulated, 4 * (size_t)v.n_lines);
    }
    // the hash only finds the candidate; the decision is an exact comparison of the hashed integers (a collision would reuse
    // lane / unit / K-step tables of another track layout: out-of-range reads -- ADVICE r3)
    std::vector<int> key;
    key.reserve(16 + (size_t)W * 8);
    key.insert(key.end(), hdr, hdr + 8);
    for (size_t w = 0; w < W; ++w) {
      const vpl_window& v = win[w];
      key.push_back(v.n_points); key.push_back(v.n_lines);
      const bool hp = v.n_points > 0 && v.point_start && v.point_nobs, hl = v.n_lines > 0 && v.line_start && v.line_nobs;
      key.push_back((hp ? 1 : 0) | (hl ? 2 : 0) | ((v.n_lines > 0 && v.line_triangulated) ? 4 : 0));   // which arrays follow
      if (hp) { key.insert(key.end(), v.point_start, v.point_start + v.n_points); key.insert(key.end(), v.point_nobs, v.point_nobs + v.n_points); }
      if (hl) { key.insert(key.end(), v.line_start, v.line_start + v.n_lines); key.insert(key.end(), v.line_nobs, v.line_nobs + v.n_lines); }
      if (v.n_lines > 0 && v.line_triangulated) key.insert(key.end(), v.line_triangulated, v.line_triangulated + v.n_lines);
    }
    same_layout = c->layout_valid && h == c->layout_sig && key == c->layout_key && std::getenv("VPL_BA_NO_LAYOUT_CACHE") == nullptr;
    c->layout_sig = h;
    c->layout_key.swap(key);
    c->layout_valid = false;       // becomes valid when this upload has gone through
  }
  for (size_t w = 0; w < W; ++w) {
    const vpl_window& v = win[w];
    if (v.n_points > B.maxP || v.n_lines > B.maxL) return fail(c, VPL_E_CAPACITY, "too many tracks for the context");
    std::memcpy(&pose[w * 77], v.pose, 77 * 8);
    std::memcpy(&sb[w * 99], v.speed_bias, 99 * 8);
    std::memcpy(&ex[w * 7], v.ex_pose, 7 * 8);
    if (v.failure_occur) {
      fail_ref[w * 13] = 1.0;
      std::memcpy(&fail_ref[w * 13 + 1], v.last_P0, 3 * 8);
      std::memcpy(&fail_ref[w * 13 + 4], v.last_R0, 9 * 8);
    }
    nP[w] = v.n_points;
    c->h_nP[w] = v.n_points;
    int off = 0;
    for (int p = 0; p < v.n_points; ++p) {
      const int s = v.point_start[p], no = v.point_nobs[p];
      if (s < 0 || no < 2 || s + no > NF) return fail(c, VPL_E_INVALID, "point track outside the window");
      if (off + no > B.maxPO) return fail(c, VPL_E_CAPACITY, "too many point observations");
      pt_start[w * B.maxP + p] = s; pt_nobs[w * B.maxP + p] = no; pt_off[w * B.maxP + p] = off;
      std::memcpy(&pt_obs[(w * B.maxPO + off) * 3], v.point_obs + (size_t)off * 3, (size_t)no * 3 * 8);
      invd[w * B.maxP + p] = v.inv_depth[p];
      off += no;
    }
    if (!same_layout) {   // counting sort of the point tracks by start frame
      int cnt[NF + 1] = {0};
      for (int p = 0; p < v.n_points; ++p) cnt[v.point_start[p] + 1]++;
      for (int f = 0; f < NF; ++f) cnt[f + 1] += cnt[f];
      for (int f = 0; f <= NF; ++f) ps_cnt[w * (NF + 1) + f] = cnt[f];
      int pos[NF + 1];
      for (int f = 0; f <= NF; ++f) pos[f] = cnt[f];
      for (int p = 0; p < v.n_points; ++p) ps_list[w * B.maxP + pos[v.point_start[p]]++] = p;
      // inside a start frame the longer tracks come first: the tracks observed at index k are then a prefix of the list
      for (int f = 0; f < NF; ++f)
        std::stable_sort(&ps_list[w * B.maxP + cnt[f]], &ps_list[w * B.maxP + cnt[f + 1]],
                         [&](int a, int b) { return v.point_nobs[a] > v.point_nobs[b]; });
      // work units of the point phase of k_lin and their commit tickets (ba_pack.h)
      PointUnitLayout PL;
      if (!pack_point_units(v.point_nobs, &pt_off[w * B.maxP], &ps_list[w * B.maxP], cnt, B.maxPR, &pu_lane[w * B.maxPR * 1024],
                            &pu_sub[w * B.maxPR * 512], &PL))
        return fail(c, VPL_E_CAPACITY, "point work-unit table too small");
      pu_cnt[w] = PL.rounds;
      pu_cnt0[w] = PL.rounds0;
      // a pass that waited for a ticket owned by a unit it never runs would spin forever on the device: replay both chains
      if (!point_unit_chains_finish(&pu_sub[w * B.maxPR * 512], PL, false) || !point_unit_chains_finish(&pu_sub[w * B.maxPR * 512], PL, true))
        return fail(c, VPL_E_INVALID, "internal: commit-ticket order of the point work units is not executable");
    }
    off = 0;
    int woff = 0, nl = 0;
    std::vector<int>& lmap = c->h_lmap[w];
    lmap.clear();
    for (int l = 0; l < v.n_lines; ++l) {
      const int s = v.line_start[l], no = v.line_nobs[l];
      if (s < 0 || no < 1 || s + no > NF) return fail(c, VPL_E_INVALID, "line track outside the window");
      const int tri_flag = (!v.line_triangulated || v.line_triangulated[l]) ? 1 : 0;
      // lines that are not triangulated take no part in the solves (estimator.cpp:1133); they travel only for
      // vpl_ba_triangulate_lines
      if (all_lines || tri_flag) {
        if (off + no > B.maxLO) return fail(c, VPL_E_CAPACITY, "too many line observations");
        const size_t dl = w * B.maxL + nl;
        ln_start[dl] = s; ln_nobs[dl] = no; ln_off[dl] = off; ln_tri[dl] = tri_flag;
        std::memcpy(&ln_obs[(w * B.maxLO + off) * 8], v.line_obs + (size_t)woff * 8, (size_t)no * 8 * 8);
        if (v.line_orth) std::memcpy(&orth[dl * 4], v.line_orth + (size_t)l * 4, 4 * 8);
        else std::memcpy(&plk[dl * 6], v.line_plk + (size_t)l * 6, 6 * 8);
        for (int k = 0; k < no; ++k) lo_ln[w * B.maxLO + off + k] = nl;
        off += no;
        lmap.push_back(l);
        ++nl;
      }
      woff += no;
    }
    nL[w] = nl;
    if (!same_layout) {   // K-steps of k_schur: landmark rows by start frame, dealt to the waves, flush tickets (ba_pack.h)
      int cnt[NF + 1];
      for (int f = 0; f <= NF; ++f) cnt[f] = ps_cnt[w * (NF + 1) + f];
      // rows wider than the 6-frame view of k_schur<3>: entries with a longer track are flagged wide (k_schur_mixed)
      if (pack_schur_ksteps(v.n_points, &ps_list[w * B.maxP], cnt, nl, &ln_start[w * B.maxL], B.maxKS, &sk_tab[w * B.maxKS * 4],
                            &sk_wave[w * 8 * SK_WSTRIDE], SCHUR_THREADS / 64, v.point_nobs, &ln_nobs[w * B.maxL],
                            B.WS + 2 > 48 ? SCHUR_NARROW_FRAMES : 0, &schur_wide_w, &schur_total_w) < 0)
        return fail(c, VPL_E_CAPACITY, "K-step table of the landmark elimination too small");
    }
    if (!same_layout) {   // lane layout of the line phase: llNLW whole tracks per wave, k-major; tracks in the caller's order (tracks that start
        // in the same frame side by side would pile their LDS adds onto the same addresses)
      std::vector<int> ord(nl);
      for (int i = 0; i < nl; ++i) ord[i] = i;
      const int NLW = B.llNLW, perPass = 8 * NLW;
      ll_np[w] = (nl + perPass - 1) / perPass;
      if (ll_np[w] * 512 > B.llSlots) return fail(c, VPL_E_CAPACITY, "line layout table too small");
      for (int q = 0; q < nl; ++q) {
        const int dl = ord[q], pass = q / perPass, wave = (q % perPass) / NLW, i = q % NLW;
        const int no = std::min(ln_nobs[w * B.maxL + dl], (int)NF);
        for (int k = 0; k < no; ++k) {      // (observation, line | k << 16 | start frame << 20): everything a lane's loads need
          int* e = &ll_tab[2 * ((size_t)w * B.llSlots + pass * 512 + wave * 64 + k * NLW + i)];
          e[0] = ln_off[w * B.maxL + dl] + k;
          e[1] = dl | k << 16 | ln_start[w * B.maxL + dl] << 20;
        }
      }
    }
    orth_in[w] = v.line_orth ? 1 : 0;
    c->h_nL[w] = nl;
    nLO[w] = off;
    for (int j = 0; j < NF; ++j) to_dev_preint(v.preint[j], pre[w * NF + j]);
    if (chained) {
      const HostTab& T = ctab[w];
      pr_n[w] = T.n; pr_nb[w] = T.nb;
      c->maxPriorN = std::max(c->maxPriorN, T.n);
      for (int b = 0; b < T.nb; ++b) {
        pr_kind[w * MAXPB + b] = T.kind[b]; pr_frame[w * MAXPB + b] = T.frame[b]; pr_idx[w * MAXPB + b] = T.idx[b];
        if (T.kind[b] == 1 && T.frame[b] != 0) path[w] = 1;
      }
    } else if (v.has_prior && v.prior) {
      const vpl_prior& pr = *v.prior;
      if (pr.n < 0 || pr.n > MAXPN || pr.n_blocks < 0 || pr.n_blocks > MAXPB) return fail(c, VPL_E_INVALID, "bad prior");
      pr_n[w] = pr.n; pr_nb[w] = pr.n_blocks;
      // k_chol's elimination order keeps speed/bias 0 in its dense part; a prior that ties another frame's speed/bias block
      // (never produced by the reference's marginalisation) takes the general path
      for (int b = 0; b < pr.n_blocks; ++b)
        if (pr.block_kind[b] == 1 && pr.block_frame[b] != 0) path[w] = 1;
      c->maxPriorN = std::max(c->maxPriorN, pr.n);
      for (int b = 0; b < pr.n_blocks; ++b) {
        pr_kind[w * MAXPB + b] = pr.block_kind[b];
        pr_frame[w * MAXPB + b] = pr.block_frame[b];
        pr_idx[w * MAXPB + b] = pr.block_idx[b];
        std::memcpy(&pr_x0[(w * MAXPB + b) * 9], pr.x0[b], 9 * 8);
        if (pr.block_kind[b] < 0 || pr.block_kind[b] > 2 || pr.block_frame[b] < 0 || pr.block_frame[b] >= NF)
          return fail(c, VPL_E_INVALID, "bad prior block");
      }
      std::memcpy(&pr_r0[w * MAXPN], pr.r0, (size_t)pr.n * 8);
      if (pr.n > 0) {
        Span<double> j0 = SG.take<double>((size_t)pr.n * pr.n);
        if (SG.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
        std::memcpy(j0.p, pr.J0, (size_t)pr.n * pr.n * 8);
        SG.to_device(B.pr_J0 + w * (size_t)B.prS, j0.p, j0.n);
      }
    }
    // kept blocks of the next prior in the canonical (address) order of the reference's para_* layout
    {
      int* kd = &mg_kind[w * MAXPB]; int* fr = &mg_frame[w * MAXPB]; int* ix = &mg_idx[w * MAXPB]; int* cm = &mg_cam[w * MAXPB];
      const int pnb = chained ? ctab[w].nb : ((v.has_prior && v.prior) ? v.prior->n_blocks : 0);
      const int* pk = pnb ? (chained ? ctab[w].kind : v.prior->block_kind) : nullptr;
      const int* pf = pnb ? (chained ? ctab[w].frame : v.prior->block_frame) : nullptr;
      c->h_passthrough[w] = -1;
      if (opt->marginalization_flag == VPL_MARGIN_OLD) {
        KeepSrc S;
        S.nP = v.n_points; S.pt_start = v.point_start; S.pt_nobs = v.point_nobs;
        S.nL = nL[w]; S.ln_start = &ln_start[w * B.maxL]; S.ln_nobs = &ln_nobs[w * B.maxL]; S.ln_removed = nullptr;
        S.pr_nb = pnb; S.pr_kind = pk; S.pr_frame = pf;
        S.imu01 = v.preint[1].sum_dt < 10.0;
        int mm = 0;
        keep_tables_old(S, kd, fr, ix, cm, &mg_n[w], &mg_nb[w], &mm);
        c->h_mg_m[w] = mm;
      } else if (opt->marginalization_flag == VPL_MARGIN_SECOND_NEW) {
        const int rc = pnb ? keep_tables_second_new(pnb, pk, pf, kd, fr, ix, cm, &mg_n[w], &mg_nb[w]) : 0;
        if (rc < 0) return fail(c, VPL_E_INVALID, "MARGIN_SECOND_NEW: the prior holds the speed/bias of frame WINDOW_SIZE-1");
        if (rc == 1) { c->h_mg_m[w] = 6; c->any_second_new = true; }
        else if (pnb) {   // the reference leaves last_marginalization_info as it is (estimator.cpp:1385)
          c->h_passthrough[w] = (int)c->h_pass_priors.size();
          if (chained) {    // the prior lives on the device only: the pass-through entry carries its table (download refetches the rest)
            vpl_prior hp;
            std::memset(&hp, 0, sizeof(int) * (2 + 3 * VPL_MAX_PRIOR_BLOCKS));
            hp.n = -1;      // marker: not available on the host
            c->h_pass_priors.push_back(hp);
          } else {
            c->h_pass_priors.push_back(*v.prior);
          }
        }
      }
      if (mg_n[w] > MAXKEEP) return fail(c, VPL_E_CAPACITY, "marginalisation keeps more than MAXKEEP dims");
    }
  }
  {
    int nmax = 0;
    for (size_t w = 0; w < W; ++w) nmax = std::max(nmax, mg_n[w]);
    // kept blocks of up to 48 dims (the one-wave factorisations) whose workspace fits 79 KB: 256 threads, two work-groups per CU
    c->marg_small = nmax <= 48 && B.maxP <= 256 && B.maxL <= 256 &&
                    (size_t)marg_layout(nmax, true).total * sizeof(double) <= MARG_LDS_SMALL && std::getenv("VPL_BA_MARG_BIG") == nullptr;
    c->marg_smem = (size_t)marg_layout(nmax, c->marg_small).total * sizeof(double);
    if (c->marg_smem > 159 * 1024) return fail(c, VPL_E_CAPACITY, "marginalisation workspace exceeds LDS");
  }
  HIPCHK(c, up(c, B.pose, pose)); HIPCHK(c, up(c, B.sb, sb)); HIPCHK(c, up(c, B.ex, ex)); HIPCHK(c, up(c, B.invd, invd));
  HIPCHK(c, up(c, B.plk, plk)); HIPCHK(c, up(c, B.fail_ref, fail_ref)); HIPCHK(c, up(c, B.orth_in, orth_in));
  HIPCHK(c, up(c, B.orth, orth));
  HIPCHK(c, up(c, B.pose_0, pose)); HIPCHK(c, up(c, B.sb_0, sb)); HIPCHK(c, up(c, B.ex_0, ex)); HIPCHK(c, up(c, B.invd_0, invd));
  HIPCHK(c, up(c, B.plk_0, plk));
  HIPCHK(c, up(c, B.pt_obs, pt_obs));
  if (!same_layout) {
  HIPCHK(c, up(c, B.nP, nP)); HIPCHK(c, up(c, B.nL, nL));
  HIPCHK(c, up(c, B.pt_start, pt_start)); HIPCHK(c, up(c, B.pt_nobs, pt_nobs)); HIPCHK(c, up(c, B.pt_off, pt_off));
  HIPCHK(c, up(c, B.ps_list, ps_list)); HIPCHK(c, up(c, B.ps_cnt, ps_cnt));
  {   // the first max-over-the-batch rounds of every window, packed into the arena
    int rmax = 0;
    for (size_t q = 0; q < W; ++q) rmax = std::max(rmax, pu_cnt[q]);
    if (rmax > 0) {
      Span<int> pl = SG.take<int>(W * (size_t)rmax * 1024), psb = SG.take<int>(W * (size_t)rmax * 512);
      if (SG.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
      for (size_t q = 0; q < W; ++q) {   // rounds a window does not use travel too: idle lanes, empty slots
        int* l = pl.p + q * (size_t)rmax * 1024;
        int* u = psb.p + q * (size_t)rmax * 512;
        std::memcpy(l, &pu_lane[q * B.maxPR * 1024], (size_t)pu_cnt[q] * 4096);
        std::memcpy(u, &pu_sub[q * B.maxPR * 512], (size_t)pu_cnt[q] * 2048);
        std::fill(l + (size_t)pu_cnt[q] * 1024, l + (size_t)rmax * 1024, -1);
        std::fill(u + (size_t)pu_cnt[q] * 512, u + (size_t)rmax * 512, 0);
        SG.to_device(B.pu_lane + q * (size_t)B.maxPR * 1024, l, (size_t)rmax * 1024);
        SG.to_device(B.pu_sub + q * (size_t)B.maxPR * 512, u, (size_t)rmax * 512);
      }
    }
  } HIPCHK(c, up(c, B.pu_cnt, pu_cnt)); HIPCHK(c, up(c, B.pu_cnt0, pu_cnt0));
  HIPCHK(c, up(c, B.ln_start, ln_start)); HIPCHK(c, up(c, B.ln_nobs, ln_nobs)); HIPCHK(c, up(c, B.ln_off, ln_off));
  HIPCHK(c, up(c, B.nLO, nLO)); HIPCHK(c, up(c, B.lo_ln, lo_ln));
  HIPCHK(c, up(c, B.ll_tab, ll_tab)); HIPCHK(c, up(c, B.ll_np, ll_np));
  HIPCHK(c, up(c, B.sk_tab, sk_tab)); HIPCHK(c, up(c, B.sk_wave, sk_wave));
  }
  HIPCHK(c, up(c, B.ln_obs, ln_obs)); HIPCHK(c, up(c, B.ln_tri, ln_tri));
  HIPCHK(c, up(c, B.pre, pre));
  HIPCHK(c, up(c, B.pr_n, pr_n)); HIPCHK(c, up(c, B.pr_nb, pr_nb)); HIPCHK(c, up(c, B.pr_kind, pr_kind));
  HIPCHK(c, up(c, B.pr_frame, pr_frame)); HIPCHK(c, up(c, B.pr_idx, pr_idx));
  if (!chained) { HIPCHK(c, up(c, B.pr_x0, pr_x0)); HIPCHK(c, up(c, B.pr_r0, pr_r0)); }
  else {
    // values of the priors: device to device, before the scatter below overwrites mg_* with the next solve's tables
    DevBuf dkeep;
    HIPCHK(c, dkeep.alloc(W * 4));
    HIPCHK(c, hipMemcpyAsync(dkeep.p, keep_prior.data(), W * 4, hipMemcpyHostToDevice, c->stream));
    if (B.prS != prev_prS && std::any_of(keep_prior.begin(), keep_prior.end(), [](int k) { return k != 0; })) {
      // windows that keep their prior (MARGIN_SECOND_NEW pass-through) have J0 at the PREVIOUS batch stride: move them through
      // a scratch copy (old and new locations of different windows overlap) -- ADVICE r3
      DevBuf tmp;
      HIPCHK(c, tmp.alloc(W * (size_t)MAXPN * MAXPN * 8));
      hipLaunchKernelGGL(k_prior_restride, dim3(nW), dim3(256), 0, c->stream, B, (const int*)dkeep.p, prev_prS, tmp.d(), 0);
      hipLaunchKernelGGL(k_prior_restride, dim3(nW), dim3(256), 0, c->stream, B, (const int*)dkeep.p, prev_prS, tmp.d(), 1);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    hipLaunchKernelGGL(k_prior_handoff, dim3(nW), dim3(256), 0, c->stream, B, (const int*)dkeep.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, up(c, B.mg_n, mg_n)); HIPCHK(c, up(c, B.mg_nb, mg_nb)); HIPCHK(c, up(c, B.mg_kind, mg_kind));
  HIPCHK(c, up(c, B.mg_frame, mg_frame)); HIPCHK(c, up(c, B.mg_idx, mg_idx)); HIPCHK(c, up(c, B.mg_cam, mg_cam));
  {
    Span<int> mm = SG.take<int>(W);
    if (SG.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
    std::memcpy(mm.p, c->h_mg_m.data(), W * 4);
    HIPCHK(c, up(c, B.mg_m, mm));
  }
  c->h_mg_n.assign(mg_n.begin(), mg_n.end());
  // k_schur_mixed pays five passes over the rows of a wide entry: measured (tools/bench_tracks.py, 512 windows) 0.52 ms per step
  // with a tenth of the point tracks long, 3.2 ms with every track long, against 1.5 - 1.65 ms of k_schur<5> either way
  if (!same_layout) c->schur_mostly_wide = schur_total_w > 0 && 20 * schur_wide_w > 7 * schur_total_w;
  if (c->force_general) std::fill(path.begin(), path.end(), 1);
  HIPCHK(c, up(c, B.path, path));
  // ONE host-to-device copy of the arena, ONE kernel that scatters its pieces into the batch's arrays
  HIPCHK(c, stage_run(c, true));
  if (c->leg_timing) HIPCHK(c, hipEventRecord(c->leg_ev[1], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the arena is free for the next call from here on
  c->layout_valid = true;
  return VPL_OK;
}

int vpl_ba_reset_state(vpl_ctx* c) {
  if (c) { const int rs = settle(c); if (rs) return rs; }
  if (!c || c->nW < 1) return VPL_E_INVALID;
  DevBatch& B = c->B;
  const size_t W = c->nW;
  HIPCHK(c, hipMemcpyAsync(B.pose, B.pose_0, W * 77 * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(B.sb, B.sb_0, W * 99 * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(B.ex, B.ex_0, W * 7 * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(B.invd, B.invd_0, W * B.maxP * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(B.plk, B.plk_0, W * B.maxL * 6 * 8, hipMemcpyDeviceToDevice, c->stream));
  return VPL_OK;
}

int vpl_ba_upload(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt) {
  return upload_impl(c, nW, win, opt, false);
}
int vpl_ba_upload_chained(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt) {
  return upload_impl(c, nW, win, opt, false, true);
}

// FeatureManager::triangulateLine for a batch: upload (every line), k_triangulate, flags + Pluecker vectors back
static int triangulate_lines_impl(vpl_ctx* c, int nW, vpl_window* win, bool async) {
  if (!c || !win || nW < 1) return VPL_E_INVALID;
  for (int w = 0; w < nW; ++w)
    if (win[w].n_lines > 0 && !win[w].line_triangulated) return fail(c, VPL_E_INVALID, "line_triangulated is required");
  vpl_ba_options opt;
  vpl_ba_default_options(&opt);
  opt.marginalization_flag = VPL_MARGIN_NONE;
  int rc = upload_impl(c, nW, win, &opt, true);
  if (rc) return rc;
  DevBatch& B = c->B;
  hipStream_t s = c->stream;
  { KTimer t(c, "k_triangulate"); hipLaunchKernelGGL(k_triangulate, dim3(nW), dim3(128), 0, s, B); }
  HIPCHK(c, hipGetLastError());
  const size_t W = nW;
  auto plk = std::make_shared<std::vector<double>>(W * B.maxL * 6);
  auto tri = std::make_shared<std::vector<int>>(W * B.maxL);
  HIPCHK(c, hipMemcpyAsync(plk->data(), B.plk, plk->size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(tri->data(), B.ln_tri, tri->size() * 4, hipMemcpyDeviceToHost, s));
  const int maxL = B.maxL;
  return finish_or_defer(c, async, [c, win, W, plk, tri, maxL, s]() -> int {
    HIPCHK(c, hipStreamSynchronize(s));
    for (size_t w = 0; w < W; ++w) {
      vpl_window& v = win[w];
      const std::vector<int>& lmap = c->h_lmap[w];
      for (size_t dl = 0; dl < lmap.size(); ++dl) {
        const int l = lmap[dl];
        if (!v.line_triangulated[l] && (*tri)[w * maxL + dl]) {
          std::memcpy(v.line_plk + (size_t)l * 6, &(*plk)[(w * maxL + dl) * 6], 6 * 8);
          v.line_triangulated[l] = 1;
        }
      }
    }
    return VPL_OK;
  });
}
int vpl_ba_triangulate_lines(vpl_ctx* c, int nW, vpl_window* win) { return triangulate_lines_impl(c, nW, win, false); }
int vpl_ba_triangulate_lines_async(vpl_ctx* c, int nW, vpl_window* win) { return triangulate_lines_impl(c, nW, win, true); }

// Estimator::slideWindow for a batch.  The list bookkeeping (start frames, dropped observations, erased tracks) is integer
// work on the caller's arrays and is done here on the host; the re-anchoring arithmetic of removeBackShiftDepth runs in
// k_slide_shift over the gathered start-frame-0 survivors of all windows.
static int slide_window_impl(vpl_ctx* c, int nW, vpl_window* win, int flag, double init_depth, vpl_slide_tracks* out, bool async) {
  if (!c || !win || !out || nW < 1 || !(init_depth > 0.0)) return VPL_E_INVALID;
  if (flag != VPL_MARGIN_OLD && flag != VPL_MARGIN_SECOND_NEW) return fail(c, VPL_E_INVALID, "slide_window: marginalization_flag");
  constexpr int WS = VPL_NFRAMES - 1;
  for (int w = 0; w < nW; ++w) {
    const vpl_window& W = win[w];
    const vpl_slide_tracks& O = out[w];
    if ((W.n_points && (!O.point_start || !O.point_nobs || !O.point_drop || !W.point_start || !W.point_nobs || !W.point_obs || !W.inv_depth)) ||
        (W.n_lines && (!O.line_start || !O.line_nobs || !O.line_drop || !W.line_start || !W.line_nobs || !W.line_plk)))
      return fail(c, VPL_E_INVALID, "slide_window: null track array");
    for (int i = 0; i < W.n_points; ++i)
      if (W.point_nobs[i] < 1 || W.point_start[i] < 0 || W.point_start[i] + W.point_nobs[i] > VPL_NFRAMES)
        return fail(c, VPL_E_INVALID, "slide_window: point track outside the window");
    for (int i = 0; i < W.n_lines; ++i)
      if (W.line_nobs[i] < 1 || W.line_start[i] < 0 || W.line_start[i] + W.line_nobs[i] > VPL_NFRAMES)
        return fail(c, VPL_E_INVALID, "slide_window: line track outside the window");
  }
  HIPCHK(c, hipSetDevice(c->device));
  { const int rs = settle(c); if (rs) return rs; }
  if (flag == VPL_MARGIN_SECOND_NEW) {
    // removeFront(frame_count = WINDOW_SIZE): no arithmetic
    auto front = [](int n, const int* start, const int* nobs, int* ostart, int* onobs, int* odrop) {
      for (int i = 0; i < n; ++i) {
        ostart[i] = start[i]; onobs[i] = nobs[i]; odrop[i] = -1;
        if (start[i] == WS) { ostart[i] = WS - 1; continue; }
        if (start[i] + nobs[i] - 1 < WS - 1) continue;
        odrop[i] = WS - 1 - start[i];
        onobs[i] = nobs[i] - 1;
      }
    };
    for (int w = 0; w < nW; ++w) {
      front(win[w].n_points, win[w].point_start, win[w].point_nobs, out[w].point_start, out[w].point_nobs, out[w].point_drop);
      front(win[w].n_lines, win[w].line_start, win[w].line_nobs, out[w].line_start, out[w].line_nobs, out[w].line_drop);
      std::memcpy(win[w].pose[WS - 1], win[w].pose[WS], sizeof(win[w].pose[0]));
      std::memcpy(win[w].speed_bias[WS - 1], win[w].speed_bias[WS], sizeof(win[w].speed_bias[0]));
    }
    return VPL_OK;
  }
  struct SlideJob {
    std::vector<double> fr, pd, ld;
    std::vector<int> pw, lw, pidx, lidx;
    DevBuf dfr, dpw, dpd, dlw, dld;
  };
  auto job = std::make_shared<SlideJob>();
  std::vector<double>&fr = job->fr, &pd = job->pd, &ld = job->ld;
  std::vector<int>&pw = job->pw, &lw = job->lw, &pidx = job->pidx, &lidx = job->lidx;
  fr.resize((size_t)nW * 21);
  for (int w = 0; w < nW; ++w) {
    const vpl_window& W = win[w];
    std::memcpy(&fr[(size_t)w * 21], W.pose[0], 56);
    std::memcpy(&fr[(size_t)w * 21 + 7], W.pose[1], 56);
    std::memcpy(&fr[(size_t)w * 21 + 14], W.ex_pose, 56);
    size_t off = 0;
    for (int i = 0; i < W.n_points; off += W.point_nobs[i], ++i) {
      out[w].point_drop[i] = -1;
      if (W.point_start[i] != 0) { out[w].point_start[i] = W.point_start[i] - 1; out[w].point_nobs[i] = W.point_nobs[i]; continue; }
      out[w].point_drop[i] = 0;
      out[w].point_start[i] = 0;
      out[w].point_nobs[i] = W.point_nobs[i] - 1 < 2 ? 0 : W.point_nobs[i] - 1;
      if (!out[w].point_nobs[i]) continue;
      pw.push_back(w); pidx.push_back(i);
      pd.insert(pd.end(), {W.point_obs[3 * off], W.point_obs[3 * off + 1], W.point_obs[3 * off + 2], W.inv_depth[i]});
    }
    for (int i = 0; i < W.n_lines; ++i) {
      out[w].line_drop[i] = -1;
      if (W.line_start[i] != 0) { out[w].line_start[i] = W.line_start[i] - 1; out[w].line_nobs[i] = W.line_nobs[i]; continue; }
      out[w].line_drop[i] = 0;
      out[w].line_start[i] = 0;
      out[w].line_nobs[i] = W.line_nobs[i] - 1 < 2 ? 0 : W.line_nobs[i] - 1;
      if (!out[w].line_nobs[i]) continue;
      lw.push_back(w); lidx.push_back(i);
      ld.insert(ld.end(), W.line_plk + 6 * i, W.line_plk + 6 * i + 6);
    }
  }
  const int nPts = (int)pw.size(), nLns = (int)lw.size();
  hipStream_t s = c->stream;
  if (nPts + nLns > 0) {
    DevBuf &dfr = job->dfr, &dpw = job->dpw, &dpd = job->dpd, &dlw = job->dlw, &dld = job->dld;
    HIPCHK(c, dfr.alloc(fr.size() * 8)); HIPCHK(c, dpw.alloc(pw.size() * 4)); HIPCHK(c, dpd.alloc(pd.size() * 8));
    HIPCHK(c, dlw.alloc(lw.size() * 4)); HIPCHK(c, dld.alloc(ld.size() * 8));
    HIPCHK(c, hipMemcpyAsync(dfr.p, fr.data(), fr.size() * 8, hipMemcpyHostToDevice, s));
    if (nPts) { HIPCHK(c, hipMemcpyAsync(dpw.p, pw.data(), pw.size() * 4, hipMemcpyHostToDevice, s));
                HIPCHK(c, hipMemcpyAsync(dpd.p, pd.data(), pd.size() * 8, hipMemcpyHostToDevice, s)); }
    if (nLns) { HIPCHK(c, hipMemcpyAsync(dlw.p, lw.data(), lw.size() * 4, hipMemcpyHostToDevice, s));
                HIPCHK(c, hipMemcpyAsync(dld.p, ld.data(), ld.size() * 8, hipMemcpyHostToDevice, s)); }
    { KTimer t(c, "k_slide_shift");
      hipLaunchKernelGGL(k_slide_shift, dim3((nPts + nLns + 255) / 256), dim3(256), 0, s, dfr.d(), nPts, (const int*)dpw.p, dpd.d(),
                         nLns, (const int*)dlw.p, dld.d(), init_depth); }
    HIPCHK(c, hipGetLastError());
    if (nPts) HIPCHK(c, hipMemcpyAsync(pd.data(), dpd.p, pd.size() * 8, hipMemcpyDeviceToHost, s));
    if (nLns) HIPCHK(c, hipMemcpyAsync(ld.data(), dld.p, ld.size() * 8, hipMemcpyDeviceToHost, s));
  }
  // (the job owns the staging vectors and the device buffers until the results have been scattered)
  return finish_or_defer(c, async, [c, win, nW, job, nPts, nLns, s]() -> int {
    if (nPts + nLns > 0) {
      HIPCHK(c, hipStreamSynchronize(s));
      for (int k = 0; k < nPts; ++k) win[job->pw[k]].inv_depth[job->pidx[k]] = job->pd[(size_t)k * 4 + 3];
      for (int k = 0; k < nLns; ++k) std::memcpy(win[job->lw[k]].line_plk + 6 * job->lidx[k], &job->ld[(size_t)k * 6], 48);
    }
    for (int w = 0; w < nW; ++w) {
      std::memmove(win[w].pose[0], win[w].pose[1], sizeof(win[w].pose[0]) * WS);            // frames 1..10 -> 0..9, 10 stays
      std::memmove(win[w].speed_bias[0], win[w].speed_bias[1], sizeof(win[w].speed_bias[0]) * WS);
    }
    return VPL_OK;
  });
}
int vpl_ba_slide_window(vpl_ctx* c, int nW, vpl_window* win, int flag, double init_depth, vpl_slide_tracks* out) {
  return slide_window_impl(c, nW, win, flag, init_depth, out, false);
}
int vpl_ba_slide_window_async(vpl_ctx* c, int nW, vpl_window* win, int flag, double init_depth, vpl_slide_tracks* out) {
  return slide_window_impl(c, nW, win, flag, init_depth, out, true);
}

// FeatureManager::triangulate for a batch: upload, k_triangulate_points, inverse depths back
static int triangulate_points_impl(vpl_ctx* c, int nW, vpl_window* win, double init_depth, bool async) {
  if (!c || !win || nW < 1 || !(init_depth > 0.0)) return VPL_E_INVALID;
  vpl_ba_options opt;
  vpl_ba_default_options(&opt);
  opt.marginalization_flag = VPL_MARGIN_NONE;
  int rc = upload_impl(c, nW, win, &opt, true);
  if (rc) return rc;
  DevBatch& B = c->B;
  hipStream_t s = c->stream;
  { KTimer t(c, "k_triangulate_points"); hipLaunchKernelGGL(k_triangulate_points, dim3(nW), dim3(128), 0, s, B, init_depth); }
  HIPCHK(c, hipGetLastError());
  const size_t W = nW;
  auto invd = std::make_shared<std::vector<double>>(W * B.maxP);
  HIPCHK(c, hipMemcpyAsync(invd->data(), B.invd, invd->size() * 8, hipMemcpyDeviceToHost, s));
  const int maxP = B.maxP;
  return finish_or_defer(c, async, [c, win, W, invd, maxP, s]() -> int {
    HIPCHK(c, hipStreamSynchronize(s));
    for (size_t w = 0; w < W; ++w)
      for (int p = 0; p < win[w].n_points; ++p) win[w].inv_depth[p] = (*invd)[w * maxP + p];
    return VPL_OK;
  });
}
int vpl_ba_triangulate_points(vpl_ctx* c, int nW, vpl_window* win, double init_depth) { return triangulate_points_impl(c, nW, win, init_depth, false); }
int vpl_ba_triangulate_points_async(vpl_ctx* c, int nW, vpl_window* win, double init_depth) { return triangulate_points_impl(c, nW, win, init_depth, true); }

// Estimator::onlyLineOpt for a batch: upload (triangulated lines), k_prep (world orth of the lines), k_line_opt (the LM
// loop), k_gauge (setLineOrth + removeLineOutlier; the gauge transform is the identity, the poses did not move)
// solve_opt != nullptr (vpl_ba_solve_odometry): the upload is made with the options of the solve that FOLLOWS, so that the
// batch -- layout tables, observations, prior, kept-block tables -- can stay where it is for that solve when onlyLineOpt
// erases no line; the line kernels run on a copy of the batch descriptor with onlyLineOpt's own flags.
static int only_line_opt_impl(vpl_ctx* c, int nW, vpl_window* win, const vpl_ba_options* opt_in, vpl_solve_report* reports, bool async,
                              const vpl_ba_options* solve_opt = nullptr) {
  if (!c || !win || !opt_in || nW < 1) return VPL_E_INVALID;
  if (c->maxL > LOPT_THREADS) return fail(c, VPL_E_CAPACITY, "onlyLineOpt handles at most 256 lines per window");
  vpl_ba_options opt = *opt_in;
  opt.marginalization_flag = VPL_MARGIN_NONE;
  opt.remove_line_outliers = 1;          // f_manager.removeLineOutlier at the end of onlyLineOpt (estimator.cpp:1037)
  int rc = upload_impl(c, nW, win, solve_opt ? solve_opt : &opt, false);
  if (rc) return rc;
  DevBatch B = c->B;
  B.opt.marginalization_flag = opt.marginalization_flag;
  B.opt.remove_line_outliers = opt.remove_line_outliers;
  const dim3 grid(nW);
  hipStream_t s = c->stream;
  { KTimer t(c, "k_prep"); hipLaunchKernelGGL(k_prep, grid, dim3(PREP_THREADS), prep_smem(c->maxPriorN), s, B, std::min(c->maxPriorN, PREP_NMAX)); }
  {
    KTimer t(c, "k_line_opt");
    if (4 * c->maxL <= LOPT_THREADS) hipLaunchKernelGGL(k_line_opt<4>, grid, dim3(LOPT_THREADS), 0, s, B);
    else if (2 * c->maxL <= LOPT_THREADS) hipLaunchKernelGGL(k_line_opt<2>, grid, dim3(LOPT_THREADS), 0, s, B);
    else hipLaunchKernelGGL(k_line_opt<1>, grid, dim3(LOPT_THREADS), 0, s, B);
  }
  { KTimer t(c, "k_gauge"); hipLaunchKernelGGL(k_gauge, grid, dim3(128), 0, s, B); }
  HIPCHK(c, hipGetLastError());
  const size_t W = nW;
  auto plk = std::make_shared<std::vector<double>>(W * B.maxL * 6);
  auto removed = std::make_shared<std::vector<int>>(W * B.maxL);
  auto tr = std::make_shared<std::vector<TrState>>(W);
  HIPCHK(c, hipMemcpyAsync(plk->data(), B.plk, plk->size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(removed->data(), B.ln_removed, removed->size() * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(tr->data(), B.tr, W * sizeof(TrState), hipMemcpyDeviceToHost, s));
  const int maxL = B.maxL;
  return finish_or_defer(c, async, [c, win, reports, W, plk, removed, tr, maxL, s]() -> int {
    HIPCHK(c, hipStreamSynchronize(s));
    for (size_t w = 0; w < W; ++w) {
      vpl_window& v = win[w];
      const std::vector<int>& lmap = c->h_lmap[w];
      if (v.line_removed)
        for (int l = 0; l < v.n_lines; ++l) v.line_removed[l] = 0;
      if (reports) std::memset(&reports[w], 0, sizeof(reports[w]));
      if (lmap.size() < 4) continue;       // "if (feature_index < 3) return;" -- nothing is touched
      for (size_t dl = 0; dl < lmap.size(); ++dl) {
        const bool rem = (*removed)[w * maxL + dl] != 0;
        if (!rem) std::memcpy(v.line_plk + (size_t)lmap[dl] * 6, &(*plk)[(w * maxL + dl) * 6], 6 * 8);
        if (v.line_removed) v.line_removed[lmap[dl]] = rem ? 1 : 0;
        if (reports) reports[w].n_lines_removed += rem ? 1 : 0;
      }
      if (reports) {
        vpl_solve_report& r = reports[w];
        const TrState& t = (*tr)[w];
        r.iterations = t.iter;
        r.num_successful_steps = t.num_successful;
        r.termination = t.status == 1 ? 1 : t.status == 2 ? 2 : 0;
        r.initial_cost = t.initial_cost;
        r.final_cost = t.x_cost;
      }
    }
    return VPL_OK;
  });
}
int vpl_ba_only_line_opt(vpl_ctx* c, int nW, vpl_window* win, const vpl_ba_options* opt, vpl_solve_report* reports) {
  return only_line_opt_impl(c, nW, win, opt, reports, false);
}
int vpl_ba_only_line_opt_async(vpl_ctx* c, int nW, vpl_window* win, const vpl_ba_options* opt, vpl_solve_report* reports) {
  return only_line_opt_impl(c, nW, win, opt, reports, true);
}

// The whole solve of the windows [w0, w0 + nw) on stream s
static void launch_solve(vpl_ctx* c, int w0, int nw, hipStream_t s) {
  DevBatch B = c->B;
  B.ord_it = 0;
  // with kernel timing on, every launch also counts the windows that did work in it (vpl_ba_launch_profile)
  B.act = c->timing ? c->d_act : nullptr;
  B.launch = 0;
  if (c->timing) { c->ltimes.clear(); hipMemsetAsync(c->d_act, 0, sizeof(int) * ACT_SLOTS * 4, s); }
  const dim3 grid(nw);
  { KTimer t(c, "k_prep"); hipLaunchKernelGGL(k_prep, grid, dim3(PREP_THREADS), prep_smem(c->maxPriorN), s, B, std::min(c->maxPriorN, PREP_NMAX)); }
  ++B.launch;
  { KTimer t(c, "k_lin"); hipLaunchKernelGGL(k_lin2, dim3(16 * ((nw + 7) / 8)), dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B); }
  ++B.launch;
  for (int it = 0; it < c->opt.num_iterations; ++it) {
    B.ord_it = it;      // k_solve / k_cost of iteration `it` walk order[it & 1]; k_cost fills order[(it + 1) & 1]
    // the step: landmark elimination -> reduced camera system -> (general path: windows flagged in B.path only) -> landmark
    // back-substitution + dogleg + candidate.
    { KTimer t(c, "k_schur");
      if (B.WS + 2 <= 48) hipLaunchKernelGGL(k_schur<3>, grid, dim3(SCHUR_THREADS), schur_smem(B.maxP, B.maxL, 3), s, B);
      else if (c->schur_wide_all || (c->schur_mostly_wide && !c->schur_never_wide)) hipLaunchKernelGGL(k_schur<5>, grid, dim3(SCHUR_THREADS), schur_smem(B.maxP, B.maxL, 5), s, B);
      // rows wider than 6 frames: narrow view for the entries of short tracks, all tiles for the flagged ones (round 4)
      else hipLaunchKernelGGL(k_schur_mixed, grid, dim3(SCHUR_THREADS), schur_smem(B.maxP, B.maxL, 3), s, B); }
    ++B.launch;
    { KTimer t(c, "k_chol"); hipLaunchKernelGGL(k_chol, grid, dim3(CHOL_THREADS), CHOL_SMEM, s, B); }
    ++B.launch;
    { KTimer t(c, "k_solve"); hipLaunchKernelGGL(k_solve, grid, dim3(SOLVE_THREADS), solve_smem(B.maxP, B.maxL), s, B); }
    ++B.launch;
    { KTimer t(c, "k_back"); hipLaunchKernelGGL(k_back, grid, dim3(BACK_THREADS), back_smem(B.maxP, B.maxL), s, B); }
    ++B.launch;
    { KTimer t(c, "k_cost"); hipLaunchKernelGGL(k_cost, grid, dim3(COST_THREADS), 0, s, B); }
    ++B.launch;
    if (it + 1 < c->opt.num_iterations) {
      B.ord_it = it + 1;
      KTimer t(c, "k_lin");
      hipLaunchKernelGGL(k_lin2, dim3(16 * ((nw + 7) / 8)), dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B);
    }
    if (it + 1 < c->opt.num_iterations) ++B.launch;
  }
  B.ord_it = 0;
  { KTimer t(c, "k_gauge"); hipLaunchKernelGGL(k_gauge, grid, dim3(128), 0, s, B); }
  ++B.launch;
  if (c->opt.marginalization_flag == VPL_MARGIN_OLD) {
    { KTimer t(c, "k_lin_marg"); hipLaunchKernelGGL(k_lin<1>, grid, dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B); }
    ++B.launch;
    { KTimer t(c, "k_marg"); if (c->marg_small) hipLaunchKernelGGL(k_marg<256>, grid, dim3(256), c->marg_smem, s, B); else hipLaunchKernelGGL(k_marg<MARG_THREADS>, grid, dim3(MARG_THREADS), c->marg_smem, s, B); }
  } else if (c->any_second_new) {
    { KTimer t(c, "k_lin_marg"); hipLaunchKernelGGL(k_lin<2>, grid, dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B); }
    ++B.launch;
    { KTimer t(c, "k_marg"); if (c->marg_small) hipLaunchKernelGGL(k_marg<256>, grid, dim3(256), c->marg_smem, s, B); else hipLaunchKernelGGL(k_marg<MARG_THREADS>, grid, dim3(MARG_THREADS), c->marg_smem, s, B); }
  }
}

// vpl_ba_solve: the kernel-per-phase sequence over the whole batch, asynchronous on the context's stream.
int vpl_ba_solve(vpl_ctx* c) {
  if (c) { const int rs = settle(c); if (rs) return rs; }
  if (!c || c->nW < 1) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->opt.marginalization_flag != VPL_MARGIN_NONE) { c->prior_resident = true; c->prior_resident_nW = c->nW; }
  if (c->leg_timing) {
    HIPCHK(c, hipEventRecord(c->leg_ev[2], c->stream));
    launch_solve(c, 0, c->nW, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->leg_ev[3], c->stream));
    return VPL_OK;
  }
  if (c->use_graph && !c->timing && c->stream != nullptr) {   // (the legacy default stream cannot be captured)
    if (!c->graph_exec) {
      hipGraph_t graph = nullptr;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      launch_solve(c, 0, c->nW, c->stream);
      HIPCHK(c, hipStreamEndCapture(c->stream, &graph));
      hipError_t e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (e != hipSuccess) { c->graph_exec = nullptr; return fail(c, VPL_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    }
    HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
    return VPL_OK;
  }
  launch_solve(c, 0, c->nW, c->stream);
  HIPCHK(c, hipGetLastError());
  return VPL_OK;
}

// priors of the last marginalisation (k_marg) of the uploaded batch -> host; mn[2 w] = m, mn[2 w + 1] = n.
// Two steps around ONE device-to-host copy of the staging arena: plan (pieces of the arena + gather segments), then -- after
// stage_run(c, false) and a stream synchronisation -- finish (scatter into the caller's vpl_prior structs).
struct PriorFetch {
  Span<int> mg_n, mg_nb, mg_kind, mg_frame, mg_idx, mg_m;
  Span<double> mg_x0, mg_r0;
  std::vector<Span<double>> J0;   // per window: the host-side bound of the kept dims, squared
};
static size_t prior_fetch_bytes(vpl_ctx* c, int nW) {
  size_t b = (size_t)nW * ((3 + 3 * MAXPB) * 4 + (9 * MAXPB + MAXKEEP) * 8 + 64 + sizeof(CopySeg)) + 16 * 64;
  for (int w = 0; w < nW; ++w) {
    const int n = (size_t)w < c->h_mg_n.size() ? std::min(std::max(c->h_mg_n[w], 0), (int)MAXKEEP) : (int)MAXKEEP;
    b += (size_t)n * n * 8;
  }
  return b;
}
static int prior_fetch_plan(vpl_ctx* c, int nW, PriorFetch& F) {
  DevBatch& B = c->B;
  Stage& S = c->stage;
  const size_t W = nW;
  F.mg_m = S.take<int>(W); F.mg_n = S.take<int>(W); F.mg_nb = S.take<int>(W);
  F.mg_kind = S.take<int>(W * MAXPB); F.mg_frame = S.take<int>(W * MAXPB); F.mg_idx = S.take<int>(W * MAXPB);
  F.mg_x0 = S.take<double>(W * MAXPB * 9); F.mg_r0 = S.take<double>(W * MAXKEEP);
  S.from_device(F.mg_m.p, B.mg_m, W); S.from_device(F.mg_n.p, B.mg_n, W); S.from_device(F.mg_nb.p, B.mg_nb, W);
  S.from_device(F.mg_kind.p, B.mg_kind, W * MAXPB); S.from_device(F.mg_frame.p, B.mg_frame, W * MAXPB);
  S.from_device(F.mg_idx.p, B.mg_idx, W * MAXPB);
  S.from_device(F.mg_x0.p, B.mg_x0, W * MAXPB * 9); S.from_device(F.mg_r0.p, B.mg_r0, W * MAXKEEP);
  F.J0.resize(W);
  for (size_t w = 0; w < W; ++w) {
    const int n = w < c->h_mg_n.size() ? std::min(std::max(c->h_mg_n[w], 0), (int)MAXKEEP) : (int)MAXKEEP;
    F.J0[w] = S.take<double>((size_t)n * n);
    S.from_device(F.J0[w].p, B.mg_J0 + w * (size_t)MAXKEEP * MAXKEEP, (size_t)n * n);
  }
  if (S.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
  return VPL_OK;
}
static int prior_fetch_finish(vpl_ctx* c, int nW, const PriorFetch& F, vpl_prior* priors, std::vector<int>& mn) {
  DevBatch& B = c->B;
  const size_t W = nW;
  const Span<int>&mg_n = F.mg_n, &mg_nb = F.mg_nb, &mg_kind = F.mg_kind, &mg_frame = F.mg_frame, &mg_idx = F.mg_idx, &mg_m = F.mg_m;
  const Span<double>& mg_x0 = F.mg_x0, &mg_r0 = F.mg_r0;
  mn.assign(2 * W, 0);
  for (size_t w = 0; w < W; ++w) {
    mn[2 * w] = mg_m[w];
    if (c->h_passthrough[w] >= 0) {   // MARGIN_SECOND_NEW without pose WINDOW_SIZE-1 in the prior: the prior stays (estimator.cpp:1385)
      priors[w] = c->h_pass_priors[c->h_passthrough[w]];
      if (priors[w].n < 0) {          // chained upload: the untouched prior lives on the device only
        vpl_prior& p = priors[w];
        std::memset(&p, 0, sizeof(int) * (2 + 3 * VPL_MAX_PRIOR_BLOCKS));
        HIPCHK(c, hipMemcpy(&p.n, B.pr_n + w, 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&p.n_blocks, B.pr_nb + w, 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.block_kind, B.pr_kind + w * MAXPB, MAXPB * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.block_frame, B.pr_frame + w * MAXPB, MAXPB * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.block_idx, B.pr_idx + w * MAXPB, MAXPB * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.x0, B.pr_x0 + w * MAXPB * 9, MAXPB * 9 * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.J0, B.pr_J0 + w * (size_t)B.prS, (size_t)p.n * p.n * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(p.r0, B.pr_r0 + w * MAXPN, (size_t)p.n * 8, hipMemcpyDeviceToHost));
      }
      mn[2 * w + 1] = priors[w].n;
      continue;
    }
    vpl_prior& p = priors[w];
    std::memset(&p, 0, sizeof(int) * (2 + 3 * VPL_MAX_PRIOR_BLOCKS));
    const int n = mg_n[w];
    p.n = n; p.n_blocks = mg_nb[w];
    mn[2 * w + 1] = n;
    for (int b = 0; b < p.n_blocks; ++b) {
      p.block_kind[b] = mg_kind[w * MAXPB + b];
      p.block_frame[b] = mg_frame[w * MAXPB + b];
      p.block_idx[b] = mg_idx[w * MAXPB + b];
      std::memcpy(p.x0[b], &mg_x0[(w * MAXPB + b) * 9], 9 * 8);
    }
    if ((size_t)n * n > F.J0[w].n) return fail(c, VPL_E_HIP, "internal: the device kept more prior dims than the host's bound");
    std::memcpy(p.J0, F.J0[w].p, (size_t)n * n * 8);
    std::memcpy(p.r0, &mg_r0[w * MAXKEEP], (size_t)n * 8);
  }
  return VPL_OK;
}
static int fetch_priors(vpl_ctx* c, int nW, vpl_prior* priors, std::vector<int>& mn) {
  HIPCHK(c, c->stage.reserve(prior_fetch_bytes(c, nW) + 4096));
  PriorFetch F;
  int rc = prior_fetch_plan(c, nW, F);
  if (rc) return rc;
  HIPCHK(c, stage_run(c, false));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return prior_fetch_finish(c, nW, F, priors, mn);
}

// States of the uploaded batch -> caller's DEVICE buffer [nW][183], asynchronous on the context's stream
int vpl_ba_pack_states_device(vpl_ctx* c, int nW, void* d_states) {
  if (c) { const int rs = settle(c); if (rs) return rs; }
  if (!c || nW != c->nW || !d_states) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_pack_states, dim3(nW), dim3(192), 0, c->stream, c->B, (double*)d_states);
  HIPCHK(c, hipGetLastError());
  return VPL_OK;
}

int vpl_ba_download(vpl_ctx* c, int nW, vpl_window* win, vpl_prior* priors, vpl_solve_report* reports) {
  if (c) { const int rs = settle(c); if (rs) return rs; }
  if (!c || nW != c->nW || !win) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  DevBatch& B = c->B;
  const size_t W = nW;
  const bool marg = priors != nullptr && c->opt.marginalization_flag != VPL_MARGIN_NONE;
  hipStream_t s = c->stream;
  if (c->leg_timing) HIPCHK(c, hipEventRecord(c->leg_ev[4], s));
  // one gather kernel into the device arena, ONE device-to-host copy into the pinned arena
  Stage& S = c->stage;
  HIPCHK(c, S.reserve(W * ((77 + 99 + 7 + B.maxP + 6 * B.maxL) * 8 + sizeof(TrState) + 4 + B.maxL * 4) + 16 * 64 + 16 * sizeof(CopySeg) +
                      (marg ? prior_fetch_bytes(c, nW) : 0) + 4096));
  Span<double> pose = S.take<double>(W * 77), sb = S.take<double>(W * 99), ex = S.take<double>(W * 7), invd = S.take<double>(W * B.maxP),
               plk = S.take<double>(W * B.maxL * 6);
  Span<TrState> tr = S.take<TrState>(W);
  Span<int> mg_m = S.take<int>(W), removed = S.take<int>(W * B.maxL);
  std::vector<int> mn;
  S.from_device(pose.p, B.pose, W * 77); S.from_device(sb.p, B.sb, W * 99); S.from_device(ex.p, B.ex, W * 7);
  S.from_device(invd.p, B.invd, W * B.maxP); S.from_device(plk.p, B.plk, W * B.maxL * 6); S.from_device(tr.p, B.tr, W);
  S.from_device(mg_m.p, B.mg_m, W); S.from_device(removed.p, B.ln_removed, W * B.maxL);
  PriorFetch F;
  if (marg) { const int rc = prior_fetch_plan(c, nW, F); if (rc) return rc; }
  if (S.overflow) return fail(c, VPL_E_CAPACITY, "internal: staging arena bound too small");
  HIPCHK(c, stage_run(c, false));
  if (c->leg_timing) HIPCHK(c, hipEventRecord(c->leg_ev[5], s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (marg) {
    const int rc = prior_fetch_finish(c, nW, F, priors, mn);
    if (rc) return rc;
  }
  for (size_t w = 0; w < W; ++w) {
    vpl_window& v = win[w];
    std::memcpy(v.pose, &pose[w * 77], 77 * 8);
    std::memcpy(v.speed_bias, &sb[w * 99], 99 * 8);
    std::memcpy(v.ex_pose, &ex[w * 7], 7 * 8);
    for (int p = 0; p < v.n_points; ++p) v.inv_depth[p] = invd[w * B.maxP + p];
    const std::vector<int>& lmap = c->h_lmap[w];
    for (size_t dl = 0; dl < lmap.size(); ++dl)   // erased tracks keep the caller's value (they are gone from f_manager)
      if (!removed[w * B.maxL + dl]) std::memcpy(v.line_plk + (size_t)lmap[dl] * 6, &plk[(w * B.maxL + dl) * 6], 6 * 8);
    if (reports) {
      vpl_solve_report& r = reports[w];
      std::memset(&r, 0, sizeof(r));
      r.iterations = tr[w].iter;
      r.num_successful_steps = tr[w].num_successful;
      r.termination = tr[w].status == 1 ? 1 : tr[w].status == 2 ? 2 : 0;
      r.initial_cost = tr[w].initial_cost;
      r.final_cost = tr[w].x_cost;
      r.prior_m = mg_m[w];
      r.prior_n = marg ? mn[2 * w + 1] : 0;
      for (size_t dl = 0; dl < lmap.size(); ++dl) r.n_lines_removed += removed[w * B.maxL + dl] ? 1 : 0;
    }
    if (v.line_removed) {
      for (int l = 0; l < v.n_lines; ++l) v.line_removed[l] = 0;
      for (size_t dl = 0; dl < lmap.size(); ++dl) v.line_removed[lmap[dl]] = removed[w * B.maxL + dl] ? 1 : 0;
    }
  }
  return VPL_OK;
}

// MarginalizationInfo::{addResidualBlockInfo, preMarginalize, marginalize} for a batch of windows WITHOUT a solve
// (marginalization_factor.cpp:89-129,177-363 as driven by estimator.cpp:1229-1447): the factor subset of the flag is
// linearised at the windows' current states (k_lin<1|2>), the landmarks and the dropped frame are eliminated and the kept
// block is factored into (J0, r0) (k_marg).  The states are not touched.
static int marginalize_impl(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt_in, int marginalization_flag,
                            vpl_prior* priors, int* m_out, int* n_out, bool async) {
  if (!c || !win || !opt_in || !priors || nW < 1) return VPL_E_INVALID;
  if (marginalization_flag != VPL_MARGIN_OLD && marginalization_flag != VPL_MARGIN_SECOND_NEW)
    return fail(c, VPL_E_INVALID, "vpl_ba_marginalize: flag must be VPL_MARGIN_OLD or VPL_MARGIN_SECOND_NEW");
  vpl_ba_options opt = *opt_in;
  opt.marginalization_flag = marginalization_flag;
  opt.remove_line_outliers = 0;
  int rc = upload_impl(c, nW, win, &opt, false);
  if (rc) return rc;
  DevBatch B = c->B;
  B.ord_it = 0; B.act = nullptr; B.launch = 0;
  const dim3 grid(nW);
  hipStream_t s = c->stream;
  // no line of this call is erased (remove_line_outliers = 0 above): the flags k_lin<MARG> and k_marg read are written by
  // k_gauge, which runs in a solve only -- without this they would be those of the batch the context solved before
  HIPCHK(c, hipMemsetAsync(B.ln_removed, 0, (size_t)nW * B.maxL * sizeof(int), s));
  // k_prep: whitening matrices, q <- Quaterniond(R(q)) and the world orth of the lines (the vector2double() the reference
  // runs before it marginalises, estimator.cpp:1233), J0^T J0 of the incoming prior
  { KTimer t(c, "k_prep"); hipLaunchKernelGGL(k_prep, grid, dim3(PREP_THREADS), prep_smem(c->maxPriorN), s, B, std::min(c->maxPriorN, PREP_NMAX)); }
  bool ran = false;
  if (marginalization_flag == VPL_MARGIN_OLD) {
    { KTimer t(c, "k_lin_marg"); hipLaunchKernelGGL(k_lin<1>, grid, dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B); }
    ran = true;
  } else if (c->any_second_new) {
    { KTimer t(c, "k_lin_marg"); hipLaunchKernelGGL(k_lin<2>, grid, dim3(LIN_THREADS), lin_smem(c->maxP, c->maxL), s, B); }
    ran = true;
  }
  c->prior_resident = true; c->prior_resident_nW = nW;
  if (ran) { KTimer t(c, "k_marg"); if (c->marg_small) hipLaunchKernelGGL(k_marg<256>, grid, dim3(256), c->marg_smem, s, B); else hipLaunchKernelGGL(k_marg<MARG_THREADS>, grid, dim3(MARG_THREADS), c->marg_smem, s, B); }
  HIPCHK(c, hipGetLastError());
  // (the priors are fetched from the device when the call completes: any later call on the context completes this one first)
  return finish_or_defer(c, async, [c, nW, priors, m_out, n_out]() -> int {
    std::vector<int> mn;
    const int rf = fetch_priors(c, nW, priors, mn);
    if (rf) return rf;
    for (int w = 0; w < nW; ++w) {
      if (m_out) m_out[w] = mn[2 * w];
      if (n_out) n_out[w] = mn[2 * w + 1];
    }
    return VPL_OK;
  });
}
int vpl_ba_marginalize(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt, int marginalization_flag,
                       vpl_prior* priors, int* m_out, int* n_out) {
  return marginalize_impl(c, nW, win, opt, marginalization_flag, priors, m_out, n_out, false);
}
int vpl_ba_marginalize_async(vpl_ctx* c, int nW, const vpl_window* win, const vpl_ba_options* opt, int marginalization_flag,
                             vpl_prior* priors, int* m_out, int* n_out) {
  return marginalize_impl(c, nW, win, opt, marginalization_flag, priors, m_out, n_out, true);
}

int vpl_ba_solve_windows(vpl_ctx* c, int nW, vpl_window* win, const vpl_ba_options* opt, vpl_prior* priors,
                         vpl_solve_report* reports) {
  int rc = vpl_ba_upload(c, nW, win, opt);
  if (rc) return rc;
  rc = vpl_ba_solve(c);
  if (rc) return rc;
  rc = vpl_ctx_synchronize(c);
  if (rc) return rc;
  return vpl_ba_download(c, nW, win, priors, reports);
}

// f_manager.triangulate and f_manager.triangulateLine on ONE upload (both work on the window with every line, and touch
// different arrays): what solveOdometry's two threads start with
static int triangulate_points_and_lines(vpl_ctx* c, int nW, vpl_window* win, double init_depth) {
  if (!(init_depth > 0.0)) return VPL_E_INVALID;
  vpl_ba_options opt;
  vpl_ba_default_options(&opt);
  opt.marginalization_flag = VPL_MARGIN_NONE;
  int rc = upload_impl(c, nW, win, &opt, true);
  if (rc) return rc;
  DevBatch& B = c->B;
  hipStream_t s = c->stream;
  { KTimer t(c, "k_triangulate_points"); hipLaunchKernelGGL(k_triangulate_points, dim3(nW), dim3(128), 0, s, B, init_depth); }
  { KTimer t(c, "k_triangulate"); hipLaunchKernelGGL(k_triangulate, dim3(nW), dim3(128), 0, s, B); }
  HIPCHK(c, hipGetLastError());
  const size_t W = nW;
  std::vector<double> invd(W * B.maxP), plk(W * B.maxL * 6);
  std::vector<int> tri(W * B.maxL);
  HIPCHK(c, hipMemcpyAsync(invd.data(), B.invd, invd.size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(plk.data(), B.plk, plk.size() * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(tri.data(), B.ln_tri, tri.size() * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  for (size_t w = 0; w < W; ++w) {
    vpl_window& v = win[w];
    for (int p = 0; p < v.n_points; ++p) v.inv_depth[p] = invd[w * B.maxP + p];
    const std::vector<int>& lmap = c->h_lmap[w];
    for (size_t dl = 0; dl < lmap.size(); ++dl) {
      const int l = lmap[dl];
      if (!v.line_triangulated[l] && tri[w * B.maxL + dl]) {
        std::memcpy(v.line_plk + (size_t)l * 6, &plk[(w * B.maxL + dl) * 6], 6 * 8);
        v.line_triangulated[l] = 1;
      }
    }
  }
  return VPL_OK;
}

// Estimator::solveOdometry (estimator.cpp:624-648) for a batch, in one call: triangulate || (triangulateLine -> onlyLineOpt)
// -> optimizationwithLine.  The two line stages change WHICH lines take part (newly triangulated ones join, the ones
// removeLineOutlier erases leave), and the lane / unit / K-step tables of the kernels are built on the host from that set: the
// stages are the entry points above run back to back on the caller's arrays (the two triangulations share one upload).
int vpl_ba_solve_odometry(vpl_ctx* c, int nW, vpl_window* win, const vpl_ba_options* opt, double init_depth,
                          vpl_prior* priors, vpl_solve_report* line_reports, vpl_solve_report* reports) {
  if (!c || !win || !opt || nW < 1) return VPL_E_INVALID;
  bool any_lines = false;
  for (int w = 0; w < nW; ++w)
    if (win[w].n_lines > 0) {
      any_lines = true;
      if (!win[w].line_triangulated || !win[w].line_removed)
        return fail(c, VPL_E_INVALID, "solve_odometry: line_triangulated and line_removed are required for windows with lines");
    }
  using oclk = std::chrono::steady_clock;
  const auto t0 = oclk::now();
  int rc = any_lines ? triangulate_points_and_lines(c, nW, win, init_depth) : vpl_ba_triangulate_points(c, nW, win, init_depth);
  if (rc) return rc;
  const auto t1 = oclk::now();
  c->odo_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
  c->odo_ms[1] = 0.0;
  bool batch_resident = false;
  if (any_lines) {
    bool orth_given = false;
    for (int w = 0; w < nW; ++w) orth_given = orth_given || win[w].line_orth != nullptr;
    // uploaded with the FINAL solve's options: when no line is erased the same batch is solved where it lies (round 4)
    rc = only_line_opt_impl(c, nW, win, opt, line_reports, false, (orth_given || std::getenv("VPL_BA_ODO_REUPLOAD")) ? nullptr : opt);
    c->odo_ms[1] = std::chrono::duration<double, std::milli>(oclk::now() - t1).count();
    if (rc) return rc;
    // f_manager.removeLineOutlier erased these tracks (estimator.cpp:1037): they take no part in the solve
    bool erased = false;
    for (int w = 0; w < nW; ++w)
      for (int l = 0; l < win[w].n_lines; ++l)
        if (win[w].line_removed[l]) { win[w].line_triangulated[l] = 0; erased = true; }
    batch_resident = !erased && !orth_given && !std::getenv("VPL_BA_ODO_REUPLOAD");
  } else if (line_reports) {
    std::memset(line_reports, 0, sizeof(vpl_solve_report) * (size_t)nW);
  }
  const auto t2 = oclk::now();
  if (batch_resident) {
    // What a fresh upload of the caller's arrays would put on the device is there already: the layout tables, observations,
    // prior and kept-block tables of this very line set, the optimised Pluecker vectors (the caller's copy was downloaded
    // from B.plk), and -- after this restore -- the states k_prep re-normalised in place.  Same bits as the four-stage call.
    DevBatch& B = c->B;
    const size_t W = nW;
    HIPCHK(c, hipMemcpyAsync(B.pose, B.pose_0, W * 77 * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.sb, B.sb_0, W * 99 * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.ex, B.ex_0, W * 7 * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.invd, B.invd_0, W * B.maxP * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(B.ln_removed, 0, W * B.maxL * 4, c->stream));
    rc = vpl_ba_solve(c);
    if (!rc) rc = vpl_ctx_synchronize(c);
    if (!rc) rc = vpl_ba_download(c, nW, win, priors, reports);
  } else {
    rc = vpl_ba_solve_windows(c, nW, win, opt, priors, reports);
  }
  c->odo_ms[2] = std::chrono::duration<double, std::milli>(oclk::now() - t2).count();
  return rc;
}
// wall clock of the three stages of the last vpl_ba_solve_odometry: triangulation | onlyLineOpt | optimizationwithLine (ms)
int vpl_ba_debug_odometry_ms(vpl_ctx* c, double* ms3) {
  if (!c || !ms3) return VPL_E_INVALID;
  for (int k = 0; k < 3; ++k) ms3[k] = c->odo_ms[k];
  return VPL_OK;
}

// Debug/test access to the marginalisation invariants (A, b before the final eigen-decomposition),
// mirroring the reference's commented check at marginalization_factor.cpp:361-362.
int vpl_ba_debug_marg_Ab(vpl_ctx* c, int w, double* A, double* b) {
  if (!c || w < 0 || w >= c->nW) return VPL_E_INVALID;
  int n = 0;
  HIPCHK(c, hipMemcpy(&n, c->B.mg_n + w, 4, hipMemcpyDeviceToHost));
  std::vector<double> t((size_t)n * n);
  HIPCHK(c, hipMemcpy(A, c->B.mg_A + (size_t)w * MAXKEEP * MAXKEEP, (size_t)n * n * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(b, c->B.mg_b + (size_t)w * MAXKEEP, (size_t)n * 8, hipMemcpyDeviceToHost));
  return n;
}

// Debug aid of the randomised sweeps (tools/fuzz_*.py with VPL_DEBUG_GUARDS=1 in the environment when the context is made): the
// 64 bytes behind every device array then hold 0xA5; returns how many arrays have had theirs written to (a kernel ran past the
// end of an array), the first one named in vpl_last_error by its allocation index and size.
int vpl_ba_debug_guards(vpl_ctx* c) {
  if (!c) return VPL_E_INVALID;
  if (!c->guards) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  int bad = 0;
  unsigned char pad[64];
  for (size_t i = 0; i < c->allocs.size(); ++i) {
    HIPCHK(c, hipMemcpy(pad, (char*)c->allocs[i] + c->alloc_bytes[i], 64, hipMemcpyDeviceToHost));
    bool hit = false;
    for (int k = 0; k < 64; ++k) hit |= pad[k] != 0xA5;
    if (hit && !bad++) c->err = "guard behind device array #" + std::to_string(i) + " (" + std::to_string(c->alloc_bytes[i]) + " bytes) overwritten";
  }
  return bad;
}

int vpl_ba_debug_stamps(vpl_ctx* c, int w, long long* out) {
  HIPCHK(c, hipMemcpy(out, c->B.dbg + (size_t)w * 64, 64 * 8, hipMemcpyDeviceToHost));
  return VPL_OK;
}

// Host-only (no device call): the point work-unit tables upload builds for one window, and a replay of their commit chains.
int vpl_ba_debug_point_units(int n_points, const int* point_start, const int* point_nobs, int max_rounds, int* lane_table,
                             int* unit_table, int* rounds, int* rounds0) {
  if (n_points < 0 || !point_start || !point_nobs || !lane_table || !unit_table || !rounds || !rounds0 || max_rounds < 1) return VPL_E_INVALID;
  std::vector<int> off(n_points + 1, 0), list(n_points + 1, 0);
  int cnt[NF + 1] = {0};
  int o = 0;
  for (int p = 0; p < n_points; ++p) {
    if (point_start[p] < 0 || point_nobs[p] < 2 || point_start[p] + point_nobs[p] > NF) return VPL_E_INVALID;
    off[p] = o; o += point_nobs[p];
    cnt[point_start[p] + 1]++;
  }
  for (int f = 0; f < NF; ++f) cnt[f + 1] += cnt[f];
  int pos[NF + 1];
  for (int f = 0; f <= NF; ++f) pos[f] = cnt[f];
  for (int p = 0; p < n_points; ++p) list[pos[point_start[p]]++] = p;
  for (int f = 0; f < NF; ++f)
    std::stable_sort(&list[cnt[f]], &list[cnt[f + 1]], [&](int a, int b) { return point_nobs[a] > point_nobs[b]; });
  PointUnitLayout PL;
  if (!pack_point_units(point_nobs, off.data(), list.data(), cnt, max_rounds, lane_table, unit_table, &PL)) return VPL_E_CAPACITY;
  *rounds = PL.rounds; *rounds0 = PL.rounds0;
  return VPL_OK;
}
int vpl_ba_debug_point_chains(const int* unit_table, int rounds, int rounds0, int marg_pass) {
  if (!unit_table || rounds < 0 || rounds0 < 0 || rounds0 > rounds) return VPL_E_INVALID;
  PointUnitLayout PL;
  PL.rounds = rounds; PL.rounds0 = rounds0;
  return point_unit_chains_finish(unit_table, PL, marg_pass != 0) ? 1 : 0;
}

int vpl_ba_enable_kernel_timing(vpl_ctx* c, int enable) {
  if (!c) return VPL_E_INVALID;
  c->timing = enable != 0;
  c->ktimes.clear();
  return VPL_OK;
}
int vpl_ba_kernel_times(vpl_ctx* c, int* count, const char** names, double* total_ms, int* launches) {
  if (!c || !count) return VPL_E_INVALID;
  int cap = *count, i = 0;
  c->kname_store.clear();
  for (auto& kv : c->ktimes) c->kname_store.push_back(kv.first);
  for (auto& kv : c->ktimes) {
    if (i >= cap) break;
    names[i] = c->kname_store[i].c_str();
    total_ms[i] = kv.second.first;
    launches[i] = kv.second.second;
    ++i;
  }
  *count = i;
  return VPL_OK;
}

// Per-launch profile of the LAST solve run with kernel timing enabled: kernel name, device time and the number of windows
// that did work in the launch (k_lin: linearised; k_solve: [1] computed a new Gauss-Newton step, [2] re-used the step of a
// rejected iteration; k_cost: evaluated a candidate).  active is [count][4] in the order k_lin, k_solve new, k_solve
// re-used, k_cost.
int vpl_ba_launch_profile(vpl_ctx* c, int* count, const char** names, double* ms, int* active) {
  if (!c || !count || !names || !ms || !active) return VPL_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int> act((size_t)ACT_SLOTS * 4);
  HIPCHK(c, hipMemcpy(act.data(), c->d_act, act.size() * sizeof(int), hipMemcpyDeviceToHost));
  const int n = std::min(*count, (int)c->ltimes.size());
  for (int i = 0; i < n; ++i) {
    names[i] = c->ltimes[i].first;
    ms[i] = c->ltimes[i].second;
    for (int k = 0; k < 4; ++k) active[4 * i + k] = i < ACT_SLOTS ? act[4 * i + k] : 0;
  }
  *count = n;
  return VPL_OK;
}

}  // extern "C"
