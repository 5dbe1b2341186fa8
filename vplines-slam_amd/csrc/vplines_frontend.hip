// C ABI of the MI355X-native line front-end (include/vplines_frontend.h): EDLines extractor + KLT line matcher.
// No CPU compute path: every call that computes launches HIP kernels or returns an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "vplines_ba.h"        // VPL_E_* codes
#include "vplines_frontend.h"
#include "ed_kernels.h"
#include "lm_kernels.h"
#include "pre_kernels.h"
#include "vp_kernels.h"

using namespace vpl;

struct vpl_fe_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int maxN = 0, W = 0, H = 0, maxLines = 0;
  int n = 0;
  size_t routeSmem = 0;
  EdBatch B;
  LmBatch M;
  int maxPairs = 0, nPairs = 0;
  bool lmReserved = false;
  int *d_refImg = nullptr, *d_curImg = nullptr, *d_nRef = nullptr, *d_nCur = nullptr;
  vpl_line *d_linesRef = nullptr, *d_linesCur = nullptr;
  int blurMode = VPL_BLUR_NORMALISED; // vpl_fe_set_blur_kernel
  uint8_t* d_blur = nullptr;          // [maxN][H][W] copy of the blurred frames (vpl_fe_keep_blurred; tests)
  vpl_line* d_sorted = nullptr;       // [maxN][maxLines] the detected lines in the reference's order (k_ed_sort_lines)
  int* d_sortedCnt = nullptr;         // [maxN]
  bool matchFromDetected = false;     // the pending match took its lines from the detector's table (overflow is checked at its download)
  // image preparation (remap + CLAHE)
  uint8_t *d_raw = nullptr, *d_mid = nullptr, *d_lut = nullptr;
  float *d_mapx = nullptr, *d_mapy = nullptr;
  bool haveMaps = false;
  int lutTiles = 0;
  // vanishing points
  VpBatch V;
  bool vpReserved = false;
  float *d_vpHyp = nullptr, *d_vpAll = nullptr;
  int *d_vpNHyp = nullptr, *d_vpNAll = nullptr, *d_vpFirst = nullptr;
  uint32_t* d_vpSeed = nullptr;
  int vpN = 0;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;   // payload of allocs[i]; 64 pad bytes follow (VPL_DEBUG_GUARDS=1: 0xA5, vpl_fe_debug_guards)
  bool guards = false;
  std::string err;
  bool timing = false;                                    // vpl_fe_enable_kernel_timing
  std::vector<std::pair<const char*, double>> ktimes;     // (kernel, ms) of the launches since timing was enabled
};

// times one kernel launch with hipEvents on the context's stream when timing is on (bench.py: k_ed_grad GB/s)
struct FeTimer {
  vpl_fe_ctx* c;
  const char* name;
  hipEvent_t a = nullptr, b = nullptr;
  FeTimer(vpl_fe_ctx* c_, const char* n) : c(c_), name(n) {
    if (c->timing) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, c->stream); }
  }
  ~FeTimer() {
    if (c->timing) {
      hipEventRecord(b, c->stream);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      c->ktimes.emplace_back(name, (double)ms);
      hipEventDestroy(a);
      hipEventDestroy(b);
    }
  }
};

static int fe_fail(vpl_fe_ctx* c, int code, const std::string& m) {
  if (c) c->err = m;
  return code;
}
#define FECHK(ctx, call)                                                                                \
  do {                                                                                                  \
    hipError_t e__ = (call);                                                                            \
    if (e__ != hipSuccess) return fe_fail(ctx, VPL_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

template <typename T>
static hipError_t fe_alloc(vpl_fe_ctx* c, T** p, size_t n) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, n * sizeof(T) + 64);
  if (e != hipSuccess) return e;
  c->allocs.push_back(q);
  c->alloc_bytes.push_back(n * sizeof(T));
  *p = (T*)q;
  e = hipMemset(q, 0, n * sizeof(T) + 64);
  if (e == hipSuccess && c->guards) e = hipMemset((char*)q + n * sizeof(T), 0xA5, 64);
  return e;
}

extern "C" {

void vpl_edline_default_param(vpl_edline_param* p) {
  p->ksize = 5; p->sigma = 1.0f; p->gradientThreshold = 30.f; p->anchorThreshold = 5.f; p->scanIntervals = 2;
  p->minLineLen = 35; p->lineFitErrThreshold = 1.8;
}

// The detected lines of a frame in their deterministic order -- by (edge chain, ordinal inside the chain), the order the
// reference's sequential loop produces -- and in the vpl_line layout, ON THE DEVICE: vpl_edlines_download copies this table, and
// vpl_match_from_detected hands it to the matcher without a host round trip.  One work-group per frame; the rank of a line is
// the number of keys below its own (keys are unique; <= maxLines lines).
__global__ __launch_bounds__(256) void k_ed_sort_lines(EdBatch B, vpl_line* out, int* outCnt, int ML) {
  const int n = blockIdx.x;
  const int m = min(B.nLines[n], ML);
  if (threadIdx.x == 0) outCnt[n] = m;
  const uint32_t* K = B.lkey + (size_t)n * ML;
  for (int k = threadIdx.x; k < m; k += blockDim.x) {
    const uint32_t key = K[k];
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += K[j] < key ? 1 : 0;
    const double* o = B.lines + ((size_t)n * ML + k) * 10;
    vpl_line& dst = out[(size_t)n * ML + rank];
    for (int q = 0; q < 4; ++q) dst.line_endpoint[q] = (float)o[q];
    for (int q = 0; q < 3; ++q) dst.line_equation[q] = o[4 + q];
    dst.center[0] = (float)o[7]; dst.center[1] = (float)o[8];
    dst.length = (float)o[9];
  }
}
// lines of the pairs' frames out of the sorted table (at most `cap` per frame)
__global__ __launch_bounds__(256) void k_lm_take_lines(const vpl_line* sorted, const int* cnt, int ML, int cap, const int* refImg,
                                                       const int* curImg, vpl_line* linesRef, vpl_line* linesCur, int* nRef, int* nCur) {
  const int i = blockIdx.x, side = blockIdx.y;
  const int img = side ? curImg[i] : refImg[i];
  const int m = min(cnt[img], cap);
  if (threadIdx.x == 0) (side ? nCur : nRef)[i] = m;
  vpl_line* dst = (side ? linesCur : linesRef) + (size_t)i * ML;
  const vpl_line* src = sorted + (size_t)img * ML;
  for (int k = threadIdx.x; k < m; k += blockDim.x) dst[k] = src[k];
}

int vpl_fe_create(vpl_fe_ctx** out, int device, int max_images, int width, int height, int max_lines_per_image) {
  if (!out || max_images < 1 || width < 8 || height < 8 || max_lines_per_image < 1) return VPL_E_INVALID;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0 || device >= nd) return VPL_E_NODEVICE;
  if (hipSetDevice(device) != hipSuccess) return VPL_E_NODEVICE;
  vpl_fe_ctx* c = new vpl_fe_ctx();
  { const char* g = getenv("VPL_DEBUG_GUARDS"); c->guards = g && g[0] == '1'; }
  c->device = device; c->maxN = max_images; c->W = width; c->H = height; c->maxLines = max_lines_per_image;
  EdBatch& B = c->B;
  std::memset(&B, 0, sizeof(B));
  B.W = width; B.H = height;
  B.cap = width * height / 5;      // edgePixelArraySize (edline_detector.cpp:94)
  B.capEdges = B.cap / 20;         // maxNumOfEdge (:95)
  B.maxLines = max_lines_per_image;
  const size_t N = max_images, PX = (size_t)width * height;
  hipError_t e = hipSuccess;
  uint8_t* img = nullptr;
#define AL(ptr, n) if (e == hipSuccess) e = fe_alloc(c, &ptr, (size_t)(n))
  AL(img, N * PX); B.img = img;
  AL(B.dx, N * PX); AL(B.dy, N * PX); AL(B.g, N * PX); AL(B.dir, N * PX);
  B.Wc = (width + ED_TILE - 1) / ED_TILE * ED_TILE;
  AL(B.code, N * (size_t)height * B.Wc); AL(B.rstats, N * 4);
  AL(B.anchX, N * B.cap); AL(B.anchY, N * B.cap); AL(B.nAnch, N);
  AL(B.fX, N * B.cap); AL(B.fY, N * B.cap); AL(B.cX, N * 2 * B.cap); AL(B.cY, N * 2 * B.cap);
  AL(B.sId, N * (B.capEdges + 2)); AL(B.nEdges, N);
  AL(B.lines, N * B.maxLines * 10); AL(B.lkey, N * B.maxLines); AL(B.nLines, N);
  AL(c->d_sorted, N * B.maxLines); AL(c->d_sortedCnt, N);
#undef AL
  // k_ed_route keeps the frame's edge bitmap and one tile of routing codes in LDS
  {
    // edge bitmap + routing strip [HS][128] bytes + request block; HS = the frame height rounded up to 32 when that fits into
    // 159 KB, else what is left (at least 128 rows: frames above ~1.1 Mpixel are refused)
    const size_t bits = (size_t)(((((size_t)width * height + 31) >> 5) + 3) & ~(size_t)3) * 4;
    int hs = (height + 31) / 32 * 32;
    while (hs > 128 && bits + (size_t)hs * ED_TILE + 64 * 4 > 159 * 1024) hs -= 32;
    // More frames in flight than the device has CUs: a strip of at most half the LDS lets TWO frames' walkers share a CU (the
    // walk is one wave; the strip then moves vertically as well, which the helper waves make cheap) -- measured below.
    {
      int ncu = 0;
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      const char* ev = std::getenv("VPL_FE_ROUTE_HS");
      int want = ev ? std::atoi(ev) : 0;
      if (!ev && ncu > 0 && max_images > ncu) {
        want = hs;
        while (want > 128 && bits + (size_t)want * ED_TILE + 64 * 4 > 79 * 1024) want -= 32;
        if (bits + (size_t)want * ED_TILE + 64 * 4 > 79 * 1024) want = hs;      // does not fit twice anyway
      }
      if (want >= 32 && want < hs && want % 32 == 0) hs = want;
    }
    // (rows of the strip past the frame are zero-filled by the loader)
    B.routeHS = hs;
    c->routeSmem = bits + (size_t)hs * ED_TILE + 64 * 4;
  }
  if (e == hipSuccess && c->routeSmem > 159 * 1024) e = hipErrorInvalidValue;   // frames above ~1.1 Mpixel
  static size_t route_max = 0;   // per kernel, not per context: never lowered by a later, smaller context
  if (e == hipSuccess && c->routeSmem > 48 * 1024 && c->routeSmem > route_max) {
    e = hipFuncSetAttribute((const void*)k_ed_route, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->routeSmem);
    if (e == hipSuccess) route_max = c->routeSmem;
  }
  if (e != hipSuccess) {
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return e == hipErrorInvalidValue ? VPL_E_CAPACITY : VPL_E_HIP;
  }
  *out = c;
  return VPL_OK;
}

void vpl_fe_destroy(vpl_fe_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (void* p : c->allocs) hipFree(p);
  delete c;
}
int vpl_fe_set_stream(vpl_fe_ctx* c, void* s) {
  if (!c) return VPL_E_INVALID;
  if (c->stream != (hipStream_t)s) {   // what was enqueued on the stream so far is completed first (downloads follow on the new one)
    FECHK(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
  }
  c->stream = (hipStream_t)s;
  return VPL_OK;
}
int vpl_fe_enable_kernel_timing(vpl_fe_ctx* c, int enable) {
  if (!c) return VPL_E_INVALID;
  c->timing = enable != 0;
  c->ktimes.clear();
  return VPL_OK;
}
int vpl_fe_kernel_times(vpl_fe_ctx* c, int* count, const char** names, double* ms) {
  if (!c || !count || !names || !ms) return VPL_E_INVALID;
  const int n = std::min(*count, (int)c->ktimes.size());
  for (int i = 0; i < n; ++i) { names[i] = c->ktimes[i].first; ms[i] = c->ktimes[i].second; }
  *count = n;
  return VPL_OK;
}
int vpl_fe_synchronize(vpl_fe_ctx* c) { if (!c) return VPL_E_INVALID; FECHK(c, hipStreamSynchronize(c->stream)); return VPL_OK; }
const char* vpl_fe_last_error(const vpl_fe_ctx* c) { return c ? c->err.c_str() : "null context"; }
// Debug aid of the randomised sweeps (VPL_DEBUG_GUARDS=1 when the context is made: the 64 bytes behind every device array hold
// 0xA5): how many arrays have had theirs written to; the first is named in vpl_fe_last_error.
int vpl_fe_debug_guards(vpl_fe_ctx* c) {
  if (!c) return VPL_E_INVALID;
  if (!c->guards) return 0;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipDeviceSynchronize());
  int bad = 0;
  unsigned char pad[64];
  for (size_t i = 0; i < c->allocs.size(); ++i) {
    FECHK(c, hipMemcpy(pad, (char*)c->allocs[i] + c->alloc_bytes[i], 64, hipMemcpyDeviceToHost));
    bool hit = false;
    for (int k = 0; k < 64; ++k) hit |= pad[k] != 0xA5;
    if (hit && !bad++) c->err = "guard behind device array #" + std::to_string(i) + " (" + std::to_string(c->alloc_bytes[i]) + " bytes) overwritten";
  }
  return bad;
}

int vpl_edlines_upload(vpl_fe_ctx* c, int n, const uint8_t* images) {
  if (!c || !images || n < 1) return VPL_E_INVALID;
  if (n > c->maxN) return fe_fail(c, VPL_E_CAPACITY, "more images than max_images");
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipMemcpyAsync((void*)c->B.img, images, (size_t)n * c->W * c->H, hipMemcpyHostToDevice, c->stream));
  c->n = n;
  c->B.N = n;
  return VPL_OK;
}

// ---- image preparation: cv::remap + CLAHE of LineFeatureTracker::readImage (line_feature_tracker.cpp:62-68) ----
int vpl_pre_set_maps(vpl_fe_ctx* c, const float* map_x, const float* map_y) {
  if (!c || (map_x == nullptr) != (map_y == nullptr)) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  if (!map_x) { c->haveMaps = false; return VPL_OK; }
  const size_t PX = (size_t)c->W * c->H;
  if (!c->d_mapx) { FECHK(c, fe_alloc(c, &c->d_mapx, PX)); FECHK(c, fe_alloc(c, &c->d_mapy, PX)); }
  FECHK(c, hipMemcpyAsync(c->d_mapx, map_x, PX * 4, hipMemcpyHostToDevice, c->stream));
  FECHK(c, hipMemcpyAsync(c->d_mapy, map_y, PX * 4, hipMemcpyHostToDevice, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));   // the caller's maps may go away
  c->haveMaps = true;
  return VPL_OK;
}

int vpl_pre_upload(vpl_fe_ctx* c, int n, const uint8_t* raw) {
  if (!c || !raw || n < 1) return VPL_E_INVALID;
  if (n > c->maxN) return fe_fail(c, VPL_E_CAPACITY, "more images than max_images");
  FECHK(c, hipSetDevice(c->device));
  const size_t PX = (size_t)c->W * c->H;
  if (!c->d_raw) { FECHK(c, fe_alloc(c, &c->d_raw, (size_t)c->maxN * PX)); FECHK(c, fe_alloc(c, &c->d_mid, (size_t)c->maxN * PX)); }
  FECHK(c, hipMemcpyAsync(c->d_raw, raw, (size_t)n * PX, hipMemcpyHostToDevice, c->stream));
  c->n = n;
  c->B.N = n;
  return VPL_OK;
}

int vpl_pre_run(vpl_fe_ctx* c, int equalize, double clip_limit, int tiles_x, int tiles_y) {
  if (!c || c->n < 1 || !c->d_raw) return VPL_E_INVALID;
  if (equalize && (tiles_x < 1 || tiles_y < 1 || tiles_x > c->W || tiles_y > c->H)) return fe_fail(c, VPL_E_INVALID, "bad CLAHE grid");
  FECHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t PX = (size_t)c->W * c->H;
  PreBatch P;
  std::memset(&P, 0, sizeof(P));
  P.N = c->n; P.W = c->W; P.H = c->H;
  P.raw = c->d_raw; P.mapx = c->d_mapx; P.mapy = c->d_mapy; P.mid = c->d_mid; P.out = (uint8_t*)c->B.img;
  const int W4 = (c->W + 3) / 4;
  const dim3 gpx((W4 * c->H + 255) / 256, c->n);
  // stage 1: remap into `mid` (or straight into the frame batch when no CLAHE follows)
  const uint8_t* clahe_in = c->d_raw;
  if (c->haveMaps) {
    uint8_t* dst = equalize ? c->d_mid : P.out;
    hipLaunchKernelGGL(k_pre_remap, gpx, dim3(256), 0, s, P, dst);
    clahe_in = dst;
  } else if (!equalize) {
    FECHK(c, hipMemcpyAsync(P.out, c->d_raw, (size_t)c->n * PX, hipMemcpyDeviceToDevice, s));
  }
  if (equalize) {
    const int tiles = tiles_x * tiles_y;
    if (tiles > c->lutTiles) { FECHK(c, fe_alloc(c, &c->d_lut, (size_t)c->maxN * tiles * 256)); c->lutTiles = tiles; }
    P.lut = c->d_lut;
    int extW = c->W, extH = c->H;
    if (c->W % tiles_x != 0 || c->H % tiles_y != 0) { extW += tiles_x - c->W % tiles_x; extH += tiles_y - c->H % tiles_y; }
    P.tilesX = tiles_x; P.tilesY = tiles_y; P.tw = extW / tiles_x; P.th = extH / tiles_y;
    const int area = P.tw * P.th;
    P.lutScale = (float)255 / area;
    P.clipLimit = clip_limit > 0.0 ? std::max((int)(clip_limit * area / 256), 1) : 0;   // clahe.cpp: clipLimit_ * tileSizeTotal / histSize
    P.inv_tw = 1.0f / P.tw; P.inv_th = 1.0f / P.th;
    hipLaunchKernelGGL(k_pre_clahe_lut, dim3(tiles, c->n), dim3(256), 0, s, P, clahe_in);
    hipLaunchKernelGGL(k_pre_clahe_interp, gpx, dim3(256), 0, s, P, clahe_in);
  }
  FECHK(c, hipGetLastError());
  return VPL_OK;
}

int vpl_pre_download(vpl_fe_ctx* c, int n, uint8_t* images) {
  if (!c || !images || n < 1 || n > c->n) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipMemcpyAsync(images, c->B.img, (size_t)n * c->W * c->H, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}

int vpl_pre_batch(vpl_fe_ctx* c, int n, const uint8_t* raw, int equalize, double clip_limit, int tiles_x, int tiles_y,
                  uint8_t* images) {
  int rc = vpl_pre_upload(c, n, raw);
  if (rc) return rc;
  rc = vpl_pre_run(c, equalize, clip_limit, tiles_x, tiles_y);
  if (rc) return rc;
  return images ? vpl_pre_download(c, n, images) : vpl_fe_synchronize(c);
}

// The taps of cv::GaussianBlur's 8-bit path in 8.8 fixed point (imgproc/smooth.cpp: createGaussianKernels +
// getFixedpointGaussianKernel<ufixedpoint16>): ksize <= 0 with sigma > 0 picks cvRound(6 sigma + 1) | 1; sigma <= 0 takes the
// fixed tables (n = 1, 3, 5, 7) or sigma = 0.15 n + 0.35.  VPL_BLUR_OPENCV_341: k_i = cvRound(256 g_i) (3.4.1 - 3.4.8, 4.0 - 4.1);
// VPL_BLUR_NORMALISED: getGaussianKernelFixedPoint_ED (3.4.9+, 4.2+): the rounding error is carried from the ends towards the
// centre, the centre tap takes what is left of 256.  Returns the kernel size, or -1 (even, or larger than `cap`).
static int gauss_kernel_q8(int ksize, double sigma, int mode, int* k, int cap) {
  if (sigma < 0) sigma = 0;
  if (ksize <= 0 && sigma > 0) ksize = (int)std::nearbyint(sigma * 3 * 2 + 1) | 1;
  if (ksize <= 0 || (ksize & 1) == 0 || ksize > cap) return -1;
  const int n = ksize, h = n / 2;
  double v[16];
  if (sigma <= 0 && n <= 7) {
    static const double t1[] = {1.0}, t3[] = {0.25, 0.5, 0.25}, t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                        t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    const double* t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
    for (int i = 0; i < n; ++i) v[i] = t[i];
  } else {
    const double sx = sigma > 0 ? sigma : 0.15 * n + 0.35;
    const double scale2X = -0.5 * 0.25 / (sx * sx);
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n; ++i, x += 2) { v[i] = std::exp((double)(x * x) * scale2X); sum += v[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) v[i] *= sum;
  }
  if (mode == VPL_BLUR_OPENCV_341) {
    for (int i = 0; i < n; ++i) k[i] = (int)std::nearbyint(v[i] * 256.0);
  } else {
    double err = 0;
    int sum = 0;
    for (int i = 0; i < h; ++i) {
      const double adj = v[i] * 256.0 + err;
      const int v0 = (int)std::nearbyint(adj);
      err = adj - v0;
      k[i] = k[n - 1 - i] = v0;
      sum += v0;
    }
    k[h] = 256 - 2 * sum;
  }
  return n;
}

int vpl_fe_set_blur_kernel(vpl_fe_ctx* c, int mode) {
  if (!c || (mode != VPL_BLUR_NORMALISED && mode != VPL_BLUR_OPENCV_341)) return VPL_E_INVALID;
  c->blurMode = mode;
  return VPL_OK;
}

int vpl_fe_keep_blurred(vpl_fe_ctx* c, int enable) {
  if (!c) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  if (enable && !c->d_blur) FECHK(c, fe_alloc(c, &c->d_blur, (size_t)c->maxN * c->W * c->H));
  c->B.blurOut = enable ? c->d_blur : nullptr;
  return VPL_OK;
}

// EDLineDetector::EDline(image, lines, smoothed) for the uploaded batch (edline_detector.cpp:1176-1198 -> EdgeDrawing :81-710)
int vpl_edlines_detect_ex(vpl_fe_ctx* c, const vpl_edline_param* p, int smoothed) {
  if (!c || !p || c->n < 1) return VPL_E_INVALID;
  if (p->scanIntervals < 1 || p->minLineLen < 2) return fe_fail(c, VPL_E_INVALID, "bad EDLine parameters");
  FECHK(c, hipSetDevice(c->device));
  EdBatch& B = c->B;
  // member types of EDLineDetector: short gradienThreshold_, unsigned char anchorThreshold_ (edline_detector.h:113-117)
  B.gradTh = (int)(short)p->gradientThreshold;
  B.anchorTh = (int)(unsigned char)p->anchorThreshold;
  B.scan = p->scanIntervals;
  B.minLineLen = p->minLineLen;
  B.fitErr = p->lineFitErrThreshold;
  const int PX = c->W * c->H;
  hipStream_t s = c->stream;
  FECHK(c, hipMemsetAsync(B.nLines, 0, c->n * sizeof(int), s));
  int ksz = 1;
  if (!smoothed) {   // cv::GaussianBlur(image, image_, cv::Size(ksize_, ksize_), sigma_), edline_detector.cpp:82-84
    ksz = gauss_kernel_q8(p->ksize, (double)p->sigma, c->blurMode, B.blurK, 2 * EDB_RMAX + 1);
    if (ksz < 0) return fe_fail(c, VPL_E_INVALID, "Gaussian kernel size must be odd and at most 7");
    if (c->W < 8 || c->H < 8) return fe_fail(c, VPL_E_INVALID, "frames smaller than 8 x 8 are not supported with smoothed = false");
    B.blurR = ksz / 2;
  }
  if (smoothed || ksz == 1) {   // a 1 x 1 kernel copies (GaussianBlur's early return)
    FeTimer t(c, "k_ed_grad");
    hipLaunchKernelGGL(k_ed_grad, dim3((PX + 255) / 256, c->n), dim3(256), 0, s, B);
    if (!smoothed && B.blurOut) FECHK(c, hipMemcpyAsync(B.blurOut, B.img, (size_t)c->n * PX, hipMemcpyDeviceToDevice, s));
  } else {
    FeTimer t(c, "k_ed_blur_grad");
    hipLaunchKernelGGL(k_ed_blur_grad, dim3((c->W + EDB_TW - 1) / EDB_TW, (c->H + EDB_TH - 1) / EDB_TH, c->n), dim3(256), 0, s, B);
  }
  {
    FeTimer t(c, "k_ed_anchor");
    const int nWs = (c->W - 2 + B.scan - 1) / B.scan, nHs = (c->H - 2 + B.scan - 1) / B.scan;
    const size_t abytes = (size_t)nWs * ((nHs + 31) / 32) * 4;
    if (abytes > 60 * 1024) return fe_fail(c, VPL_E_CAPACITY, "anchor bitmask exceeds LDS (frame too large for this scanIntervals)");
    hipLaunchKernelGGL(k_ed_anchor, dim3(c->n), dim3(1024), abytes, s, B);
  }
  { FeTimer t(c, "k_ed_code"); hipLaunchKernelGGL(k_ed_code, dim3((PX + 255) / 256, c->n), dim3(256), 0, s, B); }
  { FeTimer t(c, "k_ed_route"); hipLaunchKernelGGL(k_ed_route, dim3(c->n), dim3(64 * ED_ROUTE_WAVES), c->routeSmem, s, B); }
  { FeTimer t(c, "k_ed_fit"); hipLaunchKernelGGL(k_ed_fit, dim3(ED_FIT_BLOCKS, c->n), dim3(64), 0, s, B); }
  { FeTimer t(c, "k_ed_sort_lines"); hipLaunchKernelGGL(k_ed_sort_lines, dim3(c->n), dim3(256), 0, s, B, c->d_sorted, c->d_sortedCnt, c->maxLines); }
  FECHK(c, hipGetLastError());
  return VPL_OK;
}

// the production call: smoothed = true (feature_tracker/src/line_feature_tracker.cpp:87)
int vpl_edlines_detect(vpl_fe_ctx* c, const vpl_edline_param* p) { return vpl_edlines_detect_ex(c, p, 1); }

int vpl_edlines_debug_blurred(vpl_fe_ctx* c, int img, uint8_t* out) {
  if (!c || !out || img < 0 || img >= c->n || !c->B.blurOut) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  const size_t PX = (size_t)c->W * c->H;
  FECHK(c, hipMemcpyAsync(out, c->B.blurOut + img * PX, PX, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}

// The detector's line table holds max_lines_per_image lines per frame; the reference's std::vector has no such limit.  A frame
// with more lines keeps the ones that arrived first (the order of arrival is not defined), so the result is refused, loudly,
// wherever it reaches the host: found[i] = lines the detector found in frame i.
static int lines_overflow_check(vpl_fe_ctx* c, int n) {
  std::vector<int> found(n);
  FECHK(c, hipMemcpyAsync(found.data(), c->B.nLines, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i)
    if (found[i] > c->maxLines)
      return fe_fail(c, VPL_E_CAPACITY, "frame " + std::to_string(i) + ": " + std::to_string(found[i]) + " lines found, max_lines_per_image is " +
                                            std::to_string(c->maxLines));
  return VPL_OK;
}

int vpl_edlines_download(vpl_fe_ctx* c, int n, vpl_line* lines, int* counts) {
  if (!c || n != c->n || !lines || !counts) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  const size_t ML = c->maxLines;
  // (the deterministic order -- edge chain, ordinal inside the chain -- and the vpl_line layout are made on the device by
  // k_ed_sort_lines; rows beyond counts[i] are not defined)
  FECHK(c, hipMemcpyAsync(counts, c->d_sortedCnt, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(lines, c->d_sorted, (size_t)n * ML * sizeof(vpl_line), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  return lines_overflow_check(c, n);
}

int vpl_edlines_detect_batch_ex(vpl_fe_ctx* c, int n, const uint8_t* images, const vpl_edline_param* p, int smoothed,
                                vpl_line* lines, int* counts) {
  int rc = vpl_edlines_upload(c, n, images);
  if (rc) return rc;
  rc = vpl_edlines_detect_ex(c, p, smoothed);
  if (rc) return rc;
  rc = vpl_fe_synchronize(c);
  if (rc) return rc;
  return vpl_edlines_download(c, n, lines, counts);
}
int vpl_edlines_detect_batch(vpl_fe_ctx* c, int n, const uint8_t* images, const vpl_edline_param* p, vpl_line* lines,
                             int* counts) {
  return vpl_edlines_detect_batch_ex(c, n, images, p, 1, lines, counts);
}

// LineMatching::LineFilter (line_matching.cpp:167-264) on the lines of the last detect, where they lie (asynchronous):
// vpl_edlines_download and vpl_match_from_detected then see the filtered lists
int vpl_line_filter_detected(vpl_fe_ctx* c, float distance_threshold, float parallel_threshold) {
  if (!c || c->n < 1) return VPL_E_INVALID;
  if (c->maxLines > 8192) return fe_fail(c, VPL_E_INVALID, "LineFilter: max_lines_per_image above 8192");
  FECHK(c, hipSetDevice(c->device));
  FeTimer t(c, "k_lm_line_filter");
  hipLaunchKernelGGL(k_lm_line_filter, dim3(c->n), dim3(256), (size_t)c->maxLines * 8, c->stream, c->d_sorted, c->d_sortedCnt,
                     c->maxLines, distance_threshold, parallel_threshold);
  FECHK(c, hipGetLastError());
  return VPL_OK;
}

// the same for caller-owned line lists [n][max_lines] (in / out), through the context's line table
int vpl_line_filter_batch(vpl_fe_ctx* c, int n, vpl_line* lines, int* counts, float distance_threshold, float parallel_threshold) {
  if (!c || !lines || !counts || n < 1 || n > c->maxN) return VPL_E_INVALID;
  if (c->maxLines > 8192) return fe_fail(c, VPL_E_INVALID, "LineFilter: max_lines_per_image above 8192");
  for (int i = 0; i < n; ++i)
    if (counts[i] < 0 || counts[i] > c->maxLines) return fe_fail(c, VPL_E_INVALID, "line count beyond max_lines_per_image");
  FECHK(c, hipSetDevice(c->device));
  const size_t ML = c->maxLines;
  FECHK(c, hipMemcpyAsync(c->d_sorted, lines, (size_t)n * ML * sizeof(vpl_line), hipMemcpyHostToDevice, c->stream));
  FECHK(c, hipMemcpyAsync(c->d_sortedCnt, counts, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_lm_line_filter, dim3(n), dim3(256), ML * 8, c->stream, c->d_sorted, c->d_sortedCnt, (int)ML,
                     distance_threshold, parallel_threshold);
  FECHK(c, hipGetLastError());
  FECHK(c, hipMemcpyAsync(counts, c->d_sortedCnt, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(lines, c->d_sorted, (size_t)n * ML * sizeof(vpl_line), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}

int vpl_edlines_debug_stage(vpl_fe_ctx* c, int img, int16_t* dx, int16_t* dy, int16_t* g, uint8_t* dir, uint32_t* anchors,
                            int* n_anchors, uint32_t* chain_x, uint32_t* chain_y, uint32_t* sId, int* n_edges) {
  if (!c || img < 0 || img >= c->n) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipStreamSynchronize(c->stream));
  const EdBatch& B = c->B;
  const size_t PX = (size_t)c->W * c->H;
  if (dx) FECHK(c, hipMemcpy(dx, B.dx + img * PX, PX * 2, hipMemcpyDeviceToHost));
  if (dy) FECHK(c, hipMemcpy(dy, B.dy + img * PX, PX * 2, hipMemcpyDeviceToHost));
  if (g) FECHK(c, hipMemcpy(g, B.g + img * PX, PX * 2, hipMemcpyDeviceToHost));
  if (dir) FECHK(c, hipMemcpy(dir, B.dir + img * PX, PX, hipMemcpyDeviceToHost));
  int nA = 0, nE = 0;
  FECHK(c, hipMemcpy(&nA, B.nAnch + img, 4, hipMemcpyDeviceToHost));
  FECHK(c, hipMemcpy(&nE, B.nEdges + img, 4, hipMemcpyDeviceToHost));
  if (n_anchors) *n_anchors = nA;
  if (n_edges) *n_edges = nE;
  if (anchors) {
    std::vector<uint32_t> ax(nA), ay(nA);
    FECHK(c, hipMemcpy(ax.data(), B.anchX + (size_t)img * B.cap, nA * 4, hipMemcpyDeviceToHost));
    FECHK(c, hipMemcpy(ay.data(), B.anchY + (size_t)img * B.cap, nA * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < nA; ++i) { anchors[2 * i] = ax[i]; anchors[2 * i + 1] = ay[i]; }
  }
  if (sId) FECHK(c, hipMemcpy(sId, B.sId + (size_t)img * (B.capEdges + 2), (nE + 1) * 4, hipMemcpyDeviceToHost));
  if (chain_x && chain_y) {
    uint32_t npx = 0;
    FECHK(c, hipMemcpy(&npx, B.sId + (size_t)img * (B.capEdges + 2) + nE, 4, hipMemcpyDeviceToHost));
    FECHK(c, hipMemcpy(chain_x, B.cX + (size_t)img * 2 * B.cap, npx * 4, hipMemcpyDeviceToHost));
    FECHK(c, hipMemcpy(chain_y, B.cY + (size_t)img * 2 * B.cap, npx * 4, hipMemcpyDeviceToHost));
  }
  return VPL_OK;
}


int vpl_edlines_debug_route_stats(vpl_fe_ctx* c, int img, unsigned long long* out4) {
  if (!c || img < 0 || img >= c->n || !out4) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipStreamSynchronize(c->stream));
  FECHK(c, hipMemcpy(out4, c->B.rstats + (size_t)img * 4, 32, hipMemcpyDeviceToHost));
  return VPL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// KLT line matching
// ---------------------------------------------------------------------------------------------------------------
void vpl_match_default_param(vpl_match_param* p) {
  p->step = 10; p->closest_line_threshold = 0.5f; p->line_matching_ratio = 0.4f; p->line_distance_error_ratio = 3.f;
  p->klt_error_threshold = 40.f; p->illumination_adapt = 1; p->topological_filter = 1;
  p->topo_distance_threshold = 15.f; p->topo_length_tolerate_ratio = 0.2f; p->topo_violation_ratio = 0.05f;
}

int vpl_match_reserve(vpl_fe_ctx* c, int max_pairs, int max_kps) {
  if (!c || max_pairs < 1 || max_kps < 1) return VPL_E_INVALID;
  if (c->lmReserved) return fe_fail(c, VPL_E_INVALID, "vpl_match_reserve called twice");
  if (c->W <= LM_WIN || c->H <= LM_WIN) return fe_fail(c, VPL_E_INVALID, "image smaller than the KLT window");
  FECHK(c, hipSetDevice(c->device));
  LmBatch& M = c->M;
  std::memset(&M, 0, sizeof(M));
  M.W = c->W; M.H = c->H; M.img = c->B.img;
  // buildOpticalFlowPyramid: halve until maxLevel or until the next level would not exceed the window
  int w = c->W, h = c->H, l = 0;
  size_t off = 0;
  for (;;) {
    M.lw[l] = w; M.lh[l] = h; M.ls[l] = w + 2 * LM_WIN; M.loff[l] = off;
    off += (size_t)(w + 2 * LM_WIN) * (h + 2 * LM_WIN);
    off = (off + 63) & ~(size_t)63;
    if (l == LM_LEVELS - 1) break;
    const int nw = (w + 1) / 2, nh = (h + 1) / 2;
    if (nw <= LM_WIN || nh <= LM_WIN) break;
    w = nw; h = nh; ++l;
  }
  M.nLevels = l + 1;
  M.pyrSize = off;
  M.maxLines = c->maxLines; M.maxK = max_kps;
  c->maxPairs = max_pairs;
  const size_t N = c->maxN, P = max_pairs, ML = c->maxLines, MK = max_kps;
  hipError_t e = hipSuccess;
#define AL(ptr, n) if (e == hipSuccess) e = fe_alloc(c, &ptr, (size_t)(n))
  AL(M.pyr, N * M.pyrSize); AL(M.der, N * M.pyrSize * 2);
  AL(c->d_refImg, P); AL(c->d_curImg, P); AL(c->d_nRef, P); AL(c->d_nCur, P);
  AL(c->d_linesRef, P * ML); AL(c->d_linesCur, P * ML);
  AL(M.kpsRef, P * MK); AL(M.kpsCur, P * MK); AL(M.status, P * MK); AL(M.err, P * MK); AL(M.kp2lineCur, P * MK);
  AL(M.kpOff, P * ML); AL(M.kpNum, P * ML); AL(M.nK, P); AL(M.r2c, P * ML); AL(M.valid, P);
  AL(M.chunkOff, P + 1); AL(M.workCounter, 1); AL(M.winScratch, (size_t)LM_KLT_GRID * LM_NPX * 64);
#undef AL
  if (e != hipSuccess) return fe_fail(c, VPL_E_HIP, std::string("vpl_match_reserve: ") + hipGetErrorString(e));
  M.refImg = c->d_refImg; M.curImg = c->d_curImg; M.nRef = c->d_nRef; M.nCur = c->d_nCur;
  M.linesRef = c->d_linesRef; M.linesCur = c->d_linesCur;
  c->lmReserved = true;
  return VPL_OK;
}

int vpl_match_upload(vpl_fe_ctx* c, int n_pairs, const int* ref_image, const int* cur_image, const vpl_line* lines_ref,
                     const int* n_ref, const vpl_line* lines_cur, const int* n_cur) {
  if (!c || !c->lmReserved || n_pairs < 1 || !ref_image || !cur_image || !lines_ref || !n_ref || !lines_cur || !n_cur)
    return VPL_E_INVALID;
  if (n_pairs > c->maxPairs) return fe_fail(c, VPL_E_CAPACITY, "more pairs than max_pairs");
  if (c->n < 1) return fe_fail(c, VPL_E_INVALID, "no images uploaded");
  c->matchFromDetected = false;
  for (int i = 0; i < n_pairs; ++i) {
    if (ref_image[i] < 0 || ref_image[i] >= c->n || cur_image[i] < 0 || cur_image[i] >= c->n)
      return fe_fail(c, VPL_E_INVALID, "pair names an image that was not uploaded");
    if (n_ref[i] < 0 || n_ref[i] > c->maxLines || n_cur[i] < 0 || n_cur[i] > c->maxLines)
      return fe_fail(c, VPL_E_CAPACITY, "more lines than max_lines_per_image");
  }
  FECHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t P = n_pairs, ML = c->maxLines;
  FECHK(c, hipMemcpyAsync(c->d_refImg, ref_image, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_curImg, cur_image, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_nRef, n_ref, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_nCur, n_cur, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_linesRef, lines_ref, P * ML * sizeof(vpl_line), hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_linesCur, lines_cur, P * ML * sizeof(vpl_line), hipMemcpyHostToDevice, s));
  FECHK(c, hipStreamSynchronize(s));   // the caller's buffers may be pageable and short-lived
  c->nPairs = n_pairs;
  return VPL_OK;
}

// The lines of the last vpl_edlines_detect of this context as the matcher's input, device to device (LineFeatureTracker::readImage
// keeps the previous frame's lines and matches the new frame's against them: line_feature_tracker.cpp:110-118, :290-322): no download /
// upload of the lines between the two stages.  At most max_lines lines per frame take part (the first ones in the
// detector's order).  Asynchronous on the context's stream.
int vpl_match_from_detected(vpl_fe_ctx* c, int n_pairs, const int* ref_image, const int* cur_image, int max_lines) {
  if (!c || !c->lmReserved || n_pairs < 1 || !ref_image || !cur_image || max_lines < 1) return VPL_E_INVALID;
  if (n_pairs > c->maxPairs) return fe_fail(c, VPL_E_CAPACITY, "more pairs than max_pairs");
  if (c->n < 1) return fe_fail(c, VPL_E_INVALID, "no images uploaded");
  for (int i = 0; i < n_pairs; ++i)
    if (ref_image[i] < 0 || ref_image[i] >= c->n || cur_image[i] < 0 || cur_image[i] >= c->n)
      return fe_fail(c, VPL_E_INVALID, "pair names an image that was not uploaded");
  c->matchFromDetected = true;
  FECHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t P = n_pairs;
  FECHK(c, hipMemcpyAsync(c->d_refImg, ref_image, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_curImg, cur_image, P * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipStreamSynchronize(s));   // (two small index arrays; the caller's buffers may be short-lived)
  hipLaunchKernelGGL(k_lm_take_lines, dim3(n_pairs, 2), dim3(256), 0, s, c->d_sorted, c->d_sortedCnt, c->maxLines,
                     std::min(max_lines, c->maxLines), c->d_refImg, c->d_curImg, c->d_linesRef, c->d_linesCur, c->d_nRef, c->d_nCur);
  FECHK(c, hipGetLastError());
  c->nPairs = n_pairs;
  return VPL_OK;
}
// number of lines of every pair's two frames that take part in the match (after vpl_match_upload / vpl_match_from_detected)
int vpl_match_counts(vpl_fe_ctx* c, int n_pairs, int* n_ref, int* n_cur) {
  if (!c || !c->lmReserved || n_pairs != c->nPairs || !n_ref || !n_cur) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipMemcpyAsync(n_ref, c->d_nRef, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(n_cur, c->d_nCur, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  return VPL_OK;
}

int vpl_match_run(vpl_fe_ctx* c, const vpl_match_param* p) {
  if (!c || !p || !c->lmReserved || c->nPairs < 1) return VPL_E_INVALID;
  if (p->step < 1) return fe_fail(c, VPL_E_INVALID, "bad LineMatching parameters");
  FECHK(c, hipSetDevice(c->device));
  LmBatch& M = c->M;
  M.N = c->n; M.nPairs = c->nPairs; M.prm = *p;
  double eps = std::min(std::max(0.001, 0.), 10.);   // TermCriteria(COUNT|EPS, 30, 0.001) through KLT::KLT, klt.cpp:28-33
  M.epsilon = eps * eps;
  hipStream_t s = c->stream;
  auto blocks = [&](int l) { return (unsigned)(((size_t)M.ls[l] * (M.lh[l] + 2 * LM_WIN) + 255) / 256); };
  { FeTimer t(c, "k_lm_pyramid");
    hipLaunchKernelGGL(k_lm_level0, dim3(blocks(0), c->n), dim3(256), 0, s, M);
    for (int l = 1; l < M.nLevels; ++l) hipLaunchKernelGGL(k_lm_down, dim3(blocks(l), c->n), dim3(256), 0, s, M, l); }
  { FeTimer t(c, "k_lm_scharr"); hipLaunchKernelGGL(k_lm_scharr, dim3(blocks(0), c->n, M.nLevels), dim3(256), 0, s, M); }
  { FeTimer t(c, "k_lm_anchors"); hipLaunchKernelGGL(k_lm_anchors, dim3(c->nPairs), dim3(256), (size_t)M.maxLines * sizeof(int), s, M);
    hipLaunchKernelGGL(k_lm_plan, dim3(1), dim3(64), 0, s, M); }
  { FeTimer t(c, "k_lm_klt"); hipLaunchKernelGGL(k_lm_klt, dim3(LM_KLT_GRID), dim3(64), LM_KLT_SMEM, s, M); }
  { FeTimer t(c, "k_lm_vote"); hipLaunchKernelGGL(k_lm_vote, dim3(c->nPairs), dim3(256), (2 * M.maxLines + 1) * sizeof(int), s, M); }
  FECHK(c, hipGetLastError());
  return VPL_OK;
}

int vpl_match_download(vpl_fe_ctx* c, int n_pairs, int* r2c, int* matched) {
  if (!c || !c->lmReserved || n_pairs != c->nPairs || !r2c || !matched) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  const size_t ML = c->maxLines;
  std::vector<int> valid(n_pairs), nref(n_pairs), buf((size_t)n_pairs * ML);
  FECHK(c, hipMemcpyAsync(valid.data(), c->M.valid, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(nref.data(), c->d_nRef, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(buf.data(), c->M.r2c, buf.size() * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  if (c->matchFromDetected) { const int ro = lines_overflow_check(c, c->n); if (ro) return ro; }
  for (int i = 0; i < n_pairs; ++i)
    if (valid[i] < 0) return fe_fail(c, VPL_E_CAPACITY, "pair " + std::to_string(i) + ": more key points than max_kps");
  for (int i = 0; i < n_pairs; ++i) {
    matched[i] = valid[i];
    if (valid[i] == 1) std::memcpy(r2c + (size_t)i * ML, buf.data() + (size_t)i * ML, (size_t)nref[i] * 4);
  }
  return VPL_OK;
}

int vpl_line_match_batch(vpl_fe_ctx* c, int n_images, const uint8_t* images, int n_pairs, const int* ref_image,
                         const int* cur_image, const vpl_line* lines_ref, const int* n_ref, const vpl_line* lines_cur,
                         const int* n_cur, const vpl_match_param* param, int* r2c, int* matched) {
  int rc = vpl_edlines_upload(c, n_images, images);
  if (rc) return rc;
  rc = vpl_match_upload(c, n_pairs, ref_image, cur_image, lines_ref, n_ref, lines_cur, n_cur);
  if (rc) return rc;
  rc = vpl_match_run(c, param);
  if (rc) return rc;
  rc = vpl_fe_synchronize(c);
  if (rc) return rc;
  return vpl_match_download(c, n_pairs, r2c, matched);
}

int vpl_match_debug_kps(vpl_fe_ctx* c, int pair, int cap, float* kps_ref, float* kps_cur, uint8_t* status, float* err,
                        int* kp2line_cur, int* n_kps) {
  if (!c || !c->lmReserved || pair < 0 || pair >= c->nPairs) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipStreamSynchronize(c->stream));
  const LmBatch& M = c->M;
  int nk = 0;
  FECHK(c, hipMemcpy(&nk, M.nK + pair, 4, hipMemcpyDeviceToHost));
  if (n_kps) *n_kps = nk;
  const size_t m = (size_t)std::min(nk, cap), o = (size_t)pair * M.maxK;
  if (m == 0) return VPL_OK;
  if (kps_ref) FECHK(c, hipMemcpy(kps_ref, M.kpsRef + o, m * 8, hipMemcpyDeviceToHost));
  if (kps_cur) FECHK(c, hipMemcpy(kps_cur, M.kpsCur + o, m * 8, hipMemcpyDeviceToHost));
  if (status) FECHK(c, hipMemcpy(status, M.status + o, m, hipMemcpyDeviceToHost));
  if (err) FECHK(c, hipMemcpy(err, M.err + o, m * 4, hipMemcpyDeviceToHost));
  if (kp2line_cur) FECHK(c, hipMemcpy(kp2line_cur, M.kp2lineCur + o, m * 4, hipMemcpyDeviceToHost));
  return VPL_OK;
}

int vpl_match_debug_level(vpl_fe_ctx* c, int img, int level, uint8_t* pixels, int16_t* deriv, int* w, int* h) {
  if (!c || !c->lmReserved || img < 0 || img >= c->n || level < 0 || level >= c->M.nLevels) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipStreamSynchronize(c->stream));
  const LmBatch& M = c->M;
  const int lw = M.lw[level], lh = M.lh[level], S = M.ls[level];
  if (w) *w = lw;
  if (h) *h = lh;
  const size_t base = (size_t)img * M.pyrSize + M.loff[level] + (size_t)LM_WIN * S + LM_WIN;
  if (pixels) FECHK(c, hipMemcpy2D(pixels, lw, M.pyr + base, S, lw, lh, hipMemcpyDeviceToHost));
  if (deriv) FECHK(c, hipMemcpy2D(deriv, (size_t)lw * 4, M.der + base * 2, (size_t)S * 4, (size_t)lw * 4, lh, hipMemcpyDeviceToHost));
  return VPL_OK;
}

// ---- vanishing points (vanishing_point_detection.cpp) ----
static int vp_reserve(vpl_fe_ctx* c) {
  if (c->vpReserved) return VPL_OK;
  if (c->maxLines > 1024) return fe_fail(c, VPL_E_CAPACITY, "vanishing points: max_lines_per_image above 1024");
  const size_t N = c->maxN, ML = c->maxLines;
  VpBatch& V = c->V;
  std::memset(&V, 0, sizeof(V));
  FECHK(c, fe_alloc(c, &c->d_vpHyp, N * ML * 4)); FECHK(c, fe_alloc(c, &c->d_vpAll, N * ML * 4));
  FECHK(c, fe_alloc(c, &c->d_vpNHyp, N)); FECHK(c, fe_alloc(c, &c->d_vpNAll, N)); FECHK(c, fe_alloc(c, &c->d_vpFirst, N));
  FECHK(c, fe_alloc(c, &c->d_vpSeed, N));
  FECHK(c, fe_alloc(c, &V.g, N * VP_CELLS)); FECHK(c, fe_alloc(c, &V.grid, N * VP_CELLS));
  FECHK(c, fe_alloc(c, &V.pairs, N * VP_IT * 2)); FECHK(c, fe_alloc(c, &V.rng, N * 36)); FECHK(c, fe_alloc(c, &V.status, N));
  FECHK(c, fe_alloc(c, &V.partScore, N * VP_SCORE_BLOCKS)); FECHK(c, fe_alloc(c, &V.partIdx, N * VP_SCORE_BLOCKS));
  FECHK(c, fe_alloc(c, &V.vps, N * 9)); FECHK(c, fe_alloc(c, &V.ids, N * ML)); FECHK(c, fe_alloc(c, &V.bestIdx, N));
  V.maxL = (int)ML;
  V.hypEnds = c->d_vpHyp; V.allEnds = c->d_vpAll; V.nHyp = c->d_vpNHyp; V.nAll = c->d_vpNAll; V.seed = c->d_vpSeed;
  V.firstFrame = c->d_vpFirst;
  c->vpReserved = true;
  return VPL_OK;
}

int vpl_vp_detect_batch(vpl_fe_ctx* c, int n, const vpl_line* hyp_lines, const int* n_hyp, const vpl_line* all_lines,
                        const int* n_all, float f, float cx, float cy, const uint32_t* seeds, const int* first_frame, double* vps,
                        int* vp_ids, int* status) {
  if (!c || n < 1 || !hyp_lines || !n_hyp || !all_lines || !n_all || !seeds || !first_frame || !vps || !vp_ids || !status)
    return VPL_E_INVALID;
  if (n > c->maxN) return fe_fail(c, VPL_E_CAPACITY, "more frames than max_images");
  FECHK(c, hipSetDevice(c->device));
  int rc = vp_reserve(c);
  if (rc) return rc;
  const size_t ML = c->maxLines;
  for (int i = 0; i < n; ++i)
    if (n_hyp[i] < 0 || n_all[i] < 0 || n_hyp[i] > (int)ML || n_all[i] > (int)ML) return fe_fail(c, VPL_E_CAPACITY, "more lines than max_lines");
  std::vector<float> eh((size_t)n * ML * 4, 0.f), ea((size_t)n * ML * 4, 0.f);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < n_hyp[i]; ++k) std::memcpy(&eh[((size_t)i * ML + k) * 4], hyp_lines[(size_t)i * ML + k].line_endpoint, 16);
    for (int k = 0; k < n_all[i]; ++k) std::memcpy(&ea[((size_t)i * ML + k) * 4], all_lines[(size_t)i * ML + k].line_endpoint, 16);
  }
  hipStream_t s = c->stream;
  VpBatch& V = c->V;
  V.N = n; V.f = f; V.ppx = cx; V.ppy = cy;
  FECHK(c, hipMemcpyAsync(c->d_vpHyp, eh.data(), eh.size() * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_vpAll, ea.data(), ea.size() * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_vpNHyp, n_hyp, (size_t)n * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_vpNAll, n_all, (size_t)n * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_vpSeed, seeds, (size_t)n * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemcpyAsync(c->d_vpFirst, first_frame, (size_t)n * 4, hipMemcpyHostToDevice, s));
  FECHK(c, hipMemsetAsync(V.g, 0, (size_t)n * VP_CELLS * 8, s));
  hipLaunchKernelGGL(k_vp_grid, dim3(n, 2), dim3(64), 0, s, V);
  hipLaunchKernelGGL(k_vp_smooth, dim3((VP_CELLS + 255) / 256, n), dim3(256), 0, s, V);
  hipLaunchKernelGGL(k_vp_score, dim3(VP_SCORE_BLOCKS, n), dim3(256), 0, s, V);
  const size_t pickSmem = ML * (3 * 8 + 4 + 3 * 4);
  hipLaunchKernelGGL(k_vp_pick, dim3(n), dim3(64), pickSmem, s, V);
  FECHK(c, hipGetLastError());
  FECHK(c, hipMemcpyAsync(vps, V.vps, (size_t)n * 72, hipMemcpyDeviceToHost, s));
  FECHK(c, hipMemcpyAsync(status, V.status, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  std::vector<int> ids((size_t)n * ML);
  FECHK(c, hipMemcpyAsync(ids.data(), V.ids, ids.size() * 4, hipMemcpyDeviceToHost, s));
  FECHK(c, hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) std::memcpy(vp_ids + (size_t)i * ML, &ids[(size_t)i * ML], (size_t)n_all[i] * 4);
  c->vpN = n;
  return VPL_OK;
}

int vpl_vp_debug(vpl_fe_ctx* c, int frame, double* grid, int* pairs, int* best_idx, int* drawn) {
  if (!c || !c->vpReserved || frame < 0 || frame >= c->vpN) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipStreamSynchronize(c->stream));
  const VpBatch& V = c->V;
  if (grid) FECHK(c, hipMemcpy(grid, V.grid + (size_t)frame * VP_CELLS, VP_CELLS * 8, hipMemcpyDeviceToHost));
  if (pairs) FECHK(c, hipMemcpy(pairs, V.pairs + (size_t)frame * VP_IT * 2, VP_IT * 2 * 4, hipMemcpyDeviceToHost));
  if (best_idx) FECHK(c, hipMemcpy(best_idx, V.bestIdx + frame, 4, hipMemcpyDeviceToHost));
  if (drawn) FECHK(c, hipMemcpy(drawn, V.rng + (size_t)frame * 36 + 35, 4, hipMemcpyDeviceToHost));
  return VPL_OK;
}

// LineFeatureTracker::readImage, the lists after the match (line_feature_tracker.cpp:109-229).  Plain host code, as in the
// reference: a few hundred integers per frame.
int vpl_line_track_ids(int n_new, const float* ends, int n_prev, const int* id_prev, const int* tcnt_prev, int n_tcnt_prev,
                       const int* prev_to_new, int max_h, int max_v, int* allfeature_cnt, int* keep, int* id_out,
                       int* tcnt_out, int* vertical_new, int* n_vertical_new) {
  if (n_new < 0 || n_prev < 0 || !allfeature_cnt || (n_new && (!ends || !keep || !id_out || !tcnt_out)) ||
      (n_prev && (!id_prev || !prev_to_new)) || (n_tcnt_prev && !tcnt_prev))
    return VPL_E_INVALID;
  std::vector<int> id(n_new, -1);
  for (int i = 0; i < n_new; ++i) tcnt_out[i] = 0;
  for (int k = 0; k < n_prev; ++k) {
    const int mt = prev_to_new[k];
    if (mt <= 0 || mt >= n_new) continue;                     // `if (mt > 0)`: detection 0 never inherits an id
    id[mt] = id_prev[k];
    tcnt_out[mt] = (mt < n_tcnt_prev ? tcnt_prev[mt] : 0) + 1;
  }
  // segAngle (:20-25) and the "h" class of :166 / :185
  auto horizontal_class = [&](int i) {
    const float* e = ends + 4 * i;
    const double a = e[2] > e[0] ? std::atan2(e[3] - e[1], e[2] - e[0]) : std::atan2(e[1] - e[3], e[0] - e[2]);
    return (a >= 3.14 / 4.0 && a <= 3 * 3.14 / 4.0) || (a <= -3.14 / 4.0 && a >= -3 * 3.14 / 4.0);
  };
  int n_keep = 0, h_tracked = 0, v_tracked = 0;
  std::vector<int> fresh_h, fresh_v;
  for (int i = 0; i < n_new; ++i) {
    if (id[i] == -1) {
      id[i] = (*allfeature_cnt)++;
      (horizontal_class(i) ? fresh_h : fresh_v).push_back(i);
    } else {
      keep[n_keep] = i; id_out[n_keep] = id[i]; ++n_keep;
      (horizontal_class(i) ? h_tracked : v_tracked)++;
    }
  }
  const int take_h = std::min<int>(std::max(max_h - h_tracked, 0), (int)fresh_h.size());
  const int take_v = std::min<int>(std::max(max_v - v_tracked, 0), (int)fresh_v.size());
  for (int k = 0; k < take_h; ++k, ++n_keep) { keep[n_keep] = fresh_h[k]; id_out[n_keep] = id[fresh_h[k]]; }
  for (int k = 0; k < take_v; ++k, ++n_keep) { keep[n_keep] = fresh_v[k]; id_out[n_keep] = id[fresh_v[k]]; }
  if (vertical_new && n_vertical_new) {                       // verticalLine of :151-176: every new line of the v class
    for (size_t k = 0; k < fresh_v.size(); ++k) vertical_new[k] = fresh_v[k];
    *n_vertical_new = (int)fresh_v.size();
  }
  return n_keep;
}

}  // extern "C"
