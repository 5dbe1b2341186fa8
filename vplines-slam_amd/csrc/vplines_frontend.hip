// C ABI of the MI355X-native line front-end (include/vplines_frontend.h): EDLines extractor.
// No CPU compute path: every call that computes launches HIP kernels or returns an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "vplines_ba.h"        // VPL_E_* codes
#include "vplines_frontend.h"
#include "ed_kernels.h"

using namespace vpl;

struct vpl_fe_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int maxN = 0, W = 0, H = 0, maxLines = 0;
  int n = 0;
  EdBatch B;
  std::vector<void*> allocs;
  std::string err;
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
  *p = (T*)q;
  return hipMemset(q, 0, n * sizeof(T) + 64);
}

extern "C" {

void vpl_edline_default_param(vpl_edline_param* p) {
  p->ksize = 5; p->sigma = 1.0f; p->gradientThreshold = 30.f; p->anchorThreshold = 5.f; p->scanIntervals = 2;
  p->minLineLen = 35; p->lineFitErrThreshold = 1.8;
}

int vpl_fe_create(vpl_fe_ctx** out, int device, int max_images, int width, int height, int max_lines_per_image) {
  if (!out || max_images < 1 || width < 8 || height < 8 || max_lines_per_image < 1) return VPL_E_INVALID;
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0 || device >= nd) return VPL_E_NODEVICE;
  if (hipSetDevice(device) != hipSuccess) return VPL_E_NODEVICE;
  vpl_fe_ctx* c = new vpl_fe_ctx();
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
  AL(B.dx, N * PX); AL(B.dy, N * PX); AL(B.g, N * PX); AL(B.dir, N * PX); AL(B.edge, N * PX);
  AL(B.anchX, N * B.cap); AL(B.anchY, N * B.cap); AL(B.nAnch, N);
  AL(B.fX, N * B.cap); AL(B.fY, N * B.cap); AL(B.cX, N * 2 * B.cap); AL(B.cY, N * 2 * B.cap);
  AL(B.sId, N * (B.capEdges + 2)); AL(B.nEdges, N);
  AL(B.lines, N * B.maxLines * 10); AL(B.lkey, N * B.maxLines); AL(B.nLines, N);
#undef AL
  if (e != hipSuccess) {
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return VPL_E_HIP;
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
int vpl_fe_set_stream(vpl_fe_ctx* c, void* s) { if (!c) return VPL_E_INVALID; c->stream = (hipStream_t)s; return VPL_OK; }
int vpl_fe_synchronize(vpl_fe_ctx* c) { if (!c) return VPL_E_INVALID; FECHK(c, hipStreamSynchronize(c->stream)); return VPL_OK; }
const char* vpl_fe_last_error(const vpl_fe_ctx* c) { return c ? c->err.c_str() : "null context"; }

int vpl_edlines_upload(vpl_fe_ctx* c, int n, const uint8_t* images) {
  if (!c || !images || n < 1) return VPL_E_INVALID;
  if (n > c->maxN) return fe_fail(c, VPL_E_CAPACITY, "more images than max_images");
  FECHK(c, hipSetDevice(c->device));
  FECHK(c, hipMemcpyAsync((void*)c->B.img, images, (size_t)n * c->W * c->H, hipMemcpyHostToDevice, c->stream));
  c->n = n;
  c->B.N = n;
  return VPL_OK;
}

int vpl_edlines_detect(vpl_fe_ctx* c, const vpl_edline_param* p) {
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
  hipLaunchKernelGGL(k_ed_grad, dim3((PX + 255) / 256, c->n), dim3(256), 0, s, B);
  hipLaunchKernelGGL(k_ed_anchor, dim3(c->n), dim3(1024), 0, s, B);
  hipLaunchKernelGGL(k_ed_route, dim3(c->n), dim3(64), 0, s, B);
  hipLaunchKernelGGL(k_ed_fit, dim3((B.capEdges + 63) / 64, c->n), dim3(64), 0, s, B);
  FECHK(c, hipGetLastError());
  return VPL_OK;
}

int vpl_edlines_download(vpl_fe_ctx* c, int n, vpl_line* lines, int* counts) {
  if (!c || n != c->n || !lines || !counts) return VPL_E_INVALID;
  FECHK(c, hipSetDevice(c->device));
  const size_t ML = c->maxLines;
  std::vector<double> L((size_t)n * ML * 10);
  std::vector<uint32_t> K((size_t)n * ML);
  std::vector<int> cnt(n);
  FECHK(c, hipMemcpyAsync(cnt.data(), c->B.nLines, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(L.data(), c->B.lines, L.size() * 8, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipMemcpyAsync(K.data(), c->B.lkey, K.size() * 4, hipMemcpyDeviceToHost, c->stream));
  FECHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) {
    const int m = std::min<int>(cnt[i], (int)ML);
    counts[i] = m;
    // deterministic order: (edge chain, ordinal inside the chain) -- plumbing, the lines themselves come from the device
    std::vector<int> ord(m);
    for (int k = 0; k < m; ++k) ord[k] = k;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return K[i * ML + a] < K[i * ML + b]; });
    for (int k = 0; k < m; ++k) {
      const double* o = &L[((size_t)i * ML + ord[k]) * 10];
      vpl_line& dst = lines[(size_t)i * ML + k];
      for (int q = 0; q < 4; ++q) dst.line_endpoint[q] = (float)o[q];
      for (int q = 0; q < 3; ++q) dst.line_equation[q] = o[4 + q];
      dst.center[0] = (float)o[7]; dst.center[1] = (float)o[8];
      dst.length = (float)o[9];
    }
  }
  return VPL_OK;
}

int vpl_edlines_detect_batch(vpl_fe_ctx* c, int n, const uint8_t* images, const vpl_edline_param* p, vpl_line* lines,
                             int* counts) {
  int rc = vpl_edlines_upload(c, n, images);
  if (rc) return rc;
  rc = vpl_edlines_detect(c, p);
  if (rc) return rc;
  rc = vpl_fe_synchronize(c);
  if (rc) return rc;
  return vpl_edlines_download(c, n, lines, counts);
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

}  // extern "C"
