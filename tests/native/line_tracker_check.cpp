// GPU test program for the LineFeatureTracker mirror (vplines-slam_amd/host/vpl_frontend.hpp): readImage over a short
// sequence of raw frames written by tests/test_gpu_host_adapter.py.  For every frame it prints the prepared image's
// checksum, the kept lines, their ids, t_cnt, the match vector and the estimator-side observations, so the Python side
// can replay each step with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../vplines-slam_amd/host/vpl_frontend.hpp"

using namespace vplhost;

template <typename T>
static std::vector<T> slurp(const char* path, size_t n) {
  std::vector<T> b(n);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(b.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
  std::fclose(f);
  return b;
}

int main(int argc, char** argv) {
  // frames.raw n W H mapx.f32 mapy.f32 max_h max_v
  if (argc < 9) return 2;
  const int n = std::atoi(argv[2]), W = std::atoi(argv[3]), H = std::atoi(argv[4]);
  const size_t px = (size_t)W * H;
  std::vector<uint8_t> frames = slurp<uint8_t>(argv[1], px * n);
  std::vector<float> mx = slurp<float>(argv[5], px), my = slurp<float>(argv[6], px);
  FrontendDevice dev(W, H, 1024, 16384);
  EDLineParam param = {5, 1.0f, 30.f, 5.f, 2, 35, 1.8};
  LineFeatureTracker tracker(dev, param, std::atoi(argv[7]), std::atoi(argv[8]), true);
  uint32_t next_seed = 1000;
  tracker.vp_seed = [&next_seed] { return next_seed++; };
  tracker.setUndistortMaps(mx.data(), my.data(), 458.654f, 457.296f, (float)(W / 2), (float)(H / 2));
  for (int f = 0; f < n; ++f) {
    tracker.readImage(frames.data() + f * px);
    const FrameLines& F = *tracker.curframe_;
    unsigned long long sum = 0;
    for (size_t i = 0; i < px; ++i) sum = sum * 1315423911ull + F.img[i];
    std::printf("frame%d_img %llu\n", f, sum);
    std::printf("frame%d_lines", f);
    for (const Line& l : F.vecLine) std::printf(" %.9g %.9g %.9g %.9g", l.line_endpoint[0], l.line_endpoint[1], l.line_endpoint[2], l.line_endpoint[3]);
    std::printf("\nframe%d_det", f);
    for (const Line& l : tracker.last_detected)
      std::printf(" %.9g %.9g %.9g %.9g %.17g %.17g %.17g %.9g %.9g %.9g", l.line_endpoint[0], l.line_endpoint[1], l.line_endpoint[2],
                  l.line_endpoint[3], l.line_equation[0], l.line_equation[1], l.line_equation[2], l.center[0], l.center[1], l.length);
    std::printf("\nframe%d_ids", f);
    for (int v : F.lineID) std::printf(" %d", v);
    std::printf("\nframe%d_tcnt", f);
    for (int v : F.t_cnt) std::printf(" %d", v);
    std::printf("\nframe%d_match", f);
    for (int v : tracker.last_match) std::printf(" %d", v);
    std::printf("\nframe%d_obs", f);
    for (const auto& ob : tracker.lineObservations()) {
      std::printf(" %d", ob.id);
      for (int k = 0; k < 8; ++k) std::printf(" %.9g", ob.v[k]);
    }
    std::printf("\nframe%d_vpids", f);
    for (int v : tracker.last_vp_ids) std::printf(" %d", v);
    std::printf("\nframe%d_vps %u", f, tracker.last_vp_seed);
    for (int k = 0; k < 9; ++k) std::printf(" %.17g", tracker.last_vps[k]);
    std::printf("\nframe%d_vp4", f);
    for (const auto& a : F.vps) std::printf(" %.17g %.17g %.17g %.17g", a[0], a[1], a[2], a[3]);
    std::printf("\nframe%d_cnt %d %d\n", f, tracker.allfeature_cnt, tracker.lines_exit ? 1 : 0);
  }
  return 0;
}
