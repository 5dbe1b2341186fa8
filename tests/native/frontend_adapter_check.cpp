// GPU test program for the C++ front-end adapter (vplines-slam_amd/host/vpl_frontend.hpp): the tracker's two calls,
// EDLineDetector::EDline on both frames and LineMatching::Matching between them, on raw frame files written by
// tests/test_gpu_host_adapter.py.  Prints the lines and the matches for comparison with the oracle.
#include <cstdio>
#include <vector>
#include "../../vplines-slam_amd/host/vpl_frontend.hpp"

using namespace vplhost;

static std::vector<uint8_t> slurp(const char* path, size_t n) {
  std::vector<uint8_t> b(n);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(b.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
  std::fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
  std::vector<uint8_t> a = slurp(argv[1], (size_t)W * H), b = slurp(argv[2], (size_t)W * H);
  FrontendDevice dev(W, H, 512, 8192);
  EDLineParam param = {5, 1.0f, 30.f, 5.f, 2, 35, 1.8};   // line_feature_tracker_node.cpp:203 + euroc_config.yaml
  EDLineDetector det(dev, param);
  LineMatching lm(dev);
  std::vector<Line> la, lb;
  if (det.EDline(a.data(), la, true) != 1 || det.EDline(b.data(), lb, true) != 1) return 3;
  // the reference's demo (test_edline_detector.cpp:15,27; test_line_matching.cpp:57): EDline with its default smoothed = false
  // (Gaussian pre-blur first), then LineFilter(lines, 3.0)
  EDLineParam demo = {5, 1.0f, 30.f, 5.f, 2, 25, 1.8};
  EDLineDetector det2(dev, demo);
  std::vector<Line> ld;
  if (det2.EDline(a.data(), ld) != 1) return 4;
  const size_t n_demo = ld.size();
  lm.LineFilter(ld, 3.0f);
  std::printf("demo %zu %zu\n", n_demo, ld.size());
  std::printf("demolines");
  for (const Line& l : ld) std::printf(" %.9g %.9g %.9g %.9g", l.line_endpoint[0], l.line_endpoint[1], l.line_endpoint[2], l.line_endpoint[3]);
  std::printf("\n");
  for (int k = 0; k < 2; ++k) {
    const std::vector<Line>& L = k ? lb : la;
    std::printf("lines%d", k);
    for (const Line& l : L)
      std::printf(" %.9g %.9g %.9g %.9g %.17g %.17g %.17g %.9g %.9g %.9g", l.line_endpoint[0], l.line_endpoint[1],
                  l.line_endpoint[2], l.line_endpoint[3], l.line_equation[0], l.line_equation[1], l.line_equation[2],
                  l.center[0], l.center[1], l.length);
    std::printf("\n");
  }
  std::vector<int> r2c(3, 77);
  const bool ok = lm.Matching(a.data(), b.data(), la, lb, r2c, true, true);
  std::printf("match %d", ok ? 1 : 0);
  for (int v : r2c) std::printf(" %d", v);
  std::printf("\n");
  std::vector<Line> none;
  std::vector<int> keep(2, 55);
  const bool ok2 = lm.Matching(a.data(), b.data(), none, lb, keep, true, true);
  std::printf("empty %d %d %d\n", ok2 ? 1 : 0, keep[0], keep[1]);
  return 0;
}
