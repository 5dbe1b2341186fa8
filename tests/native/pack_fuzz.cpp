// Host-only sweep of the layout packers of csrc/ba_pack.h under the address sanitizer (tests/test_point_units.py builds and
// runs it with g++ -fsanitize=address,undefined): random track layouts, every output table allocated at exactly the size the
// library gives it (vplines_ba.hip: maxPR, maxKS), so that a write or read past a table is an error here and not a silent
// corruption of the neighbouring table in the staging arena.  Also replays the commit-ticket chains of every layout.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ba_pack.h"

using namespace vpl;
constexpr int SCHUR_THREADS = 256, SCHUR_NARROW_FRAMES = 6;   // csrc/ba_step.h:26, :178 (device header: not includable here)

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937 rng(argc > 2 ? atoi(argv[2]) : 1);
  auto U = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
  int refused = 0;
  for (int t = 0; t < trials; ++t) {
    const int caps[4] = {16, 64, 256, 200};
    const int maxP = caps[U(0, 3)], maxL = U(0, 1) ? 128 : 16;
    const int maxPO = maxP * U(2, NF);
    const int maxPR = std::min((maxP / 16 + NF) * (NF - 1), maxPO / 16 + NF * (NF - 1) / 2) / 32 + 2;
    const int maxKS = maxP / 4 + maxL + NF + 2;
    const int nP = U(0, maxP), nL = U(0, maxL);
    const int shape = U(0, 3);    // 0 uniform length, 1 ragged, 2 everything starts in frame 0, 3 long tracks
    std::vector<int> start(nP), nobs(nP), off(nP);
    int total = 0;
    for (int p = 0; p < nP; ++p) {
      start[p] = shape == 2 ? 0 : U(0, NF - 2);
      const int room = NF - start[p];
      nobs[p] = shape == 0 ? std::min(6, room) : shape == 3 ? room : U(2, room);
      if (nobs[p] < 2) { start[p] = NF - 2; nobs[p] = 2; }
      off[p] = total;
      total += nobs[p];
    }
    if (total > maxPO) continue;   // (the library refuses such a window before it packs)
    int cnt[NF + 1] = {0};
    for (int p = 0; p < nP; ++p) cnt[start[p] + 1]++;
    for (int f = 0; f < NF; ++f) cnt[f + 1] += cnt[f];
    std::vector<int> ps(nP);
    {
      int pos[NF + 1];
      for (int f = 0; f <= NF; ++f) pos[f] = cnt[f];
      for (int p = 0; p < nP; ++p) ps[pos[start[p]]++] = p;
      for (int f = 0; f < NF; ++f) std::stable_sort(ps.begin() + cnt[f], ps.begin() + cnt[f + 1], [&](int a, int b) { return nobs[a] > nobs[b]; });
    }
    int* lt = new int[(size_t)maxPR * 1024];
    int* st = new int[(size_t)maxPR * 512];
    PointUnitLayout PL;
    if (!pack_point_units(nobs.data(), off.data(), ps.data(), cnt, maxPR, lt, st, &PL)) ++refused;
    else {
      if (PL.rounds > maxPR || PL.rounds0 > PL.rounds) { printf("trial %d: rounds %d / %d of %d\n", t, PL.rounds0, PL.rounds, maxPR); return 1; }
      if (!point_unit_chains_finish(st, PL, false) || !point_unit_chains_finish(st, PL, true)) { printf("trial %d: a commit chain does not finish\n", t); return 1; }
      // every factor (track, k >= 1) appears exactly once in the lane table, under its start frame
      std::vector<int> seen(total, 0);
      for (int r = 0; r < PL.rounds; ++r)
        for (int l = 0; l < 512; ++l) {
          const int a = lt[((size_t)r * 512 + l) * 2];
          if (a < 0) continue;
          const int p = a & 0xffff, k = (a >> 16) & 15, f = (a >> 20) & 15;
          if (p >= nP || k < 1 || k >= nobs[p] || f != start[p]) { printf("trial %d: lane (%d, %d) names track %d k %d f %d\n", t, r, l, p, k, f); return 1; }
          seen[off[p] + k]++;
        }
      for (int p = 0; p < nP; ++p)
        for (int k = 1; k < nobs[p]; ++k)
          if (seen[off[p] + k] != 1) { printf("trial %d: factor (%d, %d) packed %d times\n", t, p, k, seen[off[p] + k]); return 1; }
    }
    delete[] lt;
    delete[] st;
    // landmark elimination: entries of four landmarks of one start frame, dealt to the waves in chunks
    std::vector<int> ls(nL), lno(nL);
    for (int l = 0; l < nL; ++l) { ls[l] = U(0, NF - 2); lno[l] = U(1, NF - ls[l]); }
    int* tab = new int[(size_t)maxKS * 4];
    int* wave = new int[(size_t)8 * SK_WSTRIDE];
    long ww = 0, tw = 0;
    const int rc = pack_schur_ksteps(nP, ps.data(), cnt, nL, ls.data(), maxKS, tab, wave, SCHUR_THREADS / 64, nobs.data(), lno.data(),
                                     U(0, 1) ? SCHUR_NARROW_FRAMES : 0, &ww, &tw);
    if (rc > maxKS) { printf("trial %d: %d K-steps of %d\n", t, rc, maxKS); return 1; }
    delete[] tab;
    delete[] wave;
  }
  printf("pack_fuzz: %d layouts, %d refused for table size, no table overrun, every factor packed once, every chain finishes\n", trials, refused);
  return 0;
}
