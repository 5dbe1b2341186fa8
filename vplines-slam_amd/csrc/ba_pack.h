// Host-side layout tables of k_lin's point phase (pure C++: no device call, so the packing rules are testable without a GPU).
//
// Work unit = (start frame f, observation index k >= 1, <= 16 of the tracks that start in f and are seen at k), packed
// first-fit into quarter-wave slots (full units take a slot, small ones share one on even lane boundaries).  All units of
// a chunk of tracks go to the same HALF of the work-group (waves 0..3 or 4..7): the per-track sums over k are then ordered
// by one four-wave chain per half and need no second copy.
//
//   lane table  lt[round][512][2]      : (track | k << 16 | f << 20, observation offset) of every lane, -1 = idle
//   unit table  st[round][32][8][2]    : per quarter-wave slot up to 8 x (descriptor, tickets); descriptor 0 ends the list
//       descriptor = f | (f + k) << 4 | ks0 << 8 | ks1 << 12 | 1 << 16   (ks0..ks1: range of MFMA K-steps = lane pairs)
//       tickets    = seq | seq0 << 16
//          seq  : position of the unit in the commit chain of its half in the solve pass (every round)
//          seq0 : position in the chain of the MARGIN_OLD pass, which runs the first `rounds0` rounds only (the units of
//                 start frame 0 sit there); 0xffff for units of later rounds.
// A wave waits for its unit's ticket before it adds the unit's Hessian tile, so BOTH sequences must be consistent with the
// order in which each wave reaches its own units (round, quarter, unit) and each must number exactly the units its pass
// executes: a pass that waits for a ticket owned by a unit it never runs would spin forever.  seq0 is the rank of the
// unit among the units of the first rounds0 rounds in seq order -- a sub-sequence of a consistent order is consistent.
#pragma once
#include <algorithm>
#include <vector>

#include "ba_types.h"

namespace vpl {

struct PointUnitLayout {
  int rounds = 0, rounds0 = 0;
};

// ps_list / cnt: the window's tracks sorted by start frame (cnt[f] .. cnt[f+1]), inside a start frame by decreasing length.
// lt / st must hold maxPR rounds.  Returns false when the table is too small.
inline bool pack_point_units(const int* point_nobs, const int* pt_off, const int* ps_list, const int* cnt, int maxPR,
                             int* lt, int* st, PointUnitLayout* out) {
  struct Slot { int used, nsub, desc[8], first[8], cnt[8], lane0[8], k[8]; };
  std::vector<Slot> slots[2];
  std::vector<int> open[2];
  int slots0[2] = {0, 0}, load[2] = {0, 0};
  for (int f = 0; f < NF; ++f) {
    const int c0 = cnt[f], c1 = cnt[f + 1];
    const int maxno = c1 > c0 ? point_nobs[ps_list[c0]] : 0;
    std::vector<int> half_of((c1 - c0 + 15) / 16, -1);
    for (int k = 1; k < maxno; ++k) {
      int ck = 0;
      while (c0 + ck < c1 && point_nobs[ps_list[c0 + ck]] > k) ++ck;
      for (int q = 0; q < ck; q += 16) {
        const int n = std::min(16, ck - q), need = (n + 1) & ~1;
        int& hf = half_of[q / 16];
        if (hf < 0) {   // the chunk's lanes over all k go to the lighter half
          hf = load[1] < load[0] ? 1 : 0;
          for (int m = q; m < std::min(q + 16, c1 - c0); ++m) load[hf] += point_nobs[ps_list[c0 + m]] - 1;
        }
        std::vector<Slot>& SL = slots[hf];
        std::vector<int>& OP = open[hf];
        int si = -1;
        for (size_t o = 0; o < OP.size() && si < 0; ++o)
          if (16 - SL[OP[o]].used >= need) si = OP[o];
        if (si < 0) {
          si = (int)SL.size();
          SL.push_back(Slot{});
          OP.push_back(si);
        }
        Slot& S = SL[si];
        const int i = S.nsub++;
        S.lane0[i] = S.used; S.first[i] = c0 + q; S.cnt[i] = n; S.k[i] = k;
        S.desc[i] = f | (f + k) << 4 | (S.used / 2) << 8 | ((S.used + need) / 2) << 12 | 1 << 16;
        S.used += need;
        if (S.used == 16 || S.nsub == 8) OP.erase(std::find(OP.begin(), OP.end(), si));
        if (f == 0) slots0[hf] = std::max(slots0[hf], si + 1);
      }
    }
  }
  const int rounds = (int)(std::max(slots[0].size(), slots[1].size()) + 15) / 16;
  if (rounds > maxPR) return false;
  const int rounds0 = (std::max(slots0[0], slots0[1]) + 15) / 16;
  out->rounds = rounds;
  out->rounds0 = rounds0;
  std::fill(lt, lt + (size_t)rounds * 1024, -1);
  std::fill(st, st + (size_t)rounds * 512, 0);
  for (int hf = 0; hf < 2; ++hf)
    for (size_t si = 0; si < slots[hf].size(); ++si) {   // slot i of a half: round i / 16, wave 4 hf + i % 4, quarter (i % 16) / 4
      const int rnd = (int)si / 16, wave = 4 * hf + (int)si % 4, qq = ((int)si % 16) / 4;
      const Slot& S = slots[hf][si];
      for (int i = 0; i < S.nsub; ++i) {
        for (int m = 0; m < S.cnt[i]; ++m) {
          const int p = ps_list[S.first[i] + m];
          int* e = &lt[(rnd * 512 + wave * 64 + qq * 16 + S.lane0[i] + m) * 2];
          e[0] = p | S.k[i] << 16 | (S.desc[i] & 15) << 20;
          e[1] = pt_off[p];
        }
        st[((rnd * 32 + wave * 4 + qq) * 8 + i) * 2] = S.desc[i];
      }
    }
  // commit tickets: two chains (waves 0..3 / 4..7).  A wave reaches its units in (round, quarter, unit) order; the
  // chain serves, among the four waves' next units, the one a rough cycle model expects to be ready first, so that a
  // slot packed with many small units does not hold up the waves whose slots are full ones
  for (int hf = 0; hf < 2; ++hf) {
    const int F = 11000, STAGE = 1200, KS = 170, SUB = 200, COMMIT = 700;   // factor math, staging, per MFMA step, per unit, per commit
    int pos[4] = {0, 0, 0, 0};          // next item of each wave: round * 32 + qq * 8 + i
    long clk[4] = {F, F, F, F}, chain = 0;
    bool staged[4] = {false, false, false, false};
    int seq = 0, seq0 = 0;
    auto entry = [&](int wv, int ps) { return &st[(((ps / 32) * 32 + (hf * 4 + wv) * 4 + (ps % 32) / 8) * 8 + ps % 8) * 2]; };
    auto advance = [&](int wv) {        // skip to the wave's next existing unit, charging round starts and staging
      while (pos[wv] < rounds * 32 && entry(wv, pos[wv])[0] == 0) {
        pos[wv] = (pos[wv] / 8 + 1) * 8;           // descriptor 0 ends a slot's list
        if (pos[wv] % 32 == 0 && pos[wv] < rounds * 32) clk[wv] += F;
        staged[wv] = false;
      }
    };
    for (int wv = 0; wv < 4; ++wv) advance(wv);
    for (;;) {
      int best = -1;
      long bt = 0;
      for (int wv = 0; wv < 4; ++wv) {
        if (pos[wv] >= rounds * 32) continue;
        const int d = entry(wv, pos[wv])[0];
        const long t = clk[wv] + (staged[wv] ? 0 : STAGE) + SUB + KS * (((d >> 12) & 15) - ((d >> 8) & 15));
        if (best < 0 || t < bt) { best = wv; bt = t; }
      }
      if (best < 0) break;
      // the marginalisation pass numbers only the units it runs, in the same relative order
      const bool in0 = pos[best] / 32 < rounds0;
      if (seq > 0xfffe) return false;
      entry(best, pos[best])[1] = seq | (in0 ? seq0 : 0xffff) << 16;
      ++seq;
      if (in0) ++seq0;
      chain = std::max(chain, bt) + COMMIT;
      clk[best] = chain;
      staged[best] = true;
      ++pos[best];
      if (pos[best] % 8 == 0) {
        staged[best] = false;
        if (pos[best] % 32 == 0 && pos[best] < rounds * 32) clk[best] += F;
      }
      advance(best);
    }
  }
  return true;
}

// Replays the commit chains of one pass the way the device runs them (four waves per half, each walking its own units in
// (round, quarter, unit) order and spinning on the half's ticket counter) and reports whether every wave finishes.
// marg = true: the MARGIN_OLD pass (first rounds0 rounds, seq0 tickets).
inline bool point_unit_chains_finish(const int* st, const PointUnitLayout& L, bool marg) {
  const int nR = marg ? L.rounds0 : L.rounds;
  for (int hf = 0; hf < 2; ++hf) {
    int pos[4] = {0, 0, 0, 0};
    int tick = 0;
    auto entry = [&](int wv, int ps) { return &st[(((ps / 32) * 32 + (hf * 4 + wv) * 4 + (ps % 32) / 8) * 8 + ps % 8) * 2]; };
    auto advance = [&](int wv) {
      while (pos[wv] < nR * 32 && entry(wv, pos[wv])[0] == 0) pos[wv] = (pos[wv] / 8 + 1) * 8;
    };
    for (int wv = 0; wv < 4; ++wv) advance(wv);
    for (;;) {
      bool progressed = false, pending = false;
      for (int wv = 0; wv < 4; ++wv) {
        if (pos[wv] >= nR * 32) continue;
        pending = true;
        const int t = entry(wv, pos[wv])[1];
        const int want = marg ? (t >> 16) & 0xffff : t & 0xffff;
        if (want == tick) { ++tick; ++pos[wv]; advance(wv); progressed = true; }
      }
      if (!pending) break;
      if (!progressed) return false;   // every pending wave waits for a ticket nobody will pass
    }
  }
  return true;
}

// ---- entry table of k_schur (ba_step.h) ---------------------------------------------------------------------------
// Landmark rows sorted by group (points by start frame, then lines by start frame).  An entry holds four landmarks of one
// group: four points = one K-step of the FP64 matrix-core instruction, four lines = four K-steps (K-step a takes row a of
// each line; weight 4).
// The entries are dealt to the NWV (<= 8) waves of k_schur in CHUNKS -- a whole group, or a piece of a group that is heavier
// than a wave's fair share; a wave adds its accumulators to the shared system at the end of every chunk.  Those adds are
// committed in ticket order, sorted by the K-steps the wave has done when it reaches the add (ties: by wave): the waves
// start together and advance at about the same rate, so a wave seldom finds its ticket not yet due -- and the order is a
// fixed function of the table.
//   tab  [maxKS][4] : {group (0..10: points of that start frame, 32 | f: lines, 64: WIDE), id0 | id1 << 16, id2 | id3 << 16, 0}, 0xffff = none
//   wave [8][SK_WSTRIDE] : {number of chunks n, then n x (first entry, end, ticket)}
// Mixed track lengths (narrow_frames > 0: the batch's rows are wider than narrow_frames frames, round 4): an entry that holds a
// track of more than narrow_frames observations is WIDE (bit 6 of the group; its product needs every column tile of the
// compact row), the others are narrow (3 column tiles); wide and narrow entries of a start frame are different groups, so a
// chunk is one or the other.  The points of a start frame come sorted by length (ps_list), so its wide entries are a prefix;
// the lines of a start frame are dealt in two passes.  point_nobs / ln_nobs are only read when narrow_frames > 0.
// Returns the number of entries or -1 when a table is too small.
inline int pack_schur_ksteps(int nP, const int* ps_list, const int* cnt, int nL, const int* ln_start, int maxKS, int* tab,
                             int* wave, int NWV, const int* point_nobs = nullptr, const int* ln_nobs = nullptr, int narrow_frames = 0,
                             long* wide_weight = nullptr, long* total_weight = nullptr) {
  std::vector<int> grp, wgt;
  int nks = 0;
  const bool mixed = narrow_frames > 0;
  auto push = [&](int g, const int* id) {
    if (nks >= maxKS) return false;
    tab[4 * nks] = g; tab[4 * nks + 1] = id[0] | id[1] << 16; tab[4 * nks + 2] = id[2] | id[3] << 16; tab[4 * nks + 3] = 0;
    grp.push_back(g);
    // weight ~ time of the entry: 6 tile products narrow; wide = five passes of five products, six loads and transforms each
    // (x 4 K-steps for lines)
    wgt.push_back(((g & 32) ? 4 : 1) * (mixed ? ((g & 64) ? 10 : 2) : 1));
    ++nks;
    return true;
  };
  (void)nP;
  for (int f = 0; f < NF; ++f)
    for (int q = cnt[f]; q < cnt[f + 1]; q += 4) {
      int id[4];
      bool wide = false;
      for (int i = 0; i < 4; ++i) {
        id[i] = q + i < cnt[f + 1] ? ps_list[q + i] : 0xffff;
        if (mixed && id[i] != 0xffff && point_nobs[id[i]] > narrow_frames) wide = true;
      }
      if (!push(f | (wide ? 64 : 0), id)) return -1;
    }
  for (int f = 0; f < NF; ++f)
    for (int pass = mixed ? 0 : 1; pass < 2; ++pass) {   // pass 0: the long tracks of the start frame (mixed only)
      int id[4], n = 0;
      const int g = 32 | f | (pass == 0 ? 64 : 0);
      for (int l = 0; l < nL; ++l) {
        if (ln_start[l] != f) continue;
        if (mixed && (ln_nobs[l] > narrow_frames) != (pass == 0)) continue;
        id[n++] = l;
        if (n == 4) { if (!push(g, id)) return -1; n = 0; }
      }
      if (n) {
        for (int i = n; i < 4; ++i) id[i] = 0xffff;
        if (!push(g, id)) return -1;
      }
    }
  // Chunks: a group, or a piece of one when the group is heavier than a wave's fair share.  A chunk is worked by ONE wave
  // and ends with ONE add into the shared system; chunks go to the waves longest-first onto the least loaded wave.
  std::vector<long> pre(nks + 1, 0);
  for (int k = 0; k < nks; ++k) pre[k + 1] = pre[k] + wgt[k];
  const long total = pre[nks];
  if (total_weight) *total_weight += total;
  if (wide_weight)
    for (int k = 0; k < nks; ++k) if (grp[k] & 64) *wide_weight += wgt[k];
  const long fair = std::max(1L, (total + NWV - 1) / NWV);
  struct Chunk { int k0, k1; long w; };
  std::vector<Chunk> chunks;
  for (int k = 0; k < nks;) {
    int e = k;
    while (e < nks && grp[e] == grp[k]) ++e;
    const long gw = pre[e] - pre[k];
    const int pieces = (int)((gw + fair - 1) / fair);
    int c0 = k;
    for (int pc = 0; pc < pieces; ++pc) {
      const long lim = pre[k] + gw * (pc + 1) / pieces;
      int c1 = c0;
      while (c1 < e && (pc == pieces - 1 || pre[c1 + 1] <= lim)) ++c1;
      if (c1 > c0) chunks.push_back(Chunk{c0, c1, pre[c1] - pre[c0]});
      c0 = c1;
    }
    k = e;
  }
  std::vector<int> ord(chunks.size());
  for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int)i;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return chunks[a].w > chunks[b].w; });
  std::vector<std::vector<int>> mine(NWV);
  std::vector<long> load(NWV, 0);
  for (int ci : ord) {
    int best = 0;
    for (int wv = 1; wv < NWV; ++wv) if (load[wv] < load[best]) best = wv;
    mine[best].push_back(ci);
    load[best] += chunks[ci].w;
  }
  // a wave works its chunks in table order; ticket order = by the weight the wave has done when the chunk ends (ties: wave)
  struct Inc { int wv, ci; long when; };
  std::vector<Inc> inc;
  for (int wv = 0; wv < NWV; ++wv) {
    // table order; WIDE chunks last: a wide chunk holds its ticket from the adds of its first pass to those of its fifth, so it
    // should be the last one anybody waits for
    std::sort(mine[wv].begin(), mine[wv].end(), [&](int a, int b) {
      const bool wa = (grp[chunks[a].k0] & 64) != 0, wb = (grp[chunks[b].k0] & 64) != 0;
      if (wa != wb) return wb;
      return a < b;
    });
    long done = 0;
    for (int ci : mine[wv]) { done += chunks[ci].w; inc.push_back(Inc{wv, ci, (grp[chunks[ci].k0] & 64) ? total + done : done}); }
  }
  std::vector<int> order(inc.size());
  for (size_t i = 0; i < inc.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (inc[a].when != inc[b].when) return inc[a].when < inc[b].when;
    return inc[a].wv < inc[b].wv;
  });
  std::vector<int> ticket(inc.size());
  for (size_t t = 0; t < order.size(); ++t) ticket[order[t]] = (int)t;
  for (int wv = 0; wv < 8; ++wv) {
    int* o = wave + wv * SK_WSTRIDE;
    std::fill(o, o + SK_WSTRIDE, -1);
    o[0] = 0;
  }
  for (size_t i = 0; i < inc.size(); ++i) {
    int* o = wave + inc[i].wv * SK_WSTRIDE;
    const int n = o[0];
    if (1 + 3 * (n + 1) > SK_WSTRIDE) return -1;
    o[1 + 3 * n] = chunks[inc[i].ci].k0; o[2 + 3 * n] = chunks[inc[i].ci].k1; o[3 + 3 * n] = ticket[i];
    o[0] = n + 1;
  }
  return nks;
}

}  // namespace vpl
