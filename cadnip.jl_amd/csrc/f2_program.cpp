// f2_program.cpp -- host side of the fused kernel's linear solve: the entry program (csrc/fused2.hip executes it).
// Pure C++ (no HIP): also reachable through the host-only C ABI (cadnip_host_f2_*) so that the CPU test-suite can
// emulate the program pass by pass.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "internal.hpp"

namespace cadnip {

typedef unsigned long long u64;
#define NOPOS 0xFFFFu

// ---- the linear-solve program -------------------------------------------------------------------------------
// Work array of one instance:  W = [ sparse L\U (all pattern entries outside the core block) | dense core NC x NC | rhs n | trash ].
// The last NC pivots form the *core*: its Schur complement is accumulated by the entry program like everything else
// and then eliminated by dense_core_solve in registers (NC = 0: no core, the entry program does it all).
// Levels -> passes of 64 lane descriptors.  An entry with nt terms gets a group of min(cap, pow2ceil(nt)) lanes (aligned,
// so a DPP butterfly sums it); groups are packed widest first; cap is chosen per level to minimise passes*2 + iterations.
static bool f2_emit_passes(std::vector<F2Ent>& ents, F2Program& G, int& n_passes) {
  auto p2c = [](size_t x) { size_t r = 1; while (r < x) r *= 2; return r; };
  std::stable_sort(ents.begin(), ents.end(), [](const F2Ent& p, const F2Ent& q) { return p.lvl < q.lvl; });
  n_passes = 0;
  for (size_t i = 0; i < ents.size();) {
    size_t j = i;
    while (j < ents.size() && ents[j].lvl == ents[i].lvl) ++j;
    const size_t E = j - i;
    size_t best_cost = (size_t)-1;
    int best_cap = 1;
    for (int cap = 16; cap >= 1; cap >>= 1) {
      std::vector<size_t> L(E), ord(E);
      size_t total = 0;
      for (size_t e = 0; e < E; ++e) { L[e] = std::min((size_t)cap, p2c(std::max<size_t>(ents[i + e].a.size(), 1))); total += L[e]; ord[e] = e; }
      std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return L[x] > L[y]; });
      const size_t passes = (total + 63) / 64;
      std::vector<size_t> it(passes, 1);
      size_t off = 0;
      for (size_t e : ord) { const size_t nt = ents[i + e].a.size(); it[off / 64] = std::max(it[off / 64], (nt + L[e] - 1) / L[e]); off += L[e]; }
      size_t cost = 0;
      for (size_t p = 0; p < passes; ++p) cost += 2 + it[p] - 1;
      if (cost < best_cost) { best_cost = cost; best_cap = cap; }
    }
    std::vector<size_t> L(E), ord(E);
    for (size_t e = 0; e < E; ++e) { L[e] = std::min((size_t)best_cap, p2c(std::max<size_t>(ents[i + e].a.size(), 1))); ord[e] = e; }
    std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return L[x] > L[y]; });
    const size_t base = G.lanes.size();
    std::vector<int> lane_lg, lane_nt, lane_div;
    for (size_t e : ord) {
      const F2Ent& x = ents[i + e];
      int lg = 0;
      while (((size_t)1 << lg) < L[e]) ++lg;
      for (size_t sub = 0; sub < L[e]; ++sub) {
        const size_t t0 = G.terms.size();
        size_t nt = 0;
        for (size_t t = sub; t < x.a.size(); t += L[e], ++nt) G.terms.push_back((unsigned)x.a[t] | ((unsigned)x.b[t] << 16));
        if (nt > 255 || t0 >= 65535) return false;
        G.lanes.push_back(pack4(x.pos, x.dg < 0 ? NOPOS : x.dg, (unsigned)t0, (unsigned)(nt | ((unsigned)lg << 8) | ((sub == 0 ? 1u : 0u) << 12))));
        lane_lg.push_back(lg); lane_nt.push_back((int)nt); lane_div.push_back(x.dg >= 0);
      }
    }
    const size_t total = G.lanes.size() - base;
    for (size_t off = 0; off < total; off += 64) {
      const size_t T = std::min<size_t>(64, total - off);
      unsigned maxlg = 0, hasdiv = 0, multi = 0;
      for (size_t l = off; l < off + T; ++l) { maxlg = std::max(maxlg, (unsigned)lane_lg[l]); hasdiv |= (unsigned)lane_div[l]; multi |= lane_nt[l] > 1 ? 1u : 0u; }
      const unsigned fence = off + 64 >= total ? 1u : 0u;
      if (base + off >= ((size_t)1 << 31)) return false;
      G.passes.push_back((u64)(base + off) | ((u64)(T | (maxlg << 8) | (hasdiv << 11) | (fence << 12) | (multi << 13)) << 32));
      ++n_passes;
    }
    i = j;
  }
  return true;
}

// Entry program for core size nc, straight from the L\U pattern (permuted indices, rows sorted by column):
//   pre  : every L\U entry outside the core block (left-looking recurrence, L entries divided by their pivot), the Schur
//          updates of the core block by the leaf pivots, forward substitution as the recurrence of an extra column (for core
//          rows: their leaf part);
//   post : back substitution of the leaf rows, in place (the core rows of x are written by the dense solve).
// max_terms > 0 (team program): an entry with more multiply-add terms than that is split into a chain of partial entries on consecutive
// levels (all but the last only subtract their share; the last one also divides), so that one lane never carries more than one term.
static void f2_build_entries(const LUProgram& P, int n, int nc, int max_terms, F2Program& G, std::vector<F2Ent>& pre, std::vector<F2Ent>& post, std::vector<F2Ent>& fwd) {
  G = F2Program();
  G.nc = nc;
  const int cs0 = n - nc;
  std::vector<int> rowof(P.nnz_lu);
  for (int i = 0; i < n; ++i) for (int p = P.lu_rowptr[i]; p < P.lu_rowptr[i + 1]; ++p) rowof[p] = i;
  G.posW.assign(P.nnz_lu, -1);
  int nsp = 0;
  for (int p = 0; p < P.nnz_lu; ++p) if (!(rowof[p] >= cs0 && P.lu_col[p] >= cs0)) G.posW[p] = nsp++;
  G.dn0 = nsp;
  for (int p = 0; p < P.nnz_lu; ++p) if (G.posW[p] < 0) G.posW[p] = G.dn0 + (rowof[p] - cs0) * nc + (P.lu_col[p] - cs0);
  G.lu_words = nsp + nc * nc;
  if ((G.lu_words + n) & 1) ++G.lu_words;   // pad: W = [LU | rhs n | trash] has an even word count (zeroed 16 bytes at a time)
  const int y0 = G.lu_words;
  auto find = [&](int i, int j) -> int {     // pattern position of (i, j) or -1
    const int* b = &P.lu_col[P.lu_rowptr[i]]; const int* e = &P.lu_col[P.lu_rowptr[i + 1]];
    const int* it = std::lower_bound(b, e, j);
    return (it != e && *it == j) ? (int)(it - &P.lu_col[0]) : -1;
  };
  auto unit = [&](int k) { return !P.unit.empty() && P.unit[k] != 0; };   // pivot k is the stamped constant 1 (symbolic.cpp: leaf phase)
  std::vector<int> lev(P.nnz_lu, -1);        // level at which a sparse entry is final (-1: as assembled)
  pre.clear(); post.clear(); fwd.clear();
  // append entry x (its level set); returns the level at which its value is final
  // (x.tl[t]: the level at which both factors of term t are final, -1 = as assembled; x.dl: that of the pivot.)  A long entry is cut into
  // chunks of max_terms terms in the order in which its terms become available: a chunk runs one level behind its latest term and behind
  // the previous chunk (they update the same word), so only the terms that arrive last lengthen the critical path.
  auto push = [&](std::vector<F2Ent>& list, F2Ent&& x) -> int {
    const size_t nt = x.a.size();
    if (max_terms <= 0 || nt <= (size_t)max_terms) { const int l = x.lvl; list.push_back(std::move(x)); return l; }
    std::vector<size_t> ord(nt);
    for (size_t t = 0; t < nt; ++t) ord[t] = t;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t p, size_t q) { return x.tl[p] < x.tl[q]; });
    int prev = -1, l = -1;
    for (size_t c0 = 0; c0 < nt; c0 += max_terms) {
      const size_t c1 = std::min(nt, c0 + (size_t)max_terms);
      const bool last = c1 == nt;
      F2Ent y; y.pos = x.pos; y.dg = last ? x.dg : -1;
      int ready = prev;
      for (size_t t = c0; t < c1; ++t) { y.a.push_back(x.a[ord[t]]); y.b.push_back(x.b[ord[t]]); ready = std::max(ready, x.tl[ord[t]]); }
      if (last) ready = std::max(ready, x.dl);
      l = ready + 1;
      y.lvl = l; prev = l;
      list.push_back(std::move(y));
    }
    return l;
  };
  for (int i = 0; i < n; ++i)
    for (int p = P.lu_rowptr[i]; p < P.lu_rowptr[i + 1]; ++p) {
      const int j = P.lu_col[p];
      if (i >= cs0 && j >= cs0) continue;
      F2Ent x; x.pos = G.posW[p]; x.dg = -1; x.lvl = -1;
      for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
        const int k = P.lu_col[pl];
        if (k >= j) break;
        const int pu = find(k, j);
        if (pu < 0) continue;
        x.a.push_back(G.posW[pl]); x.b.push_back(G.posW[pu]); x.tl.push_back(std::max(lev[pl], lev[pu]));
        x.lvl = std::max(x.lvl, std::max(lev[pl], lev[pu]));
      }
      if (j < i && !unit(j)) { x.dg = G.posW[P.lu_diag[j]]; x.dl = lev[P.lu_diag[j]]; x.lvl = std::max(x.lvl, lev[P.lu_diag[j]]); }   // L entry / pivot (a constant-1 pivot divides nothing)
      if (x.a.empty() && x.dg < 0) continue;
      x.lvl += 1;
      lev[p] = push(pre, std::move(x));
    }
  for (int i = cs0; i < n; ++i)
    for (int j = cs0; j < n; ++j) {
      F2Ent x; x.pos = G.dn0 + (i - cs0) * nc + (j - cs0); x.dg = -1; x.lvl = -1;
      for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
        const int k = P.lu_col[pl];
        if (k >= cs0) break;
        const int pu = find(k, j);
        if (pu < 0) continue;
        x.a.push_back(G.posW[pl]); x.b.push_back(G.posW[pu]); x.tl.push_back(std::max(lev[pl], lev[pu]));
        x.lvl = std::max(x.lvl, std::max(lev[pl], lev[pu]));
      }
      if (x.a.empty()) continue;
      x.lvl += 1;
      push(pre, std::move(x));
    }
  std::vector<int> ylev(n, -1);
  for (int i = 0; i < n; ++i) {
    F2Ent x; x.pos = y0 + i; x.dg = -1; x.lvl = -1;
    for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
      const int k = P.lu_col[pl];
      if (k >= cs0) break;
      x.a.push_back(G.posW[pl]); x.b.push_back(y0 + k); x.tl.push_back(std::max(lev[pl], ylev[k]));
      x.lvl = std::max(x.lvl, std::max(lev[pl], ylev[k]));
    }
    if (x.a.empty()) continue;
    x.lvl += 1;
    ylev[i] = push(pre, std::move(x));
  }
  std::vector<int> xlev(n, -1);
  for (int i = cs0 - 1; i >= 0; --i) {
    F2Ent x; x.pos = y0 + i; x.dg = unit(i) ? -1 : G.posW[P.lu_diag[i]]; x.lvl = -1;
    for (int pu = P.lu_diag[i] + 1; pu < P.lu_rowptr[i + 1]; ++pu) {
      const int j = P.lu_col[pu];
      x.a.push_back(G.posW[pu]); x.b.push_back(y0 + j); x.tl.push_back(j < cs0 ? xlev[j] : -1);
      if (j < cs0) x.lvl = std::max(x.lvl, xlev[j]);
    }
    if (x.a.empty() && x.dg < 0) continue;     // x_i = y_i: nothing to do (xlev stays -1: final as forward substitution left it)
    x.lvl += 1;
    xlev[i] = push(post, std::move(x));
  }
  // forward substitution on KEPT factors (Newton mode 1, a round that does not refactor): the y recurrences alone; every L entry is
  // final, so a row's level is one more than the deepest y it reads
  {
    std::vector<int> fl(n, -1);
    for (int i = 0; i < n; ++i) {
      F2Ent x; x.pos = y0 + i; x.dg = -1; x.lvl = -1;
      for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
        const int k = P.lu_col[pl];
        if (k >= cs0) break;
        x.a.push_back(G.posW[pl]); x.b.push_back(y0 + k); x.tl.push_back(fl[k]);
        x.lvl = std::max(x.lvl, fl[k]);
      }
      if (x.a.empty()) continue;
      x.lvl += 1;
      fl[i] = push(fwd, std::move(x));
    }
  }
}

bool f2_build_program(const LUProgram& P, int n, int nc, F2Program& G) {
  std::vector<F2Ent> pre, post, fwd;
  f2_build_entries(P, n, nc, 0, G, pre, post, fwd);
  if (!f2_emit_passes(pre, G, G.n_pre) || !f2_emit_passes(post, G, G.n_post) || !f2_emit_passes(fwd, G, G.n_fwd)) return false;
  G.terms.push_back(0);   // a lane without terms still prefetches term[t0]
  if (G.terms.size() >= 65535) return false;
  // wave cycles measured on the DFF (tools/trace_fused2.py): ~950 per pass; dense solve 2.7 k at nc = 8, 4.3 k at nc = 12
  G.cost = 950.0 * (G.n_pre + G.n_post) + (nc ? 1450.0 + 19.5 * nc * nc : 0.0);
  return true;
}

// ---- the same program for a TEAM of nw waves per instance (fused_team_kernel.hpp) ----------------------------------------------------
// One wave alone pays for instructions, not for data (~8 cycles per instruction, ~25 per taken branch: tools/ubench/lat.hip), so a step of
// the team's linear solve is straight-line code: every lane gets ONE 8-byte descriptor per step -- four 15-bit word offsets into the work
// array: the entry, its pivot, and the two factors of its single multiply-add term; the leader flag and the lane-group width sit in the
// top bits of the four 16-bit fields -- staged in LDS (160 KB = 20 480 doubles: 15 bits address all of it); no term lists, no loops, no
// per-pass variants.  Lanes without work point at the constant words behind the work array: W[zero] = 0.0 (factors, entry) and
// W[zero + 1] = 1.0 (pivot of an entry that is not divided).  A step is nw * 64 lanes of ONE dependency level (a level with more lanes
// takes several steps); every step ends with a workgroup barrier.
bool f2_build_team(const LUProgram& P, int n, int nc, int nw, F2Team& T) {
  F2Program G;
  std::vector<F2Ent> lists[3];
  f2_build_entries(P, n, nc, 16, G, lists[0], lists[1], lists[2]);
  T = F2Team();
  T.nw = nw; T.nc = G.nc; T.lu_words = G.lu_words; T.dn0 = G.dn0; T.posW = G.posW;
  const unsigned zero_w = (unsigned)(G.lu_words + n + 64 /* F2_TRASH */), one_w = zero_w + 1u;
  if (one_w >= (1u << 15)) return false;
  const size_t lanes_per_step = 64 * (size_t)nw;
  auto p2c = [](size_t x) { size_t r = 1; while (r < x) r *= 2; return r; };
  auto pack = [](unsigned pos, unsigned leader, unsigned piv, unsigned a, unsigned b, unsigned lg) -> unsigned long long {
    return (unsigned long long)(pos | leader << 15) | (unsigned long long)(piv | (lg & 1u) << 15) << 16 |
           (unsigned long long)(a | ((lg >> 1) & 1u) << 15) << 32 | (unsigned long long)(b | ((lg >> 2) & 1u) << 15) << 48;
  };
  const unsigned long long idle = pack(zero_w, 0, one_w, zero_w, zero_w, 0);
  for (int li = 0; li < 3; ++li) {
    auto& ents = lists[li];
    std::stable_sort(ents.begin(), ents.end(), [](const F2Ent& p, const F2Ent& q) { return p.lvl < q.lvl; });
    for (size_t i = 0; i < ents.size();) {
      size_t j = i;
      while (j < ents.size() && ents[j].lvl == ents[i].lvl) ++j;
      std::vector<size_t> ord;
      for (size_t e = i; e < j; ++e) ord.push_back(e);
      auto width = [&](size_t e) { return std::min<size_t>(16, p2c(std::max<size_t>(ents[e].a.size(), 1))); };
      std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return width(x) > width(y); });   // widest first: groups stay aligned
      size_t used = lanes_per_step;          // lanes filled in the current step (full: the level opens a new one)
      for (size_t e : ord) {
        const F2Ent& x = ents[e];
        const size_t L = width(e);
        if (used + L > lanes_per_step) { T.desc.resize(T.desc.size() + lanes_per_step, idle); used = 0; ++T.n_steps[li]; }
        unsigned lg = 0;
        while (((size_t)1 << lg) < L) ++lg;
        for (size_t sub = 0; sub < L; ++sub)
          T.desc[T.desc.size() - lanes_per_step + used + sub] =
              pack((unsigned)x.pos, sub == 0 ? 1u : 0u, x.dg < 0 ? one_w : (unsigned)x.dg, sub < x.a.size() ? (unsigned)x.a[sub] : zero_w,
                   sub < x.a.size() ? (unsigned)x.b[sub] : zero_w, lg);
        used += L;
      }
      i = j;
    }
  }
  T.desc.resize(T.desc.size() + lanes_per_step, idle);   // the kernel reads one step beyond a list's end
  if (T.desc.size() & 1) T.desc.push_back(idle);         // (what follows the descriptors in LDS stays 16-byte aligned)
  return true;
}

// ---- ... and for ONE wave per instance (k_fused2's lean variant, fused2_kernel.hpp; k_lu_f2) ----------------------------------------
// The pass program above costs a wave ~75 vector instructions per pass plus two LDS round trips per extra term of a lane (term list ->
// operands), and the kernel is bound by instruction issue.  Here a step is straight-line code as for the teams, but a lane carries up to
// THREE multiply-add terms in a 16-byte descriptor -- eight 15-bit word offsets: entry, pivot, (a0, b0), (a1, b1), (a2, b2) -- so that the
// typical entry of a circuit matrix (one to three terms) needs one lane and no lane-group sum at all, and a wave without barriers need
// not align steps with dependency levels: entries are list-scheduled into the earliest step behind their operands' last writers that
// has an aligned lane group free.  The top bits of the eight fields: leader flag, log2 of the group width (3 bits), and -- the same in
// all lanes of a step, read through v_readfirstlane -- the step's widest group (3 bits) and whether any of its entries divides.
// DFF (core of 8): 12 + 10 steps (refactor + solve / forward + backward on kept factors: 10) instead of 17 / 13 passes.
bool f2_build_steps(const LUProgram& P, int n, int nc, int nw, F2Team& T) {
  F2Program G;
  std::vector<F2Ent> lists[3];
  constexpr int TPL = 3;                     // terms per lane
  f2_build_entries(P, n, nc, 16 * TPL, G, lists[0], lists[1], lists[2]);
  T = F2Team();
  T.nw = nw; T.nc = G.nc; T.lu_words = G.lu_words; T.dn0 = G.dn0; T.posW = G.posW;
  const unsigned zero_w = (unsigned)(G.lu_words + n + 64 /* F2_TRASH */), one_w = zero_w + 1u;
  if (one_w >= (1u << 15)) return false;
  const size_t NL = 64 * (size_t)nw;         // lanes of a step (a team's step ends with a workgroup barrier: list scheduling holds for it as well)
  auto p2c = [](size_t x) { size_t r = 1; while (r < x) r *= 2; return r; };
  struct Lane { unsigned pos, leader, piv, lg, a[TPL], b[TPL]; };
  const Lane idle{zero_w, 0, one_w, 0, {zero_w, zero_w, zero_w}, {zero_w, zero_w, zero_w}};
  std::vector<int> fs(one_w + 1);            // step (within the current list) of a word's last writer so far, -1 = none
  for (int li = 0; li < 3; ++li) {
    auto& ents = lists[li];
    std::stable_sort(ents.begin(), ents.end(), [](const F2Ent& p, const F2Ent& q) { return p.lvl < q.lvl; });
    std::fill(fs.begin(), fs.end(), -1);
    std::vector<std::vector<Lane>> steps;
    std::vector<std::vector<unsigned short>> occ;   // per step: occupancy of its 16-lane rows
    std::vector<int> hasdiv, maxlg;
    const bool aligned = getenv("CADNIP_F2_LEVEL_ALIGNED") != nullptr;     // (diagnostic: steps that do not mix dependency levels)
    for (size_t i = 0; i < ents.size();) {
      size_t j = i;
      while (j < ents.size() && ents[j].lvl == ents[i].lvl) ++j;
      const int floor_step = aligned ? (int)steps.size() - 1 : -1;
      std::vector<size_t> ord;
      for (size_t e = i; e < j; ++e) ord.push_back(e);
      auto width = [&](size_t e) { return std::min<size_t>(16, p2c(std::max<size_t>((ents[e].a.size() + TPL - 1) / TPL, 1))); };
      std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return width(x) > width(y); });   // the wide groups of a level share steps
      for (size_t e : ord) {
        const F2Ent& x = ents[e];
        const size_t L = width(e);
        int ready = std::max(fs[x.pos], floor_step);
        if (x.dg >= 0) ready = std::max(ready, fs[x.dg]);
        for (size_t t = 0; t < x.a.size(); ++t) ready = std::max(ready, std::max(fs[x.a[t]], fs[x.b[t]]));
        const unsigned m = (1u << L) - 1u;                  // (L <= 16: a group lies inside one row)
        int s = ready + 1, at = -1;
        for (;; ++s) {
          if ((size_t)s >= steps.size()) { steps.emplace_back(NL, idle); occ.emplace_back(NL / 16, (unsigned short)0); hasdiv.push_back(0); maxlg.push_back(0); }
          for (size_t o = 0; o < NL && at < 0; o += L) if (!(occ[s][o >> 4] & (m << (o & 15)))) at = (int)o;
          if (at >= 0) break;
        }
        occ[s][at >> 4] |= (unsigned short)(m << (at & 15));
        unsigned lg = 0;
        while (((size_t)1 << lg) < L) ++lg;
        maxlg[s] = std::max(maxlg[s], (int)lg);
        if (x.dg >= 0) hasdiv[s] = 1;
        for (size_t sub = 0; sub < L; ++sub) {
          Lane& ln = steps[s][at + sub];
          ln.pos = (unsigned)x.pos; ln.leader = sub == 0; ln.piv = x.dg < 0 ? one_w : (unsigned)x.dg; ln.lg = lg;
          for (int k = 0; k < TPL; ++k) {
            const size_t t = sub + (size_t)k * L;
            ln.a[k] = t < x.a.size() ? (unsigned)x.a[t] : zero_w; ln.b[k] = t < x.a.size() ? (unsigned)x.b[t] : zero_w;
          }
        }
        fs[x.pos] = s;
      }
      i = j;
    }
    T.n_steps[li] = (int)steps.size();
    for (size_t s = 0; s < steps.size(); ++s)
      for (const Lane& ln : steps[s]) {
        const unsigned f0 = maxlg[s] & 1, f1 = (maxlg[s] >> 1) & 1, f2 = (maxlg[s] >> 2) & 1, f3 = hasdiv[s];
        T.desc.push_back((unsigned long long)(ln.pos | ln.leader << 15) | (unsigned long long)(ln.piv | (ln.lg & 1u) << 15) << 16 |
                         (unsigned long long)(ln.a[0] | ((ln.lg >> 1) & 1u) << 15) << 32 | (unsigned long long)(ln.b[0] | ((ln.lg >> 2) & 1u) << 15) << 48);
        T.desc.push_back((unsigned long long)(ln.a[1] | f0 << 15) | (unsigned long long)(ln.b[1] | f1 << 15) << 16 |
                         (unsigned long long)(ln.a[2] | f2 << 15) << 32 | (unsigned long long)(ln.b[2] | f3 << 15) << 48);
      }
  }
  for (size_t l = 0; l < NL; ++l) {          // the kernel reads one step beyond a list's end
    T.desc.push_back((unsigned long long)zero_w | (unsigned long long)one_w << 16 | (unsigned long long)zero_w << 32 | (unsigned long long)zero_w << 48);
    T.desc.push_back((unsigned long long)zero_w | (unsigned long long)zero_w << 16 | (unsigned long long)zero_w << 32 | (unsigned long long)zero_w << 48);
  }
  return true;
}

}  // namespace cadnip
