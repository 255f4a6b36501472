// f2_program.cpp -- host side of the fused kernel's linear solve: the entry program (csrc/fused2.hip executes it).
// Pure C++ (no HIP): also reachable through the host-only C ABI (cadnip_host_f2_*) so that the CPU test-suite can
// emulate the program pass by pass.
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
bool f2_build_program(const LUProgram& P, int n, int nc, F2Program& G) {
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
  std::vector<F2Ent> pre, post;
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
        x.a.push_back(G.posW[pl]); x.b.push_back(G.posW[pu]);
        x.lvl = std::max(x.lvl, std::max(lev[pl], lev[pu]));
      }
      if (j < i && !unit(j)) { x.dg = G.posW[P.lu_diag[j]]; x.lvl = std::max(x.lvl, lev[P.lu_diag[j]]); }   // L entry / pivot (a constant-1 pivot divides nothing)
      if (x.a.empty() && x.dg < 0) continue;
      x.lvl += 1;
      lev[p] = x.lvl;
      pre.push_back(std::move(x));
    }
  for (int i = cs0; i < n; ++i)
    for (int j = cs0; j < n; ++j) {
      F2Ent x; x.pos = G.dn0 + (i - cs0) * nc + (j - cs0); x.dg = -1; x.lvl = -1;
      for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
        const int k = P.lu_col[pl];
        if (k >= cs0) break;
        const int pu = find(k, j);
        if (pu < 0) continue;
        x.a.push_back(G.posW[pl]); x.b.push_back(G.posW[pu]);
        x.lvl = std::max(x.lvl, std::max(lev[pl], lev[pu]));
      }
      if (x.a.empty()) continue;
      x.lvl += 1;
      pre.push_back(std::move(x));
    }
  std::vector<int> ylev(n, -1);
  for (int i = 0; i < n; ++i) {
    F2Ent x; x.pos = y0 + i; x.dg = -1; x.lvl = -1;
    for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
      const int k = P.lu_col[pl];
      if (k >= cs0) break;
      x.a.push_back(G.posW[pl]); x.b.push_back(y0 + k);
      x.lvl = std::max(x.lvl, std::max(lev[pl], ylev[k]));
    }
    if (x.a.empty()) continue;
    x.lvl += 1;
    ylev[i] = x.lvl;
    pre.push_back(std::move(x));
  }
  std::vector<int> xlev(n, -1);
  for (int i = cs0 - 1; i >= 0; --i) {
    F2Ent x; x.pos = y0 + i; x.dg = unit(i) ? -1 : G.posW[P.lu_diag[i]]; x.lvl = -1;
    for (int pu = P.lu_diag[i] + 1; pu < P.lu_rowptr[i + 1]; ++pu) {
      const int j = P.lu_col[pu];
      x.a.push_back(G.posW[pu]); x.b.push_back(y0 + j);
      if (j < cs0) x.lvl = std::max(x.lvl, xlev[j]);
    }
    if (x.a.empty() && x.dg < 0) continue;     // x_i = y_i: nothing to do (xlev stays -1: final as forward substitution left it)
    x.lvl += 1;
    xlev[i] = x.lvl;
    post.push_back(std::move(x));
  }
  // forward substitution on KEPT factors (Newton mode 1, a round that does not refactor): the y recurrences alone; every L entry is
  // final, so a row's level is one more than the deepest y it reads
  std::vector<F2Ent> fwd;
  {
    std::vector<int> fl(n, -1);
    for (int i = 0; i < n; ++i) {
      F2Ent x; x.pos = y0 + i; x.dg = -1; x.lvl = -1;
      for (int pl = P.lu_rowptr[i]; pl < P.lu_diag[i]; ++pl) {
        const int k = P.lu_col[pl];
        if (k >= cs0) break;
        x.a.push_back(G.posW[pl]); x.b.push_back(y0 + k);
        x.lvl = std::max(x.lvl, fl[k]);
      }
      if (x.a.empty()) continue;
      x.lvl += 1;
      fl[i] = x.lvl;
      fwd.push_back(std::move(x));
    }
  }
  if (!f2_emit_passes(pre, G, G.n_pre) || !f2_emit_passes(post, G, G.n_post) || !f2_emit_passes(fwd, G, G.n_fwd)) return false;
  G.terms.push_back(0);   // a lane without terms still prefetches term[t0]
  if (G.terms.size() >= 65535) return false;
  // wave cycles measured on the DFF (tools/trace_fused2.py): ~950 per pass; dense solve 2.7 k at nc = 8, 4.3 k at nc = 12
  G.cost = 950.0 * (G.n_pre + G.n_post) + (nc ? 1450.0 + 19.5 * nc * nc : 0.0);
  return true;
}


}  // namespace cadnip
