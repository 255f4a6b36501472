// symbolic.cpp -- host-side symbolic phase of the GPU sparse LU (run once per structure).
//
// Stands in for KLU's analyse + first factor (SuiteSparse, third-party, reached in the
// reference through Sundials IDA `linear_solver=:KLU`, /root/reference/src/sweeps.jl:600, and
// LinearSolve's KLUFactorization with the symbolic factorisation reused,
// /root/reference/src/mna/solve.jl:612-613,667-670).  Like klu_refactor, every later numeric
// factorisation on the GPU reuses the pivot sequence chosen here.
//
// MNA matrices have structurally zero diagonals (V-source / inductor rows,
// /root/reference/src/mna/devices.jl:619-633), so the order is chosen numerically on a sample
// Jacobian: Markowitz cost (r-1)(c-1) among entries passing a relative threshold test,
// diagonal entries preferred on ties (node diagonals are sums of conductances and stay
// dominant across operating points, which is what a static order needs).
//
// Output: permuted row-major L+U pattern with fill, the J -> LU load map, an entry-wise
// left-looking program  lu[e] = (lu[e] - sum_k lu[a_k]*lu[b_k]) [/ pivot]  whose entries are
// grouped into dependency levels (each LU entry is written exactly once, by one lane, no
// atomics), and level schedules for the two triangular solves.
#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <unordered_map>
#include <stdio.h>
#include <stdlib.h>
#include "internal.hpp"

namespace cadnip {

static const int LU_EXHAUSTIVE_MAX = 4096;   // unknowns up to which every remaining entry is a pivot candidate
static const int LU_SEARCH_ROWS = 12;        // beyond: candidates come from this many shortest rows

static int lu_analyze_mode(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& vals,
                           double pivot_tol, bool magnitudes, const LULeaves* leaves, LUProgram& out, std::string& err);

// The pivot search eliminates numerically on the sample.  A sample assembled from several operating points is not
// the Jacobian of any one state, so cancellation in it can fake a singular matrix (two nodes tied by the same huge
// sampled conductance).  When the signed elimination fails, the search is repeated on magnitudes
// (|a_ij| + |l_ik u_kj|, no cancellation): a structurally non-singular pattern then always yields an order.
// `sample` = vals is such a composite; for the Jacobian of one state a singular result is reported as is.
// `leaves` (optional): the unknowns from leaves->q_begin on are charge states, from leaves->lim_begin on limit variables -- the
// device-local unknowns of Cadnip's formulation (contrib.jl:356-375 charge rows q - 1e12 Q(V) = 0, vasim.jl:3110-3138 tracking rows
// u_l - V(probe) = 0).  Their diagonal is a stamped constant 1, their rows and columns touch only their own device's unknowns, and they
// do not touch each other except charge rows reading limit columns.  They are therefore pivoted FIRST, on their own diagonals -- limits,
// then charges: no fill (every update lands on an entry the device stamps anyway), dependency depth two, and what remains for the
// Markowitz search is the node / branch-current system (DFF: 25 of 235 unknowns; c6288: 5 124 of 75 908).  A leaf whose diagonal fails
// the threshold test (or is missing) is left to the general search.  `unit[k]` marks pivots whose value is the constant 1 (structure says
// one constant stamp, the sample says exactly 1.0): entries divided by them need no division.
int lu_analyze(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& vals,
               double pivot_tol, bool sample, LUProgram& out, std::string& err, const LULeaves* leaves) {
  // The program is built aside and moved into `out` only on success: a failed analysis (a singular re-pivot victim,
  // driver.hip) must leave the caller's previous program -- and the device arrays uploaded from it -- usable.
  LUProgram fresh;
  int rc = lu_analyze_mode(n, rowptr, colidx, vals, pivot_tol, false, leaves, fresh, err);
  if (rc == CADNIP_SINGULAR && sample) {
    std::string err2;
    int rc2 = lu_analyze_mode(n, rowptr, colidx, vals, pivot_tol, true, leaves, fresh, err2);
    if (rc2 == CADNIP_OK) { err.clear(); rc = rc2; }
  }
  if (rc == CADNIP_OK) out = std::move(fresh);
  return rc;
}

static int lu_analyze_mode(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& vals,
                           double pivot_tol, bool magnitudes, const LULeaves* leaves, LUProgram& out, std::string& err) {
  out = LUProgram();
  out.n = n;
  out.unit.assign(n, 0);
  // leaf unknowns in the order they are tried: limits, then charges
  std::vector<int> leaf_order;
  const bool have_leaves = leaves && leaves->q_begin >= 0 && leaves->q_begin <= leaves->lim_begin && leaves->lim_begin <= n;
  std::vector<char> unit_cand(n, 0);          // pivot k sits on a leaf's own diagonal whose stamped value is the constant 1
  std::vector<double> diag0(n, 0.0);          // the sample's diagonal before any elimination
  for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) if (colidx[p] == i) diag0[i] = vals[p];
  if (have_leaves && leaves->first) {
    for (int i = leaves->lim_begin; i < n; ++i) leaf_order.push_back(i);
    for (int i = leaves->q_begin; i < leaves->lim_begin; ++i) leaf_order.push_back(i);
  }
  size_t leaf_next = 0;
  std::vector<std::map<int, double>> rows(n);
  std::vector<std::set<int>> cols(n);
  for (int i = 0; i < n; ++i)
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
      rows[i][colidx[p]] = magnitudes ? std::fabs(vals[p]) : vals[p];
      cols[colidx[p]].insert(i);
    }
  std::vector<char> rdone(n, 0);
  std::vector<std::vector<int>> Urow(n), Lcol(n);
  out.rperm.assign(n, -1);
  out.cperm.assign(n, -1);
  std::vector<double> colmax(n);
  std::vector<int> colmax_stamp(n, -1);
  // Exhaustive Markowitz search costs O(n * nnz): fine for the circuits the fused kernel takes (the DFF: 235 unknowns),
  // minutes at 76 k unknowns (a 16 x 16 multiplier).  Large systems search the LU_SEARCH_ROWS shortest remaining rows
  // only (rows are kept ordered by their current length), the classic restricted Markowitz strategy; the pivot rule inside
  // the searched rows is the same.
  const bool restricted = n > LU_EXHAUSTIVE_MAX;
  // CADNIP_LU_ORDER=klu: KLU's ordering instead of the Markowitz search (klu_order.cpp) -- block triangular form, minimum degree inside the
  // blocks; the pivot row of every column by threshold partial pivoting on the sample, the matched entry preferred (KLU's diagonal preference)
  std::vector<int> klu_cols, klu_match, klu_blocks;
  bool klu = false;
  // Default: the Markowitz search for the circuits of a sweep (n <= 4 096: on the flip-flop it gives 14 dependency levels and 194 multiply-adds
  // where the KLU-style order gives 23 and 339 -- device-local unknowns first, constant-1 pivots), KLU's ordering for a single large circuit
  // (c6288, n = 75 908: 241 levels and 272 k multiply-adds instead of 322 and 694 k, 4 % less fill).  CADNIP_LU_ORDER = klu | markowitz forces one.
  bool want_klu = restricted;
  if (const char* e = getenv("CADNIP_LU_ORDER")) want_klu = e[0] == 'k' || e[0] == 'K';
  std::vector<double> klu_rscale;
  if (want_klu) {
    klu_rscale.assign(n, 1.0);
    for (int i = 0; i < n; ++i) {
      double mx = 0;
      for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) mx = std::max(mx, std::fabs(vals[p]));
      if (mx > 0.0 && std::isfinite(mx)) klu_rscale[i] = 1.0 / mx;
    }
    klu = klu_style_order(n, rowptr, colidx, klu_cols, klu_match, klu_blocks);
    if (!klu) { err = "matrix is structurally singular (no perfect matching)"; return CADNIP_SINGULAR; }
    out.n_blocks = (int)klu_blocks.size() - 1;
    leaf_order.clear();
    if (getenv("CADNIP_LU_DEBUG")) {
      int big = 0; for (size_t b = 0; b + 1 < klu_blocks.size(); ++b) big = std::max(big, klu_blocks[b + 1] - klu_blocks[b]);
      fprintf(stderr, "[cadnip lu] KLU-style order: %d diagonal blocks, the largest of %d unknowns\n", out.n_blocks, big);
    }
  }
  std::set<std::pair<int, int>> by_len;      // (current row length, row) of the rows not yet eliminated
  if (restricted) for (int i = 0; i < n; ++i) by_len.insert({(int)rows[i].size(), i});
  std::vector<int> cand;
  for (int k = 0; k < n; ++k) {
    long best_cost = -1;
    int bi = -1, bj = -1;
    bool bdiag = false;
    double brel = 0;
    cand.clear();
    // ---- leaf phase: the next device-local unknown whose own diagonal passes the threshold test
    while (leaf_next < leaf_order.size() && bi < 0) {
      const int i = leaf_order[leaf_next++];
      if (rdone[i]) continue;
      auto it = rows[i].find(i);
      if (it == rows[i].end()) continue;
      const double av = std::fabs(it->second);
      if (!(av > 0.0) || !std::isfinite(av)) continue;
      double m = 0;
      for (int ii : cols[i]) m = std::max(m, std::fabs(rows[ii][i]));
      if (av < pivot_tol * m) continue;
      bi = bj = i;
    }
    if (bi < 0 && restricted) {
      // the shortest rows; keep going while nothing acceptable was found (threshold test) -- done below by widening
      for (auto it = by_len.begin(); it != by_len.end() && (int)cand.size() < LU_SEARCH_ROWS; ++it) cand.push_back(it->second);
    }
    // one candidate entry (i, j) against the best so far: Markowitz cost, threshold test, diagonal preferred on ties, then the larger
    // relative magnitude
    auto consider = [&](int i, int j, double val, long cost) {
      const double av = std::fabs(val);
      if (!(av > 0.0) || !std::isfinite(av)) return;
      if (best_cost >= 0 && cost > best_cost) return;
      if (colmax_stamp[j] < 0) {          // (valid until an elimination step touches the column: a rail's column is long and asked for often)
        double m = 0;
        for (int ii : cols[j]) m = std::max(m, std::fabs(rows[ii][j]));
        colmax[j] = m;
        colmax_stamp[j] = 0;
      }
      if (av < pivot_tol * colmax[j]) return;
      const double rel = av / colmax[j];
      const bool diag = (i == j);
      bool better = false;
      if (best_cost < 0 || cost < best_cost) better = true;
      else if (diag != bdiag) better = diag;
      else if (rel > brel * (1 + 1e-12)) better = true;
      if (better) { best_cost = cost; bi = i; bj = j; bdiag = diag; brel = rel; }
    };
    if (klu) {
      // partial pivoting in column j on ROW-SCALED magnitudes (KLU scales rows by their largest entry before it factors: MNA rows differ by
      // twenty orders of magnitude -- a charge row carries 1 and 1e12 dQ/dV, a node row conductances down to gmin -- and unscaled column
      // pivoting loses the flip-flop's Jacobian entirely); the matched entry is taken when it is within pivot_tol of the best
      const int j = klu_cols[k];
      double m = 0;
      for (int ii : cols[j]) m = std::max(m, std::fabs(rows[ii][j]) * klu_rscale[ii]);
      const int pref = klu_match[j];
      if (m > 0.0 && std::isfinite(m)) {
        auto it = rows[pref].find(j);
        if (!rdone[pref] && it != rows[pref].end() && std::fabs(it->second) * klu_rscale[pref] >= pivot_tol * m && std::fabs(it->second) > 0.0) { bi = pref; bj = j; }
        else for (int ii : cols[j]) if (std::fabs(rows[ii][j]) * klu_rscale[ii] == m) { bi = ii; bj = j; break; }
      }
    } else if (bi < 0 && !restricted) {
      for (int i = 0; i < n; ++i) {
        if (rdone[i]) continue;
        const long r = (long)rows[i].size();
        for (auto& kv : rows[i]) consider(i, kv.first, kv.second, (r - 1) * ((long)cols[kv.first].size() - 1));
      }
    } else if (bi < 0) {
      // The entries of the searched rows in ascending cost (stable: equal costs keep the row-by-row order, so the choice is the one a plain
      // scan makes): the column maximum of the threshold test is then taken for cheap candidates only -- a scan in row order asked for it
      // whenever no acceptable candidate had been seen yet, and for an entry in a supply rail's column (10 k rows in the c6288 multiplier)
      // that is 10 k look-ups, at nearly every pivot step: 9.5 of the 10 s the search took there.
      struct Cnd { long cost; int i, j; double v; };
      std::vector<Cnd> cs;
      auto gather = [&]() {
        cs.clear();
        for (int i : cand) {
          if (rdone[i]) continue;
          const long r = (long)rows[i].size();
          for (auto& kv : rows[i]) cs.push_back(Cnd{(r - 1) * ((long)cols[kv.first].size() - 1), i, kv.first, kv.second});
        }
        std::stable_sort(cs.begin(), cs.end(), [](const Cnd& x, const Cnd& y) { return x.cost < y.cost; });
      };
      gather();
      for (const Cnd& c : cs) { if (best_cost >= 0 && c.cost > best_cost) break; consider(c.i, c.j, c.v, c.cost); }
      if (bi < 0 && cand.size() < by_len.size()) {      // widen: nothing passed the threshold among the shortest rows
        cand.clear();
        for (auto& pr : by_len) cand.push_back(pr.second);
        gather();
        for (const Cnd& c : cs) { if (best_cost >= 0 && c.cost > best_cost) break; consider(c.i, c.j, c.v, c.cost); }
      }
    }
    if (bi < 0) { err = "matrix is singular at pivot step " + std::to_string(k); return CADNIP_SINGULAR; }
    out.rperm[k] = bi;
    out.cperm[k] = bj;
    if (have_leaves && bi == bj && bi >= leaves->q_begin && leaves->unit_ok && leaves->unit_ok[bi] && diag0[bi] == 1.0) unit_cand[k] = 1;
    double piv = rows[bi][bj];
    std::vector<int> lrows;
    for (int i : cols[bj]) if (i != bi) lrows.push_back(i);
    for (auto& kv : rows[bi]) Urow[k].push_back(kv.first);
    Lcol[k] = lrows;
    if (restricted) {
      by_len.erase({(int)rows[bi].size(), bi});
      for (int i : lrows) by_len.erase({(int)rows[i].size(), i});
    }
    for (int i : lrows) {
      double f = rows[i][bj] / piv;
      for (auto& kv : rows[bi]) {
        int j = kv.first;
        if (j == bj) continue;
        auto it = rows[i].find(j);
        if (it == rows[i].end()) { rows[i][j] = magnitudes ? std::fabs(f * kv.second) : -f * kv.second; cols[j].insert(i); }
        else if (magnitudes) it->second = std::fabs(it->second) + std::fabs(f * kv.second);
        else it->second -= f * kv.second;
        colmax_stamp[j] = -1;
      }
      rows[i].erase(bj);
      if (restricted) by_len.insert({(int)rows[i].size(), i});
    }
    for (auto& kv : rows[bi]) { cols[kv.first].erase(bi); colmax_stamp[kv.first] = -1; }
    cols[bj].clear();
    rdone[bi] = 1;
  }
  std::vector<int> pinv(n), qinv(n);
  for (int k = 0; k < n; ++k) { pinv[out.rperm[k]] = k; qinv[out.cperm[k]] = k; }
  // permuted row-major L+U pattern
  std::vector<std::vector<int>> prow(n);
  for (int k = 0; k < n; ++k) {
    for (int j : Urow[k]) prow[k].push_back(qinv[j]);
    for (int i : Lcol[k]) prow[pinv[i]].push_back(k);
  }
  out.lu_rowptr.assign(n + 1, 0);
  out.lu_diag.assign(n, -1);
  for (int i = 0; i < n; ++i) {
    std::sort(prow[i].begin(), prow[i].end());
    prow[i].erase(std::unique(prow[i].begin(), prow[i].end()), prow[i].end());
    out.lu_rowptr[i + 1] = out.lu_rowptr[i] + (int)prow[i].size();
  }
  out.nnz_lu = out.lu_rowptr[n];
  out.lu_col.resize(out.nnz_lu);
  std::unordered_map<long long, int> posmap;
  posmap.reserve((size_t)out.nnz_lu * 2);
  for (int i = 0; i < n; ++i)
    for (size_t t = 0; t < prow[i].size(); ++t) {
      int p = out.lu_rowptr[i] + (int)t;
      out.lu_col[p] = prow[i][t];
      posmap[(long long)i * n + prow[i][t]] = p;
      if (prow[i][t] == i) out.lu_diag[i] = p;
    }
  for (int i = 0; i < n; ++i)
    if (out.lu_diag[i] < 0) { err = "internal: missing diagonal in LU pattern"; return CADNIP_SINGULAR; }
  // load map
  for (int i = 0; i < n; ++i)
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
      auto it = posmap.find((long long)pinv[i] * n + qinv[colidx[p]]);
      if (it == posmap.end()) { err = "internal: J entry outside LU pattern"; return CADNIP_BADARG; }
      out.load_src.push_back(p);
      out.load_dst.push_back(it->second);
    }
  // entry-wise program + levels
  std::vector<int> level(out.nnz_lu, 0);
  struct Ent { int pos, diag, lvl; std::vector<int> a, b; };
  std::vector<Ent> ents;
  // a candidate pivot IS the constant 1 if, besides, no elimination step updates its diagonal entry (no term in its recurrence):
  // decided here, row by row -- row j's diagonal is complete before any later row divides by it
  // Terms of row i: for every L(i,k), k ascending, the entries (k,j), j > k, of row k that row i also holds -- found through a scatter of
  // row i's positions (O(multiply-adds); looking every (k,j) pair of a row up in a hash map was quadratic in the row length: 167 M
  // look-ups, half of the 13 s this phase took for the c6288 multiplier, whose supply rails have rows of thousands of entries).  An
  // entry's terms arrive in ascending k, the order the look-up produced.
  {
    std::vector<int> wpos(n, -1);
    std::vector<std::vector<int>> ta, tb;
    for (int i = 0; i < n; ++i) {
      const int r0 = out.lu_rowptr[i], r1 = out.lu_rowptr[i + 1], dp = out.lu_diag[i];
      for (int p = r0; p < r1; ++p) wpos[out.lu_col[p]] = p;
      ta.assign((size_t)(r1 - r0), std::vector<int>()); tb.assign((size_t)(r1 - r0), std::vector<int>());
      for (int pl = r0; pl < dp; ++pl) {
        const int kk = out.lu_col[pl];
        for (int pu = out.lu_diag[kk] + 1; pu < out.lu_rowptr[kk + 1]; ++pu) {
          const int q = wpos[out.lu_col[pu]];
          if (q >= 0) { ta[(size_t)(q - r0)].push_back(pl); tb[(size_t)(q - r0)].push_back(pu); }
        }
      }
      for (int p = r0; p < r1; ++p) {
        const int j = out.lu_col[p];
        Ent e;
        e.pos = p;
        e.diag = (j < i && !out.unit[j]) ? out.lu_diag[j] : -1;     // L entry: divided by its pivot, unless that is the constant 1
        e.a = std::move(ta[(size_t)(p - r0)]); e.b = std::move(tb[(size_t)(p - r0)]);
        int lvl = 0;
        for (size_t t = 0; t < e.a.size(); ++t) lvl = std::max(lvl, std::max(level[e.a[t]], level[e.b[t]]) + 1);
        if (j == i) out.unit[i] = (unit_cand[i] && e.a.empty()) ? 1 : 0;
        if (e.diag >= 0) lvl = std::max(lvl, level[e.diag] + 1);
        level[p] = lvl;
        e.lvl = lvl;
        if (!e.a.empty() || e.diag >= 0) ents.push_back(std::move(e));
      }
      for (int p = r0; p < r1; ++p) wpos[out.lu_col[p]] = -1;
    }
  }
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.lvl < y.lvl; });
  int nlev = 0;
  for (auto& e : ents) nlev = std::max(nlev, e.lvl + 1);
  out.lev_ptr.assign(1, 0);
  out.ent_ptr.assign(1, 0);
  {
    int cur = ents.empty() ? 0 : ents[0].lvl;
    // levels start at the first level that has work
    for (size_t t = 0; t < ents.size(); ++t) {
      if (ents[t].lvl != cur) { out.lev_ptr.push_back((int)t); cur = ents[t].lvl; }
      out.ent_pos.push_back(ents[t].pos);
      out.ent_diag.push_back(ents[t].diag);
      out.term_a.insert(out.term_a.end(), ents[t].a.begin(), ents[t].a.end());
      out.term_b.insert(out.term_b.end(), ents[t].b.begin(), ents[t].b.end());
      out.ent_ptr.push_back((int)out.term_a.size());
    }
    out.lev_ptr.push_back((int)ents.size());
  }
  // triangular solve schedules
  std::vector<int> fl(n, 0), bl(n, 0);
  std::vector<std::pair<int, int>> fr, br;
  for (int i = 0; i < n; ++i) {
    int lv = 0;
    bool any = false;
    for (int p = out.lu_rowptr[i]; p < out.lu_diag[i]; ++p) { lv = std::max(lv, fl[out.lu_col[p]] + 1); any = true; }
    fl[i] = lv;
    if (any) fr.push_back({lv, i});
  }
  for (int i = n - 1; i >= 0; --i) {
    int lv = 0;
    for (int p = out.lu_diag[i] + 1; p < out.lu_rowptr[i + 1]; ++p) lv = std::max(lv, bl[out.lu_col[p]] + 1);
    bl[i] = lv;
    br.push_back({lv, i});
  }
  auto pack = [](std::vector<std::pair<int, int>>& v, std::vector<int>& rws, std::vector<int>& ptr) {
    std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
    ptr.assign(1, 0);
    for (size_t t = 0; t < v.size(); ++t) {
      if (t > 0 && v[t].first != v[t - 1].first) ptr.push_back((int)t);
      rws.push_back(v[t].second);
    }
    ptr.push_back((int)v.size());
  };
  pack(fr, out.fwd_rows, out.fwd_lev_ptr);
  pack(br, out.bwd_rows, out.bwd_lev_ptr);
  return CADNIP_OK;
}

}  // namespace cadnip
