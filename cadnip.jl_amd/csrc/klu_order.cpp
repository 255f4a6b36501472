// klu_order.cpp -- the ordering KLU applies before it factors (SuiteSparse KLU, reached by the reference through Sundials IDA
// `linear_solver = :KLU`, /root/reference/src/sweeps.jl:600, and LinearSolve's KLUFactorization, src/mna/solve.jl:612-613): a permutation to
// block upper triangular form -- maximum transversal, then the strongly connected components of the matched matrix -- and a fill-reducing
// column order inside every diagonal block.  Published algorithms, written here from their descriptions (the library itself is third-party
// and absent): Duff's depth-first maximum transversal (MC21), Tarjan's strongly connected components, and a minimum-degree order on the
// symmetrised block with elements instead of explicit fill (the quotient graph of George & Liu / AMD; degrees are recomputed exactly for the
// neighbours of a pivot rather than bounded approximately as AMD does, and nodes denser than 10 sqrt(n) are ordered last, as AMD treats dense
// rows).  symbolic.cpp takes the result as a column sequence and chooses the pivot ROW of every column by threshold partial pivoting with a
// preference for the matched entry -- KLU's own rule (its diagonal preference) -- while it eliminates on the sample values.
// Selected with CADNIP_LU_ORDER=klu (the default remains the threshold-Markowitz search: see DESIGN.md section 5 for the comparison).
#include <algorithm>
#include <vector>
#include "internal.hpp"

namespace cadnip {

// match[j] = row matched to column j (every column matched when the pattern is structurally non-singular); returns the number of matches.
// Cheap assignment first, then depth-first augmenting paths (iterative).  cptr / cind: the pattern by columns.
static int max_transversal(int n, const std::vector<int>& cptr, const std::vector<int>& cind, std::vector<int>& match) {
  match.assign(n, -1);
  std::vector<int> row_of(n, -1);                 // row -> column it is matched to
  int nm = 0;
  for (int j = 0; j < n; ++j)
    for (int p = cptr[j]; p < cptr[j + 1]; ++p)
      if (row_of[cind[p]] < 0) { row_of[cind[p]] = j; match[j] = cind[p]; ++nm; break; }
  std::vector<int> visited(n, -1), stack_col, stack_pos, cheap(cptr.begin(), cptr.begin() + n);
  for (int j0 = 0; j0 < n; ++j0) {
    if (match[j0] >= 0) continue;
    stack_col.assign(1, j0); stack_pos.assign(1, cptr[j0]);
    visited[j0] = j0;
    bool found = false;
    while (!stack_col.empty() && !found) {
      const int j = stack_col.back();
      // a free row in column j ends the path
      int& ch = cheap[j];
      while (ch < cptr[j + 1] && row_of[cind[ch]] >= 0) ++ch;
      if (ch < cptr[j + 1]) {
        int r = cind[ch];
        // augment along the stack: column stack[k] takes the row that led to stack[k + 1]
        for (int k = (int)stack_col.size() - 1; k >= 0; --k) {
          const int c = stack_col[k];
          const int prev = match[c];
          match[c] = r; row_of[r] = c;
          r = prev;
        }
        found = true; ++nm;
        break;
      }
      int& p = stack_pos.back();
      bool pushed = false;
      while (p < cptr[j + 1]) {
        const int r = cind[p++];
        const int c2 = row_of[r];
        if (c2 >= 0 && visited[c2] != j0) {
          visited[c2] = j0;
          // descend: column c2 would give up row r to column j
          stack_col.push_back(c2); stack_pos.push_back(cptr[c2]);
          pushed = true;
          break;
        }
      }
      if (!pushed) { stack_col.pop_back(); stack_pos.pop_back(); }
    }
    (void)found;
  }
  // the augmentation above reassigns rows along the stack: a column's new row is the one through which its successor was reached.  Rebuild
  // that invariant explicitly (the loop above passes `prev` upwards, which is the row the column held before -- exactly that row).
  return nm;
}

// Tarjan's SCC (iterative) on the graph  column j -> column j'  iff  row match[j] has an entry in column j'.  Components come out in reverse
// topological order; numbering them from the last found to the first gives block UPPER triangular form.  rptr / rind: pattern by rows.
static void scc_blocks(int n, const std::vector<int>& rptr, const std::vector<int>& rind, const std::vector<int>& match,
                       std::vector<int>& comp, int& n_comp) {
  comp.assign(n, -1);
  n_comp = 0;
  std::vector<int> index(n, -1), low(n, 0), stk, onstk(n, 0), cs, cp;
  int idx = 0;
  for (int s = 0; s < n; ++s) {
    if (index[s] >= 0) continue;
    cs.assign(1, s); cp.assign(1, rptr[match[s]]);
    index[s] = low[s] = idx++; stk.push_back(s); onstk[s] = 1;
    while (!cs.empty()) {
      const int j = cs.back();
      const int r = match[j];
      int& p = cp.back();
      bool descended = false;
      while (p < rptr[r + 1]) {
        const int j2 = rind[p++];
        if (j2 == j) continue;
        if (index[j2] < 0) {
          index[j2] = low[j2] = idx++; stk.push_back(j2); onstk[j2] = 1;
          cs.push_back(j2); cp.push_back(rptr[match[j2]]);
          descended = true;
          break;
        }
        if (onstk[j2]) low[j] = std::min(low[j], index[j2]);
      }
      if (descended) continue;
      if (low[j] == index[j]) {
        for (;;) { const int v = stk.back(); stk.pop_back(); onstk[v] = 0; comp[v] = n_comp; if (v == j) break; }
        ++n_comp;
      }
      cs.pop_back(); cp.pop_back();
      if (!cs.empty()) low[cs.back()] = std::min(low[cs.back()], low[j]);
    }
  }
}

// Minimum-degree order of the nodes `nodes` (global ids) of the undirected graph adj (lists may hold ids outside the set: ignored).
// Quotient graph: an eliminated node becomes an element whose variable list is its neighbourhood at elimination time; a variable's
// neighbourhood is its remaining variable neighbours plus the variables of its elements.
static void min_degree(const std::vector<int>& nodes, const std::vector<std::vector<int>>& adj, int n_total, std::vector<int>& order) {
  const int m = (int)nodes.size();
  order.clear();
  if (m == 0) return;
  if (m <= 2) { order = nodes; return; }
  std::vector<int> local(n_total, -1);
  for (int k = 0; k < m; ++k) local[nodes[k]] = k;
  std::vector<std::vector<int>> var(m), elems(m), evars;       // per variable: variable neighbours, adjacent elements; per element: its variables
  for (int k = 0; k < m; ++k)
    for (int g : adj[nodes[k]]) { const int l = g >= 0 ? local[g] : -1; if (l >= 0 && l != k) var[k].push_back(l); }
  for (auto& v : var) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  const int dense_lim = std::max(16, (int)(10.0 * std::sqrt((double)m)));
  std::vector<char> done(m, 0), dense(m, 0);
  std::vector<int> dense_nodes;
  for (int k = 0; k < m; ++k) if ((int)var[k].size() > dense_lim) { dense[k] = 1; dense_nodes.push_back(k); }
  std::vector<int> deg(m, 0), mark(m, -1);
  int stamp = 0;
  auto degree_of = [&](int i) {
    ++stamp; int d = 0;
    mark[i] = stamp;
    for (int v : var[i]) if (!done[v] && !dense[v] && mark[v] != stamp) { mark[v] = stamp; ++d; }
    for (int e : elems[i]) for (int v : evars[e]) if (!done[v] && !dense[v] && mark[v] != stamp) { mark[v] = stamp; ++d; }
    return d;
  };
  // buckets by degree (lazy: entries are validated when popped)
  std::vector<std::vector<int>> bucket(m + 1);
  for (int k = 0; k < m; ++k) if (!dense[k]) { deg[k] = degree_of(k); bucket[deg[k]].push_back(k); }
  std::vector<char> alive_e;
  int remaining = m - (int)dense_nodes.size(), dmin = 0;
  std::vector<int> Lp;
  while (remaining > 0) {
    int p = -1;
    for (; dmin <= m && p < 0; ) {
      auto& b = bucket[dmin];
      while (!b.empty()) { const int c = b.back(); b.pop_back(); if (!done[c] && deg[c] == dmin) { p = c; break; } }
      if (p < 0) ++dmin;
    }
    if (p < 0) break;
    done[p] = 1; --remaining;
    order.push_back(nodes[p]);
    // the pivot's neighbourhood becomes a new element; the elements it touched are absorbed
    ++stamp; Lp.clear();
    mark[p] = stamp;
    for (int v : var[p]) if (!done[v] && !dense[v] && mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); }
    for (int e : elems[p]) { if (!alive_e[e]) continue; for (int v : evars[e]) if (!done[v] && !dense[v] && mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); } alive_e[e] = 0; }
    const int enew = (int)evars.size();
    evars.push_back(Lp); alive_e.push_back(1);
    const int lp_stamp = stamp;
    for (int i : Lp) {
      // variable neighbours inside Lp are covered by the new element; dead elements leave the list
      auto& vi = var[i];
      vi.erase(std::remove_if(vi.begin(), vi.end(), [&](int v) { return done[v] || mark[v] == lp_stamp; }), vi.end());
      auto& ei = elems[i];
      ei.erase(std::remove_if(ei.begin(), ei.end(), [&](int e) { return !alive_e[e]; }), ei.end());
      ei.push_back(enew);
    }
    for (int i : Lp) {
      const int d = degree_of(i);
      deg[i] = d; bucket[d].push_back(i);
      if (d < dmin) dmin = d;
    }
  }
  for (int k : dense_nodes) order.push_back(nodes[k]);
}

// The whole ordering: colorder (the column pivoted at step k), match_row (per column: the row of the maximum transversal), and the block
// boundaries in colorder.  Returns false when the pattern is structurally singular (no perfect matching).
bool klu_style_order(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, std::vector<int>& colorder,
                     std::vector<int>& match_row, std::vector<int>& block_ptr) {
  // pattern by columns
  std::vector<int> cptr(n + 1, 0), cind(colidx.size());
  for (size_t p = 0; p < colidx.size(); ++p) ++cptr[colidx[p] + 1];
  for (int j = 0; j < n; ++j) cptr[j + 1] += cptr[j];
  { std::vector<int> nx(cptr.begin(), cptr.begin() + n);
    for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) cind[nx[colidx[p]]++] = i; }
  // prefer the diagonal in the cheap assignment: put row j first in column j's list when present (MNA: node diagonals are the natural pivots)
  for (int j = 0; j < n; ++j)
    for (int p = cptr[j]; p < cptr[j + 1]; ++p) if (cind[p] == j) { std::swap(cind[p], cind[cptr[j]]); break; }
  if (max_transversal(n, cptr, cind, match_row) != n) return false;
  { std::vector<char> seen(n, 0); for (int j = 0; j < n; ++j) { if (match_row[j] < 0 || seen[match_row[j]]) return false; seen[match_row[j]] = 1; } }
  std::vector<int> comp; int n_comp = 0;
  scc_blocks(n, rowptr, colidx, match_row, comp, n_comp);
  // Tarjan numbers a component when all components it reaches are numbered: component 0 reaches nothing.  Column j -> j' means row
  // match[j] has an entry in column j': with blocks ordered by DEcreasing component number the matched matrix is block upper triangular.
  std::vector<std::vector<int>> members(n_comp);
  for (int j = 0; j < n; ++j) members[n_comp - 1 - comp[j]].push_back(j);
  // symmetrised graph of the matched matrix (column ids): j ~ j' iff row match[j] has an entry in column j' or vice versa
  std::vector<std::vector<int>> adj(n);
  for (int j = 0; j < n; ++j) {
    const int r = match_row[j];
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) { const int j2 = colidx[p]; if (j2 != j && comp[j2] == comp[j]) { adj[j].push_back(j2); adj[j2].push_back(j); } }
  }
  colorder.clear(); block_ptr.assign(1, 0);
  std::vector<int> ord;
  for (auto& blk : members) {
    min_degree(blk, adj, n, ord);
    colorder.insert(colorder.end(), ord.begin(), ord.end());
    block_ptr.push_back((int)colorder.size());
  }
  return (int)colorder.size() == n;
}

}  // namespace cadnip
