// fused2.hip -- the fused transient Newton kernel: WPB sweep instances per workgroup, one 64-lane wave each,
// the circuit *structure* resident in LDS and shared by the workgroup's instances, the instance's Jacobian,
// right-hand side, solution and BDF history term resident in LDS for all the Newton rounds of a launch.
//
// One Newton round of one instance (DFF class: n = 235, 1091 LU entries) is a chain of short dependent phases.
// Nothing here is HBM-bound; the cost is LDS round trips on the critical path, so the kernel is organised to
// keep that chain short:
//   * no slot buffer and no G / C: each stamp value is accumulated straight into the instance's LDS-resident
//        J = G + a0*C           (at its LU position, ds_add_f64)
//        r = C*du + G*u - b = J*u + C*beta - b      (du = a0*u + beta, BDF)   (in pivot-row order)
//     Stamps are branch-free: a slot whose row or column is ground accumulates into a per-lane trash word, so
//     the table read and the atomic of consecutive stamps pipeline instead of waiting on each other.
//   * refactorisation, forward substitution and back substitution are ONE entry-wise program
//        W[pos] = (W[pos] - sum_k W[a_k]*W[b_k]) [/ W[piv]]
//     over the work array W = [ LU | rhs ]: forward substitution is the LU recurrence of an extra column, so its
//     entries join the factorisation's dependency levels instead of forming a second sweep; back substitution
//     follows in place.  Entries and terms are packed (8 B / 4 B), the dot product of an entry is spread over up
//     to 16 lanes and summed with DPP, level descriptors are prefetched one level ahead.
//   * the step controller's scalars live in registers for the whole launch, u and beta in LDS; HBM is touched
//     for device parameters, the predictor / history vectors and the outputs.
//   * a wave never waits for another wave: all synchronisation is wave-level (tran_ctrl.hpp).
// Summation order inside an nz differs from the per-op path (slot-major instead of COO order), so results
// agree with it to rounding (1e-13 relative), not bit for bit; the per-op path remains the reference ABI.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "fused_team_kernel.hpp"

namespace cadnip {

// ---- host side: build the packed tables once per (structure, LU program) ------------------------------------
struct F2Tables {
  std::vector<unsigned> data;   // 32-bit words
  int off[S_NSEC];
  void begin(int which) { if (data.size() & 1) data.push_back(0); off[which] = (int)data.size(); }
  void add16(const std::vector<int>& v) {
    for (size_t i = 0; i < v.size(); i += 2) data.push_back((unsigned)(v[i] & 0xFFFF) | ((i + 1 < v.size() ? (unsigned)(v[i + 1] & 0xFFFF) : 0u) << 16));
  }
  void add32(unsigned x) { data.push_back(x); }
  void add64(u64 x) { data.push_back((unsigned)x); data.push_back((unsigned)(x >> 32)); }
};

static bool f2_prepare(CadnipHandle* h, F2Tables& T, const std::vector<int>& g_ptr, const std::vector<int>& g_slots, const std::vector<int>& c_ptr,
                       const std::vector<int>& c_slots, const std::vector<int>& b_ptr, const std::vector<int>& b_slots) {
  const LUProgram& P = h->lu;
  const int n = h->n;
  if (n >= 32767) return false;
  // core size: the cheapest of {0, 8, 12, 16}; CADNIP_F2_NC forces one (diagnostic)
  F2Program G;
  {
    bool any = false;
    const char* force = getenv("CADNIP_F2_NC");
    for (int nc : {0, 8, 12, F2_NCMAX}) {
      if (nc > n || (force && atoi(force) != nc)) continue;
      F2Program C;
      if (!f2_build_program(P, n, nc, C)) continue;
      if (!any || C.cost < G.cost) { G = std::move(C); any = true; }
    }
    if (!any) return false;
  }
  const int y0 = G.lu_words, trash0 = G.lu_words + n;                 // W offsets
  if (trash0 + F2_TRASH >= 65535) return false;
  h->f2_lu_words = G.lu_words; h->f2_nc = G.nc; h->f2_dn0 = G.dn0; h->f2_n_pre = G.n_pre; h->f2_n_post = G.n_post; h->f2_n_fwd = G.n_fwd;
  std::vector<int> pinv(n), qinv(n);
  for (int k = 0; k < n; ++k) { pinv[P.rperm[k]] = k; qinv[P.cperm[k]] = k; }
  std::vector<int> dst(h->nnz, 0);
  for (size_t k = 0; k < P.load_src.size(); ++k) dst[P.load_src[k]] = G.posW[P.load_dst[k]];
  // slot -> W offset.  A slot that no nz gathers (ground row / column) goes to the trash word of its lane.
  std::vector<int> gs(h->ns_g, -1), cpos(h->ns_c, -1), crow(h->ns_c, 0), ccol(h->ns_c, 0), br(h->ns_b, -1);
  for (int i = 0; i < n; ++i)
    for (int e = h->h_rowptr[i]; e < h->h_rowptr[i + 1]; ++e) {
      for (int p = g_ptr[e]; p < g_ptr[e + 1]; ++p) gs[g_slots[p]] = dst[e];
      for (int p = c_ptr[e]; p < c_ptr[e + 1]; ++p) { cpos[c_slots[p]] = dst[e]; crow[c_slots[p]] = y0 + pinv[i]; ccol[c_slots[p]] = h->h_colidx[e]; }
    }
  for (int i = 0; i < n; ++i) for (int p = b_ptr[i]; p < b_ptr[i + 1]; ++p) br[b_slots[p]] = y0 + pinv[i];
  // the lane that writes slot s of a block is ((s - base) % count) % 64
  std::vector<int> lane_g(h->ns_g, 0), lane_c(h->ns_c, 0), lane_b(h->ns_b, 0);
  for (auto& b : h->blocks) {
    if (b.count == 0) continue;
    for (int s = 0; s < b.n_g * b.count; ++s) lane_g[b.g_base + s] = (s % b.count) & 63;
    for (int s = 0; s < b.n_c * b.count; ++s) lane_c[b.c_base + s] = (s % b.count) & 63;
    for (int s = 0; s < b.n_b * b.count; ++s) lane_b[b.b_base + s] = (s % b.count) & 63;
  }
  for (int s = 0; s < h->ns_g; ++s) if (gs[s] < 0) gs[s] = trash0 + lane_g[s];
  for (int s = 0; s < h->ns_b; ++s) if (br[s] < 0) br[s] = trash0 + lane_b[s];
  // ---- section order: [pass program, load map | permutations] = the prefix the per-op program LU (lu_f2.hip) stages; [permutations | stamp
  // tables, node tables] = the range the lean kernels stage (their linear solve runs from step descriptors, fused2_kernel.hpp: run_steps);
  // the J*u list of the assembled-residual variant last.  Range boundaries are multiples of 4 words (16-byte copies, aligned work arrays).
  auto pad4 = [&]() { while (T.data.size() & 3) T.data.push_back(0); };
  T.begin(S_ENT); for (u64 wv : G.lanes) T.add64(wv);
  T.begin(S_TERM); for (unsigned t : G.terms) T.add32(t);
  T.begin(S_LEV); for (u64 wv : G.passes) T.add64(wv);
  T.add64(0); T.add64(0);   // two empty passes: the kernel reads pass descriptors two ahead
  T.begin(S_LOADPOS); T.add16(dst);                 // csr entry -> W word: the per-op program LU loads J = G + gamma C through it
  pad4();
  h->f2_lean_lo = (int)T.data.size();
  std::vector<int> qoff(n);
  for (int j = 0; j < n; ++j) qoff[j] = y0 + qinv[j];
  T.begin(S_QINV); T.add16(qoff);
  {
    std::vector<int> rowoff(n);                       // unknown index -> rhs word of its row (direct residuals, devices.hpp Rn)
    for (int i = 0; i < n; ++i) rowoff[i] = y0 + pinv[i];
    T.begin(S_ROWOF); T.add16(rowoff);
  }
  pad4();
  h->f2_lu_len = (int)T.data.size();
  T.begin(S_GPOS); T.add16(gs);
  T.begin(S_CDESC);
  for (int s = 0; s < h->ns_c; ++s) {
    if (cpos[s] < 0) T.add64(pack4(trash0 + lane_c[s], trash0 + lane_c[s], 0, 0));
    else T.add64(pack4(cpos[s], crow[s], ccol[s], 0));
  }
  T.begin(S_BROW); T.add16(br);
  T.begin(S_NODES);
  h->f2_nodes_off.clear();
  {
    std::vector<int> all;
    for (auto& b : h->blocks) { h->f2_nodes_off.push_back((int)all.size()); all.insert(all.end(), b.h_nodes.begin(), b.h_nodes.end()); }
    T.add16(all);
  }
  pad4();
  h->f2_lean_end = (int)T.data.size();
  T.begin(S_NZ);
  {
    // J*u adds every entry's product into its row with an LDS atomic, 64 entries per instruction.  In CSR order a long
    // row would put up to 64 same-address atomics into one instruction; ordered by (position within the row, row) the
    // entries of one instruction belong to different rows.
    std::vector<std::pair<int, int>> order;   // (rank in row, csr entry)
    for (int i = 0; i < n; ++i) for (int e = h->h_rowptr[i]; e < h->h_rowptr[i + 1]; ++e) order.push_back({e - h->h_rowptr[i], e});
    std::stable_sort(order.begin(), order.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
    std::vector<int> row_of(h->nnz);
    for (int i = 0; i < n; ++i) for (int e = h->h_rowptr[i]; e < h->h_rowptr[i + 1]; ++e) row_of[e] = i;
    for (auto& oe : order) { const int e = oe.second; T.add64(pack4(dst[e], y0 + pinv[row_of[e]], h->h_colidx[e], 0)); }
  }
  pad4();                                          // the work arrays behind the tables stay 16-byte aligned
  h->f2_lds_len = (int)T.data.size();              // what the full-table kernels copy to LDS ends here
  // ---- one wave per instance, lean variant: list-scheduled steps with three terms per lane (f2_build_steps); the kernel stages them in LDS
  {
    F2Team TM;
    if (h->d_steps1) { (void)hipFree(h->d_steps1); h->d_steps1 = nullptr; }
    h->steps1_len = 0;
    if (f2_build_steps(P, n, G.nc, 1, TM) && TM.lu_words == G.lu_words &&
        hipMalloc((void**)&h->d_steps1, TM.desc.size() * sizeof(unsigned long long)) == hipSuccess) {
      if (hipMemcpy(h->d_steps1, TM.desc.data(), TM.desc.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) == hipSuccess) {
        for (int li = 0; li < 3; ++li) h->steps1[li] = TM.n_steps[li];
        h->steps1_len = (int)TM.desc.size();
      } else { (void)hipFree(h->d_steps1); h->d_steps1 = nullptr; }
    }
  }
  // ... and for a team of four waves per instance: the per-op step LU (lu_f2.hip: k_lu_steps), descriptors read from global memory
  {
    F2Team TM;
    if (h->d_steps4) { (void)hipFree(h->d_steps4); h->d_steps4 = nullptr; }
    if (f2_build_steps(P, n, G.nc, 4, TM) && TM.lu_words == G.lu_words &&
        hipMalloc((void**)&h->d_steps4, TM.desc.size() * sizeof(unsigned long long)) == hipSuccess) {
      if (hipMemcpy(h->d_steps4, TM.desc.data(), TM.desc.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) == hipSuccess) {
        for (int li = 0; li < 3; ++li) h->steps4[li] = TM.n_steps[li];
      } else { (void)hipFree(h->d_steps4); h->d_steps4 = nullptr; }
    }
  }
  // ---- team kernel (fused_team_kernel.hpp): the same program as straight-line steps for teams of 2 and 4 waves (f2_build_team); the
  // kernel stages the descriptors in LDS
  for (int k = 0; k < 2; ++k) {
    F2Team TM;
    if (h->d_team_desc[k]) { (void)hipFree(h->d_team_desc[k]); h->d_team_desc[k] = nullptr; }
    // teams of two: three-term list-scheduled steps; teams of four: one-term level-aligned steps (fused_team_kernel.hpp: run_steps)
    if (!(k == 0 ? f2_build_steps(P, n, G.nc, 2, TM) : f2_build_team(P, n, G.nc, 4, TM)) || TM.lu_words != G.lu_words) continue;     // (no team kernel for this circuit then)
    if (hipMalloc((void**)&h->d_team_desc[k], TM.desc.size() * sizeof(unsigned long long)) != hipSuccess) { h->d_team_desc[k] = nullptr; continue; }
    if (hipMemcpy(h->d_team_desc[k], TM.desc.data(), TM.desc.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(h->d_team_desc[k]); h->d_team_desc[k] = nullptr; continue; }
    for (int li = 0; li < 3; ++li) h->team_steps[k][li] = TM.n_steps[li];
    h->team_desc_len[k] = (int)TM.desc.size();
  }
  return true;
}

struct F2DcOpts { double abstol; int maxiters, use_pcnr, mode, initjct; int* dcstate; };
struct F2StepOpts { int refresh; double *resid, *norm; int reps, skip; };   // cadnip_newton_step_fused: one Newton iteration in the team kernel (STEP mode)

// Build (or rebuild) the structure tables; CADNIP_BADARG if the circuit cannot be expressed in them (16-bit offsets)
static int fused2_tables(CadnipHandle* h) {
  if (!h->analyzed) return CADNIP_NOTREADY;
  if (h->d_f2tab && !h->fused2_dirty) return CADNIP_OK;
  // host copies of the gather lists are needed to invert them: read back once
  std::vector<int> g_ptr(h->nnz + 1), c_ptr(h->nnz + 1), b_ptr(h->n + 1);
  HIP_TRY(hipMemcpy(g_ptr.data(), h->d_g_ptr, g_ptr.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c_ptr.data(), h->d_c_ptr, c_ptr.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b_ptr.data(), h->d_b_ptr, b_ptr.size() * 4, hipMemcpyDeviceToHost));
  std::vector<int> g_slots(g_ptr.back()), c_slots(c_ptr.back()), b_slots(b_ptr.back());
  if (!g_slots.empty()) HIP_TRY(hipMemcpy(g_slots.data(), h->d_g_slots, g_slots.size() * 4, hipMemcpyDeviceToHost));
  if (!c_slots.empty()) HIP_TRY(hipMemcpy(c_slots.data(), h->d_c_slots, c_slots.size() * 4, hipMemcpyDeviceToHost));
  if (!b_slots.empty()) HIP_TRY(hipMemcpy(b_slots.data(), h->d_b_slots, b_slots.size() * 4, hipMemcpyDeviceToHost));
  F2Tables T;
  if (!f2_prepare(h, T, g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots)) return CADNIP_BADARG;
  if (h->d_f2tab) (void)hipFree(h->d_f2tab);
  h->d_f2tab = nullptr;
  HIP_TRY(hipMalloc((void**)&h->d_f2tab, T.data.size() * sizeof(unsigned)));
  HIP_TRY(hipMemcpy(h->d_f2tab, T.data.data(), T.data.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  for (int i = 0; i < S_NSEC; ++i) h->f2off[i] = T.off[i];
  h->f2len = h->f2_lds_len;                        // words the kernels stage in LDS (the team's step lists behind them stay in global memory)
  h->fused2_dirty = false;
  h->f2_blk_dirty = true;
  return CADNIP_OK;
}

bool fused2_tables_ready(CadnipHandle* h) { return fused2_tables(h) == CADNIP_OK; }

// Does one instance of this circuit (tables + work array) fit into a CU's LDS?  The drivers fall back to the per-op
// kernels (still on the GPU) when it does not, or when the tables cannot address it.
bool fused2_fits(CadnipHandle* h) {
  if (fused2_tables(h) != CADNIP_OK) return false;
  int nb = 0;
  for (auto& b : h->blocks) nb += b.count > 0;
  if (nb > F2_MAX_BLOCKS) return false;
  const size_t per = (size_t)h->f2_lu_words + 3 * (size_t)h->n + F2_TRASH + 2;
  return ((size_t)h->f2len / 2 + per) * 8 <= 160 * 1024;
}

// device-block descriptors of the fused kernels: rebuilt when the tables were, or when cadnip_set_params changed a block (sp_mos1 pairing);
// also decides the kernel variant (f2_lean, f2_direct)
static int fused2_blocks(CadnipHandle* h) {
  { int rc = fused2_tables(h); if (rc) return rc; }
  if (h->f2_blk_dirty || !h->d_f2blk) {
    F2Block hb[F2_MAX_BLOCKS];
    int nb = 0;
    for (size_t bi = 0; bi < h->blocks.size() && nb < F2_MAX_BLOCKS; ++bi) {
      auto& b = h->blocks[bi];
      if (b.count == 0) continue;
      hb[nb++] = F2Block{b.d_ipar, b.d_par, b.type, b.count, b.n_par, b.g_base, b.c_base, b.b_base, h->f2_nodes_off[bi], b.mos1_plain ? 1 : 0, -1};
    }
    // the heaviest device type first
    for (int i = 0; i < nb; ++i)
      for (int j = i + 1; j < nb; ++j)
        if ((hb[j].type == CADNIP_DEV_MOS1) > (hb[i].type == CADNIP_DEV_MOS1)) { F2Block tmp = hb[i]; hb[i] = hb[j]; hb[j] = tmp; }
    h->f2_rc_blk = -1;
    for (int i = 0; i < nb; ++i)
      if (hb[i].type == CADNIP_DEV_CAPACITOR || hb[i].type == CADNIP_DEV_RESISTOR) { h->f2_rc_blk = i; break; }
    h->f2_src_blk = -1;
    for (int i = 0; i < nb; ++i)
      if (hb[i].type == CADNIP_DEV_VSOURCE || hb[i].type == CADNIP_DEV_ISOURCE) { h->f2_src_blk = i; break; }
    h->f2_lean = true;
    for (int i = 0; i < nb; ++i) {
      const int ty = hb[i].type;
      const bool heavy = ty == CADNIP_DEV_DIODE || ty == CADNIP_DEV_DIODECAP || ty == CADNIP_DEV_SIMPLEMOS || ty == CADNIP_DEV_BVSOURCE ||
                         ty == CADNIP_DEV_BISOURCE || ty == CADNIP_DEV_VA || (ty == CADNIP_DEV_MOS1 && !hb[i].mos1_plain);
      if (heavy) h->f2_lean = false;
    }
    // ... and its linear solve runs from step descriptors staged in LDS: a circuit whose steps do not fit beside the tables and eight
    // instances (long dependency chains: an RC ladder has one step per section) takes the full variant, whose pass program is compact
    if (!h->d_steps1 || ((size_t)(h->f2_lean_end - h->f2_lean_lo) / 2 + (size_t)h->steps1_len + 8 * ((size_t)h->f2_lu_words + 3 * (size_t)h->n + F2_TRASH + 2)) * 8 > 160 * 1024)
      h->f2_lean = false;
    h->f2_n_blk = nb;
    // team kernel (fused_team_kernel.hpp): the parameter rows of the lane-paired sp_mos1 blocks are staged in LDS
    // ... of what the register-resident first pass over the first block does not cover (more than 32 MOSFETs, several blocks)
    h->f2_par_words = 0;
    { bool first = true;
      for (int i = 0; i < nb; ++i)
        if (hb[i].type == CADNIP_DEV_MOS1 && hb[i].mos1_plain) {
          if (!first || hb[i].count > 32) { hb[i].lds_par = h->f2_par_words; h->f2_par_words += hb[i].n_par * hb[i].count; }
          first = false;
        } }
    // every device type emits its residual directly (devices.hpp, Rn); CADNIP_F2_NODIRECT=1 selects the assembled form
    // r = J u + C beta - b instead (diagnostic: the two must agree)
    h->f2_direct = !getenv("CADNIP_F2_NODIRECT");
    if (!h->d_f2blk) HIP_TRY(hipMalloc((void**)&h->d_f2blk, sizeof(hb)));
    HIP_TRY(hipStreamSynchronize(h->stream));               // no launch in flight may still read the old descriptors
    HIP_TRY(hipMemcpy(h->d_f2blk, hb, sizeof(F2Block) * (size_t)nb, hipMemcpyHostToDevice));
    h->f2_blk_dirty = false;
  }
  return CADNIP_OK;
}

// Newton mode 1 (Jacobian reuse) exists in the lean direct-residual variant of the fused kernels only; a circuit with other device types
// (diodes, behavioural sources, sp_mos1 with series resistances, built-in Verilog-A modules ...) takes the per-op kernels, which support
// the mode (driver.hip), instead of failing
bool fused2_mode1_ok(CadnipHandle* h) { return fused2_blocks(h) == CADNIP_OK && h->f2_direct && h->f2_lean; }

static int launch_fused2(CadnipHandle* h, const TranArgs& t, int rounds, const F2DcOpts* dc, const F2StepOpts* step = nullptr) {
  if (!h->analyzed) return CADNIP_NOTREADY;
  if (h->homotopy || h->spec.gshunt != 0.0 || h->spec.srcFact < 1.0) return CADNIP_BADARG;   // homotopies run on the per-op path
  { int rc = fused2_blocks(h); if (rc) return rc; }
  ProfScope ps(h, dc ? "fused2_dc" : step ? "fused_step" : "fused2_newton");
  const LUProgram& P = h->lu;
  F2Args f;
  f.n_blk = h->f2_n_blk; f.rc_blk = h->f2_rc_blk; f.src_blk = h->f2_src_blk;
  f.blk = (const F2Block*)h->d_f2blk;
  f.wave = h->d_wave;
  f.tab = h->d_f2tab;
  for (int i = 0; i < S_NSEC; ++i) f.off[i] = h->f2off[i];
  f.tab_len = h->f2len; f.tab_lo = 0;
  f.n = h->n; f.nnz = h->nnz; f.nnz_lu = h->f2_lu_words;
  f.n_pre = h->f2_n_pre; f.n_post = h->f2_n_post; f.nc = h->f2_nc; f.dn0 = h->f2_dn0; f.n_fwd = h->f2_n_fwd;
  f.lufac = nullptr; f.team_desc = nullptr; f.team_desc_len = 0; f.par_words = 0; f.ts_pre = f.ts_post = f.ts_fwd = 0;
  f.step_refresh = step ? step->refresh : 0; f.step_resid = step ? step->resid : nullptr; f.step_norm = step ? step->norm : nullptr;
  f.step_reps = step ? step->reps : 1; f.step_skip = step ? step->skip : 0;
  if (!dc && (t.newton_mode || step)) {
    // IDA-style Jacobian reuse exists in the lean direct-residual variant (fused2_kernel.hpp); the kept factors of instances that
    // are not resident live in HBM
    if (!(h->f2_direct && h->f2_lean)) return CADNIP_BADARG;
    const size_t need = (size_t)h->B * h->f2_lu_words;
    if (need > h->f2_lufac_cap) {
      if (h->d_f2_lufac) (void)hipFree(h->d_f2_lufac);
      h->d_f2_lufac = nullptr; h->f2_lufac_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->d_f2_lufac, need * sizeof(double)));
      h->f2_lufac_cap = need;
    }
    f.lufac = h->d_f2_lufac;
  }
  f.rounds = rounds; f.B = h->B; f.t = t; f.cold = h->d_cold;
  f.dc_abstol = 0; f.dc_maxiters = 0; f.dc_pcnr = 0; f.dc_mode = 1; f.dc_initjct = 0; f.dcstate = nullptr;
  if (dc) { f.dc_abstol = dc->abstol; f.dc_maxiters = dc->maxiters; f.dc_pcnr = dc->use_pcnr; f.dc_mode = dc->mode; f.dc_initjct = dc->initjct; f.dcstate = dc->dcstate; }
  const bool lean = h->f2_direct && h->f2_lean;
  if (lean) { f.tab_lo = h->f2_lean_lo; f.tab_len = h->f2_lean_end - h->f2_lean_lo; }   // the lean kernels stage [permutations | stamp and node tables] only
  const size_t tab_dbl = (size_t)f.tab_len / 2;
  const size_t per = (size_t)h->f2_lu_words + 3 * (size_t)h->n + F2_TRASH + 2;           // (+ the steps' constant words 0.0, 1.0 behind the trash words)
  const size_t lds_cap = 160 * 1024;
  if (h->n_cu <= 0) {
    int cu = 0;
    HIP_TRY(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, h->device));
    h->n_cu = cu > 0 ? cu : 256;
  }
  if (!h->d_f2queue) HIP_TRY(hipMalloc((void**)&h->d_f2queue, sizeof(int)));
  // Few instances: a team of waves per instance (fused_team_kernel.hpp) -- the latency of ONE transient is what counts when the batch
  // cannot fill the chip.  Transient, direct residuals, lean device set.  CADNIP_F2_TEAM = 0 | 2 | 4 forces the choice (diagnostic, tests).
  if (!dc && h->f2_direct && h->f2_lean) {
    // at most one instance per CU: a team of four waves (one per SIMD); at most two: teams of two waves, two workgroups per CU (their LDS allows it);
    // beyond that the sweep kernel's one wave per instance
    int nw = h->B <= h->n_cu ? 4 : h->B <= 2 * h->n_cu ? 2 : 0;
    if (const char* e = getenv("CADNIP_F2_TEAM")) nw = atoi(e) >= 4 ? 4 : atoi(e) >= 2 ? 2 : 0;
    if (step) nw = 4;
    const size_t shmem_t = (tab_dbl + per + 4 * (size_t)nw + (((size_t)h->f2_par_words + 1) & ~(size_t)1) + (nw ? (size_t)h->team_desc_len[nw / 4] : 0) +
                            (nw ? (size_t)(nw - 1) * ((size_t)h->f2_lu_words + h->n + F2_TRASH) : 0)) * 8;     // (+ the two constant words behind the trash words)
    if (nw && h->d_team_desc[nw / 4] && shmem_t <= lds_cap) {
      f.team_desc = h->d_team_desc[nw / 4]; f.team_desc_len = h->team_desc_len[nw / 4]; f.par_words = h->f2_par_words;
      f.ts_pre = h->team_steps[nw / 4][0]; f.ts_post = h->team_steps[nw / 4][1]; f.ts_fwd = h->team_steps[nw / 4][2];
      TRY_RC(dev_zero_async(h, h->d_f2queue, sizeof(int)));
      f.queue = h->d_f2queue;
      const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(lds_cap / shmem_t, (size_t)(16 / nw)));
      const int grid = std::min(h->B, h->n_cu * wg_per_cu);
      if (getenv("CADNIP_F2_DEBUG")) fprintf(stderr, "[cadnip f2] team of %d waves: B %d n_cu %d grid %d shmem %zu rounds %d nc %d steps %d+%d / %d\n", nw, h->B, h->n_cu, grid, shmem_t, rounds, h->f2_nc, f.ts_pre, f.ts_post, f.ts_fwd);
      if (step) TRY_RC(fteam_launch_step(std::min(h->B, h->n_cu), shmem_t, h->stream, f));
      else TRY_RC(fteam_launch(nw, grid, shmem_t, h->stream, f));
      HIP_TRY(hipGetLastError());
      return CADNIP_OK;
    }
  }
  if (step) return CADNIP_BADARG;                       // (no team kernel for this circuit: the caller takes the per-op kernels)
  // waves (= instances) per workgroup: 8 (two waves per SIMD) when they fit into LDS; fewer when the whole batch is then
  // still resident in one generation with a workgroup on every CU -- a wave runs about 20 % faster with half as many
  // neighbours on its CU (1024 instances: 4 per workgroup on 256 CUs, 57.7 M iterations/s, against 48.1 M as 8 x 128)
  size_t desc_dbl = 0;
  if (lean) {       // the lean variant's linear solve runs from step descriptors (f2_build_steps) staged behind the tables
    if (!h->d_steps1) return CADNIP_BADARG;
    f.team_desc = h->d_steps1; f.team_desc_len = h->steps1_len; f.ts_pre = h->steps1[0]; f.ts_post = h->steps1[1]; f.ts_fwd = h->steps1[2];
    desc_dbl = (size_t)h->steps1_len;
  }
  int wpb = 8;
  if (const char* e = getenv("CADNIP_F2_WPB")) wpb = atoi(e) >= 8 ? 8 : atoi(e) >= 4 ? 4 : atoi(e) >= 2 ? 2 : 1;   // diagnostic: cap the waves per workgroup
  while (wpb > 1 && ((tab_dbl + desc_dbl + wpb * per) * 8 > lds_cap || h->n_cu * (wpb / 2) >= h->B)) wpb >>= 1;
  size_t shmem = (tab_dbl + desc_dbl + wpb * per) * 8;
  if (shmem > lds_cap) return CADNIP_BADARG;
  // resident workgroups only: the instances beyond them are handed out by the in-kernel queue as waves become free
  TRY_RC(dev_zero_async(h, h->d_f2queue, sizeof(int)));
  f.queue = h->d_f2queue;
  const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(lds_cap / shmem, (size_t)(32 / wpb)));
  int grid = std::min((h->B + wpb - 1) / wpb, h->n_cu * wg_per_cu);
  if (getenv("CADNIP_F2_DEBUG")) fprintf(stderr, "[cadnip f2] B %d n_cu %d wpb %d grid %d shmem %zu (tables %zu, steps %zu, per instance %zu) rounds %d nc %d passes %d+%d steps %d+%d / %d variant %d\n", h->B, h->n_cu, wpb, grid, shmem, tab_dbl * 8, desc_dbl * 8, per * 8, rounds, h->f2_nc, h->f2_n_pre, h->f2_n_post, f.ts_pre, f.ts_post, f.ts_fwd, !h->f2_direct ? 2 : h->f2_lean ? 0 : 1);
  const int var = !h->f2_direct ? 2 : h->f2_lean ? 0 : 1;
  { int rc = var == 0 ? f2_launch_variant<0>(wpb, dc != nullptr, grid, shmem, h->stream, f)
           : var == 1 ? f2_launch_variant<1>(wpb, dc != nullptr, grid, shmem, h->stream, f)
                      : f2_launch_variant<2>(wpb, dc != nullptr, grid, shmem, h->stream, f);
    if (rc) return rc; }
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

int launch_fused2_rounds(CadnipHandle* h, const TranArgs& t, int rounds) { return launch_fused2(h, t, rounds, nullptr); }

// one Newton iteration of every instance in the team kernel (STEP mode): residual -> io.resid / io.norm (optional), Newton step -> io.delta,
// io.flags[inst] = 1 on a failed solve (the caller clears them); CADNIP_BADARG when the circuit has no team kernel (not lean, too large for LDS)
int launch_fused_step(CadnipHandle* h, int refresh, const FusedStepIO& io) {
  if (h->va_ext || !fused2_fits(h)) return CADNIP_BADARG;
  TranArgs t;
  memset(&t, 0, sizeof(t));
  t.u = (double*)io.u; t.du = (double*)io.du; t.delta = io.delta; t.limit_w = h->d_limit_w; t.tcur = (double*)io.t; t.gamma = (double*)io.gamma; t.active = h->d_active; t.flags = io.flags;
  t.t = io.t_keep; t.h = io.gamma_keep;                   // (STEP mode: where the caller's times / leading coefficients are also kept)
  t.B = h->B; t.n = h->n; t.n_limits = h->n_limits;
  F2StepOpts so{refresh, io.resid, io.norm, io.reps > 0 ? io.reps : 1, io.skip};
  return launch_fused2(h, t, 1 << 30, nullptr, &so);
}

int launch_fused2_dc(CadnipHandle* h, const TranArgs& t, int rounds, double abstol, int maxiters, int use_pcnr, int mode, int initjct, int* d_dcstate) {
  F2DcOpts dc{abstol, maxiters, use_pcnr, mode, initjct, d_dcstate};
  return launch_fused2(h, t, rounds, &dc);
}

}  // namespace cadnip

