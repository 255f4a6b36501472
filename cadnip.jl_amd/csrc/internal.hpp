// internal.hpp -- handle layout shared by the kernel, symbolic and driver translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/cadnip_hip.h"
// rows of the derived sp_mos1 parameter card (devices.hpp: enum M1_*) that cadnip_set_params inspects
#define CADNIP_MOS1_PAR_OXCAP 8
#define CADNIP_MOS1_PAR_GD 30
#define CADNIP_MOS1_PAR_GS 31

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t _e = (expr);                                               \
    if (_e != hipSuccess) {                                               \
      cadnip::set_last_error(#expr, _e);                                  \
      return CADNIP_HIPERROR;                                             \
    }                                                                     \
  } while (0)

namespace cadnip {

void set_last_error(const char* what, hipError_t e);

// host-side result of the symbolic phase (symbolic.cpp)
struct LUProgram {
  int n = 0;
  int n_blocks = 0;                    // CADNIP_LU_ORDER=klu: diagonal blocks of the block triangular form (0: the Markowitz search was used)
  std::vector<int> rperm, cperm;       // pivot k uses original row rperm[k], column cperm[k]
  std::vector<char> unit;              // pivot k is the stamped constant 1 (a charge / limit row's own diagonal): no division by it
  int nnz_lu = 0;
  // LU stored row-major in permuted indices; row i occupies [lu_rowptr[i], lu_rowptr[i+1]) with sorted columns
  std::vector<int> lu_rowptr, lu_col, lu_diag;   // lu_diag[i] = position of U(i,i)
  std::vector<int> load_src, load_dst;           // J csr position -> LU position
  // entry-wise left-looking factorisation program, entries sorted by dependency level
  std::vector<int> ent_pos, ent_diag, ent_ptr;   // ent_diag = -1 for U entries, else position of the pivot
  std::vector<int> term_a, term_b;               // LU positions: acc -= lu[a]*lu[b]
  std::vector<int> lev_ptr;                      // factor levels -> ranges of entries
  // triangular solves, row-wise gather, rows sorted by level
  std::vector<int> fwd_rows, fwd_lev_ptr, bwd_rows, bwd_lev_ptr;
};

// device-local unknowns for the leaf-first pivot order (symbolic.cpp): charges [q_begin, lim_begin), limits [lim_begin, n);
// unit_ok[i]: the diagonal of unknown i is one constant stamp (no C part, a single G slot)
struct LULeaves { int q_begin = -1, lim_begin = -1; const unsigned char* unit_ok = nullptr; bool first = false; };   // first: pivot the leaves before everything else (diagnostic, CADNIP_LU_LEAF_FIRST=1)

static inline unsigned long long pack4(unsigned a, unsigned b, unsigned c, unsigned d) {
  return (unsigned long long)(a & 0xFFFFu) | ((unsigned long long)(b & 0xFFFFu) << 16) | ((unsigned long long)(c & 0xFFFFu) << 32) | ((unsigned long long)(d & 0xFFFFu) << 48);
}

// ---- linear-solve program of the fused kernel (f2_program.cpp) ----
struct F2Program {
  int nc = 0, lu_words = 0, dn0 = 0;       // core size, words before the rhs (sparse + dense), first dense word
  std::vector<int> posW;                   // LU pattern position -> W offset
  std::vector<unsigned long long> lanes, passes;          // lane descriptors, pass descriptors (pre-dense passes, then post-dense)
  std::vector<unsigned> terms;
  int n_pre = 0, n_post = 0, n_fwd = 0;    // (n_fwd: forward substitution alone, on kept factors: behind the other two lists)
  double cost = 0;                         // issue-slot estimate used to choose nc
};

struct F2Ent { int pos, dg, lvl; std::vector<int> a, b; std::vector<int> tl; int dl = -1; };   // tl / dl: level at which term t's factors / the pivot are final

bool f2_build_program(const LUProgram& P, int n, int nc, F2Program& G);

// the same program laid out for a team of nw waves per instance (f2_program.cpp: f2_build_team; fused_team_kernel.hpp)
struct F2Team {
  int nw = 0, nc = 0, lu_words = 0, dn0 = 0;
  int n_steps[3] = {0, 0, 0};              // steps of the pre-core, post-core and forward-only (kept factors) lists
  std::vector<int> posW;                   // as F2Program::posW (same layout for the same nc)
  std::vector<unsigned long long> desc;    // [step][nw * 64 lanes] packed word offsets into W (f2_program.cpp: f2_build_team)
};
bool f2_build_team(const LUProgram& P, int n, int nc, int nw, F2Team& T);
// ... for one wave per instance: list-scheduled steps of 64 lanes with up to three terms per lane, 16-byte descriptors (two words per lane in `desc`)
bool f2_build_steps(const LUProgram& P, int n, int nc, int nw, F2Team& T);

struct DeviceBlock {
  int type, count, n_nodes, n_ipar, n_par;
  int g_base, c_base, b_base, n_g, n_c, n_b;
  int* d_nodes = nullptr;
  std::vector<int> h_nodes;  // host copy (the fused kernel keeps an int16 copy in LDS)
  double* d_cache = nullptr; int n_cache = 0;   // generated external model: [B][n_cache][count] results of its setup pass (bias-independent statements)
  int va_model = -1;         // CADNIP_DEV_VA: the block's model id (ipar row 0)
  int va_tl = 0;             // generated external model: lanes per device when evaluated with one derivative direction per lane (16 / 32; stamp_csr.hip)
  bool mos1_plain = false;   // sp_mos1 block: every instance has gd = gs = OxideCap = 0 (set by cadnip_set_params)
  int* d_ipar = nullptr;
  double* d_par = nullptr;   // [B][n_par][count]
  // reduction plan of the stamping kernel (stamp_csr.hip): devices per tile, tiles per instance, and per tile the targets
  // (CSR entries of G / C, rows of b) it contributes to with the LDS offsets of their contributions in COO order
  int sp_cs = 0, sp_chunks = 0, sp_n_targets = 0, sp_levels = 1, sp_scratch = 0;
  struct Target { int chunk; unsigned word; std::vector<unsigned short> offs; };
  std::vector<Target> sp_targets;            // build-time only
  int *d_sp_tptr = nullptr, *d_sp_info = nullptr; uint4* d_sp_rec = nullptr;
  // sp_mos1 blocks carry two plans -- the lane-pair path (mos1_plain: the rows of the external d / g / s terminals and the d / s columns
  // are structural zeros there) stages fewer rows -- and the launch picks the one that matches the parameters in force
  struct PlanSet { int n_targets = 0, levels = 1, scratch = 0, rows = 0; int *tptr = nullptr, *info = nullptr; uint4* rec = nullptr; unsigned short* rowoff = nullptr; };
  PlanSet sp_gen, sp_plain;
  int sp_rows = 0;                           // staged rows of a tile: the slots some target reads (+ one trash row for the rest); see build_stamp_plan
  unsigned short* d_sp_rowoff = nullptr;     // [n_g + n_c + n_b] word offset of every slot's row inside a tile
};

struct ProfEntry { const char* name; double ms = 0; int64_t calls = 0; };

}  // namespace cadnip

struct CadnipHandle {
  int device = 0;
  hipStream_t stream = nullptr;
  int B = 0;
  // structure (host copies kept for the symbolic phase)
  int n = 0, n_nodes = 0, n_currents = 0, n_charges = 0, n_limits = 0, nnz = 0;
  std::vector<int> h_rowptr, h_colidx, h_to_ref;
  int ns_g = 0, ns_c = 0, ns_b = 0, ns = 0;
  std::vector<cadnip::DeviceBlock> blocks;
  std::vector<double> h_limit_init;
  CadnipSpec spec{1, 1e-12, 0.0, 1.0};
  int initjct = 0;
  // device: structure
  int *d_rowptr = nullptr, *d_colidx = nullptr, *d_to_ref = nullptr;
  int *d_g_ptr = nullptr, *d_g_slots = nullptr, *d_c_ptr = nullptr, *d_c_slots = nullptr, *d_b_ptr = nullptr, *d_b_slots = nullptr;
  unsigned char* d_diag_flag = nullptr;   // [nnz] 1 where the entry is G[i,i] of a voltage node
  int* d_long_rows = nullptr;   // rows with more than LONG_LIST entries (kernels.hip: k_residual_long)
  int n_long_rows = 0;
  double* d_dump = nullptr;     // non-null only inside cadnip_get_contributions: [B][ns] staged contributions (stamp_csr.hip)
  unsigned* d_prep = nullptr;   // words the stamping kernels do not store themselves (stamp_csr.hip: k_stamp_prep)
  int n_prep = 0, n_prep_atomic = 0;   // the first n_prep_atomic words are rewritten before every restamp
  bool prep_stale = true;              // unstamped node diagonals may still hold a gshunt of an earlier restamp
  double* d_wave = nullptr;
  double* d_limit_init = nullptr;
  // per-instance homotopy parameters of the DC fallback chain (solve.jl:720-850); equal to spec.gshunt / spec.srcFact
  // except while cadnip_dc_run walks an instance through gshunt / source stepping
  double *d_gshunt = nullptr, *d_srcfact = nullptr;
  bool homotopy = false;     // some instance has gshunt != 0 or srcFact < 1: the fused kernel (no homotopy terms) must not run
  // device: per-instance state [B][..]
  double *d_u = nullptr, *d_du = nullptr, *d_t = nullptr, *d_gamma = nullptr;
  double *d_G = nullptr, *d_C = nullptr, *d_b = nullptr, *d_J = nullptr, *d_resid = nullptr, *d_delta = nullptr;
  double *d_limit_w = nullptr, *d_LU = nullptr, *d_tmp = nullptr;
  int* d_flags = nullptr;        // [B] per-instance status bits (1 = singular pivot, 2 = non-finite)
  int* d_active = nullptr;       // [B] 1 = instance takes part in the next launches
  int* d_nonfinite = nullptr;    // [B] raised by the assemble kernels when a stamped value is NaN / Inf
  int* d_cold = nullptr;         // [B] 1 = the instance's DC solve is a cold start: initjct (armed per launch) applies to it (solve.jl:615-625: iszero(u0))
  // LU
  bool analyzed = false;
  cadnip::LUProgram lu;
  int *d_load_src = nullptr, *d_load_dst = nullptr, *d_ent_pos = nullptr, *d_ent_diag = nullptr, *d_ent_ptr = nullptr;
  int *d_term_a = nullptr, *d_term_b = nullptr, *d_lev_ptr = nullptr;
  int *d_lu_rowptr = nullptr, *d_lu_col = nullptr, *d_lu_diag = nullptr, *d_rperm = nullptr, *d_cperm = nullptr;
  int *d_fwd_rows = nullptr, *d_fwd_lev_ptr = nullptr, *d_bwd_rows = nullptr, *d_bwd_lev_ptr = nullptr;
  // fused kernel: packed structure tables (fused2.hip), rebuilt when the LU program changes
  unsigned int* d_f2tab = nullptr;
  int f2off[16] = {0};      // section offsets in 32-bit words
  int f2len = 0;            // 32-bit words
  int f2_lu_words = 0, f2_nc = 0, f2_dn0 = 0, f2_n_pre = 0, f2_n_post = 0, f2_n_fwd = 0;
  double* d_f2_lufac = nullptr; size_t f2_lufac_cap = 0;   // Newton mode 1: kept factors of the non-resident instances [B][f2_lu_words]   // linear-solve program of the fused kernel
  std::vector<int> f2_nodes_off;   // per device block: offset (int16 units) of its node table inside the NODES section
  bool fused2_dirty = true;
  int* d_f2queue = nullptr;   // fused kernel: dynamic instance queue
  void* d_f2blk = nullptr;    // fused kernel: device-block descriptors
  bool f2_blk_dirty = true;
  int f2_n_blk = 0, f2_rc_blk = -1;
  unsigned long long* d_team_desc[2] = {nullptr, nullptr};   // team kernel: step descriptors for teams of 2 / 4 waves (f2_build_team), built with the tables
  int team_steps[2][3] = {{0, 0, 0}, {0, 0, 0}};
  int team_desc_len[2] = {0, 0};   // 64-bit words each
  unsigned long long* d_steps1 = nullptr;   // one wave per instance, lean variant: step descriptors (f2_build_steps), two words per lane
  int steps1[3] = {0, 0, 0}, steps1_len = 0;
  unsigned long long* d_steps4 = nullptr;   // ... for a team of four waves (the per-op step LU, lu_f2.hip: k_lu_steps)
  int steps4[3] = {0, 0, 0};
  int f2_lean_lo = 0, f2_lean_end = 0;      // the table words [lo, end) the lean kernels stage in LDS (permutations, stamp tables, node tables)
  int f2_lds_len = 0;         // 32-bit words of the table that the fused kernels copy to LDS (f2len; the team kernel's step lists lie behind)
  int f2_par_words = 0;       // team kernel: doubles of the LDS-staged sp_mos1 parameter rows of one instance
  bool f2_direct = false;     // devices emit their residuals directly: no J*u pass (off: CADNIP_F2_NODIRECT=1)
  std::vector<unsigned char> leaf_unit_ok;   // per unknown: its diagonal is one constant G stamp, no C stamp (leaf-first pivot order, symbolic.cpp)
  cadnip::LULeaves leaves;    // charge / limit ranges of the unknown layout [V | I | q | lim]
  int f2_lu_len = 0;          // 32-bit words of the table's linear-solve prefix (entry program, permutations, load map)
  int n_cu_hint() const { return n_cu > 0 ? n_cu : 256; }   // compute units of the device (queried by the first fused launch; MI355X: 256)
  bool va_ext = false;        // the circuit uses an external generated model (va_generated_ext.hpp): not compiled into the fused kernel
  bool f2_lean = false;       // only device types of the lean kernel variant (fused2.hip: dispatch_stamp2)
  int f2_src_blk = -1;        // first independent-source block of the fused block list
  int n_cu = 0;
  // driver state (allocated lazily)
  struct Driver* drv = nullptr;
  // profiling
  bool prof_on = false;
  std::vector<cadnip::ProfEntry> prof;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // pinned scratch
  int* h_pinned = nullptr;
  int* d_pinned = nullptr;
  // mapped pinned staging area of the host-pointer entry points (api.hip: stage_up / stage_down / stage_finish): small transfers are moved
  // by a copy KERNEL on the stream that reads / writes the staging area across the bus; large ones are blocking copies
  char* h_stage = nullptr; char* d_stage = nullptr; size_t stage_bytes = 0, stage_off = 0;   // (d_stage: the same memory as the device sees it)
  struct PendingDown { void* dst; const void* src; size_t bytes; };
  // cadnip_newton_step: the launch sequence of one call as an instantiated HIP graph per variant (refresh or not); `graph_epoch` moves whenever
  // something a captured kernel argument depends on changes (spec, parameters' structure, LU program): a stale graph is captured again
  struct StepGraph { hipGraphExec_t exec = nullptr; unsigned long long epoch = 0, warmed = 0; };
  StepGraph step_graph[4];      // per-op: no refresh / refresh; fused: no refresh / refresh
  unsigned long long graph_epoch = 1;
  std::vector<PendingDown> stage_pending;   // the same words as the device sees them: small results are PUBLISHED there by a kernel (driver.hip: k_publish_int), not copied
};

namespace cadnip {
// symbolic.cpp
// KLU's ordering (klu_order.cpp): column sequence, the maximum transversal's row per column, block boundaries; false = structurally singular
bool klu_style_order(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, std::vector<int>& colorder, std::vector<int>& match_row, std::vector<int>& block_ptr);
int lu_analyze(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& vals,
               double pivot_tol, bool sample, LUProgram& out, std::string& err, const LULeaves* leaves = nullptr);
// kernels.hip launchers (all asynchronous on h->stream)
int launch_rebuild(CadnipHandle* h);                       // stamp_csr.hip: one stamp + reduce kernel per device type at (d_u, d_t)
int build_stamp_plan(CadnipHandle* h, const CadnipStructure* s);   // stamp_csr.hip, once per structure
int launch_stamp_block(CadnipHandle* h, int block);                // stamp_csr.hip: the stamping kernel of one device block
int launch_residual(CadnipHandle* h, const double* d_du);  // d_resid = C du + G u - b
int launch_jacobian(CadnipHandle* h);                      // d_J = G + gamma C
int launch_factor(CadnipHandle* h, bool fuse_jacobian);    // LU of J (or of G + gamma C)
int launch_solve(CadnipHandle* h, const double* d_rhs, double* d_x);
int launch_factor_solve(CadnipHandle* h, bool fuse_jacobian, const double* d_rhs, double* d_x);
int upload_lu(CadnipHandle* h);
int upload_homotopy(CadnipHandle* h, const double* gshunt /* [B] or null = spec */, const double* srcfact /* [B] or null = spec */);
int launch_calib_copy(CadnipHandle* h, long n, int reps);
#define TRY_RC(x) do { int _rc_ = (x); if (_rc_) return _rc_; } while (0)
struct MultiCopy { struct Seg { unsigned* dst; const unsigned* src; size_t words; }; Seg seg[8]; int n = 0;
  void add(void* dst, const void* src, size_t bytes) { seg[n].dst = (unsigned*)dst; seg[n].src = (const unsigned*)src; seg[n].words = bytes / 4; ++n; } };
int dev_multi_async(CadnipHandle* h, const MultiCopy& m, bool to_host);   // kernels.hip: up to 8 word copies / clears (src = null) in one launch
int launch_norm2(CadnipHandle* h, const double* d_x, double* d_out);      // kernels.hip: per-instance 2-norm
int dev_zero_async(CadnipHandle* h, void* p, size_t bytes);
int dev_copy_async(CadnipHandle* h, void* dst, const void* src, size_t bytes, bool to_host);   // kernels.hip: word copy as a kernel on the handle's stream       // kernels.hip: zero-fill as a kernel on the handle's stream (ordered with the other kernels)
int launch_negate(CadnipHandle* h, double* d_x, long n);
struct TranArgs;                                                          // tran_ctrl.hpp
int launch_fused2_rounds(CadnipHandle* h, const TranArgs& t, int rounds); // fused2.hip
// one Newton iteration in the team kernel (fused2.hip): device-visible pointers (device memory or mapped pinned host memory) of its inputs and outputs;
// gamma_keep / t_keep: device arrays that also receive the caller's gamma / t (the handle's state stays what the per-op entry points would leave)
struct FusedStepIO { const double *u, *du, *gamma, *t; double *gamma_keep, *t_keep, *delta, *resid, *norm; int* flags; int reps = 1, skip = 0; };   // reps / skip: measurement (cadnip_debug_step_time)
int launch_fused_step(CadnipHandle* h, int refresh, const FusedStepIO& io);
int launch_va_setup(CadnipHandle* h, DeviceBlock& b);                        // stamp_csr.hip: the setup pass of a generated external model's block
bool fused2_tables_ready(CadnipHandle* h);                                 // the packed tables exist (built on demand)
bool fused2_fits(CadnipHandle* h);                                        // false: circuit too large for the LDS-resident kernel
bool fused2_mode1_ok(CadnipHandle* h);                                    // Newton mode 1 can run in the fused kernel (lean device set, direct residuals)
int launch_fused2_dc(CadnipHandle* h, const TranArgs& t, int rounds, double abstol, int maxiters, int use_pcnr, int mode, int initjct, int* d_dcstate);
struct ProfScope {
  CadnipHandle* h; int idx;
  ProfScope(CadnipHandle* h, const char* name);
  ~ProfScope();
};
}  // namespace cadnip
