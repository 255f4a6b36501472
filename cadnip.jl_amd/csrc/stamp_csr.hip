// stamp_csr.hip -- the per-op stamping kernels: one kernel per device type that evaluates the devices AND reduces their
// contributions into the CSR arrays G, C and b.  ≡ reset_direct_stamp! + the builder pass + the deferred-b / srcFact /
// gshunt steps of fast_rebuild! (/root/reference/src/mna/precompile.jl:493-537, value_only.jl:238-261, 395-478).
//
// One 64-lane wave owns a *tile*: one chunk of (up to 64) devices of one type, for one sweep instance (several instances when
// the type has few devices).  Phases of a tile:
//   1. stamp   -- every lane evaluates its device (coalesced reads of its parameter rows, node voltages from the instance's
//                 u) and stages its per-element contributions in LDS, slot-major ([slot][device]: conflict-free writes).
//   2. reduce  -- segmented reduction in LDS: the tile's *targets* (the CSR entries of G / C and the rows of b that receive
//                 anything from this chunk, in CSR order) are dealt to the lanes; a lane sums its target's staged
//                 contributions in the reference's COO order (nzval[map[pos]] += v, value_only.jl:414-418, addition for
//                 addition) and writes ONE value to HBM.  Consecutive lanes write consecutive CSR positions.  A target
//                 with more than five contributions (node diagonals, supply rails) is reduced as a 5-ary tree over
//                 consecutive runs of its list: partial sums go to LDS scratch words and are combined on the next level,
//                 so no lane ever walks a long list alone (one lane summing the 120 stamps of a rail made the whole wave
//                 wait 40 k cycles).
// No slot buffer in HBM, no separate assemble pass, no zero-fill of G / C / b: a target whose contributions all come from
// one tile is stored; a target that an earlier kernel of the stream has already stored is read-modify-written (kernels of a
// stream run in order, and within this kernel the tile is its only writer); only a target that receives contributions from
// several tiles of the SAME kernel -- a boundary between tiles, e.g. a supply rail fed by every chunk of a large circuit --
// is accumulated with a global fp64 atomic, on a word that k_stamp_prep has pre-set when no earlier kernel stores it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CADNIP_VA_WITH_EXT   // this translation unit carries the external generated models too (va_generated_ext.hpp)
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"   // CADNIP_WAVE_SYNC

namespace cadnip {

#define N_CLS 5   // record classes of the reduction (k_stamp_csr); 5 = padding step
enum { TGT_STORE = 0, TGT_RMW = 1, TGT_ATOMIC = 2, TGT_PARTIAL = 3 };   // bits 30-31 of a target word; bits 28-29: 0 G, 1 C, 2 b; bits 0-27: index
                                                                       // (TGT_PARTIAL: index = LDS scratch word of the tile, relative to the tile)

// LDS staging writer: slot (k, dev) of this lane's instance tile; same interface as SlotOut (devices.hpp)
struct LdsOut {
  static constexpr bool DIRECT = false;
  __device__ __forceinline__ void Rn(int, double) const {}
  __device__ __forceinline__ double du(int) const { return 0.0; }
  double *g, *c, *b;     // tile + 0, + n_g * cs, + (n_g + n_c) * cs
  int cs, ldev;          // devices per tile row, this lane's device within the tile
  bool on;               // false: no device behind this lane (or an inactive instance)
  __device__ __forceinline__ void G(int k, double v) const { if (on) g[k * cs + ldev] = v; }
  __device__ __forceinline__ void C(int k, double v) const { if (on) c[k * cs + ldev] = v; }
  __device__ __forceinline__ void B(int k, double v) const { if (on) b[k * cs + ldev] = v; }
  template <int N> __device__ __forceinline__ void Gv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) G(k0 + i, v[i]);
  }
  template <int N> __device__ __forceinline__ void Gk(const int (&k)[N], const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) G(k[i], v[i]);
  }
  template <int N> __device__ __forceinline__ void Cv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) C(k0 + i, v[i]);
  }
  template <int N> __device__ __forceinline__ void Bv(int k0, const double (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) B(k0 + i, v[i]);
  }
};

struct CsrStampArgs {
  const int* nodes; const int* ipar; const double* par; const double* wave;
  const double* u; const double* t; const int* active; const int* cold;
  double *G, *C, *b, *limit_w; int* nonfinite;
  const unsigned char* diag_flag; const double* gshunt; const double* srcFact;
  const int* step_ptr; const int* step_info; const uint4* tgt_rec;   // [n_chunks + 1] step ranges, per step class | new-level flag << 8, STEP_W records (16 B each) per step (build_stamp_plan)
  int B, count, n, nnz, n_par, n_g, n_c, n_b, cs, n_chunks, ipw, lpd, mode, initjct, zero_first, n_levels, n_scratch;
  int u_lds;   // the unknowns of the tile's instances are staged in LDS (small circuits): node voltages are then LDS reads
  const double* cache; int n_cache;                 // generated external models: setup-pass results [B][n_cache][count] (k_va_setup)
  double* dump; int ns, dump_g, dump_c, dump_b;   // operating-point read-out only (cadnip_get_contributions): the staged per-device
                                                    // contributions written out as [B][ns], slot (k, dev) of array A at A_base + k * count + dev

};

#ifdef CADNIP_TRACE
// diagnostic build: cycles between the phases of the sp_mos1 kernel, summed over all waves and launches
static __device__ unsigned long long g_sc_sum[8], g_sc_cnt;
#define SC_POINT(id) do { if (TYPE == CADNIP_DEV_MOS1) { unsigned long long _t = clock64(); if (threadIdx.x == 0) atomicAdd(&g_sc_sum[id], _t - sc_last); sc_last = _t; } } while (0)
#else
#define SC_POINT(id) do {} while (0)
#endif

#define STEP_W 128   // records per reduction step (two per lane)
#define PIPE 8       // steps whose records are in flight / in registers

template <int TYPE>
__global__ void __launch_bounds__(64) k_stamp_csr(CsrStampArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
#ifdef CADNIP_TRACE
  unsigned long long sc_last = clock64();
  if (TYPE == CADNIP_DEV_MOS1 && threadIdx.x == 0) atomicAdd(&g_sc_cnt, 1ull);
#endif
  const int chunk = blockIdx.x % a.n_chunks, grp = blockIdx.x / a.n_chunks;
  // ---- the reduction's records.  The chunk's records form *steps* of STEP_W = 128 (two per lane), every step homogeneous
  // in class and level (build_stamp_plan pads).  The reduce loop is ROLLED -- this kernel runs once per wave, so every
  // instruction it executes is an instruction-cache miss waiting to happen, and a 32-fold unrolled loop ran at ~800 cycles
  // per step for that reason alone -- and keeps a shift register of PIPE steps' records: the first PIPE are requested
  // here, before the stamp phase, a new one enters with every step.  step_lane: lane s holds the descriptor of step s
  // (class | first-step-of-a-level flag << 8), read with v_readlane.
  const int s0 = a.step_ptr[chunk], n_steps = a.step_ptr[chunk + 1] - s0;
  const int step_lane = a.step_info[s0 + (lane < n_steps ? lane : 0)];   // steps beyond 64 read their descriptor from memory
  const uint4* recs = a.tgt_rec + (size_t)s0 * STEP_W + lane;
  uint4 pipe[PIPE][2];                                                    // stage p holds the records of step q + p
#pragma unroll
  for (int p = 0; p < PIPE; ++p) {
    const int st = p < n_steps ? p : 0;
    pipe[p][0] = recs[(size_t)st * STEP_W]; pipe[p][1] = recs[(size_t)st * STEP_W + 64];
  }
  const int nslots = a.n_g + a.n_c + a.n_b, tile_words = nslots * a.cs + a.n_scratch;   // staged slots | scratch of the reduction tree
  // lane -> (instance of the tile, device of the chunk, side of a lane pair)
  const int dl = lane / a.lpd, side = lane - dl * a.lpd;
  const int ii = a.n_chunks == 1 ? dl / a.count : 0;
  const int ldev = a.n_chunks == 1 ? dl - ii * a.count : dl;
  const int dev = chunk * a.cs + ldev;
  const int inst = grp * a.ipw + ii;
  const bool lane_on = ii < a.ipw && inst < a.B && dev < a.count && ldev < a.cs;
  const bool valid = lane_on && a.active[inst] != 0;
  const int inst_c = inst < a.B ? inst : a.B - 1;                 // clamped: every lane runs the device code (lane-pair DPP)
  double* tile = lds + (size_t)(ii < a.ipw ? ii : 0) * tile_words;
  if (a.zero_first) {
    // (tile_words is even for every type that needs this: n_g + n_c + n_b of sp_mos1 = 114; an odd tail is covered below)
    const int words = a.ipw * tile_words;
    for (int i = lane; i < (words >> 1); i += 64) ((double2*)lds)[i] = make_double2(0.0, 0.0);
    if ((words & 1) && lane == 0) lds[words - 1] = 0.0;
    CADNIP_WAVE_SYNC();
  }
  if (lane < a.ipw) lds[(size_t)(lane + 1) * tile_words - 1] = 0.0;   // the tile's zero word: operand slots a record does not use
  // per-instance scalars of the tile's instances, behind the tiles: the reduction reads them per instance
  double* inst_par = lds + (size_t)a.ipw * tile_words;             // [ipw][3]: active, gshunt, srcFact
  if (lane < a.ipw) {
    const int i2 = grp * a.ipw + lane;
    const bool in = i2 < a.B;
    inst_par[3 * lane] = in && a.active[i2] ? 1.0 : 0.0;
    inst_par[3 * lane + 1] = in ? a.gshunt[i2] : 0.0;
    inst_par[3 * lane + 2] = in ? a.srcFact[i2] : 1.0;
  }
  double* u_tile = inst_par + 3 * a.ipw;                           // [ipw][n] when u_lds
  if (a.u_lds) {
    for (int r = 0; r < a.ipw; ++r) {
      const int i2 = grp * a.ipw + r < a.B ? grp * a.ipw + r : a.B - 1;
      const double* ug = a.u + (size_t)i2 * a.n;
      for (int i = lane; i < a.n; i += 64) u_tile[(size_t)r * a.n + i] = ug[i];
    }
    CADNIP_WAVE_SYNC();
  }
  SC_POINT(0);
  {
    DevCtx d{a.nodes, a.ipar, a.par + (size_t)inst_c * a.n_par * a.count, a.wave, a.count, dev < a.count ? dev : a.count - 1, a.t[inst_c], a.mode, (a.initjct && a.cold[inst_c]) ? 1 : 0,
             a.cache ? a.cache + (size_t)inst_c * a.n_cache * a.count : nullptr};
    LdsOut s{tile, tile + (size_t)a.n_g * a.cs, tile + (size_t)(a.n_g + a.n_c) * a.cs, a.cs, ldev, valid};
    const double* u = a.u_lds ? u_tile + (size_t)(ii < a.ipw ? ii : 0) * a.n : a.u + (size_t)inst_c * a.n;
    double* lw = valid ? a.limit_w + (size_t)inst_c * a.n : nullptr;
    if (TYPE == CADNIP_DEV_RESISTOR) stamp_resistor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CAPACITOR) stamp_capacitor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_INDUCTOR) stamp_inductor(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VSOURCE) stamp_vsource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_ISOURCE) stamp_isource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VCVS) stamp_vcvs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VCCS) stamp_vccs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CCVS) stamp_ccvs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_CCCS) stamp_cccs(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_DIODE) stamp_diode(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_DIODECAP) stamp_diodecap(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_SIMPLEMOS) stamp_simplemos(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_MOS1) {
      if (a.lpd == 2) stamp_mos1_pair(d, u, s, lw, side, valid);   // two lanes per MOSFET (devices.hpp)
      else stamp_mos1(d, u, s, lw);
    }
    else if (TYPE == CADNIP_DEV_BVSOURCE) stamp_bvsource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_BISOURCE) stamp_bisource(d, u, s, lw);
    else if (TYPE == CADNIP_DEV_VA) {
      if (a.lpd >= VA_TL_LANES) stamp_va_tl(d, u, s, lw, side);   // external model: lane `side` of the device's group carries direction `side`
      else stamp_va(d, u, s, lw);
    }
  }
  CADNIP_WAVE_SYNC();
  if (a.dump && valid && side == 0) {
    double* D = a.dump + (size_t)inst * a.ns;
    for (int k = 0; k < a.n_g; ++k) D[a.dump_g + k * a.count + dev] = tile[k * a.cs + ldev];
    for (int k = 0; k < a.n_c; ++k) D[a.dump_c + k * a.count + dev] = tile[(a.n_g + k) * a.cs + ldev];
    for (int k = 0; k < a.n_b; ++k) D[a.dump_b + k * a.count + dev] = tile[(a.n_g + a.n_c + k) * a.cs + ldev];
  }
  SC_POINT(1);
  // ---- segmented reduction: one target per lane and step, contributions summed in COO order out of LDS.  A target is
  // one 16-byte record (destination, count, up to five staging offsets inline); RED_U records per lane are requested
  // before the first is used, so the batch costs one memory latency, not one per target.
  // Step by step: class 0 / 1 / 2 = sole writer of a word of G / C / b (plain store), 3 = partial sum into LDS scratch,
  // 4 = the rest (read-modify-write behind an earlier kernel, atomics between tiles), 5 = padding.  Class, destination and
  // the instance's scalars are wave-uniform, so a record costs five LDS reads issued together (operand slots it does not
  // use point at the tile's zero word), four adds and one store.
  // everything a record needs about its instance, gathered once (not per record: the compiler would re-read kernel
  // arguments and LDS words in every step)
  struct InstCtx { double *Gb, *Cb, *bb; int* nf; double sf, gsh; int tile_off; };   // (the tile is addressed by offset: an LDS pointer inside a struct decays to a generic one)
  auto inst_ctx = [&](int ri) {
    const int rinst = grp * a.ipw + ri;
    InstCtx c;
    c.tile_off = ri * tile_words;
    c.Gb = a.G + (size_t)rinst * a.nnz; c.Cb = a.C + (size_t)rinst * a.nnz; c.bb = a.b + (size_t)rinst * a.n;
    c.nf = a.nonfinite + rinst;
    c.gsh = uniform_f64(inst_par[3 * ri + 1]); c.sf = uniform_f64(inst_par[3 * ri + 2]);
    return c;
  };
  const unsigned char* const diag_flag = a.diag_flag;
  auto reduce_instance = [&](const InstCtx& c, const uint4& ra, const uint4& rb, int cls) {
    double* src = lds + c.tile_off;
    const double a0 = src[ra.y >> 16], a1 = src[ra.z & 0xFFFFu], a2 = src[ra.z >> 16], a3 = src[ra.w & 0xFFFFu], a4 = src[ra.w >> 16];
    const double b0 = src[rb.y >> 16], b1 = src[rb.z & 0xFFFFu], b2 = src[rb.z >> 16], b3 = src[rb.w & 0xFFFFu], b4 = src[rb.w >> 16];
    double acc[2] = {(((a0 + a1) + a2) + a3) + a4, (((b0 + b1) + b2) + b3) + b4};
    const unsigned e[2] = {ra.x & 0x0FFFFFFFu, rb.x & 0x0FFFFFFFu};
    const bool on[2] = {(ra.y & 0xFFFFu) != 0u, (rb.y & 0xFFFFu) != 0u};   // padding records inside a step have count 0
    const unsigned wx[2] = {ra.x, rb.x};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!on[k]) continue;
      double v = acc[k];
      switch (cls) {                                 // wave-uniform
        case 0:
          if (c.gsh != 0.0 && diag_flag[e[k]]) v += c.gsh;                                     // precompile.jl:529-534
          if (!isfinite(v)) *c.nf = 1;
          c.Gb[e[k]] = v;
          break;
        case 1:
          if (!isfinite(v)) *c.nf = 1;
          c.Cb[e[k]] = v;
          break;
        case 2:
          if (c.sf < 1.0) v *= c.sf;                                                           // precompile.jl:524-527
          if (!isfinite(v)) *c.nf = 1;
          c.bb[e[k]] = v;
          break;
        case 3: src[e[k]] = v; break;
        default: {
          const unsigned arr = (wx[k] >> 28) & 3u, md = wx[k] >> 30;
          if (arr == 2u && c.sf < 1.0) v *= c.sf;
          if (!isfinite(v)) *c.nf = 1;
          double* dst = (arr == 0u ? c.Gb : arr == 1u ? c.Cb : c.bb) + e[k];
          if (md == TGT_RMW) *dst += v;
          else unsafeAtomicAdd(dst, v);
        }
      }
    }
  };
  const bool single = a.ipw == 1;
  const bool single_on = single && grp < a.B && inst_par[0] != 0.0;                 // (ipw == 1: the tile's instance is grp)
  const InstCtx c0 = inst_ctx(0);
  // The loop body is PIPE steps (stage p of the register window serves steps p, p + PIPE, ...): a stage is refilled right
  // after it has been consumed.  Result stores are acknowledged in order with the record reads (one vmcnt counter), so the
  // window also decides how many steps of stores may be in flight: with 4 the loop ran at the store latency / 4 per step.
  auto step = [&](int q, const uint4& ra, const uint4& rb) {
    const int info = q < 64 ? __builtin_amdgcn_readlane(step_lane, q) : a.step_info[s0 + q];
    const int cls = info & 0xFF;
    if (info & 0x100) CADNIP_WAVE_SYNC();          // first step of a level: the partial sums below it are complete
    if (cls >= 5) return;
    if (single) { if (single_on) reduce_instance(c0, ra, rb, cls); return; }
    for (int ri = 0; ri < a.ipw; ++ri)
      if (grp * a.ipw + ri < a.B && inst_par[3 * ri] != 0.0) reduce_instance(inst_ctx(ri), ra, rb, cls);   // wave-uniform
  };
  for (int q0 = 0; q0 < n_steps; q0 += PIPE) {
#pragma unroll
    for (int p = 0; p < PIPE; ++p) {
      const int q = q0 + p;
      const uint4 ra = pipe[p][0], rb = pipe[p][1];
      const int st = q + PIPE < n_steps ? q + PIPE : 0;                              // past the end: a harmless re-read of step 0
      pipe[p][0] = recs[(size_t)st * STEP_W]; pipe[p][1] = recs[(size_t)st * STEP_W + 64];
      if (q < n_steps) step(q, ra, rb);                                              // wave-uniform
    }
  }
  CADNIP_WAVE_SYNC();
  SC_POINT(7);
}

// Pre-set words: (a) targets accumulated with atomics whose first contributions come from that same kernel, (b) diagonal
// entries of voltage nodes that no device stamps into G (a node held by capacitors only): they carry gshunt alone.
struct PrepArgs { const unsigned* words; int n_words; const unsigned char* diag_flag; const double* gshunt; const int* active; double *G, *C, *b; int B, n, nnz; };
__global__ void __launch_bounds__(256) k_stamp_prep(PrepArgs a) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)a.B * a.n_words) return;
  const int inst = (int)(tid / a.n_words);
  if (!a.active[inst]) return;
  const unsigned w = a.words[tid - (long)inst * a.n_words];
  const unsigned e = w & 0x0FFFFFFFu, arr = (w >> 28) & 3u;
  double v = 0.0;
  if (arr == 0u && a.diag_flag[e]) v = a.gshunt[inst];
  double* dst = arr == 0u ? a.G + (size_t)inst * a.nnz + e : arr == 1u ? a.C + (size_t)inst * a.nnz + e : a.b + (size_t)inst * a.n + e;
  *dst = v;
}

// ------------------------------------------------------------------------------------------
// host: the reduction plan, built once per structure (cadnip_create)
// ------------------------------------------------------------------------------------------
template <class T> static int upload_vec(T** p, const std::vector<T>& v) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIP_TRY(hipMalloc((void**)p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return CADNIP_OK;
}

int build_stamp_plan(CadnipHandle* h, const CadnipStructure* s) {
  const int n = h->n, nnz = h->nnz;
  // slot -> (block, k, dev) per array; blocks own disjoint slot ranges
  struct Owner { int blk, k, dev; };
  auto owners = [&](int which, int total) {
    std::vector<Owner> o((size_t)total, Owner{-1, 0, 0});
    for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
      const DeviceBlock& b = h->blocks[bi];
      if (b.count == 0) continue;
      const int base = which == 0 ? b.g_base : which == 1 ? b.c_base : b.b_base, nk = which == 0 ? b.n_g : which == 1 ? b.n_c : b.n_b;
      for (int k = 0; k < nk; ++k) for (int d = 0; d < b.count; ++d) o[(size_t)base + (size_t)k * b.count + d] = Owner{(int)bi, k, d};
    }
    return o;
  };
  // tile geometry per block
  for (auto& b : h->blocks) {
    if (b.count == 0) continue;
    const int nslots = b.n_g + b.n_c + b.n_b;
    int cs = b.type == CADNIP_DEV_MOS1 ? 32 : 64;                       // sp_mos1: room for two lanes per device
    if (b.type == CADNIP_DEV_VA && b.va_tl) cs = 64 / b.va_tl;           // external models: 16 or 32 direction lanes per device (va_runtime.hpp)
    while (cs > 1 && (size_t)cs * nslots * 8 > 96 * 1024) cs >>= 1;     // big generated models: smaller chunks
    if ((size_t)cs * nslots > 65535) return CADNIP_BADARG;              // 16-bit staging offsets
    if (b.count <= cs) { b.sp_cs = b.count; b.sp_chunks = 1; }
    else { b.sp_cs = cs; b.sp_chunks = (b.count + cs - 1) / cs; }
    b.sp_targets.clear();
  }
  // contributions of every target, in COO order, tagged with (block, chunk)
  struct Contrib { int blk, chunk; unsigned short off; };
  const int* ptrs[3] = {s->g_ptr, s->c_ptr, s->b_ptr};
  const int* slots[3] = {s->g_slots, s->c_slots, s->b_slots};
  const int n_tgt[3] = {nnz, nnz, n};
  const int totals[3] = {h->ns_g, h->ns_c, h->ns_b};
  std::vector<unsigned> prep, prep_orphan;   // atomically accumulated words nobody stores first | unstamped node diagonals
  std::vector<char> is_diag((size_t)nnz, 0);
  for (int i = 0; i < s->n_nodes; ++i) if (s->diag_nz[i] >= 0 && s->diag_nz[i] < nnz) is_diag[s->diag_nz[i]] = 1;
  std::vector<Contrib> cl;
  std::vector<int> order;
  for (int arr = 0; arr < 3; ++arr) {
    const std::vector<Owner> own = owners(arr, totals[arr]);
    for (int e = 0; e < n_tgt[arr]; ++e) {
      cl.clear();
      for (int p = ptrs[arr][e]; p < ptrs[arr][e + 1]; ++p) {
        const Owner& o = own[(size_t)slots[arr][p]];
        if (o.blk < 0) return CADNIP_BADARG;                            // a gather list names a slot no block owns
        const DeviceBlock& b = h->blocks[o.blk];
        const int kk = o.k + (arr == 0 ? 0 : arr == 1 ? b.n_g : b.n_g + b.n_c);
        cl.push_back(Contrib{o.blk, o.dev / b.sp_cs, (unsigned short)(kk * b.sp_cs + o.dev % b.sp_cs)});
      }
      if (cl.empty()) {
        // a G entry nobody stamps stays zero for ever -- unless it is a node diagonal, which carries gshunt
        if (arr == 0 && is_diag[e]) prep_orphan.push_back((unsigned)e);
        continue;
      }
      // group by tile (block, chunk) in launch order; inside a tile the COO order is kept
      order.resize(cl.size());
      for (size_t i = 0; i < cl.size(); ++i) order[i] = (int)i;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cl[x].blk != cl[y].blk ? cl[x].blk < cl[y].blk : cl[x].chunk < cl[y].chunk; });
      const int first_blk = cl[order[0]].blk;
      for (size_t i = 0; i < order.size();) {
        const int blk = cl[order[i]].blk, chunk = cl[order[i]].chunk;
        size_t j = i;
        DeviceBlock::Target t;
        t.chunk = chunk;
        while (j < order.size() && cl[order[j]].blk == blk && cl[order[j]].chunk == chunk) t.offs.push_back(cl[order[j++]].off);
        // does another tile of the same kernel contribute too?
        const bool shared_in_kernel = (i > 0 && cl[order[i - 1]].blk == blk) || (j < order.size() && cl[order[j]].blk == blk);
        const int mode = shared_in_kernel ? TGT_ATOMIC : (blk == first_blk ? TGT_STORE : TGT_RMW);
        if (mode == TGT_ATOMIC && blk == first_blk && i == 0) prep.push_back(((unsigned)arr << 28) | (unsigned)e);   // nobody stores it first
        t.word = ((unsigned)mode << 30) | ((unsigned)arr << 28) | (unsigned)e;
        h->blocks[blk].sp_targets.push_back(std::move(t));
        i = j;
      }
    }
  }
  // per block: targets grouped by chunk (stable: array, then CSR position), uploaded as flat arrays
  for (auto& b : h->blocks) {
    if (b.count == 0) continue;
    std::stable_sort(b.sp_targets.begin(), b.sp_targets.end(), [](const DeviceBlock::Target& x, const DeviceBlock::Target& y) { return x.chunk < y.chunk; });
    // record = {destination word, count | off0 << 16, off1 | off2 << 16, off3 | off4 << 16}: up to five operands inline.
    // A target with more contributions becomes a 5-ary tree: level-0 records sum consecutive runs of five staged words into
    // scratch words of the tile, the next level combines five of those, ... until one record is left, which carries the
    // target's real destination.  Records are grouped per chunk and level.
    const int nslots = b.n_g + b.n_c + b.n_b, stage_words = nslots * b.sp_cs;
    struct Rec { int chunk, level, cls; uint4 r; };
    auto cls_of = [](unsigned word) { const unsigned md = word >> 30, arr = (word >> 28) & 3u; return md == TGT_PARTIAL ? 3 : md == TGT_STORE ? (int)arr : 4; };
    std::vector<Rec> recs;
    std::vector<int> scratch_used(b.sp_chunks, 0);
    int n_levels = 1;
    // (the zero word's offset is known only after the scratch words are counted: unused operand slots are patched below)
    const unsigned UNUSED = 0xFFFFu;
    auto pack = [&](unsigned word, const unsigned* o, unsigned cnt) {
      uint4 r; unsigned v[5] = {UNUSED, UNUSED, UNUSED, UNUSED, UNUSED};
      for (unsigned i = 0; i < cnt; ++i) v[i] = o[i];
      r.x = word; r.y = cnt | (v[0] << 16); r.z = v[1] | (v[2] << 16); r.w = v[3] | (v[4] << 16);
      return r;
    };
    for (auto& t : b.sp_targets) {
      std::vector<unsigned> cur(t.offs.begin(), t.offs.end());
      int level = 0;
      while (cur.size() > 5) {
        std::vector<unsigned> next;
        for (size_t i = 0; i < cur.size(); i += 5) {
          const unsigned cnt = (unsigned)std::min<size_t>(5, cur.size() - i);
          if (cnt == 1) { next.push_back(cur[i]); continue; }             // a lone tail word moves up as it is
          const unsigned so = (unsigned)(stage_words + scratch_used[t.chunk]++);
          recs.push_back(Rec{t.chunk, level, 3, pack(((unsigned)TGT_PARTIAL << 30) | so, &cur[i], cnt)});
          next.push_back(so);
        }
        cur.swap(next);
        ++level;
      }
      recs.push_back(Rec{t.chunk, level, cls_of(t.word), pack(t.word, cur.data(), (unsigned)cur.size())});
      n_levels = std::max(n_levels, level + 1);
    }
    int n_scratch = 0;
    for (int c = 0; c < b.sp_chunks; ++c) n_scratch = std::max(n_scratch, scratch_used[c]);
    n_scratch += 1;                                                         // + the tile's zero word (its last word)
    if ((stage_words + n_scratch) & 1) n_scratch += 1;                      // tiles stay 16-byte aligned (zeroing uses 16-byte stores)
    if ((size_t)stage_words + n_scratch >= 65535) return CADNIP_BADARG;
    {
      const unsigned zero_off = (unsigned)(stage_words + n_scratch - 1);
      auto fix = [&](unsigned half) { return half == UNUSED ? zero_off : half; };
      for (auto& rc_ : recs) {
        uint4& r = rc_.r;
        r.y = (r.y & 0xFFFFu) | (fix(r.y >> 16) << 16);
        r.z = fix(r.z & 0xFFFFu) | (fix(r.z >> 16) << 16);
        r.w = fix(r.w & 0xFFFFu) | (fix(r.w >> 16) << 16);
      }
    }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.chunk != y.chunk ? x.chunk < y.chunk : x.level != y.level ? x.level < y.level : x.cls < y.cls; });
    // steps: runs of STEP_W records (two per lane), every step within one (level, class); a short last step of a group is
    // padded with count-0 records that sum the zero word and write nothing
    std::vector<int> sptr(b.sp_chunks + 1, 0), sinfo;
    std::vector<uint4> rec;
    {
      const unsigned zo = (unsigned)(stage_words + n_scratch - 1);
      uint4 padrec; padrec.x = 0; padrec.y = 0u | (zo << 16); padrec.z = zo | (zo << 16); padrec.w = zo | (zo << 16);
      size_t k = 0;
      for (int c = 0; c < b.sp_chunks; ++c) {
        sptr[c] = (int)sinfo.size();
        for (int l = 0; l < n_levels; ++l) {
          bool first_of_level = l > 0;
          for (int cls = 0; cls < N_CLS; ++cls) {
            size_t k1 = k;
            while (k1 < recs.size() && recs[k1].chunk == c && recs[k1].level == l && recs[k1].cls == cls) ++k1;
            for (size_t p = k; p < k1; p += STEP_W) {
              sinfo.push_back(cls | (first_of_level ? 0x100 : 0));
              first_of_level = false;
              for (size_t j = p; j < p + STEP_W; ++j) rec.push_back(j < k1 ? recs[j].r : padrec);
            }
            k = k1;
          }
          if (first_of_level) sinfo.push_back(5 | 0x100), rec.insert(rec.end(), STEP_W, padrec);   // an empty level still fences
        }
      }
      sptr[b.sp_chunks] = (int)sinfo.size();
    }
    b.sp_n_targets = (int)rec.size();
    b.sp_levels = n_levels; b.sp_scratch = n_scratch;
    int rc;
    if ((rc = upload_vec(&b.d_sp_tptr, sptr))) return rc;
    if ((rc = upload_vec(&b.d_sp_info, sinfo))) return rc;
    if ((rc = upload_vec(&b.d_sp_rec, rec))) return rc;
    b.sp_targets.clear(); b.sp_targets.shrink_to_fit();
  }
  h->n_prep_atomic = (int)prep.size();
  prep.insert(prep.end(), prep_orphan.begin(), prep_orphan.end());
  h->n_prep = (int)prep.size();
  if (h->n_prep) { int rc = upload_vec(&h->d_prep, prep); if (rc) return rc; }
  return CADNIP_OK;
}

// Setup pass of a generated external model (va_generated_ext.hpp: setup_va_<module>): every bias-independent statement of the module,
// one thread per (instance, device), whenever the block's parameters change (cadnip_set_params).  What the per-call stamp function
// reads of it lies in the block's cache [B][n_cache][count].
__global__ void __launch_bounds__(64) k_va_setup(const int* nodes, const int* ipar, const double* par, const double* wave, double* cache, int B, int count, int n_par, int n_cache, int mode) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= B * count) return;
  const int inst = idx / count, dev = idx - inst * count;
  DevCtx d{nodes, ipar, par + (size_t)inst * n_par * count, wave, count, dev, 0.0, mode, 0, nullptr};
  setup_va(d, cache + (size_t)inst * n_cache * count + dev, count);
}

int launch_va_setup(CadnipHandle* h, DeviceBlock& b) {
  if (b.n_cache <= 0 || !b.d_cache) return CADNIP_OK;
  const int total = h->B * b.count;
  hipLaunchKernelGGL(k_va_setup, dim3((total + 63) / 64), dim3(64), 0, h->stream, b.d_nodes, b.d_ipar, b.d_par, h->d_wave, b.d_cache, h->B, b.count, b.n_par, b.n_cache, h->spec.mode);
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

template <int TYPE>
static int launch_stamp_csr_t(CadnipHandle* h, DeviceBlock& b) {
  const int nslots = b.n_g + b.n_c + b.n_b;
  const bool pair = TYPE == CADNIP_DEV_MOS1 && b.mos1_plain;
  const int lpd = pair ? 2 : (TYPE == CADNIP_DEV_VA && b.va_tl) ? b.va_tl : 1;
  int ipw = 1;
  if (b.sp_chunks == 1) ipw = std::min(8, std::max(1, 64 / (b.count * lpd)));   // (a wave reduces its instances one after the other: few per wave)
  const size_t tile_words = (size_t)nslots * b.sp_cs + b.sp_scratch;
  while (ipw > 1 && (size_t)ipw * tile_words * 8 > 64 * 1024) --ipw;
  const int u_lds = (size_t)ipw * h->n * 8 <= 16 * 1024 ? 1 : 0;
  const size_t shmem = ((size_t)ipw * tile_words + 3 * (size_t)ipw + (u_lds ? (size_t)ipw * h->n : 0)) * 8;   // tiles (staged slots + tree scratch), per-instance scalars, u
  CsrStampArgs a{b.d_nodes, b.d_ipar, b.d_par, h->d_wave, h->d_u, h->d_t, h->d_active, h->d_cold, h->d_G, h->d_C, h->d_b, h->d_limit_w, h->d_nonfinite,
                 h->d_diag_flag, h->d_gshunt, h->d_srcfact, b.d_sp_tptr, b.d_sp_info, b.d_sp_rec,
                 h->B, b.count, h->n, h->nnz, b.n_par, b.n_g, b.n_c, b.n_b, b.sp_cs, b.sp_chunks, ipw, lpd, h->spec.mode, h->initjct,
                 (TYPE == CADNIP_DEV_MOS1 || TYPE == CADNIP_DEV_VA) ? 1 : 0, b.sp_levels, b.sp_scratch, u_lds,
                 b.d_cache, b.n_cache, h->d_dump, h->ns, b.g_base, h->ns_g + b.c_base, h->ns_g + h->ns_c + b.b_base};
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_stamp_csr<TYPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const unsigned grid = (unsigned)b.sp_chunks * (unsigned)((h->B + ipw - 1) / ipw);
  if (getenv("CADNIP_SC_DEBUG")) fprintf(stderr, "[cadnip stamp] type %d count %d cs %d chunks %d ipw %d lpd %d nslots %d scratch %d tile_words %zu shmem %zu grid %u steps %d levels %d\n",
                                         TYPE, b.count, b.sp_cs, b.sp_chunks, ipw, lpd, nslots, b.sp_scratch, tile_words, shmem, grid, 0, b.sp_levels);
  hipLaunchKernelGGL(k_stamp_csr<TYPE>, dim3(grid), dim3(64), shmem, h->stream, a);
  return CADNIP_OK;
}

// one stamping kernel alone (bench.py times it back to back for the stamp-kernel roofline line)
int launch_stamp_block(CadnipHandle* h, int block) {
  if (block < 0 || block >= (int)h->blocks.size() || h->blocks[block].count == 0) return CADNIP_BADARG;
  DeviceBlock& blk = h->blocks[block];
  switch (blk.type) {
#define CASE(T) case T: return launch_stamp_csr_t<T>(h, blk);
    CASE(CADNIP_DEV_RESISTOR) CASE(CADNIP_DEV_CAPACITOR) CASE(CADNIP_DEV_INDUCTOR) CASE(CADNIP_DEV_VSOURCE) CASE(CADNIP_DEV_ISOURCE)
    CASE(CADNIP_DEV_VCVS) CASE(CADNIP_DEV_VCCS) CASE(CADNIP_DEV_CCVS) CASE(CADNIP_DEV_CCCS) CASE(CADNIP_DEV_DIODE) CASE(CADNIP_DEV_DIODECAP)
    CASE(CADNIP_DEV_SIMPLEMOS) CASE(CADNIP_DEV_MOS1) CASE(CADNIP_DEV_BVSOURCE) CASE(CADNIP_DEV_BISOURCE) CASE(CADNIP_DEV_VA)
#undef CASE
  }
  return CADNIP_BADARG;
}

int launch_rebuild(CadnipHandle* h) {
  // the pre-set pass runs when some word is accumulated with atomics from scratch; the unstamped node diagonals carry
  // gshunt alone, so they are rewritten only while a homotopy is on and once after it has been switched off
  const bool prep_now = h->n_prep_atomic > 0 || (h->n_prep > 0 && (h->homotopy || h->spec.gshunt != 0.0 || h->prep_stale));
  h->prep_stale = h->n_prep > h->n_prep_atomic && (h->homotopy || h->spec.gshunt != 0.0);
  if (prep_now) {
    ProfScope ps(h, "stamp_prep");
    PrepArgs p{h->d_prep, h->n_prep, h->d_diag_flag, h->d_gshunt, h->d_active, h->d_G, h->d_C, h->d_b, h->B, h->n, h->nnz};
    const long total = (long)h->B * h->n_prep;
    hipLaunchKernelGGL(k_stamp_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, p);
  }
  for (auto& blk : h->blocks) {
    if (blk.count == 0) continue;
    int rc = CADNIP_OK;
    switch (blk.type) {
#define CASE(T, NAME) case T: { ProfScope ps(h, NAME); rc = launch_stamp_csr_t<T>(h, blk); } break;
      CASE(CADNIP_DEV_RESISTOR, "stamp_resistor") CASE(CADNIP_DEV_CAPACITOR, "stamp_capacitor")
      CASE(CADNIP_DEV_INDUCTOR, "stamp_inductor") CASE(CADNIP_DEV_VSOURCE, "stamp_vsource")
      CASE(CADNIP_DEV_ISOURCE, "stamp_isource") CASE(CADNIP_DEV_VCVS, "stamp_vcvs") CASE(CADNIP_DEV_VCCS, "stamp_vccs")
      CASE(CADNIP_DEV_CCVS, "stamp_ccvs") CASE(CADNIP_DEV_CCCS, "stamp_cccs") CASE(CADNIP_DEV_DIODE, "stamp_diode")
      CASE(CADNIP_DEV_DIODECAP, "stamp_diodecap") CASE(CADNIP_DEV_SIMPLEMOS, "stamp_simplemos")
      CASE(CADNIP_DEV_MOS1, "stamp_mos1") CASE(CADNIP_DEV_BVSOURCE, "stamp_bvsource") CASE(CADNIP_DEV_BISOURCE, "stamp_bisource")
      CASE(CADNIP_DEV_VA, "stamp_va")
#undef CASE
      default: return CADNIP_BADARG;
    }
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

}  // namespace cadnip

#ifdef CADNIP_TRACE
extern "C" int cadnip_debug_stamp_trace(unsigned long long* sum8, unsigned long long* cnt, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return CADNIP_HIPERROR;
  if (hipMemcpyFromSymbol(sum8, HIP_SYMBOL(cadnip::g_sc_sum), 8 * sizeof(unsigned long long)) != hipSuccess) return CADNIP_HIPERROR;
  if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(cadnip::g_sc_cnt), sizeof(unsigned long long)) != hipSuccess) return CADNIP_HIPERROR;
  if (reset) { unsigned long long z[9] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(cadnip::g_sc_sum), z, 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(cadnip::g_sc_cnt), z, 8); }
  return CADNIP_OK;
}
#endif
