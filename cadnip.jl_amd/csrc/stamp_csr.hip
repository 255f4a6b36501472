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
//                 addition) and writes ONE value to HBM.  Consecutive lanes write consecutive CSR positions.
// No slot buffer in HBM, no separate assemble pass, no zero-fill of G / C / b: a target whose contributions all come from
// one tile is stored; a target that an earlier kernel of the stream has already stored is read-modify-written (kernels of a
// stream run in order, and within this kernel the tile is its only writer); only a target that receives contributions from
// several tiles of the SAME kernel -- a boundary between tiles, e.g. a supply rail fed by every chunk of a large circuit --
// is accumulated with a global fp64 atomic, on a word that k_stamp_prep has pre-set when no earlier kernel stores it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "devices.hpp"
#include "internal.hpp"
#include "tran_ctrl.hpp"   // CADNIP_WAVE_SYNC

namespace cadnip {

enum { TGT_STORE = 0, TGT_RMW = 1, TGT_ATOMIC = 2 };   // bits 30-31 of a target word; bits 28-29: 0 G, 1 C, 2 b; bits 0-27: index

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
  const double* u; const double* t; const int* active;
  double *G, *C, *b, *limit_w; int* nonfinite;
  const unsigned char* diag_flag; const double* gshunt; const double* srcFact;
  const int* tgt_ptr; const unsigned* tgt_dst; const int* tgt_lptr; const unsigned short* lst;
  int B, count, n, nnz, n_par, n_g, n_c, n_b, cs, n_chunks, ipw, lpd, mode, initjct, zero_first;
};

template <int TYPE>
__global__ void __launch_bounds__(64) k_stamp_csr(CsrStampArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int chunk = blockIdx.x % a.n_chunks, grp = blockIdx.x / a.n_chunks;
  const int nslots = a.n_g + a.n_c + a.n_b, tile_words = nslots * a.cs;
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
    for (int i = lane; i < a.ipw * tile_words; i += 64) lds[i] = 0.0;
    CADNIP_WAVE_SYNC();
  }
  {
    DevCtx d{a.nodes, a.ipar, a.par + (size_t)inst_c * a.n_par * a.count, a.wave, a.count, dev < a.count ? dev : a.count - 1, a.t[inst_c], a.mode, a.initjct};
    LdsOut s{tile, tile + (size_t)a.n_g * a.cs, tile + (size_t)(a.n_g + a.n_c) * a.cs, a.cs, ldev, valid};
    const double* u = a.u + (size_t)inst_c * a.n;
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
    else if (TYPE == CADNIP_DEV_VA) stamp_va(d, u, s, lw);
  }
  CADNIP_WAVE_SYNC();
  // ---- segmented reduction: one target per lane and step, contributions summed in COO order out of LDS
  const int t0 = a.tgt_ptr[chunk], nt = a.tgt_ptr[chunk + 1] - t0;
  for (int idx = lane; idx < a.ipw * nt; idx += 64) {
    const int ri = idx / nt, t = t0 + (idx - ri * nt);
    const int rinst = grp * a.ipw + ri;
    if (rinst >= a.B || !a.active[rinst]) continue;
    const unsigned w = a.tgt_dst[t];
    const unsigned e = w & 0x0FFFFFFFu, arr = (w >> 28) & 3u, md = w >> 30;
    const double* src = lds + (size_t)ri * tile_words;
    double acc = 0.0;
    for (int p = a.tgt_lptr[t]; p < a.tgt_lptr[t + 1]; ++p) acc += src[a.lst[p]];
    if (arr == 2u) { const double sf = a.srcFact[rinst]; if (sf < 1.0) acc *= sf; }          // precompile.jl:524-527
    if (!isfinite(acc)) a.nonfinite[rinst] = 1;
    double* dst = arr == 0u ? a.G + (size_t)rinst * a.nnz + e : arr == 1u ? a.C + (size_t)rinst * a.nnz + e : a.b + (size_t)rinst * a.n + e;
    if (md == TGT_STORE) {
      if (arr == 0u) { const double gsh = a.gshunt[rinst]; if (gsh != 0.0 && a.diag_flag[e]) acc += gsh; }   // precompile.jl:529-534
      *dst = acc;
    } else if (md == TGT_RMW) *dst += acc;
    else unsafeAtomicAdd(dst, acc);
  }
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
  std::vector<unsigned> prep;
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
        if (arr == 0 && is_diag[e]) prep.push_back((unsigned)e);
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
    std::vector<int> tptr(b.sp_chunks + 1, 0), lptr(1, 0);
    std::vector<unsigned> dst;
    std::vector<unsigned short> lst;
    for (auto& t : b.sp_targets) {
      tptr[t.chunk + 1] += 1;
      dst.push_back(t.word);
      lst.insert(lst.end(), t.offs.begin(), t.offs.end());
      lptr.push_back((int)lst.size());
    }
    for (int c = 0; c < b.sp_chunks; ++c) tptr[c + 1] += tptr[c];
    b.sp_n_targets = (int)dst.size();
    int rc;
    if ((rc = upload_vec(&b.d_sp_tptr, tptr))) return rc;
    if ((rc = upload_vec(&b.d_sp_dst, dst))) return rc;
    if ((rc = upload_vec(&b.d_sp_lptr, lptr))) return rc;
    if ((rc = upload_vec(&b.d_sp_lst, lst))) return rc;
    b.sp_targets.clear(); b.sp_targets.shrink_to_fit();
  }
  h->n_prep = (int)prep.size();
  if (h->n_prep) { int rc = upload_vec(&h->d_prep, prep); if (rc) return rc; }
  return CADNIP_OK;
}

template <int TYPE>
static int launch_stamp_csr_t(CadnipHandle* h, DeviceBlock& b) {
  const int nslots = b.n_g + b.n_c + b.n_b;
  const bool pair = TYPE == CADNIP_DEV_MOS1 && b.mos1_plain;
  const int lpd = pair ? 2 : 1;
  int ipw = 1;
  if (b.sp_chunks == 1) ipw = std::max(1, 64 / (b.count * lpd));
  while (ipw > 1 && (size_t)ipw * nslots * b.sp_cs * 8 > 64 * 1024) --ipw;
  const size_t shmem = (size_t)ipw * nslots * b.sp_cs * 8;
  CsrStampArgs a{b.d_nodes, b.d_ipar, b.d_par, h->d_wave, h->d_u, h->d_t, h->d_active, h->d_G, h->d_C, h->d_b, h->d_limit_w, h->d_nonfinite,
                 h->d_diag_flag, h->d_gshunt, h->d_srcfact, b.d_sp_tptr, b.d_sp_dst, b.d_sp_lptr, b.d_sp_lst,
                 h->B, b.count, h->n, h->nnz, b.n_par, b.n_g, b.n_c, b.n_b, b.sp_cs, b.sp_chunks, ipw, lpd, h->spec.mode, h->initjct,
                 (TYPE == CADNIP_DEV_MOS1 || TYPE == CADNIP_DEV_VA) ? 1 : 0};
  if (shmem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_stamp_csr<TYPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const unsigned grid = (unsigned)b.sp_chunks * (unsigned)((h->B + ipw - 1) / ipw);
  hipLaunchKernelGGL(k_stamp_csr<TYPE>, dim3(grid), dim3(64), shmem, h->stream, a);
  return CADNIP_OK;
}

int launch_rebuild(CadnipHandle* h) {
  if (h->n_prep > 0) {
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
