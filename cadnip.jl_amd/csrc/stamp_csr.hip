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
#include "stamp_csr_kernel.hpp"   // CsrStampArgs, LdsOut, k_stamp_csr<TYPE, EXT>

namespace cadnip {

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

static int build_plan_variant(CadnipHandle* h, const CadnipStructure* s, bool plain) {
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
  // Rows of a tile.  A tile stages [row][device]; only the slots that some target reads need a row of their own (a stamp into a
  // ground row / column has no target, nor has the unused form -- charge state or linear -- of a reactive branch): they are packed,
  // every other slot writes into one shared trash row, and a slot whose value is structurally zero is read from the tile's zero
  // word.  Fewer rows = less LDS per wave = more waves per CU: this kernel's duration follows its occupancy.
  const unsigned short ROW_ZERO = 0xFFFEu, ROW_NONE = 0xFFFFu;
  std::vector<std::vector<unsigned short>> rowmap(h->blocks.size());
  for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
    const DeviceBlock& b = h->blocks[bi];
    rowmap[bi].assign((size_t)(b.n_g + b.n_c + b.n_b), ROW_NONE);
  }
  auto packs = [](const DeviceBlock& b) { return b.type == CADNIP_DEV_MOS1 || b.type == CADNIP_DEV_VA; };   // (stamp_csr_kernel.hpp: REMAP)
  for (int arr = 0; arr < 3; ++arr) {
    const std::vector<Owner> own = owners(arr, totals[arr]);
    for (int e = 0; e < n_tgt[arr]; ++e)
      for (int p = ptrs[arr][e]; p < ptrs[arr][e + 1]; ++p) {
        const Owner& o = own[(size_t)slots[arr][p]];
        if (o.blk < 0) return CADNIP_BADARG;
        const DeviceBlock& b = h->blocks[o.blk];
        rowmap[o.blk][(size_t)(o.k + (arr == 0 ? 0 : arr == 1 ? b.n_g : b.n_g + b.n_c))] = 0;      // live
      }
  }
  for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
    DeviceBlock& b = h->blocks[bi];
    if (b.count == 0) continue;
    if (b.type == CADNIP_DEV_MOS1)                      // devices.hpp: the charge rows' d/dV_d and d/dV_s entries (gq[1], gq[3]) and their linear
      for (int r = 0; r < 4; ++r)                       // form (dq[0], dq[2]) are zeros whatever the parameters
        for (int k : {48 + 7 * r + 1, 48 + 7 * r + 3, b.n_g + 4 + 6 * r, b.n_g + 4 + 6 * r + 2})
          if (rowmap[bi][(size_t)k] != ROW_NONE) rowmap[bi][(size_t)k] = ROW_ZERO;
    if (b.type == CADNIP_DEV_MOS1 && plain) {           // the lane-pair path (stamp_mos1_pair; gd = gs = OxideCap = 0): the KCL rows of the external
      auto zero = [&](int k) { if (rowmap[bi][(size_t)k] != ROW_NONE) rowmap[bi][(size_t)k] = ROW_ZERO; };   // d, g, s terminals carry nothing ...
      for (int k = 12; k < 12 + 18; ++k) zero(k);
      for (int br = 3; br < 6; ++br) { zero(12 + 6 * br); zero(12 + 6 * br + 2); }      // ... and no row has an entry in the d / s columns
      for (int k = 0; k < 3; ++k) zero(b.n_g + b.n_c + k);                               // b of rows d, g, s
    }
    if (!packs(b)) {                                    // one row per slot, no table
      for (size_t k = 0; k < rowmap[bi].size(); ++k) rowmap[bi][k] = (unsigned short)k;
      b.sp_rows = (int)rowmap[bi].size();
      if (b.d_sp_rowoff) { (void)hipFree(b.d_sp_rowoff); b.d_sp_rowoff = nullptr; }
      continue;
    }
    int rows = 0;
    for (auto& r : rowmap[bi]) if (r == 0) r = (unsigned short)rows++;
    b.sp_rows = rows + 1;                               // + the trash row
    if ((size_t)b.sp_rows * b.sp_cs > 65000) return CADNIP_BADARG;   // 16-bit staging offsets
    std::vector<unsigned short> ro(rowmap[bi].size());
    for (size_t k = 0; k < ro.size(); ++k) ro[k] = (unsigned short)((rowmap[bi][k] >= ROW_ZERO ? rows : rowmap[bi][k]) * b.sp_cs);
    int rc = upload_vec(&b.d_sp_rowoff, ro);
    if (rc) return rc;
  }
  const unsigned OFF_ZERO = 0xFFFFu;                    // operand that reads the tile's zero word (patched below, like an unused operand slot)
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
        const unsigned short row = rowmap[o.blk][(size_t)kk];
        cl.push_back(Contrib{o.blk, o.dev / b.sp_cs, row == ROW_ZERO ? (unsigned short)OFF_ZERO : (unsigned short)(row * b.sp_cs + o.dev % b.sp_cs)});
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
    const int stage_words = b.sp_rows * b.sp_cs;
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

int build_stamp_plan(CadnipHandle* h, const CadnipStructure* s) {
  auto take = [](DeviceBlock& b) {
    DeviceBlock::PlanSet p;
    p.n_targets = b.sp_n_targets; p.levels = b.sp_levels; p.scratch = b.sp_scratch; p.rows = b.sp_rows;
    p.tptr = b.d_sp_tptr; p.info = b.d_sp_info; p.rec = b.d_sp_rec; p.rowoff = b.d_sp_rowoff;
    b.d_sp_tptr = nullptr; b.d_sp_info = nullptr; b.d_sp_rec = nullptr; b.d_sp_rowoff = nullptr;
    return p;
  };
  bool any_mos1 = false;
  for (auto& b : h->blocks) any_mos1 = any_mos1 || (b.type == CADNIP_DEV_MOS1 && b.count > 0);
  int rc = build_plan_variant(h, s, false);
  if (rc || !any_mos1) return rc;
  for (auto& b : h->blocks) if (b.type == CADNIP_DEV_MOS1 && b.count > 0) b.sp_gen = take(b);
  rc = build_plan_variant(h, s, true);
  if (rc) return rc;
  for (auto& b : h->blocks) if (b.type == CADNIP_DEV_MOS1 && b.count > 0) b.sp_plain = take(b);
  return CADNIP_OK;
}

// The external generated models: one translation unit each (va_ext/<module>.hip, written by va/hipgen.py) with the model's own
// instantiation of k_stamp_csr and of its setup kernel; reached through these tables (model id - CADNIP_VA_NBUILTIN).
#define X(i, nm) int va_ext_stamp_launch_##nm(const CsrStampArgs&, unsigned, size_t, hipStream_t); int va_ext_setup_launch_##nm(const VaSetupArgs&, hipStream_t);
CADNIP_VA_EXT_LIST(X)
#undef X
typedef int (*VaExtStampFn)(const CsrStampArgs&, unsigned, size_t, hipStream_t);
typedef int (*VaExtSetupFn)(const VaSetupArgs&, hipStream_t);
#define X(i, nm) va_ext_stamp_launch_##nm,
static const VaExtStampFn VA_EXT_STAMP[CADNIP_VA_NEXT + 1] = {CADNIP_VA_EXT_LIST(X) nullptr};
#undef X
#define X(i, nm) va_ext_setup_launch_##nm,
static const VaExtSetupFn VA_EXT_SETUP[CADNIP_VA_NEXT + 1] = {CADNIP_VA_EXT_LIST(X) nullptr};
#undef X

int launch_va_setup(CadnipHandle* h, DeviceBlock& b) {
  if (b.n_cache <= 0 || !b.d_cache) return CADNIP_OK;
  const int ext = b.va_model - CADNIP_VA_NBUILTIN;
  if (ext < 0 || ext >= CADNIP_VA_NEXT) return CADNIP_BADARG;
  VaSetupArgs a{b.d_nodes, b.d_ipar, b.d_par, h->d_wave, b.d_cache, h->B, b.count, b.n_par, b.n_cache, h->spec.mode};
  int rc = VA_EXT_SETUP[ext](a, h->stream);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return CADNIP_OK;
}

template <int TYPE>
static int launch_stamp_csr_pass(CadnipHandle* h, DeviceBlock& b, bool dump_only) {
  const int nslots = b.n_g + b.n_c + b.n_b;
  const bool pair = TYPE == CADNIP_DEV_MOS1 && b.mos1_plain;
  const int lpd = pair ? 2 : (TYPE == CADNIP_DEV_VA && b.va_tl) ? b.va_tl : 1;
  DeviceBlock::PlanSet P;                              // the plan in force
  if (TYPE == CADNIP_DEV_MOS1) P = b.mos1_plain ? b.sp_plain : b.sp_gen;
  else { P.levels = b.sp_levels; P.scratch = b.sp_scratch; P.rows = b.sp_rows; P.tptr = b.d_sp_tptr; P.info = b.d_sp_info; P.rec = b.d_sp_rec; P.rowoff = b.d_sp_rowoff; }
  const int rows = dump_only ? nslots : P.rows;        // the read-out pass stages every slot in a row of its own
  int ipw = 1;
  if (b.sp_chunks == 1) ipw = std::min(8, std::max(1, 64 / (b.count * lpd)));   // (a wave reduces its instances one after the other: few per wave)
  const size_t tile_words = (size_t)rows * b.sp_cs + P.scratch;
  while (ipw > 1 && (size_t)ipw * tile_words * 8 > 64 * 1024) --ipw;
  const int u_lds = (size_t)ipw * h->n * 8 <= 16 * 1024 ? 1 : 0;
  static const size_t lds_pad = getenv("CADNIP_SC_PAD") ? (size_t)atol(getenv("CADNIP_SC_PAD")) : 0;        // experiments: occupancy as a function of the LDS request
  // tiles (staged rows + tree scratch), per-instance scalars, u, the slots' row offsets
  const size_t shmem = ((size_t)ipw * tile_words + 3 * (size_t)ipw + (u_lds ? (size_t)ipw * h->n : 0)) * 8 + (((size_t)nslots * 2 + 7) & ~(size_t)7) + lds_pad;
  CsrStampArgs a{b.d_nodes, b.d_ipar, b.d_par, h->d_wave, h->d_u, h->d_t, h->d_active, h->d_cold, h->d_G, h->d_C, h->d_b, h->d_limit_w, h->d_nonfinite,
                 h->d_diag_flag, h->d_gshunt, h->d_srcfact, P.tptr, P.info, P.rec,
                 h->B, b.count, h->n, h->nnz, b.n_par, b.n_g, b.n_c, b.n_b, b.sp_cs, b.sp_chunks, ipw, lpd, h->spec.mode, h->initjct,
                 (TYPE == CADNIP_DEV_MOS1 || TYPE == CADNIP_DEV_VA) ? 1 : 0, P.levels, P.scratch, u_lds,
                 b.d_cache, b.n_cache, dump_only ? h->d_dump : nullptr, h->ns, b.g_base, h->ns_g + b.c_base, h->ns_g + h->ns_c + b.b_base,
                 dump_only ? nullptr : P.rowoff, rows, dump_only ? 1 : 0};
  const unsigned grid = (unsigned)b.sp_chunks * (unsigned)((h->B + ipw - 1) / ipw);
  if (getenv("CADNIP_SC_DEBUG")) fprintf(stderr, "[cadnip stamp] type %d count %d cs %d chunks %d ipw %d lpd %d slots %d rows %d scratch %d tile_words %zu shmem %zu grid %u levels %d%s\n",
                                         TYPE, b.count, b.sp_cs, b.sp_chunks, ipw, lpd, nslots, rows, P.scratch, tile_words, shmem, grid, P.levels, dump_only ? " (read-out pass)" : "");
  if (TYPE == CADNIP_DEV_VA && b.va_tl) {           // external model: its own kernel (va_ext/<module>.hip)
    const int ext = b.va_model - CADNIP_VA_NBUILTIN;
    if (ext < 0 || ext >= CADNIP_VA_NEXT) return CADNIP_BADARG;
    return VA_EXT_STAMP[ext](a, grid, shmem, h->stream);
  }
  return launch_stamp_kernel<TYPE>(a, grid, shmem, h->stream);
}

// The stamping pass of one block; with the operating-point read-out armed (cadnip_get_contributions), a second pass that stages every
// slot -- also those no target reads: the current into a grounded terminal is one of them -- and writes them out instead of reducing.
template <int TYPE>
static int launch_stamp_csr_t(CadnipHandle* h, DeviceBlock& b) {
  int rc = launch_stamp_csr_pass<TYPE>(h, b, false);
  if (!rc && h->d_dump) rc = launch_stamp_csr_pass<TYPE>(h, b, true);
  return rc;
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
