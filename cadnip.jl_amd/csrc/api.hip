// api.hip -- the C ABI of include/cadnip_hip.h (everything except the two host drivers).
#include <hip/hip_runtime.h>
#include <functional>
#include <algorithm>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "internal.hpp"
#include "va_generated.hpp"   // CADNIP_VA_SHAPES: the generated Verilog-A modules this library was built with

using namespace cadnip;

namespace {
template <class T>
int dev_alloc(T** p, size_t count) {
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
  HIP_TRY(hipMemset(*p, 0, count * sizeof(T)));
  return CADNIP_OK;
}
template <class T>
int dev_upload(T** p, const T* src, size_t count) {
  int rc = dev_alloc(p, count);
  if (rc) return rc;
  if (count) HIP_TRY(hipMemcpy(*p, src, count * sizeof(T), hipMemcpyHostToDevice));
  return CADNIP_OK;
}
template <class T>
int dev_upload(T** p, const std::vector<T>& v) { return dev_upload(p, v.data(), v.size()); }
#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
}  // namespace

extern "C" {

const char* cadnip_version(void) { return "cadnip_hip 0.1.0 (gfx950)"; }

void cadnip_destroy(CadnipHandle* h);
// inside cadnip_create, once the handle exists: a failing step releases everything created so far
#define CREATE_TRY(x) do { int _rc = (x); if (_rc) { cadnip_destroy(h); return _rc; } } while (0)
#define CREATE_HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cadnip::set_last_error(#expr, _e); cadnip_destroy(h); return CADNIP_HIPERROR; } } while (0)

int cadnip_create(const CadnipStructure* s, int32_t n_instances, int32_t device, CadnipHandle** out) {
  if (!s || !out || n_instances <= 0 || s->n <= 0 || s->nnz < 0 || s->n_limits < 0 || s->n_nodes < 0 || s->n_nodes > s->n) return CADNIP_BADARG;
  // Host-side shape checks first, before anything is allocated: every pointer the structure must carry, every prefix
  // array (start 0, non-decreasing, end = length of its list) and every index a kernel will dereference.
  if (!s->rowptr || !s->colidx || !s->to_ref_nz || !s->g_ptr || !s->c_ptr || !s->b_ptr || !s->diag_nz) return CADNIP_BADARG;
  if (s->n_limits > 0 && !s->limit_init) return CADNIP_BADARG;
  if (s->n_blocks < 0 || (s->n_blocks > 0 && !s->blocks) || s->n_wave_data < 0 || (s->n_wave_data > 0 && !s->wave_data)) return CADNIP_BADARG;
  if (s->ns_g < 0 || s->ns_c < 0 || s->ns_b < 0) return CADNIP_BADARG;
  {
    auto prefix_ok = [](const int32_t* p, int len) {
      if (p[0] != 0) return false;
      for (int i = 0; i < len; ++i) if (p[i] > p[i + 1]) return false;
      return true;
    };
    if (!prefix_ok(s->rowptr, s->n) || s->rowptr[s->n] != s->nnz) return CADNIP_BADARG;
    if (!prefix_ok(s->g_ptr, s->nnz) || !prefix_ok(s->c_ptr, s->nnz) || !prefix_ok(s->b_ptr, s->n)) return CADNIP_BADARG;
    if ((s->g_ptr[s->nnz] > 0 && !s->g_slots) || (s->c_ptr[s->nnz] > 0 && !s->c_slots) || (s->b_ptr[s->n] > 0 && !s->b_slots)) return CADNIP_BADARG;
    for (int k = 0; k < s->nnz; ++k) if (s->colidx[k] < 0 || s->colidx[k] >= s->n || s->to_ref_nz[k] < 0 || s->to_ref_nz[k] >= s->nnz) return CADNIP_BADARG;
    for (int k = 0; k < s->g_ptr[s->nnz]; ++k) if (s->g_slots[k] < 0 || s->g_slots[k] >= s->ns_g) return CADNIP_BADARG;
    for (int k = 0; k < s->c_ptr[s->nnz]; ++k) if (s->c_slots[k] < 0 || s->c_slots[k] >= s->ns_c) return CADNIP_BADARG;
    for (int k = 0; k < s->b_ptr[s->n]; ++k) if (s->b_slots[k] < 0 || s->b_slots[k] >= s->ns_b) return CADNIP_BADARG;
  }
  HIP_TRY(hipSetDevice(device));
  CadnipHandle* h = new CadnipHandle();
  h->device = device;
  h->B = n_instances;
  h->n = s->n; h->n_nodes = s->n_nodes; h->n_currents = s->n_currents; h->n_charges = s->n_charges; h->n_limits = s->n_limits;
  h->nnz = s->nnz;
  h->ns_g = s->ns_g; h->ns_c = s->ns_c; h->ns_b = s->ns_b; h->ns = s->ns_g + s->ns_c + s->ns_b;
  // Host-pointer transfers of this ABI are BLOCKING copies -- uploads at API entry, where the handle's stream is idle; downloads after a stream
  // synchronisation -- not asynchronous copies on the stream: an asynchronous copy from pageable host memory is not reliably ordered with
  // the kernels of a non-blocking stream (observed under `rocprofv3 --pmc`: cadnip_dc_run's start state landed after the first stamping
  // kernels had run, the residual of the stale state was not finite and every instance failed; driver.hip: k_publish_int has the same story
  // for the other direction).
  CREATE_HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CREATE_HIP_TRY(hipEventCreate(&h->ev0));
  CREATE_HIP_TRY(hipEventCreate(&h->ev1));
  CREATE_HIP_TRY(hipHostMalloc((void**)&h->h_pinned, 64 * sizeof(int), hipHostMallocMapped));
  CREATE_HIP_TRY(hipHostGetDevicePointer((void**)&h->d_pinned, h->h_pinned, 0));
  h->stage_bytes = (size_t)2 << 20;
  CREATE_HIP_TRY(hipHostMalloc((void**)&h->h_stage, h->stage_bytes, hipHostMallocMapped));
  CREATE_HIP_TRY(hipHostGetDevicePointer((void**)&h->d_stage, h->h_stage, 0));
  h->h_rowptr.assign(s->rowptr, s->rowptr + s->n + 1);
  h->h_colidx.assign(s->colidx, s->colidx + s->nnz);
  h->h_to_ref.assign(s->to_ref_nz, s->to_ref_nz + s->nnz);
  h->h_limit_init.assign(s->limit_init, s->limit_init + s->n_limits);
  CREATE_TRY(dev_upload(&h->d_rowptr, s->rowptr, (size_t)s->n + 1));
  CREATE_TRY(dev_upload(&h->d_colidx, s->colidx, (size_t)s->nnz));
  CREATE_TRY(dev_upload(&h->d_to_ref, s->to_ref_nz, (size_t)s->nnz));
  CREATE_TRY(dev_upload(&h->d_g_ptr, s->g_ptr, (size_t)s->nnz + 1));
  CREATE_TRY(dev_upload(&h->d_g_slots, s->g_slots, (size_t)s->g_ptr[s->nnz]));
  CREATE_TRY(dev_upload(&h->d_c_ptr, s->c_ptr, (size_t)s->nnz + 1));
  CREATE_TRY(dev_upload(&h->d_c_slots, s->c_slots, (size_t)s->c_ptr[s->nnz]));
  CREATE_TRY(dev_upload(&h->d_b_ptr, s->b_ptr, (size_t)s->n + 1));
  CREATE_TRY(dev_upload(&h->d_b_slots, s->b_slots, (size_t)s->b_ptr[s->n]));
  {
    // the device-local unknowns (charges, limits) and which of them carry a constant-1 diagonal (one G stamp, no C stamp on their own
    // diagonal entry): the symbolic phase marks such pivots `unit` -- nothing divides by them.  CADNIP_LU_NOLEAF=1 switches that off,
    // CADNIP_LU_LEAF_FIRST=1 additionally pivots them before everything else (diagnostics).
    h->leaf_unit_ok.assign(s->n, 0);
    const int q0 = s->n_nodes + s->n_currents, l0 = s->n - s->n_limits;
    for (int i = q0; i < s->n; ++i)
      for (int e = s->rowptr[i]; e < s->rowptr[i + 1]; ++e)
        if (s->colidx[e] == i) h->leaf_unit_ok[i] = (s->g_ptr[e + 1] - s->g_ptr[e] == 1 && s->c_ptr[e + 1] == s->c_ptr[e]) ? 1 : 0;
    if (!getenv("CADNIP_LU_NOLEAF") && q0 >= 0 && q0 <= l0 && l0 <= s->n && s->n_charges == l0 - q0) {
      h->leaves.q_begin = q0; h->leaves.lim_begin = l0; h->leaves.unit_ok = h->leaf_unit_ok.data(); h->leaves.first = getenv("CADNIP_LU_LEAF_FIRST") != nullptr;
    }
  }
  std::vector<unsigned char> dflag(s->nnz, 0);
  for (int i = 0; i < s->n_nodes; ++i) { int p = s->diag_nz[i]; if (p >= 0 && p < s->nnz) dflag[p] = 1; }
  CREATE_TRY(dev_upload(&h->d_diag_flag, dflag.data(), dflag.size()));
  {
    // rows longer than kernels.hip's LONG_LIST get a workgroup each in the residual (supply rails of large circuits)
    const int LONG = 512;
    std::vector<int> lr;
    for (int i = 0; i < s->n; ++i) if (s->rowptr[i + 1] - s->rowptr[i] > LONG) lr.push_back(i);
    h->n_long_rows = (int)lr.size();
    if (!lr.empty()) CREATE_TRY(dev_upload(&h->d_long_rows, lr.data(), lr.size()));
  }
  CREATE_TRY(dev_upload(&h->d_wave, s->wave_data, (size_t)s->n_wave_data));
  CREATE_TRY(dev_upload(&h->d_limit_init, s->limit_init, (size_t)s->n_limits));
  for (int bi = 0; bi < s->n_blocks; ++bi) {
    const CadnipDeviceBlock& sb = s->blocks[bi];
    DeviceBlock b;
    b.type = sb.type; b.count = sb.count; b.n_nodes = sb.n_nodes; b.n_ipar = sb.n_ipar; b.n_par = sb.n_par;
    b.g_base = sb.g_base; b.c_base = sb.c_base; b.b_base = sb.b_base; b.n_g = sb.n_g; b.n_c = sb.n_c; b.n_b = sb.n_b;
    if (b.type < 0 || b.type >= CADNIP_DEV_NTYPES || b.count < 0 || b.n_nodes < 0 || b.n_ipar < 0 || b.n_par < 0 || b.n_g < 0 || b.n_c < 0 || b.n_b < 0 ||
        b.g_base < 0 || b.c_base < 0 || b.b_base < 0) { cadnip_destroy(h); return CADNIP_BADARG; }
    if ((b.n_nodes * b.count > 0 && !sb.nodes) || (b.n_ipar * b.count > 0 && !sb.ipar)) { cadnip_destroy(h); return CADNIP_BADARG; }
    if (b.g_base + b.n_g * b.count > s->ns_g || b.c_base + b.n_c * b.count > s->ns_c || b.b_base + b.n_b * b.count > s->ns_b) { cadnip_destroy(h); return CADNIP_BADARG; }
    for (int k = 0; k < b.n_nodes * b.count; ++k) if (sb.nodes[k] < -1 || sb.nodes[k] >= s->n) { cadnip_destroy(h); return CADNIP_BADARG; }
    if ((b.type == CADNIP_DEV_VSOURCE || b.type == CADNIP_DEV_ISOURCE)) {
      if (b.n_ipar < 3) { cadnip_destroy(h); return CADNIP_BADARG; }
      for (int d = 0; d < b.count; ++d) {
        int kind = sb.ipar[d], off = sb.ipar[b.count + d], len = sb.ipar[2 * b.count + d];
        int need = kind == CADNIP_WAVE_PWL ? 2 * len : (kind == CADNIP_WAVE_DC ? 0 : len);
        if (kind < 0 || kind > 3 || off < 0 || off + need > s->n_wave_data || (kind == CADNIP_WAVE_PWL && len < 1)) { cadnip_destroy(h); return CADNIP_BADARG; }
      }
    }
    if (b.type == CADNIP_DEV_BVSOURCE || b.type == CADNIP_DEV_BISOURCE) {
      // walk every postfix program once: operands in range, stack depth within bounds, exactly one result
      if (b.n_ipar < 2 || b.n_par < 1) { cadnip_destroy(h); return CADNIP_BADARG; }
      for (int d = 0; d < b.count; ++d) {
        int off = sb.ipar[d], len = sb.ipar[b.count + d], sp = 0;
        bool ok = off >= 0 && len > 0 && off + len <= s->n_wave_data;
        for (int i = 0; ok && i < len;) {
          int op = (int)s->wave_data[off + i++];
          if (op == CADNIP_BOP_CONST) { ok = i < len; ++i; ++sp; }
          else if (op == CADNIP_BOP_V) {
            ok = i + 1 < len;
            if (ok) { double a = s->wave_data[off + i], c = s->wave_data[off + i + 1]; ok = a >= -1 && a < s->n && c >= -1 && c < s->n; }
            i += 2; ++sp;
          } else if (op == CADNIP_BOP_TIME) ++sp;
          else if (op >= CADNIP_BOP_ADD && op <= CADNIP_BOP_MAX) { ok = sp >= 2; --sp; }
          else if (op >= CADNIP_BOP_NEG && op <= CADNIP_BOP_COS) ok = sp >= 1;
          else ok = false;
          if (sp > CADNIP_BSRC_MAX_STACK) ok = false;
        }
        if (!ok || sp != 1) { cadnip_destroy(h); return CADNIP_BADARG; }
      }
    }
    if (b.type == CADNIP_DEV_MOS1 && (b.n_par != CADNIP_MOS1_NPAR || b.n_nodes != 14)) { cadnip_destroy(h); return CADNIP_BADARG; }
    if (b.type == CADNIP_DEV_VA) {
      // one generated module per block; its shape must be the one the library was generated with (va_generated.hpp)
      bool ok = b.n_ipar >= 2 && b.count > 0;
      const int id = ok ? sb.ipar[0] : -1;
      ok = ok && id >= 0 && id < CADNIP_VA_NMODELS;
      for (int d = 0; ok && d < b.count; ++d) ok = sb.ipar[d] == id;
      if (ok) {
        const auto& sh = CADNIP_VA_SHAPES[id];
        ok = b.n_nodes == sh.n_nodes && b.n_g == sh.n_g && b.n_c == sh.n_c && b.n_b == sh.n_b && b.n_par == sh.n_par;
      }
      if (!ok) { cadnip_destroy(h); return CADNIP_BADARG; }
      b.va_model = id;
      if (id >= CADNIP_VA_NBUILTIN) { static const int tl_lanes[] = {CADNIP_VA_EXT_TL_LANES 0}, n_cache[] = {CADNIP_VA_EXT_NCACHE 0}; h->va_ext = true; b.va_tl = tl_lanes[id - CADNIP_VA_NBUILTIN]; b.n_cache = n_cache[id - CADNIP_VA_NBUILTIN]; }   // a large external model (PSP103): per-op kernels only (fused2_fits)
    }
    b.h_nodes.assign(sb.nodes, sb.nodes + (size_t)b.n_nodes * b.count);
    h->blocks.push_back(b);                      // registered first: a failing upload below is cleaned up by cadnip_destroy
    DeviceBlock& hb = h->blocks.back();
    CREATE_TRY(dev_upload(&hb.d_nodes, sb.nodes, (size_t)b.n_nodes * b.count));
    CREATE_TRY(dev_upload(&hb.d_ipar, sb.ipar, (size_t)(b.n_ipar > 0 ? b.n_ipar : 1) * b.count));
    CREATE_TRY(dev_alloc(&hb.d_par, (size_t)h->B * b.n_par * b.count));
    if (b.n_cache > 0) CREATE_TRY(dev_alloc(&hb.d_cache, (size_t)h->B * b.n_cache * b.count));
  }
  size_t B = h->B, n = h->n, nnz = h->nnz;
  CREATE_TRY(dev_alloc(&h->d_u, B * n)); CREATE_TRY(dev_alloc(&h->d_du, B * n)); CREATE_TRY(dev_alloc(&h->d_t, B)); CREATE_TRY(dev_alloc(&h->d_gamma, B));
  CREATE_TRY(dev_alloc(&h->d_G, B * nnz)); CREATE_TRY(dev_alloc(&h->d_C, B * nnz)); CREATE_TRY(dev_alloc(&h->d_b, B * n));
  CREATE_TRY(dev_alloc(&h->d_J, B * nnz)); CREATE_TRY(dev_alloc(&h->d_resid, B * n)); CREATE_TRY(dev_alloc(&h->d_delta, B * n));
  CREATE_TRY(dev_alloc(&h->d_limit_w, B * n)); CREATE_TRY(dev_alloc(&h->d_tmp, B * n));
  CREATE_TRY(dev_alloc(&h->d_flags, B)); CREATE_TRY(dev_alloc(&h->d_active, B)); CREATE_TRY(dev_alloc(&h->d_nonfinite, B));
  CREATE_TRY(dev_alloc(&h->d_gshunt, B)); CREATE_TRY(dev_alloc(&h->d_srcfact, B)); CREATE_TRY(dev_alloc(&h->d_cold, B));
  CREATE_TRY(upload_homotopy(h, nullptr, nullptr));
  CREATE_TRY(build_stamp_plan(h, s));
  std::vector<int> ones(B, 1);
  CREATE_HIP_TRY(hipMemcpy(h->d_active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
  CREATE_HIP_TRY(hipMemcpy(h->d_cold, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
  *out = h;
  return CADNIP_OK;
}

void cadnip_driver_free(CadnipHandle* h);   // driver.hip

void cadnip_destroy(CadnipHandle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  cadnip_driver_free(h);
  void* ptrs[] = {h->d_rowptr, h->d_colidx, h->d_to_ref, h->d_g_ptr, h->d_g_slots, h->d_c_ptr, h->d_c_slots, h->d_b_ptr, h->d_b_slots,
                  h->d_diag_flag, h->d_prep, h->d_long_rows, h->d_wave, h->d_limit_init, h->d_u, h->d_du, h->d_t, h->d_gamma, h->d_G, h->d_C, h->d_b, h->d_J,
                  h->d_resid, h->d_delta, h->d_limit_w, h->d_LU, h->d_tmp, h->d_flags, h->d_active, h->d_nonfinite, h->d_gshunt, h->d_srcfact, h->d_cold, h->d_load_src, h->d_load_dst, h->d_ent_pos,
                  h->d_ent_diag, h->d_ent_ptr, h->d_term_a, h->d_term_b, h->d_lev_ptr, h->d_lu_rowptr, h->d_lu_col, h->d_lu_diag, h->d_rperm,
                  h->d_cperm, h->d_fwd_rows, h->d_fwd_lev_ptr, h->d_bwd_rows, h->d_bwd_lev_ptr};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->d_f2tab) (void)hipFree(h->d_f2tab);
  for (int k = 0; k < 2; ++k) if (h->d_team_desc[k]) (void)hipFree(h->d_team_desc[k]);
  if (h->d_steps1) (void)hipFree(h->d_steps1);
  if (h->d_steps4) (void)hipFree(h->d_steps4);
  for (auto& g : h->step_graph) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  for (auto& b : h->blocks) {
    void* bp[] = {b.d_nodes, b.d_ipar, b.d_par, b.d_sp_tptr, b.d_sp_info, b.d_sp_rec, b.d_cache, b.d_sp_rowoff,
                  b.sp_gen.tptr, b.sp_gen.info, b.sp_gen.rec, b.sp_gen.rowoff, b.sp_plain.tptr, b.sp_plain.info, b.sp_plain.rec, b.sp_plain.rowoff};
    for (void* p : bp) if (p) (void)hipFree(p);
  }
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->d_f2queue) (void)hipFree(h->d_f2queue);
  if (h->d_f2_lufac) (void)hipFree(h->d_f2_lufac);
  if (h->d_f2blk) (void)hipFree(h->d_f2blk);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int cadnip_set_params(CadnipHandle* h, int32_t block, const double* par_host) {
  if (h) ++h->graph_epoch;
  if (!h || block < 0 || block >= (int)h->blocks.size() || !par_host) return CADNIP_BADARG;
  auto& b = h->blocks[block];
  HIP_TRY(hipMemcpy(b.d_par, par_host, (size_t)h->B * b.n_par * b.count * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (b.type == CADNIP_DEV_MOS1) {
    // the fused kernel's two-lanes-per-MOSFET stamp (devices.hpp: stamp_mos1_pair) applies when no instance has series
    // resistances (gd, gs) or a Meyer gate charge (OxideCap): parameter rows 30, 31 and 8 of the derived card
    bool plain = true;
    for (size_t i = 0; i < (size_t)h->B && plain; ++i) {
      const double* p = par_host + i * b.n_par * b.count;
      for (int d = 0; d < b.count; ++d)
        if (p[(size_t)CADNIP_MOS1_PAR_GD * b.count + d] != 0.0 || p[(size_t)CADNIP_MOS1_PAR_GS * b.count + d] != 0.0 ||
            p[(size_t)CADNIP_MOS1_PAR_OXCAP * b.count + d] != 0.0) { plain = false; break; }
    }
    b.mos1_plain = plain;
  }
  h->f2_blk_dirty = true;
  if (b.n_cache > 0) {       // generated external model: its bias-independent statements run now, once for this parameter set
    int rc = cadnip::launch_va_setup(h, b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return CADNIP_OK;
}

int cadnip_set_spec(CadnipHandle* h, const CadnipSpec* spec) {
  if (!h || !spec || spec->mode < 0 || spec->mode > 2) return CADNIP_BADARG;
  h->spec = *spec;
  ++h->graph_epoch;
  return upload_homotopy(h, nullptr, nullptr);
}

int cadnip_set_initjct(CadnipHandle* h, int32_t on) { if (!h) return CADNIP_BADARG; h->initjct = on ? 1 : 0; ++h->graph_epoch; return CADNIP_OK; }

// ---- host-pointer transfers of the callback entry points.  Small ones are staged through mapped pinned memory and moved by a copy KERNEL
// on the stream (kernels of a stream stay ordered among themselves; asynchronous copies were not ordered with them under a counter pass -- see
// cadnip_create); what does not fit the staging area is a blocking copy (uploads: the stream holds at most earlier staged uploads, to other buffers; downloads:
// after a stream synchronisation).  An entry point calls stage_begin first and stage_finish last (which synchronises and hands out the downloads).
static void stage_begin(CadnipHandle* h) { h->stage_off = 0; h->stage_pending.clear(); }
static char* stage_take(CadnipHandle* h, size_t bytes) {
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (!h->h_stage || h->stage_off + need > h->stage_bytes) return nullptr;
  char* p = h->h_stage + h->stage_off;
  h->stage_off += need;
  return p;
}
static int stage_up(CadnipHandle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (char* s = stage_take(h, bytes)) {
    memcpy(s, src_host, bytes);
    return dev_copy_async(h, dst_dev, h->d_stage + (s - h->h_stage), bytes, false);
  }
  HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return CADNIP_OK;
}
static int stage_down(CadnipHandle* h, void* dst_host, const void* src_dev, size_t bytes) {
  if (char* s = stage_take(h, bytes)) {
    h->stage_pending.push_back({dst_host, s, bytes});
    return dev_copy_async(h, h->d_stage + (s - h->h_stage), src_dev, bytes, true);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return CADNIP_OK;
}
static int stage_finish(CadnipHandle* h) {
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (auto& p : h->stage_pending) memcpy(p.dst, p.src, p.bytes);
  h->stage_pending.clear();
  h->stage_off = 0;                 // nothing of the staging area is in flight any more
  return CADNIP_OK;
}

static int upload_state(CadnipHandle* h, const double* u_host, const double* t_host) {
  size_t B = h->B, n = h->n;
  if (u_host) TRY(stage_up(h, h->d_u, u_host, B * n * sizeof(double)));
  if (t_host) TRY(stage_up(h, h->d_t, t_host, B * sizeof(double)));
  return CADNIP_OK;
}

// CADNIP_NONFINITE when the last restamp produced a NaN / Inf in G, C or b of any instance (the assemble kernels raise
// d_nonfinite; the Julia shim maps the status to DomainError, which _dc_solve_with_fallbacks catches, solve.jl:887-897)
static int check_nonfinite(CadnipHandle* h) {
  std::vector<int> nf((size_t)h->B);
  TRY(stage_down(h, nf.data(), h->d_nonfinite, nf.size() * sizeof(int)));
  TRY(stage_finish(h));
  for (int f : nf) if (f) return CADNIP_NONFINITE;
  return CADNIP_OK;
}

int cadnip_rebuild(CadnipHandle* h, const double* u_host, const double* t_host) {
  if (!h) return CADNIP_BADARG;
  stage_begin(h);
  TRY(upload_state(h, u_host, t_host));
  TRY_RC(dev_zero_async(h, h->d_nonfinite, (size_t)h->B * sizeof(int)));
  TRY(launch_rebuild(h));
  return check_nonfinite(h);
}

int cadnip_residual(CadnipHandle* h, const double* du_host, const double* u_host, double* resid_host) {
  if (!h || !du_host || !resid_host) return CADNIP_BADARG;
  size_t B = h->B, n = h->n;
  stage_begin(h);
  TRY(upload_state(h, u_host, nullptr));
  TRY(stage_up(h, h->d_du, du_host, B * n * sizeof(double)));
  TRY(launch_residual(h, h->d_du));
  TRY(stage_down(h, resid_host, h->d_resid, B * n * sizeof(double)));
  TRY(stage_finish(h));
  for (size_t k = 0; k < B * n; ++k) if (!(resid_host[k] == resid_host[k]) || resid_host[k] - resid_host[k] != 0.0) return CADNIP_NONFINITE;
  return CADNIP_OK;
}

static int readback_ref_order(CadnipHandle* h, const double* d_src, double* host_out) {
  size_t B = h->B, nnz = h->nnz;
  std::vector<double> tmp(B * nnz);
  TRY(stage_down(h, tmp.data(), d_src, B * nnz * sizeof(double)));
  TRY(stage_finish(h));
  for (size_t i = 0; i < B; ++i)
    for (size_t k = 0; k < nnz; ++k) host_out[i * nnz + h->h_to_ref[k]] = tmp[i * nnz + k];
  return CADNIP_OK;
}

int cadnip_jacobian(CadnipHandle* h, const double* gamma_host, double* J_ref_nz_host) {
  if (!h || !gamma_host) return CADNIP_BADARG;
  stage_begin(h);
  TRY(stage_up(h, h->d_gamma, gamma_host, (size_t)h->B * sizeof(double)));
  TRY(launch_jacobian(h));
  if (J_ref_nz_host) TRY(readback_ref_order(h, h->d_J, J_ref_nz_host));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CADNIP_OK;
}

// dense fast_jacobian! (precompile.jl:588-603): the CSR entries of J = G + gamma C scattered into a zeroed column-major n x n block per instance
__global__ void __launch_bounds__(256) k_dense_scatter(const double* J, const int* rowptr, const int* colidx, double* D, int B, int n, int nnz) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)B * n) return;
  const int inst = (int)(tid / n), i = (int)(tid - (long)inst * n);
  double* Di = D + (size_t)inst * n * n;
  const double* Ji = J + (size_t)inst * nnz;
  for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) Di[(size_t)colidx[p] * n + i] = Ji[p];
}
int cadnip_jacobian_dense(CadnipHandle* h, const double* gamma_host, double* J_dense_host) {
  if (!h || !gamma_host || !J_dense_host) return CADNIP_BADARG;
  const size_t words = (size_t)h->B * h->n * h->n;
  if (words > ((size_t)1 << 28)) return CADNIP_BADARG;            // (2 GiB of dense Jacobians: this entry point is for small systems)
  stage_begin(h);
  TRY(stage_up(h, h->d_gamma, gamma_host, (size_t)h->B * sizeof(double)));
  TRY(launch_jacobian(h));
  double* d_dense = nullptr;
  HIP_TRY(hipMalloc((void**)&d_dense, words * sizeof(double)));
  int rc = dev_zero_async(h, d_dense, words * sizeof(double));
  if (!rc) {
    const long total = (long)h->B * h->n;
    hipLaunchKernelGGL(k_dense_scatter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->d_J, (const int*)h->d_rowptr, (const int*)h->d_colidx,
                       d_dense, h->B, h->n, h->nnz);
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(J_dense_host, d_dense, words * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = CADNIP_HIPERROR;
  }
  (void)hipFree(d_dense);
  return rc;
}

// ODE form (src/mna/solve.jl:2241-2276): du = b - G u,  J = -G
int cadnip_ode_rhs(CadnipHandle* h, const double* u_host, const double* t_host, double* du_host) {
  if (!h || !du_host) return CADNIP_BADARG;
  stage_begin(h);          // (an earlier entry point that failed between stage_down and stage_finish must not leave its downloads pending)
  size_t B = h->B, n = h->n;
  if (u_host) { TRY(upload_state(h, u_host, t_host)); TRY(launch_rebuild(h)); }
  TRY_RC(dev_zero_async(h, h->d_du, B * n * sizeof(double)));
  TRY(launch_residual(h, h->d_du));                       // G u - b
  TRY(launch_negate(h, h->d_resid, (long)(B * n)));
  HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(du_host, h->d_resid, B * n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t k = 0; k < B * n; ++k) if (du_host[k] - du_host[k] != 0.0) return CADNIP_NONFINITE;
  return CADNIP_OK;
}

int cadnip_ode_jacobian(CadnipHandle* h, const double* u_host, const double* t_host, double* J_ref_nz_host) {
  if (!h || !J_ref_nz_host) return CADNIP_BADARG;
  stage_begin(h);          // (an earlier entry point that failed between stage_down and stage_finish must not leave its downloads pending)
  TRY_RC(dev_zero_async(h, h->d_nonfinite, (size_t)h->B * sizeof(int)));
  if (u_host) { TRY(upload_state(h, u_host, t_host)); TRY(launch_rebuild(h)); }
  TRY_RC(dev_zero_async(h, h->d_gamma, (size_t)h->B * sizeof(double)));
  TRY(launch_jacobian(h));                                // G + 0*C
  TRY(launch_negate(h, h->d_J, (long)((size_t)h->B * h->nnz)));
  TRY(readback_ref_order(h, h->d_J, J_ref_nz_host));
  return check_nonfinite(h);
}

int cadnip_get_GCb(CadnipHandle* h, double* G_ref_nz, double* C_ref_nz, double* b, double* limit_w) {
  if (!h) return CADNIP_BADARG;
  stage_begin(h);          // (an earlier entry point that failed between stage_down and stage_finish must not leave its downloads pending)
  size_t B = h->B, n = h->n;
  if (G_ref_nz) TRY(readback_ref_order(h, h->d_G, G_ref_nz));
  if (C_ref_nz) TRY(readback_ref_order(h, h->d_C, C_ref_nz));
  if (b) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(b, h->d_b, B * n * sizeof(double), hipMemcpyDeviceToHost)); }
  if (limit_w) {
    std::vector<double> tmp(B * n);
    HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(tmp.data(), h->d_limit_w, B * n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipStreamSynchronize(h->stream));
    size_t L = h->n_limits, l0 = n - L;
    for (size_t i = 0; i < B; ++i) for (size_t k = 0; k < L; ++k) limit_w[i * L + k] = tmp[i * n + l0 + k];
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CADNIP_OK;
}

// Operating-point read-out (context.jl:1200-1342: terminal currents, op variables): the per-device contributions of one
// restamp at the handle's current state, [B][ns_g + ns_c + ns_b] in the slot layout of CadnipStructure (slot (k, dev) of
// a block at base + k * count + dev).  The stamping kernels keep these in LDS; here they write them out once.
int cadnip_get_contributions(CadnipHandle* h, double* slots_host) {
  if (!h || !slots_host) return CADNIP_BADARG;
  const size_t cnt = (size_t)h->B * h->ns;
  HIP_TRY(hipMalloc((void**)&h->d_dump, std::max<size_t>(cnt, 1) * sizeof(double)));
  int rc = CADNIP_OK;
  if (dev_zero_async(h, h->d_dump, cnt * sizeof(double)) != CADNIP_OK) rc = CADNIP_HIPERROR;
  if (!rc) rc = launch_rebuild(h);
  if (!rc && (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(slots_host, h->d_dump, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) rc = CADNIP_HIPERROR;
  if (hipStreamSynchronize(h->stream) != hipSuccess) rc = CADNIP_HIPERROR;
  (void)hipFree(h->d_dump);
  h->d_dump = nullptr;
  return rc;
}

int cadnip_analyze(CadnipHandle* h, int32_t sample_instance) {
  if (!h || sample_instance < 0 || sample_instance >= h->B) return CADNIP_BADARG;
  std::vector<double> vals(h->nnz);
  HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(vals.data(), h->d_J + (size_t)sample_instance * h->nnz, (size_t)h->nnz * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::string err;
  int rc = lu_analyze(h->n, h->h_rowptr, h->h_colidx, vals, 1e-3, false, h->lu, err, h->leaves.q_begin >= 0 ? &h->leaves : nullptr);
  if (rc) { fprintf(stderr, "[cadnip_hip] analyze: %s\n", err.c_str()); return rc; }
  return upload_lu(h);
}

int cadnip_analyze_values(CadnipHandle* h, const double* J_csr_host) {
  if (!h || !J_csr_host) return CADNIP_BADARG;
  std::vector<double> vals(J_csr_host, J_csr_host + h->nnz);
  std::string err;
  int rc = lu_analyze(h->n, h->h_rowptr, h->h_colidx, vals, 1e-3, true, h->lu, err, h->leaves.q_begin >= 0 ? &h->leaves : nullptr);
  if (rc) { fprintf(stderr, "[cadnip_hip] analyze: %s\n", err.c_str()); return rc; }
  if (getenv("CADNIP_LU_DEBUG")) {
    unsigned long long hsh = 1469598103934665603ull;
    for (int k = 0; k < h->n; ++k) { hsh = (hsh ^ (unsigned)h->lu.rperm[k]) * 1099511628211ull; hsh = (hsh ^ (unsigned)h->lu.cperm[k]) * 1099511628211ull; }
    fprintf(stderr, "[cadnip lu] B %d pivot order hash %016llx nnz_lu %d\n", h->B, hsh, h->lu.nnz_lu);
  }
  return upload_lu(h);
}

struct CadnipHostLU { cadnip::LUProgram p; };
static const std::vector<int>* host_lu_array(const CadnipHostLU* lu, int which) {
  const cadnip::LUProgram& P = lu->p;
  switch (which) {
    case CADNIP_LU_RPERM: return &P.rperm; case CADNIP_LU_CPERM: return &P.cperm; case CADNIP_LU_ROWPTR: return &P.lu_rowptr;
    case CADNIP_LU_COL: return &P.lu_col; case CADNIP_LU_DIAG: return &P.lu_diag; case CADNIP_LU_LOAD_SRC: return &P.load_src;
    case CADNIP_LU_LOAD_DST: return &P.load_dst; case CADNIP_LU_ENT_POS: return &P.ent_pos; case CADNIP_LU_ENT_DIAG: return &P.ent_diag;
    case CADNIP_LU_ENT_PTR: return &P.ent_ptr; case CADNIP_LU_TERM_A: return &P.term_a; case CADNIP_LU_TERM_B: return &P.term_b;
    case CADNIP_LU_LEV_PTR: return &P.lev_ptr; case CADNIP_LU_FWD_ROWS: return &P.fwd_rows; case CADNIP_LU_FWD_LEV_PTR: return &P.fwd_lev_ptr;
    case CADNIP_LU_BWD_ROWS: return &P.bwd_rows; case CADNIP_LU_BWD_LEV_PTR: return &P.bwd_lev_ptr;
  }
  return nullptr;
}
int cadnip_host_lu_analyze(int32_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, double pivot_tol, int32_t sample, CadnipHostLU** out) {
  return cadnip_host_lu_analyze_leaves(n, rowptr, colidx, vals, pivot_tol, sample, -1, -1, nullptr, out);
}
int cadnip_host_lu_analyze_leaves(int32_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, double pivot_tol, int32_t sample,
                                  int32_t q_begin, int32_t lim_begin, const uint8_t* unit_ok, CadnipHostLU** out) {
  if (n <= 0 || !rowptr || !colidx || !vals || !out) return CADNIP_BADARG;
  CadnipHostLU* lu = new CadnipHostLU();
  std::vector<int> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + rowptr[n]);
  std::vector<double> v(vals, vals + rowptr[n]);
  std::string err;
  cadnip::LULeaves lv; lv.q_begin = q_begin; lv.lim_begin = lim_begin; lv.unit_ok = unit_ok; lv.first = getenv("CADNIP_LU_LEAF_FIRST") != nullptr;
  int rc = lu_analyze(n, rp, ci, v, pivot_tol, sample != 0, lu->p, err, q_begin >= 0 ? &lv : nullptr);
  if (rc) { delete lu; return rc; }
  *out = lu;
  return CADNIP_OK;
}
int32_t cadnip_host_lu_size(const CadnipHostLU* lu, int32_t which) { auto* v = lu ? host_lu_array(lu, which) : nullptr; return v ? (int32_t)v->size() : -1; }
int cadnip_host_lu_get(const CadnipHostLU* lu, int32_t which, int32_t* dst) {
  auto* v = lu ? host_lu_array(lu, which) : nullptr;
  if (!v || !dst) return CADNIP_BADARG;
  memcpy(dst, v->data(), v->size() * sizeof(int));
  return CADNIP_OK;
}
struct CadnipHostF2 { cadnip::F2Program g; std::vector<int> meta; };
int cadnip_host_f2_build(const CadnipHostLU* lu, int32_t nc, CadnipHostF2** out) {
  if (!lu || !out || nc < 0 || nc > lu->p.n) return CADNIP_BADARG;
  CadnipHostF2* f = new CadnipHostF2();
  if (!cadnip::f2_build_program(lu->p, lu->p.n, nc, f->g)) { delete f; return CADNIP_BADARG; }
  f->meta = {f->g.nc, f->g.lu_words, f->g.dn0, f->g.n_pre, f->g.n_post};
  *out = f;
  return CADNIP_OK;
}
int32_t cadnip_host_f2_size(const CadnipHostF2* f, int32_t which) {
  if (!f) return -1;
  switch (which) {
    case CADNIP_F2_POSW: return (int32_t)f->g.posW.size(); case CADNIP_F2_LANES: return (int32_t)f->g.lanes.size();
    case CADNIP_F2_PASSES: return (int32_t)f->g.passes.size(); case CADNIP_F2_TERMS: return (int32_t)f->g.terms.size();
    case CADNIP_F2_META: return (int32_t)f->meta.size();
  }
  return -1;
}
int cadnip_host_f2_get(const CadnipHostF2* f, int32_t which, void* dst) {
  if (!f || !dst) return CADNIP_BADARG;
  switch (which) {
    case CADNIP_F2_POSW: memcpy(dst, f->g.posW.data(), f->g.posW.size() * 4); return CADNIP_OK;
    case CADNIP_F2_LANES: memcpy(dst, f->g.lanes.data(), f->g.lanes.size() * 8); return CADNIP_OK;
    case CADNIP_F2_PASSES: memcpy(dst, f->g.passes.data(), f->g.passes.size() * 8); return CADNIP_OK;
    case CADNIP_F2_TERMS: memcpy(dst, f->g.terms.data(), f->g.terms.size() * 4); return CADNIP_OK;
    case CADNIP_F2_META: memcpy(dst, f->meta.data(), f->meta.size() * 4); return CADNIP_OK;
  }
  return CADNIP_BADARG;
}
// steps of the team layout of the same program (f2_program.cpp: f2_build_team): out[0..2] = steps of the pre-core, post-core and forward-only lists,
// out[3] = descriptor words (64-bit)
int cadnip_host_f2_team_steps(const CadnipHostLU* lu, int32_t nc, int32_t nw, int32_t* out) {
  if (!lu || !out || nc < 0 || nc > lu->p.n || (nw != 1 && nw != 2 && nw != 4)) return CADNIP_BADARG;
  cadnip::F2Team T;
  if (!cadnip::f2_build_team(lu->p, lu->p.n, nc, nw, T)) return CADNIP_BADARG;
  out[0] = T.n_steps[0]; out[1] = T.n_steps[1]; out[2] = T.n_steps[2]; out[3] = (int32_t)T.desc.size();
  return CADNIP_OK;
}
int cadnip_host_f2_steps(const CadnipHostLU* lu, int32_t nc, int32_t nw, int32_t* out, uint64_t* dst) {
  // nw = 1, 2, 4: the 8-byte one-term team layouts for 2 / 4 waves, the 16-byte three-term layout for one wave; nw = 12, 14: the three-term layout for 2 / 4 waves
  const bool three = nw == 1 || nw == 12 || nw == 14;
  if (!lu || !out || nc < 0 || nc > lu->p.n || (!three && nw != 2 && nw != 4)) return CADNIP_BADARG;
  cadnip::F2Team T;
  if (!(three ? cadnip::f2_build_steps(lu->p, lu->p.n, nc, nw == 1 ? 1 : nw - 10, T) : cadnip::f2_build_team(lu->p, lu->p.n, nc, nw, T))) return CADNIP_BADARG;
  out[0] = T.n_steps[0]; out[1] = T.n_steps[1]; out[2] = T.n_steps[2]; out[3] = (int32_t)T.desc.size();
  if (dst) memcpy(dst, T.desc.data(), T.desc.size() * sizeof(unsigned long long));
  return CADNIP_OK;
}
void cadnip_host_f2_free(CadnipHostF2* f) { delete f; }
int32_t cadnip_host_lu_blocks(const CadnipHostLU* lu) { return lu ? lu->p.n_blocks : -1; }
void cadnip_host_lu_free(CadnipHostLU* lu) { delete lu; }

// Replay of a fixed launch sequence as an instantiated HIP graph: the first call in a configuration (`key`) enqueues plainly (lazy set-up
// inside the launchers runs there), the second captures and instantiates, later ones replay; profiling and CADNIP_NO_GRAPH=1 enqueue plainly.
static int run_graphed(CadnipHandle* h, CadnipHandle::StepGraph& g, unsigned long long key, const std::function<int()>& enqueue) {
  if (h->prof_on || getenv("CADNIP_NO_GRAPH")) return enqueue();
  if (g.exec && g.epoch == key) { HIP_TRY(hipGraphLaunch(g.exec, h->stream)); return CADNIP_OK; }
  if (g.warmed != key) { TRY_RC(enqueue()); g.warmed = key; return CADNIP_OK; }
  if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  const int rc = enqueue();
  const hipError_t ec = hipStreamEndCapture(h->stream, &graph);
  if (rc || ec != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); if (rc) return rc; cadnip::set_last_error("hipStreamEndCapture", ec); return CADNIP_HIPERROR; }
  const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) { g.exec = nullptr; cadnip::set_last_error("hipGraphInstantiate", ei); return CADNIP_HIPERROR; }
  g.epoch = key;
  HIP_TRY(hipGraphLaunch(g.exec, h->stream));
  return CADNIP_OK;
}

// One Newton iteration of the DAE form in ONE call: what a host integrator that keeps the Newton loop to itself (IDA through the Julia
// shim: residual callback, Jacobian callback, KLU refactor / solve -- src/mna/precompile.jl:546-585, src/mna/solve.jl:2138-2160) otherwise
// does with five entry points and five stream synchronisations.  The SAME kernels in the same order as cadnip_rebuild -> cadnip_residual ->
// [cadnip_jacobian -> cadnip_factor] -> cadnip_solve, so the results are the same doubles; all transfers are staged (one upload kernel,
// one download kernel), and the launch sequence is an instantiated HIP graph that is replayed while nothing it depends on changes.
//   refresh != 0: J = G + gamma C is formed and refactored (factors stay in HBM); 0: the factors of the last refreshing call solve.
//   delta [B][n] = J^-1 resid with resid = C du + G u - b;  resid_norm [B] (optional) = ||resid||_2;  resid [B][n] (optional).
int cadnip_newton_step(CadnipHandle* h, const double* u_host, const double* du_host, const double* gamma_host, const double* t_host, int32_t refresh,
                       double* delta_host, double* resid_norm_host, double* resid_host) {
  if (!h || !u_host || !du_host || !delta_host || (refresh && !gamma_host)) return CADNIP_BADARG;
  if (!h->analyzed) return CADNIP_NOTREADY;
  const size_t B = h->B, n = h->n, vec = B * n * sizeof(double);
  stage_begin(h);
  char *s_u = stage_take(h, vec), *s_du = stage_take(h, vec), *s_g = stage_take(h, B * 8), *s_t = stage_take(h, B * 8), *s_d = stage_take(h, vec),
       *s_r = stage_take(h, vec), *s_nrm = stage_take(h, B * 8), *s_nf = stage_take(h, B * 4), *s_fl = stage_take(h, B * 4);
  if (!s_fl) {
    // too large for the staging area: the five entry points (blocking copies), same results
    TRY(cadnip_rebuild(h, u_host, t_host));
    std::vector<double> r(B * n);
    TRY(cadnip_residual(h, du_host, nullptr, r.data()));
    if (refresh) { TRY(cadnip_jacobian(h, gamma_host, nullptr)); TRY(cadnip_factor(h)); }
    TRY(cadnip_solve(h, r.data(), delta_host));
    if (resid_host) memcpy(resid_host, r.data(), vec);
    if (resid_norm_host) for (size_t i = 0; i < B; ++i) { double a = 0; for (size_t k = 0; k < n; ++k) a += r[i * n + k] * r[i * n + k]; resid_norm_host[i] = sqrt(a); }
    return CADNIP_OK;
  }
  auto dv = [&](char* p) { return h->d_stage + (p - h->h_stage); };
  memcpy(s_u, u_host, vec); memcpy(s_du, du_host, vec);
  if (gamma_host) memcpy(s_g, gamma_host, B * 8);
  if (t_host) memcpy(s_t, t_host, B * 8);
  const bool have_g = gamma_host != nullptr, have_t = t_host != nullptr;
  auto enqueue = [&]() -> int {
    MultiCopy up;
    up.add(h->d_u, dv(s_u), vec); up.add(h->d_du, dv(s_du), vec);
    if (have_g) up.add(h->d_gamma, dv(s_g), B * 8);
    if (have_t) up.add(h->d_t, dv(s_t), B * 8);
    up.add(h->d_nonfinite, nullptr, B * 4); up.add(h->d_flags, nullptr, B * 4);
    TRY_RC(dev_multi_async(h, up, false));
    TRY_RC(launch_rebuild(h));
    TRY_RC(launch_residual(h, h->d_du));
    if (refresh) { TRY_RC(launch_jacobian(h)); TRY_RC(launch_factor(h, false)); }
    TRY_RC(launch_solve(h, h->d_resid, h->d_delta));
    TRY_RC(launch_norm2(h, h->d_resid, h->d_tmp));
    MultiCopy down;
    down.add(dv(s_d), h->d_delta, vec); down.add(dv(s_r), h->d_resid, vec); down.add(dv(s_nrm), h->d_tmp, B * 8);
    down.add(dv(s_nf), h->d_nonfinite, B * 4); down.add(dv(s_fl), h->d_flags, B * 4);
    TRY_RC(dev_multi_async(h, down, true));
    return CADNIP_OK;
  };
  // graph variants: refresh x (gamma given) x (t given) change the launch sequence: keyed by the first only, the other two recorded in the epoch
  TRY(run_graphed(h, h->step_graph[refresh ? 1 : 0], h->graph_epoch * 4 + (have_g ? 2 : 0) + (have_t ? 1 : 0), enqueue));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage_off = 0;
  memcpy(delta_host, s_d, vec);
  if (resid_host) memcpy(resid_host, s_r, vec);
  if (resid_norm_host) memcpy(resid_norm_host, s_nrm, B * 8);
  const int* nf = (const int*)s_nf; const int* fl = (const int*)s_fl;
  for (size_t i = 0; i < B; ++i) if (nf[i]) return CADNIP_NONFINITE;
  for (size_t i = 0; i < B; ++i) if (fl[i] & 1) return CADNIP_SINGULAR;
  for (size_t k = 0; k < B * n; ++k) if (delta_host[k] - delta_host[k] != 0.0) return CADNIP_NONFINITE;
  return CADNIP_OK;
}

// The same iteration in the fused team kernel (csrc/fused_team_kernel.hpp, STEP mode): stamping, residual, refactorisation (or the kept
// factors) and the solve of every instance in ONE kernel between the staged upload and download.  Same mathematics, another summation order
// (the devices add into the LDS-resident work array): the results agree with cadnip_newton_step to rounding, not bit for bit.  For circuits
// the team kernel runs (linear elements, sources, plain sp_mos1; tables within LDS); CADNIP_BADARG otherwise -- call cadnip_newton_step then.
int cadnip_newton_step_fused(CadnipHandle* h, const double* u_host, const double* du_host, const double* gamma_host, const double* t_host, int32_t refresh,
                             double* delta_host, double* resid_norm_host, double* resid_host) {
  if (!h || !u_host || !du_host || !delta_host || (refresh && !gamma_host)) return CADNIP_BADARG;
  if (!h->analyzed) return CADNIP_NOTREADY;
  if (h->homotopy || h->spec.gshunt != 0.0 || h->spec.srcFact < 1.0) return CADNIP_BADARG;
  const size_t B = h->B, n = h->n, vec = B * n * sizeof(double);
  stage_begin(h);
  char *s_u = stage_take(h, vec), *s_du = stage_take(h, vec), *s_g = stage_take(h, B * 8), *s_t = stage_take(h, B * 8), *s_d = stage_take(h, vec),
       *s_r = stage_take(h, vec), *s_nrm = stage_take(h, B * 8), *s_fl = stage_take(h, B * 4);
  if (!s_fl) return CADNIP_BADARG;
  auto dv = [&](char* p) { return h->d_stage + (p - h->h_stage); };
  // ONE launch: the kernel reads u, du, gamma, t from the mapped pinned staging area (a few KB across the bus, once) and writes the Newton
  // step, the residual, its norm and the failure flags back there; gamma / t stay what they were on the device when not given
  memcpy(s_u, u_host, vec); memcpy(s_du, du_host, vec);
  if (gamma_host) memcpy(s_g, gamma_host, B * 8);
  if (t_host) memcpy(s_t, t_host, B * 8);
  memset(s_fl, 0, B * 4);
  FusedStepIO io;
  io.u = (const double*)dv(s_u); io.du = (const double*)dv(s_du);
  io.gamma = gamma_host ? (const double*)dv(s_g) : h->d_gamma; io.t = t_host ? (const double*)dv(s_t) : h->d_t;
  io.gamma_keep = gamma_host ? h->d_gamma : nullptr; io.t_keep = t_host ? h->d_t : nullptr;
  io.delta = (double*)dv(s_d); io.resid = resid_host ? (double*)dv(s_r) : nullptr; io.norm = (double*)dv(s_nrm); io.flags = (int*)dv(s_fl);
  { const int rc = launch_fused_step(h, refresh ? 1 : 0, io); if (rc) { (void)hipStreamSynchronize(h->stream); h->stage_off = 0; return rc; } }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage_off = 0;
  memcpy(delta_host, s_d, vec);
  if (resid_host) memcpy(resid_host, s_r, vec);
  if (resid_norm_host) memcpy(resid_norm_host, s_nrm, B * 8);
  const int* fl = (const int*)s_fl;
  for (size_t i = 0; i < B; ++i) if (fl[i] & 1) return CADNIP_SINGULAR;
  for (size_t k = 0; k < B * n; ++k) if (delta_host[k] - delta_host[k] != 0.0) return CADNIP_NONFINITE;
  return CADNIP_OK;
}

int cadnip_factor(CadnipHandle* h) {
  if (!h) return CADNIP_BADARG;
  TRY_RC(dev_zero_async(h, h->d_flags, (size_t)h->B * sizeof(int)));
  TRY(launch_factor(h, false));
  std::vector<int> fl(h->B);
  stage_begin(h);
  TRY(stage_down(h, fl.data(), h->d_flags, (size_t)h->B * sizeof(int)));
  TRY(stage_finish(h));
  for (int f : fl) if (f & 1) return CADNIP_SINGULAR;
  return CADNIP_OK;
}

int cadnip_solve(CadnipHandle* h, const double* rhs_host, double* x_host) {
  if (!h || !rhs_host || !x_host) return CADNIP_BADARG;
  size_t B = h->B, n = h->n;
  stage_begin(h);
  TRY(stage_up(h, h->d_resid, rhs_host, B * n * sizeof(double)));
  TRY(launch_solve(h, h->d_resid, h->d_delta));
  TRY(stage_down(h, x_host, h->d_delta, B * n * sizeof(double)));
  return stage_finish(h);
}

int cadnip_lu_stats(CadnipHandle* h, int32_t* nnz_lu, int32_t* n_terms, int32_t* n_levels, int32_t* n_fwd, int32_t* n_bwd) {
  if (!h || !h->analyzed) return CADNIP_NOTREADY;
  if (nnz_lu) *nnz_lu = h->lu.nnz_lu;
  if (n_terms) *n_terms = (int)h->lu.term_a.size();
  if (n_levels) *n_levels = (int)h->lu.lev_ptr.size() - 1;
  if (n_fwd) *n_fwd = (int)h->lu.fwd_lev_ptr.size() - 1;
  if (n_bwd) *n_bwd = (int)h->lu.bwd_lev_ptr.size() - 1;
  return CADNIP_OK;
}

void* cadnip_dev_ptr(CadnipHandle* h, int32_t which) {
  if (!h) return nullptr;
  switch (which) {
    case CADNIP_BUF_U: return h->d_u; case CADNIP_BUF_G: return h->d_G; case CADNIP_BUF_C: return h->d_C; case CADNIP_BUF_B: return h->d_b;
    case CADNIP_BUF_J: return h->d_J; case CADNIP_BUF_RESID: return h->d_resid; case CADNIP_BUF_SLOTS: return nullptr;   /* the slot buffer is gone: contributions are staged in LDS (stamp_csr.hip) */ case CADNIP_BUF_LU: return h->d_LU;
    case CADNIP_BUF_FLAGS: return h->d_flags;
  }
  return nullptr;
}
void* cadnip_stream(CadnipHandle* h) { return h ? (void*)h->stream : nullptr; }

int cadnip_set_u(CadnipHandle* h, const double* u_host) {
  if (!h || !u_host) return CADNIP_BADARG;
  HIP_TRY(hipMemcpy(h->d_u, u_host, (size_t)h->B * h->n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CADNIP_OK;
}
int cadnip_get_u(CadnipHandle* h, double* u_host) {
  if (!h || !u_host) return CADNIP_BADARG;
  HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(u_host, h->d_u, (size_t)h->B * h->n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CADNIP_OK;
}
int cadnip_get_flags(CadnipHandle* h, int32_t* flags_host) {
  if (!h || !flags_host) return CADNIP_BADARG;
  HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(flags_host, h->d_flags, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CADNIP_OK;
}
int cadnip_debug_copy(CadnipHandle* h, int64_t n_doubles, int32_t reps) {
  if (!h || n_doubles <= 0 || reps <= 0) return CADNIP_BADARG;
  return launch_calib_copy(h, (long)n_doubles, reps);
}
// `reps` back-to-back launches of the stamping kernel of one device block (block < 0: the whole restamp, every block) at the
// handle's current state, timed with one pair of HIP events on the handle's stream: *ms_total = elapsed milliseconds
int cadnip_debug_stamp_time(CadnipHandle* h, int32_t block, int32_t reps, double* ms_total) {
  if (!h || reps <= 0 || !ms_total) return CADNIP_BADARG;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  for (int r = 0; r < reps; ++r) TRY(block < 0 ? launch_rebuild(h) : launch_stamp_block(h, block));
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_total = ms;
  // a single block launched on its own adds to the words an earlier kernel of a restamp wrote (read-modify-write / atomic targets):
  // leave G, C, b as one full restamp leaves them
  if (block >= 0) { TRY(launch_rebuild(h)); HIP_TRY(hipStreamSynchronize(h->stream)); }
  return CADNIP_OK;
}
// Measurement: the cost of ONE Newton iteration inside the team kernel, phase by phase.  The STEP-mode kernel repeats the iteration at the handle's
// resident state (d_u, d_du, d_gamma, d_t) `reps` times in one launch; `skip` leaves phases out (1 stamping, 2 adding the waves' private
// sums, 4 the linear solve's steps, 8 the dense core), so that differences of two calls are a phase's cost.  ms_total: the launch (HIP events).
int cadnip_debug_step_time(CadnipHandle* h, int32_t refresh, int32_t reps, int32_t skip, double* ms_total) {
  if (!h || reps <= 0 || !ms_total) return CADNIP_BADARG;
  FusedStepIO io;
  io.u = h->d_u; io.du = h->d_du; io.gamma = h->d_gamma; io.t = h->d_t; io.gamma_keep = nullptr; io.t_keep = nullptr;
  io.delta = h->d_delta; io.resid = nullptr; io.norm = h->d_tmp; io.flags = h->d_flags; io.reps = reps; io.skip = skip;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  TRY(launch_fused_step(h, refresh ? 1 : 0, io));
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_total = ms;
  return CADNIP_OK;
}
int cadnip_sync(CadnipHandle* h) { if (!h) return CADNIP_BADARG; HIP_TRY(hipStreamSynchronize(h->stream)); return CADNIP_OK; }

int cadnip_profile_enable(CadnipHandle* h, int32_t on) {
  if (!h) return CADNIP_BADARG;
  h->prof_on = on != 0;
  h->prof.clear();
  return CADNIP_OK;
}
int cadnip_profile_read(CadnipHandle* h, int32_t max_entries, const char** names, double* ms, int64_t* calls) {
  if (!h) return 0;
  int k = 0;
  for (auto& e : h->prof) { if (k >= max_entries) break; names[k] = e.name; ms[k] = e.ms; calls[k] = e.calls; ++k; }
  return k;
}

}  // extern "C"

namespace cadnip {
int upload_homotopy(CadnipHandle* h, const double* gshunt, const double* srcfact) {
  ++h->graph_epoch;
  const size_t B = (size_t)h->B;
  std::vector<double> g(B, h->spec.gshunt), sf(B, h->spec.srcFact);
  if (gshunt) g.assign(gshunt, gshunt + B);
  if (srcfact) sf.assign(srcfact, srcfact + B);
  bool any = false;
  for (size_t i = 0; i < B; ++i) if (g[i] != 0.0 || sf[i] < 1.0) any = true;
  HIP_TRY(hipMemcpy(h->d_gshunt, g.data(), B * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_srcfact, sf.data(), B * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->stream));   // the host vectors go out of scope
  h->homotopy = any;
  return CADNIP_OK;
}
int upload_lu(CadnipHandle* h) {
  ++h->graph_epoch;
  LUProgram& P = h->lu;
  int** olds[] = {&h->d_load_src, &h->d_load_dst, &h->d_ent_pos, &h->d_ent_diag, &h->d_ent_ptr, &h->d_term_a, &h->d_term_b, &h->d_lev_ptr,
                  &h->d_lu_rowptr, &h->d_lu_col, &h->d_lu_diag, &h->d_rperm, &h->d_cperm, &h->d_fwd_rows, &h->d_fwd_lev_ptr, &h->d_bwd_rows, &h->d_bwd_lev_ptr};
  for (int** p : olds) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->d_LU) { (void)hipFree(h->d_LU); h->d_LU = nullptr; }
  // load_dst indexed by csr position (load_src is 0..nnz-1 in order)
  std::vector<int> dst(h->nnz, 0);
  for (size_t k = 0; k < P.load_src.size(); ++k) dst[P.load_src[k]] = P.load_dst[k];
  TRY(dev_upload(&h->d_load_dst, dst));
  TRY(dev_upload(&h->d_ent_pos, P.ent_pos)); TRY(dev_upload(&h->d_ent_diag, P.ent_diag)); TRY(dev_upload(&h->d_ent_ptr, P.ent_ptr));
  TRY(dev_upload(&h->d_term_a, P.term_a)); TRY(dev_upload(&h->d_term_b, P.term_b)); TRY(dev_upload(&h->d_lev_ptr, P.lev_ptr));
  TRY(dev_upload(&h->d_lu_rowptr, P.lu_rowptr)); TRY(dev_upload(&h->d_lu_col, P.lu_col)); TRY(dev_upload(&h->d_lu_diag, P.lu_diag));
  TRY(dev_upload(&h->d_rperm, P.rperm)); TRY(dev_upload(&h->d_cperm, P.cperm));
  TRY(dev_upload(&h->d_fwd_rows, P.fwd_rows)); TRY(dev_upload(&h->d_fwd_lev_ptr, P.fwd_lev_ptr));
  TRY(dev_upload(&h->d_bwd_rows, P.bwd_rows)); TRY(dev_upload(&h->d_bwd_lev_ptr, P.bwd_lev_ptr));
  TRY(dev_alloc(&h->d_LU, (size_t)h->B * P.nnz_lu));
  h->analyzed = true;
  h->fused2_dirty = true;
  return CADNIP_OK;
}
}  // namespace cadnip
