// driver.hip -- host drivers: DC operating point (PCNR Newton + fallbacks) and the transient
// loop (variable-step BDF1/BDF2 + Newton), standing in for the integrator that calls the hot
// path in the reference (Sundials IDA via SciML, /root/reference/src/sweeps.jl:588-665; DC:
// _dc_solve_with_fallbacks, /root/reference/src/mna/solve.jl:599-929).
//
// The loops run on the host and launch the hot-path kernels once per Newton iteration for
// the whole batch.  Each sweep instance walks its own step sequence (own t, h, order,
// Newton count); its convergence test, error test, step-size choice, history rotation,
// predictor and output interpolation are evaluated by one wave per instance in
// k_tran_update (per-op path) or inside the fused kernel (fused2.hip), so desynchronised instances never round-trip
// over PCIe.  The host reads one counter (instances still running) every few launches.
//
// Integration method (identical in oracle/cpu_port.cpp, which is what the 1e-9 parity bar is
// defined against -- SURVEY.md section 7 "hard parts"):
//   * BDF1 with constant predictor on the first step after a (re)start, no error test;
//     BDF1 + linear predictor on the second; variable-step BDF2 + quadratic predictor after.
//   * local error estimate from the predictor-corrector difference (Newton divided
//     difference):  order 1: h/(h+h1) (u-up);  order 2: (1+w)h/((1+2w)(h+h1+h2)) (u-up), w=h/h1.
//   * weighted RMS norms with w_i = 1/(atol_i + rtol*|u_i|); step factor 0.9*err^(-1/(k+1))
//     clipped to [0.2, 2]; Newton failure -> h/4; restart at order 1 on every breakpoint.
#include <hip/hip_runtime.h>
#include <chrono>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include "internal.hpp"

using namespace cadnip;

struct Driver {
  int n_save = 0, n_obs = 0, n_break = 0;
  double *t = nullptr, *h = nullptr, *hprev = nullptr, *hpp = nullptr;
  int *nhist = nullptr, *order = nullptr, *k = nullptr, *status = nullptr, *bp_idx = nullptr, *save_idx = nullptr, *dcstate = nullptr, *action = nullptr;
  long long* cnt = nullptr;   // [B][4]
  double *mn_a0f = nullptr, *mn_ss = nullptr, *mn_dnp = nullptr; int* mn_flags = nullptr;   // Newton mode 1 (tran_ctrl.hpp)
  double *u0 = nullptr, *u1 = nullptr, *u2 = nullptr, *up = nullptr, *beta = nullptr;
  double *u3 = nullptr, *hp3 = nullptr;   // max_order = 3 (allocated when first asked for)
  double *atol = nullptr, *emask = nullptr, *breaks = nullptr, *save_t = nullptr, *out = nullptr;
  int* obs = nullptr;
  int* nactive = nullptr;
  int* part = nullptr;        // [B] 1 = the instance takes part in the DC Newton run being started (k_dc_init)
  size_t out_cap = 0, brk_cap = 0, save_cap = 0, obs_cap = 0;
  // what the DC fallback chain did, one entry per (instance, Newton run): cadnip_dc_log_*
  struct DCLogEntry { int inst, stage; double value; int ok; long long iters; };
  std::vector<DCLogEntry> dc_log;
};

namespace {
#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
template <class T> int dalloc(T** p, size_t c) { if (*p) return CADNIP_OK; HIP_TRY(hipMalloc((void**)p, (c ? c : 1) * sizeof(T))); HIP_TRY(hipMemset(*p, 0, (c ? c : 1) * sizeof(T))); return CADNIP_OK; }
template <class T> int drealloc(T** p, size_t* cap, size_t c) { if (*p && *cap >= c) return CADNIP_OK; if (*p) (void)hipFree(*p); *p = nullptr; *cap = c; return dalloc(p, c); }

int ensure_driver(CadnipHandle* h) {
  if (!h->drv) h->drv = new Driver();
  Driver* d = h->drv;
  size_t B = h->B, n = h->n;
  TRY(dalloc(&d->t, B)); TRY(dalloc(&d->h, B)); TRY(dalloc(&d->hprev, B)); TRY(dalloc(&d->hpp, B));
  TRY(dalloc(&d->nhist, B)); TRY(dalloc(&d->order, B)); TRY(dalloc(&d->k, B)); TRY(dalloc(&d->status, B));
  TRY(dalloc(&d->bp_idx, B)); TRY(dalloc(&d->save_idx, B)); TRY(dalloc(&d->dcstate, B)); TRY(dalloc(&d->action, B));
  TRY(dalloc(&d->cnt, B * 4));
  TRY(dalloc(&d->mn_a0f, B)); TRY(dalloc(&d->mn_ss, B)); TRY(dalloc(&d->mn_dnp, B)); TRY(dalloc(&d->mn_flags, B));
  TRY(dalloc(&d->u0, B * n)); TRY(dalloc(&d->u1, B * n)); TRY(dalloc(&d->u2, B * n)); TRY(dalloc(&d->up, B * n)); TRY(dalloc(&d->beta, B * n));
  TRY(dalloc(&d->atol, n)); TRY(dalloc(&d->emask, n)); TRY(dalloc(&d->nactive, 2)); TRY(dalloc(&d->part, B));
  return CADNIP_OK;
}

}  // namespace
#include "tran_ctrl.hpp"
namespace {
__global__ void __launch_bounds__(64) k_tran_init(TranArgs a) {
  const int inst = blockIdx.x, tid = threadIdx.x, n = a.n;
  double* u = a.u + (size_t)inst * n;
  double* u0 = a.u0 + (size_t)inst * n;
  for (int i = tid; i < n; i += 64) { double v = u[i]; u0[i] = v; a.u1[(size_t)inst * n + i] = v; a.u2[(size_t)inst * n + i] = v; }
  int bp = 0;
  while (bp < a.n_break && a.breaks[bp] <= a.t0) ++bp;
  int si = 0;
  while (si < a.n_save && a.save_t[si] <= a.t0) {
    double* o = a.out + ((size_t)inst * a.n_save + si) * a.n_obs;
    for (int j = tid; j < a.n_obs; j += 64) o[j] = u[a.obs[j]];
    ++si;
  }
  __syncthreads();
  if (tid == 0) {
    a.flags[inst] = 0;
    for (int c = 0; c < 4; ++c) a.cnt[(size_t)inst * 4 + c] = 0;
  }
  __syncthreads();
  StepState s;
  s.t = a.t0; s.h = a.h0; s.hprev = a.h0; s.hpp = a.h0; s.hp3 = a.h0; s.tn = a.t0; s.a0 = 0.0;
  s.nhist = 1; s.ord = 1; s.k = 0; s.status = 0; s.bp = bp; s.si = si;
  s.c_newton = s.c_accept = s.c_reject = s.c_fail = 0;
  s.t_break = next_break(a, bp); s.t_save = next_save(a, si);
  s.a0f = 0.0; s.ss = 20.0; s.dnp = 0.0; s.dsc = 1.0; s.mflags = MN_NEED;
  GlobalVecs v(a, inst);
  prepare_step(a, v, s, tid, a.t0, a.h0, 1, a.h0, a.h0);
  store_state(a, inst, tid, s);
}

// NT threads per instance: one wave for the circuits of a sweep (n of a few hundred), a workgroup of four for a large one
template <int NT>
__global__ void __launch_bounds__(NT) k_tran_update(TranArgs a) {
  const int inst = blockIdx.x, tid = threadIdx.x;
  StepState s = load_state(a, inst);
  if (s.status != 0) return;
  const int bad = a.flags[inst] & 1;
  grp_sync<GlobalVecsT<NT>>();
  if (tid == 0) a.flags[inst] = 0;
  GlobalVecsT<NT> v(a, inst);
  if (a.newton_mode) {
    // per-op path: every round restamps and refactors (the factors live in LDS for the length of one k_lu_f2 launch), so the Jacobian is
    // always current -- a failed iteration halves the step at once, nothing is scaled.  The rate constant follows the same events as in
    // the fused kernel (reset to 20 where that one would refactor: first round, a0 outside IDA's window, 20 steps), so both paths accept
    // iterates by the same rule.  oracle/cpu_port.cpp mirrors this as newton_mode 2.
    const bool setup = !(s.mflags & MN_VALID) || (s.k == 0 && (s.a0 < 0.6 * s.a0f || s.a0 * 0.6 > s.a0f || (s.mflags >> MN_SINCE_SHIFT) >= 20));
    if (setup) { s.a0f = s.a0; s.ss = 20.0; s.mflags = MN_VALID | MN_JCUR; }
    else s.mflags |= MN_JCUR;
    s.dsc = 1.0;
  }
  tran_update_body(a, v, s, inst, tid, bad);
  store_state(a, inst, tid, s);
}

__global__ void k_count_running(const int* status, int B, int* nactive) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = (i < B && status[i] == 0) ? 1 : 0;
  unsigned long long m = __ballot(r);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(nactive, __popcll(m));
}

// The running-instance count reaches the host through a store of this kernel into mapped pinned memory, not through an asynchronous
// device-to-host copy: under `rocprofv3 --pmc` (which serialises and re-queues the application's kernels) a copy queued behind the counting
// kernel was observed to run BEFORE it -- the host read 0, the Newton loop of cadnip_dc_run stopped at round 0 and every instance counted
// as failed.  Kernels of one stream stay ordered among themselves, and a finished kernel's stores to host memory are visible after the
// stream / event synchronisation.
__global__ void k_publish_int(const int* src, int* dst_host) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { *dst_host = *src; __threadfence_system(); }
}

// ------------------------------------------------------------------------------------------
// DC: PCNR Newton state machine (solve.jl:599-698) / plain Newton (solve.jl:542-578)
// ------------------------------------------------------------------------------------------
struct DCArgs {
  double *u, *resid, *delta, *limit_w; const double* limit_init;
  int *active, *flags, *status, *dcstate, *action, *cold; long long* cnt;
  int B, n, n_limits, use_pcnr, maxiters; double abstol;
};

// `part` (may be null = everyone): an instance that sits this run out is parked -- status 2, inactive -- and none of the
// kernels of the run touches its state
__global__ void __launch_bounds__(64) k_dc_init(DCArgs a, int cold_start, const int* part) {
  const int inst = blockIdx.x, tid = threadIdx.x, n = a.n;
  if (part && !part[inst]) {
    if (tid == 0) { a.status[inst] = 2; a.dcstate[inst] = 0; a.action[inst] = 0; a.active[inst] = 0; }
    return;
  }
  double* u = a.u + (size_t)inst * n;
  int nz = 0;
  for (int i = tid; i < n; i += 64) if (u[i] != 0.0) nz = 1;
  nz = wave_any(nz);
  // cold start (iszero(u0)): seed the limit variables (solve.jl:620-625)
  if (cold_start && !nz && a.use_pcnr)
    for (int k = tid; k < a.n_limits; k += 64) u[n - a.n_limits + k] = a.limit_init[k];
  // initjct (armed by the host for the first stamping of the run) applies to the instances that start cold; a warm start --
  // a sweep point continued from its neighbour's solution -- is stamped at its start state (solve.jl:615-625)
  if (tid == 0) a.cold[inst] = (cold_start && !nz) ? 1 : 0;
  if (tid == 0) { a.status[inst] = 0; a.dcstate[inst] = 0; a.action[inst] = 0; a.active[inst] = 1; a.flags[inst] = 0; for (int c = 0; c < 4; ++c) a.cnt[(size_t)inst * 4 + c] = 0; }
}

__global__ void __launch_bounds__(64) k_dc_check(DCArgs a) {
  const int inst = blockIdx.x, tid = threadIdx.x, n = a.n;
  if (a.status[inst] != 0) return;
  const double* F = a.resid + (size_t)inst * n;
  double* u = a.u + (size_t)inst * n;
  double s = 0.0; int bad = 0;
  for (int i = tid; i < n; i += 64) { double f = F[i]; if (!isfinite(f)) bad = 1; s += f * f; }
  s = wave_sum(s); bad = wave_any(bad);
  const double nrm = sqrt(s);
  int st = a.dcstate[inst];
  long long it = a.cnt[(size_t)inst * 4 + 0];
  int action = 0, status = 0;
  // The PCNR loop tests convergence at the top of iterations 1..maxiters only (solve.jl:630-663): after its last solve it
  // returns unconverged without looking at the residual again.  The plain Newton stage gets the check after its last step.
  const bool pcnr = a.use_pcnr && a.n_limits > 0;
  if (bad) status = -1;
  else if (pcnr && st == 0 && it >= a.maxiters) status = -3;
  else if (nrm < a.abstol) {
    if (!a.use_pcnr || a.n_limits == 0) status = 1;
    else if (st == 0) {   // settle the limit slots, verify on the next rebuild (solve.jl:640-657)
      const double* lw = a.limit_w + (size_t)inst * n;
      for (int i = n - a.n_limits + tid; i < n; i += 64) u[i] = lw[i];
      st = 1; action = 1;
    } else status = 1;
  } else st = 0;
  if (status == 0 && action == 0 && it >= a.maxiters) status = -3;
  if (tid == 0) {
    a.dcstate[inst] = st; a.action[inst] = action; a.status[inst] = status;
    a.active[inst] = (status == 0 && action == 0) ? 1 : 0;   // mask for the LU + update kernels of this round
  }
}

__global__ void __launch_bounds__(64) k_dc_update(DCArgs a) {
  const int inst = blockIdx.x, tid = threadIdx.x, n = a.n;
  const int status = a.status[inst];
  if (status == 0 && a.action[inst] == 0) {
    double* u = a.u + (size_t)inst * n;
    const double* d = a.delta + (size_t)inst * n;
    int bad = (a.flags[inst] & 1);     // the refactorisation met a zero / non-finite pivot: leave u alone (the host may re-pivot)
    if (!bad)
      for (int i = tid; i < n; i += 64) { double dd = d[i]; if (!isfinite(dd)) bad = 1; u[i] -= dd; }
    bad = wave_any(bad);
    if (a.use_pcnr && a.n_limits > 0) {
      const double* lw = a.limit_w + (size_t)inst * n;
      for (int i = n - a.n_limits + tid; i < n; i += 64) u[i] = lw[i];
    }
    // a solve that failed (zero / non-finite pivot, non-finite step) is not an iteration: the reference's count is of the
    // updates applied (solve.jl:667-690; a SingularException leaves the loop before the counter moves)
    if (tid == 0) { if (bad) a.status[inst] = -2; else a.cnt[(size_t)inst * 4 + 0] += 1; }
  }
  __syncthreads();
  if (tid == 0) a.active[inst] = (a.status[inst] == 0) ? 1 : 0;   // next round's rebuild mask
}

int count_running(CadnipHandle* h, int* out) {
  Driver* d = h->drv;
  TRY_RC(dev_zero_async(h, d->nactive, sizeof(int)));
  hipLaunchKernelGGL(k_count_running, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, d->status, h->B, d->nactive);
  hipLaunchKernelGGL(k_publish_int, dim3(1), dim3(64), 0, h->stream, (const int*)d->nactive, h->d_pinned);
  HIP_TRY(hipStreamSynchronize(h->stream));
  *out = ((volatile int*)h->h_pinned)[0];
  return CADNIP_OK;
}

// one DC Newton run on the whole batch with the handle's current spec; returns per-instance status in drv->status
int dc_newton(CadnipHandle* h, double abstol, int maxiters, int use_pcnr, int cold_start, long long* iters_total, int fused = 0, const int* d_part = nullptr) {
  Driver* d = h->drv;
  DCArgs a{h->d_u, h->d_resid, h->d_delta, h->d_limit_w, h->d_limit_init, h->d_active, h->d_flags, d->status, d->dcstate, d->action, h->d_cold, d->cnt,
           h->B, h->n, h->n_limits, (use_pcnr && h->n_limits > 0) ? 1 : 0, maxiters, abstol};
  hipLaunchKernelGGL(k_dc_init, dim3(h->B), dim3(64), 0, h->stream, a, cold_start, d_part);
  TRY_RC(dev_zero_async(h, h->d_gamma, (size_t)h->B * sizeof(double)));
  TRY_RC(dev_zero_async(h, h->d_du, (size_t)h->B * h->n * sizeof(double)));
  TRY_RC(dev_zero_async(h, h->d_t, (size_t)h->B * sizeof(double)));
  int saved_initjct = h->initjct;
  h->initjct = (cold_start && a.use_pcnr) ? 1 : 0;   // armed for the first stamping only (solve.jl:624,632)
  int rc = CADNIP_OK;
  if (fused && h->analyzed && !h->homotopy && h->spec.gshunt == 0.0 && h->spec.srcFact >= 1.0 && !h->va_ext && fused2_fits(h)) {
    // the whole Newton loop of every instance in the fused kernel; the host only looks at the running count
    TranArgs ta{};
    ta.u = h->d_u; ta.limit_w = h->d_limit_w; ta.status = d->status; ta.cnt = d->cnt; ta.active = h->d_active; ta.flags = h->d_flags;
    ta.B = h->B; ta.n = h->n; ta.n_limits = h->n_limits;
    int first = h->initjct;
    for (int launch = 0; launch < 4; ++launch) {
      rc = launch_fused2_dc(h, ta, 2 * maxiters + 4, abstol, maxiters, a.use_pcnr, h->spec.mode, first, d->dcstate); if (rc) break;
      first = 0;
      int running = 0;
      rc = count_running(h, &running); if (rc) break;
      if (running == 0) break;
    }
    h->initjct = saved_initjct;
    return rc;
  }
  int prev_running = h->B, repivots = 0;
  for (int round = 0; round < 2 * maxiters + 4; ++round) {
    rc = launch_rebuild(h); if (rc) break;
    h->initjct = 0;
    rc = launch_residual(h, h->d_du); if (rc) break;
    hipLaunchKernelGGL(k_dc_check, dim3(h->B), dim3(64), 0, h->stream, a);
    rc = launch_factor_solve(h, true, h->d_resid, h->d_delta); if (rc) break;
    hipLaunchKernelGGL(k_dc_update, dim3(h->B), dim3(64), 0, h->stream, a);
    int running = 0;
    rc = count_running(h, &running); if (rc) break;
    if (running < prev_running && repivots < 2) {
      // Some instance stopped.  If its refactorisation hit a zero pivot, the static pivot order does not fit this operating
      // point: choose a new one on the Jacobian at hand and let the instance repeat the iteration -- what KLU does when
      // klu_refactor fails and the caller falls back to klu_factor (solve.jl:667-670 keeps the symbolic object only as
      // long as refactoring works).  The new order serves every instance from here on.
      std::vector<int> stat((size_t)h->B);
      HIP_TRY(hipMemcpy(stat.data(), d->status, stat.size() * sizeof(int), hipMemcpyDeviceToHost));
      int victim = -1;
      for (int i = 0; i < h->B && victim < 0; ++i) if (stat[i] == -2) victim = i;
      if (victim >= 0) {
        ++repivots;
        std::vector<int> act((size_t)h->B);
        for (int i = 0; i < h->B; ++i) act[i] = (stat[i] == 0 || stat[i] == -2) ? 1 : 0;
        HIP_TRY(hipMemcpy(h->d_active, act.data(), act.size() * sizeof(int), hipMemcpyHostToDevice));
        rc = launch_jacobian(h); if (rc) break;                 // J = G + gamma C of this round (G, C are still the round's stamps)
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (cadnip_analyze(h, victim) == CADNIP_OK) {
          for (int i = 0; i < h->B; ++i) if (stat[i] == -2) stat[i] = 0;
          HIP_TRY(hipMemcpy(d->status, stat.data(), stat.size() * sizeof(int), hipMemcpyHostToDevice));
          rc = count_running(h, &running); if (rc) break;
        } else {
          for (int i = 0; i < h->B; ++i) act[i] = stat[i] == 0 ? 1 : 0;
          HIP_TRY(hipMemcpy(h->d_active, act.data(), act.size() * sizeof(int), hipMemcpyHostToDevice));
        }
      }
    }
    prev_running = running;
    if (running == 0) break;
  }
  h->initjct = saved_initjct;
  return rc;
}

}  // namespace

extern "C" {

void cadnip_driver_free(CadnipHandle* h) {
  if (!h || !h->drv) return;
  Driver* d = h->drv;
  void* ptrs[] = {d->t, d->h, d->hprev, d->hpp, d->nhist, d->order, d->k, d->status, d->bp_idx, d->save_idx, d->dcstate, d->action, d->cnt,
                  d->u0, d->u1, d->u2, d->up, d->beta, d->atol, d->emask, d->breaks, d->save_t, d->out, d->obs, d->nactive, d->part,
                  d->mn_a0f, d->mn_ss, d->mn_dnp, d->mn_flags, d->u3, d->hp3};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete d;
  h->drv = nullptr;
}

int cadnip_tran_state(CadnipHandle* h, double* t_host, double* h_host, int32_t* order_host) {
  if (!h || !h->drv) return CADNIP_NOTREADY;
  Driver* d = h->drv;
  size_t B = h->B;
  if (t_host) HIP_TRY(hipMemcpy(t_host, d->t, B * sizeof(double), hipMemcpyDeviceToHost));
  if (h_host) HIP_TRY(hipMemcpy(h_host, d->h, B * sizeof(double), hipMemcpyDeviceToHost));
  if (order_host) HIP_TRY(hipMemcpy(order_host, d->order, B * sizeof(int), hipMemcpyDeviceToHost));
  return CADNIP_OK;
}

// DC operating point with the reference's fallback chain (_dc_solve_with_fallbacks, solve.jl:871-929), PER INSTANCE: sweep
// points are independent circuits (sweeps.jl:696-703), so an instance leaves the chain at the first stage that converges for
// it and its solution is never touched again; only the instances still unsolved take part in the later stages, each on its
// own homotopy ladder (per-instance gshunt / srcFact, kernels.hip: k_assemble).
//   stage 0  PCNR Newton from the caller's start point (solve.jl:599-698)  -- or plain Newton when use_pcnr is off
//   stage 1  plain Newton from the caller's start point (solve.jl:899-903)
//   stage 2  gshunt stepping from zero: 1e-3, /10 ... 1e-12, then the target; a failed rung restores the last solution
//            and takes the square root of the factor until it is <= 1.5 (solve.jl:720-783)
//   stage 3  source stepping from zero: srcFact 0, +0.1 ... 1; a failed rung halves the raise (solve.jl:805-850)
// Every Newton run of every instance is logged (cadnip_dc_log_*): stage, rung value, converged, Newton solves.
int cadnip_dc_run(CadnipHandle* h, const CadnipDCOpts* o, double* u_host, int32_t* converged_host, CadnipRunStats* st) {
  if (!h || !o || !u_host) return CADNIP_BADARG;
  TRY(ensure_driver(h));
  Driver* d = h->drv;
  const size_t B = h->B, n = h->n;
  auto w0 = std::chrono::steady_clock::now();
  d->dc_log.clear();
  // the symbolic phase needs one numeric Jacobian: stamp once at the start point
  if (!h->analyzed) {
    HIP_TRY(hipMemcpy(h->d_u, u_host, B * n * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> ones(B, 1);
    HIP_TRY(hipMemcpy(h->d_active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
    TRY_RC(dev_zero_async(h, h->d_gamma, B * sizeof(double)));
    TRY_RC(dev_zero_async(h, h->d_t, B * sizeof(double)));
    // pattern-complete sample: G + 1e9*C makes every structural entry of the unified pattern visible
    std::vector<double> g(B, 1e9);
    int ij = h->initjct;
    h->initjct = (o->cold_start && o->use_pcnr && h->n_limits > 0) ? 1 : 0;
    int rc = launch_rebuild(h);
    h->initjct = ij;
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h->d_gamma, g.data(), B * sizeof(double), hipMemcpyHostToDevice));
    TRY(launch_jacobian(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    TRY(cadnip_analyze(h, 0));
  }
  // whatever happens below, the handle leaves with its own spec and no homotopy terms
  struct HomotopyGuard {
    CadnipHandle* h;
    ~HomotopyGuard() { (void)upload_homotopy(h, nullptr, nullptr); }
  } guard{h};
  // per-instance start state of the next run / states after the last run.  `start` and `U` are built only when the first run leaves
  // someone unsolved: the usual case -- everybody converges in stage 0 -- moves the state once up and once down, nothing more
  std::vector<double> start, U, R(B * n);
  std::vector<int> fin(B, 0), status(B), part(B, 1), out_of_run(B, 0);
  if (o->participate)
    for (size_t i = 0; i < B; ++i) if (!o->participate[i]) { part[i] = 0; out_of_run[i] = 1; fin[i] = 1; }   // (fin: no stage picks them up)
  std::vector<long long> cnt(B * 4);
  std::vector<double> gsh(B, h->spec.gshunt), sfc(B, h->spec.srcFact);
  long long iters = 0;
  bool direct = false;
  // one Newton run of the instances in `part`, each from U[i] with its own (gsh[i], sfc[i]); results in status / R / cnt
  auto run = [&](int use_pcnr, int cold_start, int fused, int stage, const std::vector<double>& rung) -> int {
    // (blocking copies: the stream is idle here, and the start state must be in place before the first kernel is queued -- see k_publish_int)
    HIP_TRY(hipMemcpy(h->d_u, U.empty() ? u_host : U.data(), B * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->part, part.data(), B * sizeof(int), hipMemcpyHostToDevice));
    TRY(upload_homotopy(h, gsh.data(), sfc.data()));
    TRY(dc_newton(h, o->abstol, o->maxiters, use_pcnr, cold_start, nullptr, fused, d->part));
    HIP_TRY(hipMemcpy(status.data(), d->status, B * sizeof(int), hipMemcpyDeviceToHost));
    direct = stage == 0 && !o->participate;
    for (size_t i = 0; i < B && direct; ++i) direct = status[i] == 1;
    HIP_TRY(hipMemcpy(direct ? u_host : R.data(), h->d_u, B * n * sizeof(double), hipMemcpyDeviceToHost));   // direct: the first run solved everybody -- straight to the caller
    HIP_TRY(hipMemcpy(cnt.data(), d->cnt, B * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < B; ++i)
      if (part[i]) { iters += cnt[i * 4]; d->dc_log.push_back({(int)i, stage, rung[i], status[i] == 1 ? 1 : 0, cnt[i * 4]}); }
    return CADNIP_OK;
  };
  auto take = [&](size_t i) { std::copy(R.begin() + i * n, R.begin() + (i + 1) * n, U.begin() + i * n); };
  auto n_open = [&]() { int k = 0; for (size_t i = 0; i < B; ++i) k += !fin[i]; return k; };
  const std::vector<double> none(B, 0.0);
  // ---- stage 0: PCNR (or plain Newton) from the caller's start point
  TRY(run(o->use_pcnr, o->cold_start, o->fused, 0, none));
  if (direct) {
    // everybody converged in the first run: the device holds the solutions already, the caller's array has them too
    if (converged_host) for (size_t i = 0; i < B; ++i) converged_host[i] = 1;
    std::vector<int> ones(B, 1);
    HIP_TRY(hipMemcpy(h->d_cold, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));      // (d_active: dc_newton left every instance active)
    HIP_TRY(hipMemcpy(h->d_active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
    if (st) { memset(st, 0, sizeof(*st)); st->newton_iters = iters; st->n_failed = 0; st->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); }
    return CADNIP_OK;
  }
  start.assign(u_host, u_host + B * n);
  U = start;
  for (size_t i = 0; i < B; ++i) if (part[i]) { take(i); if (status[i] == 1) fin[i] = 1; }
  // ---- stage 1: plain Newton from the caller's start point, for those PCNR did not solve
  if (n_open() && o->use_pcnr && h->n_limits > 0) {
    for (size_t i = 0; i < B; ++i) { part[i] = !fin[i]; if (part[i]) std::copy(start.begin() + i * n, start.begin() + (i + 1) * n, U.begin() + i * n); }
    TRY(run(0, 0, 0, 1, none));
    for (size_t i = 0; i < B; ++i) if (part[i]) { take(i); if (status[i] == 1) fin[i] = 1; }
  }
  if (n_open() && o->use_stepping) {
    // ---- stage 2: gshunt stepping, one ladder per open instance
    const double target = h->spec.gshunt, gfloor = fmax(target, 1e-12);
    struct Rung { double g = 1e-3, factor = 10.0; int steps = 0; bool finalizing = false, over = false; std::vector<double> u, saved; };
    std::vector<Rung> L(B);
    for (size_t i = 0; i < B; ++i) { L[i].over = fin[i]; if (!fin[i]) { L[i].u.assign(n, 0.0); L[i].saved.assign(n, 0.0); } }
    for (;;) {
      int k = 0;
      for (size_t i = 0; i < B; ++i) {
        part[i] = !L[i].over;
        if (!part[i]) { gsh[i] = h->spec.gshunt; continue; }
        ++k;
        gsh[i] = L[i].finalizing ? target : L[i].g;
        std::copy(L[i].u.begin(), L[i].u.end(), U.begin() + i * n);
      }
      if (!k) break;
      TRY(run(0, 0, 0, 2, gsh));
      for (size_t i = 0; i < B; ++i) {
        if (!part[i]) continue;
        Rung& r = L[i];
        const bool ok = status[i] == 1;
        if (r.finalizing) {                                   // the solve at the exact target ends the ladder either way
          if (ok) { take(i); fin[i] = 1; } else std::copy(r.u.begin(), r.u.end(), U.begin() + i * n);
          r.over = true;
          continue;
        }
        ++r.steps;
        if (ok) {
          r.u.assign(R.begin() + i * n, R.begin() + (i + 1) * n);
          r.saved = r.u;
          if (r.g <= gfloor) {
            if (r.g != target) r.finalizing = true;
            else { take(i); fin[i] = 1; r.over = true; }
          } else { r.g /= r.factor; if (r.g < gfloor) r.g = gfloor; }
        } else {
          if (r.factor <= 1.5) r.over = true;                 // cannot make progress
          else { r.factor = sqrt(r.factor); r.u = r.saved; }
        }
        if (!r.over && !r.finalizing && r.steps >= 20) r.over = true;   // max_steps
        if (r.over && !fin[i]) std::copy(r.u.begin(), r.u.end(), U.begin() + i * n);
      }
    }
    for (size_t i = 0; i < B; ++i) gsh[i] = h->spec.gshunt;
    // ---- stage 3: source stepping for the rest
    if (n_open()) {
      struct Ramp { double src = 0.0, conv = 0.0, raise = 0.1; int steps = 0; bool over = false; std::vector<double> u, saved; };
      std::vector<Ramp> S(B);
      for (size_t i = 0; i < B; ++i) { S[i].over = fin[i]; if (!fin[i]) { S[i].u.assign(n, 0.0); S[i].saved.assign(n, 0.0); } }
      for (;;) {
        int k = 0;
        for (size_t i = 0; i < B; ++i) {
          part[i] = !S[i].over;
          if (!part[i]) { sfc[i] = h->spec.srcFact; continue; }
          ++k;
          sfc[i] = S[i].src;
          std::copy(S[i].u.begin(), S[i].u.end(), U.begin() + i * n);
        }
        if (!k) break;
        TRY(run(0, 0, 0, 3, sfc));
        for (size_t i = 0; i < B; ++i) {
          if (!part[i]) continue;
          Ramp& r = S[i];
          ++r.steps;
          if (status[i] == 1) {
            r.conv = r.src;
            r.u.assign(R.begin() + i * n, R.begin() + (i + 1) * n);
            r.saved = r.u;
            if (r.src >= 1.0) { take(i); fin[i] = 1; r.over = true; }
            else r.src = fmin(r.src + r.raise, 1.0);
          } else {
            if (r.src - r.conv < 1e-6) r.over = true;
            else { r.raise /= 2.0; r.src = r.conv + r.raise; r.u = r.saved; }
          }
          if (!r.over && r.steps >= 50) r.over = true;
          if (r.over && !fin[i]) std::copy(r.u.begin(), r.u.end(), U.begin() + i * n);
        }
      }
      for (size_t i = 0; i < B; ++i) sfc[i] = h->spec.srcFact;
    }
  }
  const int n_failed = n_open();
  memcpy(u_host, U.data(), B * n * sizeof(double));
  HIP_TRY(hipMemcpy(h->d_u, U.data(), B * n * sizeof(double), hipMemcpyHostToDevice));
  if (converged_host) for (size_t i = 0; i < B; ++i) converged_host[i] = out_of_run[i] ? 0 : fin[i];
  // leave every instance active for subsequent ABI calls (and subject to a cadnip_set_initjct of the caller's)
  std::vector<int> ones(B, 1);
  HIP_TRY(hipMemcpy(h->d_active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_cold, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
  if (st) {
    memset(st, 0, sizeof(*st));
    st->newton_iters = iters;
    st->n_failed = n_failed;
    st->wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  }
  return n_failed ? CADNIP_NOCONV : CADNIP_OK;
}

// the log of the last cadnip_dc_run: entries in execution order, per (instance, Newton run)
int32_t cadnip_dc_log_size(CadnipHandle* h) { return (h && h->drv) ? (int32_t)h->drv->dc_log.size() : 0; }
int cadnip_dc_log_get(CadnipHandle* h, int32_t* inst, int32_t* stage, double* value, int32_t* ok, int64_t* iters) {
  if (!h || !h->drv) return CADNIP_NOTREADY;
  size_t k = 0;
  for (auto& e : h->drv->dc_log) {
    if (inst) inst[k] = e.inst; if (stage) stage[k] = e.stage; if (value) value[k] = e.value; if (ok) ok[k] = e.ok; if (iters) iters[k] = e.iters;
    ++k;
  }
  return CADNIP_OK;
}

int cadnip_tran_run(CadnipHandle* h, const CadnipTranOpts* o, double* out_host, int64_t* per_inst_host, CadnipRunStats* st) {
  if (!h || !o || !o->abstol || o->t1 <= o->t0 || o->n_save < 0 || o->n_break < 0) return CADNIP_BADARG;
  if (!h->analyzed) return CADNIP_NOTREADY;
  for (int i = 1; i < o->n_break; ++i) if (!(o->breaks[i] > o->breaks[i - 1])) return CADNIP_BADARG;
  for (int i = 1; i < o->n_save; ++i) if (!(o->save_t[i] >= o->save_t[i - 1])) return CADNIP_BADARG;
  for (int i = 0; i < o->n_obs; ++i) if (o->obs[i] < 0 || o->obs[i] >= h->n) return CADNIP_BADARG;
  TRY(ensure_driver(h));
  Driver* d = h->drv;
  const size_t B = h->B, n = h->n;
  const int n_obs = o->n_obs > 0 ? o->n_obs : (int)n;
  TRY(drealloc(&d->breaks, &d->brk_cap, (size_t)o->n_break));
  TRY(drealloc(&d->save_t, &d->save_cap, (size_t)o->n_save));
  TRY(drealloc(&d->obs, &d->obs_cap, (size_t)n_obs));
  TRY(drealloc(&d->out, &d->out_cap, B * (size_t)o->n_save * n_obs));
  if (o->n_break) HIP_TRY(hipMemcpy(d->breaks, o->breaks, o->n_break * sizeof(double), hipMemcpyHostToDevice));
  if (o->n_save) HIP_TRY(hipMemcpy(d->save_t, o->save_t, o->n_save * sizeof(double), hipMemcpyHostToDevice));
  std::vector<int> obs(n_obs);
  for (int i = 0; i < n_obs; ++i) obs[i] = o->n_obs > 0 ? o->obs[i] : i;
  HIP_TRY(hipMemcpy(d->obs, obs.data(), n_obs * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d->atol, o->abstol, n * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double> emask(n, 1.0);
  int n_err = (int)n;
  if (o->err_mask) { n_err = 0; for (size_t i = 0; i < n; ++i) { emask[i] = o->err_mask[i] != 0.0 ? 1.0 : 0.0; n_err += emask[i] != 0.0; } }
  HIP_TRY(hipMemcpy(d->emask, emask.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double span = o->t1 - o->t0;
  double hmax = o->hmax > 0 ? o->hmax : span / 50.0;
  double h0 = o->h0 > 0 ? o->h0 : span * 1e-6;
  double hmin = o->hmin > 0 ? o->hmin : span * 1e-14;
  TranArgs a{h->d_u, h->d_du, h->d_delta, h->d_limit_w, h->d_t, h->d_gamma, h->d_active, h->d_flags,
             d->t, d->h, d->hprev, d->hpp, d->nhist, d->order, d->k, d->status, d->bp_idx, d->save_idx, d->cnt,
             d->u0, d->u1, d->u2, d->up, d->beta, d->atol, d->emask, d->breaks, d->save_t, d->obs, d->out, d->nactive,
             h->B, h->n, h->n_limits, o->n_break, o->n_save, n_obs, n_err,
             o->t0, o->t1, o->reltol, h0, hmin, hmax, o->newton_tol > 0 ? o->newton_tol : 1e-3,
             o->max_newton > 0 ? o->max_newton : 10, o->max_order > 0 ? o->max_order : 2, o->use_pcnr,
             o->newton_mode ? 1 : 0, d->mn_a0f, d->mn_ss, d->mn_dnp, d->mn_flags, o->step_rule ? 1 : 0, nullptr, nullptr};
  if (a.max_order > 3) a.max_order = 3;
  if (a.max_order >= 3) {                                 // variable-step BDF3: a fourth history vector and a third step size per instance
    TRY(dalloc(&d->u3, (size_t)h->B * h->n)); TRY(dalloc(&d->hp3, (size_t)h->B));
    a.u3 = d->u3; a.hp3 = d->hp3;
  }
  if (!h->analyzed) return CADNIP_NOTREADY;              // the symbolic LU phase (cadnip_analyze*) comes first
  // the per-op kernels take over where the fused kernel cannot run: external generated models (they exist in the per-op stamping kernel only), a
  // circuit too large for the LDS-resident kernel, Newton mode 1 on a circuit outside the lean device set
  const bool use_fused = o->fused && !h->va_ext && fused2_fits(h) && (!o->newton_mode || fused2_mode1_ok(h));
  struct ModeGuard { CadnipHandle* h; int saved; ~ModeGuard() { h->spec.mode = saved; } } mode_guard{h, h->spec.mode};
  h->spec.mode = 1;   // :tran (restored on every exit path)
  hipLaunchKernelGGL(k_tran_init, dim3(h->B), dim3(64), 0, h->stream, a);
  HIP_TRY(hipStreamSynchronize(h->stream));
  auto w0 = std::chrono::steady_clock::now();
  int64_t launches = 0;
  const int64_t max_it = o->max_iterations > 0 ? o->max_iterations : 50000000;
  int rc = CADNIP_OK, running = h->B;
  // Newton rounds between two looks at the running-instance count.  A fused launch loads the structure tables and each
  // instance's state once and ends when its slowest wave has done its rounds (measured, DFF sweep: 8 rounds per launch
  // 38.8 M iterations/s, 64: 43.2, 1024: 44.8, one launch for the whole transient: 45.2).
  const int check_every = use_fused ? 1024 : 8;
  if (use_fused) {
    // The host stays one launch ahead: launch k+1 is queued before the running-instance count of launch k is read,
    // so the GPU never waits for the host between launches.  Once every instance has finished, the one launch
    // already in the queue finds nothing to do (each wave reads its status and exits).
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int slot = 0;
    bool have_prev = false;
    while (launches < max_it) {
      rc = launch_fused2_rounds(h, a, check_every); if (rc) break;
      launches += check_every;
      if (dev_zero_async(h, d->nactive + slot, sizeof(int)) != CADNIP_OK) { rc = CADNIP_HIPERROR; break; }
      hipLaunchKernelGGL(k_count_running, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, d->status, h->B, d->nactive + slot);
      hipLaunchKernelGGL(k_publish_int, dim3(1), dim3(64), 0, h->stream, (const int*)(d->nactive + slot), h->d_pinned + slot);
      if (hipEventRecord(ev[slot], h->stream) != hipSuccess) { rc = CADNIP_HIPERROR; break; }
      if (have_prev) {
        if (hipEventSynchronize(ev[1 - slot]) != hipSuccess) { rc = CADNIP_HIPERROR; break; }
        running = ((volatile int*)h->h_pinned)[1 - slot];
        if (running == 0) break;
      }
      have_prev = true;
      slot ^= 1;
    }
    hipError_t es = hipStreamSynchronize(h->stream);
    (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
    if (es != hipSuccess) { set_last_error("hipStreamSynchronize", es); return CADNIP_HIPERROR; }
    if (!rc && running > 0) rc = count_running(h, &running);
  }
  while (!use_fused && running > 0 && launches < max_it) {
    for (int c = 0; c < check_every; ++c) {
      rc = launch_rebuild(h); if (rc) break;
      rc = launch_residual(h, h->d_du); if (rc) break;
      rc = launch_factor_solve(h, true, h->d_resid, h->d_delta); if (rc) break;
      { ProfScope ps(h, "tran_update");
        if (h->n >= 4096) hipLaunchKernelGGL(k_tran_update<256>, dim3(h->B), dim3(256), 0, h->stream, a);
        else hipLaunchKernelGGL(k_tran_update<64>, dim3(h->B), dim3(64), 0, h->stream, a); }
      ++launches;
    }
    if (rc) break;
    rc = count_running(h, &running); if (rc) break;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  if (rc) return rc;
  std::vector<long long> cnt(B * 4);
  std::vector<int> status(B);
  HIP_TRY(hipMemcpy(cnt.data(), d->cnt, B * 4 * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(status.data(), d->status, B * sizeof(int), hipMemcpyDeviceToHost));
  if (out_host && o->n_save) HIP_TRY(hipMemcpy(out_host, d->out, B * (size_t)o->n_save * n_obs * sizeof(double), hipMemcpyDeviceToHost));
  CadnipRunStats s; memset(&s, 0, sizeof(s));
  for (size_t i = 0; i < B; ++i) {
    s.newton_iters += cnt[i * 4 + 0]; s.steps_accepted += cnt[i * 4 + 1]; s.steps_rejected += cnt[i * 4 + 2]; s.newton_failures += cnt[i * 4 + 3];
    if (status[i] != 1) s.n_failed += 1;
    if (per_inst_host) { per_inst_host[i * 4 + 0] = cnt[i * 4 + 0]; per_inst_host[i * 4 + 1] = cnt[i * 4 + 1]; per_inst_host[i * 4 + 2] = cnt[i * 4 + 2]; per_inst_host[i * 4 + 3] = status[i]; }
  }
  s.launches = launches; s.wall_seconds = wall;
  if (st) *st = s;
  std::vector<int> ones(B, 1);
  HIP_TRY(hipMemcpy(h->d_active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice));
  return s.n_failed ? CADNIP_NOCONV : CADNIP_OK;
}

}  // extern "C"
