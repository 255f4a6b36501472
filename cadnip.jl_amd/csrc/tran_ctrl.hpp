// tran_ctrl.hpp -- device-side transient controller shared by the per-op driver (driver.hip: k_tran_init,
// k_tran_update) and the fused Newton kernel (fused2.hip).  See driver.hip for the integration method.
//
// The controller is written once, over two small policies:
//   * StepState   -- the per-instance scalars (t, h, order, Newton counter, ...).  A kernel loads them into
//                    registers once, runs any number of Newton rounds on them, and stores them back once.
//   * a vector policy V -- where the per-unknown vectors live.  GlobalVecs: everything in HBM (per-op path).
//                    The fused kernel supplies its own (u, beta and the Newton step in LDS, history in HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace cadnip {

// Wave-level synchronisation: every controller function is executed by ONE 64-lane wave per sweep
// instance (several instances may share a workgroup in the fused kernel, each taking its own control
// path), so ordering is needed only among the lanes of the calling wave.  A wave's LDS operations, and
// its vector-memory operations, execute in issue order, so data written by one lane is seen by a later
// read of another lane of the same wave without waiting: a wavefront-scope fence (no instructions, it
// only stops the compiler from moving memory operations across it) plus a wave barrier is enough.
// Never s_barrier here.
#define CADNIP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- cross-lane sums on DPP (no LDS round trips) -----------------------------------------------------
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // (bound_ctrl: every lane of these permutations has a source lane, so the "old" operand is dead -- saying so saves its zero-fill)
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of `width` consecutive lanes (width = 1, 2, 4, 8 or 16; uniform), result in every
// lane of the group.  Must be called with all 64 lanes active.
__device__ __forceinline__ double group_sum16(double v, int width) {
  if (width >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  if (width >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (width >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
  if (width >= 16) v += dpp_f64<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = group_sum16(v, 16);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ int wave_any(int v) { return __any(v); }

// a / b for finite, non-zero, normal-range b (pivots, positive error weights, step sizes): v_rcp_f64 + two Newton steps +
// one residual correction; none of IEEE division's range scaling.  For normal-range operands and quotients it IS the correctly rounded
// quotient -- bit for bit the result of `a / b` over 1.3e9 random pairs with exponents up to +-500 apart (tools/ubench/divcheck.hip) --
// so the CPU port's plain divisions stay its exact counterpart.  One wave alone pays ~45 cycles for it, 117 for the compiler's IEEE
// sequence (tools/ubench/lat.hip), and the controller's scalars are a chain of such divisions.
__device__ __forceinline__ double fast_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// x^(-1/(ord+1)) for the step-size rule, ord = 1 or 2, x > 0 finite.  `pow` costs one wave ~870 cycles (tools/ubench/lat.hip) once per
// accepted step.  Order 1: 1 / sqrt(x), both IEEE-exact.  Order 2: x = m 8^q with m in [0.5, 4), y ~ m^(-1/3) by four Newton steps
// y <- y (4 - m y^3) / 3 from a linear start (relative error 1e-13: a step-size heuristic does not need the last bits, it needs the same
// bits on both sides) -- written with single-rounding operations only (products, fma, exact frexp / ldexp), so that oracle/cpu_port.cpp's
// copy (step_root) returns the same double.
__device__ __forceinline__ double step_root(double x, int ord) {
  if (ord == 1) return fast_div(1.0, sqrt(x));
  if (ord == 3) return fast_div(1.0, sqrt(sqrt(x)));
  int e;
  double m = frexp(x, &e);                 // x = m 2^e, m in [0.5, 1)
  const int q = (e >= 0 ? e : e - 2) / 3;  // floor(e / 3)
  m = ldexp(m, e - 3 * q);                 // [0.5, 4)
  double y = fma(m, -0.17, 1.18);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const double y3 = (y * y) * y;
    y = (y * fma(-m, y3, 4.0)) * (1.0 / 3.0);
  }
  return ldexp(y, -q);
}

// ------------------------------------------------------------------------------------------
#ifndef CADNIP_TRACE_POINT
#define CADNIP_TRACE_POINT(id) do {} while (0)
#endif
// transient controller
// ------------------------------------------------------------------------------------------
struct TranArgs {
  double *u, *du, *delta, *limit_w, *tcur, *gamma;
  int *active, *flags;
  double *t, *h, *hprev, *hpp; int *nhist, *order, *k, *status, *bp_idx, *save_idx; long long* cnt;
  double *u0, *u1, *u2, *up, *beta;
  const double *atol, *emask, *breaks, *save_t; const int* obs; double* out; int* nactive;
  int B, n, n_limits, n_break, n_save, n_obs, n_err;
  double t0, t1, reltol, h0, hmin, hmax, newton_tol;
  int max_newton, max_order, use_pcnr;
  // Newton mode 1 = the nonlinear iteration as IDA runs it (the reference's integrator, sweeps.jl:600): the Jacobian is refactored
  // only on a "setup" -- first round, a0 outside [0.6, 1 / 0.6] of its value at the last one (IDA's XRATE window), 20 steps without one, or a failed iteration on
  // a stale one, which is then repeated at the same step -- the rounds between solve with the kept factors (residual-only device
  // pass); the update is scaled by 2 / (1 + a0 / a0_setup) on a stale Jacobian; convergence is IDA's rate test
  // ss * ||delta|| <= 0.33 with ss = rate / (1 - rate) carried from round to round and reset to 20 at a setup (ida.c: IDANls /
  // IDANewtonIter; the rate is taken between successive updates).  Fused kernel and CPU port only.  Per-instance state:
  int newton_mode;
  double *mn_a0f, *mn_ss, *mn_dnp; int* mn_flags;
  int step_rule;   // 0 = classical step controller, 1 = IDA's eta rule (CadnipTranOpts::step_rule)
  double *u3, *hp3;   // max_order >= 3: u at the fourth-last accepted point [B][n], the third-last step size [B] (set by the driver behind the aggregate)
};
#define MN_NEED 1     // the next round must refactor
#define MN_JCUR 2     // a refactorisation happened in this step attempt
#define MN_VALID 4    // kept factors exist (in the wave's LDS work array, or saved to HBM between launches)
#define MN_SINCE_SHIFT 8   // accepted steps since the last refactorisation

struct StepState {
  double t, h, hprev, hpp, tn, a0;   // accepted time, step in flight, the two previous steps, t + h, BDF leading coefficient
  int nhist, ord, k, status, bp, si; // history depth, order, Newton counter, 0 running / 1 done / <0 failed, next breakpoint / save index
  int c_newton, c_accept, c_reject, c_fail;   // counter increments since load
  double t_break, t_save;                     // breaks[bp] / save_t[si], +inf past the end: derived, refreshed when bp / si move
  double a0f, ss, dnp, dsc;                   // Newton mode 1: a0 of the kept factors, rate constant, previous update norm; scale of this round's update
  int mflags;
  double hp3;                                 // max_order >= 3: the step before hpp
};
template <class A> __device__ __forceinline__ double next_break(const A& a, int bp) { return bp < a.n_break ? a.breaks[bp] : __builtin_inf(); }
template <class A> __device__ __forceinline__ double next_save(const A& a, int si) { return si < a.n_save ? a.save_t[si] : __builtin_inf(); }

// All lanes of the wave hold the same StepState; saying so (readfirstlane) lets it live in scalar registers, which
// matters in the fused kernel where vector registers are the scarce resource.
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void make_uniform(StepState& s) {
  s.t = uniform_f64(s.t); s.h = uniform_f64(s.h); s.hprev = uniform_f64(s.hprev); s.hpp = uniform_f64(s.hpp);
  s.tn = uniform_f64(s.tn); s.a0 = uniform_f64(s.a0); s.t_break = uniform_f64(s.t_break); s.t_save = uniform_f64(s.t_save);
  s.nhist = __builtin_amdgcn_readfirstlane(s.nhist); s.ord = __builtin_amdgcn_readfirstlane(s.ord); s.k = __builtin_amdgcn_readfirstlane(s.k);
  s.status = __builtin_amdgcn_readfirstlane(s.status); s.bp = __builtin_amdgcn_readfirstlane(s.bp); s.si = __builtin_amdgcn_readfirstlane(s.si);
  s.c_newton = __builtin_amdgcn_readfirstlane(s.c_newton); s.c_accept = __builtin_amdgcn_readfirstlane(s.c_accept);
  s.c_reject = __builtin_amdgcn_readfirstlane(s.c_reject); s.c_fail = __builtin_amdgcn_readfirstlane(s.c_fail);
  s.a0f = uniform_f64(s.a0f); s.ss = uniform_f64(s.ss); s.dnp = uniform_f64(s.dnp); s.dsc = uniform_f64(s.dsc);
  s.mflags = __builtin_amdgcn_readfirstlane(s.mflags);
  s.hp3 = uniform_f64(s.hp3);
}

// A: TranArgs, or a view with the same member names (the fused kernel fetches these rarely used pointers on demand)
template <class A> __device__ inline StepState load_state(const A& a, int inst) {
  StepState s;
  s.t = a.t[inst]; s.h = a.h[inst]; s.hprev = a.hprev[inst]; s.hpp = a.hpp[inst]; s.tn = a.tcur[inst]; s.a0 = a.gamma[inst];
  s.nhist = a.nhist[inst]; s.ord = a.order[inst]; s.k = a.k[inst]; s.status = a.status[inst]; s.bp = a.bp_idx[inst]; s.si = a.save_idx[inst];
  s.c_newton = s.c_accept = s.c_reject = s.c_fail = 0;
  s.t_break = next_break(a, s.bp); s.t_save = next_save(a, s.si);
  s.a0f = 0.0; s.ss = 20.0; s.dnp = 0.0; s.dsc = 1.0; s.mflags = MN_NEED;
  if (a.newton_mode) { s.a0f = a.mn_a0f[inst]; s.ss = a.mn_ss[inst]; s.dnp = a.mn_dnp[inst]; s.mflags = a.mn_flags[inst]; }
  s.hp3 = a.max_order >= 3 ? a.hp3[inst] : 0.0;
  return s;
}
template <class A> __device__ inline void store_state(const A& a, int inst, int tid, const StepState& s) {
  if (tid != 0) return;
  a.t[inst] = s.t; a.h[inst] = s.h; a.hprev[inst] = s.hprev; a.hpp[inst] = s.hpp; a.tcur[inst] = s.tn; a.gamma[inst] = s.a0;
  a.nhist[inst] = s.nhist; a.order[inst] = s.ord; a.k[inst] = s.k; a.status[inst] = s.status; a.bp_idx[inst] = s.bp; a.save_idx[inst] = s.si;
  a.active[inst] = s.status == 0 ? 1 : 0;
  if (a.newton_mode) { a.mn_a0f[inst] = s.a0f; a.mn_ss[inst] = s.ss; a.mn_dnp[inst] = s.dnp; a.mn_flags[inst] = s.mflags; }
  if (a.max_order >= 3) a.hp3[inst] = s.hp3;
  long long* c = a.cnt + (size_t)inst * 4;
  c[0] += s.c_newton; c[1] += s.c_accept; c[2] += s.c_reject; c[3] += s.c_fail;
}

// Group helpers of the controller: a sweep instance is handled by V::NT threads -- one wave (the fused kernel, small systems on the per-op
// path) or a workgroup of several (large systems on the per-op path: one wave walking 76 k unknowns of the c6288 multiplier took 0.8 ms).
template <class V> __device__ __forceinline__ void grp_sync() {
  if constexpr (V::NT == 64) CADNIP_WAVE_SYNC(); else __syncthreads();
}
template <class V> __device__ __forceinline__ double grp_sum(double v) {
  if constexpr (V::NT == 64) return wave_sum(v);
  else {
    __shared__ double red[V::NT / 64];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < V::NT / 64; ++k) t += red[k];
    return t;
  }
}
template <class V> __device__ __forceinline__ int grp_any(int v) {
  if constexpr (V::NT == 64) return wave_any(v);
  else return __syncthreads_or(v);
}
// the update's three reductions at once (two norms and the failure flag): for a group of several waves one LDS exchange and ONE barrier
// instead of five.  Every wave adds the partial sums in the same order, so all of them take the same decisions.  The exchange words are
// rewritten by the next call only: callers have a barrier between two updates (a Newton round has several).
template <class V> __device__ __forceinline__ void grp_reduce3(V& v, double& s1, double& s2, int& bad) {
  if constexpr (V::NT == 64) { s1 = wave_sum(s1); s2 = wave_sum(s2); bad = wave_any(bad); }
  else if constexpr (V::OWN_REDUCE) v.reduce3(s1, s2, bad);
  else {
    __shared__ double red[V::NT / 64][3];
    const double p1 = wave_sum(s1), p2 = wave_sum(s2);
    const int pb = wave_any(bad);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = p1; red[threadIdx.x >> 6][1] = p2; red[threadIdx.x >> 6][2] = pb ? 1.0 : 0.0; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < V::NT / 64; ++k) { t1 += red[k][0]; t2 += red[k][1]; tb += red[k][2]; }
    s1 = t1; s2 = t2; bad = tb != 0.0;
  }
}

// every per-unknown vector in HBM (per-op path)
template <int NT_>
struct GlobalVecsT {
  static constexpr int KPF = 0;
  static constexpr int NT = NT_;
  static constexpr bool OWN_REDUCE = false;
  double *__restrict__ u, *__restrict__ du, *__restrict__ up, *__restrict__ beta, *__restrict__ u0, *__restrict__ u1, *__restrict__ u2;
  double* u3;
  const double *__restrict__ delta, *__restrict__ lw;
  __device__ GlobalVecsT(const TranArgs& a, int inst) {
    const size_t o = (size_t)inst * a.n;
    u = a.u + o; du = a.du + o; up = a.up + o; beta = a.beta + o; u0 = a.u0 + o; u1 = a.u1 + o; u2 = a.u2 + o; u3 = a.max_order >= 3 ? a.u3 + o : nullptr; delta = a.delta + o; lw = a.limit_w + o;
  }
  __device__ __forceinline__ double get_delta(int i, int) const { return delta[i]; }
  __device__ __forceinline__ void step_consumed(int) const {}
  __device__ __forceinline__ double get_u(int i) const { return u[i]; }
  __device__ __forceinline__ void set_u(int i, double v) const { u[i] = v; }
  __device__ __forceinline__ double get_beta(int i) const { return beta[i]; }
  __device__ __forceinline__ void set_beta(int i, double v) const { beta[i] = v; }
  __device__ __forceinline__ void set_du(int i, double v) const { du[i] = v; }
  __device__ __forceinline__ double get_lw(int i) const { return lw[i]; }
  // history (u at the last three accepted points), predictor, error weights.  `k` is the element's per-lane ordinal when
  // the policy keeps the first KPF elements of each lane in registers (fused kernel), -1 otherwise: unused here.
  __device__ __forceinline__ double h0(int i, int) const { return u0[i]; }
  __device__ __forceinline__ double h1(int i, int) const { return u1[i]; }
  __device__ __forceinline__ double h2(int i, int) const { return u2[i]; }
  __device__ __forceinline__ double hp(int i, int) const { return up[i]; }
  __device__ __forceinline__ double h3(int i, int) const { return u3[i]; }
  __device__ __forceinline__ void set_h3(int i, int, double v) { u3[i] = v; }
  __device__ __forceinline__ void set_h0(int i, int, double v) { u0[i] = v; }
  __device__ __forceinline__ void set_h1(int i, int, double v) { u1[i] = v; }
  __device__ __forceinline__ void set_h2(int i, int, double v) { u2[i] = v; }
  __device__ __forceinline__ void set_hp(int i, int, double v) { up[i] = v; }
  __device__ __forceinline__ double atol_of(const TranArgs& a, int i, int) const { return a.atol[i]; }
  __device__ __forceinline__ double emask_of(const TranArgs& a, int i, int) const { return a.emask[i]; }
  __device__ __forceinline__ void history_to_memory(int, int) const {}   // u0 / u1 are in memory: always
  __device__ __forceinline__ double mem_u0(int i) const { return u0[i]; }   // any unknown's u0 / u1 (valid after history_to_memory)
  __device__ __forceinline__ double mem_u1(int i) const { return u1[i]; }
};
typedef GlobalVecsT<64> GlobalVecs;

// f(i, k) for the elements i = tid, tid + 64, ... < n of one lane; k = the ordinal for the first V::KPF of them (a
// compile-time constant after unrolling, so that V's per-lane register arrays are indexed statically), -1 beyond
template <class V, class F>
__device__ __forceinline__ void each_elem(int n, int tid, F&& f) {
  if constexpr (V::KPF > 0) {
#pragma unroll
    for (int k = 0; k < V::KPF; ++k) { const int i = tid + V::NT * k; if (i < n) f(i, k); }
  }
  // several elements in flight per lane: with every vector in HBM (per-op path, large n) an element is a chain of memory
  // latencies, and one wave walks n / 64 of them
  if constexpr (V::KPF == 0) {
#pragma unroll 4
    for (int i = tid; i < n; i += V::NT) f(i, -1);
  } else {
    for (int i = tid + V::NT * V::KPF; i < n; i += V::NT) f(i, -1);
  }
}

// Set up the step that starts at (t, history) with proposed size h: clip to the next stop,
// pick order from the available history, extrapolate the predictor, BDF coefficients.
template <class V>
__device__ inline void prepare_step(const TranArgs& a, V& v, StepState& s, int tid, double t, double h, int nhist, double hprev, double hpp) {
  const int n = a.n;
  double tstop = a.t1;
  if (s.t_break < tstop) tstop = s.t_break;
  double rem = tstop - t, tn;
  if (h >= rem * (1.0 - 1e-9)) { h = rem; tn = tstop; }
  else if (2.0 * h > rem) { h = 0.5 * rem; tn = t + h; }
  else tn = t + h;
  int ord;
  double a0;
  if (nhist <= 1) {
    ord = 1; a0 = fast_div(1.0, h);
    each_elem<V>(n, tid, [&](int i, int k) { double p = v.h0(i, k); v.set_hp(i, k, p); v.set_u(i, p); double b = fast_div(-p, h); v.set_beta(i, b); v.set_du(i, a0 * p + b); });
  } else if (nhist == 2 || a.max_order < 2) {
    ord = 1; a0 = fast_div(1.0, h);
    double w = fast_div(h, hprev);
    each_elem<V>(n, tid, [&](int i, int k) {
      double x0 = v.h0(i, k);
      double p = x0 + w * (x0 - v.h1(i, k));
      v.set_hp(i, k, p); v.set_u(i, p);
      double b = fast_div(-x0, h);
      v.set_beta(i, b); v.set_du(i, a0 * p + b);
    });
  } else if (nhist >= 4 && a.max_order >= 3) {
    // variable-step BDF3: the derivative at t_n of the cubic through (t_n, u) and the three last accepted points (d1 < d2 < d3: their distances
    // from t_n); the predictor is the cubic through the FOUR last accepted points.  oracle/cpu_port.cpp carries the same operations in the same order.
    ord = 3;
    const double hr = s.hp3;
    const double d1 = h, d2 = h + hprev, d3 = d2 + hpp, s12 = hprev + hpp;
    a0 = (fast_div(1.0, d1) + fast_div(1.0, d2)) + fast_div(1.0, d3);
    const double a1 = -fast_div(d2 * d3, d1 * (hprev * s12)), a2 = fast_div(d1 * d3, d2 * (hprev * hpp)), a3 = -fast_div(d1 * d2, d3 * (s12 * hpp));
    const double x1 = -hprev, x2 = -s12, x3 = -(s12 + hr), x = h;
    const double L0 = fast_div(((x - x1) * (x - x2)) * (x - x3), ((0.0 - x1) * (0.0 - x2)) * (0.0 - x3));
    const double L1 = fast_div(((x - 0.0) * (x - x2)) * (x - x3), ((x1 - 0.0) * (x1 - x2)) * (x1 - x3));
    const double L2 = fast_div(((x - 0.0) * (x - x1)) * (x - x3), ((x2 - 0.0) * (x2 - x1)) * (x2 - x3));
    const double L3 = fast_div(((x - 0.0) * (x - x1)) * (x - x2), ((x3 - 0.0) * (x3 - x1)) * (x3 - x2));
    each_elem<V>(n, tid, [&](int i, int k) {
      const double x0 = v.h0(i, k), xm1 = v.h1(i, k), xm2 = v.h2(i, k);
      const double p = ((L0 * x0 + L1 * xm1) + L2 * xm2) + L3 * v.h3(i, k);
      v.set_hp(i, k, p); v.set_u(i, p);
      const double b = (a1 * x0 + a2 * xm1) + a3 * xm2;
      v.set_beta(i, b); v.set_du(i, a0 * p + b);
    });
  } else {
    ord = 2;
    double w = fast_div(h, hprev);
    a0 = fast_div(1.0 + 2.0 * w, (1.0 + w) * h);
    double a1 = fast_div(-(1.0 + w), h), a2 = fast_div(w * w, (1.0 + w) * h);
    double x1 = -hprev, x2 = -(hprev + hpp), x = h;
    double L0 = fast_div((x - x1) * (x - x2), (0.0 - x1) * (0.0 - x2));
    double L1 = fast_div((x - 0.0) * (x - x2), (x1 - 0.0) * (x1 - x2));
    double L2 = fast_div((x - 0.0) * (x - x1), (x2 - 0.0) * (x2 - x1));
    each_elem<V>(n, tid, [&](int i, int k) {
      double x0 = v.h0(i, k), xm1 = v.h1(i, k);
      double p = L0 * x0 + L1 * xm1 + L2 * v.h2(i, k);
      v.set_hp(i, k, p); v.set_u(i, p);
      double b = a1 * x0 + a2 * xm1;
      v.set_beta(i, b); v.set_du(i, a0 * p + b);
    });
  }
  s.h = h; s.ord = ord; s.k = 0; s.tn = tn; s.a0 = a0;
  s.mflags &= ~MN_JCUR;          // a new step attempt: no refactorisation in it yet
}

template <class V>
__device__ inline void save_outputs(const TranArgs& a, V& v, StepState& s, int inst, int tid) {
  const double told = s.t, tn = s.tn, hh = tn - told;
  int si = s.si;
  // observers read u0 / u1 at arbitrary unknowns: a policy that holds them in registers writes them out first (rare:
  // once per save point)
  if (s.t_save <= tn * (1.0 + 1e-15)) v.history_to_memory(a.n, tid);
  while (s.t_save <= tn * (1.0 + 1e-15)) {
    double ts = s.t_save;
    double* o = a.out + ((size_t)inst * a.n_save + si) * a.n_obs;
    if (s.nhist >= 2) {   // quadratic through (tn,unew) (told,u0) (told-hprev,u1)
      double x = ts - told, xa = hh, xc = -s.hprev;
      double La = fast_div((x - 0.0) * (x - xc), (xa - 0.0) * (xa - xc));
      double Lb = fast_div((x - xa) * (x - xc), (0.0 - xa) * (0.0 - xc));
      double Lc = fast_div((x - xa) * (x - 0.0), (xc - xa) * (xc - 0.0));
      for (int j = tid; j < a.n_obs; j += V::NT) { int i = a.obs[j]; o[j] = La * v.get_u(i) + Lb * v.mem_u0(i) + Lc * v.mem_u1(i); }
    } else {
      double sc = fast_div(ts - told, hh);
      for (int j = tid; j < a.n_obs; j += V::NT) { int i = a.obs[j]; double x0 = v.mem_u0(i); o[j] = x0 + sc * (v.get_u(i) - x0); }
    }
    ++si;
    s.t_save = next_save(a, si);
  }
  s.si = si;
}

// One Newton update + step controller for one sweep instance, executed by one 64-lane wave.  `bad` is the
// wave-uniform "linear solve failed" flag of this round.  The caller guarantees s.status == 0.
template <class V>
__device__ inline void tran_update_body(const TranArgs& a, V& v, StepState& s, int inst, int tid, int bad) {
  const int n = a.n;
  const double h = s.h, hprev = s.hprev, hpp = s.hpp;
  double s1 = 0.0, s2 = 0.0;
  each_elem<V>(n, tid, [&](int i, int k) {
    const double x0 = v.h0(i, k), at = v.atol_of(a, i, k), upv = v.hp(i, k), em = v.emask_of(a, i, k);
    double d = v.get_delta(i, k) * s.dsc;
    double un = v.get_u(i) - d;
    if (!isfinite(d)) bad = 1;
    double w = fast_div(1.0, at + a.reltol * fabs(x0));
    s1 += (d * w) * (d * w);
    double e = un - upv;
    double w2 = fast_div(em, at + a.reltol * fmax(fabs(x0), fabs(un)));
    s2 += (e * w2) * (e * w2);
    v.set_u(i, un);
  });
  grp_reduce3<V>(v, s1, s2, bad);
  v.step_consumed(tid);          // every thread of the group has read its share of the Newton step (the fused kernels may clear its storage now)
  const double dnorm = sqrt(fast_div(s1, (double)n));
  CADNIP_TRACE_POINT(30);
  s.c_newton += 1;
  bool conv = !bad && dnorm < a.newton_tol, diverge = false;
  if (a.newton_mode) {
    conv = false;
    if (!bad) {
      if (s.k == 0) conv = dnorm <= 0.33e-4;
      else {
        const double rate = s.dnp > 0.0 ? fast_div(dnorm, s.dnp) : 0.0;
        if (rate > 0.9) diverge = true; else s.ss = fast_div(rate, 1.0 - rate);
      }
      if (!diverge && s.ss * dnorm <= 0.33) conv = true;
      s.dnp = dnorm;
    }
  }
  if (conv) {
    double errn = 0.0;
    bool accept = true;
    const bool tested = s.nhist >= 2 && a.n_err > 0;
    if (tested) {
      double errc;
      if (s.ord == 1) errc = fast_div(h, h + hprev);
      else if (s.ord == 3) errc = fast_div(fast_div(1.0, s.a0), ((h + hprev) + hpp) + s.hp3);
      else { double w = fast_div(h, hprev); errc = fast_div(fast_div((1.0 + w) * h, 1.0 + 2.0 * w), h + hprev + hpp); }
      errn = errc * sqrt(fast_div(s2, (double)a.n_err));
      accept = errn <= 1.0;
    }
    if (accept) {
      grp_sync<V>();
      save_outputs(a, v, s, inst, tid);
      if (a.max_order >= 3) each_elem<V>(n, tid, [&](int i, int k) { v.set_h3(i, k, v.h2(i, k)); });
      each_elem<V>(n, tid, [&](int i, int k) { double v1 = v.h1(i, k), v0 = v.h0(i, k); v.set_h2(i, k, v1); v.set_h1(i, k, v0); v.set_h0(i, k, v.get_u(i)); });
      const double tn = s.tn;
      bool landed = tn == s.t_break;
      const int nh_cap = a.max_order >= 3 ? 4 : 3;
      int nh_new = s.nhist + 1 > nh_cap ? nh_cap : s.nhist + 1;
      double hnext;
      if (tested) {
        double fac;
        if (a.step_rule == 0) {
          fac = errn > 0.0 ? 0.9 * step_root(errn, s.ord) : 2.0;
          fac = fmin(2.0, fmax(0.2, fac));
        } else {
          // IDA (ida.c: IDACompleteStep / IDASetEta): double the step when the estimate allows it, shrink it by 0.5 .. 0.9 when it must, keep it otherwise
          const double eta = errn > 0.0 ? fast_div(1.0, fast_div(1.0, step_root(2.0 * errn, s.ord)) + 1e-4) : 2.0;
          fac = eta >= 2.0 ? 2.0 : (eta <= 1.0 ? fmax(0.5, fmin(0.9, eta)) : 1.0);
        }
        hnext = h * fac;
      } else hnext = 2.0 * h;
      if (landed) {
        ++s.bp;
        s.t_break = next_break(a, s.bp);
        nh_new = 1;
        double tstop = a.t1;
        if (s.t_break < tstop) tstop = s.t_break;
        hnext = 0.1 * fmin(h, tstop - tn);
      }
      hnext = fmin(hnext, a.hmax);
      s.hp3 = hpp;
      s.t = tn; s.hpp = hprev; s.hprev = h; s.nhist = nh_new;
      s.c_accept += 1;
      s.mflags += 1 << MN_SINCE_SHIFT;
      CADNIP_TRACE_POINT(31);
      if (tn >= a.t1) { s.status = 1; return; }
      if (hnext < a.hmin) hnext = a.hmin;
      grp_sync<V>();
      prepare_step(a, v, s, tid, tn, hnext, nh_new, s.hprev, s.hpp);
      CADNIP_TRACE_POINT(32);
    } else {
      double fac;
      if (a.step_rule == 0) { fac = 0.9 * step_root(errn, s.ord); fac = fmin(0.9, fmax(0.1, fac)); }
      else { fac = fast_div(0.9, fast_div(1.0, step_root(2.0 * errn, s.ord)) + 1e-4); fac = fmin(0.9, fmax(0.25, fac)); }   // IDA after a failed error test
      double hn = h * fac;
      s.c_reject += 1;
      if (hn < a.hmin) { s.status = -1; return; }
      grp_sync<V>();
      prepare_step(a, v, s, tid, s.t, hn, s.nhist, hprev, hpp);
    }
  } else {
    if (a.newton_mode && (bad || diverge || s.k + 1 >= a.max_newton) && !(s.mflags & MN_JCUR)) {
      // the iteration failed on a stale Jacobian: same step again, refactored first (IDA: IDA_NLS recoverable with callSetup)
      s.mflags |= MN_NEED;
      grp_sync<V>();
      prepare_step(a, v, s, tid, s.t, h, s.nhist, hprev, hpp);
    } else if (bad || diverge || s.k + 1 >= a.max_newton) {
      double hn = 0.25 * h;
      s.c_fail += 1;
      if (hn < a.hmin) { s.status = -2; return; }
      grp_sync<V>();
      prepare_step(a, v, s, tid, s.t, hn, s.nhist, hprev, hpp);
    } else {
      grp_sync<V>();
      if (a.use_pcnr && a.n_limits > 0)
        for (int i = n - a.n_limits + tid; i < n; i += V::NT) v.set_u(i, v.get_lw(i));
      grp_sync<V>();
      const double a0 = s.a0;
      for (int i = tid; i < n; i += V::NT) v.set_du(i, a0 * v.get_u(i) + v.get_beta(i));
      s.k += 1;
    }
  }
}

}  // namespace cadnip
