// tran_ctrl.hpp -- device-side transient controller shared by the per-op driver (driver.hip) and the
// fused per-instance Newton kernel (fused.hip).  See driver.hip for the integration method.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace cadnip {

// Wave-level synchronisation: every controller function is executed by ONE 64-lane wave per sweep
// instance (several instances may share a workgroup in the fused kernel, each taking its own control
// path), so ordering is needed only among the lanes of the calling wave: a workgroup-scope fence
// (waits for the wave's outstanding LDS / global traffic; a workgroup's waves share one L1) plus a
// compiler-level wave barrier.  Never s_barrier here.
#define CADNIP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_any(int v) { return __any(v); }

// ------------------------------------------------------------------------------------------
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
};

// Set up the step that starts at (t, history) with proposed size h: clip to the next stop,
// pick order from the available history, extrapolate the predictor, BDF coefficients.
__device__ inline void prepare_step(const TranArgs& a, int inst, int tid, double t, double h, int nhist, double hprev, double hpp) {
  const int n = a.n;
  double tstop = a.t1;
  int bp = a.bp_idx[inst];
  if (bp < a.n_break && a.breaks[bp] < tstop) tstop = a.breaks[bp];
  double rem = tstop - t, tn;
  if (h >= rem * (1.0 - 1e-9)) { h = rem; tn = tstop; }
  else if (2.0 * h > rem) { h = 0.5 * rem; tn = t + h; }
  else tn = t + h;
  double* u = a.u + (size_t)inst * n;
  double* du = a.du + (size_t)inst * n;
  double* up = a.up + (size_t)inst * n;
  double* beta = a.beta + (size_t)inst * n;
  const double* u0 = a.u0 + (size_t)inst * n;
  const double* u1 = a.u1 + (size_t)inst * n;
  const double* u2 = a.u2 + (size_t)inst * n;
  int ord;
  double a0;
  if (nhist <= 1) {
    ord = 1; a0 = 1.0 / h;
    for (int i = tid; i < n; i += 64) { double p = u0[i]; up[i] = p; u[i] = p; double b = -u0[i] / h; beta[i] = b; du[i] = a0 * p + b; }
  } else if (nhist == 2 || a.max_order < 2) {
    ord = 1; a0 = 1.0 / h;
    double w = h / hprev;
    for (int i = tid; i < n; i += 64) { double p = u0[i] + w * (u0[i] - u1[i]); up[i] = p; u[i] = p; double b = -u0[i] / h; beta[i] = b; du[i] = a0 * p + b; }
  } else {
    ord = 2;
    double w = h / hprev;
    a0 = (1.0 + 2.0 * w) / ((1.0 + w) * h);
    double a1 = -(1.0 + w) / h, a2 = (w * w) / ((1.0 + w) * h);
    double x1 = -hprev, x2 = -(hprev + hpp), x = h;
    double L0 = (x - x1) * (x - x2) / ((0.0 - x1) * (0.0 - x2));
    double L1 = (x - 0.0) * (x - x2) / ((x1 - 0.0) * (x1 - x2));
    double L2 = (x - 0.0) * (x - x1) / ((x2 - 0.0) * (x2 - x1));
    for (int i = tid; i < n; i += 64) {
      double p = L0 * u0[i] + L1 * u1[i] + L2 * u2[i];
      up[i] = p; u[i] = p;
      double b = a1 * u0[i] + a2 * u1[i];
      beta[i] = b; du[i] = a0 * p + b;
    }
  }
  if (tid == 0) { a.h[inst] = h; a.order[inst] = ord; a.k[inst] = 0; a.tcur[inst] = tn; a.gamma[inst] = a0; }
}

__device__ inline void save_outputs(const TranArgs& a, int inst, int tid, double told, double tn, const double* unew, const double* u0, const double* u1,
                             int nhist_before, double hprev) {
  int si = a.save_idx[inst];
  double hh = tn - told;
  while (si < a.n_save && a.save_t[si] <= tn * (1.0 + 1e-15)) {
    double ts = a.save_t[si];
    double* o = a.out + ((size_t)inst * a.n_save + si) * a.n_obs;
    if (nhist_before >= 2) {   // quadratic through (tn,unew) (told,u0) (told-hprev,u1)
      double x = ts - told, xa = hh, xc = -hprev;
      double La = (x - 0.0) * (x - xc) / ((xa - 0.0) * (xa - xc));
      double Lb = (x - xa) * (x - xc) / ((0.0 - xa) * (0.0 - xc));
      double Lc = (x - xa) * (x - 0.0) / ((xc - xa) * (xc - 0.0));
      for (int j = tid; j < a.n_obs; j += 64) { int i = a.obs[j]; o[j] = La * unew[i] + Lb * u0[i] + Lc * u1[i]; }
    } else {
      double s = (ts - told) / hh;
      for (int j = tid; j < a.n_obs; j += 64) { int i = a.obs[j]; o[j] = u0[i] + s * (unew[i] - u0[i]); }
    }
    ++si;
  }
  if (tid == 0) a.save_idx[inst] = si;
}

// One Newton update + step controller for sweep instance `inst`, executed by one 64-lane wave.
__device__ inline void tran_update_body(const TranArgs& a, int inst, int tid) {
  const int n = a.n;
  if (a.status[inst] != 0) return;
  double* u = a.u + (size_t)inst * n;
  double* du = a.du + (size_t)inst * n;
  const double* delta = a.delta + (size_t)inst * n;
  double* u0 = a.u0 + (size_t)inst * n;
  double* u1 = a.u1 + (size_t)inst * n;
  double* u2 = a.u2 + (size_t)inst * n;
  const double* up = a.up + (size_t)inst * n;
  const double* beta = a.beta + (size_t)inst * n;
  const double t = a.t[inst], h = a.h[inst], hprev = a.hprev[inst], hpp = a.hpp[inst], tn = a.tcur[inst], a0 = a.gamma[inst];
  const int nhist = a.nhist[inst], ord = a.order[inst], k = a.k[inst];
  int bad = (a.flags[inst] & 1);
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < n; i += 64) {
    double d = delta[i];
    double un = u[i] - d;
    if (!isfinite(d)) bad = 1;
    double w = 1.0 / (a.atol[i] + a.reltol * fabs(u0[i]));
    s1 += (d * w) * (d * w);
    double e = un - up[i];
    double w2 = a.emask[i] / (a.atol[i] + a.reltol * fmax(fabs(u0[i]), fabs(un)));
    s2 += (e * w2) * (e * w2);
    u[i] = un;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  bad = wave_any(bad);
  const double dnorm = sqrt(s1 / n);
  if (tid == 0) { a.cnt[(size_t)inst * 4 + 0] += 1; a.flags[inst] = 0; }
  const bool conv = !bad && dnorm < a.newton_tol;
  if (conv) {
    double errn = 0.0;
    bool accept = true;
    if (nhist >= 2 && a.n_err > 0) {
      double errc;
      if (ord == 1) errc = h / (h + hprev);
      else { double w = h / hprev; errc = ((1.0 + w) * h / (1.0 + 2.0 * w)) / (h + hprev + hpp); }
      errn = errc * sqrt(s2 / a.n_err);
      accept = errn <= 1.0;
    }
    if (accept) {
      save_outputs(a, inst, tid, t, tn, u, u0, u1, nhist, hprev);
      for (int i = tid; i < n; i += 64) { double v1 = u1[i], v0 = u0[i]; u2[i] = v1; u1[i] = v0; u0[i] = u[i]; }
      int bp = a.bp_idx[inst];
      bool landed = (bp < a.n_break && tn == a.breaks[bp]);
      int nh_new = nhist + 1 > 3 ? 3 : nhist + 1;
      double hnext;
      if (nhist >= 2 && a.n_err > 0) {
        double fac = errn > 0.0 ? 0.9 * pow(errn, -1.0 / (ord + 1)) : 2.0;
        fac = fmin(2.0, fmax(0.2, fac));
        hnext = h * fac;
      } else hnext = 2.0 * h;
      double new_hprev = h, new_hpp = hprev;
      CADNIP_WAVE_SYNC();
      if (landed) {
        ++bp;
        nh_new = 1;
        double tstop = a.t1;
        if (bp < a.n_break && a.breaks[bp] < tstop) tstop = a.breaks[bp];
        hnext = 0.1 * fmin(h, tstop - tn);
      }
      hnext = fmin(hnext, a.hmax);
      if (tid == 0) {
        a.t[inst] = tn; a.hprev[inst] = new_hprev; a.hpp[inst] = new_hpp; a.nhist[inst] = nh_new; a.bp_idx[inst] = bp;
        a.cnt[(size_t)inst * 4 + 1] += 1;
      }
      if (tn >= a.t1) {
        if (tid == 0) { a.status[inst] = 1; a.active[inst] = 0; }
        return;
      }
      if (hnext < a.hmin) hnext = a.hmin;
      CADNIP_WAVE_SYNC();
      prepare_step(a, inst, tid, tn, hnext, nh_new, new_hprev, new_hpp);
    } else {
      double fac = 0.9 * pow(errn, -1.0 / (ord + 1));
      fac = fmin(0.9, fmax(0.1, fac));
      double hn = h * fac;
      if (tid == 0) a.cnt[(size_t)inst * 4 + 2] += 1;
      if (hn < a.hmin) { if (tid == 0) { a.status[inst] = -1; a.active[inst] = 0; } return; }
      CADNIP_WAVE_SYNC();
      prepare_step(a, inst, tid, t, hn, nhist, hprev, hpp);
    }
  } else {
    if (bad || k + 1 >= a.max_newton) {
      double hn = 0.25 * h;
      if (tid == 0) a.cnt[(size_t)inst * 4 + 3] += 1;
      if (hn < a.hmin) { if (tid == 0) { a.status[inst] = -2; a.active[inst] = 0; } return; }
      CADNIP_WAVE_SYNC();
      prepare_step(a, inst, tid, t, hn, nhist, hprev, hpp);
    } else {
      if (a.use_pcnr && a.n_limits > 0) {
        const double* lw = a.limit_w + (size_t)inst * n;
        for (int i = n - a.n_limits + tid; i < n; i += 64) u[i] = lw[i];
      }
      CADNIP_WAVE_SYNC();
      for (int i = tid; i < n; i += 64) du[i] = a0 * u[i] + beta[i];
      if (tid == 0) a.k[inst] = k + 1;
    }
  }
}


}  // namespace cadnip
