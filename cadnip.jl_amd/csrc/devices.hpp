// devices.hpp -- per-device-type stamp functions for gfx950 (device code only).
//
// One function per reference `stamp!` method; each is called by one GPU kernel per device
// type (kernels.hip: k_stamp<T>) and by the fused per-instance Newton kernel.  A stamp
// function reads the unknowns it needs from `u` (one sweep instance's solution vector),
// its parameters from the SoA parameter block, and writes the values of its fixed G / C / b
// slots; slot (k, dev) lives at S[base + k*count + dev] so that consecutive device lanes
// write consecutive addresses.  No branch changes which slots are written: the stamp
// sequence of the reference is branch-independent by construction
// (/root/reference/src/mna/value_only.jl:395-421, src/vasim.jl:1984-2134).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/cadnip_hip.h"

namespace cadnip {

#define CADNIP_CHARGE_SCALE 1e12   // /root/reference/src/mna/contrib.jl:39

// Diagnostic build only (build.sh --trace -> libcadnip_hip_trace.so): cycle timeline of wave 0 of workgroup 0.
// Each point adds the cycles since the previous point to its own bucket; read back with cadnip_debug_trace.
#ifdef CADNIP_TRACE
static __device__ unsigned long long g_trace_sum[64], g_trace_cnt[64];
static __device__ int g_trace_wave;   // the traced wave of workgroup 0 (team kernel: one role per wave)
__shared__ unsigned long long g_trace_last;
// -DCADNIP_TRACE=1: the phase boundaries of a Newton round only (a point costs ~250 cycles: the fine points inside the device functions and
// the update shift what they measure); -DCADNIP_TRACE=2: every point.
#define CADNIP_TRACE_FINE(id) (((id) >= 8 && (id) <= 15) || (id) >= 20 || (id) == 0 || (id) == 2)
#define CADNIP_TRACE_POINT(id)                                                                          \
  do {                                                                                                  \
    if ((CADNIP_TRACE + 0 >= 2 || !CADNIP_TRACE_FINE(id)) && blockIdx.x == 0 && (int)(threadIdx.x >> 6) == g_trace_wave) { \
      unsigned long long _t = clock64();                                                                \
      if ((threadIdx.x & 63) == 0) {                                                                    \
        atomicAdd(&g_trace_sum[id], _t - g_trace_last); atomicAdd(&g_trace_cnt[id], 1ull); g_trace_last = _t; \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#elif defined(CADNIP_MARKS)
// analysis build (tools/kernel_regs.sh ... -DCADNIP_MARKS): the trace points become comments in the assembly, so that the instructions of a
// phase can be counted statically (tools/phase_instr.py)
#define CADNIP_TRACE_POINT(id) asm volatile("; @@MARK " #id)
#else
#define CADNIP_TRACE_POINT(id) do {} while (0)
#endif

// swap a double between lanes 2j and 2j+1 (DPP quad_perm [1,0,3,2]); all lanes of the wave must be active
__device__ __forceinline__ double dpp_pair_swap(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// NodeT: int in HBM (per-op kernels) or int16 in LDS (fused kernel); -1 = ground either way
template <class NodeT> struct DevCtxT {
  const NodeT* __restrict__ nodes; // [n_nodes][count]
  const int* __restrict__ ipar;    // [n_ipar][count]
  const double* __restrict__ par;  // this instance: [n_par][count]
  const double* __restrict__ wave; // shared wave tables
  int count;
  int dev;
  double t;
  int mode;                        // 0 dcop, 1 tran, 2 tranop
  int initjct;
  const double* __restrict__ cache = nullptr;   // generated external models: [n_cache][count] values of this instance's setup pass (va_generated_ext.hpp)
};
typedef DevCtxT<int> DevCtx;

// slot writers: S points at this instance's slot buffer; bases are per block
struct SlotOut {
  // DIRECT writers (fused kernel) take each device's residual contribution r = C du + G u - b straight from the device
  // (Rn, by unknown index) instead of assembling it from b, C*beta and a J*u product; the slot writer never does.
  static constexpr bool DIRECT = false;
  __device__ __forceinline__ void Rn(int, double) const {}
  __device__ __forceinline__ double du(int) const { return 0.0; }
  double* __restrict__ g;  // S + g_base
  double* __restrict__ c;  // S + ns_g + c_base
  double* __restrict__ b;  // S + ns_g + ns_c + b_base
  int count, dev;
  __device__ __forceinline__ void G(int k, double v) const { g[k * count + dev] = v; }
  __device__ __forceinline__ void C(int k, double v) const { c[k * count + dev] = v; }
  __device__ __forceinline__ void B(int k, double v) const { b[k * count + dev] = v; }
  // batch forms: slots k0 .. k0+N-1 (the fused kernel's writer pipelines its table reads over a batch)
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

template <class Ctx> __device__ __forceinline__ int node_of(const Ctx& d, int k) { return d.nodes[k * d.count + d.dev]; }
template <class Ctx> __device__ __forceinline__ double par_of(const Ctx& d, int k) { return d.par[k * d.count + d.dev]; }
template <class Ctx> __device__ __forceinline__ double va_cache(const Ctx& d, int k) { return d.cache[k * d.count + d.dev]; }
__device__ __forceinline__ double volt(const double* u, int node) { return node < 0 ? 0.0 : u[node]; }

// ------------------------------------------------------------------------------------------
// source waves (devices.jl:30-103, 155-203)
// ------------------------------------------------------------------------------------------
// `seg` (optional): the segment index i found by the previous call.  When t lies strictly inside that segment again
// (time mostly advances in small steps) the search -- a chain of dependent memory reads -- is skipped; same result.
__device__ inline double pwl_at_time(const double* ts, const double* ys, int n, double t, int* seg = nullptr) {
  if (seg) {
    const int i = *seg;
    if (i >= 2 && i <= n) {
      const double t_lo = ts[i - 2], t_hi = ts[i - 1], y_lo = ys[i - 2], y_hi = ys[i - 1];
      if (t_lo < t && t < t_hi) {
        if (y_lo == y_hi) return y_hi;
        return y_lo + (t - t_lo) * ((y_hi - y_lo) / (t_hi - t_lo));
      }
    }
  }
  // find_t_in_ts: searchsortedfirst, +1 on an exact hit (devices.jl:30-36); i is 1-based
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (ts[mid] < t) lo = mid + 1; else hi = mid; }
  int i = lo + 1;
  if (i <= n && ts[i - 1] == t) i += 1;
  if (seg) *seg = i;
  if (i <= 1) return ys[0];
  if (i > n) return ys[n - 1];
  if (ys[i - 2] == ys[i - 1]) return ys[i - 1];
  if (ts[i - 1] == ts[i - 2]) return (ys[i - 2] + ys[i - 1]) / 2;
  double slope = (ys[i - 1] - ys[i - 2]) / (ts[i - 1] - ts[i - 2]);
  return ys[i - 2] + (t - ts[i - 2]) * slope;
}

__device__ inline double pulse_at_time(double v1, double v2, double td, double tr, double tf, double pw, double per, double t) {
  if (t < td) return v1;
  double phase;
  if (per > 0) { phase = fmod(t - td, per); if (phase < 0) phase += per; } else phase = t - td;
  if (phase < tr) return tr > 0 ? v1 + (v2 - v1) * (phase / tr) : v2;
  else if (phase < tr + pw) return v2;
  else if (phase < tr + pw + tf) return tf > 0 ? v2 + (v1 - v2) * ((phase - tr - pw) / tf) : v1;
  return v1;
}

__device__ inline double sind_deg(double deg) {
  double r = fmod(deg, 360.0);
  if (r == 0.0 || r == 180.0 || r == -180.0) return 0.0;
  if (r == 90.0 || r == -270.0) return 1.0;
  if (r == -90.0 || r == 270.0) return -1.0;
  return sin(r * (3.14159265358979323846 / 180.0));
}

// get_source_value (devices.jl:352-360): :dcop -> dc, otherwise tran(t)
template <class Ctx> __device__ inline double source_value(const Ctx& d, double dc, double scale, int* seg = nullptr) {
  int kind = d.ipar[0 * d.count + d.dev];
  if (kind == 0 || d.mode == 0) return dc;
  int off = d.ipar[1 * d.count + d.dev];
  int len = d.ipar[2 * d.count + d.dev];
  const double* w = d.wave + off;
  double v;
  if (kind == 1) v = pwl_at_time(w, w + len, len, d.t, seg);
  else if (kind == 2) v = pulse_at_time(w[0], w[1], w[2], w[3], w[4], w[5], w[6], d.t);
  else {
    double vo = w[0], va = w[1], freq = w[2], td = w[3], theta = w[4], phase = w[5];
    if (d.t < td) v = vo + va * sind_deg(phase);
    else v = vo + va * exp(-theta * (d.t - td)) * sind_deg(360 * freq * (d.t - td) + phase);
  }
  return scale * v;
}

// ------------------------------------------------------------------------------------------
// linear devices
// ------------------------------------------------------------------------------------------
template <class Out> __device__ inline void conductance4(const Out& s, int k0, double g) { const double v[4] = {g, -g, -g, g}; s.Gv(k0, v); }
template <class Out> __device__ inline void capacitance4(const Out& s, int k0, double c) { const double v[4] = {c, -c, -c, c}; s.Cv(k0, v); }
template <class Out> __device__ inline void branch4(const Out& s) { const double v[4] = {1.0, -1.0, 1.0, -1.0}; s.Gv(0, v); }

// residual of a two-terminal branch current i flowing p -> n
template <class Out> __device__ __forceinline__ void residual2(const Out& s, int p, int n, double i) { s.Rn(p, i); s.Rn(n, -i); }

template <class Ctx, class Out> __device__ inline void stamp_resistor(const Ctx& d, const double* u, const Out& s, double*) {
  const double g = par_of(d, 0);
  conductance4(s, 0, g);
  if constexpr (Out::DIRECT) { const int p = node_of(d, 0), n = node_of(d, 1); residual2(s, p, n, g * (volt(u, p) - volt(u, n))); }
}
template <class Ctx, class Out> __device__ inline void stamp_capacitor(const Ctx& d, const double*, const Out& s, double*) {
  const double c = par_of(d, 0);
  capacitance4(s, 0, c);
  if constexpr (Out::DIRECT) { const int p = node_of(d, 0), n = node_of(d, 1); residual2(s, p, n, c * (s.du(p) - s.du(n))); }
}
// branch element with its own current unknown I between p and n: KCL rows carry u[I]; `vbranch` is what the branch row
// equates V(p) - V(n) to
template <class Out> __device__ __forceinline__ void residual_branch(const Out& s, const double* u, int p, int n, int I, double vbranch) {
  residual2(s, p, n, u[I]);
  s.Rn(I, volt(u, p) - volt(u, n) - vbranch);
}
template <class Ctx, class Out> __device__ inline void stamp_inductor(const Ctx& d, const double* u, const Out& s, double*) {
  const double L = par_of(d, 0);
  branch4(s); s.C(0, -L);
  if constexpr (Out::DIRECT) { const int I = node_of(d, 2); residual_branch(s, u, node_of(d, 0), node_of(d, 1), I, L * s.du(I)); }
}
// independent sources, split into value (a function of time only) and stamp so that a caller may keep the value
// across the Newton rounds of one time point
template <class Ctx, class Out> __device__ inline void stamp_vsource_value(const Ctx& d, const double* u, const Out& s, double v) {
  branch4(s);
  s.B(0, v);
  if constexpr (Out::DIRECT) residual_branch(s, u, node_of(d, 0), node_of(d, 1), node_of(d, 2), v);
}
template <class Ctx, class Out> __device__ inline void stamp_isource_value(const Ctx& d, const Out& s, double i) {
  s.B(0, i); s.B(1, -i);
  if constexpr (Out::DIRECT) residual2(s, node_of(d, 0), node_of(d, 1), -i);
}
template <class Ctx, class Out> __device__ inline void stamp_vsource(const Ctx& d, const double* u, const Out& s, double*) {
  stamp_vsource_value(d, u, s, source_value(d, par_of(d, 0), par_of(d, 1)));
}
template <class Ctx, class Out> __device__ inline void stamp_isource(const Ctx& d, const double*, const Out& s, double*) {
  stamp_isource_value(d, s, source_value(d, par_of(d, 0), par_of(d, 1)));
}
template <class Ctx, class Out> __device__ inline void stamp_vcvs(const Ctx& d, const double* u, const Out& s, double*) {
  double a = par_of(d, 0);
  s.G(0, 1.0); s.G(1, -1.0); s.G(2, 1.0); s.G(3, -1.0); s.G(4, -a); s.G(5, a);
  if constexpr (Out::DIRECT)      // nodes op on ip in I: V(op) - V(on) = a (V(ip) - V(in))
    residual_branch(s, u, node_of(d, 0), node_of(d, 1), node_of(d, 4), a * (volt(u, node_of(d, 2)) - volt(u, node_of(d, 3))));
}
template <class Ctx, class Out> __device__ inline void stamp_vccs(const Ctx& d, const double* u, const Out& s, double*) {
  double gm = par_of(d, 0);
  s.G(0, -gm); s.G(1, gm); s.G(2, gm); s.G(3, -gm);
  if constexpr (Out::DIRECT) residual2(s, node_of(d, 0), node_of(d, 1), -gm * (volt(u, node_of(d, 2)) - volt(u, node_of(d, 3))));
}
template <class Ctx, class Out> __device__ inline void stamp_ccvs(const Ctx& d, const double* u, const Out& s, double*) {
  const double rm = par_of(d, 0);
  s.G(0, 1.0); s.G(1, -1.0); s.G(2, 1.0); s.G(3, -1.0);
  s.G(4, 1.0); s.G(5, -1.0); s.G(6, 1.0); s.G(7, -1.0); s.G(8, -rm);
  if constexpr (Out::DIRECT) {   // nodes op on ip in Iin Iout: 0 V sense branch ip -> in, output branch V(op) - V(on) = rm I_in
    const int Iin = node_of(d, 4);
    residual_branch(s, u, node_of(d, 2), node_of(d, 3), Iin, 0.0);
    residual_branch(s, u, node_of(d, 0), node_of(d, 1), node_of(d, 5), rm * u[Iin]);
  }
}
template <class Ctx, class Out> __device__ inline void stamp_cccs(const Ctx& d, const double* u, const Out& s, double*) {
  double a = par_of(d, 0);
  s.G(0, 1.0); s.G(1, -1.0); s.G(2, 1.0); s.G(3, -1.0); s.G(4, -a); s.G(5, a);
  if constexpr (Out::DIRECT) {   // nodes op on ip in Iin: sense branch ip -> in; G(op,Iin) = -a, G(on,Iin) = +a
    const int Iin = node_of(d, 4);
    residual_branch(s, u, node_of(d, 2), node_of(d, 3), Iin, 0.0);
    residual2(s, node_of(d, 0), node_of(d, 1), -a * u[Iin]);
  }
}

// ------------------------------------------------------------------------------------------
// behavioural sources (devices.jl:1079-1131): value_fn(get_voltage) as a postfix program (include/cadnip_hip.h)
// ------------------------------------------------------------------------------------------
template <class Ctx> __device__ inline double bsrc_value(const Ctx& d, const double* u) {
  const int off = d.ipar[0 * d.count + d.dev], len = d.ipar[1 * d.count + d.dev];
  const double* pr = d.wave + off;
  double st[CADNIP_BSRC_MAX_STACK];
  int sp = 0;
  for (int i = 0; i < len;) {
    const int op = (int)pr[i++];
    if (op == CADNIP_BOP_CONST) st[sp++] = pr[i++];
    else if (op == CADNIP_BOP_V) { const int a = (int)pr[i], b = (int)pr[i + 1]; i += 2; st[sp++] = volt(u, a) - volt(u, b); }
    else if (op == CADNIP_BOP_TIME) st[sp++] = d.t;
    else if (op < CADNIP_BOP_NEG) {
      const double y = st[--sp], x = st[sp - 1];
      double r;
      switch (op) {
        case CADNIP_BOP_ADD: r = x + y; break;
        case CADNIP_BOP_SUB: r = x - y; break;
        case CADNIP_BOP_MUL: r = x * y; break;
        case CADNIP_BOP_DIV: r = x / y; break;
        case CADNIP_BOP_POW: r = pow(x, y); break;
        case CADNIP_BOP_MIN: r = fmin(x, y); break;
        default: r = fmax(x, y); break;
      }
      st[sp - 1] = r;
    } else {
      const double x = st[sp - 1];
      double r;
      switch (op) {
        case CADNIP_BOP_NEG: r = -x; break;
        case CADNIP_BOP_EXP: r = exp(x); break;
        case CADNIP_BOP_LOG: r = log(x); break;
        case CADNIP_BOP_SQRT: r = sqrt(x); break;
        case CADNIP_BOP_ABS: r = fabs(x); break;
        case CADNIP_BOP_TANH: r = tanh(x); break;
        case CADNIP_BOP_SIN: r = sin(x); break;
        default: r = cos(x); break;
      }
      st[sp - 1] = r;
    }
  }
  return par_of(d, 0) * st[0];
}
template <class Ctx, class Out> __device__ inline void stamp_bvsource(const Ctx& d, const double* u, const Out& s, double*) {
  branch4(s);
  const double v = bsrc_value(d, u);
  s.B(0, v);
  if constexpr (Out::DIRECT) residual_branch(s, u, node_of(d, 0), node_of(d, 1), node_of(d, 2), v);
}
template <class Ctx, class Out> __device__ inline void stamp_bisource(const Ctx& d, const double* u, const Out& s, double*) {
  const double i = bsrc_value(d, u);
  s.B(0, i); s.B(1, -i);
  if constexpr (Out::DIRECT) residual2(s, node_of(d, 0), node_of(d, 1), -i);
}

// ------------------------------------------------------------------------------------------
// diode (devices.jl:1169-1189 pnjlim, :1209-1234 limit!, :1333-1345 _diode_iv, :1370-1428)
// ------------------------------------------------------------------------------------------
__device__ inline double pnjlim(double vnew, double vold, double vt, double vcrit) {
  if (vnew > vcrit && fabs(vnew - vold) > vt + vt) {
    if (vold > 0.0) {
      double arg = (vnew - vold) / vt;
      if (arg > 0.0) return vold + vt * (2.0 + log(arg - 2.0));
      return vold - vt * (2.0 + log(2.0 - arg));
    }
    return vt * log(vnew / vt);
  } else if (vnew < 0.0) {
    double arg = vold > 0.0 ? -vold - 1.0 : 2.0 * vold - 1.0;
    if (vnew < arg) return arg;
  }
  return vnew;
}

template <class Ctx, class Out> __device__ inline void stamp_diode(const Ctx& d, const double* u, const Out& s, double* limit_w_base) {
  int p = node_of(d, 0), n = node_of(d, 1), l = node_of(d, 2);
  double Is = par_of(d, 0), nVt = par_of(d, 1), vcrit = par_of(d, 2);
  double V0 = volt(u, p) - volt(u, n);
  double I0, Gd, Ieq;
  if (d.ipar[d.dev]) {
    double vold = u[l];
    double w = d.initjct ? vcrit : pnjlim(V0, vold, nVt, vcrit);
    if (limit_w_base) limit_w_base[l] = w;     // record_limit_w! (value_only.jl:384); null = nobody reads it (fused transient without PCNR)
    s.G(0, 1.0); s.G(1, -1.0); s.G(2, 1.0);    // g_lim row (devices.jl:1228-1231)
    double xarg = w / nVt;
    if (xarg > 80.0) { double e80 = exp(80.0); I0 = Is * (e80 * (1.0 + (xarg - 80.0)) - 1.0); Gd = Is / nVt * e80; }
    else { double e = exp(xarg); I0 = Is * (e - 1.0); Gd = Is / nVt * e; }
    Ieq = I0 - Gd * w;                         // anchored at w (devices.jl:1251-1258)
    if constexpr (Out::DIRECT) s.Rn(l, vold - V0);   // limit row: u_l - (V(p) - V(n))
  } else {
    s.G(0, 0.0); s.G(1, 0.0); s.G(2, 0.0);
    double e = exp(V0 / nVt);
    I0 = Is * (e - 1.0); Gd = Is / nVt * e;
    Ieq = I0 - Gd * V0;
  }
  conductance4(s, 3, Gd);
  s.B(0, -Ieq); s.B(1, Ieq);
  if constexpr (Out::DIRECT) residual2(s, p, n, Gd * V0 + Ieq);   // = I0 + Gd (V0 - w): the companion model's current at V0
}

template <class Ctx, class Out> __device__ inline void stamp_diodecap(const Ctx& d, const double* u, const Out& s, double*) {
  int p = node_of(d, 0), n = node_of(d, 1);
  double Is = par_of(d, 0), nVt = par_of(d, 1), Cj0 = par_of(d, 2), Vj = par_of(d, 3), m = par_of(d, 4);
  double V0 = volt(u, p) - volt(u, n);
  double e = exp(V0 / nVt);
  double I0 = Is * (e - 1.0), G = Is / nVt * e, Ieq = I0 - G * V0;
  conductance4(s, 0, G);
  s.B(0, -Ieq); s.B(1, Ieq);
  double Vmax = 0.9 * Vj, C;              // diode_junction_cap devices.jl:1505-1516
  if (V0 < Vmax) C = Cj0 / pow(1 - V0 / Vj, m);
  else { double Ca = Cj0 / pow(1 - Vmax / Vj, m); double dC = Cj0 * m / Vj / pow(1 - Vmax / Vj, m + 1); C = Ca + dC * (V0 - Vmax); }
  capacitance4(s, 0, C);
  if constexpr (Out::DIRECT) residual2(s, p, n, I0 + C * (s.du(p) - s.du(n)));
}

// SimpleMOSFET (devices.jl:1667-1749)
template <class Ctx, class Out> __device__ inline void stamp_simplemos(const Ctx& d, const double* u, const Out& s, double*) {
  double Vd = volt(u, node_of(d, 0)), Vg = volt(u, node_of(d, 1)), Vs = volt(u, node_of(d, 2));
  double Vth = par_of(d, 0), K = par_of(d, 1), lam = par_of(d, 2), Cgd = par_of(d, 3), Cgs = par_of(d, 4);
  double Vgs = Vg - Vs, Vds = Vd - Vs, Ids, gm, gds;
  if (Vgs <= Vth) { Ids = 0; gm = 0; gds = 0; }
  else if (Vds <= Vgs - Vth) { Ids = K * ((Vgs - Vth) * Vds - Vds * Vds / 2); gm = K * Vds; gds = K * (Vgs - Vth - Vds); }
  else { double vov = Vgs - Vth; Ids = K / 2 * (vov * vov) * (1 + lam * Vds); gm = K * vov * (1 + lam * Vds); gds = K / 2 * (vov * vov) * lam; }
  double Ieq = Ids - gm * Vgs - gds * Vds;
  s.G(0, gds); s.G(1, gm); s.G(2, -(gds + gm)); s.G(3, -gds); s.G(4, -gm); s.G(5, gds + gm);
  s.B(0, -Ieq); s.B(1, Ieq);
  capacitance4(s, 0, Cgs);
  capacitance4(s, 4, Cgd);
  if constexpr (Out::DIRECT) {   // channel current d -> s, capacitor currents g -> s and g -> d
    const int nd = node_of(d, 0), ng = node_of(d, 1), ns = node_of(d, 2);
    residual2(s, nd, ns, Ids);
    residual2(s, ng, ns, Cgs * (s.du(ng) - s.du(ns)));
    residual2(s, ng, nd, Cgd * (s.du(ng) - s.du(nd)));
  }
}

// ------------------------------------------------------------------------------------------
// forward-mode dual numbers (the JacobianTag dual of contrib.jl:54-101, restricted to the N
// independent voltages a model actually depends on)
// ------------------------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double p[N];
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(double x) : v(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = 0.0;
  }
  __device__ __forceinline__ static Dual seed(double x, int k) { Dual r(x); r.p[k] = 1.0; return r; }
};
#define DUAL_BIN(op, vexpr, pexpr)                                                        \
  template <int N> __device__ __forceinline__ Dual<N> operator op(const Dual<N>& a, const Dual<N>& b) { \
    Dual<N> r; r.v = vexpr;                                                                \
    _Pragma("unroll") for (int i = 0; i < N; ++i) r.p[i] = pexpr;                          \
    return r; }
DUAL_BIN(+, a.v + b.v, a.p[i] + b.p[i])
DUAL_BIN(-, a.v - b.v, a.p[i] - b.p[i])
DUAL_BIN(*, a.v * b.v, a.p[i] * b.v + b.p[i] * a.v)
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; double ib = 1.0 / b.v, q = a.v * ib; r.v = q;   // one division per dual quotient
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = (a.p[i] - q * b.p[i]) * ib;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(double b, const Dual<N>& a) {
  Dual<N> r; r.v = b - a.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = -a.p[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a) { return 0.0 - a; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, double b) {
  Dual<N> r; r.v = a.v * b;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * b;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, double b) {
  Dual<N> r; double ib = 1.0 / b; r.v = a.v * ib;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * ib;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator/(double b, const Dual<N>& a) {
  Dual<N> r; double ia = 1.0 / a.v, q = b * ia; r.v = q;
  const double f = -q * ia;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = f * a.p[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> dsqrt(const Dual<N>& a) {
  Dual<N> r; double s = sqrt(a.v); r.v = s; double f = 0.5 / s;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * f;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> dexp(const Dual<N>& a) {
  Dual<N> r; double e = exp(a.v); r.v = e;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * e;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> dlog(const Dual<N>& a) {
  Dual<N> r; r.v = log(a.v); const double ia = 1.0 / a.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * ia;
  return r;
}

// ------------------------------------------------------------------------------------------
// sp_mos1 (models/VADistillerModels.jl/va/mos1.va) as Cadnip stamps it (vasim.jl:3319-3521)
//
// Every node-voltage dependence of the load section enters through the three limited,
// type-adjusted branch voltages vgs, vds, vbs returned by the `$limit ... DEVlimitNewSet`
// sites (mos1.va:976-980); their duals are pass-through (vasim.jl:1319-1330: value = w,
// d/dV_p = +1, d/dV_n = -1, unit partial in the site's own slot), and every partial produced
// by the limiter functions themselves is discarded there.  So a Dual<3> over (vgs, vds, vbs)
// carries exactly the information of the reference's width-14 JacobianTag dual:
//   d/dV_g = type*da, d/dV_dint = type*db, d/dV_b = type*dc, d/dV_sint = -type*(da+db+dc),
//   d/dW(site vgs,vds,vbs) = type*(da,db,dc)  ->  lim_rhs terms (vasim.jl:2957-2966).
// ------------------------------------------------------------------------------------------
enum { M1_TYPE = 0, M1_VT, M1_TPHI, M1_TVBI, M1_TVTO, M1_GAMMA, M1_LAMBDA, M1_BETA, M1_OXCAP, M1_SSATCUR, M1_DSATCUR,
       M1_SVCRIT, M1_DVCRIT, M1_CBS, M1_CBSSW, M1_CBD, M1_CBDSW, M1_TBULKPOT, M1_TDEPCAP, M1_F2S, M1_F3S, M1_F4S,
       M1_F2D, M1_F3D, M1_F4D, M1_MJ, M1_MJSW, M1_CGSOV, M1_CGDOV, M1_CGBOV, M1_GD, M1_GS, M1_MFACTOR, M1_GMIN };
static_assert(M1_GMIN == 33 && M1_OXCAP == 8 && M1_GD == 30 && M1_GS == 31, "keep CADNIP_MOS1_PAR_* (internal.hpp) in step");

__device__ inline double m1_fetlim(double vnew, double vold, double vto) {   // DEVfetlim mos1.va:542-605
  double vlimited = vnew;
  double vtsthi = fabs(2 * (vold - vto)) + 2;
  double vtstlo = fabs(vold - vto) + 1;
  double vtox = vto + 3.5;
  double delv = vnew - vold;
  if (vold >= vto) {
    if (vold >= vtox) {
      if (delv <= 0) {
        if (vlimited >= vtox) { if (-delv > vtstlo) vlimited = vold - vtstlo; }
        else vlimited = fmax(vnew, vto + 2);
      } else { if (delv >= vtsthi) vlimited = vold + vtsthi; }
    } else {
      if (delv <= 0) vlimited = fmax(vnew, vto - 0.5);
      else vlimited = fmin(vnew, vto + 4);
    }
  } else {
    if (delv <= 0) { if (-delv > vtsthi) vlimited = vold - vtsthi; }
    else {
      double vtemp = vto + 0.5;
      if (vnew <= vtemp) { if (delv > vtstlo) vlimited = vold + vtstlo; }
      else vlimited = vtemp;
    }
  }
  return vlimited;
}
__device__ inline double m1_limvds(double vnew, double vold) {               // DEVlimvds mos1.va:607-635
  if (vold >= 3.5) {
    if (vnew > vold) return fmin(vnew, (3 * vold) + 2);
    if (vnew < 3.5) return fmax(vnew, 2.0);
    return vnew;
  }
  if (vnew > vold) return fmin(vnew, 4.0);
  return fmax(vnew, -0.5);
}
__device__ inline double m1_pnjlim(double vnew, double vold, double vt, double vcrit) {   // DEVpnjlim mos1.va:503-540
  double limited = vnew;
  if ((vnew > vcrit) && (fabs(vnew - vold) > (vt + vt))) {
    if (vold > 0) {
      double arg = (vnew - vold) / vt;
      if (arg > 0) limited = vold + vt * log(1 + arg);
      else limited = vold - vt * log(1 - arg);
    } else limited = vt * log(vnew / vt);
  } else if (vnew < 0) {
    double arg = vold > 0 ? -1 * vold - 1 : 2 * vold - 1;
    if (vnew < arg) limited = arg;
  }
  return limited;
}

typedef Dual<3> D3;

// DEVqmeyer (mos1.va:401-465): Meyer capacitances, evaluated on duals
__device__ inline void m1_qmeyer(const D3& vgs, const D3& vgd, const D3& von, const D3& vdsat_in, double phi, double cox,
                                 D3& capgs, D3& capgd, D3& capgb) {
  D3 vdsat = vdsat_in;
  D3 vgst = vgs - von;
  if (!(vdsat.v > 0.025)) vdsat = D3(0.025);
  if (vgst.v <= -phi) { capgb = D3(cox / 2); capgs = D3(0.0); capgd = D3(0.0); }
  else if (vgst.v <= -phi / 2) { capgb = -1.0 * vgst * cox / (2 * phi); capgs = D3(0.0); capgd = D3(0.0); }
  else if (vgst.v <= 0) {
    capgb = -1.0 * vgst * cox / (2 * phi);
    capgs = vgst * cox / (1.5 * phi) + cox / 3;
    D3 vds = vgs - vgd;
    if (vds.v >= vdsat.v) capgd = D3(0.0);
    else {
      D3 vddif = 2.0 * vdsat - vds, vddif1 = vdsat - vds, vddif2 = vddif * vddif;
      capgd = capgs * (1.0 - vdsat * vdsat / vddif2);
      capgs = capgs * (1.0 - vddif1 * vddif1 / vddif2);
    }
  } else {
    D3 vds = vgs - vgd;
    if (vdsat.v <= vds.v) { capgs = D3(cox / 3); capgd = D3(0.0); capgb = D3(0.0); }
    else {
      D3 vddif = 2.0 * vdsat - vds, vddif1 = vdsat - vds, vddif2 = vddif * vddif;
      capgd = cox * (1.0 - vdsat * vdsat / vddif2) / 3;
      capgs = cox * (1.0 - vddif1 * vddif1 / vddif2) / 3;
      capgb = D3(0.0);
    }
  }
}

// depletion charge (mos1.va:1049-1109)
__device__ inline D3 m1_qdep(const D3& v, double Cb, double Cbsw, double tBulkPot, double tDepCap, double mj, double mjsw,
                             double f2, double f3, double f4) {
  if (Cb != 0 || Cbsw != 0) {
    if (v.v < tDepCap) {
      D3 arg = 1.0 - v / tBulkPot;
      D3 sarg = (mj == 0.5) ? 1.0 / dsqrt(arg) : dexp(-mj * dlog(arg));
      D3 sargsw = (mjsw == mj) ? sarg : ((mjsw == 0.5) ? 1.0 / dsqrt(arg) : dexp(-mjsw * dlog(arg)));
      return tBulkPot * (Cb * (1.0 - arg * sarg) / (1 - mj) + Cbsw * (1.0 - arg * sargsw) / (1 - mjsw));
    }
    return f4 + v * (f2 + v * (f3 / 2));
  }
  return D3(0.0);
}

// $limit sites of the load section (mos1.va:919-980): limited branch voltages w = (vgs, vds, vbs, vbd), sign-adjusted
template <class Ctx>
__device__ inline void m1_limit(const Ctx& d, const double* u, double type, double vt, double tPhi, double tVbi, double gamma, double Vg, double Vb,
                                double Vdi, double Vsi, int l0, int l1, int l2, int l3, double& w_gs, double& w_ds, double& w_bs, double& w_bd) {
  double o_vgs = type * u[l0], o_vds = type * u[l1], o_vbs = type * u[l2], o_vbd = type * u[l3];
  double vbs = type * (Vb - Vsi), vgs = type * (Vg - Vsi), vds = type * (Vdi - Vsi);
  double vbd = vbs - vds, vgd = vgs - vds, vgdo = o_vgs - o_vds;
  // Quiet rounds.  The three limiters return their argument unchanged when the iterate has moved little since the last one: DEVfetlim
  // whenever |vnew - vold| <= 0.5 whatever the threshold voltage (every clamp of mos1.va:542-605 needs a larger move -- the closest is the
  // step across vto + 0.5 from below vto), DEVlimvds whenever |vnew - vold| <= 0.5 and vnew >= -0.5 (mos1.va:607-635; the second holds by itself
  // since the old value has the sign the branch was chosen by), DEVpnjlim whenever |vnew - vold| <= 2 vt (its first branch is that very
  // test, its second one needs a step below -1 V; mos1.va:503-540).  Most Newton rounds of a transient are like that for every device at
  // once; then the limiting code -- a square root, a division and some forty branches per device -- reduces to the four subtractions that
  // rebuild vds / vgd / vbd, with the same operations in the same order as the full code, so the results are the same doubles.  (0.4
  // instead of 0.5: no rounding case to argue about.)
  {
    const bool fwd = o_vds >= 0;
    const double q_vgs = fwd ? vgs : vgd + (vgs - vgd), q_vgd = fwd ? vgs - (vgs - vgd) : vgd, q_vds = vgs - vgd;
    const bool c1 = fabs(fwd ? vgs - o_vgs : vgd - vgdo) <= 0.4, c2 = fabs(q_vds - o_vds) <= 0.4;
    const bool c3 = q_vds >= 0 ? !(fabs(vbs - o_vbs) > (vt + vt)) : !(fabs(vbd - o_vbd) > (vt + vt));
    if (__all(c1 && c2 && c3)) {
      double r_vgs = q_vgs, r_vds = q_vds, r_vbs, r_vbd;
      (void)q_vgd;
      if (r_vds >= 0) { r_vbs = vbs; r_vbd = r_vbs - r_vds; }
      else { r_vbd = vbd; r_vbs = r_vbd + r_vds; }
      if (d.initjct) { r_vbs = -1; r_vgs = type * par_of(d, M1_TVTO); r_vds = 0; r_vbd = r_vbs - r_vds; }   // mos1.va:969-974
      w_gs = type * r_vgs; w_ds = type * r_vds; w_bs = type * r_vbs; w_bd = type * r_vbd;
      return;
    }
  }
  int omode = o_vds >= 0 ? 1 : -1;
  double osel = omode == 1 ? o_vbs : o_vbd, osarg;
  if (osel <= 0) osarg = sqrt(tPhi - osel);
  else { osarg = sqrt(tPhi); osarg = osarg - o_vbs / (osarg + osarg); osarg = fmax(0.0, osarg); }   // mos1.va:940 (sic: always vbs)
  double o_von = (tVbi * type) + gamma * osarg;
  double von = type * o_von;
  if (o_vds >= 0) {
    vgs = m1_fetlim(vgs, o_vgs, von);
    vds = vgs - vgd;
    vds = m1_limvds(vds, o_vds);
    vgd = vgs - vds;
  } else {
    vgd = m1_fetlim(vgd, vgdo, von);
    vds = vgs - vgd;
    vds = -m1_limvds(-vds, -o_vds);
    vgs = vgd + vds;
  }
  if (vds >= 0) { vbs = m1_pnjlim(vbs, o_vbs, vt, par_of(d, M1_SVCRIT)); vbd = vbs - vds; }
  else { vbd = m1_pnjlim(vbd, o_vbd, vt, par_of(d, M1_DVCRIT)); vbs = vbd + vds; }
  if (d.initjct) { vbs = -1; vgs = type * par_of(d, M1_TVTO); vds = 0; vbd = vbs - vds; }   // mos1.va:969-974
  w_gs = type * vgs; w_ds = type * vds; w_bs = type * vbs; w_bd = type * vbd;
}

// junction current of one bulk diode (mos1.va:983-998)
__device__ inline D3 m1_junction(const D3& v, double vt, double gmin_m, double isat) {
  if (v.v <= -3 * vt) return gmin_m * v - isat;
  D3 x = v / vt;
  D3 e = dexp(709.0 < x.v ? D3(709.0) : x);
  return isat * (e - 1.0) + gmin_m * v;
}

// threshold, saturation voltage and drain current (mos1.va:1000-1040)
__device__ inline void m1_channel(const D3& a, const D3& b, const D3& c, const D3& dvbd, const D3& dvgd, double tPhi, double tVbi, double type,
                                  double gamma, double lambda, double Beta, int& mode, D3& dvon, D3& vdsat, D3& cdrain) {
  mode = b.v >= 0 ? 1 : -1;
  D3 sel = mode == 1 ? c : dvbd, sarg;
  if (sel.v <= 0) sarg = dsqrt(tPhi - sel);
  else { double s0 = sqrt(tPhi); sarg = s0 - sel / (s0 + s0); if (0 > sarg.v) sarg = D3(0.0); }
  dvon = tVbi * type + gamma * sarg;
  D3 vgst = (mode == 1 ? a : dvgd) - dvon;
  vdsat = vgst.v > 0 ? vgst : D3(0.0);
  if (vgst.v <= 0) cdrain = D3(0.0);
  else {
    D3 vdsm = b * (double)mode;
    D3 betap = Beta * (1.0 + lambda * vdsm);
    if (vgst.v <= vdsm.v) cdrain = betap * vgst * vgst * 0.5;
    else cdrain = betap * vdsm * (vgst - 0.5 * vdsm);
  }
}

template <class Ctx, class Out> __device__ inline void stamp_mos1(const Ctx& d, const double* u, const Out& s, double* limit_w_base) {
  const double CS = CADNIP_CHARGE_SCALE;
  int nd = node_of(d, 0), ng = node_of(d, 1), ns = node_of(d, 2), nb = node_of(d, 3), ndi = node_of(d, 4), nsi = node_of(d, 5);
  int l0 = node_of(d, 6), l1 = node_of(d, 7), l2 = node_of(d, 8), l3 = node_of(d, 9);
  double Vd = volt(u, nd), Vg = volt(u, ng), Vs = volt(u, ns), Vb = volt(u, nb), Vdi = volt(u, ndi), Vsi = volt(u, nsi);
  double type = par_of(d, M1_TYPE), vt = par_of(d, M1_VT), tPhi = par_of(d, M1_TPHI), tVbi = par_of(d, M1_TVBI);
  double gamma = par_of(d, M1_GAMMA), lambda = par_of(d, M1_LAMBDA), Beta = par_of(d, M1_BETA), OxideCap = par_of(d, M1_OXCAP);
  double mf = par_of(d, M1_MFACTOR), gmin_m = par_of(d, M1_GMIN) / mf;
  CADNIP_TRACE_POINT(20);
  // ---- limiting (mos1.va:919-980), on values
  double w_gs, w_ds, w_bs, w_bd;
  m1_limit(d, u, type, vt, tPhi, tVbi, gamma, Vg, Vb, Vdi, Vsi, l0, l1, l2, l3, w_gs, w_ds, w_bs, w_bd);
  if (limit_w_base) { limit_w_base[l0] = w_gs; limit_w_base[l1] = w_ds; limit_w_base[l2] = w_bs; limit_w_base[l3] = w_bd; }
  CADNIP_TRACE_POINT(21);
  // g_lim rows (vasim.jl:3134-3136)
  {
    const double gl[12] = {1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0};
    s.Gv(0, gl);
  }
  CADNIP_TRACE_POINT(22);
  // ---- evaluation on pass-through duals anchored at w
  D3 a = D3::seed(type * w_gs, 0), b = D3::seed(type * w_ds, 1), c = D3::seed(type * w_bs, 2);   // load_vgs, load_vds, load_vbs
  D3 dvbd = c - b, dvgd = a - b, dvgb = a - c;
  D3 cbs = m1_junction(c, vt, gmin_m, par_of(d, M1_SSATCUR));       // junction currents mos1.va:983-998
  D3 cbd = m1_junction(dvbd, vt, gmin_m, par_of(d, M1_DSATCUR));
  CADNIP_TRACE_POINT(23);
  int mode;
  D3 dvon, vdsat, cdrain;
  m1_channel(a, b, c, dvbd, dvgd, tPhi, tVbi, type, gamma, lambda, Beta, mode, dvon, vdsat, cdrain);
  CADNIP_TRACE_POINT(24);
  double ms = OxideCap == 0 ? 0.0 : OxideCap, mu = OxideCap == 0 ? 1.0 : OxideCap;   // meyer_scale / meyer_unscale mos1.va:1042-1048
  D3 qbs = m1_qdep(c, par_of(d, M1_CBS), par_of(d, M1_CBSSW), par_of(d, M1_TBULKPOT), par_of(d, M1_TDEPCAP), par_of(d, M1_MJ),
                   par_of(d, M1_MJSW), par_of(d, M1_F2S), par_of(d, M1_F3S), par_of(d, M1_F4S));
  D3 qbd = m1_qdep(dvbd, par_of(d, M1_CBD), par_of(d, M1_CBDSW), par_of(d, M1_TBULKPOT), par_of(d, M1_TDEPCAP), par_of(d, M1_MJ),
                   par_of(d, M1_MJSW), par_of(d, M1_F2D), par_of(d, M1_F3D), par_of(d, M1_F4D));
  // Meyer gate charges.  With OxideCap == 0 (no tox) meyer_scale is 0 (mos1.va:1042-1048): every gate-charge
  // term is exactly zero (value and partials), so the whole block is skipped -- same numbers, no work.
  D3 qgs(0.0), qgd(0.0), qgb(0.0);
  if (OxideCap != 0.0) {
    D3 mcgs, mcgd, mcgb;
    if (mode > 0) m1_qmeyer(a, dvgd, dvon, vdsat, tPhi, OxideCap, mcgs, mcgd, mcgb);
    else m1_qmeyer(dvgd, a, dvon, vdsat, tPhi, OxideCap, mcgd, mcgs, mcgb);
    D3 capgs = mcgs + mcgs + par_of(d, M1_CGSOV), capgd = mcgd + mcgd + par_of(d, M1_CGDOV), capgb = mcgb + mcgb + par_of(d, M1_CGBOV);
    qgs = capgs * ((ms * a) / mu); qgd = capgd * ((ms * dvgd) / mu); qgb = capgb * ((ms * dvgb) / mu);   // reactive part of ceqg* mos1.va:1140-1147
  }
  CADNIP_TRACE_POINT(25);
  D3 cdreq = (mode >= 0 ? type : -type) * cdrain;
  // branch contributions (mos1.va:1164-1169), split into resistive (Ir) and reactive (q) parts; I(d), I(s) are linear
  double gd = par_of(d, M1_GD), gs = par_of(d, M1_GS);
  D3 Ir[6], q[4];
  Ir[0] = D3(0.0); Ir[1] = D3(0.0); Ir[2] = D3(0.0);
  Ir[3] = type * cbs + type * cbd;
  Ir[4] = -1.0 * (type * cbd - cdreq);
  Ir[5] = -1.0 * (cdreq + type * cbs);
  q[0] = type * (qgs + qgb + qgd);
  q[1] = (type * qbs + type * qbd) - type * qgb;
  q[2] = -1.0 * (type * qbd + type * qgd);
  q[3] = -1.0 * (type * qbs + type * qgs);
  double dW_gs = (Vg - Vsi) - w_gs, dW_ds = (Vdi - Vsi) - w_ds, dW_bs = (Vb - Vsi) - w_bs;   // lim_rhs deltas
  const double Vk[6] = {Vd, Vg, Vs, Vb, Vdi, Vsi};
  // linear conductance terms of I(d), I(s), I(d_int), I(s_int) in node space: value and d/dV_k
  double Iv[6], dI[6][6];
#pragma unroll
  for (int br = 0; br < 6; ++br) {
    double fa = type * Ir[br].p[0], fb = type * Ir[br].p[1], fc = type * Ir[br].p[2];
    dI[br][0] = 0.0; dI[br][1] = fa; dI[br][2] = 0.0; dI[br][3] = fc; dI[br][4] = fb; dI[br][5] = -(fa + fb + fc);
    Iv[br] = Ir[br].v;
  }
  Iv[0] += gd * (Vd - Vdi); dI[0][0] += gd; dI[0][4] -= gd;
  Iv[2] += gs * (Vs - Vsi); dI[2][2] += gs; dI[2][5] -= gs;
  Iv[4] += gd * (Vdi - Vd); dI[4][4] += gd; dI[4][0] -= gd;
  Iv[5] += gs * (Vsi - Vs); dI[5][5] += gs; dI[5][2] -= gs;
  double Ib[6];
#pragma unroll
  for (int br = 0; br < 6; ++br) {
    double Ieq = mf * Iv[br];
    double g[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { g[k] = mf * dI[br][k]; Ieq = Ieq + (-g[k] * Vk[k]); }
    s.Gv(12 + 6 * br, g);
    Ieq = Ieq + (mf * type * Ir[br].p[0]) * dW_gs;
    Ieq = Ieq + (mf * type * Ir[br].p[1]) * dW_ds;
    Ieq = Ieq + (mf * type * Ir[br].p[2]) * dW_bs;
    Ib[br] = -Ieq;
    if constexpr (Out::DIRECT) {   // the row's residual: the branch current plus the lim_rhs anchoring terms
      const int nrow = br == 0 ? nd : br == 1 ? ng : br == 2 ? ns : br == 3 ? nb : br == 4 ? ndi : nsi;
      s.Rn(nrow, mf * Iv[br] + (mf * type * Ir[br].p[0]) * dW_gs + (mf * type * Ir[br].p[1]) * dW_ds + (mf * type * Ir[br].p[2]) * dW_bs);
    }
  }
  s.Bv(0, Ib);
  if constexpr (Out::DIRECT) {     // limit rows: u_l - (V_p - V_n) for (g,s_int), (d_int,s_int), (b,s_int), (b,d_int)
    s.Rn(l0, u[l0] - (Vg - Vsi)); s.Rn(l1, u[l1] - (Vdi - Vsi)); s.Rn(l2, u[l2] - (Vb - Vsi)); s.Rn(l3, u[l3] - (Vb - Vdi));
  }
  CADNIP_TRACE_POINT(26);
  int vdep = d.ipar[d.dev];
  {
    const double cs4[4] = {1.0 / CS, 1.0 / CS, 1.0 / CS, 1.0 / CS};   // charge-state formulation (vasim.jl:3433-3472)
    s.Cv(0, cs4);
  }
  double qb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double fa = mf * type * q[r].p[0], fb = mf * type * q[r].p[1], fc = mf * type * q[r].p[2];
    const double dq[6] = {0.0, fa, 0.0, fc, fb, -(fa + fb + fc)};
    const double gq[7] = {1.0, -CS * dq[0], -CS * dq[1], -CS * dq[2], -CS * dq[3], -CS * dq[4], -CS * dq[5]};
    s.Gv(48 + 7 * r, gq);
    double bc = mf * q[r].v;
#pragma unroll
    for (int k = 0; k < 6; ++k) bc -= dq[k] * Vk[k];
    bc += fa * dW_gs; bc += fb * dW_ds; bc += fc * dW_bs;
    qb[r] = CS * bc;
    // linear form (vasim.jl:3474-3482), used when the branch was not flagged voltage dependent
    s.Cv(4 + 6 * r, dq);
    if constexpr (Out::DIRECT) {
      const int np = r == 0 ? ng : r == 1 ? nb : r == 2 ? ndi : nsi;
      if ((vdep >> r) & 1) {
        const int nq = node_of(d, 10 + r);
        s.Rn(nq, u[nq] - CS * (mf * q[r].v + fa * dW_gs + fb * dW_ds + fc * dW_bs));
        s.Rn(np, s.du(nq) * (1.0 / CS));
      } else {
        s.Rn(np, dq[1] * s.du(ng) + dq[3] * s.du(nb) + dq[4] * s.du(ndi) + dq[5] * s.du(nsi));
      }
    }
  }
  s.Bv(6, qb);
  (void)vdep;
  CADNIP_TRACE_POINT(27);
}


// ---- sp_mos1 on a lane PAIR (fused kernel) -----------------------------------------------------------------------
// The DFF has 30 MOSFETs: one lane per device leaves half of a 64-lane wave idle.  Here lanes 2j and 2j+1 serve
// device j.  Both lanes evaluate the shared part (limiting, channel current); the source-side lane (side 0) and the
// drain-side lane (side 1) each evaluate ONE bulk junction and ONE depletion charge -- the same code on different
// operands -- and swap the results with a DPP quad_perm; then each lane emits half of the stamps, again with one
// instruction stream: rows (b | d_int), half of row s_int each, charge rows (g | b) and (d_int | s_int), half of the
// g_lim block each.  Applies when no device of the wave has series resistances (gd = gs = 0, the usual card);
// otherwise the caller falls back to stamp_mos1 on the even lanes.  `valid` = this lane's device exists.
__device__ __forceinline__ D3 m1_swap_pair(const D3& x) {
  D3 r;
  r.v = dpp_pair_swap(x.v);
#pragma unroll
  for (int i = 0; i < 3; ++i) r.p[i] = dpp_pair_swap(x.p[i]);
  return r;
}
__device__ __forceinline__ D3 m1_sel(bool c, const D3& x, const D3& y) {
  D3 r;
  r.v = c ? x.v : y.v;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.p[i] = c ? x.p[i] : y.p[i];
  return r;
}


template <class Ctx, class Out>
__device__ inline void stamp_mos1_pair(const Ctx& d, const double* u, const Out& s, double* limit_w_base, int side, bool valid) {
  const double CS = CADNIP_CHARGE_SCALE;
  const bool D = side != 0;
  int nd = node_of(d, 0), ng = node_of(d, 1), ns = node_of(d, 2), nb = node_of(d, 3), ndi = node_of(d, 4), nsi = node_of(d, 5);
  int l0 = node_of(d, 6), l1 = node_of(d, 7), l2 = node_of(d, 8), l3 = node_of(d, 9);
  double Vd = volt(u, nd), Vg = volt(u, ng), Vs = volt(u, ns), Vb = volt(u, nb), Vdi = volt(u, ndi), Vsi = volt(u, nsi);
  // parameters are fetched where they are first needed, not up front: 36 values held across the limiting code would not
  // fit into the registers of a two-waves-per-SIMD kernel
  double type = par_of(d, M1_TYPE), vt = par_of(d, M1_VT), tPhi = par_of(d, M1_TPHI), tVbi = par_of(d, M1_TVBI);
  double gamma = par_of(d, M1_GAMMA);
  // second group, needed after the limiting code: requested now so that the memory latency (the parameter blocks of the
  // resident instances exceed the L2) passes behind the limiting arithmetic
  const double mf = par_of(d, M1_MFACTOR), gmin_p = par_of(d, M1_GMIN), isat = par_of(d, D ? M1_DSATCUR : M1_SSATCUR);
  const double lambda = par_of(d, M1_LAMBDA), Beta = par_of(d, M1_BETA);
  const int vdep = Out::DIRECT ? d.ipar[d.dev] : 0;       // bit r: reactive branch r uses a charge unknown
  __builtin_amdgcn_sched_barrier(0);
  CADNIP_TRACE_POINT(20);
  double w_gs, w_ds, w_bs, w_bd;
  m1_limit(d, u, type, vt, tPhi, tVbi, gamma, Vg, Vb, Vdi, Vsi, l0, l1, l2, l3, w_gs, w_ds, w_bs, w_bd);
  if (limit_w_base && valid && !D) { limit_w_base[l0] = w_gs; limit_w_base[l1] = w_ds; limit_w_base[l2] = w_bs; limit_w_base[l3] = w_bd; }
  CADNIP_TRACE_POINT(21);
  {   // g_lim rows (vasim.jl:3134-3136): two limit variables per lane
    const double gl[6] = {1.0, -1.0, 1.0, 1.0, -1.0, 1.0};
    s.Gv(6 * side, gl);
    if constexpr (Out::DIRECT) {   // r_l = u_l - (V_p - V_n) for (g,s_int), (d_int,s_int) | (b,s_int), (b,d_int)
      s.Rn(D ? l2 : l0, u[D ? l2 : l0] - ((D ? Vb : Vg) - Vsi));
      s.Rn(D ? l3 : l1, u[D ? l3 : l1] - (D ? Vb - Vdi : Vdi - Vsi));
    }
  }
  CADNIP_TRACE_POINT(22);
  D3 a = D3::seed(type * w_gs, 0), b = D3::seed(type * w_ds, 1), c = D3::seed(type * w_bs, 2);
  D3 dvbd = c - b, dvgd = a - b;
  const D3 vj = m1_sel(D, dvbd, c);
  const double gmin_m = gmin_p / mf;                               // isat: this lane's junction, source side or drain side
  // third group (depletion charge of this lane's junction), in flight during the junction and channel arithmetic
  const double q_cb = par_of(d, D ? M1_CBD : M1_CBS), q_cbsw = par_of(d, D ? M1_CBDSW : M1_CBSSW), q_pot = par_of(d, M1_TBULKPOT), q_dep = par_of(d, M1_TDEPCAP);
  const double q_mj = par_of(d, M1_MJ), q_mjsw = par_of(d, M1_MJSW), q_f2 = par_of(d, D ? M1_F2D : M1_F2S), q_f3 = par_of(d, D ? M1_F3D : M1_F3S), q_f4 = par_of(d, D ? M1_F4D : M1_F4S);
  __builtin_amdgcn_sched_barrier(0);
  const D3 cj = m1_junction(vj, vt, gmin_m, isat), co = m1_swap_pair(cj);
  const D3 cbs = m1_sel(D, co, cj), cbd = m1_sel(D, cj, co);
  CADNIP_TRACE_POINT(23);
  int mode;
  D3 dvon, vdsat, cdrain;
  m1_channel(a, b, c, dvbd, dvgd, tPhi, tVbi, type, gamma, lambda, Beta, mode, dvon, vdsat, cdrain);
  CADNIP_TRACE_POINT(24);
  const D3 cdreq = (mode >= 0 ? type : -type) * cdrain;
  const double dW_gs = (Vg - Vsi) - w_gs, dW_ds = (Vdi - Vsi) - w_ds, dW_bs = (Vb - Vsi) - w_bs;   // lim_rhs deltas
  const double Vk[6] = {Vd, Vg, Vs, Vb, Vdi, Vsi};
  // one current row: G entries of d/dV_g, d/dV_b, d/dV_dint, d/dV_sint (columns 1, 3, 4, 5) and the equivalent current
  auto row = [&](const D3& I, double (&g)[6], double& Ieq) {
    const double fa = type * I.p[0], fb = type * I.p[1], fc = type * I.p[2];
    g[0] = 0.0; g[1] = mf * fa; g[2] = 0.0; g[3] = mf * fc; g[4] = mf * fb; g[5] = mf * -(fa + fb + fc);
    Ieq = mf * I.v;
#pragma unroll
    for (int k = 0; k < 6; ++k) Ieq = Ieq + (-g[k] * Vk[k]);
    Ieq = Ieq + (mf * type * I.p[0]) * dW_gs;
    Ieq = Ieq + (mf * type * I.p[1]) * dW_ds;
    Ieq = Ieq + (mf * type * I.p[2]) * dW_bs;
  };
  {   // rows I(b) | I(d_int)   (mos1.va:1164-1169)
    const D3 I3 = type * cbs + type * cbd, I4 = -1.0 * (type * cbd - cdreq);
    double g[6], Ieq;
    row(m1_sel(D, I4, I3), g, Ieq);
    const int br = 3 + side;
    s.Gv(12 + 6 * br, g);
    s.B(br, -Ieq);
    if constexpr (Out::DIRECT) {   // the row's residual is the branch current itself plus the lim_rhs anchoring terms
      const D3 I = m1_sel(D, I4, I3);
      s.Rn(D ? ndi : nb, mf * I.v + (mf * type * I.p[0]) * dW_gs + (mf * type * I.p[1]) * dW_ds + (mf * type * I.p[2]) * dW_bs);
    }
  }
  {   // row I(s_int): both lanes form it, each emits two of its four entries; the source-side lane owns the b entry
    const D3 I5 = -1.0 * (cdreq + type * cbs);
    double g[6], Ieq;
    row(I5, g, Ieq);
    const int k2[2] = {12 + 30 + (D ? 4 : 1), 12 + 30 + (D ? 5 : 3)};
    const double v2[2] = {D ? g[4] : g[1], D ? g[5] : g[3]};
    s.Gk(k2, v2);
    if (Out::DIRECT || !D) s.B(5, -Ieq);   // the source-side lane owns the b entry (a store-type writer must see one writer per slot)
    if constexpr (Out::DIRECT)
      s.Rn(nsi, D ? 0.0 : mf * I5.v + (mf * type * I5.p[0]) * dW_gs + (mf * type * I5.p[1]) * dW_ds + (mf * type * I5.p[2]) * dW_bs);
  }
  CADNIP_TRACE_POINT(26);
  // depletion charge of this lane's junction (after the current rows: fewer values live at once), swapped like the current
  const D3 qj = m1_qdep(vj, q_cb, q_cbsw, q_pot, q_dep, q_mj, q_mjsw, q_f2, q_f3, q_f4);
  const D3 qo = m1_swap_pair(qj);
  const D3 qbs = m1_sel(D, qo, qj), qbd = m1_sel(D, qj, qo);
  CADNIP_TRACE_POINT(25);
  {   // charge-state columns (vasim.jl:3433-3472): two per lane
    const double cs2[2] = {1.0 / CS, 1.0 / CS};
    s.Cv(2 * side, cs2);
  }
  // reactive rows: branch charges of g, b, d_int, s_int.  Meyer charges vanish with OxideCap == 0; the pair path is only
  // taken then (caller), so q_g = 0, q_b = type (qbs + qbd), q_dint = -type qbd, q_sint = -type qbs
  const D3 q0(0.0), q1 = type * qbs + type * qbd, q2 = -1.0 * (type * qbd), q3 = -1.0 * (type * qbs);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = 2 * it + side;
    const D3 q = it == 0 ? m1_sel(D, q1, q0) : m1_sel(D, q3, q2);
    const double fa = mf * type * q.p[0], fb = mf * type * q.p[1], fc = mf * type * q.p[2];
    const double dq[6] = {0.0, fa, 0.0, fc, fb, -(fa + fb + fc)};
    const double gq[7] = {1.0, -CS * dq[0], -CS * dq[1], -CS * dq[2], -CS * dq[3], -CS * dq[4], -CS * dq[5]};
    // Only one of the two forms of a branch lands in the matrix: the slots of the other one lead to trash words (no
    // charge unknown / no linear entries in the pattern).  The direct-residual variants know which (vdep) and leave the
    // dead half out; q_g = 0 here, so branch 0 stamps nothing at all.
    const bool cs_form = !Out::DIRECT || ((vdep >> r) & 1), lin_form = !Out::DIRECT || (!((vdep >> r) & 1) && r != 0);
    if (cs_form) s.Gv(48 + 7 * r, gq);
    double bc = mf * q.v;
#pragma unroll
    for (int k = 0; k < 6; ++k) bc -= dq[k] * Vk[k];
    bc += fa * dW_gs; bc += fb * dW_ds; bc += fc * dW_bs;
    s.B(6 + r, CS * bc);
    if (lin_form) s.Cv(4 + 6 * r, dq);
    if constexpr (Out::DIRECT) {
      // branch node of reactive branch r: g, b, d_int, s_int.  Charge-state form (vasim.jl:3433-3472) when the branch was
      // flagged voltage dependent: r_q = u_q - CS (q + lim terms), r_p += du_q / CS; else the linear form r_p += sum dq_k du_k.
      const int np = r == 0 ? ng : r == 1 ? nb : r == 2 ? ndi : nsi;
      if ((vdep >> r) & 1) {
        const int nq = node_of(d, 10 + r);
        s.Rn(nq, u[nq] - CS * (mf * q.v + fa * dW_gs + fb * dW_ds + fc * dW_bs));
        s.Rn(np, s.du(nq) * (1.0 / CS));
      } else if (r != 0) {
        s.Rn(np, dq[1] * s.du(ng) + dq[3] * s.du(nb) + dq[4] * s.du(ndi) + dq[5] * s.du(nsi));
      }
    }
  }
  CADNIP_TRACE_POINT(27);
}

// ---- sp_mos1 for a TEAM of waves (fused_team_kernel.hpp: several waves per sweep instance) --------------------------
// A single transient is a latency problem: one wave evaluates a MOSFET as one serial instruction stream of ~1 400
// instructions, whatever the number of devices.  The load section of mos1.va has three independent parts behind the
// limiting code -- the bulk-junction currents (mos1.va:983-998), the channel current (:1000-1040) and the depletion
// charges (:1049-1109) -- and every matrix / residual entry is a SUM of contributions of those parts (I(b) = type (cbs +
// cbd), I(d_int) = -(type cbd - cdreq), I(s_int) = -(cdreq + type cbs); q_b = type (qbs + qbd), q_dint = -type qbd,
// q_sint = -type qbs).  So the parts run on different waves (`roles`, wave-uniform), each wave on lane pairs -- lane 2j + s
// serves junction s (0 = source side, 1 = drain side) of device j, or one of the channel's two rows -- and adds its share into
// the work array with LDS atomics.  Limiting is evaluated by every role (same inputs, same result).  No cross-lane traffic.
// Plain cards only (gd = gs = OxideCap = 0), direct residuals only: the caller checks.
// Register-resident view of ONE device for one lane (team kernel, first pass over the first sp_mos1 block): the node indices are a property
// of the circuit (read once per launch), the 34 derived parameters of the instance (read once per residence).  Side-dependent parameters --
// the source-side lane needs the source junction's, the drain-side lane the drain junction's -- are resolved when they are loaded: the slot
// of the source-side parameter holds this lane's value (par_side).  Indices are literals at every use, so the arrays are registers.
#define M1_NPAR 34
struct M1RegView {
  double p[M1_NPAR];
  int nd[14];
  int initjct;
};
__device__ __forceinline__ int node_of(const M1RegView& d, int k) { return d.nd[k]; }
__device__ __forceinline__ double par_of(const M1RegView& d, int k) { return d.p[k]; }
__device__ __forceinline__ double par_side(const M1RegView& d, int ks, int, bool) { return d.p[ks]; }
template <class Ctx> __device__ __forceinline__ double par_side(const Ctx& d, int ks, int kd, bool D) { return par_of(d, D ? kd : ks); }
__device__ __forceinline__ int node_side(const M1RegView& d, int ks, int kd, bool D) { return D ? d.nd[kd] : d.nd[ks]; }
template <class Ctx> __device__ __forceinline__ int node_side(const Ctx& d, int ks, int kd, bool D) { return node_of(d, D ? kd : ks); }
// (one sweep instance: its parameter block `par` [M1_NPAR][count], device `dev`, this lane's junction side)
__device__ __forceinline__ void m1_load_params(M1RegView& v, const double* par, int count, int dev, bool D) {
#pragma unroll
  for (int k = 0; k < M1_NPAR; ++k) v.p[k] = par[k * count + dev];
  if (D) { v.p[M1_SSATCUR] = v.p[M1_DSATCUR]; v.p[M1_CBS] = v.p[M1_CBD]; v.p[M1_CBSSW] = v.p[M1_CBDSW]; v.p[M1_F2S] = v.p[M1_F2D]; v.p[M1_F3S] = v.p[M1_F3D]; v.p[M1_F4S] = v.p[M1_F4D]; }
}

#define M1_ROLE_J 1    // junction currents: rows b, d_int | s_int
#define M1_ROLE_Q 2    // depletion charges: reactive rows b, d_int | s_int (charge-state or linear form)
#define M1_ROLE_CH 4   // channel current: rows d_int | s_int; the limit rows (g_lim block, limit residuals, limit_w)
template <class Ctx, class Out>
__device__ inline void stamp_mos1_team(const Ctx& d, const double* u, const Out& s, double* limit_w_base, int side, bool valid, int roles, int vdep) {
  static_assert(Out::DIRECT, "team stamping emits direct residuals");
  const double CS = CADNIP_CHARGE_SCALE;
  const bool D = side != 0;
  int ng = node_of(d, 1), nb = node_of(d, 3), ndi = node_of(d, 4), nsi = node_of(d, 5);
  int l0 = node_of(d, 6), l1 = node_of(d, 7), l2 = node_of(d, 8), l3 = node_of(d, 9);
  double Vg = volt(u, ng), Vb = volt(u, nb), Vdi = volt(u, ndi), Vsi = volt(u, nsi);
  double type = par_of(d, M1_TYPE), vt = par_of(d, M1_VT), tPhi = par_of(d, M1_TPHI), tVbi = par_of(d, M1_TVBI);
  double gamma = par_of(d, M1_GAMMA);
  const double mf = par_of(d, M1_MFACTOR);
  CADNIP_TRACE_POINT(20);               // (vdep, bit r: reactive branch r uses a charge unknown -- a property of the circuit, fetched by the caller)
  double w_gs, w_ds, w_bs, w_bd;
  m1_limit(d, u, type, vt, tPhi, tVbi, gamma, Vg, Vb, Vdi, Vsi, l0, l1, l2, l3, w_gs, w_ds, w_bs, w_bd);
  CADNIP_TRACE_POINT(21);
  D3 a = D3::seed(type * w_gs, 0), b = D3::seed(type * w_ds, 1), c = D3::seed(type * w_bs, 2);
  D3 dvbd = c - b, dvgd = a - b;
  const D3 vj = m1_sel(D, dvbd, c);                                  // this lane's junction voltage (J, Q roles)
  const double dW_gs = (Vg - Vsi) - w_gs, dW_ds = (Vdi - Vsi) - w_ds, dW_bs = (Vb - Vsi) - w_bs;   // lim_rhs deltas
  // one current contribution I (a dual over vgs, vds, vbs): its Jacobian entries d/dV_g, d/dV_b, d/dV_dint, d/dV_sint and
  // its residual share I + lim_rhs anchoring terms (vasim.jl:2957-2966)
  auto row = [&](const D3& I, double (&g)[6], double& res) {
    const double fa = type * I.p[0], fb = type * I.p[1], fc = type * I.p[2];
    g[0] = 0.0; g[1] = mf * fa; g[2] = 0.0; g[3] = mf * fc; g[4] = mf * fb; g[5] = mf * -(fa + fb + fc);
    res = mf * I.v + (mf * type * I.p[0]) * dW_gs + (mf * type * I.p[1]) * dW_ds + (mf * type * I.p[2]) * dW_bs;
  };
  if (roles & M1_ROLE_CH) {
    if (limit_w_base && valid && !D) { limit_w_base[l0] = w_gs; limit_w_base[l1] = w_ds; limit_w_base[l2] = w_bs; limit_w_base[l3] = w_bd; }
    {   // g_lim rows (vasim.jl:3134-3136) and the limit rows' residuals r_l = u_l - (V_p - V_n): two limit variables per lane
      const double gl[6] = {1.0, -1.0, 1.0, 1.0, -1.0, 1.0};
      s.Gv(6 * side, gl);
      s.Rn(D ? l2 : l0, u[D ? l2 : l0] - ((D ? Vb : Vg) - Vsi));
      s.Rn(D ? l3 : l1, u[D ? l3 : l1] - (D ? Vb - Vdi : Vdi - Vsi));
    }
    CADNIP_TRACE_POINT(22);
    const double lambda = par_of(d, M1_LAMBDA), Beta = par_of(d, M1_BETA);
    int mode;
    D3 dvon, vdsat, cdrain;
    m1_channel(a, b, c, dvbd, dvgd, tPhi, tVbi, type, gamma, lambda, Beta, mode, dvon, vdsat, cdrain);
    CADNIP_TRACE_POINT(24);
    const D3 cdreq = (mode >= 0 ? type : -type) * cdrain;
    const D3 I = m1_sel(D, cdreq, -1.0 * cdreq);              // row d_int carries +cdreq, row s_int -cdreq (mos1.va:1168-1169)
    double g[6], res;
    row(I, g, res);
    s.Gv(12 + 6 * (D ? 4 : 5), g);
    s.Rn(D ? ndi : nsi, res);
    CADNIP_TRACE_POINT(26);
  }
  if (roles & M1_ROLE_J) {
    const double gmin_m = par_of(d, M1_GMIN) / mf, isat = par_side(d, M1_SSATCUR, M1_DSATCUR, D);
    const D3 cj = m1_junction(vj, vt, gmin_m, isat);
    CADNIP_TRACE_POINT(23);
    const D3 Ib = type * cj;                                  // into row b; the junction's own terminal row takes the negative
    double g[6], res;
    row(Ib, g, res);
    s.Gv(12 + 6 * 3, g);
    s.Rn(nb, res);
    const double gn[6] = {0.0, -g[1], 0.0, -g[3], -g[4], -g[5]};
    s.Gv(12 + 6 * (D ? 4 : 5), gn);
    s.Rn(D ? ndi : nsi, -res);
    CADNIP_TRACE_POINT(26);
  }
  if (roles & M1_ROLE_Q) {
    const double q_cb = par_side(d, M1_CBS, M1_CBD, D), q_cbsw = par_side(d, M1_CBSSW, M1_CBDSW, D), q_pot = par_of(d, M1_TBULKPOT), q_dep = par_of(d, M1_TDEPCAP);
    const double q_mj = par_of(d, M1_MJ), q_mjsw = par_of(d, M1_MJSW), q_f2 = par_side(d, M1_F2S, M1_F2D, D), q_f3 = par_side(d, M1_F3S, M1_F3D, D), q_f4 = par_side(d, M1_F4S, M1_F4D, D);
    const D3 qj = m1_qdep(vj, q_cb, q_cbsw, q_pot, q_dep, q_mj, q_mjsw, q_f2, q_f3, q_f4);
    CADNIP_TRACE_POINT(25);
    {   // charge-state columns C[p, q_r] = 1 / CS (vasim.jl:3433-3472): two per lane
      const double cs2[2] = {1.0 / CS, 1.0 / CS};
      s.Cv(2 * side, cs2);
    }
    // this junction's charge enters reactive branch 1 (node b) with +type, and its own terminal's branch (2: d_int, 3: s_int)
    // with -type.  Each branch has one owner lane for what is stamped once: the charge unknown's unit diagonal, its own
    // value in the residual and the du_q / CS term of the node row.  Branch 0 (gate) carries no charge on a plain card.
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = it == 0 ? 1 : (D ? 2 : 3);
      const bool owner = it == 0 ? !D : true;
      const D3 q = it == 0 ? type * qj : -1.0 * (type * qj);
      const double fa = mf * type * q.p[0], fb = mf * type * q.p[1], fc = mf * type * q.p[2];
      const double dq[6] = {0.0, fa, 0.0, fc, fb, -(fa + fb + fc)};
      const int np = r == 1 ? nb : r == 2 ? ndi : nsi;
      if ((vdep >> r) & 1) {
        const double gq[7] = {owner ? 1.0 : 0.0, 0.0, -CS * dq[1], 0.0, -CS * dq[3], -CS * dq[4], -CS * dq[5]};
        s.Gv(48 + 7 * r, gq);
        const int nq = it == 0 ? node_of(d, 11) : node_side(d, 13, 12, D);      // the charge unknown of branch r
        s.Rn(nq, (owner ? u[nq] : 0.0) - CS * (mf * q.v + fa * dW_gs + fb * dW_ds + fc * dW_bs));
        s.Rn(np, owner ? s.du(nq) * (1.0 / CS) : 0.0);
      } else {
        s.Cv(4 + 6 * r, dq);
        s.Rn(np, dq[1] * s.du(ng) + dq[3] * s.du(nb) + dq[4] * s.du(ndi) + dq[5] * s.du(nsi));
      }
    }
    if (!D && (vdep & 1)) {   // a gate charge unknown exists in the pattern (never on a plain card's own detection, kept for safety): q_g = 0
      const double g1[1] = {1.0};
      s.Gv(48, g1);
      const int nq = node_of(d, 10);
      s.Rn(nq, u[nq]);
      s.Rn(ng, s.du(nq) * (1.0 / CS));
    }
    CADNIP_TRACE_POINT(27);
  }
}

}  // namespace cadnip

#include "va_runtime.hpp"
