// va_runtime.hpp -- what the generated Verilog-A stamp functions (va_generated.hpp, written by cadnip.jl_amd/va/hipgen.py)
// are built from: elementary functions on Dual<N> / double, and the branch stamp of the reference's generated stamp!
// (/root/reference/src/vasim.jl:3319-3521; charge-state formulation :3433-3472, constant capacitance :3474-3482).
// Included at the end of devices.hpp.
#pragma once

namespace cadnip {

// what $temperature, $mfactor, $simparam("gmin" | "initjct" / "iniLim"), analysis() and $abstime read
struct VaSys { double temp, mf, gmin, initjct; int mode; double time; };

template <class X> __device__ __forceinline__ double va_val(X x) { return (double)x; }
template <int N> __device__ __forceinline__ double va_val(const Dual<N>& x) { return x.v; }
template <class X> __device__ __forceinline__ bool va_true(X x) { return va_val(x) != 0.0; }

// f(a) with derivative df: chain rule over the partials
template <int N> __device__ __forceinline__ Dual<N> va_chain(const Dual<N>& a, double f, double df) {
  Dual<N> r; r.v = f;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * df;
  return r;
}
// The expensive elementary functions (software fp64: exp ~40 instructions, log ~50, pow ~170) behind one name each.  PSP103 has ~500 call
// sites and its tangent-lane stamp function is 157 k vector instructions of straight-line code (1.2 MB) that every wave streams through the
// instruction cache once; calling these instead of inlining them was tried in the translation unit that carries it and lost.
#ifdef CADNIP_VA_WITH_EXT
#define VA_MATH_FN static __device__ __forceinline__   // (called, not inlined: measured 7 % slower -- the call sequences cost more than the instruction fetches they save)
#else
#define VA_MATH_FN __device__ __forceinline__
#endif
VA_MATH_FN double va_m_exp(double x) { return exp(x); }
VA_MATH_FN double va_m_log(double x) { return log(x); }
VA_MATH_FN double va_m_pow(double a, double b) { return pow(a, b); }
#define VA_UNARY(name, fexpr, dfexpr)                                                                     \
  __device__ __forceinline__ double va_##name(double x) { return fexpr; }                                 \
  template <int N> __device__ __forceinline__ Dual<N> va_##name(const Dual<N>& a) {                        \
    const double x = a.v; const double f = fexpr; (void)f; return va_chain(a, f, dfexpr); }
VA_UNARY(exp, va_m_exp(x), f)
VA_UNARY(ln, va_m_log(x), 1.0 / x)
VA_UNARY(log, log10(x), 1.0 / (x * 2.302585092994046))
VA_UNARY(sqrt, sqrt(x), 0.5 / f)
VA_UNARY(tanh, tanh(x), 1.0 - f * f)
VA_UNARY(sinh, sinh(x), cosh(x))
VA_UNARY(cosh, cosh(x), sinh(x))
VA_UNARY(sin, sin(x), cos(x))
VA_UNARY(cos, cos(x), -sin(x))
VA_UNARY(atan, atan(x), 1.0 / (1.0 + x * x))
VA_UNARY(tan, tan(x), 1.0 + f * f)
VA_UNARY(asin, asin(x), 1.0 / sqrt(1.0 - x * x))
VA_UNARY(acos, acos(x), -1.0 / sqrt(1.0 - x * x))
VA_UNARY(asinh, asinh(x), 1.0 / sqrt(x * x + 1.0))
VA_UNARY(acosh, acosh(x), 1.0 / sqrt(x * x - 1.0))
VA_UNARY(atanh, atanh(x), 1.0 / (1.0 - x * x))
// limexp: exp below 80, its tangent above (the usual SPICE continuation)
VA_UNARY(limexp, (x < 80.0 ? va_m_exp(x) : 5.54062238439351e+34 * (1.0 + x - 80.0)), (x < 80.0 ? f : 5.54062238439351e+34))
#undef VA_UNARY
// piecewise constant functions: plain numbers
template <class X> __device__ __forceinline__ double va_floor(const X& x) { return floor(va_val(x)); }
template <class X> __device__ __forceinline__ double va_ceil(const X& x) { return ceil(va_val(x)); }
template <class X> __device__ __forceinline__ double va_int(const X& x) { return trunc(va_val(x)); }
// ddx(expr, V(node k)): the partial itself, a plain number
__device__ __forceinline__ double va_ddx(double, int) { return 0.0; }
template <int N> __device__ __forceinline__ double va_ddx(const Dual<N>& a, int k) { return a.p[k]; }
__device__ __forceinline__ double va_abs(double x) { return fabs(x); }
template <int N> __device__ __forceinline__ Dual<N> va_abs(const Dual<N>& a) { return a.v >= 0.0 ? a : -a; }

// two-argument functions: every double / dual combination
// pow: the exponents compact models use are mostly parameters with simple values (JUNCAP's grading coefficients default to 1/2,
// mobility exponents to 1 or 2).  A software fp64 pow is ~170 instructions on this ISA, a square root ~20: the common exponents take
// their closed forms (exact or within the same ulp as pow), everything else goes to pow.  The derivative of a^b with respect to a is
// b f / a -- one division instead of a second pow (b a^(b-1) when a = 0).
__device__ __forceinline__ double va_pow(double a, double b) {
  if (b == 0.5) return sqrt(a);
  if (b == 1.0) return a;
  if (b == 2.0) return a * a;
  if (b == -1.0) return 1.0 / a;
  if (b == -0.5) return 1.0 / sqrt(a);
  if (b == 0.0) return 1.0;
  if (b == 1.5) return a * sqrt(a);
  if (b == 3.0) return a * a * a;
  return va_m_pow(a, b);
}
__device__ __forceinline__ double va_dpow(double a, double b, double f) {   // d(a^b)/da given f = a^b
  if (b == 0.0) return 0.0;
  return a != 0.0 ? b * f / a : b * va_pow(a, b - 1.0);
}
template <int N> __device__ __forceinline__ Dual<N> va_pow(const Dual<N>& a, double b) {
  const double f = va_pow(a.v, b);
  return va_chain(a, f, va_dpow(a.v, b, f));
}
template <int N> __device__ __forceinline__ Dual<N> va_pow(double a, const Dual<N>& b) {
  const double f = va_pow(a, b.v);
  return va_chain(b, f, f * va_m_log(a));
}
template <int N> __device__ __forceinline__ Dual<N> va_pow(const Dual<N>& a, const Dual<N>& b) {
  const double f = va_pow(a.v, b.v), da = va_dpow(a.v, b.v, f);
  bool bdep = false;
#pragma unroll
  for (int i = 0; i < N; ++i) bdep = bdep || b.p[i] != 0.0;
  const double db = bdep ? f * log(a.v) : 0.0;             // (the exponent is usually a parameter carried in a dual: no log then)
  Dual<N> r; r.v = f;
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = a.p[i] * da + (b.p[i] != 0.0 ? b.p[i] * db : 0.0);
  return r;
}
__device__ __forceinline__ double va_atan2(double y, double x) { return atan2(y, x); }
template <int N> __device__ __forceinline__ Dual<N> va_atan2(const Dual<N>& y, const Dual<N>& x) {
  const double d = x.v * x.v + y.v * y.v;
  Dual<N> r; r.v = atan2(y.v, x.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.p[i] = (y.p[i] * x.v - x.p[i] * y.v) / d;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> va_atan2(const Dual<N>& y, double x) { return va_atan2(y, Dual<N>(x)); }
template <int N> __device__ __forceinline__ Dual<N> va_atan2(double y, const Dual<N>& x) { return va_atan2(Dual<N>(y), x); }
__device__ __forceinline__ double va_hypot(double a, double b) { return hypot(a, b); }
template <class A, class B> __device__ __forceinline__ auto va_hypot(const A& a, const B& b) -> decltype(va_sqrt(a * a + b * b)) { return va_sqrt(a * a + b * b); }
// min / max select by value and carry the selected operand's partials (ForwardDiff semantics); ties take the first
__device__ __forceinline__ double va_max(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ double va_min(double a, double b) { return b < a ? b : a; }
template <int N> __device__ __forceinline__ Dual<N> va_max(const Dual<N>& a, const Dual<N>& b) { return b.v > a.v ? b : a; }
template <int N> __device__ __forceinline__ Dual<N> va_min(const Dual<N>& a, const Dual<N>& b) { return b.v < a.v ? b : a; }
template <int N> __device__ __forceinline__ Dual<N> va_max(const Dual<N>& a, double b) { return b > a.v ? Dual<N>(b) : a; }
template <int N> __device__ __forceinline__ Dual<N> va_max(double a, const Dual<N>& b) { return b.v > a ? b : Dual<N>(a); }
template <int N> __device__ __forceinline__ Dual<N> va_min(const Dual<N>& a, double b) { return b < a.v ? Dual<N>(b) : a; }
template <int N> __device__ __forceinline__ Dual<N> va_min(double a, const Dual<N>& b) { return b.v < a ? b : Dual<N>(a); }

// A $limit call site (vasim.jl:1316-1330): value w, the probe's node partials kept ("undamped"), unit partial in the site's
// own slot so that the companion model can be re-anchored at w (lim_rhs).
template <int W> __device__ __forceinline__ Dual<W> va_site(const Dual<W>& probe, double w, int slot) {
  Dual<W> r = probe;
  r.v = w;
  r.p[slot] = 1.0;
  return r;
}
// rows of the limit unknown l of probe branch (p, n): u_l - (V_p - V_n) = 0 (the hoisted preamble, vasim.jl:3110-3138)
template <int N, int B, class Out> __device__ __forceinline__ void va_emit_limit_rows(const Out& s, int l) {
  const int g0 = 2 * N * B + (N + 1) * B + 3 * l;
  s.G(g0, 1.0);
  s.G(g0 + 1, -1.0);
  s.G(g0 + 2, 1.0);
}

// An executed potential contribution V(p,n) <+ 0 with its branch current I (local unknown `ui`; vasim.jl:2363-2393): KCL columns
// G[p,I] = 1, G[n,I] = -1, the constraint row G[I,p] = 1, G[I,n] = -1 with its (zero) partials -dV/dV_k in every node column --
// the reference stamps them, so they are part of the pattern -- and b[I] = 0.  g0: first G slot of the short (VAModule.g_short),
// bk: its b slot.  `on`: the statement executes for this instance with two distinct unknowns behind p and n.
template <int N, class Out>
__device__ __forceinline__ void va_emit_short(const double* u, const Out& s, const double (&Vf)[N], const int* nd, int ui, int pl, int nl, int g0, int bk, bool on) {
  if (!on) return;
  s.G(g0, 1.0);
  s.G(g0 + 1, -1.0);
  s.G(g0 + 2, 1.0);
  s.G(g0 + 3, -1.0);
  if constexpr (!Out::DIRECT) {
#pragma unroll
    for (int k = 0; k < N; ++k) s.G(g0 + 4 + k, 0.0);
    s.B(bk, 0.0);
  }
  if constexpr (Out::DIRECT) {
    const int ni = nd[ui];
    const double cur = u[ni];
    s.Rn(pl < 0 ? -1 : nd[pl], cur);
    s.Rn(nl < 0 ? -1 : nd[nl], -cur);
    s.Rn(ni, (pl < 0 ? 0.0 : Vf[pl]) - (nl < 0 ? 0.0 : Vf[nl]));
  }
}

// A potential contribution with a value.  Two-node form V(p,n) <+ X (vasim.jl:2340-2393, 3780-3812): KCL columns, the constraint row
// G[I,p] = 1, G[I,n] = -1, G[I,k] = -dX/dV_k in every node column, b[I] = X - sum_k dX/dV_k V_k.  Per-op path only.
__device__ __forceinline__ double va_probe_current(const double* u, int node, bool on) { return on && node >= 0 ? u[node] : 0.0; }
template <int N, int W, class Out>
__device__ __forceinline__ void va_emit_vcontrib(const Out& s, const double (&Vf)[N], int g0, int bk, bool on, const Dual<W>& X) {
  static_assert(!Out::DIRECT, "potential contributions with a value exist on the per-op path only");
  if (!on) return;
  s.G(g0, 1.0); s.G(g0 + 1, -1.0); s.G(g0 + 2, 1.0); s.G(g0 + 3, -1.0);
  double beq = X.v;
#pragma unroll
  for (int k = 0; k < N; ++k) { s.G(g0 + 4 + k, -X.p[k]); beq -= X.p[k] * Vf[k]; }
  s.B(bk, beq);
}
// Named branch V(br) <+ X at the top level of the analog block (vasim.jl:3669-3746): no partials -- b[I] = the resistive value,
// C[I,I] = -(the value under ddt()) when there is one.
template <class Out>
__device__ __forceinline__ void va_emit_vnamed(const Out& s, int g0, int bk, int ck, bool on, double val, double react) {
  static_assert(!Out::DIRECT, "potential contributions with a value exist on the per-op path only");
  if (!on) return;
  s.G(g0, 1.0); s.G(g0 + 1, -1.0); s.G(g0 + 2, 1.0); s.G(g0 + 3, -1.0);
  s.B(bk, val);
  if (ck >= 0) s.C(ck, -react);
}

// One branch (p, n) of a generated module: I = resistive current (value + d/dV_k), Q = charge (value + d/dV_k), both
// already scaled by the multiplicity factor.  Slot layout: VAModule.shape / .program (cadnip.jl_amd/va/frontend.py).
// Both reactive forms are written; the circuit's pattern holds the one the host's voltage-dependence detection chose, the
// other's slots are never gathered (per-op path) / lead to trash words (fused path).  `pl`, `nl`: local node index or -1.
// `ld[j]` = V(probe_j) - w_j of $limit site j: the partials N + j of I and Q re-anchor the companion model at w.
template <int N, int S, int B, bool REACTIVE, class Ctx, class Out>
__device__ __forceinline__ void va_emit_branch(const Ctx& d, const double* u, const Out& s, const double (&Vf)[N], const double* ld, const int* nd, int b, int pl,
                                               int nl, const Dual<N + S>& I, const Dual<N + S>& Q, bool vdep) {
  const double CS = CADNIP_CHARGE_SCALE;
  double Ieq = I.v, Ilim = 0.0, Qlim = 0.0;
#pragma unroll
  for (int j = 0; j < S; ++j) { Ilim += I.p[N + j] * ld[j]; Qlim += Q.p[N + j] * ld[j]; }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    s.G(2 * N * b + 2 * k, I.p[k]);
    s.G(2 * N * b + 2 * k + 1, -I.p[k]);
    Ieq += -I.p[k] * Vf[k];
  }
  if (REACTIVE) {
    s.C(2 * b, 1.0 / CS);
    s.C(2 * b + 1, -1.0 / CS);
    s.G(2 * N * B + (N + 1) * b, 1.0);
    double bq = Q.v;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      s.G(2 * N * B + (N + 1) * b + 1 + k, -CS * Q.p[k]);
      bq -= Q.p[k] * Vf[k];
      s.C(2 * B + 2 * N * b + 2 * k, Q.p[k]);
      s.C(2 * B + 2 * N * b + 2 * k + 1, -Q.p[k]);
    }
    bq += Qlim;                       // same summation order as the reference: value, node terms, then the lim_rhs terms
    s.B(3 * b + 2, CS * bq);
  }
  Ieq += Ilim;
  s.B(3 * b, -Ieq);
  s.B(3 * b + 1, Ieq);
  if constexpr (Out::DIRECT) {   // the branch's share of r = C du + G u - b, straight from I and Q
    double r = I.v + Ilim;
    if (REACTIVE) {
      if (vdep) {
        const int nq = nd[N + b];
        s.Rn(nq, u[nq] - CS * (Q.v + Qlim));
        r += s.du(nq) * (1.0 / CS);
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) r += Q.p[k] * s.du(nd[k]);
      }
    }
    s.Rn(pl < 0 ? -1 : nd[pl], r);
    s.Rn(nl < 0 ? -1 : nd[nl], -r);
  }
}

// ---- tangent lanes: one derivative direction per lane ---------------------------------------------------------------------
// The large external models (PSP103: ~3 000 statements on duals of width 12) are evaluated by a group of VA_TL_LANES = 16 lanes per
// device: every lane of the group computes the same values and ONE partial -- lane `dir` carries d/dV_dir for dir < N and the
// $limit-site partial N + j above -- so a dual operation costs the value plus one multiply-add instead of N + S of them, the
// duals live in registers instead of scratch memory, and the device's latency chain is ~5 x shorter.  Sums over the directions
// (the equivalent current I - sum_k dI/dV_k V_k) are DPP row sums (tran_ctrl.hpp: group_sum16); a lane writes the stamps of its
// own node column.  Per-op stamping kernel only (stamp_csr.hip); the summation order of the equivalent currents differs from
// the reference's left-to-right sum, i.e. results agree with the oracle to rounding, not bit for bit.
#define VA_TL_LANES 16          // lanes per device: 16 (one DPP row) for models with up to 16 directions, 32 (two rows) up to 32
__device__ __forceinline__ Dual<1> va_seed_tl(double x, bool mine) { Dual<1> r; r.v = x; r.p[0] = mine ? 1.0 : 0.0; return r; }
__device__ __forceinline__ Dual<1> va_site_tl(const Dual<1>& probe, double w, bool mine) { Dual<1> r = probe; r.v = w; if (mine) r.p[0] = 1.0; return r; }
// ddx(expr, V(node k)): the partial held by lane k of this device's group
template <int LANES> __device__ __forceinline__ double va_ddx_tl(double, int) { return 0.0; }
template <int LANES> __device__ __forceinline__ double va_ddx_tl(const Dual<1>& a, int k) { return __shfl(a.p[0], (int)((__lane_id() & ~(LANES - 1)) + k)); }
template <int CTRL> __device__ __forceinline__ double dpp_row_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over one device's direction lanes -- the 16 lanes of a DPP row, or two neighbouring rows -- result in every lane; all lanes of the
// wave must be active
template <int LANES> __device__ __forceinline__ double va_group_sum(double v) {
  v += dpp_row_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_row_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_row_f64<0x141>(v);   // row_half_mirror
  v += dpp_row_f64<0x140>(v);   // row_mirror
  if (LANES == 32) v += __shfl_xor(v, 16);
  return v;
}

// two-node potential contribution on tangent lanes: wgt_node = -V_dir for a node direction, 0 otherwise (the constraint row takes the node
// partials only, vasim.jl:3795-3810); every lane of the device's group takes part in the sum
template <int N, int LANES, class Out>
__device__ __forceinline__ void va_emit_vcontrib_tl(const Out& s, int g0, int bk, bool on, const Dual<1>& X, double wgt_node, int dir) {
  const double beq = X.v + va_group_sum<LANES>(X.p[0] * wgt_node);
  if (!on) return;
  if (dir == 0) { s.G(g0, 1.0); s.G(g0 + 1, -1.0); s.G(g0 + 2, 1.0); s.G(g0 + 3, -1.0); s.B(bk, beq); }
  if (dir < N) s.G(g0 + 4 + dir, -X.p[0]);
}

// wgt: this direction's weight in  X - sum_k dX/dV_k V_k + sum_j dX/dsite_j (V(probe_j) - w_j):  -V_dir for a node direction,
// V(probe_j) - w_j for $limit site j, 0 for an idle lane (the generated function computes it once, after the body)
template <int N, int S, int B, bool REACTIVE, int LANES, class Out>
__device__ __forceinline__ void va_emit_branch_tl(const Out& s, int b, const Dual<1>& I, const Dual<1>& Q, double wgt, int dir) {
  static_assert(!Out::DIRECT, "tangent-lane stamping exists for the per-op path only");
  const double CS = CADNIP_CHARGE_SCALE;
  const bool node_dir = dir < N;
  const double Ieq = I.v + va_group_sum<LANES>(I.p[0] * wgt);
  if (node_dir) {
    s.G(2 * N * b + 2 * dir, I.p[0]);
    s.G(2 * N * b + 2 * dir + 1, -I.p[0]);
  }
  if (REACTIVE) {
    const double bq = Q.v + va_group_sum<LANES>(Q.p[0] * wgt);
    if (node_dir) {
      s.G(2 * N * B + (N + 1) * b + 1 + dir, -CS * Q.p[0]);
      s.C(2 * B + 2 * N * b + 2 * dir, Q.p[0]);
      s.C(2 * B + 2 * N * b + 2 * dir + 1, -Q.p[0]);
    }
    if (dir == 0) {
      s.C(2 * b, 1.0 / CS);
      s.C(2 * b + 1, -1.0 / CS);
      s.G(2 * N * B + (N + 1) * b, 1.0);
      s.B(3 * b + 2, CS * bq);
    }
  }
  if (dir == 0) {
    s.B(3 * b, -Ieq);
    s.B(3 * b + 1, Ieq);
  }
}

}  // namespace cadnip

#define CADNIP_VA_DEVICE_CODE
#include "va_generated.hpp"
