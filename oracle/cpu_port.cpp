// TEST INFRASTRUCTURE (oracle) -- compiled CPU restatement ("port") of the hot path.
//
// Single-threaded C++ restatement of Cadnip.jl's transient inner loop on the flattened device
// table: stamping (/root/reference/src/mna/devices.jl, models/VADistillerModels.jl/va/mos1.va
// through the stamp pattern of src/vasim.jl:3319-3521), fast_rebuild!/fast_residual!/
// fast_jacobian! (src/mna/precompile.jl:493-585), a KLU-style refactor/solve with a fixed pivot
// sequence (SuiteSparse KLU is third-party and absent; klu_refactor semantics), the PCNR DC
// Newton (src/mna/solve.jl:599-698) and the variable-step BDF1/BDF2 transient driver described
// in cadnip.jl_amd/csrc/driver.hip (the reference's integrator is Sundials IDA, third-party and
// absent).  Used (a) as the full-size parity oracle for the GPU path -- validated itself against
// the literal Python oracle (oracle/mna_ref.py, oracle/va_mos1_ref.py) in tests/ -- and (b) as
// bench.py's cpu_baseline (kind "port", 1 core).  Never linked into or called by the product.
//
// Build: g++ -O3 -march=native -shared -fPIC -o oracle/_build/libcpu_port.so oracle/cpu_port.cpp
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

// x^(-1/(ord+1)) of the step-size rule: operation for operation the product's step_root (csrc/tran_ctrl.hpp) -- IEEE sqrt and division
// for order 1; for order 2 exact frexp / ldexp, products and fused multiply-adds only, so that both sides get the same double.
static double step_root(double x, int ord) {
  if (ord == 1) return 1.0 / std::sqrt(x);
  if (ord == 3) return 1.0 / std::sqrt(std::sqrt(x));
  int e;
  double m = std::frexp(x, &e);
  const int q = (e >= 0 ? e : e - 2) / 3;
  m = std::ldexp(m, e - 3 * q);
  double y = std::fma(m, -0.17, 1.18);
  for (int it = 0; it < 4; ++it) {
    const double y3 = (y * y) * y;
    y = (y * std::fma(-m, y3, 4.0)) * (1.0 / 3.0);
  }
  return std::ldexp(y, -q);
}

const double CS = 1e12;  // contrib.jl:39

struct Block {
  int type, count, n_nodes, n_ipar, n_par, g_base, c_base, b_base, n_g, n_c, n_b;
  std::vector<int> nodes, ipar;
  std::vector<double> par;  // [n_par][count]
};

struct LU {
  int nnz_lu = 0;
  std::vector<int> rperm, cperm, rowptr, col, diag, load_dst, ent_pos, ent_diag, ent_ptr, term_a, term_b;
  std::vector<double> v, y;
};

struct Port {
  int n, n_nodes, n_limits, nnz, ns_g, ns_c, ns_b;
  std::vector<int> rowptr, colidx, g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots, diag_nz;
  std::vector<double> limit_init, wave;
  std::vector<Block> blocks;
  int mode = 1, initjct = 0;
  double gmin = 1e-12, gshunt = 0, srcFact = 1;
  std::vector<double> S, G, C, b, limit_w;
  LU lu;
  bool has_lu = false;
  // External stamping: when set, fast_rebuild! is this callback (u, t -> G, C in the structure's CSR order, b, limit_w) instead of the
  // port's own device code -- how a circuit of generated Verilog-A models (which the port does not know) gets a transient oracle: the
  // literal Python interpreter (oracle/va_ref.py through oracle/mna_ref.py's fast_rebuild!) stamps, the port's controller and LU run.
  void (*ext_stamp)(const double* u, double t, double* G, double* C, double* b, double* limit_w) = nullptr;
};

// ---- waves (devices.jl:30-103,155-203) ---------------------------------------------------------
double pwl(const double* ts, const double* ys, int n, double t) {
  int i = (int)(std::lower_bound(ts, ts + n, t) - ts) + 1;
  if (i <= n && ts[i - 1] == t) i += 1;
  if (i <= 1) return ys[0];
  if (i > n) return ys[n - 1];
  if (ys[i - 2] == ys[i - 1]) return ys[i - 1];
  if (ts[i - 1] == ts[i - 2]) return (ys[i - 2] + ys[i - 1]) / 2;
  return ys[i - 2] + (t - ts[i - 2]) * ((ys[i - 1] - ys[i - 2]) / (ts[i - 1] - ts[i - 2]));
}
double pulse(const double* w, double t) {
  double v1 = w[0], v2 = w[1], td = w[2], tr = w[3], tf = w[4], pw = w[5], per = w[6];
  if (t < td) return v1;
  double ph = per > 0 ? std::fmod(t - td, per) : t - td;
  if (per > 0 && ph < 0) ph += per;
  if (ph < tr) return tr > 0 ? v1 + (v2 - v1) * (ph / tr) : v2;
  if (ph < tr + pw) return v2;
  if (ph < tr + pw + tf) return tf > 0 ? v2 + (v1 - v2) * ((ph - tr - pw) / tf) : v1;
  return v1;
}
double sind(double deg) {
  double r = std::fmod(deg, 360.0);
  if (r == 0 || r == 180 || r == -180) return 0;
  if (r == 90 || r == -270) return 1;
  if (r == -90 || r == 270) return -1;
  return std::sin(r * (3.14159265358979323846 / 180.0));
}
double source(const Port& P, const Block& B, int d, double t) {
  double dc = B.par[0 * B.count + d], scale = B.par[1 * B.count + d];
  int kind = B.ipar[d];
  if (kind == 0 || P.mode == 0) return dc;
  const double* w = P.wave.data() + B.ipar[B.count + d];
  int len = B.ipar[2 * B.count + d];
  double v;
  if (kind == 1) v = pwl(w, w + len, len, t);
  else if (kind == 2) v = pulse(w, t);
  else v = t < w[3] ? w[0] + w[1] * sind(w[5]) : w[0] + w[1] * std::exp(-w[4] * (t - w[3])) * sind(360 * w[2] * (t - w[3]) + w[5]);
  return scale * v;
}

// behavioural source value: postfix program in the wave table (include/cadnip_hip.h CadnipBsrcOp), devices.jl:1079-1131
double bsrc(const Port& P, const Block& B, int d, const double* u, double t) {
  const double* pr = P.wave.data() + B.ipar[d];
  const int len = B.ipar[B.count + d];
  std::vector<double> st;
  for (int i = 0; i < len;) {
    int op = (int)pr[i++];
    if (op == 0) st.push_back(pr[i++]);
    else if (op == 1) { int a = (int)pr[i], b = (int)pr[i + 1]; i += 2; st.push_back((a < 0 ? 0.0 : u[a]) - (b < 0 ? 0.0 : u[b])); }
    else if (op == 2) st.push_back(t);
    else if (op < 20) {
      double y = st.back(); st.pop_back();
      double x = st.back(), r;
      switch (op) { case 10: r = x + y; break; case 11: r = x - y; break; case 12: r = x * y; break; case 13: r = x / y; break;
                    case 14: r = std::pow(x, y); break; case 15: r = std::fmin(x, y); break; default: r = std::fmax(x, y); }
      st.back() = r;
    } else {
      double x = st.back(), r;
      switch (op) { case 20: r = -x; break; case 21: r = std::exp(x); break; case 22: r = std::log(x); break; case 23: r = std::sqrt(x); break;
                    case 24: r = std::fabs(x); break; case 25: r = std::tanh(x); break; case 26: r = std::sin(x); break; default: r = std::cos(x); }
      st.back() = r;
    }
  }
  return B.par[d] * st[0];
}

// ---- 3-wide dual (JacobianTag dual restricted to vgs, vds, vbs) ---------------------------------
struct D3 { double v, a, b, c; };
inline D3 mk(double v) { return {v, 0, 0, 0}; }
inline D3 operator+(D3 x, D3 y) { return {x.v + y.v, x.a + y.a, x.b + y.b, x.c + y.c}; }
inline D3 operator-(D3 x, D3 y) { return {x.v - y.v, x.a - y.a, x.b - y.b, x.c - y.c}; }
inline D3 operator*(D3 x, D3 y) { return {x.v * y.v, x.a * y.v + y.a * x.v, x.b * y.v + y.b * x.v, x.c * y.v + y.c * x.v}; }
inline D3 operator/(D3 x, D3 y) { double q = x.v / y.v; return {q, (x.a - q * y.a) / y.v, (x.b - q * y.b) / y.v, (x.c - q * y.c) / y.v}; }
inline D3 operator+(D3 x, double s) { return {x.v + s, x.a, x.b, x.c}; }
inline D3 operator+(double s, D3 x) { return x + s; }
inline D3 operator-(D3 x, double s) { return {x.v - s, x.a, x.b, x.c}; }
inline D3 operator-(double s, D3 x) { return {s - x.v, -x.a, -x.b, -x.c}; }
inline D3 operator*(D3 x, double s) { return {x.v * s, x.a * s, x.b * s, x.c * s}; }
inline D3 operator*(double s, D3 x) { return x * s; }
inline D3 operator/(D3 x, double s) { return {x.v / s, x.a / s, x.b / s, x.c / s}; }
inline D3 operator/(double s, D3 x) { double q = s / x.v, f = -q / x.v; return {q, f * x.a, f * x.b, f * x.c}; }
inline D3 dsqrt(D3 x) { double s = std::sqrt(x.v), f = 0.5 / s; return {s, x.a * f, x.b * f, x.c * f}; }
inline D3 dexp(D3 x) { double e = std::exp(x.v); return {e, x.a * e, x.b * e, x.c * e}; }
inline D3 dlog(D3 x) { return {std::log(x.v), x.a / x.v, x.b / x.v, x.c / x.v}; }

// ---- mos1 limiters (mos1.va:503-635) -------------------------------------------------------------
double fetlim(double vnew, double vold, double vto) {
  double vl = vnew, hi = std::fabs(2 * (vold - vto)) + 2, lo = std::fabs(vold - vto) + 1, vtox = vto + 3.5, dv = vnew - vold;
  if (vold >= vto) {
    if (vold >= vtox) {
      if (dv <= 0) { if (vl >= vtox) { if (-dv > lo) vl = vold - lo; } else vl = std::max(vnew, vto + 2); }
      else if (dv >= hi) vl = vold + hi;
    } else vl = dv <= 0 ? std::max(vnew, vto - 0.5) : std::min(vnew, vto + 4);
  } else {
    if (dv <= 0) { if (-dv > hi) vl = vold - hi; }
    else { double vt = vto + 0.5; if (vnew <= vt) { if (dv > lo) vl = vold + lo; } else vl = vt; }
  }
  return vl;
}
double limvds(double vnew, double vold) {
  if (vold >= 3.5) { if (vnew > vold) return std::min(vnew, 3 * vold + 2); return vnew < 3.5 ? std::max(vnew, 2.0) : vnew; }
  return vnew > vold ? std::min(vnew, 4.0) : std::max(vnew, -0.5);
}
double pnjlim_va(double vnew, double vold, double vt, double vcrit) {
  if (vnew > vcrit && std::fabs(vnew - vold) > vt + vt) {
    if (vold > 0) { double a = (vnew - vold) / vt; return a > 0 ? vold + vt * std::log(1 + a) : vold - vt * std::log(1 - a); }
    return vt * std::log(vnew / vt);
  }
  if (vnew < 0) { double a = vold > 0 ? -vold - 1 : 2 * vold - 1; if (vnew < a) return a; }
  return vnew;
}
double pnjlim_native(double vnew, double vold, double vt, double vcrit) {   // devices.jl:1169-1189
  if (vnew > vcrit && std::fabs(vnew - vold) > vt + vt) {
    if (vold > 0) { double a = (vnew - vold) / vt; return a > 0 ? vold + vt * (2 + std::log(a - 2)) : vold - vt * (2 + std::log(2 - a)); }
    return vt * std::log(vnew / vt);
  }
  if (vnew < 0) { double a = vold > 0 ? -vold - 1 : 2 * vold - 1; if (vnew < a) return a; }
  return vnew;
}
void qmeyer(D3 vgs, D3 vgd, D3 von, D3 vdsat, double phi, double cox, D3& cgs, D3& cgd, D3& cgb) {   // mos1.va:401-465
  D3 vgst = vgs - von;
  if (!(vdsat.v > 0.025)) vdsat = mk(0.025);
  if (vgst.v <= -phi) { cgb = mk(cox / 2); cgs = mk(0); cgd = mk(0); return; }
  if (vgst.v <= -phi / 2) { cgb = (-1.0 * vgst) * cox / (2 * phi); cgs = mk(0); cgd = mk(0); return; }
  if (vgst.v <= 0) {
    cgb = (-1.0 * vgst) * cox / (2 * phi);
    cgs = vgst * cox / (1.5 * phi) + cox / 3;
    D3 vds = vgs - vgd;
    if (vds.v >= vdsat.v) { cgd = mk(0); return; }
    D3 d = 2.0 * vdsat - vds, d1 = vdsat - vds, d2 = d * d;
    cgd = cgs * (1.0 - vdsat * vdsat / d2);
    cgs = cgs * (1.0 - d1 * d1 / d2);
    return;
  }
  D3 vds = vgs - vgd;
  if (vdsat.v <= vds.v) { cgs = mk(cox / 3); cgd = mk(0); cgb = mk(0); return; }
  D3 d = 2.0 * vdsat - vds, d1 = vdsat - vds, d2 = d * d;
  cgd = cox * (1.0 - vdsat * vdsat / d2) / 3.0;
  cgs = cox * (1.0 - d1 * d1 / d2) / 3.0;
  cgb = mk(0);
}
D3 qdep(D3 v, double Cb, double Cbsw, double pot, double dep, double mj, double mjsw, double f2, double f3, double f4) {   // mos1.va:1049-1109
  if (Cb == 0 && Cbsw == 0) return mk(0);
  if (v.v < dep) {
    D3 arg = 1.0 - v / pot;
    D3 s = mj == 0.5 ? 1.0 / dsqrt(arg) : dexp(-mj * dlog(arg));
    D3 ssw = mjsw == mj ? s : (mjsw == 0.5 ? 1.0 / dsqrt(arg) : dexp(-mjsw * dlog(arg)));
    return pot * (Cb * (1.0 - arg * s) / (1 - mj) + Cbsw * (1.0 - arg * ssw) / (1 - mjsw));
  }
  return f4 + v * (f2 + v * (f3 / 2));
}

enum { T_TYPE = 0, T_VT, T_TPHI, T_TVBI, T_TVTO, T_GAMMA, T_LAMBDA, T_BETA, T_OXCAP, T_SSAT, T_DSAT, T_SVCRIT, T_DVCRIT, T_CBS, T_CBSSW, T_CBD,
       T_CBDSW, T_POT, T_DEP, T_F2S, T_F3S, T_F4S, T_F2D, T_F3D, T_F4D, T_MJ, T_MJSW, T_CGSO, T_CGDO, T_CGBO, T_GD, T_GS, T_MF, T_GMIN };

struct Slots {
  double *g, *c, *b; int count, dev;
  void G(int k, double v) const { g[k * count + dev] = v; }
  void C(int k, double v) const { c[k * count + dev] = v; }
  void B(int k, double v) const { b[k * count + dev] = v; }
};

void stamp_mos1(Port& P, const Block& B, int d, const double* u, const Slots& s) {
  auto nd = [&](int k) { return B.nodes[k * B.count + d]; };
  auto par = [&](int k) { return B.par[k * B.count + d]; };
  auto V = [&](int node) { return node < 0 ? 0.0 : u[node]; };
  const double Vd = V(nd(0)), Vg = V(nd(1)), Vs = V(nd(2)), Vb = V(nd(3)), Vdi = V(nd(4)), Vsi = V(nd(5));
  const int l0 = nd(6), l1 = nd(7), l2 = nd(8), l3 = nd(9);
  const double ty = par(T_TYPE), vt = par(T_VT), tPhi = par(T_TPHI), tVbi = par(T_TVBI), gam = par(T_GAMMA), lam = par(T_LAMBDA);
  const double Beta = par(T_BETA), ox = par(T_OXCAP), mf = par(T_MF), gmin = par(T_GMIN) / mf;
  // previous (vold) values and von (mos1.va:919-943)
  double ovgs = ty * u[l0], ovds = ty * u[l1], ovbs = ty * u[l2], ovbd = ty * u[l3];
  double osel = ovds >= 0 ? ovbs : ovbd, osarg;
  if (osel <= 0) osarg = std::sqrt(tPhi - osel);
  else { osarg = std::sqrt(tPhi); osarg = std::max(0.0, osarg - ovbs / (osarg + osarg)); }
  double von = ty * ((tVbi * ty) + gam * osarg);
  double vbs = ty * (Vb - Vsi), vgs = ty * (Vg - Vsi), vds = ty * (Vdi - Vsi), vbd = vbs - vds, vgd = vgs - vds, vgdo = ovgs - ovds;
  if (ovds >= 0) { vgs = fetlim(vgs, ovgs, von); vds = vgs - vgd; vds = limvds(vds, ovds); vgd = vgs - vds; }
  else { vgd = fetlim(vgd, vgdo, von); vds = vgs - vgd; vds = -limvds(-vds, -ovds); vgs = vgd + vds; }
  if (vds >= 0) { vbs = pnjlim_va(vbs, ovbs, vt, par(T_SVCRIT)); vbd = vbs - vds; }
  else { vbd = pnjlim_va(vbd, ovbd, vt, par(T_DVCRIT)); vbs = vbd + vds; }
  if (P.initjct) { vbs = -1; vgs = ty * par(T_TVTO); vds = 0; vbd = vbs - vds; }
  const double wgs = ty * vgs, wds = ty * vds, wbs = ty * vbs, wbd = ty * vbd;
  P.limit_w[l0] = wgs; P.limit_w[l1] = wds; P.limit_w[l2] = wbs; P.limit_w[l3] = wbd;
  for (int lb = 0; lb < 4; ++lb) { s.G(3 * lb, 1.0); s.G(3 * lb + 1, -1.0); s.G(3 * lb + 2, 1.0); }
  // evaluation at the limited voltages with pass-through partials (vasim.jl:1319-1330)
  D3 a = {ty * wgs, 1, 0, 0}, b = {ty * wds, 0, 1, 0}, c = {ty * wbs, 0, 0, 1};
  D3 xbd = c - b, xgd = a - b, xgb = a - c, cbs, cbd;
  if (c.v <= -3 * vt) cbs = gmin * c - par(T_SSAT);
  else { D3 x = c / vt; D3 e = dexp(709.0 < x.v ? mk(709.0) : x); cbs = par(T_SSAT) * (e - 1.0) + gmin * c; }
  if (xbd.v <= -3 * vt) cbd = gmin * xbd - par(T_DSAT);
  else { D3 x = xbd / vt; D3 e = dexp(709.0 < x.v ? mk(709.0) : x); cbd = par(T_DSAT) * (e - 1.0) + gmin * xbd; }
  int mode = b.v >= 0 ? 1 : -1;
  D3 sel = mode == 1 ? c : xbd, sarg;
  if (sel.v <= 0) sarg = dsqrt(tPhi - sel);
  else { double s0 = std::sqrt(tPhi); sarg = s0 - sel / (s0 + s0); if (0 > sarg.v) sarg = mk(0); }
  D3 lvon = tVbi * ty + gam * sarg;
  D3 vgst = (mode == 1 ? a : xgd) - lvon;
  D3 vdsat = vgst.v > 0 ? vgst : mk(0), cdrain = mk(0);
  if (vgst.v > 0) {
    D3 vm = b * (double)mode, bp = Beta * (1.0 + lam * vm);
    cdrain = vgst.v <= vm.v ? bp * vgst * vgst * 0.5 : bp * vm * (vgst - 0.5 * vm);
  }
  double ms = ox == 0 ? 0.0 : ox, mu = ox == 0 ? 1.0 : ox;
  D3 qbs = qdep(c, par(T_CBS), par(T_CBSSW), par(T_POT), par(T_DEP), par(T_MJ), par(T_MJSW), par(T_F2S), par(T_F3S), par(T_F4S));
  D3 qbd = qdep(xbd, par(T_CBD), par(T_CBDSW), par(T_POT), par(T_DEP), par(T_MJ), par(T_MJSW), par(T_F2D), par(T_F3D), par(T_F4D));
  D3 mgs, mgd, mgb;
  if (mode > 0) qmeyer(a, xgd, lvon, vdsat, tPhi, ox, mgs, mgd, mgb);
  else qmeyer(xgd, a, lvon, vdsat, tPhi, ox, mgd, mgs, mgb);
  D3 cgs = mgs + mgs + par(T_CGSO), cgd = mgd + mgd + par(T_CGDO), cgb = mgb + mgb + par(T_CGBO);
  D3 qgs = cgs * ((ms * a) / mu), qgd = cgd * ((ms * xgd) / mu), qgb = cgb * ((ms * xgb) / mu);
  D3 cdreq = (mode >= 0 ? ty : -ty) * cdrain;
  D3 Ir[6] = {mk(0), mk(0), mk(0), ty * cbs + ty * cbd, -1.0 * (ty * cbd - cdreq), -1.0 * (cdreq + ty * cbs)};
  D3 q[4] = {ty * (qgs + qgb + qgd), (ty * qbs + ty * qbd) - ty * qgb, -1.0 * (ty * qbd + ty * qgd), -1.0 * (ty * qbs + ty * qgs)};
  const double gd = par(T_GD), gs = par(T_GS);
  const double dWgs = (Vg - Vsi) - wgs, dWds = (Vdi - Vsi) - wds, dWbs = (Vb - Vsi) - wbs;
  const double Vk[6] = {Vd, Vg, Vs, Vb, Vdi, Vsi};
  for (int br = 0; br < 6; ++br) {
    double fa = ty * Ir[br].a, fb = ty * Ir[br].b, fc = ty * Ir[br].c;
    double dI[6] = {0, fa, 0, fc, fb, -(fa + fb + fc)}, Iv = Ir[br].v;
    if (br == 0) { Iv += gd * (Vd - Vdi); dI[0] += gd; dI[4] -= gd; }
    if (br == 2) { Iv += gs * (Vs - Vsi); dI[2] += gs; dI[5] -= gs; }
    if (br == 4) { Iv += gd * (Vdi - Vd); dI[4] += gd; dI[0] -= gd; }
    if (br == 5) { Iv += gs * (Vsi - Vs); dI[5] += gs; dI[2] -= gs; }
    double Ieq = mf * Iv;
    for (int k = 0; k < 6; ++k) { double g = mf * dI[k]; s.G(12 + 6 * br + k, g); Ieq = Ieq + (-g * Vk[k]); }
    Ieq = Ieq + (mf * fa) * dWgs; Ieq = Ieq + (mf * fb) * dWds; Ieq = Ieq + (mf * fc) * dWbs;
    s.B(br, -Ieq);
  }
  for (int r = 0; r < 4; ++r) {
    double fa = mf * ty * q[r].a, fb = mf * ty * q[r].b, fc = mf * ty * q[r].c;
    double dq[6] = {0, fa, 0, fc, fb, -(fa + fb + fc)};
    s.C(r, 1.0 / CS);
    s.G(48 + 7 * r, 1.0);
    double bc = mf * q[r].v;
    for (int k = 0; k < 6; ++k) { s.G(48 + 7 * r + 1 + k, -CS * dq[k]); bc -= dq[k] * Vk[k]; }
    bc += fa * dWgs; bc += fb * dWds; bc += fc * dWbs;
    s.B(6 + r, CS * bc);
    for (int k = 0; k < 6; ++k) s.C(4 + 6 * r + k, dq[k]);
  }
}

void rebuild(Port& P, const double* u, double t) {
  if (P.ext_stamp) { P.ext_stamp(u, t, P.G.data(), P.C.data(), P.b.data(), P.limit_w.data()); return; }
  for (auto& B : P.blocks) {
    for (int d = 0; d < B.count; ++d) {
      Slots s{P.S.data() + B.g_base, P.S.data() + P.ns_g + B.c_base, P.S.data() + P.ns_g + P.ns_c + B.b_base, B.count, d};
      auto nd = [&](int k) { return B.nodes[k * B.count + d]; };
      auto par = [&](int k) { return B.par[k * B.count + d]; };
      auto V = [&](int node) { return node < 0 ? 0.0 : u[node]; };
      auto cond4 = [&](int k0, double g) { s.G(k0, g); s.G(k0 + 1, -g); s.G(k0 + 2, -g); s.G(k0 + 3, g); };
      auto cap4 = [&](int k0, double c) { s.C(k0, c); s.C(k0 + 1, -c); s.C(k0 + 2, -c); s.C(k0 + 3, c); };
      auto br4 = [&]() { s.G(0, 1); s.G(1, -1); s.G(2, 1); s.G(3, -1); };
      switch (B.type) {
        case 0: cond4(0, par(0)); break;
        case 1: cap4(0, par(0)); break;
        case 2: br4(); s.C(0, -par(0)); break;
        case 3: br4(); s.B(0, source(P, B, d, t)); break;
        case 4: { double i = source(P, B, d, t); s.B(0, i); s.B(1, -i); } break;
        case 5: { double a = par(0); s.G(0, 1); s.G(1, -1); s.G(2, 1); s.G(3, -1); s.G(4, -a); s.G(5, a); } break;
        case 6: { double g = par(0); s.G(0, -g); s.G(1, g); s.G(2, g); s.G(3, -g); } break;
        case 7: s.G(0, 1); s.G(1, -1); s.G(2, 1); s.G(3, -1); s.G(4, 1); s.G(5, -1); s.G(6, 1); s.G(7, -1); s.G(8, -par(0)); break;
        case 8: { double a = par(0); s.G(0, 1); s.G(1, -1); s.G(2, 1); s.G(3, -1); s.G(4, -a); s.G(5, a); } break;
        case 9: {   // Diode devices.jl:1370-1428
          double Is = par(0), nVt = par(1), vcrit = par(2), V0 = V(nd(0)) - V(nd(1)), I0, Gd, Ieq;
          if (B.ipar[d]) {
            int l = nd(2);
            double w = P.initjct ? vcrit : pnjlim_native(V0, u[l], nVt, vcrit);
            P.limit_w[l] = w;
            s.G(0, 1); s.G(1, -1); s.G(2, 1);
            double x = w / nVt;
            if (x > 80) { double e80 = std::exp(80.0); I0 = Is * (e80 * (1 + (x - 80)) - 1); Gd = Is / nVt * e80; }
            else { double e = std::exp(x); I0 = Is * (e - 1); Gd = Is / nVt * e; }
            Ieq = I0 - Gd * w;
          } else {
            s.G(0, 0); s.G(1, 0); s.G(2, 0);
            double e = std::exp(V0 / nVt); I0 = Is * (e - 1); Gd = Is / nVt * e; Ieq = I0 - Gd * V0;
          }
          cond4(3, Gd); s.B(0, -Ieq); s.B(1, Ieq);
        } break;
        case 10: {  // DiodeWithCap devices.jl:1558-1602
          double Is = par(0), nVt = par(1), Cj0 = par(2), Vj = par(3), m = par(4), V0 = V(nd(0)) - V(nd(1));
          double e = std::exp(V0 / nVt), I0 = Is * (e - 1), G = Is / nVt * e, Ieq = I0 - G * V0;
          cond4(0, G); s.B(0, -Ieq); s.B(1, Ieq);
          double Vmax = 0.9 * Vj, Cc;
          if (V0 < Vmax) Cc = Cj0 / std::pow(1 - V0 / Vj, m);
          else Cc = Cj0 / std::pow(1 - Vmax / Vj, m) + Cj0 * m / Vj / std::pow(1 - Vmax / Vj, m + 1) * (V0 - Vmax);
          cap4(0, Cc);
        } break;
        case 11: {  // SimpleMOSFET devices.jl:1667-1749
          double Vd = V(nd(0)), Vg = V(nd(1)), Vs = V(nd(2)), Vth = par(0), K = par(1), lam = par(2);
          double Vgs = Vg - Vs, Vds = Vd - Vs, Ids, gm, gds;
          if (Vgs <= Vth) Ids = gm = gds = 0;
          else if (Vds <= Vgs - Vth) { Ids = K * ((Vgs - Vth) * Vds - Vds * Vds / 2); gm = K * Vds; gds = K * (Vgs - Vth - Vds); }
          else { double ov = Vgs - Vth; Ids = K / 2 * (ov * ov) * (1 + lam * Vds); gm = K * ov * (1 + lam * Vds); gds = K / 2 * (ov * ov) * lam; }
          double Ieq = Ids - gm * Vgs - gds * Vds;
          s.G(0, gds); s.G(1, gm); s.G(2, -(gds + gm)); s.G(3, -gds); s.G(4, -gm); s.G(5, gds + gm);
          s.B(0, -Ieq); s.B(1, Ieq);
          cap4(0, par(4)); cap4(4, par(3));
        } break;
        case 12: stamp_mos1(P, B, d, u, s); break;
        case 13: br4(); s.B(0, bsrc(P, B, d, u, t)); break;
        case 14: { double i = bsrc(P, B, d, u, t); s.B(0, i); s.B(1, -i); } break;
      }
    }
  }
  // slot -> nz gather in COO order == nzval[map[pos]] += v  (value_only.jl:414-418), then precompile.jl:508-534
  const double* Sg = P.S.data(); const double* Sc = Sg + P.ns_g; const double* Sb = Sc + P.ns_c;
  for (int e = 0; e < P.nnz; ++e) {
    double acc = 0; for (int p = P.g_ptr[e]; p < P.g_ptr[e + 1]; ++p) acc += Sg[P.g_slots[p]]; P.G[e] = acc;
    acc = 0; for (int p = P.c_ptr[e]; p < P.c_ptr[e + 1]; ++p) acc += Sc[P.c_slots[p]]; P.C[e] = acc;
  }
  for (int i = 0; i < P.n; ++i) {
    double acc = 0; for (int p = P.b_ptr[i]; p < P.b_ptr[i + 1]; ++p) acc += Sb[P.b_slots[p]];
    if (P.srcFact < 1.0) acc *= P.srcFact;
    P.b[i] = acc;
  }
  if (P.gshunt != 0.0) for (int i = 0; i < P.n_nodes; ++i) if (P.diag_nz[i] >= 0) P.G[P.diag_nz[i]] += P.gshunt;
}

void residual(const Port& P, const double* u, const double* du, double* r) {   // precompile.jl:546-557
  for (int i = 0; i < P.n; ++i) {
    double aC = 0, aG = 0;
    for (int p = P.rowptr[i]; p < P.rowptr[i + 1]; ++p) { int j = P.colidx[p]; aC += P.C[p] * du[j]; aG += P.G[p] * u[j]; }
    r[i] = (aC + aG) - P.b[i];
  }
}

// numeric refactor with the fixed pivot sequence + solve; returns false on a zero / non-finite pivot
bool lu_solve_kept(Port& P, const double* rhs, double* x);
bool lu_factor(Port& P, double gamma);
bool factor_solve(Port& P, double gamma, const double* rhs, double* x) { return lu_factor(P, gamma) && lu_solve_kept(P, rhs, x); }
bool lu_factor(Port& P, double gamma) {
  LU& L = P.lu;
  std::fill(L.v.begin(), L.v.end(), 0.0);
  for (int k = 0; k < P.nnz; ++k) L.v[L.load_dst[k]] = P.G[k] + gamma * P.C[k];
  const int ne = (int)L.ent_pos.size();
  for (int e = 0; e < ne; ++e) {   // entries are sorted by dependency level, so sequential order is valid
    double acc = L.v[L.ent_pos[e]];
    for (int t = L.ent_ptr[e]; t < L.ent_ptr[e + 1]; ++t) acc -= L.v[L.term_a[t]] * L.v[L.term_b[t]];
    if (L.ent_diag[e] >= 0) acc /= L.v[L.ent_diag[e]];
    L.v[L.ent_pos[e]] = acc;
  }
  const int n = P.n;
  for (int i = 0; i < n; ++i) { double d = L.v[L.diag[i]]; if (d == 0.0 || !std::isfinite(d)) return false; }
  return true;
}
// the two triangular sweeps on the factors lu_factor left (klu_solve after klu_refactor; Newton mode 1 keeps them over several rounds)
bool lu_solve_kept(Port& P, const double* rhs, double* x) {
  LU& L = P.lu;
  const int n = P.n;
  for (int i = 0; i < n; ++i) L.y[i] = rhs[L.rperm[i]];
  for (int i = 0; i < n; ++i) { double acc = L.y[i]; for (int p = L.rowptr[i]; p < L.diag[i]; ++p) acc -= L.v[p] * L.y[L.col[p]]; L.y[i] = acc; }
  for (int i = n - 1; i >= 0; --i) { double acc = L.y[i]; for (int p = L.diag[i] + 1; p < L.rowptr[i + 1]; ++p) acc -= L.v[p] * L.y[L.col[p]]; L.y[i] = acc / L.v[L.diag[i]]; }
  for (int i = 0; i < n; ++i) x[L.cperm[i]] = L.y[i];
  return true;
}

}  // namespace

extern "C" {

void* port_create(int n, int n_nodes, int n_limits, int nnz, const int* rowptr, const int* colidx, int ns_g, int ns_c, int ns_b,
                  const int* g_ptr, const int* g_slots, const int* c_ptr, const int* c_slots, const int* b_ptr, const int* b_slots,
                  const int* diag_nz, const double* limit_init, int n_wave, const double* wave) {
  Port* P = new Port();
  P->n = n; P->n_nodes = n_nodes; P->n_limits = n_limits; P->nnz = nnz; P->ns_g = ns_g; P->ns_c = ns_c; P->ns_b = ns_b;
  P->rowptr.assign(rowptr, rowptr + n + 1); P->colidx.assign(colidx, colidx + nnz);
  P->g_ptr.assign(g_ptr, g_ptr + nnz + 1); P->g_slots.assign(g_slots, g_slots + g_ptr[nnz]);
  P->c_ptr.assign(c_ptr, c_ptr + nnz + 1); P->c_slots.assign(c_slots, c_slots + c_ptr[nnz]);
  P->b_ptr.assign(b_ptr, b_ptr + n + 1); P->b_slots.assign(b_slots, b_slots + b_ptr[n]);
  P->diag_nz.assign(diag_nz, diag_nz + n_nodes); P->limit_init.assign(limit_init, limit_init + n_limits);
  P->wave.assign(wave, wave + n_wave);
  P->S.assign((size_t)ns_g + ns_c + ns_b, 0.0); P->G.assign(nnz, 0.0); P->C.assign(nnz, 0.0); P->b.assign(n, 0.0); P->limit_w.assign(n, 0.0);
  return P;
}
void port_destroy(void* p) { delete (Port*)p; }
void port_add_block(void* p, int type, int count, int n_nodes, const int* nodes, int n_ipar, const int* ipar, int n_par, const double* par,
                    int g_base, int c_base, int b_base, int n_g, int n_c, int n_b) {
  Port* P = (Port*)p;
  Block B{type, count, n_nodes, n_ipar, n_par, g_base, c_base, b_base, n_g, n_c, n_b, {}, {}, {}};
  B.nodes.assign(nodes, nodes + (size_t)n_nodes * count);
  B.ipar.assign(ipar, ipar + (size_t)std::max(n_ipar, 1) * count);
  B.par.assign(par, par + (size_t)n_par * count);
  P->blocks.push_back(B);
}
void port_set_stamp_callback(void* p, void (*cb)(const double*, double, double*, double*, double*, double*)) { ((Port*)p)->ext_stamp = cb; }
void port_set_spec(void* p, int mode, double gmin, double gshunt, double srcFact, int initjct) {
  Port* P = (Port*)p; P->mode = mode; P->gmin = gmin; P->gshunt = gshunt; P->srcFact = srcFact; P->initjct = initjct;
}
void port_set_lu(void* p, int nnz_lu, const int* rperm, const int* cperm, const int* rowptr, const int* col, const int* diag, const int* load_src,
                 const int* load_dst, int n_ent, const int* ent_pos, const int* ent_diag, const int* ent_ptr, const int* term_a, const int* term_b) {
  Port* P = (Port*)p; LU& L = P->lu; int n = P->n;
  L.nnz_lu = nnz_lu; L.rperm.assign(rperm, rperm + n); L.cperm.assign(cperm, cperm + n); L.rowptr.assign(rowptr, rowptr + n + 1);
  L.col.assign(col, col + nnz_lu); L.diag.assign(diag, diag + n);
  L.load_dst.assign(P->nnz, 0); for (int k = 0; k < P->nnz; ++k) L.load_dst[load_src[k]] = load_dst[k];
  L.ent_pos.assign(ent_pos, ent_pos + n_ent); L.ent_diag.assign(ent_diag, ent_diag + n_ent); L.ent_ptr.assign(ent_ptr, ent_ptr + n_ent + 1);
  L.term_a.assign(term_a, term_a + ent_ptr[n_ent]); L.term_b.assign(term_b, term_b + ent_ptr[n_ent]);
  L.v.assign(nnz_lu, 0.0); L.y.assign(n, 0.0); P->has_lu = true;
}
// fast_rebuild!: outputs in CSR order of the structure (any pointer may be null)
void port_rebuild(void* p, const double* u, double t, double* G, double* C, double* b, double* limit_w) {
  Port* P = (Port*)p; rebuild(*P, u, t);
  if (G) memcpy(G, P->G.data(), P->nnz * sizeof(double)); if (C) memcpy(C, P->C.data(), P->nnz * sizeof(double));
  if (b) memcpy(b, P->b.data(), P->n * sizeof(double)); if (limit_w) memcpy(limit_w, P->limit_w.data(), P->n * sizeof(double));
}
int port_factor_solve(void* p, double gamma, const double* rhs, double* x) { return factor_solve(*(Port*)p, gamma, rhs, x) ? 0 : 2; }

// PCNR / plain DC Newton (solve.jl:599-698, 542-578).  returns 1 = converged; *iters = Newton solves
int port_dc(void* p, double* u, double abstol, int maxiters, int use_pcnr, int cold_start, int* iters) {
  Port& P = *(Port*)p; const int n = P.n, L = P.n_limits, l0 = n - L;
  const bool pcnr = use_pcnr && L > 0;
  std::vector<double> F(n), d(n), du(n, 0.0);
  bool allzero = true; for (int i = 0; i < n; ++i) if (u[i] != 0.0) allzero = false;
  int saved = P.initjct; P.initjct = 0;
  if (pcnr && cold_start && allzero) { for (int k = 0; k < L; ++k) u[l0 + k] = P.limit_init[k]; P.initjct = 1; }
  int it = 0, state = 0, result = 0;
  for (int round = 0; round < 2 * maxiters + 4; ++round) {
    rebuild(P, u, 0.0); P.initjct = 0;
    residual(P, u, du.data(), F.data());
    double s = 0; bool bad = false; for (int i = 0; i < n; ++i) { if (!std::isfinite(F[i])) bad = true; s += F[i] * F[i]; }
    if (bad) { result = 0; break; }
    if (pcnr && state == 0 && it >= maxiters) { result = 0; break; }   // PCNR tests convergence before solves 1..maxiters only (solve.jl:630-663)
    if (std::sqrt(s) < abstol) {
      if (!pcnr) { result = 1; break; }
      if (state == 0) { for (int k = 0; k < L; ++k) u[l0 + k] = P.limit_w[l0 + k]; state = 1; continue; }
      result = 1; break;
    } else state = 0;
    if (it >= maxiters) { result = 0; break; }
    if (!factor_solve(P, 0.0, F.data(), d.data())) { result = 0; break; }
    bad = false; for (int i = 0; i < n; ++i) { if (!std::isfinite(d[i])) bad = true; u[i] -= d[i]; }
    ++it;
    if (bad) { result = 0; break; }
    if (pcnr) for (int k = 0; k < L; ++k) u[l0 + k] = P.limit_w[l0 + k];
  }
  P.initjct = saved; if (iters) *iters = it; return result;
}

struct TranOpts {
  double t0, t1, reltol; const double* abstol; const double* err_mask; double h0, hmin, hmax; int max_newton, max_order, use_pcnr; double newton_tol;
  int n_break; const double* breaks; int n_save; const double* save_t; int n_obs; const int* obs;
  int newton_mode;   // 1 = IDA's nonlinear iteration: Jacobian reuse + rate test (cadnip.jl_amd/csrc/tran_ctrl.hpp, the same policy statement for statement);
                     // 2 = the same test with a refactorisation every round (what the per-op GPU path does with newton_mode 1)
  int step_rule;     // 0 = classical step controller, 1 = IDA's eta rule (tran_ctrl.hpp, CadnipTranOpts::step_rule)
};
struct TranStats { int64_t newton_iters, accepted, rejected, newton_failures; int status; double wall_seconds; int64_t refactorisations; };

// the transient driver of cadnip.jl_amd/csrc/driver.hip, one instance, sequential
int port_tran(void* p, double* u_io, const TranOpts* o, double* out, TranStats* st, double* trace_t, int trace_cap, int* trace_n) {
  Port& P = *(Port*)p; const int n = P.n, L = P.n_limits;
  auto w0 = std::chrono::steady_clock::now();
  const double span = o->t1 - o->t0, hmax = o->hmax > 0 ? o->hmax : span / 50.0, h0 = o->h0 > 0 ? o->h0 : span * 1e-6, hmin = o->hmin > 0 ? o->hmin : span * 1e-14;
  const double ntol = o->newton_tol > 0 ? o->newton_tol : 1e-3; const int maxn = o->max_newton > 0 ? o->max_newton : 10, maxo = o->max_order > 0 ? o->max_order : 2;
  std::vector<double> emask(n, 1.0); int n_err = n;
  if (o->err_mask) { n_err = 0; for (int i = 0; i < n; ++i) { emask[i] = o->err_mask[i] != 0 ? 1.0 : 0.0; n_err += emask[i] != 0; } }
  const int n_obs = o->n_obs > 0 ? o->n_obs : n;
  std::vector<double> u(u_io, u_io + n), u0 = u, u1 = u, u2 = u, u3 = u, up(n), beta(n), du(n), r(n), delta(n);
  double t = o->t0, h = h0, hprev = h0, hpp = h0, hp3 = h0, tn = 0, a0 = 0; int nhist = 1, ord = 1, k = 0, bp = 0, si = 0, status = 0;
  int savedmode = P.mode; P.mode = 1;
  while (bp < o->n_break && o->breaks[bp] <= o->t0) ++bp;
  while (si < o->n_save && o->save_t[si] <= o->t0) { for (int j = 0; j < n_obs; ++j) out[(size_t)si * n_obs + j] = u[o->n_obs > 0 ? o->obs[j] : j]; ++si; }
  TranStats S{0, 0, 0, 0, 0, 0.0, 0}; int ntrace = 0;
  const char* dbg_env = getenv("PORT_DEBUG"); const double dbg_from = dbg_env ? atof(dbg_env) : 0.0; bool dbg = false;
  auto prepare = [&](double tt, double hh, int nh, double hp, double hq, double hr) {
    double tstop = o->t1; if (bp < o->n_break && o->breaks[bp] < tstop) tstop = o->breaks[bp];
    double rem = tstop - tt;
    if (hh >= rem * (1.0 - 1e-9)) { hh = rem; tn = tstop; } else if (2.0 * hh > rem) { hh = 0.5 * rem; tn = tt + hh; } else tn = tt + hh;
    if (nh <= 1) { ord = 1; a0 = 1.0 / hh; for (int i = 0; i < n; ++i) { double pv = u0[i]; up[i] = pv; u[i] = pv; double bb = -u0[i] / hh; beta[i] = bb; du[i] = a0 * pv + bb; } }
    else if (nh == 2 || maxo < 2) { ord = 1; a0 = 1.0 / hh; double w = hh / hp; for (int i = 0; i < n; ++i) { double pv = u0[i] + w * (u0[i] - u1[i]); up[i] = pv; u[i] = pv; double bb = -u0[i] / hh; beta[i] = bb; du[i] = a0 * pv + bb; } }
    else if (nh >= 4 && maxo >= 3) {
      // variable-step BDF3 (tran_ctrl.hpp: prepare_step, operation for operation): the derivative at t_n of the cubic through (t_n, u) and the three
      // last accepted points, d1 < d2 < d3 their distances from t_n; the predictor is the cubic through the FOUR last accepted points
      ord = 3;
      const double d1 = hh, d2 = hh + hp, d3 = d2 + hq, s12 = hp + hq;
      a0 = (1.0 / d1 + 1.0 / d2) + 1.0 / d3;
      const double a1 = -((d2 * d3) / (d1 * (hp * s12))), a2 = (d1 * d3) / (d2 * (hp * hq)), a3 = -((d1 * d2) / (d3 * (s12 * hq)));
      const double x1 = -hp, x2 = -s12, x3 = -(s12 + hr), x = hh;
      const double L0 = ((x - x1) * (x - x2)) * (x - x3) / (((0.0 - x1) * (0.0 - x2)) * (0.0 - x3));
      const double L1 = ((x - 0.0) * (x - x2)) * (x - x3) / (((x1 - 0.0) * (x1 - x2)) * (x1 - x3));
      const double L2 = ((x - 0.0) * (x - x1)) * (x - x3) / (((x2 - 0.0) * (x2 - x1)) * (x2 - x3));
      const double L3 = ((x - 0.0) * (x - x1)) * (x - x2) / (((x3 - 0.0) * (x3 - x1)) * (x3 - x2));
      for (int i = 0; i < n; ++i) {
        double pv = ((L0 * u0[i] + L1 * u1[i]) + L2 * u2[i]) + L3 * u3[i]; up[i] = pv; u[i] = pv;
        double bb = (a1 * u0[i] + a2 * u1[i]) + a3 * u2[i]; beta[i] = bb; du[i] = a0 * pv + bb;
      }
    }
    else {
      ord = 2; double w = hh / hp; a0 = (1.0 + 2.0 * w) / ((1.0 + w) * hh); double a1 = -(1.0 + w) / hh, a2 = (w * w) / ((1.0 + w) * hh);
      double x1 = -hp, x2 = -(hp + hq), x = hh;
      double L0 = (x - x1) * (x - x2) / ((0.0 - x1) * (0.0 - x2)), L1 = (x - 0.0) * (x - x2) / ((x1 - 0.0) * (x1 - x2)), L2 = (x - 0.0) * (x - x1) / ((x2 - 0.0) * (x2 - x1));
      for (int i = 0; i < n; ++i) { double pv = L0 * u0[i] + L1 * u1[i] + L2 * u2[i]; up[i] = pv; u[i] = pv; double bb = a1 * u0[i] + a2 * u1[i]; beta[i] = bb; du[i] = a0 * pv + bb; }
    }
    h = hh; k = 0;
  };
  // Newton mode 1 (tran_ctrl.hpp): kept factors, their a0, the rate constant, the previous update norm, flags
  const int mode = o->newton_mode;
  double a0f = 0.0, ss = 20.0, dnp = 0.0; bool need = true, jcur = false, valid = false; int since = 0;
  auto prepare0 = prepare;
  auto prepare_attempt = [&](double tt, double hh, int nh, double hp, double hq) { prepare0(tt, hh, nh, hp, hq, hp3); jcur = false; };
  prepare_attempt(t, h, nhist, hprev, hpp);
  while (status == 0) {
    dbg = dbg_env && tn >= dbg_from;
    rebuild(P, u.data(), tn);
    residual(P, u.data(), du.data(), r.data());
    bool refresh = true; double dsc = 1.0;
    if (mode) {
      refresh = need || !valid || (k == 0 && (a0 < 0.6 * a0f || a0 * 0.6 > a0f || since >= 20));
      if (refresh) { a0f = a0; ss = 20.0; need = false; valid = true; jcur = true; since = 0; }
      if (mode == 2) { refresh = true; jcur = true; }        // the per-op GPU path: same events for the rate constant, but it refactors every round
      if (!refresh) dsc = a0 == a0f ? 1.0 : 2.0 / (1.0 + a0 / a0f);
    }
    bool ok = refresh ? factor_solve(P, a0, r.data(), delta.data()) : lu_solve_kept(P, r.data(), delta.data());
    S.refactorisations += refresh ? 1 : 0;
    S.newton_iters += 1;
    double s1 = 0, s2 = 0; bool bad = !ok;
    for (int i = 0; i < n; ++i) {
      double d = delta[i] * dsc, un = u[i] - d; if (!std::isfinite(d)) bad = true;
      double w = 1.0 / (o->abstol[i] + o->reltol * std::fabs(u0[i])); s1 += (d * w) * (d * w);
      double e = un - up[i], w2 = emask[i] / (o->abstol[i] + o->reltol * std::max(std::fabs(u0[i]), std::fabs(un))); s2 += (e * w2) * (e * w2);
      u[i] = un;
    }
    const double dnorm = std::sqrt(s1 / n);
    if (dbg) {
      int im = 0; double wm = 0;
      for (int i = 0; i < n; ++i) { double w = std::fabs(delta[i]) / (o->abstol[i] + o->reltol * std::fabs(u0[i])); if (w > wm) { wm = w; im = i; } }
      fprintf(stderr, "t=%.9e h=%.3e ord=%d k=%d dnorm=%.3e worst i=%d delta=%.3e u=%.6e bad=%d\n", tn, h, ord, k, dnorm, im, delta[im], u[im], (int)bad);
      static int dumped = 0;   // PORT_DUMP=<file>: dump (tn,h,a0,n | u | beta) of the first badly started iteration
      const char* dump_path = getenv("PORT_DUMP");
      if (dump_path && k == 0 && dnorm > 1e3 && dumped < 1) {
        FILE* f = fopen(dump_path, dumped++ == 0 ? "wb" : "ab");
        double hdr[4] = {tn, h, a0, (double)n}; fwrite(hdr, 8, 4, f); fwrite(u.data(), 8, n, f); fwrite(beta.data(), 8, n, f); fclose(f);
      }
    }
    bool conv = !bad && dnorm < ntol, diverge = false;
    if (mode) {
      conv = false;
      if (!bad) {
        if (k == 0) conv = dnorm <= 0.33e-4;
        else { const double rate = dnp > 0.0 ? dnorm / dnp : 0.0; if (rate > 0.9) diverge = true; else ss = rate / (1.0 - rate); }
        if (!diverge && ss * dnorm <= 0.33) conv = true;
        dnp = dnorm;
      }
    }
    if (conv) {
      double errn = 0; bool accept = true;
      if (nhist >= 2 && n_err > 0) {
        double errc; if (ord == 1) errc = h / (h + hprev); else if (ord == 3) errc = (1.0 / a0) / (((h + hprev) + hpp) + hp3);
        else { double w = h / hprev; errc = ((1.0 + w) * h / (1.0 + 2.0 * w)) / (h + hprev + hpp); }
        errn = errc * std::sqrt(s2 / n_err); accept = errn <= 1.0;
      }
      if (accept) {
        while (si < o->n_save && o->save_t[si] <= tn * (1.0 + 1e-15)) {
          double ts = o->save_t[si], hh = tn - t; double* oo = out + (size_t)si * n_obs;
          if (nhist >= 2) {
            double x = ts - t, xa = hh, xc = -hprev;
            double La = (x - 0.0) * (x - xc) / ((xa - 0.0) * (xa - xc)), Lb = (x - xa) * (x - xc) / ((0.0 - xa) * (0.0 - xc)), Lc = (x - xa) * (x - 0.0) / ((xc - xa) * (xc - 0.0));
            for (int j = 0; j < n_obs; ++j) { int i = o->n_obs > 0 ? o->obs[j] : j; oo[j] = La * u[i] + Lb * u0[i] + Lc * u1[i]; }
          } else { double s = (ts - t) / hh; for (int j = 0; j < n_obs; ++j) { int i = o->n_obs > 0 ? o->obs[j] : j; oo[j] = u0[i] + s * (u[i] - u0[i]); } }
          ++si;
        }
        if (maxo >= 3) u3 = u2;
        u2 = u1; u1 = u0; u0 = u;
        bool landed = bp < o->n_break && tn == o->breaks[bp];
        int nh_new = std::min(nhist + 1, maxo >= 3 ? 4 : 3); double hnext;
        if (nhist >= 2 && n_err > 0) {
          double fac;
          if (o->step_rule == 0) { fac = errn > 0.0 ? 0.9 * step_root(errn, ord) : 2.0; fac = std::min(2.0, std::max(0.2, fac)); }
          else { const double eta = errn > 0.0 ? 1.0 / (1.0 / step_root(2.0 * errn, ord) + 1e-4) : 2.0; fac = eta >= 2.0 ? 2.0 : (eta <= 1.0 ? std::max(0.5, std::min(0.9, eta)) : 1.0); }
          hnext = h * fac;
        }
        else hnext = 2.0 * h;
        double new_hprev = h, new_hpp = hprev;
        if (landed) { ++bp; nh_new = 1; double tstop = o->t1; if (bp < o->n_break && o->breaks[bp] < tstop) tstop = o->breaks[bp]; hnext = 0.1 * std::min(h, tstop - tn); }
        hnext = std::min(hnext, hmax);
        hp3 = hpp;
        t = tn; hprev = new_hprev; hpp = new_hpp; nhist = nh_new; S.accepted += 1; since += 1;
        if (trace_t && ntrace < trace_cap) trace_t[ntrace++] = t;
        if (tn >= o->t1) { status = 1; break; }
        if (hnext < hmin) hnext = hmin;
        prepare_attempt(t, hnext, nhist, hprev, hpp);
      } else {
        double fac;
        if (o->step_rule == 0) { fac = 0.9 * step_root(errn, ord); fac = std::min(0.9, std::max(0.1, fac)); }
        else { fac = 0.9 / (1.0 / step_root(2.0 * errn, ord) + 1e-4); fac = std::min(0.9, std::max(0.25, fac)); }
        double hn = h * fac; S.rejected += 1;
        if (hn < hmin) { status = -1; break; }
        prepare_attempt(t, hn, nhist, hprev, hpp);
      }
    } else {
      if (mode && (bad || diverge || k + 1 >= maxn) && !jcur) {
        need = true;                                   // failed on a stale Jacobian: the same step again, refactored first
        prepare_attempt(t, h, nhist, hprev, hpp);
      } else if (bad || diverge || k + 1 >= maxn) {
        double hn = 0.25 * h; S.newton_failures += 1;
        if (hn < hmin) { status = -2; break; }
        prepare_attempt(t, hn, nhist, hprev, hpp);
      } else {
        if (o->use_pcnr && L > 0) for (int i = n - L; i < n; ++i) u[i] = P.limit_w[i];
        for (int i = 0; i < n; ++i) du[i] = a0 * u[i] + beta[i];
        k += 1;
      }
    }
  }
  P.mode = savedmode;
  memcpy(u_io, u0.data(), n * sizeof(double));
  S.status = status; S.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  if (st) *st = S; if (trace_n) *trace_n = ntrace;
  return status == 1 ? 0 : 6;
}

}  // extern "C"
