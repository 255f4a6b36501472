"""Verilog-A module -> HIP stamp function for gfx950 (the device half of SURVEY.md section 8f-3).

Counterpart of the reference's ``generate_mna_stamp_method_nterm`` (/root/reference/src/vasim.jl:2993-3985): the module
body becomes straight-line C++ over forward-mode duals ``Dual<N>`` (value + one partial per module node, devices.hpp);
``ddt()`` splits an expression into a resistive and a reactive (charge) part, tracked statically here instead of by a
second dual tag (contrib.jl:356-375).  Expressions that cannot depend on a voltage stay plain ``double``.  The branch
stamps themselves (Jacobian rows, equivalent currents, charge-state or constant-capacitance form, direct residuals)
are the hand-written ``va_emit_branch`` in devices.hpp.

    python -m cadnip_jl_amd.va.hipgen  model1.va model2.va ...  > csrc/va_generated.hpp      (csrc/build.sh does this)
"""
import os
import sys

from .frontend import VAError, parse_file, parse_module

K_OVER_Q = 1.380649e-23 / 1.602176634e-19
# $simparam names that are MNASpec fields (vasim.jl:1190-1218) and are the same for every instance of a handle
SPEC_CONST = {"tnom": 27.0, "gshunt": 0.0, "srcFact": 1.0, "abstol": 1e-12, "reltol": 1e-3, "vntol": 1e-6, "iabstol": 1e-12}


def _lit(x):
    r = repr(float(x))
    return r if ("." in r or "e" in r or "inf" in r or "nan" in r) else r + ".0"


class _Gen:
    def __init__(self, m, func=None, tl=False):
        self.m = m
        self.tl = tl           # tangent-lane variant: T = Dual<1>, the lane's direction is `dir` (va_runtime.hpp)
        self.phase = "all"     # "setup": bias-independent statements only; "eval": without the assignments to hoisted variables
        self.lines = []
        self.pre = []          # definitions of $limit sites: emitted in front of the statement that uses them
        self.func = func       # name of the analog function being generated: every variable has the template type X
        self.tname = "X" if func else "T"

    def T(self, code, is_t):
        return code if is_t else "%s(%s)" % (self.tname, code)

    def fname(self, f):
        return "vaf_%s_%s" % (self.m.name, f)

    # expression -> (resistive code, it is a dual, reactive code or None)
    def g(self, e):
        m, k = self.m, e[0]
        if k == "num":
            return _lit(e[1]), False, None
        if k == "str":
            raise VAError("%s: a string value has no meaning on the device" % m.name)
        if k == "given":
            pn = m.aliasparams.get(e[1], e[1])
            return "(g_%s != 0.0 ? 1.0 : 0.0)" % pn, False, None
        if k == "analysis":
            # vasim.jl:1220-1250: "dc" / "static" <-> :dcop (mode 0), "tran" / "transient" <-> :tran (mode 1); :tranop is neither
            tests = []
            if any(a in ("dc", "static") for a in e[1]):
                tests.append("sys.mode == 0")
            if any(a in ("tran", "transient") for a in e[1]):
                tests.append("sys.mode == 1")
            return "((%s) ? 1.0 : 0.0)" % (" || ".join(tests) if tests else "false"), False, None
        if k == "noise":
            return "0.0", False, None
        if k == "Iprobe":
            j = m.probe_short(e)
            if j is None:
                return "0.0", False, None          # a branch that carries noise only: zero on the DC / transient path
            return "va_probe_current(u, nd[N + B + NL + %d], ((smask >> %d) & 1) != 0)" % (j, j), False, None
        if k == "ddx":
            c, t, _ = self.g(e[1])
            a = m.node_index(e[2])
            one = lambda k: ("va_ddx%s(%s, %d)" % (("_tl<%d>" % tl_lanes(m)) if self.tl else "", c, k)) if (t and k >= 0) else "0.0"
            if e[3] is not None:           # with respect to a branch potential V(a,b): (d/dV_a - d/dV_b) / 2 (vasim.jl:1168-1180)
                return "((%s - %s) / 2.0)" % (one(a), one(m.node_index(e[3]))), False, None
            return one(a), False, None
        if k == "var":
            if self.func:
                return "f_" + e[1], True, None
            if e[1] in m.params:
                return "p_" + e[1], False, None
            return "v_" + e[1], m.var_is_dual[e[1]], ("v_%s_q" % e[1]) if m.var_is_reactive[e[1]] else None
        if k == "ucall":
            return self.call(e[1], e[2])
        if k == "limit":
            # vasim.jl:1258-1330: w = limiter(vnew, vold, args...) on values; the site's dual is anchored at w, keeps the
            # probe's node partials and owns partial N + j
            j, lb = e[5][0], m.limit_sites[e[5][0]]
            probe = self.g(("V", e[1], e[2]))[0]
            args = ", ".join(["va_val(%s)" % probe, "vold%d" % lb] + ["va_val(%s)" % self.g(a)[0] for a in e[4]])
            self.pre.append("const double lim_w%d = %s<double>(%s, sys);" % (j, self.fname(e[3]), args))
            self.pre.append("if (lw) lw[nd[N + B + %d]] = lim_w%d;" % (lb, j))
            self.pre.append("ld[%d] = va_val(%s) - lim_w%d;" % (j, probe, j))
            if self.tl:
                self.pre.append("const T site%d = va_site_tl(%s, lim_w%d, dir == N + %d);" % (j, probe, j, j))
            else:
                self.pre.append("const T site%d = va_site(%s, lim_w%d, N + %d);" % (j, probe, j, j))
            return "site%d" % j, True, None
        if k == "V":
            a, b = m.node_index(e[1]), m.node_index(e[2])
            if a >= 0 and b >= 0:
                return "(V%d - V%d)" % (a, b), True, None
            if a >= 0:
                return "V%d" % a, True, None
            return "(-V%d)" % b, True, None
        if k == "ddt":
            c, t, _ = self.g(e[1])
            return "0.0", False, self.T(c, t)
        if k == "un":
            c, t, q = self.g(e[2])
            if e[1] == "!":
                return "(va_true(%s) ? 0.0 : 1.0)" % c, False, None
            if e[1] == "~":
                return "(double)(~(long long)va_val(%s))" % c, False, None
            return "(-%s)" % c, t, ("(-%s)" % q) if q else None
        if k == "cond":
            cc = self.g(e[1])[0]
            a, ta, qa = self.g(e[2])
            b, tb, qb = self.g(e[3])
            t = ta or tb
            r = "(va_true(%s) ? %s : %s)" % (cc, self.T(a, ta) if t else a, self.T(b, tb) if t else b)
            q = None
            if qa or qb:
                q = "(va_true(%s) ? %s : %s)" % (cc, qa or "T(0.0)", qb or "T(0.0)")
            return r, t, q
        if k == "call":
            args = [self.g(a) for a in e[2]]
            return "va_%s(%s)" % (e[1], ", ".join(a[0] for a in args)), any(a[1] for a in args) and e[1] not in ("floor", "ceil", "int"), None
        if k == "sys":
            if e[1] == "$table_model":          # decided by the parameters: evaluated on the host, one more entry of the parameter block
                NPm = len(m.params)
                return "par_of(d, %d)" % (NPm + 3 + (NPm if m.uses_given else 0) + len(m.string_tests) + m.table_calls.index(e)), False, None
            if e[1] == "$temperature":
                return "sys.temp", False, None
            if e[1] == "$mfactor":
                return "sys.mf", False, None
            if e[1] == "$vt":
                if e[2]:
                    c, t, _ = self.g(e[2][0])
                    return "(%s * %s)" % (_lit(K_OVER_Q), c), t, None
                return "(%s * sys.temp)" % _lit(K_OVER_Q), False, None
            if e[1] in ("$abstime", "$realtime"):
                return "sys.time", False, None
            name = e[2][0][1] if e[2] and e[2][0][0] == "str" else None
            if name == "gmin":
                return "sys.gmin", False, None
            if name in ("initjct", "iniLim"):
                return "sys.initjct", False, None
            if name in SPEC_CONST:
                return _lit(SPEC_CONST[name]), False, None
            if len(e[2]) > 1:
                return self.g(e[2][1])
            raise VAError("$simparam(%r) has no value on the device" % name)
        op = e[1]
        if op in ("==", "!="):                       # a string parameter against a literal: decided on the host, one flag per test
            for x, y in ((e[2], e[3]), (e[3], e[2])):
                if x[0] == "var" and m.param_kind.get(x[1]) == "string" and y[0] == "str":
                    i = m.string_tests.index((x[1], y[1]))
                    return "(st_%d %s 0.0 ? 1.0 : 0.0)" % (i, "!=" if op == "==" else "=="), False, None
        l, tl, ql = self.g(e[2])
        r, tr, qr = self.g(e[3])
        if op in ("+", "-"):
            q = None
            if ql and qr:
                q = "(%s %s %s)" % (ql, op, qr)
            elif ql:
                q = ql
            elif qr:
                q = qr if op == "+" else "(-%s)" % qr
            return "(%s %s %s)" % (l, op, r), tl or tr, q
        if op == "*":
            q = ("(%s * %s)" % (ql, r)) if ql else ("(%s * %s)" % (l, qr)) if qr else None
            return "(%s * %s)" % (l, r), tl or tr, q
        if op == "/":
            return "(%s / %s)" % (l, r), tl or tr, ("(%s / %s)" % (ql, r)) if ql else None
        if op in ("&&", "||"):
            return "((va_true(%s) %s va_true(%s)) ? 1.0 : 0.0)" % (l, op, r), False, None
        if op == "%":
            return "fmod(va_val(%s), va_val(%s))" % (l, r), False, None
        if op in ("&", "|", "^", "<<", ">>"):
            return "(double)((long long)va_val(%s) %s (long long)va_val(%s))" % (l, op, r), False, None
        return "((va_val(%s) %s va_val(%s)) ? 1.0 : 0.0)" % (l, op, r), False, None

    def call(self, fname, arg_exprs):
        """an analog-function call; output / inout arguments are passed by reference (they are variables of the call's
        numeric type: _analyse types a variable as a dual as soon as one call writes a dual into it)"""
        m = self.m
        dirs = m.func_dirs.get(fname) or ["in"] * len(arg_exprs)
        args = [self.g(a) for a in arg_exprs]
        dual = self.func is not None or any(a[1] for a in args)
        ty = ("X" if self.func else "T") if dual else "double"
        parts = []
        for a, d, ex in zip(args, dirs, arg_exprs):
            if d == "in":
                parts.append(self.T(a[0], a[1]) if dual else a[0])
            else:
                if ex[0] != "var":
                    raise VAError("%s: %s: an output argument must be a variable" % (m.name, fname))
                if not self.func and bool(m.var_is_dual.get(ex[1], False)) != dual:
                    raise VAError("%s: %s: output argument %s is %s, the call is %s" % (m.name, fname, ex[1],
                                  "a dual" if m.var_is_dual.get(ex[1]) else "a plain number", "on duals" if dual else "on plain numbers"))
                if self.phase == "eval" and not self.func and ex[1] in m.hoist_vars:
                    # the variable belongs to the setup pass (read here through the per-device cache, not assignable): the call writes
                    # the same bias-independent value the setup pass already left there, into a scratch copy
                    self.n_scratch = getattr(self, "n_scratch", 0) + 1
                    nm = "t_out%d" % self.n_scratch
                    self.pre.append("double %s = %s;" % (nm, a[0] if ex[1] in m.cache_vars else "0.0"))
                    parts.append(nm)
                else:
                    parts.append(a[0])
        return "%s<%s>(%s)" % (self.fname(fname), ty, ", ".join(parts + ["sys"])), dual, None

    def emit_vcontrib(self, j, value):
        """the stamps of the two-node potential contribution `vshorts[j]` with the (dual) value ``value``"""
        m = self.m
        a, b = m.shorts[m.vshorts[j]][0], m.shorts[m.vshorts[j]][1]
        on = "((smask >> %d) & 1) != 0" % j
        if self.tl:
            return "va_emit_vcontrib_tl<N, %d>(s, %d, %d, %s, %s, dir < N ? -Vf[dir < N ? dir : 0] : 0.0, dir)" % (tl_lanes(m), m.g_short(j), 3 * len(m.branches) + j, on, value)
        return "va_emit_vcontrib<N>(s, Vf, %d, %d, %s, %s)" % (m.g_short(j), 3 * len(m.branches) + j, on, value)

    def stmts(self, body, ind):
        m, pad = self.m, "  " * ind
        for s in body:
            if self.phase == "setup":
                if s[0] in ("contrib", "short", "fatal"):
                    continue
                if s[0] == "assign" and not m.var_is_static.get(s[1], False):
                    continue
                if s[0] == "callstmt" and not all(m.expr_is_static(a) for a in s[2]):
                    continue
                if s[0] in ("if", "while") and not m.expr_is_static(s[1]):
                    continue
                if s[0] == "case" and not m.expr_is_static(s[1]):
                    continue
                if s[0] == "for" and not m.expr_is_static(s[2]):
                    continue
            elif self.phase == "eval" and s[0] == "assign" and s[1] in m.hoist_vars:
                continue
            if s[0] == "callstmt":
                c = self.call(s[1], s[2])[0]
                self.flush(pad)
                self.lines.append("%s(void)%s;" % (pad, c))
            elif s[0] == "fatal":
                self.lines.append("%s// %s: not raised on the device (the host evaluation of the instance's parameters raises it)" % (pad, s[1]))
            elif s[0] == "case":
                sel = self.g(s[1])[0]
                self.flush(pad)
                self.lines.append("%s{ const double case_sel = va_val(%s);" % (pad, sel))
                first = True
                default = None
                for vals, body_ in s[2]:
                    if vals is None:
                        default = body_
                        continue
                    cond = " || ".join("case_sel == va_val(%s)" % self.g(v)[0] for v in vals)
                    self.lines.append("%s%sif (%s) {" % (pad, "" if first else "} else ", cond))
                    first = False
                    self.stmts([body_], ind + 1)
                if default is not None:
                    self.lines.append("%s%s{" % (pad, "" if first else "} else "))
                    self.stmts([default], ind + 1)
                    first = False
                if not first:
                    self.lines.append("%s}" % pad)
                self.lines.append("%s}" % pad)
            elif s[0] == "while":
                self.lines.append("%swhile (va_true(%s)) {" % (pad, self.g(s[1])[0]))
                self.stmts([s[2]], ind + 1)
                self.lines.append("%s}" % pad)
            elif s[0] == "for":
                self.stmts([s[1]], ind)
                self.lines.append("%swhile (va_true(%s)) {" % (pad, self.g(s[2])[0]))
                self.stmts([s[4], s[3]], ind + 1)
                self.lines.append("%s}" % pad)
            elif s[0] == "assign":
                c, t, q = self.g(s[2])
                self.flush(pad)
                if self.func:
                    self.lines.append("%sf_%s = %s;" % (pad, s[1], c))
                    continue
                self.lines.append("%sv_%s = %s;" % (pad, s[1], c))
                if m.var_is_reactive[s[1]]:
                    self.lines.append("%sv_%s_q = %s;" % (pad, s[1], q or "T(0.0)"))
            elif s[0] == "contrib":
                if s[3][0] == "noise":
                    continue                                  # noise sources: no current on the DC / transient path
                b = m.branches.index((m.node_index(s[1]), m.node_index(s[2])))
                c, t, q = self.g(s[3])
                self.flush(pad)
                self.lines.append("%sbr%d_r = br%d_r + %s;" % (pad, b, b, c))
                if q:
                    self.lines.append("%sbr%d_q = br%d_q + %s;" % (pad, b, b, q))
            elif s[0] == "short":
                j = next((jj for jj, si in enumerate(m.vshorts) if m.shorts[si][3] is s), -1)
                if j < 0:
                    self.lines.append("%s// V(%s,%s) <+ 0: the two nets are one unknown for this instance (collapsed at structure discovery)" % (pad, s[1], s[2]))
                elif m.short_kind[m.vshorts[j]] == "cond" and s[3] == ("num", 0.0):
                    a, b = m.shorts[m.vshorts[j]][0], m.shorts[m.vshorts[j]][1]
                    self.lines.append("%sva_emit_short<N>(u, s, Vf, nd, N + B + NL + %d, %d, %d, %d, %d, ((smask >> %d) & 1) != 0%s);   // V(%s,%s) <+ 0 with a branch current"
                                      % (pad, j, a, b, m.g_short(j), 3 * len(m.branches) + j, j, " && dir == 0" if self.tl else "", s[1], s[2]))
                else:
                    c, t, q = self.g(s[3])
                    self.flush(pad)
                    if m.short_kind[m.vshorts[j]] == "cond":       # stamped where it stands (vasim.jl:2340-2393)
                        self.lines.append("%s%s;   // V(%s,%s) <+ ... with a branch current" % (pad, self.emit_vcontrib(j, self.T(c, t)), s[1], s[2]))
                    else:                                          # top level of the analog block: stamped behind the branches
                        self.lines.append("%ssv%d = %s;" % (pad, j, self.T(c, t)))
                        if m.short_reactive[j]:
                            self.lines.append("%ssv%d_q = %s;" % (pad, j, q if q else "0.0"))
            elif s[0] == "block":
                self.stmts(s[1], ind)
            elif s[0] == "if":
                self.lines.append("%sif (va_true(%s)) {" % (pad, self.g(s[1])[0]))
                self.stmts([s[2]], ind + 1)
                if s[3] != ("block", []):
                    self.lines.append("%s} else {" % pad)
                    self.stmts([s[3]], ind + 1)
                self.lines.append("%s}" % pad)


def _flush(self, pad):
    for l in self.pre:
        self.lines.append(pad + l)
    self.pre = []


_Gen.flush = _flush


def generate_analog_functions(m):
    """``template <class X> X vaf_<module>_<name>(X inputs..., X& outputs..., const VaSys& sys)`` per analog function:
    instantiated with double ($limit limiters, voltage-independent calls) and with the module's dual type.  Functions may
    call each other in any order: all are declared before the first is defined."""
    out, defs = [], []
    for fname, (f_args, f_loc, f_body) in m.functions.items():
        dirs = m.func_dirs.get(fname) or ["in"] * len(f_args)
        sig = "__device__ inline X vaf_%s_%s(%s)" % (m.name, fname, ", ".join(["X%s f_%s" % ("" if d == "in" else "&", a) for a, d in zip(f_args, dirs)] + ["const VaSys& sys"]))
        out.append("template <class X> %s;" % sig)
        g = _Gen(m, func=fname)
        defs.append("template <class X>")
        defs.append(sig + " {")
        for v in f_loc + [fname]:
            defs.append("  X f_%s = 0.0;" % v)
        g.stmts(f_body, 1)
        defs.extend(g.lines)
        defs.append("  return f_%s;" % fname)
        defs.append("}")
    return "\n".join(out + defs)


def tl_lanes(m):
    """lanes per device of the tangent-lane variant: one DPP row when the directions (nodes + $limit sites) fit it, else two"""
    return 16 if m.n_nodes + m.n_sites <= 16 else 32


def generate_function(m, tl=False, with_functions=True):
    """tl: the tangent-lane variant ``stamp_va_<module>_tl(d, u, s, lw, dir)`` -- one derivative direction per lane, 16 lanes per
    device (va_runtime.hpp); the statements are the same text on Dual<1>."""
    N, B, NP, S, NL = m.n_nodes, len(m.branches), len(m.params), m.n_sites, len(m.limit_branches)
    if tl and N + S > 32:
        raise VAError("%s: %d derivative directions do not fit the 32 lanes of a device group" % (m.name, N + S))
    lanes = tl_lanes(m)
    g = _Gen(m, tl=tl)
    hoist = tl and bool(m.cache_vars)        # the external models run a setup pass once per parameter set (generate_setup)
    if hoist:
        g.phase = "eval"
    L = g.lines
    if m.functions and with_functions:
        L.append(generate_analog_functions(m))
    L.append("// module %s: nodes (%s), %d parameter(s), branches %s" % (
        m.name, ", ".join(m.nodes), NP, ", ".join("(%s,%s)%s" % (m.nodes[p] if p >= 0 else "gnd", m.nodes[n] if n >= 0 else "gnd",
                                                               " reactive" if r else "") for (p, n), r in zip(m.branches, m.reactive))))
    L.append("template <class Ctx, class Out>")
    if tl:
        L.append("__device__ inline void stamp_va_%s_tl(const Ctx& d, const double* u, const Out& s, double* lw, const int dir) {" % m.name)
        # the function is too large to be inlined: through `d` the compiler sees generic pointers and every parameter / cache reference
        # would be a flat load (PSP103: 1 700 of them); typed pointers make them global loads
        L.append("  typedef const __attribute__((address_space(1))) double* VaGlobalPtr;   // the parameter rows and the setup cache lie in global memory: say so")
        L.append("  const VaGlobalPtr va_gpar = (VaGlobalPtr)d.par + d.dev, va_gcache = (VaGlobalPtr)d.cache + d.dev;")
        L.append("  const int va_count = d.count;")
    else:
        L.append("__device__ inline void stamp_va_%s(const Ctx& d, const double* u, const Out& s, double* lw) {" % m.name)
    L.append("  constexpr int N = %d, B = %d, S = %d, NL = %d;   // nodes, branches, $limit sites, limit unknowns" % (N, B, S, NL))
    L.append("  typedef Dual<%s> T;" % ("1" if tl else "N + S"))
    L.append("  constexpr int NV = %d;   // V(a,b) <+ 0 statements that own a branch current" % len(m.vshorts))
    L.append("  int nd[N + B + NL + NV];")
    L.append("#pragma unroll")
    L.append("  for (int k = 0; k < N + B + NL + NV; ++k) nd[k] = node_of(d, k);")
    if m.vshorts:
        L.append("  const int smask = d.ipar[2 * d.count + d.dev];   // bit j: short j executes for this instance")
    L.append("  double Vf[N];")
    L.append("#pragma unroll")
    L.append("  for (int k = 0; k < N; ++k) Vf[k] = volt(u, nd[k]);")
    for k in range(N):
        if tl:
            L.append("  const T V%d = va_seed_tl(Vf[%d], dir == %d);   // V(%s)" % (k, k, k, m.nodes[k]))
        else:
            L.append("  const T V%d = T::seed(Vf[%d], %d);   // V(%s)" % (k, k, k, m.nodes[k]))
    # parameters: small modules read them once, up front.  A large external model (782 parameters for PSP103) would keep hundreds of
    # them alive from the top of the function to their uses -- the first build spilled 638 doubles to scratch memory and reloaded each
    # at its use, one exposed memory latency apiece (80 % of the wave's time was s_waitcnt) -- so there every reference loads its
    # parameter where it stands (macros, undefined again behind the function).
    macros = []
    for i, p in enumerate(m.params):
        if m.param_kind.get(p) == "string":
            continue
        if tl:
            macros.append(("p_%s" % p, "va_gpar[%d * va_count]" % i))
        else:
            L.append("  const double p_%s = par_of(d, %d);" % (p, i))
    L.append("  const VaSys sys{par_of(d, %d), par_of(d, %d), par_of(d, %d), d.initjct ? 1.0 : 0.0, d.mode, d.t};   // $temperature, $mfactor, gmin, initjct, analysis(), $abstime" % (NP, NP + 1, NP + 2))
    if m.uses_given:
        for i, p in enumerate(m.params):
            if tl:
                macros.append(("g_%s" % p, "va_gpar[%d * va_count]" % (NP + 3 + i)))
            else:
                L.append("  const double g_%s = par_of(d, %d);   // $param_given(%s)" % (p, NP + 3 + i, p))
    for i, (p, lit) in enumerate(m.string_tests):
        if tl:
            macros.append(("st_%d" % i, "va_gpar[%d * va_count]" % (NP + 3 + (NP if m.uses_given else 0) + i)))
        else:
            L.append("  const double st_%d = par_of(d, %d);   // %s == \"%s\"" % (i, NP + 3 + (NP if m.uses_given else 0) + i, p, lit))
    L.append("  double ld[S > 0 ? S : 1] = {0.0};    // per $limit site: V(probe) - w, the lim_rhs delta (vasim.jl:2957-2966)")
    for lb, (p, n) in enumerate(m.limit_branches):
        L.append("  const double vold%d = u[nd[N + B + %d]];   // limit unknown of probe branch (%s,%s)" % (
            lb, lb, m.nodes[p] if p >= 0 else "gnd", m.nodes[n] if n >= 0 else "gnd"))
        L.append("  %sva_emit_limit_rows<N, B>(s, %d);" % ("if (dir == 0) " if tl else "", lb))
    for v in m.locals_:
        if hoist and v in m.hoist_vars:
            if v in m.cache_vars:
                macros.append(("v_%s" % v, "va_gcache[%d * va_count]" % m.cache_vars.index(v)))
            continue                                    # computed by the setup pass; read from the per-device cache where used
        L.append("  %s v_%s = 0.0;" % ("T" if m.var_is_dual[v] else "double", v))
        if m.var_is_reactive[v]:
            L.append("  T v_%s_q = 0.0;" % v)
    for b in range(B):
        L.append("  T br%d_r = 0.0, br%d_q = 0.0;" % (b, b))
    tops = [(j, si) for kind in ("named", "top") for j, si in enumerate(m.vshorts) if m.short_kind[si] == kind]
    for j, si in tops:                                                 # potential contributions at the top level: value now, stamps behind the branches
        L.append("  T sv%d = 0.0%s;" % (j, (", sv%d_q = 0.0" % j) if m.short_reactive[j] else ""))
    g.stmts([("assign", nm, ie) for nm, ie in m.local_init], 1)       # module-scope initialisers, in declaration order
    g.stmts(m.body, 1)
    if tl:
        L.append("  // this lane's weight in the equivalent currents: -V_dir for a node direction, V(probe) - w for a $limit site, 0 when idle")
        L.append("  double wgt = 0.0;")
        for k in range(N):
            L.append("  if (dir == %d) wgt = -Vf[%d];" % (k, k))
        for j in range(S):
            L.append("  if (dir == N + %d) wgt = ld[%d];" % (j, j))
        for b, ((p, n), r) in enumerate(zip(m.branches, m.reactive)):
            L.append("  va_emit_branch_tl<N, S, B, %s, %d>(s, %d, sys.mf * br%d_r, sys.mf * br%d_q, wgt, dir);" % ("true" if r else "false", lanes, b, b, b))
        for j, si in tops:
            if m.short_kind[si] == "named":
                L.append("  va_emit_vnamed(s, %d, %d, %d, ((smask >> %d) & 1) != 0 && dir == 0, va_val(sv%d), %s);   // V(%s) <+ ..." % (
                    m.g_short(j), 3 * B + j, m.c_short(j) if m.short_reactive[j] else -1, j, j, ("va_val(sv%d_q)" % j) if m.short_reactive[j] else "0.0", m.shorts[si][3][4]))
            else:
                L.append("  %s;" % g.emit_vcontrib(j, "sv%d" % j))
        L.append("}")
        k = next(i for i, l in enumerate(L) if l.startswith("template <class Ctx, class Out>"))
        return "\n".join(L[:k] + ["#define %s %s" % mc for mc in macros] + L[k:] + ["#undef %s" % mc[0] for mc in macros])
    L.append("  const int vdep = d.ipar[1 * d.count + d.dev];   // bit b: branch b uses a charge unknown")
    for b, ((p, n), r) in enumerate(zip(m.branches, m.reactive)):
        L.append("  va_emit_branch<N, S, B, %s>(d, u, s, Vf, ld, nd, %d, %d, %d, sys.mf * br%d_r, sys.mf * br%d_q, ((vdep >> %d) & 1) != 0);"
                 % ("true" if r else "false", b, p, n, b, b, b))
    for j, si in tops:
        if m.short_kind[si] == "named":
            L.append("  va_emit_vnamed(s, %d, %d, %d, ((smask >> %d) & 1) != 0, va_val(sv%d), %s);   // V(%s) <+ ..." % (
                m.g_short(j), 3 * B + j, m.c_short(j) if m.short_reactive[j] else -1, j, j, ("va_val(sv%d_q)" % j) if m.short_reactive[j] else "0.0", m.shorts[si][3][4]))
        else:
            L.append("  %s;" % g.emit_vcontrib(j, "sv%d" % j))
    for lb, (p, n) in enumerate(m.limit_branches):
        L.append("  if constexpr (Out::DIRECT) s.Rn(nd[N + B + %d], vold%d - (%s));   // limit row: u_l - V(probe)" % (
            lb, lb, " - ".join(["Vf[%d]" % p if p >= 0 else "0.0", "Vf[%d]" % n if n >= 0 else "0.0"])))
    L.append("}")
    return "\n".join(L)


def generate_setup(m):
    """``setup_va_<module>(d, cache, stride)``: every bias-independent statement of the module, once per (instance, device) and parameter
    set; writes the hoisted variables the per-call code reads (VAModule.cache_vars) to ``cache[k * stride]``."""
    NP = len(m.params)
    g = _Gen(m)
    g.phase = "setup"
    L = g.lines
    macros = []
    for i, p in enumerate(m.params):
        if m.param_kind.get(p) != "string":
            macros.append(("p_%s" % p, "par_of(d, %d)" % i))
    if m.uses_given:
        for i, p in enumerate(m.params):
            macros.append(("g_%s" % p, "par_of(d, %d)" % (NP + 3 + i)))
    for i, (p, lit) in enumerate(m.string_tests):
        macros.append(("st_%d" % i, "par_of(d, %d)" % (NP + 3 + (NP if m.uses_given else 0) + i)))
    L.append("template <class Ctx>")
    L.append("__device__ inline void setup_va_%s(const Ctx& d, double* cache, const int stride) {" % m.name)
    L.append("  typedef double T;")
    L.append("  const VaSys sys{par_of(d, %d), par_of(d, %d), par_of(d, %d), 0.0, d.mode, 0.0};" % (NP, NP + 1, NP + 2))
    for v in m.locals_:
        if m.var_is_static.get(v, False):
            L.append("  double v_%s = 0.0;" % v)
    g.stmts([("assign", nm, ie) for nm, ie in m.local_init if m.var_is_static.get(nm, False)], 1)
    g.stmts(m.body, 1)
    for k, v in enumerate(m.cache_vars):
        L.append("  cache[%d * stride] = v_%s;" % (k, v))
    L.append("}")
    return "\n".join(["#define %s %s" % mc for mc in macros] + L + ["#undef %s" % mc[0] for mc in macros])


def generate_header(modules):
    """The whole va_generated.hpp: one function per built-in module, the dispatcher and the host-side shape table.  The modules
    whose sources are not part of this repository (va.EXTERNAL) live in va_generated_ext.hpp (generate_ext_header), which this
    header includes: their model ids follow the built-in ones."""
    out = ["// GENERATED by cadnip.jl_amd/va/hipgen.py from the Verilog-A sources listed below -- do not edit.",
           "// Included by devices.hpp (device code) and api.hip (shape table).", "#pragma once", ""]
    out.append("#define CADNIP_VA_NBUILTIN %d" % len(modules))
    out.append('#include "va_generated_ext.hpp"   // CADNIP_VA_NEXT, CADNIP_VA_EXT_LIST, CADNIP_VA_EXT_SHAPES: the external models (device code: va_ext/<module>.hip)')
    out.append("#ifdef CADNIP_VA_DEVICE_CODE   // set by va_runtime.hpp (device translation units); api.hip takes the shape table only")
    out.append("namespace cadnip {")
    for m in modules:
        out.append(generate_function(m))
        out.append("")
    out.append("// model id = position in the list the generator was given (ipar row 0 of a CADNIP_DEV_VA block)")
    out.append("template <class Ctx, class Out>")
    out.append("__device__ inline void stamp_va(const Ctx& d, const double* u, const Out& s, double* lw) {")
    out.append("  switch (d.ipar[0 * d.count + d.dev]) {")
    for i, m in enumerate(modules):
        out.append("    case %d: stamp_va_%s(d, u, s, lw); break;" % (i, m.name))
    out.append("    default: break;")
    out.append("  }")
    out.append("}")
    out.append("}  // namespace cadnip")
    out.append("#endif")
    out.append("")
    out.append("// n_nodes (local unknowns), n_g, n_c, n_b, n_par, n_ipar per generated model: what cadnip_create checks a block against")
    out.append("#define CADNIP_VA_NMODELS (CADNIP_VA_NBUILTIN + CADNIP_VA_NEXT)")
    out.append("static const struct { const char* name; int n_nodes, n_g, n_c, n_b, n_par, n_ipar; } CADNIP_VA_SHAPES[CADNIP_VA_NMODELS > 0 ? CADNIP_VA_NMODELS : 1] = {")
    for m in modules:
        out.append('  {"%s", %d, %d, %d, %d, %d, %d},' % ((m.name,) + m.shape()))
    out.append("  CADNIP_VA_EXT_SHAPES")
    out.append("};")
    return "\n".join(out) + "\n"


def generate_ext_header(modules):
    """va_generated_ext.hpp: the table of the external models (the reference's own model files: their sources are third-party text
    inside the reference and are not copied -- what the repository keeps is generated: this table and one translation unit per model,
    generate_ext_unit, regenerated by csrc/build.sh whenever the sources are present).  Model ids follow the built-in ones."""
    out = ["// GENERATED by cadnip.jl_amd/va/hipgen.py from %s -- do not edit." % (", ".join(m.name for m in modules) or "(no external model)"),
           "// Included by va_generated.hpp.", "#pragma once", ""]
    out.append("#define CADNIP_VA_NEXT %d" % len(modules))
    out.append("#define CADNIP_VA_EXT_LIST(X) " + " ".join("X(%d, %s)" % (i, m.name) for i, m in enumerate(modules)))
    out.append("#define CADNIP_VA_EXT_SHAPES " + " ".join('{"%s", %d, %d, %d, %d, %d, %d},' % ((m.name,) + m.shape()) for m in modules))
    out.append("// evaluated with one derivative direction per lane, 16 or 32 lanes per device (va_runtime.hpp: tangent lanes)")
    out.append("#define CADNIP_VA_EXT_TL_LANES " + " ".join("%d," % tl_lanes(m) for m in modules))
    out.append("// doubles per device that the setup pass leaves for the per-call code (VAModule.cache_vars)")
    out.append("#define CADNIP_VA_EXT_NCACHE " + " ".join("%d," % len(m.cache_vars) for m in modules))
    return "\n".join(out) + "\n"


def generate_ext_unit(m):
    """va_ext/<module>.hip: the device code of one external model -- its analog functions, its setup pass, its tangent-lane stamp
    function -- with the model's own instantiation of the per-op stamping kernel and of the setup kernel (stamp_csr_kernel.hpp).  These
    functions are thousands of statements long: a kernel per model keeps each one's register budget and compile time its own; circuits
    that use them never run in the fused kernel."""
    out = ["// GENERATED by cadnip.jl_amd/va/hipgen.py from the Verilog-A source of %s -- do not edit." % m.name,
           '#include "../stamp_csr_kernel.hpp"', "", "namespace cadnip {"]
    if m.functions:
        out.append(generate_analog_functions(m))
    out.append(generate_setup(m))
    out.append("")
    out.append(generate_function(m, tl=True, with_functions=False))
    out.append("")
    out.append("struct VaExt_%s {" % m.name)
    out.append("  template <class Ctx, class Out> static __device__ __forceinline__ void stamp(const Ctx& d, const double* u, const Out& s, double* lw, const int dir) { stamp_va_%s_tl(d, u, s, lw, dir); }" % m.name)
    out.append("  template <class Ctx> static __device__ __forceinline__ void setup(const Ctx& d, double* cache, const int stride) { setup_va_%s(d, cache, stride); }" % m.name)
    out.append("};")
    out.append("int va_ext_stamp_launch_%s(const CsrStampArgs& a, unsigned grid, size_t shmem, hipStream_t stream) { return launch_stamp_kernel<CADNIP_DEV_VA, VaExt_%s>(a, grid, shmem, stream); }" % (m.name, m.name))
    out.append("int va_ext_setup_launch_%s(const VaSetupArgs& a, hipStream_t stream) { return launch_va_setup_kernel<VaExt_%s>(a, stream); }" % (m.name, m.name))
    out.append("}  // namespace cadnip")
    return "\n".join(out) + "\n"


def _write_if_changed(path, text):
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)


def main(argv):
    """hipgen.py a.va b.va ...                 -> va_generated.hpp on stdout
       hipgen.py --ext DIR x.va y.va ...       -> va_generated_ext.hpp on stdout, DIR/<module>.hip per model (rewritten only when changed;
                                                  units of models no longer listed are removed)"""
    if argv and argv[0] == "--ext":
        d = argv[1]
        mods = [parse_file(fn) for fn in argv[2:]]
        os.makedirs(d, exist_ok=True)
        for m in mods:
            _write_if_changed(os.path.join(d, m.name + ".hip"), generate_ext_unit(m))
        for fn in os.listdir(d):
            if fn.endswith(".hip") and fn[:-4] not in [m.name for m in mods]:
                os.remove(os.path.join(d, fn))
        sys.stdout.write(generate_ext_header(mods))
        return
    mods = [parse_file(fn) for fn in argv]
    sys.stdout.write(generate_header(mods))


if __name__ == "__main__":
    main(sys.argv[1:])
