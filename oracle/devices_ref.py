"""TEST INFRASTRUCTURE (oracle) -- builtin device stamps and source waves.

Literal restatement of /root/reference/src/mna/devices.jl (line numbers cited per
function).  ``ctx`` is either oracle.mna_ref.MNAContext or DirectStampContext.
"""
import bisect
import math

from .mna_ref import stamp_conductance, stamp_capacitance, x_at
from .dual import val


# -- waves --------------------------------------------------------------------
def find_t_in_ts(ts, t):  # devices.jl:30-36 (returns 1-based insertion index)
    idx = bisect.bisect_left(ts, t) + 1
    if idx <= len(ts) and ts[idx - 1] == t:
        return idx + 1
    return idx


def pwl_at_time(ts, ys, t):  # devices.jl:47-71
    i = find_t_in_ts(ts, t)
    if i <= 1:
        return ys[0]
    if i > len(ts):
        return ys[-1]
    if ys[i - 2] == ys[i - 1]:
        return ys[i - 1]
    if ts[i - 1] == ts[i - 2]:
        return (ys[i - 2] + ys[i - 1]) / 2
    slope = (ys[i - 1] - ys[i - 2]) / (ts[i - 1] - ts[i - 2])
    return ys[i - 2] + (t - ts[i - 2]) * slope


def pulse_at_time(v1, v2, td, tr, tf, pw, per, t):  # devices.jl:85-103
    if t < td:
        return v1
    phase = math.fmod(t - td, per) if per > 0 else (t - td)
    if per > 0 and phase < 0:
        phase += per
    if phase < tr:
        return v1 + (v2 - v1) * (phase / tr) if tr > 0 else v2
    elif phase < tr + pw:
        return v2
    elif phase < tr + pw + tf:
        return v2 + (v1 - v2) * ((phase - tr - pw) / tf) if tf > 0 else v1
    return v1


def sind(deg):
    """Julia sind: exact at multiples of 90 degrees."""
    r = math.fmod(deg, 360.0)
    if r == 0.0 or r == 180.0 or r == -180.0:
        return 0.0
    if r == 90.0 or r == -270.0:
        return 1.0
    if r == -90.0 or r == 270.0:
        return -1.0
    return math.sin(math.radians(r))


class PWLWave:  # devices.jl:130-145
    def __init__(self, ts, ys):
        self.ts = list(map(float, ts))
        self.ys = list(map(float, ys))

    def __call__(self, t):
        return pwl_at_time(self.ts, self.ys, t)

    def breakpoints(self):
        return ("list", list(self.ts))


class SinWave:  # devices.jl:155-180
    def __init__(self, vo, va, freq, td=0.0, theta=0.0, phase=0.0):
        self.vo, self.va, self.freq, self.td, self.theta, self.phase = map(float, (vo, va, freq, td, theta, phase))

    def __call__(self, t):
        if t < self.td:
            return self.vo + self.va * sind(self.phase)
        return self.vo + self.va * math.exp(-self.theta * (t - self.td)) * sind(360 * self.freq * (t - self.td) + self.phase)

    def breakpoints(self):
        return ("list", [self.td]) if self.td > 0 else None


class PulseWave:  # devices.jl:189-214
    def __init__(self, v1, v2, td, tr, tf, pw, per):
        self.v1, self.v2, self.td, self.tr, self.tf, self.pw, self.per = map(float, (v1, v2, td, tr, tf, pw, per))

    def __call__(self, t):
        return pulse_at_time(self.v1, self.v2, self.td, self.tr, self.tf, self.pw, self.per, t)

    def breakpoints(self):
        edges = [self.td, self.td + self.tr, self.td + self.tr + self.pw, self.td + self.tr + self.pw + self.tf]
        return ("periodic", edges, self.per) if self.per > 0 else ("list", edges)


def expand_breakpoints(specs, tspan, max_points=100000):
    """solve.jl:1847-1900: sorted, de-duplicated times strictly inside tspan."""
    t0, t1 = float(tspan[0]), float(tspan[1])
    out = []
    for s in specs:
        if s is None or not s[1]:
            continue
        if s[0] == "list":
            out.extend(t for t in s[1] if t0 < t < t1)
        else:
            _, times, period = s
            tmin, tmax = min(times), max(times)
            k_start = int(min(max(math.floor((t0 - tmax) / period), 0.0), 1e15))
            k_end = int(min(max(math.ceil((t1 - tmin) / period), -1.0), 1e15))
            if k_end < k_start:
                continue
            if k_end - k_start + 1 > max_points:
                k_end = k_start + max_points - 1
            for k in range(k_start, k_end + 1):
                base = k * period
                out.extend(t + base for t in times if t0 < t + base < t1)
    if not out:
        return out
    out.sort()
    del out[max_points:]
    eps = lambda v: math.ulp(v)
    dedup = [out[0]]
    for t in out[1:]:
        if t - dedup[-1] > 4 * max(eps(dedup[-1]), eps(t)):
            dedup.append(t)
    return dedup


def get_source_value(dc, tran, t, mode):  # devices.jl:352-360
    if tran is None:
        return dc
    if mode in ("dcop", "ac"):
        return dc
    return tran(t)


# -- linear devices ---------------------------------------------------------------
def stamp_resistor(ctx, p, n, r, name="R"):  # devices.jl:498-510
    stamp_conductance(ctx, p, n, 1.0 / r)
    if hasattr(ctx, "register_thermal_noise"):           # Johnson-Nyquist 4kT G (a no-op on the direct-stamp context)
        ctx.register_thermal_noise(p, n, 1.0 / r, name)


def stamp_capacitor(ctx, p, n, c):  # devices.jl:531-534
    stamp_capacitance(ctx, p, n, c)


def stamp_inductor(ctx, p, n, l, name="L"):  # devices.jl:569-586
    I = ctx.alloc_current("I_" + name)
    ctx.stamp_G(p, I, 1.0)
    ctx.stamp_G(n, I, -1.0)
    ctx.stamp_G(I, p, 1.0)
    ctx.stamp_G(I, n, -1.0)
    ctx.stamp_C(I, I, -l)
    return I


def stamp_vsource(ctx, p, n, dc, tran=None, t=0.0, mode="dcop", name="V"):  # devices.jl:619-663
    I = ctx.alloc_current("I_" + name)
    ctx.register_breakpoints(tran)
    ctx.stamp_G(p, I, 1.0)
    ctx.stamp_G(n, I, -1.0)
    ctx.stamp_G(I, p, 1.0)
    ctx.stamp_G(I, n, -1.0)
    ctx.stamp_b(I, get_source_value(dc, tran, t, mode))
    return I


def stamp_isource(ctx, p, n, dc, tran=None, t=0.0, mode="dcop", name="I"):  # devices.jl:698-737
    ctx.register_breakpoints(tran)
    i = get_source_value(dc, tran, t, mode)
    ctx.stamp_b(p, i)
    ctx.stamp_b(n, -i)


def stamp_behavioral_vsource(ctx, p, n, value_fn, name="B", get_voltage=None):  # devices.jl:1079-1102
    I = ctx.alloc_current("I_" + name)
    ctx.stamp_G(p, I, 1.0)
    ctx.stamp_G(n, I, -1.0)
    ctx.stamp_G(I, p, 1.0)
    ctx.stamp_G(I, n, -1.0)
    v = value_fn(get_voltage) if get_voltage is not None else 0.0
    ctx.stamp_b(I, v)
    return I


def stamp_behavioral_isource(ctx, p, n, value_fn, get_voltage=None):  # devices.jl:1118-1131
    i = value_fn(get_voltage) if get_voltage is not None else 0.0
    ctx.stamp_b(p, i)
    ctx.stamp_b(n, -i)


def stamp_vcvs(ctx, op, on, ip, in_, gain, name="E"):  # devices.jl:760-775
    I = ctx.alloc_current("I_" + name)
    ctx.stamp_G(op, I, 1.0)
    ctx.stamp_G(on, I, -1.0)
    ctx.stamp_G(I, op, 1.0)
    ctx.stamp_G(I, on, -1.0)
    ctx.stamp_G(I, ip, -gain)
    ctx.stamp_G(I, in_, gain)
    return I


def stamp_vccs(ctx, op, on, ip, in_, gm):  # devices.jl:797-808
    ctx.stamp_G(op, ip, -gm)
    ctx.stamp_G(op, in_, gm)
    ctx.stamp_G(on, ip, gm)
    ctx.stamp_G(on, in_, -gm)


def stamp_ccvs(ctx, op, on, ip, in_, rm, name="H"):  # devices.jl:824-849
    Iin = ctx.alloc_current("I_" + name + "_in")
    Iout = ctx.alloc_current("I_" + name + "_out")
    ctx.stamp_G(ip, Iin, 1.0)
    ctx.stamp_G(in_, Iin, -1.0)
    ctx.stamp_G(Iin, ip, 1.0)
    ctx.stamp_G(Iin, in_, -1.0)
    ctx.stamp_G(op, Iout, 1.0)
    ctx.stamp_G(on, Iout, -1.0)
    ctx.stamp_G(Iout, op, 1.0)
    ctx.stamp_G(Iout, on, -1.0)
    ctx.stamp_G(Iout, Iin, -rm)
    return Iout, Iin


def stamp_cccs(ctx, op, on, ip, in_, gain, name="F"):  # devices.jl:865-881
    Iin = ctx.alloc_current("I_" + name + "_in")
    ctx.stamp_G(ip, Iin, 1.0)
    ctx.stamp_G(in_, Iin, -1.0)
    ctx.stamp_G(Iin, ip, 1.0)
    ctx.stamp_G(Iin, in_, -1.0)
    ctx.stamp_G(op, Iin, -gain)
    ctx.stamp_G(on, Iin, gain)
    return Iin


# -- PCNR limiting ------------------------------------------------------------------
def pnjlim(vnew, vold, vt, vcrit):  # devices.jl:1169-1189
    if vnew > vcrit and abs(vnew - vold) > vt + vt:
        if vold > 0.0:
            arg = (vnew - vold) / vt
            if arg > 0.0:
                return vold + vt * (2.0 + math.log(arg - 2.0)), True
            return vold - vt * (2.0 + math.log(2.0 - arg)), True
        return vt * math.log(vnew / vt), True
    elif vnew < 0.0:
        arg = -vold - 1.0 if vold > 0.0 else 2.0 * vold - 1.0
        if vnew < arg:
            return arg, True
    return vnew, False


def limit(ctx, name, p, n, vnew, x, fn, init=0.0):  # limit!  devices.jl:1209-1234
    lidx = ctx.alloc_limit(name, p, n, init=init)
    li = ctx.resolve_index(lidx)
    vold = 0.0 if len(x) == 0 else float(x[li - 1])
    if ctx.initjct:
        w = init
    else:
        w = fn(vnew, vold)
    ctx.record_limit_w(lidx, w)
    ctx.stamp_G(lidx, lidx, 1.0)
    ctx.stamp_G(lidx, p, -1.0)
    ctx.stamp_G(lidx, n, 1.0)
    return w


def stamp_limited_companion(ctx, p, n, w, I0, Gd):  # devices.jl:1251-1258
    stamp_conductance(ctx, p, n, Gd)
    Ieq = I0 - Gd * w
    ctx.stamp_b(p, -Ieq)
    ctx.stamp_b(n, Ieq)


def diode_vcrit(Is, Vt, n):  # devices.jl:1319-1320
    nVt = n * Vt
    return nVt * math.log(nVt / (math.sqrt(2.0) * Is))


def _exp(x):
    """exp with IEEE overflow (Inf), as Julia's: math.exp raises instead.  A non-limited junction driven far forward makes
    the stamps non-finite, which the Newton loops detect (solve.jl:560-566 retcode / :636 all(isfinite, F))."""
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def diode_iv(Is, nVt, v):  # _diode_iv  devices.jl:1333-1345
    xarg = v / nVt
    if xarg > 80.0:
        e80 = math.exp(80.0)
        return Is * (e80 * (1.0 + (xarg - 80.0)) - 1.0), Is / nVt * e80
    e = _exp(xarg)
    return Is * (e - 1.0), Is / nVt * e


def _register_diode_flicker(ctx, p, n, I0, KF, AF, FFE, name):  # devices.jl:1435-1443: KF |I0|^AF / f^FFE, only with KF > 0
    if KF > 0 and hasattr(ctx, "register_flicker_noise"):
        ctx.register_flicker_noise(p, n, KF * abs(I0) ** AF, FFE, name)


def stamp_diode(ctx, p, n, x, Is=1e-14, Vt=0.026, nf=1.0, limit_=True, name="D", KF=0.0, AF=1.0, FFE=1.0):  # devices.jl:1370-1428
    V0 = x_at(x, p) - x_at(x, n)
    nVt = nf * Vt
    if limit_:
        vcrit = diode_vcrit(Is, Vt, nf)
        w = limit(ctx, name + "_vdlim", p, n, V0, x, lambda vn, vo: pnjlim(vn, vo, nVt, vcrit)[0], init=vcrit)
        I0, Gd = diode_iv(Is, nVt, w)
        stamp_limited_companion(ctx, p, n, w, I0, Gd)
        if hasattr(ctx, "register_shot_noise"):          # devices.jl:1393-1397
            ctx.register_shot_noise(p, n, I0, name)
        _register_diode_flicker(ctx, p, n, I0, KF, AF, FFE, name)
    else:
        e = _exp(V0 / nVt)
        I0 = Is * (e - 1.0)
        Gd = Is / nVt * e
        Ieq = I0 - Gd * V0
        stamp_conductance(ctx, p, n, Gd)
        ctx.stamp_b(p, -Ieq)
        ctx.stamp_b(n, Ieq)
        if hasattr(ctx, "register_shot_noise"):          # devices.jl:1418-1419
            ctx.register_shot_noise(p, n, I0, name)
        _register_diode_flicker(ctx, p, n, I0, KF, AF, FFE, name)


def diode_junction_cap(V, Cj0, Vj, m):  # devices.jl:1505-1516
    Vmax = 0.9 * Vj
    if V < Vmax:
        return Cj0 / (1 - V / Vj) ** m
    C_at = Cj0 / (1 - Vmax / Vj) ** m
    dC = Cj0 * m / Vj / (1 - Vmax / Vj) ** (m + 1)
    return C_at + dC * (V - Vmax)


def stamp_diode_with_cap(ctx, p, n, x, Is=1e-14, Vt=0.026, nf=1.0, Cj0=1e-12, Vj=0.7, m=0.5, name="D", KF=0.0, AF=1.0, FFE=1.0):  # devices.jl:1558-1602
    V0 = x_at(x, p) - x_at(x, n)
    nVt = nf * Vt
    e = _exp(V0 / nVt)
    I0 = Is * (e - 1.0)
    G = Is / nVt * e
    Ieq = I0 - G * V0
    stamp_conductance(ctx, p, n, G)
    ctx.stamp_b(p, -Ieq)
    ctx.stamp_b(n, Ieq)
    if hasattr(ctx, "register_shot_noise"):              # devices.jl:1582-1585: shot and flicker noise at the junction bias
        ctx.register_shot_noise(p, n, I0, name)
    _register_diode_flicker(ctx, p, n, I0, KF, AF, FFE, name)
    stamp_capacitance(ctx, p, n, diode_junction_cap(V0, Cj0, Vj, m))


def stamp_simple_mosfet(ctx, d, g, s, x, Vth=0.5, K=1e-3, lam=0.0, Cgd=1e-15, Cgs=1e-15, name="M", KF=0.0, AF=1.0, FFE=1.0):  # devices.jl:1667-1749
    Vd, Vg, Vs = x_at(x, d), x_at(x, g), x_at(x, s)
    Vgs = Vg - Vs
    Vds = Vd - Vs
    if Vgs <= Vth:
        Ids = gm = gds = 0.0
    elif Vds <= Vgs - Vth:
        Ids = K * ((Vgs - Vth) * Vds - Vds ** 2 / 2)
        gm = K * Vds
        gds = K * (Vgs - Vth - Vds)
    else:
        Ids = K / 2 * (Vgs - Vth) ** 2 * (1 + lam * Vds)
        gm = K * (Vgs - Vth) * (1 + lam * Vds)
        gds = K / 2 * (Vgs - Vth) ** 2 * lam
    Ieq = Ids - gm * Vgs - gds * Vds
    ctx.stamp_G(d, d, gds)
    ctx.stamp_G(d, g, gm)
    ctx.stamp_G(d, s, -(gds + gm))
    ctx.stamp_G(s, d, -gds)
    ctx.stamp_G(s, g, -gm)
    ctx.stamp_G(s, s, gds + gm)
    ctx.stamp_b(d, -Ieq)
    ctx.stamp_b(s, Ieq)
    if gm > 0 and hasattr(ctx, "register_channel_thermal_noise"):     # devices.jl:1718-1724: 4kT (2/3) gm between drain and source, not in cutoff
        ctx.register_channel_thermal_noise(d, s, gm, name)
    if KF > 0 and Ids != 0.0 and hasattr(ctx, "register_flicker_noise"):   # devices.jl:1725-1732: KF |Ids|^AF / f^FFE
        ctx.register_flicker_noise(d, s, KF * abs(Ids) ** AF, FFE, name)
    stamp_capacitance(ctx, g, s, Cgs)
    stamp_capacitance(ctx, g, d, Cgd)
