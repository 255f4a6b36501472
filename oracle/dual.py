"""TEST INFRASTRUCTURE (oracle) -- forward-mode dual numbers.

Restates the two ForwardDiff tags the reference uses for Verilog-A stamping
(/root/reference/src/mna/contrib.jl:54-101, 356-375):

* ``Dual``  == ``Dual{JacobianTag}``: value + W partials (d/dV_k and one slot
  per ``$limit`` call site, /root/reference/src/vasim.jl:3012-3017).
* ``CDual`` == ``Dual{ContributionTag}`` ("s-dual"): resistive part + reactive
  (charge) part; ``va_ddt(x) = CDual(0, x)`` (contrib.jl:356-375).

Comparison operators compare values only (ForwardDiff semantics); ``dmax`` /
``dmin`` / ``dabs`` select by value and carry the selected operand's partials.
Pure-Python scalar arithmetic: meant for small cases.
"""
import math
import numpy as np


def val(x):
    """extract_value (context.jl:910): strip every dual layer."""
    if isinstance(x, Dual):
        return x.v
    if isinstance(x, CDual):
        return val(x.r)
    return float(x)


class Dual:
    __slots__ = ("v", "p")
    __array_priority__ = 1000

    def __init__(self, v, p):
        self.v = float(v)
        self.p = p

    @staticmethod
    def seed(v, k, width):
        p = np.zeros(width)
        p[k] = 1.0
        return Dual(v, p)

    # -- arithmetic ---------------------------------------------------------
    def __add__(self, o):
        if isinstance(o, CDual):
            return NotImplemented
        if isinstance(o, Dual):
            return Dual(self.v + o.v, self.p + o.p)
        return Dual(self.v + o, self.p)

    __radd__ = __add__

    def __neg__(self):
        return Dual(-self.v, -self.p)

    def __sub__(self, o):
        if isinstance(o, CDual):
            return NotImplemented
        if isinstance(o, Dual):
            return Dual(self.v - o.v, self.p - o.p)
        return Dual(self.v - o, self.p)

    def __rsub__(self, o):
        return Dual(o - self.v, -self.p)

    def __mul__(self, o):
        if isinstance(o, CDual):
            return NotImplemented
        if isinstance(o, Dual):
            return Dual(self.v * o.v, self.p * o.v + o.p * self.v)
        return Dual(self.v * o, self.p * o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, CDual):
            return NotImplemented
        if isinstance(o, Dual):
            q = self.v / o.v
            return Dual(q, (self.p - q * o.p) / o.v)
        return Dual(self.v / o, self.p / o)

    def __rtruediv__(self, o):
        q = o / self.v
        return Dual(q, (-q / self.v) * self.p)

    # -- comparisons on value ----------------------------------------------
    def __lt__(self, o):
        return self.v < val(o)

    def __le__(self, o):
        return self.v <= val(o)

    def __gt__(self, o):
        return self.v > val(o)

    def __ge__(self, o):
        return self.v >= val(o)

    def __eq__(self, o):
        return self.v == val(o)

    def __ne__(self, o):
        return self.v != val(o)

    __hash__ = None

    def __repr__(self):
        return "Dual(%r, %r)" % (self.v, self.p)


class CDual:
    """Dual{ContributionTag}: r = resistive part, q = reactive (charge) part."""
    __slots__ = ("r", "q")

    def __init__(self, r, q):
        self.r = r
        self.q = q

    def __add__(self, o):
        if isinstance(o, CDual):
            return CDual(self.r + o.r, self.q + o.q)
        return CDual(self.r + o, self.q)

    __radd__ = __add__

    def __neg__(self):
        return CDual(-self.r, -self.q)

    def __sub__(self, o):
        if isinstance(o, CDual):
            return CDual(self.r - o.r, self.q - o.q)
        return CDual(self.r - o, self.q)

    def __rsub__(self, o):
        return CDual(o - self.r, -self.q)

    def __mul__(self, o):
        if isinstance(o, CDual):
            return CDual(self.r * o.r, self.r * o.q + self.q * o.r)
        return CDual(self.r * o, self.q * o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, CDual):
            raise TypeError("division by a ContributionTag dual is not used by any model")
        return CDual(self.r / o, self.q / o)


def va_ddt(x):
    """contrib.jl:356-375."""
    if isinstance(x, CDual):
        return CDual(0.0 * x.r, x.r)
    return CDual(0.0 * x, x)


# -- elementary functions ------------------------------------------------------
def dsqrt(x):
    if isinstance(x, Dual):
        s = math.sqrt(x.v)
        return Dual(s, x.p * (0.5 / s)) if s != 0.0 else Dual(s, x.p * math.inf)
    return math.sqrt(x)


def dexp(x):
    if isinstance(x, Dual):
        e = math.exp(x.v)
        return Dual(e, x.p * e)
    return math.exp(x)


def dln(x):
    if isinstance(x, Dual):
        return Dual(math.log(x.v), x.p / x.v)
    return math.log(x)


def dabs(x):
    if isinstance(x, Dual):
        if x.v >= 0:
            return x
        return -x
    return abs(x)


def dmax(a, b):
    return a if val(a) > val(b) else b if val(b) > val(a) else a


def dmin(a, b):
    return a if val(a) < val(b) else b if val(b) < val(a) else a


def partials(x, width):
    if isinstance(x, Dual):
        return x.p
    return np.zeros(width)
