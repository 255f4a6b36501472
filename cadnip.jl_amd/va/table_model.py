"""$table_model (Verilog-AMS LRM 9.21) as the reference supports it (src/vasim.jl:752-860, src/mna/table_model.jl): a whitespace-separated
table file on a regular product grid, control string "<interp per dimension>;<column>" with interpolation '1' (multilinear) and one
extrapolation code for all dimensions -- 'L' linear, 'C' constant, 'E' error.  ('D', the discrete look-up, is refused: its tie rule is not
pinned by any fixture.)  Used on the host only: this build accepts $table_model calls whose inputs are decided by the parameters, evaluates
them when the parameters are packed and hands the value to the device code as one more entry of the instance's parameter block."""
import os

import numpy as np

_cache = {}


def parse_control(ctrl, n_inputs):
    parts = ctrl.split(";")
    if len(parts) != 2:
        raise ValueError('$table_model control string must be "<interp>;<col>"; got "%s"' % ctrl)
    dims = [d.strip() for d in parts[0].split(",")]
    if len(dims) != n_inputs:
        raise ValueError('$table_model control string specifies %d dim(s) but the call has %d input argument(s); got "%s"' % (len(dims), n_inputs, ctrl))
    extrap = set()
    for d in dims:
        if not d or d[0] not in "1D" or len(d) > 2 or (len(d) == 2 and d[1] not in "LCE"):
            raise ValueError('$table_model: unsupported interpolation spec "%s" in "%s"' % (d, ctrl))
        if d[0] == "D":
            raise ValueError('$table_model: discrete look-up ("D") is not supported')
        extrap.add(d[1] if len(d) == 2 else "L")
    if len(extrap) != 1:
        raise ValueError('$table_model requires uniform extrapolation across dimensions; got "%s"' % ctrl)
    return extrap.pop(), int(parts[1])


def parse_file(path, n_inputs):
    key = (os.path.abspath(path), n_inputs)
    if key not in _cache:
        rows = []
        for line in open(path):
            s = line.split("#", 1)[0].strip()
            if s:
                rows.append([float(t) for t in s.split()])
        if not rows or any(len(r) != len(rows[0]) for r in rows) or len(rows[0]) <= n_inputs:
            raise ValueError("$table_model file %s: need a rectangular table with %d input column(s) and at least one more" % (path, n_inputs))
        a = np.array(rows)
        axes = [np.unique(a[:, k]) for k in range(n_inputs)]
        if any(len(ax) < 2 for ax in axes) or len(rows) != int(np.prod([len(ax) for ax in axes])):
            raise ValueError("$table_model file %s: not a regular product grid with at least two points per dimension" % path)
        out = np.full([len(ax) for ax in axes] + [a.shape[1] - n_inputs], np.nan)
        for r in a:
            out[tuple(int(np.searchsorted(axes[k], r[k])) for k in range(n_inputs))] = r[n_inputs:]
        if np.isnan(out).any():
            raise ValueError("$table_model file %s: duplicate or missing grid points" % path)
        _cache[key] = (axes, out)
    return _cache[key]


def lookup(path, ctrl, xs):
    """value of column ``col`` (1-based among the dependent columns) at the point ``xs``"""
    xs = [float(x) for x in xs]
    extrap, col = parse_control(ctrl, len(xs))
    axes, out = parse_file(path, len(xs))
    ys = out[..., col - 1]
    xc = [min(max(x, ax[0]), ax[-1]) for x, ax in zip(xs, axes)]
    if extrap == "E" and xc != xs:
        raise ValueError("$table_model: %s lies outside the table %s" % (xs, path))
    cell = [min(int(np.searchsorted(ax, x, side="right")) - 1, len(ax) - 2) for x, ax in zip(xc, axes)]
    frac = [(x - ax[i]) / (ax[i + 1] - ax[i]) for x, ax, i in zip(xc, axes, cell)]

    def multilinear(fr):
        v = 0.0
        for corner in range(1 << len(xs)):
            w, idx = 1.0, []
            for d in range(len(xs)):
                hi = (corner >> d) & 1
                w *= fr[d] if hi else 1.0 - fr[d]
                idx.append(cell[d] + hi)
            v += w * ys[tuple(idx)]
        return v
    v = multilinear(frac)
    if extrap == "L":                         # the boundary value continued along the interpolant's gradient there
        for d in range(len(xs)):
            if xs[d] != xc[d]:
                f1 = list(frac); f1[d] = 1.0
                f0 = list(frac); f0[d] = 0.0
                slope = (multilinear(f1) - multilinear(f0)) / (axes[d][cell[d] + 1] - axes[d][cell[d]])
                v += slope * (xs[d] - xc[d])
    return v
