"""Operating-point read-out: device terminal currents and the models' small-signal variables at a DC solution.

Counterpart of the reference's operating-point channel (/root/reference/src/mna/context.jl:1200-1342; surfaced as
``op[:i_m1_d]``, ``op[:m1_gm]`` -- test/opinfo.jl).  The reference has every ``stamp!`` register its terminal currents and
op variables on the context; here they are recovered from the per-device contributions of one restamp at the solution
(``cadnip_get_contributions``): at a converged point the companion model of a device IS the device, so

    current into terminal t   =  sum over the KCL rows r of t (t itself and the internal nodes collapsed onto it)
                                 of   sum_k G_dev[r, k] u_k  -  b_dev[r]

and the small-signal conductances are entries of the device's own Jacobian block (for ``sp_mos1``: gm = dI(d_int)/dV(g), ...).
Names follow the reference: ``i_<device>_<terminal>`` and ``<device>_<variable>``, lower case.
"""
import numpy as np

from .structure import GND


def _index(st, node):
    """global unknown index of a typed local node handle"""
    if node == GND:
        return -1
    kind, k = node
    return {"n": 0, "c": st.n_nodes, "q": st.n_nodes + st.n_currents, "l": st.n_nodes + st.n_currents + st.n_charges}[kind] + k


def _blocks_by_type(st):
    return {b.type: b for b in st.blocks}


def device_matrix(st, info, Sg, Sb):
    """(local Jacobian block [n_local, n_local], local right-hand side [n_local]) of one device from the contribution arrays
    of one instance (ground rows and columns included: the program is local)."""
    blk = _blocks_by_type(st)[info["type"]]
    nl = len(info["nodes"])
    J, rhs = np.zeros((nl, nl)), np.zeros(nl)
    d = info["dev"]
    for stream, k, rl, cl in info["prog"]:
        if stream == "G":
            J[rl, cl] += Sg[blk.g_base + k * blk.count + d]
        elif stream == "b":
            rhs[rl] += Sb[blk.b_base + k * blk.count + d]
    return J, rhs


def operating_point(st, u, Sg, Sb, packed=None, instance=0):
    """``{name: value}`` of every terminal current and op variable of one instance; ``u`` [n], ``Sg`` / ``Sb`` the G / b
    contributions of that instance, ``packed``: the parameter blocks (pack_params) for the variables that need a model
    parameter (sp_mos1's vdsat / von)."""
    out = {}
    btypes = list(_blocks_by_type(st))
    for info in st.opinfo:
        J, rhs = device_matrix(st, info, Sg, Sb)
        ul = np.array([0.0 if nd == GND else u[_index(st, nd)] for nd in info["nodes"]])
        row_current = J @ ul - rhs                      # current the device draws out of every local KCL row
        name = info["name"].lower()
        for t, rows in zip(info["terminals"], info["groups"]):
            out["i_%s_%s" % (name, t)] = float(sum(row_current[r] for r in rows))
        ty = info["type"]
        if ty == "MOS1":
            # local unknowns: d g s b d_int s_int | limits (g,s_int) (d_int,s_int) (b,s_int) (b,d_int) | charges
            vgs, vds, vbs = ul[6], ul[7], ul[8]
            par = None
            if packed is not None:
                par = packed[btypes.index("MOS1")][instance][:, info["dev"]]
            typ = 1.0 if par is None else float(par[0])
            mode = 1.0 if typ * vds >= 0 else -1.0
            gbd, gbs = -J[3, 4], -J[3, 5]
            out.update({name + "_gm": mode * J[4, 1], name + "_gds": J[4, 4] + J[3, 4], name + "_gmbs": mode * (J[4, 3] + gbd) if mode > 0 else -(J[4, 3] + gbd),
                        name + "_gbd": gbd, name + "_gbs": gbs, name + "_vgs": vgs, name + "_vds": vds, name + "_vbs": vbs,
                        name + "_id": out["i_%s_d" % name]})
            if par is not None:
                # DEVfetlim's threshold (mos1.va:999-1010): von = type tVbi + gamma sarg, vdsat = max(vgs - von, 0) at level 1
                t_phi, t_vbi, gamma = float(par[2]), float(par[3]), float(par[5])
                vb = typ * (vbs if mode > 0 else vbs - vds)
                sarg = np.sqrt(t_phi - vb) if vb <= 0 else max(0.0, np.sqrt(t_phi) - vb / (2.0 * np.sqrt(t_phi)))
                von = t_vbi * typ + gamma * sarg
                vg = typ * (vgs if mode > 0 else vgs - vds)
                out[name + "_von"] = typ * von
                out[name + "_vdsat"] = typ * max(vg - von, 0.0)
        elif ty == "D":
            out[name + "_gd"] = J[0, 0]                   # the junction conductance the AC analysis linearises around
            out[name + "_vd"] = ul[0] - ul[1]
        elif ty == "R":
            out[name + "_i"] = out["i_%s_p" % name]
    return out
