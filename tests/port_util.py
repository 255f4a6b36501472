"""Helpers that put one sweep instance of a product Structure onto the CPU port (oracle)."""
import numpy as np
from cadnip_jl_amd import api

import cadnip_jl_amd as cj
from cadnip_jl_amd import hip
from cadnip_jl_amd.structure import TYPE_ID
from oracle import cpu_port


def make_port(circ, params, temp=27.0, mode="tran", gmin=1e-12):
    st = cj.discover(circ, params)
    packed = cj.pack_params(st, circ, {k: np.array([float(v)]) for k, v in params.items()}, np.array([float(temp)]), 1, gmin=gmin)
    port = cpu_port.Port(st, [p[0] for p in packed], TYPE_ID)
    port.set_spec(mode=mode, gmin=gmin)
    return st, port


def analyze_port(st, port, vscale, gamma=1e9, seed=1234, n_samples=6):
    """Same sample construction as BatchSimulator.analyze (api.py), evaluated with the port's stamps;
    the symbolic phase itself is the product's host code (cadnip_host_lu_analyze)."""
    rng = np.random.default_rng(seed)
    acc = np.zeros(st.nnz)
    for k in range(n_samples):
        if k == 0:
            u = np.zeros(st.n)
            u[st.n - st.n_limits:] = st.limit_init
            port.set_spec(initjct=1)
        elif k == 1:
            u = np.zeros(st.n)
        else:
            u = (rng.random(st.n) * 1.2 - 0.1) * vscale
            u[st.n_nodes:st.n_nodes + st.n_currents] = 0.0
        G, C, b, lw = port.rebuild(u, 0.0)
        port.set_spec(initjct=0)
        acc = np.maximum(acc, api.clip_sample(G + gamma * C)[0])
    prog = hip.host_lu_analyze(st.n, st.rowptr, st.colidx, acc, sample=True, leaves=hip.leaves_of(st))   # the pivot order a handle uses
    port.set_lu(prog)
    return prog
