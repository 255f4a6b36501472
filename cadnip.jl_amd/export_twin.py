"""Python twin of ``export_structure`` in julia/CadnipHIP.jl: the algorithm a Julia host runs to hand a compiled circuit
to the library, statement for statement, so that it can be executed and checked where Julia is absent.

Input is exactly what the reference's ``CompiledStructure`` holds (src/mna/precompile.jl:75-124) -- the shared CSC pattern ``colptr`` /
``rowval`` of G and C, the positional maps ``G_coo_to_idx`` / ``C_coo_to_idx`` (the k-th ``stamp_G!`` / ``stamp_C!`` call of a builder pass
writes ``nzval[map[k]]``, src/mna/value_only.jl:395-421), ``b_deferred_resolved`` (the row of the k-th ``stamp_b!``), the unknown counts --
plus the flattened device table in builder order (``DeviceRow``: type, resolved local unknowns, integer parameters, and the stamp program
of its ``stamp!`` method with the stamps into ground left out, as the reference skips them before the positional counter).  One walk over
the table splits the three positional streams into per-target slot lists in COO order; the output is the content of ``CadnipStructure``
(include/cadnip_hip.h): CSR pattern with the permutation back to the reference's nzval order, device blocks, gather lists.

The product's own structure discovery (structure.py: discover) reaches the same arrays by emulating the builder passes itself;
tests/test_host_cpu.py feeds THIS exporter the oracle's CompiledStructure and checks that both routes agree entry for entry, and
tests/test_gpu_parity.py runs the kernels on a handle built from its output.
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .structure import Block, Structure, shape_of


@dataclass
class DeviceRow:
    type: str                                   # device type key (structure.TYPE_ID maps it to CadnipDeviceType)
    nodes: List[int]                            # resolved local unknown indices, 0-based, -1 = ground
    ipar: List[int]
    program: List[Tuple[str, int, int, object]]  # (stream "G" | "C" | "b", local slot, local row, local col) in stamp order, no stamps into ground
    shape: Tuple[int, int, int, int, int, int]   # n_nodes, n_g, n_c, n_b, n_par, n_ipar of the type


def device_table(st: Structure) -> List[DeviceRow]:
    """The device table of a discovered circuit in builder (netlist) order -- what a Julia host derives from the netlist / sema result."""
    blk_of = {b.type: b for b in st.blocks}
    rows = []
    for info in st.opinfo:
        blk = blk_of[info["type"]]
        j = info["dev"]
        nodes = [int(v) for v in blk.nodes[:, j]]
        nl, ng, nc, nb, npar, nip = shape_of(info["type"])
        prog = [(s, k, rl, cl) for (s, k, rl, cl) in info["prog"] if nodes[rl] >= 0 and (cl is None or nodes[cl] >= 0)]
        rows.append(DeviceRow(info["type"], nodes, [int(v) for v in blk.ipar[:, j]], prog, (nl, ng, nc, nb, npar, nip)))
    return rows


def export_structure(cs, table: List[DeviceRow]):
    """CadnipHIP.jl: export_structure(cs, ctx, table).  ``cs``: an object with the fields of the reference's CompiledStructure
    (``n, n_nodes, n_currents, n_limits, colptr, rowval`` 0-based CSC, ``G_coo_to_idx, C_coo_to_idx, b_deferred_resolved`` 1-based as in
    Julia, ``G_n_coo, C_n_coo, limit_init``).  Returns a dict of the CadnipStructure arrays."""
    n = int(cs.n)
    colptr, rowval = np.asarray(cs.colptr), np.asarray(cs.rowval)
    nnz = len(rowval)
    # CSC -> CSR with the permutation back to the reference's nzval order
    rowcount = np.zeros(n + 1, dtype=np.int64)
    for r in rowval:
        rowcount[r + 1] += 1
    rowptr = np.cumsum(rowcount)
    fill = rowptr[:n].copy()
    colidx = np.zeros(nnz, dtype=np.int32)
    to_ref = np.zeros(nnz, dtype=np.int32)
    csr_of = np.zeros(nnz, dtype=np.int64)
    for j in range(n):
        for p in range(colptr[j], colptr[j + 1]):
            r = rowval[p]
            e = fill[r]
            fill[r] += 1
            colidx[e] = j
            to_ref[e] = p
            csr_of[p] = e
    # blocks: instances grouped by type in order of first appearance (one kernel per device type)
    order = []
    for d in table:
        if d.type not in order:
            order.append(d.type)
    members = {t: [i for i, d in enumerate(table) if d.type == t] for t in order}
    gb = cb = bb = 0
    bases, blocks = {}, []
    for t in order:
        idx = members[t]
        sh = table[idx[0]].shape
        cnt = len(idx)
        nodes = np.array([[table[i].nodes[k] for i in idx] for k in range(sh[0])], dtype=np.int32).reshape(sh[0], cnt)
        ipar = np.array([[table[i].ipar[k] if k < len(table[i].ipar) else 0 for i in idx] for k in range(max(sh[5], 1))], dtype=np.int32)
        bases[t] = (gb, cb, bb)
        blocks.append(dict(type=t, count=cnt, dev_index=idx, nodes=nodes, ipar=ipar, g_base=gb, c_base=cb, b_base=bb, n_g=sh[1], n_c=sh[2], n_b=sh[3], n_par=sh[4]))
        gb += sh[1] * cnt
        cb += sh[2] * cnt
        bb += sh[3] * cnt
    # one walk over the builder order: positional streams -> per-target slot lists, COO order preserved
    g_lists = [[] for _ in range(nnz)]
    c_lists = [[] for _ in range(nnz)]
    b_lists = [[] for _ in range(n)]
    kg = kc = kb = 0
    seen = {t: 0 for t in order}
    for d in table:
        dev = seen[d.type]
        seen[d.type] += 1
        cnt = len(members[d.type])
        g0, c0, b0 = bases[d.type]
        for (stream, slot, lrow, lcol) in d.program:
            if stream == "G":
                g_lists[csr_of[cs.G_coo_to_idx[kg] - 1]].append(g0 + slot * cnt + dev)
                kg += 1
            elif stream == "C":
                c_lists[csr_of[cs.C_coo_to_idx[kc] - 1]].append(c0 + slot * cnt + dev)
                kc += 1
            else:
                b_lists[cs.b_deferred_resolved[kb] - 1].append(b0 + slot * cnt + dev)      # the row the reference resolved for this stamp_b!
                kb += 1
    if kg != cs.G_n_coo or kc != cs.C_n_coo or kb != len(cs.b_deferred_resolved):
        raise ValueError("device table does not account for every stamp of the builder pass (%d/%d G, %d/%d C, %d/%d b)" % (
            kg, cs.G_n_coo, kc, cs.C_n_coo, kb, len(cs.b_deferred_resolved)))

    def flat(ls):
        ptr = np.zeros(len(ls) + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(l) for l in ls])
        return ptr, np.array([s for l in ls for s in l], dtype=np.int32)

    g_ptr, g_slots = flat(g_lists)
    c_ptr, c_slots = flat(c_lists)
    b_ptr, b_slots = flat(b_lists)
    diag = np.full(int(cs.n_nodes), -1, dtype=np.int32)
    for i in range(int(cs.n_nodes)):
        for e in range(rowptr[i], rowptr[i + 1]):
            if colidx[e] == i:
                diag[i] = e
                break
    return dict(n=n, rowptr=rowptr.astype(np.int32), colidx=colidx, to_ref_nz=to_ref, blocks=blocks, ns=(gb, cb, bb), g_ptr=g_ptr, g_slots=g_slots,
                c_ptr=c_ptr, c_slots=c_slots, b_ptr=b_ptr, b_slots=b_slots, diag_nz=diag)


def to_structure(ex, st: Structure) -> Structure:
    """A Structure (what hip.Handle takes) from an exported dict; names, wave data and breakpoints -- which the exporter does not produce --
    from ``st``."""
    import dataclasses
    blocks = [Block(b["type"], b["count"], b["dev_index"], b["nodes"], b["ipar"], b["g_base"], b["c_base"], b["b_base"], b["n_g"], b["n_c"], b["n_b"], b["n_par"])
              for b in ex["blocks"]]
    return dataclasses.replace(st, rowptr=ex["rowptr"], colidx=ex["colidx"], to_ref_nz=ex["to_ref_nz"], blocks=blocks, ns_g=ex["ns"][0], ns_c=ex["ns"][1],
                               ns_b=ex["ns"][2], g_ptr=ex["g_ptr"], g_slots=ex["g_slots"], c_ptr=ex["c_ptr"], c_slots=ex["c_slots"], b_ptr=ex["b_ptr"],
                               b_slots=ex["b_slots"], diag_nz=ex["diag_nz"])
