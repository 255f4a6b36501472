# CadnipHIP.jl -- the reference-side binding of libcadnip_hip.so (include/cadnip_hip.h).
#
# This is the `ccall` shim a Cadnip.jl maintainer adds to route the transient hot path through the
# MI355X library.  It subtypes nothing new: it provides a `GPUEvalWorkspace` that is passed as SciML's
# `p` exactly where `EvalWorkspace` is passed today (src/mna/solve.jl:2138-2160, 2497-2519), with
# `fast_rebuild!` / `fast_residual!` / `fast_jacobian!` methods of the same signatures
# (src/mna/precompile.jl:493-603) and a KLU-shaped linear solver (src/mna/solve.jl:612-613,667-670).
#
# Julia is not installed in the build environment, so this file is shipped as source only (it is not
# executed by the test-suite); every `ccall` below matches a prototype in include/cadnip_hip.h and the
# Python ctypes binding cadnip.jl_amd/hip.py exercises the same entry points on the GPU.
module CadnipHIP

using SparseArrays, LinearAlgebra

const LIB = get(ENV, "CADNIP_HIP_LIB", "libcadnip_hip.so")

const CADNIP_OK, CADNIP_BADARG, CADNIP_SINGULAR, CADNIP_NONFINITE, CADNIP_HIPERROR, CADNIP_NOTREADY, CADNIP_NOCONV = 0:6

# error convention of the reference: exceptions (src/mna/solve.jl:887-897 catches these two)
function check(rc::Cint, what)
    rc == CADNIP_OK && return nothing
    rc == CADNIP_SINGULAR && throw(LinearAlgebra.SingularException(0))
    rc == CADNIP_NONFINITE && throw(DomainError(what, "non-finite stamp value"))
    error("$what failed with status $rc")
end

struct CadnipDeviceBlock            # == typedef struct CadnipDeviceBlock
    type::Int32; count::Int32
    n_nodes::Int32; nodes::Ptr{Int32}
    n_ipar::Int32; ipar::Ptr{Int32}
    n_par::Int32
    g_base::Int32; c_base::Int32; b_base::Int32
    n_g::Int32; n_c::Int32; n_b::Int32
end

struct CadnipStructure              # == typedef struct CadnipStructure
    n::Int32; n_nodes::Int32; n_currents::Int32; n_charges::Int32; n_limits::Int32
    nnz::Int32; rowptr::Ptr{Int32}; colidx::Ptr{Int32}; to_ref_nz::Ptr{Int32}
    n_blocks::Int32; blocks::Ptr{CadnipDeviceBlock}
    n_wave_data::Int32; wave_data::Ptr{Float64}
    ns_g::Int32; ns_c::Int32; ns_b::Int32
    g_ptr::Ptr{Int32}; g_slots::Ptr{Int32}; c_ptr::Ptr{Int32}; c_slots::Ptr{Int32}; b_ptr::Ptr{Int32}; b_slots::Ptr{Int32}
    diag_nz::Ptr{Int32}; limit_init::Ptr{Float64}
end

struct CadnipSpec
    mode::Int32; gmin::Float64; gshunt::Float64; srcFact::Float64
end

"""
    GPUEvalWorkspace

Drop-in for `MNA.EvalWorkspace` (src/mna/precompile.jl:168-172).  Holds the library handle plus the
host mirrors the integrator reads (`b`, `limit_w`), and remembers the (u, t) of the last stamping so
that `fast_residual!` and `fast_jacobian!` at the same point share one restamp (the reference restamps
in each: precompile.jl:548,572).
"""
mutable struct GPUEvalWorkspace
    handle::Ptr{Cvoid}
    n::Int; nnz::Int; n_limits::Int
    keep::Vector{Any}                 # arrays the C structure points into
    last_u::Vector{Float64}; last_t::Float64; stamped::Bool
    gamma::Vector{Float64}; tbuf::Vector{Float64}
    J_ref::Vector{Float64}            # J in the reference's CSC nzval order
end

"""
    export_structure(cs::MNA.CompiledStructure, ctx::MNA.MNAContext, table) -> GPUEvalWorkspace

Build the `CadnipStructure` from what Cadnip already has: the unified CSC pattern of `cs.G`
(precompile.jl:413-421) converted to CSR + `to_ref_nz`, and the device table `table` (one entry per
instance: type id, resolved node indices, slot bases) produced by walking `ctx`'s COO streams in stamp
order -- `cs.G_coo_to_idx[k]` is exactly the nz each G slot gathers into (value_only.jl:395-421).
"""
function export_structure(n, n_nodes, n_currents, n_charges, n_limits,
                          rowptr::Vector{Int32}, colidx::Vector{Int32}, to_ref_nz::Vector{Int32},
                          blocks::Vector{CadnipDeviceBlock}, wave_data::Vector{Float64},
                          ns::NTuple{3,Int32}, g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots,
                          diag_nz::Vector{Int32}, limit_init::Vector{Float64}; n_instances=1, device=0)
    keep = Any[rowptr, colidx, to_ref_nz, blocks, wave_data, g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots, diag_nz, limit_init]
    s = Ref(CadnipStructure(n, n_nodes, n_currents, n_charges, n_limits, length(colidx), pointer(rowptr), pointer(colidx),
            pointer(to_ref_nz), length(blocks), pointer(blocks), length(wave_data), pointer(wave_data), ns[1], ns[2], ns[3],
            pointer(g_ptr), pointer(g_slots), pointer(c_ptr), pointer(c_slots), pointer(b_ptr), pointer(b_slots),
            pointer(diag_nz), pointer(limit_init)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:cadnip_create, LIB), Cint, (Ref{CadnipStructure}, Int32, Int32, Ref{Ptr{Cvoid}}), s, n_instances, device, h), "cadnip_create")
    GPUEvalWorkspace(h[], n, length(colidx), n_limits, keep, zeros(n), NaN, false, zeros(1), zeros(1), zeros(length(colidx)))
end

set_params!(ws::GPUEvalWorkspace, block::Integer, par::Array{Float64,3}) =
    check(ccall((:cadnip_set_params, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ws.handle, block, par), "cadnip_set_params")

function set_spec!(ws::GPUEvalWorkspace; mode::Symbol=:tran, gmin=1e-12, gshunt=0.0, srcFact=1.0)
    m = mode === :dcop ? 0 : mode === :tran ? 1 : 2
    check(ccall((:cadnip_set_spec, LIB), Cint, (Ptr{Cvoid}, Ref{CadnipSpec}), ws.handle, Ref(CadnipSpec(m, gmin, gshunt, srcFact))), "cadnip_set_spec")
end

# ---- the three callbacks ---------------------------------------------------------------------------
"fast_rebuild!(ws, u, t)  -- src/mna/precompile.jl:493-537"
function fast_rebuild!(ws::GPUEvalWorkspace, u::AbstractVector, t::Real)
    ws.tbuf[1] = Float64(t)
    check(ccall((:cadnip_rebuild, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ws.handle, u, ws.tbuf), "cadnip_rebuild")
    copyto!(ws.last_u, u); ws.last_t = Float64(t); ws.stamped = true
    return nothing
end

_same_point(ws, u, t) = ws.stamped && ws.last_t == Float64(t) && ws.last_u == u

"fast_residual!(resid, du, u, ws, t)  -- src/mna/precompile.jl:546-557"
function fast_residual!(resid::AbstractVector, du::AbstractVector, u::AbstractVector, ws::GPUEvalWorkspace, t::Real)
    _same_point(ws, u, t) || fast_rebuild!(ws, u, t)
    check(ccall((:cadnip_residual, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ws.handle, du, u, resid), "cadnip_residual")
    return nothing
end

"fast_jacobian!(J, du, u, ws, gamma, t)  -- src/mna/precompile.jl:568-585; J shares cs.G's colptr/rowval"
function fast_jacobian!(J::SparseMatrixCSC, du::AbstractVector, u::AbstractVector, ws::GPUEvalWorkspace, gamma::Real, t::Real)
    _same_point(ws, u, t) || fast_rebuild!(ws, u, t)
    ws.gamma[1] = Float64(gamma)
    check(ccall((:cadnip_jacobian, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ws.handle, ws.gamma, nonzeros(J)), "cadnip_jacobian")
    return nothing
end

"fast_jacobian!(J::Matrix, ...) -- the dense form, src/mna/precompile.jl:588-603 (test/mna/audio_integration.jl:505-520 solves `J \\ resid` with it)"
function fast_jacobian!(J::Matrix{Float64}, du::AbstractVector, u::AbstractVector, ws::GPUEvalWorkspace, gamma::Real, t::Real)
    _same_point(ws, u, t) || fast_rebuild!(ws, u, t)
    ws.gamma[1] = Float64(gamma)
    size(J) == (ws.n, ws.n) || throw(DimensionMismatch("J must be n x n"))
    check(ccall((:cadnip_jacobian_dense, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ws.handle, ws.gamma, J), "cadnip_jacobian_dense")
    return nothing
end

# ---- ODE form (src/mna/solve.jl:2241-2276): mass matrix cs.C, rhs! = b - G*u, jac! = -G ------------------------
function ode_rhs!(du::AbstractVector, u::AbstractVector, ws::GPUEvalWorkspace, t::Real)
    ws.tbuf[1] = Float64(t)
    check(ccall((:cadnip_ode_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ws.handle, u, ws.tbuf, du), "cadnip_ode_rhs")
end
function ode_jac!(J::SparseMatrixCSC, u::AbstractVector, ws::GPUEvalWorkspace, t::Real)
    ws.tbuf[1] = Float64(t)
    check(ccall((:cadnip_ode_jacobian, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ws.handle, u, ws.tbuf, nonzeros(J)), "cadnip_ode_jacobian")
end

# ---- KLU-shaped linear solver: symbolic once, numeric refactor per Jacobian, solve per iteration -----------
analyze!(ws::GPUEvalWorkspace) = check(ccall((:cadnip_analyze, LIB), Cint, (Ptr{Cvoid}, Int32), ws.handle, 0), "cadnip_analyze")
factor!(ws::GPUEvalWorkspace) = check(ccall((:cadnip_factor, LIB), Cint, (Ptr{Cvoid},), ws.handle), "cadnip_factor")
function solve!(x::Vector{Float64}, ws::GPUEvalWorkspace, rhs::Vector{Float64})
    check(ccall((:cadnip_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ws.handle, rhs, x), "cadnip_solve")
    return x
end

# One Newton iteration in one ccall (cadnip_newton_step): resid = C du + G u - b, [J = G + gamma C refactored when `refresh`,] delta = J^-1 resid.
# For a hand-written Newton loop around the library (the shape of _dc_pcnr_newton, src/mna/solve.jl:599-698, on the DAE residual) and for an
# integrator whose nonlinear solver can be replaced: five entry points and five synchronisations become one.  Returns ||resid||_2.
function newton_step!(delta::Vector{Float64}, du::Vector{Float64}, u::Vector{Float64}, ws::GPUEvalWorkspace, gamma::Real, t::Real; refresh::Bool=true)
    ws.tbuf[1] = Float64(t); ws.gamma[1] = Float64(gamma)
    nrm = Ref{Float64}(0.0)
    check(ccall((:cadnip_newton_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ws.handle, u, du, ws.gamma, ws.tbuf, refresh ? 1 : 0, delta, nrm, C_NULL), "cadnip_newton_step")
    return nrm[]
end

# SciML closures identical in shape to make_workspace_dae_residual / _jacobian (src/mna/solve.jl:2497-2519)
make_gpu_dae_residual() = (resid, du, u, ws::GPUEvalWorkspace, t) -> fast_residual!(resid, du, u, ws, t)
make_gpu_dae_jacobian() = (J, du, u, ws::GPUEvalWorkspace, gamma, t) -> fast_jacobian!(J, du, u, ws, gamma, t)

# ---- walking CompiledStructure into CadnipStructure (SURVEY.md section 8f-1) --------------------------------------------
# What the library needs beyond the matrices is, per device, WHICH stamp of the builder pass lands WHERE.  The reference
# already has that in positional form: the k-th stamp_G! call of a builder pass writes nzval[cs.G_coo_to_idx[k]]
# (src/mna/value_only.jl:395-421, src/mna/precompile.jl:75-124), stamp_C! likewise, and the k-th deferred b stamp goes to row
# cs.b_deferred_resolved[k] (direct b stamps: the row itself).  A device table row says how many stamps its stamp! method
# issues and in which local order (`program`: the same fixed per-type programs as cadnip.jl_amd/structure.py PROGRAMS, i.e.
# the statement order of src/mna/devices.jl / the generated stamp! of src/vasim.jl:3319-3521), so one walk over the table in
# builder order splits the three positional streams into (block, local slot, device) -> target.

"One instance of the flattened netlist: type id (include/cadnip_hip.h CadnipDeviceType), resolved local unknown indices
(0-based, -1 = ground), integer parameters, and its stamp program: (stream, local slot, local row, local col) with stream
:G / :C / :b in the order the stamp! method issues them; stamps into ground are NOT listed (the reference skips them before
the positional counter, value_only.jl:395-397)."
struct DeviceRow
    type::Int32
    nodes::Vector{Int32}
    ipar::Vector{Int32}
    program::Vector{Tuple{Symbol,Int32,Int32,Int32}}
    shape::NTuple{6,Int32}            # n_nodes, n_g, n_c, n_b, n_par, n_ipar of the type (CADNIP_VA_SHAPES for generated modules)
end

"""
    export_structure(cs::MNA.CompiledStructure, ctx::MNA.MNAContext, table::Vector{DeviceRow}; n_instances=1, device=0)

(The algorithm below has a Python twin, `cadnip.jl_amd/export_twin.py`, statement for statement: Julia is not installed where this
library is built, so the twin is what the test-suite executes -- on a restatement of `compile_structure` -- and what must stay in
step with this function.)

`cs`, `ctx` as `compile_structure` leaves them (precompile.jl:312-443); `table` in builder order.  Returns the
`GPUEvalWorkspace` whose CSR pattern, `to_ref_nz` permutation and per-nz gather lists reproduce `cs.G` / `cs.C` / `b`
addition for addition (COO order), so `nonzeros(J)` comes back in `cs.G`'s own nzval order.
"""
function export_structure(cs, ctx, table::Vector{DeviceRow}; n_instances=1, device=0)
    n = cs.n; G = cs.G
    colptr, rowval = G.colptr, G.rowval                       # cs.C shares the pattern (precompile.jl:413-421)
    nnz_ = length(rowval)
    # CSC -> CSR with the permutation back to the reference's nzval order
    rowcount = zeros(Int32, n + 1)
    for r in rowval; rowcount[r + 1] += 1; end
    rowptr = cumsum(rowcount)                                  # 0-based row pointer
    fill_ = copy(rowptr[1:n]); colidx = zeros(Int32, nnz_); to_ref = zeros(Int32, nnz_); csr_of = zeros(Int32, nnz_)
    for j in 1:n, p in colptr[j]:(colptr[j + 1] - 1)
        r = rowval[p]; e = (fill_[r] += 1)
        colidx[e] = j - 1; to_ref[e] = p - 1; csr_of[p] = e - 1
    end
    # blocks: instances grouped by type in order of first appearance (one kernel per device type)
    order = unique(Int32[d.type for d in table])
    members = Dict(t => findall(d -> d.type == t, table) for t in order)
    gb = cb = bb = Int32(0); bases = Dict{Int32,NTuple{3,Int32}}()
    blocks = CadnipDeviceBlock[]; keep = Any[]
    for t in order
        idx = members[t]; sh = table[idx[1]].shape; cnt = Int32(length(idx))
        nodes = Int32[table[i].nodes[k] for k in 1:sh[1], i in idx] |> permutedims |> vec        # [n_local][count]
        ipar = Int32[get(table[i].ipar, k, 0) for k in 1:max(sh[6], 1), i in idx] |> permutedims |> vec
        push!(keep, nodes, ipar)
        bases[t] = (gb, cb, bb)
        push!(blocks, CadnipDeviceBlock(t, cnt, sh[1], pointer(nodes), sh[6], pointer(ipar), sh[5], gb, cb, bb, sh[2], sh[3], sh[4]))
        gb += sh[2] * cnt; cb += sh[3] * cnt; bb += sh[4] * cnt
    end
    # one walk over the builder order: positional streams -> per-target slot lists, COO order preserved
    g_lists = [Int32[] for _ in 1:nnz_]; c_lists = [Int32[] for _ in 1:nnz_]; b_lists = [Int32[] for _ in 1:n]
    kg = kc = kb = 0
    seen = Dict(t => 0 for t in order)
    for d in table
        dev = seen[d.type]; seen[d.type] += 1
        cnt = length(members[d.type]); (g0, c0, b0) = bases[d.type]
        for (stream, slot, lrow, lcol) in d.program
            if stream === :G
                kg += 1; push!(g_lists[csr_of[cs.G_coo_to_idx[kg]] + 1], g0 + slot * cnt + dev)
            elseif stream === :C
                kc += 1; push!(c_lists[csr_of[cs.C_coo_to_idx[kc]] + 1], c0 + slot * cnt + dev)
            else
                kb += 1; push!(b_lists[cs.b_deferred_resolved[kb]], b0 + slot * cnt + dev)     # the row the reference resolved for the kb-th stamp_b! (value_only.jl:440-478: every b stamp is positional; ground rows never reach the counter)
            end
        end
    end
    (kg == cs.G_n_coo && kc == cs.C_n_coo && kb == length(cs.b_deferred_resolved)) ||
        error("device table does not account for every stamp of the builder pass ($kg/$(cs.G_n_coo) G, $kc/$(cs.C_n_coo) C, $kb/$(length(cs.b_deferred_resolved)) b)")
    flat(ls) = (Int32[0; cumsum(length.(ls))], reduce(vcat, ls; init=Int32[]))
    (g_ptr, g_slots), (c_ptr, c_slots), (b_ptr, b_slots) = flat(g_lists), flat(c_lists), flat(b_lists)
    diag = Int32[(e = findfirst(==(i - 1), view(colidx, rowptr[i] + 1:rowptr[i + 1])); e === nothing ? -1 : rowptr[i] + e - 1) for i in 1:n]
    ws = export_structure(n, cs.n_nodes, cs.n_currents, ctx.n_charges, cs.n_limits, Int32.(rowptr), colidx, to_ref, blocks, Float64[],
                          (gb, cb, bb), g_ptr, g_slots, c_ptr, c_slots, b_ptr, b_slots, diag, Vector{Float64}(cs.limit_init);
                          n_instances=n_instances, device=device)
    append!(ws.keep, keep)
    return ws
end

# The GPU LU behind LinearSolve.jl -- what `KLUFactorization()` is to src/mna/solve.jl:612-613 and to DFBDF / FBDF's `linsolve` -- lives in
# CadnipHIPLinearSolve.jl beside this file (it needs LinearSolve.jl and SciMLBase.jl, which this module does not).

Base.close(ws::GPUEvalWorkspace) = (ccall((:cadnip_destroy, LIB), Cvoid, (Ptr{Cvoid},), ws.handle); ws.handle = C_NULL; nothing)

end # module
