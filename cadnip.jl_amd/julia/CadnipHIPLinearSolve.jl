# CadnipHIPLinearSolve.jl -- the library's sparse LU as a LinearSolve.jl algorithm: what `KLUFactorization()` is to the reference's PCNR loop
# (`LinearSolve.init(LinearProblem(cs.G, F), KLUFactorization())` + `solve!`, src/mna/solve.jl:612-613, 667-670) and to the `linsolve=`
# keyword of DFBDF / FBDF / Rodas (src/sweeps.jl:588-665 passes the solver through).  Load after CadnipHIP.jl:
#
#     include("CadnipHIP.jl"); include("CadnipHIPLinearSolve.jl")
#     using .CadnipHIPLinearSolve: CadnipLU
#     sol = solve(prob, FBDF(linsolve = CadnipLU(ws)))              # ws::CadnipHIP.GPUEvalWorkspace of the same circuit
#
# The Jacobian never travels: `fast_jacobian!` leaves J = G + gamma C on the device (the nzval copy it returns is for the host's own
# use), `factor!` refactors it there with the pivot sequence of the symbolic phase (klu_refactor semantics; CADNIP_SINGULAR becomes a
# `SingularException`, which the reference's callers catch: src/mna/solve.jl:887-897), `solve!` runs the two triangular sweeps.
# Sundials' IDA takes its linear solver by name (`IDA(linear_solver=:KLU)`, src/sweeps.jl:600); a host that wants this factorisation
# inside IDA wraps the same three calls in a SUNLinearSolver through Sundials.jl's `LinSolHandle` (setup = factor!, solve = solve!), or
# replaces IDA's Newton iteration by `CadnipHIP.newton_step!` (one call per iteration).
#
# Never executed where this repository is built (no Julia there): kept to the documented LinearSolve.jl v2 / v3 interface
# (`init_cacheval`, `SciMLBase.solve!(::LinearCache, alg)`, `cache.isfresh`, `build_linear_solution`).
module CadnipHIPLinearSolve

using LinearSolve, SciMLBase
using ..CadnipHIP: GPUEvalWorkspace, analyze!, factor!, solve!

export CadnipLU

"LinearSolve algorithm backed by one `GPUEvalWorkspace` (one circuit structure, one GPU)."
struct CadnipLU <: LinearSolve.SciMLLinearSolveAlgorithm
    ws::GPUEvalWorkspace
end

# symbolic phase once per cache: pivot order and fill from the Jacobian the workspace holds (cadnip_analyze)
function LinearSolve.init_cacheval(alg::CadnipLU, A, b, u, Pl, Pr, maxiters::Int, abstol, reltol, verbose::Bool, assumptions::LinearSolve.OperatorAssumptions)
    analyze!(alg.ws)
    return alg.ws
end

LinearSolve.needs_concrete_A(::CadnipLU) = true

function SciMLBase.solve!(cache::LinearSolve.LinearCache, alg::CadnipLU; kwargs...)
    ws = cache.cacheval
    if cache.isfresh                      # a new Jacobian was written into cache.A by fast_jacobian!: it is already on the device
        factor!(ws)
        cache.isfresh = false
    end
    rhs = cache.b isa Vector{Float64} ? cache.b : Vector{Float64}(cache.b)
    x = cache.u isa Vector{Float64} ? cache.u : Vector{Float64}(undef, length(rhs))
    solve!(x, ws, rhs)
    x === cache.u || copyto!(cache.u, x)
    return SciMLBase.build_linear_solution(alg, cache.u, nothing, cache)
end

end # module
