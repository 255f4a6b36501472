"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the Cadnip.jl transient hot path
(src/mna: fast_rebuild!/fast_residual!/fast_jacobian! + PCNR Newton + KLU solve).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package.  The product (cadnip.jl_amd/) never does.

Pinning status: the reference is Julia-only and Julia is absent from this image,
so the oracle cannot be run against the reference itself.  It is pinned against
the closed-form / exact-entry fixtures the reference's own tests hold
(test/mna/core.jl, test/mna/precompile.jl, test/mna/pcnr.jl, test/transients.jl,
test/sweep.jl); see tests/test_oracle_golden.py.  Results none of those fixtures
cover (the sp_mos1 Verilog-A device at full DFF scale) are "parity unpinned"
against the reference and are pinned only between this oracle's two independent
restatements (literal-dual Python vs. C++ port).
"""
