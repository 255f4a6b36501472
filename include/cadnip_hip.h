/*
 * cadnip_hip.h -- C ABI of libcadnip_hip.so, the MI355X (gfx950) replacement for the
 * CPU hot path of Cadnip.jl's transient analysis (reference: /root/reference/src/mna).
 *
 * Every entry point names the reference interface it replaces (file:line relative to
 * /root/reference).  The reference-side binding (Julia `ccall`) is shown in
 * INTEGRATION.md and shipped as source in cadnip.jl_amd/julia/CadnipHIP.jl.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; all indices are 0-based int32;
 *     all values are IEEE fp64.
 *   - every function returns a CadnipStatus (0 = ok); nothing throws across the ABI.
 *     The Julia shim maps CADNIP_SINGULAR -> LinearAlgebra.SingularException and
 *     CADNIP_NONFINITE -> DomainError so `_dc_solve_with_fallbacks`
 *     (src/mna/solve.jl:887-897) keeps working.
 *   - a handle owns all device memory; the caller owns every host array it passes.
 *     `*_host` arguments are host pointers (copied over PCIe inside the call);
 *     `cadnip_dev_ptr` exposes the handle's device buffers for zero-copy callers.
 *   - one handle = one (structure, GPU, HIP stream); handles are independent.
 *   - `n_instances` > 1 batches sweep points / Monte-Carlo instances that share one
 *     structure (CircuitSweep, src/sweeps.jl:387-424); all per-instance arrays are laid
 *     out instance-major: x[inst * n + i].
 */
#ifndef CADNIP_HIP_H
#define CADNIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  CADNIP_OK = 0,
  CADNIP_BADARG = 1,
  CADNIP_SINGULAR = 2,   /* zero / non-finite pivot in some instance (KLU: SingularException) */
  CADNIP_NONFINITE = 3,  /* non-finite stamp value (reference: DomainError from device math) */
  CADNIP_HIPERROR = 4,
  CADNIP_NOTREADY = 5,   /* e.g. factor before analyze */
  CADNIP_NOCONV = 6      /* driver: some instance did not converge / step size underflow */
} CadnipStatus;

/* Device types: one stamping kernel per type (src/mna/devices.jl `stamp!` methods and the
 * Verilog-A stamp pattern src/vasim.jl:3319-3521). */
typedef enum {
  CADNIP_DEV_RESISTOR = 0,   /* devices.jl:498-510   nodes p,n        par: g=1/r                */
  CADNIP_DEV_CAPACITOR = 1,  /* devices.jl:531-534   nodes p,n        par: c                    */
  CADNIP_DEV_INDUCTOR = 2,   /* devices.jl:569-586   nodes p,n,I      par: l                    */
  CADNIP_DEV_VSOURCE = 3,    /* devices.jl:619-663   nodes p,n,I      par: dc,scale  ipar: wave */
  CADNIP_DEV_ISOURCE = 4,    /* devices.jl:698-737   nodes p,n        par: dc,scale  ipar: wave */
  CADNIP_DEV_VCVS = 5,       /* devices.jl:760-775   nodes op,on,ip,in,I          par: gain     */
  CADNIP_DEV_VCCS = 6,       /* devices.jl:797-808   nodes op,on,ip,in            par: gm       */
  CADNIP_DEV_CCVS = 7,       /* devices.jl:824-849   nodes op,on,ip,in,Iin,Iout   par: rm       */
  CADNIP_DEV_CCCS = 8,       /* devices.jl:865-881   nodes op,on,ip,in,Iin        par: gain     */
  CADNIP_DEV_DIODE = 9,      /* devices.jl:1370-1428 nodes p,n,lim    par: Is,nVt,vcrit  ipar: limit flag */
  CADNIP_DEV_DIODECAP = 10,  /* devices.jl:1558-1602 nodes p,n        par: Is,nVt,Cj0,Vj,m      */
  CADNIP_DEV_SIMPLEMOS = 11, /* devices.jl:1667-1749 nodes d,g,s      par: Vth,K,lambda,Cgd,Cgs */
  CADNIP_DEV_MOS1 = 12,      /* models/VADistillerModels.jl/va/mos1.va via vasim.jl:3319-3521;
                                nodes d,g,s,b,d_int,s_int,lim[4],q[4]; par: CADNIP_MOS1_NPAR derived
                                (setup+temp hoisted, mos1.va:695-897) values; ipar: flags       */
  CADNIP_DEV_BVSOURCE = 13,  /* devices.jl:1079-1102 BehavioralVoltageSource  nodes p,n,I   par: scale   ipar: program off,len */
  CADNIP_DEV_BISOURCE = 14,  /* devices.jl:1118-1131 BehavioralCurrentSource  nodes p,n     par: scale   ipar: program off,len */
  CADNIP_DEV_VA = 15,        /* generated Verilog-A module (src/vasim.jl:2993-3985 generate_mna_stamp_method_nterm): nodes = ports, internal
                              * nodes, one charge unknown per branch (-1 = none); par = module parameters, temperature [K], mfactor,
                              * gmin; ipar: model id (position in the list the library was generated from), voltage-dependent-charge
                              * bit mask.  One block per module; slot layout: cadnip.jl_amd/va/frontend.py */
  CADNIP_DEV_NTYPES = 16
} CadnipDeviceType;

#define CADNIP_MOS1_NPAR 36

typedef enum { CADNIP_WAVE_DC = 0, CADNIP_WAVE_PWL = 1, CADNIP_WAVE_PULSE = 2, CADNIP_WAVE_SIN = 3 } CadnipWaveKind;

/* Behavioural sources: `value_fn(get_voltage)` of the reference is a Julia closure; across the C ABI it is a
 * postfix program of doubles stored in wave_data[off .. off+len): opcode [operands].  The value is evaluated at the
 * current iterate and stamped as a fixed source (no Jacobian entries), exactly as devices.jl:1079-1131 does. */
typedef enum CadnipBsrcOp {
  CADNIP_BOP_CONST = 0,   /* + value                     */
  CADNIP_BOP_V = 1,       /* + unknown index p, n (-1 = ground): pushes u[p] - u[n] */
  CADNIP_BOP_TIME = 2,
  CADNIP_BOP_ADD = 10, CADNIP_BOP_SUB = 11, CADNIP_BOP_MUL = 12, CADNIP_BOP_DIV = 13, CADNIP_BOP_POW = 14,
  CADNIP_BOP_MIN = 15, CADNIP_BOP_MAX = 16,
  CADNIP_BOP_NEG = 20, CADNIP_BOP_EXP = 21, CADNIP_BOP_LOG = 22, CADNIP_BOP_SQRT = 23, CADNIP_BOP_ABS = 24,
  CADNIP_BOP_TANH = 25, CADNIP_BOP_SIN = 26, CADNIP_BOP_COS = 27
} CadnipBsrcOp;
#define CADNIP_BSRC_MAX_STACK 16

/* One block per device type.  Node / ipar arrays are shared by all instances (structure);
 * parameters are per instance and set with cadnip_set_params.
 *   nodes[k * count + d]  : unknown index (0-based) read/written by device d's k-th node
 *                           slot, -1 = ground
 *   ipar[k * count + d]   : integer parameters (wave kind, offset/length into wave_data, flags)
 *   slot ids              : device d's k-th G stamp owns slot  g_base + k*count + d  of the
 *                           per-instance slot buffer (likewise C, b); limit_w index
 *                           lim_base + k*count + d is written by limiting devices. */
typedef struct {
  int32_t type;      /* CadnipDeviceType */
  int32_t count;
  int32_t n_nodes;   const int32_t* nodes;
  int32_t n_ipar;    const int32_t* ipar;
  int32_t n_par;     /* doubles per device per instance */
  int32_t g_base, c_base, b_base; /* first slot of this block in the G / C / b slot ranges */
  int32_t n_g, n_c, n_b;          /* slots per device */
} CadnipDeviceBlock;

/* Fixed structure of one circuit == the reference's CompiledStructure
 * (src/mna/precompile.jl:75-124), in CSR and with slot->nz gather lists instead of the
 * positional COO->nz maps (value_only.jl:395-478). */
typedef struct {
  int32_t n, n_nodes, n_currents, n_charges, n_limits;
  /* unified G u C pattern (precompile.jl:413-421), CSR, column indices sorted */
  int32_t nnz;
  const int32_t* rowptr;    /* [n+1] */
  const int32_t* colidx;    /* [nnz] */
  const int32_t* to_ref_nz; /* [nnz] position of CSR entry k in the reference's CSC nzval order */
  /* device blocks */
  int32_t n_blocks;
  const CadnipDeviceBlock* blocks;
  int32_t n_wave_data; const double* wave_data; /* PWL (t,y) tables etc. */
  /* slot -> nz gather lists, each list in the reference's COO (stamp) order so that the
   * floating-point summation order equals `nzval[map[pos]] += v` (value_only.jl:414-418) */
  int32_t ns_g, ns_c, ns_b;
  const int32_t* g_ptr; const int32_t* g_slots; /* [nnz+1], [..] */
  const int32_t* c_ptr; const int32_t* c_slots;
  const int32_t* b_ptr; const int32_t* b_slots; /* [n+1],  [..]  deferred b (precompile.jl:508-515) */
  const int32_t* diag_nz;   /* [n_nodes] CSR position of G[i,i] or -1 (precompile.jl:451-467) */
  const double*  limit_init;/* [n_limits] (precompile.jl:119-123) */
} CadnipStructure;

/* MNASpec scalars (src/mna/solve.jl:57-70).  temp is per instance (corner sweeps); the
 * temperature dependence of device parameters is hoisted into the per-instance parameter
 * blocks, so only mode / gmin / gshunt / srcFact reach the kernels. */
typedef struct {
  int32_t mode;      /* 0 = :dcop, 1 = :tran, 2 = :tranop */
  double gmin, gshunt, srcFact;
} CadnipSpec;

typedef struct CadnipHandle CadnipHandle;

/* ---- lifetime -------------------------------------------------------------------------
 * replaces compile_structure + create_workspace (precompile.jl:312-443, 193-222) */
int cadnip_create(const CadnipStructure* s, int32_t n_instances, int32_t device, CadnipHandle** out);
void cadnip_destroy(CadnipHandle* h);
int cadnip_set_params(CadnipHandle* h, int32_t block, const double* par_host /* [B][n_par][count] */);
int cadnip_set_spec(CadnipHandle* h, const CadnipSpec* spec);
int cadnip_set_initjct(CadnipHandle* h, int32_t on);            /* DirectStampContext.initjct, value_only.jl:93-95 */

/* ---- the three hot-path callbacks ------------------------------------------------------
 * cadnip_rebuild   == fast_rebuild!(ws, u, t)                  precompile.jl:493-537
 * cadnip_residual  == fast_residual!(resid, du, u, ws, t)      precompile.jl:546-557
 * cadnip_jacobian  == fast_jacobian!(J, du, u, ws, gamma, t)   precompile.jl:568-585
 * Unlike the reference, residual and jacobian do NOT restamp: they reuse the stamps of the
 * last cadnip_rebuild (the reference restamps the whole circuit in each, precompile.jl:548,572).
 * u_host/du_host: [B][n]; t_host: [B] (instances may sit at different times); gamma_host: [B]. */
int cadnip_rebuild(CadnipHandle* h, const double* u_host, const double* t_host);
int cadnip_residual(CadnipHandle* h, const double* du_host, const double* u_host, double* resid_host);
int cadnip_jacobian(CadnipHandle* h, const double* gamma_host, double* J_ref_nz_host /* may be NULL */);
/* fast_jacobian!(J::Matrix, ...) of a structure compiled with dense = true (precompile.jl:588-603; the form the reference's canonical boundary
 * test drives, test/mna/audio_integration.jl:505-520): J = G + gamma C as a dense matrix per instance, [B][n * n] in Julia's column-major
 * layout (J[i, j] at j * n + i), structural zeros written as 0.  Meant for the small systems that test uses (n^2 words per instance). */
int cadnip_jacobian_dense(CadnipHandle* h, const double* gamma_host, double* J_dense_host);
/* ODE-form callbacks (mass matrix C constant; used by the reference with FBDF / QNDF / Rodas):
 *   cadnip_ode_rhs      == rhs!(du, u, p, t):  restamp at (u, t), du = b - G u      src/mna/solve.jl:2241-2248
 *   cadnip_ode_jacobian == jac!(J, u, p, t):   restamp at (u, t), J = -G            src/mna/solve.jl:2251-2276
 * J comes back in the reference's CSC nzval order (jac_prototype = -cs.G, solve.jl:2285); the mass matrix is
 * cadnip_get_GCb's C.  u_host NULL = reuse the stamps of the last rebuild. */
int cadnip_ode_rhs(CadnipHandle* h, const double* u_host, const double* t_host, double* du_host);
int cadnip_ode_jacobian(CadnipHandle* h, const double* u_host, const double* t_host, double* J_ref_nz_host);
/* operating-point read-out: the per-device contributions of one restamp at the handle's current u / t,
 * [B][ns_g + ns_c + ns_b] in CadnipStructure's slot layout (G slots, then C, then b; slot (k, dev) of a block at
 * base + k * count + dev).  Terminal currents and small-signal variables are sums over them (context.jl:1200-1342). */
int cadnip_get_contributions(CadnipHandle* h, double* slots_host);
/* parity / debug read-back in the reference's CSC nzval order (any pointer may be NULL) */
int cadnip_get_GCb(CadnipHandle* h, double* G_ref_nz, double* C_ref_nz, double* b, double* limit_w);

/* ---- sparse LU == KLU at src/sweeps.jl:600 (IDA linear_solver=:KLU) and
 * src/mna/solve.jl:612-613,667-670 (LinearSolve KLUFactorization, symbolic reused) ----------
 * cadnip_analyze: host symbolic phase, once: threshold-Markowitz pivot order chosen on the
 *   current J of instance `sample_instance`, symbolic fill, level schedules.
 * cadnip_factor : numeric refactor of J for every instance on the GPU with the static pivot
 *   order; returns CADNIP_SINGULAR if any instance hit a bad pivot (flags via cadnip_get_flags).
 * cadnip_solve  : x = J^-1 rhs per instance. */
int cadnip_analyze(CadnipHandle* h, int32_t sample_instance);
/* same, on caller-supplied sample values (CSR order of CadnipStructure.colidx), e.g. the element-wise
 * max |J| over several operating points so that the static pivot order suits all of them */
int cadnip_analyze_values(CadnipHandle* h, const double* J_csr_host);
int cadnip_factor(CadnipHandle* h);
int cadnip_solve(CadnipHandle* h, const double* rhs_host, double* x_host);

/* One Newton iteration of the DAE form in one call -- what a host integrator that keeps the nonlinear loop to itself (IDA behind the Julia
 * shim: residual!, then jacobian! + klu_refactor when it decides on a set-up, then klu_solve; src/mna/precompile.jl:546-585,
 * src/mna/solve.jl:2138-2160, src/sweeps.jl:600) otherwise does with five entry points and five synchronisations:
 *   resid = C du + G u - b at (u, t)          [cadnip_rebuild, cadnip_residual]
 *   refresh != 0: J = G + gamma C, refactored  [cadnip_jacobian, cadnip_factor]; 0: the factors of the last refreshing call are used
 *   delta = J^-1 resid                         [cadnip_solve]
 * with the same kernels in the same order (the results are the same doubles as the five calls'), one staged upload, one staged download,
 * one synchronisation; the launch sequence is an instantiated HIP graph, replayed while the handle's configuration stays as it is.
 * u, du, delta, resid: [B][n]; gamma, t, resid_norm: [B].  t_host NULL = times unchanged; gamma_host may be NULL when refresh == 0;
 * resid_norm_host (||resid||_2 per instance) and resid_host are optional.  CADNIP_NONFINITE / CADNIP_SINGULAR as the separate calls. */
int cadnip_newton_step(CadnipHandle* h, const double* u_host, const double* du_host, const double* gamma_host, const double* t_host, int32_t refresh,
                       double* delta_host, double* resid_norm_host, double* resid_host);
/* The same iteration in ONE kernel (the fused team kernel, csrc/fused_team_kernel.hpp): stamping, residual, refactorisation or kept factors,
 * solve.  Same arguments; the results agree with cadnip_newton_step to rounding (the devices add into an LDS-resident work array in
 * another order), not bit for bit.  CADNIP_BADARG for circuits the team kernel does not run (device types outside linear elements /
 * sources / plain sp_mos1, tables beyond LDS, external generated models, homotopy in force): use cadnip_newton_step there.  The stamps use
 * the transient device mode (a DAE residual is a transient's). */
int cadnip_newton_step_fused(CadnipHandle* h, const double* u_host, const double* du_host, const double* gamma_host, const double* t_host, int32_t refresh,
                             double* delta_host, double* resid_norm_host, double* resid_host);
int cadnip_lu_stats(CadnipHandle* h, int32_t* nnz_lu, int32_t* n_terms, int32_t* n_levels, int32_t* n_fwd_levels, int32_t* n_bwd_levels);

/* ---- host drivers (Newton loop + step controller; stand-in for IDA / _dc_pcnr_newton) ----
 * The loops run on the host and launch the kernels above on the handle's stream; the
 * per-instance convergence / step decisions are evaluated on the device (one lane-group
 * per instance) so that B desynchronised instances need no per-iteration PCIe traffic. */
typedef struct {
  double abstol;          /* ||G u - b||_2 < abstol      solve.jl:640 (1e-10 dc!, 1e-9 CedarTranOp) */
  int32_t maxiters;       /* solve.jl:600 */
  int32_t use_pcnr;       /* PCNR corrector u[lim] = limit_w  solve.jl:682-688 */
  int32_t cold_start;     /* seed limit vars + initjct       solve.jl:615-625 */
  int32_t use_stepping;   /* gshunt / source stepping fallbacks solve.jl:909-925 */
  int32_t fused;          /* non-zero: the first attempt (PCNR / Newton at the handle's spec) runs in the fused kernel
                             (csrc/fused2.hip, DC mode); the fallback homotopies always use the per-op kernels */
  const int32_t* participate; /* [B] or NULL (= everyone): instances with 0 sit the run out -- their u is left as it is and they
                             report converged = 0.  dc!(cs) continuation solves a sweep in stages (sweeps.jl:511-532): the
                             points of a stage start from converged neighbours of the earlier stages */
} CadnipDCOpts;

typedef struct {
  double t0, t1;
  double reltol;
  const double* abstol;     /* [n] per-unknown absolute tolerance (state_abstol, build.jl:276-283) */
  const double* err_mask;   /* [n] 1 = unknown takes part in the local-error test, 0 = excluded; NULL = all.
                               The driver's default excludes algebraic unknowns (V-source currents jump at
                               source breakpoints), i.e. it tests the differential unknowns -- the columns of C
                               (cf. detect_differential_vars, solve.jl:2041-2058, and IDA's suppressalg) */
  double h0;                /* initial step (<=0: automatic) */
  double hmin, hmax;        /* hmax <= 0: (t1-t0)/50 */
  int32_t max_newton;       /* IDA max_nonlinear_iters = 10, sweeps.jl:600 */
  int32_t max_order;        /* 1 = backward Euler only, 2 = variable-step BDF2 (default), 3 = variable-step BDF3 where four accepted points exist
                             * (the reference's IDA goes to 5, src/sweeps.jl:600; orders restart at 1 on every source breakpoint) */
  int32_t use_pcnr;         /* apply the PCNR corrector inside transient Newton */
  double newton_tol;        /* weighted-RMS norm of the Newton update that counts as converged */
  int32_t n_break; const double* breaks; /* sorted tstops from source breakpoints (solve.jl:1847-1960) */
  int32_t n_save;  const double* save_t; /* sorted output times (saveat) */
  int32_t n_obs;   const int32_t* obs;   /* unknown indices to record; n_obs = 0 -> all n */
  int64_t max_iterations;   /* safety bound on lock-step Newton launches */
  int32_t fused;            /* non-zero = fused per-instance Newton kernel, 0 = one kernel per op.  The fused kernel is chosen by batch size: one wave per
                               instance for sweeps (csrc/fused2_kernel.hpp), a team of four waves per instance for batches of at most one instance per
                               compute unit (csrc/fused_team_kernel.hpp: a single transient's latency).  A circuit the fused kernel cannot run -- too
                               large for LDS, an external generated model, or newton_mode 1 with device types outside its lean set (diodes, behavioural
                               sources, sp_mos1 with series resistances, built-in Verilog-A modules) -- runs on the per-op kernels instead */
  int32_t newton_mode;      /* 0 = full Newton, converged when the weighted update norm < newton_tol (every round restamps and refactors);
                               1 = the nonlinear iteration as IDA runs it (the reference's integrator, src/sweeps.jl:600): refactor on a setup
                               only (first round, a0 outside [0.6, 1/0.6] of its last setup value, 20 steps, failure on a stale Jacobian), kept factors in between, rate-based
                               convergence test ss ||delta|| <= 0.33.  The fused kernel keeps the factors; the per-op kernels refactor every round (their
                               factors live in LDS for one launch) and take the convergence test only */
  int32_t step_rule;        /* step size after the error test.  0 = the classical controller: factor 0.9 err^(-1/(k+1)) in [0.2, 2] after an accepted step, in
                               [0.1, 0.9] after a rejected one.  1 = IDA's rule (ida.c: IDACompleteStep / IDASetEta, the reference's integrator, src/sweeps.jl:600):
                               eta = 1 / ((2 err)^(1/(k+1)) + 1e-4); the step DOUBLES when eta >= 2, shrinks by max(0.5, min(0.9, eta)) when eta <= 1 and
                               otherwise STAYS (fewer rejected steps, and a constant step keeps the kept factors of newton_mode 1 valid); after a failed
                               error test the factor is 0.9 eta in [0.25, 0.9] */
} CadnipTranOpts;

typedef struct {
  int64_t newton_iters;     /* == sol.stats.nnonliniter summed over instances */
  int64_t steps_accepted, steps_rejected, newton_failures;
  int64_t launches;         /* lock-step Newton launches */
  int32_t n_failed;         /* instances that did not reach t1 / did not converge */
  double  wall_seconds;     /* host wall clock of the driver loop (device-synchronised) */
} CadnipRunStats;

/* u_host [B][n]: in = initial guess (zeros = cold start), out = solution; converged_host [B] */
/* What the fallback chain of the last cadnip_dc_run did: one entry per (instance, Newton run) in execution order.
 * stage 0 = PCNR (or plain Newton when use_pcnr is off), 1 = plain Newton, 2 = gshunt stepping (value = the rung's gshunt),
 * 3 = source stepping (value = srcFact); ok = the run converged; iters = its Newton solves.  Arrays of cadnip_dc_log_size
 * entries; any pointer may be NULL.  (The reference reports the same through @debug lines, solve.jl:887-927.) */
int32_t cadnip_dc_log_size(CadnipHandle* h);
int cadnip_dc_log_get(CadnipHandle* h, int32_t* inst, int32_t* stage, double* value, int32_t* ok, int64_t* iters);
int cadnip_dc_run(CadnipHandle* h, const CadnipDCOpts* o, double* u_host, int32_t* converged_host, CadnipRunStats* st);
/* starts from the handle's current state u (e.g. left by cadnip_dc_run in :tranop mode);
 * out_host [B][n_save][n_obs]; per_inst_host [B][4] = {newton_iters, accepted, rejected, status} */
int cadnip_tran_run(CadnipHandle* h, const CadnipTranOpts* o, double* out_host, int64_t* per_inst_host, CadnipRunStats* st);

/* per-instance integrator state after / during a run: time reached, current step size, order */
int cadnip_tran_state(CadnipHandle* h, double* t_host, double* h_host, int32_t* order_host);

/* ---- misc -------------------------------------------------------------------------------- */
typedef enum { CADNIP_BUF_U = 0, CADNIP_BUF_G = 1, CADNIP_BUF_C = 2, CADNIP_BUF_B = 3, CADNIP_BUF_J = 4,
               CADNIP_BUF_RESID = 5, CADNIP_BUF_SLOTS = 6, CADNIP_BUF_LU = 7, CADNIP_BUF_FLAGS = 8 } CadnipBuffer;
void* cadnip_dev_ptr(CadnipHandle* h, int32_t which);  /* device pointer of a handle-owned buffer */
void* cadnip_stream(CadnipHandle* h);                  /* hipStream_t the handle launches on */
int cadnip_set_u(CadnipHandle* h, const double* u_host);
int cadnip_get_u(CadnipHandle* h, double* u_host);
int cadnip_get_flags(CadnipHandle* h, int32_t* flags_host /* [B] */);
int cadnip_sync(CadnipHandle* h);
/* timing of the kernels launched since the last reset, measured with hipEvents on the handle's
 * stream: names[i] (static strings), ms[i] total, calls[i]; returns number of entries */
int cadnip_profile_enable(CadnipHandle* h, int32_t on);
int cadnip_profile_read(CadnipHandle* h, int32_t max_entries, const char** names, double* ms, int64_t* calls);
const char* cadnip_version(void);
/* PMC calibration aid (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE must be calibrated on a known
 * byte count in the kernel's own access width): streams n_doubles fp64 values src -> dst with 8 B per lane,
 * `reps` times, through a kernel named k_calib_copy_f64.  Known traffic: 8*n_doubles read + 8*n_doubles written per rep. */
/* measurement aid: `reps` back-to-back launches of the stamping kernel of device block `block` (< 0: all blocks = one restamp)
 * at the handle's current u / t, timed with HIP events on the handle's stream */
int cadnip_debug_stamp_time(CadnipHandle* h, int32_t block, int32_t reps, double* ms_total);
/* diagnostic: one Newton iteration of the team kernel (STEP mode) repeated `reps` times at the resident state, phases left out by `skip`
 * (1 stamping, 2 adding the waves' private sums, 4 linear-solve steps, 8 dense core); tools/team_phases.py */
int cadnip_debug_step_time(CadnipHandle* h, int32_t refresh, int32_t reps, int32_t skip, double* ms_total);
int cadnip_debug_copy(CadnipHandle* h, int64_t n_doubles, int32_t reps);

/* ---- host-only helpers (no GPU needed): run the symbolic phase on any CSR matrix and read the
 * resulting program back; used by the CPU test-suite to validate the LU program --------------- */
typedef struct CadnipHostLU CadnipHostLU;
typedef enum { CADNIP_LU_RPERM = 0, CADNIP_LU_CPERM, CADNIP_LU_ROWPTR, CADNIP_LU_COL, CADNIP_LU_DIAG, CADNIP_LU_LOAD_SRC,
               CADNIP_LU_LOAD_DST, CADNIP_LU_ENT_POS, CADNIP_LU_ENT_DIAG, CADNIP_LU_ENT_PTR, CADNIP_LU_TERM_A, CADNIP_LU_TERM_B,
               CADNIP_LU_LEV_PTR, CADNIP_LU_FWD_ROWS, CADNIP_LU_FWD_LEV_PTR, CADNIP_LU_BWD_ROWS, CADNIP_LU_BWD_LEV_PTR,
               CADNIP_LU_NARRAYS } CadnipLUArray;
/* sample != 0: `vals` is a composite magnitude sample (as cadnip_analyze_values takes), not one state's Jacobian: a
 * numerically singular sample is re-analysed on magnitudes instead of being reported (csrc/symbolic.cpp) */
int cadnip_host_lu_analyze(int32_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, double pivot_tol, int32_t sample, CadnipHostLU** out);
/* ... with the leaf-first pivot order a handle uses (csrc/symbolic.cpp): unknowns [q_begin, lim_begin) are charge states, [lim_begin, n)
 * limit variables (the reference's layout [V | I | q | lim], src/mna/context.jl); unit_ok[i] != 0: the diagonal of unknown i is one
 * constant stamp.  q_begin < 0: plain Markowitz search (as cadnip_host_lu_analyze). */
int cadnip_host_lu_analyze_leaves(int32_t n, const int32_t* rowptr, const int32_t* colidx, const double* vals, double pivot_tol, int32_t sample,
                                  int32_t q_begin, int32_t lim_begin, const uint8_t* unit_ok, CadnipHostLU** out);
int32_t cadnip_host_lu_size(const CadnipHostLU* lu, int32_t which);
int32_t cadnip_host_lu_blocks(const CadnipHostLU* lu);   /* diagonal blocks of the block triangular form when KLU's ordering was used (csrc/klu_order.cpp), 0 = Markowitz search */
int cadnip_host_lu_get(const CadnipHostLU* lu, int32_t which, int32_t* dst);
void cadnip_host_lu_free(CadnipHostLU* lu);

/* The fused kernel's linear-solve program for core size nc (0, 8, 12 or 16; csrc/f2_program.cpp), built from a host LU:
 * arrays POSW (int32: LU pattern position -> work-array offset), LANES / PASSES (uint64 descriptors), TERMS (uint32) and
 * META (int32: nc, lu_words, dn0, n_pre, n_post).  cadnip_host_f2_get copies raw bytes; _size returns the element count. */
typedef struct CadnipHostF2 CadnipHostF2;
typedef enum { CADNIP_F2_POSW = 0, CADNIP_F2_LANES, CADNIP_F2_PASSES, CADNIP_F2_TERMS, CADNIP_F2_META, CADNIP_F2_NARRAYS } CadnipF2Array;
int cadnip_host_f2_build(const CadnipHostLU* lu, int32_t nc, CadnipHostF2** out);
int32_t cadnip_host_f2_size(const CadnipHostF2* prog, int32_t which);
int cadnip_host_f2_get(const CadnipHostF2* prog, int32_t which, void* dst);
int cadnip_host_f2_team_steps(const CadnipHostLU* lu, int32_t nc, int32_t nw, int32_t* out4);   /* steps of the team layout (pre, post, forward-only) and its descriptor words */
/* The same program as straight-line steps (csrc/f2_program.cpp): nw = 1 -- the one-wave layout of the fused sweep kernel (list-scheduled steps of 64
 * lanes, three terms per lane, 16-byte descriptors = two uint64 per lane), nw = 2 / 4 -- the team layouts (8-byte descriptors), nw = 12 / 14 -- the
 * three-term layout for teams of 2 / 4 waves (the per-op step LU).  out4 = steps of the
 * pre-core, post-core and forward-only lists and the number of uint64 words; dst (may be NULL) receives the words. */
int cadnip_host_f2_steps(const CadnipHostLU* lu, int32_t nc, int32_t nw, int32_t* out4, uint64_t* dst);
void cadnip_host_f2_free(CadnipHostF2* prog);

#ifdef __cplusplus
}
#endif
#endif /* CADNIP_HIP_H */
