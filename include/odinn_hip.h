/*
 * odinn_hip.h -- C ABI of libodinn_hip.so: the MI355X (gfx950) implementation of
 * ODINN.jl's SIA2D(+NN_theta) time-stepping hot path and its discrete adjoint.
 *
 * This is the drop-in boundary.  The reference has no FFI today; its plug-in
 * seams for this path are Julia dispatch points, each replaced here by one
 * entry point (paths relative to the ODINN.jl tree):
 *
 *   odinn_sia2d_dhdt        <- SIA2D_UDE!(dH,H,container,t) -> Huginn.SIA2D!
 *                              src/simulations/inversions/inversion_utils.jl:691-699
 *   odinn_sia2d_vjp_H       <- VJP_lambda_dSIAdH(::DiscreteVJP, lam, H, theta, sim, t)
 *                              src/inverse/SIA2D/VJPs.jl:2-5 -> adjoint.jl:31-151
 *   odinn_sia2d_vjp_theta   <- VJP_lambda_dSIAdtheta(::DiscreteVJP, ...)
 *                              src/inverse/SIA2D/VJPs.jl:30-33 -> adjoint.jl:178-255
 *   odinn_mb_vjp_H          <- VJP_lambda_dMBdH(::DiscreteVJP, ...)  VJPs.jl:107-151
 *   odinn_surface_V         <- Huginn.surface_V / V_from_H (called src/losses/Losses.jl:314,358)
 *   odinn_surface_V_vjp_H/_theta <- VJP_lambda_dsurface_VdH / dtheta (::DiscreteVJP)
 *                              src/inverse/SIA2D/VJPs.jl:61-69 -> adjoint.jl:268-413
 *   odinn_solve             <- _batch_iceflow_UDE / simulate_iceflow_UDE!
 *                              src/simulations/inversions/inversion_utils.jl:472-572
 *   odinn_loss              <- batch_loss_iceflow_transient  inversion_utils.jl:383-461
 *   odinn_loss_grad         <- SIA2D_grad_batch! (DiscreteAdjoint + DiscreteVJP)
 *                              src/inverse/SIA2D/gradient.jl:45-275, summed over the
 *                              glaciers of the batch as SIA2D_grad! does (:6-31)
 *
 * Conventions
 *   - every array is contiguous float64, Julia column-major: element [i,j] of an
 *     nx-by-ny field is at i + nx*j (i = x = Julia dim 1).  Dual-grid fields are
 *     (nx-1)-by-(ny-1).
 *   - pointers are HOST pointers unless the name ends in _dev.
 *   - every function returns 0 on success, non-zero on error;
 *     odinn_last_error() returns a thread-local message.  No exceptions cross.
 *   - an odinn_batch owns the device state of G glaciers on ONE device and one
 *     HIP stream.  A batch is not thread-safe; different batches are independent.
 *   - there is NO CPU fallback: if no gfx950 device is present every compute entry
 *     point fails with ODINN_ERR_NO_DEVICE.
 */
#ifndef ODINN_HIP_H
#define ODINN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODINN_OK 0
#define ODINN_ERR_ARG 1
#define ODINN_ERR_NO_DEVICE 2
#define ODINN_ERR_HIP 3
#define ODINN_ERR_STATE 4
#define ODINN_ERR_MAXITERS 5
#define ODINN_ERR_NONFINITE 6
#define ODINN_ERR_UNSUPPORTED 7
#define ODINN_ERR_DTMIN 8 /* an adaptive solve (forward or reverse) that is stuck: 256 attempts in a row without advancing its time
                             variable -- rejections, or accepted steps so small that t + dt == t (step size at the resolution of t).
                             It would spin until maxiters.  OrdinaryDiffEq gives up earlier, at the first dt <= dtmin = eps(t)
                             (ReturnCode.DtLessThanMin), also where the step size recovers after a few such attempts */

#define ODINN_MAX_LAYERS 8
#define ODINN_MAX_WIDTH 32

/* which quantity theta drives (src/laws/Laws.jl; targets src/models/target/) */
enum odinn_law_kind {
  ODINN_LAW_CONST_A = 0,      /* A given per glacier (scalar) or as a dual-grid field       */
  ODINN_LAW_NN_A_SCALAR = 1,  /* A = minA+(maxA-minA)*MLP(T), scalar T   Laws.jl:348-358    */
  ODINN_LAW_NN_A_GRIDDED = 2, /* same on a gridded T (dual grid), evaluated once per theta  */
  ODINN_LAW_NN_Y = 3,         /* Y = max*exp((y-1)/y), y=MLP(T,Hbar)     Laws.jl:258-265    */
  ODINN_LAW_NN_U = 4          /* U = post(MLP(Hbar,|gradS|)), D=Hbar*U   Laws.jl:114-123    */
};

enum odinn_act { ODINN_ACT_IDENTITY = 0, ODINN_ACT_SOFTPLUS = 1, ODINN_ACT_SIGMOID = 2,
                 ODINN_ACT_GELU = 3, ODINN_ACT_TANH = 4, ODINN_ACT_RELU = 5 };

enum odinn_post { ODINN_POST_NONE = 0,   /* identity                                          */
                  ODINN_POST_AFFINE = 1, /* lo+(hi-lo)*y     target_utils.jl:109-113          */
                  ODINN_POST_EXPMAX = 2, /* hi*exp((y-1)/y)  target_utils.jl:86-93            */
                  ODINN_POST_SCALE = 3   /* hi*y                                              */ };

/* Sleipnir.PhysicalParameters fields used on the path (test/params_construction.jl:24-34) */
typedef struct odinn_phys {
  double rho, g, eta0, n, p, q, C, minA, maxA;
} odinn_phys;

typedef struct odinn_glacier_desc {
  int32_t nx, ny;
  double dx, dy;
  odinn_phys phys;
  double A; /* creep coefficient for ODINN_LAW_CONST_A [Pa^-n yr^-1]                        */
  double T; /* scalar long-term air temperature input of LawA / LawY [deg C]                */
} odinn_glacier_desc;

/* Lux.Chain(Dense...) descriptor; theta is flattened layer by layer as
 * [vec(W_l) column-major (out x in), b_l]  (ML_utils.jl:31-36,54). */
typedef struct odinn_mlp_desc {
  int32_t n_layers;
  int32_t widths[ODINN_MAX_LAYERS + 1]; /* widths[0] = n_in (1 or 2), widths[n_layers] = 1 */
  int32_t acts[ODINN_MAX_LAYERS];
  int32_t has_prescale;                 /* x~ = (x-lo)/(hi-lo)-0.5  target_utils.jl:131-141 */
  double pre_lo[2], pre_hi[2];
  int32_t post_kind;
  double post_lo, post_hi;
} odinn_mlp_desc;

typedef struct odinn_solver_opts {
  double reltol;    /* params.solver.reltol (inversion_utils.jl:565)                        */
  double abstol;    /* OrdinaryDiffEq default 1e-6                                          */
  double dtmax;     /* <=0: unbounded                                                       */
  double dt0;       /* <=0: Hairer-Wanner automatic initial step                            */
  double fixed_dt;  /* >0: non-adaptive RDPK3Sp35 with this step (clipped at tstops)        */
  int64_t maxiters; /* params.solver.maxiters (:567); <=0 -> 1e6                            */
  int32_t scheme;   /* 0 auto, 1/2: kernel schedule of one RDPK3Sp35 step (same arithmetic either way):
                       1 five per-stage kernels (HBM-bound, 264 B/cell/step),
                       2 one temporally fused kernel (~24 B/cell/step, fp64-VALU-bound);
                       ODINN_SCHEDULE="scheme=1|2" overrides 0.
                       3 ODINN_SCHEME_EULER_CFL: explicit Euler with the CFL-limited step
                         dt = cfl * min(dx,dy)^2 / (4 max D) (the classical SIA stepping; max D by an
                         in-kernel wavefront reduction; 24 B per cell-step).  Not a reference scheme:
                         own definition, restated in oracle/ (solve_euler_cfl).                  */
  int32_t dense;    /* 0 (default): workgroups of the fused step whose whole halo region is ice-free
                       (u == 0) take an exact shortcut (bit-identical result); 1: always run the
                       five stages (what bench.py times for `value`)                              */
  double cfl;       /* scheme 3 only: CFL safety factor in (0, 1]; <= 0 -> 0.25                  */
} odinn_solver_opts;

typedef struct odinn_solve_stats {
  int64_t naccept, nreject, nrhs;
  double t_final, dt_last;
} odinn_solve_stats;

/* H-VJP stencil used by odinn_sia2d_vjp_H and inside both adjoints (VJPTypes.jl:29-50):
 * DiscreteVJP = exact transpose of the discretised RHS (adjoint.jl:31-151);
 * ContinuousVJP = discretisation of the continuous adjoint operator (adjoint.jl:442-553; all targets).
 * VJP_lambda_dSIA/dtheta_continuous (adjoint.jl:583-662) is the forward form of the same bilinear
 * expression as the discrete theta-VJP, so odinn_sia2d_vjp_theta serves both methods. */
#define ODINN_SCHEME_EULER_CFL 3
#define ODINN_VJP_DISCRETE 0
#define ODINN_VJP_CONTINUOUS 1

/* options of the reverse solve of the continuous adjoint; defaults = ContinuousAdjoint()
 * (src/inverse/AdjointTypes.jl:58-67): RDPK3Sp35, reltol = abstol = 1e-8, dtmax = 1/12,
 * linear interpolation of H in time, n_quadrature = 200.  Fields <= 0 take the default. */
typedef struct odinn_adjoint_opts {
  double reltol;
  double abstol;
  double dtmax;
  int32_t n_quadrature;
  int32_t reserved;
  int64_t maxiters; /* params.solver.maxiters (gradient.jl:468) */
} odinn_adjoint_opts;

/* Kernel schedule of a batch: which of the library's equivalent kernel forms run (same arithmetic, results agree to rounding --
 * see DESIGN.md "Different compilations of the same per-cell expression sequence").  Every field: -1 = automatic (the library's
 * measured rule, the default), otherwise the forced choice.  The environment variable ODINN_SCHEDULE="field=value,field=value,..."
 * (field names as below, plus scheme=1|2), if set, overrides the named fields of every batch of the process (measurement / A-B aid;
 * read at every call).  `scheme` and `dense` of odinn_solver_opts stay per solve.
 *
 * ALL environment variables the library reads (14):
 *   ODINN_SCHEDULE            the schedule override above
 *   ODINN_INTERP_SELECT=0     `:Linear` law gradients by the radix-sorted contractions instead of the sort-free ones (Y and U law)
 *   ODINN_INTERP_ACTIVE=0     ... over all dual nodes instead of the list of nodes that ever carry ice
 *   ODINN_UTAB_LEVEL=0..3     U law: force the table resolution (16 x 8 ... 128 x 64 patches) instead of the coarsest that passes
 *   ODINN_UT_LDS=0            U law: table patches from global memory even where a kernel stages them in LDS (test aid)
 *   ODINN_LAW_TABLE_HMAX=x    first range of the law tables (test aid: small enough to overflow and be widened)
 *   ODINN_LAW_TABLE_VERBOSE   one line per table build on stderr (size, measured deviation, table / network)
 *   ODINN_DTMIN=0             no ODINN_ERR_DTMIN exit of a stuck adaptive solve (diagnostics)
 *   ODINN_TRACE_STEPS=n       the first n attempts (t, dt, error estimate, step factor) of one glacier's solves on stderr
 *   ODINN_TRACE_GLACIER=g     ... of glacier g instead of glacier 0
 *   ODINN_PROFILE_HOST        host-side phase timings of odinn_solve / the gradients on stderr
 *   ODINN_TIMED_ADJ_SKIP      odinn_time_kernel(ODINN_TIMED_ADJ_FUSED_STEP) times the kernel with its ice-free shortcut
 *   ODINN_RCCL_LIB=path       librccl.so to dlopen for odinn_comm_* (default: the loader's search path, then torch's copy)
 *   ODINN_REQUEST_DEV_KERNARG=1  sets HIP_FORCE_DEV_KERNARG=1 at load time (kernel arguments in device memory) */
typedef struct odinn_schedule {
  int32_t step_sc;         /* step_sc: 1 = self-controlled step loop (no controller / post-step launches), 0 = off          */
  int32_t fused_tiles;     /* fused_tiles (s|l|t|u): fused step kernel -- 1 54x8 latency tiles, 2 54x40 tiles, 3 strip kernel
                              with 7 rows per thread, 4 strip kernel with 8 rows per thread                                        */
  int32_t dhdt_strip;      /* dhdt_strip: 0 = keep the 64x16-tile RHS / CFL-Euler kernels instead of the strip layout        */
  int32_t vjph_strip;      /* vjph_strip: H-VJP in the strip layout (1) or on 64x16 LDS tiles (0)                            */
  int32_t vjpth_strip;     /* vjpth_strip: the same for the theta-VJP reduction                                              */
  int32_t snap_on_load;    /* snap_on_load: 0 = post-step launch per step instead of the snapshot-on-load two-launch loop    */
  int32_t interp_streams;  /* interp_streams: side streams of the per-glacier `:Linear` interpolation sequences (1 ... 8)     */
  int32_t interp_batch;    /* interp_batch: 0 = one interpolation sequence per glacier instead of one per call (Y law)      */
  int32_t lawgrad_wave;    /* lawgrad_wave: 0 = per-thread accumulators for the gridded law's theta-gradient                */
  int32_t vq_onepass;      /* vq_onepass: 0 = interpolate / scale / pull-back / reduce sequence at the quadrature nodes of a
                              velocity loss instead of the one-pass node kernel                                                    */
  int32_t adj_fused;       /* adj_fused: 0 = five k_adj_stage launches per reverse step instead of the fused reverse step   */
  int32_t adj_skip;        /* adj_skip: 0 = no ice-free shortcut in the fused reverse step                                   */
  int32_t adj_segs;        /* adj_segs: 0 = read the two snapshots instead of the interleaved {H_j, H_j+1 - H_j} pairs       */
  int32_t adj_rows;        /* adj_rows: 4 | 7 | 8 rows per thread of the fused reverse step (8: gridded A only, else ignored) */
  int32_t adj_theta_fused; /* adj_theta_fused: 0 = theta-VJP of a quadrature node in launches of its own                     */
  int32_t law_table;       /* law_table: 0 = the stencil kernels of a batch with the Y law (ODINN_LAW_NN_Y: inputs = the
                              glacier's scalar temperature and Hbar) or the U law (ODINN_LAW_NN_U: inputs = Hbar and |grad S|)
                              evaluate the network at every dual node and stage.  Default (-1 / 1): inside the forward solve and
                              both adjoints they read the law from a table -- Y(Hbar) per glacier in 1024 quintics, U(Hbar, |grad S|)
                              for the batch in bi-quintic patches, the coarsest of 16 x 8 ... 128 x 64 that passes the check (a
                              smooth law: 37 KB that stay in the L1 instead of 2.4 MB) -- rebuilt from the network whenever theta changes, used
                              only while its measured deviation from the network is < 1e-12 relative (else the network); a solve
                              that leaves the table's range is repeated with a wider one.  The seam calls always evaluate the
                              network.  What the 1e-12 bounds are the VALUES of the law: the finite-difference terms of the adjoints
                              ((Y(Hbar + 1e-4) - Y(Hbar)) / 1e-4, the U law's central differences) are then differences of the
                              interpolant, whose error is amplified by 1 / step -- 1e-8 relative at the bound, the accuracy those
                              differences have in fp64 anyway; gradients with and without the table agree to 1e-8 ... 5e-8
                              (tests/test_gpu_law_table*.py)                                                                       */
  int32_t interp_async;    /* interp_async: 0 = the Y law's `:Linear` contraction of a stop (sort, knots, interval sums, knot
                              backprop) on the batch's own stream; n = 1 ... 4: overlapped with the following reverse steps of both
                              adjoints on n lane streams (default: 3 in the DiscreteAdjoint, 1 or 4 in the ContinuousAdjoint; results
                              bit-identical: every contribution has its own slot, the slots are added in the order of the stops)  */
  int32_t adj_sc;          /* adj_sc: 1 = self-controlled reverse step of the ContinuousAdjoint (the fused reverse step decides
                              the previous attempt and does the post-step of a stop itself: one launch per reverse step), 0 = the
                              three-launch loop (fused step, controller, post-step); same decisions, same arithmetic             */
  int32_t adj_ut_fused;    /* adj_ut_fused: reverse step of the ContinuousAdjoint for the U law (target :D) through its table: 0 = five
                              k_adj_stage launches, 1 = the strip kernel's UT form (measured 3 x slower, kept as a cross-check), 2 = the
                              fused step on LDS tiles (k_adj_fused_lds, sia2d_adj_lds.hpp); -1 = the library's measured rule             */
  int32_t reserved[1];     /* zero                                                                                                 */
} odinn_schedule;

typedef struct odinn_batch odinn_batch;

const char* odinn_last_error(void);
int odinn_device_count(int* n);
/* name of device `dev` (e.g. "gfx950...") into buf */
int odinn_device_name(int dev, char* buf, int buflen);

/* ---- lifetime / inputs (H2D once; state stays resident in HBM) ---------------------- */
int odinn_batch_create(int device, int n_glaciers, const odinn_glacier_desc* descs, odinn_batch** out);
int odinn_batch_destroy(odinn_batch* b);
int odinn_batch_sync(odinn_batch* b);
/* sc == NULL: everything automatic.  odinn_get_schedule returns what is in effect (environment overrides applied). */
int odinn_set_schedule(odinn_batch* b, const odinn_schedule* sc);
int odinn_get_schedule(odinn_batch* b, odinn_schedule* out);
/* State of the Y law's / U law's table (odinn_schedule.law_table; builds it for the current theta if needed): *usable = 1 when the stencil
 * kernels of the next solve / gradient will read it, its interval count (Y law: per glacier; U law: patches of the batch's bivariate table at the
 * resolution the build settled on, 16 x 8 ... 128 x 64), its largest measured deviation from the network
 * (relative; values below 1e-3 of the table's largest value: relative to that value) and the Hbar range per glacier
 * (metres; doubled by a solve that left it).  Any pointer may be NULL.  No counterpart in the reference. */
int odinn_get_law_table(odinn_batch* b, int* usable, int* n_intervals, double* max_rel_dev, double* hmax_per_glacier);
int odinn_set_fields(odinn_batch* b, int g, const double* H0, const double* B);
int odinn_set_A(odinn_batch* b, int g, double A);
int odinn_set_A_field(odinn_batch* b, int g, const double* A_dual);   /* CONST_A, gridded  */
int odinn_set_T_field(odinn_batch* b, int g, const double* T_dual);   /* NN_A_GRIDDED input */
/* n_H / n_gradS: D_hybrid exponents (target_D_hybrid.jl:180-185); pass <0 to use phys.n */
int odinn_set_law(odinn_batch* b, int kind, const odinn_mlp_desc* mlp, const double* theta, int P,
                  double n_H, double n_gradS);
int odinn_set_theta(odinn_batch* b, const double* theta, int P);
/* spatial evaluation of d law / d theta inside the theta-VJP of the Y and U laws (SIA2D_D_hybrid_target.interpolation,
 * src/models/target/target_D_hybrid.jl:12-15,121-160; SIA2D_D_target.interpolation, target_D_pure.jl:34-39,163-193):
 * ODINN_GRAD_INTERP_NONE = exact backprop at every dual node (`:None`); ODINN_GRAD_INTERP_LINEAR (`:Linear`) =
 *   Y law: gradients on the <= 2 n_interp_half knots of create_interpolation(Hbar) (target_utils.jl:245-293),
 *          interpolated linearly in Hbar;
 *   U law: gradients on the fixed (2 n_interp_half)^2 node grid of LawU's p_VJP! (Laws.jl:128-169; both axes
 *          LinRange(0, 100, 2 n_interp_half), as the law's cache is constructed), interpolated bilinearly in
 *          (Hbar, |grad S|).  The interpolant does not extrapolate: a dual node with Hbar > 100 makes the call that
 *          evaluates the theta-VJP fail with ODINN_ERR_ARG ("BoundsError"), as the reference throws.
 * odinn_set_law selects the reference's default: LINEAR with n_interp_half = 75 for ODINN_LAW_NN_Y, NONE for every
 * other law.  n_interp_half <= 256; ODINN_ERR_UNSUPPORTED for the A-type laws (no spatial law gradient). */
#define ODINN_GRAD_INTERP_NONE 0
#define ODINN_GRAD_INTERP_LINEAR 1
int odinn_set_grad_interpolation(odinn_batch* b, int kind, int n_interp_half);
/* thickness data glacier.thicknessData: n_ref fields at times t_ref; the loss mask is
 * is_in_glacier(H_ref, distance) (Losses.jl:266) */
int odinn_set_reference(odinn_batch* b, int g, int n_ref, const double* t_ref, const double* H_ref,
                        int distance);
/* mass-balance increment per step_MB: mb = min(mb0 + dmb_dS*(S - S_ref), mb_max), masked and
 * clipped as VJPs.jl:129-139.  S_ref may be NULL when dmb_dS == 0. Pass mb0 == NULL to disable. */
int odinn_set_mass_balance(odinn_batch* b, int g, const double* mb0, double dmb_dS,
                           const double* S_ref, double mb_max);

/* surface-velocity data glacier.velocityData: n_ref triples (|V|, Vx, Vy) of nx*ny fields at times
 * t_ref; the loss mask is V_ref > 0 (Losses.jl:316,361).  Pass n_ref = 0 to clear. */
int odinn_set_velocity_reference(odinn_batch* b, int g, int n_ref, const double* t_ref, const double* Vabs,
                                 const double* Vx, const double* Vy);
/* empirical loss function (src/losses/Losses.jl): LossH(L2Sum) [default], LossV(L2Sum, component,
 * scale_loss) :293-390, LossHV(hLoss, vLoss, scaling) :395-440 */
enum odinn_loss_kind { ODINN_LOSS_H = 0, ODINN_LOSS_V = 1, ODINN_LOSS_HV = 2 };
int odinn_set_loss(odinn_batch* b, int kind, int v_component_abs, int v_scale_loss, double hv_scaling);
/* the simple loss inside LossV (and LossHV's velocity part): L2Sum (default) or LogSum(eps) = log^2((a + eps) / (b + eps)) /
 * normalization (Losses.jl:34-49,207-229; Morlighem et al. 2010).  LogSum asserts non-negative fields in the reference, i.e.
 * it goes with component :abs; with :xy the gradient entry points fail with ODINN_ERR_ARG. */
/* target :D (U law): Velocity^ = U / f with f = parameters.simulation.f_surface_velocity_factor (target_D_pure.jl:206-255;
 * Sleipnir's default is out of tree, the reference's test sets 0.8): the surface-velocity entry points and LossV / LossHV
 * then differentiate the law by central differences (1e-4 in Hbar, 1e-6 in |grad S|) and backpropagate dU/dtheta per node. */
int odinn_set_surface_velocity_factor(odinn_batch* b, double f);
#define ODINN_SIMPLE_L2SUM 0
#define ODINN_SIMPLE_LOGSUM 1
int odinn_set_velocity_loss_function(odinn_batch* b, int simple_loss, double eps);
/* the same for LossH (and LossHV's thickness part): L2Sum (default) or LogSum(eps) on the mask is_in_glacier(H_ref, distance);
 * the reference asserts H_pred >= 0 for LogSum (Losses.jl:214) -- here a negative cell yields the NaN of log. */
int odinn_set_thickness_loss_function(odinn_batch* b, int simple_loss, double eps);
/* LossDhdt, a time-aggregated loss (src/losses/TimeAggregatedLosses.jl:38-113): glacier.dhdtData = (t0, t1, dhdt_ref);
 * with H0, H1 the predicted thickness at t0, t1 (both must be tstops of the solve), mask = H0 > 1e-2 and
 * dhdt = mean((H1 - H0)[mask]) / (t1 - t0), the term weight * (dhdt - dhdt_ref)^2 joins the loss of odinn_loss /
 * odinn_loss_grad / odinn_loss_grad_continuous and +-2 weight (dhdt - dhdt_ref) mask / (N_mask (t1 - t0)) joins lambda at
 * t1 / t0 (gradient.jl:170-215, :369-449).  `weight` is the MultiLoss lambda of the term relative to the data loss;
 * weight = 0 (default) switches the term off; t1 <= t0 clears a glacier's data.  A glacier without data does not contribute
 * to the term; a non-zero weight with NO glacier carrying the data makes the loss / gradient entry points fail with
 * ODINN_ERR_STATE (the same rule holds for LossAvgV and VelocityRegularization below). */
int odinn_set_dhdt_reference(odinn_batch* b, int g, double t0, double t1, double dhdt_ref);
int odinn_set_dhdt_loss(odinn_batch* b, double weight);
/* LossAvgV, the other time-aggregated loss (src/losses/TimeAggregatedLosses.jl:115-258): glacier.velocityData holds ONE
 * sample (date1 = t1, date2 = t2; vabs, vx, vy as nx*ny column-major arrays with V_from_H's pairing).  With
 * tLoss = t1:step:t2 (last point dropped), dt = diff, T = sum(dt), the predicted velocities V_from_H(H(tLoss_i)) are
 * averaged with weights dt_i / T and compared with the sample by L2Sum on mask = vabs > 0 (component :xy, or :abs when
 * component_abs != 0; normalization nx * ny); the term weight * loss joins odinn_loss / odinn_loss_grad /
 * odinn_loss_grad_continuous, its cotangent dl/dV dt_i / T is pulled back through surface_V at every tLoss_i
 * (VJP_lambda_dsurface_V/dH joins lambda at that stop, /dtheta joins dtheta; gradient.jl:170-215,274, :369-449,538).
 * Every tLoss_i must be among the glacier's stops of the solve (ODINN_ERR_ARG otherwise); every law with a surface-velocity path:
 * A-type (target :A) and the U law (target :D), like LossV.
 * weight = 0 (default) switches the term off; t2 <= t1 clears a glacier's sample. */
int odinn_set_avgv_reference(odinn_batch* b, int g, double t1, double t2, const double* Vabs, const double* Vx,
                             const double* Vy);
int odinn_set_avgv_loss(odinn_batch* b, double weight, double step, int component_abs);
/* VelocityRegularization(reg = TikhonovRegularization(), components = :abs, distance) (src/losses/Regularization.jl:64-79,
 * 192-245; the regulariser of the reference's documented multi-objective example): at every velocity-data time t_m, m >= 2,
 * of a glacier (the times given to odinn_set_velocity_reference; the maps themselves are not used) the term
 * weight * (t_m - t_{m-1}) * sum_mask (lap V)^2 with V = |V_from_H(H(t_m))| and mask = is_in_glacier(H(t_m), distance) & V > 0
 * joins the loss; its dL/dH joins lambda at that stop, its dL/dtheta is summed over the stops (odinn_loss_grad) or integrated
 * over the quadrature nodes on the interpolated state with Delta-t = 1 (odinn_loss_grad_continuous, gradient.jl:475-503).
 * `weight` is the MultiLoss lambda of the term relative to the data loss; 0 (default) switches it off.  A-type laws and the U law. */
int odinn_set_velocity_regularization(odinn_batch* b, double weight, int distance);

/* ---- fine-grained seams (host in / host out; parity + drop-in, not the fast path) ---- */
int odinn_sia2d_dhdt(odinn_batch* b, int g, const double* H, double t, double* dH);
int odinn_sia2d_vjp_H(odinn_batch* b, int g, const double* lam, const double* H, double t, double* dlam);
/* dtheta has P entries (P = 1 for CONST_A: d/dA) */
int odinn_sia2d_vjp_theta(odinn_batch* b, int g, const double* lam, const double* H, double t,
                          double* dtheta, int P);
/* Huginn.surface_V / V_from_H (restated from adjoint.jl:268-350): Vx, Vy are nx*ny with the
 * reference's inn1 pairing (dual node (i,j) at element [i,j]; last row and column 0).  Every law: Velocity^ of target :A
 * (target_A.jl:94-142), :D (U / f, target_D_pure.jl:206-255) and :D_hybrid AS THE REFERENCE WRITES IT
 * (target_D_hybrid.jl:210-372: the value uses the diffusivity's Gamma = 2 (rho g)^n / (n + 2), its H-partial a forward
 * difference of compute_D, its slope partial and theta-weight Gamma^ = 2 (rho g)^n / (n + 1); dY/dtheta exact per node or on
 * the knots of create_interpolation per odinn_set_grad_interpolation -- upstream has no test of that target's velocity path). */
int odinn_surface_V(odinn_batch* b, int g, const double* H, double* Vx, double* Vy);
/* VJP_lambda_dsurface_V/dH and /dtheta (DiscreteVJP), src/inverse/SIA2D/VJPs.jl:61-69 */
int odinn_surface_V_vjp_H(odinn_batch* b, int g, const double* dVx, const double* dVy, const double* H, double* out);
int odinn_surface_V_vjp_theta(odinn_batch* b, int g, const double* dVx, const double* dVy, const double* H,
                              double* dtheta, int P);
int odinn_mb_apply(odinn_batch* b, int g, const double* H, double* H_new, double* MB_applied);
int odinn_mb_vjp_H(odinn_batch* b, int g, const double* lam, const double* H_pre, double* out);
/* value of the law on the dual grid (scalar laws: out[0]) */
int odinn_eval_law(odinn_batch* b, int g, const double* H, double* out, int n_out);

/* ---- device-resident time loop -------------------------------------------------------- */
/* Per-glacier stop table.  The reference builds tstops PER GLACIER -- the `step` grid and solver.tstops, which all glaciers
 * share, plus the glacier's own thickness / velocity data times and the stops of its time-aggregated losses
 * (src/simulations/inversions/inversion_utils.jl:487-495, src/inverse/SIA2D/gradient.jl:96-107): a glacier's integrator
 * never lands on another glacier's data times, its result holds its own stops only, and the reverse loops of both adjoints
 * walk exactly those.  odinn_set_glacier_stops gives glacier g its own table t[0..n-1] (strictly increasing; t[0] and
 * t[n-1] must equal tstops[0] and tstops[n_stops-1] of the calls that follow: every glacier covers the same tspan); the
 * `tstops` argument of odinn_solve / odinn_loss_grad* then is the table of the glaciers WITHOUT one.  n = 0 clears.
 * Snapshot indices (odinn_get_snapshot) and every per-stop quantity of glacier g count ITS stops. */
int odinn_set_glacier_stops(odinn_batch* b, int g, int n, const double* t);
/* Integrates every glacier of the batch from tstops[0] to tstops[n_stops-1] with
 * RDPK3Sp35 + PID (adaptive, all control on the device), storing a snapshot at every
 * stop of the glacier and applying the mass balance at mb_times, strictly increasing times in (tstops[0], tstops[end]].
 * A mass-balance time that is not a stop of the glacier makes its integrator land there and apply the mass balance without
 * adding a snapshot to the result -- PeriodicCallback(mb_action!, step_MB) of inversion_utils.jl:498-517 with
 * step_MB not a multiple of solver.step.  (odinn_loss_grad rejects such times like the reference's DiscreteAdjoint,
 * gradient.jl:131; odinn_loss_grad_continuous stops its reverse solve there too and adds VJP_MB(lambda, H_itp(t) - MB_t), the
 * reverse PeriodicCallback of gradient.jl:413-432.)
 * stats: array of n_glaciers entries (may be NULL). */
int odinn_solve(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times,
                const odinn_solver_opts* opts, odinn_solve_stats* stats);
int odinn_get_snapshot(odinn_batch* b, int g, int istop, double* H_out);
int odinn_get_H(odinn_batch* b, int g, double* H_out); /* current state */
/* loss_per_glacier[g] = sum_tau w_tau * L2Sum(H_tau, Href_tau, mask)/N over stored snapshots */
int odinn_loss(odinn_batch* b, double* loss_per_glacier);
/* forward solve + discrete adjoint; loss and dtheta are SUMS over the batch's glaciers
 * (aggregate of Model.jl:208-224 for a FunctionalModel).  theta may be NULL (keep current).
 * lam0_out (optional) receives lambda at t0 of glacier 0 .. G-1 concatenated. */
int odinn_loss_grad(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops,
                    int n_mb, const double* mb_times, const odinn_solver_opts* opts,
                    double* loss, double* dtheta, odinn_solve_stats* stats);
/* Same contract with the reference's DEFAULT gradient method (UDEparameters.jl:63):
 * ContinuousAdjoint(VJP_method = DiscreteVJP()) of SIA2D_grad_batch! (gradient.jl:276-539):
 * reverse ODE dlam/dtau = J_H(H_itp(-tau))^T lam integrated on the device with adaptive RDPK3Sp35,
 * H interpolated linearly between the forward snapshots (:287), loss and mass-balance terms added
 * at the snapshot times (:331-365, :413-432), dL/dtheta = Gauss-Legendre quadrature of
 * J_theta(H_itp(t))^T lam(t) (:497-503); with LossV / LossHV also of dl_V/dtheta on reference velocities
 * interpolated linearly in time (:291-301, :475-503; the data must then span tspan).
 * stats / stats_rev: forward / reverse solve statistics per glacier (may be NULL). */
int odinn_loss_grad_continuous(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops,
                               int n_mb, const double* mb_times, const odinn_solver_opts* opts,
                               const odinn_adjoint_opts* adjoint_opts, double* loss, double* dtheta,
                               odinn_solve_stats* stats, odinn_solve_stats* stats_rev);
int odinn_get_lambda0(odinn_batch* b, int g, double* lam0);
/* selects ODINN_VJP_DISCRETE (default) or ODINN_VJP_CONTINUOUS for this batch */
int odinn_set_vjp_method(odinn_batch* b, int method);
/* TikhonovRegularization(operator = :laplacian) of src/losses/Regularization.jl:92-126 on one field
 * a[nx*ny] (column-major): *loss = sum_mask (lap a)^2 with lap = the reference's staggered Laplacian
 * (:330-352), grad = VJP_lap(2 mask lap a) (:372-382).  mask: nx*ny bytes or NULL (all true).
 * Used for InitialThicknessRegularization (a = H0) and RheologyRegularization (a = A on the dual grid). */
int odinn_tikhonov(odinn_batch* b, int nx, int ny, double dx, double dy, const double* a,
                   const unsigned char* mask, double* loss, double* grad);
/* Per-glacier pieces of the last odinn_loss_grad, for per-glacier parameters (PerGlacierModel:
 * GlacierWideInv / GriddedInv, classical LawA(params), Laws.jl:402-460; aggregate rule
 * Model.jl:208-224):  loss_g, and G_g = dL/dA_g for a glacier-wide scalar A.  With a gridded A
 * (odinn_set_A_field or NN_A_GRIDDED) odinn_get_grad_field returns dL/dA on the dual grid. */
int odinn_get_grad_parts(odinn_batch* b, double* loss_per_glacier, double* G_per_glacier);
int odinn_get_grad_field(odinn_batch* b, int g, double* dLdA_dual);

/* ---- multi-GPU: glaciers shard across ranks (one process per GPU), the only exchange of the path is the sum of
 * [loss, dtheta...] over ranks -- SIA2D_grad! (src/inverse/SIA2D/gradient.jl:6-31: pmap over glacier batches,
 * sum(losses), aggregate_grad of Model.jl:208-224).  The communicator is RCCL (ncclAllReduce over xGMI); the caller
 * distributes the 128-byte unique id of rank 0 by whatever means it has (Julia: Distributed/MPI; Python: torch.distributed
 * or a file) -- exactly NCCL's bootstrap contract. */
typedef struct odinn_comm odinn_comm;
#define ODINN_COMM_ID_BYTES 128
int odinn_comm_get_unique_id(void* id_out /* ODINN_COMM_ID_BYTES */);
/* collective over all ranks: binds the calling process's rank to `device` */
int odinn_comm_init_rank(int device, int nranks, int rank, const void* id, odinn_comm** out);
int odinn_comm_destroy(odinn_comm* c);
int odinn_comm_rank(const odinn_comm* c, int* rank, int* nranks);
/* in-place sum over ranks of n doubles: host buffer (staged through the communicator's device buffer) ... */
int odinn_comm_allreduce_sum(odinn_comm* c, double* inout, int n);
/* ... or device buffer, enqueued on `hip_stream` (a hipStream_t; NULL = the communicator's own stream), no host copy */
int odinn_comm_allreduce_sum_dev(odinn_comm* c, double* inout_dev, int n, void* hip_stream);
/* SIA2D_grad!: odinn_loss_grad (adjoint = 0, DiscreteAdjoint) or odinn_loss_grad_continuous (adjoint = 1, adjoint_opts
 * may be NULL) on this rank's batch, then ONE ncclAllReduce(sum, ncclDouble, 1 + P) of [loss, dtheta] on the batch's
 * stream; every rank returns the global loss and gradient.  comm == NULL: no reduction (single rank). */
int odinn_batch_loss_grad(odinn_batch* b, odinn_comm* comm, int adjoint, const double* theta, int P, int n_stops,
                          const double* tstops, int n_mb, const double* mb_times, const odinn_solver_opts* opts,
                          const odinn_adjoint_opts* adjoint_opts, double* loss, double* dtheta,
                          odinn_solve_stats* stats, odinn_solve_stats* stats_rev);

/* ---- measurement (HIP events on the batch's own stream, state already in HBM) -------- */
enum odinn_timed {
  ODINN_TIMED_DHDT = 0,     /* RHS only: read H,B write dH                    24 B/cell     */
  ODINN_TIMED_RK_STEP = 1,  /* one full RDPK3Sp35 step = 5 fused stage kernels              */
  ODINN_TIMED_VJP_H = 2,    /* read lam,H,B write dlam                        32 B/cell     */
  ODINN_TIMED_VJP_THETA = 3,/* read lam,H,B -> reduction                      24 B/cell     */
  ODINN_TIMED_RK_STAGE2 = 4,/* one interior stage kernel (stage 2)            56 B/cell     */
  ODINN_TIMED_SOLVE_STEP = 5,/* what odinn_solve launches per step under the default scheme:
                               RK step kernel(s) + controller (error-norm reduce, PID) + post-step */
  ODINN_TIMED_FUSED_STEP = 6,/* the temporally fused RDPK3Sp35 step kernel alone   24 B/cell */
  ODINN_TIMED_SOLVE_STEP_STAGED = 7,/* SOLVE_STEP forced onto the five per-stage kernels       */
  ODINN_TIMED_FUSED_STEP_SKIP = 8,  /* fused step kernel with the ice-free-tile shortcut enabled */
  ODINN_TIMED_EULER_CFL = 9,        /* the CFL Euler step kernel alone (RHS + update + max D)  24 B/cell */
  ODINN_TIMED_ADJ_STAGE2 = 10,      /* stage 2 of the reverse ODE of the continuous adjoint   72 B/cell */
  ODINN_TIMED_ADJ_FUSED_STEP = 11,  /* a whole reverse RDPK3Sp35 step in one kernel (integer-power law) 40 B/cell */
  ODINN_TIMED_LAW_FIELD = 12        /* hoisted LawA: A = NN_theta(T) on the dual grid, once per theta (Laws.jl:339-358);
                                       NN_A_GRIDDED only                                      16 B/node */
};
/* runs `iters` back-to-back launches over ALL glaciers of the batch after `warmup`
 * untimed ones; *ms_total is the elapsed time of the timed launches. */
int odinn_time_kernel(odinn_batch* b, int which, int warmup, int iters, double* ms_total);
/* the same launches without the event bracket, for callers that time with their own clock:
 * prepare resets the synthetic state (synchronous); enqueue launches iterations
 * first_iter .. first_iter+n-1 asynchronously on the batch's stream (sync with odinn_batch_sync). */
int odinn_bench_prepare(odinn_batch* b);
int odinn_bench_enqueue(odinn_batch* b, int which, int first_iter, int n);
/* on != 0: odinn_bench_enqueue(ODINN_TIMED_SOLVE_STEP) records a pair of HIP events around the fused step kernel of every step it
 * launches (on the batch's stream); odinn_bench_kernel_ms synchronises, returns the summed kernel time of the pairs recorded since
 * the last call and the number of launches, and starts over.  bench.py: the kernel's time and the step's time from ONE loop. */
int odinn_bench_kernel_events(odinn_batch* b, int on);
int odinn_bench_kernel_ms(odinn_batch* b, double* ms_total, int* launches);
/* total primal cells in the batch */
int64_t odinn_batch_cells(odinn_batch* b);

#ifdef __cplusplus
}
#endif
#endif /* ODINN_HIP_H */
