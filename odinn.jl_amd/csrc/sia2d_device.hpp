// sia2d_device.hpp -- CDNA4 (gfx950) device code of the SIA2D(+NN_theta) hot path.
//
// What is computed is ODINN.jl's SIA2D right-hand side and its hand-written discrete
// adjoint (reference: src/inverse/SIA2D/adjoint.jl:52-151,199-252 and the staggered
// transposes of src/inverse/SIA2D/inversion_utils.jl:3-66).  How it is computed is
// not a translation of that allocation-per-operator Julia code:
//
//   * one fused kernel per RHS evaluation / RK stage / VJP: a 64x16-cell tile (+1 halo)
//     of H and S=B+max(H,0) is staged once through LDS with coalesced 512-B row loads;
//     every dual-grid quantity (grad S, Hbar, D) is evaluated ONCE per dual node into LDS
//     -- a node depends only on its own 2x2 corner cells -- and every primal cell then
//     gathers from its 4 nodes and its 5-point neighbourhood.  In the H-VJP each node
//     computes what it contributes to its four corner cells and a cell adds four numbers.
//     No transpose is a scatter: the adjoint needs no atomics and is bitwise deterministic.
//   * the default time step is ONE kernel (sia2d_fused.hpp: the five RDPK3Sp35 stages
//     temporally fused on a 64-wide halo region kept in LDS / registers).
//   * all glaciers of a batch run in ONE launch: a tile table maps blockIdx -> (glacier,
//     tile), XCD-swizzled so that the tiles one XCD's L2 sees form a contiguous band.
//   * the adaptive time loop (RDPK3Sp35 3S*+ registers, embedded error norm, PID step
//     controller, tstops, mass-balance source, snapshots) lives on the device: stage
//     kernels read dt / accept flags from a per-glacier state record, block partial sums
//     of the error norm are combined in a fixed order by a one-block-per-glacier
//     controller kernel; the host only polls an "active glaciers" counter per chunk.
//
// Layout: element [i,j] of an nx*ny field at i + nx*j (i = x contiguous, as in Julia).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace odinn {

constexpr int TX = 64;         // tile width  (x, contiguous)  = one wavefront
constexpr int TY = 16;         // tile height (y)
constexpr int NT = 256;        // threads per block (4 wavefronts)
constexpr int NW = NT / 64;    // wavefronts per block
constexpr int RPT = TY / NW;   // rows per thread
constexpr int LDW = TX + 2;    // LDS row stride of cell tiles (with halo)
constexpr int LDN = TX + 1;    // LDS row stride of node tiles
constexpr int NNODE = (TX + 1) * (TY + 1);
constexpr int MAXP = 2048;     // max #theta supported by the in-kernel reductions
constexpr int MAXL = 8;        // max Dense layers
// fused-step kernel geometry (sia2d_fused.hpp): the halo REGION of a workgroup is exactly one
// wavefront wide, so a wavefront owns whole region rows (row index wave-uniform, column = lane)
constexpr int FH = 5;                 // halo = number of stages
#ifndef ODINN_FOY
#define ODINN_FOY 40                  // 64 x 50 region: 78 KB of LDS, 2 workgroups / CU
#endif
#ifndef ODINN_FNT
#define ODINN_FNT 512
#endif
constexpr int FRX = 64;               // region width = one wavefront
constexpr int FOX = FRX - 2 * FH, FOY = ODINN_FOY;  // 54 x FOY output tile
constexpr int FOYS = 8;               // "latency" tile height: used when the batch has too few tiles to fill
                                      // the GPU -- a workgroup then walks 18 region rows instead of 50
constexpr int FNT = ODINN_FNT;        // threads per block
constexpr int FNW = FNT / 64;
#ifndef ODINN_TNW
#define ODINN_TNW 8
#endif
#ifndef ODINN_TRPT
#define ODINN_TRPT 7
#endif
constexpr int TRPT = ODINN_TRPT;               // "strip" variant of the fused kernel (integer-power law): a wavefront owns
constexpr int TNW = ODINN_TNW;        // TRPT CONTIGUOUS region rows, so the y-neighbours of a cell live in the
constexpr int TNT = 64 * TNW;         // same thread's registers: TNW wavefronts, 64 x 56 region, 54 x 46 output tile
constexpr int TRY = TRPT * TNW;
constexpr int FOYT = TRY - 2 * FH;
constexpr int FOYT4 = 4 * TNW - 2 * FH;  // fused reverse step of small batches with 4 rows per thread: 54 x 22 output tiles
constexpr int FOYT2 = 2 * TNW - 2 * FH;  // ... of the smallest batches with 2 rows per thread: 54 x 6 output tiles (see odinn_hip.hip: rows of the reverse step)
constexpr int FOYT8 = 8 * TNW - 2 * FH;  // forward strip kernel with 8 rows per thread: 54 x 54 output tiles
constexpr int FLD = FRX + 1;          // LDS row stride (odd)



// ---- RDPK3Sp35 (Ranocha, Dalcin, Parsani, Ketcheson 2022), 3S*+ form ---------------
constexpr double c_g1[5] = {0.0, 2.587771979725733308135192812685323706e-01,
                                          -1.324380360140723382965420909764953437e-01,
                                          5.056033948190826045833606441415585735e-02,
                                          5.670532000739313812633197158607642990e-01};
constexpr double c_g2[5] = {1.0, 5.528354909301389892439698870483746541e-01,
                                          6.731871608203061824849561782794643600e-01,
                                          2.803103963297672407841316576323901761e-01,
                                          5.521525447020610386070346724931300367e-01};
constexpr double c_g3[5] = {0.0, 0.0, 0.0, 2.752563273304676380891217287572780582e-01,
                                          -8.950526174674033822276061734289327568e-01};
constexpr double c_dl[5] = {1.0, 3.407655879334525365094815965895763636e-01,
                                          3.414382655003386206551709871126405331e-01,
                                          7.229275366787987419692007421895451953e-01, 0.0};
constexpr double c_bt[5] = {2.300298624518076223899418286314123354e-01,
                                          3.021434166948288809034402119555380003e-01,
                                          8.025606185416310937583009085873554681e-01,
                                          4.362158943603440930655148245148766471e-01,
                                          1.129272530455059129782111662594436580e-01};
constexpr double c_bh[5] = {1.046363371354093758897668305991705199e-01,
                                          9.520431574956758809511173383346476348e-02,
                                          4.482446645568668405072421350300379357e-01,
                                          2.449030295461310135957132640369862245e-01,
                                          1.070116530120251819121660365003405564e-01};

// stage times c_i of RDPK3Sp35 (only the reverse ODE of the continuous adjoint is non-autonomous)
constexpr double c_cc[5] = {0.0, 2.300298624518076223899418286314123354e-01,
                                          4.050046072094990912268498160116125481e-01,
                                          8.947822893693433545220710894560512805e-01,
                                          7.235136928826589010272834603680114769e-01};

// ---- records living in device memory ------------------------------------------------
struct GDev {  // per-glacier constants
  int nx, ny, ntx, nty, tile0, ntiles;
  int tile0F, ntilesF;  // range in the fused-step tile table (FOX x FOY output tiles)
  int tile0Fs, ntilesFs; // ... and in the table of FOX x FOYS "latency" tiles
  int tile0Ft, ntilesFt; // ... and in the table of FOX x FOYT "strip" tiles
  int tile0Fu, ntilesFu; // ... and in the table of FOX x FOYT8 strip tiles (forward kernel, 8 rows per thread)
  int tile0Fv, ntilesFv; // ... and in the table of FOX x FOYT4 strip tiles (fused reverse step of small batches, 4 rows per thread)
  int tile0Fw, ntilesFw; // ... and in the table of FOX x FOYT2 strip tiles (fused reverse step of the smallest batches, 2 rows per thread)
  int tile0D, ntilesD;   // ... and in the table of 62 x 62 tiles of the RHS-only / CFL-Euler strip kernel
  long long off;   // offset of this glacier in the pooled primal arrays  [doubles]
  long long offd;  // offset in the pooled dual arrays
  double dx, dy, inv_dx, inv_dy, eta0;
  double hinv_dx, hinv_dy;    // 0.5/dx, 0.5/dy
  double hinv_dx2, hinv_dy2;  // 0.5/dx^2, 0.5/dy^2
  double A;        // scalar creep coefficient in use (CONST_A or hoisted NN_A_SCALAR)
  double Gam;      // 2 (rho g)^n / (n+2)      target_utils.jl:3-12
  double Sc;       // C (rho g)^(p-q)          target_utils.jl:14-18
  double n, p, q, T, nH, nS;
  double minA, maxA;
  int fast;        // n == 3 && Sc == 0 && eta0 == 1 -> integer-power path, no sqrt/pow, no eta0 products
  int use_Afield;  // A read from the dual-grid field
  int has_mb;
  double dmb_dS, mb_max;
  // tabulated Y law (law mode LM_YTAB): this glacier's table starts at yt_off doubles, interval i covers
  // [i, i + 1) / yt_inv_h metres of Hbar
  double yt_inv_h;
  long long yt_off;
  int yt_fast;  // n_H = n_gradS = 3, no sliding: the integer-power form of the Y law's geometry factor (no upow / spow)
  int yt_pad;
};

struct GState {  // per-glacier integrator state (written by the controller kernel)
  double t, dt, e2, e3, EEst;
  int accepted;   // last step accepted -> stage 1 reads the new state, else S3 (uprev)
  int at_stop;    // the accepted step landed on tstops[istop-1]: post-step must run
  int mb_now;     // ... and the mass balance is applied there
  int mb_slot;    // index into the pre-MB snapshot store
  int done, istop, clipped, cur;  // cur: which ping-pong buffer holds the accepted state
  long long naccept, nreject;
  int nonfinite;
  int pad;
  int snap_slot;  // snapshot slot of the stop just reached (forward solve: CtrlArgs::snap_slot of that stop)
  int pad2;       // adaptive solves: consecutive attempts that did not advance t (rejections, accepted steps below the resolution of t)
};

struct AdjState {  // per-glacier state of the reverse (continuous-adjoint) solve, written by the controller
  int seg;        // H_itp segment [tsnap[seg], tsnap[seg+1]] that contains the coming step
  int seg_stop;   // segment that contains the stop just reached
  int snapj;      // forward snapshot index of the stop just reached (-1: not a snapshot time)
  int pad;        // > 0: the stop is a mass-balance time that is not a result stop; pad - 1 = the hidden snapshot slot that
                  // holds the forward state right after that mass balance (0 otherwise)
  double qw;      // Gauss-Legendre weight of the stop just reached (0: not a quadrature node)
  double s_stop;  // interpolation weight of the stop inside seg_stop
  double sitp[5]; // interpolation weights at the five stage times tau + c_i dt of the coming step
};

struct LawDev {  // passed by value to kernels
  int kind, n_layers, has_pre, post_kind, P, maxw;
  int widths[9];
  int acts[8];
  double pre_lo[2], pre_inv[2];
  double post_lo, post_hi;
  const double* theta;  // device
  // run-time architectures: one zero-padded row of `padw` (16 | 32) weights per unit, rows in layer / unit order, and the
  // biases in the same order (odinn_set_theta repacks them) -- a unit's weights are then ONE aligned scalar load
  const double* theta_pad;
  const double* bias_pad;
  // tabulated Y law (LM_YTAB): ytab_ni intervals per glacier, six Chebyshev-basis-free monomial coefficients each (ytab_eval);
  // *ytab_over is raised by any node whose Hbar lies beyond the table
  const double* ytab;
  int* ytab_over;
  int ytab_ni;
  // tabulated U law (LM_UTAB): ONE table for the batch (the law's inputs are both fields: Hbar and |grad S|) of utab_nh x utab_ns
  // patches, 36 coefficients c[a][b] of u^a v^b each (utab_eval); patch (i, j) covers [i, i + 1) / ut_inv_h metres of Hbar and
  // [j, j + 1) / ut_inv_s of |grad S|; overflow raises *ytab_over as well
  const double* utab;
  int utab_nh, utab_ns;
  int ut_nolds;  // ODINN_UT_LDS=0 (test aid): the tile kernels read the patches from global memory even where utab_stage could stage them
  double ut_inv_h, ut_inv_s;
};

struct Pools {  // pooled device arrays (all glaciers concatenated)
  const int4* tiles;
  const GDev* gd;
  GState* gs;
  const double* B;
  const double* Afield;  // dual
  double* part;          // block partial sums, 4 doubles per tile
};

// ---- small helpers ------------------------------------------------------------------
// index of this wavefront in its workgroup, as a SCALAR: rows of a tile are dealt to wavefronts, so
// row indices, row predicates and the row part of every address then live in SGPRs / SALU
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// One-instruction min/max for the clamp stencils.  fmin/fmax lower to llvm.minnum/maxnum, for which
// the compiler (IEEE mode) first canonicalises every operand that comes from memory with an extra
// v_max_f64 x,x,x -- 9 % of the VALU work of the step kernel.  No NaN reaches these sites (a NaN
// state is caught by the error norm), and for non-NaN operands the results are identical.
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(a, -b) and max(a, 0) with the source modifier / inline constant the compiler would have used
__device__ __forceinline__ double vmax_neg(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax0(double a) {
  double r;
  asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
  return r;
}
// clamp of a slope between -lom and up: max(min(e, up), -lom)   (inversion_utils.jl:17-20,31-34)
__device__ __forceinline__ double clampn(double e, double up, double lom) { return vmax_neg(vmin(e, up), lom); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// deterministic block sum; result valid in thread 0.  red: LDS scratch of NW doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NW; ++k) s += red[k];
  }
  return s;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double block_max(double v, double* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NW; ++k) s = fmax(s, red[k]);
  }
  return s;
}

// ---- fast fp64 transcendentals for the activations ---------------------------------------
// ocml's exp/log1p cost ~140 fp64 instructions each (double-double internals); an MLP node
// evaluation is dominated by them.  softplus/sigmoid only need exp on (-inf, 0] and log1p on
// [0, 1]; the routines below are <= 2 ulp there (checked against 40-digit arithmetic) at
// ~20 / ~28 instructions.
__device__ __forceinline__ double fast_div(double n, double d) {  // d normal, <= 1 ulp
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return fma(fma(-d, q, n), r, q);
}
__device__ __forceinline__ double exp_nonpos(double y) {  // y <= 0
  y = fmax(y, -750.0);
  const double n = rint(y * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;          // 1/13!
  p = fma(p, r, 2.08767569878681e-09);
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 1.984126984126984e-04);
  p = fma(p, r, 1.3888888888888889e-03);
  p = fma(p, r, 8.333333333333333e-03);
  p = fma(p, r, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
// log1p on [0, 1] from a 65-entry table {log1p(i/64), 1/(1 + i/64)} (one 16-byte gather, L1-resident) and a degree-8
// remainder: 1 + t = (1 + t_i)(1 + r), r = (t - t_i)/(1 + t_i), |r| <= 1/128, log1p(t) = L_i + r P(r).  t - t_i is exact
// (Sterbenz), the truncation r^8/9 < 2e-18 relative even where L_i = 0; <= 2 ulp like the atanh form it replaces (kept
// below as log1p_01_series, ODINN_LOG1P_TABLE=0) at ~19 instead of ~31 instructions: 32 of the 33 activations of the
// 2 x 16 net are softplus, each one exp + one log1p.
// Two forms: ODINN_LOG1P_TABLE == 1 reads the table from global memory (L1-resident; the one-node-per-thread kernels: hoisted
// law field, knot / node-grid gradients, velocity kernels), == 2 from a copy in LDS that the tile loaders fill (the stencil
// kernels with an inlined network: k_fwd.hip, k_adj.hip, k_fused.hip define it before including this header).  With the
// global form those stencil kernels are gather-bound for the 16-wide net (32 softplus per node at 2 waves per SIMD:
// 2797 -> 5031 us per solve step at 8 x 1024^2); with the LDS form every net gains (us per solve step at 8 x 1024^2,
// series -> LDS table: 2-16-16-1 Y law per-stage 2797 -> 2194, fused 3614 -> 2475; default 2-3-10-3-1 Y law 1852 -> 1244
// and 2431 -> 1530; U law 1394 -> 1185; 4 alpine glaciers, fused: 128 -> 103, 78 -> 54, 73 -> 56); the hoisted-law field
// 0.49 -> 0.38 ms.  0 selects the series everywhere.
#ifndef ODINN_LOG1P_TABLE
#define ODINN_LOG1P_TABLE 1
#endif
__device__ static const double LOG1P_TAB[65][2] __attribute__((aligned(16))) = {
    {0.0, 1.0}, {0.015504186535965254, 0.9846153846153847},
    {0.030771658666753687, 0.9696969696969697}, {0.0458095360312942, 0.9552238805970149},
    {0.06062462181643484, 0.9411764705882353}, {0.07522342123758753, 0.927536231884058},
    {0.08961215868968714, 0.9142857142857143}, {0.10379679368164356, 0.9014084507042254},
    {0.11778303565638346, 0.8888888888888888}, {0.13157635778871926, 0.8767123287671232},
    {0.1451820098444979, 0.8648648648648649}, {0.15860503017663857, 0.8533333333333334},
    {0.17185025692665923, 0.8421052631578947}, {0.184922338494012, 0.8311688311688312},
    {0.19782574332991987, 0.8205128205128205}, {0.21056476910734964, 0.810126582278481},
    {0.22314355131420976, 0.8}, {0.2355660713127669, 0.7901234567901234},
    {0.24783616390458124, 0.7804878048780488}, {0.25995752443692605, 0.7710843373493976},
    {0.27193371548364176, 0.7619047619047619}, {0.2837681731306446, 0.7529411764705882},
    {0.2954642128938359, 0.7441860465116279}, {0.3070250352949119, 0.735632183908046},
    {0.3184537311185346, 0.7272727272727273}, {0.329753286372468, 0.7191011235955056},
    {0.3409265869705932, 0.7111111111111111}, {0.3519764231571782, 0.7032967032967034},
    {0.3629054936893685, 0.6956521739130435}, {0.37371640979358406, 0.6881720430107527},
    {0.38441169891033206, 0.6808510638297872}, {0.394993808240869, 0.6736842105263158},
    {0.4054651081081644, 0.6666666666666666}, {0.415827895143711, 0.6597938144329897},
    {0.42608439531090003, 0.6530612244897959}, {0.43623676677491807, 0.6464646464646465},
    {0.44628710262841953, 0.64}, {0.45623743348158763, 0.6336633663366337},
    {0.46608972992459924, 0.6274509803921569}, {0.4758459048699639, 0.6213592233009708},
    {0.48550781578170077, 0.6153846153846154}, {0.4950772667978515, 0.6095238095238096},
    {0.5045560107523953, 0.6037735849056604}, {0.5139457511022343, 0.5981308411214953},
    {0.5232481437645479, 0.5925925925925926}, {0.5324647988694718, 0.5871559633027523},
    {0.5415972824327444, 0.5818181818181818}, {0.5506471179526623, 0.5765765765765766},
    {0.5596157879354227, 0.5714285714285714}, {0.5685047353526687, 0.5663716814159292},
    {0.5773153650348236, 0.5614035087719298}, {0.5860490450035782, 0.5565217391304348},
    {0.5947071077466928, 0.5517241379310345}, {0.6032908514380843, 0.5470085470085471},
    {0.6118015411059929, 0.5423728813559322}, {0.6202404097518576, 0.5378151260504201},
    {0.6286086594223741, 0.5333333333333333}, {0.6369074622370692, 0.5289256198347108},
    {0.6451379613735847, 0.5245901639344263}, {0.6533012720127457, 0.5203252032520326},
    {0.661398482245365, 0.5161290322580645}, {0.6694306539426292, 0.512},
    {0.6773988235918061, 0.5079365079365079}, {0.6853040030989194, 0.5039370078740157},
    {0.6931471805599453, 0.5}};
#if ODINN_LOG1P_TABLE == 2
// mode 2 (the stencil kernels with an inlined network: k_fwd.hip, k_adj.hip, k_fused.hip): the table in LDS, filled by the
// tile loader -- an L1 gather per activation makes the throughput-bound ones gather-bound, a ds_read_b128 does not
__shared__ double2 g_log1p_lds[65];
#endif
__device__ __forceinline__ double log1p_01_series(double t) {  // 0 <= t <= 1
  const bool big = t > 0.41421356237309503;
  const double num = big ? t - 1.0 : t;
  const double den = big ? t + 3.0 : t + 2.0;
  const double s_ = fast_div(num, den);  // log1p(t) = [ln 2 +] 2 atanh(s)
  const double z = s_ * s_;
  double p = 1.0 / 23.0;
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = p * z;
  double r = fma(s_, p, s_);
  r = r + r;
  return big ? r + 0.6931471805599453 : r;
}
__device__ __forceinline__ double log1p_01(double t) {  // 0 <= t <= 1
#if ODINN_LOG1P_TABLE
  const double k = rint(t * 64.0);
#if ODINN_LOG1P_TABLE == 2
  const double2 e = g_log1p_lds[(int)k];
#else
  const double2 e = *reinterpret_cast<const double2*>(&LOG1P_TAB[(int)k][0]);
#endif
  const double r = (t - k * 0.015625) * e.y;
  double p = -0.125;
  p = fma(p, r, 1.0 / 7.0);
  p = fma(p, r, -1.0 / 6.0);
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p, r, 1.0);
  return fma(r, p, e.x);
#else
  return log1p_01_series(t);
#endif
}
__device__ __forceinline__ double softplus_f(double x) { return log1p_01(exp_nonpos(-fabs(x))) + fmax(x, 0.0); }
__device__ __forceinline__ double sigmoid_f(double x) {
  const double t = exp_nonpos(-fabs(x));
  return fast_div(x >= 0.0 ? 1.0 : t, 1.0 + t);
}

__device__ __forceinline__ double act_f(int code, double x) {
  switch (code) {
    case 1: return softplus_f(x);  // NNlib.softplus = log1p(exp(-|x|)) + relu(x)
    case 2: return sigmoid_f(x);
    case 3: { const double c = 0.7978845608028654; return 0.5 * x * (1.0 + tanh(c * (x + 0.044715 * x * x * x))); }
    case 4: return tanh(x);
    case 5: return fmax(x, 0.0);
    default: return x;
  }
}
__device__ __forceinline__ double dact_f(int code, double x) {
  switch (code) {
    case 1: return sigmoid_f(x);
    case 2: { const double s = sigmoid_f(x); return s * (1.0 - s); }
    case 3: { const double c = 0.7978845608028654; double u = c * (x + 0.044715 * x * x * x); double th = tanh(u);
              double du = c * (1.0 + 3.0 * 0.044715 * x * x); return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * du; }
    case 4: { double th = tanh(x); return 1.0 - th * th; }
    case 5: return x > 0.0 ? 1.0 : 0.0;
    default: return 1.0;
  }
}
__device__ __forceinline__ double postscale_f(const LawDev& L, double y) {
  switch (L.post_kind) {
    case 1: return L.post_lo + (L.post_hi - L.post_lo) * y;
    case 2: return L.post_hi * exp_nonpos((y - 1.0) / y);  // y in (0, 1]
    case 3: return L.post_hi * y;
    default: return y;
  }
}
__device__ __forceinline__ double dpostscale_f(const LawDev& L, double y) {
  switch (L.post_kind) {
    case 1: return L.post_hi - L.post_lo;
    case 2: return L.post_hi * exp_nonpos((y - 1.0) / y) / (y * y);
    case 3: return L.post_hi;
    default: return 1.0;
  }
}

// Per-lane MLP evaluation (_pred_NN, src/laws/Laws.jl:34-36) for run-time architectures; weights are read with wave-uniform
// addresses (scalar loads through the constant cache), activations live in registers.
#define ODINN_RT_PREFETCH 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
// Rolled: ONE copy of the unit code (a padded row of weights through one scalar load, MW multiply-adds at compile-time register
// indices -- the padding multiplies zeros -- one activation, chosen by a uniform branch) looped over the units of a layer and
// over the layers; the unit's result goes to z[o] with a wave-uniform dynamic register index (s_set_gpr_idx).  Summation order
// = bias, then the inputs in ascending order, as everywhere else.  (Until round 4 this was a fully unrolled, width-predicated
// evaluator behind a function call: 16 inlined copies of EVERY activation per layer, its descriptor and weights read with
// per-lane flat loads because nothing was provably uniform across the call -- 7.5x the default architecture's forward stage
// and 13x its reverse stage at 8 x 512^2; now 2.3x and 3.2x, profiles/r04/rt_arch.txt.)
template <int MW>
__device__ __forceinline__ double mlp_eval_rt(const LawDev& L, double x0, double x1) {
  typedef double vec __attribute__((ext_vector_type(MW)));
  vec h = 0.0;
  h[0] = L.has_pre ? (x0 - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0;
  if (L.widths[0] > 1) h[1] = L.has_pre ? (x1 - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1;
  const double* __restrict__ w = L.theta_pad;
  const double* __restrict__ bz = L.bias_pad;
#if ODINN_RT_PREFETCH
  // the NEXT unit's row is requested before this unit's arithmetic (the table ends with one spare row): at the two
  // wavefronts per SIMD these kernels run at, nothing else hides the scalar-cache latency
  constexpr bool PF = MW <= 16;  // (two rows of 32 weights do not fit the scalar registers)
#else
  constexpr bool PF = false;
#endif
  double wc[MW], bc = 0.0;
  if constexpr (PF) {
#pragma unroll
    for (int i = 0; i < MW; ++i) wc[i] = w[i];
    bc = bz[0];
  }
#pragma nounroll
  for (int l = 0; l < L.n_layers; ++l) {
    const int nout = L.widths[l + 1], a = L.acts[l];
    vec z = 0.0;
#pragma nounroll
    for (int o = 0; o < nout; ++o, w += MW, ++bz) {
      double acc;
      if constexpr (PF) {
        double wn[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) wn[i] = w[MW + i];
        const double bn = bz[1];
        acc = bc;
#pragma unroll
        for (int i = 0; i < MW; ++i) acc = fma(wc[i], h[i], acc);
        z[o] = act_f(a, acc);
#pragma unroll
        for (int i = 0; i < MW; ++i) wc[i] = wn[i];
        bc = bn;
      } else {
        acc = bz[0];
#pragma unroll
        for (int i = 0; i < MW; ++i) acc = fma(w[i], h[i], acc);
        z[o] = act_f(a, acc);
      }
    }
    h = z;
  }
  return postscale_f(L, h[0]);
}

// (kernels that are not instantiated per law mode: the velocity kernels, the hoisted law field, the seams)
__device__ __forceinline__ double mlp_eval_any(const LawDev& L, double x0, double x1) {
  if (L.maxw <= 16) return mlp_eval_rt<16>(L, x0, x1);
  return mlp_eval_rt<32>(L, x0, x1);
}

// g[k*stride] += wgt * d out / d theta_k at one input (exact backprop; stands in for the
// Zygote/Mooncake pass of src/laws/auto_VJP.jl:114-122).  `g` is a thread-private
// accumulator in global memory laid out g[k*stride] so that a wavefront's accesses coalesce.
inline __device__ __noinline__ void mlp_grad(const LawDev& L, double x0, double x1, double wgt, double* g, long long stride) {
  constexpr int MAXW = 32;
  double hs[MAXL + 1][MAXW];
  double zs[MAXL][MAXW];
  const double* __restrict__ th = L.theta;
  hs[0][0] = L.has_pre ? (x0 - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0;
  hs[0][1] = L.has_pre ? (x1 - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1;
  int offs[MAXL + 1];
  offs[0] = 0;
  for (int l = 0; l < L.n_layers; ++l) {
    const int nin = L.widths[l], nout = L.widths[l + 1], a = L.acts[l];
    const int off = offs[l];
    for (int o = 0; o < nout; ++o) {
      double acc = th[off + nin * nout + o];
      for (int i = 0; i < nin; ++i) acc = fma(th[off + o + nout * i], hs[l][i], acc);
      zs[l][o] = acc;
      hs[l + 1][o] = act_f(a, acc);
    }
    offs[l + 1] = off + nout * (nin + 1);
  }
  double gv[MAXW], gn[MAXW];
  gv[0] = wgt * dpostscale_f(L, hs[L.n_layers][0]);
  for (int l = L.n_layers - 1; l >= 0; --l) {
    const int nin = L.widths[l], nout = L.widths[l + 1], a = L.acts[l];
    const int off = offs[l];
    for (int i = 0; i < nin; ++i) gn[i] = 0.0;
    for (int o = 0; o < nout; ++o) {
      const double dz = gv[o] * dact_f(a, zs[l][o]);
      g[(long long)(off + nin * nout + o) * stride] += dz;
      for (int i = 0; i < nin; ++i) {
        g[(long long)(off + o + nout * i) * stride] += dz * hs[l][i];
        gn[i] = fma(th[off + o + nout * i], dz, gn[i]);
      }
    }
    for (int i = 0; i < nin; ++i) gv[i] = gn[i];
  }
}

// ---- wave-reduced backprop: sum over the 64 lanes of a wavefront of wgt * d out / d theta, accumulated in LDS ----------
// (users: k_law_field_grad_wave -- the hoisted gridded law -- and the knot contraction of the `:Linear` interpolation)
constexpr int WG_SLOTS = 16, WG_LD = 65;
struct WaveAcc {
  double (*stage)[WG_LD];  // the wavefront's [WG_SLOTS][WG_LD] staging area
  const int* order;        // parameter index of the n-th contribution of a backward pass (mlp_grad_order)
  double* acc;             // the wavefront's [P] accumulators
};
struct ArchRT { static constexpr int NL = MAXL, MAXW = 32; static constexpr int W[MAXL + 1] = {}; static constexpr int A[MAXL] = {}; };
// order in which mlp_grad_wave emits its contributions: layers last to first, per output unit the bias, then its weights
__device__ inline void mlp_grad_order(const LawDev& L, int* order) {
  int offs[MAXL + 1];
  offs[0] = 0;
  for (int l = 0; l < L.n_layers; ++l) offs[l + 1] = offs[l] + L.widths[l + 1] * (L.widths[l] + 1);
  int n = 0;
  for (int l = L.n_layers - 1; l >= 0; --l) {
    const int nin = L.widths[l], nout = L.widths[l + 1];
    for (int o = 0; o < nout; ++o) {
      order[n++] = offs[l] + nin * nout + o;
      for (int i = 0; i < nin; ++i) order[n++] = offs[l] + o + nout * i;
    }
  }
}
__device__ __forceinline__ void wave_acc_flush(const WaveAcc& A, int base, int nslot, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int kk = lane & 15, q = lane >> 4;
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += A.stage[kk][16 * q + j];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (lane < nslot) A.acc[A.order[base + lane]] += s;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// activation and its derivative from one exponential (softplus / sigmoid: the formulas of act_f and dact_f)
__device__ __forceinline__ void act_dact(int code, double z, double& a, double& da) {
  if (code == 1) {
    const double t = exp_nonpos(-fabs(z));
    a = log1p_01(t) + fmax(z, 0.0);
    da = fast_div(z >= 0.0 ? 1.0 : t, 1.0 + t);
  } else if (code == 2) {
    a = sigmoid_f(z);
    da = a * (1.0 - a);
  } else {
    a = act_f(code, z);
    da = dact_f(code, z);
  }
}
// acc[k] += sum over the wavefront's 64 lanes of wgt * d out / d theta_k at the lane's input (mlp_grad's arithmetic per
// lane).  ALL 64 lanes call it together (a lane without a node passes wgt = 0).  FIXED: compile-time architecture AR,
// fully unrolled, activations in registers; otherwise the run-time architecture of L (AR = ArchRT).
template <class AR, bool FIXED>
__device__ __forceinline__ void mlp_grad_wave(const LawDev& L, double x0, double x1, double wgt, const WaveAcc& A, int lane) {
  constexpr int MW = AR::MAXW, ML = AR::NL;
  const int nl = FIXED ? AR::NL : L.n_layers;
  double hs[ML + 1][MW], ds[ML][MW];
  const double* __restrict__ th = L.theta;
  hs[0][0] = L.has_pre ? (x0 - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0;
  if (MW > 1) hs[0][1] = L.has_pre ? (x1 - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1;
  int offs[ML + 1];
  offs[0] = 0;
#pragma unroll
  for (int l = 0; l < nl; ++l) {
    const int nin = FIXED ? AR::W[l] : L.widths[l], nout = FIXED ? AR::W[l + 1] : L.widths[l + 1];
    const int a = FIXED ? AR::A[l] : L.acts[l];
    const int off = offs[l];
#pragma unroll
    for (int o = 0; o < nout; ++o) {
      double acc = th[off + nin * nout + o];
#pragma unroll
      for (int i = 0; i < nin; ++i) acc = fma(th[off + o + nout * i], hs[l][i], acc);
      act_dact(a, acc, hs[l + 1][o], ds[l][o]);
    }
    offs[l + 1] = off + nout * (nin + 1);
  }
  double gv[MW], gn[MW];
  gv[0] = wgt * dpostscale_f(L, hs[nl][0]);
  int slot = 0, base = 0;
#pragma unroll
  for (int l = nl - 1; l >= 0; --l) {
    const int nin = FIXED ? AR::W[l] : L.widths[l], nout = FIXED ? AR::W[l + 1] : L.widths[l + 1];
    const int off = offs[l];
#pragma unroll
    for (int i = 0; i < nin; ++i) gn[i] = 0.0;
#pragma unroll
    for (int o = 0; o < nout; ++o) {
      const double dz = gv[o] * ds[l][o];
#pragma unroll
      for (int i = -1; i < nin; ++i) {  // i == -1: the bias
        A.stage[slot][lane] = i < 0 ? dz : dz * hs[l][i < 0 ? 0 : i];
        if (i >= 0) gn[i] = fma(th[off + o + nout * i], dz, gn[i]);
        if (++slot == WG_SLOTS) {
          wave_acc_flush(A, base, WG_SLOTS, lane);
          base += WG_SLOTS;
          slot = 0;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < nin; ++i) gv[i] = gn[i];
  }
  if (slot) wave_acc_flush(A, base, slot, lane);
}

// ---- law modes of the stencil kernels ---------------------------------------------------------
// 0 integer-power A law, 1 generic-pow A law, 2 inlined MLP with run-time architecture,
// 3..5 inlined MLP with a compile-time architecture (fully unrolled, activations in registers):
//   3: 2 -> 3 -> 10 -> 3 -> 1, softplus x3 + sigmoid   (build_default_NN, ML_utils.jl:31-36)
//   4: 2 -> 16 -> 16 -> 1,     softplus x2 + sigmoid   (BASELINE configs[2], "2 layers x 16 units")
//   5: 2 -> 3 -> 1,            softplus + sigmoid      (test_mode light net, ML_utils.jl:26-29)
//   7: the Y law of target :D_hybrid through a per-glacier TABLE of Y(Hbar) (its other input, the glacier's temperature, is a
//      scalar: LawY's inputs are (T, Hbar), Laws.jl:240-273) -- structurally a closed-form law: no network in the kernel
//   8: the U law of target :D through ONE bivariate table U(Hbar, |grad S|) for the batch -- same idea, two variables
constexpr int LM_FAST = 0, LM_POW = 1, LM_NN = 2, LM_NN_DEF = 3, LM_NN_16 = 4, LM_NN_LIGHT = 5, LM_NN_WIDE = 6, LM_YTAB = 7, LM_UTAB = 8;
constexpr bool lm_is_nn(int lm) { return lm >= LM_NN && lm <= LM_NN_WIDE; }  // (7, 8: tables -- structurally closed-form laws)
// run-time architectures: LM_NN (every layer has <= 16 inputs) and LM_NN_WIDE (<= 32) -- kernels of their own, the 32-wide
// evaluator holds 128 registers of activations and would set the register allocation of the common case
constexpr bool lm_is_rt(int lm) { return lm == LM_NN || lm == LM_NN_WIDE; }

struct ArchDef   { static constexpr int NL = 4, MAXW = 10; static constexpr int W[5] = {2, 3, 10, 3, 1}; static constexpr int A[4] = {1, 1, 1, 2}; };
struct Arch16    { static constexpr int NL = 3, MAXW = 16; static constexpr int W[4] = {2, 16, 16, 1};   static constexpr int A[3] = {1, 1, 2}; };
struct ArchLight { static constexpr int NL = 2, MAXW = 3;  static constexpr int W[3] = {2, 3, 1};        static constexpr int A[2] = {1, 2}; };

template <class AR>
constexpr int arch_off(int l) {
  int o = 0;
  for (int k = 0; k < l; ++k) o += AR::W[k + 1] * (AR::W[k] + 1);
  return o;
}

// softplus and sigmoid share t = exp(-|x|)
template <int CODE>
__device__ __forceinline__ double act_c(double x) {
  if (CODE == 1) return softplus_f(x);
  if (CODE == 2) return sigmoid_f(x);
  return act_f(CODE, x);
}

template <class AR, int l>
__device__ __forceinline__ void mlp_layer_fixed(const double* __restrict__ th, const double (&hin)[AR::MAXW],
                                                double (&hout)[AR::MAXW]) {
  constexpr int nin = AR::W[l], nout = AR::W[l + 1], off = arch_off<AR>(l);
#pragma unroll
  for (int o = 0; o < nout; ++o) {
    double acc = th[off + nin * nout + o];
#pragma unroll
    for (int i = 0; i < nin; ++i) acc = fma(th[off + o + nout * i], hin[i], acc);
    hout[o] = act_c<AR::A[l]>(acc);
  }
}

template <class AR>
__device__ __forceinline__ double mlp_eval_fixed(const LawDev& L, double x0, double x1) {
  double h0[AR::MAXW], h1[AR::MAXW];
  h0[0] = L.has_pre ? (x0 - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0;
  h0[1] = L.has_pre ? (x1 - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1;
  const double* __restrict__ th = L.theta;
  mlp_layer_fixed<AR, 0>(th, h0, h1);
  if constexpr (AR::NL > 1) mlp_layer_fixed<AR, 1>(th, h1, h0);
  if constexpr (AR::NL > 2) mlp_layer_fixed<AR, 2>(th, h0, h1);
  if constexpr (AR::NL > 3) mlp_layer_fixed<AR, 3>(th, h1, h0);
  return postscale_f(L, (AR::NL & 1) ? h1[0] : h0[0]);
}

template <int LM>
__device__ __forceinline__ double mlp_eval_lm(const LawDev& L, double x0, double x1) {
  if constexpr (LM == LM_NN_DEF) return mlp_eval_fixed<ArchDef>(L, x0, x1);
  else if constexpr (LM == LM_NN_16) return mlp_eval_fixed<Arch16>(L, x0, x1);
  else if constexpr (LM == LM_NN_LIGHT) return mlp_eval_fixed<ArchLight>(L, x0, x1);
  else if constexpr (LM == LM_NN_WIDE) return mlp_eval_rt<32>(L, x0, x1);
  else return mlp_eval_rt<16>(L, x0, x1);
}

// ---- the network at x AND at x + delta e_d for a few tiny perturbations, for the price of ~1.5 evaluations ----------------
// The reference forms dD/dHbar and dD/d|grad S| of the per-node-MLP laws by FINITE DIFFERENCES of the law (Y: one forward
// difference with 1e-4, target_D_hybrid.jl:58-71; U: central differences with 1e-4 and 1e-6, target_D_pure.jl:105-137):
// 2 resp. 5 network evaluations per dual node and stage, which is what bounds the reverse kernels of these laws (fp64
// transcendentals: ~45 instructions per softplus).  The perturbed evaluations differ from the central one by |dz| ~ 1e-6 in
// every pre-activation, so they are evaluated through the local Taylor expansion of each activation around the central
// pre-activation,
//     a(z + dz) - a(z) = dz (a1 + dz (a2 / 2 + dz (a3 / 6 + dz (a4 / 24 + dz a5 / 120)))),   ak = k-th derivative of a at z,
// whose coefficients are polynomials of sigma(z) = 1 / (1 + exp(-z)) -- the same exponential the activation itself needs --
// and the perturbation dz of the next layer is the exact linear map W da.  Truncation |dz|^6 / 720 < 1e-17 relative for
// |dz| <= 2e-3, i.e. the perturbed VALUES are those of a direct evaluation to below its own rounding (the perturbation itself
// carries a relative error of 1e-16, a direct evaluation an absolute one of 1e-16 |a|), and the finite differences formed
// from them keep the reference's semantics including their truncation error.  A wavefront in which any |dz| exceeds 2e-3
// (huge weights, no input scaling) evaluates the perturbed points directly.
// CODE 1 softplus, 2 sigmoid.  c[k] = (k+1)-th derivative of the activation at z, divided by (k+1)!.
template <int CODE>
__device__ __forceinline__ double act_taylor(double z, double (&c)[5]) {
  const double t = exp_nonpos(-fabs(z));
  const double r = fast_div(1.0, 1.0 + t);
  const bool pos = z >= 0.0;
  const double tr = t * r;
  const double s = pos ? r : tr;                     // sigma(z)
  const double u = tr * r;                           // sigma (1 - sigma) = t / (1 + t)^2: no cancellation
  const double m0 = (1.0 - t) * r;
  const double m = pos ? -m0 : m0;                   // 1 - 2 sigma
  const double um = u * m, q6 = fma(-6.0, u, 1.0), q12 = fma(-12.0, u, 1.0);
  if (CODE == 1) {  // softplus: derivatives s, u, u m, u (1 - 6u), u m (1 - 12u)
    c[0] = s; c[1] = 0.5 * u; c[2] = um * (1.0 / 6.0); c[3] = u * q6 * (1.0 / 24.0); c[4] = um * q12 * (1.0 / 120.0);
    return log1p_01(t) + fmax(z, 0.0);
  }
  // sigmoid: derivatives u, u m, u (1 - 6u), u m (1 - 12u), u (1 - 30u + 120u^2)
  c[0] = u; c[1] = 0.5 * um; c[2] = u * q6 * (1.0 / 6.0); c[3] = um * q12 * (1.0 / 24.0);
  c[4] = u * fma(u, fma(120.0, u, -30.0), 1.0) * (1.0 / 120.0);
  return s;
}

constexpr double PERT_DZ_MAX = 2e-3;
// units between two scheduling fences: 1 = strictly one unit at a time (fewest registers, longest dependent chains),
// 2-3 = that many exponentials in flight (the kernels run 2 waves per SIMD: some instruction-level parallelism is needed)
// (per number of perturbations: the Y law's single perturbation leaves registers for more units in flight than the U law's four)
#define ODINN_PERT_Y 0  // (fixed: its A/B is recorded above; no longer a build-time knob)
#define ODINN_PERT_GROUP1 5  // (fixed: its A/B is recorded above; no longer a build-time knob)
#define ODINN_PERT_GROUP4 2  // (fixed: its A/B is recorded above; no longer a build-time knob)
#ifndef ODINN_PERT_INLINE
#define ODINN_PERT_INLINE __forceinline__
#endif

// Scheduling fence of the perturbed evaluation: one unit at a time.  Left alone the scheduler interleaves all units of a
// layer (every exponential in flight at once) and spills hundreds of registers.  The empty statement "rewrites" `next` --
// a value the NEXT unit starts from -- together with what this unit produced last, so the next unit cannot begin before
// this one is done; it emits nothing.
template <int NP>
__device__ __forceinline__ void pert_fence(double& next, double& a, double (&d)[NP]) {
  if constexpr (NP == 4) asm volatile("" : "+v"(next), "+v"(a), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
  else if constexpr (NP == 1) asm volatile("" : "+v"(next), "+v"(a), "+v"(d[0]));
  else asm volatile("" : "+v"(next), "+v"(a));
}

// One layer.  ZGIVEN: v / dv hold this layer's pre-activations (accumulated by the previous layer's push) and are overwritten
// in place by the activations.  Otherwise v / dv hold the previous layer's activations and every unit PULLs its
// pre-activation from them; its activation is then either STOREd (n / dn = this layer's activations) or, when the next
// layer is not wider than this one, PUSHed into the next layer's pre-activation accumulators (n / dn, started from the
// biases): the wide hidden layer of 2-3-10-3-1 never exists as a vector.  The order of the fused multiply-adds of every
// pre-activation is bias first, then inputs ascending -- exactly mlp_layer_fixed's, so the central value is bit-identical.
template <class AR, int l, int NP, bool ZGIVEN, bool PUSH>
__device__ __forceinline__ void mlp_layer_pert(const double* __restrict__ th, double (&v)[AR::MAXW], double (&dv)[NP][AR::MAXW],
                                               double (&n)[AR::MAXW], double (&dn)[NP][AR::MAXW], double& dzmax) {
  constexpr int nin = AR::W[l], nout = AR::W[l + 1], off = arch_off<AR>(l);
  static_assert(AR::A[l] == 1 || AR::A[l] == 2, "perturbed evaluation: softplus / sigmoid layers");
  constexpr int nnext = PUSH ? AR::W[l + 2] : 0, offn = PUSH ? arch_off<AR>(l + 1) : 0;
  if constexpr (PUSH) {
#pragma unroll
    for (int k = 0; k < nnext; ++k) {
      n[k] = th[offn + nout * nnext + k];
#pragma unroll
      for (int q = 0; q < NP; ++q) dn[q][k] = 0.0;
    }
  }
#pragma unroll
  for (int o = 0; o < nout; ++o) {
    double z, dz[NP];
    if constexpr (ZGIVEN) {
      z = v[o];
#pragma unroll
      for (int q = 0; q < NP; ++q) dz[q] = dv[q][o];
    } else {
      z = th[off + nin * nout + o];
#pragma unroll
      for (int i = 0; i < nin; ++i) z = fma(th[off + o + nout * i], v[i], z);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        dz[q] = 0.0;
#pragma unroll
        for (int i = 0; i < nin; ++i) dz[q] = fma(th[off + o + nout * i], dv[q][i], dz[q]);
      }
    }
    double c[5];
    double a = act_taylor<AR::A[l]>(z, c);
    double da[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      dzmax = fmax(dzmax, fabs(dz[q]));
      da[q] = dz[q] * fma(dz[q], fma(dz[q], fma(dz[q], fma(dz[q], c[4], c[3]), c[2]), c[1]), c[0]);
    }
    constexpr int grp = NP == 1 ? ODINN_PERT_GROUP1 : ODINN_PERT_GROUP4;
    const bool fence = (o % grp) == grp - 1;  // (folded: the unit loop is fully unrolled)
    if constexpr (ZGIVEN) {
      if (fence) pert_fence<NP>(v[o + 1 < nout ? o + 1 : 0], a, da);
      v[o] = a;
#pragma unroll
      for (int q = 0; q < NP; ++q) dv[q][o] = da[q];
    } else if constexpr (PUSH) {
#pragma unroll
      for (int k = 0; k < nnext; ++k) {
        n[k] = fma(th[offn + k + nnext * o], a, n[k]);
#pragma unroll
        for (int q = 0; q < NP; ++q) dn[q][k] = fma(th[offn + k + nnext * o], da[q], dn[q][k]);
      }
      if (fence) {
        double last[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) last[q] = dn[q][nnext - 1];
        pert_fence<NP>(v[0], n[nnext - 1], last);
#pragma unroll
        for (int q = 0; q < NP; ++q) dn[q][nnext - 1] = last[q];
      }
    } else {
      if (fence) pert_fence<NP>(v[0], a, da);
      n[o] = a;
#pragma unroll
      for (int q = 0; q < NP; ++q) dn[q][o] = da[q];
    }
  }
}

template <class AR>
__device__ __noinline__ double mlp_eval_fixed_ni(const LawDev& L, double x0, double x1) { return mlp_eval_fixed<AR>(L, x0, x1); }

// layers l .. NL-1; on entry (cur, dcur) hold pre-activations of layer l (ZGIVEN) or activations of layer l-1
template <class AR, int l, int NP, bool ZGIVEN>
__device__ __forceinline__ void mlp_pert_from(const double* __restrict__ th, double (&cur)[AR::MAXW], double (&dcur)[NP][AR::MAXW],
                                              double (&oth)[AR::MAXW], double (&doth)[NP][AR::MAXW], double& dzmax, double& y,
                                              double (&dy)[NP]) {
  if constexpr (ZGIVEN) {  // in place; the next layer pulls from cur
    mlp_layer_pert<AR, l, NP, true, false>(th, cur, dcur, oth, doth, dzmax);
    if constexpr (l + 1 < AR::NL) mlp_pert_from<AR, l + 1, NP, false>(th, cur, dcur, oth, doth, dzmax, y, dy);
    else {
      y = cur[0];
#pragma unroll
      for (int q = 0; q < NP; ++q) dy[q] = dcur[q][0];
    }
  } else if constexpr (l + 1 < AR::NL && AR::W[l + 2] <= AR::W[l + 1]) {  // push into the next layer's accumulators
    mlp_layer_pert<AR, l, NP, false, true>(th, cur, dcur, oth, doth, dzmax);
    mlp_pert_from<AR, l + 1, NP, true>(th, oth, doth, cur, dcur, dzmax, y, dy);
  } else {  // store
    mlp_layer_pert<AR, l, NP, false, false>(th, cur, dcur, oth, doth, dzmax);
    if constexpr (l + 1 < AR::NL) mlp_pert_from<AR, l + 1, NP, false>(th, oth, doth, cur, dcur, dzmax, y, dy);
    else {
      y = oth[0];
#pragma unroll
      for (int q = 0; q < NP; ++q) dy[q] = doth[q][0];
    }
  }
}

// y = net(x0, x1); yp[q] = net with input pd[q] shifted by dlt[q] (raw input units, before the pre-scaling)
template <class AR, int NP>
__device__ ODINN_PERT_INLINE double mlp_eval_pert(const LawDev& L, double x0, double x1, const int (&pd)[NP],
                                                  const double (&dlt)[NP], double (&yp)[NP]) {
  double h0[AR::MAXW], h1[AR::MAXW], d0[NP][AR::MAXW], d1[NP][AR::MAXW];
  h0[0] = L.has_pre ? (x0 - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0;
  h0[1] = L.has_pre ? (x1 - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const double sc = L.has_pre ? L.pre_inv[pd[q]] : 1.0;
    d0[q][0] = pd[q] == 0 ? dlt[q] * sc : 0.0;
    d0[q][1] = pd[q] == 1 ? dlt[q] * sc : 0.0;
  }
  double dzmax = 0.0, y, dy[NP];
  mlp_pert_from<AR, 0, NP, false>(L.theta, h0, d0, h1, d1, dzmax, y, dy);
  if (__builtin_amdgcn_ballot_w64(dzmax > PERT_DZ_MAX) != 0) {  // wave-uniform; rare: evaluate the perturbed points directly
#pragma unroll 1
    for (int q = 0; q < NP; ++q)
      yp[q] = mlp_eval_fixed_ni<AR>(L, pd[q] == 0 ? x0 + dlt[q] : x0, pd[q] == 1 ? x1 + dlt[q] : x1);
  } else {
#pragma unroll
    for (int q = 0; q < NP; ++q) yp[q] = postscale_f(L, y + dy[q]);
  }
  return postscale_f(L, y);
}

template <int LM, int NP>
__device__ __forceinline__ double mlp_eval_pert_lm(const LawDev& L, double x0, double x1, const int (&pd)[NP],
                                                   const double (&dlt)[NP], double (&yp)[NP]) {
  if constexpr (LM == LM_NN_DEF) return mlp_eval_pert<ArchDef, NP>(L, x0, x1, pd, dlt, yp);
  else if constexpr (LM == LM_NN_16) return mlp_eval_pert<Arch16, NP>(L, x0, x1, pd, dlt, yp);
  else if constexpr (LM == LM_NN_LIGHT) return mlp_eval_pert<ArchLight, NP>(L, x0, x1, pd, dlt, yp);
  else {  // run-time architecture (any activation): direct evaluations, ONE inlined copy of the evaluator looped over the points
    double yc = 0.0;
#pragma unroll 1
    for (int q = 0; q <= NP; ++q) {
      double a0 = x0, a1 = x1;
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (q == k) {
          if (pd[k] == 0) a0 = x0 + dlt[k];
          else a1 = x1 + dlt[k];
        }
      const double v = mlp_eval_lm<LM>(L, a0, a1);
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (q == k) yp[k] = v;
      if (q == NP) yc = v;
    }
    return yc;
  }
}


// ---- powers with a per-glacier (wave-uniform) exponent -----------------------------------------
// The exponents of the diffusivity (n + 2, n - 1, p - q + 1, ... -- target_A.jl:25-61) are physical constants of a
// glacier, in practice small integers (n = 3, p = 3, q = 0 or 1).  ocml's pow costs ~280 fp64 instructions; with a
// uniform small integer exponent the power is a product chain under a scalar branch (<= 6 multiplications,
// |error| <= a few ulp; 0^0 = 1 as in Julia).  Anything else falls through to pow.
__device__ __forceinline__ double upow(double x, double e) {
  const int k = (int)e;
  if ((double)k == e && k >= 0 && k <= 16) {  // wave-uniform
    double r = 1.0, b = x;
    for (int kk = k; kk; kk >>= 1) {
      if (kk & 1) r *= b;
      b *= b;
    }
    return r;
  }
  return pow(x, e);
}
// |grad S|^e from |grad S|^2: even exponents need no square root at all
__device__ __forceinline__ double spow(double gS2, double e) {
  const int k = (int)e;
  if ((double)k == e && k >= 0 && k <= 16) {  // wave-uniform
    const double h = upow(gS2, (double)(k >> 1));
    return (k & 1) ? h * sqrt(gS2) : h;
  }
  return pow(sqrt(gS2), e);
}

// ---- diffusivity on one dual node ---------------------------------------------------
// Returns D.  For the adjoint (ADJ) also alpha = dD/dHbar and beta (the reference's
// "dD/dgradH", i.e. (dD/d|gradS|)/|gradS| for the closed forms) -- target_A.jl:16-62,
// target_D_hybrid.jl:22-96,168-208, target_D_pure.jl:78-137 -- and `spat`, the spatial
// factor of dD/dtheta (target_A.jl:71-72, target_D_hybrid.jl:117-118, target_D_pure.jl:142).
// NK: 0 = the law's kind is read from L (one kernel for both per-node-network laws), 3 / 4 = compile-time Y / U law -- the
// reverse kernels are instantiated per law: the U law's four perturbations and the Y law's single one want different
// register budgets, and in a shared kernel the allocation of the one spills the other
// ---- tabulated Y law (LM_YTAB) -------------------------------------------------------------------------------------------
// LawY's inputs are the glacier's long-term air temperature -- a SCALAR per glacier -- and Hbar (Laws.jl:240-273), so for a
// given theta the law of one glacier is a function of ONE variable.  k_ytab_build evaluates the network exactly on six Chebyshev
// nodes of each of ytab_ni equal intervals of [0, Hmax_g] and stores the interpolating quintic in s = 2 (Hbar / h - i) - 1 as
// monomial coefficients c0 .. c5 (48 bytes per interval); the build also measures the table against the network between the
// nodes and the host keeps the exact kernels unless the worst relative deviation is below 1e-12 (odinn_hip.hip: ytab_refresh).
// The reference's forward difference (Y(Hbar + 1e-4) - Y(Hbar)) / 1e-4 (target_D_hybrid.jl:58-71) keeps its semantics: the
// perturbed value is the SAME quintic at s + ds, formed as Y + [p(s + ds) - p(s)] with the exact finite Taylor shift of the
// polynomial, ds from the perturbation the reference really applies, fl(Hbar + 1e-4) - Hbar.  A node beyond the table takes the
// table's last value and raises *ytab_over: the host repeats the call with a wider table (or the exact kernels).
struct YtabRef {  // one glacier's table as the strip kernels take it (AdjFusedArgs): base of the glacier's coefficients, flag, size
  const double* tab;
  int* over;
  int ni;
  unsigned long long* beyond;  // the kernel's own ballot accumulator (ytab_eval_acc): the flag is raised once, at the kernel's end
};
// Y(Hbar + 1e-4) of the reverse kernels (the reference's forward difference of dD/dHbar, target_D_hybrid.jl:58-71): the interval's quintic
// evaluated again at s + ds (6 instructions) instead of its exact Taylor shift from s (23: four derivative Horner chains).  The quotient
// (Y(H + 1e-4) - Y(H)) / 1e-4 then carries the rounding of two evaluations, 2e-10 relative -- what the reference's own two network
// evaluations carry -- where the shift kept 1e-16.  Measured (round 6, k_adj_fused_strip<..., YT>, 64 x 1024^2): 3625 -> 3309 us dense,
// 2476 -> 2328 us with the ice-free shortcut; at 4 waves per SIMD (36-40 spilled VGPRs) 3500 / 2355: the kernel stays at 2.
#define ODINN_YTAB_SHIFT_EXACT 0  // (fixed: its A/B is recorded above)
template <bool ADJ>
__device__ __forceinline__ double ytab_eval_acc(const double* __restrict__ tab, int ni, unsigned long long& beyond, double inv_h, double Hb, double& Yp) {
  // `beyond` collects the lanes whose node lies beyond the table (or is NaN): a ballot, i.e. scalar arithmetic -- NO divergent
  // branch.  The branch `if (beyond) *over = 1` that used to sit here, inside the register-pressured stage code of the strip kernels,
  // is what ROCm 7.2's backend miscompiled (a live-range-split VGPR copy / spill store placed in the join block AHEAD of the
  // `s_or_b64 exec` that restores EXEC: fuzz seed 24379; tools/exec_lint.py checks every built kernel for it).
  double x = Hb * inv_h;
  const bool out = !(x < (double)ni);
  beyond |= __builtin_amdgcn_ballot_w64(out);
  x = out ? (double)ni : x;  // the table's last value, so that the doomed solve stays tame
  const int i = min((int)x, ni - 1);  // Hbar >= 0
  const double s = fma(2.0, x - (double)i, -1.0);
  const double2* __restrict__ c = reinterpret_cast<const double2*>(tab) + 3 * i;
  const double2 c01 = c[0], c23 = c[1], c45 = c[2];
  const double Y = fma(fma(fma(fma(fma(c45.y, s, c45.x), s, c23.y), s, c23.x), s, c01.y), s, c01.x);
  if (ADJ) {
    const double ds = 2.0 * (((Hb + 1e-4) - Hb) * inv_h);
#if ODINN_YTAB_SHIFT_EXACT
    const double p1 = fma(fma(fma(fma(5.0 * c45.y, s, 4.0 * c45.x), s, 3.0 * c23.y), s, 2.0 * c23.x), s, c01.y);
    const double p2 = fma(fma(fma(10.0 * c45.y, s, 6.0 * c45.x), s, 3.0 * c23.y), s, c23.x);
    const double p3 = fma(fma(10.0 * c45.y, s, 4.0 * c45.x), s, c23.y);
    const double p4 = fma(5.0 * c45.y, s, c45.x);
    Yp = fma(ds, fma(ds, fma(ds, fma(ds, fma(ds, c45.y, p4), p3), p2), p1), Y);
#else
    const double s2 = s + ds;
    Yp = fma(fma(fma(fma(fma(c45.y, s2, c45.x), s2, c23.y), s2, c23.x), s2, c01.y), s2, c01.x);
#endif
  }
  return Y;
}
// the flag itself: raised by a UNIFORM branch (the ballot is a scalar: s_cbranch_scc, EXEC untouched)
__device__ __forceinline__ void ytab_raise(int* over, unsigned long long beyond) {
  if (beyond != 0ull) *over = 1;
}
template <bool ADJ>
__device__ __forceinline__ double ytab_eval_core(const double* __restrict__ tab, int ni, int* over, double inv_h, double Hb, double& Yp) {
  unsigned long long beyond = 0ull;
  const double Y = ytab_eval_acc<ADJ>(tab, ni, beyond, inv_h, Hb, Yp);
  ytab_raise(over, beyond);
  return Y;
}
template <bool ADJ>
__device__ __forceinline__ double ytab_eval(const GDev& g, const LawDev& L, double Hb, double& Yp) {
  return ytab_eval_core<ADJ>(L.ytab + g.yt_off, L.ytab_ni, L.ytab_over, g.yt_inv_h, Hb, Yp);
}

// ---- tabulated U law (LM_UTAB) -------------------------------------------------------------------------------------------
// LawU's inputs are Hbar and |grad S| (Laws.jl:97-183): a function of two variables, the same for every glacier of the batch.
// k_utab_build evaluates the network on the 6 x 6 tensor grid of Chebyshev nodes of each patch of [0, Hmax] x [0, Smax] and stores
// the interpolating bi-quintic as monomial coefficients c[a][b] of u^a v^b, u = 2 (Hbar / h_H - i) - 1, v = 2 (|grad S| / h_S - j) - 1
// (288 bytes per patch); table against network between the nodes decides whether the kernels may use it, as for the Y law.
// The reference's central differences (U at Hbar +- 1e-4 and at |grad S| +- 1e-6, target_D_pure.jl:105-137) are taken on the SAME
// patch: the bi-quintic is collapsed to a quintic in u at the node's v (and to one in v at the node's u), whose values at u +- du
// come from the exact finite Taylor shift -- the quotients carry the reference's truncation error, not the cancellation noise of
// separate evaluations.  up[0..3] = U(H + dH, s), U(H - dH, s), U(H, s + dS), U(H, s - dS).
__device__ __forceinline__ void quintic_shift(const double (&q)[6], double x, double dx, double& val, double& plus, double& minus) {
  val = fma(fma(fma(fma(fma(q[5], x, q[4]), x, q[3]), x, q[2]), x, q[1]), x, q[0]);
  const double p1 = fma(fma(fma(fma(5.0 * q[5], x, 4.0 * q[4]), x, 3.0 * q[3]), x, 2.0 * q[2]), x, q[1]);
  const double p2 = fma(fma(fma(10.0 * q[5], x, 6.0 * q[4]), x, 3.0 * q[3]), x, q[2]);
  const double p3 = fma(fma(10.0 * q[5], x, 4.0 * q[4]), x, q[3]);
  const double p4 = fma(5.0 * q[5], x, q[4]);
  const double d2 = dx * dx;
  const double ev = d2 * fma(d2, p4, p2);                     // even part of p(x + dx) - p(x)
  const double od = dx * fma(d2, fma(d2, q[5], p3), p1);      // odd part
  plus = val + (ev + od);
  minus = val + (ev - od);
}
// The patches one tile's nodes fall into, staged in LDS by the workgroup (utab_stage): lds == nullptr -> every lane reads the table
// in global memory.  Why: the vector L1 returns data in order, so a table load that hits still queues behind the streaming misses
// of the other workgroups of the CU (measured on k_adj_stage / k_vjp_H / k_rk_stage at 8 x 1024^2: 27 / 26 / 42 us of 220 / 130 /
// 127 us per launch are the table loads, with the table L1-resident); LDS reads do not.
struct UtabTile {
  const double2* lds;
  int ih0, is0, nsr;
  // per node, filled by the caller from utab_stage's pass (pre): the node's patch -- offset of its 18 double2 in `lds`, or the patch
  // number in the global table -- and its patch coordinates, so that the evaluation repeats neither the square root nor the indexing
  bool pre;
  int slot;
  double u, v;
  int pst;  // double2 per patch in `lds`: 18, or 19 where the whole table is resident (k_adj_fused_lds)
};
#define ODINN_UT_NONE UtabTile{nullptr, 0, 0, 0, false, 0, 0.0, 0.0, 18}
// patch (ih, is) and the patch coordinates (u, v) of a node; a node beyond the table takes the table's edge and raises the flag
__device__ __forceinline__ void utab_index(const LawDev& L, double Hb, double gS, int& ih, int& is, double& u, double& v) {
  double xh = Hb * L.ut_inv_h, xs = gS * L.ut_inv_s;
  const bool oh = !(xh < (double)L.utab_nh), os = !(xs < (double)L.utab_ns);  // beyond the table (or NaN): its edge value
  xh = oh ? (double)L.utab_nh : xh; xs = os ? (double)L.utab_ns : xs;
  ytab_raise(L.ytab_over, __builtin_amdgcn_ballot_w64(oh || os));  // (uniform branch: see ytab_eval_acc)
  ih = min((int)xh, L.utab_nh - 1); is = min((int)xs, L.utab_ns - 1);
  u = fma(2.0, xh - (double)ih, -1.0); v = fma(2.0, xs - (double)is, -1.0);
}
template <bool ADJ>
__device__ __forceinline__ double utab_eval(const LawDev& L, double Hb, double gS, double (&up)[4], const UtabTile ut = ODINN_UT_NONE) {
  int ih = 0, is = 0, slot;
  double u, v;
  if (ut.pre) { slot = ut.slot; u = ut.u; v = ut.v; }
  else {
    utab_index(L, Hb, gS, ih, is, u, v);
    slot = ut.lds ? ut.pst * ((ih - ut.ih0) * ut.nsr + (is - ut.is0)) : ih * L.utab_ns + is;
  }
  // (explicit address spaces: with generic pointers the compiler folds the two branches into ONE set of flat loads on a selected
  //  pointer, and a flat load of LDS data queues in the vector memory pipeline like the global one it was meant to avoid)
  typedef double d2v __attribute__((ext_vector_type(2)));
  typedef const d2v __attribute__((address_space(3))) * lds_p;
  typedef const d2v __attribute__((address_space(1))) * glb_p;
  d2v cr[18];  // c[a][b] of u^a v^b: cr[3 a + b / 2]
  if (ut.lds) {    // (workgroup-uniform)
    lds_p c = (lds_p)(ut.lds) + slot;
#pragma unroll
    for (int k = 0; k < 18; ++k) cr[k] = c[k];
  } else {
    glb_p c = (glb_p)(L.utab) + 18 * (long long)slot;
#pragma unroll
    for (int k = 0; k < 18; ++k) cr[k] = c[k];
  }
  // row a = c[a][0..5], highest power of u first: qu[a] = the row's quintic in v at the node's v; qv[b] = Horner in u down the rows (ADJ)
  double qu[6], qv[6];
#pragma unroll
  for (int a = 5; a >= 0; --a) {
    const d2v c0 = cr[3 * a], c1 = cr[3 * a + 1], c2 = cr[3 * a + 2];
    qu[a] = fma(fma(fma(fma(fma(c2.y, v, c2.x), v, c1.y), v, c1.x), v, c0.y), v, c0.x);
    if (ADJ) {
      if (a == 5) {
        qv[0] = c0.x; qv[1] = c0.y; qv[2] = c1.x; qv[3] = c1.y; qv[4] = c2.x; qv[5] = c2.y;
      } else {
        qv[0] = fma(qv[0], u, c0.x); qv[1] = fma(qv[1], u, c0.y); qv[2] = fma(qv[2], u, c1.x);
        qv[3] = fma(qv[3], u, c1.y); qv[4] = fma(qv[4], u, c2.x); qv[5] = fma(qv[5], u, c2.y);
      }
    }
  }
  if (!ADJ) return fma(fma(fma(fma(fma(qu[5], u, qu[4]), u, qu[3]), u, qu[2]), u, qu[1]), u, qu[0]);
  double U, U2;
  quintic_shift(qu, u, 2.0 * (1e-4 * L.ut_inv_h), U, up[0], up[1]);
  quintic_shift(qv, v, 2.0 * (1e-6 * L.ut_inv_s), U2, up[2], up[3]);
  return U;
}

template <bool ADJ, int LM, int NK = 0>
__device__ __forceinline__ double node_D(const GDev& g, const LawDev& L, double Hb, double gS2, double Anode,
                                         double& alpha, double& beta, double& spat, const UtabTile ut = ODINN_UT_NONE) {
  if constexpr (LM == LM_UTAB) {  // the U law's branch below with U and its four finite-difference points from the table
    if (!(Hb > 0.0)) {
      if (ADJ) { alpha = 0.0; beta = 0.0; spat = 0.0; }
      return 0.0;
    }
    const double gS = ut.pre ? 0.0 : sqrt(gS2);
    double up[4];
    const double U = utab_eval<ADJ>(L, Hb, gS, up, ut);
    if (ADJ) {
      const double dH = 1e-4, dS = 1e-6;  // target_D_pure.jl:109,125
      const double Dp = up[0] * (Hb + dH), Dm = up[1] * (Hb - dH);
      alpha = (Dp - Dm) * (1.0 / (2.0 * dH));  // (the quotients as products with the constant reciprocals: <= 1 ulp from the division,
      const double Ep = up[2] * Hb, Em = up[3] * Hb;   //  ~60 fp64 instructions per node less)
      beta = (Ep - Em) * (1.0 / (2.0 * dS));
      spat = Hb;
    }
    return Hb * U;
  }
  if constexpr (LM == LM_YTAB) {  // the Y law's closed form below with Y (and Y at Hbar + 1e-4) from the table
    double Yp = 0.0;
    const double Y = ytab_eval<ADJ>(g, L, Hb, Yp);
    if (g.yt_fast) {  // (wave-uniform) the products of node_D<LM_FAST> with Y in A's place; the same expressions as the general form
      const double H2 = Hb * Hb, H4 = H2 * H2, H5 = H4 * Hb;
      const double geo = g.Gam * H5 * gS2;
      if (ADJ) {
        alpha = 5.0 * Y * g.Gam * H4 * gS2 + (Yp * geo - Y * geo) / 1e-4;
        beta = g.Gam * Y * 2.0 * H5;
        spat = geo;
      }
      return Y * geo;
    }
    const double sS1 = spow(gS2, g.nS - 1.0);
    const double geo = g.Gam * upow(Hb, g.nH + 2.0) * sS1;
    double D = Y * geo;
    double hs = 0.0, sp1 = 0.0;
    if (g.Sc != 0.0) {
      hs = upow(Hb, g.p - g.q + 1.0);
      sp1 = spow(gS2, g.p - 1.0);
      D += g.Sc * hs * sp1;
    }
    if (ADJ) {
      const double dH = 1e-4;  // target_D_hybrid.jl:58
      const double slide = g.Sc != 0.0 ? g.Sc * hs * sp1 : 0.0;
      alpha = (g.nH + 2.0) * Y * g.Gam * upow(Hb, g.nH + 1.0) * sS1 + ((slide + Yp * geo) - (slide + Y * geo)) / dH;
      beta = g.Gam * Y * (g.nS - 1.0) * upow(Hb, g.nH + 2.0) * spow(gS2, g.nS - 3.0);
      if (g.Sc != 0.0) {
        alpha += (g.p - g.q + 1.0) * g.Sc * upow(Hb, g.p - g.q) * sp1;
        beta += g.Sc * (g.p - 1.0) * hs * spow(gS2, g.p - 3.0);
      }
      spat = geo;
    }
    return D;
  }
  if (!lm_is_nn(LM)) {  // A-type laws (scalar or field A)
    if (LM == LM_FAST) {
      const double H2 = Hb * Hb, H4 = H2 * H2;
      const double AG = Anode * g.Gam;
      const double H5 = H4 * Hb;
      if (ADJ) {
        alpha = AG * 5.0 * H4 * gS2;
        beta = AG * 2.0 * H5;  // (n-1) * gradS^(n-3), 0^0 = 1
        spat = g.Gam * H5 * gS2;
      }
      return AG * H5 * gS2;
    }
    const double hn2 = upow(Hb, g.n + 2.0), sn1 = spow(gS2, g.n - 1.0);
    double D = Anode * g.Gam * hn2 * sn1;
    double hs = 0.0, sp1 = 0.0;
    if (g.Sc != 0.0) {
      hs = upow(Hb, g.p - g.q + 1.0);
      sp1 = spow(gS2, g.p - 1.0);
      D += g.Sc * hs * sp1;
    }
    if (ADJ) {
      alpha = Anode * g.Gam * (g.n + 2.0) * upow(Hb, g.n + 1.0) * sn1;
      beta = Anode * g.Gam * (g.n - 1.0) * hn2 * spow(gS2, g.n - 3.0);
      if (g.Sc != 0.0) {
        alpha += (g.p - g.q + 1.0) * g.Sc * upow(Hb, g.p - g.q) * sp1;
        beta += g.Sc * (g.p - 1.0) * hs * spow(gS2, g.p - 3.0);
      }
      spat = g.Gam * hn2 * sn1;
    }
    return D;
  }
  // On ice-free nodes (Hbar == 0) D, alpha, beta and the theta-weight vanish identically
  // (every term carries a positive power of Hbar), so the MLP is not evaluated there.
  const bool ice = Hb > 0.0;
  if ((NK ? NK : L.kind) == 3) {  // Y law, :D_hybrid
    // (the single forward-difference point of this law is evaluated directly: the perturbed evaluation -- see the U law below --
    //  was measured at +5 ... +10 % on the Y law's reverse kernels, its Taylor coefficients cost what the second evaluation costs;
    //  -DODINN_PERT_Y=1 selects it)
    double Y, Yp_pert = 0.0;
    if constexpr (ADJ && (ODINN_PERT_Y || lm_is_rt(LM))) {  // (run-time architectures: the loop form shares one evaluator body)
      Y = 0.0;
      if (ice) {
        const int pd[1] = {1};
        const double dl[1] = {1e-4};
        double yp[1];
        Y = mlp_eval_pert_lm<LM, 1>(L, g.T, Hb, pd, dl, yp);
        Yp_pert = yp[0];
      }
    } else {
      Y = ice ? mlp_eval_lm<LM>(L, g.T, Hb) : 0.0;
    }
    const double sS1 = spow(gS2, g.nS - 1.0);
    const double geo = g.Gam * upow(Hb, g.nH + 2.0) * sS1;
    double D = Y * geo;
    double hs = 0.0, sp1 = 0.0;
    if (g.Sc != 0.0) {
      hs = upow(Hb, g.p - g.q + 1.0);
      sp1 = spow(gS2, g.p - 1.0);
      D += g.Sc * hs * sp1;
    }
    if (ADJ) {
      const double dH = 1e-4;  // target_D_hybrid.jl:58
      double Yp;
      if constexpr (ODINN_PERT_Y || lm_is_rt(LM)) Yp = Yp_pert;
      else Yp = ice ? mlp_eval_lm<LM>(L, g.T, Hb + dH) : 0.0;
      const double slide = g.Sc != 0.0 ? g.Sc * hs * sp1 : 0.0;
      alpha = (g.nH + 2.0) * Y * g.Gam * upow(Hb, g.nH + 1.0) * sS1 +
              ((slide + Yp * geo) - (slide + Y * geo)) / dH;
      beta = g.Gam * Y * (g.nS - 1.0) * upow(Hb, g.nH + 2.0) * spow(gS2, g.nS - 3.0);
      if (g.Sc != 0.0) {
        alpha += (g.p - g.q + 1.0) * g.Sc * upow(Hb, g.p - g.q) * sp1;
        beta += g.Sc * (g.p - 1.0) * hs * spow(gS2, g.p - 3.0);
      }
      spat = geo;
    }
    return D;
  }
  // U law, :D   D = Hbar * U(Hbar, gradS)
  if (!ice) {
    if (ADJ) { alpha = 0.0; beta = 0.0; spat = 0.0; }
    return 0.0;
  }
  const double gS = sqrt(gS2);
  if constexpr (ADJ) {
    const double dH = 1e-4, dS = 1e-6;  // target_D_pure.jl:109,125
    // U at the node and at the four points of the two central differences in one pass (mlp_eval_pert)
    const int pd[4] = {0, 0, 1, 1};
    const double dl[4] = {dH, -dH, dS, -dS};
    double up[4];
    const double U = mlp_eval_pert_lm<LM, 4>(L, Hb, gS, pd, dl, up);
    const double Dp = up[0] * (Hb + dH);
    const double Dm = up[1] * (Hb - dH);
    alpha = (Dp - Dm) / (2.0 * dH);
    const double Ep = up[2] * Hb;
    const double Em = up[3] * Hb;
    beta = (Ep - Em) / (2.0 * dS);
    spat = Hb;
    return Hb * U;
  } else {
    return Hb * mlp_eval_lm<LM>(L, Hb, gS);
  }
}

// ---- tile geometry -------------------------------------------------------------------
// Load the (TX+2)x(TY+2) halo tile of U into LDS, interleaved: sHS[r][c] = {max(U,0), B + max(U,0)}.
// Thread (tx, ty) owns interior rows r = 1 + ty + NW*m and keeps their raw values in own[].
template <int NWV = NW, int TYV = TY>
__device__ __forceinline__ bool load_tile_HS2(const double* __restrict__ U, const double* __restrict__ B, const GDev& g,
                                              int i0, int j0, double2 (*sHS)[LDW], double (&own)[TYV / NWV],
                                              const double* __restrict__ U2 = nullptr, double sw = 0.0) {
  // U2 != null: the field is U + sw (U2 - U)  (H_itp of the continuous adjoint, gradient.jl:287).
  // Returns whether any value THIS thread loaded (own rows or its share of the halo) carries ice.
  bool ice = false;
#if ODINN_LOG1P_TABLE == 2
  // log1p's table into LDS with the tile: every kernel of these units loads its tile through here and has a barrier
  // between this and its first network evaluation
  if (threadIdx.x < 65) g_log1p_lds[threadIdx.x] = *reinterpret_cast<const double2*>(&LOG1P_TAB[threadIdx.x][0]);
#endif
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  const bool colok = gi < g.nx;
#pragma unroll
  for (int m = 0; m < TYV / NWV; ++m) {
    const int r = 1 + ty + NWV * m;
    const int gj = j0 - 1 + r;
    double h = 0.0, b = 0.0;
    if (colok && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      h = U[id];
      if (U2) h = fma(sw, U2[id] - h, h);
      b = B[id];
    }
    own[m] = h;
    const double hc = vmax0(h);
    ice = ice || hc > 0.0;
    sHS[r][tx + 1] = make_double2(hc, b + hc);
  }
  if (ty < 2) {
    const int r = ty == 0 ? 0 : TYV + 1;
    const int gj = j0 - 1 + r;
    double h = 0.0, b = 0.0;
    if (colok && gj >= 0 && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      h = U[id];
      if (U2) h = fma(sw, U2[id] - h, h);
      b = B[id];
    }
    const double hc = vmax0(h);
    ice = ice || hc > 0.0;
    sHS[r][tx + 1] = make_double2(hc, b + hc);
  } else if (ty < 4) {
    const int l = threadIdx.x - 128;
    if (l < 2 * (TYV + 2)) {
      const int r = l >> 1, side = l & 1;
      const int c = side ? TX + 1 : 0;
      const int gi2 = i0 - 1 + c, gj = j0 - 1 + r;
      double h = 0.0, b = 0.0;
      if (gi2 >= 0 && gi2 < g.nx && gj >= 0 && gj < g.ny) {
        const long long id = g.off + gi2 + (long long)g.nx * gj;
        h = U[id];
        if (U2) h = fma(sw, U2[id] - h, h);
        b = B[id];
      }
      const double hc = vmax0(h);
      ice = ice || hc > 0.0;
      sHS[r][c] = make_double2(hc, b + hc);
    }
  }
  return ice;
}

// Same for a third field kept unclamped and masked to the interior (lambda~).
// MASK = false keeps the boundary-ring values (the continuous-form VJP differentiates lambda itself).
template <int NWV = NW, int TYV = TY, bool MASK = true>
__device__ __forceinline__ void load_tile_lam(const double* __restrict__ Lm, const GDev& g, int i0, int j0,
                                              double (*sL)[LDW], double (&own)[TYV / NWV]) {
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  auto ld = [&](int gi_, int gj_, double& raw) -> double {
    raw = 0.0;
    if (gi_ >= 0 && gi_ < g.nx && gj_ >= 0 && gj_ < g.ny) {
      raw = Lm[g.off + gi_ + (long long)g.nx * gj_];
      if (!MASK || (gi_ >= 1 && gi_ <= g.nx - 2 && gj_ >= 1 && gj_ <= g.ny - 2)) return raw;
    }
    return 0.0;
  };
#pragma unroll
  for (int m = 0; m < TYV / NWV; ++m) {
    const int r = 1 + ty + NWV * m;
    sL[r][tx + 1] = ld(gi, j0 - 1 + r, own[m]);
  }
  double dummy;
  if (ty < 2) {
    const int r = ty == 0 ? 0 : TYV + 1;
    sL[r][tx + 1] = ld(gi, j0 - 1 + r, dummy);
  } else if (ty < 4) {
    const int l = threadIdx.x - 128;
    if (l < 2 * (TYV + 2)) {
      const int r = l >> 1, c = (l & 1) ? TX + 1 : 0;
      sL[r][c] = ld(i0 - 1 + c, j0 - 1 + r, dummy);
    }
  }
}

// ---- shared stencil arithmetic (per-stage and fused kernels use the same expressions) ------
// The reference divides every slope by dx before clamping (adjoint.jl:87-94).  The clamp is
// monotone and dx > 0, so  clamp(dS/dx, eta H+/dx, -eta H-/dx) == clamp(dS, eta H+, -eta H-)/dx
// with the SAME branch taken (ties included); the 1/dx factors are applied once per cell.
// pS/pH point at the node's lower-left cell; LD = row stride of the cell tiles.
// Cell tiles are stored interleaved as double2 {Hc, S} (16-B aligned) so that every stencil
// access is ONE full-rate ds_read_b128 instead of two half-rate ds_read2_b64.
__device__ __forceinline__ void node_geom_vals(const GDev& g, double2 c00, double2 c10, double2 c01, double2 c11,
                                               double& gx, double& gy, double& Hb) {
  gx = ((c10.y - c00.y) + (c11.y - c01.y)) * g.hinv_dx;
  gy = ((c01.y - c00.y) + (c11.y - c10.y)) * g.hinv_dy;
  Hb = 0.25 * ((c00.x + c10.x) + (c01.x + c11.x));
}
template <int LD>
__device__ __forceinline__ void node_geom(const GDev& g, const double2* p, double& gx, double& gy, double& Hb) {
  node_geom_vals(g, p[0], p[1], p[LD], p[LD + 1], gx, gy, Hb);
}
// dH/dt of an interior cell from its 5-point {Hc, S} values and the D of its four corner nodes.
// ETA1: eta0 == 1 (the integer-power law mode requires it), so eta0*H is H bit for bit.
template <bool ETA1>
__device__ __forceinline__ double cell_div_vals(const GDev& g, double2 c0, double2 ce_, double2 cw_, double2 cn_,
                                                double2 cs_, double Dsw, double Dse, double Dnw, double Dne) {
  const double e0 = ETA1 ? 1.0 : g.eta0;
  const double S0 = c0.y, eH0 = ETA1 ? c0.x : e0 * c0.x;
  const double ce = clampn(ce_.y - S0, ETA1 ? ce_.x : e0 * ce_.x, eH0);
  const double cw = clampn(S0 - cw_.y, eH0, (ETA1 ? cw_.x : e0 * cw_.x));
  const double cn = clampn(cn_.y - S0, ETA1 ? cn_.x : e0 * cn_.x, eH0);
  const double cs = clampn(S0 - cs_.y, eH0, (ETA1 ? cs_.x : e0 * cs_.x));
  const double qx = (Dse + Dne) * ce - (Dsw + Dnw) * cw;
  const double qy = (Dnw + Dne) * cn - (Dsw + Dse) * cs;
  return fma(g.hinv_dx2, qx, g.hinv_dy2 * qy);
}
// p at the cell, pD at its north-east node; LDD = node row stride.
template <int LD, int LDD, bool ETA1 = false>
__device__ __forceinline__ double cell_div(const GDev& g, const double2* p, const double* pD) {
  return cell_div_vals<ETA1>(g, p[0], p[1], p[-1], p[LD], p[-LD], pD[-LDD - 1], pD[-LDD], pD[-1], pD[0]);
}

// D on every dual node of the tile -> sD (0 on nodes outside the glacier's dual grid).
template <int LM>
__device__ __forceinline__ void nodes_forward(const GDev& g, const LawDev& L, const double* __restrict__ Afield, int i0,
                                              int j0, const double2 (*sHS)[LDW], double (*sD)[LDN]) {
  auto node = [&](int idx) {
    const int b = idx / (TX + 1), a = idx - b * (TX + 1);
    const int gi = i0 - 1 + a, gj = j0 - 1 + b;
    double D = 0.0;
    if (gi >= 0 && gi <= g.nx - 2 && gj >= 0 && gj <= g.ny - 2) {
      double gx, gy, Hb;
      node_geom<LDW>(g, &sHS[b][a], gx, gy, Hb);
      const double gS2 = gx * gx + gy * gy;
      double An = g.A;
      if (g.use_Afield) An = Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
      double al, be, sp;
      D = node_D<false, LM>(g, L, Hb, gS2, An, al, be, sp);
    }
    sD[b][a] = D;
  };
  if constexpr (lm_is_nn(LM)) {  // one copy of the inlined network, whatever the surrounding control flow
#pragma unroll 1
    for (int idx = threadIdx.x; idx < NNODE; idx += NT) node(idx);
  } else {
    for (int idx = threadIdx.x; idx < NNODE; idx += NT) node(idx);
  }
}

__device__ __forceinline__ double clampf(double e, double up, double lo) { return vmax(vmin(e, up), lo); }

// Barrier after the tile loads + "does the tile (with its halo) carry any ice?".  The closed-form laws
// use the answer for the exact ice-free shortcut; the inlined-MLP laws always answer yes: they skip
// the network per node where Hbar = 0 anyway, and a conditional stencil made the register allocator
// keep the whole network live (300 VGPRs instead of 131 for the 2x16 architecture).
template <int LM>
__device__ __forceinline__ bool tile_has_ice(bool loaded_ice) {
  if constexpr (lm_is_nn(LM)) {
    __syncthreads();
    return true;
  } else {
    return __syncthreads_or(loaded_ice) != 0;
  }
}

// dH/dt of the cell at halo coordinates (c, r); caller guarantees the cell is interior.
template <int LM>
__device__ __forceinline__ double cell_rhs(const GDev& g, int c, int r, const double2 (*sHS)[LDW],
                                           const double (*sD)[LDN]) {
  return cell_div<LDW, LDN, LM == LM_FAST>(g, &sHS[r][c], &sD[r][c]);
}

// =====================================================================================
// K1: RHS only.  dH = SIA2D(H)   (Huginn.SIA2D!, restated from adjoint.jl:52-97)
// =====================================================================================
template <int LM>
__global__ __launch_bounds__(NT) void k_dhdt(Pools P, LawDev L, const double* __restrict__ U, double* __restrict__ dH,
                                             int tile_base) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double sD[TY + 1][LDN];
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double own[RPT];
  // no ice on the tile and its halo: every clamped slope and D vanish, dH/dt = 0 exactly -- skip the stencil
  const bool ice = tile_has_ice<LM>(load_tile_HS2(U, P.B, g, i0, j0, sHS, own));
  if (ice) {
    nodes_forward<LM>(g, L, P.Afield, i0, j0, sHS, sD);
    __syncthreads();
  }
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      double k = 0.0;
      if (ice && gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2) k = cell_rhs<LM>(g, tx + 1, r, sHS, sD);
      dH[g.off + gi + (long long)g.nx * gj] = k;
    }
  }
}

// =====================================================================================
// K1b: one explicit-Euler step with a CFL-limited step size (the north star's "CFL" mode;
// read H,B; write H_new = 24 B per cell-step):  u' = u + dt dH/dt(u), and the
// tile's max D -- wavefront shuffle -> LDS -> one partial per tile, reduced in fixed order by
// the controller, which sets the NEXT step to dt = cfl * min(dx,dy)^2 / (4 max D).  dt == 0
// (priming launch) just measures max D(u0).
// =====================================================================================
template <int LM>
__global__ __launch_bounds__(NT) void k_euler_cfl(Pools P, LawDev L, const double* __restrict__ Usrc,
                                                  double* __restrict__ Udst) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double sD[TY + 1][LDN];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (gs->done) return;
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const double dt = gs->dt;
  double own[RPT];
  const bool ice = tile_has_ice<LM>(load_tile_HS2(Usrc, P.B, g, i0, j0, sHS, own));
  if (ice) {
    nodes_forward<LM>(g, L, P.Afield, i0, j0, sHS, sD);
    __syncthreads();
  }
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double dmax = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      double k = 0.0;
      if (ice && gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2) k = cell_rhs<LM>(g, tx + 1, r, sHS, sD);
      Udst[g.off + gi + (long long)g.nx * gj] = fma(dt, k, own[m]);
    }
    if (ice) dmax = fmax(dmax, sD[r][tx + 1]);  // the node north-east of the cell (0 where it does not exist)
  }
  const double tot = block_max(dmax, red);
  if (threadIdx.x == 0) P.part[4 * (long long)t4.w] = tot;
}

// The 3S*+ registers tmp/uprev/utilde are pure streams (read once, written once, no halo):
// they are accessed with the non-temporal policy so that they do not evict the u/B halo rows
// other tiles are about to re-read from L2 (+10 % on the HBM-bound stage kernels).
// =====================================================================================
// K2: one fused RDPK3Sp35 stage (RHS + 3S*+ register update + embedded-error partial).
//   tmp (S2), uprev (S3), utilde (E); u ping-pongs between Usrc and Udst.
//   stage 1 : reads X = accepted ? Ucur : S3 ; writes Udst, S3 (<- X), E
//   stage 2 : tmp_old == uprev (never stored by stage 1); writes Udst, S2, E
//   stage 3 : reads S2, E           ; writes Udst, S2, E
//   stage 4 : reads S2, S3, E       ; writes Udst, S2, E
//   stage 5 : reads S2, S3, E       ; writes Udst ; error partial -> part[4*tile]
// =====================================================================================
template <int STAGE, int LM>
__global__ __launch_bounds__(NT) void k_rk_stage(Pools P, LawDev L, const double* __restrict__ Usrc,
                                                 double* __restrict__ Udst, double* __restrict__ S2,
                                                 double* __restrict__ S3, double* __restrict__ E, double abstol,
                                                 double reltol) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double sD[TY + 1][LDN];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (gs->done) return;
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const double dt = gs->dt;
  const double* __restrict__ X = Usrc;
  if (STAGE == 1 && !gs->accepted) X = S3;  // rejected step: restart from uprev
  double own[RPT];
  // no ice on the tile and its halo: every clamped slope and D vanish, dH/dt = 0 exactly -- skip the stencil
  const bool ice = tile_has_ice<LM>(load_tile_HS2(X, P.B, g, i0, j0, sHS, own));
  if (ice) {
    nodes_forward<LM>(g, L, P.Afield, i0, j0, sHS, sD);
    __syncthreads();
  }
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  constexpr int s = STAGE - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  double errsq = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      double k = 0.0;
      if (ice && gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2) k = cell_rhs<LM>(g, tx + 1, r, sHS, sD);
      const double u = own[m];
      const double dtk = dt * k;
      if (STAGE == 1) {
        Udst[id] = fma(bt, dtk, u);
        if (gs->accepted) __builtin_nontemporal_store(u, &S3[id]);
        __builtin_nontemporal_store(bh * dtk, &E[id]);
      } else {
        const double up = (STAGE == 2 || STAGE >= 4) ? __builtin_nontemporal_load(&S3[id]) : 0.0;
        const double tmp_old = (STAGE == 2) ? up : __builtin_nontemporal_load(&S2[id]);
        const double tmp = fma(dl, u, tmp_old);
        double un = fma(g1, u, g2 * tmp);
        if (STAGE >= 4) un = fma(g3, up, un);
        un = fma(bt, dtk, un);
        Udst[id] = un;
        const double e = fma(bh, dtk, __builtin_nontemporal_load(&E[id]));
        if (STAGE < 5) {
          if (STAGE < 5 && dl != 0.0) __builtin_nontemporal_store(tmp, &S2[id]);
          __builtin_nontemporal_store(e, &E[id]);
        } else {
          const double err = (un - up) - e;
          const double sk = abstol + fmax(fabs(up), fabs(un)) * reltol;
          const double q = err / sk;
          errsq = fma(q, q, errsq);
        }
      }
    }
  }
  if (STAGE == 5) {
    const double tot = block_sum(errsq, red);
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w] = tot;
  }
}

// controller: one block (64 threads) per glacier.  Sums the error partials in a fixed
// order, applies the PID controller (beta = 0.64,-0.31,0.04; limiter 1+atan(x-1);
// accept iff factor >= 0.81), advances t / tstops, and proposes the next dt.
// Stop tables are PER GLACIER (the reference builds tstops per glacier: gradient.jl:96-107, inversion_utils.jl:487-495):
// entry i of glacier g lives at [i * G + g]; glacier g has nstops[g] entries (t_0 ... t_end, strictly increasing).
struct CtrlArgs {
  const double* tstops;   // [imax][G]
  const int* nstops;      // [G]
  int G;
  const int* mb_flag;     // [imax][G]: 1 if the mass balance is applied at that stop
  const int* mb_slot;     // [imax][G]: index of the pre-MB snapshot
  const int* snap_slot;   // [imax][G] forward solve: slot of the stop's snapshot (result stops: the glacier's own result
                          // index; stops that exist only for the mass balance: hidden slots behind them); null: i
  __device__ __forceinline__ double tstop(int i, int g) const { return tstops[(long long)i * G + g]; }
  __device__ __forceinline__ int at(const int* tab, int i, int g) const { return tab[(long long)i * G + g]; }
  double dtmax;
  int adaptive;
  double fixed_dt;
  int* n_active;
  int next_cur;  // ping-pong buffer that holds u_new of this step; -1: flip the glacier's own `cur`
  const double* errpart;  // per-tile error partials: errpart[stride * tile]
  int stride;
  int fused;              // partials are indexed by the fused-step tile table (1: FOY tiles, 2: FOYS tiles, 3: FOYT tiles, 4: FOYT8 tiles, 5: 62 x 62 tiles, 6: FOYT4 tiles, 7: FOYT2 tiles)
  int* est_steps;         // [G] (nullable): estimated steps still needed, for the host's poll spacing
  double cfl;             // > 0: explicit Euler with dt = cfl*min(dx,dy)^2/(4 max D); the partials are tile maxima
  int cfl_prime;          // the launch only measured max D(u0): set the first dt, do not advance
  // reverse (continuous-adjoint) solve only; adj == null in the forward solve.  tstops are then
  // tau = -t ascending, the union of the snapshot times and the Gauss-Legendre nodes.
  AdjState* adj;
  const double* tsnap;    // [kmax][G] forward snapshot times t_0 < ... < t_{k_g - 1} of every glacier
  const int* stop_snap;   // [imax][G] per stop: forward snapshot index, -1 for a quadrature node
  const double* stop_qw;  // [imax][G] per stop: quadrature weight, 0 for a snapshot time
  const int* stop_hid;    // [imax][G] (nullable) per stop: hidden snapshot slot + 1 of a mass-balance-only stop, else 0
  double* qw_out;         // per glacier: weight of the node reached by this step (0 otherwise)
  double* trace;          // (nullable, diagnostics: ODINN_TRACE_STEPS) [trace_cap][4] of glacier 0: t, dt, EEst, +-factor per attempt
  int trace_cap, trace_g;  // trace_g: the traced glacier (ODINN_TRACE_GLACIER, default 0)
  int stuck_off;          // ODINN_DTMIN=0 (diagnostics): no exit on a collapsed step size
  int nrows;              // rows of the stop tables; t_last: the time of every glacier's last stop (self-controlled reverse step:
  double t_last;          //   CtrlPre)
};

// self-controlled fused step (sia2d_fused.hpp, k_rk_fused_strip<..., SC = true>)
struct ScArgs {
  CtrlArgs C;
  const GState* gin;
  GState* gout;
  const double* part_in;  // error partials written by the previous launch (this launch writes partF)
  double* snaps;          // [n_stops][ntot]
  long long ntot;
  double* premb;          // mass balance at a stop (applied on load): pre-MB snapshots, pooled MB fields
  const double* mb0;
  const double* Sref;     // (nullable)
  int snap_on_load;       // 1: NOT the self-controlled loop -- the controller kernel decides as usual, but the step kernel
                          // stores the snapshot of a stop the previous step reached (gs->at_stop) from the state it loads,
                          // which replaces the post-step launch of batches without a mass balance
};

// interpolation weights of H_itp at the five stage times of the step [tau, tau + dt]
__device__ __forceinline__ void adj_stage_weights(AdjState* a, const double* tsnap, int G, int gidx, double tau, double dt,
                                                  bool all_at_end, double ta, double tb) {
  const double inv = 1.0 / (tb - ta);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double t = -(tau + (all_at_end ? 1.0 : c_cc[i]) * dt);
    a->sitp[i] = fmin(fmax((t - ta) * inv, 0.0), 1.0);
  }
}
__device__ __forceinline__ void adj_stage_weights(AdjState* a, const double* tsnap, int G, int gidx, double tau, double dt,
                                                  bool all_at_end) {
  const double ta = tsnap[(long long)a->seg * G + gidx], inv = 1.0 / (tsnap[(long long)(a->seg + 1) * G + gidx] - ta);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double t = -(tau + (all_at_end ? 1.0 : c_cc[i]) * dt);
    a->sitp[i] = fmin(fmax((t - ta) * inv, 0.0), 1.0);
  }
}

// ---- the step controller, as device functions shared by k_controller and k_adj_ctrl_post -----------------------------------
// error sum of a glacier's per-tile partials (fixed order: lane l takes tiles l, l + 64, ...; then the butterfly) and the
// three powers of the PID factor, evaluated by lanes 0..2 side by side (one pow's latency instead of three on the single-
// thread critical path; same values, same order of the product).  Called by the 64 lanes of one wavefront.
__device__ __forceinline__ void controller_errsum(const CtrlArgs& C, const GDev& g, double e2, double e3, int lane, double& s_out,
                                                  double& pw0, double& pw1, double& pw2) {
  double s = 0.0;
  {
    const int t0 = C.fused == 7 ? g.tile0Fw : C.fused == 6 ? g.tile0Fv : C.fused == 5 ? g.tile0D : C.fused == 4 ? g.tile0Fu : C.fused == 3 ? g.tile0Ft : C.fused == 2 ? g.tile0Fs : (C.fused ? g.tile0F : g.tile0);
    const int nt = C.fused == 7 ? g.ntilesFw : C.fused == 6 ? g.ntilesFv : C.fused == 5 ? g.ntilesD : C.fused == 4 ? g.ntilesFu : C.fused == 3 ? g.ntilesFt : C.fused == 2 ? g.ntilesFs : (C.fused ? g.ntilesF : g.ntiles);
    if (C.cfl > 0.0) {
      for (int k = lane; k < nt; k += 64) s = fmax(s, C.errpart[(long long)C.stride * (t0 + k)]);
    } else {
      // (the loads of up to eight rounds are issued before the first addition; same order of the additions)
      for (int k0 = lane; k0 < nt; k0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = k0 + 64 * q;
          v[q] = k < nt ? C.errpart[(long long)C.stride * (t0 + k)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (k0 + 64 * q < nt) s += v[q];
      }
    }
  }
  // fixed-shape tree: lane l holds sum of tiles l, l+64, ... ; then butterfly
  s = C.cfl > 0.0 ? wave_max(s) : wave_sum(s);
  // the three powers of the PID factor are evaluated by lanes 0..2 side by side (one pow's latency instead of three on the
  // single-thread critical path of this kernel; same values, same order of the product)
  pw0 = 1.0; pw1 = 1.0; pw2 = 1.0;
  if (C.adaptive && !(C.cfl > 0.0)) {
    const double sb = __shfl(s, 0, 64);
    double EEst = sqrt(sb / ((double)g.nx * (double)g.ny));
    if (!(EEst == EEst) || isinf(EEst)) EEst = 1e300;
    if (EEst < 2.220446049250313e-16) EEst = 2.220446049250313e-16;
    const int l = lane;
    const double base = l == 0 ? 1.0 / EEst : (l == 1 ? e2 : e3);
    const double ex = l == 0 ? 0.64 / 3.0 : (l == 1 ? -0.31 / 3.0 : 0.04 / 3.0);
    const double pw = l < 3 ? pow(base, ex) : 1.0;
    pw0 = __shfl(pw, 0, 64); pw1 = __shfl(pw, 1, 64); pw2 = __shfl(pw, 2, 64);
  }
  s_out = s;
}
// the decision itself (PID controller, accept / reject, stop handling, next step size; the reverse solve's AdjState),
// on values in registers: one thread.  Returns 1 when the glacier has just finished; est: steps still needed (-1: n/a).
// pre (self-controlled reverse step): table entries requested before the decision starts -- stop `is` (the one the attempt
// aimed at), the time of stop is + 1, the last stop's time, the snapshot times of segment `seg`; the same values the tables hold
constexpr int STALL_MAX = 256;  // consecutive attempts without progress in t after which an adaptive solve is given up (controller_decide)
struct CtrlPre {
  int is, seg, n_stops, mbf, mbs, snap, hid;
  double t_is, t_is1, t_last, qw, ta, tb;
};
__device__ __forceinline__ int controller_decide(GState& st, AdjState& ad, const GDev& g, const CtrlArgs& C, int gidx, double s,
                                                 double pw0, double pw1, double pw2, int& est, const CtrlPre* pre = nullptr) {
  const int n_stops = pre ? pre->n_stops : C.nstops[gidx];
  auto tstop_at = [&](int i) {
    if (pre) {
      if (i == pre->is) return pre->t_is;
      if (i == pre->is + 1) return pre->t_is1;
      if (i == n_stops - 1) return pre->t_last;
    }
    return C.tstop(i, gidx);
  };
  est = -1;
  const double h = st.dt;
  double fac = 1.0;
  bool accept = true;
  if (C.adaptive) {
    double EEst = sqrt(s / ((double)g.nx * (double)g.ny));
    if (!(EEst == EEst) || isinf(EEst)) { st.nonfinite = 1; EEst = 1e300; }
    if (EEst < 2.220446049250313e-16) EEst = 2.220446049250313e-16;
    st.EEst = EEst;
    const double e1 = 1.0 / EEst;
    fac = pw0 * pw1 * pw2;  // pow(e1, 0.64 / 3) * pow(e2, -0.31 / 3) * pow(e3, 0.04 / 3)
    fac = 1.0 + atan(fac - 1.0);
    accept = fac >= 0.81;
    if (accept) { st.e3 = st.e2; st.e2 = e1; }
  }
  double t = st.t;
  const double t0_ = st.t;
  if (C.trace && gidx == C.trace_g) {
    const long long q = st.naccept + st.nreject;
    if (q < C.trace_cap) { C.trace[4 * q] = t; C.trace[4 * q + 1] = h; C.trace[4 * q + 2] = st.EEst; C.trace[4 * q + 3] = accept ? fac : -fac; }
  }
  st.at_stop = 0;
  st.mb_now = 0;
  if (C.adj) { ad.qw = 0.0; ad.snapj = -1; ad.pad = 0; }
  if (accept) {
    st.naccept++;
    st.accepted = 1;
    st.cur = C.next_cur >= 0 ? C.next_cur : 1 - st.cur;
    if (st.clipped) {
      t = tstop_at(st.istop);
      st.at_stop = 1;
      st.mb_now = pre ? pre->mbf : C.at(C.mb_flag, st.istop, gidx);
      st.mb_slot = pre ? pre->mbs : C.at(C.mb_slot, st.istop, gidx);
      st.snap_slot = C.snap_slot ? C.at(C.snap_slot, st.istop, gidx) : st.istop;
      if (C.adj) {
        AdjState* a = &ad;
        a->snapj = pre ? pre->snap : C.at(C.stop_snap, st.istop, gidx);
        a->pad = pre ? pre->hid : (C.stop_hid ? C.at(C.stop_hid, st.istop, gidx) : 0);
        a->qw = pre ? pre->qw : C.stop_qw[(long long)st.istop * C.G + gidx];
        a->seg_stop = a->seg;
        const double ta = pre ? pre->ta : C.tsnap[(long long)a->seg * C.G + gidx];
        a->s_stop = a->snapj >= 0 ? (a->snapj == a->seg ? 0.0 : 1.0) : (-t - ta) / ((pre ? pre->tb : C.tsnap[(long long)(a->seg + 1) * C.G + gidx]) - ta);
        if (a->snapj >= 1) a->seg = a->snapj - 1;  // the next steps run below snapshot j
      }
      st.istop++;
    } else {
      t += h;
    }
    st.t = t;
  } else {
    st.nreject++;
    st.accepted = 0;
  }
  // A solve that is stuck: STALL_MAX consecutive attempts without advancing t -- rejections, or accepted steps so small that
  // t + dt == t.  Seen where the reverse solve meets a jump of its right-hand side at a mass-balance stop a few 1e-5 yr from the next
  // stop: rejections down to dt < eps(t), an accepted step that does not move tau, growth, rejection, ... for ever (three fuzz draws in
  // 23 200 seeds spun like that to maxiters).  The glacier leaves the loop, the host reports ODINN_ERR_DTMIN (nonfinite == 2).
  // OrdinaryDiffEq aborts EARLIER, at the first dt <= dtmin = eps(t) (ReturnCode.DtLessThanMin) -- also on the ~1 % of the fuzz draws
  // whose reverse solve dips below eps(t) for a few attempts and recovers; those keep running here, as they always did.
  if (C.adaptive) {
    if (accept && st.t != t0_) st.pad2 = 0;
    else if (++st.pad2 >= STALL_MAX && !C.stuck_off) {
      if (!st.nonfinite) st.nonfinite = 2;
      st.done = 1;
      est = 0;
      return 1;
    }
  }
  if (!C.adj) {
    // forward solve, mass balance applied ON LOAD by the strip step kernel (ScArgs::snap_on_load with a mass balance): bit 2
    // of `pad` says the buffer `cur` still lacks the mass balance of the stop it sits on -- set when an accepted step lands
    // on such a stop, cleared by the next accepted step (which replaces the buffer).  Ignored by the post-step schedule.
    if (accept) st.pad &= ~4;
    if (accept && st.at_stop && st.mb_now && g.has_mb) st.pad |= 4;
  }
  if (st.istop >= n_stops) {
    st.done = 1;
    est = 0;
    return 1;
  }
  double dtn = C.adaptive ? h * fac : C.fixed_dt;
  if (C.dtmax > 0.0 && dtn > C.dtmax) dtn = C.dtmax;
  const double rem = tstop_at(st.istop) - t;
  // snap to the stop when the step would end within 100 ulp of it
  if (dtn >= rem || fabs(rem - dtn) <= 100.0 * 2.220446049250313e-16 * fabs(t)) {
    dtn = rem;
    st.clipped = 1;
  } else {
    st.clipped = 0;
  }
  st.dt = dtn;
  {  // at the current step size, and at least one step per remaining stop
    const double e = ceil((tstop_at(n_stops - 1) - t) / (C.adaptive ? h * fac : dtn));
    const int stops_left = n_stops - st.istop;
    est = e < (double)stops_left ? stops_left : (e > 1e6 ? 1000000 : (int)e);
  }
  if (C.adj) {
    if (pre && ad.seg == pre->seg) adj_stage_weights(&ad, C.tsnap, C.G, gidx, t, dtn, false, pre->ta, pre->tb);
    else adj_stage_weights(&ad, C.tsnap, C.G, gidx, t, dtn, false);
  }
  return 0;
}

#ifdef ODINN_MISC_KERNELS
__global__ __launch_bounds__(64) void k_controller(Pools P, CtrlArgs C) {
  const int gidx = blockIdx.x;
  GState* gs = P.gs + gidx;
  if (gs->done) {
    if (threadIdx.x == 0) {
      gs->at_stop = 0;  // its final post-step already ran
      if (C.qw_out) C.qw_out[gidx] = 0.0;
    }
    return;
  }
  const GDev g = P.gd[gidx];
  double s, pw0, pw1, pw2;
  controller_errsum(C, g, gs->e2, gs->e3, (int)threadIdx.x, s, pw0, pw1, pw2);
  if (threadIdx.x != 0) return;
  // one thread from here on: the per-glacier state is taken into registers once and written back once (dozens of dependent
  // global read-modify-writes through `gs->` otherwise: the kernel spent most of its 8 us on them)
  GState st = *gs;
  AdjState ad{};
  if (C.adj) ad = C.adj[gidx];
  if (C.cfl > 0.0) {
    // explicit Euler, always accepted; the next dt comes from the max diffusivity just measured
    if (!(s == s) || isinf(s)) st.nonfinite = 1;
    double t = st.t;
    st.at_stop = 0;
    st.mb_now = 0;
    st.EEst = s;
    if (!C.cfl_prime) {
      st.naccept++;
      st.accepted = 1;
      st.cur = C.next_cur;
      if (st.clipped) {
        t = C.tstop(st.istop, gidx);
        st.at_stop = 1;
        st.mb_now = C.at(C.mb_flag, st.istop, gidx);
        st.mb_slot = C.at(C.mb_slot, st.istop, gidx);
        st.snap_slot = C.snap_slot ? C.at(C.snap_slot, st.istop, gidx) : st.istop;
        st.istop++;
      } else {
        t += st.dt;
      }
      st.t = t;
      if (st.istop >= C.nstops[gidx]) {
        st.done = 1;
        atomicSub(C.n_active, 1);
        *gs = st;
        return;
      }
    }
    const double dmin = fmin(g.dx, g.dy);
    const double rem = C.tstop(st.istop, gidx) - t;
    double dtn = s > 0.0 ? C.cfl * dmin * dmin / (4.0 * s) : rem;
    if (C.dtmax > 0.0 && dtn > C.dtmax) dtn = C.dtmax;
    if (dtn >= rem || fabs(rem - dtn) <= 100.0 * 2.220446049250313e-16 * fabs(t)) {
      dtn = rem;
      st.clipped = 1;
    } else {
      st.clipped = 0;
    }
    st.dt = dtn;
    *gs = st;
    return;
  }
  int est;
  const int newly_done = controller_decide(st, ad, g, C, gidx, s, pw0, pw1, pw2, est);
  if (C.qw_out) C.qw_out[gidx] = ad.qw;
  if (newly_done) atomicSub(C.n_active, 1);
  if (C.est_steps && est >= 0) C.est_steps[gidx] = est;
  *gs = st;
  if (C.adj) C.adj[gidx] = ad;
}

#endif  // ODINN_MISC_KERNELS

// post-step: at a tstop, apply the mass balance in place (VJPs.jl:129-139) and store the
// snapshot (and the pre-MB state for the adjoint).  Pointwise; no-op for glaciers not at a stop.
struct PostArgs {
  double* snaps;           // [n_stops][Ntot]
  double* premb;           // [n_mb][Ntot]
  long long ntot;
  const double* mb0;       // pooled
  const double* Sref;      // pooled (may be null)
};

// arguments of k_vref_itp (continuous adjoint with a velocity loss)
struct VItpArgs {
  const double* Vabs;   // [n_vref][ntot]
  const double* Vxr;
  const double* Vyr;
  long long ntot;
  const int* slotA;     // [n_rstop][G]; -1: this glacier has no velocity data
  const int* slotB;
  const double* sw;     // interpolation weight inside [slotA, slotB]
  int G;
  const AdjState* adj;
  double* Vq;           // 3 x ntot out: Vabs, Vx, Vy
};
// post-step arguments of the reverse (continuous-adjoint) solve, see k_adj_poststep
struct AdjPostArgs {
  const AdjState* adj;
  const double* snaps;
  const double* premb;
  long long ntot;
  const double* mb0;
  const double* Sref;
  const double* Href;
  const unsigned char* mask;
  const double* ws;      // [n_snap][G] loss weights
  const int* refslot;    // [n_snap][G]
  int G;
  int loss_first;
  double* Hq;
  // LossDhdt (time-aggregated, TimeAggregatedLosses.jl:82-113): at the snapshot dh_i1[g] (dh_i0[g]) of glacier g the
  // field +(-) dh_coef[g] [H(t0) > 1e-2] joins lambda after the loss term (CallbackSet order, gradient.jl:437); null: off
  const int* dh_i0;
  const int* dh_i1;
  const double* dh_coef;
  // LossAvgV (TimeAggregatedLosses.jl:183-258): dL/dH of the stop snapj, precomputed after the forward solve, lives in
  // aggH[agg_slot[snapj]] (zero for glaciers whose tLoss does not contain the stop); null: off
  const int* agg_slot;
  const double* aggH;
  double h_log_eps;      // LossH's simple loss: 0 = L2Sum, > 0 = LogSum(eps)
  int hq_snap_only;      // Hq is wanted at the snapshot stops only (the quadrature nodes' consumer interpolates H itself)
};

__device__ __forceinline__ double mb_value(const GDev& g, double mb0, double sref, double H, double B, double& dmb) {
  double raw = mb0;
  dmb = 0.0;
  if (g.dmb_dS != 0.0) {
    raw = mb0 + g.dmb_dS * ((B + H) - sref);
    dmb = g.dmb_dS;
    if (raw >= g.mb_max) { raw = g.mb_max; dmb = 0.0; }
  }
  return raw;
}

#ifdef ODINN_MISC_KERNELS
__global__ __launch_bounds__(NT) void k_poststep(Pools P, PostArgs A, double* __restrict__ Ua, double* __restrict__ Ub) {
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (!gs->at_stop) return;
  const GDev g = P.gd[t4.x];
  double* __restrict__ U = gs->cur ? Ub : Ua;
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  const int slot = gs->snap_slot;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      double h = U[id];
      if (gs->mb_now && g.has_mb) {
        A.premb[(long long)gs->mb_slot * A.ntot + id] = h;
        double dmb;
        double mb = mb_value(g, A.mb0[id], A.Sref ? A.Sref[id] : 0.0, h, P.B[id], dmb);
        const bool mask = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
        if (!mask) mb = 0.0;
        if (mask && h + mb < 0.0) mb = -h;
        h += mb;
        U[id] = h;
      }
      A.snaps[(long long)slot * A.ntot + id] = h;
    }
  }
}

#endif  // ODINN_MISC_KERNELS

// D_adjoint of a dual node (adjoint.jl:99-104) from its own four edges, slopes unscaled:
//   Da = -(0.5/dx^2) sum_xedges dlam*clamp(dS) - (0.5/dy^2) sum_yedges dlam*clamp(dS)
// p: {Hc,S} at the node's lower-left cell, pl: lambda~ there; vxl/vxu/vyl/vyr: edge validity.
template <int LD>
__device__ __forceinline__ double node_Da(const GDev& g, const double2* p, const double* pl, bool vxl, bool vxu,
                                          bool vyl, bool vyr, double& gx, double& gy, double& Hb) {
  const double2 c00 = p[0], c10 = p[1], c01 = p[LD], c11 = p[LD + 1];
  const double l00 = pl[0], l10 = pl[1], l01 = pl[LD], l11 = pl[LD + 1];
  const double dxl = c10.y - c00.y, dxu = c11.y - c01.y, dyl = c01.y - c00.y, dyr = c11.y - c10.y;
  gx = (dxl + dxu) * g.hinv_dx;
  gy = (dyl + dyr) * g.hinv_dy;
  Hb = 0.25 * ((c00.x + c10.x) + (c01.x + c11.x));
  const double e00 = g.eta0 * c00.x, e10 = g.eta0 * c10.x, e01 = g.eta0 * c01.x, e11 = g.eta0 * c11.x;
  double ax = 0.0, ay = 0.0;
  if (vxl) ax = (l10 - l00) * clampn(dxl, e10, e00);
  if (vxu) ax = fma(l11 - l01, clampn(dxu, e11, e01), ax);
  if (vyl) ay = (l01 - l00) * clampn(dyl, e01, e00);
  if (vyr) ay = fma(l11 - l10, clampn(dyr, e11, e10), ay);
  return -fma(g.hinv_dx2, ax, g.hinv_dy2 * ay);
}

// =====================================================================================
// K5: H-VJP (DiscreteVJP adjoint.jl:99-148, or ContinuousVJP :442-553) in node->corner form,
// optionally fused with the reverse explicit-Euler update of gradient.jl:242:
//     out = lam + dt * J_H(H)^T lam + w * 2*mask*(H - Href)/N          (MODE 1)
//     out = J_H(H)^T lam                                               (MODE 0)
// and the masked L2 loss partial (Losses.jl:133-141) as a by-product in MODE 1.
// =====================================================================================
// the simple loss of LossH on one cell (Losses.jl:133-152 L2Sum, :207-229 LogSum(eps)): d = the factor of 2 w / N in dl/dH,
// q = the cell's share of the loss.  eps == 0: L2Sum.
__device__ __forceinline__ void simple_loss_terms(double a, double b, double eps, double& d, double& q) {
  if (eps > 0.0) {
    const double lg = log((a + eps) / (b + eps));
    d = lg / (a + eps);
    q = lg * lg;
  } else {
    d = a - b;
    q = d * d;
  }
}

struct AdjArgs {
  double h_log_eps;    // LossH's simple loss: 0 = L2Sum, > 0 = LogSum(eps)
  const double* H;     // snapshot (pooled)
  const double* lam;   // pooled
  double* out;         // pooled
  const double* Href;  // pooled references [slot][ntot] or null
  const unsigned char* mask;  // pooled masks [slot][ntot] or null
  const double* dts;   // per-glacier dt (device) or null
  const double* ws;    // per-glacier loss weight (device) or null
  const int* refslot;  // per-glacier reference slot for this stop
  long long ntot;
  // MODE 0 only, continuous adjoint: H = H_itp at stage time 0 of each glacier's AdjState
  const double* snaps; // forward snapshots [n_snap][ntot] or null (then H is used)
  const AdjState* adj;
  // k_vjp_H_strip<..., YT>: the Y law through its table (every glacier yt_fast)
  const double* ytab;
  int* ytab_over;
  int ytab_ni;
};

// One dual node of k_vjp_H: what node (a,b) of the tile contributes to the VJP of its four corner
// cells {SW, SE, NW, NE} -- the diffusivity term (adjoint.jl:123-127) AND its share D_node of the
// clamp/flux term of its four edges (adjoint.jl:130-144, inversion_utils.jl:22-43); the clamped
// slopes and their bounds are already at hand from D_adjoint (adjoint.jl:99-104).
template <int LM, int NK = 0, int LD = LDW>
__device__ __forceinline__ void vjpH_node(const GDev& g, const LawDev& L, const Pools& P, const double2 (*sHS)[LD],
                                          const double (*sL)[LD], int i0, int j0, int a, int b, double (&k)[4],
                                          const UtabTile ut = ODINN_UT_NONE) {
  // LD: row stride of the two tiles (LDW: the 64 x 16 tiles with their one-cell halo; FLD: the regions of the fused LDS-tile steps)
  const int gi = i0 - 1 + a, gj = j0 - 1 + b;
  k[0] = k[1] = k[2] = k[3] = 0.0;
  if (gi < 0 || gi > g.nx - 2 || gj < 0 || gj > g.ny - 2) return;
  const double2* p = &sHS[b][a];
  const double* pl = &sL[b][a];
  double al, be, sp, Dnn = 0.0;
  // (measured: pays for the U law's five-point evaluation, costs the Y law 18 %; the U TABLE: 200 -> 142 VGPRs, 275 -> 221 us at 8 x 1024^2)
  constexpr bool LAW_FIRST = (lm_is_nn(LM) && NK == 4) || LM == LM_UTAB;
  if constexpr (LAW_FIRST) {
    // per-node network: evaluate the law FIRST, from the node's thickness and slope alone, so that none of the node's other
    // quantities (corner values, bounds, lambda differences) is live across the ~2000 instructions of the network; the
    // corner values are read again from LDS afterwards (the memory clobber keeps the two sets of loads apart)
    const double2 c00 = p[0], c10 = p[1], c01 = p[LD], c11 = p[LD + 1];
    const double gx = ((c10.y - c00.y) + (c11.y - c01.y)) * g.hinv_dx, gy = ((c01.y - c00.y) + (c11.y - c10.y)) * g.hinv_dy;
    const double Hb = 0.25 * ((c00.x + c10.x) + (c01.x + c11.x));
    Dnn = node_D<true, LM, NK>(g, L, Hb, gx * gx + gy * gy, g.A, al, be, sp, ut);
    asm volatile("" ::: "memory");
  }
  const double2 c00 = p[0], c10 = p[1], c01 = p[LD], c11 = p[LD + 1];
  const double l00 = pl[0], l10 = pl[1], l01 = pl[LD], l11 = pl[LD + 1];
  const double dxl = c10.y - c00.y, dxu = c11.y - c01.y, dyl = c01.y - c00.y, dyr = c11.y - c10.y;
  const double gx = (dxl + dxu) * g.hinv_dx, gy = (dyl + dyr) * g.hinv_dy;
  const double Hb = 0.25 * ((c00.x + c10.x) + (c01.x + c11.x));
  const double e00 = g.eta0 * c00.x, e10 = g.eta0 * c10.x, e01 = g.eta0 * c01.x, e11 = g.eta0 * c11.x;
  // lambda differences along the four edges (0 on edges that do not exist)
  const double qxl = gj >= 1 ? l10 - l00 : 0.0, qxu = gj + 1 <= g.ny - 2 ? l11 - l01 : 0.0;
  const double qyl = gi >= 1 ? l01 - l00 : 0.0, qyr = gi + 1 <= g.nx - 2 ? l11 - l10 : 0.0;
  const double ax = fma(qxu, clampn(dxu, e11, e01), qxl * clampn(dxl, e10, e00));
  const double ay = fma(qyr, clampn(dyr, e11, e10), qyl * clampn(dyl, e01, e00));
  const double Da = -fma(g.hinv_dx2, ax, g.hinv_dy2 * ay);
  double D;
  if constexpr (LAW_FIRST) {
    D = Dnn;
  } else {
    double An = g.A;
    if (g.use_Afield) An = P.Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
    D = node_D<true, LM, NK>(g, L, Hb, gx * gx + gy * gy, An, al, be, sp, ut);
  }
  // first term: avg^T(alpha Da) + dx^T(ay^T(beta gx Da))/dx + dy^T(ax^T(beta gy Da))/dy
  const double ad = 0.25 * al * Da, bd = be * Da;
  const double bx = g.hinv_dx * (bd * gx), by = g.hinv_dy * (bd * gy);
  k[0] = ad - bx - by; k[1] = ad + bx - by; k[2] = ad - bx + by; k[3] = ad + bx + by;
  // second term.  Edge between a "minus" cell (bound -eta H-) and a "plus" cell (bound eta H+):
  // weight 1 for both inside the open interval; eta0 for the minus cell where dS < -eta H-, eta0 for
  // the plus cell where dS > eta H+ (strict inequalities, inversion_utils.jl:24-28).
  const double Dx = D * g.hinv_dx2, Dy = D * g.hinv_dy2, eta = g.eta0;
#define ODINN_EDGE(dS, em, ep, q, Dd, KM, KP)                              \
  {                                                                        \
    const bool in = dS < ep && dS > -em;                                   \
    const double t = Dd * q;                                               \
    k[KM] = fma(t, in ? 1.0 : (dS < -em ? eta : 0.0), k[KM]);              \
    k[KP] = fma(-t, in ? 1.0 : (dS > ep ? eta : 0.0), k[KP]);              \
  }
  ODINN_EDGE(dxl, e00, e10, qxl, Dx, 0, 1)
  ODINN_EDGE(dxu, e01, e11, qxu, Dx, 2, 3)
  ODINN_EDGE(dyl, e00, e01, qyl, Dy, 0, 2)
  ODINN_EDGE(dyr, e10, e11, qyr, Dy, 1, 3)
#undef ODINN_EDGE
}

// LDS of the H-VJP kernels.  Phase A: {Hc,S} and lambda tiles; phase B: the node->corner
// contributions.  The closed-form laws reuse the same 35 KB for both (4 blocks / CU); the MLP laws
// (LM >= 2) are VALU-bound: they keep both side by side (2 blocks / CU) and a rolled node loop, so
// the inlined network is instantiated once.
// VJ: 0 = DiscreteVJP (adjoint.jl:31-151), 1 = ContinuousVJP (adjoint.jl:442-553),
// which needs a third double2 per node ({alpha/4, q/4}).
template <int LM, int VJ = 0>
struct VjpHLds {
  static constexpr bool ALIAS = !lm_is_nn(LM);
  static constexpr int A_D2 = (TY + 2) * LDW + ((TY + 2) * LDW + 1) / 2;  // in double2 units
  static constexpr int B_D2 = (VJ ? 3 : 2) * (TY + 1) * LDN;
  static constexpr int SIZE = ALIAS ? (A_D2 > B_D2 ? A_D2 : B_D2) : A_D2 + B_D2;
  // tabulated U law: the tile's table patches (utab_stage) live behind the tiles, in the part of the node-result region the tiles
  // do not cover -- 4 double2 of reduction scratch, then UT_CAP patches of 18 double2 (23 for the DiscreteVJP stencil, no extra LDS)
  static constexpr int UT_CAP = (LM == LM_UTAB && ALIAS && B_D2 > A_D2 + 4) ? (B_D2 - A_D2 - 4) / 18 : 0;
  static __device__ __forceinline__ double2 (*hs(double2* m))[LDW] { return reinterpret_cast<double2(*)[LDW]>(m); }
  static __device__ __forceinline__ double (*lam(double2* m))[LDW] {
    return reinterpret_cast<double(*)[LDW]>(m + (TY + 2) * LDW);
  }
};

static_assert(VjpHLds<LM_UTAB, 0>::UT_CAP == 23 && VjpHLds<LM_UTAB, 0>::SIZE == VjpHLds<LM_FAST, 0>::SIZE && VjpHLds<LM_FAST, 0>::UT_CAP == 0,
              "the staged patches of the U table fit the node-result region the tiles leave free: no extra LDS");

// One dual node of the CONTINUOUS-form H-VJP (VJP_lambda_dSIA/dH_continuous, adjoint.jl:442-553):
//   dlam = div(D grad lam) - avg(dD/dH) avg(q) + avg_y(dx(q beta gSx))/dx + avg_x(dy(q beta gSy))/dy
// on the interior, q = <grad S, grad lam> on the dual grid (:534-538), slopes unclamped, lambda raw.
// The divergence terms are linear in the node's D, q beta gS: k[c] is what the node adds to corner
// cell c; the product term needs the corner sums of alpha and q separately (a4 = alpha/4, q4 = q/4).
template <int LM, int NK = 0>
__device__ __forceinline__ void vjpHc_node(const GDev& g, const LawDev& L, const Pools& P, const double2 (*sHS)[LDW],
                                           const double (*sL)[LDW], int i0, int j0, int a, int b, double (&k)[4],
                                           double& a4, double& q4, const UtabTile ut = ODINN_UT_NONE) {
  const int gi = i0 - 1 + a, gj = j0 - 1 + b;
  k[0] = k[1] = k[2] = k[3] = 0.0;
  a4 = 0.0; q4 = 0.0;
  if (gi < 0 || gi > g.nx - 2 || gj < 0 || gj > g.ny - 2) return;
  const double2* p = &sHS[b][a];
  const double* pl = &sL[b][a];
  const double2 c00 = p[0], c10 = p[1], c01 = p[LDW], c11 = p[LDW + 1];
  const double l00 = pl[0], l10 = pl[1], l01 = pl[LDW], l11 = pl[LDW + 1];
  const double dxl = c10.y - c00.y, dxu = c11.y - c01.y, dyl = c01.y - c00.y, dyr = c11.y - c10.y;
  const double gx = (dxl + dxu) * g.hinv_dx, gy = (dyl + dyr) * g.hinv_dy;
  const double Hb = 0.25 * ((c00.x + c10.x) + (c01.x + c11.x));
  double An = g.A;
  if (g.use_Afield) An = P.Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
  double al, be, sp;
  const double D = node_D<true, LM, NK>(g, L, Hb, gx * gx + gy * gy, An, al, be, sp, ut);
  const double mxl = l10 - l00, mxu = l11 - l01, myl = l01 - l00, myr = l11 - l10;
  const double q = fma(g.hinv_dx2, fma(dxu, mxu, dxl * mxl), g.hinv_dy2 * fma(dyr, myr, dyl * myl));
  const double wx = D * g.hinv_dx2, wy = D * g.hinv_dy2;
  const double px = g.hinv_dx * (q * be * gx), py = g.hinv_dy * (q * be * gy);
  k[0] = fma(wx, mxl, wy * myl) + px + py;    // cell SW of the node (node = its NE corner)
  k[1] = fma(-wx, mxl, wy * myr) - px + py;   // cell SE
  k[2] = fma(wx, mxu, -wy * myl) + px - py;   // cell NW
  k[3] = fma(-wx, mxu, -wy * myr) - px - py;  // cell NE
  a4 = 0.25 * al;
  q4 = 0.25 * q;
}

// Stage the table patches of this tile's nodes in LDS (tabulated U law; see UtabTile).  Every thread passes the (up to RPT + 1)
// nodes it will evaluate; the workgroup reduces the patch rectangle [ih0, ih1] x [is0, is1] of the nodes that carry ice, and, when it
// has at most `cap` patches, copies it to `buf` (18 double2 per patch, row-major in (ih, is)).  scratch: 4 * NW ints of LDS.  Contains
// two barriers, the second one after the copy; the tiles must be complete (synchronised) on entry.
__device__ __forceinline__ UtabTile utab_stage(const GDev& g, const LawDev& L, const double2 (*sHS)[LDW], int i0, int j0, int tx, int ty,
                                               int ea, int eb, bool extra, double2* buf, int cap, int* scratch, int (&slot)[RPT + 1],
                                               double (&pu)[RPT + 1], double (&pv)[RPT + 1]) {
  int lo_h = 1 << 30, hi_h = -1, lo_s = 1 << 30, hi_s = -1;
  int nih[RPT + 1], nis[RPT + 1];
#pragma unroll
  for (int m = 0; m <= RPT; ++m) {
    const int a = m < RPT ? tx : ea, b = m < RPT ? ty + NW * m : eb;
    const int gi = i0 - 1 + a, gj = j0 - 1 + b;
    nih[m] = 0; nis[m] = 0; pu[m] = 0.0; pv[m] = 0.0;
    if ((m < RPT || extra) && gi >= 0 && gi <= g.nx - 2 && gj >= 0 && gj <= g.ny - 2) {
      double gx, gy, Hb;
      node_geom<LDW>(g, &sHS[b][a], gx, gy, Hb);
      if (Hb > 0.0) {
        utab_index(L, Hb, sqrt(gx * gx + gy * gy), nih[m], nis[m], pu[m], pv[m]);
        lo_h = min(lo_h, nih[m]); hi_h = max(hi_h, nih[m]); lo_s = min(lo_s, nis[m]); hi_s = max(hi_s, nis[m]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo_h = min(lo_h, __shfl_xor(lo_h, o, 64)); hi_h = max(hi_h, __shfl_xor(hi_h, o, 64));
    lo_s = min(lo_s, __shfl_xor(lo_s, o, 64)); hi_s = max(hi_s, __shfl_xor(hi_s, o, 64));
  }
  if (tx == 0) { scratch[4 * ty] = lo_h; scratch[4 * ty + 1] = hi_h; scratch[4 * ty + 2] = lo_s; scratch[4 * ty + 3] = hi_s; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    lo_h = min(lo_h, scratch[4 * w]); hi_h = max(hi_h, scratch[4 * w + 1]);
    lo_s = min(lo_s, scratch[4 * w + 2]); hi_s = max(hi_s, scratch[4 * w + 3]);
  }
  UtabTile T = ODINN_UT_NONE;
  T.pre = true;
  const int nhr = hi_h - lo_h + 1, nsr = hi_s - lo_s + 1;
  const bool fits = hi_h >= 0 && nhr * nsr <= cap;
#pragma unroll
  for (int m = 0; m <= RPT; ++m)  // (nodes without ice never evaluate the table: whatever their slot says)
    slot[m] = fits ? 18 * ((nih[m] - lo_h) * nsr + (nis[m] - lo_s)) : nih[m] * L.utab_ns + nis[m];
  if (fits) {  // (workgroup-uniform)
    const double2* __restrict__ tab = reinterpret_cast<const double2*>(L.utab);
    for (int idx = threadIdx.x; idx < 18 * nhr * nsr; idx += NT) {
      const int p = idx / 18, k = idx - 18 * p;
      const int ph = p / nsr, ps = p - ph * nsr;
      buf[idx] = tab[18 * ((long long)(lo_h + ph) * L.utab_ns + (lo_s + ps)) + k];
    }
    T.lds = buf; T.ih0 = lo_h; T.is0 = lo_s; T.nsr = nsr;
  }
  __syncthreads();
  return T;
}

// v[m] = (J_H(H)^T lam)[cell m of this thread] from tiles already in LDS (and synchronised).
// Phase A: every thread evaluates its (up to 5) nodes; phase B: every cell adds the four numbers
// its corner nodes left for it, masked by H > 0 (adjoint.jl:148).
template <int LM, int VJ = 0, int NK = 0>
__device__ __forceinline__ void vjpH_tile(const GDev& g, const LawDev& L, const Pools& P, double2* smem, int i0,
                                          int j0, const double (&ownH)[RPT], double (&v)[RPT]) {
  using S = VjpHLds<LM, VJ>;
  double2(*sHS)[LDW] = S::hs(smem);
  double(*sL)[LDW] = S::lam(smem);
  double2* cbase = S::ALIAS ? smem : smem + S::A_D2;
  double2(*sCa)[LDN] = reinterpret_cast<double2(*)[LDN]>(cbase);                    // {SW, SE}
  double2(*sCb)[LDN] = reinterpret_cast<double2(*)[LDN]>(cbase + (TY + 1) * LDN);   // {NW, NE}
  const int tx = threadIdx.x & 63, ty = wave_id();
  // the 65th column and 17th row of nodes: wavefront 0 takes the row, wavefront 1 the column
  const int ea = ty == 0 ? tx : TX, eb = ty == 0 ? TY : tx;
  const bool extra = ty == 0 || (ty == 1 && tx <= TY);
  UtabTile ut = ODINN_UT_NONE;
  int ut_slot[RPT + 1];
  double ut_u[RPT + 1], ut_v[RPT + 1];
  if constexpr (S::UT_CAP > 0)
    ut = utab_stage(g, L, sHS, i0, j0, tx, ty, ea, eb, extra, smem + S::A_D2 + 4, L.ut_nolds ? 0 : S::UT_CAP, reinterpret_cast<int*>(smem + S::A_D2),
                    ut_slot, ut_u, ut_v);
  auto utn = [&](int m) {  // the tile descriptor with node m's patch and coordinates
    UtabTile t = ut;
    if constexpr (S::UT_CAP > 0) { t.slot = ut_slot[m]; t.u = ut_u[m]; t.v = ut_v[m]; }
    return t;
  };
  if constexpr (VJ == 1) {
    double2(*sCc)[LDN] = reinterpret_cast<double2(*)[LDN]>(cbase + 2 * (TY + 1) * LDN);  // {alpha/4, q/4}
    if constexpr (S::ALIAS) {  // closed-form A laws: the node results are staged in registers, their LDS aliases the tiles
      double kk[RPT + 1][4], aa[RPT + 1], qq[RPT + 1];
#pragma unroll
      for (int m = 0; m < RPT; ++m) vjpHc_node<LM, NK>(g, L, P, sHS, sL, i0, j0, tx, ty + NW * m, kk[m], aa[m], qq[m], utn(m));
      if (extra) vjpHc_node<LM, NK>(g, L, P, sHS, sL, i0, j0, ea, eb, kk[RPT], aa[RPT], qq[RPT], utn(RPT));
      __syncthreads();
#pragma unroll
      for (int m = 0; m < RPT; ++m) {
        sCa[ty + NW * m][tx] = make_double2(kk[m][0], kk[m][1]);
        sCb[ty + NW * m][tx] = make_double2(kk[m][2], kk[m][3]);
        sCc[ty + NW * m][tx] = make_double2(aa[m], qq[m]);
      }
      if (extra) {
        sCa[eb][ea] = make_double2(kk[RPT][0], kk[RPT][1]);
        sCb[eb][ea] = make_double2(kk[RPT][2], kk[RPT][3]);
        sCc[eb][ea] = make_double2(aa[RPT], qq[RPT]);
      }
    } else {  // per-node MLP laws (targets :D_hybrid / :D): one node at a time, results in their own LDS
#pragma unroll 1
      for (int m = 0; m <= RPT; ++m) {
        const int a = m < RPT ? tx : ea, b = m < RPT ? ty + NW * m : eb;
        if (m < RPT || extra) {
          double k[4], a4, q4;
          vjpHc_node<LM, NK>(g, L, P, sHS, sL, i0, j0, a, b, k, a4, q4);
          sCa[b][a] = make_double2(k[0], k[1]);
          sCb[b][a] = make_double2(k[2], k[3]);
          sCc[b][a] = make_double2(a4, q4);
        }
      }
    }
    __syncthreads();
    const int c = tx + 1;
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int r = 1 + ty + NW * m;
      const int gi = i0 + tx, gj = j0 - 1 + r;
      v[m] = 0.0;
      if (gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2) {  // inn(dlam) only (adjoint.jl:551-552)
        const double2 sw = sCc[r - 1][c - 1], se = sCc[r - 1][c], nw = sCc[r][c - 1], ne = sCc[r][c];
        const double lin = (sCb[r - 1][c - 1].y + sCb[r - 1][c].x) + (sCa[r][c - 1].y + sCa[r][c].x);
        v[m] = lin - ((sw.x + se.x) + (nw.x + ne.x)) * ((sw.y + se.y) + (nw.y + ne.y));
      }
    }
    return;
  }
  if constexpr (S::ALIAS) {
    double kk[RPT + 1][4];
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      vjpH_node<LM, NK>(g, L, P, sHS, sL, i0, j0, tx, ty + NW * m, kk[m], utn(m));
      if constexpr (LM == LM_UTAB) asm volatile("" ::: "memory");  // one node's 36 table loads at a time (register budget)
    }
    if (extra) vjpH_node<LM, NK>(g, L, P, sHS, sL, i0, j0, ea, eb, kk[RPT], utn(RPT));
    __syncthreads();
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      sCa[ty + NW * m][tx] = make_double2(kk[m][0], kk[m][1]);
      sCb[ty + NW * m][tx] = make_double2(kk[m][2], kk[m][3]);
    }
    if (extra) {
      sCa[eb][ea] = make_double2(kk[RPT][0], kk[RPT][1]);
      sCb[eb][ea] = make_double2(kk[RPT][2], kk[RPT][3]);
    }
  } else {
#pragma unroll 1
    for (int m = 0; m <= RPT; ++m) {
      const int a = m < RPT ? tx : ea, b = m < RPT ? ty + NW * m : eb;
      if (m < RPT || extra) {
        double k[4];
        vjpH_node<LM, NK>(g, L, P, sHS, sL, i0, j0, a, b, k);
        sCa[b][a] = make_double2(k[0], k[1]);
        sCb[b][a] = make_double2(k[2], k[3]);
      }
    }
  }
  __syncthreads();
  const int c = tx + 1;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m;
    v[m] = 0.0;
    if (ownH[m] > 0.0)  // dlam .* (H .> 0)  (adjoint.jl:148)
      v[m] = (sCb[r - 1][c - 1].y + sCb[r - 1][c].x) + (sCa[r][c - 1].y + sCa[r][c].x);
  }
}

// Barrier (the tiles are complete) + vjpH_tile, or the exact shortcut v = 0 when no own cell of the
// tile carries ice (DiscreteVJP only: its result is masked by the cell's own H > 0).  Returns whether
// the stencil ran.
template <int LM, int VJ, int NK = 0>
__device__ __forceinline__ bool vjpH_tile_or_zero(const GDev& g, const LawDev& L, const Pools& P, double2* smem, int i0,
                                                  int j0, const double (&ownH)[RPT], double (&v)[RPT]) {
  if constexpr (VJ == 0 && !lm_is_nn(LM)) {  // (a conditional stencil makes the MLP variants spill, see tile_has_ice)
    bool any = false;
#pragma unroll
    for (int m = 0; m < RPT; ++m) any = any || ownH[m] > 0.0;
    if (!__syncthreads_or(any)) {
#pragma unroll
      for (int m = 0; m < RPT; ++m) v[m] = 0.0;
      return false;
    }
  } else {
    __syncthreads();
  }
  vjpH_tile<LM, VJ, NK>(g, L, P, smem, i0, j0, ownH, v);
  return true;
}

template <int MODE, int LM, int VJ = 0, int NK = 0>
__global__ __launch_bounds__(NT, (LM == LM_FAST && VJ == 0 ? 4 : 2)) void k_vjp_H(Pools P, LawDev L, AdjArgs A, int tile_base) {
  __shared__ double2 smem[VjpHLds<LM, VJ>::SIZE];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  if (MODE == 1 && A.dts[t4.x] == 0.0) {
    // reverse-Euler loop, row m of the per-glacier stop tables: this glacier has no stop m (fewer stops of its own than the
    // longest table of the batch: dt = 0 only there, own stops are strictly increasing) -- lambda passes through untouched
    const int gi_ = i0 + (threadIdx.x & 63);
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int gj = j0 + wave_id() + NW * m;
      if (gi_ < g.nx && gj < g.ny) {
        const long long id = g.off + gi_ + (long long)g.nx * gj;
        A.out[id] = A.lam[id];
      }
    }
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = 0.0;
    return;
  }
  double ownH[RPT], ownL[RPT], v[RPT];
  if (MODE == 0 && A.snaps) {
    const AdjState a = A.adj[t4.x];
    const double* Ha = A.snaps + (long long)a.seg * A.ntot;
    load_tile_HS2(Ha, P.B, g, i0, j0, VjpHLds<LM, VJ>::hs(smem), ownH, Ha + A.ntot, a.sitp[0]);
  } else {
    load_tile_HS2(A.H, P.B, g, i0, j0, VjpHLds<LM, VJ>::hs(smem), ownH);
  }
  load_tile_lam<NW, TY, VJ == 0>(A.lam, g, i0, j0, VjpHLds<LM, VJ>::lam(smem), ownL);
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double dt = 1.0, w = 0.0;
  long long roff = 0;
  // loss data of this thread's cells (MODE 1 at a data stop), fetched before the stencil phase so
  // that its latency hides behind it: hd[m] = H - Href where the mask is set, else 0
  double hd[RPT], hq[RPT];
  bool hm[RPT];
#pragma unroll
  for (int m = 0; m < RPT; ++m) { hd[m] = 0.0; hq[m] = 0.0; hm[m] = false; }
  if (MODE == 1) {
    dt = A.dts[t4.x];
    w = A.ws ? A.ws[t4.x] : 0.0;
    if (w != 0.0) {
      roff = (long long)A.refslot[t4.x] * A.ntot;
#pragma unroll
      for (int m = 0; m < RPT; ++m) {
        const int gj = j0 + ty + NW * m;
        if (gi < g.nx && gj < g.ny) {
          const long long id = g.off + gi + (long long)g.nx * gj;
          if (A.mask[roff + id]) {
            hm[m] = true;
            simple_loss_terms(ownH[m], A.Href[roff + id], A.h_log_eps, hd[m], hq[m]);
          }
        }
      }
    }
  }
  // DiscreteVJP: the result is masked by H > 0 of the cell itself (adjoint.jl:148), so a tile whose own
  // cells are all ice-free yields exactly 0 whatever its nodes would contribute: skip the stencil
  vjpH_tile_or_zero<LM, VJ, NK>(g, L, P, smem, i0, j0, ownH, v);
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  double lsum = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      if (MODE == 0) {
        A.out[id] = v[m];
      } else {
        double o = fma(dt, v[m], ownL[m]);
        if (hm[m]) {
          o = fma(w * 2.0 * Ninv, hd[m], o);
          lsum += hq[m];
        }
        A.out[id] = o;
      }
    }
  }
  if (MODE == 1) {
    const double tot = block_sum(lsum, red);
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = tot * w * Ninv;
  }
}

// =====================================================================================
// K5b: one RDPK3Sp35 stage of the reverse ODE of the continuous adjoint
//     dlam/dtau = J_H(H_itp(-tau))^T lam            (gradient.jl:316-324)
// = k_rk_stage with the H-VJP as right-hand side and H interpolated linearly in time between
// the two forward snapshots that bracket the stage time (weights from the controller).
// =====================================================================================
struct AdjStageArgs {
  const double* snaps;  // forward snapshots [n_snap][ntot]
  long long ntot;
  const AdjState* adj;
  const double* src;    // lambda ping-pong
  double* dst;
  double* S2;
  double* S3;
  double* E;
  double abstol, reltol;
};

// One cell of the reverse solve's post-step at a stop (k_adj_poststep and the self-controlled fused reverse step share it):
// lambda after the stop's terms.  Snapshot time t_j: lam += VJP_MB(lam, H_j - MB_j) (gradient.jl:413-425) and
// lam += w_j dl/dH(H_j) (:331-365) plus the time-aggregated cotangents, mass balance first (CallbackSet order :437) except at the
// very first stop (loss_first); a mass-balance time that is not a result stop (the reverse PeriodicCallback, :413-432):
// lam += VJP_MB(lam, H_pre) with H_pre = H_itp(t) - MB_t as the reference forms it -- H_itp from the RESULT snapshots,
// MB_t = (state after) - (state before) the mass balance of the forward solve at t (hidden snapshot slot, pre-MB slot).
// id: pooled cell index; w / roff: loss weight and reference offset of the stop's snapshot (0 / 0: no thickness term).
__device__ __forceinline__ double adj_post_cell(const GDev& g, const AdjState& a, const AdjPostArgs& A, const double* __restrict__ Bp,
                                                int gidx, int mb_now, int mb_slot, double w, long long roff, double Ninv,
                                                long long id, double l) {
  if (a.snapj >= 0) {
    double dl = 0.0;
    if (w != 0.0 && A.mask[roff + id]) {
      double d, q;
      simple_loss_terms(A.snaps[(long long)a.snapj * A.ntot + id], A.Href[roff + id], A.h_log_eps, d, q);
      dl = w * 2.0 * Ninv * d;
    }
    double dagg = 0.0;
    if (A.dh_coef) {
      const int q0 = A.dh_i0[gidx], q1 = A.dh_i1[gidx];
      if (q0 >= 0 && (a.snapj == q0 || a.snapj == q1) && A.snaps[(long long)q0 * A.ntot + id] > 1e-2)
        dagg = a.snapj == q1 ? A.dh_coef[gidx] : -A.dh_coef[gidx];
    }
    if (A.aggH) {
      const int sl = A.agg_slot[a.snapj];
      if (sl >= 0) dagg += A.aggH[(long long)sl * A.ntot + id];
    }
    if (A.loss_first) l += dl + dagg;
    if (mb_now && g.has_mb) {
      const double h = A.premb[(long long)mb_slot * A.ntot + id];
      double dmb;
      const double mb = mb_value(g, A.mb0[id], A.Sref ? A.Sref[id] : 0.0, h, Bp[id], dmb);
      const bool msk = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
      double vv = 0.0;
      if (msk) vv = dmb * l;
      if (msk && h + mb < 0.0) vv = -l;
      l += vv;
    }
    if (!A.loss_first) l += dl + dagg;
  } else if (a.pad > 0 && mb_now && g.has_mb) {
    const double ha = A.snaps[(long long)a.seg_stop * A.ntot + id];
    const double hb = A.snaps[(long long)(a.seg_stop + 1) * A.ntot + id];
    const double post = A.snaps[(long long)(a.pad - 1) * A.ntot + id];
    const double pre = A.premb[(long long)mb_slot * A.ntot + id];
    const double h = fma(a.s_stop, hb - ha, ha) - (post - pre);
    double dmb;
    const double mb = mb_value(g, A.mb0[id], A.Sref ? A.Sref[id] : 0.0, h, Bp[id], dmb);
    const bool msk = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
    double vv = 0.0;
    if (msk) vv = dmb * l;
    if (msk && h + mb < 0.0) vv = -l;
    l += vv;
  }
  return l;
}

// arguments of k_adj_fused_strip (sia2d_adj_fused.hpp): the five stages of a reverse step in one kernel
struct AdjFusedArgs {
  const double2* segs;  // non-null: {H_j, H_j+1 - H_j} interleaved per segment [n_snap - 1][ntot] (k_seg_pairs): one 16-byte
                        //   load per cell and stage instead of two 8-byte ones
  const double* snaps;  // forward snapshots [n_snap][ntot]
  long long ntot;
  const AdjState* adj;
  double* lam0;         // lambda ping-pong: the step reads lam[cur] and writes lam[1 - cur]; the controller
  double* lam1;         // flips cur on acceptance, so a rejected step needs no copy
  double* partF;        // error partials, FOX x FOYT tile table
  const int4* tilesF;
  double abstol, reltol;
  double* th_part;      // non-null (A-type laws): per-tile running sums of the theta-VJP at the quadrature nodes, formed in
                        //   stage 1 of the step that follows a node (same tile table)
  double* Gacc;         // non-null (gridded A; needs th_part and segs): the dual-grid accumulator gets the node weights there too
  const double* ytab;   // non-null: the Y law through its table (LM_YTAB, every glacier yt_fast) -- the kernel's YT instantiation
  int* ytab_over;
  int ytab_ni;
  // YT, `:Linear` gradient of the law, sort-free contraction (k_interp.hip): stage 1 of the step that follows a quadrature node EMITS the
  // node's (Hbar, qw dD/dY D_adjoint) pairs -- k_vjp_theta's emit mode without a launch of its own.  Every dual node of a glacier on a
  // node is written (zeros on ice-free tiles: no memset of the arrays), and the glacier's max Hbar / max |weight| are raised by
  // atomicMax on their bit patterns (zeroed per use; still zero = the glacier emitted nothing, the contraction skips it).
  double* emitH;
  double* emitV;
  unsigned long long* emit_amax;
  unsigned long long* emit_vmax;
  // UT: the U law of target :D through its batch-wide table (LM_UTAB): D = Hbar U(Hbar, |grad S|), alpha and beta by the reference's
  // central differences on the same bi-quintic patch (node_D<LM_UTAB>); overflow raises *ytab_over like the Y table
  const double* utab;
  int utab_nh, utab_ns;
  double ut_inv_h, ut_inv_s;
  // self-controlled reverse step (the kernel's SC instantiations; see k_adj_fused_strip): launch n decides attempt n - 1 itself.
  // State, AdjState and error partials alternate between two arrays from launch to launch (nobody reads what another
  // workgroup of the same launch writes); C.errpart must point at the partials the PREVIOUS launch wrote.
  const GState* gin;
  GState* gout;
  const AdjState* adj_in;
  AdjState* adj_out;
  CtrlArgs C;
  AdjPostArgs post;     // what k_adj_poststep would get (loss_first = 0, Hq = null)
};


template <int STAGE, int LM, int VJ = 0, int NK = 0>
__global__ __launch_bounds__(NT, (LM == LM_FAST && VJ == 0 ? 4 : 2)) void k_adj_stage(Pools P, LawDev L, AdjStageArgs A) {
  __shared__ double2 smem[VjpHLds<LM, VJ>::SIZE];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (gs->done) return;
  const GDev g = P.gd[t4.x];
  const AdjState a = A.adj[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const double dt = gs->dt;
  const double* __restrict__ X = A.src;
  if (STAGE == 1 && !gs->accepted) X = A.S3;  // rejected step: restart from uprev
  double ownH[RPT], ownL[RPT], v[RPT];
  {
    const double* Ha = A.snaps + (long long)a.seg * A.ntot;
    load_tile_HS2(Ha, P.B, g, i0, j0, VjpHLds<LM, VJ>::hs(smem), ownH, Ha + A.ntot, a.sitp[STAGE - 1]);
  }
  load_tile_lam<NW, TY, VJ == 0>(X, g, i0, j0, VjpHLds<LM, VJ>::lam(smem), ownL);
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double* __restrict__ Udst = A.dst;
  double* __restrict__ S2 = A.S2;
  double* __restrict__ S3 = A.S3;
  double* __restrict__ E = A.E;
  // the 3S*+ stream registers are fetched now so that their latency hides behind the stencil work (the U law's kernels:
  // AFTER it -- they are bound by the network's arithmetic and 24 more live registers across it mean spills)
  double pup[RPT], ptm[RPT], pe[RPT];
  auto fetch_streams = [&]() {
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int gj = j0 + ty + NW * m;
      pup[m] = ptm[m] = pe[m] = 0.0;
      if (STAGE > 1 && gi < g.nx && gj < g.ny) {
        const long long id = g.off + gi + (long long)g.nx * gj;
        if (STAGE == 2 || STAGE >= 4) pup[m] = __builtin_nontemporal_load(&S3[id]);
        if (STAGE != 2) ptm[m] = __builtin_nontemporal_load(&S2[id]);
        pe[m] = __builtin_nontemporal_load(&E[id]);
      }
    }
  };
  // (the U table: fetched after the stencil as well -- 24 registers less across it; fetching them early measured 206 vs 202 us, noise)
  constexpr bool STREAMS_LATE = NK == 4 || LM == LM_UTAB;
  if constexpr (!STREAMS_LATE) fetch_streams();
  vjpH_tile_or_zero<LM, VJ, NK>(g, L, P, smem, i0, j0, ownH, v);
  if constexpr (STREAMS_LATE) fetch_streams();
  constexpr int s = STAGE - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  double errsq = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double u = ownL[m];
      const double dtk = dt * v[m];
      if (STAGE == 1) {
        Udst[id] = fma(bt, dtk, u);
        if (gs->accepted) __builtin_nontemporal_store(u, &S3[id]);
        __builtin_nontemporal_store(bh * dtk, &E[id]);
      } else {
        const double up = pup[m];
        const double tmp_old = (STAGE == 2) ? up : ptm[m];
        const double tmp = fma(dl, u, tmp_old);
        double un = fma(g1, u, g2 * tmp);
        if (STAGE >= 4) un = fma(g3, up, un);
        un = fma(bt, dtk, un);
        Udst[id] = un;
        const double e = fma(bh, dtk, pe[m]);
        if (STAGE < 5) {
          if (dl != 0.0) __builtin_nontemporal_store(tmp, &S2[id]);
          __builtin_nontemporal_store(e, &E[id]);
        } else {
          const double err = (un - up) - e;
          const double sk = A.abstol + fmax(fabs(up), fabs(un)) * A.reltol;
          const double q = err / sk;
          errsq = fma(q, q, errsq);
        }
      }
    }
  }
  if (STAGE == 5) {
    const double tot = block_sum(errsq, red);
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w] = tot;
  }
}

// =====================================================================================
// K6: discrete theta-VJP (adjoint.jl:235-250).  For A-type laws dD/dtheta = spat * dA/dtheta
// with dA/dtheta independent of the state, so the kernel reduces  sum_nodes spat*Da  (scalar
// A: one partial per tile -> part[4*tile+2]) or accumulates  Gacc += scale*spat*Da  on the
// dual grid (gridded A).  For Y/U laws the per-node law gradient is contracted in place:
// part_theta[tile][k] = sum_nodes dlaw/dtheta_k * spat * Da.
// =====================================================================================
struct ThArgs {
  const double* H;
  const double* lam;
  const double* scales;  // per-glacier multiplier (dt) or null (=1)
  double* Gacc;          // dual pooled accumulator (gridded A) or null
  double* part_theta;    // [ntiles_total][P] for Y/U laws or null
  double* gscratch;      // thread-private gradient scratch [P][grid*NT]
  const double* lam_alt; // non-null: lambda of glacier g is in lam_alt where gs[g].cur == 1 (fused reverse step: per-glacier ping-pong)
  const double* snaps;   // non-null (continuous adjoint): H = H_itp at the stop each glacier just reached, formed in the tile
  const AdjState* adj;   //   loader from the two bracketing forward snapshots (seg_stop, s_stop) instead of being read from H
  long long ntot;
  int accum;             // A-type laws: add the tile's sum onto its partial slot instead of overwriting it
                         // (one reduction after a whole reverse solve instead of one per step)
  double* emitH;         // non-null (Y law, `:Linear` gradient interpolation, k_interp.hip): instead of backpropagating
  double* emitV;         //   per node, write Hbar and the node weight scale * spat * Da to these dual pooled arrays
                         //   (pre-zeroed by the caller: tiles that leave early contribute zeros)
  double* emitS;         // U law: |grad S| of the node as well (the second axis of its gradient interpolant)
};

template <int LM>
__global__ __launch_bounds__(NT) void k_vjp_theta(Pools P, LawDev L, ThArgs A, int tile_base) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double sL[TY + 2][LDW];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const double scale = A.scales ? A.scales[t4.x] : 1.0;
  constexpr bool nn_node = lm_is_nn(LM);
  if (nn_node && A.emitH && (scale == 0.0)) return;
  if (scale == 0.0) {  // this glacier contributes nothing now (e.g. not at a quadrature node)
    if (!nn_node) {
      if (threadIdx.x == 0 && !A.accum) P.part[4 * (long long)t4.w + 2] = 0.0;
    } else {
      for (int k = threadIdx.x; k < L.P; k += NT) A.part_theta[(long long)t4.w * L.P + k] = 0.0;
    }
    return;
  }
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double ownH[RPT], ownL[RPT];
  bool ice;
  if (A.snaps) {
    const AdjState a = A.adj[t4.x];
    const double* Ha = A.snaps + (long long)a.seg_stop * A.ntot;
    ice = load_tile_HS2(Ha, P.B, g, i0, j0, sHS, ownH, Ha + A.ntot, a.s_stop);
  } else {
    ice = load_tile_HS2(A.H, P.B, g, i0, j0, sHS, ownH);
  }
  // no ice anywhere on the tile and its halo: every owned node has Hbar = 0, so its weight
  // spat * Da vanishes identically (A-type and D_hybrid laws: spat ~ Hbar^(n+2); D law: Hbar) -- exact
  if (!__syncthreads_or(ice)) {
    if (!nn_node) {
      if (threadIdx.x == 0 && !A.accum) P.part[4 * (long long)t4.w + 2] = 0.0;
    } else if (!A.emitH) {
      for (int k = threadIdx.x; k < L.P; k += NT) A.part_theta[(long long)t4.w * L.P + k] = 0.0;
    }
    return;
  }
  load_tile_lam((A.lam_alt && P.gs[t4.x].cur) ? A.lam_alt : A.lam, g, i0, j0, sL, ownL);
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = wave_id();
  double acc = 0.0;
  const long long gstride = (long long)gridDim.x * NT;
  double* gth = A.gscratch ? A.gscratch + ((long long)blockIdx.x * NT + threadIdx.x) : nullptr;
  const bool emit = nn_node && A.emitH != nullptr;
  if (nn_node && !emit)
    for (int k = 0; k < L.P; ++k) gth[(long long)k * gstride] = 0.0;
  // owned nodes: lower-left cell is an interior-of-tile cell (a = tx+1, b = r)
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int b = 1 + ty + NW * m, a = tx + 1;
    const int gi = i0 - 1 + a, gj = j0 - 1 + b;
    if (gi <= g.nx - 2 && gj <= g.ny - 2) {
      double gx, gy, Hb;
      const double Da = node_Da<LDW>(g, &sHS[b][a], &sL[b][a], gj >= 1, gj + 1 <= g.ny - 2, gi >= 1,
                                     gi + 1 <= g.nx - 2, gx, gy, Hb);
      const double gS2 = gx * gx + gy * gy;
      double spat;
      if (LM == LM_FAST) {
        const double H2 = Hb * Hb;
        spat = g.Gam * (H2 * H2 * Hb) * gS2;
      } else if (LM == LM_POW) {
        spat = g.Gam * upow(Hb, g.n + 2.0) * spow(gS2, g.n - 1.0);
      } else if (L.kind == 3) {
        spat = g.Gam * upow(Hb, g.nH + 2.0) * spow(gS2, g.nS - 1.0);
      } else {
        spat = Hb > 0.0 ? Hb : 0.0;
      }
      const double wgt = scale * spat * Da;
      if (!nn_node) {
        acc += wgt;
        if (A.Gacc) A.Gacc[g.offd + gi + (long long)(g.nx - 1) * gj] += wgt;
      } else if (emit) {
        const long long q = g.offd + gi + (long long)(g.nx - 1) * gj;
        A.emitH[q] = Hb;
        A.emitV[q] = wgt;
        if (L.kind == 4) A.emitS[q] = sqrt(gS2);
      } else if (!(L.kind == 4 && Hb == 0.0)) {  // target_D_pure.jl:166-168 skips Hbar == 0
        // accumulate wgt * dlaw/dtheta into the thread-private scratch
        const double x0 = (L.kind == 3) ? g.T : Hb, x1 = (L.kind == 3) ? Hb : sqrt(gS2);
        mlp_grad(L, x0, x1, wgt, gth, gstride);
      }
    }
  }
  if (!nn_node) {
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 2] = A.accum ? P.part[4 * (long long)t4.w + 2] + tot : tot;
  } else if (!emit) {
    for (int k = 0; k < L.P; ++k) {
      const double tot = block_sum(gth[(long long)k * gstride], red);
      if (threadIdx.x == 0) A.part_theta[(long long)t4.w * L.P + k] = tot;
    }
  }
}

#ifdef ODINN_MISC_KERNELS
// fixed-order sum of per-tile partials of one glacier: out[g*stride_out + k]
__global__ __launch_bounds__(64) void k_sum_part(Pools P, int slot, double* out, int accumulate, int g0) {
  const int gidx = blockIdx.x + g0;
  const GDev g = P.gd[gidx];
  double s = 0.0;
  for (int k = threadIdx.x; k < g.ntiles; k += 64) s += P.part[4 * (long long)(g.tile0 + k) + slot];
  s = wave_sum(s);
  if (threadIdx.x == 0) { if (accumulate) out[gidx] += s; else out[gidx] = s; }
}
// the same for a whole reverse loop at once: partials of step j live at base + j*stride; steps are
// added in the order jhi, jhi-1, ..., jlo -- exactly the sequence of per-step k_sum_part(accumulate)
// launches it replaces (2 dependent launches fewer per reverse step)
__global__ __launch_bounds__(64) void k_sum_part_steps(Pools P, const double* base, long long stride, int jhi, int jlo,
                                                       int slot, double* out) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  double acc = out[gidx];
  for (int j = jhi; j >= jlo; --j) {
    const double* p = base + (long long)j * stride;
    double s = 0.0;
    for (int k = threadIdx.x; k < g.ntiles; k += 64) s += p[4 * (long long)(g.tile0 + k) + slot];
    s = wave_sum(s);
    acc += s;
  }
  if (threadIdx.x == 0) out[gidx] = acc;
}
// out[g*Pn + k] (+)= scale_g * sum_tiles part_theta[tile][k]
__global__ __launch_bounds__(64) void k_sum_part_theta(Pools P, const double* part_theta, int Pn, double* out,
                                                       int accumulate, int g0) {
  const int gidx = blockIdx.y + g0;
  const GDev g = P.gd[gidx];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int t = threadIdx.x; t < g.ntiles; t += 64) s += part_theta[(long long)(g.tile0 + t) * Pn + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    if (accumulate) out[(long long)gidx * Pn + k] += s; else out[(long long)gidx * Pn + k] = s;
  }
}

// =====================================================================================
// pointwise helpers (grid = tiles)
// =====================================================================================
// masked L2 loss partial of one snapshot (forward loss, inversion_utils.jl:425-461)
__global__ __launch_bounds__(NT) void k_loss(Pools P, const double* __restrict__ H, const double* __restrict__ Href,
                                             const unsigned char* __restrict__ mask, const double* ws,
                                             const int* refslot, long long ntot, double log_eps) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  const double w = ws[t4.x];
  double s = 0.0;
  if (w != 0.0) {
    const long long roff = (long long)refslot[t4.x] * ntot;
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int gj = j0 + ty + NW * m;
      if (gi < g.nx && gj < g.ny) {
        const long long id = g.off + gi + (long long)g.nx * gj;
        if (mask[roff + id]) {
          double d, q;
          simple_loss_terms(H[id], Href[roff + id], log_eps, d, q);
          s += q;
        }
      }
    }
  }
  const double tot = block_sum(s, red);
  if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = tot * w / ((double)g.nx * (double)g.ny);
}

// lam += VJP_MB(lam, H_pre)   (VJPs.jl:107-151), in place; flag per glacier.  gflag / gslot (nullable): per-glacier tables of
// the discrete reverse loop -- glacier g takes part iff gflag[g], and its pre-MB state is Hpre + gslot[g] * ntot
__global__ __launch_bounds__(NT) void k_mb_vjp(Pools P, const double* __restrict__ Hpre, const double* __restrict__ mb0,
                                               const double* __restrict__ Sref, const double* __restrict__ lam_in,
                                               double* __restrict__ lam_out, int add, int tile_base,
                                               const int* __restrict__ gflag, const int* __restrict__ gslot, long long ntot) {
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  if (gflag) {
    if (!gflag[t4.x]) return;  // (in place, add = 1: nothing to do)
    Hpre += (long long)gslot[t4.x] * ntot;
  }
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double h = Hpre[id], l = lam_in[id];
      double v = 0.0;
      if (g.has_mb) {
        double dmb;
        double mb = mb_value(g, mb0[id], Sref ? Sref[id] : 0.0, h, P.B[id], dmb);
        const bool mask = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
        if (mask) v = dmb * l;
        if (mask && h + mb < 0.0) v = -l;
      }
      lam_out[id] = add ? l + v : v;
    }
  }
}

// out = H + MB(H) and the applied increment (seam for odinn_mb_apply)
__global__ __launch_bounds__(NT) void k_mb_apply(Pools P, const double* __restrict__ H, const double* __restrict__ mb0,
                                                 const double* __restrict__ Sref, double* __restrict__ Hn,
                                                 double* __restrict__ MBout, int tile_base) {
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double h = H[id];
      double mb = 0.0;
      if (g.has_mb) {
        double dmb;
        mb = mb_value(g, mb0[id], Sref ? Sref[id] : 0.0, h, P.B[id], dmb);
        const bool mask = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
        if (!mask) mb = 0.0;
        if (mask && h + mb < 0.0) mb = -h;
      }
      Hn[id] = h + mb;
      MBout[id] = mb;
    }
  }
}

// ---- LossDhdt (src/losses/TimeAggregatedLosses.jl:38-113) ------------------------------------------------------
// forward: per tile  sum_{H0 > 1e-2} (H1 - H0)  and the count, H0 / H1 = snapshots dh_i0[g] / dh_i1[g] of glacier g
__global__ __launch_bounds__(NT) void k_dhdt_sums(Pools P, const double* __restrict__ snaps, const int* __restrict__ i0s,
                                                  const int* __restrict__ i1s, long long ntot, double* __restrict__ part2) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const int q0 = i0s[t4.x], q1 = i1s[t4.x];
  double s = 0.0, c = 0.0;
  if (q0 >= 0) {
    const GDev g = P.gd[t4.x];
    const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int gj = t4.z * TY + ty + NW * m;
      if (gi < g.nx && gj < g.ny) {
        const long long id = g.off + gi + (long long)g.nx * gj;
        const double h0 = snaps[(long long)q0 * ntot + id];
        if (h0 > 1e-2) { s += snaps[(long long)q1 * ntot + id] - h0; c += 1.0; }
      }
    }
  }
  s = block_sum(s, red);
  __syncthreads();
  c = block_sum(c, red);
  if (threadIdx.x == 0) { part2[2 * (long long)t4.w] = s; part2[2 * (long long)t4.w + 1] = c; }
}
// per glacier: dhdt = mean / (t1 - t0), loss += w (dhdt - ref)^2, coef = 2 w (dhdt - ref) / (N (t1 - t0))
__global__ __launch_bounds__(64) void k_dhdt_finish(Pools P, const double* __restrict__ part2, const int* __restrict__ i0s,
                                                    const double* __restrict__ dts, const double* __restrict__ refs, double w,
                                                    double* __restrict__ coef, double* __restrict__ lossacc) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  if (i0s[gidx] < 0) {
    if (threadIdx.x == 0) coef[gidx] = 0.0;
    return;
  }
  double s = 0.0, c = 0.0;
  for (int k = threadIdx.x; k < g.ntiles; k += 64) { s += part2[2 * (long long)(g.tile0 + k)]; c += part2[2 * (long long)(g.tile0 + k) + 1]; }
  s = wave_sum(s);
  c = wave_sum(c);
  if (threadIdx.x == 0) {
    const double dh = s / c / dts[gidx];  // c == 0 (no ice at t0): NaN, as mean() of an empty set in the reference
    const double d = dh - refs[gidx];
    coef[gidx] = 2.0 * w * d / (c * dts[gidx]);
    lossacc[gidx] += w * d * d;
  }
}
// discrete reverse loop, stop j: lambda_{j-1} += (+ at t1, - at t0) coef [H(t0) > 1e-2]   (gradient.jl:212-215)
__global__ __launch_bounds__(NT) void k_dhdt_cot(Pools P, double* __restrict__ lam, const double* __restrict__ snaps,
                                                 const int* __restrict__ i0s, const int* __restrict__ i1s,
                                                 const double* __restrict__ coef, int j, long long ntot) {
  const int4 t4 = P.tiles[blockIdx.x];
  const int q0 = i0s[t4.x], q1 = i1s[t4.x];
  if (q0 < 0 || (j != q0 && j != q1)) return;
  const double cf = j == q1 ? coef[t4.x] : -coef[t4.x];
  const GDev g = P.gd[t4.x];
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      if (snaps[(long long)q0 * ntot + id] > 1e-2) lam[id] += cf;
    }
  }
}

// A on the dual grid from a gridded temperature: Afield = post(MLP(T))  (hoisted law)
__global__ __launch_bounds__(NT) void k_law_field(LawDev L, const double* __restrict__ T, double* __restrict__ Aout,
                                                  long long n) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i < n) Aout[i] = mlp_eval_any(L, T[i], 0.0);
}
// the same with a compile-time one-input architecture (activations in registers, no run-time width predicates):
// the default A(T) net 1-3-10-3-1 (ML_utils.jl:31-36) and the "2 layers x 16 units" net of BASELINE configs[2]
struct ArchDefA { static constexpr int NL = 4, MAXW = 10; static constexpr int W[5] = {1, 3, 10, 3, 1}; static constexpr int A[4] = {1, 1, 1, 2}; };
struct Arch16A  { static constexpr int NL = 3, MAXW = 16; static constexpr int W[4] = {1, 16, 16, 1};   static constexpr int A[3] = {1, 1, 2}; };
template <class AR>
__global__ __launch_bounds__(NT) void k_law_field_fixed(LawDev L, const double* __restrict__ T, double* __restrict__ Aout,
                                                        long long n) {
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i < n) Aout[i] = mlp_eval_fixed<AR>(L, T[i], 0.0);
}

// theta-gradient of the hoisted gridded law: part_theta[block][k] = sum_i G[i]*dA/dtheta_k(T[i])
__global__ __launch_bounds__(NT) void k_law_field_grad(LawDev L, const double* __restrict__ T,
                                                       const double* __restrict__ G, long long n,
                                                       double* gscratch, double* part_theta) {
  __shared__ double red[NW];
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  const long long gstride = (long long)gridDim.x * NT;
  double* gth = gscratch + ((long long)blockIdx.x * NT + threadIdx.x);
  for (int k = 0; k < L.P; ++k) gth[(long long)k * gstride] = 0.0;
  if (i < n) {
    mlp_grad(L, T[i], 0.0, G[i], gth, gstride);
  }
  for (int k = 0; k < L.P; ++k) {
    const double tot = block_sum(gth[(long long)k * gstride], red);
    if (threadIdx.x == 0) part_theta[(long long)blockIdx.x * L.P + k] = tot;
  }
}
// ---- the same, wave-reduced: no thread-private accumulators in global memory ---------------------------------------
// k_law_field_grad keeps P accumulators per THREAD in global memory (2 x 8 B x P of traffic per dual node: 43 GB and
// 62 ms for 8 x 1024^2 nodes of the 321-parameter 1-16-16-1 net).  Here a wavefront takes 64 nodes at a time, every
// parameter's 64 contributions go through an LDS staging area (16 parameters per flush: 16 ds_write + 16 ds_read +
// 2 cross-row shuffles per lane, bank-conflict free on rows padded to 65) into the wavefront's P accumulators in LDS;
// chunks whose 64 weights are all zero (no ice: Gacc == 0) are skipped.  Summation order is fixed.
// part_theta[block][k] = sum over the block's nodes of G[i] * dA/dtheta_k(T[i]); dynamic LDS: NW x P accumulators + P ints
template <class AR, bool FIXED>
__global__ __launch_bounds__(NT) void k_law_field_grad_wave(LawDev L, const double* __restrict__ T, const double* __restrict__ G,
                                                            long long n, double* __restrict__ part_theta) {
  extern __shared__ double wg_dyn[];
  __shared__ double stage[NW][WG_SLOTS][WG_LD];
  double* accs = wg_dyn;
  int* order = reinterpret_cast<int*>(wg_dyn + (size_t)NW * L.P);
  const int lane = threadIdx.x & 63, w = wave_id();
  for (int k = threadIdx.x; k < NW * L.P; k += NT) accs[k] = 0.0;
  if (threadIdx.x == 0) mlp_grad_order(L, order);
  __syncthreads();
  const WaveAcc A{stage[w], order, accs + (size_t)w * L.P};
  const long long nchunk = (n + 63) / 64;
  for (long long c = (long long)blockIdx.x * NW + w; c < nchunk; c += (long long)gridDim.x * NW) {
    const long long i = c * 64 + lane;
    const bool ok = i < n;
    const double wgt = ok ? G[i] : 0.0;
    if (__builtin_amdgcn_ballot_w64(wgt != 0.0) == 0) continue;  // no node of the chunk carries weight
    mlp_grad_wave<AR, FIXED>(L, ok ? T[i] : 0.0, 0.0, wgt, A, lane);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < L.P; k += NT) {
    double s = 0.0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += accs[(size_t)ww * L.P + k];
    part_theta[(long long)blockIdx.x * L.P + k] = s;
  }
}
__global__ __launch_bounds__(64) void k_sum_rows(const double* part, int nrows, int Pn, double* out) {
  const int k = blockIdx.x;
  double s = 0.0;
  for (int t = threadIdx.x; t < nrows; t += 64) s += part[(long long)t * Pn + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[k] = s;
}

// value of the law on the dual grid of one glacier (seam odinn_eval_law)
__global__ __launch_bounds__(NT) void k_eval_law(Pools P, LawDev L, const double* __restrict__ U, double* __restrict__ out,
                                                 int gidx) {
  const GDev g = P.gd[gidx];
  const long long nd = (long long)(g.nx - 1) * (g.ny - 1);
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= nd) return;
  const int a = (int)(i % (g.nx - 1)), b = (int)(i / (g.nx - 1));
  auto Hc = [&](int ii, int jj) { double h = U[g.off + ii + (long long)g.nx * jj]; return h > 0.0 ? h : 0.0; };
  auto Sf = [&](int ii, int jj) { return P.B[g.off + ii + (long long)g.nx * jj] + Hc(ii, jj); };
  const double gx = 0.5 * ((Sf(a + 1, b) - Sf(a, b)) * g.inv_dx + (Sf(a + 1, b + 1) - Sf(a, b + 1)) * g.inv_dx);
  const double gy = 0.5 * ((Sf(a, b + 1) - Sf(a, b)) * g.inv_dy + (Sf(a + 1, b + 1) - Sf(a + 1, b)) * g.inv_dy);
  const double Hb = 0.25 * (Hc(a, b) + Hc(a + 1, b) + Hc(a, b + 1) + Hc(a + 1, b + 1));
  double v;
  if (L.kind == 3) v = mlp_eval_any(L, g.T, Hb);
  else if (L.kind == 4) v = mlp_eval_any(L, Hb, sqrt(gx * gx + gy * gy));
  else v = g.use_Afield ? P.Afield[g.offd + i] : g.A;
  out[i] = v;
}

// ---- table of the Y law (LM_YTAB, ytab_eval) -------------------------------------------------------------------------
// One thread per interval i of glacier blockIdx.y: the network at the six Chebyshev nodes s_j = cos((2 j + 1) pi / 12) of the
// interval, the interpolating quintic's monomial coefficients c = Vinv f (Vinv: inverse Vandermonde matrix of the nodes, formed
// on the host in long double), and the table's deviation from the network at six points between the nodes.  stat[0] = largest
// |table - network| / max(|network|, 1e-6 largest |network| seen so far ...) is not order-independent, so the kernel reduces the
// two ingredients separately: stat[0] = max |table - network| / |network| over points with |network| >= floor, stat[1] =
// max |table - network| over the others, stat[2] = max |network|; the host forms the verdict.  (Non-negative doubles order like
// their bit patterns: atomicMax on the 64-bit integers.)
struct YtabVinv { double v[6][6]; double node[6]; };
__global__ __launch_bounds__(256) void k_ytab_build(Pools P, LawDev L, YtabVinv V, double* __restrict__ tab, int ni, double floor_abs,
                                                    unsigned long long* __restrict__ stat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ni) return;
  const GDev g = P.gd[blockIdx.y];
  const double h = 1.0 / g.yt_inv_h;
  double f[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) f[j] = mlp_eval_any(L, g.T, ((double)i + 0.5 + 0.5 * V.node[j]) * h);
  double c[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) a = fma(V.v[k][j], f[j], a);
    c[k] = a;
  }
  double* __restrict__ o = tab + g.yt_off + 6 * (long long)i;
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = c[k];
  const double chk[6] = {-0.985, -0.55, -0.13, 0.31, 0.68, 0.995};
  double erel = 0.0, eabs = 0.0, ymax = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double sj = chk[j];
    const double y = mlp_eval_any(L, g.T, ((double)i + 0.5 + 0.5 * sj) * h);
    const double p = fma(fma(fma(fma(fma(c[5], sj, c[4]), sj, c[3]), sj, c[2]), sj, c[1]), sj, c[0]);
    const double e = fabs(p - y), ay = fabs(y);
    if (!(e == e)) erel = 1e300;  // NaN anywhere: no table
    if (ay >= floor_abs) erel = fmax(erel, e / ay);
    else eabs = fmax(eabs, e);
    ymax = fmax(ymax, ay);
  }
  atomicMax(&stat[0], (unsigned long long)__double_as_longlong(erel));
  atomicMax(&stat[1], (unsigned long long)__double_as_longlong(eabs));
  atomicMax(&stat[2], (unsigned long long)__double_as_longlong(ymax));
}

// out[i] = ((out[i] + slot_0[i]) + slot_1[i]) + ... : the slots of the overlapped `:Linear` contractions added in the order of the stops
__global__ void k_sum_slots(long long n, int nslots, const double* __restrict__ slots, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = out[i];
  for (int q = 0; q < nslots; ++q) s += slots[(long long)q * n + i];
  out[i] = s;
}

// ---- table of the U law (LM_UTAB, utab_eval): one thread per patch; stat as in k_ytab_build --------------------------------------
__global__ __launch_bounds__(64) void k_utab_build(LawDev L, YtabVinv V, double* __restrict__ tab, int nh, int ns, double floor_abs,
                                                   unsigned long long* __restrict__ stat) {
  const long long pidx = (long long)blockIdx.x * 64 + threadIdx.x;
  if (pidx >= (long long)nh * ns) return;
  const int ih = (int)(pidx / ns), is = (int)(pidx - (long long)ih * ns);
  const double hh = 1.0 / L.ut_inv_h, hs = 1.0 / L.ut_inv_s;
  double c[6][6];  // first F[j][k], then c[a][k] = sum_j V[a][j] F[j][k], then c[a][b] = sum_k c[a][k] V[b][k]
  for (int j = 0; j < 6; ++j)
    for (int k = 0; k < 6; ++k)
      c[j][k] = mlp_eval_any(L, ((double)ih + 0.5 + 0.5 * V.node[j]) * hh, ((double)is + 0.5 + 0.5 * V.node[k]) * hs);
  for (int k = 0; k < 6; ++k) {
    double t[6];
    for (int a = 0; a < 6; ++a) {
      double acc = 0.0;
      for (int j = 0; j < 6; ++j) acc = fma(V.v[a][j], c[j][k], acc);
      t[a] = acc;
    }
    for (int a = 0; a < 6; ++a) c[a][k] = t[a];
  }
  for (int a = 0; a < 6; ++a) {
    double t[6];
    for (int b = 0; b < 6; ++b) {
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) acc = fma(V.v[b][k], c[a][k], acc);
      t[b] = acc;
    }
    for (int b = 0; b < 6; ++b) c[a][b] = t[b];
  }
  double* __restrict__ o = tab + 36 * pidx;
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) o[6 * a + b] = c[a][b];
  const double chk[4] = {-0.97, -0.31, 0.23, 0.96};
  double erel = 0.0, eabs = 0.0, ymax = 0.0;
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < 4; ++k) {
      const double u = chk[j], v = chk[(k + j) & 3];
      const double y = mlp_eval_any(L, ((double)ih + 0.5 + 0.5 * u) * hh, ((double)is + 0.5 + 0.5 * v) * hs);
      double p = 0.0;
      for (int a = 5; a >= 0; --a) {
        const double qa = fma(fma(fma(fma(fma(c[a][5], v, c[a][4]), v, c[a][3]), v, c[a][2]), v, c[a][1]), v, c[a][0]);
        p = fma(p, u, qa);
      }
      const double e = fabs(p - y), ay = fabs(y);
      if (!(e == e)) erel = 1e300;
      if (ay >= floor_abs) erel = fmax(erel, e / ay);
      else eabs = fmax(eabs, e);
      ymax = fmax(ymax, ay);
    }
  atomicMax(&stat[0], (unsigned long long)__double_as_longlong(erel));
  atomicMax(&stat[1], (unsigned long long)__double_as_longlong(eabs));
  atomicMax(&stat[2], (unsigned long long)__double_as_longlong(ymax));
}

// ---- misc elementwise over pooled arrays ---------------------------------------------
__global__ void k_axpy(long long n, double a, const double* __restrict__ x, const double* __restrict__ y,
                       double* __restrict__ z) {  // z = y + a*x
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    z[i] = fma(a, x[i], y[i]);
}
// per-glacier z = y + dt_g * x on tiles
__global__ __launch_bounds__(NT) void k_axpy_g(Pools P, const double* __restrict__ x, const double* __restrict__ y,
                                               double* __restrict__ z) {
  const int4 t4 = P.tiles[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const double a = P.gs[t4.x].dt;
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      z[id] = fma(a, x[id], y[id]);
    }
  }
}
// scaled-norm partials for the Hairer-Wanner initial step:
//  slot0 = sum (u/sk)^2, slot1 = sum (f/sk)^2, slot2 = sum ((f1-f)/sk)^2 ; sk = abstol + |u| reltol
__global__ __launch_bounds__(NT) void k_initdt_norms(Pools P, const double* __restrict__ U, const double* __restrict__ F0,
                                                     const double* __restrict__ F1, double abstol, double reltol) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double u = U[id], f0 = F0[id];
      const double sk = abstol + fabs(u) * reltol;
      const double a = u / sk, b = f0 / sk;
      s0 = fma(a, a, s0);
      s1 = fma(b, b, s1);
      if (F1) { const double c = (F1[id] - f0) / sk; s2 = fma(c, c, s2); }
    }
  }
  const double t0 = block_sum(s0, red);
  const double t1 = block_sum(s1, red);
  const double t2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    double* p = P.part + 4 * (long long)t4.w;
    p[0] = t0; p[1] = t1; p[2] = t2;
  }
}
// phase 0: dt0 from d0,d1 ; phase 1: final dt from d1,d2   (ode_determine_initdt)
__global__ __launch_bounds__(64) void k_initdt_ctrl(Pools P, int phase, double tspan, double dtmax, double* dt0store) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < g.ntiles; k += 64) {
    const double* p = P.part + 4 * (long long)(g.tile0 + k);
    s0 += p[0]; s1 += p[1]; s2 += p[2];
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (threadIdx.x != 0) return;
  const double N = (double)g.nx * (double)g.ny;
  const double d0 = sqrt(s0 / N), d1 = sqrt(s1 / N);
  GState* gs = P.gs + gidx;
  if (phase == 0) {
    double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    dt0 = fmin(dt0, tspan);
    if (dtmax > 0.0) dt0 = fmin(dt0, dtmax);
    dt0store[gidx] = dt0;
    gs->dt = dt0;
  } else {
    const double dt0 = dt0store[gidx];
    const double d2 = sqrt(s2 / N) / dt0;
    const double dm = fmax(d1, d2);
    const double dt1 = dm <= 1e-15 ? fmax(1e-6, dt0 * 1e-3) : pow(0.01 / dm, 1.0 / 4.0);
    double dt = fmin(fmin(100.0 * dt0, dt1), tspan);
    if (dtmax > 0.0) dt = fmin(dt, dtmax);
    gs->dt = dt;
  }
}

// start of a solve: reset the integrator state; dt is already in gs->dt (given or from
// k_initdt_ctrl) and is clipped to the first stop here.
__global__ void k_begin(Pools P, int n, const double* tstops /* [imax][n] */, double dtmax, double dt_given) {
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= n) return;
  GState* gs = P.gs + gidx;
  const double t0 = tstops[gidx];
  double dt = dt_given > 0.0 ? dt_given : gs->dt;
  if (dtmax > 0.0 && dt > dtmax) dt = dtmax;
  const double rem = tstops[n + gidx] - t0;
  int clipped = 0;
  if (dt >= rem || fabs(rem - dt) <= 100.0 * 2.220446049250313e-16 * fabs(t0)) { dt = rem; clipped = 1; }
  gs->t = t0; gs->dt = dt; gs->e2 = 1.0; gs->e3 = 1.0; gs->EEst = 0.0;
  gs->accepted = 1; gs->at_stop = 0; gs->mb_now = 0; gs->mb_slot = 0;
  gs->done = 0; gs->istop = 1; gs->clipped = clipped; gs->cur = 0;
  gs->naccept = 0; gs->nreject = 0; gs->nonfinite = 0; gs->pad = 0; gs->snap_slot = 0; gs->pad2 = 0;
}


__global__ void k_set_dt(Pools P, int n, double dt) {
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx < n) P.gs[gidx].dt = dt;
}

// ---- continuous adjoint: reverse-solve bookkeeping --------------------------------------
// start of the reverse solve: the glacier sits on stop 0 (tau_0 = -t_{k-1}, the last snapshot), marked
// as "just reached" so that the post-step kernel adds the loss term of t_{k-1} and then the
// mass-balance VJP (gradient.jl:441-446 and PeriodicCallback(initial_affect = true) :431-432).
// seg starts in the last snapshot interval; k_begin later resets the integrator state proper.
__global__ void k_adj_begin(Pools P, int n, AdjState* adj, const int* n_snaps /* [n] */, double tau0, const int* mb_flags,
                            const int* mb_slots /* stop 0 of the reverse tables: [n] */) {
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= n) return;
  GState* gs = P.gs + gidx;
  AdjState* a = adj + gidx;
  const int n_snap = n_snaps[gidx], mb_flag = mb_flags[gidx], mb_slot = mb_slots[gidx];
  a->seg = n_snap - 2; a->seg_stop = n_snap - 2; a->snapj = n_snap - 1; a->pad = 0;
  a->qw = 0.0; a->s_stop = 1.0;
  for (int i = 0; i < 5; ++i) a->sitp[i] = 1.0;
  gs->t = tau0; gs->done = 0; gs->at_stop = 1; gs->cur = 0; gs->mb_now = mb_flag; gs->mb_slot = mb_slot;
  gs->accepted = 1;
}
// weights for the coming step (all_at_end: every weight at tau + dt, used for the second RHS of
// the initial-step heuristic)
__global__ void k_adj_itp(Pools P, int n, AdjState* adj, const double* tsnap, const int* n_snaps, int all_at_end) {
  const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= n) return;
  const GState* gs = P.gs + gidx;
  AdjState* a = adj + gidx;
  // the reverse solve starts in the last snapshot interval; the probe of the initial-step heuristic sits at tau_0 + dt_0,
  // which may lie below the snapshot under it (H_itp interpolates over ALL snapshots, gradient.jl:287): its own interval
  a->seg = n_snaps[gidx] - 2;
  if (all_at_end) {
    const double t = -(gs->t + gs->dt);
    while (a->seg > 0 && t < tsnap[(long long)a->seg * n + gidx]) a->seg--;
  }
  adj_stage_weights(a, tsnap, n, gidx, gs->t, gs->dt, all_at_end != 0);
}

// post-step of the reverse solve (pointwise, no-op for glaciers not at a stop):
//  * snapshot time t_j: lam += VJP_MB(lam, H_j - MB_j) (:413-425) and lam += w_j dl/dH(H_j) (:331-365),
//    mass balance first (CallbackSet order :437) except at the very first stop (loss_first);
//  * quadrature node: Hq = H_itp(t_node) for the theta-VJP that follows (:497-503).

// (one workgroup takes up to ADJ_POST_TILES tiles in turn once the launch has more than 512 workgroups left: most launches
//  of the reverse loop find no glacier on a stop, and a launch of ntiles workgroups that read two words and leave costs 12 us
//  at 8 x 1024^2 against 2 us for an eighth of them; small batches keep one tile per workgroup -- the stops that do have work
//  would otherwise serialise it, 4 alpine glaciers: 54 instead of 7 us per snapshot stop)
constexpr int ADJ_POST_TILES = 8;
__device__ __forceinline__ void adj_poststep_tile(const Pools& P, const AdjPostArgs& A, double* __restrict__ Ua,
                                                  double* __restrict__ Ub, const int4 t4) {
  const GState* gs = P.gs + t4.x;
  if (!gs->at_stop) return;
  const GDev g = P.gd[t4.x];
  const AdjState a = A.adj[t4.x];
  // a quadrature node whose H_itp nobody asked for: nothing to do on this tile
  if (a.snapj < 0 && !(a.pad > 0 && gs->mb_now && g.has_mb) && !(A.Hq && a.qw != 0.0 && !A.hq_snap_only)) return;
  double* __restrict__ U = gs->cur ? Ub : Ua;
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  double w = 0.0;
  long long roff = 0;
  if (a.snapj >= 0 && A.ws) {
    w = A.ws[(long long)a.snapj * A.G + t4.x];
    if (w != 0.0) roff = (long long)A.refslot[(long long)a.snapj * A.G + t4.x] * A.ntot;
  }
  const bool touches = a.snapj >= 0 || (a.pad > 0 && gs->mb_now && g.has_mb);
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      if (touches) U[id] = adj_post_cell(g, a, A, P.B, t4.x, gs->mb_now, gs->mb_slot, w, roff, Ninv, id, U[id]);
      if (A.Hq && ((a.qw != 0.0 && !A.hq_snap_only) || a.snapj >= 0)) {  // H_itp at the stop, for the velocity loss term (the theta-VJP forms it itself)
        const double ha = A.snaps[(long long)a.seg_stop * A.ntot + id];
        const double hb = A.snaps[(long long)(a.seg_stop + 1) * A.ntot + id];
        A.Hq[id] = fma(a.s_stop, hb - ha, ha);
      }
    }
  }
}
__global__ __launch_bounds__(NT) void k_adj_poststep(Pools P, AdjPostArgs A, double* __restrict__ Ua,
                                                     double* __restrict__ Ub, int ntiles, int per_wg) {
  const int t0 = blockIdx.x * per_wg;
  const int t1 = t0 + per_wg < ntiles ? t0 + per_wg : ntiles;
  for (int t = t0; t < t1; ++t) adj_poststep_tile(P, A, Ua, Ub, P.tiles[t]);
}


// ---- Tikhonov regularisation (Regularization.jl:92-126, 330-382) ---------------------------
// The reference's Laplacian  avg_y(diff_x(avg_y(diff_x a))) + avg_x(diff_y(avg_x(diff_y a)))  on the
// interior is the 3x3 stencil  [1 2 1]^T/4 (x) d_xx/dx^2 + d_yy/dy^2 (x) [1 2 1]/4 ; its VJP
// (diff/avg adjoints of inversion_utils.jl:3-66 composed) is the same symmetric stencil applied to
// the cotangent zeroed outside the interior, evaluated on every cell.
__device__ __forceinline__ double lap9(const double* __restrict__ a, int nx, int i, int j, double wx, double wy,
                                       bool interior_only_src, int ny) {
  auto at = [&](int ii, int jj) -> double {
    if (ii < 0 || ii >= nx || jj < 0 || jj >= ny) return 0.0;
    if (interior_only_src && (ii < 1 || ii > nx - 2 || jj < 1 || jj > ny - 2)) return 0.0;
    return a[ii + (long long)nx * jj];
  };
  double sx = 0.0, sy = 0.0;
#pragma unroll
  for (int q = -1; q <= 1; ++q) {
    const double w = q == 0 ? 2.0 : 1.0;
    sx = fma(w, (at(i + 1, j + q) - 2.0 * at(i, j + q)) + at(i - 1, j + q), sx);
    sy = fma(w, (at(i + q, j + 1) - 2.0 * at(i + q, j)) + at(i + q, j - 1), sy);
  }
  return fma(wx, sx, wy * sy);
}
// pass 1: r = 2 * mask * lap(a) on the interior (0 elsewhere), partial[block] = sum mask * lap(a)^2
__global__ __launch_bounds__(NT) void k_tikhonov_fwd(const double* __restrict__ a, const unsigned char* __restrict__ mask,
                                                     double* __restrict__ r, double* __restrict__ partial, int nx,
                                                     int ny, double wx, double wy) {
  __shared__ double red[NW];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * NW + (threadIdx.x >> 6);
  double sq = 0.0;
  if (i < nx && j < ny) {
    double v = 0.0;
    if (i >= 1 && i <= nx - 2 && j >= 1 && j <= ny - 2 && (!mask || mask[i + (long long)nx * j])) {
      const double l = lap9(a, nx, i, j, wx, wy, false, ny);
      v = 2.0 * l;
      sq = l * l;
    }
    r[i + (long long)nx * j] = v;
  }
  const double tot = block_sum(sq, red);
  if (threadIdx.x == 0) partial[blockIdx.x + (long long)gridDim.x * blockIdx.y] = tot;
}
// pass 2: grad = L^T r
__global__ __launch_bounds__(NT) void k_tikhonov_bwd(const double* __restrict__ r, double* __restrict__ grad, int nx,
                                                     int ny, double wx, double wy) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * NW + (threadIdx.x >> 6);
  if (i < nx && j < ny) grad[i + (long long)nx * j] = lap9(r, nx, i, j, wx, wy, true, ny);
}


// ---- VelocityRegularization (Regularization.jl:64-79,192-245) on the pooled arrays, all glaciers of a batch at once ----
// Tikhonov penalty on the Laplacian of the predicted surface speed: w_g sum_mask (lap V)^2, mask = is_in_glacier(H, dist)
// & (V > 0).  Three passes around the velocity kernels (k_surface_V before, k_surfV_vjp<2> after): prep (V, mask),
// lap (r = 2 mask lap V, loss partial), cot (dReg/dV = L^T r; dReg/dVx = dReg/dV Vx / V, written over Vx, Vy).
// w[g] == 0: the glacier has no term at this stop.
__global__ __launch_bounds__(NT) void k_vreg_prep(Pools P, const double* __restrict__ H, const double* __restrict__ vx,
                                                  const double* __restrict__ vy, const double* __restrict__ w, int dist,
                                                  double* __restrict__ Vabs, unsigned char* __restrict__ mask) {
  const int4 t4 = P.tiles[blockIdx.x];
  if (w[t4.x] == 0.0) return;
  const GDev g = P.gd[t4.x];
  const int tx = threadIdx.x & 63;
  const int gi0 = t4.y * TX, gi = gi0 + tx, ty = wave_id();
  const double* __restrict__ Hg = H + g.off;
  // is_in_glacier: ice on the cell and on every cell within Chebyshev distance `dist` (cells outside the grid count as
  // ice-free) -- the definition the thickness loss uses for its reference mask (odinn_set_reference).  A wavefront owns
  // RPT consecutive rows; the ice flags of a row are a ballot (plus one load by the first 2 dist lanes for the dist
  // columns either side of the wavefront's 64), its horizontal erosion is 2 dist shifts and ANDs of that scalar, and
  // every eroded row is formed once and ANDed into the rows of the wavefront within `dist` of it:
  // 2 (RPT + 2 dist) loads per wavefront instead of RPT (2 dist + 1)^2 (dist = 3: 261 -> 92 us at 8 x 1024^2).
  const int gjw = t4.z * TY + ty * RPT;  // first row of this wavefront
  unsigned long long inm[RPT];
#pragma unroll
  for (int m = 0; m < RPT; ++m) inm[m] = ~0ull;
  // halo column of this lane: lanes [0, dist) take gi0 - dist + tx, lanes [dist, 2 dist) take gi0 + 64 + (tx - dist)
  const int hi_col = tx < dist ? gi0 - dist + tx : gi0 + 64 + (tx - dist);
  const bool hi_ok = tx < 2 * dist && hi_col >= 0 && hi_col < g.nx;
  for (int rr = -dist; rr < RPT + dist; ++rr) {
    const int jj = gjw + rr;
    const bool rowok = jj >= 0 && jj < g.ny;  // wave-uniform
    unsigned long long E = 0ull;
    if (rowok) {
      const bool ice = gi < g.nx && Hg[gi + (long long)g.nx * jj] > 0.0;
      const bool hice = hi_ok && Hg[hi_col + (long long)g.nx * jj] > 0.0;
      const unsigned long long M = __builtin_amdgcn_ballot_w64(ice);
      const unsigned long long Hb = __builtin_amdgcn_ballot_w64(hice);
      const unsigned long long Lh = Hb & ((1ull << dist) - 1ull);           // bit k: column gi0 - dist + k
      const unsigned long long Rh = (Hb >> dist) & ((1ull << dist) - 1ull);  // bit k: column gi0 + 64 + k
      E = M;
      for (int a = 1; a <= dist; ++a) {
        // neighbour a to the east of lane t is bit t + a of M, or bit t + a - 64 of Rh; to the west bit t - a, or Lh's top
        E &= (M >> a) | (Rh << (64 - a));
        E &= (M << a) | (Lh >> (dist - a));
      }
    }
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
      const int d = rr - m;
      if (d >= -dist && d <= dist) inm[m] &= E;
    }
  }
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = gjw + m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double v = sqrt(vx[id] * vx[id] + vy[id] * vy[id]);
      Vabs[id] = v;
      mask[id] = (((inm[m] >> tx) & 1ull) && v > 0.0) ? 1 : 0;
    }
  }
}
__global__ __launch_bounds__(NT) void k_vreg_lap(Pools P, const double* __restrict__ Vabs, const unsigned char* __restrict__ mask,
                                                 const double* __restrict__ w, double* __restrict__ r) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const double wg = w[t4.x];
  if (wg == 0.0) {
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = 0.0;
    return;
  }
  const GDev g = P.gd[t4.x];
  const double wx = g.inv_dx * g.inv_dx * 0.25, wy = g.inv_dy * g.inv_dy * 0.25;
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
  double sq = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      double v = 0.0;
      if (gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2 && mask[id]) {
        const double l = lap9(Vabs + g.off, g.nx, gi, gj, wx, wy, false, g.ny);
        v = 2.0 * l;
        sq = fma(l, l, sq);
      }
      r[id] = v;
    }
  }
  const double tot = block_sum(sq, red);
  if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = tot * wg;
}
__global__ __launch_bounds__(NT) void k_vreg_cot(Pools P, const double* __restrict__ r, const double* __restrict__ Vabs,
                                                 const double* __restrict__ w, double* __restrict__ vx, double* __restrict__ vy) {
  const int4 t4 = P.tiles[blockIdx.x];
  if (w[t4.x] == 0.0) return;
  const GDev g = P.gd[t4.x];
  const double wx = g.inv_dx * g.inv_dx * 0.25, wy = g.inv_dy * g.inv_dy * 0.25;
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double gV = lap9(r + g.off, g.nx, gi, gj, wx, wy, true, g.ny);  // VJP of the Laplacian (Regularization.jl:372-382)
      const double v = Vabs[id];
      const double cx = v > 0.0 ? gV * vx[id] / v : 0.0, cy = v > 0.0 ? gV * vy[id] / v : 0.0;
      vx[id] = cx;
      vy[id] = cy;
    }
  }
}
// segs[j][i] = {H_j[i], H_j+1[i] - H_j[i]} for the n_seg = n_snap - 1 segments of a run (fused reverse step)
__global__ void k_seg_pairs(long long ntot, int n_seg, const double* __restrict__ snaps, double2* __restrict__ segs) {
  const long long n = ntot * n_seg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = snaps[i], b = snaps[i + ntot];
    segs[i] = make_double2(a, b - a);
  }
}
// out[g] += sum of the glacier's entries of a per-tile array on the FOX x FOYT (rows = 7), FOX x FOYT4 (rows = 4) or the forward
// kernel's FOX x FOYU (rows = 8) strip-tile table (fixed order)
__global__ __launch_bounds__(64) void k_sum_tilesFt(Pools P, const double* __restrict__ part, double* __restrict__ out, int rows) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  const int t0 = rows == 2 ? g.tile0Fw : rows == 4 ? g.tile0Fv : rows == 8 ? g.tile0Fu : g.tile0Ft;
  const int nt = rows == 2 ? g.ntilesFw : rows == 4 ? g.ntilesFv : rows == 8 ? g.ntilesFu : g.ntilesFt;
  double s = 0.0;
  for (int k = threadIdx.x; k < nt; k += 64) s += part[t0 + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[gidx] += s;
}
// out = a + s (b - a) on n entries (the time interpolant of two snapshots, load_tile_HS2's formula)
// out = H_seg + s (H_seg+1 - H_seg) with the segment and the weight PER GLACIER (every glacier interpolates between its own
// result snapshots, interpolate((t,), H, Gridded(Linear())), gradient.jl:287); seg < 0: the glacier is skipped
__global__ __launch_bounds__(NT) void k_lerp_g(Pools P, const double* __restrict__ snaps, long long ntot, const int* __restrict__ seg,
                                               const double* __restrict__ sw, double* __restrict__ out) {
  const int4 t4 = P.tiles[blockIdx.x];
  const int sg = seg[t4.x];
  if (sg < 0) return;
  const GDev g = P.gd[t4.x];
  const double s = sw[t4.x];
  const double* __restrict__ a = snaps + (long long)sg * ntot;
  const double* __restrict__ b = a + ntot;
  const int gi = t4.y * TX + (threadIdx.x & 63);
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + wave_id() + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double ha = a[id];
      out[id] = s != 0.0 ? fma(s, b[id] - ha, ha) : ha;  // (s = 0: the slot above may not exist)
    }
  }
}
__global__ void k_lerp(long long n, double s, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = fma(s, b[i] - a[i], a[i]);
}

// ---- continuous adjoint with a velocity loss: the theta-part of the loss at a quadrature node ------
// (gradient.jl:475-503: backward_loss at t_node with the reference velocities interpolated linearly in
// time, :291-301).  k_vref_itp writes the interpolated (Vabs, Vx, Vy) of the glaciers that just reached
// a node and the partial sums for LossV's normalisation (Losses.jl:323-326); k_vref_scale turns them
// into per-glacier scale and weight for k_surfV_vjp<1>.
__global__ __launch_bounds__(NT) void k_vref_itp(Pools P, VItpArgs A) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  double* pp = P.part + 4 * (long long)t4.w;
  const bool at_node = gs->at_stop && A.adj[t4.x].qw != 0.0;
  const long long q = (long long)(gs->istop - 1) * A.G + t4.x;
  const int sa = at_node ? A.slotA[q] : -1;
  if (sa < 0) {
    if (threadIdx.x == 0) { pp[0] = 0.0; pp[1] = 0.0; }
    return;
  }
  const GDev g = P.gd[t4.x];
  const long long oa = (long long)sa * A.ntot, ob = (long long)A.slotB[q] * A.ntot;
  const double w = A.sw[q];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double ss = 0.0, cnt = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = j0 + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      auto lerp = [&](const double* f) { const double a = f[oa + id]; return w == 0.0 ? a : fma(w, f[ob + id] - a, a); };
      const double va = lerp(A.Vabs), vx = lerp(A.Vxr), vy = lerp(A.Vyr);
      A.Vq[id] = va;
      A.Vq[A.ntot + id] = vx;
      A.Vq[2 * A.ntot + id] = vy;
      if (va > 0.0) { ss = fma(vx, vx, fma(vy, vy, ss)); cnt += 1.0; }
    }
  }
  const double t0 = block_sum(ss, red);
  const double t1 = block_sum(cnt, red);
  if (threadIdx.x == 0) { pp[0] = t0; pp[1] = t1; }
}
// per glacier: scale = 1/sqrt(mean_{mask} |Vref|^2) (or 1), weight = quadrature weight x wq (0 when not at a node)
__global__ __launch_bounds__(64) void k_vref_scale(Pools P, const AdjState* adj, const int* slotA, int G, int scale_loss,
                                                   double wq, double* scale_out, double* w_out) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  const GState* gs = P.gs + gidx;
  const bool at_node = gs->at_stop && adj[gidx].qw != 0.0 && slotA[(long long)(gs->istop - 1) * G + gidx] >= 0;
  double ss = 0.0, cnt = 0.0;
  if (at_node)
    for (int k = threadIdx.x; k < g.ntiles; k += 64) {
      ss += P.part[4 * (long long)(g.tile0 + k)];
      cnt += P.part[4 * (long long)(g.tile0 + k) + 1];
    }
  ss = wave_sum(ss);
  cnt = wave_sum(cnt);
  if (threadIdx.x == 0) {
    scale_out[gidx] = (scale_loss && cnt > 0.0 && ss > 0.0) ? 1.0 / sqrt(ss / cnt) : 1.0;
    w_out[gidx] = at_node ? adj[gidx].qw * wq : 0.0;
  }
}

// the per-glacier end of k_surfV_theta_node: out[g] += (quadrature weight x wq) x scale x (sum of the tiles' slot 3), with
// scale = 1/sqrt(mean_{mask} |Vref|^2) (or 1) from slots 0 and 1; fixed summation order
// coef (non-null): the glacier's coefficient for k_gacc_axpy, -(weight x scale), 0 when it is not at a node
__global__ __launch_bounds__(64) void k_vq_finish(Pools P, const AdjState* adj, const int* slotA, int G, int scale_loss,
                                                  double wq, double* out, double* coef) {
  const int gidx = blockIdx.x;
  const GDev g = P.gd[gidx];
  const GState* gs = P.gs + gidx;
  const bool at_node = gs->at_stop && adj[gidx].qw != 0.0 && slotA[(long long)(gs->istop - 1) * G + gidx] >= 0;
  if (!at_node) {
    if (coef && threadIdx.x == 0) coef[gidx] = 0.0;
    return;
  }
  double ss = 0.0, cnt = 0.0, gt = 0.0;
  for (int k = threadIdx.x; k < g.ntiles; k += 64) {
    const double* pp = P.part + 4 * (long long)(g.tile0 + k);
    ss += pp[0];
    cnt += pp[1];
    gt += pp[3];
  }
  ss = wave_sum(ss);
  cnt = wave_sum(cnt);
  gt = wave_sum(gt);
  if (threadIdx.x == 0) {
    const double sc = (scale_loss && cnt > 0.0 && ss > 0.0) ? 1.0 / sqrt(ss / cnt) : 1.0;
    out[gidx] = fma((adj[gidx].qw * wq) * sc, gt, out[gidx]);
    if (coef) coef[gidx] = -((adj[gidx].qw * wq) * sc);
  }
}

#endif  // ODINN_MISC_KERNELS

}  // namespace odinn
