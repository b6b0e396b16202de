// odinn_hip.hip -- host side of libodinn_hip.so: batch context, device-resident time
// loop, discrete-adjoint reverse loop, and the C ABI declared in include/odinn_hip.h.
// No torch types, no CPU fallback: every compute entry point needs a gfx950 device.
#include "../../include/odinn_hip.h"
#include "launch.hpp"
#include "sia2d_velocity.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// The step loops are chains of dependent launches, so launch latency is part of every step.  HIP_FORCE_DEV_KERNARG=1 (kernel
// arguments in device memory) is the HIP runtime's setting for that on MI300-class parts.  It is a PROCESS-wide runtime
// setting (it changes HIP for every other user of the runtime in the process), so the library does not touch it unless
// asked to: ODINN_REQUEST_DEV_KERNARG=1 makes the loader request it (never over a value the user has set, and without effect
// once the host application has initialised HIP).  On ROCm 7.2 / gfx950 it is the runtime's default already; switched off
// explicitly it costs 0.118 -> 0.122 ms per step at 8 x 1024^2 and 9.7 -> 12.3 ms on the continuous-adjoint gradient of 4
// alpine glaciers.
namespace {
struct OdinnLoadTimeSettings {
  OdinnLoadTimeSettings() {
    const char* e = std::getenv("ODINN_REQUEST_DEV_KERNARG");
    if (e && e[0] == '1') setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  }
} odinn_load_time_settings;
}  // namespace

using namespace odinn;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(x)                                                                            \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess)                                                                    \
      return fail(ODINN_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define CHK(x)                 \
  do {                         \
    int r_ = (x);              \
    if (r_ != ODINN_OK) return r_; \
  } while (0)

// ---- host MLP (hoisted scalar law: evaluated once per theta, Laws.jl:339-358) --------
double h_act(int c, double x) {
  switch (c) {
    case 1: return std::log1p(std::exp(-std::fabs(x))) + std::fmax(x, 0.0);
    case 2: { double t = std::exp(-std::fabs(x)); return x >= 0 ? 1.0 / (1.0 + t) : t / (1.0 + t); }
    case 3: { const double k = 0.7978845608028654; return 0.5 * x * (1.0 + std::tanh(k * (x + 0.044715 * x * x * x))); }
    case 4: return std::tanh(x);
    case 5: return std::fmax(x, 0.0);
    default: return x;
  }
}
double h_dact(int c, double x) {
  switch (c) {
    case 1: return h_act(2, x);
    case 2: { double s = h_act(2, x); return s * (1.0 - s); }
    case 3: { const double k = 0.7978845608028654; double u = k * (x + 0.044715 * x * x * x); double th = std::tanh(u);
              double du = k * (1.0 + 3 * 0.044715 * x * x); return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * du; }
    case 4: { double th = std::tanh(x); return 1.0 - th * th; }
    case 5: return x > 0 ? 1.0 : 0.0;
    default: return 1.0;
  }
}
// returns post(MLP(x)); if grad != null fills d out / d theta (P entries)
double h_mlp(const odinn_mlp_desc& m, const double* th, const double* x, double* grad) {
  std::vector<std::vector<double>> hs(m.n_layers + 1), zs(m.n_layers);
  std::vector<int> offs(m.n_layers + 1, 0);
  hs[0].resize(m.widths[0]);
  for (int i = 0; i < m.widths[0]; ++i)
    hs[0][i] = m.has_prescale ? (x[i] - m.pre_lo[i]) / (m.pre_hi[i] - m.pre_lo[i]) - 0.5 : x[i];
  for (int l = 0; l < m.n_layers; ++l) {
    const int nin = m.widths[l], nout = m.widths[l + 1], off = offs[l];
    zs[l].resize(nout);
    hs[l + 1].resize(nout);
    for (int o = 0; o < nout; ++o) {
      double acc = th[off + nin * nout + o];
      for (int i = 0; i < nin; ++i) acc = std::fma(th[off + o + nout * i], hs[l][i], acc);
      zs[l][o] = acc;
      hs[l + 1][o] = h_act(m.acts[l], acc);
    }
    offs[l + 1] = off + nout * (nin + 1);
  }
  const double y = hs[m.n_layers][0];
  double out = y, dpost = 1.0;
  switch (m.post_kind) {
    case ODINN_POST_AFFINE: out = m.post_lo + (m.post_hi - m.post_lo) * y; dpost = m.post_hi - m.post_lo; break;
    case ODINN_POST_EXPMAX: out = m.post_hi * std::exp((y - 1.0) / y); dpost = out / (y * y); break;
    case ODINN_POST_SCALE: out = m.post_hi * y; dpost = m.post_hi; break;
    default: break;
  }
  if (grad) {
    std::vector<double> gv(1, dpost), gn;
    for (int l = m.n_layers - 1; l >= 0; --l) {
      const int nin = m.widths[l], nout = m.widths[l + 1], off = offs[l];
      gn.assign(nin, 0.0);
      for (int o = 0; o < nout; ++o) {
        const double dz = gv[o] * h_dact(m.acts[l], zs[l][o]);
        grad[off + nin * nout + o] = dz;
        for (int i = 0; i < nin; ++i) {
          grad[off + o + nout * i] = dz * hs[l][i];
          gn[i] = std::fma(th[off + o + nout * i], dz, gn[i]);
        }
      }
      gv = gn;
    }
  }
  return out;
}
int mlp_nparams(const odinn_mlp_desc& m) {
  int p = 0;
  for (int l = 0; l < m.n_layers; ++l) p += m.widths[l + 1] * (m.widths[l] + 1);
  return p;
}

}  // namespace

// Kernel-schedule switch: the environment variable (a measurement / A-B override) if it is set, else the batch's
// odinn_schedule field, else -1 = the library's own measured rule.  Flags are '0' / '1' (any other digit string: atoi).
static int sched_val(int field, const char* env) {
  if (const char* e = std::getenv(env))
    if (e[0] >= '0' && e[0] <= '9') return std::atoi(e);
  return field;
}

struct odinn_batch {
  odinn_schedule sched = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, {0}};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int G = 0;
  std::vector<odinn_glacier_desc> descs;
  std::vector<GDev> gd;
  long long ntot = 0, ntotd = 0;
  int ntiles = 0;
  // device pools
  int* d_est = nullptr;        // per-glacier estimate of the steps still needed (written by the controller)
  std::vector<int> h_est;
  int4 *d_tiles = nullptr, *d_tiles_nat = nullptr, *d_tilesF = nullptr, *d_tilesFs = nullptr, *d_tilesFt = nullptr, *d_tilesFu = nullptr, *d_tilesFv = nullptr, *d_tilesFw = nullptr;
  int4* d_tilesD = nullptr;  // 62 x 62 tiles of the RHS-only strip kernel (all glaciers, XCD-banded)
  double* d_partD = nullptr; // per-tile max D of the CFL Euler step in that layout
  int ntilesD = 0;
  int ntilesF = 0, ntilesFs = 0, ntilesFt = 0, ntilesFu = 0, ntilesFv = 0, ntilesFw = 0;
  double *d_partF = nullptr, *d_partFs = nullptr, *d_partFt = nullptr, *d_partFu = nullptr, *d_partFv = nullptr, *d_partFw = nullptr;
  int sc_env() const {  // odinn_schedule::step_sc / ODINN_STEP_SC: -1 automatic, 0 off, 1 forced on
    const int v = sched_val(sched.step_sc, "ODINN_STEP_SC");
    return v < 0 ? -1 : (v == 1 ? 1 : 0);
  }
  static int n_cus() {  // compute units of the current device (256 on MI355X)
    static int n = 0;
    if (!n) {
      int dev = 0, v = 0;
      n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n;
  }
  // odinn_schedule::fused_tiles / ODINN_FUSED_TILES (s | l | t | u, or the digit): 0 automatic, 1 small, 2 large,
  // 3 strip with 7 rows per thread, 4 strip with 8 rows per thread
  int fused_override() const {
    if (const char* e = std::getenv("ODINN_FUSED_TILES")) {
      const int v = e[0] == 's' ? 1 : e[0] == 'l' ? 2 : e[0] == 't' ? 3 : e[0] == 'u' ? 4 : (e[0] >= '1' && e[0] <= '4') ? e[0] - '0' : 0;
      if (v) return v;
    }
    return sched.fused_tiles >= 1 && sched.fused_tiles <= 4 ? sched.fused_tiles : 0;
  }
  // which fused-step kernel / tile table: 0 = FOX x FOY row-interleaved kernel, 1 = FOX x FOYS latency tiles of the
  // same kernel (a workgroup walks 18 region rows instead of 50: batches too small to fill the 256 CUs), 2 = FOX x FOYT
  // strip kernel (integer-power law).  Measured crossover small <-> strip: ~100 strip tiles (one 512^2 glacier);
  // ODINN_FUSED_TILES=small|large|t overrides.
  // the Y law through its table where it is the integer-power law with Y(Hbar) in A's place (n_H = n_gradS = 3, no sliding on every
  // glacier): the strip kernels' YT instantiations take it (ODINN_YT_STRIP=0: the LDS-tile kernels of law mode LM_YTAB)
  bool ytab_strip() const {
    if (lm_kern() != LM_YTAB || gd.empty()) return false;
    for (const GDev& r : gd) if (!r.yt_fast) return false;
    static const bool off = std::getenv("ODINN_YT_STRIP") && std::getenv("ODINN_YT_STRIP")[0] == '0';
    return !off;
  }
  bool strip_law() const { return lm() == 0 || ytab_strip(); }
  int fused_kind() const {
    const int o = fused_override();
    if (o == 1) return 1;
    if (o == 2) return 0;
    if (o == 3) return strip_law() ? 2 : 0;
    if (o == 4) return strip_law() ? 3 : 0;
    // 2 / 3 = strip kernel with 7 / 8 rows per thread (54 x 46 / 54 x 54 tiles).  The kernel is VALU-bound, so a launch
    // lasts about (tiles on the busiest CU) x (rows per thread): 8 rows do less halo work per cell but quantise worse.
    // That model reproduces the measured order on 11 batch shapes (1 x 1024^2: 7 rows win 8 %, 2 x 1024^2: 8 rows win
    // 10 %, 8 x 512^2: 7 rows win 7 %, ties and everything large: 8 rows win 1..5 %).
    // Below ~100 strip tiles the 54 x 8 latency tiles win as a KERNEL, but the strip kernel can run the self-controlled
    // step loop (no controller / post-step launches), which wins as a STEP (4 alpine glaciers: 0.64 -> 0.55 ms for 25
    // steps)
    if (strip_law()) {
      if (ntilesFt < 96 && sc_env() == 0) return 1;
      const long cu = n_cus();
      return 8 * ((ntilesFu + cu - 1) / cu) <= 7 * ((ntilesFt + cu - 1) / cu) ? 3 : 2;
    }
    return ntilesF <= 256 ? 1 : 0;
  }
  const int4* fused_tiles() const { const int k = fused_kind(); return k == 3 ? d_tilesFu : k == 2 ? d_tilesFt : k == 1 ? d_tilesFs : d_tilesF; }
  double* fused_part() const { const int k = fused_kind(); return k == 3 ? d_partFu : k == 2 ? d_partFt : k == 1 ? d_partFs : d_partF; }
  int fused_ntiles() const { const int k = fused_kind(); return k == 3 ? ntilesFu : k == 2 ? ntilesFt : k == 1 ? ntilesFs : ntilesF; }
  int fused_ctrl() const { const int k = fused_kind(); return k == 3 ? 4 : k == 2 ? 3 : k == 1 ? 2 : 1; }
  GDev* d_gd = nullptr;
  GState* d_gs = nullptr;
  GState* d_gs2 = nullptr;   // second state array and second error-partial array of the self-controlled step loop
  double* d_part2 = nullptr;
  double *d_B = nullptr, *d_H0 = nullptr, *d_Afield = nullptr, *d_Tfield = nullptr, *d_Gacc = nullptr;
  double *d_part = nullptr, *d_U[2] = {nullptr, nullptr}, *d_S2 = nullptr, *d_S3 = nullptr, *d_E = nullptr;
  double *d_lam[2] = {nullptr, nullptr}, *d_tmpA = nullptr, *d_tmpB = nullptr;
  double *d_mb0 = nullptr, *d_Sref = nullptr;
  bool any_mb = false, any_sref = false;
  // snapshots
  int nstops_alloc = 0, nmb_alloc = 0;
  double *d_snaps = nullptr, *d_premb = nullptr;
  // reference thickness data
  std::vector<std::vector<double>> t_ref;  // per glacier
  int nref_alloc = 0;
  double* d_Href = nullptr;
  unsigned char* d_mask = nullptr;
  // surface-velocity data and loss selection
  std::vector<std::vector<double>> t_vref;                // per glacier
  // f_surface_velocity_factor of the simulation parameters (target :D: Velocity^ = U / f, target_D_pure.jl:206-255)
  double fV = 1.0;
  // surface-velocity path: every law has one -- A-type (target :A, closed form), U (target :D) and Y (target :D_hybrid, as the
  // reference writes it) through the per-node network in the velocity kernels
  bool vel_nn() const { return law_kind == ODINN_LAW_NN_U || law_kind == ODINN_LAW_NN_Y; }
  // Y law with the target's default `:Linear` interpolation of dY/dtheta: the velocity kernels emit (Hbar, node weight)
  // theta-part of the surface-velocity pull-backs through the target's `:Linear` law-gradient interpolation: the kernels emit
  // (Hbar, weight[, |grad S|]) per dual node for k_interp.hip instead of backpropagating per node (Y law: the knots of
  // create_interpolation, target_D_hybrid.jl:321-345; U law: LawU's node grid, dU/dtheta / f of target_D_pure.jl:179-193,247-255)
  bool vel_emit() const { return law_kind >= ODINN_LAW_NN_Y && grad_interp == ODINN_GRAD_INTERP_LINEAR; }
  bool vel_emit_U() const { return vel_emit() && law_kind == ODINN_LAW_NN_U; }
  // LossV's simple loss: 0 = L2Sum, > 0 = LogSum(eps) (component :abs only; Losses.jl:34-49,207-229); h_log_eps: LossH's
  double v_log_eps = 0.0, h_log_eps = 0.0;
  std::vector<std::vector<std::vector<double>>> v_edge;  // per glacier per slot: V_ref > 0 on the last row / column
  // data-only part of LossV on the last row / column of slot m (V_pred = 0 there by construction), divided by nx ny
  double v_const(int g, int m) const {
    if (!v_abs) return v_cxy[g][m];
    if (!(v_log_eps > 0.0)) return v_cabs[g][m];
    double s_ = 0.0;
    for (double va : v_edge[g][m]) { const double l = std::log(v_log_eps / (va + v_log_eps)); s_ += l * l; }
    return s_ / ((double)gd[g].nx * (double)gd[g].ny);
  }
  std::vector<std::vector<double>> v_scale, v_cxy, v_cabs;  // per glacier per slot: 1/sqrt(mean|Vref|^2),
                                                           // constant loss of the last row/column (:xy, :abs)
  int nvref_alloc = 0;
  double *d_Vabs = nullptr, *d_Vxr = nullptr, *d_Vyr = nullptr;
  int loss_kind = ODINN_LOSS_H, v_abs = 0, v_scale_loss = 1;
  int vjp_method = 0;  // 0 DiscreteVJP, 1 ContinuousVJP (H-VJP stencil used by the seams and both adjoints)
  double hv_scaling = 1.0;
  double *d_wv = nullptr, *d_vsc = nullptr;
  int* d_vslot = nullptr;
  std::vector<double> wv_h, vsc_h;  // [k][G] host copies
  std::vector<int> vslot_h;
  // law
  int law_kind = ODINN_LAW_CONST_A;
  odinn_mlp_desc mlp{};
  std::vector<double> theta;
  double* d_theta = nullptr;
  double* d_theta_pad = nullptr;  // run-time architectures: padded weight rows + biases (LawDev::theta_pad / bias_pad)
  int pad_rows = 0, pad_w = 16;
  int P = 0;
  double nH = -1, nS = -1;
  bool has_Afield_const = false;
  // LossDhdt (time-aggregated loss): per-glacier data, weight, device tables (stop indices, coefficients, tile partials)
  std::vector<double> dh_t0, dh_t1, dh_ref;
  double dhdt_weight = 0.0;
  int *d_dh_i0 = nullptr, *d_dh_i1 = nullptr;
  double *d_dh_coef = nullptr, *d_dh_part = nullptr, *d_dh_dt = nullptr, *d_dh_ref = nullptr;
  std::vector<int> dh_i0_h, dh_i1_h;
  bool dhdt_on() const {
    if (!(dhdt_weight != 0.0)) return false;
    for (size_t g = 0; g < dh_t0.size(); ++g) if (dh_t1[g] > dh_t0[g]) return true;
    return false;
  }
  // LossAvgV (time-aggregated loss): one velocity sample per glacier (pooled), per-stop weights dt_i / T, the time-averaged
  // velocity / its cotangents (d_avg[0..1]; [2..3]: V of one stop), dL/dH of every tLoss stop (d_aggH[slot])
  std::vector<double> av_t1, av_t2;
  double avgv_weight = 0.0, avgv_step = 1.0 / 12.0;
  int avgv_abs = 0;
  double *d_aVabs = nullptr, *d_aVx = nullptr, *d_aVy = nullptr, *d_avg = nullptr, *d_wA = nullptr, *d_aggH = nullptr;
  int* d_agg_slot = nullptr;
  unsigned char* d_av_on = nullptr;
  size_t wA_cap = 0, aggH_cap = 0, agg_slot_cap = 0;
  double* d_partTh = nullptr;  // fused reverse step: per-tile running sums of the theta-VJP at the quadrature nodes
  size_t partTh_cap = 0;
  double2* d_segs = nullptr;  // {H_j, H_j+1 - H_j} per segment for the fused reverse step
  size_t segs_cap = 0;
  std::vector<double> wA_h, wR_h;
  std::vector<int> agg_slot_h;
  int agg_nslots = 0;
  // VelocityRegularization: MultiLoss weight (0: off), distance to the margin, per-stop weights, mask scratch, node weights
  double vreg_weight = 0.0;
  int vreg_dist = 3;
  double *d_wR = nullptr, *d_wRq = nullptr, *d_swq = nullptr;
  int* d_sgq = nullptr;
  size_t wRq_cap = 0;
  unsigned char* d_vrm = nullptr;
  bool vreg_on() const {
    if (!(vreg_weight != 0.0)) return false;
    for (const auto& tv : t_vref) if (tv.size() >= 2) return true;
    return false;
  }
  bool avgv_on() const {
    if (!(avgv_weight != 0.0) || !d_aVabs) return false;
    for (size_t g = 0; g < av_t1.size(); ++g) if (av_t2[g] > av_t1[g]) return true;
    return false;
  }
  // `:Linear` interpolation of d law / d theta (Y law): knots of Hbar, see k_interp.hip
  int grad_interp = ODINN_GRAD_INTERP_NONE, n_interp_half = 75;
  double *d_nodeS = nullptr, *d_ucell = nullptr;  // U law: slope of the dual nodes, corner sums of the node-grid cells
  int* d_interp_err = nullptr;                    // sticky: a node fell outside the U law's interpolant (it does not extrapolate)
  double *d_nodeH = nullptr, *d_nodeV = nullptr, *d_sortH = nullptr, *d_sortV = nullptr, *d_knots = nullptr, *d_knotG = nullptr,
         *d_knotab = nullptr;
  int* d_knotM = nullptr;
  void* d_sorttmp = nullptr;
  // the per-glacier sort / knots / interval-sums sequences of one evaluation run side by side on interp_lanes streams, each
  // with its own scratch set (lane l: offset l * <size> into the arrays above)
  // Y law: the whole batch in one sequence of launches (launch_interp_theta_batch); scratch over the pooled dual nodes
  unsigned *d_ib_gid = nullptr, *d_ib_iota = nullptr, *d_ib_iA = nullptr, *d_ib_iB = nullptr, *d_ib_kA = nullptr, *d_ib_kB = nullptr;
  double *d_ib_sH = nullptr, *d_ib_sV = nullptr, *d_ib_knots = nullptr, *d_ib_ab = nullptr;
  int* d_ib_M = nullptr;
  void* d_ib_tmp = nullptr;
  size_t ib_tmp_bytes = 0;
  static constexpr int INTERP_LANES_MAX = 8;
  int interp_lanes = 0;
  long long interp_ndmax = 0;
  hipStream_t side[INTERP_LANES_MAX] = {};
  hipEvent_t ev_fork = nullptr, ev_join[INTERP_LANES_MAX] = {};
  // Y law's `:Linear` gradient interpolation inside the adjoints: the sort / knots / interval sums / knot backprop of a stop run on
  // lane streams of their own while the batch's stream goes on with the reverse steps (interp_async_*): per lane a set of emitted
  // node arrays and sort scratch (lane 0: the batch's own), ev_emit[l] = "lane l's node arrays are written", ev_done[l] = "... have
  // been contracted"; contribution q lands in its own slot of d_dthq and the slots are added onto d_dth in order at the join
  static constexpr int IA_LANES_MAX = 4, IA_SLOTS = 256;
  struct IaLane {
    double *sH = nullptr, *sV = nullptr, *knots = nullptr, *ab = nullptr;
    unsigned *iA = nullptr, *iB = nullptr, *kA = nullptr, *kB = nullptr;
    void* tmp = nullptr;
    int* M = nullptr;
    void* sel = nullptr;  // scratch of the sort-free contraction (launch_interp_theta_select)
  } ia_lane[IA_LANES_MAX];
  // node arrays: one SET more than lanes (set q % ia_sets, lane q % ia_lanes), so that the emitting kernel does not wait for the
  // contraction that is still running on the lane it will use
  static constexpr int IA_SETS_MAX = IA_LANES_MAX + 1;
  double *ia_nodeH[IA_SETS_MAX] = {}, *ia_nodeV[IA_SETS_MAX] = {};
  unsigned long long* ia_setmax[IA_SETS_MAX] = {};  // [2 G] per set: max Hbar, max |weight| per glacier, raised by the emitting kernel
  bool ia_emit_fused = false;                        // the node arrays of the current contraction carry those maxima
  hipEvent_t ev_set_done[IA_SETS_MAX] = {};
  bool ia_set_pending[IA_SETS_MAX] = {};
  int ia_sets = 0;
  // the dual nodes that carry ice in at least one snapshot of the current forward solve (launch_interp_active): the contractions of
  // this gradient sort / gather / sum these only
  unsigned char* ia_flags = nullptr;
  unsigned *ia_act = nullptr, *ia_gid_act = nullptr, *ia_nact_dev = nullptr;
  long long* ia_aoff = nullptr;
  void* ia_sel_tmp = nullptr;
  size_t ia_sel_bytes = 0;
  long long ia_nact = 0;
  bool ia_select = true;  // the sort-free contraction (ODINN_INTERP_SELECT=0: the radix sort of the active nodes)  // 0: the dense sequence
  bool interp_async = false;
  int ia_lanes = 0, ia_q = 0, ia_alloc = 0;
  bool ia_pending[IA_LANES_MAX] = {};
  double* d_dthq = nullptr;
  size_t dthq_cap = 0;
  hipStream_t ia_stream[IA_LANES_MAX] = {};
  hipEvent_t ev_emit[IA_LANES_MAX] = {}, ev_done[IA_LANES_MAX] = {};
  size_t sorttmp_bytes = 0, knotG_cap = 0;
  double *d_part_theta = nullptr, *d_gscratch = nullptr, *d_dth = nullptr;
  size_t part_theta_cap = 0, gscratch_cap = 0, dth_cap = 0;
  // solve bookkeeping.  Stop tables are PER GLACIER, as the reference builds them (gradient.jl:96-107,
  // inversion_utils.jl:487-495): ts_g[g] = the result stops of glacier g (own_stops[g] if odinn_set_glacier_stops gave any,
  // the tstops of the call otherwise); row m of every [kmax][G] table refers to glacier g's OWN m-th result stop, and a
  // glacier with fewer than kmax stops is idle in the rows it does not have.  The integrator's table (it_*) also holds the
  // mass-balance times that are not result stops (PeriodicCallback of :498-517: the integrator lands there, the state is
  // not part of the result); their post-MB states go to hidden snapshot slots behind the kmax result slots.
  std::vector<double> tstops;                   // the tstops of the last call (common table)
  std::vector<std::vector<double>> own_stops;   // [G] per-glacier override
  std::vector<std::vector<double>> ts_g;        // [G] result stops in effect
  int kmax = 0, imax = 0, nhid = 0, nmb_slots = 0;
  long long stops_version = 0;
  std::vector<int> mbf_res, mbs_res;            // [kmax][G]: mass balance applied at result stop m / its pre-MB slot
  std::vector<double> it_t;                     // [imax][G] integrator stops (padded with the last one)
  std::vector<int> it_mbf, it_mbs, it_snap, it_n;  // [imax][G] mb flag, pre-MB slot, snapshot slot; [G] number of stops
  bool ragged = false;                          // some glacier has a table of its own or a hidden stop
  int K() const { return kmax; }
  int nres(int g) const { return (int)ts_g[g].size(); }
  double* d_tstops = nullptr;
  int *d_mb_flag = nullptr, *d_mb_slot = nullptr, *d_nactive = nullptr;
  int *d_snapslot = nullptr, *d_nst = nullptr, *d_mbf_res = nullptr, *d_mbs_res = nullptr;
  double* d_dt0 = nullptr;
  double *d_dts = nullptr, *d_ws = nullptr, *d_lossacc = nullptr, *d_Gsum = nullptr;
  int* d_refslot = nullptr;
  int tab_cap = 0;
  double* d_partsteps = nullptr;  // [k][4*ntiles] per-step partials of the discrete reverse loop
  size_t partsteps_cap = 0;
  // continuous adjoint with a velocity loss
  int *d_rvA = nullptr, *d_rvB = nullptr, *d_zeroslot = nullptr;
  double *d_rvs = nullptr, *d_Vq = nullptr, *d_vscq = nullptr, *d_wvq = nullptr;
  size_t rv_cap = 0;
  // Tikhonov regulariser scratch
  double *d_rega = nullptr, *d_regr = nullptr, *d_regg = nullptr, *d_regp = nullptr;
  unsigned char* d_regm = nullptr;
  size_t reg_cap = 0, regp_cap = 0;
  // reverse (continuous-adjoint) solve tables
  double *d_rtau = nullptr, *d_rqw = nullptr, *d_tsnap = nullptr, *d_qw = nullptr;
  int *d_rsnap = nullptr, *d_rmbf = nullptr, *d_rmbs = nullptr, *d_rhid = nullptr, *d_nr = nullptr, *d_ksn = nullptr, *d_lastseg = nullptr;
  double* d_zerow = nullptr;
  AdjState* d_adj = nullptr;
  AdjState* d_adj2 = nullptr;  // second AdjState array of the self-controlled reverse step loop
  int rev_cap = 0, tsnap_cap = 0;
  std::vector<unsigned char> rev_host;  // byte image of the reverse tables on the device (loss_grad_continuous_impl)
  std::vector<double> gl_x, gl_w;       // Gauss-Legendre rule of the last continuous-adjoint call
  // key of the loss tables currently on the device (upload_loss_tables)
  const double* tab_key_ptr = nullptr;
  long long tab_key_ver = -1, refs_version = 0;
  std::vector<std::vector<double>> tab_key_stops;
  bool solved = false;
  bool gd_dirty = true;
  std::vector<double> last_loss_g, last_G_g;
  bool grad_field_valid = false;

  // law mode of the stencil kernels: 0 integer-power fast path (n==3, C==0 for every glacier),
  // 1 generic pow path, 2 inlined per-node MLP (Y / U laws)
  int lm() const {
    if (law_kind >= ODINN_LAW_NN_Y) {
      auto is = [&](int nl, std::initializer_list<int> w, std::initializer_list<int> a) {
        if (mlp.n_layers != nl) return false;
        int k = 0;
        for (int v : w) if (mlp.widths[k++] != v) return false;
        k = 0;
        for (int v : a) if (mlp.acts[k++] != v) return false;
        return true;
      };
      if (is(4, {2, 3, 10, 3, 1}, {1, 1, 1, 2})) return LM_NN_DEF;
      if (is(3, {2, 16, 16, 1}, {1, 1, 2})) return LM_NN_16;
      if (is(2, {2, 3, 1}, {1, 2})) return LM_NN_LIGHT;
      for (int l = 0; l < mlp.n_layers; ++l)
        if (mlp.widths[l] > 16) return LM_NN_WIDE;  // (a layer's INPUTS: the padded weight rows are 32 wide)
      return LM_NN;
    }
    for (const GDev& r : gd)
      if (!r.fast) return 1;
    return 0;
  }
  // ---- tabulated Y law (LM_YTAB; sia2d_device.hpp: ytab_eval, k_ytab_build) ----
  // Inside the forward solve and both adjoints the stencil kernels of a batch with the Y law read Y(Hbar) from a per-glacier
  // table instead of evaluating the network at every dual node and stage (odinn_schedule.law_table = 0 / ODINN_LAW_TABLE=0:
  // always the network).  The table is rebuilt whenever theta / the table range changes (refresh_gd) and is only used when its measured
  // deviation from the network is below YTAB_TOL; the seam calls (arbitrary fields from the caller) always take the network.
  double* d_ytab = nullptr;
  int* d_ytab_over = nullptr;
  unsigned long long* d_ytab_stat = nullptr;
  int ytab_ni = 1024;
  size_t ytab_cap = 0;
  std::vector<double> ytab_hmax, h0max;  // per glacier: table range (0: not chosen yet), largest initial / reference thickness seen
  bool ytab_ok = false;       // the table on the device belongs to the current theta / ranges and passed the check
  bool ytab_blocked = false;  // given up for this batch's current fields (range overflow after two widenings)
  int ytab_scope = 0;         // > 0 inside do_solve / odinn_loss_grad / odinn_loss_grad_continuous
  double ytab_err_rel = 0.0, ytab_err_abs = 0.0, ytab_ymax = 0.0;
  bool ytab_wanted() const {
    return (law_kind == ODINN_LAW_NN_Y || law_kind == ODINN_LAW_NN_U) && !ytab_blocked && sched_val(sched.law_table, "ODINN_LAW_TABLE") != 0;
  }
  // U law: ONE bivariate table U(Hbar, |grad S|) for the batch (LM_UTAB), the same life cycle -- built from the network at every
  // theta update, used while it agrees with it to YTAB_TOL, a solve that leaves [0, utab_hmax] x [0, utab_smax] is repeated with
  // twice the range (d_ytab / d_ytab_over / d_ytab_stat are shared with the Y law's table)
  int utab_nh = 128, utab_ns = 64;
  double utab_hmax = 0.0, utab_smax = 0.0;
  // law mode of the stencil kernels that evaluate the law per node and stage (forward stages, H-VJP, reverse stages)
  int lm_kern() const { return (ytab_scope > 0 && ytab_ok) ? (law_kind == ODINN_LAW_NN_U ? LM_UTAB : LM_YTAB) : lm(); }
  // dL/dA is accumulated on the dual grid when A is a field (hoisted NN or prescribed)
  bool wants_Gacc() const {
    return law_kind == ODINN_LAW_NN_A_GRIDDED || (law_kind == ODINN_LAW_CONST_A && has_Afield_const);
  }
  Pools pools(bool swz = true) const {
    Pools p;
    p.tiles = swz ? d_tiles : d_tiles_nat;
    p.gd = d_gd;
    p.gs = d_gs;
    p.B = d_B;
    p.Afield = d_Afield;
    p.part = d_part;
    return p;
  }
  LawDev lawdev() const {
    LawDev L{};
    L.kind = law_kind;
    L.n_layers = mlp.n_layers;
    L.has_pre = mlp.has_prescale;
    L.post_kind = mlp.post_kind;
    L.P = P;
    int mw = 1;
    for (int l = 0; l <= mlp.n_layers && l < 9; ++l) { L.widths[l] = mlp.widths[l]; mw = std::max(mw, mlp.widths[l]); }
    L.maxw = mw;
    for (int l = 0; l < mlp.n_layers; ++l) L.acts[l] = mlp.acts[l];
    for (int i = 0; i < 2; ++i) {
      L.pre_lo[i] = mlp.pre_lo[i];
      L.pre_inv[i] = mlp.has_prescale ? 1.0 / (mlp.pre_hi[i] - mlp.pre_lo[i]) : 1.0;
    }
    L.post_lo = mlp.post_lo;
    L.post_hi = mlp.post_hi;
    L.theta = d_theta;
    L.theta_pad = d_theta_pad;
    L.bias_pad = d_theta_pad ? d_theta_pad + (size_t)pad_rows * pad_w : nullptr;
    L.ytab = d_ytab;
    L.ytab_over = d_ytab_over;
    L.ytab_ni = ytab_ni;
    L.utab = d_ytab;
    L.utab_nh = utab_nh; L.utab_ns = utab_ns;
    L.ut_inv_h = utab_hmax > 0.0 ? (double)utab_nh / utab_hmax : 1.0;
    L.ut_inv_s = utab_smax > 0.0 ? (double)utab_ns / utab_smax : 1.0;
    return L;
  }
};

static int vel_theta_args(odinn_batch* b, VArgs& A, int g);
static int vel_theta_finish(odinn_batch* b, int g, bool accumulate, const Pools& P);
static int interp_prepare(odinn_batch* b, int g, bool linU);

namespace {

int use_dev(odinn_batch* b) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  HIPCHK(hipSetDevice(b->device));
  return ODINN_OK;
}

template <class T>
int dalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  // hipMemset on device memory is asynchronous on the NULL stream, with which the batch's non-blocking
  // stream does not synchronise: without the wait the zero fill can land AFTER the first
  // hipMemcpyAsync into the new buffer (seen as an all-zero tstops table, 1 run in ~15)
  HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
  HIPCHK(hipStreamSynchronize(nullptr));
  return ODINN_OK;
}
template <class T>
void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// derived per-glacier constants + hoisted scalar law; uploads d_gd when dirty
// Build the Y law's table for the current theta / ranges and decide whether the kernels may use it: the deviation from the
// network measured by k_ytab_build between the interpolation nodes must stay below YTAB_TOL relative to the law's value
// (values below 1e-3 of the largest value in the table: relative to that largest value).
constexpr double YTAB_TOL = 1e-12;
int ytab_refresh(odinn_batch* b) {
  const bool isU = b->law_kind == ODINN_LAW_NN_U;
  const size_t need = isU ? (size_t)36 * b->utab_nh * b->utab_ns : (size_t)b->G * 6 * b->ytab_ni;
  if (need > b->ytab_cap || !b->d_ytab) {
    dfree(b->d_ytab);
    b->d_ytab = nullptr;
    CHK(dalloc(&b->d_ytab, need));
    b->ytab_cap = need;
  }
  if (!b->d_ytab_over) {
    CHK(dalloc(&b->d_ytab_over, (size_t)1));
    HIPCHK(hipMemsetAsync(b->d_ytab_over, 0, sizeof(int), b->stream));
  }
  if (!b->d_ytab_stat) CHK(dalloc(&b->d_ytab_stat, (size_t)3));
  // two passes: the first finds the law's largest value on the table's range, the second measures the deviation -- relative to the
  // law's value where that is at least a thousandth of the largest, relative to the largest below (where the law all but vanishes,
  // e.g. exp((y - 1) / y) for y -> 0 far outside the range the network was scaled for, its RELATIVE curvature is unbounded and
  // nothing flows)
  double st[3] = {0.0, 0.0, 0.0};
  double floor_abs = 1e300;
  for (int pass = 0; pass < 2; ++pass) {
    HIPCHK(hipMemsetAsync(b->d_ytab_stat, 0, 3 * sizeof(unsigned long long), b->stream));
    if (isU) launch_utab_build(b->stream, b->lawdev(), b->d_ytab, b->utab_nh, b->utab_ns, floor_abs, b->d_ytab_stat);
    else launch_ytab_build(b->stream, b->pools(false), b->lawdev(), b->G, b->d_ytab, b->ytab_ni, floor_abs, b->d_ytab_stat);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(st, b->d_ytab_stat, sizeof(st), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    floor_abs = 1e-3 * st[2];
  }
  b->ytab_err_rel = st[0]; b->ytab_err_abs = st[1]; b->ytab_ymax = st[2];
  b->ytab_ok = st[0] <= YTAB_TOL && st[1] <= YTAB_TOL * st[2] && std::isfinite(st[2]) && st[2] > 0.0;
  static const bool verbose = std::getenv("ODINN_LAW_TABLE_VERBOSE") != nullptr;
  if (verbose)
    std::fprintf(stderr, "[odinn %s] %d x %d %s, rel %.3g abs %.3g max %.3g -> %s\n", isU ? "utab" : "ytab", isU ? b->utab_nh : b->G,
                 isU ? b->utab_ns : b->ytab_ni, isU ? "patches" : "intervals", st[0], st[1], st[2], b->ytab_ok ? "table" : "network");
  return ODINN_OK;
}
// did any node leave the table since the last call?  (clears the flag)
int ytab_overflowed(odinn_batch* b, bool* over) {
  *over = false;
  if (!b->d_ytab_over) return ODINN_OK;
  int f = 0;
  HIPCHK(hipMemcpyAsync(&f, b->d_ytab_over, sizeof(int), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (f) {
    HIPCHK(hipMemsetAsync(b->d_ytab_over, 0, sizeof(int), b->stream));
    *over = true;
  }
  return ODINN_OK;
}

int refresh_gd(odinn_batch* b) {
  if (!b->gd_dirty) return ODINN_OK;
  for (int g = 0; g < b->G; ++g) {
    const odinn_glacier_desc& d = b->descs[g];
    GDev& r = b->gd[g];
    const odinn_phys& ph = d.phys;
    r.dx = d.dx; r.dy = d.dy; r.inv_dx = 1.0 / d.dx; r.inv_dy = 1.0 / d.dy; r.eta0 = ph.eta0;
    r.hinv_dx = 0.5 / d.dx; r.hinv_dy = 0.5 / d.dy; r.hinv_dx2 = 0.5 / (d.dx * d.dx); r.hinv_dy2 = 0.5 / (d.dy * d.dy);
    r.n = ph.n; r.p = ph.p; r.q = ph.q; r.T = d.T;
    r.Gam = 2.0 * std::pow(ph.rho * ph.g, ph.n) / (ph.n + 2.0);
    r.Sc = ph.C * std::pow(ph.rho * ph.g, ph.p - ph.q);
    r.fast = (ph.n == 3.0 && r.Sc == 0.0 && ph.eta0 == 1.0) ? 1 : 0;
    r.nH = b->nH >= 0 ? b->nH : ph.n;
    r.nS = b->nS >= 0 ? b->nS : ph.n;
    r.minA = ph.minA; r.maxA = ph.maxA;
    if (b->law_kind == ODINN_LAW_NN_A_SCALAR) {
      r.A = h_mlp(b->mlp, b->theta.data(), &d.T, nullptr);
      r.use_Afield = 0;
    } else if (b->law_kind == ODINN_LAW_NN_A_GRIDDED) {
      r.A = 0.0;
      r.use_Afield = 1;
    } else if (b->law_kind == ODINN_LAW_CONST_A) {
      r.A = d.A;
      r.use_Afield = b->has_Afield_const ? 1 : 0;
    } else {
      r.A = 0.0;
      r.use_Afield = 0;
    }
  }
  const bool ytab = b->ytab_wanted();
  b->ytab_ok = false;
  if (ytab && b->law_kind == ODINN_LAW_NN_U) {
    b->h0max.resize(b->G, 0.0);
    if (!(b->utab_hmax > 0.0)) {
      double m = 0.0;
      for (int g = 0; g < b->G; ++g) m = std::max(m, b->h0max[g]);
      b->utab_hmax = std::max(200.0, 1.25 * m + 50.0);
      // slopes: twice the upper bound of the law's input scaling, at least 1 (a solve that meets a steeper node widens it)
      b->utab_smax = std::max(1.0, b->mlp.has_prescale ? 2.0 * b->mlp.pre_hi[1] : 1.0);
      if (const char* e = std::getenv("ODINN_LAW_TABLE_HMAX"))  // (test aid)
        if (std::atof(e) > 0.0) b->utab_hmax = std::atof(e);
    }
  } else if (ytab) {
    b->ytab_hmax.resize(b->G, 0.0);
    b->h0max.resize(b->G, 0.0);
    for (int g = 0; g < b->G; ++g) {
      // range of the table: generously above the thickest ice the glacier's data shows; a solve that leaves it raises the
      // overflow flag and is repeated with twice the range (do_solve)
      if (!(b->ytab_hmax[g] > 0.0)) {
        b->ytab_hmax[g] = std::max(200.0, 1.25 * b->h0max[g] + 50.0);
        if (const char* e = std::getenv("ODINN_LAW_TABLE_HMAX"))  // (test aid: a first range small enough to overflow)
          if (std::atof(e) > 0.0) b->ytab_hmax[g] = std::atof(e);
      }
      b->gd[g].yt_inv_h = (double)b->ytab_ni / b->ytab_hmax[g];
      b->gd[g].yt_off = (long long)g * 6 * b->ytab_ni;
      b->gd[g].yt_fast = (b->gd[g].fast && b->gd[g].nH == 3.0 && b->gd[g].nS == 3.0 && !std::getenv("ODINN_LAW_TABLE_NOFAST")) ? 1 : 0;
    }
  }
  HIPCHK(hipMemcpyAsync(b->d_gd, b->gd.data(), sizeof(GDev) * b->G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->gd_dirty = false;
  if (ytab) CHK(ytab_refresh(b));
  return ODINN_OK;
}

int refresh_law_field(odinn_batch* b) {
  if (b->law_kind != ODINN_LAW_NN_A_GRIDDED) return ODINN_OK;
  launch_law_field(b->stream, b->lawdev(), b->d_Tfield, b->d_Afield, b->ntotd);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

int check_g(odinn_batch* b, int g) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  if (g < 0 || g >= b->G) return fail(ODINN_ERR_ARG, "glacier index %d out of range [0,%d)", g, b->G);
  return ODINN_OK;
}

int ensure_theta_scratch(odinn_batch* b, int grid_blocks, bool thread_scratch = true) {
  const int P = std::max(b->P, 1);
  const size_t need_pt = (size_t)std::max(b->ntiles, grid_blocks) * P;
  if (need_pt > b->part_theta_cap) {
    dfree(b->d_part_theta);
    CHK(dalloc(&b->d_part_theta, need_pt));
    b->part_theta_cap = need_pt;
  }
  const size_t need_gs = thread_scratch ? (size_t)grid_blocks * NT * P : 0;
  if (need_gs > b->gscratch_cap) {
    dfree(b->d_gscratch);
    CHK(dalloc(&b->d_gscratch, need_gs));
    b->gscratch_cap = need_gs;
  }
  const size_t need_dth = (size_t)b->G * P;
  if (need_dth > b->dth_cap) {
    dfree(b->d_dth);
    CHK(dalloc(&b->d_dth, need_dth));
    b->dth_cap = need_dth;
  }
  return ODINN_OK;
}

// upload one glacier-sized host field into a pooled device array
int up_field(odinn_batch* b, int g, double* dpool, const double* h, bool dual = false) {
  const GDev& r = b->gd[g];
  const long long n = dual ? (long long)(r.nx - 1) * (r.ny - 1) : (long long)r.nx * r.ny;
  const long long off = dual ? r.offd : r.off;
  HIPCHK(hipMemcpyAsync(dpool + off, h, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ODINN_OK;
}
int down_field(odinn_batch* b, int g, const double* dpool, double* h, bool dual = false) {
  const GDev& r = b->gd[g];
  const long long n = dual ? (long long)(r.nx - 1) * (r.ny - 1) : (long long)r.nx * r.ny;
  const long long off = dual ? r.offd : r.off;
  HIPCHK(hipMemcpyAsync(h, dpool + off, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ODINN_OK;
}

// ---- launches -------------------------------------------------------------------------
int launch_dhdt(odinn_batch* b, const double* U, double* dH, int g /* -1: all */) {
  // integer-power law, whole batch (or a batch of one glacier): the strip-layout RHS kernel; ODINN_DHDT_STRIP=0 keeps k_dhdt
  const bool strip_on = sched_val(b->sched.dhdt_strip, "ODINN_DHDT_STRIP") != 0;
  if (strip_on && b->lm() == 0 && (g < 0 || b->G == 1)) {
    launch_dhdt_strip(b->ntilesD, b->gd[0].use_Afield, 1, b->stream, b->pools(true), b->d_tilesD, U, dH);
    HIPCHK(hipGetLastError());
    return ODINN_OK;
  }
  const Pools P = b->pools(g < 0);
  const int base = g < 0 ? 0 : b->gd[g].tile0, n = g < 0 ? b->ntiles : b->gd[g].ntiles;
  static void (*const tab[9])(int, hipStream_t, Pools, LawDev, const double*, double*, int) = {
      launch_dhdt_lm0, launch_dhdt_lm1, launch_dhdt_lm2, launch_dhdt_lm3, launch_dhdt_lm4, launch_dhdt_lm5, launch_dhdt_lm6, launch_dhdt_lm7, launch_dhdt_lm8};
  tab[b->lm_kern()](n, b->stream, P, b->lawdev(), U, dH, base);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

template <int S>
void launch_stage(odinn_batch* b, const Pools& P, const LawDev& L, const double* src, double* dst, double abstol,
                  double reltol) {
  static void (*const tab[9])(int, int, hipStream_t, Pools, LawDev, const double*, double*, double*, double*, double*,
                              double, double) = {launch_rk_stage_lm0, launch_rk_stage_lm1, launch_rk_stage_lm2,
                                                 launch_rk_stage_lm3, launch_rk_stage_lm4, launch_rk_stage_lm5, launch_rk_stage_lm6, launch_rk_stage_lm7, launch_rk_stage_lm8};
  tab[b->lm_kern()](S, b->ntiles, b->stream, P, L, src, dst, b->d_S2, b->d_S3, b->d_E, abstol, reltol);
}
// vj < 0: the batch's VJP method (odinn_set_vjp_method)
void launch_vjp_H(odinn_batch* b, int mode, int nblk, const Pools& P, const LawDev& L, const AdjArgs& A, int base,
                  int vj = -1) {
  static void (*const tab[9])(int, int, int, hipStream_t, Pools, LawDev, AdjArgs, int) = {
      launch_vjp_H_lm0, launch_vjp_H_lm1, launch_vjp_H_lm2, launch_vjp_H_lm3, launch_vjp_H_lm4, launch_vjp_H_lm5, launch_vjp_H_lm6, launch_vjp_H_lm7, launch_vjp_H_lm8};
  // integer-power law, DiscreteVJP, all glaciers at once: the strip-layout kernel on the 62 x 62 tile table
  // (sia2d_adj_fused.hpp: k_vjp_H_strip; ODINN_VJPH_STRIP=0 keeps the 64 x 16 LDS-tile kernel) ...
  // ... where its 62 x 62 tiles are reasonably full: batches of small glaciers (alpine: 96 x 80 ... 192 x 160 fill them to
  // 50-67 %) stay on the 64 x 16 tiles (measured: 512 alpine glaciers 53.9 k vs 48.9 k gradients/s); ODINN_VJPH_STRIP=1 forces it
  const int se = sched_val(b->sched.vjph_strip, "ODINN_VJPH_STRIP");
  const bool strip_on = se >= 0 ? se != 0 : (double)b->ntot >= 0.75 * (double)b->ntilesD * (DHDT_OX * DHDT_OY);
  const int vje = vj < 0 ? b->vjp_method : vj;
  if (strip_on && b->strip_law() && vje == ODINN_VJP_DISCRETE && !A.snaps && base == 0 && nblk == b->ntiles &&
      !(mode == 1 && b->h_log_eps > 0.0) &&  // (LossH with LogSum: the tile kernel carries that branch)
      (P.tiles == b->d_tiles || b->G == 1)) {
    AdjArgs As = A;
    if (b->lm() != 0) { As.ytab = L.ytab; As.ytab_over = L.ytab_over; As.ytab_ni = L.ytab_ni; }  // (the Y law through its table)
    launch_vjp_H_strip(mode, b->gd[0].use_Afield ? 1 : 0, b->ntilesD, b->stream, P, b->d_tilesD, As);
    return;
  }
  tab[b->lm_kern()](mode, vje, nblk, b->stream, P, L, A, base);
}
void launch_adj_stage(int lm, int vj, int stage, int nblk, hipStream_t st, const Pools& P, const LawDev& L,
                      const AdjStageArgs& A) {
  static void (*const tab[9])(int, int, int, hipStream_t, Pools, LawDev, AdjStageArgs) = {
      launch_adj_stage_lm0, launch_adj_stage_lm1, launch_adj_stage_lm2, launch_adj_stage_lm3, launch_adj_stage_lm4,
      launch_adj_stage_lm5, launch_adj_stage_lm6, launch_adj_stage_lm7, launch_adj_stage_lm8};
  tab[lm](stage, vj, nblk, st, P, L, A);
}
void launch_vjp_theta(odinn_batch* b, int nblk, const Pools& P, const LawDev& L, const ThArgs& A, int base) {
  static void (*const tab[7])(int, hipStream_t, Pools, LawDev, ThArgs, int) = {
      launch_vjp_theta_lm0, launch_vjp_theta_lm1, launch_vjp_theta_lm2, launch_vjp_theta_lm3, launch_vjp_theta_lm4,
      launch_vjp_theta_lm5, launch_vjp_theta_lm6};
  // integer-power A-type laws, all glaciers at once: the strip-layout reduction (k_vjp_theta_strip), under the same
  // tile-fullness rule as k_vjp_H_strip; ODINN_VJPTH_STRIP=0/1 forces the choice
  const int se = sched_val(b->sched.vjpth_strip, "ODINN_VJPTH_STRIP");
  const bool strip_on = se >= 0 ? se != 0 : (double)b->ntot >= 0.75 * (double)b->ntilesD * (DHDT_OX * DHDT_OY);
  // (the Y law through its table in emit mode: the same geometry factor, the node pairs written instead of reduced)
  const bool yt_emit = b->lm() != 0 && b->ytab_strip() && A.emitH && !A.emitS && !A.Gacc;
  if (strip_on && (yt_emit || (b->lm() == 0 && !A.emitH)) && base == 0 && nblk == b->ntiles && (P.tiles == b->d_tiles || b->G == 1)) {
    launch_vjp_theta_strip(A.Gacc ? 1 : 0, A.snaps ? 1 : 0, b->ntilesD, b->stream, P, b->d_tilesD, A);
    return;
  }
  tab[b->lm()](nblk, b->stream, P, L, A, base);
}

// integer-power law: the CFL Euler step runs in the strip layout (k_dhdt_strip<.., EULER>) with its own tile table and
// per-tile max-D partials; ODINN_DHDT_STRIP=0 keeps the 64 x 16 tile kernel
static bool euler_strip(const odinn_batch* b) {
  return sched_val(b->sched.dhdt_strip, "ODINN_DHDT_STRIP") != 0 && b->lm() == 0;
}
void launch_euler_cfl(odinn_batch* b, const Pools& P, const LawDev& L, const double* src, double* dst) {
  if (euler_strip(b)) {
    launch_euler_cfl_strip(b->ntilesD, b->gd[0].use_Afield, b->stream, P, b->d_tilesD, src, dst, b->d_partD);
    return;
  }
  static void (*const tab[9])(int, hipStream_t, Pools, LawDev, const double*, double*) = {
      launch_euler_cfl_lm0, launch_euler_cfl_lm1, launch_euler_cfl_lm2, launch_euler_cfl_lm3, launch_euler_cfl_lm4,
      launch_euler_cfl_lm5, launch_euler_cfl_lm6, launch_euler_cfl_lm7, launch_euler_cfl_lm8};
  tab[b->lm_kern()](b->ntiles, b->stream, P, L, src, dst);
}

// one RDPK3Sp35 step for all glaciers: 5 fused stage kernels.  parity p: state in U[p].
int launch_step(odinn_batch* b, int p, double abstol, double reltol) {
  const Pools P = b->pools(true);
  const LawDev L = b->lawdev();
  double* Ua = b->d_U[p];
  double* Ub = b->d_U[1 - p];
  launch_stage<1>(b, P, L, Ua, Ub, abstol, reltol);
  launch_stage<2>(b, P, L, Ub, Ua, abstol, reltol);
  launch_stage<3>(b, P, L, Ua, Ub, abstol, reltol);
  launch_stage<4>(b, P, L, Ub, Ua, abstol, reltol);
  launch_stage<5>(b, P, L, Ua, Ub, abstol, reltol);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// self-controlled reverse step (continuous adjoint): used up to this many tiles of the fused reverse step (see the rule's
// measurements at its use)
#ifndef ODINN_ADJ_SC_MAX_TILES
#define ODINN_ADJ_SC_MAX_TILES 640
#endif
#ifndef ODINN_NN_FUSED_MAX_TILES
#define ODINN_NN_FUSED_MAX_TILES 512  // latency (54 x 8) tiles; measured crossover, see pick_scheme
#endif
// scheme actually used: 1 = five per-stage kernels, 2 = one fused kernel per step
int pick_scheme(const odinn_batch* b, int requested) {
  int s = requested;
  if (s == 0) {
    const char* e = std::getenv("ODINN_SCHEME");
    if (e && (e[0] == '1' || e[0] == '2')) s = e[0] - '0';
  }
  // inlined-MLP laws (LawY / LawU): the fused step kernel evaluates the network once per dual node and stage inside the
  // stencil like the per-stage kernels do, but pays ~1.3x redundant evaluations in its halo; it wins where a step is
  // launch-bound (one launch instead of five), i.e. on batches that do not fill the GPU.  scheme = 2 (or
  // ODINN_SCHEME=2) forces it at any size.
  // Measured (2x16 Y law, us per step, per-stage vs fused): 4 alpine glaciers 201 vs 128, 1 x 512^2 216 vs 357,
  // 64 alpine 600 vs 755, 8 x 1024^2 2797 vs 3614 -- fused while its latency tiles do not fill the GPU
  if (lm_is_nn(b->lm_kern()) && s == 0) s = b->ntilesFs <= ODINN_NN_FUSED_MAX_TILES ? 2 : 1;
  if (s == 0) s = 2;
  return s;
}

int launch_fused_step(odinn_batch* b, double abstol, double reltol, int skip, const ScArgs* sc = nullptr,
                      double* part_override = nullptr) {
  const Pools P = b->pools(true);
  const LawDev L = b->lawdev();
  const int small = b->fused_kind();
  const int nblk = b->fused_ntiles();
  const int4* tiles = b->fused_tiles();
  double* part = part_override ? part_override : b->fused_part();
  if (small >= 2) {
    bool sq = true;  // square cells everywhere: the kernel instantiation without the dx / dy ratio (bit-identical, 2.5 % fewer VALU instructions)
    for (const GDev& r : b->gd) sq = sq && r.dx == r.dy;
    launch_rk_fused_strip(nblk, b->gd[0].use_Afield, small == 3 ? 8 : TRPT, b->stream, P, L, tiles, b->d_U[0], b->d_U[1], part, abstol,
                          reltol, skip, sc, sq ? 1 : 0, b->lm() != 0 ? 1 : 0);
  } else {
    static void (*const tab[9])(int, hipStream_t, Pools, LawDev, const int4*, double*, double*, double*, double, double, int, int) = {
        launch_rk_fused_lm0, launch_rk_fused_lm1, launch_rk_fused_lm2, launch_rk_fused_lm3, launch_rk_fused_lm4, launch_rk_fused_lm5, launch_rk_fused_lm6, launch_rk_fused_lm7, launch_rk_fused_lm8};
    tab[b->lm_kern()](nblk, b->stream, P, L, tiles, b->d_U[0], b->d_U[1], part, abstol, reltol, skip, small);
  }
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// self-controlled step loop (ScArgs): strip kernel, no mass balance.  Every workgroup repeats the controller's
// work (~2 us), which pays while launch latency is a large part of a step (1 x 512^2: 20.0 -> 15.6 us per step,
// 64 alpine glaciers: 40 -> 35, 2 x 1024^2: 38.8 -> 37.7) and costs slightly more than the two launches it saves on
// the largest batches (8 x 1024^2: +2 %, 512 alpine glaciers: +3 %).  ODINN_STEP_SC=0|1 overrides.
// Against the two-launch loop with the faster controller (end of round 2, us per step, two-launch vs self-controlled):
// 4 alpine 37-43 vs 34, 64 alpine (544 tiles) 50-53 vs 48, 1 x 512^2 19 vs 16, 4 x 512^2 (400) 25 vs 22, 1 x 1024^2 (361)
// 27 vs 26, 128 alpine (1088) 71-75 vs 72, 2 x 1024^2 (722) 37.5 vs 39.6, 4 x 1024^2 (1444) 50 vs 56: self-controlled
// up to 640 strip tiles.
static bool sc_mode(const odinn_batch* b, int scheme) {
  if (scheme != 2 || b->fused_kind() < 2 || (b->any_mb && b->gd[0].use_Afield)) return false;  // MB on load: constant-A path
  if (b->sc_env() >= 0) return b->sc_env() == 1;
  return b->fused_ntiles() <= 640;  // re-measured with the 5.9 us controller and the two-launch loop (tools/sc_probe3.py): see above
}
// Large batches (no self-controlled loop) without a mass balance: the strip step kernel stores the snapshot of a stop
// from the state it loads (ScArgs::snap_on_load), so a step is TWO dependent launches (step kernel, controller) instead of
// three; the snapshot of the LAST stop (after which no launch loads the state again) is a device copy at the end of the
// solve.  ODINN_SNAP_ON_LOAD=0 restores the post-step launch.
static bool snap_on_load_mode(const odinn_batch* b, int scheme, bool sc) {
  const int e = sched_val(b->sched.snap_on_load, "ODINN_SNAP_ON_LOAD");  // read per call: tests toggle it
  // (with a mass balance: the constant-A strip kernels apply it on load -- GState::pad bit 2, kept by the controller)
  return e != 0 && scheme == 2 && !sc && b->fused_kind() >= 2 && (!b->any_mb || !b->gd[0].use_Afield);
}
static int sc_buffers(odinn_batch* b) {
  if (!b->d_gs2) CHK(dalloc(&b->d_gs2, (size_t)b->G));
  const size_t need = (size_t)std::max(std::max(b->ntilesFt, b->ntilesFu), std::max(b->ntilesFv, b->ntilesFw));
  if (!b->d_part2) CHK(dalloc(&b->d_part2, need));
  return ODINN_OK;
}

int ensure_tables(odinn_batch* b, int n_stops) {
  if (n_stops > b->tab_cap) {
    dfree(b->d_tstops); dfree(b->d_mb_flag); dfree(b->d_mb_slot); dfree(b->d_snapslot); dfree(b->d_mbf_res); dfree(b->d_mbs_res);
    dfree(b->d_dts); dfree(b->d_ws); dfree(b->d_refslot); dfree(b->d_wv); dfree(b->d_vsc); dfree(b->d_vslot);
    CHK(dalloc(&b->d_wv, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_vsc, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_vslot, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_tstops, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_mb_flag, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_mb_slot, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_snapslot, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_mbf_res, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_mbs_res, (size_t)n_stops * b->G));
    if (!b->d_nst) CHK(dalloc(&b->d_nst, (size_t)b->G));
    CHK(dalloc(&b->d_dts, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_ws, (size_t)n_stops * b->G));
    CHK(dalloc(&b->d_refslot, (size_t)n_stops * b->G));
    b->tab_cap = n_stops;
    b->tab_key_ptr = nullptr;  // the loss tables went with the old buffers (a new buffer may get the old address)
  }
  return ODINN_OK;
}

// Per-glacier stop tables of a solve (see odinn_batch::ts_g).  tstops: the table of the call (every glacier without a table
// of its own); mb_times: the mass-balance times (PeriodicCallback(step_MB), inversion_utils.jl:498-517), in (t0, t1]; they need
// not be result stops.  Fills the host tables and uploads the integrator's.
int build_stop_tables(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times) {
  const int G = b->G;
  const double t0 = tstops[0], t1 = tstops[n_stops - 1];
  // validate everything BEFORE the batch's tables are touched: a rejected call leaves the previous solve's tables (and its
  // snapshots) consistent
  b->own_stops.resize(G);
  for (int g = 0; g < G; ++g) {
    const std::vector<double>& o = b->own_stops[g];
    if (!o.empty() && (o.front() != t0 || o.back() != t1))
      return fail(ODINN_ERR_ARG, "the stops of glacier %d span [%.12g, %.12g], the call's tstops [%.12g, %.12g]: every glacier "
                                 "covers the same tspan", g, o.front(), o.back(), t0, t1);
  }
  std::vector<double> mbt;
  if (b->any_mb)
    for (int m = 0; m < n_mb; ++m) {
      if (!(mb_times[m] > t0) || mb_times[m] > t1)
        return fail(ODINN_ERR_ARG, "mb_times[%d]=%g is not inside (tstops[0], tstops[end]]", m, mb_times[m]);
      if (m > 0 && !(mb_times[m] > mb_times[m - 1])) return fail(ODINN_ERR_ARG, "mb_times must be strictly increasing");
      mbt.push_back(mb_times[m]);
    }
  b->solved = false;  // from here on the tables no longer describe the stored snapshots; do_solve sets it again at its end
  b->tstops.assign(tstops, tstops + n_stops);
  b->ts_g.assign(G, std::vector<double>());
  b->ragged = false;
  int kmax = 0;
  for (int g = 0; g < G; ++g) {
    const std::vector<double>& o = b->own_stops[g];
    if (o.empty()) {
      b->ts_g[g] = b->tstops;
    } else {
      b->ts_g[g] = o;
      if (o != b->tstops) b->ragged = true;
    }
    kmax = std::max(kmax, (int)b->ts_g[g].size());
  }
  struct It { double t; int res, mb, mbs, snap; };
  std::vector<std::vector<It>> its(G);
  int imax = 0, nhid = 0, nmbs = 0;
  for (int g = 0; g < G; ++g) {
    const std::vector<double>& r = b->ts_g[g];
    const bool mbg = b->gd[g].has_mb != 0;
    size_t a = 0, c = 0;
    int hid = 0, nm = 0;
    while (a < r.size() || (mbg && c < mbt.size())) {
      const bool has_r = a < r.size(), has_m = mbg && c < mbt.size();
      It e{0.0, -1, 0, 0, 0};
      if (has_r && (!has_m || r[a] <= mbt[c])) {
        e.t = r[a]; e.res = (int)a;
        if (has_m && mbt[c] == r[a]) { e.mb = 1; ++c; }
        ++a;
      } else {
        e.t = mbt[c]; e.mb = 1; ++c;
      }
      if (e.mb) e.mbs = nm++;
      e.snap = e.res >= 0 ? e.res : kmax + hid++;
      its[g].push_back(e);
    }
    if (hid) b->ragged = true;
    imax = std::max(imax, (int)its[g].size()); nhid = std::max(nhid, hid); nmbs = std::max(nmbs, nm);
  }
  b->kmax = kmax; b->imax = imax; b->nhid = nhid; b->nmb_slots = nmbs;
  b->stops_version++;
  CHK(ensure_tables(b, std::max(imax, kmax)));
  const size_t ni = (size_t)imax * G, nk = (size_t)kmax * G;
  b->it_t.assign(ni, t1); b->it_mbf.assign(ni, 0); b->it_mbs.assign(ni, 0); b->it_snap.assign(ni, 0); b->it_n.assign(G, 0);
  b->mbf_res.assign(nk, 0); b->mbs_res.assign(nk, 0);
  for (int g = 0; g < G; ++g) {
    b->it_n[g] = (int)its[g].size();
    for (size_t i = 0; i < its[g].size(); ++i) {
      const It& e = its[g][i];
      const size_t q = i * G + g;
      b->it_t[q] = e.t; b->it_mbf[q] = e.mb; b->it_mbs[q] = e.mbs; b->it_snap[q] = e.snap;
      if (e.res >= 0 && e.mb) { b->mbf_res[(size_t)e.res * G + g] = 1; b->mbs_res[(size_t)e.res * G + g] = e.mbs; }
    }
  }
  HIPCHK(hipMemcpyAsync(b->d_tstops, b->it_t.data(), ni * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_mb_flag, b->it_mbf.data(), ni * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_mb_slot, b->it_mbs.data(), ni * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_snapslot, b->it_snap.data(), ni * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_nst, b->it_n.data(), (size_t)G * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_mbf_res, b->mbf_res.data(), nk * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_mbs_res, b->mbs_res.data(), nk * sizeof(int), hipMemcpyHostToDevice, b->stream));
  return ODINN_OK;
}

// loss weights and reference slots per stop and glacier (safe_slice rule, gradient.jl:38-40,144-149).
// LossH: wH = dtH; LossV: wV = dtV; LossHV: wH = dtH^2, wV = scaling*dtV^2 (Losses.jl:407,424-431
// multiply by Dt once more on top of the inner losses).
int upload_loss_tables(odinn_batch* b) {
  // unchanged stops / reference data / loss selection since the last upload (every iteration of an
  // inversion): the tables on the device are still valid -- saves six small copies and a sync per solve
  if (b->tab_key_ptr == b->d_ws && b->d_ws && b->tab_key_ver == b->refs_version && b->tab_key_stops == b->ts_g)
    return ODINN_OK;
  const int k = b->K();
  const size_t n = (size_t)k * b->G;
  std::vector<double> dts(n, 0.0), ws(n, 0.0);
  std::vector<int> slot(n, 0);
  b->wv_h.assign(n, 0.0); b->vsc_h.assign(n, 1.0); b->vslot_h.assign(n, 0);
  const bool useH = b->loss_kind != ODINN_LOSS_V, useV = b->loss_kind != ODINN_LOSS_H;
  for (int j = 0; j < k; ++j)
    for (int g = 0; g < b->G; ++g) {
      const size_t q = (size_t)j * b->G + g;
      const std::vector<double>& tsg = b->ts_g[g];
      if (j >= (int)tsg.size()) continue;  // the glacier has no stop j: dt = 0 marks the row as idle for the reverse kernels
      dts[q] = j > 0 ? tsg[j] - tsg[j - 1] : 0.0;
      if (useH) {
        const std::vector<double>& tr = b->t_ref[g];
        for (size_t m = 0; m < tr.size(); ++m)
          if (tr[m] == tsg[j]) {
            slot[q] = (int)m;
            const double d = m >= 1 ? tr[m] - tr[m - 1] : 0.0;
            ws[q] = b->loss_kind == ODINN_LOSS_HV ? d * d : d;
            break;
          }
      }
      if (useV) {
        const std::vector<double>& tv = b->t_vref[g];
        for (size_t m = 0; m < tv.size(); ++m)
          if (tv[m] == tsg[j]) {
            b->vslot_h[q] = (int)m;
            const double d = m >= 1 ? tv[m] - tv[m - 1] : 0.0;
            b->wv_h[q] = b->loss_kind == ODINN_LOSS_HV ? b->hv_scaling * d * d : d;
            b->vsc_h[q] = b->v_scale_loss ? b->v_scale[g][m] : 1.0;
            break;
          }
      }
    }
  HIPCHK(hipMemcpyAsync(b->d_dts, dts.data(), n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_ws, ws.data(), n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_refslot, slot.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_wv, b->wv_h.data(), n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_vsc, b->vsc_h.data(), n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_vslot, b->vslot_h.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->tab_key_ptr = b->d_ws; b->tab_key_ver = b->refs_version; b->tab_key_stops = b->ts_g;
  return ODINN_OK;
}

// LossV contribution of stop j: accumulates wV * dl/dH into `out`, the loss partial into d_lossacc
// and the theta weight into d_Gsum / d_Gacc.  Returns the constant loss of the last row/column.
int launch_lossV(odinn_batch* b, int j, const double* Hj, double* out, bool with_grad, double* const_loss) {
  *const_loss = 0.0;
  bool any = false;
  for (int g = 0; g < b->G; ++g) {
    const size_t q = (size_t)j * b->G + g;
    if (b->wv_h[q] != 0.0) {
      any = true;
      const int m = b->vslot_h[q];
      *const_loss += b->wv_h[q] * b->vsc_h[q] * b->v_const(g, m);
    }
  }
  if (!any) return ODINN_OK;
  if (b->v_log_eps > 0.0 && !b->v_abs)
    return fail(ODINN_ERR_ARG, "LogSum needs non-negative fields (Losses.jl:214): use it with component :abs");
  VArgs A{};
  A.H = Hj; A.out = out; A.Vabs = b->d_Vabs; A.Vxr = b->d_Vxr; A.Vyr = b->d_Vyr;
  A.wv = b->d_wv + (size_t)j * b->G; A.scale = b->d_vsc + (size_t)j * b->G; A.refslot = b->d_vslot + (size_t)j * b->G;
  A.ntot = b->ntot; A.component_abs = b->v_abs; A.log_eps = b->v_abs ? b->v_log_eps : 0.0;
  A.Gacc = (with_grad && b->wants_Gacc()) ? b->d_Gacc : nullptr;
  A.finv = 1.0 / b->fV;
  // U / Y law: the network's theta-gradient per node (backprop, or the knot interpolation of the Y law), reduced into d_dth
  if (with_grad) CHK(vel_theta_args(b, A, -1));
  const Pools P = b->pools(true);
  launch_surfV_vjp(b->lm(), 1, b->ntiles, b->stream, P, b->lawdev(), A, 0);
  launch_sum_part(b->G, b->stream, P, 1, b->d_lossacc, 1, 0);
  if (with_grad && b->vel_nn()) CHK(vel_theta_finish(b, -1, true, P));
  else if (with_grad) launch_sum_part(b->G, b->stream, P, 3, b->d_Gsum, 1, 0);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

int do_solve_once(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times,
                  const odinn_solver_opts* o, odinn_solve_stats* stats);
// The forward solve; with the tabulated Y law (lm_kern) a solve in which some node left the table's range is repeated with
// twice the range, after two widenings with the network itself.
int do_solve(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times,
             const odinn_solver_opts* o, odinn_solve_stats* stats) {
  for (int attempt = 0;; ++attempt) {
    ++b->ytab_scope;
    const int rc = do_solve_once(b, n_stops, tstops, n_mb, mb_times, o, stats);
    --b->ytab_scope;
    if (!b->ytab_ok) return rc;
    bool over = false;
    if (ytab_overflowed(b, &over) != ODINN_OK || !over) return rc;  // (a failed solve that left the table is repeated as well)
    if (attempt >= 2) b->ytab_blocked = true;
    else if (b->law_kind == ODINN_LAW_NN_U) { b->utab_hmax *= 2.0; b->utab_smax *= 2.0; }
    else for (double& h : b->ytab_hmax) h *= 2.0;
    b->gd_dirty = true;
    b->solved = false;
  }
}
// The gradient drivers run inside the same scope (their reverse passes read snapshots of a forward solve that stayed inside
// the table, and H interpolated between two of them, so they cannot leave it; should the flag be raised all the same, the
// call is repeated with the network).
template <class F>
int with_law_table(odinn_batch* b, F&& f) {
  for (;;) {
    ++b->ytab_scope;
    const int rc = f();
    --b->ytab_scope;
    if (!b->ytab_ok) return rc;
    bool over = false;
    if (ytab_overflowed(b, &over) != ODINN_OK || !over) return rc;
    b->ytab_blocked = true;
    b->gd_dirty = true;
    b->solved = false;
  }
}
int do_solve_once(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times,
                  const odinn_solver_opts* o, odinn_solve_stats* stats) {
  if (n_stops < 2) return fail(ODINN_ERR_ARG, "need at least 2 tstops");
  for (int j = 1; j < n_stops; ++j)
    if (!(tstops[j] > tstops[j - 1])) return fail(ODINN_ERR_ARG, "tstops must be strictly increasing");
  static const bool prof = std::getenv("ODINN_PROFILE_HOST") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp0 = prof ? now() : 0.0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
  CHK(use_dev(b));
  CHK(refresh_gd(b));
  CHK(refresh_law_field(b));
  odinn_solver_opts opt{1e-8, 1e-6, 0.0, 0.0, 0.0, 1000000, 0, 0, 0.0};  // dense = 0: ice-free tiles take the exact shortcut
  if (o) opt = *o;
  if (opt.maxiters <= 0) opt.maxiters = 1000000;
  if (opt.abstol <= 0) opt.abstol = 1e-6;
  if (opt.reltol <= 0) opt.reltol = 1e-8;
  const bool euler = opt.scheme == ODINN_SCHEME_EULER_CFL;
  if (euler && !(opt.cfl > 0.0)) opt.cfl = 0.25;
  if (euler && opt.cfl > 1.0) return fail(ODINN_ERR_ARG, "cfl must be in (0, 1]");
  const bool adaptive = !(opt.fixed_dt > 0.0) && !euler;
  // stop tables (per glacier)
  CHK(build_stop_tables(b, n_stops, tstops, n_mb, mb_times));
  const int nslots = b->kmax + b->nhid, nmb = b->nmb_slots;
  if (nslots > b->nstops_alloc) {
    dfree(b->d_snaps);
    CHK(dalloc(&b->d_snaps, (size_t)nslots * b->ntot));
    // (rows a glacier does not own are never written by the solve: keep them finite for the kernels that read them under a
    //  zero weight)
    HIPCHK(hipMemsetAsync(b->d_snaps, 0, (size_t)nslots * b->ntot * sizeof(double), b->stream));
    b->nstops_alloc = nslots;
  }
  if (nmb > b->nmb_alloc) {
    dfree(b->d_premb);
    CHK(dalloc(&b->d_premb, (size_t)nmb * b->ntot));
    HIPCHK(hipMemsetAsync(b->d_premb, 0, (size_t)nmb * b->ntot * sizeof(double), b->stream));
    b->nmb_alloc = nmb;
  }
  CHK(upload_loss_tables(b));
  if (prof) { HIPCHK(hipStreamSynchronize(b->stream)); tp1 = now(); }
  // initial state and first snapshot
  const size_t fb = (size_t)b->ntot * sizeof(double);
  HIPCHK(hipMemcpyAsync(b->d_U[0], b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_snaps, b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  const Pools P = b->pools(true);
  long long nrhs_extra = 0;
  const double tspan = tstops[n_stops - 1] - tstops[0];
  if (adaptive && !(opt.dt0 > 0.0)) {
    // Hairer-Wanner initial step (OrdinaryDiffEq ode_determine_initdt), all on device
    CHK(launch_dhdt(b, b->d_U[0], b->d_S2, -1));  // f0 -> S2
    launch_initdt_norms(b->ntiles, b->stream, P, b->d_U[0], b->d_S2, nullptr, opt.abstol, opt.reltol);
    launch_initdt_ctrl(b->G, b->stream, P, 0, tspan, opt.dtmax, b->d_dt0);
    launch_axpy_g(b->ntiles, b->stream, P, b->d_S2, b->d_U[0], b->d_U[1]);
    CHK(launch_dhdt(b, b->d_U[1], b->d_E, -1));  // f1 -> E
    launch_initdt_norms(b->ntiles, b->stream, P, b->d_U[0], b->d_S2, b->d_E, opt.abstol, opt.reltol);
    launch_initdt_ctrl(b->G, b->stream, P, 1, tspan, opt.dtmax, b->d_dt0);
    nrhs_extra = 2;
  }
  launch_begin(b->G, b->stream, P, b->d_tstops, opt.dtmax, adaptive ? opt.dt0 : (euler ? 1.0 : opt.fixed_dt));
  if (prof) { HIPCHK(hipStreamSynchronize(b->stream)); tp2 = now(); }
  int nact = b->G;
  HIPCHK(hipMemcpyAsync(b->d_nactive, &nact, sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipGetLastError());

  const int scheme = euler ? 3 : pick_scheme(b, opt.scheme);
  CtrlArgs C{};
  C.tstops = b->d_tstops; C.nstops = b->d_nst; C.G = b->G; C.mb_flag = b->d_mb_flag; C.mb_slot = b->d_mb_slot;
  C.snap_slot = b->d_snapslot;
  C.dtmax = opt.dtmax; C.adaptive = adaptive ? 1 : 0; C.fixed_dt = opt.fixed_dt; C.n_active = b->d_nactive;
  C.errpart = scheme == 2 ? b->fused_part() : b->d_part;
  C.stride = scheme == 2 ? 1 : 4; C.fused = scheme == 2 ? b->fused_ctrl() : 0;
  if (euler && euler_strip(b)) { C.errpart = b->d_partD; C.stride = 1; C.fused = 5; }  // max-D partials of the strip-layout Euler step
  PostArgs A{};
  A.snaps = b->d_snaps; A.premb = b->d_premb; A.ntot = b->ntot; A.mb0 = b->d_mb0;
  A.Sref = b->any_sref ? b->d_Sref : nullptr;
  if (euler) {  // priming launch: dt = 0, measures max D(u0) and sets the first step
    C.cfl = opt.cfl; C.cfl_prime = 1; C.next_cur = 0;
    launch_set_dt(b->G, b->stream, P, 0.0);
    launch_euler_cfl(b, P, b->lawdev(), b->d_U[0], b->d_U[1]);
    launch_controller(b->G, b->stream, P, C);
    C.cfl_prime = 0;
  }
  // steps between host polls of the active-glacier counter: at least one step per stop is needed, so
  // the first batch is n_stops-1 steps (short solves that land on a stop every step finish with one
  // poll and no wasted launches); afterwards 16, with parity kept even for the ping-pong buffers
  const bool sc = !euler && sc_mode(b, scheme);
  if (sc) CHK(sc_buffers(b));
  const bool snapload = !euler && snap_on_load_mode(b, scheme, sc);
  ScArgs SL{};
  SL.snaps = b->d_snaps; SL.ntot = b->ntot; SL.snap_on_load = 1;
  if (b->any_mb) { SL.premb = b->d_premb; SL.mb0 = b->d_mb0; SL.Sref = b->any_sref ? b->d_Sref : nullptr; }
  long long steps = 0;
  int p = 0;
  int chunk = std::max(2, std::min(256, (b->imax - 1 + 1) & ~1));
  int polls = 0;
  if (!euler) {
    if (!b->d_est) CHK(dalloc(&b->d_est, (size_t)b->G));
    C.est_steps = b->d_est;
    b->h_est.assign(b->G, 0);
  }
  while (nact > 0) {
    for (int s = 0; s < chunk; ++s) {
      if (scheme == 3) {
        launch_euler_cfl(b, P, b->lawdev(), b->d_U[p], b->d_U[1 - p]);
        C.next_cur = 1 - p;
      } else if (sc) {
        // launch n reads state / partials [n & 1 ... (n - 1) & 1], writes the other ones; it decides attempt n - 1
        ScArgs SA{};
        SA.C = C; SA.C.next_cur = -1;
        SA.gin = (steps & 1) ? b->d_gs2 : b->d_gs; SA.gout = (steps & 1) ? b->d_gs : b->d_gs2;
        SA.part_in = (steps & 1) ? b->fused_part() : b->d_part2;
        SA.snaps = b->d_snaps; SA.ntot = b->ntot;
        SA.premb = b->d_premb; SA.mb0 = b->d_mb0; SA.Sref = b->any_sref ? b->d_Sref : nullptr;
        CHK(launch_fused_step(b, opt.abstol, opt.reltol, opt.dense ? 0 : 1, &SA, (steps & 1) ? b->d_part2 : b->fused_part()));
        p = 1 - p;
        ++steps;
        continue;
      } else if (scheme == 2) {
        CHK(launch_fused_step(b, opt.abstol, opt.reltol, opt.dense ? 0 : 1, snapload ? &SL : nullptr));
        C.next_cur = -1;
      } else {
        CHK(launch_step(b, p, opt.abstol, opt.reltol));
        C.next_cur = 1 - p;
      }
      launch_controller(b->G, b->stream, P, C);
      if (!snapload) launch_poststep(b->ntiles, b->stream, P, A, b->d_U[0], b->d_U[1]);
      p = 1 - p;
      ++steps;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&nact, b->d_nactive, sizeof(int), hipMemcpyDeviceToHost, b->stream));
    if (C.est_steps) HIPCHK(hipMemcpyAsync(b->h_est.data(), b->d_est, sizeof(int) * b->G, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (steps >= opt.maxiters && nact > 0) return fail(ODINN_ERR_MAXITERS, "maxiters (%lld) reached with %d glaciers active", (long long)opt.maxiters, nact);
    if (C.est_steps) {
      // next poll when the slowest glacier should be done at its current step size (+2 for a rejection or two)
      int est = 0;
      for (int g = 0; g < b->G; ++g) est = std::max(est, b->h_est[g]);
      chunk = std::max(2, std::min(64, (est + 2 + 1) & ~1));
    } else {
      chunk = polls == 0 ? 4 : (polls == 1 ? 8 : 16);
    }
    if (opt.maxiters - steps < chunk) chunk = (int)std::max<long long>(2, (opt.maxiters - steps + 1) & ~1LL);  // never run far past maxiters
    ++polls;
  }
  if (sc && (steps & 1))  // the last launch wrote its state to d_gs2
    HIPCHK(hipMemcpyAsync(b->d_gs, b->d_gs2, sizeof(GState) * b->G, hipMemcpyDeviceToDevice, b->stream));
  if (sc) HIPCHK(hipStreamSynchronize(b->stream));
  if (prof) tp3 = now();
  std::vector<GState> gs(b->G);
  HIPCHK(hipMemcpy(gs.data(), b->d_gs, sizeof(GState) * b->G, hipMemcpyDeviceToHost));
  if (prof) {
    tp4 = now();
    std::fprintf(stderr, "[odinn do_solve] setup %.0f us, initdt+begin %.0f us, steps(%lld) %.0f us, state readback %.0f us\n",
                 tp1 - tp0, tp2 - tp1, steps, tp3 - tp2, tp4 - tp3);
  }
  if (snapload) {
    // the last stop's snapshot of the glacier(s) decided by the very last controller call: one more launch of the step
    // kernel, in which finished glaciers only store what is pending (no controller behind it)
    CHK(launch_fused_step(b, opt.abstol, opt.reltol, opt.dense ? 0 : 1, &SL));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  for (int g = 0; g < b->G; ++g) {
    if (gs[g].nonfinite) return fail(ODINN_ERR_NONFINITE, "non-finite error estimate in glacier %d", g);
    if (stats) {
      stats[g].naccept = gs[g].naccept;
      stats[g].nreject = gs[g].nreject;
      stats[g].nrhs = (euler ? 1 : 5) * (gs[g].naccept + gs[g].nreject) + nrhs_extra + (euler ? 1 : 0);
      stats[g].t_final = gs[g].t;
      stats[g].dt_last = gs[g].dt;
    }
  }
  b->solved = true;
  return ODINN_OK;
}

// LossDhdt after a forward solve: stop indices of every glacier's (t0, t1), the masked mean thickness change, the
// loss term (added onto d_lossacc[g]) and the coefficient of its cotangent fields (d_dh_coef[g])
int dhdt_forward(odinn_batch* b) {
  if (!b->dhdt_on()) return ODINN_OK;
  b->dh_i0_h.assign(b->G, -1); b->dh_i1_h.assign(b->G, -1);
  std::vector<double> dts(b->G, 1.0);
  for (int g = 0; g < b->G; ++g) {
    if (!(b->dh_t1[g] > b->dh_t0[g])) continue;
    const std::vector<double>& tsg = b->ts_g[g];
    for (int j = 0; j < (int)tsg.size(); ++j) {
      if (tsg[j] == b->dh_t0[g]) b->dh_i0_h[g] = j;
      if (tsg[j] == b->dh_t1[g]) b->dh_i1_h[g] = j;
    }
    if (b->dh_i0_h[g] < 0 || b->dh_i1_h[g] < 0)
      return fail(ODINN_ERR_ARG, "dhdtData times (%g, %g) of glacier %d are not among the tstops", b->dh_t0[g], b->dh_t1[g], g);
    dts[g] = b->dh_t1[g] - b->dh_t0[g];
  }
  if (!b->d_dh_i0) {
    CHK(dalloc(&b->d_dh_i0, (size_t)b->G)); CHK(dalloc(&b->d_dh_i1, (size_t)b->G)); CHK(dalloc(&b->d_dh_coef, (size_t)b->G));
    CHK(dalloc(&b->d_dh_dt, (size_t)b->G)); CHK(dalloc(&b->d_dh_ref, (size_t)b->G)); CHK(dalloc(&b->d_dh_part, (size_t)2 * b->ntiles));
  }
  HIPCHK(hipMemcpyAsync(b->d_dh_i0, b->dh_i0_h.data(), sizeof(int) * b->G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_dh_i1, b->dh_i1_h.data(), sizeof(int) * b->G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_dh_dt, dts.data(), sizeof(double) * b->G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_dh_ref, b->dh_ref.data(), sizeof(double) * b->G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));  // the host vectors above are temporaries
  launch_dhdt_sums(b->ntiles, b->G, b->stream, b->pools(true), b->d_snaps, b->d_dh_i0, b->d_dh_i1, b->ntot, b->d_dh_part,
                   b->d_dh_dt, b->d_dh_ref, b->dhdt_weight, b->d_dh_coef, b->d_lossacc);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// Terms of the loss whose gradient does not involve lambda (LossAvgV, VelocityRegularization) are formed right after the
// forward solve: their dL/dH of stop j lands in d_aggH[agg_slot[j]] and is added to lambda at that stop by the reverse loops,
// their dL/dtheta goes into d_Gsum / d_Gacc, which the reverse loops go on accumulating into.  agg_tables builds the
// per-stop per-glacier weight tables of both terms, assigns the slots and clears the fields.
int agg_tables(odinn_batch* b, bool with_grad) {
  const int k = b->K(), G = b->G;
  b->agg_slot_h.assign(k, -1);
  b->agg_nslots = 0;
  const bool av = b->avgv_on(), vr = b->vreg_on();
  if (!av && !vr) return ODINN_OK;
  // generic over the targets in the reference (TimeAggregatedLosses.jl:115-258, Regularization.jl:192-245): every law with a
  // surface-velocity path here -- A-type (target :A) and the U law (target :D, per-node backprop of dU/dtheta)
  b->wA_h.assign((size_t)k * G, 0.0);
  b->wR_h.assign((size_t)k * G, 0.0);
  std::vector<unsigned char> on(G, 0);
  if (av)
    for (int g = 0; g < G; ++g) {  // LossAvgV: tLoss = collect(t1:step:t2) without its last point, weights dt_i / T
      const double t1 = b->av_t1[g], t2 = b->av_t2[g], st = b->avgv_step;
      if (!(t2 > t1)) continue;
      // length of t1:step:t2 as Julia's float ranges find it: the nearest integer to (t2 - t1) / step, one less when that
      // point lies beyond t2 by more than rounding
      int n = (int)std::llround((t2 - t1) / st);
      if (t1 + n * st > t2 + 4.0 * 2.220446049250313e-16 * std::fmax(std::fabs(t1), std::fabs(t2))) --n;
      if (n < 1) return fail(ODINN_ERR_ARG, "LossAvgV: (t1, t2) = (%g, %g) of glacier %d holds no interval of length step = %g", t1, t2, g, st);
      double T = 0.0;
      for (int i = 0; i < n; ++i) T += (t1 + (i + 1) * st) - (t1 + i * st);
      for (int i = 0; i < n; ++i) {
        const double x = t1 + i * st;
        int jj = -1;
        for (int j = 0; j < b->nres(g); ++j)
          if (std::fabs(b->ts_g[g][j] - x) <= 1e-9) { jj = j; break; }
        if (jj < 0) return fail(ODINN_ERR_ARG, "LossAvgV: time %.10g of glacier %d is not among the tstops", x, g);
        b->wA_h[(size_t)jj * G + g] = ((t1 + (i + 1) * st) - x) / T;
      }
      on[g] = 1;
    }
  if (vr)
    for (int g = 0; g < G; ++g) {  // VelocityRegularization: Delta-t.V of the velocity-data times (gradient.jl:144-163)
      const std::vector<double>& tv = b->t_vref[g];
      for (int j = 0; j < b->nres(g); ++j)
        for (size_t m = 1; m < tv.size(); ++m)
          if (tv[m] == b->ts_g[g][j]) b->wR_h[(size_t)j * G + g] = b->vreg_weight * (tv[m] - tv[m - 1]);
    }
  for (int j = 0; j < k; ++j) {
    bool any = false;
    for (int g = 0; g < G; ++g) any = any || b->wA_h[(size_t)j * G + g] != 0.0 || b->wR_h[(size_t)j * G + g] != 0.0;
    if (any) b->agg_slot_h[j] = b->agg_nslots++;
  }
  if (!b->d_avg) CHK(dalloc(&b->d_avg, (size_t)4 * b->ntot));
  if (!b->d_av_on) HIPCHK(hipMalloc(&b->d_av_on, (size_t)G));
  if ((size_t)k * G > b->wA_cap) {
    dfree(b->d_wA); dfree(b->d_wR);
    CHK(dalloc(&b->d_wA, (size_t)k * G)); CHK(dalloc(&b->d_wR, (size_t)k * G));
    b->wA_cap = (size_t)k * G;
  }
  if ((size_t)k > b->agg_slot_cap) {
    if (b->d_agg_slot) (void)hipFree(b->d_agg_slot);
    b->d_agg_slot = nullptr;
    HIPCHK(hipMalloc(&b->d_agg_slot, sizeof(int) * k));
    b->agg_slot_cap = (size_t)k;
  }
  if (with_grad && (size_t)b->agg_nslots * b->ntot > b->aggH_cap) {
    dfree(b->d_aggH);
    CHK(dalloc(&b->d_aggH, (size_t)b->agg_nslots * b->ntot));
    b->aggH_cap = (size_t)b->agg_nslots * b->ntot;
  }
  HIPCHK(hipMemcpyAsync(b->d_wA, b->wA_h.data(), sizeof(double) * k * G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_wR, b->wR_h.data(), sizeof(double) * k * G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_agg_slot, b->agg_slot_h.data(), sizeof(int) * k, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_av_on, on.data(), (size_t)G, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));  // `on` is a temporary
  if (with_grad && b->agg_nslots > 0)
    HIPCHK(hipMemsetAsync(b->d_aggH, 0, (size_t)b->agg_nslots * b->ntot * sizeof(double), b->stream));
  return ODINN_OK;
}

// LossAvgV (TimeAggregatedLosses.jl:146-258): the time-averaged velocity over the stops of the time grid, its loss (added
// onto d_lossacc[g]) and, with_grad, the pull-back of dt_i / T dl/dV through surface_V at every stop of the grid
int avgv_forward(odinn_batch* b, bool with_grad) {
  if (!b->avgv_on()) return ODINN_OK;
  const int k = b->K(), G = b->G;
  auto stop_on = [&](int j) {
    for (int g = 0; g < G; ++g) if (b->wA_h[(size_t)j * G + g] != 0.0) return true;
    return false;
  };
  double *ax = b->d_avg, *ay = b->d_avg + b->ntot, *vx = b->d_avg + 2 * b->ntot, *vy = b->d_avg + 3 * b->ntot;
  HIPCHK(hipMemsetAsync(ax, 0, (size_t)2 * b->ntot * sizeof(double), b->stream));
  const Pools P = b->pools(true);
  for (int j = 0; j < k; ++j) {
    if (!stop_on(j)) continue;
    launch_surface_V(b->lm(), b->ntiles, b->stream, P, b->lawdev(), b->d_snaps + (size_t)j * b->ntot, vx, vy, 0, 1.0 / b->fV);
    launch_avgv_axpy(b->ntiles, b->stream, P, vx, vy, ax, ay, b->d_wA + (size_t)j * G);
  }
  launch_avgv_cot(b->ntiles, b->stream, P, ax, ay, b->d_aVabs, b->d_aVx, b->d_aVy, b->d_av_on, b->avgv_abs, b->avgv_weight);
  launch_sum_part(G, b->stream, P, 1, b->d_lossacc, 1, 0);
  if (with_grad) {
    for (int j = 0; j < k; ++j) {
      if (!stop_on(j)) continue;
      VArgs A{};
      A.H = b->d_snaps + (size_t)j * b->ntot; A.dVx = ax; A.dVy = ay; A.out = b->d_aggH + (size_t)b->agg_slot_h[j] * b->ntot;
      A.wv = b->d_wA + (size_t)j * G; A.ntot = b->ntot;
      A.Gacc = b->wants_Gacc() ? b->d_Gacc : nullptr;
      A.finv = 1.0 / b->fV;
      CHK(vel_theta_args(b, A, -1));  // U / Y law: the network's theta-gradient per node, reduced into d_dth
      launch_surfV_vjp(b->lm(), 2, b->ntiles, b->stream, P, b->lawdev(), A, 0);
      if (b->vel_nn()) CHK(vel_theta_finish(b, -1, true, P));
      else launch_sum_part(G, b->stream, P, 3, b->d_Gsum, 1, 0);
    }
  }
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// VelocityRegularization (Regularization.jl:192-245) at one state H: loss partial onto d_lossacc (w_loss), the pull-back of
// dReg/dV through surface_V accumulated into `outH` (null: none) and into d_Gsum / d_Gacc (theta: false: not), all scaled
// per glacier by w[g]
static int vreg_at(odinn_batch* b, const double* H, const double* w, bool add_loss, double* outH, bool theta) {
  const Pools P = b->pools(true);
  double *vx = b->d_avg, *vy = b->d_avg + b->ntot, *va = b->d_avg + 2 * b->ntot, *r = b->d_avg + 3 * b->ntot;
  launch_surface_V(b->lm(), b->ntiles, b->stream, P, b->lawdev(), H, vx, vy, 0, 1.0 / b->fV);
  launch_vreg_prep(b->ntiles, b->stream, P, H, vx, vy, w, b->vreg_dist, va, b->d_vrm);
  launch_vreg_lap(b->ntiles, b->stream, P, va, b->d_vrm, w, r);
  if (add_loss) launch_sum_part(b->G, b->stream, P, 1, b->d_lossacc, 1, 0);
  if (outH || theta) {
    launch_vreg_cot(b->ntiles, b->stream, P, r, va, w, vx, vy);
    VArgs A{};
    A.H = H; A.dVx = vx; A.dVy = vy; A.out = outH ? outH : r;  // (r is dead by now: a sink for the unused H-part)
    A.wv = w; A.ntot = b->ntot;
    A.Gacc = (theta && b->wants_Gacc()) ? b->d_Gacc : nullptr;
    A.finv = 1.0 / b->fV;
    if (theta) CHK(vel_theta_args(b, A, -1));  // U / Y law: the network's theta-gradient per node, reduced into d_dth
    // nobody wants dL/dH (the quadrature nodes of the continuous adjoint), closed-form law: the theta-part alone
    if (!outH && theta && b->lm() <= 1 && !b->vel_nn()) launch_surfV_theta_only(b->lm(), b->ntiles, b->stream, P, A);
    else launch_surfV_vjp(b->lm(), 2, b->ntiles, b->stream, P, b->lawdev(), A, 0);
    if (theta && b->vel_nn()) CHK(vel_theta_finish(b, -1, true, P));
    else if (theta) launch_sum_part(b->G, b->stream, P, 3, b->d_Gsum, 1, 0);
  }
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// The VelocityRegularization term over a run: loss and dL/dH at the velocity-data stops with the weights Delta-t.V of the
// discrete loss; dL/dtheta summed over the same stops (DiscreteAdjoint, gradient.jl:252) or -- nq > 0 -- integrated over
// the Gauss-Legendre nodes on the interpolated state with Delta-t = 1 (ContinuousAdjoint, gradient.jl:475-503)
int vreg_forward(odinn_batch* b, bool with_grad, bool add_loss, int nq, const double* qt, const double* qw) {
  if (!b->vreg_on()) return ODINN_OK;
  const int k = b->K(), G = b->G;
  if (!b->d_vrm) HIPCHK(hipMalloc(&b->d_vrm, (size_t)b->ntot));
  for (int j = 0; j < k; ++j) {
    bool any = false;
    for (int g = 0; g < G; ++g) any = any || b->wR_h[(size_t)j * G + g] != 0.0;
    if (!any) continue;
    CHK(vreg_at(b, b->d_snaps + (size_t)j * b->ntot, b->d_wR + (size_t)j * G, add_loss,
                with_grad ? b->d_aggH + (size_t)b->agg_slot_h[j] * b->ntot : nullptr, with_grad && nq == 0));
  }
  if (with_grad && nq > 0) {
    std::vector<double> wq((size_t)nq * G, 0.0);
    for (int g = 0; g < G; ++g)
      if (b->t_vref[g].size() >= 2)
        for (int n = 0; n < nq; ++n) wq[(size_t)n * G + g] = b->vreg_weight * qw[n];
    // segment and weight of every node in every glacier's own snapshots, interpolate((t,), H, Gridded(Linear())) (gradient.jl:287)
    // (glaciers without two velocity dates carry weight 0 but are interpolated like the others: the kernels below run over the
    //  whole batch, and 0 * whatever-the-buffer-held-before must not be 0 * NaN)
    std::vector<int> sg((size_t)nq * G, 0);
    std::vector<double> sw((size_t)nq * G, 0.0);
    for (int g = 0; g < G; ++g) {
      const std::vector<double>& tsg = b->ts_g[g];
      const int kg = (int)tsg.size();
      for (int n = 0; n < nq; ++n) {
        int j = 0;
        while (j + 2 < kg && qt[n] >= tsg[j + 1]) ++j;
        sg[(size_t)n * G + g] = j;
        sw[(size_t)n * G + g] = (qt[n] - tsg[j]) / (tsg[j + 1] - tsg[j]);
      }
    }
    if (wq.size() > b->wRq_cap) {
      dfree(b->d_wRq); dfree(b->d_swq); dfree(b->d_sgq);
      CHK(dalloc(&b->d_wRq, wq.size())); CHK(dalloc(&b->d_swq, wq.size())); CHK(dalloc(&b->d_sgq, wq.size()));
      b->wRq_cap = wq.size();
    }
    HIPCHK(hipMemcpyAsync(b->d_wRq, wq.data(), sizeof(double) * wq.size(), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_swq, sw.data(), sizeof(double) * sw.size(), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_sgq, sg.data(), sizeof(int) * sg.size(), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));  // the staging vectors are temporaries
    for (int n = 0; n < nq; ++n) {
      launch_lerp_g(b->ntiles, b->stream, b->pools(true), b->d_snaps, b->ntot, b->d_sgq + (size_t)n * G, b->d_swq + (size_t)n * G, b->d_tmpA);
      CHK(vreg_at(b, b->d_tmpA, b->d_wRq + (size_t)n * G, false, nullptr, true));
    }
  }
  return ODINN_OK;
}

// A MultiLoss term with a non-zero weight whose data no glacier carries would silently drop out of the loss: refuse instead
// (glaciers WITHOUT the data of a term that others carry simply do not contribute to it -- the per-glacier opt-out)
int check_loss_terms(odinn_batch* b) {
  if (b->dhdt_weight != 0.0 && !b->dhdt_on())
    return fail(ODINN_ERR_STATE, "LossDhdt has weight %g but no glacier carries dhdtData (odinn_set_dhdt_reference)", b->dhdt_weight);
  if (b->avgv_weight != 0.0 && !b->avgv_on())
    return fail(ODINN_ERR_STATE, "LossAvgV has weight %g but no glacier carries a velocity sample (odinn_set_avgv_reference)", b->avgv_weight);
  if (b->vreg_weight != 0.0 && !b->vreg_on())
    return fail(ODINN_ERR_STATE, "VelocityRegularization has weight %g but no glacier carries two velocity-data dates "
                                 "(odinn_set_velocity_reference)", b->vreg_weight);
  return ODINN_OK;
}

// forward loss over the stored snapshots -> d_lossacc[g]; *const_loss: data-only part of LossV
int do_loss(odinn_batch* b, double* const_loss) {
  CHK(check_loss_terms(b));
  const int k = b->K();
  *const_loss = 0.0;
  HIPCHK(hipMemsetAsync(b->d_lossacc, 0, sizeof(double) * b->G, b->stream));
  const Pools P = b->pools(true);
  for (int j = 1; j < k; ++j) {
    if (b->d_Href && b->loss_kind != ODINN_LOSS_V) {
      launch_loss(b->ntiles, b->stream, P, b->d_snaps + (size_t)j * b->ntot, b->d_Href, b->d_mask,
                  b->d_ws + (size_t)j * b->G, b->d_refslot + (size_t)j * b->G, b->ntot, b->h_log_eps);
      launch_sum_part(b->G, b->stream, P, 1, b->d_lossacc, 1, 0);
    }
    if (b->d_Vabs && b->loss_kind != ODINN_LOSS_H) {
      double c = 0.0;
      CHK(launch_lossV(b, j, b->d_snaps + (size_t)j * b->ntot, b->d_tmpB, false, &c));
      *const_loss += c;
    }
  }
  CHK(dhdt_forward(b));  // time-aggregated terms (inversion_utils.jl:457-460)
  CHK(agg_tables(b, false));
  CHK(avgv_forward(b, false));
  CHK(vreg_forward(b, false, true, 0, nullptr, nullptr));
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

}  // namespace

// =========================================================================================
extern "C" {

const char* odinn_last_error(void) { return g_err.c_str(); }

int odinn_device_count(int* n) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  if (n) *n = c;
  return ODINN_OK;
}

int odinn_device_name(int dev, char* buf, int buflen) {
  hipDeviceProp_t pr;
  HIPCHK(hipGetDeviceProperties(&pr, dev));
  snprintf(buf, buflen, "%s (%s)", pr.name, pr.gcnArchName);
  return ODINN_OK;
}

int odinn_batch_create(int device, int n_glaciers, const odinn_glacier_desc* descs, odinn_batch** out) {
  if (!out || !descs || n_glaciers <= 0) return fail(ODINN_ERR_ARG, "bad arguments to odinn_batch_create");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(ODINN_ERR_NO_DEVICE, "no HIP device visible: libodinn_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(ODINN_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
  for (int g = 0; g < n_glaciers; ++g)
    if (descs[g].nx < 3 || descs[g].ny < 3 || !(descs[g].dx > 0) || !(descs[g].dy > 0))
      return fail(ODINN_ERR_ARG, "glacier %d: need nx,ny >= 3 and dx,dy > 0", g);
  for (int g = 0; g < n_glaciers; ++g)  // the strip kernels address a glacier's cells by 32-bit byte offsets
    if ((long long)descs[g].nx * descs[g].ny >= (1LL << 29))
      return fail(ODINN_ERR_ARG, "glacier %d: %d x %d cells exceed the 2^29 cells a single glacier may have", g, descs[g].nx,
                  descs[g].ny);
  odinn_batch* b = new odinn_batch();
  b->device = device;
  b->G = n_glaciers;
  b->descs.assign(descs, descs + n_glaciers);
  b->gd.resize(n_glaciers);
  b->t_ref.resize(n_glaciers);
  b->dh_t0.assign(n_glaciers, 0.0); b->dh_t1.assign(n_glaciers, 0.0); b->dh_ref.assign(n_glaciers, 0.0);
  b->av_t1.assign(n_glaciers, 0.0); b->av_t2.assign(n_glaciers, 0.0);
  b->t_vref.resize(n_glaciers); b->v_scale.resize(n_glaciers); b->v_cxy.resize(n_glaciers); b->v_cabs.resize(n_glaciers);
  b->v_edge.resize(n_glaciers);
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&b->ev0));
  HIPCHK(hipEventCreate(&b->ev1));
  std::vector<int4> nat;
  long long off = 0, offd = 0;
  for (int g = 0; g < n_glaciers; ++g) {
    GDev& r = b->gd[g];
    memset(&r, 0, sizeof r);
    r.nx = descs[g].nx; r.ny = descs[g].ny;
    r.ntx = (r.nx + TX - 1) / TX; r.nty = (r.ny + TY - 1) / TY;
    r.tile0 = (int)nat.size(); r.ntiles = r.ntx * r.nty;
    // 64-double alignment of every glacier so that rows of nx%64==0 grids stay 512-B aligned
    off = (off + 63) & ~63LL; offd = (offd + 63) & ~63LL;
    r.off = off; r.offd = offd;
    off += (long long)r.nx * r.ny;
    offd += (long long)(r.nx - 1) * (r.ny - 1);
    r.mb_max = INFINITY;
    for (int ty = 0; ty < r.nty; ++ty)
      for (int tx = 0; tx < r.ntx; ++tx) nat.push_back(make_int4(g, tx, ty, (int)nat.size()));
  }
  b->ntot = off; b->ntotd = offd; b->ntiles = (int)nat.size();
  // XCD-aware order: block k runs on XCD k%8; give each XCD a contiguous band of tiles
  std::vector<int4> swz(nat.size());
  {
    const int n = (int)nat.size(), X = 8;
    const int per = (n + X - 1) / X;
    int k = 0;
    std::vector<int> order;
    order.reserve(n);
    for (int r = 0; r < per; ++r)
      for (int x = 0; x < X; ++x) {
        const int t = x * per + r;
        if (t < n) order.push_back(t);
      }
    for (int t : order) swz[k++] = nat[t];
  }
  CHK(dalloc(&b->d_tiles, nat.size()));
  CHK(dalloc(&b->d_tiles_nat, nat.size()));
  HIPCHK(hipMemcpy(b->d_tiles, swz.data(), sizeof(int4) * nat.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_tiles_nat, nat.data(), sizeof(int4) * nat.size(), hipMemcpyHostToDevice));
  // tile tables of the fused-step kernel, same XCD-banded order: FOX x FOY "throughput" tiles and
  // FOX x FOYS "latency" tiles (used when the batch has too few throughput tiles to fill the GPU)
  for (int small = 0; small < 6; ++small) {
    const int foy = small == 5 ? FOYT2 : small == 4 ? FOYT4 : small == 3 ? FOYT8 : small == 2 ? FOYT : small ? FOYS : FOY;
    std::vector<int4> natF;
    for (int g = 0; g < n_glaciers; ++g) {
      GDev& r = b->gd[g];
      const int fx = (r.nx + FOX - 1) / FOX, fy = (r.ny + foy - 1) / foy;
      if (small == 5) { r.tile0Fw = (int)natF.size(); r.ntilesFw = fx * fy; }
      else if (small == 4) { r.tile0Fv = (int)natF.size(); r.ntilesFv = fx * fy; }
      else if (small == 3) { r.tile0Fu = (int)natF.size(); r.ntilesFu = fx * fy; }
      else if (small == 2) { r.tile0Ft = (int)natF.size(); r.ntilesFt = fx * fy; }
      else if (small) { r.tile0Fs = (int)natF.size(); r.ntilesFs = fx * fy; }
      else { r.tile0F = (int)natF.size(); r.ntilesF = fx * fy; }
      for (int ty = 0; ty < fy; ++ty)
        for (int tx = 0; tx < fx; ++tx) natF.push_back(make_int4(g, tx, ty, (int)natF.size()));
    }
    const int nF = (int)natF.size(), X = 8, per = (nF + X - 1) / X;
    std::vector<int4> swzF;
    swzF.reserve(nF);
    for (int r = 0; r < per; ++r)
      for (int x = 0; x < X; ++x) {
        const int t = x * per + r;
        if (t < nF) swzF.push_back(natF[t]);
      }
    if (small == 5) {
      b->ntilesFw = nF;
      CHK(dalloc(&b->d_tilesFw, (size_t)nF));
      CHK(dalloc(&b->d_partFw, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesFw, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    } else if (small == 4) {
      b->ntilesFv = nF;
      CHK(dalloc(&b->d_tilesFv, (size_t)nF));
      CHK(dalloc(&b->d_partFv, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesFv, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    } else if (small == 3) {
      b->ntilesFu = nF;
      CHK(dalloc(&b->d_tilesFu, (size_t)nF));
      CHK(dalloc(&b->d_partFu, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesFu, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    } else if (small == 2) {
      b->ntilesFt = nF;
      CHK(dalloc(&b->d_tilesFt, (size_t)nF));
      CHK(dalloc(&b->d_partFt, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesFt, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    } else if (small) {
      b->ntilesFs = nF;
      CHK(dalloc(&b->d_tilesFs, (size_t)nF));
      CHK(dalloc(&b->d_partFs, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesFs, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    } else {
      b->ntilesF = nF;
      CHK(dalloc(&b->d_tilesF, (size_t)nF));
      CHK(dalloc(&b->d_partF, (size_t)nF));
      HIPCHK(hipMemcpy(b->d_tilesF, swzF.data(), sizeof(int4) * nF, hipMemcpyHostToDevice));
    }
  }
  {  // tile table of k_dhdt_strip
    std::vector<int4> natD;
    for (int g = 0; g < n_glaciers; ++g) {
      GDev& r = b->gd[g];
      const int fx = (r.nx + DHDT_OX - 1) / DHDT_OX, fy = (r.ny + DHDT_OY - 1) / DHDT_OY;
      r.tile0D = (int)natD.size(); r.ntilesD = fx * fy;
      for (int ty = 0; ty < fy; ++ty)
        for (int tx = 0; tx < fx; ++tx) natD.push_back(make_int4(g, tx, ty, (int)natD.size()));
    }
    const int nD = (int)natD.size(), X = 8, per = (nD + X - 1) / X;
    std::vector<int4> swzD;
    swzD.reserve(nD);
    for (int r = 0; r < per; ++r)
      for (int x = 0; x < X; ++x) {
        const int t = x * per + r;
        if (t < nD) swzD.push_back(natD[t]);
      }
    b->ntilesD = nD;
    CHK(dalloc(&b->d_tilesD, (size_t)nD));
    CHK(dalloc(&b->d_partD, (size_t)nD));
    HIPCHK(hipMemcpy(b->d_tilesD, swzD.data(), sizeof(int4) * nD, hipMemcpyHostToDevice));
  }
  CHK(dalloc(&b->d_gd, n_glaciers));
  CHK(dalloc(&b->d_gs, n_glaciers));
  const size_t n = (size_t)b->ntot, nd = (size_t)b->ntotd;
  CHK(dalloc(&b->d_B, n)); CHK(dalloc(&b->d_H0, n));
  CHK(dalloc(&b->d_U[0], n)); CHK(dalloc(&b->d_U[1], n));
  CHK(dalloc(&b->d_S2, n)); CHK(dalloc(&b->d_S3, n)); CHK(dalloc(&b->d_E, n));
  CHK(dalloc(&b->d_lam[0], n)); CHK(dalloc(&b->d_lam[1], n));
  CHK(dalloc(&b->d_tmpA, n)); CHK(dalloc(&b->d_tmpB, n));
  CHK(dalloc(&b->d_mb0, n)); CHK(dalloc(&b->d_Sref, n));
  CHK(dalloc(&b->d_Afield, nd)); CHK(dalloc(&b->d_Tfield, nd)); CHK(dalloc(&b->d_Gacc, nd));
  CHK(dalloc(&b->d_part, 4 * nat.size()));
  CHK(dalloc(&b->d_nactive, 1));
  CHK(dalloc(&b->d_dt0, n_glaciers));
  CHK(dalloc(&b->d_lossacc, n_glaciers));
  CHK(dalloc(&b->d_Gsum, n_glaciers));
  CHK(dalloc(&b->d_theta, 1));
  b->mlp.n_layers = 0;
  b->gd_dirty = true;
  *out = b;
  return ODINN_OK;
}

int odinn_batch_destroy(odinn_batch* b) {
  if (!b) return ODINN_OK;
  (void)hipSetDevice(b->device);
  (void)hipStreamSynchronize(b->stream);
  dfree(b->d_tilesD); dfree(b->d_partD);
  dfree(b->d_dh_i0); dfree(b->d_dh_i1); dfree(b->d_dh_coef); dfree(b->d_dh_part); dfree(b->d_dh_dt); dfree(b->d_dh_ref);
  dfree(b->d_tiles); dfree(b->d_tiles_nat); dfree(b->d_tilesF); dfree(b->d_partF); dfree(b->d_gd); dfree(b->d_gs);
  dfree(b->d_ytab); dfree(b->d_ytab_over); dfree(b->d_ytab_stat);
  dfree(b->d_B); dfree(b->d_H0); dfree(b->d_U[0]); dfree(b->d_U[1]); dfree(b->d_S2); dfree(b->d_S3); dfree(b->d_E);
  dfree(b->d_lam[0]); dfree(b->d_lam[1]); dfree(b->d_tmpA); dfree(b->d_tmpB); dfree(b->d_mb0); dfree(b->d_Sref);
  dfree(b->d_Afield); dfree(b->d_Tfield); dfree(b->d_Gacc); dfree(b->d_part); dfree(b->d_nactive); dfree(b->d_dt0); dfree(b->d_tilesFs); dfree(b->d_partFs); dfree(b->d_tilesFt); dfree(b->d_partFt); dfree(b->d_tilesFu); dfree(b->d_partFu); dfree(b->d_tilesFv); dfree(b->d_partFv); dfree(b->d_tilesFw); dfree(b->d_partFw); dfree(b->d_est); dfree(b->d_gs2); dfree(b->d_part2);
  dfree(b->d_rtau); dfree(b->d_rqw); dfree(b->d_tsnap); dfree(b->d_qw); dfree(b->d_rsnap); dfree(b->d_rmbf);
  dfree(b->d_rmbs); dfree(b->d_adj); dfree(b->d_adj2);
  dfree(b->d_partsteps);
  dfree(b->d_rvA); dfree(b->d_rvB); dfree(b->d_zeroslot); dfree(b->d_rvs); dfree(b->d_Vq); dfree(b->d_vscq); dfree(b->d_wvq);
  dfree(b->d_rega); dfree(b->d_regr); dfree(b->d_regg); dfree(b->d_regp); dfree(b->d_regm);
  dfree(b->d_lossacc); dfree(b->d_Gsum); dfree(b->d_theta); dfree(b->d_theta_pad); dfree(b->d_snaps); dfree(b->d_premb); dfree(b->d_Href);
  dfree(b->d_mask); dfree(b->d_part_theta); dfree(b->d_gscratch); dfree(b->d_dth); dfree(b->d_tstops);
  dfree(b->d_snapslot); dfree(b->d_nst); dfree(b->d_mbf_res); dfree(b->d_mbs_res); dfree(b->d_nr); dfree(b->d_ksn); dfree(b->d_lastseg);
  dfree(b->d_zerow); dfree(b->d_swq); dfree(b->d_sgq); dfree(b->d_rhid);
  dfree(b->d_aVabs); dfree(b->d_aVx); dfree(b->d_aVy); dfree(b->d_avg); dfree(b->d_wA); dfree(b->d_aggH);
  if (b->d_agg_slot) (void)hipFree(b->d_agg_slot);
  if (b->d_av_on) (void)hipFree(b->d_av_on);
  dfree(b->d_wR); dfree(b->d_wRq);
  if (b->d_segs) (void)hipFree(b->d_segs);
  dfree(b->d_partTh);
  if (b->d_vrm) (void)hipFree(b->d_vrm);
  dfree(b->d_nodeS); dfree(b->d_ucell); if (b->d_interp_err) (void)hipFree(b->d_interp_err);
  dfree(b->d_nodeH); dfree(b->d_nodeV); dfree(b->d_sortH); dfree(b->d_sortV); dfree(b->d_knots); dfree(b->d_knotG);
  dfree(b->d_knotab); dfree(b->d_knotM);
  if (b->d_sorttmp) { (void)hipFree(b->d_sorttmp); b->d_sorttmp = nullptr; }
  dfree(b->d_ib_gid); dfree(b->d_ib_iota); dfree(b->d_ib_iA); dfree(b->d_ib_iB); dfree(b->d_ib_kA); dfree(b->d_ib_kB);
  dfree(b->d_ib_sH); dfree(b->d_ib_sV); dfree(b->d_ib_knots); dfree(b->d_ib_ab); dfree(b->d_ib_M);
  if (b->d_ib_tmp) { (void)hipFree(b->d_ib_tmp); b->d_ib_tmp = nullptr; }
  for (int l = 0; l < odinn_batch::INTERP_LANES_MAX; ++l) {
    if (b->side[l]) (void)hipStreamDestroy(b->side[l]);
    if (b->ev_join[l]) (void)hipEventDestroy(b->ev_join[l]);
  }
  if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
  for (int l = 0; l < odinn_batch::IA_LANES_MAX; ++l) {
    if (b->ia_stream[l]) { (void)hipStreamSynchronize(b->ia_stream[l]); (void)hipStreamDestroy(b->ia_stream[l]); }
    if (b->ev_emit[l]) (void)hipEventDestroy(b->ev_emit[l]);
    if (b->ev_done[l]) (void)hipEventDestroy(b->ev_done[l]);
    if (b->ia_lane[l].sel) (void)hipFree(b->ia_lane[l].sel);
    if (l > 0) {  // (lane 0 aliases the batch's own arrays)
      odinn_batch::IaLane& a = b->ia_lane[l];
      dfree(a.sH); dfree(a.sV); dfree(a.knots); dfree(a.ab); dfree(a.iA); dfree(a.iB); dfree(a.kA); dfree(a.kB);
      dfree(a.M);
      if (a.tmp) (void)hipFree(a.tmp);
    }
  }
  dfree(b->ia_flags); dfree(b->ia_act); dfree(b->ia_gid_act); dfree(b->ia_nact_dev); dfree(b->ia_aoff);
  if (b->ia_sel_tmp) (void)hipFree(b->ia_sel_tmp);
  for (int q_ = 0; q_ < odinn_batch::IA_SETS_MAX; ++q_) {
    if (q_ > 0) { dfree(b->ia_nodeH[q_]); dfree(b->ia_nodeV[q_]); }  // (set 0 aliases d_nodeH / d_nodeV)
    dfree(b->ia_setmax[q_]);
    if (b->ev_set_done[q_]) (void)hipEventDestroy(b->ev_set_done[q_]);
  }
  dfree(b->d_dthq);
  dfree(b->d_mb_flag); dfree(b->d_mb_slot); dfree(b->d_dts); dfree(b->d_ws); dfree(b->d_refslot);
  dfree(b->d_Vabs); dfree(b->d_Vxr); dfree(b->d_Vyr); dfree(b->d_wv); dfree(b->d_vsc); dfree(b->d_vslot);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return ODINN_OK;
}

int odinn_batch_sync(odinn_batch* b) {
  CHK(use_dev(b));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ODINN_OK;
}

int64_t odinn_batch_cells(odinn_batch* b) {
  if (!b) return 0;
  int64_t n = 0;
  for (int g = 0; g < b->G; ++g) n += (int64_t)b->gd[g].nx * b->gd[g].ny;
  return n;
}

int odinn_set_fields(odinn_batch* b, int g, const double* H0, const double* B) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H0 || !B) return fail(ODINN_ERR_ARG, "null field");
  CHK(up_field(b, g, b->d_H0, H0));
  CHK(up_field(b, g, b->d_B, B));
  b->solved = false;
  {  // range of the Y law's table (refresh_gd): follows the thickest ice of the new state
    b->h0max.resize(b->G, 0.0);
    b->ytab_hmax.resize(b->G, 0.0);
    double m = 0.0;
    const size_t n = (size_t)b->gd[g].nx * b->gd[g].ny;
    for (size_t i = 0; i < n; ++i) if (H0[i] > m) m = H0[i];
    b->h0max[g] = m;
    b->ytab_hmax[g] = 0.0;
    b->utab_hmax = 0.0;
    b->ytab_blocked = false;
    if (b->law_kind == ODINN_LAW_NN_Y || b->law_kind == ODINN_LAW_NN_U) b->gd_dirty = true;
  }
  return ODINN_OK;
}

int odinn_set_A(odinn_batch* b, int g, double A) {
  CHK(check_g(b, g));
  b->descs[g].A = A;
  b->gd_dirty = true;
  if (b->has_Afield_const) {
    // once any glacier of the batch carries a gridded A every kernel reads the field: a scalar A replaces this
    // glacier's slice of it (it used to be ignored silently)
    CHK(use_dev(b));
    const GDev& r = b->gd[g];
    std::vector<double> a((size_t)(r.nx - 1) * (r.ny - 1), A);
    CHK(up_field(b, g, b->d_Afield, a.data(), true));
  }
  return ODINN_OK;
}

int odinn_set_A_field(odinn_batch* b, int g, const double* A_dual) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!A_dual) return fail(ODINN_ERR_ARG, "null field");
  if (b->law_kind != ODINN_LAW_CONST_A) return fail(ODINN_ERR_STATE, "odinn_set_A_field requires ODINN_LAW_CONST_A");
  if (!b->has_Afield_const) {
    // every glacier reads the field once any glacier sets one: initialise all to their scalar A
    for (int q = 0; q < b->G; ++q) {
      const GDev& r = b->gd[q];
      std::vector<double> a((size_t)(r.nx - 1) * (r.ny - 1), b->descs[q].A);
      CHK(up_field(b, q, b->d_Afield, a.data(), true));
    }
    b->has_Afield_const = true;
  }
  CHK(up_field(b, g, b->d_Afield, A_dual, true));
  b->gd_dirty = true;
  return ODINN_OK;
}

int odinn_set_T_field(odinn_batch* b, int g, const double* T_dual) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!T_dual) return fail(ODINN_ERR_ARG, "null field");
  return up_field(b, g, b->d_Tfield, T_dual, true);
}

int odinn_set_theta(odinn_batch* b, const double* theta, int P) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  CHK(use_dev(b));
  if (b->law_kind == ODINN_LAW_CONST_A) return fail(ODINN_ERR_STATE, "no trainable law set");
  if (P != b->P || !theta) return fail(ODINN_ERR_ARG, "theta has %d entries, law expects %d", P, b->P);
  b->theta.assign(theta, theta + P);
  HIPCHK(hipMemcpyAsync(b->d_theta, theta, sizeof(double) * P, hipMemcpyHostToDevice, b->stream));
  {  // the padded rows mlp_eval_rt reads: unit o of layer l -> row (units of the layers before) + o; theta keeps Lux's
     // layout (weight out x in column-major, then bias, layer after layer)
    int rows = 0, mw = 1;
    for (int l = 0; l < b->mlp.n_layers; ++l) { rows += b->mlp.widths[l + 1]; mw = std::max(mw, b->mlp.widths[l]); }
    const int W = mw <= 16 ? 16 : 32;
    if (rows != b->pad_rows || W != b->pad_w || !b->d_theta_pad) {
      dfree(b->d_theta_pad);
      b->d_theta_pad = nullptr;
      CHK(dalloc(&b->d_theta_pad, (size_t)rows * (W + 1) + W + 1));  // (+ one spare row and bias: the evaluator reads one unit ahead)
      b->pad_rows = rows; b->pad_w = W;
    }
    std::vector<double> pad((size_t)rows * (W + 1) + W + 1, 0.0);
    int off = 0, r0 = 0;
    for (int l = 0; l < b->mlp.n_layers; ++l) {
      const int nin = b->mlp.widths[l], nout = b->mlp.widths[l + 1];
      for (int o = 0; o < nout; ++o) {
        for (int i = 0; i < nin; ++i) pad[(size_t)(r0 + o) * W + i] = theta[off + o + nout * i];
        pad[(size_t)rows * W + r0 + o] = theta[off + nin * nout + o];
      }
      off += nout * (nin + 1);
      r0 += nout;
    }
    HIPCHK(hipMemcpyAsync(b->d_theta_pad, pad.data(), sizeof(double) * pad.size(), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));  // (pad is a local)
  }
  HIPCHK(hipStreamSynchronize(b->stream));
  b->gd_dirty = true;
  return ODINN_OK;
}

int odinn_set_law(odinn_batch* b, int kind, const odinn_mlp_desc* mlp, const double* theta, int P, double n_H,
                  double n_gradS) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  CHK(use_dev(b));
  if (kind < ODINN_LAW_CONST_A || kind > ODINN_LAW_NN_U) return fail(ODINN_ERR_ARG, "unknown law kind %d", kind);
  if (kind == ODINN_LAW_CONST_A) {
    b->law_kind = kind;
    b->grad_interp = ODINN_GRAD_INTERP_NONE;
    b->P = 0;
    b->mlp.n_layers = 0;
    b->gd_dirty = true;
    return ODINN_OK;
  }
  if (!mlp || !theta) return fail(ODINN_ERR_ARG, "NN law needs an MLP descriptor and theta");
  if (mlp->n_layers < 1 || mlp->n_layers > ODINN_MAX_LAYERS) return fail(ODINN_ERR_ARG, "n_layers out of range");
  for (int l = 0; l <= mlp->n_layers; ++l)
    if (mlp->widths[l] < 1 || mlp->widths[l] > ODINN_MAX_WIDTH) return fail(ODINN_ERR_ARG, "layer width out of range");
  if (mlp->widths[mlp->n_layers] != 1) return fail(ODINN_ERR_ARG, "the MLP must have one output");
  const int nin_expected = (kind == ODINN_LAW_NN_Y || kind == ODINN_LAW_NN_U) ? 2 : 1;
  if (mlp->widths[0] != nin_expected) return fail(ODINN_ERR_ARG, "law kind %d expects %d MLP inputs", kind, nin_expected);
  if (mlp_nparams(*mlp) != P) return fail(ODINN_ERR_ARG, "theta has %d entries, architecture needs %d", P, mlp_nparams(*mlp));
  if (P > MAXP) return fail(ODINN_ERR_ARG, "P=%d exceeds %d", P, MAXP);
  b->law_kind = kind;
  b->mlp = *mlp;
  b->P = P;
  b->ytab_blocked = false;  // (a new law: its table gets its chance)
  b->utab_hmax = 0.0;
  b->nH = n_H; b->nS = n_gradS;
  // the reference's defaults: SIA2D_D_hybrid_target(interpolation = :Linear, n_interp_half = 75) (target_D_hybrid.jl:12-15),
  // SIA2D_D_target(interpolation = :None) (target_D_pure.jl:34-39); A-type laws have no spatial law gradient
  b->grad_interp = kind == ODINN_LAW_NN_Y ? ODINN_GRAD_INTERP_LINEAR : ODINN_GRAD_INTERP_NONE;
  b->n_interp_half = 75;
  dfree(b->d_theta);
  CHK(dalloc(&b->d_theta, (size_t)P));
  return odinn_set_theta(b, theta, P);
}

int odinn_set_reference(odinn_batch* b, int g, int n_ref, const double* t_ref, const double* H_ref, int distance) {
  if (b) b->refs_version++;
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (n_ref < 0 || (n_ref > 0 && (!t_ref || !H_ref))) return fail(ODINN_ERR_ARG, "bad reference data");
  if (n_ref > b->nref_alloc) {
    // grow, keeping what other glaciers already uploaded
    double* nh = nullptr; unsigned char* nm = nullptr;
    auto grow = [&]() -> int {
      CHK(dalloc(&nh, (size_t)n_ref * b->ntot));
      CHK(dalloc(&nm, (size_t)n_ref * b->ntot));
      if (b->d_Href) {
        HIPCHK(hipMemcpy(nh, b->d_Href, (size_t)b->nref_alloc * b->ntot * sizeof(double), hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nm, b->d_mask, (size_t)b->nref_alloc * b->ntot, hipMemcpyDeviceToDevice));
        HIPCHK(hipStreamSynchronize(nullptr));  // D2D copies on the NULL stream do not block the host
      }
      return ODINN_OK;
    };
    if (const int rc = grow()) { dfree(nh); dfree(nm); return rc; }  // nothing leaks when an allocation or copy fails
    dfree(b->d_Href); dfree(b->d_mask);
    b->d_Href = nh; b->d_mask = nm; b->nref_alloc = n_ref;
  }
  const GDev& r = b->gd[g];
  const long long n = (long long)r.nx * r.ny;
  b->t_ref[g].assign(t_ref, t_ref + n_ref);
  std::vector<unsigned char> mask((size_t)n);
  for (int m = 0; m < n_ref; ++m) {
    const double* Hr = H_ref + (size_t)m * n;
    // is_in_glacier(Href, distance): Href>0 on the whole (2d+1)^2 Chebyshev neighbourhood
    for (int j = 0; j < r.ny; ++j)
      for (int i = 0; i < r.nx; ++i) {
        bool in = true;
        for (int dj = -distance; dj <= distance && in; ++dj)
          for (int di = -distance; di <= distance; ++di) {
            const int ii = i + di, jj = j + dj;
            if (ii < 0 || ii >= r.nx || jj < 0 || jj >= r.ny || !(Hr[ii + (size_t)r.nx * jj] > 0.0)) { in = false; break; }
          }
        mask[i + (size_t)r.nx * j] = in ? 1 : 0;
      }
    HIPCHK(hipMemcpy(b->d_Href + (size_t)m * b->ntot + r.off, Hr, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_mask + (size_t)m * b->ntot + r.off, mask.data(), (size_t)n, hipMemcpyHostToDevice));
  }
  return ODINN_OK;
}

int odinn_set_mass_balance(odinn_batch* b, int g, const double* mb0, double dmb_dS, const double* S_ref, double mb_max) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  GDev& r = b->gd[g];
  if (!mb0) {
    r.has_mb = 0;
  } else {
    if (dmb_dS != 0.0 && !S_ref) return fail(ODINN_ERR_ARG, "S_ref required when dmb_dS != 0");
    CHK(up_field(b, g, b->d_mb0, mb0));
    if (S_ref) { CHK(up_field(b, g, b->d_Sref, S_ref)); b->any_sref = true; }
    r.has_mb = 1; r.dmb_dS = dmb_dS; r.mb_max = mb_max;
  }
  b->any_mb = false;
  for (int q = 0; q < b->G; ++q) b->any_mb = b->any_mb || b->gd[q].has_mb;
  b->gd_dirty = true;
  return ODINN_OK;
}

// ---- seams ------------------------------------------------------------------------------
int odinn_sia2d_dhdt(odinn_batch* b, int g, const double* H, double t, double* dH) {
  (void)t;  // SIA2D is autonomous
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !dH) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  CHK(launch_dhdt(b, b->d_tmpA, b->d_tmpB, g));
  return down_field(b, g, b->d_tmpB, dH);
}

int odinn_sia2d_vjp_H(odinn_batch* b, int g, const double* lam, const double* H, double t, double* dlam) {
  (void)t;
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !lam || !dlam) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  CHK(up_field(b, g, b->d_lam[0], lam));
  AdjArgs A{};
  A.H = b->d_tmpA; A.lam = b->d_lam[0]; A.out = b->d_tmpB;
  launch_vjp_H(b, 0, b->gd[g].ntiles, b->pools(false), b->lawdev(), A, b->gd[g].tile0);
  HIPCHK(hipGetLastError());
  return down_field(b, g, b->d_tmpB, dlam);
}

static int ensure_interp_scratch(odinn_batch* b) {
  if (!b->d_nodeH) {
    long long ndmax = 1;
    for (const GDev& r : b->gd) ndmax = std::max(ndmax, (long long)(r.nx - 1) * (r.ny - 1));
    int lanes = std::min(b->G, (int)odinn_batch::INTERP_LANES_MAX);
    if (const int e = sched_val(b->sched.interp_streams, "ODINN_INTERP_STREAMS"); e >= 1) lanes = std::max(1, std::min(lanes, e));
    b->interp_lanes = lanes;
    b->interp_ndmax = ndmax;
    CHK(dalloc(&b->d_nodeH, (size_t)b->ntotd)); CHK(dalloc(&b->d_nodeV, (size_t)b->ntotd));
    CHK(dalloc(&b->d_sortH, (size_t)ndmax * lanes)); CHK(dalloc(&b->d_sortV, (size_t)ndmax * lanes));
    CHK(dalloc(&b->d_knots, (size_t)INTERP_KMAX * lanes)); CHK(dalloc(&b->d_knotab, (size_t)2 * INTERP_KMAX * lanes));
    CHK(dalloc(&b->d_knotM, (size_t)lanes));
    b->sorttmp_bytes = (interp_sort_temp_bytes(ndmax) + 255) & ~(size_t)255;
    HIPCHK(hipMalloc(&b->d_sorttmp, std::max<size_t>(b->sorttmp_bytes, 256) * lanes));
    if (lanes > 1) {
      HIPCHK(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
      for (int l = 0; l < lanes; ++l) {
        HIPCHK(hipStreamCreateWithFlags(&b->side[l], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->ev_join[l], hipEventDisableTiming));
      }
    }
  }
  if (b->law_kind == ODINN_LAW_NN_U && !b->d_nodeS) {
    CHK(dalloc(&b->d_nodeS, (size_t)b->ntotd));
    CHK(dalloc(&b->d_ucell, (size_t)4 * (INTERP_KMAX - 1) * (INTERP_KMAX - 1) * b->interp_lanes));
    HIPCHK(hipMalloc(&b->d_interp_err, sizeof(int)));
    HIPCHK(hipMemsetAsync(b->d_interp_err, 0, sizeof(int), b->stream));
  }
  if (b->law_kind == ODINN_LAW_NN_Y && !b->d_ib_gid && b->ntotd < (1ll << 32)) {
    const size_t N = (size_t)b->ntotd;
    CHK(dalloc(&b->d_ib_gid, N)); CHK(dalloc(&b->d_ib_iota, N)); CHK(dalloc(&b->d_ib_iA, N)); CHK(dalloc(&b->d_ib_iB, N));
    CHK(dalloc(&b->d_ib_kA, N)); CHK(dalloc(&b->d_ib_kB, N)); CHK(dalloc(&b->d_ib_sH, N)); CHK(dalloc(&b->d_ib_sV, N));
    CHK(dalloc(&b->d_ib_knots, (size_t)b->G * INTERP_KMAX)); CHK(dalloc(&b->d_ib_ab, (size_t)b->G * 2 * INTERP_KMAX));
    CHK(dalloc(&b->d_ib_M, (size_t)b->G));
    b->ib_tmp_bytes = interp_batch_temp_bytes(b->ntotd);
    HIPCHK(hipMalloc(&b->d_ib_tmp, std::max<size_t>(b->ib_tmp_bytes, 256)));
    launch_fill_gid(b->stream, b->pools(true), b->G, b->ntotd, b->d_ib_gid, b->d_ib_iota);
  }
  const size_t need = (size_t)std::max(b->P, 1) * INTERP_KMAX * b->interp_lanes;
  if (need > b->knotG_cap) {
    dfree(b->d_knotG);
    CHK(dalloc(&b->d_knotG, need));
    b->knotG_cap = need;
  }
  return ODINN_OK;
}

// U law, `:Linear`: Interpolations.Gridded(Linear()) does not extrapolate -- the reference throws a BoundsError when a
// dual node has Hbar > 100 (or |grad S| > 100); reported once the stream has been synchronised
static int check_interp_bounds(odinn_batch* b) {
  if (!b->d_interp_err) return ODINN_OK;
  int e = 0;
  HIPCHK(hipMemcpy(&e, b->d_interp_err, sizeof(int), hipMemcpyDeviceToHost));
  if (!e) return ODINN_OK;
  HIPCHK(hipMemset(b->d_interp_err, 0, sizeof(int)));
  return fail(ODINN_ERR_ARG, "BoundsError: a dual node lies outside [0, 100] x [0, 100], the domain of the U law's gradient "
                             "interpolant (Laws.jl:128-131, interpolation = :Linear)");
}

// `:Linear` interpolation of d law / d theta (k_interp.hip): zero the (Hbar, weight[, slope]) node arrays of glacier g (-1: all)
// before a kernel emits into them ...
static int interp_prepare(odinn_batch* b, int g, bool linU) {
  CHK(ensure_interp_scratch(b));
  const int g0 = g < 0 ? 0 : g, g1 = g < 0 ? b->G : g + 1;
  const long long lo = b->gd[g0].offd;
  const long long hi = g1 < b->G ? b->gd[g1].offd : b->ntotd;
  HIPCHK(hipMemsetAsync(b->d_nodeH + lo, 0, (size_t)(hi - lo) * sizeof(double), b->stream));
  HIPCHK(hipMemsetAsync(b->d_nodeV + lo, 0, (size_t)(hi - lo) * sizeof(double), b->stream));
  if (linU) HIPCHK(hipMemsetAsync(b->d_nodeS + lo, 0, (size_t)(hi - lo) * sizeof(double), b->stream));
  return ODINN_OK;
}
static int interp_contract(odinn_batch* b, int g, bool linU, bool accumulate, const Pools& P);

// theta-part of a surface-velocity pull-back with a per-node network (U and Y laws): where the node contributions go ...
static int vel_theta_args(odinn_batch* b, VArgs& A, int g) {
  if (!b->vel_nn()) return ODINN_OK;
  CHK(ensure_theta_scratch(b, std::max(g < 0 ? b->ntiles : b->gd[g].ntiles, (int)node_backprop_part_count(g < 0 ? b->G : 1, 1)), false));
  // the kernels EMIT (Hbar, weight[, |grad S|]) per dual node; vel_theta_finish turns them into dtheta: through the knot / node-grid
  // interpolation (`:Linear`) or by exact backprop at every node (`:None`, launch_node_backprop)
  const bool U = b->law_kind == ODINN_LAW_NN_U;
  CHK(interp_prepare(b, g, U));
  A.emitH = b->d_nodeH; A.emitV = b->d_nodeV; A.emitS = U ? b->d_nodeS : nullptr;
  return ODINN_OK;
}
// ... and their reduction into d_dth after the launch (added onto what is there unless !accumulate)
static int vel_theta_finish(odinn_batch* b, int g, bool accumulate, const Pools& P) {
  if (!b->vel_nn()) return ODINN_OK;
  if (b->vel_emit()) return interp_contract(b, g, b->vel_emit_U(), accumulate, P);
  const int g0 = g < 0 ? 0 : g, ng = g < 0 ? b->G : 1;
  if (launch_node_backprop(b->stream, P, b->lawdev(), g0, ng, b->ntotd, b->d_nodeH, b->d_nodeS, b->d_nodeV, b->d_part_theta, b->d_dth,
                           accumulate ? 1 : 0))
    return fail(ODINN_ERR_UNSUPPORTED, "the network has too many parameters for the per-node backprop of the velocity pull-back");
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// ---- the Y law's `:Linear` contraction off the critical path of the adjoints --------------------------------------------------
// The theta-integrand of a stop does not feed back into the reverse solve, so its contraction (a radix sort of all dual nodes,
// the knots, the interval sums, the backprop at the knots: ~35 dependent launches that leave most of the GPU idle) runs on a
// lane stream while the batch's stream takes the next reverse steps, and the contractions of consecutive stops run side by
// side on up to IA_LANES_MAX lanes (own node arrays and sort scratch each).  Contribution q is WRITTEN to slot q of d_dthq and
// the slots are added onto d_dth in the order of the stops when the lanes are joined -- the additions, and their order, of the
// sequence on one stream: bit-identical gradients.  odinn_schedule.interp_async / ODINN_INTERP_ASYNC: 0 = off, n = n lanes.
static int interp_async_lanes(odinn_batch* b, bool useV, int lanes_default) {
  if (useV || b->law_kind != ODINN_LAW_NN_Y || b->grad_interp != ODINN_GRAD_INTERP_LINEAR) return 0;
  const int e = sched_val(b->sched.interp_async, "ODINN_INTERP_ASYNC");
  if (e == 0) return 0;
  if (sched_val(b->sched.interp_batch, "ODINN_INTERP_BATCH") == 0) return 0;
  if (!b->d_ib_gid || interp_batch_lds_bytes(b->P) > 30 * 1024) return 0;
  int lanes = e < 0 ? lanes_default : std::min(e, (int)odinn_batch::IA_LANES_MAX);
  // a lane's arrays: ~72 B per dual node + the sort's scratch (4.9 GB at 64 x 1024^2); the lanes not yet allocated must fit into a
  // quarter of what is free of the 288 GB
  const double per_lane = 72.0 * (double)b->ntotd + (double)b->ib_tmp_bytes;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  int have = 1;
  for (int l = 1; l < (int)odinn_batch::IA_LANES_MAX; ++l) have += b->ia_lane[l].sH ? 1 : 0;
  while (lanes > have && per_lane * (lanes - have) > 0.25 * (double)free_b) --lanes;
  return lanes;
}
static int interp_async_setup(odinn_batch* b, int lanes) {
  const size_t N = (size_t)b->ntotd;
  for (int l = 0; l < lanes; ++l) {
    if (!b->ia_stream[l]) {
      // (measured: low-priority lane streams starve -- 8 x 512^2 continuous gradient 150 -> 470 ms, the batch's stream ends up
      //  waiting for their node arrays)
      HIPCHK(hipStreamCreateWithFlags(&b->ia_stream[l], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&b->ev_emit[l], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&b->ev_done[l], hipEventDisableTiming));
    }
    odinn_batch::IaLane& a = b->ia_lane[l];
    if (l == 0) {
      a.sH = b->d_ib_sH; a.sV = b->d_ib_sV; a.knots = b->d_ib_knots; a.ab = b->d_ib_ab;
      a.iA = b->d_ib_iA; a.iB = b->d_ib_iB; a.kA = b->d_ib_kA; a.kB = b->d_ib_kB; a.tmp = b->d_ib_tmp; a.M = b->d_ib_M;
    } else if (!a.sH) {
      CHK(dalloc(&a.sH, N)); CHK(dalloc(&a.sV, N));
      CHK(dalloc(&a.iA, N)); CHK(dalloc(&a.iB, N)); CHK(dalloc(&a.kA, N)); CHK(dalloc(&a.kB, N));
      CHK(dalloc(&a.knots, (size_t)b->G * INTERP_KMAX)); CHK(dalloc(&a.ab, (size_t)b->G * 2 * INTERP_KMAX)); CHK(dalloc(&a.M, (size_t)b->G));
      HIPCHK(hipMalloc(&a.tmp, std::max<size_t>(b->ib_tmp_bytes, 256)));
    }
    if (!a.sel) HIPCHK(hipMalloc(&a.sel, interp_select_scratch_bytes(b->G, nullptr)));
    b->ia_pending[l] = false;
  }
  int sets = lanes + 1;
  if (const char* e = std::getenv("ODINN_INTERP_SETS")) sets = std::max(lanes, std::min(std::atoi(e), (int)odinn_batch::IA_SETS_MAX));  // (A/B aid)
  for (int q = 0; q < sets; ++q) {
    if (q == 0) { b->ia_nodeH[0] = b->d_nodeH; b->ia_nodeV[0] = b->d_nodeV; }
    else if (!b->ia_nodeH[q]) { CHK(dalloc(&b->ia_nodeH[q], N)); CHK(dalloc(&b->ia_nodeV[q], N)); }
    if (!b->ia_setmax[q]) CHK(dalloc(&b->ia_setmax[q], (size_t)2 * b->G));
    if (!b->ev_set_done[q]) HIPCHK(hipEventCreateWithFlags(&b->ev_set_done[q], hipEventDisableTiming));
    b->ia_set_pending[q] = false;
  }
  b->ia_sets = sets;
  const size_t need = (size_t)odinn_batch::IA_SLOTS * b->G * b->P;
  if (need > b->dthq_cap) {
    dfree(b->d_dthq);
    CHK(dalloc(&b->d_dthq, need));
    b->dthq_cap = need;
  }
  b->ia_lanes = lanes;
  b->ia_q = 0;
  // the active nodes of this gradient's snapshots (ODINN_INTERP_ACTIVE=0: the dense sequence over all dual nodes)
  b->ia_nact = 0;
  static const bool dense = std::getenv("ODINN_INTERP_ACTIVE") && std::getenv("ODINN_INTERP_ACTIVE")[0] == '0';
  int gbits = 0;
  while ((1ll << gbits) < (long long)b->G) ++gbits;
  // (the composite (glacier, Hbar) sort key has room for 64 glaciers; the sort-free contraction has no such limit)
  b->ia_select = sched_val(-1, "ODINN_INTERP_SELECT") != 0;
  if (!dense && (gbits <= 6 || b->ia_select) && b->solved) {
    if (!b->ia_flags) {
      CHK(dalloc(&b->ia_flags, N)); CHK(dalloc(&b->ia_act, N)); CHK(dalloc(&b->ia_gid_act, N)); CHK(dalloc(&b->ia_nact_dev, (size_t)1));
      CHK(dalloc(&b->ia_aoff, (size_t)b->G + 1));
      b->ia_sel_bytes = interp_active_temp_bytes(b->ntotd);
      HIPCHK(hipMalloc(&b->ia_sel_tmp, std::max<size_t>(b->ia_sel_bytes, 256)));
    }
    if (launch_interp_active(b->stream, b->pools(false), b->G, b->ntotd, b->d_ib_gid, b->d_ib_iota, b->d_snaps, b->kmax + b->nhid, b->ntot,
                             b->ia_flags, b->ia_sel_tmp, b->ia_sel_bytes, b->ia_act, b->ia_gid_act, b->ia_aoff, b->ia_nact_dev))
      return fail(ODINN_ERR_HIP, "selection of the active dual nodes failed");
    HIPCHK(hipGetLastError());
    unsigned na = 0;
    HIPCHK(hipMemcpyAsync(&na, b->ia_nact_dev, sizeof(unsigned), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->ia_nact = (long long)na;
  }
  return ODINN_OK;
}
// the batch's stream waits for every contraction issued so far and adds their slots onto d_dth, in order
static int interp_async_join(odinn_batch* b) {
  for (int l = 0; l < b->ia_lanes; ++l)
    if (b->ia_pending[l]) {
      HIPCHK(hipStreamWaitEvent(b->stream, b->ev_done[l], 0));
      b->ia_pending[l] = false;
    }
  for (int q = 0; q < b->ia_sets; ++q) b->ia_set_pending[q] = false;  // (covered by the lanes' last events)
  if (b->ia_q > 0) {
    launch_sum_slots(b->stream, (long long)b->G * b->P, b->ia_q, b->d_dthq, b->d_dth);
    HIPCHK(hipGetLastError());
    b->ia_q = 0;
  }
  return ODINN_OK;
}
// node arrays the next emitting launch may write (zeroed, on the batch's stream, once their previous contraction is through)
static int interp_async_begin(odinn_batch* b, double** nH, double** nV) {
  if (b->ia_q >= odinn_batch::IA_SLOTS) CHK(interp_async_join(b));
  const int q = b->ia_q % b->ia_sets;
  *nH = b->ia_nodeH[q];
  *nV = b->ia_nodeV[q];
  if (b->ia_set_pending[q]) HIPCHK(hipStreamWaitEvent(b->stream, b->ev_set_done[q], 0));
  HIPCHK(hipMemsetAsync(*nH, 0, (size_t)b->ntotd * sizeof(double), b->stream));
  HIPCHK(hipMemsetAsync(*nV, 0, (size_t)b->ntotd * sizeof(double), b->stream));
  return ODINN_OK;
}
// ... the same for the fused reverse step's in-kernel emission: the kernel writes EVERY dual node of the glaciers that emit (zeros on
// their ice-free tiles) and raises their maxima, so only the 16 G bytes of maxima are cleared -- a glacier whose maxima stay zero did
// not emit, and the contraction skips it whatever its (stale) node entries hold
static int interp_async_begin_fused(odinn_batch* b, double** nH, double** nV, unsigned long long** mx) {
  if (b->ia_q >= odinn_batch::IA_SLOTS) CHK(interp_async_join(b));
  const int q = b->ia_q % b->ia_sets;
  *nH = b->ia_nodeH[q];
  *nV = b->ia_nodeV[q];
  *mx = b->ia_setmax[q];
  if (b->ia_set_pending[q]) HIPCHK(hipStreamWaitEvent(b->stream, b->ev_set_done[q], 0));
  HIPCHK(hipMemsetAsync(*mx, 0, (size_t)2 * b->G * sizeof(unsigned long long), b->stream));
  return ODINN_OK;
}
static int interp_async_contract(odinn_batch* b, const Pools& P) {
  const int l = b->ia_q % b->ia_lanes, qs = b->ia_q % b->ia_sets;
  const odinn_batch::IaLane& a = b->ia_lane[l];
  HIPCHK(hipEventRecord(b->ev_emit[l], b->stream));
  HIPCHK(hipStreamWaitEvent(b->ia_stream[l], b->ev_emit[l], 0));
  const int rc = (b->ia_nact > 0 && b->ia_select)
                     ? launch_interp_theta_select(b->ia_stream[l], P, b->lawdev(), b->n_interp_half, b->G, b->ia_nact, b->ia_nodeH[qs],
                                                  b->ia_nodeV[qs], b->ia_act, b->ia_aoff, a.sH, a.sel, a.knots, a.M, a.ab,
                                                  b->d_dthq + (size_t)b->ia_q * b->G * b->P, 0,
                                                  b->ia_emit_fused ? b->ia_setmax[qs] : nullptr,
                                                  b->ia_emit_fused ? b->ia_setmax[qs] + b->G : nullptr)
                 : b->ia_nact > 0
                     ? launch_interp_theta_active(b->ia_stream[l], P, b->lawdev(), b->n_interp_half, b->G, b->ia_nact, b->ia_nodeH[qs],
                                                  b->ia_nodeV[qs], b->ia_act, b->ia_gid_act, b->ia_aoff, b->d_ib_iota, a.sH, a.sV, a.iA, a.tmp,
                                                  b->ib_tmp_bytes, a.knots, a.M, a.ab, b->d_dthq + (size_t)b->ia_q * b->G * b->P, 0)
                     : launch_interp_theta_batch(b->ia_stream[l], P, b->lawdev(), b->n_interp_half, 0, b->G, 0, b->ntotd, b->ia_nodeH[qs],
                                           b->ia_nodeV[qs],
                                           b->d_ib_gid, b->d_ib_iota, a.sH, a.sV, a.iA, a.iB, a.kA, a.kB, a.tmp, b->ib_tmp_bytes, a.knots,
                                           a.M, a.ab, b->d_dthq + (size_t)b->ia_q * b->G * b->P, 0);
  if (rc) return fail(ODINN_ERR_HIP, "gradient interpolation failed (code %d)", rc);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b->ev_done[l], b->ia_stream[l]));
  HIPCHK(hipEventRecord(b->ev_set_done[qs], b->ia_stream[l]));
  b->ia_pending[l] = true;
  b->ia_set_pending[qs] = true;
  ++b->ia_q;
  return ODINN_OK;
}
struct InterpAsyncScope {  // leaves no work behind on the lane streams, whichever way the driver returns
  odinn_batch* b;
  ~InterpAsyncScope() {
    if (!b->interp_async) return;
    b->interp_async = false;
    for (int l = 0; l < b->ia_lanes; ++l) {
      if (b->ia_stream[l]) (void)hipStreamSynchronize(b->ia_stream[l]);
      b->ia_pending[l] = false;
    }
    for (int q = 0; q < b->ia_sets; ++q) b->ia_set_pending[q] = false;
    b->ia_q = 0;
  }
};
// the adjoint drivers switch the overlap on for their reverse loop ...
// Lanes by default (8 x 512^2, table path, ms per gradient; profiles/r04/ytab_lanes.txt): the reverse-Euler loop of the
// DiscreteAdjoint is short and its contractions are the longer chain -- 0 / 1 / 2 / 3 lanes: 9.7 / 9.3 / 8.4 / 8.4; the reverse
// ODE of the ContinuousAdjoint keeps the GPU busy itself and more than one lane only takes bandwidth from it: 212 / 151 / 190 / 179
// (low-priority lanes: 470; a high-priority batch stream brings 2 - 3 lanes back to 152, no better than one lane) -- that is with
// five stage launches per reverse step; with ONE fused launch (k_adj_fused_strip<..., YT>) the lanes are the longer chain: 4.
static int interp_async_enable(odinn_batch* b, bool useV, int lanes_default) {
  b->ia_emit_fused = false;  // (the continuous adjoint's driver raises it for its fused emission)
  if (b->law_kind != ODINN_LAW_NN_Y) return ODINN_OK;
  CHK(ensure_interp_scratch(b));
  const int lanes = interp_async_lanes(b, useV, lanes_default);
  if (lanes < 1) return ODINN_OK;
  CHK(interp_async_setup(b, lanes));
  b->interp_async = true;
  return ODINN_OK;
}

static int theta_vjp_launch(odinn_batch* b, const double* H, const double* lam, const double* scales, int g,
                            bool accumulate, double* part_deferred = nullptr, bool inplace = false,
                            const double* lam_alt = nullptr, const double* snaps = nullptr, const AdjState* adj = nullptr) {
  // part_deferred (A-type laws only): the per-tile partials go there and are NOT summed here;
  // inplace: they are ADDED onto part_deferred (which the caller zeroed and reduces at the end)
  // g < 0: all glaciers (swizzled table); result of A-type laws lands in d_Gsum / d_Gacc,
  // of Y/U laws in d_dth[g][P]
  const bool nn_node = b->law_kind >= ODINN_LAW_NN_Y;
  const int base = g < 0 ? 0 : b->gd[g].tile0, nblk = g < 0 ? b->ntiles : b->gd[g].ntiles;
  const int ng = g < 0 ? b->G : 1, g0 = g < 0 ? 0 : g;
  const bool linear = b->law_kind >= ODINN_LAW_NN_Y && b->grad_interp == ODINN_GRAD_INTERP_LINEAR;
  const bool isU = b->law_kind == ODINN_LAW_NN_U;
  // run-time architectures (law mode 2) with exact per-node backprop: the kernel emits (Hbar, weight[, |grad S|]) and
  // k_node_backprop contracts them (wave-reduced); the compile-time architectures backpropagate inside k_vjp_theta_nn
  const bool emit_rt = nn_node && !linear && lm_is_rt(b->lm()) && (size_t)NW * b->P * sizeof(double) + (size_t)b->P * sizeof(int) <= 30 * 1024;
  const bool emit = linear || emit_rt;
  if (nn_node) CHK(ensure_theta_scratch(b, std::max(nblk, (int)node_backprop_part_count(ng, 1)), !emit));
  const bool async = linear && !isU && g < 0 && accumulate && b->interp_async;
  double *nH = b->d_nodeH, *nV = b->d_nodeV;
  if (async) CHK(interp_async_begin(b, &nH, &nV));
  else if (emit) CHK(interp_prepare(b, g, isU));
  if (emit && !async) { nH = b->d_nodeH; nV = b->d_nodeV; }  // (allocated by interp_prepare)
  ThArgs A{};
  A.emitH = emit ? nH : nullptr; A.emitV = emit ? nV : nullptr; A.emitS = (emit && isU) ? b->d_nodeS : nullptr;
  A.H = H; A.lam = lam; A.lam_alt = lam_alt; A.scales = scales;
  A.snaps = snaps; A.adj = adj; A.ntot = b->ntot;
  A.Gacc = b->wants_Gacc() ? b->d_Gacc : nullptr;
  A.part_theta = nn_node ? b->d_part_theta : nullptr;
  A.gscratch = (nn_node && !emit) ? b->d_gscratch : nullptr;
  A.accum = (inplace && part_deferred && !nn_node) ? 1 : 0;
  Pools P = b->pools(g < 0);
  if (part_deferred) P.part = part_deferred;
  launch_vjp_theta(b, nblk, P, b->lawdev(), A, base);
  if (async) {
    CHK(interp_async_contract(b, P));
  } else if (linear) {
    CHK(interp_contract(b, g, isU, accumulate, P));
  } else if (emit_rt) {
    if (launch_node_backprop(b->stream, P, b->lawdev(), g0, ng, b->ntotd, b->d_nodeH, b->d_nodeS, b->d_nodeV, b->d_part_theta, b->d_dth,
                             accumulate ? 1 : 0))
      return fail(ODINN_ERR_HIP, "per-node backprop of the emitted node weights failed");
  } else if (part_deferred) {
  } else if (nn_node)
    launch_sum_part_theta(b->P, ng, b->stream, P, b->d_part_theta, b->d_dth, accumulate ? 1 : 0, g0);
  else
    launch_sum_part(ng, b->stream, P, 2, b->d_Gsum, accumulate ? 1 : 0, g0);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// ... and contract them: dtheta_g (+)= sum_knots c_k d law / d theta (knot_k) -- sort the glacier's nodes by Hbar, build its knots,
// sum the node weights per interval, backpropagate at the knots only
static int interp_contract(odinn_batch* b, int g, bool linU, bool accumulate, const Pools& P) {
  const int ng = g < 0 ? b->G : 1, g0 = g < 0 ? 0 : g;
  {
    // dtheta_g (+)= sum_knots c_k dY/dtheta(T_g, knot_k): sort the glacier's nodes by Hbar, build its knots, sum per interval
    const LawDev L = b->lawdev();
    // Y law: all glaciers of the call in one sequence of launches (ODINN_INTERP_BATCH=0: one sequence per glacier)
    const int eb = sched_val(b->sched.interp_batch, "ODINN_INTERP_BATCH");
    if (!linU && b->d_ib_gid && eb != 0 && interp_batch_lds_bytes(b->P) <= 30 * 1024) {
      const long long lo = b->gd[g0].offd;
      const long long hi = g0 + ng < b->G ? b->gd[g0 + ng].offd : b->ntotd;
      const int rc = launch_interp_theta_batch(b->stream, P, L, b->n_interp_half, g0, ng, lo, hi - lo, b->d_nodeH, b->d_nodeV,
                                               b->d_ib_gid, b->d_ib_iota, b->d_ib_sH, b->d_ib_sV, b->d_ib_iA, b->d_ib_iB, b->d_ib_kA,
                                               b->d_ib_kB, b->d_ib_tmp, b->ib_tmp_bytes, b->d_ib_knots, b->d_ib_M, b->d_ib_ab,
                                               b->d_dth, accumulate ? 1 : 0);
      if (rc) return fail(ODINN_ERR_HIP, "gradient interpolation failed (code %d)", rc);
      HIPCHK(hipGetLastError());
      return ODINN_OK;
    }
    // glaciers are independent (own nodes, own dtheta slot): their sequences run side by side on the lane streams, forked
    // from and joined back into the batch's stream by events
    const int lanes = std::min(ng, b->interp_lanes);
    if (lanes > 1) {
      HIPCHK(hipEventRecord(b->ev_fork, b->stream));
      for (int l = 0; l < lanes; ++l) HIPCHK(hipStreamWaitEvent(b->side[l], b->ev_fork, 0));
    }
    const size_t Pk = (size_t)std::max(b->P, 1) * INTERP_KMAX, ucn = (size_t)4 * (INTERP_KMAX - 1) * (INTERP_KMAX - 1);
    for (int q = g0; q < g0 + ng; ++q) {
      const GDev& r = b->gd[q];
      const long long nd = (long long)(r.nx - 1) * (r.ny - 1);
      const int l = lanes > 1 ? (q - g0) % lanes : 0;
      hipStream_t st = lanes > 1 ? b->side[l] : b->stream;
      double* sH = b->d_sortH + (size_t)l * b->interp_ndmax;
      double* sV = b->d_sortV + (size_t)l * b->interp_ndmax;
      void* tmp = static_cast<char*>(b->d_sorttmp) + (size_t)l * std::max<size_t>(b->sorttmp_bytes, 256);
      const int rc = linU ? launch_interp_theta_U(st, L, b->n_interp_half, b->d_nodeH + r.offd, b->d_nodeS + r.offd,
                                                  b->d_nodeV + r.offd, nd, sH, sV, tmp, b->sorttmp_bytes,
                                                  b->d_ucell + (size_t)l * ucn, b->d_knotG + (size_t)l * Pk, b->d_interp_err,
                                                  b->d_dth + (size_t)q * b->P, accumulate ? 1 : 0)
                          : launch_interp_theta(st, L, b->descs[q].T, b->n_interp_half, b->d_nodeH + r.offd, b->d_nodeV + r.offd, nd,
                                                sH, sV, tmp, b->sorttmp_bytes, b->d_knots + (size_t)l * INTERP_KMAX, b->d_knotM + l,
                                                b->d_knotG + (size_t)l * Pk, b->d_knotab + (size_t)l * 2 * INTERP_KMAX,
                                                b->d_dth + (size_t)q * b->P, accumulate ? 1 : 0);
      if (rc) return fail(ODINN_ERR_HIP, "gradient interpolation failed (code %d)", rc);
    }
    if (lanes > 1)
      for (int l = 0; l < lanes; ++l) {
        HIPCHK(hipEventRecord(b->ev_join[l], b->side[l]));
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_join[l], 0));
      }
  }
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

// gridded hoisted law: dtheta = sum_nodes Gacc * dA/dtheta(T)  over dual range [lo, lo+n)
static int gridded_law_grad(odinn_batch* b, long long lo, long long n, double* dtheta_host) {
  int nblk = (int)((n + NT - 1) / NT);
  // wave-reduced kernel (accumulators in LDS) while they fit; ODINN_LAWGRAD_WAVE=0: per-thread accumulators in global memory
  const int ew = sched_val(b->sched.lawgrad_wave, "ODINN_LAWGRAD_WAVE");
  const size_t dyn = (size_t)NW * b->P * sizeof(double) + (size_t)b->P * sizeof(int);
  if (ew != 0 && dyn <= 30 * 1024) {  // + 33 KB of static staging area: within the 64 KB of a workgroup
    const int max_rows = 2048;
    CHK(ensure_theta_scratch(b, max_rows, false));
    nblk = launch_law_field_grad(b->stream, b->lawdev(), b->d_Tfield + lo, b->d_Gacc + lo, n, b->d_part_theta, max_rows);
  } else {
    CHK(ensure_theta_scratch(b, nblk));
    launch_law_field_grad_scratch(nblk, b->stream, b->lawdev(), b->d_Tfield + lo, b->d_Gacc + lo, n, b->d_gscratch,
                                  b->d_part_theta);
  }
  launch_sum_rows(b->P, b->stream, b->d_part_theta, nblk, b->d_dth);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dtheta_host, b->d_dth, sizeof(double) * b->P, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ODINN_OK;
}

int odinn_sia2d_vjp_theta(odinn_batch* b, int g, const double* lam, const double* H, double t, double* dtheta, int P) {
  (void)t;
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !lam || !dtheta) return fail(ODINN_ERR_ARG, "null argument");
  const int Pexp = b->law_kind == ODINN_LAW_CONST_A ? 1 : b->P;
  if (P != Pexp) return fail(ODINN_ERR_ARG, "dtheta has %d entries, expected %d", P, Pexp);
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  CHK(up_field(b, g, b->d_lam[0], lam));
  const GDev& r = b->gd[g];
  const long long nd = (long long)(r.nx - 1) * (r.ny - 1);
  if (b->law_kind == ODINN_LAW_NN_A_GRIDDED)
    HIPCHK(hipMemsetAsync(b->d_Gacc + r.offd, 0, nd * sizeof(double), b->stream));
  CHK(theta_vjp_launch(b, b->d_tmpA, b->d_lam[0], nullptr, g, false));
  if (b->law_kind >= ODINN_LAW_NN_Y) {
    HIPCHK(hipMemcpyAsync(dtheta, b->d_dth + (size_t)g * b->P, sizeof(double) * b->P, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return check_interp_bounds(b);
  }
  if (b->law_kind == ODINN_LAW_NN_A_GRIDDED) return gridded_law_grad(b, r.offd, nd, dtheta);
  double Gs = 0.0;
  HIPCHK(hipMemcpyAsync(&Gs, b->d_Gsum + g, sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->law_kind == ODINN_LAW_CONST_A) { dtheta[0] = Gs; return ODINN_OK; }
  std::vector<double> dA(b->P);
  h_mlp(b->mlp, b->theta.data(), &b->descs[g].T, dA.data());
  for (int k = 0; k < b->P; ++k) dtheta[k] = dA[k] * Gs;  // cartesian_tensor contraction, target_utils.jl:156-161
  return ODINN_OK;
}

int odinn_set_grad_interpolation(odinn_batch* b, int kind, int n_interp_half) {
  if (!b || (kind != ODINN_GRAD_INTERP_NONE && kind != ODINN_GRAD_INTERP_LINEAR)) return fail(ODINN_ERR_ARG, "bad interpolation kind");
  if (kind == ODINN_GRAD_INTERP_LINEAR && (n_interp_half < 2 || 2 * n_interp_half > INTERP_KMAX))
    return fail(ODINN_ERR_ARG, "n_interp_half must be in [2, %d]", INTERP_KMAX / 2);
  if (kind == ODINN_GRAD_INTERP_LINEAR && b->law_kind < ODINN_LAW_NN_Y)
    return fail(ODINN_ERR_UNSUPPORTED, "the A-type laws have no spatial law gradient to interpolate (Y and U laws only)");
  b->grad_interp = kind;
  if (kind == ODINN_GRAD_INTERP_LINEAR) b->n_interp_half = n_interp_half;
  return ODINN_OK;
}

int odinn_set_vjp_method(odinn_batch* b, int method) {
  if (!b || (method != ODINN_VJP_DISCRETE && method != ODINN_VJP_CONTINUOUS)) return fail(ODINN_ERR_ARG, "bad VJP method");
  b->vjp_method = method;
  return ODINN_OK;
}

int odinn_set_surface_velocity_factor(odinn_batch* b, double f) {
  if (!b || !(f > 0.0)) return fail(ODINN_ERR_ARG, "f_surface_velocity_factor must be positive");
  b->fV = f;
  return ODINN_OK;
}

int odinn_set_thickness_loss_function(odinn_batch* b, int simple_loss, double eps) {
  if (!b || (simple_loss != ODINN_SIMPLE_L2SUM && simple_loss != ODINN_SIMPLE_LOGSUM)) return fail(ODINN_ERR_ARG, "unknown simple loss");
  if (simple_loss == ODINN_SIMPLE_LOGSUM && !(eps > 0.0)) return fail(ODINN_ERR_ARG, "LogSum needs eps > 0");
  b->h_log_eps = simple_loss == ODINN_SIMPLE_LOGSUM ? eps : 0.0;
  return ODINN_OK;
}

int odinn_set_velocity_loss_function(odinn_batch* b, int simple_loss, double eps) {
  if (b) b->refs_version++;
  if (!b || (simple_loss != ODINN_SIMPLE_L2SUM && simple_loss != ODINN_SIMPLE_LOGSUM)) return fail(ODINN_ERR_ARG, "unknown simple loss");
  if (simple_loss == ODINN_SIMPLE_LOGSUM && !(eps > 0.0)) return fail(ODINN_ERR_ARG, "LogSum needs eps > 0");
  b->v_log_eps = simple_loss == ODINN_SIMPLE_LOGSUM ? eps : 0.0;
  return ODINN_OK;
}

int odinn_set_loss(odinn_batch* b, int kind, int v_component_abs, int v_scale_loss, double hv_scaling) {
  if (b) b->refs_version++;
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  if (kind < ODINN_LOSS_H || kind > ODINN_LOSS_HV) return fail(ODINN_ERR_ARG, "unknown loss kind %d", kind);
  b->loss_kind = kind; b->v_abs = v_component_abs ? 1 : 0; b->v_scale_loss = v_scale_loss ? 1 : 0;
  b->hv_scaling = hv_scaling;
  return ODINN_OK;
}

int odinn_set_dhdt_reference(odinn_batch* b, int g, double t0, double t1, double dhdt_ref) {
  CHK(check_g(b, g));
  b->dh_t0[g] = t0; b->dh_t1[g] = t1; b->dh_ref[g] = dhdt_ref;
  return ODINN_OK;
}

int odinn_set_dhdt_loss(odinn_batch* b, double weight) {
  if (!b || !(weight >= 0.0)) return fail(ODINN_ERR_ARG, "bad LossDhdt weight");
  b->dhdt_weight = weight;
  return ODINN_OK;
}

int odinn_set_avgv_reference(odinn_batch* b, int g, double t1, double t2, const double* Vabs, const double* Vx,
                             const double* Vy) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!(t2 > t1)) { b->av_t1[g] = 0.0; b->av_t2[g] = 0.0; return ODINN_OK; }
  if (!Vabs || !Vx || !Vy) return fail(ODINN_ERR_ARG, "null velocity sample");
  if (!b->d_aVabs) {
    CHK(dalloc(&b->d_aVabs, (size_t)b->ntot)); CHK(dalloc(&b->d_aVx, (size_t)b->ntot)); CHK(dalloc(&b->d_aVy, (size_t)b->ntot));
    HIPCHK(hipMemset(b->d_aVabs, 0, (size_t)b->ntot * sizeof(double)));
    HIPCHK(hipMemset(b->d_aVx, 0, (size_t)b->ntot * sizeof(double)));
    HIPCHK(hipMemset(b->d_aVy, 0, (size_t)b->ntot * sizeof(double)));
  }
  const GDev& r = b->gd[g];
  const size_t nb = (size_t)r.nx * r.ny * sizeof(double);
  HIPCHK(hipMemcpy(b->d_aVabs + r.off, Vabs, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_aVx + r.off, Vx, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_aVy + r.off, Vy, nb, hipMemcpyHostToDevice));
  b->av_t1[g] = t1; b->av_t2[g] = t2;
  return ODINN_OK;
}

int odinn_set_velocity_regularization(odinn_batch* b, double weight, int distance) {
  if (!b || !(weight >= 0.0) || distance < 0 || distance > 32) return fail(ODINN_ERR_ARG, "bad VelocityRegularization weight / distance");
  b->vreg_weight = weight; b->vreg_dist = distance;
  return ODINN_OK;
}

int odinn_set_avgv_loss(odinn_batch* b, double weight, double step, int component_abs) {
  if (!b || !(weight >= 0.0) || !(step > 0.0)) return fail(ODINN_ERR_ARG, "bad LossAvgV weight / step");
  b->avgv_weight = weight; b->avgv_step = step; b->avgv_abs = component_abs ? 1 : 0;
  return ODINN_OK;
}

int odinn_set_velocity_reference(odinn_batch* b, int g, int n_ref, const double* t_ref, const double* Vabs,
                                 const double* Vx, const double* Vy) {
  if (b) b->refs_version++;
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (n_ref < 0 || (n_ref > 0 && (!t_ref || !Vabs || !Vx || !Vy))) return fail(ODINN_ERR_ARG, "bad velocity data");
  if (n_ref > b->nvref_alloc) {
    double *na = nullptr, *nx_ = nullptr, *ny_ = nullptr;
    const size_t nb = (size_t)n_ref * b->ntot;
    auto grow = [&]() -> int {
      CHK(dalloc(&na, nb)); CHK(dalloc(&nx_, nb)); CHK(dalloc(&ny_, nb));
      if (b->d_Vabs) {
        const size_t ob = (size_t)b->nvref_alloc * b->ntot * sizeof(double);
        HIPCHK(hipMemcpy(na, b->d_Vabs, ob, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nx_, b->d_Vxr, ob, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(ny_, b->d_Vyr, ob, hipMemcpyDeviceToDevice));
        HIPCHK(hipStreamSynchronize(nullptr));  // D2D copies on the NULL stream do not block the host
      }
      return ODINN_OK;
    };
    if (const int rc = grow()) { dfree(na); dfree(nx_); dfree(ny_); return rc; }
    dfree(b->d_Vabs); dfree(b->d_Vxr); dfree(b->d_Vyr);
    b->d_Vabs = na; b->d_Vxr = nx_; b->d_Vyr = ny_; b->nvref_alloc = n_ref;
  }
  const GDev& r = b->gd[g];
  const size_t n = (size_t)r.nx * r.ny;
  b->t_vref[g].assign(t_ref, t_ref + n_ref);
  b->v_scale[g].assign(n_ref, 1.0); b->v_cxy[g].assign(n_ref, 0.0); b->v_cabs[g].assign(n_ref, 0.0);
  b->v_edge[g].assign(n_ref, std::vector<double>());
  for (int m = 0; m < n_ref; ++m) {
    const double *va = Vabs + m * n, *vx = Vx + m * n, *vy = Vy + m * n;
    double s2 = 0.0, cxy = 0.0, cabs = 0.0;
    long long cnt = 0;
    for (int j = 0; j < r.ny; ++j)
      for (int i = 0; i < r.nx; ++i) {
        const size_t c = i + (size_t)r.nx * j;
        if (va[c] > 0.0) {
          s2 += vx[c] * vx[c] + vy[c] * vy[c];
          ++cnt;
          if (i == r.nx - 1 || j == r.ny - 1) { cxy += vx[c] * vx[c] + vy[c] * vy[c]; cabs += va[c] * va[c]; b->v_edge[g][m].push_back(va[c]); }
        }
      }
    b->v_scale[g][m] = cnt > 0 ? 1.0 / std::sqrt(s2 / (double)cnt) : 1.0;
    b->v_cxy[g][m] = cxy / (double)n;
    b->v_cabs[g][m] = cabs / (double)n;
    HIPCHK(hipMemcpy(b->d_Vabs + (size_t)m * b->ntot + r.off, va, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_Vxr + (size_t)m * b->ntot + r.off, vx, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_Vyr + (size_t)m * b->ntot + r.off, vy, n * sizeof(double), hipMemcpyHostToDevice));
  }
  return ODINN_OK;
}

int odinn_surface_V(odinn_batch* b, int g, const double* H, double* Vx, double* Vy) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !Vx || !Vy) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  launch_surface_V(b->lm(), b->gd[g].ntiles, b->stream, b->pools(false), b->lawdev(), b->d_tmpA, b->d_tmpB, b->d_lam[1], b->gd[g].tile0, 1.0 / b->fV);
  HIPCHK(hipGetLastError());
  CHK(down_field(b, g, b->d_tmpB, Vx));
  return down_field(b, g, b->d_lam[1], Vy);
}

static int surfV_vjp_common(odinn_batch* b, int g, const double* dVx, const double* dVy, const double* H) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !dVx || !dVy) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  CHK(up_field(b, g, b->d_lam[0], dVx));
  CHK(up_field(b, g, b->d_lam[1], dVy));
  const GDev& r = b->gd[g];
  if (b->law_kind == ODINN_LAW_NN_A_GRIDDED)
    HIPCHK(hipMemsetAsync(b->d_Gacc + r.offd, 0, (size_t)(r.nx - 1) * (r.ny - 1) * sizeof(double), b->stream));
  VArgs A{};
  A.H = b->d_tmpA; A.dVx = b->d_lam[0]; A.dVy = b->d_lam[1]; A.out = b->d_tmpB;
  A.Gacc = b->law_kind == ODINN_LAW_NN_A_GRIDDED ? b->d_Gacc : nullptr;
  A.finv = 1.0 / b->fV;
  CHK(vel_theta_args(b, A, g));
  const Pools P = b->pools(false);
  launch_surfV_vjp(b->lm(), 0, r.ntiles, b->stream, P, b->lawdev(), A, r.tile0);
  if (b->vel_nn()) CHK(vel_theta_finish(b, g, false, P));
  else launch_sum_part(1, b->stream, P, 3, b->d_Gsum, 0, g);
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

int odinn_surface_V_vjp_H(odinn_batch* b, int g, const double* dVx, const double* dVy, const double* H, double* out) {
  if (!out) return fail(ODINN_ERR_ARG, "null field");
  CHK(surfV_vjp_common(b, g, dVx, dVy, H));
  return down_field(b, g, b->d_tmpB, out);
}

int odinn_surface_V_vjp_theta(odinn_batch* b, int g, const double* dVx, const double* dVy, const double* H,
                              double* dtheta, int P) {
  if (!dtheta) return fail(ODINN_ERR_ARG, "null argument");
  CHK(check_g(b, g));
  const int Pexp = b->law_kind == ODINN_LAW_CONST_A ? 1 : b->P;
  if (P != Pexp) return fail(ODINN_ERR_ARG, "dtheta has %d entries, expected %d", P, Pexp);
  CHK(surfV_vjp_common(b, g, dVx, dVy, H));
  const GDev& r = b->gd[g];
  if (b->law_kind == ODINN_LAW_NN_A_GRIDDED) return gridded_law_grad(b, r.offd, (long long)(r.nx - 1) * (r.ny - 1), dtheta);
  if (b->vel_nn()) {
    HIPCHK(hipMemcpyAsync(dtheta, b->d_dth + (size_t)g * b->P, sizeof(double) * b->P, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return check_interp_bounds(b);  // (U law with `:Linear`: a node outside the gradient interpolant's grid)
  }
  double Gs = 0.0;
  HIPCHK(hipMemcpyAsync(&Gs, b->d_Gsum + g, sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->law_kind == ODINN_LAW_CONST_A) { dtheta[0] = Gs; return ODINN_OK; }
  std::vector<double> dA(b->P);
  h_mlp(b->mlp, b->theta.data(), &b->descs[g].T, dA.data());
  for (int k = 0; k < b->P; ++k) dtheta[k] = dA[k] * Gs;
  return ODINN_OK;
}

int odinn_mb_apply(odinn_batch* b, int g, const double* H, double* H_new, double* MB_applied) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!H || !H_new) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b));
  CHK(up_field(b, g, b->d_tmpA, H));
  launch_mb_apply(b->gd[g].ntiles, b->stream, b->pools(false), b->d_tmpA, b->d_mb0,
                  b->any_sref ? b->d_Sref : nullptr, b->d_tmpB, b->d_lam[1], b->gd[g].tile0);
  HIPCHK(hipGetLastError());
  CHK(down_field(b, g, b->d_tmpB, H_new));
  if (MB_applied) CHK(down_field(b, g, b->d_lam[1], MB_applied));
  return ODINN_OK;
}

int odinn_mb_vjp_H(odinn_batch* b, int g, const double* lam, const double* H_pre, double* out) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!lam || !H_pre || !out) return fail(ODINN_ERR_ARG, "null field");
  CHK(refresh_gd(b));
  CHK(up_field(b, g, b->d_tmpA, H_pre));
  CHK(up_field(b, g, b->d_lam[0], lam));
  launch_mb_vjp(b->gd[g].ntiles, b->stream, b->pools(false), b->d_tmpA, b->d_mb0,
                b->any_sref ? b->d_Sref : nullptr, b->d_lam[0], b->d_tmpB, 0, b->gd[g].tile0);
  HIPCHK(hipGetLastError());
  return down_field(b, g, b->d_tmpB, out);
}

int odinn_eval_law(odinn_batch* b, int g, const double* H, double* out, int n_out) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  const GDev& r = b->gd[g];
  const long long nd = (long long)(r.nx - 1) * (r.ny - 1);
  const bool scalar = (b->law_kind == ODINN_LAW_NN_A_SCALAR) || (b->law_kind == ODINN_LAW_CONST_A && !r.use_Afield);
  if (scalar) {
    if (n_out < 1) return fail(ODINN_ERR_ARG, "n_out too small");
    out[0] = r.A;
    return ODINN_OK;
  }
  if (n_out < nd) return fail(ODINN_ERR_ARG, "n_out=%d < %lld dual nodes", n_out, nd);
  if (!H) return fail(ODINN_ERR_ARG, "null field");
  CHK(up_field(b, g, b->d_tmpA, H));
  launch_eval_law(b->stream, b->pools(false), b->lawdev(), b->d_tmpA, b->d_Gacc + r.offd, g, nd);
  HIPCHK(hipGetLastError());
  return down_field(b, g, b->d_Gacc, out, true);
}

// ---- time loop ----------------------------------------------------------------------------
int odinn_solve(odinn_batch* b, int n_stops, const double* tstops, int n_mb, const double* mb_times,
                const odinn_solver_opts* opts, odinn_solve_stats* stats) {
  if (!b || !tstops) return fail(ODINN_ERR_ARG, "null argument");
  return do_solve(b, n_stops, tstops, n_mb, mb_times, opts, stats);
}

int odinn_set_glacier_stops(odinn_batch* b, int g, int n, const double* t) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  CHK(check_g(b, g));
  if (n != 0 && (n < 2 || !t)) return fail(ODINN_ERR_ARG, "a glacier needs at least 2 stops (n = 0 clears its table)");
  for (int j = 1; j < n; ++j)
    if (!(t[j] > t[j - 1])) return fail(ODINN_ERR_ARG, "the stops of glacier %d must be strictly increasing", g);
  b->own_stops.resize(b->G);
  b->own_stops[g].assign(t, t + n);
  return ODINN_OK;
}

int odinn_set_schedule(odinn_batch* b, const odinn_schedule* sc) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  const odinn_schedule automatic = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, {0}};
  const odinn_schedule want = sc ? *sc : automatic;  // (validated as a local: a rejected schedule leaves the old one in effect)
  if (want.adj_rows >= 0 && want.adj_rows != 2 && want.adj_rows != 4 && want.adj_rows != 7 && want.adj_rows != 8)
    return fail(ODINN_ERR_ARG, "odinn_schedule.adj_rows must be -1, 2, 4, 7 or 8");
  if (want.fused_tiles > 4) return fail(ODINN_ERR_ARG, "odinn_schedule.fused_tiles must be -1 ... 4");
  if (want.law_table != b->sched.law_table) b->gd_dirty = true;  // (the Y law's table is built by refresh_gd)
  b->sched = want;
  return ODINN_OK;
}

int odinn_get_schedule(odinn_batch* b, odinn_schedule* out) {
  if (!b || !out) return fail(ODINN_ERR_ARG, "null argument");
  *out = b->sched;
  // what is in effect: an environment override wins over the field
  out->step_sc = sched_val(b->sched.step_sc, "ODINN_STEP_SC");
  out->fused_tiles = b->fused_override() ? b->fused_override() : -1;
  out->dhdt_strip = sched_val(b->sched.dhdt_strip, "ODINN_DHDT_STRIP");
  out->vjph_strip = sched_val(b->sched.vjph_strip, "ODINN_VJPH_STRIP");
  out->vjpth_strip = sched_val(b->sched.vjpth_strip, "ODINN_VJPTH_STRIP");
  out->snap_on_load = sched_val(b->sched.snap_on_load, "ODINN_SNAP_ON_LOAD");
  out->interp_streams = sched_val(b->sched.interp_streams, "ODINN_INTERP_STREAMS");
  out->interp_batch = sched_val(b->sched.interp_batch, "ODINN_INTERP_BATCH");
  out->lawgrad_wave = sched_val(b->sched.lawgrad_wave, "ODINN_LAWGRAD_WAVE");
  out->vq_onepass = sched_val(b->sched.vq_onepass, "ODINN_VQ_ONEPASS");
  out->adj_fused = sched_val(b->sched.adj_fused, "ODINN_ADJ_FUSED");
  out->adj_skip = sched_val(b->sched.adj_skip, "ODINN_ADJ_SKIP");
  out->adj_segs = sched_val(b->sched.adj_segs, "ODINN_ADJ_SEGS");
  out->adj_rows = sched_val(b->sched.adj_rows, "ODINN_ADJ_ROWS");
  out->adj_theta_fused = sched_val(b->sched.adj_theta_fused, "ODINN_ADJ_THETA_FUSED");
  out->law_table = sched_val(b->sched.law_table, "ODINN_LAW_TABLE");
  out->interp_async = sched_val(b->sched.interp_async, "ODINN_INTERP_ASYNC");
  out->adj_sc = sched_val(b->sched.adj_sc, "ODINN_ADJ_SC");
  return ODINN_OK;
}

int odinn_get_law_table(odinn_batch* b, int* usable, int* n_intervals, double* max_rel_dev, double* hmax_per_glacier) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  CHK(use_dev(b));
  CHK(refresh_gd(b));
  if (usable) *usable = b->ytab_ok ? 1 : 0;
  if (n_intervals) *n_intervals = b->ytab_ni;
  if (max_rel_dev) *max_rel_dev = b->ytab_ok || b->ytab_wanted() ? std::max(b->ytab_err_rel, b->ytab_ymax > 0.0 ? b->ytab_err_abs / b->ytab_ymax : 0.0) : 0.0;
  if (hmax_per_glacier)
    for (int g = 0; g < b->G; ++g)
      hmax_per_glacier[g] = !b->ytab_wanted() ? 0.0 : b->law_kind == ODINN_LAW_NN_U ? b->utab_hmax : g < (int)b->ytab_hmax.size() ? b->ytab_hmax[g] : 0.0;
  if (n_intervals && b->law_kind == ODINN_LAW_NN_U) *n_intervals = b->utab_nh * b->utab_ns;
  return ODINN_OK;
}

int odinn_get_snapshot(odinn_batch* b, int g, int istop, double* H_out) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!b->solved || g >= (int)b->ts_g.size()) return fail(ODINN_ERR_STATE, "no solve has been run");
  // (istop counts the glacier's OWN result stops: the table of odinn_set_glacier_stops, or the tstops of the solve)
  if (istop < 0 || istop >= b->nres(g)) return fail(ODINN_ERR_ARG, "istop out of range");
  return down_field(b, g, b->d_snaps + (size_t)istop * b->ntot, H_out);
}

int odinn_get_H(odinn_batch* b, int g, double* H_out) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!b->solved || g >= (int)b->ts_g.size() || b->nres(g) < 1) return down_field(b, g, b->d_H0, H_out);
  return down_field(b, g, b->d_snaps + (size_t)(b->nres(g) - 1) * b->ntot, H_out);
}

int odinn_loss(odinn_batch* b, double* loss_per_glacier) {
  if (!b || !loss_per_glacier) return fail(ODINN_ERR_ARG, "null argument");
  CHK(use_dev(b));
  if (!b->solved) return fail(ODINN_ERR_STATE, "no solve has been run");
  CHK(upload_loss_tables(b));
  double cl = 0.0;
  CHK(do_loss(b, &cl));
  HIPCHK(hipMemcpyAsync(loss_per_glacier, b->d_lossacc, sizeof(double) * b->G, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  // data-only part of LossV (cells of the last row / column, where V_pred is 0 by construction)
  const int k = b->K();
  for (int j = 1; j < k; ++j)
    for (int g = 0; g < b->G; ++g) {
      const size_t q = (size_t)j * b->G + g;
      if (b->loss_kind != ODINN_LOSS_H && b->wv_h[q] != 0.0)
        loss_per_glacier[g] += b->wv_h[q] * b->vsc_h[q] * b->v_const(g, b->vslot_h[q]);
    }
  return ODINN_OK;
}

// forward solve + zeroed gradient accumulators (common to both adjoints)
static int grad_prepare(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                        const double* mb_times, const odinn_solver_opts* opts, odinn_solve_stats* stats, int n_quadrature = 0) {
  CHK(use_dev(b));
  if (theta) CHK(odinn_set_theta(b, theta, P));
  const int Pexp = b->law_kind == ODINN_LAW_CONST_A ? 1 : b->P;
  if (P != Pexp) return fail(ODINN_ERR_ARG, "dtheta has %d entries, expected %d", P, Pexp);
  CHK(check_loss_terms(b));
  if (b->loss_kind != ODINN_LOSS_V && !b->d_Href && !b->dhdt_on() && !b->avgv_on() && !b->vreg_on()) return fail(ODINN_ERR_STATE, "no reference thickness data set");
  if (b->loss_kind != ODINN_LOSS_H && !b->d_Vabs) return fail(ODINN_ERR_STATE, "no reference velocity data set");
  CHK(do_solve(b, n_stops, tstops, n_mb, mb_times, opts, stats));
  const size_t fb = (size_t)b->ntot * sizeof(double);
  HIPCHK(hipMemsetAsync(b->d_lam[0], 0, fb, b->stream));  // lambda_k = 0   (gradient.jl:140)
  HIPCHK(hipMemsetAsync(b->d_lossacc, 0, sizeof(double) * b->G, b->stream));
  HIPCHK(hipMemsetAsync(b->d_Gsum, 0, sizeof(double) * b->G, b->stream));
  if (b->wants_Gacc()) HIPCHK(hipMemsetAsync(b->d_Gacc, 0, (size_t)b->ntotd * sizeof(double), b->stream));
  b->grad_field_valid = b->wants_Gacc();
  if (b->law_kind >= ODINN_LAW_NN_Y) {
    CHK(ensure_theta_scratch(b, b->ntiles));
    HIPCHK(hipMemsetAsync(b->d_dth, 0, sizeof(double) * b->G * b->P, b->stream));
  }
  CHK(dhdt_forward(b));  // LossDhdt: loss term and the coefficients of its cotangent fields (gradient.jl:170-188)
  CHK(agg_tables(b, true));
  CHK(avgv_forward(b, true));  // LossAvgV: loss term, dL/dH of its stops, dL/dtheta
  if (n_quadrature == 0) CHK(vreg_forward(b, true, true, 0, nullptr, nullptr));  // (ContinuousAdjoint: once its nodes exist)
  return ODINN_OK;
}
static int grad_finish(odinn_batch* b, int k, int P, double const_loss, double* loss, double* dtheta);

static int loss_grad_impl(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                          const double* mb_times, const odinn_solver_opts* opts, double* loss, double* dtheta,
                          odinn_solve_stats* stats);
int odinn_loss_grad(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                    const double* mb_times, const odinn_solver_opts* opts, double* loss, double* dtheta,
                    odinn_solve_stats* stats) {
  if (!b || !tstops || !loss || !dtheta) return fail(ODINN_ERR_ARG, "null argument");
  return with_law_table(b, [&] { return loss_grad_impl(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, loss, dtheta, stats); });
}
static int loss_grad_impl(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                          const double* mb_times, const odinn_solver_opts* opts, double* loss, double* dtheta,
                          odinn_solve_stats* stats) {
  CHK(grad_prepare(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, stats));
  // ---- reverse loop: gradient.jl:191-253 -------------------------------------------------
  // Row j of the per-glacier tables = the glacier's own j-th stop (t = result.t of THAT glacier, gradient.jl:71-73): a
  // glacier with fewer stops than the longest table is idle (dt = 0: lambda = 0 passes through) until its last stop comes up.
  if (b->nhid > 0)
    return fail(ODINN_ERR_ARG, "When using the DiscreteAdjoint the tstops of the MB callback must all be included in the "
                               "tstops from the results (gradient.jl:131)");
  const int k = b->K();
  const size_t fb = (size_t)b->ntot * sizeof(double);
  const Pools Psw = b->pools(true);
  const LawDev L = b->lawdev();
  int cur = 0;
  double const_loss = 0.0;
  // LossH with an A-type law: the loss and dL/dA partials of all steps are reduced by ONE kernel after
  // the loop (same summation order) instead of two dependent ~5 us launches per reverse step
  const bool defer = b->loss_kind == ODINN_LOSS_H && b->law_kind < ODINN_LAW_NN_Y;
  const long long pstride = 4LL * b->ntiles;
  if (defer && (size_t)k * pstride > b->partsteps_cap) {
    dfree(b->d_partsteps);
    CHK(dalloc(&b->d_partsteps, (size_t)k * pstride));
    b->partsteps_cap = (size_t)k * pstride;
  }
  InterpAsyncScope ia_scope{b};
  CHK(interp_async_enable(b, b->loss_kind != ODINN_LOSS_H, 3));
  for (int j = k - 1; j >= 1; --j) {
    double* lam = b->d_lam[cur];
    double* lam_new = b->d_lam[1 - cur];
    const double* Hj = b->d_snaps + (size_t)j * b->ntot;
    if (b->any_mb) {  // :201-207
      bool any = false;
      for (int g = 0; g < b->G; ++g) any = any || b->mbf_res[(size_t)j * b->G + g] != 0;
      if (any)
        launch_mb_vjp(b->ntiles, b->stream, Psw, b->d_premb, b->d_mb0, b->any_sref ? b->d_Sref : nullptr, lam, lam, 1, 0,
                      b->d_mbf_res + (size_t)j * b->G, b->d_mbs_res + (size_t)j * b->G, b->ntot);
    }
    AdjArgs A{};
    A.H = Hj; A.lam = lam; A.out = lam_new; A.Href = b->d_Href; A.mask = b->d_mask; A.h_log_eps = b->h_log_eps;
    A.dts = b->d_dts + (size_t)j * b->G; A.ws = b->d_ws + (size_t)j * b->G;
    A.refslot = b->d_refslot + (size_t)j * b->G; A.ntot = b->ntot;
    Pools Pj = Psw;
    if (defer) Pj.part = b->d_partsteps + (size_t)j * pstride;
    launch_vjp_H(b, 1, b->ntiles, Pj, L, A, 0);  // :235-242
    if (!defer) launch_sum_part(b->G, b->stream, Psw, 1, b->d_lossacc, 1, 0);
    if (b->dhdt_on()) {  // dl/dH of the time-aggregated loss at this stop (:212-215), before the theta-VJP uses lambda_{j-1}
      bool any = false;
      for (int g = 0; g < b->G; ++g) any = any || b->dh_i0_h[g] == j || b->dh_i1_h[g] == j;
      if (any) launch_dhdt_cot(b->ntiles, b->stream, Psw, lam_new, b->d_snaps, b->d_dh_i0, b->d_dh_i1, b->d_dh_coef, j, b->ntot);
    }
    if (b->agg_slot_h[j] >= 0)  // LossAvgV: dL/dH of this stop (:212-215)
      launch_axpy(b->ntot, b->stream, 1.0, b->d_aggH + (size_t)b->agg_slot_h[j] * b->ntot, lam_new, lam_new);
    if (b->loss_kind != ODINN_LOSS_H) {  // backward_loss(::LossV): dl/dH into lambda_{j-1}, dl/dtheta into dtheta
      double c = 0.0;
      CHK(launch_lossV(b, j, Hj, lam_new, true, &c));
      const_loss += c;
    }
    CHK(theta_vjp_launch(b, Hj, lam_new, b->d_dts + (size_t)j * b->G, -1, true, defer ? Pj.part : nullptr));  // :245-249
    cur = 1 - cur;
  }
  if (b->interp_async) CHK(interp_async_join(b));
  if (defer && k > 1) {
    launch_sum_part_steps(b->G, b->stream, Psw, b->d_partsteps, pstride, k - 1, 1, 1, b->d_lossacc);
    launch_sum_part_steps(b->G, b->stream, Psw, b->d_partsteps, pstride, k - 1, 1, 2, b->d_Gsum);
  }
  HIPCHK(hipGetLastError());
  if (cur != 0) HIPCHK(hipMemcpyAsync(b->d_lam[0], b->d_lam[cur], fb, hipMemcpyDeviceToDevice, b->stream));
  return grad_finish(b, k, P, const_loss, loss, dtheta);
}

// ---- aggregate over the batch's glaciers (Model.jl:208-224) -----------------------------
static int grad_finish(odinn_batch* b, int k, int P, double const_loss, double* loss, double* dtheta) {
  std::vector<double> lossg(b->G), Gs(b->G);
  HIPCHK(hipMemcpyAsync(lossg.data(), b->d_lossacc, sizeof(double) * b->G, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipMemcpyAsync(Gs.data(), b->d_Gsum, sizeof(double) * b->G, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  b->last_loss_g = lossg;
  b->last_G_g = Gs;
  for (int g = 0; g < b->G; ++g) {  // data-only LossV terms belong to their glacier
    for (int j = 1; j < k; ++j) {
      const size_t q = (size_t)j * b->G + g;
      if (b->loss_kind != ODINN_LOSS_H && b->wv_h[q] != 0.0)
        b->last_loss_g[g] += b->wv_h[q] * b->vsc_h[q] * b->v_const(g, b->vslot_h[q]);
    }
  }
  double Ltot = const_loss;
  for (int g = 0; g < b->G; ++g) Ltot += lossg[g];
  *loss = Ltot;
  for (int q = 0; q < P; ++q) dtheta[q] = 0.0;
  if (b->law_kind == ODINN_LAW_CONST_A) {
    for (int g = 0; g < b->G; ++g) dtheta[0] += Gs[g];
  } else if (b->law_kind == ODINN_LAW_NN_A_SCALAR) {
    std::vector<double> dA(b->P);
    for (int g = 0; g < b->G; ++g) {
      h_mlp(b->mlp, b->theta.data(), &b->descs[g].T, dA.data());
      for (int q = 0; q < P; ++q) dtheta[q] = std::fma(dA[q], Gs[g], dtheta[q]);
    }
  } else if (b->law_kind == ODINN_LAW_NN_A_GRIDDED) {
    CHK(gridded_law_grad(b, 0, b->ntotd, dtheta));
  } else {
    std::vector<double> dth((size_t)b->G * b->P);
    HIPCHK(hipMemcpy(dth.data(), b->d_dth, dth.size() * sizeof(double), hipMemcpyDeviceToHost));
    CHK(check_interp_bounds(b));
    for (int g = 0; g < b->G; ++g)
      for (int q = 0; q < P; ++q) dtheta[q] += dth[(size_t)g * b->P + q];
  }
  return ODINN_OK;
}

// Gauss-Legendre nodes (ascending) and weights on [-1, 1] by Newton iteration on P_n.
static void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0); w.assign(n, 0.0);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double z = std::cos(pi * (i + 0.75) / (n + 0.5)), pp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (std::fabs(z - z1) <= 1e-16 * std::fabs(z)) break;
    }
    {  // derivative at the converged root
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
    }
    x[i] = -z; x[n - 1 - i] = z;
    w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
  if (n % 2 == 1) x[n / 2] = 0.0;
}

// SIA2D_grad_batch! with ContinuousAdjoint(VJP_method = DiscreteVJP()) (gradient.jl:276-539).
// The reverse ODE dlam/dtau = J_H(H_itp(-tau))^T lam runs on the same device-side RDPK3Sp35 + PID
// machinery as the forward solve (one k_adj_stage per stage); its stops are the snapshot times
// (loss and mass-balance callbacks) and the Gauss-Legendre nodes (theta-VJP quadrature).
static int loss_grad_continuous_impl(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                                     const double* mb_times, const odinn_solver_opts* opts, const odinn_adjoint_opts* aopts,
                                     double* loss, double* dtheta, odinn_solve_stats* stats, odinn_solve_stats* stats_rev);
int odinn_loss_grad_continuous(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                               const double* mb_times, const odinn_solver_opts* opts, const odinn_adjoint_opts* aopts,
                               double* loss, double* dtheta, odinn_solve_stats* stats, odinn_solve_stats* stats_rev) {
  if (!b || !tstops || !loss || !dtheta) return fail(ODINN_ERR_ARG, "null argument");
  return with_law_table(b, [&] {
    return loss_grad_continuous_impl(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, aopts, loss, dtheta, stats, stats_rev);
  });
}
static int loss_grad_continuous_impl(odinn_batch* b, const double* theta, int P, int n_stops, const double* tstops, int n_mb,
                                     const double* mb_times, const odinn_solver_opts* opts, const odinn_adjoint_opts* aopts,
                                     double* loss, double* dtheta, odinn_solve_stats* stats, odinn_solve_stats* stats_rev) {
  const bool useV = b->loss_kind != ODINN_LOSS_H;
  odinn_adjoint_opts ao{1e-8, 1e-8, 1.0 / 12.0, 200, 0, 1000000};  // AdjointTypes.jl:58-67
  if (aopts) ao = *aopts;
  if (ao.reltol <= 0) ao.reltol = 1e-8;
  if (ao.abstol <= 0) ao.abstol = 1e-8;
  if (ao.n_quadrature <= 0) ao.n_quadrature = 200;
  if (ao.maxiters <= 0) ao.maxiters = 1000000;
  static const bool prof = std::getenv("ODINN_PROFILE_HOST") != nullptr;  // phase times on stderr (synchronises between phases)
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tq0 = prof ? now() : 0.0;
  double tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tq5 = 0;
  CHK(grad_prepare(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, stats, ao.n_quadrature));
  if (prof) { HIPCHK(hipStreamSynchronize(b->stream)); tq1 = now(); }
  double const_loss = 0.0;
  CHK(do_loss(b, &const_loss));  // forward loss over the snapshots -> d_lossacc
  if (prof) { HIPCHK(hipStreamSynchronize(b->stream)); tq2 = now(); }
  const int k = b->K(), G = b->G;
  const double t0 = tstops[0], t1 = tstops[n_stops - 1];
  // ---- reverse stop tables, per glacier: tau = -t ascending over the glacier's own snapshots and the quadrature nodes (:457) ----
  std::vector<double>& gx = b->gl_x;
  std::vector<double>& gw = b->gl_w;
  if ((int)gx.size() != ao.n_quadrature) gauss_legendre(ao.n_quadrature, gx, gw);  // (kept: the same rule in every call of an inversion)
  std::vector<double> qt(ao.n_quadrature), qwt(ao.n_quadrature);
  for (int i = 0; i < ao.n_quadrature; ++i) {  // GaussQuadrature, :560-566
    qt[i] = (t0 + t1) / 2.0 + gx[i] * (t1 - t0) / 2.0;
    qwt[i] = (t1 - t0) / 2.0 * gw[i];
  }
  if (b->vreg_on())  // VelocityRegularization: dL/dH at its stops, dL/dtheta by the quadrature (its loss is in do_loss)
    CHK(vreg_forward(b, true, false, ao.n_quadrature, qt.data(), qwt.data()));
  struct Stop { double tau; int snap; double qw; int hid, mbs; };  // hid > 0: mass-balance-only stop (hidden snapshot slot + 1)
  std::vector<std::vector<Stop>> stg(G);
  int nr = 0;
  for (int g = 0; g < G; ++g) {
    std::vector<Stop>& st = stg[g];
    const std::vector<double>& tsg = b->ts_g[g];
    st.reserve(tsg.size() + ao.n_quadrature);
    for (int j = 0; j < (int)tsg.size(); ++j) st.push_back({-tsg[j], j, 0.0, 0, 0});
    for (int i = 0; i < ao.n_quadrature; ++i) st.push_back({-qt[i], -1, qwt[i], 0, 0});
    // mass-balance times that are not result stops: the reverse PeriodicCallback (gradient.jl:426-432) stops there as well
    for (int i = 0; i < b->it_n[g]; ++i) {
      const size_t q = (size_t)i * G + g;
      if (b->it_snap[q] >= b->kmax) st.push_back({-b->it_t[q], -1, 0.0, b->it_snap[q] + 1, b->it_mbs[q]});
    }
    std::stable_sort(st.begin(), st.end(), [](const Stop& a, const Stop& c) { return a.tau < c.tau; });
    for (size_t i = 1; i < st.size(); ++i)
      if (!(st[i].tau > st[i - 1].tau)) return fail(ODINN_ERR_ARG, "a quadrature node coincides with a snapshot or mass-balance time");
    nr = std::max(nr, (int)st.size());
  }
  const size_t nrG = (size_t)nr * G;
  std::vector<double> h_tau(nrG, -t0), h_qw(nrG, 0.0), h_tsnap((size_t)k * G, t1);
  std::vector<int> h_snap(nrG, -1), h_mbf(nrG, 0), h_mbs(nrG, 0), h_hid(nrG, 0), h_nr(G), h_ksn(G);
  for (int g = 0; g < G; ++g) {
    const std::vector<Stop>& st = stg[g];
    h_nr[g] = (int)st.size(); h_ksn[g] = b->nres(g);
    for (int j = 0; j < b->nres(g); ++j) h_tsnap[(size_t)j * G + g] = b->ts_g[g][j];
    for (int i = 0; i < (int)st.size(); ++i) {
      const size_t q = (size_t)i * G + g;
      h_tau[q] = st[i].tau; h_qw[q] = st[i].qw; h_snap[q] = st[i].snap;
      if (st[i].snap >= 1 && b->any_mb && b->mbf_res[(size_t)st[i].snap * G + g]) { h_mbf[q] = 1; h_mbs[q] = b->mbs_res[(size_t)st[i].snap * G + g]; }
      if (st[i].hid > 0) { h_mbf[q] = 1; h_mbs[q] = st[i].mbs; h_hid[q] = st[i].hid; }
    }
  }
  if ((int)nrG > b->rev_cap) {
    dfree(b->d_rtau); dfree(b->d_rqw); dfree(b->d_rsnap); dfree(b->d_rmbf); dfree(b->d_rmbs); dfree(b->d_rhid);
    CHK(dalloc(&b->d_rtau, nrG)); CHK(dalloc(&b->d_rqw, nrG)); CHK(dalloc(&b->d_rsnap, nrG));
    CHK(dalloc(&b->d_rmbf, nrG)); CHK(dalloc(&b->d_rmbs, nrG)); CHK(dalloc(&b->d_rhid, nrG));
    b->rev_cap = (int)nrG;
    b->rev_host.clear();
  }
  if (k * G > b->tsnap_cap) { dfree(b->d_tsnap); CHK(dalloc(&b->d_tsnap, (size_t)k * G)); b->tsnap_cap = k * G; b->rev_host.clear(); }
  if (!b->d_adj) { CHK(dalloc(&b->d_adj, G)); CHK(dalloc(&b->d_qw, G)); }
  if (!b->d_nr) { CHK(dalloc(&b->d_nr, (size_t)G)); CHK(dalloc(&b->d_ksn, (size_t)G)); CHK(dalloc(&b->d_lastseg, (size_t)G)); CHK(dalloc(&b->d_zerow, (size_t)G)); b->rev_host.clear(); }
  std::vector<int> h_lastseg(G);
  for (int g = 0; g < G; ++g) h_lastseg[g] = b->nres(g) - 1;
  // The tables depend on the stops, the mass-balance times and the quadrature rule only: inside an inversion every gradient call
  // brings the same ones.  A byte image of what was uploaded last is kept; an identical image skips the ten copies and the
  // synchronisation behind them (4 alpine glaciers: 0.87 -> 0.3 ms of set-up per gradient).
  std::vector<unsigned char> img;
  {
    auto put = [&](const void* p_, size_t n) { const unsigned char* c = static_cast<const unsigned char*>(p_); img.insert(img.end(), c, c + n); };
    const long long dims[4] = {(long long)nr, (long long)G, (long long)k, (long long)b->nhid};
    put(dims, sizeof(dims));
    put(h_tau.data(), nrG * sizeof(double)); put(h_qw.data(), nrG * sizeof(double)); put(h_tsnap.data(), (size_t)k * G * sizeof(double));
    put(h_snap.data(), nrG * sizeof(int)); put(h_mbf.data(), nrG * sizeof(int)); put(h_mbs.data(), nrG * sizeof(int));
    put(h_hid.data(), nrG * sizeof(int)); put(h_nr.data(), (size_t)G * sizeof(int)); put(h_ksn.data(), (size_t)G * sizeof(int));
    put(h_lastseg.data(), (size_t)G * sizeof(int));
  }
  const bool same_tables = !b->rev_host.empty() && img == b->rev_host;
  if (!same_tables) {
    HIPCHK(hipMemcpyAsync(b->d_rtau, h_tau.data(), nrG * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rqw, h_qw.data(), nrG * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rsnap, h_snap.data(), nrG * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rmbf, h_mbf.data(), nrG * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rmbs, h_mbs.data(), nrG * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rhid, h_hid.data(), nrG * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_tsnap, h_tsnap.data(), (size_t)k * G * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_nr, h_nr.data(), (size_t)G * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_ksn, h_ksn.data(), (size_t)G * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_lastseg, h_lastseg.data(), (size_t)G * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemsetAsync(b->d_zerow, 0, (size_t)G * sizeof(double), b->stream));
  }
  HIPCHK(hipMemsetAsync(b->d_qw, 0, sizeof(double) * G, b->stream));
  // velocity loss: per (reverse stop, glacier) the bracketing reference maps of a quadrature node
  // (interpolate((tV_ref,), V_ref, Gridded(Linear())), or the single map; gradient.jl:291-301)
  std::vector<int> h_vA, h_vB;
  std::vector<double> h_vs;
  if (useV) {
    h_vA.assign((size_t)nr * G, -1); h_vB.assign((size_t)nr * G, 0); h_vs.assign((size_t)nr * G, 0.0);
    for (int g = 0; g < G; ++g)
      for (int i = 0; i < (int)stg[g].size(); ++i) {
        if (stg[g][i].snap >= 0) continue;
        const double tn = -stg[g][i].tau;
        const std::vector<double>& tv = b->t_vref[g];
        const size_t q = (size_t)i * G + g;
        if (tv.empty()) continue;
        if (tv.size() == 1) { h_vA[q] = 0; h_vB[q] = 0; continue; }
        if (tn < tv.front() || tn > tv.back())
          return fail(ODINN_ERR_ARG, "glacier %d: the velocity data must span tspan for the continuous adjoint "
                                     "(linear interpolation in time does not extrapolate)", g);
        size_t m = 0;
        while (m + 2 < tv.size() && tn > tv[m + 1]) ++m;
        h_vA[q] = (int)m; h_vB[q] = (int)m + 1;
        h_vs[q] = (tn - tv[m]) / (tv[m + 1] - tv[m]);
      }
    const size_t n = (size_t)nr * G;
    if (n > b->rv_cap) {
      dfree(b->d_rvA); dfree(b->d_rvB); dfree(b->d_rvs);
      CHK(dalloc(&b->d_rvA, n)); CHK(dalloc(&b->d_rvB, n)); CHK(dalloc(&b->d_rvs, n));
      b->rv_cap = n;
    }
    if (!b->d_Vq) { CHK(dalloc(&b->d_Vq, 3 * (size_t)b->ntot)); CHK(dalloc(&b->d_vscq, G)); CHK(dalloc(&b->d_wvq, G)); CHK(dalloc(&b->d_zeroslot, G)); }
    HIPCHK(hipMemcpyAsync(b->d_rvA, h_vA.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rvB, h_vB.data(), n * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_rvs, h_vs.data(), n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  }
  if (!same_tables || useV) HIPCHK(hipStreamSynchronize(b->stream));  // the staging vectors die with this scope
  if (!same_tables) b->rev_host.swap(img);

  const Pools Pl = b->pools(true);
  const LawDev L = b->lawdev();
  const int lm = b->lm();
  // ---- lambda(t1): loss term of the last snapshot, then the mass-balance VJP (:441-446, :431) ----
  AdjPostArgs AP{};
  AP.adj = b->d_adj; AP.snaps = b->d_snaps; AP.premb = b->d_premb; AP.ntot = b->ntot; AP.mb0 = b->d_mb0;
  AP.Sref = b->any_sref ? b->d_Sref : nullptr; AP.Href = b->d_Href; AP.mask = b->d_mask; AP.ws = b->d_ws;
  AP.h_log_eps = b->h_log_eps;
  // the theta-VJP of the A-type laws interpolates H at the quadrature node in its tile loader; the per-node MLP laws and
  // the velocity terms read it from d_tmpA, which the post-step then materialises
  // closed-form laws: the theta-part of a velocity loss takes one pass per quadrature node
  // (k_surfV_theta_node, which interpolates H itself) instead of the interpolate / scale / pull-back / reduce sequence
  // (ODINN_VQ_ONEPASS=0 selects that sequence); d_tmpA then is only needed at the snapshot stops
  bool vq_onepass = useV && lm <= 1 && !b->vel_nn();
  vq_onepass = vq_onepass && sched_val(b->sched.vq_onepass, "ODINN_VQ_ONEPASS") != 0;
  const bool theta_itp = (!useV || vq_onepass) && b->law_kind < ODINN_LAW_NN_Y;
  AP.refslot = b->d_refslot; AP.G = G; AP.loss_first = 1; AP.Hq = (theta_itp && !useV) ? nullptr : b->d_tmpA;
  AP.hq_snap_only = vq_onepass ? 1 : 0;
  if (b->dhdt_on()) { AP.dh_i0 = b->d_dh_i0; AP.dh_i1 = b->d_dh_i1; AP.dh_coef = b->d_dh_coef; }
  if (b->agg_nslots > 0) { AP.agg_slot = b->d_agg_slot; AP.aggH = b->d_aggH; }
  // (row 0 of the reverse tables is every glacier's last snapshot, tau_0 = -t1, with its mass-balance flag / slot)
  launch_adj_begin(G, b->stream, Pl, b->d_adj, b->d_ksn, -t1, b->d_rmbf, b->d_rmbs);
  // H(t1) of every glacier (its own last result slot) gathered into d_E, free until the second RHS of the initial-step heuristic
  launch_lerp_g(b->ntiles, b->stream, Pl, b->d_snaps, b->ntot, b->d_lastseg, b->d_zerow, b->d_E);
  // loss term of a velocity-data snapshot: lam += wV dl_V/dH(H_j)  (backward_loss(::LossV), Losses.jl:338-390)
  VArgs VS{};
  if (useV) {
    VS.Vabs = b->d_Vabs; VS.Vxr = b->d_Vxr; VS.Vyr = b->d_Vyr; VS.wv = b->d_wv; VS.scale = b->d_vsc; VS.refslot = b->d_vslot;
    VS.ntot = b->ntot; VS.component_abs = b->v_abs; VS.log_eps = b->v_abs ? b->v_log_eps : 0.0; VS.Gacc = nullptr; VS.adj = b->d_adj; VS.G = G;
    VS.finv = 1.0 / b->fV;  // (U law: H-part only here, the theta-part of the loss is integrated at the quadrature nodes)
    VS.H = b->d_E; VS.out = b->d_lam[0];
    launch_surfV_vjp(lm, 1, b->ntiles, b->stream, Pl, L, VS, 0);  // at t1 the losses come before the MB VJP
    VS.H = b->d_tmpA;
  }
  launch_adj_poststep(b->ntiles, b->stream, Pl, AP, b->d_lam[0], b->d_lam[1]);
  AP.loss_first = 0;
  // theta-part of the velocity loss at the quadrature nodes (Delta-t = 1; LossHV: x scaling)
  VItpArgs VI{};
  VArgs VQ{};
  if (useV) {
    VI.Vabs = b->d_Vabs; VI.Vxr = b->d_Vxr; VI.Vyr = b->d_Vyr; VI.ntot = b->ntot; VI.slotA = b->d_rvA; VI.slotB = b->d_rvB;
    VI.sw = b->d_rvs; VI.G = G; VI.adj = b->d_adj; VI.Vq = b->d_Vq;
    VQ.H = b->d_tmpA; VQ.out = b->d_tmpB; VQ.Vabs = b->d_Vq; VQ.Vxr = b->d_Vq + b->ntot; VQ.Vyr = b->d_Vq + 2 * b->ntot;
    VQ.wv = b->d_wvq; VQ.scale = b->d_vscq; VQ.refslot = b->d_zeroslot; VQ.ntot = b->ntot; VQ.component_abs = b->v_abs; VQ.log_eps = b->v_abs ? b->v_log_eps : 0.0;
    VQ.Gacc = b->wants_Gacc() ? b->d_Gacc : nullptr;
    VQ.finv = 1.0 / b->fV;
    CHK(vel_theta_args(b, VQ, -1));  // (Y law with `:Linear`: the node arrays are zeroed again before every launch below)
    HIPCHK(hipMemsetAsync(b->d_tmpB, 0, (size_t)b->ntot * sizeof(double), b->stream));
  }
  const double wq = b->loss_kind == ODINN_LOSS_HV ? b->hv_scaling : 1.0;
  // ---- initial step (ode_determine_initdt on the reverse problem) ----
  const double tspan = t1 - t0;
  long long nrhs_extra = 0;
  {
    AdjArgs A{};
    A.H = b->d_E; A.lam = b->d_lam[0]; A.out = b->d_S2; A.ntot = b->ntot;
    launch_vjp_H(b, 0, b->ntiles, Pl, L, A, 0);  // f0
    launch_initdt_norms(b->ntiles, b->stream, Pl, b->d_lam[0], b->d_S2, nullptr, ao.abstol, ao.reltol);
    launch_initdt_ctrl(G, b->stream, Pl, 0, tspan, ao.dtmax, b->d_dt0);
    launch_adj_itp(G, b->stream, Pl, b->d_adj, b->d_tsnap, b->d_ksn, 1);
    launch_axpy_g(b->ntiles, b->stream, Pl, b->d_S2, b->d_lam[0], b->d_lam[1]);
    A.snaps = b->d_snaps; A.adj = b->d_adj; A.lam = b->d_lam[1]; A.out = b->d_E;
    launch_vjp_H(b, 0, b->ntiles, Pl, L, A, 0);  // f1 at tau0 + dt0
    launch_initdt_norms(b->ntiles, b->stream, Pl, b->d_lam[0], b->d_S2, b->d_E, ao.abstol, ao.reltol);
    launch_initdt_ctrl(G, b->stream, Pl, 1, tspan, ao.dtmax, b->d_dt0);
    nrhs_extra = 2;
  }
  launch_begin(G, b->stream, Pl, b->d_rtau, ao.dtmax, 0.0);
  launch_adj_itp(G, b->stream, Pl, b->d_adj, b->d_tsnap, b->d_ksn, 0);
  int nact = G;
  HIPCHK(hipMemcpyAsync(b->d_nactive, &nact, sizeof(int), hipMemcpyHostToDevice, b->stream));
  CtrlArgs C{};
  C.tstops = b->d_rtau; C.nstops = b->d_nr; C.G = G; C.mb_flag = b->d_rmbf; C.mb_slot = b->d_rmbs; C.dtmax = ao.dtmax;
  C.adaptive = 1; C.fixed_dt = 0.0; C.n_active = b->d_nactive; C.errpart = b->d_part; C.stride = 4; C.fused = 0;
  C.adj = b->d_adj; C.tsnap = b->d_tsnap; C.stop_snap = b->d_rsnap; C.stop_qw = b->d_rqw; C.qw_out = b->d_qw;
  C.stop_hid = b->nhid > 0 ? b->d_rhid : nullptr;
  C.nrows = nr; C.t_last = -t0;
  // ODINN_TRACE_STEPS=n: the first n attempts of glacier 0's reverse solve (tau, dt, error estimate, +-step factor) on stderr
  struct TraceBuf { double* p = nullptr; ~TraceBuf() { if (p) (void)hipFree(p); } } trace_buf;
  double*& d_trace = trace_buf.p;
  const int ntrace = std::getenv("ODINN_TRACE_STEPS") ? std::atoi(std::getenv("ODINN_TRACE_STEPS")) : 0;
  if (ntrace > 0) {
    HIPCHK(hipMalloc(&d_trace, (size_t)ntrace * 4 * sizeof(double)));
    HIPCHK(hipMemsetAsync(d_trace, 0, (size_t)ntrace * 4 * sizeof(double), b->stream));
    C.trace = d_trace; C.trace_cap = ntrace;
  }
  AdjStageArgs SA{};
  SA.snaps = b->d_snaps; SA.ntot = b->ntot; SA.adj = b->d_adj; SA.S2 = b->d_S2; SA.S3 = b->d_S3; SA.E = b->d_E;
  SA.abstol = ao.abstol; SA.reltol = ao.reltol;
  const bool acc_inplace = b->law_kind < ODINN_LAW_NN_Y;
  if (acc_inplace) {
    const size_t need = 4 * (size_t)b->ntiles;
    if (need > b->partsteps_cap) {
      dfree(b->d_partsteps);
      CHK(dalloc(&b->d_partsteps, need));
      b->partsteps_cap = need;
    }
    HIPCHK(hipMemsetAsync(b->d_partsteps, 0, need * sizeof(double), b->stream));
  }
  // integer-power law + DiscreteVJP (any loss: the velocity terms are separate launches that follow each glacier's
  // current lambda buffer): the five stages of a reverse step run as ONE kernel
  // (sia2d_adj_fused.hpp) -- measured faster at every batch size, 4 alpine glaciers included; ODINN_ADJ_FUSED=0
  // selects the five k_adj_stage launches
  // (round 4: also the Y law through its table where that is the integer-power law with Y(Hbar) in A's place -- n_H = n_gradS = 3,
  //  no sliding, GDev::yt_fast on every glacier; its theta-integrand stays with theta_vjp_launch)
  bool ytab_rev = b->lm_kern() == LM_YTAB && b->ytab_ni == 1024;  // (the kernel's LDS copy of the table is laid out for that size)
  for (const GDev& r : b->gd) ytab_rev = ytab_rev && r.yt_fast;
  bool fused_rev = (lm == 0 || ytab_rev) && b->vjp_method == ODINN_VJP_DISCRETE;
  fused_rev = fused_rev && sched_val(b->sched.adj_fused, "ODINN_ADJ_FUSED") != 0;
  const int rev_skip = sched_val(b->sched.adj_skip, "ODINN_ADJ_SKIP") == 0 ? 0 : 1;
  AdjFusedArgs FA{};
  int adj_rows = TRPT;  // rows per thread of the fused reverse step
  if (fused_rev) {
    FA.snaps = b->d_snaps; FA.ntot = b->ntot; FA.adj = b->d_adj; FA.lam0 = b->d_lam[0]; FA.lam1 = b->d_lam[1];
    FA.partF = b->d_partFt; FA.tilesF = b->d_tilesFt; FA.abstol = ao.abstol; FA.reltol = ao.reltol;
    if (ytab_rev) { FA.ytab = b->d_ytab; FA.ytab_over = b->d_ytab_over; FA.ytab_ni = b->ytab_ni; }
    C.errpart = b->d_partFt; C.stride = 1; C.fused = 3;
    // the two bracketing snapshots of every segment interleaved as {H_j, H_j+1 - H_j}: one 16-byte load per cell and
    // stage instead of two 8-byte ones (ODINN_ADJ_SEGS=0: read the snapshots themselves)
    const int es = sched_val(b->sched.adj_segs, "ODINN_ADJ_SEGS");
    const size_t need = (size_t)(k - 1) * b->ntot;
    // (a second copy of the snapshots, twice their size: only while it takes less than half of what is free)
    bool fits = need <= b->segs_cap;
    if (!fits && es != 0) {
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      fits = need * sizeof(double2) <= free_b / 2;
    }
    if (es != 0 && fits) {
      if (need > b->segs_cap) {
        if (b->d_segs) (void)hipFree(b->d_segs);
        b->d_segs = nullptr; b->segs_cap = 0;
        HIPCHK(hipMalloc(&b->d_segs, need * sizeof(double2)));
        b->segs_cap = need;
      }
      launch_seg_pairs(b->ntot, k - 1, b->stream, b->d_snaps, b->d_segs);
      FA.segs = b->d_segs;
    }
    // small batches: the 4-rows-per-thread instantiation (54 x 22 output tiles).  A launch lasts about (tiles on the busiest
    // CU) x (rows per thread) while at most two 7-row workgroups share a CU: 4 rows do more halo work per cell but quantise
    // finer (measured, continuous gradient, 7 -> 4 rows: 4 / 8 / 12 alpine glaciers 11.8 -> 9.5 / 12.0 -> 9.8 / 12.1 -> 9.9 ms,
    // 16 / 24: ties within 3 %, 32: 16.3 -> 13.4, 48 / 64: 7 rows win; 1 x 256^2 ... 512^2: 8.2 -> 6.6 ms, 768^2 and up: 7 rows
    // win -- the model's order every time).  ODINN_ADJ_ROWS=4|7 forces either
    if (FA.segs) {
      const int er = sched_val(b->sched.adj_rows, "ODINN_ADJ_ROWS");
      const long cu = b->n_cus();
      const bool model = b->ntilesFt <= 2 * cu && 4 * ((b->ntilesFv + cu - 1) / cu) < 7 * ((b->ntilesFt + cu - 1) / cu);
      // the smallest batches (4 alpine glaciers: 48 four-row tiles on 256 CUs): 2 rows per thread, 54 x 6 output tiles -- 2.7 x the
      // halo work per cell on 3.4 x as many CUs, half the serial row sweeps per stage; while the two-row tiles (about) fit the CUs
      // once (measured, ms per continuous gradient, 4 -> 2 rows: 4 alpine glaciers, 259 two-row tiles, 9.08 -> 7.66; 8 alpine,
      // 518 tiles, 9.13 -> 10.93; 1 x 512^2, 860 tiles, 14.8 -> 21.9)
      const bool model2 = 4 * b->ntilesFw <= 5 * cu;
      if (er == 2 || (er < 0 && model && model2)) {
        adj_rows = 2;
        FA.partF = b->d_partFw; FA.tilesF = b->d_tilesFw;
        C.errpart = b->d_partFw; C.fused = 7;
      } else if (er == 4 || er == 7 || er == 8 ? er == 4 : model) {
        adj_rows = 4;
        FA.partF = b->d_partFv; FA.tilesF = b->d_tilesFv;
        C.errpart = b->d_partFv; C.fused = 6;
      } else if (b->gd[0].use_Afield && b->d_tilesFu &&
                 (er == 8 || (er < 0 && 8 * ((b->ntilesFu + cu - 1) / cu) <= 7 * ((b->ntilesFt + cu - 1) / cu)))) {
        // gridded A: the register-cached instantiation (one workgroup per CU, 256 VGPRs) has room for the forward kernel's
        // 8 rows per thread -- 54 x 54 output tiles: 3 % less halo work and an exact fit of 1024^2 grids (19 x 19 tiles where
        // the 54 x 46 ones need 19 x 23 and overshoot by 3 %)
        adj_rows = 8;
        FA.partF = b->d_partFu; FA.tilesF = b->d_tilesFu;
        C.errpart = b->d_partFu; C.fused = 4;
      }
    }
    const int ntilesR = adj_rows == 2 ? b->ntilesFw : adj_rows == 4 ? b->ntilesFv : adj_rows == 8 ? b->ntilesFu : b->ntilesFt;
    // A-type laws without a dual-grid accumulator: the theta-VJP of a quadrature node is formed by stage 1 of the step that
    // follows the node (same lambda, same H_itp) instead of a launch of its own (ODINN_ADJ_THETA_FUSED=0: separate launches)
    const int et = sched_val(b->sched.adj_theta_fused, "ODINN_ADJ_THETA_FUSED");
    // (with a dual-grid accumulator -- gridded A -- the same stage also adds the node weights into d_Gacc: needs the
    //  interleaved snapshot pairs, whose kernel instantiations carry that variant)
    const bool gacc_fused = b->wants_Gacc() && FA.segs && b->gd[0].use_Afield;
    if (acc_inplace && (!b->wants_Gacc() || gacc_fused) && et != 0) {
      if (gacc_fused) FA.Gacc = b->d_Gacc;
      if ((size_t)ntilesR > b->partTh_cap) {
        dfree(b->d_partTh);
        CHK(dalloc(&b->d_partTh, (size_t)ntilesR));
        b->partTh_cap = (size_t)ntilesR;
      }
      HIPCHK(hipMemsetAsync(b->d_partTh, 0, (size_t)ntilesR * sizeof(double), b->stream));
      FA.th_part = b->d_partTh;
    }
  }
  const bool theta_fused = FA.th_part != nullptr;
  const int ntilesR_launch = adj_rows == 2 ? b->ntilesFw : adj_rows == 4 ? b->ntilesFv : adj_rows == 8 ? b->ntilesFu : b->ntilesFt;
  // Self-controlled reverse step (k_adj_fused_strip<..., SC>): the step kernel decides the previous attempt itself and does the
  // post-step of a stop -- ONE launch per reverse step instead of three dependent ones (fused step, k_controller,
  // k_adj_poststep).  Needs everything a step involves inside that kernel: thickness-type losses (no velocity launches), the
  // theta-VJP of the quadrature nodes in stage 1 (A-type laws), the interleaved snapshot pairs.  The decision is repeated by every
  // workgroup (~2 us), which pays while launch latency is a large part of a step: small and medium batches (measured rule
  // below); odinn_schedule.adj_sc / ODINN_ADJ_SC = 0 | 1 forces either.
  // Y law through its table with the sort-free `:Linear` contraction on lanes: stage 1 of the fused step emits the node pairs itself
  // (AdjFusedArgs::emitH) -- no k_vjp_theta launch, no memsets of the node arrays per step
  bool emit_fused = false;
  bool rsc = false;
  auto decide_rsc = [&] {
    rsc = fused_rev && (theta_fused || emit_fused) && !useV && FA.segs != nullptr;
    const int e = sched_val(b->sched.adj_sc, "ODINN_ADJ_SC");
    rsc = rsc && (e < 0 ? ntilesR_launch <= ODINN_ADJ_SC_MAX_TILES : e != 0);
  };
  auto setup_rsc = [&]() -> int {
    decide_rsc();
    if (rsc) {
      CHK(sc_buffers(b));
      if (!b->d_adj2) CHK(dalloc(&b->d_adj2, (size_t)G));
      FA.post = AP; FA.post.loss_first = 0; FA.post.Hq = nullptr;
    }
    return ODINN_OK;
  };
  // polls as in do_solve: one step per stop at least, then the controller's estimate of what is left
  if (!b->d_est) CHK(dalloc(&b->d_est, (size_t)G));
  C.est_steps = b->d_est;
  b->h_est.assign(G, 0);
  int chunk = std::max(2, std::min(256, nr & ~1));
  long long steps = 0;
  int p = 0;
  if (prof) { HIPCHK(hipStreamSynchronize(b->stream)); tq3 = now(); }
  int polls = 0;
  InterpAsyncScope ia_scope{b};
  // (one lane while the five stage launches keep the GPU busy; with the fused reverse step of the tabulated Y law the contractions
  //  are the longer chain again -- 8 x 512^2, ms per gradient for 1 / 2 / 3 / 4 lanes: 145 / 135 / 126 / 121)
  CHK(interp_async_enable(b, useV, (fused_rev && ytab_rev) ? 4 : 1));
  emit_fused = fused_rev && ytab_rev && !useV && b->interp_async && b->ia_select && b->ia_nact > 0 &&
               b->grad_interp == ODINN_GRAD_INTERP_LINEAR && sched_val(-1, "ODINN_ADJ_EMIT_FUSED") != 0;
  b->ia_emit_fused = emit_fused;
  CHK(setup_rsc());
  while (nact > 0) {
    for (int s_ = 0; s_ < chunk; ++s_) {
      double* a0 = b->d_lam[p];
      double* a1 = b->d_lam[1 - p];
      if (emit_fused) {  // the node arrays / maxima this launch's stage 1 emits into
        unsigned long long* mx = nullptr;
        CHK(interp_async_begin_fused(b, &FA.emitH, &FA.emitV, &mx));
        FA.emit_amax = mx; FA.emit_vmax = mx + G;
      }
      if (rsc) {
        // launch n reads state / AdjState / partials [n & 1], writes the other ones; it decides attempt n - 1
        const bool odd = (steps & 1) != 0;
        AdjFusedArgs FS = FA;
        FS.gin = odd ? b->d_gs2 : b->d_gs; FS.gout = odd ? b->d_gs : b->d_gs2;
        FS.adj_in = odd ? b->d_adj2 : b->d_adj; FS.adj_out = odd ? b->d_adj : b->d_adj2;
        FS.partF = odd ? b->d_part2 : FA.partF;
        FS.C = C; FS.C.next_cur = -1; FS.C.errpart = odd ? FA.partF : b->d_part2;
        launch_adj_fused_strip(ntilesR_launch, b->gd[0].use_Afield, rev_skip, adj_rows, b->stream, Pl, FS, 1);
        if (emit_fused) CHK(interp_async_contract(b, Pl));
        p = 1 - p;
        ++steps;
        continue;
      } else if (fused_rev) {
        // the whole step in one kernel: reads lam[cur], writes lam[1 - cur] per glacier; the controller flips cur
        // on acceptance (a rejected step is simply repeated from the untouched lam[cur])
        launch_adj_fused_strip(ntilesR_launch, b->gd[0].use_Afield, rev_skip, adj_rows, b->stream, Pl, FA);
        C.next_cur = -1;
      } else {
        // five stages ping-pong lam[p] -> lam[1-p] -> ... ; the step's result lands in lam[1-p]
        const double* src = a0;
        double* dst = a1;
        for (int stg = 1; stg <= 5; ++stg) {
          SA.src = src; SA.dst = dst;
          launch_adj_stage(b->lm_kern(), b->vjp_method, stg, b->ntiles, b->stream, Pl, L, SA);
          double* t_ = const_cast<double*>(src);
          src = dst;
          dst = t_;
        }
        C.next_cur = 1 - p;
      }
      launch_controller(G, b->stream, Pl, C);
      launch_adj_poststep(b->ntiles, b->stream, Pl, AP, b->d_lam[0], b->d_lam[1]);
      if (useV) {
        if (fused_rev) { VS.out = b->d_lam[0]; VS.out_alt = b->d_lam[1]; }  // per-glacier ping-pong buffers
        else VS.out = a1;
        launch_surfV_vjp(lm, 1, b->ntiles, b->stream, Pl, L, VS, 0);                       // snapshot stops
        if (vq_onepass) {                                                               // quadrature nodes
          // (with a dual-grid accumulator the unscaled node weights go through d_tmpB, idle on this path, and are added
          //  into d_Gacc once the glacier's scale is known)
          double* tnode = b->wants_Gacc() ? b->d_tmpB : nullptr;
          launch_surfV_theta_node(lm, b->ntiles, b->stream, Pl, VI, b->d_snaps, b->v_abs, b->v_abs ? b->v_log_eps : 0.0, tnode);
          launch_vq_finish(G, b->stream, Pl, b->d_adj, b->d_rvA, b->v_scale_loss, wq, b->d_Gsum, tnode ? b->d_wvq : nullptr);
          if (tnode) launch_gacc_axpy(b->ntiles, b->stream, Pl, b->d_wvq, tnode, b->d_Gacc);
        } else {
          launch_vref_itp(b->ntiles, b->stream, Pl, VI);
          launch_vref_scale(G, b->stream, Pl, b->d_adj, b->d_rvA, b->v_scale_loss, wq, b->d_vscq, b->d_wvq);
          if (b->vel_nn()) CHK(interp_prepare(b, -1, b->law_kind == ODINN_LAW_NN_U));  // (the node arrays are re-emitted at every node)
          launch_surfV_vjp(lm, 1, b->ntiles, b->stream, Pl, L, VQ, 0);
          if (b->vel_nn()) CHK(vel_theta_finish(b, -1, true, Pl));
          else launch_sum_part(G, b->stream, Pl, 3, b->d_Gsum, 1, 0);
        }
      }
      // quadrature node reached: dtheta += w * J_theta(H_itp(t))^T lam(t)  (:497-503); A-type laws add
      // onto per-tile running sums that are reduced once after the solve
      if (theta_fused) {
      } else if (emit_fused) {
        // (the node the controller just reported is emitted by stage 1 of the NEXT launch; this launch's emission -- the node
        //  reached by the step before -- is contracted now)
        CHK(interp_async_contract(b, Pl));
      } else if (fused_rev)
        CHK(theta_vjp_launch(b, b->d_tmpA, b->d_lam[0], b->d_qw, -1, true, acc_inplace ? b->d_partsteps : nullptr, acc_inplace,
                             b->d_lam[1], theta_itp ? b->d_snaps : nullptr, b->d_adj));
      else
        CHK(theta_vjp_launch(b, b->d_tmpA, a1, b->d_qw, -1, true, acc_inplace ? b->d_partsteps : nullptr, acc_inplace, nullptr,
                             theta_itp ? b->d_snaps : nullptr, b->d_adj));
      p = 1 - p;
      ++steps;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&nact, b->d_nactive, sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(b->h_est.data(), b->d_est, sizeof(int) * G, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    ++polls;
    if (steps >= ao.maxiters && nact > 0)
      return fail(ODINN_ERR_MAXITERS, "maxiters (%lld) reached in the reverse solve with %d glaciers active",
                  (long long)ao.maxiters, nact);
    int est = 0;
    for (int g = 0; g < G; ++g) est = std::max(est, b->h_est[g]);
    chunk = std::max(2, std::min(64, (est + 2 + 1) & ~1));
    if (ao.maxiters - steps < chunk) chunk = (int)std::max<long long>(2, (ao.maxiters - steps + 1) & ~1LL);
  }
  if (prof) tq4 = now();
  if (rsc && (steps & 1)) {  // the last launch wrote its state to the second arrays
    HIPCHK(hipMemcpyAsync(b->d_gs, b->d_gs2, sizeof(GState) * G, hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_adj, b->d_adj2, sizeof(AdjState) * G, hipMemcpyDeviceToDevice, b->stream));
  }
  std::vector<GState> gs(G);
  HIPCHK(hipMemcpyAsync(gs.data(), b->d_gs, sizeof(GState) * G, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (d_trace) {
    std::vector<double> tr((size_t)ntrace * 4);
    HIPCHK(hipMemcpy(tr.data(), d_trace, tr.size() * sizeof(double), hipMemcpyDeviceToHost));
    const long long n = std::min<long long>(ntrace, gs[0].naccept + gs[0].nreject);
    for (long long q = 0; q < n; ++q)
      std::fprintf(stderr, "[odinn reverse step %lld] tau %.17g dt %.17g EEst %.17g factor %.17g\n", q, tr[4 * q], tr[4 * q + 1], tr[4 * q + 2], tr[4 * q + 3]);
  }
  bool mixed = false;
  for (int g = 0; g < G; ++g) {
    if (gs[g].nonfinite) return fail(ODINN_ERR_NONFINITE, "non-finite error estimate in the reverse solve of glacier %d", g);
    if (gs[g].cur != gs[0].cur) mixed = true;
    if (stats_rev) {
      stats_rev[g].naccept = gs[g].naccept;
      stats_rev[g].nreject = gs[g].nreject;
      stats_rev[g].nrhs = 5 * (gs[g].naccept + gs[g].nreject) + nrhs_extra;
      stats_rev[g].t_final = -gs[g].t;
      stats_rev[g].dt_last = gs[g].dt;
    }
  }
  if (b->interp_async) CHK(interp_async_join(b));
  if (acc_inplace) launch_sum_part_steps(G, b->stream, Pl, b->d_partsteps, 4LL * b->ntiles, 0, 0, 2, b->d_Gsum);
  if (theta_fused) launch_sum_tilesFt(G, adj_rows, b->stream, Pl, b->d_partTh, b->d_Gsum);
  // lambda(t0) of every glacier -> d_lam[0] (glaciers finish in different ping-pong buffers)
  if (mixed || gs[0].cur != 0) {
    for (int g = 0; g < G; ++g)
      if (gs[g].cur != 0) {
        const GDev& r = b->gd[g];
        HIPCHK(hipMemcpyAsync(b->d_lam[0] + r.off, b->d_lam[1] + r.off, (size_t)r.nx * r.ny * sizeof(double),
                              hipMemcpyDeviceToDevice, b->stream));
      }
  }
  const int rc_fin = grad_finish(b, k, P, const_loss, loss, dtheta);
  if (prof) {
    tq5 = now();
    std::fprintf(stderr, "[odinn loss_grad_continuous] forward+prepare %.0f us, loss %.0f us, reverse setup %.0f us, reverse loop(%lld launches, "
                 "%d polls, rows %d, sc %d) %.0f us, finish %.0f us\n", tq1 - tq0, tq2 - tq1, tq3 - tq2, steps, polls, adj_rows, rsc ? 1 : 0,
                 tq4 - tq3, tq5 - tq4);
  }
  return rc_fin;
}

// TikhonovRegularization(operator = :laplacian): loss = sum_mask (lap a)^2, grad = VJP_lap(2 mask lap a)
// (Regularization.jl:92-126, 330-382) for one host field of any size (H0 on the primal grid for
// InitialThicknessRegularization, A on the dual grid for RheologyRegularization).
int odinn_tikhonov(odinn_batch* b, int nx, int ny, double dx, double dy, const double* a, const unsigned char* mask,
                   double* loss, double* grad) {
  if (!b || !a || !loss || !grad) return fail(ODINN_ERR_ARG, "null argument");
  if (nx < 3 || ny < 3 || !(dx > 0.0) || !(dy > 0.0)) return fail(ODINN_ERR_ARG, "bad grid %dx%d", nx, ny);
  CHK(use_dev(b));
  const size_t n = (size_t)nx * ny;
  const int nblk = ((nx + 63) / 64) * ((ny + NW - 1) / NW);
  if (n > b->reg_cap) {
    dfree(b->d_rega); dfree(b->d_regr); dfree(b->d_regg); dfree(b->d_regm);
    CHK(dalloc(&b->d_rega, n)); CHK(dalloc(&b->d_regr, n)); CHK(dalloc(&b->d_regg, n)); CHK(dalloc(&b->d_regm, n));
    b->reg_cap = n;
  }
  if ((size_t)nblk > b->regp_cap) { dfree(b->d_regp); CHK(dalloc(&b->d_regp, nblk)); b->regp_cap = nblk; }
  HIPCHK(hipMemcpyAsync(b->d_rega, a, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
  if (mask) HIPCHK(hipMemcpyAsync(b->d_regm, mask, n, hipMemcpyHostToDevice, b->stream));
  launch_tikhonov(b->stream, b->d_rega, mask ? b->d_regm : nullptr, b->d_regr, b->d_regg, b->d_regp, nx, ny, dx, dy);
  HIPCHK(hipGetLastError());
  std::vector<double> part(nblk);
  HIPCHK(hipMemcpyAsync(grad, b->d_regg, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipMemcpyAsync(part.data(), b->d_regp, nblk * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  double s_ = 0.0;
  for (int k = 0; k < nblk; ++k) s_ += part[k];  // fixed order
  *loss = s_;
  return ODINN_OK;
}

int odinn_get_grad_parts(odinn_batch* b, double* loss_per_glacier, double* G_per_glacier) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  if ((int)b->last_loss_g.size() != b->G) return fail(ODINN_ERR_STATE, "no odinn_loss_grad has been run");
  for (int g = 0; g < b->G; ++g) {
    if (loss_per_glacier) loss_per_glacier[g] = b->last_loss_g[g];
    if (G_per_glacier) G_per_glacier[g] = b->last_G_g[g];
  }
  return ODINN_OK;
}

int odinn_get_grad_field(odinn_batch* b, int g, double* dLdA_dual) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  if (!dLdA_dual) return fail(ODINN_ERR_ARG, "null field");
  if (!b->grad_field_valid) return fail(ODINN_ERR_STATE, "no gridded-A gradient available (set an A field and run odinn_loss_grad)");
  return down_field(b, g, b->d_Gacc, dLdA_dual, true);
}

int odinn_get_lambda0(odinn_batch* b, int g, double* lam0) {
  CHK(check_g(b, g)); CHK(use_dev(b));
  return down_field(b, g, b->d_lam[0], lam0);
}

// ---- multi-GPU: RCCL communicator behind the C ABI (SIA2D_grad!, gradient.jl:6-31) ----------------------------
}  // extern "C"
struct odinn_comm {
  ncclComm_t comm = nullptr;
  int device = 0, nranks = 1, rank = 0;
  hipStream_t stream = nullptr;
  double* d_buf = nullptr;
  size_t cap = 0;
};
// RCCL is resolved on the first odinn_comm_* call (dlopen), not at link time: single-GPU users load libodinn_hip.so
// without an RCCL installation.  Search order: $ODINN_RCCL_LIB, the dynamic loader's path (librccl.so.1, librccl.so),
// $ROCM_PATH/lib, /opt/rocm/lib.  <rccl/rccl.h> is used for its types and constants only.
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (optional: odinn_comm_rank falls back to the values of
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;   //  odinn_comm_init_rank when the library lacks them)
  std::string err;
};
static RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  std::vector<std::string> cand;
  if (const char* e = std::getenv("ODINN_RCCL_LIB")) cand.push_back(e);
  cand.push_back("librccl.so.1"); cand.push_back("librccl.so");
  if (const char* r = std::getenv("ROCM_PATH")) { cand.push_back(std::string(r) + "/lib/librccl.so.1"); cand.push_back(std::string(r) + "/lib/librccl.so"); }
  cand.push_back("/opt/rocm/lib/librccl.so.1"); cand.push_back("/opt/rocm/lib/librccl.so");
  for (const std::string& c : cand) {
    api.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
    api.err = dlerror() ? dlerror() : "";
  }
  if (!api.handle) return nullptr;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
  api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.handle, "ncclCommUserRank"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
    dlclose(api.handle);
    api.handle = nullptr;
    api.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString";
    return nullptr;
  }
  return &api;
}
#define RCCL_OR_FAIL(R)                                                                                                  \
  RcclApi* R = rccl();                                                                                                   \
  if (!R) return fail(ODINN_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: the odinn_comm_* entry points need it (set ODINN_RCCL_LIB or ROCM_PATH)")
#define NCCLCHK(x)                                                                                         \
  do {                                                                                                     \
    ncclResult_t r_ = (x);                                                                                 \
    if (r_ != ncclSuccess) return fail(ODINN_ERR_HIP, "%s failed: %s (%s:%d)", #x, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)
static int comm_buf(odinn_comm* c, size_t n) {
  if (n > c->cap) {
    if (c->d_buf) (void)hipFree(c->d_buf);
    c->d_buf = nullptr;
    HIPCHK(hipMalloc((void**)&c->d_buf, n * sizeof(double)));
    c->cap = n;
  }
  return ODINN_OK;
}
extern "C" {

int odinn_comm_get_unique_id(void* id_out) {
  static_assert(sizeof(ncclUniqueId) == ODINN_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id_out) return fail(ODINN_ERR_ARG, "null id");
  ncclUniqueId id;
  RCCL_OR_FAIL(R);
  NCCLCHK(R->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return ODINN_OK;
}

int odinn_comm_init_rank(int device, int nranks, int rank, const void* id, odinn_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ODINN_ERR_ARG, "bad communicator arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(ODINN_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(ODINN_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
  RCCL_OR_FAIL(R);
  HIPCHK(hipSetDevice(device));
  odinn_comm* c = new odinn_comm();
  c->device = device; c->nranks = nranks; c->rank = rank;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclResult_t r = R->CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) { delete c; return fail(ODINN_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString(r)); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { R->CommDestroy(c->comm); delete c; return fail(ODINN_ERR_HIP, "stream creation failed"); }
  *out = c;
  return ODINN_OK;
}

int odinn_comm_destroy(odinn_comm* c) {
  if (!c) return ODINN_OK;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl()) rccl()->CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_buf) (void)hipFree(c->d_buf);
  delete c;
  return ODINN_OK;
}

int odinn_comm_rank(const odinn_comm* c, int* rank, int* nranks) {
  if (!c) return fail(ODINN_ERR_ARG, "null communicator");
  int r = c->rank, n = c->nranks;
  // what RCCL itself says about the communicator (ncclCommUserRank / ncclCommCount), so that a caller reporting "N ranks"
  // reports the group the all-reduce really runs over
  RcclApi* R = rccl();
  if (R && c->comm && R->CommCount && R->CommUserRank) {
    NCCLCHK(R->CommCount(c->comm, &n));
    NCCLCHK(R->CommUserRank(c->comm, &r));
    if (n != c->nranks || r != c->rank)
      return fail(ODINN_ERR_STATE, "communicator reports rank %d of %d, it was created as rank %d of %d", r, n, c->rank, c->nranks);
  }
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  return ODINN_OK;
}

int odinn_comm_allreduce_sum_dev(odinn_comm* c, double* inout_dev, int n, void* hip_stream) {
  if (!c || !inout_dev || n < 0) return fail(ODINN_ERR_ARG, "bad all-reduce arguments");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
  NCCLCHK(rccl()->AllReduce(inout_dev, inout_dev, (size_t)n, ncclDouble, ncclSum, c->comm, st));
  return ODINN_OK;
}

static int comm_allreduce_host(odinn_comm* c, double* inout, int n, hipStream_t st) {
  CHK(comm_buf(c, (size_t)n));
  HIPCHK(hipMemcpyAsync(c->d_buf, inout, sizeof(double) * n, hipMemcpyHostToDevice, st));
  NCCLCHK(rccl()->AllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, st));
  HIPCHK(hipMemcpyAsync(inout, c->d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return ODINN_OK;
}

int odinn_comm_allreduce_sum(odinn_comm* c, double* inout, int n) {
  if (!c || !inout || n < 0) return fail(ODINN_ERR_ARG, "bad all-reduce arguments");
  HIPCHK(hipSetDevice(c->device));
  return comm_allreduce_host(c, inout, n, c->stream);
}

int odinn_batch_loss_grad(odinn_batch* b, odinn_comm* comm, int adjoint, const double* theta, int P, int n_stops,
                          const double* tstops, int n_mb, const double* mb_times, const odinn_solver_opts* opts,
                          const odinn_adjoint_opts* adjoint_opts, double* loss, double* dtheta,
                          odinn_solve_stats* stats, odinn_solve_stats* stats_rev) {
  if (!b || !loss || !dtheta || P < 0) return fail(ODINN_ERR_ARG, "null argument");
  if (comm && comm->device != b->device) return fail(ODINN_ERR_ARG, "communicator is bound to device %d, batch to device %d", comm->device, b->device);
  if (adjoint != 0 && adjoint != 1) return fail(ODINN_ERR_ARG, "adjoint must be 0 (DiscreteAdjoint) or 1 (ContinuousAdjoint)");
  // This rank's part.  A failure here is rank-local and data-dependent (maxiters, a BoundsError of the U law's interpolant,
  // a stop table that does not hold a loss term's times, ...): the rank must still take part in the collective, or every
  // other rank blocks in ncclAllReduce forever.  So the buffer carries a status slot, [n_failed, loss, dtheta...]: a failed
  // rank contributes {1, 0, 0...}, and after the reduction EVERY rank returns an error if n_failed > 0.
  const int rc = adjoint == 0
                     ? odinn_loss_grad(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, loss, dtheta, stats)
                     : odinn_loss_grad_continuous(b, theta, P, n_stops, tstops, n_mb, mb_times, opts, adjoint_opts, loss, dtheta, stats, stats_rev);
  if (!comm || comm->nranks == 1) return rc;
  const std::string local_err = rc != ODINN_OK ? g_err : std::string();
  std::vector<double> buf(2 + (size_t)P, 0.0);
  buf[0] = rc != ODINN_OK ? 1.0 : 0.0;
  if (rc == ODINN_OK) {
    buf[1] = *loss;
    for (int q = 0; q < P; ++q) buf[2 + q] = dtheta[q];
  }
  // one ncclAllReduce(sum, ncclDouble, 2 + P) on the batch's stream (H2D, reduce, D2H enqueued back to back, one sync)
  const int rc2 = comm_allreduce_host(comm, buf.data(), 2 + P, b->stream);
  if (rc != ODINN_OK) return fail(rc, "%s", local_err.c_str());
  if (rc2 != ODINN_OK) return rc2;
  if (buf[0] > 0.0)
    return fail(ODINN_ERR_STATE, "odinn_batch_loss_grad: %d of %d ranks failed in their local gradient evaluation (this rank did "
                                 "not); loss and gradient are not valid", (int)buf[0], comm->nranks);
  *loss = buf[1];
  for (int q = 0; q < P; ++q) dtheta[q] = buf[2 + q];
  return ODINN_OK;
}

// ---- measurement --------------------------------------------------------------------------
// (measurement aid, read once per process: never inside a timed launch sequence)
static bool timed_adj_skip() {
  static const bool v = std::getenv("ODINN_TIMED_ADJ_SKIP") != nullptr;
  return v;
}
static int timed_prepare(odinn_batch* b) {
  CHK(use_dev(b));
  CHK(refresh_gd(b)); CHK(refresh_law_field(b));
  const Pools P = b->pools(true);
  // state: H0 in U[0]; a fixed small dt so that the RK registers stay finite
  const size_t fb = (size_t)b->ntot * sizeof(double);
  HIPCHK(hipMemcpyAsync(b->d_U[0], b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_S3, b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_lam[0], b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  const double ts[2] = {0.0, 1e30};
  {
    std::vector<std::vector<double>> keep;  // the synthetic two-stop table overrides the glaciers' own tables here
    keep.swap(b->own_stops);
    const int rc = build_stop_tables(b, 2, ts, 0, nullptr);
    b->own_stops.swap(keep);
    CHK(rc);
    b->solved = false;  // (the snapshot slots no longer belong to a solve's stop tables)
  }
  // two identical forward snapshots + reverse-solve state for ODINN_TIMED_ADJ_STAGE2
  if (b->nstops_alloc < 2) {
    dfree(b->d_snaps);
    CHK(dalloc(&b->d_snaps, (size_t)2 * b->ntot));
    b->nstops_alloc = 2;
  }
  HIPCHK(hipMemcpyAsync(b->d_snaps, b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(b->d_snaps + b->ntot, b->d_H0, fb, hipMemcpyDeviceToDevice, b->stream));
  if (!b->d_adj) { CHK(dalloc(&b->d_adj, b->G)); CHK(dalloc(&b->d_qw, b->G)); }
  if (!b->d_nr) { CHK(dalloc(&b->d_nr, (size_t)b->G)); CHK(dalloc(&b->d_ksn, (size_t)b->G)); CHK(dalloc(&b->d_lastseg, (size_t)b->G)); CHK(dalloc(&b->d_zerow, (size_t)b->G)); }
  b->rev_host.clear();  // (this path overwrites parts of the continuous adjoint's tables)
  {
    std::vector<int> two(b->G, 2);
    HIPCHK(hipMemcpyAsync(b->d_ksn, two.data(), (size_t)b->G * sizeof(int), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
  }
  launch_adj_begin(b->G, b->stream, P, b->d_adj, b->d_ksn, 0.0, b->d_mb_flag, b->d_mb_slot);
  if ((size_t)b->ntot > b->segs_cap) {  // the {H_j, dH} pairs of the one segment, as the reverse solve builds them
    if (b->d_segs) (void)hipFree(b->d_segs);
    b->d_segs = nullptr; b->segs_cap = 0;
    HIPCHK(hipMalloc(&b->d_segs, (size_t)b->ntot * sizeof(double2)));
    b->segs_cap = (size_t)b->ntot;
  }
  launch_seg_pairs(b->ntot, 1, b->stream, b->d_snaps, b->d_segs);
  launch_begin(b->G, b->stream, P, b->d_tstops, 0.0, 1e-6);
  HIPCHK(hipStreamSynchronize(b->stream));
  return ODINN_OK;
}

static int timed_one(odinn_batch* b, int which, int it) {
  const Pools P = b->pools(true);
  const LawDev L = b->lawdev();
  switch (which) {
    case ODINN_TIMED_FUSED_STEP: return launch_fused_step(b, 1e-6, 1e-8, 0);
    case ODINN_TIMED_FUSED_STEP_SKIP: return launch_fused_step(b, 1e-6, 1e-8, 1);
    case ODINN_TIMED_SOLVE_STEP:
    case ODINN_TIMED_SOLVE_STEP_STAGED: {
      const int scheme = which == ODINN_TIMED_SOLVE_STEP_STAGED ? 1 : pick_scheme(b, 0);
      CtrlArgs C{};
      C.tstops = b->d_tstops; C.nstops = b->d_nst; C.G = b->G; C.mb_flag = b->d_mb_flag; C.mb_slot = b->d_mb_slot;
      C.snap_slot = b->d_snapslot;
      C.dtmax = 0.0; C.adaptive = 0; C.fixed_dt = 1e-6; C.n_active = b->d_nactive;
      C.errpart = scheme == 2 ? b->fused_part() : b->d_part;
      C.stride = scheme == 2 ? 1 : 4; C.fused = scheme == 2 ? b->fused_ctrl() : 0;
      PostArgs PA;
      PA.snaps = b->d_tmpA; PA.premb = b->d_tmpB; PA.ntot = b->ntot; PA.mb0 = b->d_mb0; PA.Sref = nullptr;
      // exactly what do_solve enqueues per step for this batch: with snapshot-on-load there is no post-step launch
      const bool snapload = snap_on_load_mode(b, scheme, false) && !sc_mode(b, scheme);
      ScArgs SL{};
      SL.snaps = b->d_tmpA; SL.ntot = b->ntot; SL.snap_on_load = 1;
      if (scheme == 2) {
        CHK(launch_fused_step(b, 1e-6, 1e-8, 0, snapload ? &SL : nullptr));
        C.next_cur = -1;
      } else {
        CHK(launch_step(b, it & 1, 1e-6, 1e-8));
        C.next_cur = 1 - (it & 1);
      }
      launch_controller(b->G, b->stream, P, C);
      if (!snapload) launch_poststep(b->ntiles, b->stream, P, PA, b->d_U[0], b->d_U[1]);
      return ODINN_OK;
    }
    case ODINN_TIMED_DHDT: return launch_dhdt(b, b->d_U[0], b->d_tmpB, -1);
    case ODINN_TIMED_RK_STEP: return launch_step(b, it & 1, 1e-6, 1e-8);
    case ODINN_TIMED_RK_STAGE2: launch_stage<2>(b, P, L, b->d_U[0], b->d_U[1], 1e-6, 1e-8); return ODINN_OK;
    case ODINN_TIMED_VJP_H: {
      AdjArgs A{};
      A.H = b->d_U[0]; A.lam = b->d_lam[0]; A.out = b->d_tmpB;
      launch_vjp_H(b, 0, b->ntiles, P, L, A, 0, 0);
      return ODINN_OK;
    }
    case ODINN_TIMED_VJP_THETA: return theta_vjp_launch(b, b->d_U[0], b->d_lam[0], nullptr, -1, false);
    case ODINN_TIMED_EULER_CFL: launch_euler_cfl(b, P, L, b->d_U[0], b->d_U[1]); return ODINN_OK;
    case ODINN_TIMED_ADJ_STAGE2: {
      AdjStageArgs SA{};
      SA.snaps = b->d_snaps; SA.ntot = b->ntot; SA.adj = b->d_adj; SA.S2 = b->d_S2; SA.S3 = b->d_S3; SA.E = b->d_E;
      SA.abstol = 1e-8; SA.reltol = 1e-8; SA.src = b->d_lam[0]; SA.dst = b->d_lam[1];
      launch_adj_stage(b->lm_kern(), 0, 2, b->ntiles, b->stream, P, L, SA);
      return ODINN_OK;
    }
    case ODINN_TIMED_ADJ_FUSED_STEP: {
      if (b->lm() != 0) return fail(ODINN_ERR_STATE, "the fused reverse step exists for the integer-power law only");
      AdjFusedArgs FA{};
      FA.snaps = b->d_snaps; FA.ntot = b->ntot; FA.adj = b->d_adj; FA.lam0 = b->d_lam[0]; FA.lam1 = b->d_lam[1];
      FA.partF = b->d_partFt; FA.tilesF = b->d_tilesFt; FA.abstol = 1e-8; FA.reltol = 1e-8;
      if (sched_val(b->sched.adj_segs, "ODINN_ADJ_SEGS") != 0) FA.segs = b->d_segs;
      // (odinn_schedule.adj_rows = 4 times the 4-rows-per-thread instantiation: 54 x 22 tiles, which needs the segment pairs)
      const bool rows4 = sched_val(b->sched.adj_rows, "ODINN_ADJ_ROWS") == 4 && FA.segs;
      const bool rows8 = sched_val(b->sched.adj_rows, "ODINN_ADJ_ROWS") == 8 && FA.segs && b->gd[0].use_Afield && b->d_tilesFu;
      if (rows4) { FA.partF = b->d_partFv; FA.tilesF = b->d_tilesFv; }
      if (rows8) { FA.partF = b->d_partFu; FA.tilesF = b->d_tilesFu; }
      launch_adj_fused_strip(rows4 ? b->ntilesFv : rows8 ? b->ntilesFu : b->ntilesFt, b->gd[0].use_Afield, timed_adj_skip() ? 1 : 0,
                             rows4 ? 4 : rows8 ? 8 : TRPT, b->stream, P, FA);
      return ODINN_OK;
    }
    case ODINN_TIMED_LAW_FIELD:
      if (b->law_kind != ODINN_LAW_NN_A_GRIDDED) return fail(ODINN_ERR_STATE, "ODINN_TIMED_LAW_FIELD needs the NN_A_GRIDDED law");
      return refresh_law_field(b);
    default: return fail(ODINN_ERR_ARG, "unknown timed kernel %d", which);
  }
}

int odinn_bench_prepare(odinn_batch* b) {
  if (!b) return fail(ODINN_ERR_ARG, "null batch");
  return timed_prepare(b);
}

// The timed launches evaluate the Y law's network, as the seams do -- unless the schedule FIELD law_table is 1 (not just the
// default): then they run the table's kernels, which is how bench.py times k_rk_stage / k_adj_stage<., LM_YTAB>.
struct TimedTableScope {
  odinn_batch* b;
  bool on;
  explicit TimedTableScope(odinn_batch* b_) : b(b_), on(b_->sched.law_table == 1) { if (on) ++b->ytab_scope; }
  ~TimedTableScope() { if (on) --b->ytab_scope; }
};
int odinn_bench_enqueue(odinn_batch* b, int which, int first_iter, int n) {
  if (!b || n < 0) return fail(ODINN_ERR_ARG, "bad arguments");
  CHK(use_dev(b));
  TimedTableScope tts(b);
  for (int i = 0; i < n; ++i) CHK(timed_one(b, which, first_iter + i));
  HIPCHK(hipGetLastError());
  return ODINN_OK;
}

int odinn_time_kernel(odinn_batch* b, int which, int warmup, int iters, double* ms_total) {
  if (!b || !ms_total || iters <= 0) return fail(ODINN_ERR_ARG, "bad arguments");
  CHK(timed_prepare(b));
  TimedTableScope tts(b);
  for (int i = 0; i < warmup; ++i) CHK(timed_one(b, which, i));
  const char* eg = std::getenv("ODINN_TIME_GRAPH");
  if (eg && eg[0] == '1') {
    // measurement aid (tools/graph_probe.py): the same `iters` launches captured once into a hipGraph
    // and replayed.  Finding on MI355X / ROCm 7.2: no gain (24.0 -> 23.6 us per 3-kernel step at G = 4);
    // dependent kernels cost ~5 us each either way, so the solve loop stays on plain stream launches.
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIPCHK(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) CHK(timed_one(b, which, warmup + i));
    HIPCHK(hipStreamEndCapture(b->stream, &graph));
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphLaunch(exec, b->stream));  // warm
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    HIPCHK(hipGraphLaunch(exec, b->stream));
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    HIPCHK(hipEventSynchronize(b->ev1));
    HIPCHK(hipGraphExecDestroy(exec));
    HIPCHK(hipGraphDestroy(graph));
    float msg = 0.f;
    HIPCHK(hipEventElapsedTime(&msg, b->ev0, b->ev1));
    *ms_total = msg;
    return ODINN_OK;
  }
  HIPCHK(hipEventRecord(b->ev0, b->stream));
  for (int i = 0; i < iters; ++i) CHK(timed_one(b, which, warmup + i));
  HIPCHK(hipEventRecord(b->ev1, b->stream));
  HIPCHK(hipEventSynchronize(b->ev1));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
  *ms_total = ms;
  return ODINN_OK;
}

}  // extern "C"
