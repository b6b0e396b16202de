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

// Kernel-schedule switch: ONE environment variable, ODINN_SCHEDULE="field=value,field=value,..." (a measurement / A-B override of the
// odinn_schedule fields of every batch of the process: step_sc, fused_tiles, ..., adj_ut_fused, plus scheme = 1 | 2), if it names the
// field; else the batch's odinn_schedule field; else -1 = the library's own measured rule.  Values are digit strings (fused_tiles also
// s | l | t | u).  `env` is the historical per-field variable name: its suffix in lower case is the key.  Read per call: tests toggle it.
static const char* sched_env_str(const char* env, char* buf, size_t nbuf) {
  const char* all = std::getenv("ODINN_SCHEDULE");
  if (!all || !*all) return nullptr;
  char key[48];
  size_t n = 0;
  for (const char* c = env + 6; *c && n + 1 < sizeof(key); ++c) key[n++] = (char)(*c >= 'A' && *c <= 'Z' ? *c - 'A' + 'a' : *c);
  key[n] = 0;
  for (const char* p = all; *p;) {
    const char* e = p;
    while (*e && *e != ',') ++e;
    const char* q = p;
    while (q < e && *q != '=') ++q;
    while (p < q && *p == ' ') ++p;
    if ((size_t)(q - p) == n && std::strncmp(p, key, n) == 0 && q < e) {
      size_t m = std::min((size_t)(e - q - 1), nbuf - 1);
      std::memcpy(buf, q + 1, m);
      buf[m] = 0;
      return buf;
    }
    p = *e ? e + 1 : e;
  }
  return nullptr;
}
static int sched_val(int field, const char* env) {
  char buf[32];
  if (const char* e = sched_env_str(env, buf, sizeof(buf)))
    if (e[0] >= '0' && e[0] <= '9') return std::atoi(e);
  return field;
}

struct odinn_batch {
  odinn_schedule sched = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, {0}};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // odinn_bench_kernel_events: HIP events around the dominant kernel of every step that odinn_bench_enqueue launches
  bool bench_ev_on = false;
  std::vector<hipEvent_t> bench_ev;  // pool, pairs
  size_t bench_ev_used = 0;
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
    char ebuf[32];
    if (const char* e = sched_env_str("ODINN_FUSED_TILES", ebuf, sizeof(ebuf))) {
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
    return true;
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
  // The table's resolution is chosen by measurement (ytab_refresh): the coarsest of 16 x 8, 32 x 16, 64 x 32, 128 x 64 patches whose
  // deviation from the network passes -- a smooth law needs 37 ... 147 KB instead of 2.4 MB, so that the 64 nodes of a wavefront row
  // gather from a handful of patches that stay in the L1 (utab_level: where the search starts; reset with the range).
  int utab_nh = 16, utab_ns = 8, utab_level = 0;
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
    {
      const char* e = std::getenv("ODINN_UT_LDS");  // (read per call: tests toggle it)
      L.ut_nolds = e && e[0] == '0' ? 1 : 0;
    }
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
constexpr int UTAB_LEVELS = 4;
int ytab_refresh_level(odinn_batch* b, double gate);
int ytab_refresh(odinn_batch* b) {
  if (b->law_kind != ODINN_LAW_NN_U) return ytab_refresh_level(b, 1.0);
  // U law: the coarsest level that passes (coarser levels must pass with a margin of 10: their 16 check points per patch are sparser
  // in absolute terms); a level that failed is never tried again for this law / range
  const int forced = std::getenv("ODINN_UTAB_LEVEL") ? std::atoi(std::getenv("ODINN_UTAB_LEVEL")) : -1;  // (A/B and test aid; read per call)
  if (forced >= 0 && forced < UTAB_LEVELS) b->utab_level = forced;
  for (int lev = b->utab_level;; ++lev) {
    b->utab_level = lev;
    b->utab_nh = 16 << lev; b->utab_ns = 8 << lev;
    const bool last = lev == UTAB_LEVELS - 1 || forced >= 0;
    CHK(ytab_refresh_level(b, last ? 1.0 : 0.1));
    if (b->ytab_ok || last) return ODINN_OK;
  }
}
int ytab_refresh_level(odinn_batch* b, double gate) {
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
  b->ytab_ok = st[0] <= gate * YTAB_TOL && st[1] <= gate * YTAB_TOL * st[2] && std::isfinite(st[2]) && st[2] > 0.0;
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
      b->gd[g].yt_fast = (b->gd[g].fast && b->gd[g].nH == 3.0 && b->gd[g].nS == 3.0) ? 1 : 0;
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

#include "host_solve.inc"
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
  for (hipEvent_t e : b->bench_ev) (void)hipEventDestroy(e);
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
    b->utab_hmax = 0.0; b->utab_level = 0;
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
  b->utab_hmax = 0.0; b->utab_level = 0;
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

#include "host_seams.inc"
#include "host_interp.inc"
#include "host_discrete_adjoint.inc"
#include "host_continuous_adjoint.inc"
// ---- multi-GPU: RCCL communicator behind the C ABI (SIA2D_grad!, gradient.jl:6-31) ----------------------------
}  // extern "C"
#include "host_comm.inc"
#include "host_timing.inc"