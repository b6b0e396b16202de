// k_misc.hip -- controller, post-step, reductions, mass balance, hoisted-law kernels
#define ODINN_MISC_KERNELS 1
#include <algorithm>
#include <cmath>
#include <utility>
#include "launch.hpp"
namespace odinn {
void launch_controller(int G, hipStream_t st, Pools P, CtrlArgs C) { hipLaunchKernelGGL(k_controller, dim3(G), dim3(64), 0, st, P, C); }
void launch_poststep(int nblk, hipStream_t st, Pools P, PostArgs A, double* Ua, double* Ub) {
  hipLaunchKernelGGL(k_poststep, dim3(nblk), dim3(NT), 0, st, P, A, Ua, Ub);
}
void launch_sum_part(int ng, hipStream_t st, Pools P, int slot, double* out, int accumulate, int g0) {
  hipLaunchKernelGGL(k_sum_part, dim3(ng), dim3(64), 0, st, P, slot, out, accumulate, g0);
}
void launch_sum_part_steps(int ng, hipStream_t st, Pools P, const double* base, long long stride, int jhi, int jlo,
                           int slot, double* out) {
  hipLaunchKernelGGL(k_sum_part_steps, dim3(ng), dim3(64), 0, st, P, base, stride, jhi, jlo, slot, out);
}
void launch_sum_part_theta(int Pn, int ng, hipStream_t st, Pools P, const double* part_theta, double* out,
                           int accumulate, int g0) {
  hipLaunchKernelGGL(k_sum_part_theta, dim3(Pn, ng), dim3(64), 0, st, P, part_theta, Pn, out, accumulate, g0);
}
void launch_loss(int nblk, hipStream_t st, Pools P, const double* H, const double* Href, const unsigned char* mask,
                 const double* ws, const int* refslot, long long ntot, double log_eps) {
  hipLaunchKernelGGL(k_loss, dim3(nblk), dim3(NT), 0, st, P, H, Href, mask, ws, refslot, ntot, log_eps);
}
void launch_mb_vjp(int nblk, hipStream_t st, Pools P, const double* Hpre, const double* mb0, const double* Sref,
                   const double* lam_in, double* lam_out, int add, int base, const int* gflag, const int* gslot, long long ntot) {
  hipLaunchKernelGGL(k_mb_vjp, dim3(nblk), dim3(NT), 0, st, P, Hpre, mb0, Sref, lam_in, lam_out, add, base, gflag, gslot, ntot);
}
void launch_mb_apply(int nblk, hipStream_t st, Pools P, const double* H, const double* mb0, const double* Sref,
                     double* Hn, double* MBout, int base) {
  hipLaunchKernelGGL(k_mb_apply, dim3(nblk), dim3(NT), 0, st, P, H, mb0, Sref, Hn, MBout, base);
}
template <class AR>
static bool law_is(const LawDev& L) {
  if (L.n_layers != AR::NL) return false;
  for (int l = 0; l <= AR::NL; ++l) if (L.widths[l] != AR::W[l]) return false;
  for (int l = 0; l < AR::NL; ++l) if (L.acts[l] != AR::A[l]) return false;
  return true;
}
void launch_dhdt_sums(int nblk, int G, hipStream_t st, Pools P, const double* snaps, const int* i0s, const int* i1s, long long ntot,
                      double* part2, const double* dts, const double* refs, double w, double* coef, double* lossacc) {
  hipLaunchKernelGGL(k_dhdt_sums, dim3(nblk), dim3(NT), 0, st, P, snaps, i0s, i1s, ntot, part2);
  hipLaunchKernelGGL(k_dhdt_finish, dim3(G), dim3(64), 0, st, P, part2, i0s, dts, refs, w, coef, lossacc);
}
void launch_dhdt_cot(int nblk, hipStream_t st, Pools P, double* lam, const double* snaps, const int* i0s, const int* i1s,
                     const double* coef, int j, long long ntot) {
  hipLaunchKernelGGL(k_dhdt_cot, dim3(nblk), dim3(NT), 0, st, P, lam, snaps, i0s, i1s, coef, j, ntot);
}
void launch_law_field(hipStream_t st, LawDev L, const double* T, double* Aout, long long n) {
  const dim3 grid((unsigned)((n + NT - 1) / NT));
  if (law_is<Arch16A>(L)) hipLaunchKernelGGL(k_law_field_fixed<Arch16A>, grid, dim3(NT), 0, st, L, T, Aout, n);
  else if (law_is<ArchDefA>(L)) hipLaunchKernelGGL(k_law_field_fixed<ArchDefA>, grid, dim3(NT), 0, st, L, T, Aout, n);
  else hipLaunchKernelGGL(k_law_field, grid, dim3(NT), 0, st, L, T, Aout, n);
}
int launch_law_field_grad(hipStream_t st, LawDev L, const double* T, const double* G, long long n, double* part_theta,
                          int max_rows) {
  // wave-reduced kernel: 8 workgroups per CU at most, every wavefront strides over the 64-node chunks; returns the rows written
  const long long nchunk = (n + 63) / 64;
  int nblk = (int)std::min<long long>((nchunk + NW - 1) / NW, 2048);
  nblk = std::max(1, std::min(nblk, max_rows));
  const size_t dyn = (size_t)NW * L.P * sizeof(double) + (size_t)L.P * sizeof(int);
  if (law_is<Arch16A>(L)) hipLaunchKernelGGL((k_law_field_grad_wave<Arch16A, true>), dim3(nblk), dim3(NT), dyn, st, L, T, G, n, part_theta);
  else if (law_is<ArchDefA>(L)) hipLaunchKernelGGL((k_law_field_grad_wave<ArchDefA, true>), dim3(nblk), dim3(NT), dyn, st, L, T, G, n, part_theta);
  else hipLaunchKernelGGL((k_law_field_grad_wave<ArchRT, false>), dim3(nblk), dim3(NT), dyn, st, L, T, G, n, part_theta);
  return nblk;
}
void launch_law_field_grad_scratch(int nblk, hipStream_t st, LawDev L, const double* T, const double* G, long long n,
                                   double* gscratch, double* part_theta) {
  hipLaunchKernelGGL(k_law_field_grad, dim3(nblk), dim3(NT), 0, st, L, T, G, n, gscratch, part_theta);
}
void launch_sum_rows(int Pn, hipStream_t st, const double* part, int nrows, double* out) {
  hipLaunchKernelGGL(k_sum_rows, dim3(Pn), dim3(64), 0, st, part, nrows, Pn, out);
}
void launch_sum_slots(hipStream_t st, long long n, int nslots, const double* slots, double* out) {
  hipLaunchKernelGGL(k_sum_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, nslots, slots, out);
}
// table of the Y law for G glaciers x ni intervals (k_ytab_build); stat: 3 zeroed 64-bit words
static const YtabVinv& ytab_vinv() {
  static const YtabVinv V = [] {
    YtabVinv v{};
    long double a[6][12];
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int j = 0; j < 6; ++j) {
      const long double x = cosl((2 * j + 1) * pi / 12.0L);
      v.node[j] = (double)x;
      const long double xd = (long double)v.node[j];  // the nodes the kernel really uses
      long double pw = 1.0L;
      for (int k = 0; k < 6; ++k) { a[j][k] = pw; pw *= xd; }
      for (int k = 0; k < 6; ++k) a[j][6 + k] = j == k ? 1.0L : 0.0L;
    }
    for (int c = 0; c < 6; ++c) {  // Gauss-Jordan with partial pivoting: [V | I] -> [I | V^-1]
      int pv = c;
      for (int r = c + 1; r < 6; ++r) if (fabsl(a[r][c]) > fabsl(a[pv][c])) pv = r;
      for (int k = 0; k < 12; ++k) std::swap(a[c][k], a[pv][k]);
      const long double d = a[c][c];
      for (int k = 0; k < 12; ++k) a[c][k] /= d;
      for (int r = 0; r < 6; ++r) if (r != c) { const long double f = a[r][c]; for (int k = 0; k < 12; ++k) a[r][k] -= f * a[c][k]; }
    }
    for (int k = 0; k < 6; ++k) for (int j = 0; j < 6; ++j) v.v[k][j] = (double)a[k][6 + j];
    return v;
  }();
  return V;
}
void launch_ytab_build(hipStream_t st, Pools P, LawDev L, int G, double* tab, int ni, double floor_abs, unsigned long long* stat) {
  hipLaunchKernelGGL(k_ytab_build, dim3((unsigned)((ni + 255) / 256), (unsigned)G), dim3(256), 0, st, P, L, ytab_vinv(), tab, ni, floor_abs, stat);
}
// table of the U law: nh x ns patches (k_utab_build)
void launch_utab_build(hipStream_t st, LawDev L, double* tab, int nh, int ns, double floor_abs, unsigned long long* stat) {
  const long long np = (long long)nh * ns;
  hipLaunchKernelGGL(k_utab_build, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, st, L, ytab_vinv(), tab, nh, ns, floor_abs, stat);
}
void launch_eval_law(hipStream_t st, Pools P, LawDev L, const double* U, double* out, int gidx, long long nd) {
  hipLaunchKernelGGL(k_eval_law, dim3((unsigned)((nd + NT - 1) / NT)), dim3(NT), 0, st, P, L, U, out, gidx);
}
void launch_axpy(long long n, hipStream_t st, double a, const double* x, const double* y, double* z) {
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)(nb < 65536 ? (nb < 1 ? 1 : nb) : 65536)), dim3(256), 0, st, n, a, x, y, z);
}
void launch_axpy_g(int nblk, hipStream_t st, Pools P, const double* x, const double* y, double* z) {
  hipLaunchKernelGGL(k_axpy_g, dim3(nblk), dim3(NT), 0, st, P, x, y, z);
}
void launch_initdt_norms(int nblk, hipStream_t st, Pools P, const double* U, const double* F0, const double* F1,
                         double abstol, double reltol) {
  hipLaunchKernelGGL(k_initdt_norms, dim3(nblk), dim3(NT), 0, st, P, U, F0, F1, abstol, reltol);
}
void launch_initdt_ctrl(int G, hipStream_t st, Pools P, int phase, double tspan, double dtmax, double* dt0store) {
  hipLaunchKernelGGL(k_initdt_ctrl, dim3(G), dim3(64), 0, st, P, phase, tspan, dtmax, dt0store);
}
void launch_begin(int G, hipStream_t st, Pools P, const double* tstops, double dtmax, double dt_given) {
  hipLaunchKernelGGL(k_begin, dim3((G + 63) / 64), dim3(64), 0, st, P, G, tstops, dtmax, dt_given);
}
void launch_set_dt(int G, hipStream_t st, Pools P, double dt) {
  hipLaunchKernelGGL(k_set_dt, dim3((G + 63) / 64), dim3(64), 0, st, P, G, dt);
}
void launch_vref_itp(int nblk, hipStream_t st, Pools P, VItpArgs A) {
  hipLaunchKernelGGL(k_vref_itp, dim3(nblk), dim3(NT), 0, st, P, A);
}
void launch_vq_finish(int G, hipStream_t st, Pools P, const AdjState* adj, const int* slotA, int scale_loss, double wq, double* out,
                      double* coef) {
  hipLaunchKernelGGL(k_vq_finish, dim3(G), dim3(64), 0, st, P, adj, slotA, G, scale_loss, wq, out, coef);
}
void launch_vref_scale(int G, hipStream_t st, Pools P, const AdjState* adj, const int* slotA, int scale_loss, double wq,
                       double* scale_out, double* w_out) {
  hipLaunchKernelGGL(k_vref_scale, dim3(G), dim3(64), 0, st, P, adj, slotA, G, scale_loss, wq, scale_out, w_out);
}
void launch_adj_begin(int G, hipStream_t st, Pools P, AdjState* adj, const int* n_snaps, double tau0, const int* mb_flags,
                      const int* mb_slots) {
  hipLaunchKernelGGL(k_adj_begin, dim3((G + 63) / 64), dim3(64), 0, st, P, G, adj, n_snaps, tau0, mb_flags, mb_slots);
}
void launch_adj_itp(int G, hipStream_t st, Pools P, AdjState* adj, const double* tsnap, const int* n_snaps, int all_at_end) {
  hipLaunchKernelGGL(k_adj_itp, dim3((G + 63) / 64), dim3(64), 0, st, P, G, adj, tsnap, n_snaps, all_at_end);
}
void launch_adj_poststep(int nblk, hipStream_t st, Pools P, AdjPostArgs A, double* Ua, double* Ub) {
  const int per_wg = nblk / 512 < 1 ? 1 : (nblk / 512 > ADJ_POST_TILES ? ADJ_POST_TILES : nblk / 512);
  hipLaunchKernelGGL(k_adj_poststep, dim3((nblk + per_wg - 1) / per_wg), dim3(NT), 0, st, P, A, Ua, Ub, nblk, per_wg);
}
void launch_vreg_prep(int nblk, hipStream_t st, Pools P, const double* H, const double* vx, const double* vy, const double* w,
                      int dist, double* Vabs, unsigned char* mask) {
  hipLaunchKernelGGL(k_vreg_prep, dim3(nblk), dim3(NT), 0, st, P, H, vx, vy, w, dist, Vabs, mask);
}
void launch_vreg_lap(int nblk, hipStream_t st, Pools P, const double* Vabs, const unsigned char* mask, const double* w, double* r) {
  hipLaunchKernelGGL(k_vreg_lap, dim3(nblk), dim3(NT), 0, st, P, Vabs, mask, w, r);
}
void launch_vreg_cot(int nblk, hipStream_t st, Pools P, const double* r, const double* Vabs, const double* w, double* vx, double* vy) {
  hipLaunchKernelGGL(k_vreg_cot, dim3(nblk), dim3(NT), 0, st, P, r, Vabs, w, vx, vy);
}
void launch_seg_pairs(long long ntot, int n_seg, hipStream_t st, const double* snaps, double2* segs) {
  hipLaunchKernelGGL(k_seg_pairs, dim3(65536), dim3(256), 0, st, ntot, n_seg, snaps, segs);
}
void launch_sum_tilesFt(int G, int rows, hipStream_t st, Pools P, const double* part, double* out) {
  hipLaunchKernelGGL(k_sum_tilesFt, dim3(G), dim3(64), 0, st, P, part, out, rows);
}
void launch_lerp_g(int nblk, hipStream_t st, Pools P, const double* snaps, long long ntot, const int* seg, const double* sw, double* out) {
  hipLaunchKernelGGL(k_lerp_g, dim3(nblk), dim3(NT), 0, st, P, snaps, ntot, seg, sw, out);
}
void launch_lerp(long long n, hipStream_t st, double s, const double* a, const double* b, double* out) {
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_lerp, dim3((unsigned)(nb < 65536 ? (nb < 1 ? 1 : nb) : 65536)), dim3(256), 0, st, n, s, a, b, out);
}
void launch_tikhonov(hipStream_t st, const double* a, const unsigned char* mask, double* r, double* grad,
                     double* partial, int nx, int ny, double dx, double dy) {
  const dim3 grid((nx + 63) / 64, (ny + NW - 1) / NW);
  const double wx = 0.25 / (dx * dx), wy = 0.25 / (dy * dy);
  hipLaunchKernelGGL(k_tikhonov_fwd, grid, dim3(NT), 0, st, a, mask, r, partial, nx, ny, wx, wy);
  hipLaunchKernelGGL(k_tikhonov_bwd, grid, dim3(NT), 0, st, r, grad, nx, ny, wx, wy);
}
}  // namespace odinn
