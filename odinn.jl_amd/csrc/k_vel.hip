// k_vel.hip -- surface-velocity kernels (A-type law modes)
#define ODINN_VEL_KERNELS 1
// U law (target :D): the surface-velocity kernels evaluate the network five times per dual node; log1p's table lives in LDS
// here too (mode 2: filled by load_tile_HS2, which every kernel of this unit calls before its first barrier)
#ifndef ODINN_LOG1P_TABLE
#define ODINN_LOG1P_TABLE 2
#endif
#include "launch.hpp"
#include "sia2d_velocity.hpp"
namespace odinn {
void launch_surface_V(int lm, int nblk, hipStream_t st, Pools P, LawDev L, const double* U, double* Vx, double* Vy, int base,
                      double finv) {
  if (lm == 0) hipLaunchKernelGGL(k_surface_V<0>, dim3(nblk), dim3(NT), 0, st, P, L, U, Vx, Vy, base, finv);
  else if (lm == 1) hipLaunchKernelGGL(k_surface_V<1>, dim3(nblk), dim3(NT), 0, st, P, L, U, Vx, Vy, base, finv);
  else hipLaunchKernelGGL(k_surface_V<LM_NN>, dim3(nblk), dim3(NT), 0, st, P, L, U, Vx, Vy, base, finv);
}
// lm >= 2: the U law (target :D) through the run-time MLP evaluation, whatever its architecture
void launch_surfV_vjp(int lm, int mode, int nblk, hipStream_t st, Pools P, LawDev L, const VArgs& A, int base) {
#define ODINN_VJP_CASE(M, LMV) hipLaunchKernelGGL((k_surfV_vjp<M, LMV>), dim3(nblk), dim3(NT), 0, st, P, L, A, base)
  const int l = lm >= 2 ? 2 : lm;
  if (l == 0) { if (mode == 0) ODINN_VJP_CASE(0, 0); else if (mode == 1) ODINN_VJP_CASE(1, 0); else ODINN_VJP_CASE(2, 0); }
  else if (l == 1) { if (mode == 0) ODINN_VJP_CASE(0, 1); else if (mode == 1) ODINN_VJP_CASE(1, 1); else ODINN_VJP_CASE(2, 1); }
  else { if (mode == 0) ODINN_VJP_CASE(0, LM_NN); else if (mode == 1) ODINN_VJP_CASE(1, LM_NN); else ODINN_VJP_CASE(2, LM_NN); }
#undef ODINN_VJP_CASE
}
void launch_surfV_theta_node(int lm, int nblk, hipStream_t st, Pools P, VItpArgs I, const double* snaps, int component_abs, double log_eps,
                             double* tnode) {
  if (lm == 0) hipLaunchKernelGGL((k_surfV_theta_node<0>), dim3(nblk), dim3(NT), 0, st, P, I, snaps, component_abs, log_eps, tnode);
  else hipLaunchKernelGGL((k_surfV_theta_node<1>), dim3(nblk), dim3(NT), 0, st, P, I, snaps, component_abs, log_eps, tnode);
}
void launch_surfV_theta_only(int lm, int nblk, hipStream_t st, Pools P, const VArgs& A) {
  if (lm == 0) hipLaunchKernelGGL((k_surfV_theta_only<0>), dim3(nblk), dim3(NT), 0, st, P, A);
  else hipLaunchKernelGGL((k_surfV_theta_only<1>), dim3(nblk), dim3(NT), 0, st, P, A);
}
void launch_gacc_axpy(int nblk, hipStream_t st, Pools P, const double* coef, const double* tnode, double* Gacc) {
  hipLaunchKernelGGL(k_gacc_axpy, dim3(nblk), dim3(NT), 0, st, P, coef, tnode, Gacc);
}
void launch_avgv_axpy(int nblk, hipStream_t st, Pools P, const double* Vx, const double* Vy, double* ax, double* ay, const double* w) {
  hipLaunchKernelGGL(k_avgv_axpy, dim3(nblk), dim3(NT), 0, st, P, Vx, Vy, ax, ay, w);
}
void launch_avgv_cot(int nblk, hipStream_t st, Pools P, double* ax, double* ay, const double* Vabs, const double* Vxr,
                     const double* Vyr, const unsigned char* on, int component_abs, double weight) {
  hipLaunchKernelGGL(k_avgv_cot, dim3(nblk), dim3(NT), 0, st, P, ax, ay, Vabs, Vxr, Vyr, on, component_abs, weight);
}
}  // namespace odinn
