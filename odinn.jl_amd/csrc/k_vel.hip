// k_vel.hip -- surface-velocity kernels (A-type law modes)
#define ODINN_VEL_KERNELS 1
#include "launch.hpp"
#include "sia2d_velocity.hpp"
namespace odinn {
void launch_surface_V(int lm, int nblk, hipStream_t st, Pools P, const double* U, double* Vx, double* Vy, int base) {
  if (lm == 0) hipLaunchKernelGGL(k_surface_V<0>, dim3(nblk), dim3(NT), 0, st, P, U, Vx, Vy, base);
  else hipLaunchKernelGGL(k_surface_V<1>, dim3(nblk), dim3(NT), 0, st, P, U, Vx, Vy, base);
}
void launch_surfV_vjp(int lm, int mode, int nblk, hipStream_t st, Pools P, const VArgs& A, int base) {
  if (lm == 0) {
    if (mode == 0) hipLaunchKernelGGL((k_surfV_vjp<0, 0>), dim3(nblk), dim3(NT), 0, st, P, A, base);
    else if (mode == 1) hipLaunchKernelGGL((k_surfV_vjp<1, 0>), dim3(nblk), dim3(NT), 0, st, P, A, base);
    else hipLaunchKernelGGL((k_surfV_vjp<2, 0>), dim3(nblk), dim3(NT), 0, st, P, A, base);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_surfV_vjp<0, 1>), dim3(nblk), dim3(NT), 0, st, P, A, base);
    else if (mode == 1) hipLaunchKernelGGL((k_surfV_vjp<1, 1>), dim3(nblk), dim3(NT), 0, st, P, A, base);
    else hipLaunchKernelGGL((k_surfV_vjp<2, 1>), dim3(nblk), dim3(NT), 0, st, P, A, base);
  }
}
void launch_avgv_axpy(int nblk, hipStream_t st, Pools P, const double* Vx, const double* Vy, double* ax, double* ay, const double* w) {
  hipLaunchKernelGGL(k_avgv_axpy, dim3(nblk), dim3(NT), 0, st, P, Vx, Vy, ax, ay, w);
}
void launch_avgv_cot(int nblk, hipStream_t st, Pools P, double* ax, double* ay, const double* Vabs, const double* Vxr,
                     const double* Vyr, const unsigned char* on, int component_abs, double weight) {
  hipLaunchKernelGGL(k_avgv_cot, dim3(nblk), dim3(NT), 0, st, P, ax, ay, Vabs, Vxr, Vyr, on, component_abs, weight);
}
}  // namespace odinn
