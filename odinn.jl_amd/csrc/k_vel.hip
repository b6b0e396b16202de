// k_vel.hip -- surface-velocity kernels (A-type law modes)
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
    else hipLaunchKernelGGL((k_surfV_vjp<1, 0>), dim3(nblk), dim3(NT), 0, st, P, A, base);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_surfV_vjp<0, 1>), dim3(nblk), dim3(NT), 0, st, P, A, base);
    else hipLaunchKernelGGL((k_surfV_vjp<1, 1>), dim3(nblk), dim3(NT), 0, st, P, A, base);
  }
}
}  // namespace odinn
