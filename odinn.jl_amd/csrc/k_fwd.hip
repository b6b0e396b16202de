// k_fwd.hip -- forward stencil kernels for one law mode (compile with -DODINN_LM=0 ... 8)
// inlined-MLP laws: log1p's table lives in LDS in these kernels (mode 2: filled by the tile loader; see sia2d_device.hpp)
#if defined(ODINN_LM) && ODINN_LM >= 2 && ODINN_LM <= 6 && !defined(ODINN_LOG1P_TABLE)
#define ODINN_LOG1P_TABLE 2
#endif
#include "launch.hpp"
#ifndef ODINN_LM
#error "define ODINN_LM"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
namespace odinn {
void CAT(launch_dhdt_lm, ODINN_LM)(int nblk, hipStream_t st, Pools P, LawDev L, const double* U, double* dH, int base) {
  hipLaunchKernelGGL(k_dhdt<ODINN_LM>, dim3(nblk), dim3(NT), 0, st, P, L, U, dH, base);
}
void CAT(launch_euler_cfl_lm, ODINN_LM)(int nblk, hipStream_t st, Pools P, LawDev L, const double* src, double* dst) {
  hipLaunchKernelGGL(k_euler_cfl<ODINN_LM>, dim3(nblk), dim3(NT), 0, st, P, L, src, dst);
}
void CAT(launch_rk_stage_lm, ODINN_LM)(int stage, int nblk, hipStream_t st, Pools P, LawDev L, const double* src,
                                        double* dst, double* S2, double* S3, double* E, double abstol, double reltol) {
  switch (stage) {
    case 1: hipLaunchKernelGGL((k_rk_stage<1, ODINN_LM>), dim3(nblk), dim3(NT), 0, st, P, L, src, dst, S2, S3, E, abstol, reltol); break;
    case 2: hipLaunchKernelGGL((k_rk_stage<2, ODINN_LM>), dim3(nblk), dim3(NT), 0, st, P, L, src, dst, S2, S3, E, abstol, reltol); break;
    case 3: hipLaunchKernelGGL((k_rk_stage<3, ODINN_LM>), dim3(nblk), dim3(NT), 0, st, P, L, src, dst, S2, S3, E, abstol, reltol); break;
    case 4: hipLaunchKernelGGL((k_rk_stage<4, ODINN_LM>), dim3(nblk), dim3(NT), 0, st, P, L, src, dst, S2, S3, E, abstol, reltol); break;
    default: hipLaunchKernelGGL((k_rk_stage<5, ODINN_LM>), dim3(nblk), dim3(NT), 0, st, P, L, src, dst, S2, S3, E, abstol, reltol); break;
  }
}
}  // namespace odinn
