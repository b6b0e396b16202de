// k_adj.hip -- discrete-adjoint stencil kernels for one law mode (-DODINN_LM=0 ... 8)
// inlined-MLP laws: log1p's table lives in LDS in these kernels (mode 2: filled by the tile loader; see sia2d_device.hpp)
#if defined(ODINN_LM) && ODINN_LM >= 2 && ODINN_LM <= 6 && !defined(ODINN_LOG1P_TABLE)
#define ODINN_LOG1P_TABLE 2
#endif
#include "launch.hpp"
#include "sia2d_nn_grad.hpp"
#ifndef ODINN_LM
#error "define ODINN_LM"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
namespace odinn {
// Compile-time architectures (law modes 3..5): one instantiation per law kind (NK = 3: Y law, 4: U law; see node_D); the other
// law modes read the kind from the descriptor (NK = 0)
#if ODINN_LM >= 3 && ODINN_LM <= 5
#define ODINN_NK_DISPATCH(CALL) { if (L.kind == 3) { constexpr int NK = 3; CALL; } else { constexpr int NK = 4; CALL; } }
#else
#define ODINN_NK_DISPATCH(CALL) { constexpr int NK = 0; CALL; }
#endif
// vj: 0 = DiscreteVJP stencil, 1 = ContinuousVJP stencil
void CAT(launch_vjp_H_lm, ODINN_LM)(int mode, int vj, int nblk, hipStream_t st, Pools P, LawDev L, AdjArgs A, int base) {
  if (vj) {
    if (mode == 0) ODINN_NK_DISPATCH(hipLaunchKernelGGL((k_vjp_H<0, ODINN_LM, 1, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A, base))
    else ODINN_NK_DISPATCH(hipLaunchKernelGGL((k_vjp_H<1, ODINN_LM, 1, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A, base))
    return;
  }
  if (mode == 0) ODINN_NK_DISPATCH(hipLaunchKernelGGL((k_vjp_H<0, ODINN_LM, 0, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A, base))
  else ODINN_NK_DISPATCH(hipLaunchKernelGGL((k_vjp_H<1, ODINN_LM, 0, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A, base))
}
template <int VJ, int NK>
static void adj_stage_dispatch(int stage, int nblk, hipStream_t st, Pools P, LawDev L, AdjStageArgs A) {
  switch (stage) {
    case 1: hipLaunchKernelGGL((k_adj_stage<1, ODINN_LM, VJ, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A); break;
    case 2: hipLaunchKernelGGL((k_adj_stage<2, ODINN_LM, VJ, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A); break;
    case 3: hipLaunchKernelGGL((k_adj_stage<3, ODINN_LM, VJ, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A); break;
    case 4: hipLaunchKernelGGL((k_adj_stage<4, ODINN_LM, VJ, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A); break;
    default: hipLaunchKernelGGL((k_adj_stage<5, ODINN_LM, VJ, NK>), dim3(nblk), dim3(NT), 0, st, P, L, A); break;
  }
}
void CAT(launch_adj_stage_lm, ODINN_LM)(int stage, int vj, int nblk, hipStream_t st, Pools P, LawDev L, AdjStageArgs A) {
  if (vj) { ODINN_NK_DISPATCH((adj_stage_dispatch<1, NK>(stage, nblk, st, P, L, A))) return; }
  ODINN_NK_DISPATCH((adj_stage_dispatch<0, NK>(stage, nblk, st, P, L, A)))
}
void CAT(launch_vjp_theta_lm, ODINN_LM)(int nblk, hipStream_t st, Pools P, LawDev L, ThArgs A, int base) {
#if ODINN_LM >= 3 && ODINN_LM <= 5
  if (A.emitH) {  // `:Linear` gradient interpolation: the kernel only emits (Hbar, node weight), no per-node backprop
    hipLaunchKernelGGL(k_vjp_theta<ODINN_LM>, dim3(nblk), dim3(NT), 0, st, P, L, A, base);
    return;
  }
#endif
#if ODINN_LM == 3
  hipLaunchKernelGGL(k_vjp_theta_nn<ArchDef>, dim3(nblk), dim3(NT), 0, st, P, L, A, base);
#elif ODINN_LM == 4
  hipLaunchKernelGGL(k_vjp_theta_nn<Arch16>, dim3(nblk), dim3(NT), 0, st, P, L, A, base);
#elif ODINN_LM == 5
  hipLaunchKernelGGL(k_vjp_theta_nn<ArchLight>, dim3(nblk), dim3(NT), 0, st, P, L, A, base);
#else
  hipLaunchKernelGGL(k_vjp_theta<ODINN_LM>, dim3(nblk), dim3(NT), 0, st, P, L, A, base);
#endif
}
}  // namespace odinn
