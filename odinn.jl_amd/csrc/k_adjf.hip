// k_adjf.hip -- temporally fused reverse step of the continuous adjoint (integer-power A law, DiscreteVJP)
#include "launch.hpp"
#include "sia2d_adj_fused.hpp"
namespace odinn {
void launch_adj_fused_strip(int nblk, int afield, int skip, hipStream_t st, Pools P, AdjFusedArgs A) {
  if (afield) {
    if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true>), dim3(nblk), dim3(TNT), 0, st, P, A);
    else hipLaunchKernelGGL((k_adj_fused_strip<true, false>), dim3(nblk), dim3(TNT), 0, st, P, A);
  } else {
    if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true>), dim3(nblk), dim3(TNT), 0, st, P, A);
    else hipLaunchKernelGGL((k_adj_fused_strip<false, false>), dim3(nblk), dim3(TNT), 0, st, P, A);
  }
}
}  // namespace odinn
