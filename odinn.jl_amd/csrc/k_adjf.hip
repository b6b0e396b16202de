// k_adjf.hip -- temporally fused reverse step of the continuous adjoint (integer-power A law, DiscreteVJP)
#include "launch.hpp"
#include "sia2d_adj_fused.hpp"
namespace odinn {
void launch_adj_fused_strip(int nblk, int afield, hipStream_t st, Pools P, AdjFusedArgs A) {
  if (afield) hipLaunchKernelGGL((k_adj_fused_strip<true>), dim3(nblk), dim3(TNT), 0, st, P, A);
  else hipLaunchKernelGGL((k_adj_fused_strip<false>), dim3(nblk), dim3(TNT), 0, st, P, A);
}
}  // namespace odinn
