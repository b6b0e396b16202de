// k_adjl.hip -- temporally fused reverse step of the continuous adjoint on LDS tiles (the U law through its table; sia2d_adj_lds.hpp)
#include "launch.hpp"
#include "sia2d_adj_lds.hpp"
namespace odinn {
// 54 x 22 output tiles (the table Fv of the strip kernels' 4-row form: tilesF / partF of A belong to it), LNW = 12 waves per workgroup
void launch_adj_fused_lds(int nblk, int skip, hipStream_t st, Pools P, LawDev L, AdjFusedArgs A) {
  if (skip) hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, true, FOYT4, LNW>), dim3(nblk), dim3(64 * LNW), 0, st, P, L, A);
  else hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, false, FOYT4, LNW>), dim3(nblk), dim3(64 * LNW), 0, st, P, L, A);
}
}  // namespace odinn
