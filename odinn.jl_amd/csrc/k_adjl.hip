// k_adjl.hip -- temporally fused reverse step of the continuous adjoint on LDS tiles (the U law through its table; sia2d_adj_lds.hpp)
#include "launch.hpp"
#include "sia2d_adj_lds.hpp"
namespace odinn {
// rows: 22 = the 54 x 22 tile table (Fv), 8 = the 54 x 8 latency tiles (Fs); tilesF / partF of A belong to that table
void launch_adj_fused_lds(int nblk, int skip, int foy, hipStream_t st, Pools P, LawDev L, AdjFusedArgs A) {
  if (foy == FOYS) {
    if (skip) hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, true, FOYS>), dim3(nblk), dim3(FNT), 0, st, P, L, A);
    else hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, false, FOYS>), dim3(nblk), dim3(FNT), 0, st, P, L, A);
  } else {
    if (skip) hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, true, FOYT4>), dim3(nblk), dim3(FNT), 0, st, P, L, A);
    else hipLaunchKernelGGL((k_adj_fused_lds<LM_UTAB, false, FOYT4>), dim3(nblk), dim3(FNT), 0, st, P, L, A);
  }
}
}  // namespace odinn
