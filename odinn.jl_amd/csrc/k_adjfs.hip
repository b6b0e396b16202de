// k_adjfs.hip -- the self-controlled instantiations of the fused reverse step (k_adj_fused_strip<..., SC = true>)
#include "adjf_dispatch.hpp"
namespace odinn {
void launch_adj_fused_strip_sc(int nblk, int afield, int skip, int rows, hipStream_t st, const Pools& P, const AdjFusedArgs& A) {
  adjf_dispatch<true>(nblk, afield, skip, rows, st, P, A);
}
}  // namespace odinn
