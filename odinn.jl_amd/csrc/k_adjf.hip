// k_adjf.hip -- temporally fused reverse step of the continuous adjoint (integer-power A law, DiscreteVJP)
#include "adjf_dispatch.hpp"
namespace odinn {
void launch_adj_fused_strip_sc(int nblk, int afield, int skip, int rows, hipStream_t st, const Pools& P, const AdjFusedArgs& A);
void launch_adj_fused_strip(int nblk, int afield, int skip, int rows, hipStream_t st, Pools P, AdjFusedArgs A, int sc) {
  if (sc) launch_adj_fused_strip_sc(nblk, afield, skip, rows, st, P, A);  // (k_adjfs.hip; the caller guarantees A.segs)
  else adjf_dispatch<false>(nblk, afield, skip, rows, st, P, A);
}
void launch_vjp_H_strip(int mode, int afield, int nblk, hipStream_t st, Pools P, const int4* tilesD, AdjArgs A) {
  if (A.ytab) {  // the Y law through its table (the caller guarantees !afield)
    if (mode) hipLaunchKernelGGL((k_vjp_H_strip<false, 1, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_H_strip<false, 0, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    return;
  }
  if (afield) {
    if (mode) hipLaunchKernelGGL((k_vjp_H_strip<true, 1>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_H_strip<true, 0>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  } else {
    if (mode) hipLaunchKernelGGL((k_vjp_H_strip<false, 1>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_H_strip<false, 0>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  }
}
void launch_vjp_theta_strip(int gacc, int itp, int nblk, hipStream_t st, Pools P, const int4* tilesD, ThArgs A) {
  if (A.emitH) {  // emit mode (Y law through its table, yt_fast)
    if (itp) hipLaunchKernelGGL((k_vjp_theta_strip<false, true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_theta_strip<false, false, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    return;
  }
  if (gacc) {
    if (itp) hipLaunchKernelGGL((k_vjp_theta_strip<true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_theta_strip<true, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  } else {
    if (itp) hipLaunchKernelGGL((k_vjp_theta_strip<false, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_theta_strip<false, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  }
}
}  // namespace odinn
