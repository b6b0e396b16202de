// k_adjf.hip -- temporally fused reverse step of the continuous adjoint (integer-power A law, DiscreteVJP)
#include <cstdlib>
#include "launch.hpp"
#include "sia2d_adj_fused.hpp"
namespace odinn {
void launch_adj_fused_strip(int nblk, int afield, int skip, int rows, hipStream_t st, Pools P, AdjFusedArgs A) {
  // measurement aid: ODINN_ADJ_LDS_PAD=<bytes> of unused dynamic LDS per workgroup (> 2 KB: one workgroup per CU instead of two,
  // i.e. half the per-XCD working set against the 4 MB L2 at half the occupancy)
  static const unsigned pad = std::getenv("ODINN_ADJ_LDS_PAD") ? (unsigned)std::atoi(std::getenv("ODINN_ADJ_LDS_PAD")) : 0u;
  if (A.ytab) {  // the Y law through its table (the caller guarantees !afield, no th_part / Gacc, rows 4 or 7)
    if (rows == 4) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true, true, 4, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<false, false, true, 4, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else if (A.segs) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true, true, TRPT, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<false, false, true, TRPT, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true, false, TRPT, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<false, false, false, TRPT, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    }
    return;
  }
  if (rows == 8) {  // gridded A, register-cached (ODINN_ADJ_RC), the forward kernel's 54 x 54 tiles; the caller guarantees afield and A.segs
    if (A.Gacc) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true, 8, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true, 8, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true, 8>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true, 8>), dim3(nblk), dim3(TNT), pad, st, P, A);
    }
    return;
  }
  if (A.Gacc) {  // gridded A with the dual-grid accumulator fed by stage 1 (the caller guarantees afield, A.segs, A.th_part)
    if (rows == 4) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true, 4, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true, 4, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true, TRPT, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true, TRPT, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    }
    return;
  }
  if (rows == 4) {  // small batches: 4 rows per thread (54 x 22 output tiles); the caller guarantees A.segs
    if (afield) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true, 4>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true, 4>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true, true, 4>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<false, false, true, 4>), dim3(nblk), dim3(TNT), pad, st, P, A);
    }
    return;
  }
  if (A.segs) {  // interleaved {H_j, dH} pairs
    if (afield) {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<true, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    } else {
      if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
      else hipLaunchKernelGGL((k_adj_fused_strip<false, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    }
    return;
  }
  if (afield) {
    if (skip) hipLaunchKernelGGL((k_adj_fused_strip<true, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    else hipLaunchKernelGGL((k_adj_fused_strip<true, false>), dim3(nblk), dim3(TNT), pad, st, P, A);
  } else {
    if (skip) hipLaunchKernelGGL((k_adj_fused_strip<false, true>), dim3(nblk), dim3(TNT), pad, st, P, A);
    else hipLaunchKernelGGL((k_adj_fused_strip<false, false>), dim3(nblk), dim3(TNT), pad, st, P, A);
  }
}
void launch_vjp_H_strip(int mode, int afield, int nblk, hipStream_t st, Pools P, const int4* tilesD, AdjArgs A) {
  if (afield) {
    if (mode) hipLaunchKernelGGL((k_vjp_H_strip<true, 1>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_H_strip<true, 0>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  } else {
    if (mode) hipLaunchKernelGGL((k_vjp_H_strip<false, 1>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_H_strip<false, 0>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  }
}
void launch_vjp_theta_strip(int gacc, int itp, int nblk, hipStream_t st, Pools P, const int4* tilesD, ThArgs A) {
  if (gacc) {
    if (itp) hipLaunchKernelGGL((k_vjp_theta_strip<true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_theta_strip<true, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  } else {
    if (itp) hipLaunchKernelGGL((k_vjp_theta_strip<false, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
    else hipLaunchKernelGGL((k_vjp_theta_strip<false, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, A);
  }
}
}  // namespace odinn
