// adjf_dispatch.hpp -- instantiation tree of k_adj_fused_strip, shared by k_adjf.hip (SC = false) and k_adjfs.hip (the
// self-controlled instantiations, SC = true: interleaved snapshot pairs only), so that the two sets compile side by side
#pragma once
#include <cstdlib>
#include "launch.hpp"
#include "sia2d_adj_fused.hpp"
namespace odinn {
#define ODINN_ADJF(AF, SK, SG, NR, GA, YT) \
  hipLaunchKernelGGL((k_adj_fused_strip<AF, SK, SG, NR, GA, YT, SC>), dim3(nblk), dim3(TNT), pad, st, P, A)
template <bool SC>
static void adjf_dispatch(int nblk, int afield, int skip, int rows, hipStream_t st, const Pools& P, const AdjFusedArgs& A) {
  constexpr unsigned pad = 0u;  // (dynamic LDS; the one-workgroup-per-CU experiment it once served is recorded in DESIGN section 5)
  if (A.utab) {  // the U law (target :D) through its table: own law block, no SC (the caller guarantees A.segs, !afield, rows 2 / 4 / 7)
    if constexpr (!SC) {
#define ODINN_ADJF_UT(SK, NR) \
  hipLaunchKernelGGL((k_adj_fused_strip<false, SK, true, NR, false, false, false, true>), dim3(nblk), dim3(TNT), pad, st, P, A)
      if (rows == 2) { if (skip) ODINN_ADJF_UT(true, 2); else ODINN_ADJF_UT(false, 2); }
      else if (rows == 4) { if (skip) ODINN_ADJF_UT(true, 4); else ODINN_ADJF_UT(false, 4); }
      else { if (skip) ODINN_ADJF_UT(true, TRPT); else ODINN_ADJF_UT(false, TRPT); }
#undef ODINN_ADJF_UT
    }
    return;
  }
  if (rows == 2) {  // the smallest batches: 2 rows per thread (54 x 6 output tiles); the caller guarantees A.segs
    if (A.ytab) {
      if (skip) ODINN_ADJF(false, true, true, 2, false, true); else ODINN_ADJF(false, false, true, 2, false, true);
    } else if (A.Gacc) {
      if (skip) ODINN_ADJF(true, true, true, 2, true, false); else ODINN_ADJF(true, false, true, 2, true, false);
    } else if (afield) {
      if (skip) ODINN_ADJF(true, true, true, 2, false, false); else ODINN_ADJF(true, false, true, 2, false, false);
    } else {
      if (skip) ODINN_ADJF(false, true, true, 2, false, false); else ODINN_ADJF(false, false, true, 2, false, false);
    }
    return;
  }
  if (A.ytab) {  // the Y law through its table (the caller guarantees !afield, no th_part / Gacc, rows 2, 4 or 7)
    if (rows == 4) {
      if (skip) ODINN_ADJF(false, true, true, 4, false, true); else ODINN_ADJF(false, false, true, 4, false, true);
    } else if (A.segs) {
      if (skip) ODINN_ADJF(false, true, true, TRPT, false, true); else ODINN_ADJF(false, false, true, TRPT, false, true);
    } else if constexpr (!SC) {
      if (skip) ODINN_ADJF(false, true, false, TRPT, false, true); else ODINN_ADJF(false, false, false, TRPT, false, true);
    }
    return;
  }
  if (rows == 8) {  // gridded A, register-cached (ODINN_ADJ_RC), the forward kernel's 54 x 54 tiles; the caller guarantees afield and A.segs
    if (A.Gacc) {
      if (skip) ODINN_ADJF(true, true, true, 8, true, false); else ODINN_ADJF(true, false, true, 8, true, false);
    } else {
      if (skip) ODINN_ADJF(true, true, true, 8, false, false); else ODINN_ADJF(true, false, true, 8, false, false);
    }
    return;
  }
  if (A.Gacc) {  // gridded A with the dual-grid accumulator fed by stage 1 (the caller guarantees afield, A.segs, A.th_part)
    if (rows == 4) {
      if (skip) ODINN_ADJF(true, true, true, 4, true, false); else ODINN_ADJF(true, false, true, 4, true, false);
    } else {
      if (skip) ODINN_ADJF(true, true, true, TRPT, true, false); else ODINN_ADJF(true, false, true, TRPT, true, false);
    }
    return;
  }
  if (rows == 4) {  // small batches: 4 rows per thread (54 x 22 output tiles); the caller guarantees A.segs
    if (afield) {
      if (skip) ODINN_ADJF(true, true, true, 4, false, false); else ODINN_ADJF(true, false, true, 4, false, false);
    } else {
      if (skip) ODINN_ADJF(false, true, true, 4, false, false); else ODINN_ADJF(false, false, true, 4, false, false);
    }
    return;
  }
  if (A.segs) {  // interleaved {H_j, dH} pairs
    if (afield) {
      if (skip) ODINN_ADJF(true, true, true, TRPT, false, false); else ODINN_ADJF(true, false, true, TRPT, false, false);
    } else {
      if (skip) ODINN_ADJF(false, true, true, TRPT, false, false); else ODINN_ADJF(false, false, true, TRPT, false, false);
    }
    return;
  }
  if constexpr (!SC) {
    if (afield) {
      if (skip) ODINN_ADJF(true, true, false, TRPT, false, false); else ODINN_ADJF(true, false, false, TRPT, false, false);
    } else {
      if (skip) ODINN_ADJF(false, true, false, TRPT, false, false); else ODINN_ADJF(false, false, false, TRPT, false, false);
    }
  }
}
#undef ODINN_ADJF
}  // namespace odinn
