// k_fused.hip -- temporally fused RDPK3Sp35 step for one law mode (-DODINN_LM=0 ... 8)
#include <cstdlib>
// inlined-MLP laws: log1p's table lives in LDS in these kernels (mode 2: filled with the tile; see sia2d_device.hpp)
#if defined(ODINN_LM) && ODINN_LM >= 2 && ODINN_LM <= 6 && !defined(ODINN_LOG1P_TABLE)
#define ODINN_LOG1P_TABLE 2
#endif
#include "launch.hpp"
#include "sia2d_fused.hpp"
#ifndef ODINN_LM
#error "define ODINN_LM"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
namespace odinn {
void CAT(launch_rk_fused_lm, ODINN_LM)(int nblk, hipStream_t st, Pools P, LawDev L, const int4* tilesF, double* U0,
                                        double* U1, double* partF, double abstol, double reltol, int skip, int small) {
  // small: 1 = FOX x FOYS "latency" tiles (tilesF / partF then belong to that table)
  // inlined-MLP laws: no ice-free shortcut variant (the network is skipped per node where Hbar = 0 anyway, and the
  // conditional stencil makes the register allocator keep the whole network live: 95-166 spilled VGPRs)
  if (lm_is_nn(ODINN_LM)) skip = 0;
  if (small) {
    if (skip) hipLaunchKernelGGL((k_rk_fused<ODINN_LM, true, FOYS>), dim3(nblk), dim3(FNT), 0, st, P, L, tilesF, U0, U1, partF, abstol, reltol);
    else hipLaunchKernelGGL((k_rk_fused<ODINN_LM, false, FOYS>), dim3(nblk), dim3(FNT), 0, st, P, L, tilesF, U0, U1, partF, abstol, reltol);
  } else {
    if (skip) hipLaunchKernelGGL((k_rk_fused<ODINN_LM, true, FOY>), dim3(nblk), dim3(FNT), 0, st, P, L, tilesF, U0, U1, partF, abstol, reltol);
    else hipLaunchKernelGGL((k_rk_fused<ODINN_LM, false, FOY>), dim3(nblk), dim3(FNT), 0, st, P, L, tilesF, U0, U1, partF, abstol, reltol);
  }
}
#if ODINN_LM == 0
// strip kernel (integer-power law) on the FOX x FOYT (rows = 7) or FOX x FOYT8 (rows = 8) tile table; sc != null:
// self-controlled step (no controller / post-step launches, see ScArgs)
void launch_rk_fused_strip(int nblk, int afield, int rows, hipStream_t st, Pools P, LawDev L, const int4* tilesF, double* U0,
                           double* U1, double* partF, double abstol, double reltol, int skip, const ScArgs* sc, int sq, int ytab) {
  const ScArgs A = sc ? *sc : ScArgs{};
  constexpr unsigned pad = 0u;  // (dynamic LDS; the occupancy A/B it once served is recorded in DESIGN section 5)
  // sq: dx == dy on every glacier of the batch (one multiplication less per node, bit-identical)
#define ODINN_STRIP_Q(SK, AF, NR, SCV, SQV) \
  hipLaunchKernelGGL((k_rk_fused_strip<SK, AF, NR, SCV, SQV>), dim3(nblk), dim3(TNT), pad, st, P, L, tilesF, U0, U1, partF, abstol, reltol, A)
#define ODINN_STRIP(SK, AF, NR, SCV) \
  do { if (sq) ODINN_STRIP_Q(SK, AF, NR, SCV, true); else ODINN_STRIP_Q(SK, AF, NR, SCV, false); } while (0)
#define ODINN_STRIP_S(SK, AF, NR) \
  do { if (sc && !sc->snap_on_load) ODINN_STRIP(SK, AF, NR, true); else ODINN_STRIP(SK, AF, NR, false); } while (0)
#define ODINN_STRIP_R(SK, AF) \
  do { if (rows == 8) ODINN_STRIP_S(SK, AF, 8); else ODINN_STRIP_S(SK, AF, TRPT); } while (0)
  if (ytab) {  // the Y law through its table (the caller guarantees !afield and GDev::yt_fast on every glacier)
#define ODINN_STRIP_YQ(SK, NR, SCV, SQV) \
  hipLaunchKernelGGL((k_rk_fused_strip<SK, false, NR, SCV, SQV, true>), dim3(nblk), dim3(TNT), pad, st, P, L, tilesF, U0, U1, partF, abstol, reltol, A)
#define ODINN_STRIP_Y(SK, NR, SCV) \
  do { if (sq) ODINN_STRIP_YQ(SK, NR, SCV, true); else ODINN_STRIP_YQ(SK, NR, SCV, false); } while (0)
#define ODINN_STRIP_YS(SK, NR) \
  do { if (sc && !sc->snap_on_load) ODINN_STRIP_Y(SK, NR, true); else ODINN_STRIP_Y(SK, NR, false); } while (0)
    if (rows == 8) { if (skip) ODINN_STRIP_YS(true, 8); else ODINN_STRIP_YS(false, 8); }
    else { if (skip) ODINN_STRIP_YS(true, TRPT); else ODINN_STRIP_YS(false, TRPT); }
#undef ODINN_STRIP_YS
#undef ODINN_STRIP_Y
#undef ODINN_STRIP_YQ
  } else if (afield) {
    if (skip) ODINN_STRIP_R(true, true); else ODINN_STRIP_R(false, true);
  } else {
    if (skip) ODINN_STRIP_R(true, false); else ODINN_STRIP_R(false, false);
  }
#undef ODINN_STRIP_R
#undef ODINN_STRIP_S
#undef ODINN_STRIP
#undef ODINN_STRIP_Q
}
// RHS-only strip kernel on the DOX x DOY tile table
void launch_dhdt_strip(int nblk, int afield, int skip, hipStream_t st, Pools P, const int4* tilesD, const double* U, double* dH) {
  if (afield) {
    if (skip) hipLaunchKernelGGL((k_dhdt_strip<true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, U, dH);
    else hipLaunchKernelGGL((k_dhdt_strip<true, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, U, dH);
  } else {
    if (skip) hipLaunchKernelGGL((k_dhdt_strip<false, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, U, dH);
    else hipLaunchKernelGGL((k_dhdt_strip<false, false>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, U, dH);
  }
}
// CFL-limited explicit Euler step (scheme 3) in the same layout: dst = u + dt k, partD[tile] = max D
void launch_euler_cfl_strip(int nblk, int afield, hipStream_t st, Pools P, const int4* tilesD, const double* src, double* dst, double* partD) {
  if (afield) hipLaunchKernelGGL((k_dhdt_strip<true, true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, src, dst, partD);
  else hipLaunchKernelGGL((k_dhdt_strip<false, true, true>), dim3(nblk), dim3(TNT), 0, st, P, tilesD, src, dst, partD);
}
#endif
}  // namespace odinn