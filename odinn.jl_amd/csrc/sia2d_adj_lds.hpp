// sia2d_adj_lds.hpp -- one whole REVERSE RDPK3Sp35 step of the continuous adjoint in ONE kernel, on LDS tiles, for the laws whose node
// does not fit the register strips of k_adj_fused_strip: the U law of target :D through its bi-quintic table (LM_UTAB).
//
// The per-stage path (k_adj_stage<S, LM>) moves 72 B/cell per stage -- 360 B/cell per reverse step -- and its time is HBM time plus
// VALU issue time (docs/HISTORY.md, round 5 item 9).  The strip form of the fused step (k_adj_fused_strip<..., UT>) lost by 3 x: the node's
// 36 coefficients, two quintic collapses and two Taylor shifts do not fit beside seven rows of state.  Here a workgroup owns a
// 54 x FOYV output tile like the forward k_rk_fused<LM>: its 64 x (FOYV + 10) halo region of {H_j, dH}, B and lambda is loaded ONCE
// (40 B/cell), every thread keeps the 3S*+ registers of its cells, and each stage
//   1. writes {Hc, S} of H_itp(tau_stage) and the interior-masked lambda of the region to LDS,
//   2. evaluates ONE dual node per thread at a time (vjpH_node<LM>: law first, then the four corner contributions) into LDS,
//   3. adds the four corner contributions of every cell, masked by the cell's own H > 0 (adjoint.jl:148), and advances lambda,
// on regions that shrink by one ring per stage (average redundancy 1.28 on 54 x 22 tiles).  Same expressions per node and cell as
// k_adj_stage<S, LM> / vjpH_tile: results equal to the staged solve to rounding (tests/test_gpu_law_table_U.py).
// Reference: SIA2D_adjoint! with the DiscreteVJP (src/inverse/SIA2D/gradient.jl:276-539, adjoint.jl:31-151), target :D
// (src/models/target/target_D_pure.jl:78-137).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

template <int S, int LM, int FOYV, int NWV>
__device__ __forceinline__ void adj_lds_stage(const GDev& g, const LawDev& L, const Pools& P, int gi, int gi0, int gj0, int w, int lane,
                                               double dt, double sw, double2 (*sHS)[FLD], double (*sL)[FLD], double2 (*sCa)[FLD],
                                               double2 (*sCb)[FLD], double (&u)[(FOYV + 2 * FH + NWV - 1) / NWV],
                                               double (&tmp)[(FOYV + 2 * FH + NWV - 1) / NWV], const double (&up)[(FOYV + 2 * FH + NWV - 1) / NWV],
                                               double (&E)[(FOYV + 2 * FH + NWV - 1) / NWV], const double (&ha)[(FOYV + 2 * FH + NWV - 1) / NWV],
                                               const double (&dh)[(FOYV + 2 * FH + NWV - 1) / NWV], const double (&bb)[(FOYV + 2 * FH + NWV - 1) / NWV],
                                               const UtabTile ut) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + NWV - 1) / NWV;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
  // ---- 1. the stage's tiles: H_itp(tau_S) = H_j + s_S (H_j+1 - H_j) (gradient.jl:287: linear in the forward snapshots), lambda masked
  //         to the interior (adjoint.jl:52-97 acts on inn(lambda)); cells outside the grid hold zeros
  double hcS[FSLOT];
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + NWV * m;
    hcS[m] = 0.0;
    if (r < FRY) {
      const int gj = gj0 + r;
      const double h = fma(sw, dh[m], ha[m]);
      const double hc = vmax0(h);
      hcS[m] = hc;
      sHS[r][lane] = make_double2(hc, bb[m] + hc);
      sL[r][lane] = (intx && gj >= 1 && gj <= g.ny - 2) ? u[m] : 0.0;
    }
  }
  __syncthreads();
  // ---- 2. nodes needed by region_S: columns [S-1, 63-S], rows [S-1, FRY-1-S]; node (c, r) = north-east corner of cell (c, r)
  //         (measured and rejected: TWO nodes of the thread's column inside one divergent region, so that the scheduler can interleave the two
  //          bi-quintic evaluations -- 256 VGPRs with 13-37 spilled, both nodes evaluated wherever one of them has ice: 1.46 against 1.32 ms
  //          at 16 x 1024^2, 5.18 against 4.65 ms at 64 x 1024^2 with the shortcut)
  //         (measured and rejected: the stage's (65 - 2 S) x (FRY + 1 - 2 S) nodes dealt FLAT over the 768 threads -- 3, 3, 3, 2, 2 rounds where
  //          whole rows need 3 each; 168 VGPRs with 5-8 spilled, waves that straddle two rows: 1.23 against 1.18 ms, 4.27 against 4.16 ms)
  const bool ncol = lane >= S - 1 && lane <= FRX - 1 - S;
#pragma unroll 1
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + NWV * m;
    if (r >= S - 1 && r <= FRY - 1 - S) {  // wave-uniform
      double k[4] = {0.0, 0.0, 0.0, 0.0};
      if (ncol) vjpH_node<LM, 0, FLD>(g, L, P, sHS, sL, gi0 + 1, gj0 + 1, lane, r, k, ut);
      sCa[r][lane] = make_double2(k[0], k[1]);   // {SW, SE}
      sCb[r][lane] = make_double2(k[2], k[3]);   // {NW, NE}
    }
  }
  __syncthreads();
  // ---- 3. cells of region_S: columns [S, 63-S], rows [S, FRY-1-S]
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  const bool ccol = lane >= S && lane <= FRX - 1 - S;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + NWV * m;
    if (r >= S && r <= FRY - 1 - S) {
      const int gj = gj0 + r;
      if (ccol && inx && gj >= 0 && gj < g.ny) {
        double v = 0.0;
        if (hcS[m] > 0.0)  // dlam .* (H .> 0)  (adjoint.jl:148)
          v = (sCb[r - 1][lane - 1].y + sCb[r - 1][lane].x) + (sCa[r][lane - 1].y + sCa[r][lane].x);
        const double dtk = dt * v;
        const double uo = u[m];
        double un;
        if (S == 1) {
          un = fma(bt, dtk, uo);
          E[m] = bh * dtk;
        } else {
          const double t = fma(dl, uo, tmp[m]);
          un = fma(g1, uo, g2 * t);
          if (S >= 4) un = fma(g3, up[m], un);
          un = fma(bt, dtk, un);
          if (dl != 0.0) tmp[m] = t;
          E[m] = fma(bh, dtk, E[m]);
        }
        u[m] = un;
      }
    }
  }
  // (the next stage's tile writes touch sHS / sL only, whose last readers -- the node pass -- are behind the barrier above; its node
  //  pass overwrites sCa / sCb behind its own first barrier, which every cell-pass read of this stage precedes)
}

// FOYV = 22: the 54 x 22 tile table Fv of the strip kernels' 4-row form, 117 KB of tiles + 36 KB of table = one workgroup per CU.
// NWV waves per workgroup, rows dealt round-robin.  Measured at 8 / 12 / 16 waves (184 / 167 / 128 VGPRs, the last with 49 spilled):
// 1.31 / 1.19 / 1.48 ms per reverse step at 16 x 1024^2, 4.62 / 4.19 / 5.33 ms at 64 x 1024^2 with the shortcut -- three waves per
// SIMD hide more of the node's dependent fp64 chains than two, and 32 rows over 12 waves leave three slots of state per thread, not four.
constexpr int LNW = 12;
template <int LM, bool SKIP, int FOYV, int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void k_adj_fused_lds(Pools P, LawDev L, AdjFusedArgs A) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + NWV - 1) / NWV;
  __shared__ double2 sHS[FRY][FLD];
  __shared__ double sL[FRY][FLD];
  __shared__ double2 sCa[FRY][FLD];
  __shared__ double2 sCb[FRY][FLD];
  __shared__ double red[NWV];
  // the U law's WHOLE table where it is the coarsest level (16 x 8 bi-quintic patches, 36 KB: what ytab_refresh picks for a law as smooth as
  // the reference's scaled LawU) -- the LDS the tiles leave free on a CU that holds one workgroup anyway; every patch gather of the
  // five stages is then an LDS read (the vector L1 returns data in order: a table load that hits still queues behind the misses)
  constexpr int UT_LDS_PATCHES = LM == LM_UTAB ? 128 : 0;
  // (19 double2 per patch, not 18: neighbouring patches along Hbar are 8 patches = 8 x 72 words = a multiple of the 64 banks apart at 18,
  //  so that a wave whose nodes straddle two of them -- the common case -- reads every coefficient twice)
  constexpr int UT_PST = 19;
  __shared__ double2 sTab[UT_LDS_PATCHES > 0 ? UT_PST * UT_LDS_PATCHES : 1];
  const int4 t4 = A.tilesF[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (gs->done) return;
  const GDev g = P.gd[t4.x];
  const AdjState a = A.adj[t4.x];
  UtabTile ut = ODINN_UT_NONE;
  if constexpr (LM == LM_UTAB) {
    const int np = L.utab_nh * L.utab_ns;
    if (np <= UT_LDS_PATCHES && !L.ut_nolds) {  // (block-uniform; published by the first barrier of stage 1)
      const double2* __restrict__ tg = reinterpret_cast<const double2*>(L.utab);
      for (int k = threadIdx.x; k < 18 * np; k += 64 * NWV) sTab[k + k / 18] = tg[k];
      ut.lds = sTab; ut.ih0 = 0; ut.is0 = 0; ut.nsr = L.utab_ns; ut.pst = UT_PST;
    }
  }
  const double dt = gs->dt;
  const int cur = gs->cur;
  const double* __restrict__ src = (cur ? A.lam1 : A.lam0) + g.off;   // the step reads lam[cur] and writes lam[1 - cur]: the controller
  double* __restrict__ dst = (cur ? A.lam0 : A.lam1) + g.off;         // flips cur on acceptance, a rejected step needs no copy
  const double2* __restrict__ seg = A.segs + (long long)a.seg * A.ntot + g.off;  // {H_j, H_j+1 - H_j} of the segment of the coming step
  const double* __restrict__ Bg = P.B + g.off;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * FOX - FH, gj0 = t4.z * FOYV - FH;
  const int gi = gi0 + lane;
  const bool inx = gi >= 0 && gi < g.nx;
  double u[FSLOT], tmp[FSLOT], up[FSLOT], E[FSLOT], ha[FSLOT], dh[FSLOT], bb[FSLOT];
  bool ice = false;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + NWV * m, gj = gj0 + r;
    double l = 0.0, h0 = 0.0, d0 = 0.0, b = 0.0;
    if (r < FRY && inx && gj >= 0 && gj < g.ny) {
      const unsigned id = (unsigned)(gi + g.nx * gj);
      l = src[id];
      const double2 hd = seg[id];
      h0 = hd.x; d0 = hd.y;
      b = Bg[id];
    }
    u[m] = l; tmp[m] = l; up[m] = l; E[m] = 0.0; ha[m] = h0; dh[m] = d0; bb[m] = b;
    ice = ice || h0 > 0.0 || (h0 + d0) > 0.0;
  }
  const bool ocol = lane >= FH && lane < FH + FOX && inx;
  bool run = true;
  if (SKIP) {
    // Exact shortcut (k_adj_fused_strip's): neither bracketing snapshot has ice anywhere on the halo region and the five stage
    // weights lie in [0, 1] -- Hc = 0 on the region at every stage, J_H^T lam = +0 on all cells: the step is the 3S*+ update of
    // lambda with a zero right-hand side, which is what the stages would compute bit for bit.
#pragma unroll
    for (int k = 0; k < 5; ++k) ice = ice || !(a.sitp[k] >= 0.0 && a.sitp[k] <= 1.0);
    run = __syncthreads_or(ice);
    if (!run) {
#pragma unroll
      for (int m = 0; m < FSLOT; ++m) {
        double un = fma(c_bt[0], 0.0, up[m]), tm = up[m];
#pragma unroll
        for (int sg = 1; sg < 5; ++sg) {
          const double uo = un;
          const double t = fma(c_dl[sg], uo, tm);
          un = fma(c_g1[sg], uo, c_g2[sg] * t);
          if (sg >= 3) un = fma(c_g3[sg], up[m], un);
          un = fma(c_bt[sg], 0.0, un);
          if (c_dl[sg] != 0.0) tm = t;
        }
        u[m] = un;
      }
    }
  }
  if (run) {
    adj_lds_stage<1, LM, FOYV, NWV>(g, L, P, gi, gi0, gj0, w, lane, dt, a.sitp[0], sHS, sL, sCa, sCb, u, tmp, up, E, ha, dh, bb, ut);
    adj_lds_stage<2, LM, FOYV, NWV>(g, L, P, gi, gi0, gj0, w, lane, dt, a.sitp[1], sHS, sL, sCa, sCb, u, tmp, up, E, ha, dh, bb, ut);
    adj_lds_stage<3, LM, FOYV, NWV>(g, L, P, gi, gi0, gj0, w, lane, dt, a.sitp[2], sHS, sL, sCa, sCb, u, tmp, up, E, ha, dh, bb, ut);
    adj_lds_stage<4, LM, FOYV, NWV>(g, L, P, gi, gi0, gj0, w, lane, dt, a.sitp[3], sHS, sL, sCa, sCb, u, tmp, up, E, ha, dh, bb, ut);
    adj_lds_stage<5, LM, FOYV, NWV>(g, L, P, gi, gi0, gj0, w, lane, dt, a.sitp[4], sHS, sL, sCa, sCb, u, tmp, up, E, ha, dh, bb, ut);
  }
  // ---- output tile = region_5: lambda' from the registers, embedded error partial (k_adj_stage<5>'s expression)
  double errsq = 0.0;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + NWV * m, gj = gj0 + r;
    if (r >= FH && r <= FRY - 1 - FH && ocol && gj < g.ny) {
      dst[(unsigned)(gi + g.nx * gj)] = u[m];
      const double err = (u[m] - up[m]) - E[m];
      const double sk = A.abstol + fmax(fabs(up[m]), fabs(u[m])) * A.reltol;
      const double q = err / sk;
      errsq = fma(q, q, errsq);
    }
  }
  errsq = wave_sum(errsq);
  if (lane == 0) red[w] = errsq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < NWV; ++k) sum += red[k];
    A.partF[t4.w] = sum;
  }
}

}  // namespace odinn
