// sia2d_fused.hpp -- one whole RDPK3Sp35 step in ONE kernel (temporal fusion of the 5 stages).
//
// The per-stage path (k_rk_stage) is HBM-bound: 264 B/cell per step.  Here a workgroup owns a
// 54 x FOY output tile, loads its 64 x (FOY+10) halo region of u and B ONCE, keeps Hc = max(u,0) and
// S = B+Hc of the region in LDS and the per-cell 3S*+ registers (u, tmp, uprev, utilde, B) in VGPRs
// of the owning thread, and runs the five stages on regions shrinking by one ring per stage.  HBM
// traffic drops to ~24 B/cell per step (read u,B; write u'); the price is ~1.5x redundant stencil
// work in the halo.  A rejected step needs no uprev copy: the kernel reads U[cur] and writes
// U[1-cur]; the controller flips `cur` only on acceptance.  Arithmetic per cell is the same
// expression sequence as k_rk_stage.
//
// Work decomposition: the region is exactly 64 columns wide, so wavefront w owns the region rows
// w, w+8, w+16, ...; the row index (and with it every row predicate and the LDS row address) is
// wave-uniform -> SALU, the column is the lane -> ring / grid-edge tests are one compare on the
// lane id, no per-cell offset or flag words (the first version mapped a 74-wide region row-major
// onto the threads and spent 56 % of its VALU slots on predicates, addresses and moves).
// u' is stored straight from registers (row segments of 54 doubles).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

template <int S, int LM, int FOYV>
__device__ __forceinline__ void fused_stage(const GDev& g, const LawDev& L, const double* __restrict__ Afield,
                                             int gi, int gj0, int w, int lane, double dt, double2 (*sHS)[FLD],
                                             double (*sD)[FLD], double (&u)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&tmp)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                             const double (&up)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&E)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                            const double (&bb)[(FOYV + 2 * FH + FNW - 1) / FNW]) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + FNW - 1) / FNW;

  // ---- nodes needed by region_S: columns [S-1, 63-S], rows [S-1, FRY-1-S]; node (c, r) = north-east
  //      corner of cell (c, r) ------------------------------------------------------------------
  const bool ncol = lane >= S - 1 && lane <= FRX - 1 - S;
  const bool nodex = gi >= 0 && gi <= g.nx - 2;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    if (r >= S - 1 && r <= FRY - 1 - S) {  // wave-uniform
      const int gj = gj0 + r;
      if (ncol) {
        double D = 0.0;
        if (nodex && gj >= 0 && gj <= g.ny - 2) {
          double gx, gy, Hb;
          node_geom<FLD>(g, &sHS[r][lane], gx, gy, Hb);
          const double gS2 = gx * gx + gy * gy;
          double An = g.A;
          if (g.use_Afield) An = Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
          double al, be, sp;
          D = node_D<false, LM>(g, L, Hb, gS2, An, al, be, sp);
        }
        sD[r][lane] = D;
      }
    }
  }
  __syncthreads();
  // ---- cells of region_S: columns [S, 63-S], rows [S, FRY-1-S] ----------------------------------
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  const bool ccol = lane >= S && lane <= FRX - 1 - S;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    if (r >= S && r <= FRY - 1 - S) {
      const int gj = gj0 + r;
      if (ccol && inx && gj >= 0 && gj < g.ny) {
        double k = 0.0;
        if (intx && gj >= 1 && gj <= g.ny - 2) k = cell_div<FLD, FLD, LM == LM_FAST>(g, &sHS[r][lane], &sD[r][lane]);
        const double dtk = dt * k;
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
  __syncthreads();  // every read of sHS of this stage is done
  if (S < 5) {
#pragma unroll
    for (int m = 0; m < FSLOT; ++m) {
      const int r = w + FNW * m;
      if (r >= S && r <= FRY - 1 - S && ccol) {
        const double hc = vmax0(u[m]);
        sHS[r][lane] = make_double2(hc, bb[m] + hc);
      }
    }
    __syncthreads();
  }
}

#ifndef ODINN_FWPE
#define ODINN_FWPE 4
#endif
template <int LM, bool SKIP, int FOYV>
__global__ __launch_bounds__(FNT, ODINN_FWPE) void k_rk_fused(Pools P, LawDev L, const int4* __restrict__ tilesF,
                                                     double* __restrict__ U0, double* __restrict__ U1,
                                                     double* __restrict__ partF, double abstol, double reltol) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + FNW - 1) / FNW;
  __shared__ double2 sHS[FRY][FLD];
  __shared__ double sD[FRY][FLD];
  __shared__ double red[FNW];
  const int4 t4 = tilesF[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  if (gs->done) return;
  const GDev g = P.gd[t4.x];
  const double dt = gs->dt;
  const double* __restrict__ src = gs->cur ? U1 : U0;
  double* __restrict__ dst = gs->cur ? U0 : U1;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * FOX - FH, gj0 = t4.z * FOYV - FH;
  const int gi = gi0 + lane;
  const bool inx = gi >= 0 && gi < g.nx;
  double u[FSLOT], tmp[FSLOT], up[FSLOT], E[FSLOT], bb[FSLOT];
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    double h = 0.0, b = 0.0;
    if (r < FRY) {
      const int gj = gj0 + r;
      if (inx && gj >= 0 && gj < g.ny) {
        const long long id = g.off + gi + (long long)g.nx * gj;
        h = src[id];
        b = P.B[id];
      }
      const double hc = vmax0(h);
      sHS[r][lane] = make_double2(hc, b + hc);
    }
    u[m] = h; tmp[m] = h; up[m] = h; E[m] = 0.0; bb[m] = b;
  }
  const int gjo = gj0 + FH;  // first output row
  if (SKIP) {
    // Exact shortcut: if u == 0 on the whole halo region every clamped slope, D and k vanish in all
    // five stages, so u' = 0 and the error estimate is 0 -- bit-identical to running the stages.
    bool nz = false;
#pragma unroll
    for (int m = 0; m < FSLOT; ++m) nz = nz || (u[m] != 0.0);
    if (!__syncthreads_or(nz)) {
      if (lane >= FH && lane < FH + FOX && inx) {
        for (int rr = w; rr < FOYV; rr += FNW) {
          const int gj = gjo + rr;
          if (gj < g.ny) dst[g.off + gi + (long long)g.nx * gj] = 0.0;
        }
      }
      if (threadIdx.x == 0) partF[t4.w] = 0.0;
      return;
    }
  } else {
    __syncthreads();
  }
  fused_stage<1, LM, FOYV>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  fused_stage<2, LM, FOYV>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  fused_stage<3, LM, FOYV>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  fused_stage<4, LM, FOYV>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  fused_stage<5, LM, FOYV>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  // ---- output tile = region_5 (columns [5, 58], rows [5, FRY-6]): u' straight from the registers,
  //      embedded error partial ------------------------------------------------------------------
  double errsq = 0.0;
  const bool ocol = lane >= FH && lane < FH + FOX && inx;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    if (r >= FH && r <= FRY - 1 - FH && ocol) {
      const int gj = gj0 + r;
      if (gj < g.ny) {
        dst[g.off + gi + (long long)g.nx * gj] = u[m];
        const double err = (u[m] - up[m]) - E[m];
        const double sk = abstol + fmax(fabs(up[m]), fabs(u[m])) * reltol;
        const double q = err / sk;
        errsq = fma(q, q, errsq);
      }
    }
  }
  // deterministic block sum over the wavefronts
  errsq = wave_sum(errsq);
  if (lane == 0) red[w] = errsq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < FNW; ++k) s += red[k];
    partF[t4.w] = s;
  }
}

}  // namespace odinn
