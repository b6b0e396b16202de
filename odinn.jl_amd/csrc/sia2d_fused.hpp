// sia2d_fused.hpp -- one whole RDPK3Sp35 step in ONE kernel (temporal fusion of the 5 stages).
//
// The per-stage path (k_rk_stage) is HBM-bound: 264 B/cell per step.  Here a workgroup owns a
// 64x32 output tile, loads its (64+10)x(32+10) halo region of u and B ONCE, keeps Hc = max(u,0)
// and S = B+Hc of the region in LDS and the per-cell 3S*+ registers (u, tmp, uprev, utilde, B)
// in VGPRs of the owning thread, and runs the five stages on regions shrinking by one ring per
// stage.  HBM traffic drops to ~24 B/cell per step (read u,B; write u'); the price is ~1.3x
// redundant stencil work in the halo, which the otherwise idle fp64 VALU absorbs.  A rejected
// step needs no uprev copy: the kernel reads U[cur] and writes U[1-cur]; the controller flips
// `cur` only on acceptance.  Arithmetic per cell is the same expression sequence as k_rk_stage.
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

// meta[m]: bits 0-7 ring index of the cell inside the region (distance to the region border,
// 0 for slots past the region), bit 8: cell inside the glacier grid, bit 9: interior cell,
// bit 10: the node north-east of the cell exists in the glacier's dual grid,
// bits 16-23: node ring = min(c+1, r+1, FRX-1-c, FRY-1-r) (the node is needed by stage S iff >= S).
template <int S, int LM>
__device__ __forceinline__ void fused_stage(const GDev& g, const LawDev& L, const double* __restrict__ Afield, int gi0,
                                            int gj0, double dt, double2* sHS, double* sD,
                                            const int (&off)[FCPT], const int (&meta)[FCPT], double (&u)[FCPT],
                                            double (&tmp)[FCPT], const double (&up)[FCPT], double (&E)[FCPT],
                                            const double (&bb)[FCPT]) {
  // ---- nodes needed by region_S: a in [S-1, FRX-S), b in [S-1, FRY-S).  Each thread evaluates
  //      the node north-east of each of its own cells: no index arithmetic in the loop ---------
#pragma unroll
  for (int m = 0; m < FCPT; ++m) {
    if (((meta[m] >> 16) & 0xff) >= S) {
      double D = 0.0;
      if (meta[m] & 0x400) {
        double gx, gy, Hb;
        node_geom<FLD>(g, sHS + off[m], gx, gy, Hb);
        const double gS2 = gx * gx + gy * gy;
        double An = g.A;
        if (g.use_Afield) {
          const int idx = threadIdx.x + FNT * m;
          const int r = idx / FRX, c = idx - r * FRX;
          An = Afield[g.offd + (gi0 + c) + (long long)(g.nx - 1) * (gj0 + r)];
        }
        double al, be, sp;
        D = node_D<false, LM>(g, L, Hb, gS2, An, al, be, sp);
      }
      sD[off[m]] = D;
    }
  }
  __syncthreads();
  // ---- cells of region_S owned by this thread -----------------------------------------------
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
#pragma unroll
  for (int m = 0; m < FCPT; ++m) {
    if ((meta[m] & 0xff) >= S && (meta[m] & 0x100)) {
      double k = 0.0;
      if (meta[m] & 0x200) k = cell_div<FLD, FLD, LM == LM_FAST>(g, sHS + off[m], sD + off[m]);
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
  __syncthreads();  // every read of sH/sS of this stage is done
  if (S < 5) {
#pragma unroll
    for (int m = 0; m < FCPT; ++m) {
      if ((meta[m] & 0xff) >= S) {
        const double hc = u[m] > 0.0 ? u[m] : 0.0;
        sHS[off[m]] = make_double2(hc, bb[m] + hc);
      }
    }
    __syncthreads();
  }
}

template <int LM, bool SKIP>
__global__ __launch_bounds__(FNT, (FNT == 512 ? 4 : 4)) void k_rk_fused(Pools P, LawDev L, const int4* __restrict__ tilesF,
                                                  double* __restrict__ U0, double* __restrict__ U1,
                                                  double* __restrict__ partF, double abstol, double reltol) {
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
  const int gi0 = t4.y * FOX - FH, gj0 = t4.z * FOY - FH;
  double u[FCPT], tmp[FCPT], up[FCPT], E[FCPT], bb[FCPT];
  int off[FCPT], meta[FCPT];
  double2* pHS = &sHS[0][0];
  double* pD = &sD[0][0];
#pragma unroll
  for (int m = 0; m < FCPT; ++m) {
    const int idx = threadIdx.x + FNT * m;
    const int r = idx / FRX, c = idx - r * FRX;
    double h = 0.0, b = 0.0;
    int mt = 0;
    off[m] = r * FLD + c;
    if (idx < FNC) {
      const int gi = gi0 + c, gj = gj0 + r;
      int ring = min(min(c, FRX - 1 - c), min(r, FRY - 1 - r));
      if (gi >= 0 && gi < g.nx && gj >= 0 && gj < g.ny) {
        const long long id = g.off + gi + (long long)g.nx * gj;
        h = src[id];
        b = P.B[id];
        mt = 0x100;
        if (gi >= 1 && gi <= g.nx - 2 && gj >= 1 && gj <= g.ny - 2) mt |= 0x200;
      }
      mt |= ring;
      const int nring = min(min(c + 1, FRX - 1 - c), min(r + 1, FRY - 1 - r));
      mt |= nring << 16;
      if (gi >= 0 && gi <= g.nx - 2 && gj >= 0 && gj <= g.ny - 2) mt |= 0x400;
      const double hc = h > 0.0 ? h : 0.0;
      pHS[off[m]] = make_double2(hc, b + hc);
    } else {
      off[m] = 0;
    }
    meta[m] = mt;
    u[m] = h; tmp[m] = h; up[m] = h; E[m] = 0.0; bb[m] = b;
  }
  if (SKIP) {
    // Exact shortcut: if u == 0 on the whole halo region every clamped slope, D and k vanish in all
    // five stages, so u' = 0 and the error estimate is 0 -- bit-identical to running the stages.
    bool nz = false;
#pragma unroll
    for (int m = 0; m < FCPT; ++m) nz = nz || (u[m] != 0.0);
    if (!__syncthreads_or(nz)) {
      const int tx = threadIdx.x & 63, wy = threadIdx.x >> 6;
      const int gi = gi0 + FH + tx;
#pragma unroll
      for (int rr = wy; rr < FOY; rr += FNW) {
        const int gj = gj0 + FH + rr;
        if (gi < g.nx && gj < g.ny) dst[g.off + gi + (long long)g.nx * gj] = 0.0;
      }
      if (threadIdx.x == 0) partF[t4.w] = 0.0;
      return;
    }
  } else {
    __syncthreads();
  }
  fused_stage<1, LM>(g, L, P.Afield, gi0, gj0, dt, pHS, pD, off, meta, u, tmp, up, E, bb);
  fused_stage<2, LM>(g, L, P.Afield, gi0, gj0, dt, pHS, pD, off, meta, u, tmp, up, E, bb);
  fused_stage<3, LM>(g, L, P.Afield, gi0, gj0, dt, pHS, pD, off, meta, u, tmp, up, E, bb);
  fused_stage<4, LM>(g, L, P.Afield, gi0, gj0, dt, pHS, pD, off, meta, u, tmp, up, E, bb);
  fused_stage<5, LM>(g, L, P.Afield, gi0, gj0, dt, pHS, pD, off, meta, u, tmp, up, E, bb);
  // ---- output tile = region_5: embedded error partial; u' is staged through LDS so that the
  //      global stores are full, aligned 512-B rows (the region mapping is 74 wide) ------------
  double errsq = 0.0;
  __syncthreads();  // stage 5 is done with sD
#pragma unroll
  for (int m = 0; m < FCPT; ++m) {
    if ((meta[m] & 0xff) >= FH) {
      pD[off[m]] = u[m];
      if (meta[m] & 0x100) {
        const double err = (u[m] - up[m]) - E[m];
        const double sk = abstol + fmax(fabs(up[m]), fabs(u[m])) * reltol;
        const double q = err / sk;
        errsq = fma(q, q, errsq);
      }
    }
  }
  __syncthreads();
  {
    const int tx = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int gi = gi0 + FH + tx;
#pragma unroll
    for (int rr = wy; rr < FOY; rr += FNW) {
      const int gj = gj0 + FH + rr;
      if (gi < g.nx && gj < g.ny) dst[g.off + gi + (long long)g.nx * gj] = sD[FH + rr][FH + tx];
    }
  }
  // deterministic block sum over 8 wavefronts
  errsq = wave_sum(errsq);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = errsq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < FNW; ++k) s += red[k];
    partF[t4.w] = s;
  }
}

}  // namespace odinn
