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

#define ODINN_FFAST 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
// Variant of fused_stage for the integer-power A law (LM_FAST), trimmed for the LDS pipe and the
// scalar unit, which co-limit the kernel with the fp64 VALU (LDS array busy 53 %, VALU 57 %, waves
// parked 40 % of their life in s_waitcnt / s_barrier):
//  * lane predicates are selects, not exec-mask branches; only the wave-uniform row test branches;
//  * the thread's own cell {Hc, S} is rebuilt from its registers (max(u,0), B + max(u,0) -- bit for bit
//    what it stored): 2 of 9 ds_read_b128 go;
//  * the four D reads are single ds_read_b64 (2 LDS cycles each); the compiler would pair them into
//    ds_read2_b64, which costs 8;
//  * INNER: the whole halo region lies strictly inside the glacier grid (83 % of the tiles of a 1024^2
//    glacier), every predicate is true and the selects and compare chains vanish.
// Lanes outside region_S compute on stale neighbours; what they produce is never read by a lane
// inside region_{S+1} (a cell of region_{S+1} only touches cells and nodes of region_S), and cells
// outside the glacier keep u = tmp = utilde = 0 because their k is selected to 0.  Valid cells
// execute exactly the expression sequence of fused_stage.
template <int S, int FOYV, bool AF, bool INNER>
__device__ __forceinline__ void fused_stage_fast(const GDev& g, const LawDev& L, const double* __restrict__ Afield,
                                                  int gi, int gj0, int w, int lane, double dt, double2 (*sHS)[FLD],
                                                  double (*sD)[FLD], double (&u)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&tmp)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                                  const double (&up)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&E)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                                  const double (&bb)[(FOYV + 2 * FH + FNW - 1) / FNW]) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + FNW - 1) / FNW;
  const bool nodex = gi >= 0 && gi <= g.nx - 2;
  // ---- nodes of region_S: rows [S-1, FRY-1-S] ---------------------------------------------------
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    if (r >= S - 1 && r <= FRY - 1 - S) {  // wave-uniform
      const int gj = gj0 + r;
      const bool ok = INNER || (nodex && gj >= 0 && gj <= g.ny - 2);
      const double2* p = &sHS[r][lane];
      const double hc0 = vmax0(u[m]);
      double gx, gy, Hb;
      node_geom_vals(g, make_double2(hc0, bb[m] + hc0), p[1], p[FLD], p[FLD + 1], gx, gy, Hb);
      const double gS2 = gx * gx + gy * gy;
      double An = g.A;
      if (AF) An = Afield[g.offd + (ok ? gi + (long long)(g.nx - 1) * gj : 0LL)];
      double al, be, sp;
      const double D = node_D<false, LM_FAST>(g, L, Hb, gS2, An, al, be, sp);
      sD[r][lane] = ok ? D : 0.0;
    }
  }
  __syncthreads();
  // ---- cells of region_S: rows [S, FRY-1-S] ------------------------------------------------------
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  const bool intx = gi >= 1 && gi <= g.nx - 2;
#pragma unroll
  for (int m = 0; m < FSLOT; ++m) {
    const int r = w + FNW * m;
    if (r >= S && r <= FRY - 1 - S) {
      const int gj = gj0 + r;
      const bool interior = INNER || (intx && gj >= 1 && gj <= g.ny - 2);
      const double2* p = &sHS[r][lane];
      typedef const volatile __attribute__((address_space(3))) double* lds_vptr;  // volatile: keep three ds_read_b64
      lds_vptr pD = (lds_vptr)&sD[r][lane];
      const double hc0 = vmax0(u[m]);
      const double Dsw = pD[-FLD - 1], Dse = pD[-FLD], Dnw = pD[-1], Dne = pD[0];
      double k = cell_div_vals<true>(g, make_double2(hc0, bb[m] + hc0), p[1], p[-1], p[FLD], p[-FLD], Dsw, Dse, Dnw, Dne);
      k = interior ? k : 0.0;
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
  __syncthreads();  // every read of sHS of this stage is done
  if (S < 5) {
#pragma unroll
    for (int m = 0; m < FSLOT; ++m) {
      const int r = w + FNW * m;
      if (r >= S && r <= FRY - 1 - S) {
        const double hc = vmax0(u[m]);
        sHS[r][lane] = make_double2(hc, bb[m] + hc);
      }
    }
    __syncthreads();
  }
}

// Inlined-MLP laws (LawY / LawU, Laws.jl:258-265, 114-123): the five stages as a run-time loop with ONE copy of the
// network in the code (the node loop is not unrolled either) -- the stage bodies above, instantiated five times with the
// slot loop unrolled, would inline the 2x16 network 35 times.  A node's D = Y(T, Hbar) Gam Hbar^(nH+2) |grad S|^(nS-1)
// (or Hbar U(Hbar, |grad S|)) is evaluated once per stage from the {Hc, S} tile in LDS exactly as k_rk_stage does;
// the cell phase is fused_stage's with the RDPK3Sp35 coefficients read from the constant tables.
template <int LM, int FOYV>
__device__ __forceinline__ void fused_stages_nn(const GDev& g, const LawDev& L, int gi, int gj0, int w, int lane, double dt,
                                                 double2 (*sHS)[FLD], double (*sD)[FLD], double (&u)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                                 double (&tmp)[(FOYV + 2 * FH + FNW - 1) / FNW], const double (&up)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                                 double (&E)[(FOYV + 2 * FH + FNW - 1) / FNW], const double (&bb)[(FOYV + 2 * FH + FNW - 1) / FNW]) {
  constexpr int FRY = FOYV + 2 * FH, FSLOT = (FRY + FNW - 1) / FNW;
  const bool nodex = gi >= 0 && gi <= g.nx - 2;
  const bool inx = gi >= 0 && gi < g.nx, intx = gi >= 1 && gi <= g.nx - 2;
#pragma unroll 1
  for (int S = 1; S <= 5; ++S) {
    const bool ncol = lane >= S - 1 && lane <= FRX - 1 - S;
#pragma unroll 1
    for (int r = w; r < FRY; r += FNW) {
      if (r >= S - 1 && r <= FRY - 1 - S) {  // wave-uniform
        const int gj = gj0 + r;
        double D = 0.0;
        if (ncol && nodex && gj >= 0 && gj <= g.ny - 2) {
          double gx, gy, Hb;
          node_geom<FLD>(g, &sHS[r][lane], gx, gy, Hb);
          double al, be, sp;
          D = node_D<false, LM>(g, L, Hb, gx * gx + gy * gy, 0.0, al, be, sp);
        }
        if (ncol) sD[r][lane] = D;
      }
    }
    __syncthreads();
    const int s = S - 1;
    const double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
    const bool ccol = lane >= S && lane <= FRX - 1 - S;
#pragma unroll
    for (int m = 0; m < FSLOT; ++m) {
      const int r = w + FNW * m;
      if (r >= S && r <= FRY - 1 - S) {
        const int gj = gj0 + r;
        if (ccol && inx && gj >= 0 && gj < g.ny) {
          double k = 0.0;
          if (intx && gj >= 1 && gj <= g.ny - 2) k = cell_div<FLD, FLD, false>(g, &sHS[r][lane], &sD[r][lane]);
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
}

template <int LM, int FOYV, bool AF, bool INNER>
__device__ __forceinline__ void fused_stages(const GDev& g, const LawDev& L, const double* __restrict__ Afield,
                                              int gi, int gj0, int w, int lane, double dt, double2 (*sHS)[FLD],
                                              double (*sD)[FLD], double (&u)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&tmp)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                              const double (&up)[(FOYV + 2 * FH + FNW - 1) / FNW], double (&E)[(FOYV + 2 * FH + FNW - 1) / FNW],
                                              const double (&bb)[(FOYV + 2 * FH + FNW - 1) / FNW]) {
  if constexpr (lm_is_nn(LM)) {
    fused_stages_nn<LM, FOYV>(g, L, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  } else if constexpr (LM == LM_FAST && ODINN_FFAST) {
    fused_stage_fast<1, FOYV, AF, INNER>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage_fast<2, FOYV, AF, INNER>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage_fast<3, FOYV, AF, INNER>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage_fast<4, FOYV, AF, INNER>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage_fast<5, FOYV, AF, INNER>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  } else {
    fused_stage<1, LM, FOYV>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage<2, LM, FOYV>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage<3, LM, FOYV>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage<4, LM, FOYV>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
    fused_stage<5, LM, FOYV>(g, L, Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  }
}

#define ODINN_FINNER 1  // (fixed: its A/B is recorded above; no longer a build-time knob)
#ifndef ODINN_FWPE
#define ODINN_FWPE 4
#endif
template <int LM, bool SKIP, int FOYV>
__global__ __launch_bounds__(FNT, (lm_is_nn(LM) ? 2 : ODINN_FWPE)) void k_rk_fused(Pools P, LawDev L, const int4* __restrict__ tilesF,
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
#if ODINN_LOG1P_TABLE == 2
  // inlined-MLP laws: log1p's table into LDS with the tile (a barrier follows before the first network evaluation)
  if (lm_is_nn(LM) && threadIdx.x < 65) g_log1p_lds[threadIdx.x] = *reinterpret_cast<const double2*>(&LOG1P_TAB[threadIdx.x][0]);
#endif
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
  const bool inner = gi0 >= 1 && gi0 + FRX - 1 <= g.nx - 2 && gj0 >= 1 && gj0 + FRY - 1 <= g.ny - 2;  // block-uniform
  if (LM == LM_FAST && g.use_Afield)
    fused_stages<LM, FOYV, true, false>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  else if (LM == LM_FAST && ODINN_FINNER && inner)
    fused_stages<LM, FOYV, false, true>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
  else
    fused_stages<LM, FOYV, false, false>(g, L, P.Afield, gi, gj0, w, lane, dt, sHS, sD, u, tmp, up, E, bb);
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

// ================= strip variant (integer-power A law) ==========================================
// Measured on the row-interleaved kernel above: its time is VALU time + LDS time, not their maximum
// (removing the seven ds_read_b128 of a node+cell pair drops 190 us to 140 us; barriers and occupancy
// matter little).  Here wavefront w owns the TRPT contiguous region rows TRPT*w .. TRPT*w+TRPT-1 and a
// thread one column of them, so a cell's neighbourhood is already in the wavefront's registers:
//  * y-neighbours are the same thread's rows ({Hc, S} rebuilt from u and B, bit for bit);
//  * x-neighbours are the adjacent lanes' registers, fetched by DPP wave shifts (v_mov_b32_dpp, 4 per
//    {Hc,S}, 2 per D) -- dearer in VALU slots than an LDS read looks, but an LDS round trip costs a
//    13-cycle ds_write_b128 plus a ds_read_b128 per neighbour on a pipe shared by four SIMDs;
//  * only the first and last row of each strip cross wavefronts: 2 ds_write_b128 + 2 ds_read_b128 per
//    wavefront and stage (was 7 + 49), double-buffered so that a stage needs ONE barrier (was 3);
//  * D never leaves the registers: the wavefront recomputes the node row below its strip itself.
// LDS shrinks to 32 KB.  The previous-step value u_n needed by stages 4, 5 and the error estimate is
// re-read from global memory (L2-resident) instead of being held in 14 VGPRs.
// Every wavefront runs all its rows in every stage; rows and columns outside region_S compute on
// stale neighbours and nothing inside region_{S+1} ever reads them (see fused_stage_fast).
// base[idx] with a block-uniform base and a 32-bit BYTE offset (a glacier has < 2^29 cells), the form the
// scalar-base addressing mode of global_load / global_store needs: no 64-bit address arithmetic per access
__device__ __forceinline__ double ldg32(const double* __restrict__ base, unsigned idx) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (idx << 3));
}
__device__ __forceinline__ void stg32(double* __restrict__ base, unsigned idx, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (idx << 3)) = v;
}
__device__ __forceinline__ double dpp_shift(double x, const bool from_west) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  if (from_west) {  // lane c receives lane c-1's value (wave_shr:1); lane 0 gets 0
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
  } else {          // lane c receives lane c+1's value (wave_shl:1); lane 63 gets 0
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_west(double x) { return dpp_shift(x, true); }
__device__ __forceinline__ double2 dpp_from_west(double2 v) { return make_double2(dpp_shift(v.x, true), dpp_shift(v.y, true)); }
__device__ __forceinline__ double2 dpp_from_east(double2 v) { return make_double2(dpp_shift(v.x, false), dpp_shift(v.y, false)); }
__device__ __forceinline__ double2 cell_HS(double uu, double b) {
  const double hc = vmax0(uu);
  return make_double2(hc, b + hc);
}

// sE[buf][w][0 | 1][lane]: {Hc, S} of the first | last row of wavefront w's strip
typedef double2 (*StripEdges)[TNW][2][FRX];

// Neighbour synchronisation of the strips (ODINN_STRIP_FLAGSYNC): a stage of wavefront w needs the edge rows of wavefronts
// w - 1 and w + 1 only, so instead of a workgroup barrier per stage every wavefront publishes a stage counter in LDS after
// its edge rows and waits for its two neighbours' counters.  (The edge rows are double-buffered: wavefront w overwrites the
// buffer of stage s at the end of stage s + 2, which it can only reach after its neighbours published stage s + 1, i.e.
// after they read the stage-s rows.)
#define ODINN_STRIP_FLAGSYNC 0  // (fixed: its A/B is recorded above; no longer a build-time knob)
__device__ __forceinline__ void strip_flag_sync(volatile int* f, int w, int stage) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) f[w] = stage;
  if (w > 0)
    while (f[w - 1] < stage) __builtin_amdgcn_s_sleep(1);
  if (w + 1 < TNW)
    while (f[w + 1] < stage) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int S, bool AF, int NR, bool UPL, bool SQ, bool YT = false>
__device__ __forceinline__ void strip_stage(const GDev& g, const LawDev& L, const double (*sA)[TNT], const double (*sUp)[TNT],
                                             const double* __restrict__ src, int gic, int gi, int gj0, int w, int lane,
                                             double dtl, StripEdges sE, double (&u)[NR], double (&tmp)[NR],
                                             double (&E)[NR], const double (&bb)[NR], volatile int* sFlag,
                                             [[maybe_unused]] unsigned long long& ovm) {
  // ovm (YT): ballot of the lanes that met a node beyond the law table in stage 1 -- collected WITHOUT a branch, raised by the kernel
  // at its end (ytab_eval_acc)
  // src, Afield: based at the glacier's first cell / dual node (block-uniform), cells addressed by 32-bit indices;
  // gic: gi clamped into the grid; dtl: dt on the lanes with 1 <= gi <= nx-2, else 0
  constexpr int rd = (S - 1) & 1, wr = S & 1;  // stage S reads the edge rows from buffer rd, publishes into wr
  const int r0 = NR * w;
  [[maybe_unused]] const bool nodex = gi >= 0 && gi <= g.nx - 2;
  const double wdx = dtl * g.hinv_dx2, wdy = dtl * g.hinv_dy2;
  constexpr int s = S - 1;
  constexpr double g1 = c_g1[s], g2 = c_g2[s], g3 = c_g3[s], dl = c_dl[s], bt = c_bt[s], bh = c_bh[s];
  // D on node (c, r) = north-east corner of cell (c, r), from the {Hc,S} of cells (c,r) (c+1,r) (c,r+1) (c+1,r+1).
  // No select to 0 on nodes outside the glacier's dual grid: such a node only feeds faces of boundary-ring
  // cells and of cells outside the glacier, and those never move (see dtk below).
  // node_geom_vals + node_D<LM_FAST> with the factor 1/4 of Hbar folded into the constant:
  // A Gam Hbar^5 = (A Gam / 1024) (4 Hbar)^5 bit for bit (powers of two commute with rounding).
  // The bottom edge of a node's cell quartet is the top edge of the previous row's: its x-difference of S and
  // its pair sum of Hc are carried up the sweep (dxb, hpb) instead of being recomputed.
  // |grad S|^2 = (a / 2dx)^2 + (b / 2dy)^2 = (1 / 2dx)^2 (a^2 + (dx/dy)^2 b^2): the factor (1 / 2dx)^2 joins the law
  // constant (folded into the thread's A slots for a gridded A), one multiplication less per node than scaling a and b
  const double Gq = g.Gam * (1.0 / 1024.0) * (g.hinv_dx * g.hinv_dx), ryx = (g.hinv_dy * g.hinv_dy) / (g.hinv_dx * g.hinv_dx);
  const double AGq = g.A * Gq;
  auto node = [&](int slot, double dxb, double hpb, double dxt, double hpt, double dyw, double dye) {
    const double a = dxb + dxt, b = dyw + dye;
    const double H4s = hpb + hpt;  // 4 Hbar
    // SQ (dx == dy on every glacier of the launch): ryx == 1.0 exactly, and (1.0 * b) * b == b * b bit for bit -- one
    // multiplication less per node (2.5 % of the kernel's VALU instructions)
    const double gS2 = SQ ? fma(a, a, b * b) : fma(a, a, (ryx * b) * b);
    const double H2 = H4s * H4s, H4 = H2 * H2;
    if constexpr (YT) {
      // the Y law of target :D_hybrid with n_H = n_gradS = 3 and no sliding (GDev::yt_fast): this law with Y(Hbar) from the glacier's
      // table in A's place (node_D<LM_YTAB>) -- three 16-byte loads from an L2-resident table and a quintic per node
      // Leaving the table's range is reported from stage 1 (every row of the region then holds real data) and from the step's
      // output cells; in stages 2-5 the rows outside region_S hold whatever the stale neighbours produced -- never read, but
      // possibly far outside the table, where the evaluation clamps silently.
      double unused;
      unsigned long long stale = 0ull;
      const double Y = ytab_eval_acc<false>(L.ytab + g.yt_off, L.ytab_ni, S == 1 ? ovm : stale, g.yt_inv_h, 0.25 * H4s, unused);
      return (Y * Gq) * (H4 * H4s) * gS2;
    }
    return (AF ? sA[slot][threadIdx.x] : AGq) * (H4 * H4s) * gS2;  // sA: A Gam / (1024 (2dx)^2) of the thread's own nodes
  };
  // Flux form of cell_div_vals<true>: the flux through the face between two cells is the same number seen
  // from either side (same D sum, same clamped slope -- ties included), so a thread computes only its EAST and
  // NORTH face, takes the west flux from the lane to the west (DPP) and the south flux from its previous row:
  //   F_e(c,r) = (D(c,r-1) + D(c,r)) clamp(S(c+1,r) - S(c,r);  H(c+1,r), -H(c,r))
  //   F_n(c,r) = (D(c-1,r) + D(c,r)) clamp(S(c,r+1) - S(c,r);  H(c,r+1), -H(c,r))
  //   k = (F_e - F_w) / (2 dx^2) + (F_n - F_s) / (2 dy^2)
  // Two clamps and two products per cell instead of four; rounding differs from cell_div_vals only in
  // where the compiler contracts a*b - c*d.
  auto face = [&](double Da, double Db, double slope, double Hhi, double Hlo) { return (Da + Db) * clampn(slope, Hhi, Hlo); };
  // the rows just outside the strip: last row of the wavefront below, first row of the one above (the
  // outermost wavefronts read their own edge instead: rows 0 and (NR * TNW)-1 are never in region_S)
  const double2 hs_s = sE[rd][w > 0 ? w - 1 : 0][w > 0 ? 1 : 0][lane];
  const double2 hs_top = sE[rd][w + 1 < TNW ? w + 1 : w][w + 1 < TNW ? 0 : 1][lane];
  // One sweep up the strip: node row r, then cell row r; the node row and the north faces below the strip first.
  double2 hs_c = cell_HS(u[0], bb[0]);
  double2 e_c = dpp_from_east(hs_c);
  double dx_c = e_c.y - hs_c.y, hp_c = hs_c.x + e_c.x;
  double D_s, F_s;
  {
    const double2 e_s = dpp_from_east(hs_s);
    const double dyw = hs_c.y - hs_s.y;
    D_s = node(0, e_s.y - hs_s.y, hs_s.x + e_s.x, dx_c, hp_c, dyw, e_c.y - e_s.y);
    F_s = face(dpp_from_west(D_s), D_s, dyw, hs_c.x, hs_s.x);
  }
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int gj = gj0 + r0 + m;
    const double2 hs_n = m + 1 < NR ? cell_HS(u[m + 1 < NR ? m + 1 : m], bb[m + 1 < NR ? m + 1 : m]) : hs_top;
    const double2 e_n = dpp_from_east(hs_n);
    const double dx_n = e_n.y - hs_n.y, hp_n = hs_n.x + e_n.x, dyw = hs_n.y - hs_c.y;
    const double D_c = node(m + 1, dx_c, hp_c, dx_n, hp_n, dyw, e_n.y - e_c.y);
    const double F_e = face(D_s, D_c, dx_c, e_c.x, hs_c.x);
    const double F_n = face(dpp_from_west(D_c), D_c, dyw, hs_n.x, hs_c.x);
    const double F_w = dpp_from_west(F_e);
    // dt k with dt folded into the two divergence weights (wdx = dt / 2dx^2, wdy = dt / 2dy^2; 0 on frozen lanes)
    const double dtk = fma(wdx, F_e - F_w, wdy * (F_n - F_s));
    // cells that must not move (boundary ring, outside the glacier): the weights are 0 on their lanes and the weights of
    // dt k are 0 on their rows (wave-uniform, scalar selects) -- fma(0, dt k, x) = x = fma(bt, 0, x) exactly
    const bool rowint = gj >= 1 && gj <= g.ny - 2;
    const double btm = rowint ? bt : 0.0, bhm = rowint ? bh : 0.0;
    const double uo = u[m];
    double un;
    if (S == 1) {
      un = fma(btm, dtk, uo);
      E[m] = bhm * dtk;
    } else {
      const double t = fma(dl, uo, tmp[m]);
      un = fma(g1, uo, g2 * t);
      if (S >= 4) {  // u_n from global memory; what cells outside the glacier pick up is never read (see node)
        const int gjc = gj < 0 ? 0 : (gj > g.ny - 1 ? g.ny - 1 : gj);
        // u_n: from the thread's LDS slots where the launch keeps them (UPL), else re-read from global memory
        un = fma(g3, UPL ? sUp[m][threadIdx.x] : ldg32(src, (unsigned)(gic + g.nx * gjc)), un);
      }
      un = fma(btm, dtk, un);
      if (dl != 0.0) tmp[m] = t;
      E[m] = fma(bhm, dtk, E[m]);
    }
    u[m] = un;
    hs_c = hs_n; e_c = e_n; dx_c = dx_n; hp_c = hp_n; D_s = D_c; F_s = F_n;
    // row fence: the stage body is one basic block and, left alone, the scheduler interleaves all seven rows
    // and spills ~130 VGPRs.  An empty asm that "rewrites" what the row produced and what the next row starts
    // from pins the row order without emitting an instruction.
    if (S == 1)
      asm volatile("" : "+v"(u[m]), "+v"(E[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(e_c.x), "+v"(e_c.y), "+v"(dx_c), "+v"(hp_c), "+v"(D_s), "+v"(F_s));
    else
      asm volatile("" : "+v"(u[m]), "+v"(E[m]), "+v"(tmp[m]), "+v"(hs_c.x), "+v"(hs_c.y), "+v"(e_c.x), "+v"(e_c.y), "+v"(dx_c), "+v"(hp_c), "+v"(D_s), "+v"(F_s));
  }
  if (S < 5) {
    sE[wr][w][0][lane] = cell_HS(u[0], bb[0]);
    sE[wr][w][1][lane] = cell_HS(u[NR - 1], bb[NR - 1]);
    if (ODINN_STRIP_FLAGSYNC) strip_flag_sync(sFlag, w, S);
    else __syncthreads();
  }
}

template <bool AF, int NR, bool UPL, bool SQ, bool YT = false>
__device__ __forceinline__ void strip_stages(const GDev& g, const LawDev& L, const double (*sA)[TNT], const double (*sUp)[TNT],
                                              const double* __restrict__ src, int gic, int gi, int gj0, int w, int lane,
                                              double dtl, StripEdges sE, double (&u)[NR], double (&tmp)[NR],
                                              double (&E)[NR], const double (&bb)[NR], volatile int* sFlag, unsigned long long& ovm) {
  strip_stage<1, AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, dtl, sE, u, tmp, E, bb, sFlag, ovm);
  strip_stage<2, AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, dtl, sE, u, tmp, E, bb, sFlag, ovm);
  strip_stage<3, AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, dtl, sE, u, tmp, E, bb, sFlag, ovm);
  strip_stage<4, AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, dtl, sE, u, tmp, E, bb, sFlag, ovm);
  strip_stage<5, AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, dtl, sE, u, tmp, E, bb, sFlag, ovm);
}

// ---- self-controlled step (SC): no controller / post-step launches ---------------------------------------
// In the solve loop a step was three dependent launches (step kernel, k_controller, k_poststep); the two small
// ones cost ~16 % of a step at 8 x 1024^2 and half of it for small batches -- launch latency, not work.  With
// SC every workgroup of launch n first DECIDES the previous attempt of its glacier itself: wavefront 0 sums the
// glacier's error partials of launch n-1 (same fixed order as k_controller) and runs the PID controller on the
// state in gin (`sc_decide`, the forward branch of k_controller verbatim); all workgroups of a glacier compute
// the same decision from the same numbers, the one that owns tile (0,0) writes the new state to gout (the two
// state arrays and the two partial arrays alternate between launches, so nobody reads what another workgroup of
// the same launch writes).  If the decided step reached a stop, each workgroup stores the snapshot of its own
// output cells from the accepted buffer before stepping on.  A mass balance at a stop is applied ON LOAD to every
// launch that reads the flagged buffer (GState::pad bit 2) until a step is accepted; u_n then lives in per-thread
// LDS slots because the global copy lacks the mass balance.
// GState::pad in SC mode: bit 0 = an attempt awaits its decision, bit 1 = a snapshot awaits being stored, bit 2 = the
// state buffer `cur` still lacks the mass balance of the stop it sits on (applied on load until a step is accepted)
// Called by ALL lanes of wavefront 0 with identical arguments: the three pow() of the PID factor run in lanes 0..2 at
// once (same calls, same product order as k_controller: bit-identical), everything else is computed redundantly.
__device__ __forceinline__ void sc_decide(GState& s, const GDev& g, const CtrlArgs& C, int gidx, double errsum, int& est, int lane) {
  const double h = s.dt;
  const int n_stops = C.nstops[gidx];
  double fac = 1.0;
  bool accept = true;
  if (C.adaptive) {
    double EEst = sqrt(errsum / ((double)g.nx * (double)g.ny));
    if (!(EEst == EEst) || isinf(EEst)) { s.nonfinite = 1; EEst = 1e300; }
    if (EEst < 2.220446049250313e-16) EEst = 2.220446049250313e-16;
    s.EEst = EEst;
    const double e1 = 1.0 / EEst;
    const double pw = pow(lane == 0 ? e1 : (lane == 1 ? s.e2 : s.e3), lane == 0 ? 0.64 / 3.0 : (lane == 1 ? -0.31 / 3.0 : 0.04 / 3.0));
    fac = __shfl(pw, 0, 64) * __shfl(pw, 1, 64) * __shfl(pw, 2, 64);
    fac = 1.0 + atan(fac - 1.0);
    accept = fac >= 0.81;
    if (accept) { s.e3 = s.e2; s.e2 = e1; }
  }
  double t = s.t;
  const double t0_ = s.t;
  if (C.trace && gidx == C.trace_g && lane == 0) {  // (diagnostics, ODINN_TRACE_STEPS: every workgroup of the traced glacier writes the same four numbers)
    const long long q = s.naccept + s.nreject;
    if (q < C.trace_cap) { C.trace[4 * q] = t; C.trace[4 * q + 1] = h; C.trace[4 * q + 2] = s.EEst; C.trace[4 * q + 3] = accept ? fac : -fac; }
  }
  s.at_stop = 0;
  s.mb_now = 0;
  if (accept) {
    s.naccept++;
    s.accepted = 1;
    s.cur = 1 - s.cur;
    if (s.clipped) {
      t = C.tstop(s.istop, gidx);
      s.at_stop = 1;
      s.mb_now = C.at(C.mb_flag, s.istop, gidx);
      s.mb_slot = C.at(C.mb_slot, s.istop, gidx);
      s.snap_slot = C.snap_slot ? C.at(C.snap_slot, s.istop, gidx) : s.istop;
      s.istop++;
    } else {
      t += h;
    }
    s.t = t;
  } else {
    s.nreject++;
    s.accepted = 0;
  }
  if (C.adaptive) {  // a stuck solve: see controller_decide
    if (accept && s.t != t0_) s.pad2 = 0;
    else if (++s.pad2 >= STALL_MAX && !C.stuck_off) {
      if (!s.nonfinite) s.nonfinite = 2;
      s.done = 1;
      est = 0;
      return;
    }
  }
  if (s.istop >= n_stops) {
    s.done = 1;
    est = 0;
    return;
  }
  double dtn = C.adaptive ? h * fac : C.fixed_dt;
  if (C.dtmax > 0.0 && dtn > C.dtmax) dtn = C.dtmax;
  const double rem = C.tstop(s.istop, gidx) - t;
  if (dtn >= rem || fabs(rem - dtn) <= 100.0 * 2.220446049250313e-16 * fabs(t)) {
    dtn = rem;
    s.clipped = 1;
  } else {
    s.clipped = 0;
  }
  s.dt = dtn;
  const double e = ceil((C.tstop(n_stops - 1, gidx) - t) / (C.adaptive ? h * fac : dtn));
  const int stops_left = n_stops - s.istop;
  est = e < (double)stops_left ? stops_left : (e > 1e6 ? 1000000 : (int)e);
}

// AF: A from the dual-grid field.  One stage path per kernel: with two paths in one kernel the register allocator
// spills (a separate predicate-free kernel for the tiles strictly inside the grid was measured and lost: its
// second launch costs more than the selects it saves).
// NR: rows per thread (7: 54x46 output tiles; 8: 54x54 tiles, less halo work, for batches that fill the GPU twice over)
// YT: the Y law through its table (every glacier yt_fast; replaces the scalar A -- see strip_stage's node)
template <bool SKIP, bool AF, int NR, bool SC = false, bool SQ = false, bool YT = false>
__global__ __launch_bounds__(TNT, ODINN_FWPE) void k_rk_fused_strip(Pools P, LawDev L, const int4* __restrict__ tilesF,
                                                                    double* __restrict__ U0, double* __restrict__ U1,
                                                                    double* __restrict__ partF, double abstol, double reltol,
                                                                    ScArgs A) {
  __shared__ double2 sE[2][TNW][2][FRX];
  __shared__ double red[TNW];
  __shared__ int sFlagS[TNW];
  volatile int* sFlag = sFlagS;
  if ((threadIdx.x & 63) == 0) sFlagS[threadIdx.x >> 6] = 0;  // (a workgroup barrier follows before the first stage)
  __shared__ GState s_state;
  // gridded A: the thread's own nodes (rows r0-1 .. r0+NR-1 of its column) in private LDS slots -- read in each of the
  // five stages, fetched from global memory once (0 on nodes outside the dual grid: they only feed frozen cells)
  __shared__ double sA[AF ? NR + 1 : 1][TNT];
  // constant-A path: u_n of the thread's cells is kept in private LDS slots -- it is needed again in stages 4, 5 and by the
  // error estimate (three L2 reads per cell otherwise: -2..3 % per step), and after a mass balance applied on load (SC) the
  // global copy is not u_n any more.  With a gridded A the slots of A take that LDS.
  constexpr bool UPL = !AF;
  __shared__ double sUp[UPL ? NR : 1][TNT];
  const int4 t4 = tilesF[blockIdx.x];
  const GDev g = P.gd[t4.x];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * FOX - FH, gj0 = t4.z * (NR * TNW - 2 * FH) - FH;
  const int gi = gi0 + lane, r0 = NR * w;
  const bool inx = gi >= 0 && gi < g.nx;
  const int id0 = gi + g.nx * (gj0 + r0);
  const double* __restrict__ Bg = P.B + g.off;
  double bb[NR];  // B does not depend on the controller's decision: in SC mode its loads fly while wavefront 0 decides
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int gj = gj0 + r0 + m;
    bb[m] = (inx && gj >= 0 && gj < g.ny) ? ldg32(Bg, (unsigned)(id0 + g.nx * m)) : 0.0;
  }
  // gridded A: like B it does not depend on the state -- its loads are issued here, ahead of the (dependent) state
  // loads, and parked in the LDS slots once u is on its way (they used to start after the first barrier: PMC showed
  // the gridded-A kernel parked 44 % of its wave time against 26 % for constant A)
  [[maybe_unused]] double aa[AF ? NR + 1 : 1];
  if (AF) {
    const double* __restrict__ Afg = P.Afield + g.offd;
    const bool nodex = gi >= 0 && gi <= g.nx - 2;
#pragma unroll
    for (int m = 0; m <= NR; ++m) {
      const int gj = gj0 + r0 - 1 + m;
      const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
      aa[AF ? m : 0] = ok ? ldg32(Afg, (unsigned)(gi + (g.nx - 1) * gj)) : 0.0;
    }
  }
  double dt;
  int cur, snap_slot = -1;  // snap_slot >= 0: store the snapshot of the stop the decided step reached
  bool finished, mb_pend = false;
  int mb_slot = 0;
  if (SC) {
    if (w == 0) {
      GState sn = A.gin[t4.x];
      int est = -1;
      bool newly_done = false;
      if (!sn.done && (sn.pad & 1)) {
        const int t0 = NR == 8 ? g.tile0Fu : g.tile0Ft, nt = NR == 8 ? g.ntilesFu : g.ntilesFt;
        double sum = 0.0;
        for (int k = lane; k < nt; k += 64) sum += A.part_in[t0 + k];
        sum = __shfl(wave_sum(sum), 0, 64);
        int mbp = sn.pad & 4;
        sc_decide(sn, g, A.C, t4.x, sum, est, lane);
        if (sn.accepted) mbp = 0;                          // the accepted step replaced the buffer
        if (sn.at_stop && sn.mb_now && g.has_mb) mbp = 4;  // ... and landed on a stop with a mass balance
        sn.pad = (sn.at_stop ? 2 : 0) | mbp;
        newly_done = sn.done != 0;
      } else if (!sn.done) {
        sn.pad &= 4;
      }
      if (lane == 0) {
        s_state = sn;  // what this launch acts on
        if (t4.y == 0 && t4.z == 0) {  // the glacier's designated workgroup publishes the state for the next launch
          GState so = sn;
          // its snapshot (if any) is stored below; an attempt follows unless done; a done glacier's MB is written back below
          so.pad = sn.done ? 0 : (1 | (sn.pad & 4));
          A.gout[t4.x] = so;
          if (newly_done) atomicSub(A.C.n_active, 1);
          if (A.C.est_steps && est >= 0) A.C.est_steps[t4.x] = est;
        }
      }
    }
    __syncthreads();
    dt = s_state.dt;
    cur = s_state.cur;
    finished = s_state.done != 0;
    if (s_state.pad & 2) snap_slot = s_state.snap_slot;
    mb_pend = (s_state.pad & 4) != 0;
    mb_slot = s_state.mb_slot;
    if (finished && snap_slot < 0) return;
  } else {
    const GState* gs = P.gs + t4.x;
    // snapshot on load (ScArgs::snap_on_load): the step the controller just accepted landed on a stop -- this launch
    // loads exactly that state, so every workgroup stores its own output cells of it (no post-step launch).  A glacier
    // that has just finished does only that (the controller clears at_stop of a finished glacier at its next call).
    const bool snap = A.snap_on_load && gs->at_stop;
    if (gs->done && !snap) return;
    dt = gs->dt;
    cur = gs->cur;
    finished = gs->done != 0;
    if (snap) snap_slot = gs->snap_slot;
    // ... and with a mass balance (constant-A kernels: u_n in LDS slots) the controller flags the buffer that still lacks the
    // mass balance of its stop (GState::pad bit 2); it is applied on load below, exactly as in the self-controlled loop
    if (UPL && A.snap_on_load && A.mb0) {
      mb_pend = (gs->pad & 4) != 0;
      mb_slot = gs->mb_slot;
    }
  }
  // everything below is addressed relative to the glacier's first cell (block-uniform bases, 32-bit cell indices)
  const double* __restrict__ src = (cur ? U1 : U0) + g.off;
  double* __restrict__ dst = (cur ? U0 : U1) + g.off;
  double u[NR], tmp[NR], E[NR];
  bool nz = false;
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int gj = gj0 + r0 + m;
    double h = 0.0;
    if (inx && gj >= 0 && gj < g.ny) h = ldg32(src, (unsigned)(id0 + g.nx * m));
    u[m] = h; tmp[m] = h; E[m] = 0.0;
    nz = nz || (h != 0.0);
  }
  if (UPL && mb_pend) {
    // the buffer sits on a stop whose mass balance has not been applied to it (k_poststep's arithmetic, VJPs.jl:129-139):
    // apply it to everything loaded; on the first launch after the stop keep the pre-MB state of the own cells
    const double* __restrict__ mb0 = A.mb0 + g.off;
    const double* __restrict__ Sr = A.Sref ? A.Sref + g.off : nullptr;
    double* __restrict__ pm = A.premb + (long long)mb_slot * A.ntot + g.off;
    nz = false;
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const int r = r0 + m, gj = gj0 + r;
      if (inx && gj >= 0 && gj < g.ny) {
        const unsigned id = (unsigned)(id0 + g.nx * m);
        double h = u[m];
        if (snap_slot >= 0 && r >= FH && r <= (NR * TNW) - 1 - FH && lane >= FH && lane < FH + FOX) stg32(pm, id, h);
        double dmb;
        double mb = mb_value(g, ldg32(mb0, id), Sr ? ldg32(Sr, id) : 0.0, h, bb[m], dmb);
        const bool mask = (h > 0.0 && mb < 0.0) || (h > 10.0 && mb >= 0.0);
        if (!mask) mb = 0.0;
        if (mask && h + mb < 0.0) mb = -h;
        h += mb;
        u[m] = h; tmp[m] = h;
      }
      nz = nz || (u[m] != 0.0);
    }
  }
  if (UPL) {
#pragma unroll
    for (int m = 0; m < NR; ++m) sUp[UPL ? m : 0][threadIdx.x] = u[m];
  }
  if (snap_slot >= 0) {  // snapshot of the stop just reached: this workgroup's output cells of the accepted state
    double* __restrict__ sn = A.snaps + (long long)snap_slot * A.ntot + g.off;
    if (lane >= FH && lane < FH + FOX && inx) {
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        const int r = r0 + m, gj = gj0 + r;
        if (r >= FH && r <= (NR * TNW) - 1 - FH && gj < g.ny) {
          stg32(sn, (unsigned)(id0 + g.nx * m), u[m]);
          if (finished && mb_pend) stg32(const_cast<double*>(src), (unsigned)(id0 + g.nx * m), u[m]);  // the final state carries its MB
        }
      }
    }
    if (finished) return;
  }
  sE[0][w][0][lane] = cell_HS(u[0], bb[0]);
  sE[0][w][1][lane] = cell_HS(u[NR - 1], bb[NR - 1]);
  const bool ocol = lane >= FH && lane < FH + FOX && inx;
  if (SKIP) {
    // exact shortcut, see k_rk_fused
    if (!__syncthreads_or(nz)) {
      if (ocol) {
#pragma unroll
        for (int m = 0; m < NR; ++m) {
          const int r = r0 + m, gj = gj0 + r;
          if (r >= FH && r <= (NR * TNW) - 1 - FH && gj < g.ny) stg32(dst, (unsigned)(id0 + g.nx * m), 0.0);
        }
      }
      if (threadIdx.x == 0) partF[t4.w] = 0.0;
      return;
    }
  } else {
    __syncthreads();
  }
  const int gic = gi < 0 ? 0 : (gi > g.nx - 1 ? g.nx - 1 : gi);
  if (AF) {
    const double Gq = g.Gam * (1.0 / 1024.0) * (g.hinv_dx * g.hinv_dx);  // as in strip_stage
#pragma unroll
    for (int m = 0; m <= NR; ++m) sA[AF ? m : 0][threadIdx.x] = aa[AF ? m : 0] * Gq;
  }
  static_assert(!YT || !AF, "the table replaces the scalar A");
  unsigned long long ovm = 0ull;  // (YT) lanes that left the law table: see strip_stage
  strip_stages<AF, NR, UPL, SQ, YT>(g, L, sA, sUp, src, gic, gi, gj0, w, lane, gi >= 1 && gi <= g.nx - 2 ? dt : 0.0, sE, u, tmp, E, bb, sFlag, ovm);
  // ---- output rows [FH, (NR * TNW)-1-FH]: u' from the registers, embedded error partial -----------------------
  double errsq = 0.0;
  double upf[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) {  // all loads in flight before the first use
    const int r = r0 + m, gj = gj0 + r;
    const bool out = r >= FH && r <= (NR * TNW) - 1 - FH && ocol && gj < g.ny;
    upf[m] = UPL ? sUp[UPL ? m : 0][threadIdx.x] : ldg32(src, (unsigned)(out ? id0 + g.nx * m : 0));
  }
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int r = r0 + m, gj = gj0 + r;
    if (r >= FH && r <= (NR * TNW) - 1 - FH && ocol && gj < g.ny) {
      const double upv = upf[m];
      stg32(dst, (unsigned)(id0 + g.nx * m), u[m]);
      if constexpr (YT) ovm |= __builtin_amdgcn_ballot_w64(!(u[m] * g.yt_inv_h < (double)L.ytab_ni));  // (Hbar <= the largest of its four cells)
      const double err = (u[m] - upv) - E[m];
      const double sk = abstol + fmax(fabs(upv), fabs(u[m])) * reltol;
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
    for (int k = 0; k < TNW; ++k) sum += red[k];
    partF[t4.w] = sum;
  }
}

// ================= RHS-only strip kernel (integer-power A law) ==================================
// dH/dt = Huginn.SIA2D! (adjoint.jl:52-97) in the strip layout of k_rk_fused_strip: a wavefront owns DNR contiguous
// rows of a 64-wide region, a thread one column of them; y-neighbours are the thread's own registers, x-neighbours
// arrive by DPP, only the first and last row of a strip cross wavefronts (one LDS exchange, one barrier).  One RHS needs
// a one-cell halo: 62 x 62 outputs per 64 x 64 region (1.07 x redundant work).  Against k_dhdt (64 x 16 LDS tiles,
// {Hc,S} and D through LDS) the point is memory-level parallelism, not arithmetic: every thread has its 16 (24 with a
// gridded A) loads in flight before the first use and the workgroup needs 16 KB of LDS instead of 28, so the kernel
// sits closer to the HBM rate (PMC on k_dhdt past the Infinity Cache: 77 % of the wave time parked, 4.45 TB/s).
// Same expressions per face and node as strip_stage (flux form); results equal k_dhdt's to rounding.
constexpr int DNR = 8;                                   // rows per thread
constexpr int DOX = FRX - 2, DOY = DNR * TNW - 2;        // 62 x 62 output tile
// EULER: the north star's "CFL mode" (scheme 3, k_euler_cfl's contract): dH holds u' = u + dt k with the glacier's dt
// from its GState, and partD[tile] = max D over the tile's nodes (wavefront max -> LDS -> one partial per tile, reduced
// in a fixed order by the controller, which sets the next dt = cfl min(dx,dy)^2 / (4 max D)).
template <bool AF, bool SKIP, bool EULER = false>
__global__ __launch_bounds__(TNT, ODINN_FWPE) void k_dhdt_strip(Pools P, const int4* __restrict__ tilesD,
                                                                const double* __restrict__ U, double* __restrict__ dH,
                                                                double* __restrict__ partD = nullptr) {
  __shared__ double2 sE[TNW][2][FRX];
  __shared__ double redD[TNW];
  const int4 t4 = tilesD[blockIdx.x];
  double dt = 0.0;
  if (EULER) {
    const GState* gs = P.gs + t4.x;
    if (gs->done) return;
    dt = gs->dt;
  }
  const GDev g = P.gd[t4.x];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi0 = t4.y * DOX - 1, gj0 = t4.z * DOY - 1;
  const int gi = gi0 + lane, r0 = DNR * w;
  const bool inx = gi >= 0 && gi < g.nx;
  const int id0 = gi + g.nx * (gj0 + r0);
  const double* __restrict__ src = U + g.off;
  const double* __restrict__ Bg = P.B + g.off;
  double* __restrict__ dst = dH + g.off;
  double u[DNR], bb[DNR];
  [[maybe_unused]] double aa[AF ? DNR + 1 : 1];
#pragma unroll
  for (int m = 0; m < DNR; ++m) {
    const int gj = gj0 + r0 + m;
    const bool ok = inx && gj >= 0 && gj < g.ny;
    bb[m] = ok ? ldg32(Bg, (unsigned)(id0 + g.nx * m)) : 0.0;
    u[m] = ok ? ldg32(src, (unsigned)(id0 + g.nx * m)) : 0.0;
  }
  const double Gq = g.Gam * (1.0 / 1024.0) * (g.hinv_dx * g.hinv_dx), ryx = (g.hinv_dy * g.hinv_dy) / (g.hinv_dx * g.hinv_dx);
  if (AF) {  // A on the node rows r0-1 .. r0+DNR-1 of the thread's column (0 outside the dual grid: such nodes only feed masked cells)
    const double* __restrict__ Afg = P.Afield + g.offd;
    const bool nodex = gi >= 0 && gi <= g.nx - 2;
#pragma unroll
    for (int m = 0; m <= DNR; ++m) {
      const int gj = gj0 + r0 - 1 + m;
      const bool ok = nodex && gj >= 0 && gj <= g.ny - 2;
      aa[AF ? m : 0] = ok ? ldg32(Afg, (unsigned)(gi + (g.nx - 1) * gj)) : 0.0;
    }
  }
  bool nz = false;
#pragma unroll
  for (int m = 0; m < DNR; ++m) nz = nz || (u[m] > 0.0);
  sE[w][0][lane] = cell_HS(u[0], bb[0]);
  sE[w][1][lane] = cell_HS(u[DNR - 1], bb[DNR - 1]);
  const bool ocol = lane >= 1 && lane <= DOX && inx;
  const bool intx = gi >= 1 && gi <= g.nx - 2;
  if (SKIP) {
    // no ice on the whole region: every clamped slope and D vanish, dH/dt = 0 exactly (k_dhdt's shortcut)
    if (!__syncthreads_or(nz)) {
      if (ocol) {
#pragma unroll
        for (int m = 0; m < DNR; ++m) {
          const int r = r0 + m, gj = gj0 + r;
          if (r >= 1 && r <= DOY && gj < g.ny) stg32(dst, (unsigned)(id0 + g.nx * m), EULER ? u[m] : 0.0);
        }
      }
      if (EULER && threadIdx.x == 0) partD[t4.w] = 0.0;
      return;
    }
  } else {
    __syncthreads();
  }
  const double AGq = g.A * Gq;
  auto node = [&](int slot, double dxb, double hpb, double dxt, double hpt, double dyw, double dye) {
    const double a = dxb + dxt, b = dyw + dye;
    const double H4s = hpb + hpt;  // 4 Hbar
    const double gS2 = fma(a, a, (ryx * b) * b);
    const double H2 = H4s * H4s, H4 = H2 * H2;
    return (AF ? aa[AF ? slot : 0] * Gq : AGq) * (H4 * H4s) * gS2;
  };
  auto face = [&](double Da, double Db, double slope, double Hhi, double Hlo) { return (Da + Db) * clampn(slope, Hhi, Hlo); };
  const double2 hs_s = sE[w > 0 ? w - 1 : 0][w > 0 ? 1 : 0][lane];
  const double2 hs_top = sE[w + 1 < TNW ? w + 1 : w][w + 1 < TNW ? 0 : 1][lane];
  double2 hs_c = cell_HS(u[0], bb[0]);
  double2 e_c = dpp_from_east(hs_c);
  double dx_c = e_c.y - hs_c.y, hp_c = hs_c.x + e_c.x;
  double D_s, F_s;
  [[maybe_unused]] double dmax = 0.0;
  {
    const double2 e_s = dpp_from_east(hs_s);
    const double dyw = hs_c.y - hs_s.y;
    D_s = node(0, e_s.y - hs_s.y, hs_s.x + e_s.x, dx_c, hp_c, dyw, e_c.y - e_s.y);
    F_s = face(dpp_from_west(D_s), D_s, dyw, hs_c.x, hs_s.x);
  }
#pragma unroll
  for (int m = 0; m < DNR; ++m) {
    const int r = r0 + m, gj = gj0 + r;
    const double2 hs_n = m + 1 < DNR ? cell_HS(u[m + 1 < DNR ? m + 1 : m], bb[m + 1 < DNR ? m + 1 : m]) : hs_top;
    const double2 e_n = dpp_from_east(hs_n);
    const double dx_n = e_n.y - hs_n.y, hp_n = hs_n.x + e_n.x, dyw = hs_n.y - hs_c.y;
    const double D_c = node(m + 1, dx_c, hp_c, dx_n, hp_n, dyw, e_n.y - e_c.y);
    const double F_e = face(D_s, D_c, dx_c, e_c.x, hs_c.x);
    const double F_n = face(dpp_from_west(D_c), D_c, dyw, hs_n.x, hs_c.x);
    const double F_w = dpp_from_west(F_e);
    double k = fma(g.hinv_dx2, F_e - F_w, g.hinv_dy2 * (F_n - F_s));
    k = (intx && gj >= 1 && gj <= g.ny - 2) ? k : 0.0;  // zero on the boundary ring (adjoint.jl:52-97: dH only on the interior)
    const bool outc = ocol && r >= 1 && r <= DOY && gj < g.ny;
    if (outc) stg32(dst, (unsigned)(id0 + g.nx * m), EULER ? fma(dt, k, u[m]) : k);
    // the node north-east of an output cell, where it exists (every dual node belongs to exactly one output cell)
    if (EULER && outc && gi <= g.nx - 2 && gj <= g.ny - 2) dmax = fmax(dmax, D_c);
    hs_c = hs_n; e_c = e_n; dx_c = dx_n; hp_c = hp_n; D_s = D_c; F_s = F_n;
    asm volatile("" : "+v"(hs_c.x), "+v"(hs_c.y), "+v"(e_c.x), "+v"(e_c.y), "+v"(dx_c), "+v"(hp_c), "+v"(D_s), "+v"(F_s));
  }
  if (EULER) {
    dmax = wave_max(dmax);
    if (lane == 0) redD[w] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < TNW; ++k) mx = fmax(mx, redD[k]);
      partD[t4.w] = mx;
    }
  }
}

}  // namespace odinn
