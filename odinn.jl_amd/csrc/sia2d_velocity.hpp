// sia2d_velocity.hpp -- surface-velocity path for the A-type laws (target :A, law modes 0 / 1) and the
// U law (target :D, law mode LM_NN: Velocity^ = U / f with central finite differences for its partials and per-node
// backprop for d/dtheta, target_D_pure.jl:206-255):
//   k_surface_V   : (Vx, Vy) = -Velocity^(Hbar, |grad S|) * grad S on the dual grid, stored in
//                   nx*ny arrays with the reference's inn1 pairing (last row / column 0)
//                   [Huginn.surface_V / V_from_H, restated from adjoint.jl:268-350]
//   k_surfV_vjp<0>: VJP_lambda_dsurface_V/dH_discrete (adjoint.jl:268-350) in gather form and the
//                   reduction of VJP_lambda_dsurface_V/dtheta_discrete (adjoint.jl:352-413)
//   k_surfV_vjp<1>: the same with the cotangent taken from LossV on the fly
//                   (backward_loss(::LossV), Losses.jl:338-390: L2Sum on :xy or :abs, mask
//                   V_ref > 0, optional scaling) and accumulated into the adjoint state.
//   k_surfV_vjp<2>: explicit cotangent arrays scaled per glacier by wv and accumulated -- the per-stop pull-back of
//                   LossAvgV (TimeAggregatedLosses.jl:240-252), with k_avgv_axpy / k_avgv_cot forming the time-averaged
//                   velocity, its loss and the cotangent dl/dV (:206-231).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

// Velocity^ and its partials on one node (target_A.jl:94-142).  spat = d Velocity^/dA.
template <int LM>
__device__ __forceinline__ double node_Vup(const GDev& g, double Hb, double gS2, double An, double& alpha,
                                           double& beta, double& spat) {
  const double Gu = g.Gam * (g.n + 2.0) / (g.n + 1.0);  // 2 (rho g)^n / (n+1)
  if (LM == LM_FAST) {
    const double H2 = Hb * Hb, H3 = H2 * Hb, H4 = H2 * H2;
    const double AG = An * Gu;
    alpha = AG * 4.0 * H3 * gS2;
    beta = AG * 2.0 * H4;
    spat = Gu * H4 * gS2;
    return AG * H4 * gS2;
  }
  const double hn1 = upow(Hb, g.n + 1.0), sn1 = spow(gS2, g.n - 1.0), sn3 = spow(gS2, g.n - 3.0);
  double D = An * Gu * hn1 * sn1;
  alpha = An * Gu * (g.n + 1.0) * upow(Hb, g.n) * sn1;
  beta = An * Gu * (g.n - 1.0) * hn1 * sn3;
  spat = Gu * hn1 * sn1;
  if (g.Sc != 0.0) {  // sliding terms exactly as the reference writes them
    const double k = g.Sc * (g.p - g.q + 2.0);
    D += k * upow(Hb, g.p - g.q + 1.0) * sn1;
    alpha += k * upow(Hb, g.p - g.q) * sn1;
    beta += k * (g.p - 1.0) * upow(Hb, g.p - g.q + 1.0) * sn3;
  }
  return D;
}

// target :D (target_D_pure.jl:206-255): Velocity^ = U / f, dVelocity^/dH and /d|grad S| by central differences of the law
// with steps 1e-4 and 1e-6 (the latter NOT divided by |grad S|, as the reference returns it)
template <class AR>
__device__ __forceinline__ bool law_arch_is(const LawDev& L) {  // wave-uniform (the descriptor lives in scalar registers)
  bool ok = L.n_layers == AR::NL;
#pragma unroll
  for (int l = 0; l <= AR::NL; ++l) ok = ok && L.widths[l] == AR::W[l];
#pragma unroll
  for (int l = 0; l < AR::NL; ++l) ok = ok && L.acts[l] == AR::A[l];
  return ok;
}
__device__ __forceinline__ double node_Vup_U(const LawDev& L, double Hb, double gS, double finv, double& alpha, double& beta) {
  const double dH = 1e-4, dS = 1e-6;
  if (law_arch_is<ArchDef>(L)) {
    // the default 2-3-10-3-1 network: U at the node and at the four points of the two central differences in ONE pass
    // (mlp_eval_pert, sia2d_device.hpp) instead of five evaluations of the run-time-architecture network
    const int pd[4] = {0, 0, 1, 1};
    const double dl[4] = {dH, -dH, dS, -dS};
    double up[4];
    const double U = mlp_eval_pert<ArchDef, 4>(L, Hb, gS, pd, dl, up);
    alpha = finv * ((up[0] - up[1]) / (2.0 * dH));
    beta = finv * ((up[2] - up[3]) / (2.0 * dS));
    return U * finv;
  }
  alpha = finv * ((mlp_eval_any(L, Hb + dH, gS) - mlp_eval_any(L, Hb - dH, gS)) / (2.0 * dH));
  beta = finv * ((mlp_eval_any(L, Hb, gS + dS) - mlp_eval_any(L, Hb, gS - dS)) / (2.0 * dS));
  return mlp_eval_any(L, Hb, gS) * finv;
}

// target :D_hybrid AS WRITTEN in the reference (target_D_hybrid.jl:210-372; the CPU restatement under tests/ records the
// inconsistencies -- upstream has no test of this path):
//   Velocity^      = S H^(p-q+1) |gS|^(p-1) + Y Gamma H^(n_H+1) |gS|^(n_S-1)        with the DIFFUSIVITY's Gamma = 2 (rho g)^n / (n+2)
//   dVelocity^/dH  = (p-q+1) S H^(p-q) |gS|^(p-1) + (n_H+1) Y Gamma H^n_H |gS|^(n_S-1)
//                    + [compute_D(Y(H + 1e-4)) - compute_D(Y(H))] / 1e-4             (compute_D: H^(n_H+2), the diffusivity)
//   dVelocity^/dgS = (p-1) S H^(p-q+1) |gS|^(p-3) + Gamma^ Y (n_S-1) H^(n_H+2) |gS|^(n_S-3)   with Gamma^ = 2 (rho g)^n / (n+1)
//   spat (x dY/dtheta) = Gamma^ H^(n_H+1) |gS|^(n_S-1)
__device__ __forceinline__ double node_Vup_Y(const GDev& g, const LawDev& L, double Hb, double gS2, double& alpha, double& beta,
                                             double& spat) {
  const double dH = 1e-4;
  const double Gu = g.Gam * (g.n + 2.0) / (g.n + 1.0);  // Gamma^
  const double Y = mlp_eval_any(L, g.T, Hb), Yp = mlp_eval_any(L, g.T, Hb + dH);
  const double sS1 = spow(gS2, g.nS - 1.0);
  const double h1 = upow(Hb, g.nH + 1.0), h2 = upow(Hb, g.nH + 2.0);
  double D = Y * g.Gam * h1 * sS1;
  const double geoD = g.Gam * h2 * sS1;
  double slide = 0.0;
  alpha = (g.nH + 1.0) * Y * g.Gam * upow(Hb, g.nH) * sS1;
  beta = Gu * Y * (g.nS - 1.0) * h2 * spow(gS2, g.nS - 3.0);
  if (g.Sc != 0.0) {
    const double hs = upow(Hb, g.p - g.q + 1.0), sp1 = spow(gS2, g.p - 1.0);
    slide = g.Sc * hs * sp1;
    D += slide;
    alpha += (g.p - g.q + 1.0) * g.Sc * upow(Hb, g.p - g.q) * sp1;
    beta += (g.p - 1.0) * g.Sc * hs * spow(gS2, g.p - 3.0);
  }
  alpha += ((slide + Yp * geoD) - (slide + Y * geoD)) / dH;
  spat = Gu * h1 * sS1;
  return D;
}

template <int LM>
__global__ __launch_bounds__(NT) void k_surface_V(Pools P, LawDev L, const double* __restrict__ U, double* __restrict__ Vx,
                                                  double* __restrict__ Vy, int tile_base, double finv) {
  __shared__ double2 sHS[TY + 2][LDW];
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double own[RPT];
  load_tile_HS2(U, P.B, g, i0, j0, sHS, own);
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      double vx = 0.0, vy = 0.0;
      if (gi <= g.nx - 2 && gj <= g.ny - 2) {  // node whose lower-left cell is (gi, gj)
        double gx, gy, Hb;
        node_geom<LDW>(g, &sHS[r][tx + 1], gx, gy, Hb);
        double D;
        if constexpr (LM == LM_NN) {
          if (L.kind == 3) {  // Y law, target :D_hybrid
            double al, be, sp;
            D = node_Vup_Y(g, L, Hb, gx * gx + gy * gy, al, be, sp);
          } else {
            D = mlp_eval_any(L, Hb, sqrt(gx * gx + gy * gy)) * finv;
          }
        } else {
          double An = g.A;
          if (g.use_Afield) An = P.Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
          double al, be, sp;
          D = node_Vup<LM>(g, Hb, gx * gx + gy * gy, An, al, be, sp);
        }
        vx = -D * gx;
        vy = -D * gy;
      }
      const long long id = g.off + gi + (long long)g.nx * gj;
      Vx[id] = vx;
      Vy[id] = vy;
    }
  }
}

struct VArgs {
  const double* H;        // state (pooled)
  const double* dVx;      // MODE 0: cotangents (pooled, nx*ny, inn1 pairing)
  const double* dVy;
  double* out;            // MODE 0: written; MODE 1 / 2: accumulated (+=)
  double* out_alt;        // non-null (MODE >= 1): the glaciers whose GState says cur == 1 accumulate here instead (the fused
                          //   reverse step keeps lambda of every glacier in its own ping-pong buffer)
  // MODE 1: LossV data
  const double* Vabs;     // [slot][ntot]
  const double* Vxr;
  const double* Vyr;
  const double* wv;       // per-glacier weight (0: skip the glacier)
  const double* scale;    // per-glacier 1/sqrt(mean |V_ref|^2) or 1
  const int* refslot;     // per-glacier slot
  long long ntot;
  int component_abs;      // 0: :xy, 1: :abs
  double log_eps;         // > 0 (:abs only): LogSum(eps) instead of L2Sum -- log^2((V + eps) / (V_ref + eps)) (Losses.jl:207-229)
  double* Gacc;           // gridded-A accumulator or null
  // continuous adjoint, loss term at a snapshot time: wv / scale / refslot are the full [n_snap][G] tables and
  // the row is the snapshot the glacier's reverse solve just reached (nothing to do otherwise)
  const AdjState* adj;
  int G;
  // U law (target :D): 1 / f_surface_velocity_factor
  double finv;
  // per-node-network laws: the theta-part is EMITTED -- per owned dual node Hbar, the node weight and (U law) |grad S| into dual
  // pooled arrays the caller zeroed; the host contracts them with d law / d theta (per node or through the target's `:Linear`
  // interpolation).  null: the theta-part is not wanted
  double* emitH;
  double* emitV;
  double* emitS;
};

template <int MODE, int LM>
__global__ __launch_bounds__(NT) void k_surfV_vjp(Pools P, LawDev L, VArgs A, int tile_base) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double2 sQ[TY + 1][LDN];   // {Qx, Qy}
  __shared__ double sAW[TY + 1][LDN];   // alpha * W
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const GDev g = P.gd[t4.x];
  double wv = 1.0, sc = 1.0;
  long long roff = 0;
  if (MODE == 1 && A.adj) {
    const int j = A.adj[t4.x].snapj;
    const long long q = (long long)j * A.G + t4.x;
    wv = (P.gs[t4.x].at_stop && j >= 0) ? A.wv[q] : 0.0;
    if (wv == 0.0) {
      if (threadIdx.x == 0) { P.part[4 * (long long)t4.w + 3] = 0.0; P.part[4 * (long long)t4.w + 1] = 0.0; }
      return;
    }
    sc = A.scale[q];
    roff = (long long)A.refslot[q] * A.ntot;
  } else if (MODE >= 1) {
    wv = A.wv[t4.x];
    if (wv == 0.0) {  // no velocity data at this stop for this glacier
      if (threadIdx.x == 0) { P.part[4 * (long long)t4.w + 3] = 0.0; if (MODE == 1) P.part[4 * (long long)t4.w + 1] = 0.0; }
      return;
    }
    if (MODE == 1) {
      sc = A.scale[t4.x];
      roff = (long long)A.refslot[t4.x] * A.ntot;
    }
  }
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double* __restrict__ outp = (MODE >= 1 && A.out_alt && P.gs[t4.x].cur) ? A.out_alt : A.out;
  double own[RPT];
  load_tile_HS2(A.H, P.B, g, i0, j0, sHS, own);
  __syncthreads();
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  double gsum = 0.0, lsum = 0.0;
  for (int idx = threadIdx.x; idx < NNODE; idx += NT) {
    const int b = idx / (TX + 1), a = idx - b * (TX + 1);
    const int gi = i0 - 1 + a, gj = j0 - 1 + b;
    double aW = 0.0, Qx = 0.0, Qy = 0.0;
    if (gi >= 0 && gi <= g.nx - 2 && gj >= 0 && gj <= g.ny - 2) {
      double gx, gy, Hb;
      node_geom<LDW>(g, &sHS[b][a], gx, gy, Hb);
      double al, be, sp = 0.0, D;
      [[maybe_unused]] double gS = 0.0;
      if constexpr (LM == LM_NN) {
        if (L.kind == 3) {  // Y law, target :D_hybrid
          D = node_Vup_Y(g, L, Hb, gx * gx + gy * gy, al, be, sp);
        } else {
          gS = sqrt(gx * gx + gy * gy);
          D = node_Vup_U(L, Hb, gS, A.finv, al, be);
        }
      } else {
        double An = g.A;
        if (g.use_Afield) An = P.Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
        D = node_Vup<LM>(g, Hb, gx * gx + gy * gy, An, al, be, sp);
      }
      const long long id = g.off + gi + (long long)g.nx * gj;  // inn1 pairing: node (gi,gj) <-> element [gi,gj]
      const bool owned = (a >= 1 && b >= 1);  // lower-left cell inside the tile interior: reduced here
      double dvx, dvy;
      if (MODE != 1) {
        dvx = A.dVx[id];
        dvy = A.dVy[id];
      } else {
        dvx = 0.0; dvy = 0.0;
        const double va = A.Vabs[roff + id];
        if (va > 0.0) {  // mask = V_ref > 0 (Losses.jl:361)
          const double vx = -D * gx, vy = -D * gy;
          const double ex = vx - A.Vxr[roff + id], ey = vy - A.Vyr[roff + id];
          if (!A.component_abs) {
            dvx = 2.0 * ex * Ninv * sc;
            dvy = 2.0 * ey * Ninv * sc;
            if (owned) lsum = fma(ex, ex, fma(ey, ey, lsum));
          } else {
            const double v = sqrt(vx * vx + vy * vy), ev = v - va;
            double dv = 2.0 * ev * Ninv, lq = ev;
            if (A.log_eps > 0.0) {  // LogSum: l = log^2 q, dl/dV = 2 log q / (V + eps), q = (V + eps) / (V_ref + eps)
              lq = log((v + A.log_eps) / (va + A.log_eps));
              dv = 2.0 * lq / (v + A.log_eps) * Ninv;
            }
            dvx = dv * ex / ev * sc;  // as the reference writes it (Losses.jl:367-368)
            dvy = dv * ey / ev * sc;
            if (owned) lsum = fma(lq, lq, lsum);
          }
        }
      }
      const double W = gx * dvx + gy * dvy;
      aW = al * W;
      Qx = fma(be * gx, W, D * dvx);
      Qy = fma(be * gy, W, D * dvy);
      if (owned) {
        if constexpr (LM == LM_NN) {
          if (L.kind == 3) {
            // Y law: dVelocity^/dtheta = spat x dY/dtheta(T, Hbar): Hbar and the node weight are EMITTED; the host then contracts
            // them with dY/dtheta -- exact backprop at every node (:None, k_node_backprop) or through the knot interpolation
            // (`:Linear`, the target's default; k_interp.hip), as k_vjp_theta does
            const double wn = -wv * W * sp;
            if (A.emitH) {  // (ice-free nodes: spat = 0, the weight vanishes)
              const long long q = g.offd + gi + (long long)(g.nx - 1) * gj;
              A.emitH[q] = Hb;
              A.emitV[q] = wn;
            }
          } else if (A.emitH) {
            // U law: dVelocity^/dtheta = (Hbar > 0) dU/dtheta / f, emitted likewise: exact backprop per node (:None,
            // target_D_pure.jl:163-176,247-255) or grad_itp(Hbar, |grad S|) on LawU's node grid (`:Linear`, :179-193)
            const long long q = g.offd + gi + (long long)(g.nx - 1) * gj;
            A.emitH[q] = Hb;
            A.emitV[q] = Hb > 0.0 ? -wv * W * A.finv : 0.0;
            A.emitS[q] = gS;
          }
        } else {
          const double t = sp * W;
          gsum += t;
          if (A.Gacc) A.Gacc[g.offd + gi + (long long)(g.nx - 1) * gj] -= wv * t;
        }
      }
    }
    sAW[b][a] = aW;
    sQ[b][a] = make_double2(Qx, Qy);
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx, c = tx + 1;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      const double2 qsw = sQ[r - 1][c - 1], qse = sQ[r - 1][c], qnw = sQ[r][c - 1], qne = sQ[r][c];
      double v = 0.25 * ((sAW[r - 1][c - 1] + sAW[r - 1][c]) + (sAW[r][c - 1] + sAW[r][c]));
      v = fma(g.inv_dx * 0.5, (qsw.x + qnw.x) - (qse.x + qne.x), v);
      v = fma(g.inv_dy * 0.5, (qsw.y + qse.y) - (qnw.y + qne.y), v);
      const long long id = g.off + gi + (long long)g.nx * gj;
      if (MODE == 0) A.out[id] = -v;
      else outp[id] = fma(-wv, v, outp[id]);
    }
  }
  const double gt = block_sum(gsum, red);
  const double lt = (MODE == 1) ? block_sum(lsum, red) : 0.0;
  if (threadIdx.x == 0) {
    P.part[4 * (long long)t4.w + 3] = -wv * gt;             // enters dtheta as (dA/dtheta) * sum
    if (MODE == 1) P.part[4 * (long long)t4.w + 1] = lt * Ninv * sc * wv;
  }
}

// ---- continuous adjoint: the theta-part of the velocity loss at a quadrature node in ONE pass ---------------------------
// (gradient.jl:475-503, backward_loss(::LossV) with the reference velocities interpolated linearly in time, :291-301.)
// Closed-form laws without a dual-grid accumulator: the weight of a glacier in dL/dtheta is linear in the loss's
// normalisation 1/sqrt(mean |V_ref|^2) (Losses.jl:323-326), so one pass forms the interpolated reference, the
// normalisation sums (slots 0, 1: every cell) and the UNSCALED sum of dVelocity^/dA x W over the dual nodes (slot 3);
// k_vq_finish applies scale and quadrature weight per glacier.  Replaces k_vref_itp + k_vref_scale + k_surfV_vjp<1> +
// k_sum_part on that path: no interpolated fields (reference velocities, H) written and re-read, no H-cotangent gathered
// only to be discarded.
// tnode (non-null: a dual-grid accumulator is wanted): the unscaled weight of every dual node, which k_gacc_axpy adds into
// the accumulator once k_vq_finish knows the glacier's coefficient.
template <int LM>
__global__ __launch_bounds__(NT) void k_surfV_theta_node(Pools P, VItpArgs I, const double* __restrict__ snaps, int component_abs,
                                                         double log_eps, double* __restrict__ tnode) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const GState* gs = P.gs + t4.x;
  double* pp = P.part + 4 * (long long)t4.w;
  const bool at_node = gs->at_stop && I.adj[t4.x].qw != 0.0;
  const long long q = (long long)(gs->istop - 1) * I.G + t4.x;
  const int sa = at_node ? I.slotA[q] : -1;
  if (sa < 0) {  // not at a node, or no velocity data on this glacier
    if (threadIdx.x == 0) { pp[0] = 0.0; pp[1] = 0.0; pp[3] = 0.0; }
    return;
  }
  const GDev g = P.gd[t4.x];
  const long long oa = (long long)sa * I.ntot, ob = (long long)I.slotB[q] * I.ntot;
  const double w = I.sw[q];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double own[RPT];
  {  // H at the node: the interpolant of the two bracketing snapshots (k_adj_poststep's Hq formula, gradient.jl:287)
    const AdjState a = I.adj[t4.x];
    const double* Ha = snaps + (long long)a.seg_stop * I.ntot;
    load_tile_HS2(Ha, P.B, g, i0, j0, sHS, own, Ha + I.ntot, a.s_stop);
  }
  __syncthreads();
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double ss = 0.0, cnt = 0.0, gsum = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      auto lerp = [&](const double* f) { const double a = f[oa + id]; return w == 0.0 ? a : fma(w, f[ob + id] - a, a); };
      const double va = lerp(I.Vabs), vxr = lerp(I.Vxr), vyr = lerp(I.Vyr);  // k_vref_itp's formula
      const bool isnode = gi <= g.nx - 2 && gj <= g.ny - 2;
      double tn = 0.0;
      if (va > 0.0) {  // mask = V_ref > 0 (Losses.jl:361)
        ss = fma(vxr, vxr, fma(vyr, vyr, ss));
        cnt += 1.0;
        if (isnode) {  // the node whose lower-left cell is (gi, gj): inn1 pairing
          double gx, gy, Hb;
          node_geom<LDW>(g, &sHS[r][tx + 1], gx, gy, Hb);
          double An = g.A;
          if (g.use_Afield) An = P.Afield[g.offd + gi + (long long)(g.nx - 1) * gj];
          double al, be, sp;
          const double D = node_Vup<LM>(g, Hb, gx * gx + gy * gy, An, al, be, sp);
          const double vx = -D * gx, vy = -D * gy;
          const double ex = vx - vxr, ey = vy - vyr;
          double dvx, dvy;
          if (!component_abs) {
            dvx = 2.0 * ex * Ninv;
            dvy = 2.0 * ey * Ninv;
          } else {
            const double v = sqrt(vx * vx + vy * vy), ev = v - va;
            double dv = 2.0 * ev * Ninv;
            if (log_eps > 0.0) dv = 2.0 * log((v + log_eps) / (va + log_eps)) / (v + log_eps) * Ninv;
            dvx = dv * ex / ev;  // as the reference writes it (Losses.jl:367-368)
            dvy = dv * ey / ev;
          }
          tn = sp * (gx * dvx + gy * dvy);
          gsum += tn;
        }
      }
      if (tnode && isnode) tnode[g.offd + gi + (long long)(g.nx - 1) * gj] = tn;
    }
  }
  const double t0 = block_sum(ss, red);
  const double t1 = block_sum(cnt, red);
  const double t3 = block_sum(gsum, red);
  if (threadIdx.x == 0) { pp[0] = t0; pp[1] = t1; pp[3] = -t3; }
}

// theta-part ONLY of the pull-back of explicit cotangents (k_surfV_vjp<2> without its H-part): closed-form laws, for callers
// that discard dL/dH (VelocityRegularization at the quadrature nodes of the continuous adjoint).  Slot 3 of the tile
// partials = -wv sum_nodes dVelocity^/dA x (grad S . dV); the dual-grid accumulator gets the node terms.
template <int LM>
__global__ __launch_bounds__(NT) void k_surfV_theta_only(Pools P, VArgs A) {
  __shared__ double2 sHS[TY + 2][LDW];
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  const double wv = A.wv[t4.x];
  if (wv == 0.0) {
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 3] = 0.0;
    return;
  }
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double own[RPT];
  load_tile_HS2(A.H, P.B, g, i0, j0, sHS, own);
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = wave_id();
  const int gi = i0 + tx;
  double gsum = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int r = 1 + ty + NW * m, gj = j0 - 1 + r;
    if (gi <= g.nx - 2 && gj <= g.ny - 2) {  // the node whose lower-left cell is (gi, gj): inn1 pairing
      double gx, gy, Hb;
      node_geom<LDW>(g, &sHS[r][tx + 1], gx, gy, Hb);
      double An = g.A;
      const long long qd = g.offd + gi + (long long)(g.nx - 1) * gj;
      if (g.use_Afield) An = P.Afield[qd];
      double al, be, sp;
      (void)node_Vup<LM>(g, Hb, gx * gx + gy * gy, An, al, be, sp);
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double t = sp * (gx * A.dVx[id] + gy * A.dVy[id]);
      gsum += t;
      if (A.Gacc) A.Gacc[qd] -= wv * t;
    }
  }
  const double gt = block_sum(gsum, red);
  if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 3] = -wv * gt;
}

// Gacc[node] += coef[g] * tnode[node] on the dual nodes of the glaciers with coef[g] != 0 (k_vq_finish: the glacier just
// reached a quadrature node)
#ifdef ODINN_VEL_KERNELS
__global__ __launch_bounds__(NT) void k_gacc_axpy(Pools P, const double* __restrict__ coef, const double* __restrict__ tnode,
                                                  double* __restrict__ Gacc) {
  const int4 t4 = P.tiles[blockIdx.x];
  const double c = coef[t4.x];
  if (c == 0.0) return;
  const GDev g = P.gd[t4.x];
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi <= g.nx - 2 && gj <= g.ny - 2) {
      const long long q = g.offd + gi + (long long)(g.nx - 1) * gj;
      Gacc[q] = fma(c, tnode[q], Gacc[q]);
    }
  }
}
#endif

// ---- LossAvgV (TimeAggregatedLosses.jl:115-258) ---------------------------------------------------------------------
#ifdef ODINN_VEL_KERNELS  // non-template kernels: compiled by k_vel.hip only
// avg += w_g V  for the glaciers whose tLoss contains this stop (w_g = dt_i / T, 0: not this glacier)
__global__ __launch_bounds__(NT) void k_avgv_axpy(Pools P, const double* __restrict__ Vx, const double* __restrict__ Vy,
                                                  double* __restrict__ ax, double* __restrict__ ay, const double* __restrict__ w) {
  const int4 t4 = P.tiles[blockIdx.x];
  const double wg = w[t4.x];
  if (wg == 0.0) return;
  const GDev g = P.gd[t4.x];
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      ax[id] = fma(Vx[id], wg, ax[id]);
      ay[id] = fma(Vy[id], wg, ay[id]);
    }
  }
}
// loss of the averaged velocity against the single reference sample (L2Sum, mask V_ref > 0, normalization nx ny) into the
// tile partial slot 1, and the averages REPLACED by the cotangents weight * dl/dVx, weight * dl/dVy (:xy or :abs, :224-231)
__global__ __launch_bounds__(NT) void k_avgv_cot(Pools P, double* __restrict__ ax, double* __restrict__ ay,
                                                 const double* __restrict__ Vabs, const double* __restrict__ Vxr,
                                                 const double* __restrict__ Vyr, const unsigned char* __restrict__ on,
                                                 int component_abs, double weight) {
  __shared__ double red[NW];
  const int4 t4 = P.tiles[blockIdx.x];
  if (!on[t4.x]) {
    if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = 0.0;
    return;
  }
  const GDev g = P.gd[t4.x];
  const double Ninv = 1.0 / ((double)g.nx * (double)g.ny);
  const int gi = t4.y * TX + (threadIdx.x & 63), ty = wave_id();
  double lsum = 0.0;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int gj = t4.z * TY + ty + NW * m;
    if (gi < g.nx && gj < g.ny) {
      const long long id = g.off + gi + (long long)g.nx * gj;
      const double va = Vabs[id];
      double cx = 0.0, cy = 0.0;
      if (va > 0.0) {
        const double vx = ax[id], vy = ay[id];
        const double ex = vx - Vxr[id], ey = vy - Vyr[id];
        if (!component_abs) {
          cx = 2.0 * ex * Ninv * weight;
          cy = 2.0 * ey * Ninv * weight;
          lsum = fma(ex, ex, fma(ey, ey, lsum));
        } else {
          const double ev = sqrt(vx * vx + vy * vy) - va;
          const double dv = 2.0 * ev * Ninv;
          cx = dv * ex / ev * weight;  // as the reference writes it (:229-231)
          cy = dv * ey / ev * weight;
          lsum = fma(ev, ev, lsum);
        }
      }
      ax[id] = cx;
      ay[id] = cy;
    }
  }
  const double lt = block_sum(lsum, red);
  if (threadIdx.x == 0) P.part[4 * (long long)t4.w + 1] = lt * Ninv * weight;
}
#endif  // ODINN_VEL_KERNELS

}  // namespace odinn
