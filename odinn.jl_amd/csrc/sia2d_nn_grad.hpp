// sia2d_nn_grad.hpp -- discrete theta-VJP for per-node MLP laws (Y = NN(T,Hbar), U = NN(Hbar,|gradS|))
// with a compile-time architecture.
//
// Reference: dtheta_k = sum_nodes dD/dtheta_k * D_adjoint (adjoint.jl:235-250) with
// dD/dtheta = spat * dlaw/dtheta evaluated per node (interpolation = :None branch,
// target_D_hybrid.jl:121-131, target_D_pure.jl:163-176); the reference materialises a dense
// (nx-1)(ny-1) x P tensor and contracts it with Tullio.  Here the tensor never exists:
//   * each lane backpropagates ITS node through the MLP (activations and their derivatives stay
//     in registers, weights arrive through wave-uniform scalar loads);
//   * per layer the wave parks dz (out) and h_prev (in) of its 64 nodes in LDS and the lanes
//     switch roles: lane q owns parameter q of the layer and reduces dz[o]*h[i] over the 64 nodes
//     in a fixed order -- P/64 accumulators per lane, no atomics, bitwise deterministic;
//   * wavefronts and tiles are combined in fixed order (part_theta -> k_sum_part_theta).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

template <class AR>
struct NNG {
  static constexpr int NL = AR::NL, MAXW = AR::MAXW;
  static constexpr int ROW = 2 * MAXW + 1;  // [0,MAXW) dz | [MAXW, MAXW+nin) h_prev | 1.0 ; odd stride
  static constexpr int nparams(int l) { return AR::W[l + 1] * (AR::W[l] + 1); }
  static constexpr int nslots(int l) { return (nparams(l) + 63) / 64; }
  static constexpr int slot_base(int l) {
    int b = 0;
    for (int k = 0; k < l; ++k) b += nslots(k);
    return b;
  }
  static constexpr int NACC = slot_base(NL);
  static constexpr int P = arch_off<AR>(NL);
  static constexpr int XBUF = NW * 64 * ROW;  // doubles
};

template <int CODE>
__device__ __forceinline__ void act_and_deriv(double z, double& h, double& d) {
  if (CODE == 1) {  // softplus, derivative sigmoid: share t = exp(-|z|)
    const double t = exp_nonpos(-fabs(z));
    h = log1p_01(t) + fmax(z, 0.0);
    d = fast_div(z >= 0.0 ? 1.0 : t, 1.0 + t);
  } else if (CODE == 2) {
    h = sigmoid_f(z);
    d = h * (1.0 - h);
  } else {
    h = act_f(CODE, z);
    d = dact_f(CODE, z);
  }
}

template <class AR, int l>
__device__ __forceinline__ void nn_fwd_layer(const double* __restrict__ th, const double (&hin)[AR::MAXW],
                                             double (&hout)[AR::MAXW], double (&dout)[AR::MAXW]) {
  constexpr int nin = AR::W[l], nout = AR::W[l + 1], off = arch_off<AR>(l);
#pragma unroll
  for (int o = 0; o < nout; ++o) {
    double acc = th[off + nin * nout + o];
#pragma unroll
    for (int i = 0; i < nin; ++i) acc = fma(th[off + o + nout * i], hin[i], acc);
    act_and_deriv<AR::A[l]>(acc, hout[o], dout[o]);
  }
}

// one layer of the backward pass for the wave's 64 nodes + role switch + fixed-order reduction
template <class AR, int l>
__device__ __forceinline__ void nn_bwd_layer(const double* __restrict__ th, double* wbuf, int lane,
                                             double (&gv)[AR::MAXW], const double (&dl)[AR::MAXW],
                                             const double (&hprev)[AR::MAXW], double (&acc)[NNG<AR>::NACC]) {
  using G = NNG<AR>;
  constexpr int nin = AR::W[l], nout = AR::W[l + 1], off = arch_off<AR>(l), np = G::nparams(l);
  double dz[AR::MAXW];
  double* row = wbuf + lane * G::ROW;
#pragma unroll
  for (int o = 0; o < nout; ++o) {
    dz[o] = gv[o] * dl[o];
    row[o] = dz[o];
  }
#pragma unroll
  for (int i = 0; i < nin; ++i) row[G::MAXW + i] = hprev[i];
  row[G::MAXW + nin] = 1.0;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < G::nslots(l); ++s) {
    const int q = lane + 64 * s;
    if (q < np) {
      int o, ic;
      if (q < nin * nout) { ic = q / nout; o = q - ic * nout; } else { o = q - nin * nout; ic = nin; }
      const double* pa = wbuf + o;
      const double* pb = wbuf + G::MAXW + ic;
      double a = 0.0;
#pragma unroll 8
      for (int n = 0; n < 64; ++n) a = fma(pa[n * G::ROW], pb[n * G::ROW], a);
      acc[G::slot_base(l) + s] += a;
    }
  }
  __syncthreads();
  if (l > 0) {
    double gn[AR::MAXW];
#pragma unroll
    for (int i = 0; i < nin; ++i) {
      double t = 0.0;
#pragma unroll
      for (int o = 0; o < nout; ++o) t = fma(th[off + o + nout * i], dz[o], t);
      gn[i] = t;
    }
#pragma unroll
    for (int i = 0; i < nin; ++i) gv[i] = gn[i];
  }
}

template <class AR>
__global__ __launch_bounds__(NT) void k_vjp_theta_nn(Pools P, LawDev L, ThArgs A, int tile_base) {
  using G = NNG<AR>;
  constexpr int TILE_D = (TY + 2) * LDW * 3;  // {Hc,S} + lambda, in doubles
  constexpr int SH_D = TILE_D > G::XBUF ? TILE_D : G::XBUF;
  __shared__ double2 smem2[(SH_D + 1) / 2];
  __shared__ int any_active;
  double* smem = reinterpret_cast<double*>(smem2);
  double2(*sHS)[LDW] = reinterpret_cast<double2(*)[LDW]>(smem);
  double(*sL)[LDW] = reinterpret_cast<double(*)[LDW]>(smem + (TY + 2) * LDW * 2);
  const int4 t4 = P.tiles[blockIdx.x + tile_base];
  const double scale = A.scales ? A.scales[t4.x] : 1.0;
  if (scale == 0.0) {  // this glacier contributes nothing now (e.g. not at a quadrature node)
    for (int k = threadIdx.x; k < G::P; k += NT) A.part_theta[(long long)t4.w * G::P + k] = 0.0;
    return;
  }
  const GDev g = P.gd[t4.x];
  const int i0 = t4.y * TX, j0 = t4.z * TY;
  double ownH[RPT], ownL[RPT];
  load_tile_HS2(A.H, P.B, g, i0, j0, sHS, ownH);
  load_tile_lam((A.lam_alt && P.gs[t4.x].cur) ? A.lam_alt : A.lam, g, i0, j0, sL, ownL);
  if (threadIdx.x == 0) any_active = 0;
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = wave_id();
  double wgt[RPT], x0[RPT], x1[RPT];
  bool act = false;
#pragma unroll
  for (int m = 0; m < RPT; ++m) {
    const int b = 1 + ty + NW * m, a = tx + 1;
    const int gi = i0 - 1 + a, gj = j0 - 1 + b;
    wgt[m] = 0.0; x0[m] = 0.0; x1[m] = 0.0;
    if (gi <= g.nx - 2 && gj <= g.ny - 2) {
      double gx, gy, Hb;
      const double Da = node_Da<LDW>(g, &sHS[b][a], &sL[b][a], gj >= 1, gj + 1 <= g.ny - 2, gi >= 1,
                                     gi + 1 <= g.nx - 2, gx, gy, Hb);
      const double gS2 = gx * gx + gy * gy;
      const double gS = (L.kind == 3) ? 0.0 : sqrt(gS2);  // the U law's second input; the Y law needs |grad S| only through spow
      const double spat = (L.kind == 3) ? g.Gam * upow(Hb, g.nH + 2.0) * spow(gS2, g.nS - 1.0) : Hb;
      if (Hb > 0.0) {  // ice-free nodes carry zero weight (and target_D_pure.jl:166-168 skips them)
        wgt[m] = scale * spat * Da;
        x0[m] = (L.kind == 3) ? g.T : Hb;
        x1[m] = (L.kind == 3) ? Hb : gS;
      }
    }
    act = act || (wgt[m] != 0.0);
  }
  if (act) any_active = 1;
  __syncthreads();  // tiles are dead from here on; their LDS becomes the exchange buffers
  double acc[G::NACC];
#pragma unroll
  for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0;
  if (any_active) {
    double* wbuf = smem + (threadIdx.x >> 6) * 64 * G::ROW;
    const double* __restrict__ th = L.theta;
#pragma unroll 1
    for (int m = 0; m < RPT; ++m) {
      double h0[AR::MAXW], h1[AR::MAXW], h2[AR::MAXW], h3[AR::MAXW], h4[AR::MAXW];
      double d1[AR::MAXW], d2[AR::MAXW], d3[AR::MAXW], d4[AR::MAXW];
#pragma unroll
      for (int i = 0; i < AR::MAXW; ++i) { h0[i] = h1[i] = h2[i] = h3[i] = h4[i] = 0.0; d1[i] = d2[i] = d3[i] = d4[i] = 0.0; }
      double w = wgt[m];
      if (w != 0.0) {
        h0[0] = L.has_pre ? (x0[m] - L.pre_lo[0]) * L.pre_inv[0] - 0.5 : x0[m];
        h0[1] = L.has_pre ? (x1[m] - L.pre_lo[1]) * L.pre_inv[1] - 0.5 : x1[m];
        nn_fwd_layer<AR, 0>(th, h0, h1, d1);
        if constexpr (AR::NL > 1) nn_fwd_layer<AR, 1>(th, h1, h2, d2);
        if constexpr (AR::NL > 2) nn_fwd_layer<AR, 2>(th, h2, h3, d3);
        if constexpr (AR::NL > 3) nn_fwd_layer<AR, 3>(th, h3, h4, d4);
      }
      double gv[AR::MAXW];
#pragma unroll
      for (int i = 0; i < AR::MAXW; ++i) gv[i] = 0.0;
      {
        const double yL = AR::NL == 4 ? h4[0] : (AR::NL == 3 ? h3[0] : (AR::NL == 2 ? h2[0] : h1[0]));
        gv[0] = (w != 0.0) ? w * dpostscale_f(L, yL) : 0.0;
      }
      if constexpr (AR::NL > 3) nn_bwd_layer<AR, 3>(th, wbuf, tx, gv, d4, h3, acc);
      if constexpr (AR::NL > 2) nn_bwd_layer<AR, 2>(th, wbuf, tx, gv, d3, h2, acc);
      if constexpr (AR::NL > 1) nn_bwd_layer<AR, 1>(th, wbuf, tx, gv, d2, h1, acc);
      nn_bwd_layer<AR, 0>(th, wbuf, tx, gv, d1, h0, acc);
    }
  }
  // ---- combine the 4 wavefronts in fixed order -> part_theta[tile][k] -------------------------
  __syncthreads();
  {
    double* red = smem;  // [NW][P]
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < AR::NL; ++l) {
#pragma unroll
      for (int s = 0; s < G::nslots(l); ++s) {
        const int q = tx + 64 * s;
        if (q < G::nparams(l)) red[wv * G::P + arch_off<AR>(l) + q] = acc[G::slot_base(l) + s];
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < G::P; k += NT) {
      double s_ = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s_ += red[w * G::P + k];
      A.part_theta[(long long)t4.w * G::P + k] = s_;
    }
  }
}

}  // namespace odinn
