// sia2d_nn_grad.hpp -- discrete theta-VJP for per-node MLP laws (Y = NN(T,Hbar), U = NN(Hbar,|gradS|))
// with a compile-time architecture.
//
// Reference: dtheta_k = sum_nodes dD/dtheta_k * D_adjoint (adjoint.jl:235-250) with
// dD/dtheta = spat * dlaw/dtheta evaluated per node (interpolation = :None branch,
// target_D_hybrid.jl:121-131, target_D_pure.jl:163-176); the reference materialises a dense
// (nx-1)(ny-1) x P tensor and contracts it with Tullio.  Here the tensor never exists:
//   * each lane backpropagates ITS node through the MLP (activations and their derivatives stay
//     in registers, weights arrive through wave-uniform scalar loads);
//   * per GROUP of consecutive layers (NNG) the wave parks dz (out) and h_prev (in) of its 64 nodes in LDS and the lanes
//     switch roles: lane q owns parameter q of the group and reduces dz[o]*h[i] over the 64 nodes
//     in a fixed order -- P/64 accumulators per lane, no atomics, bitwise deterministic; the exchange buffer is private to
//     the wavefront (compiler fences only: round 5 -- one role switch per layer behind workgroup barriers before: 151 -> 109 us
//     per launch for the 2-3-10-3-1 net on 8 x 512^2);
//   * wavefronts and tiles are combined in fixed order (part_theta -> k_sum_part_theta).
#pragma once
#include "sia2d_device.hpp"

namespace odinn {

// Layers are backpropagated top-down in GROUPS of consecutive layers: the lanes park dz and the inputs of a group's layers in their
// node's LDS row and switch roles once per group (a row holds at most ROWCAP doubles, so the exchange buffers stay at the size of the
// widest single layer: 2-3-10-3-1 -> {L3, L2} and {L1, L0}, one 64-parameter slot each instead of one per layer; 2-16-16-1 -> one
// group per layer).
struct NNGroups {
  int n;
  int hi[MAXL], lo[MAXL];
};
template <class AR>
struct NNG {
  static constexpr int NL = AR::NL, MAXW = AR::MAXW;
  static constexpr int ROWCAP = 2 * MAXW + 2 > 22 ? 2 * MAXW + 2 : 22;
  static constexpr NNGroups make_groups() {
    NNGroups g{};
    int l = NL - 1;
    while (l >= 0) {
      const int hi = l;
      int sz = 1;
      while (l >= 0 && (l == hi || sz + AR::W[l + 1] + AR::W[l] <= ROWCAP)) {
        sz += AR::W[l + 1] + AR::W[l];
        --l;
      }
      g.hi[g.n] = hi;
      g.lo[g.n] = l + 1;
      ++g.n;
    }
    return g;
  }
  static constexpr NNGroups GR = make_groups();
  static constexpr int group_of(int l) {
    for (int k = 0; k < GR.n; ++k)
      if (l <= GR.hi[k] && l >= GR.lo[k]) return k;
    return 0;
  }
  // row of one node for group k: dz of its layers (top first) | their inputs | 1.0 (the biases' "input")
  static constexpr int dz_off(int l) {
    const int k = group_of(l);
    int b = 0;
    for (int j = GR.hi[k]; j > l; --j) b += AR::W[j + 1];
    return b;
  }
  static constexpr int h_off(int l) {
    const int k = group_of(l);
    int b = 0;
    for (int j = GR.hi[k]; j >= GR.lo[k]; --j) b += AR::W[j + 1];
    for (int j = GR.hi[k]; j > l; --j) b += AR::W[j];
    return b;
  }
  static constexpr int one_off(int k) {
    int b = 0;
    for (int j = GR.hi[k]; j >= GR.lo[k]; --j) b += AR::W[j + 1] + AR::W[j];
    return b;
  }
  static constexpr int row_max() {
    int m = 0;
    for (int k = 0; k < GR.n; ++k) m = one_off(k) + 1 > m ? one_off(k) + 1 : m;
    return m;
  }
  static constexpr int ROW = row_max() | 1;  // odd stride: the 64 rows of a wavefront start in different banks
  static constexpr int P = arch_off<AR>(NL);
  // parameters of group k: the contiguous range [p_lo, p_hi) of the flattened theta; lane q owns p_lo + q, p_lo + q + 64, ...
  static constexpr int p_lo(int k) { return arch_off<AR>(GR.lo[k]); }
  static constexpr int p_hi(int k) { return arch_off<AR>(GR.hi[k] + 1); }
  static constexpr int nslots(int k) { return (p_hi(k) - p_lo(k) + 63) / 64; }
  static constexpr int slot_base(int k) {
    int b = 0;
    for (int j = 0; j < k; ++j) b += nslots(j);
    return b;
  }
  static constexpr int NACC = slot_base(GR.n);
  static constexpr int XBUF = NW * 64 * ROW;  // doubles
};

// the groupings the three compile-time architectures get (a change of ROWCAP or of make_groups shows up here, not in a profile)
static_assert(NNG<ArchDef>::GR.n == 2 && NNG<ArchDef>::GR.lo[0] == 2 && NNG<ArchDef>::NACC == 2 && NNG<ArchDef>::ROW == 19 && NNG<ArchDef>::P == 86,
              "2-3-10-3-1: {L3, L2} and {L1, L0}, one 64-parameter slot each, rows of 19 doubles");
static_assert(NNG<Arch16>::GR.n == 3 && NNG<Arch16>::ROW == 33 && NNG<Arch16>::NACC == 1 + 5 + 1, "2-16-16-1: one group per layer, rows of 33 doubles");
static_assert(NNG<ArchLight>::GR.n == 1 && NNG<ArchLight>::NACC == 1 && NNG<ArchLight>::P == 13, "2-3-1: one group, one slot");

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

// one layer of the backward pass of the lane's node: dz = g * act'(z) and the layer's input go to the node's LDS row, g moves on to
// the layer below
template <class AR, int l>
__device__ __forceinline__ void nn_bwd_layer(const double* __restrict__ th, double* row, double (&gv)[AR::MAXW], const double (&dl)[AR::MAXW],
                                             const double (&hprev)[AR::MAXW]) {
  using G = NNG<AR>;
  constexpr int nin = AR::W[l], nout = AR::W[l + 1], off = arch_off<AR>(l);
  double dz[AR::MAXW];
#pragma unroll
  for (int o = 0; o < nout; ++o) {
    dz[o] = gv[o] * dl[o];
    row[G::dz_off(l) + o] = dz[o];
  }
#pragma unroll
  for (int i = 0; i < nin; ++i) row[G::h_off(l) + i] = hprev[i];
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
// where parameter q of the flattened theta finds its two factors in a node's row: dz[o] of its layer and the layer's input ic (or 1.0)
template <class AR>
__device__ __forceinline__ void nn_param_slots(int q, int& pa, int& pb) {
  using G = NNG<AR>;
  pa = 0; pb = 0;
#pragma unroll
  for (int l = 0; l < AR::NL; ++l) {
    const int off = arch_off<AR>(l), nin = AR::W[l], nout = AR::W[l + 1];
    if (q >= off && q < off + nout * (nin + 1)) {
      const int r = q - off;
      if (r < nin * nout) { const int ic = r / nout; pa = G::dz_off(l) + (r - ic * nout); pb = G::h_off(l) + ic; }
      else { pa = G::dz_off(l) + (r - nin * nout); pb = G::one_off(G::group_of(l)); }
    }
  }
}
// the role switch of group K: lane q reduces its parameters of the group over the wave's 64 nodes in a fixed order.  The exchange buffer
// is private to the wavefront and the LDS serves a wavefront's accesses in order: compiler fences, no workgroup barriers.
template <class AR, int K>
__device__ __forceinline__ void nn_reduce_group(double* wbuf, double* row, int lane, const int (&pa)[NNG<AR>::NACC], const int (&pb)[NNG<AR>::NACC],
                                                double (&acc)[NNG<AR>::NACC]) {
  using G = NNG<AR>;
  row[G::one_off(K)] = 1.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int s = 0; s < G::nslots(K); ++s) {
    const int k = G::slot_base(K) + s;
    if (G::p_lo(K) + lane + 64 * s < G::p_hi(K)) {
      const double* qa = wbuf + pa[k];
      const double* qb = wbuf + pb[k];
      double a = 0.0;
#pragma unroll 8
      for (int n = 0; n < 64; ++n) a = fma(qa[n * G::ROW], qb[n * G::ROW], a);
      acc[k] += a;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// layer l of the backward pass, followed by its group's role switch when l is the group's lowest layer
template <class AR, int l>
__device__ __forceinline__ void nn_bwd_step(const double* __restrict__ th, double* wbuf, double* row, int lane, double (&gv)[AR::MAXW],
                                            const double (&dl)[AR::MAXW], const double (&hprev)[AR::MAXW], const int (&pa)[NNG<AR>::NACC],
                                            const int (&pb)[NNG<AR>::NACC], double (&acc)[NNG<AR>::NACC]) {
  using G = NNG<AR>;
  nn_bwd_layer<AR, l>(th, row, gv, dl, hprev);
  constexpr int K = G::group_of(l);
  if constexpr (G::GR.lo[K] == l) nn_reduce_group<AR, K>(wbuf, row, lane, pa, pb, acc);
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
  // The wave's 64 nodes of a row are backpropagated lane by lane; per group of layers the lanes leave dz and the layers' inputs in their
  // node's LDS row and switch roles (nn_reduce_group).  A row of nodes without ice is skipped by its wavefront.
  double acc[G::NACC];
  int pa[G::NACC], pb[G::NACC];
#pragma unroll
  for (int k = 0; k < G::NACC; ++k) { acc[k] = 0.0; pa[k] = 0; pb[k] = 0; }
#pragma unroll
  for (int K = 0; K < G::GR.n; ++K)
#pragma unroll
    for (int s_ = 0; s_ < G::nslots(K); ++s_) nn_param_slots<AR>(G::p_lo(K) + tx + 64 * s_, pa[G::slot_base(K) + s_], pb[G::slot_base(K) + s_]);
  if (any_active) {
    double* wbuf = smem + (threadIdx.x >> 6) * 64 * G::ROW;
    double* row = wbuf + tx * G::ROW;
    const double* __restrict__ th = L.theta;
#pragma unroll 1
    for (int m = 0; m < RPT; ++m) {
      const double w = wgt[m];
      if (!__any(w != 0.0)) continue;  // (wave-uniform: no ice on this row of nodes)
      double h0[AR::MAXW], h1[AR::MAXW], h2[AR::MAXW], h3[AR::MAXW], h4[AR::MAXW];
      double d1[AR::MAXW], d2[AR::MAXW], d3[AR::MAXW], d4[AR::MAXW];
#pragma unroll
      for (int i = 0; i < AR::MAXW; ++i) { h0[i] = h1[i] = h2[i] = h3[i] = h4[i] = 0.0; d1[i] = d2[i] = d3[i] = d4[i] = 0.0; }
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
      if constexpr (AR::NL > 3) nn_bwd_step<AR, 3>(th, wbuf, row, tx, gv, d4, h3, pa, pb, acc);
      if constexpr (AR::NL > 2) nn_bwd_step<AR, 2>(th, wbuf, row, tx, gv, d3, h2, pa, pb, acc);
      if constexpr (AR::NL > 1) nn_bwd_step<AR, 1>(th, wbuf, row, tx, gv, d2, h1, pa, pb, acc);
      nn_bwd_step<AR, 0>(th, wbuf, row, tx, gv, d1, h0, pa, pb, acc);
    }
  }
  // ---- combine the 4 wavefronts in fixed order -> part_theta[tile][k] -------------------------
  __syncthreads();
  {
    double* red = smem;  // [NW][P]
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int K = 0; K < G::GR.n; ++K)
#pragma unroll
      for (int s_ = 0; s_ < G::nslots(K); ++s_) {
        const int q = G::p_lo(K) + tx + 64 * s_;
        if (q < G::p_hi(K)) red[wv * G::P + q] = acc[G::slot_base(K) + s_];
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
