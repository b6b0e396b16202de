// k_interp.hip -- the `interpolation == :Linear` branch of dDiffusivity/dtheta for the Y law (the reference's default
// for SIA2D_D_hybrid_target, src/models/target/target_D_hybrid.jl:12-15,136-160; knots: create_interpolation,
// src/models/target/target_utils.jl:245-293).
//
// The reference evaluates d law / d theta on <= 2 n_interp_half knots of Hbar and interpolates it linearly at every
// dual node, then contracts with the node weights  v = spat * D_adjoint  (adjoint.jl:235-250).  Linear interpolation
// makes that contraction a sum over KNOTS:  dtheta = sum_k c_k G_k,  c_k = sum_nodes v * (hat function of knot k)(Hbar),
// so the dense (nx-1)(ny-1) x P tensor is never formed and the network is differentiated 2 n_interp_half times instead
// of once per node.  Per glacier and evaluation:
//   1. k_vjp_theta (emit mode) has written Hbar and v of every dual node;
//   2. rocPRIM radix sort of the pairs (Hbar, v): the knots need the quantiles of Hbar, and with the nodes in Hbar order
//      every knot interval is a contiguous range, summed by one workgroup in a fixed order (deterministic, no atomics);
//   3. k_knots: uniform + quantile knots, sorted, duplicates removed;   4. k_knot_grads: exact backprop at the knots;
//   5. k_interval_sums: a_k = sum v (1 - w), b_k = sum v w per interval;   6. k_interp_contract: dtheta (+)= sum_k c_k G_k.
// Default for the Y law: the same steps for ALL glaciers of a call in one sequence of ~20 launches
// (launch_interp_theta_batch below); the per-glacier sequence remains for the U law and as the A/B reference.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include "launch.hpp"
#include <rocprim/rocprim.hpp>

namespace odinn {

constexpr int KMAX = 512;  // 2 * n_interp_half <= KMAX

__device__ __forceinline__ long long lower_bound_d(const double* __restrict__ a, long long n, double x) {  // first i: a[i] >= x
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ long long upper_bound_d(const double* __restrict__ a, long long n, double x) {  // first i: a[i] > x
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// knots[0 .. M): sorted unique union of LinRange(0, max, n) and the type-7 quantiles (probabilities j / (n + 1)) of the
// sorted values strictly inside (0, max).  M = 0 when the glacier carries no ice.
__device__ __forceinline__ void knots_body(const double* __restrict__ sH, long long nd, int n, double* __restrict__ knots,
                                           int* __restrict__ Mout, double* c, int* first) {
  const int i = threadIdx.x;
  const double amax = sH[nd - 1];
  if (!(amax > 0.0)) {
    if (i == 0) *Mout = 0;
    return;
  }
  const long long lo = upper_bound_d(sH, nd, 0.0), hi = lower_bound_d(sH, nd, amax);
  const long long m = hi - lo;  // entries strictly inside (0, max)
  bool valid = false;
  double v = 0.0;
  if (i < n) {
    const double t = (double)i / (double)(n - 1);
    v = (1.0 - t) * 0.0 + t * amax;  // LinRange: lerp with t = j / (n - 1)
    valid = true;
  } else if (i < 2 * n && m > 0) {
    const double p = (double)(i - n + 1) / (double)(n + 1);
    const double h = (double)(m - 1) * p;
    long long j = (long long)floor(h);
    const long long jmax = m >= 2 ? m - 2 : 0;
    if (j > jmax) j = jmax;
    if (j < 0) j = 0;
    double gam = h - (double)j;
    gam = gam < 0.0 ? 0.0 : (gam > 1.0 ? 1.0 : gam);
    const double a = sH[lo + j], b = sH[lo + (j + 1 < m ? j + 1 : m - 1)];
    v = a + gam * (b - a);
    valid = true;
  }
  c[i] = valid ? v : -1.0;  // knots are >= 0
  __syncthreads();
  int isfirst = valid ? 1 : 0;
  if (valid)
    for (int j = 0; j < i; ++j)
      if (c[j] == v) { isfirst = 0; break; }
  first[i] = isfirst;
  __syncthreads();
  if (isfirst) {
    int rank = 0;
    for (int j = 0; j < 2 * n; ++j) rank += (first[j] && c[j] < v) ? 1 : 0;
    knots[rank] = v;
  }
  if (i == 0) {
    int M = 0;
    for (int j = 0; j < 2 * n; ++j) M += first[j];
    *Mout = M;
  }
}

__global__ __launch_bounds__(KMAX) void k_knots(const double* __restrict__ sH, long long nd, int n, double* __restrict__ knots,
                                                int* __restrict__ Mout) {
  __shared__ double c[KMAX];
  __shared__ int first[KMAX];
  knots_body(sH, nd, n, knots, Mout, c, first);
}
// the same for the glaciers g0 + blockIdx.x of a batch whose sorted nodes lie at their pooled dual offsets (minus lo)
// (the pooled dual offsets are padded: a glacier's segment [offd_g, offd_g+1) ends in a few nodes that are never written --
//  Hbar = 0, weight 0 -- and sort to its front with the ice-free nodes, which neither the knots nor the sums look at)
__device__ __forceinline__ long long seg_len(const Pools& P, int gidx, int g_last, long long end_all) {
  return (gidx < g_last ? P.gd[gidx + 1].offd : end_all) - P.gd[gidx].offd;
}
__global__ __launch_bounds__(KMAX) void k_knots_b(Pools P, int g0, int g_last, long long lo, long long end_all,
                                                  const double* __restrict__ sHall, int n, double* __restrict__ knots_all,
                                                  int* __restrict__ M_all) {
  __shared__ double c[KMAX];
  __shared__ int first[KMAX];
  const int gidx = g0 + blockIdx.x;
  knots_body(sHall + (P.gd[gidx].offd - lo), seg_len(P, gidx, g_last, end_all), n, knots_all + (size_t)blockIdx.x * KMAX,
             M_all + blockIdx.x, c, first);
}

// G[q * KMAX + k] = d Y / d theta_q at (T, knots[k])
__global__ __launch_bounds__(64) void k_knot_grads(LawDev L, double T, const double* __restrict__ knots, const int* __restrict__ Mp,
                                                   double* __restrict__ G) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= *Mp) return;
  for (int q = 0; q < L.P; ++q) G[(long long)q * KMAX + k] = 0.0;
  mlp_grad(L, T, knots[k], 1.0, G + k, KMAX);
}

// block k: the nodes with knots[k] <= Hbar < knots[k+1] (the last interval takes Hbar == max too)
__device__ __forceinline__ void interval_sums_body(const double* __restrict__ sH, const double* __restrict__ sV, long long nd,
                                                   const double* __restrict__ knots, const int* __restrict__ Mp,
                                                   double* __restrict__ ab, double* red) {
  const int k = blockIdx.x, M = *Mp;
  if (k >= M - 1) {
    if (threadIdx.x == 0) { ab[k] = 0.0; ab[KMAX + k] = 0.0; }
    return;
  }
  // (w by a true division: the margin of an advancing glacier carries subnormal thicknesses, the lowest quantile knots of a
  //  small glacier land among them, and 1 / (x1 - x0) overflows for such an interval while (Hbar - x0) / (x1 - x0) <= 1)
  const double x0 = knots[k], x1 = knots[k + 1], wid = x1 - x0;
  // (a node with Hbar == 0 carries the weight 0 -- spat has a positive power of Hbar -- and ice-free nodes can be half of
  //  the domain, all in interval 0: they are skipped)
  const long long lo = x0 > 0.0 ? lower_bound_d(sH, nd, x0) : upper_bound_d(sH, nd, 0.0);
  const long long hi = (k == M - 2) ? nd : lower_bound_d(sH, nd, x1);
  double a = 0.0, b = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += NT) {
    const double w = (sH[i] - x0) / wid, v = sV[i];
    a = fma(v, 1.0 - w, a);
    b = fma(v, w, b);
  }
  a = block_sum(a, red);
  __syncthreads();
  b = block_sum(b, red);
  if (threadIdx.x == 0) { ab[k] = a; ab[KMAX + k] = b; }
}
__global__ __launch_bounds__(NT) void k_interval_sums(const double* __restrict__ sH, const double* __restrict__ sV, long long nd,
                                                      const double* __restrict__ knots, const int* __restrict__ Mp,
                                                      double* __restrict__ ab) {
  __shared__ double red[NW];
  interval_sums_body(sH, sV, nd, knots, Mp, ab, red);
}
__global__ __launch_bounds__(NT) void k_interval_sums_b(Pools P, int g0, int g_last, long long lo, long long end_all,
                                                        const double* __restrict__ sHall, const double* __restrict__ sVall,
                                                        const double* __restrict__ knots_all, const int* __restrict__ M_all,
                                                        double* __restrict__ ab_all) {
  __shared__ double red[NW];
  const int gidx = g0 + blockIdx.y;
  const long long off = P.gd[gidx].offd - lo;
  interval_sums_body(sHall + off, sVall + off, seg_len(P, gidx, g_last, end_all), knots_all + (size_t)blockIdx.y * KMAX,
                     M_all + blockIdx.y, ab_all + (size_t)blockIdx.y * 2 * KMAX, red);
}

// ---- the whole batch in one sequence of launches --------------------------------------------------------------------
// ONE radix sort of ALL dual nodes of the glaciers [g0, g0 + ng) by the composite key (glacier, Hbar) (k_interp_keys; values:
// node index): every glacier's nodes, sorted by Hbar, at its own pooled offset again.  (Round 3 sorted by Hbar and then
// stably by glacier: 10 instead of 7 sort passes and a third more of rocPRIM's memsets.)  Then knots and interval sums with one block row per glacier, and the contraction with the knot gradients
// as ONE wave-reduced backprop per glacier whose lane weights are the knot coefficients c_k (dtheta = sum_k c_k dY/dtheta
// (T, knot_k) is the gradient of sum_k c_k Y(T, knot_k)): ~20 launches per evaluation whatever the number of glaciers,
// instead of 23 per glacier.
__global__ void k_fill_gid(Pools P, int G, long long ntotd, unsigned* __restrict__ gid, unsigned* __restrict__ iota) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ntotd; i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = G - 1;  // last glacier with offd <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (P.gd[mid].offd <= i) lo = mid; else hi = mid - 1;
    }
    gid[i] = (unsigned)lo;
    iota[i] = (unsigned)i;
  }
}
__global__ void k_gather_gid(const unsigned* __restrict__ gid, const unsigned* __restrict__ idx, long long n,
                             unsigned* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = gid[idx[i]];
}
__global__ void k_gather_HV(const double* __restrict__ H, const double* __restrict__ V, const unsigned* __restrict__ idx, long long n,
                            double* __restrict__ sH, double* __restrict__ sV) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned j = idx[i];
    sH[i] = H[j];
    sV[i] = V[j];
  }
}
// block g: dth[(g0 + g) P + q] (+)= sum_k c_k dY/dtheta_q(T_g, knot_k); dynamic LDS: NW x P accumulators + P ints
template <class AR, bool FIXED>
__global__ __launch_bounds__(NT) void k_knot_backprop(Pools P, LawDev L, int g0, const double* __restrict__ knots_all,
                                                      const int* __restrict__ M_all, const double* __restrict__ ab_all,
                                                      double* __restrict__ dth, int accumulate) {
  extern __shared__ double kb_dyn[];
  __shared__ double stage[NW][WG_SLOTS][WG_LD];
  double* accs = kb_dyn;
  int* order = reinterpret_cast<int*>(kb_dyn + (size_t)NW * L.P);
  const int lane = threadIdx.x & 63, w = wave_id();
  for (int k = threadIdx.x; k < NW * L.P; k += NT) accs[k] = 0.0;
  if (threadIdx.x == 0) mlp_grad_order(L, order);
  __syncthreads();
  const WaveAcc A{stage[w], order, accs + (size_t)w * L.P};
  const int M = M_all[blockIdx.x];
  const double* knots = knots_all + (size_t)blockIdx.x * KMAX;
  const double* ab = ab_all + (size_t)blockIdx.x * 2 * KMAX;
  const double T = P.gd[g0 + blockIdx.x].T;
  for (int c = w; c * 64 < M; c += NW) {
    const int k = c * 64 + lane;
    double ck = 0.0, x = 0.0;
    if (k < M) {
      ck = (k < M - 1 ? ab[k] : 0.0) + (k > 0 ? ab[KMAX + k - 1] : 0.0);
      x = knots[k];
    }
    mlp_grad_wave<AR, FIXED>(L, T, x, ck, A, lane);
  }
  __syncthreads();
  double* out = dth + (size_t)(g0 + blockIdx.x) * L.P;
  for (int q = threadIdx.x; q < L.P; q += NT) {
    double s = 0.0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) s += accs[(size_t)ww * L.P + q];
    out[q] = accumulate ? out[q] + s : s;
  }
}

__global__ __launch_bounds__(64) void k_interp_contract(int P, const int* __restrict__ Mp, const double* __restrict__ ab,
                                                        const double* __restrict__ G, double* __restrict__ dth, int accumulate) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= P) return;
  const int M = *Mp;
  double s = 0.0;
  for (int k = 0; k < M; ++k) {
    const double ck = (k < M - 1 ? ab[k] : 0.0) + (k > 0 ? ab[KMAX + k - 1] : 0.0);
    s = fma(ck, G[(long long)q * KMAX + k], s);
  }
  dth[q] = accumulate ? dth[q] + s : s;
}

// ---- U law (SIA2D_D_target(interpolation = :Linear), target_D_pure.jl:179-193; node gradients: p_VJP!, Laws.jl:153-169) -------
// The gradient interpolant lives on a FIXED K x K node grid, K = 2 n_interp_half, both axes LinRange(0, 100, K) (the law's
// cache is built with the Hbar nodes on the slope axis too, Laws.jl:137-142), bilinear in (Hbar, |grad S|).  As in the Y-law
// path the contraction with the node weights becomes a sum over GRID nodes:  dtheta = sum_ij c_ij dU/dtheta(h_i, s_j),
// c_ij = sum_nodes v * (tent_i(Hbar) tent_j(|grad S|)).  Deterministic, without atomics: the dual nodes are sorted by the
// grid cell they fall in (rocPRIM, stable), one workgroup sums the four corner weights of a cell in a fixed order, the
// grid nodes gather their (up to) four cells, and only grid nodes with c_ij != 0 are differentiated.
constexpr double UNODE_MAX = 100.0;

__device__ __forceinline__ double unode(int k, int K) {  // LinRange(0.0, 100, K)[k]
  const double t = (double)k / (double)(K - 1);
  return (1.0 - t) * 0.0 + t * UNODE_MAX;
}
__device__ __forceinline__ int ucell(double x, int K) {  // k with node(k) <= x < node(k+1); the last cell is closed
  int k = (int)(x * ((double)(K - 1) / UNODE_MAX));
  k = k < 0 ? 0 : (k > K - 2 ? K - 2 : k);
  while (k > 0 && unode(k, K) > x) --k;
  while (k < K - 2 && unode(k + 1, K) <= x) ++k;
  return k;
}

// key = cell index of every dual node; a node outside [0, 100]^2 raises the flag (Gridded(Linear()) throws)
__global__ __launch_bounds__(NT) void k_ucell_keys(const double* __restrict__ H, const double* __restrict__ S, long long nd, int K,
                                                   unsigned* __restrict__ keys, unsigned* __restrict__ idx, int* __restrict__ err) {
  const long long q = (long long)blockIdx.x * NT + threadIdx.x;
  if (q >= nd) return;
  const double x = H[q], y = S[q];
  if (!(x >= 0.0 && x <= UNODE_MAX && y >= 0.0 && y <= UNODE_MAX)) {
    atomicOr(err, 1);
    keys[q] = 0u; idx[q] = (unsigned)q;
    return;
  }
  keys[q] = (unsigned)(ucell(x, K) * (K - 1) + ucell(y, K));
  idx[q] = (unsigned)q;
}

__device__ __forceinline__ long long lower_bound_u(const unsigned* __restrict__ a, long long n, unsigned x) {
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// block c: the dual nodes of grid cell c = iH (K-1) + iS; cell4[4c + {0,1,2,3}] = sum v {(1-wh)(1-ws), wh(1-ws), (1-wh)ws, wh ws}
__global__ __launch_bounds__(NT) void k_ucell_sums(const unsigned* __restrict__ skeys, const unsigned* __restrict__ sidx, long long nd,
                                                   const double* __restrict__ H, const double* __restrict__ S,
                                                   const double* __restrict__ V, int K, double* __restrict__ cell4) {
  __shared__ double red[NW];
  const unsigned c = blockIdx.x;
  const long long lo = lower_bound_u(skeys, nd, c), hi = lower_bound_u(skeys, nd, c + 1u);
  if (lo == hi) {
    if (threadIdx.x < 4) cell4[4 * (long long)c + threadIdx.x] = 0.0;
    return;
  }
  const int iH = (int)(c / (unsigned)(K - 1)), iS = (int)(c % (unsigned)(K - 1));
  const double h0 = unode(iH, K), s0 = unode(iS, K);
  const double ih = 1.0 / (unode(iH + 1, K) - h0), is = 1.0 / (unode(iS + 1, K) - s0);
  double a00 = 0.0, a10 = 0.0, a01 = 0.0, a11 = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += NT) {
    const unsigned q = sidx[i];
    const double wh = (H[q] - h0) * ih, ws = (S[q] - s0) * is, v = V[q];
    a00 = fma(v, (1.0 - wh) * (1.0 - ws), a00);
    a10 = fma(v, wh * (1.0 - ws), a10);
    a01 = fma(v, (1.0 - wh) * ws, a01);
    a11 = fma(v, wh * ws, a11);
  }
  a00 = block_sum(a00, red); __syncthreads();
  a10 = block_sum(a10, red); __syncthreads();
  a01 = block_sum(a01, red); __syncthreads();
  a11 = block_sum(a11, red);
  if (threadIdx.x == 0) {
    cell4[4 * (long long)c + 0] = a00; cell4[4 * (long long)c + 1] = a10;
    cell4[4 * (long long)c + 2] = a01; cell4[4 * (long long)c + 3] = a11;
  }
}

// slot s differentiates the grid nodes s, s + KMAX, ... whose coefficient is not zero: G[q * KMAX + s] = sum c_ij dU/dtheta_q
__global__ __launch_bounds__(64) void k_unode_grads(LawDev L, int K, const double* __restrict__ cell4, double* __restrict__ G) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= KMAX) return;
  for (int q = 0; q < L.P; ++q) G[(long long)q * KMAX + s] = 0.0;
  const int C = K - 1;
  for (int node = s; node < K * K; node += KMAX) {
    const int i = node / K, j = node - i * K;
    double c = 0.0;
    if (i < C && j < C) c += cell4[4 * ((long long)i * C + j) + 0];
    if (i > 0 && j < C) c += cell4[4 * ((long long)(i - 1) * C + j) + 1];
    if (i < C && j > 0) c += cell4[4 * ((long long)i * C + (j - 1)) + 2];
    if (i > 0 && j > 0) c += cell4[4 * ((long long)(i - 1) * C + (j - 1)) + 3];
    if (c != 0.0) mlp_grad(L, unode(i, K), unode(j, K), c, G + s, KMAX);
  }
}

__global__ __launch_bounds__(64) void k_slot_sum(int P, const double* __restrict__ G, double* __restrict__ dth, int accumulate) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= P) return;
  double s = 0.0;
  for (int k = 0; k < KMAX; ++k) s += G[(long long)q * KMAX + k];
  dth[q] = accumulate ? dth[q] + s : s;
}

// ---- the same corner sums WITHOUT a sort (round 6) --------------------------------------------------------------------------------
// The sort only grouped the dual nodes by grid cell so that a workgroup could add a cell's weights in a fixed order.  Order-free
// addition does not need the grouping: every node adds its four corner weights into the cell's accumulators with 64-bit INTEGER
// atomics -- fixed point, two limbs at 2^(e - 35) and 2^(e - 71) of the launch's largest |v| (2^e > max |v|: one reduction pass first),
// exact for 2^27 terms per cell and 2^19 x finer than a double's ulp of the largest weight: bitwise repeatable although atomic
// (the scheme of k_sel_sums, with head-room for 64 x 1024^2 nodes in one cell).  Nodes with v == 0 (no ice) add nothing.
__global__ __launch_bounds__(256) void k_ucell_vmax(const double* __restrict__ V, long long nd, unsigned long long* __restrict__ vmax) {
  double m = 0.0;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nd; q += (long long)gridDim.x * 256) m = fmax(m, fabs(V[q]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(vmax, (unsigned long long)__double_as_longlong(m));  // (positive doubles order like their patterns)
}
__device__ __forceinline__ bool ucell_scales(double vmax, double& s1, double& s2) {
  if (!(vmax > 0x1p-800)) return false;
  int e;
  (void)frexp(vmax, &e);
  s1 = ldexp(1.0, e - 35);
  s2 = ldexp(1.0, e - 71);
  return true;
}
__global__ __launch_bounds__(256) void k_ucell_accum(const double* __restrict__ H, const double* __restrict__ S, const double* __restrict__ V,
                                                     long long nd, int K, const unsigned long long* __restrict__ vmax,
                                                     unsigned long long* __restrict__ bins, int* __restrict__ err) {
  double s1, s2;
  const bool on = ucell_scales(__longlong_as_double((long long)*vmax), s1, s2);
  const double i1 = on ? 1.0 / s1 : 0.0, i2 = on ? 1.0 / s2 : 0.0;  // (powers of two: exact)
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nd; q += (long long)gridDim.x * 256) {
    const double x = H[q], y = S[q];
    if (!(x >= 0.0 && x <= UNODE_MAX && y >= 0.0 && y <= UNODE_MAX)) { atomicOr(err, 1); continue; }  // Gridded(Linear()) throws
    const double v = V[q];
    if (!on || v == 0.0) continue;
    const int iH = ucell(x, K), iS = ucell(y, K);
    const double h0 = unode(iH, K), s0 = unode(iS, K);
    const double ih = 1.0 / (unode(iH + 1, K) - h0), is = 1.0 / (unode(iS + 1, K) - s0);
    const double wh = (x - h0) * ih, ws = (y - s0) * is;   // k_ucell_sums' expressions
    unsigned long long* b = bins + 8 * ((size_t)iH * (K - 1) + iS);
    auto add = [&](int slot, double t) {
      const double hi = rint(t * i1);
      const double lo = rint(fma(-hi, s1, t) * i2);
      if (hi != 0.0) atomicAdd(b + 2 * slot, (unsigned long long)(long long)hi);
      if (lo != 0.0) atomicAdd(b + 2 * slot + 1, (unsigned long long)(long long)lo);
    };
    add(0, v * ((1.0 - wh) * (1.0 - ws)));
    add(1, v * (wh * (1.0 - ws)));
    add(2, v * ((1.0 - wh) * ws));
    add(3, v * (wh * ws));
  }
}
__global__ __launch_bounds__(256) void k_ucell_finish(int ncell4, const unsigned long long* __restrict__ vmax, const unsigned long long* __restrict__ bins,
                                                      double* __restrict__ cell4) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncell4) return;
  double s1, s2, a = 0.0;
  if (ucell_scales(__longlong_as_double((long long)*vmax), s1, s2))
    a = fma((double)(long long)bins[2 * (size_t)k], s1, (double)(long long)bins[2 * (size_t)k + 1] * s2);
  cell4[k] = a;
}

size_t interp_sort_temp_bytes(long long nd_max) {
  size_t bytes = 0, bytes_u = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const double*)nullptr, (double*)nullptr, (const double*)nullptr,
                                  (double*)nullptr, (size_t)nd_max, 0, 64, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, bytes_u, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                  (unsigned*)nullptr, (size_t)nd_max, 0, 32, nullptr);
  return bytes > bytes_u ? bytes : bytes_u;
}

// scratch: sA, sB (nd doubles each: hold the unsorted / sorted {key, index} pairs), tmp, cell4 (4 (K-1)^2), G (P * KMAX),
// err (int, sticky: set when a node lies outside the interpolant's domain)
int launch_interp_theta_U(hipStream_t st, const LawDev& L, int n_half, const double* nodeH, const double* nodeS, const double* nodeV,
                          long long nd, double* sA, double* sB, void* tmp, size_t tmp_bytes, double* cell4, double* G, int* err,
                          double* dth, int accumulate) {
  const int K = 2 * n_half;
  if (K > KMAX || n_half < 2 || nd >= (1ll << 32)) return 1;
  const char* esel = std::getenv("ODINN_INTERP_SELECT");  // (the Y law's switch; read per call: tests toggle it)
  const bool sorted = esel && esel[0] == '0';
  if (!sorted) {
    // behind cell4 (4 (KMAX - 1)^2 doubles): the fixed-point limbs, 8 words per cell, then the launch's max |v|
    unsigned long long* bins = reinterpret_cast<unsigned long long*>(cell4 + (size_t)4 * (KMAX - 1) * (KMAX - 1));
    unsigned long long* vmax = bins + (size_t)8 * (KMAX - 1) * (KMAX - 1);
    const size_t nb = (size_t)8 * (K - 1) * (K - 1);
    if (hipMemsetAsync(bins, 0, nb * sizeof(unsigned long long), st) != hipSuccess) return 2;
    if (hipMemsetAsync(vmax, 0, sizeof(unsigned long long), st) != hipSuccess) return 2;
    const unsigned nblk = (unsigned)std::min<long long>((nd + 255) / 256, 4096);
    hipLaunchKernelGGL(k_ucell_vmax, dim3(nblk), dim3(256), 0, st, nodeV, nd, vmax);
    hipLaunchKernelGGL(k_ucell_accum, dim3(nblk), dim3(256), 0, st, nodeH, nodeS, nodeV, nd, K, vmax, bins, err);
    const int n4 = 4 * (K - 1) * (K - 1);
    hipLaunchKernelGGL(k_ucell_finish, dim3((n4 + 255) / 256), dim3(256), 0, st, n4, vmax, bins, cell4);
    hipLaunchKernelGGL(k_unode_grads, dim3(KMAX / 64), dim3(64), 0, st, L, K, cell4, G);
    hipLaunchKernelGGL(k_slot_sum, dim3((L.P + 63) / 64), dim3(64), 0, st, L.P, G, dth, accumulate);
    return 0;
  }
  unsigned* keys = reinterpret_cast<unsigned*>(sA);
  unsigned* idx = keys + nd;
  unsigned* skeys = reinterpret_cast<unsigned*>(sB);
  unsigned* sidx = skeys + nd;
  hipLaunchKernelGGL(k_ucell_keys, dim3((unsigned)((nd + NT - 1) / NT)), dim3(NT), 0, st, nodeH, nodeS, nd, K, keys, idx, err);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, skeys, idx, sidx, (size_t)nd, 0, 32, st) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_ucell_sums, dim3((unsigned)((K - 1) * (K - 1))), dim3(NT), 0, st, skeys, sidx, nd, nodeH, nodeS, nodeV, K, cell4);
  hipLaunchKernelGGL(k_unode_grads, dim3(KMAX / 64), dim3(64), 0, st, L, K, cell4, G);
  hipLaunchKernelGGL(k_slot_sum, dim3((L.P + 63) / 64), dim3(64), 0, st, L.P, G, dth, accumulate);
  return 0;
}

template <class AR>
static bool interp_law_is(const LawDev& L) {
  if (L.n_layers != AR::NL) return false;
  for (int l = 0; l <= AR::NL; ++l) if (L.widths[l] != AR::W[l]) return false;
  for (int l = 0; l < AR::NL; ++l) if (L.acts[l] != AR::A[l]) return false;
  return true;
}
size_t interp_batch_temp_bytes(long long n_max) {
  size_t b1 = 0, b2 = 0, b3 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b1, (const double*)nullptr, (double*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr,
                                  (size_t)n_max, 0, 64, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, b2, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                  (unsigned*)nullptr, (size_t)n_max, 0, 32, nullptr);
  (void)rocprim::radix_sort_pairs(nullptr, b3, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                  (const unsigned*)nullptr, (unsigned*)nullptr, (size_t)n_max, 0, 64, nullptr);
  return std::max(b1, std::max(b2, b3));
}
// composite sort key of a dual node: glacier index (relative to g0) in the top gbits bits, below it the bit pattern of
// Hbar * 2^-e.  Positive doubles order like their patterns, and multiplying by a power of two is EXACT (the mantissa is
// untouched) as long as the result stays normal: e is chosen so that every thickness below 2^15 m lands in the lowest
// 2^(12 - gbits) binades, i.e. its pattern leaves the top gbits bits free.  ONE radix sort then leaves every glacier's nodes
// in exact Hbar order at the glacier's pooled offset.  With gbits <= 6 (64 glaciers per call) the exact range reaches down to
// 2^-48 m = 3.6e-15 m; thinner ice (the subnormal thicknesses an advancing margin leaves) keeps its order up to the rounding
// of the scaled value and stays behind the exact zeros.  More than 64 glaciers per call sort twice (by Hbar, then by glacier).
// (6, not 7: with 7 glacier bits the scaled thicknesses below 2^-16 m turn subnormal and lose mantissa bits -- distinct values tie
//  and come out in index order; batches of more than 64 glaciers take the sort-free contraction or the two-pass sort)
constexpr int INTERP_KEY_GBITS_MAX = 6;
__global__ void k_interp_keys(const unsigned* __restrict__ gid, const double* __restrict__ H, long long n, unsigned g0, int gbits,
                              double scale, unsigned long long* __restrict__ keys) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double h = fmin(fmax(H[i], 0.0), 32767.0);
    unsigned long long low = (unsigned long long)__double_as_longlong(h * scale);
    if (h > 0.0 && !low) low = 1;  // (underflow to zero: still behind the exact zeros)
    keys[i] = gbits ? (((unsigned long long)(gid[i] - g0) << (64 - gbits)) | low) : low;
  }
}
void launch_fill_gid(hipStream_t st, Pools P, int G, long long ntotd, unsigned* gid, unsigned* iota) {
  hipLaunchKernelGGL(k_fill_gid, dim3(1024), dim3(256), 0, st, P, G, ntotd, gid, iota);
}
size_t interp_batch_lds_bytes(int P) { return (size_t)NW * P * sizeof(double) + (size_t)P * sizeof(int); }
// the glaciers [g0, g0 + ng) whose dual nodes are the pooled range [lo, lo + n): nodeH / nodeV / gid point at the POOL's first
// node, iota holds 0, 1, 2, ...; scratch (n entries each): sH, sV, iA, iB, kA, kB; knots (ng KMAX), M (ng), ab (ng 2 KMAX)
int launch_interp_theta_batch(hipStream_t st, Pools P, const LawDev& L, int n_half, int g0, int ng, long long lo, long long n,
                              const double* nodeH, const double* nodeV, const unsigned* gid, const unsigned* iota, double* sH,
                              double* sV, unsigned* iA, unsigned* iB, unsigned* kA, unsigned* kB, void* tmp, size_t tmp_bytes,
                              double* knots, int* M, double* ab, double* dth, int accumulate) {
  if (2 * n_half > KMAX || n_half < 2 || n >= (1ll << 32)) return 1;
  const unsigned nb = (unsigned)std::min<long long>((n + 255) / 256, 4096);
  const unsigned* order = iA;
  int bits = 0;
  while ((1ll << bits) < (long long)ng) ++bits;
  if (bits <= INTERP_KEY_GBITS_MAX) {
    // one sort by (glacier, Hbar): the keys live in sH / sV, which k_gather_HV fills only afterwards
    unsigned long long* kin = reinterpret_cast<unsigned long long*>(sH);
    unsigned long long* kout = reinterpret_cast<unsigned long long*>(sV);
    const double scale = bits ? std::ldexp(1.0, -(1023 - (1 << (12 - bits)) + 15)) : 1.0;
    hipLaunchKernelGGL(k_interp_keys, dim3(nb), dim3(256), 0, st, gid + lo, nodeH + lo, n, (unsigned)g0, bits, scale, kin);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, iota, iA, (size_t)n, 0, 64, st) != hipSuccess) return 2;
  } else {  // (thousands of glaciers in one call: sort by Hbar, then stably by glacier)
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, nodeH + lo, sH, iota, iA, (size_t)n, 0, 64, st) != hipSuccess) return 2;
    int gb = 1;
    while ((1ll << gb) < (long long)(g0 + ng)) ++gb;
    hipLaunchKernelGGL(k_gather_gid, dim3(nb), dim3(256), 0, st, gid + lo, iA, n, kA);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, kA, kB, iA, iB, (size_t)n, 0, gb, st) != hipSuccess) return 2;
    order = iB;
  }
  hipLaunchKernelGGL(k_gather_HV, dim3(nb), dim3(256), 0, st, nodeH + lo, nodeV + lo, order, n, sH, sV);
  hipLaunchKernelGGL(k_knots_b, dim3(ng), dim3(KMAX), 0, st, P, g0, g0 + ng - 1, lo, lo + n, sH, n_half, knots, M);
  hipLaunchKernelGGL(k_interval_sums_b, dim3(2 * n_half, ng), dim3(NT), 0, st, P, g0, g0 + ng - 1, lo, lo + n, sH, sV, knots, M, ab);
  const size_t dyn = interp_batch_lds_bytes(L.P);
  if (interp_law_is<ArchDef>(L))
    hipLaunchKernelGGL((k_knot_backprop<ArchDef, true>), dim3(ng), dim3(NT), dyn, st, P, L, g0, knots, M, ab, dth, accumulate);
  else if (interp_law_is<Arch16>(L))
    hipLaunchKernelGGL((k_knot_backprop<Arch16, true>), dim3(ng), dim3(NT), dyn, st, P, L, g0, knots, M, ab, dth, accumulate);
  else
    hipLaunchKernelGGL((k_knot_backprop<ArchRT, false>), dim3(ng), dim3(NT), dyn, st, P, L, g0, knots, M, ab, dth, accumulate);
  return 0;
}

// ---- the same over the ACTIVE nodes only ------------------------------------------------------------------------------------------
// Inside one gradient evaluation every H the contraction sees is a snapshot of the forward solve or a convex combination of two
// of them, so a dual node none of whose four cells carries ice in ANY snapshot has Hbar = 0 and weight 0 at every stop: it is neither
// among the quantiles (strictly inside (0, max)) nor in an interval sum.  launch_interp_active builds the list of the other nodes
// once per gradient (flags from the snapshot store, a stable rocprim::select, the glaciers' offsets in the list); the per-stop
// sequence then sorts, gathers and sums n_act entries instead of all dual nodes -- half of them for an ice cap on its square grid,
// a quarter for an alpine glacier.  Same keys, same stable sort: the active nodes come out in the order they have in the full sort.
__global__ void k_active_flags(Pools P, const unsigned* __restrict__ gid, const double* __restrict__ snaps, int nslots, long long ntot,
                               long long ntotd, unsigned char* __restrict__ flags) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < ntotd; q += (long long)gridDim.x * blockDim.x) {
    const GDev& g = P.gd[gid[q]];
    const long long li = q - g.offd;
    const int a = (int)(li % (g.nx - 1)), b = (int)(li / (g.nx - 1));
    unsigned char f = 0;
    if (b <= g.ny - 2) {
      const long long c0 = g.off + a + (long long)g.nx * b;
      for (int s = 0; s < nslots && !f; ++s) {
        const double* __restrict__ H = snaps + (long long)s * ntot;
        f = (H[c0] > 0.0 || H[c0 + 1] > 0.0 || H[c0 + g.nx] > 0.0 || H[c0 + g.nx + 1] > 0.0) ? 1 : 0;
      }
    }
    flags[q] = f;
  }
}
__global__ void k_active_offsets(Pools P, int G, const unsigned* __restrict__ act, const unsigned* __restrict__ n_act_p,
                                 long long* __restrict__ aoff) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > G) return;
  const long long n = (long long)*n_act_p;
  if (g == G) { aoff[G] = n; return; }
  const long long target = P.gd[g].offd;  // first active entry with node index >= offd_g
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)act[mid] < target) lo = mid + 1; else hi = mid;
  }
  aoff[g] = lo;
}
__global__ void k_gather_gid_n(const unsigned* __restrict__ gid, const unsigned* __restrict__ idx, const unsigned* __restrict__ n_p,
                               unsigned* __restrict__ out) {
  const long long n = (long long)*n_p;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = gid[idx[i]];
}
size_t interp_active_temp_bytes(long long n) {
  size_t bytes = 0;
  (void)rocprim::select(nullptr, bytes, (const unsigned*)nullptr, (const unsigned char*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                        (size_t)n, (hipStream_t) nullptr, false);
  return bytes;
}
// act[0 .. *n_act) = the active dual nodes in ascending order, gid_act their glaciers, aoff[0 .. G] the glaciers' offsets in act
int launch_interp_active(hipStream_t st, Pools P, int G, long long ntotd, const unsigned* gid, const unsigned* iota, const double* snaps,
                         int nslots, long long ntot, unsigned char* flags, void* tmp, size_t tmp_bytes, unsigned* act, unsigned* gid_act,
                         long long* aoff, unsigned* n_act_dev) {
  const unsigned nb = (unsigned)std::min<long long>((ntotd + 255) / 256, 4096);
  hipLaunchKernelGGL(k_active_flags, dim3(nb), dim3(256), 0, st, P, gid, snaps, nslots, ntot, ntotd, flags);
  if (rocprim::select(tmp, tmp_bytes, iota, flags, act, n_act_dev, (size_t)ntotd, st, false) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_active_offsets, dim3((G + 1 + 63) / 64), dim3(64), 0, st, P, G, act, n_act_dev, aoff);
  hipLaunchKernelGGL(k_gather_gid_n, dim3(nb), dim3(256), 0, st, gid, act, n_act_dev, gid_act);
  return 0;
}
__global__ void k_interp_keys_act(const unsigned* __restrict__ gid_act, const unsigned* __restrict__ act, const double* __restrict__ H,
                                  long long n, int gbits, double scale, unsigned long long* __restrict__ keys) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double h = fmin(fmax(H[act[i]], 0.0), 32767.0);
    unsigned long long low = (unsigned long long)__double_as_longlong(h * scale);
    if (h > 0.0 && !low) low = 1;
    keys[i] = gbits ? (((unsigned long long)gid_act[i] << (64 - gbits)) | low) : low;
  }
}
__global__ void k_gather_HV_act(const double* __restrict__ H, const double* __restrict__ V, const unsigned* __restrict__ act,
                                const unsigned* __restrict__ idx, long long n, double* __restrict__ sH, double* __restrict__ sV) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned j = act[idx[i]];
    sH[i] = H[j];
    sV[i] = V[j];
  }
}
__global__ __launch_bounds__(KMAX) void k_knots_a(const long long* __restrict__ aoff, const double* __restrict__ sHall, int n,
                                                  double* __restrict__ knots_all, int* __restrict__ M_all) {
  __shared__ double c[KMAX];
  __shared__ int first[KMAX];
  knots_body(sHall + aoff[blockIdx.x], aoff[blockIdx.x + 1] - aoff[blockIdx.x], n, knots_all + (size_t)blockIdx.x * KMAX, M_all + blockIdx.x,
             c, first);
}
__global__ __launch_bounds__(NT) void k_interval_sums_a(const long long* __restrict__ aoff, const double* __restrict__ sHall,
                                                        const double* __restrict__ sVall, const double* __restrict__ knots_all,
                                                        const int* __restrict__ M_all, double* __restrict__ ab_all) {
  __shared__ double red[NW];
  const long long off = aoff[blockIdx.y];
  interval_sums_body(sHall + off, sVall + off, aoff[blockIdx.y + 1] - off, knots_all + (size_t)blockIdx.y * KMAX, M_all + blockIdx.y,
                     ab_all + (size_t)blockIdx.y * 2 * KMAX, red);
}
// all G glaciers of the batch; n_act entries (host copy of *n_act_dev); iota holds 0, 1, 2, ...
int launch_interp_theta_active(hipStream_t st, Pools P, const LawDev& L, int n_half, int G, long long n_act, const double* nodeH,
                               const double* nodeV, const unsigned* act, const unsigned* gid_act, const long long* aoff,
                               const unsigned* iota, double* sH, double* sV, unsigned* iA, void* tmp, size_t tmp_bytes, double* knots, int* M,
                               double* ab, double* dth, int accumulate) {
  int bits = 0;
  while ((1ll << bits) < (long long)G) ++bits;
  if (2 * n_half > KMAX || n_half < 2 || n_act < 1 || n_act >= (1ll << 32) || bits > INTERP_KEY_GBITS_MAX) return 1;
  const unsigned nb = (unsigned)std::min<long long>((n_act + 255) / 256, 4096);
  unsigned long long* kin = reinterpret_cast<unsigned long long*>(sH);
  unsigned long long* kout = reinterpret_cast<unsigned long long*>(sV);
  const double scale = bits ? std::ldexp(1.0, -(1023 - (1 << (12 - bits)) + 15)) : 1.0;
  hipLaunchKernelGGL(k_interp_keys_act, dim3(nb), dim3(256), 0, st, gid_act, act, nodeH, n_act, bits, scale, kin);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, iota, iA, (size_t)n_act, 0, 64, st) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_gather_HV_act, dim3(nb), dim3(256), 0, st, nodeH, nodeV, act, iA, n_act, sH, sV);
  hipLaunchKernelGGL(k_knots_a, dim3(G), dim3(KMAX), 0, st, aoff, sH, n_half, knots, M);
  hipLaunchKernelGGL(k_interval_sums_a, dim3(2 * n_half, G), dim3(NT), 0, st, aoff, sH, sV, knots, M, ab);
  const size_t dyn = interp_batch_lds_bytes(L.P);
  if (interp_law_is<ArchDef>(L))
    hipLaunchKernelGGL((k_knot_backprop<ArchDef, true>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  else if (interp_law_is<Arch16>(L))
    hipLaunchKernelGGL((k_knot_backprop<Arch16, true>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  else
    hipLaunchKernelGGL((k_knot_backprop<ArchRT, false>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  return 0;
}

// ---- the same contraction WITHOUT a sort: order statistics by selection, interval sums by exact binning ----------------------------
// What the contraction needs from "the nodes in Hbar order" is (a) 2 n order statistics of Hbar per glacier (the type-7 quantiles read
// sH[lo + j] and sH[lo + j + 1] at n probabilities) and (b) per knot interval the sums of v (1 - w) and v w.  Neither needs the
// permutation:
//   (a) SELECTION.  k_sel_max: amax = max Hbar per glacier (atomicMax on the bit patterns).  k_sel_hist: a histogram of the values
//       strictly inside (0, amax) over SEL_NB equal-width bins of [0, amax) (integer atomics: order-free).  k_sel_targets: the scanned
//       histogram gives, for every rank the quantiles ask for, its bin and its rank inside the bin.  k_sel_gather: the members of those
//       (<= 2 n) bins are copied out (in any order -- a multiset).  k_sel_pick: one workgroup per target selects the exact order
//       statistic inside its bin: MSB-first radix selection on the bit patterns until <= SEL_LDS candidates remain, then a rank count
//       in LDS.  EXACT: the same doubles the sorted array holds at those ranks, so the knots are bit-identical to the sort's.
//   (b) BINNING.  Every node finds its knot interval by binary search and adds its two terms into the interval's accumulators.  To keep
//       the sums independent of the order of the additions (run-to-run bitwise reproducibility, which the sorted, fixed-order
//       sums had) the accumulators are FIXED-POINT: a term x is split exactly into hi = rint(x / s1), lo = rint((x - hi s1) / s2)
//       with s1 = 2^(e - 39), s2 = 2^(e - 79), 2^e > max |v| of the glacier; hi and lo are added with 64-bit integer atomics (LDS per
//       workgroup, then global), |hi|, |lo| < 2^40 per term: exact integer sums for up to 2^23 terms per interval per glacier, the
//       only rounding is the 2^-80 max|v| quantisation of lo -- 2^27 times finer than a double's ulp at max |v|.
// Per evaluation: 9 launches and ~48 B per active node instead of the 64-bit-key radix / merge sort (8 passes of 32 B per node or
// ~12 merge launches), its gathers and its memsets.  ODINN_INTERP_SELECT=0 keeps the sort (A/B, and the knot-identity test).
constexpr int SEL_NB = 8192;    // histogram bins per glacier
constexpr int SEL_LDS = 256;    // candidates the final rank count handles in LDS (one per thread of k_sel_pick)
constexpr int SEL_BLK_MAX = 64; // workgroups per glacier of the streaming kernels

__device__ __forceinline__ int sel_bin(double h, double inv) {
  int k = (int)(h * inv);
  return k < 0 ? 0 : (k > SEL_NB - 1 ? SEL_NB - 1 : k);
}
// amax[g] = max Hbar, vmax[g] = max |v| over the glacier's active nodes (bit patterns of non-negative doubles order like the values)
__global__ __launch_bounds__(256) void k_sel_max(const long long* __restrict__ aoff, const unsigned* __restrict__ act,
                                                 const double* __restrict__ H, const double* __restrict__ V,
                                                 unsigned long long* __restrict__ amax, unsigned long long* __restrict__ vmax) {
  const int g = blockIdx.y;
  const long long lo = aoff[g], hi = aoff[g + 1];
  double mh = 0.0, mv = 0.0;
  for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
    const unsigned q = act[i];
    mh = fmax(mh, H[q]);
    mv = fmax(mv, fabs(V[q]));
  }
  mh = wave_max(mh); mv = wave_max(mv);
  if ((threadIdx.x & 63) == 0) {
    if (mh > 0.0) atomicMax(amax + g, (unsigned long long)__double_as_longlong(mh));
    if (mv > 0.0) atomicMax(vmax + g, (unsigned long long)__double_as_longlong(mv));
  }
}
__global__ __launch_bounds__(256) void k_sel_hist(const long long* __restrict__ aoff, const unsigned* __restrict__ act,
                                                  const double* __restrict__ H, const unsigned long long* __restrict__ amax,
                                                  unsigned* __restrict__ hist) {
  __shared__ unsigned sh[SEL_NB];
  const int g = blockIdx.y;
  const double am = __longlong_as_double((long long)amax[g]);
  if (!(am > 0.0)) return;
  for (int k = threadIdx.x; k < SEL_NB; k += 256) sh[k] = 0u;
  __syncthreads();
  const double inv = (double)SEL_NB / am;
  const long long lo = aoff[g], hi = aoff[g + 1];
  for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
    const double h = H[act[i]];
    if (h > 0.0 && h < am) atomicAdd(&sh[sel_bin(h, inv)], 1u);
  }
  __syncthreads();
  unsigned* out = hist + (size_t)g * SEL_NB;
  for (int k = threadIdx.x; k < SEL_NB; k += 256)
    if (sh[k]) atomicAdd(out + k, sh[k]);
}
// one workgroup per glacier: m = #values strictly inside (0, amax); for every quantile target its two ranks' (bin, rank in bin);
// goff[bin] = offset of a needed bin's members in the glacier's gather area (-1: not needed)
__global__ __launch_bounds__(1024) void k_sel_targets(const unsigned* __restrict__ hist, int n, unsigned* __restrict__ m_all,
                                                      int* __restrict__ goff, int4* __restrict__ tgt) {
  __shared__ unsigned cum[SEL_NB];   // inclusive
  __shared__ unsigned char need[SEL_NB];
  __shared__ unsigned wsum[16];
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  constexpr int PER = SEL_NB / 1024;
  const unsigned* hg = hist + (size_t)g * SEL_NB;
  auto block_excl_scan = [&](unsigned mine, unsigned& total) {  // exclusive prefix of one value per thread (fixed shape)
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < w) base += wsum[k];
      tot += wsum[k];
    }
    total = tot;
    return base + inc - mine;
  };
  unsigned loc[PER], sum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { loc[k] = hg[t * PER + k]; sum += loc[k]; }
  unsigned m;
  unsigned run = block_excl_scan(sum, m);
#pragma unroll
  for (int k = 0; k < PER; ++k) { run += loc[k]; cum[t * PER + k] = run; need[t * PER + k] = 0; }
  __syncthreads();
  if (t == 0) m_all[g] = m;
  auto locate = [&](unsigned r, int& bin, int& rk) {  // first bin with cum[bin] > r
    int lo = 0, hi = SEL_NB - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] > r) hi = mid; else lo = mid + 1;
    }
    bin = lo;
    rk = (int)(r - (lo > 0 ? cum[lo - 1] : 0u));
  };
  if (t < n && m > 0) {  // knots_body's quantile arithmetic
    const double p = (double)(t + 1) / (double)(n + 1);
    const double h = (double)(m - 1) * p;
    long long j = (long long)floor(h);
    const long long jmax = m >= 2 ? (long long)m - 2 : 0;
    if (j > jmax) j = jmax;
    if (j < 0) j = 0;
    const long long j1 = j + 1 < (long long)m ? j + 1 : (long long)m - 1;
    int4 q;
    locate((unsigned)j, q.x, q.y);
    locate((unsigned)j1, q.z, q.w);
    need[q.x] = 1; need[q.z] = 1;  // (benign races: every writer stores 1)
    tgt[(size_t)g * KMAX + t] = q;
  }
  __syncthreads();
  unsigned cnt = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) cnt += need[t * PER + k] ? loc[k] : 0u;
  unsigned tot;
  unsigned off = block_excl_scan(cnt, tot);
  int* go = goff + (size_t)g * SEL_NB;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const bool nd = need[t * PER + k] != 0;
    go[t * PER + k] = nd ? (int)off : -1;
    if (nd) off += loc[k];
  }
}
__global__ __launch_bounds__(256) void k_sel_gather(const long long* __restrict__ aoff, const unsigned* __restrict__ act,
                                                    const double* __restrict__ H, const unsigned long long* __restrict__ amax,
                                                    const int* __restrict__ goff, unsigned* __restrict__ cursor, double* __restrict__ buf) {
  const int g = blockIdx.y;
  const double am = __longlong_as_double((long long)amax[g]);
  if (!(am > 0.0)) return;
  const double inv = (double)SEL_NB / am;
  const long long lo = aoff[g], hi = aoff[g + 1];
  const int* go = goff + (size_t)g * SEL_NB;
  unsigned* cu = cursor + (size_t)g * SEL_NB;
  for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
    const double h = H[act[i]];
    if (h > 0.0 && h < am) {
      const int b = sel_bin(h, inv);
      const int o = go[b];
      if (o >= 0) buf[lo + o + atomicAdd(cu + b, 1u)] = h;
    }
  }
}
// the rk-th smallest (0-based) of e[0 .. c): radix selection on the bit patterns (positive doubles), then a rank count in LDS
__device__ double sel_kth(const double* __restrict__ e, unsigned c, unsigned rk, unsigned long long* cand, unsigned* cnt, unsigned* sctl) {
  unsigned long long prefix = 0ull, mask = 0ull;
  unsigned cur = c, r = rk;
  int shift = 56;
  while (cur > (unsigned)SEL_LDS && shift >= 0) {
    for (int k = threadIdx.x; k < 256; k += blockDim.x) cnt[k] = 0u;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < c; i += blockDim.x) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(e[i]);
      if ((b & mask) == prefix) atomicAdd(&cnt[(unsigned)(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned acc = 0, d = 0;
      for (; d < 256; ++d) {
        if (acc + cnt[d] > r) break;
        acc += cnt[d];
      }
      sctl[0] = d; sctl[1] = r - acc; sctl[2] = cnt[d];
    }
    __syncthreads();
    prefix |= (unsigned long long)sctl[0] << shift;
    mask |= 255ull << shift;
    r = sctl[1];
    cur = sctl[2];
    shift -= 8;
    __syncthreads();
  }
  if (cur > (unsigned)SEL_LDS) return __longlong_as_double((long long)prefix);  // all 64 bits fixed: every candidate is this value
  // compact the candidates into LDS (any order), then count ranks: the value whose rank interval contains r
  if (threadIdx.x == 0) sctl[3] = 0u;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < c; i += blockDim.x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(e[i]);
    if ((b & mask) == prefix) cand[atomicAdd(&sctl[3], 1u)] = b;
  }
  __syncthreads();
  const unsigned nc = sctl[3];
  __shared__ unsigned long long s_res;
  for (unsigned i = threadIdx.x; i < nc; i += blockDim.x) {
    const unsigned long long v = cand[i];
    unsigned less = 0, eq = 0;
#pragma unroll 8
    for (unsigned j = 0; j < nc; ++j) {
      const unsigned long long u = cand[j];
      less += u < v ? 1u : 0u;
      eq += u == v ? 1u : 0u;
    }
    if (less <= r && r < less + eq) s_res = v;  // (every thread that qualifies holds the same value)
  }
  __syncthreads();
  const double out = __longlong_as_double((long long)s_res);
  __syncthreads();
  return out;
}
__global__ __launch_bounds__(256) void k_sel_pick(const long long* __restrict__ aoff, const unsigned* __restrict__ hist,
                                                  const unsigned* __restrict__ m_all, const int* __restrict__ goff,
                                                  const int4* __restrict__ tgt, const double* __restrict__ buf, double2* __restrict__ os) {
  __shared__ unsigned long long cand[SEL_LDS];
  __shared__ unsigned cnt[256];
  __shared__ unsigned sctl[4];
  const int g = blockIdx.y, t = blockIdx.x;
  if (m_all[g] == 0u) return;
  const int4 q = tgt[(size_t)g * KMAX + t];
  const unsigned* hg = hist + (size_t)g * SEL_NB;
  const int* go = goff + (size_t)g * SEL_NB;
  const double* base = buf + aoff[g];
  const double a = sel_kth(base + go[q.x], hg[q.x], (unsigned)q.y, cand, cnt, sctl);
  const double b = (q.z == q.x && q.w == q.y) ? a : sel_kth(base + go[q.z], hg[q.z], (unsigned)q.w, cand, cnt, sctl);
  if (threadIdx.x == 0) os[(size_t)g * KMAX + t] = make_double2(a, b);
}
// knots_body with the order statistics handed in (same arithmetic, same de-duplication and ranking)
__global__ __launch_bounds__(KMAX) void k_sel_knots(const unsigned long long* __restrict__ amax, const unsigned* __restrict__ m_all,
                                                    const double2* __restrict__ os, int n, double* __restrict__ knots_all,
                                                    int* __restrict__ M_all) {
  __shared__ double c[KMAX];
  __shared__ int first[KMAX];
  const int g = blockIdx.x, i = threadIdx.x;
  double* knots = knots_all + (size_t)g * KMAX;
  const double am = __longlong_as_double((long long)amax[g]);
  if (!(am > 0.0)) {
    if (i == 0) M_all[g] = 0;
    return;
  }
  const long long m = (long long)m_all[g];
  bool valid = false;
  double v = 0.0;
  if (i < n) {
    const double t = (double)i / (double)(n - 1);
    v = (1.0 - t) * 0.0 + t * am;
    valid = true;
  } else if (i < 2 * n && m > 0) {
    const double p = (double)(i - n + 1) / (double)(n + 1);
    const double h = (double)(m - 1) * p;
    long long j = (long long)floor(h);
    const long long jmax = m >= 2 ? m - 2 : 0;
    if (j > jmax) j = jmax;
    if (j < 0) j = 0;
    double gam = h - (double)j;
    gam = gam < 0.0 ? 0.0 : (gam > 1.0 ? 1.0 : gam);
    const double2 ab = os[(size_t)g * KMAX + (i - n)];
    v = ab.x + gam * (ab.y - ab.x);
    valid = true;
  }
  c[i] = valid ? v : -1.0;
  __syncthreads();
  int isfirst = valid ? 1 : 0;
  if (valid)
    for (int j = 0; j < i; ++j)
      if (c[j] == v) { isfirst = 0; break; }
  first[i] = isfirst;
  __syncthreads();
  if (isfirst) {
    int rank = 0;
    for (int j = 0; j < 2 * n; ++j) rank += (first[j] && c[j] < v) ? 1 : 0;
    knots[rank] = v;
  }
  if (i == 0) {
    int M = 0;
    for (int j = 0; j < 2 * n; ++j) M += first[j];
    M_all[g] = M;
  }
}
// fixed-point scales of a glacier: s1 = 2^(e - 39), s2 = 2^(e - 79) with 2^e > vmax (0: the glacier contributes nothing)
__device__ __forceinline__ bool sel_scales(double vmax, double& s1, double& s2) {
  if (!(vmax > 0x1p-800)) return false;  // (weights below 2^-800 are dropped)
  int e;
  (void)frexp(vmax, &e);  // vmax = f 2^e, f in [0.5, 1)
  s1 = ldexp(1.0, e - 39);
  s2 = ldexp(1.0, e - 79);
  return true;
}
__global__ __launch_bounds__(256) void k_sel_sums(const long long* __restrict__ aoff, const unsigned* __restrict__ act,
                                                  const double* __restrict__ H, const double* __restrict__ V,
                                                  const unsigned long long* __restrict__ vmax, const double* __restrict__ knots_all,
                                                  const int* __restrict__ M_all, unsigned long long* __restrict__ bins) {
  __shared__ double kn[KMAX];
  __shared__ unsigned long long acc[KMAX][4];
  const int g = blockIdx.y, M = M_all[g];
  double s1, s2;
  if (M < 2 || !sel_scales(__longlong_as_double((long long)vmax[g]), s1, s2)) return;
  for (int k = threadIdx.x; k < M; k += 256) {
    kn[k] = knots_all[(size_t)g * KMAX + k];
    acc[k][0] = 0ull; acc[k][1] = 0ull; acc[k][2] = 0ull; acc[k][3] = 0ull;
  }
  __syncthreads();
  const double i1 = 1.0 / s1, i2 = 1.0 / s2;  // (powers of two: exact)
  auto add = [&](int k, int slot, double x) {
    const double hi = rint(x * i1);
    const double lo = rint(fma(-hi, s1, x) * i2);
    atomicAdd(&acc[k][slot], (unsigned long long)(long long)hi);
    atomicAdd(&acc[k][slot + 1], (unsigned long long)(long long)lo);
  };
  const long long lo_ = aoff[g], hi_ = aoff[g + 1];
  for (long long i = lo_ + (long long)blockIdx.x * 256 + threadIdx.x; i < hi_; i += (long long)gridDim.x * 256) {
    const unsigned q = act[i];
    const double h = H[q], v = V[q];
    if (!(h > 0.0) || v == 0.0) continue;
    int a = 0, b = M;  // first knot > h
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (kn[mid] <= h) a = mid + 1; else b = mid;
    }
    int k = a - 1;
    k = k < 0 ? 0 : (k > M - 2 ? M - 2 : k);
    const double x0 = kn[k], w = (h - x0) / (kn[k + 1] - x0);  // (a true division: see interval_sums_body)
    add(k, 0, v * (1.0 - w));
    add(k, 2, v * w);
  }
  __syncthreads();
  unsigned long long* out = bins + (size_t)g * KMAX * 4;
  for (int k = threadIdx.x; k < 4 * M; k += 256) {
    const unsigned long long x = acc[k >> 2][k & 3];
    if (x) atomicAdd(out + k, x);
  }
}
__global__ __launch_bounds__(KMAX) void k_sel_ab(const unsigned long long* __restrict__ vmax, const int* __restrict__ M_all,
                                                 const unsigned long long* __restrict__ bins, double* __restrict__ ab_all) {
  const int g = blockIdx.x, k = threadIdx.x;
  double* ab = ab_all + (size_t)g * 2 * KMAX;
  double s1, s2, a = 0.0, b = 0.0;
  if (k < M_all[g] - 1 && sel_scales(__longlong_as_double((long long)vmax[g]), s1, s2)) {
    const unsigned long long* x = bins + ((size_t)g * KMAX + k) * 4;
    a = fma((double)(long long)x[0], s1, (double)(long long)x[1] * s2);
    b = fma((double)(long long)x[2], s1, (double)(long long)x[3] * s2);
  }
  ab[k] = a; ab[KMAX + k] = b;
}
// scratch of one evaluation (per lane): zeroed block [amax G | vmax G | hist G NB | cursor G NB | bins G KMAX 4] (8-byte words first),
// then goff (G NB ints), tgt (G KMAX int4), os (G KMAX double2), m (G)
size_t interp_select_scratch_bytes(int G, size_t* zero_bytes) {
  const size_t z = (size_t)G * 8 * 2 + (size_t)G * KMAX * 4 * 8 + (size_t)G * SEL_NB * 4 * 2;
  if (zero_bytes) *zero_bytes = z;
  return z + (size_t)G * SEL_NB * 4 + (size_t)G * KMAX * 16 * 2 + (size_t)G * 4 + 64;
}
// all G glaciers; buf: n_act doubles (the gather areas); same results as launch_interp_theta_active up to the rounding of the sums
int launch_interp_theta_select(hipStream_t st, Pools P, const LawDev& L, int n_half, int G, long long n_act, const double* nodeH,
                               const double* nodeV, const unsigned* act, const long long* aoff, double* buf, void* scratch,
                               double* knots, int* M, double* ab, double* dth, int accumulate, const unsigned long long* amax_in,
                               const unsigned long long* vmax_in) {
  // amax_in / vmax_in (both or neither): the per-glacier maxima came with the node arrays (the fused reverse step's emission)
  if (2 * n_half > KMAX || n_half < 2 || n_act < 1 || n_act >= (1ll << 31)) return 1;
  size_t zb = 0;
  (void)interp_select_scratch_bytes(G, &zb);
  char* p = static_cast<char*>(scratch);
  unsigned long long* amax_own = reinterpret_cast<unsigned long long*>(p);
  unsigned long long* vmax_own = amax_own + G;
  unsigned long long* bins = vmax_own + G;
  const unsigned long long* amax = amax_in ? amax_in : amax_own;
  const unsigned long long* vmax = vmax_in ? vmax_in : vmax_own;
  unsigned* hist = reinterpret_cast<unsigned*>(bins + (size_t)G * KMAX * 4);
  unsigned* cursor = hist + (size_t)G * SEL_NB;
  int* goff = reinterpret_cast<int*>(p + zb);
  int4* tgt = reinterpret_cast<int4*>(goff + (size_t)G * SEL_NB);
  double2* os = reinterpret_cast<double2*>(tgt + (size_t)G * KMAX);
  unsigned* m = reinterpret_cast<unsigned*>(os + (size_t)G * KMAX);
  if (hipMemsetAsync(p, 0, zb, st) != hipSuccess) return 2;
  const long long per = (n_act + G - 1) / G;
  const unsigned blk = (unsigned)std::max<long long>(1, std::min<long long>(SEL_BLK_MAX, (per + 4095) / 4096));
  const dim3 grid(blk, G);
  if (!amax_in) hipLaunchKernelGGL(k_sel_max, grid, dim3(256), 0, st, aoff, act, nodeH, nodeV, amax_own, vmax_own);
  hipLaunchKernelGGL(k_sel_hist, grid, dim3(256), 0, st, aoff, act, nodeH, amax, hist);
  hipLaunchKernelGGL(k_sel_targets, dim3(G), dim3(1024), 0, st, hist, n_half, m, goff, tgt);
  hipLaunchKernelGGL(k_sel_gather, grid, dim3(256), 0, st, aoff, act, nodeH, amax, goff, cursor, buf);
  hipLaunchKernelGGL(k_sel_pick, dim3(n_half, G), dim3(256), 0, st, aoff, hist, m, goff, tgt, buf, os);
  hipLaunchKernelGGL(k_sel_knots, dim3(G), dim3(KMAX), 0, st, amax, m, os, n_half, knots, M);
  hipLaunchKernelGGL(k_sel_sums, grid, dim3(256), 0, st, aoff, act, nodeH, nodeV, vmax, knots, M, bins);
  hipLaunchKernelGGL(k_sel_ab, dim3(G), dim3(KMAX), 0, st, vmax, M, bins, ab);
  const size_t dyn = interp_batch_lds_bytes(L.P);
  if (interp_law_is<ArchDef>(L))
    hipLaunchKernelGGL((k_knot_backprop<ArchDef, true>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  else if (interp_law_is<Arch16>(L))
    hipLaunchKernelGGL((k_knot_backprop<Arch16, true>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  else
    hipLaunchKernelGGL((k_knot_backprop<ArchRT, false>), dim3(G), dim3(NT), dyn, st, P, L, 0, knots, M, ab, dth, accumulate);
  return 0;
}

// ---- exact per-node backprop of emitted node weights (`interpolation = :None` in the surface-velocity pull-backs) -----------------
// dth[g] (+)= sum over the dual nodes of glacier g of V[node] * d law / d theta at the node's inputs -- (T_g, Hbar) for the Y law,
// (Hbar, |grad S|) for the U law -- with the node weights V and inputs emitted by the velocity kernels (k_surfV_vjp: emitH / emitV /
// emitS).  Wave-reduced like k_law_field_grad_wave: P accumulators per wavefront in LDS, chunks of 64 nodes, chunks without weight
// skipped, fixed summation order.  The velocity kernels used to keep P accumulators per THREAD in global memory (2 x 8 B x P of
// traffic per node: 1.48 ms per call at 8 x 512^2 for the 83-parameter default net, ten times a reverse stage).
constexpr int NBP_BLK = 64;  // workgroups per glacier
// NWV: wavefronts per workgroup (NW; 1 for networks whose NW x P accumulators do not fit the LDS -- see launch_node_backprop)
template <class AR, bool FIXED, int NWV = NW>
__global__ __launch_bounds__(64 * NWV) void k_node_backprop(Pools P, LawDev L, int g0, int g_last, long long end_all,
                                                      const double* __restrict__ nodeH, const double* __restrict__ nodeS,
                                                      const double* __restrict__ nodeV, double* __restrict__ part) {
  extern __shared__ double nb_dyn[];
  __shared__ double stage[NWV][WG_SLOTS][WG_LD];
  double* accs = nb_dyn;
  int* order = reinterpret_cast<int*>(nb_dyn + (size_t)NWV * L.P);
  const int lane = threadIdx.x & 63, w = NWV == 1 ? 0 : wave_id();
  for (int k = threadIdx.x; k < NWV * L.P; k += 64 * NWV) accs[k] = 0.0;
  if (threadIdx.x == 0) mlp_grad_order(L, order);
  __syncthreads();
  const WaveAcc A{stage[w], order, accs + (size_t)w * L.P};
  const int gidx = g0 + blockIdx.y;
  const long long off = P.gd[gidx].offd, n = seg_len(P, gidx, g_last, end_all);
  const double T = P.gd[gidx].T;
  const long long nchunk = (n + 63) / 64;
  for (long long c = (long long)blockIdx.x * NWV + w; c < nchunk; c += (long long)gridDim.x * NWV) {
    const long long i = c * 64 + lane;
    const bool ok = i < n;
    const double wgt = ok ? nodeV[off + i] : 0.0;
    if (__builtin_amdgcn_ballot_w64(wgt != 0.0) == 0) continue;
    const double hb = ok ? nodeH[off + i] : 0.0;
    if (L.kind == 3) mlp_grad_wave<AR, FIXED>(L, T, hb, wgt, A, lane);
    else mlp_grad_wave<AR, FIXED>(L, hb, ok ? nodeS[off + i] : 0.0, wgt, A, lane);
  }
  __syncthreads();
  double* out = part + ((size_t)blockIdx.y * NBP_BLK + blockIdx.x) * L.P;
  for (int k = threadIdx.x; k < L.P; k += 64 * NWV) {
    double s = 0.0;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) s += accs[(size_t)ww * L.P + k];
    out[k] = s;
  }
}
__global__ __launch_bounds__(64) void k_node_backprop_sum(int Pn, int g0, const double* __restrict__ part, double* __restrict__ dth,
                                                          int accumulate) {
  const int k = blockIdx.x, g = blockIdx.y;
  double s = 0.0;
  for (int bl = 0; bl < NBP_BLK; ++bl) s += part[((size_t)g * NBP_BLK + bl) * Pn + k];  // fixed order
  if (threadIdx.x == 0) {
    double* o = dth + (size_t)(g0 + g) * Pn + k;
    *o = accumulate ? *o + s : s;
  }
}
size_t node_backprop_part_count(int ng, int Pn) { return (size_t)ng * NBP_BLK * Pn; }
// glaciers [g0, g0 + ng); nodeH / nodeS / nodeV point at the POOL's first dual node; part: node_backprop_part_count doubles
int launch_node_backprop(hipStream_t st, Pools P, const LawDev& L, int g0, int ng, long long end_all, const double* nodeH,
                         const double* nodeS, const double* nodeV, double* part, double* dth, int accumulate) {
  const size_t dyn = interp_batch_lds_bytes(L.P);
  const dim3 grid(NBP_BLK, ng);
  if (dyn > 30 * 1024) {
    // wide networks (the 2-5-8-20-30-10-1 net of the reference's diffusivity MWE: P = 1194): NW x P accumulators do not fit next to
    // the staging area -- one wavefront per workgroup, P accumulators (12 B per parameter: up to MAXP = 2048 parameters)
    const size_t dyn1 = (size_t)L.P * sizeof(double) + (size_t)L.P * sizeof(int);
    if (dyn1 > 30 * 1024) return 1;
    hipLaunchKernelGGL((k_node_backprop<ArchRT, false, 1>), grid, dim3(64), dyn1, st, P, L, g0, g0 + ng - 1, end_all, nodeH, nodeS, nodeV, part);
    hipLaunchKernelGGL(k_node_backprop_sum, dim3(L.P, ng), dim3(64), 0, st, L.P, g0, part, dth, accumulate);
    return 0;
  }
  if (interp_law_is<ArchDef>(L))
    hipLaunchKernelGGL((k_node_backprop<ArchDef, true>), grid, dim3(NT), dyn, st, P, L, g0, g0 + ng - 1, end_all, nodeH, nodeS, nodeV, part);
  else if (interp_law_is<Arch16>(L))
    hipLaunchKernelGGL((k_node_backprop<Arch16, true>), grid, dim3(NT), dyn, st, P, L, g0, g0 + ng - 1, end_all, nodeH, nodeS, nodeV, part);
  else
    hipLaunchKernelGGL((k_node_backprop<ArchRT, false>), grid, dim3(NT), dyn, st, P, L, g0, g0 + ng - 1, end_all, nodeH, nodeS, nodeV, part);
  hipLaunchKernelGGL(k_node_backprop_sum, dim3(L.P, ng), dim3(64), 0, st, L.P, g0, part, dth, accumulate);
  return 0;
}

// scratch: sH, sV (nd doubles each), tmp (interp_sort_temp_bytes), knots (KMAX), M (int), G (P * KMAX), ab (2 * KMAX)
int launch_interp_theta(hipStream_t st, const LawDev& L, double T, int n_half, const double* nodeH, const double* nodeV,
                        long long nd, double* sH, double* sV, void* tmp, size_t tmp_bytes, double* knots, int* M, double* G,
                        double* ab, double* dth, int accumulate) {
  if (2 * n_half > KMAX || n_half < 2) return 1;
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, nodeH, sH, nodeV, sV, (size_t)nd, 0, 64, st) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_knots, dim3(1), dim3(KMAX), 0, st, sH, nd, n_half, knots, M);
  hipLaunchKernelGGL(k_knot_grads, dim3((2 * n_half + 63) / 64), dim3(64), 0, st, L, T, knots, M, G);
  hipLaunchKernelGGL(k_interval_sums, dim3(2 * n_half), dim3(NT), 0, st, sH, sV, nd, knots, M, ab);
  hipLaunchKernelGGL(k_interp_contract, dim3((L.P + 63) / 64), dim3(64), 0, st, L.P, M, ab, G, dth, accumulate);
  return 0;
}

}  // namespace odinn
