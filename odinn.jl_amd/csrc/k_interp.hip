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
#include <cstdlib>
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
__global__ __launch_bounds__(KMAX) void k_knots(const double* __restrict__ sH, long long nd, int n, double* __restrict__ knots,
                                                int* __restrict__ Mout) {
  __shared__ double c[KMAX];
  __shared__ int first[KMAX];
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

// G[q * KMAX + k] = d Y / d theta_q at (T, knots[k])
__global__ __launch_bounds__(64) void k_knot_grads(LawDev L, double T, const double* __restrict__ knots, const int* __restrict__ Mp,
                                                   double* __restrict__ G) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= *Mp) return;
  for (int q = 0; q < L.P; ++q) G[(long long)q * KMAX + k] = 0.0;
  mlp_grad(L, T, knots[k], 1.0, G + k, KMAX);
}

// block k: the nodes with knots[k] <= Hbar < knots[k+1] (the last interval takes Hbar == max too)
__global__ __launch_bounds__(NT) void k_interval_sums(const double* __restrict__ sH, const double* __restrict__ sV, long long nd,
                                                      const double* __restrict__ knots, const int* __restrict__ Mp,
                                                      double* __restrict__ ab) {
  __shared__ double red[NW];
  const int k = blockIdx.x, M = *Mp;
  if (k >= M - 1) {
    if (threadIdx.x == 0) { ab[k] = 0.0; ab[KMAX + k] = 0.0; }
    return;
  }
  const double x0 = knots[k], x1 = knots[k + 1], inv = 1.0 / (x1 - x0);
  const long long lo = lower_bound_d(sH, nd, x0), hi = (k == M - 2) ? nd : lower_bound_d(sH, nd, x1);
  double a = 0.0, b = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += NT) {
    const double w = (sH[i] - x0) * inv, v = sV[i];
    a = fma(v, 1.0 - w, a);
    b = fma(v, w, b);
  }
  a = block_sum(a, red);
  __syncthreads();
  b = block_sum(b, red);
  if (threadIdx.x == 0) { ab[k] = a; ab[KMAX + k] = b; }
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
