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

size_t interp_sort_temp_bytes(long long nd_max) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const double*)nullptr, (double*)nullptr, (const double*)nullptr,
                                  (double*)nullptr, (size_t)nd_max, 0, 64, nullptr);
  return bytes;
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
