// Issue-cost microbenchmark for the instruction classes of the strip kernels (gfx950).
// Every kernel runs REP x 64 instructions of one class per wavefront, 4 wavefronts per SIMD (2 x 512-thread
// workgroups per CU), so the figure is the throughput cost per wave-instruction under the kernel's own occupancy.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip ; run: ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 256
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define KERNEL(name, body)                                                                                   \
  __global__ __launch_bounds__(512, 4) void name(double* out, int n) {                                       \
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double b = 1.0000001, c = 1e-9;                                                                          \
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7; \
    int addr = ((threadIdx.x + 1) & 63) << 2;                                                                \
    for (int r = 0; r < n; ++r) {                                                                            \
      asm volatile(".rept 8\n" body "\n.endr"                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),         \
                     "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)          \
                   : "v"(b), "v"(c), "v"(addr));                                                             \
    }                                                                                                        \
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7; \
  }
// operands: %0-%7 doubles, %8-%15 ints, %16 b, %17 c, %18 addr.  Each body = 8 instructions on independent registers.
KERNEL(k_fma64, "v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17")
KERNEL(k_add64, "v_add_f64 %0, %0, %17\n v_add_f64 %1, %1, %17\n v_add_f64 %2, %2, %17\n v_add_f64 %3, %3, %17\n v_add_f64 %4, %4, %17\n v_add_f64 %5, %5, %17\n v_add_f64 %6, %6, %17\n v_add_f64 %7, %7, %17")
KERNEL(k_mul64, "v_mul_f64 %0, %0, %16\n v_mul_f64 %1, %1, %16\n v_mul_f64 %2, %2, %16\n v_mul_f64 %3, %3, %16\n v_mul_f64 %4, %4, %16\n v_mul_f64 %5, %5, %16\n v_mul_f64 %6, %6, %16\n v_mul_f64 %7, %7, %16")
KERNEL(k_max64, "v_max_f64 %0, %0, %17\n v_min_f64 %1, %1, %16\n v_max_f64 %2, %2, %17\n v_min_f64 %3, %3, %16\n v_max_f64 %4, %4, %17\n v_min_f64 %5, %5, %16\n v_max_f64 %6, %6, %17\n v_min_f64 %7, %7, %16")
KERNEL(k_dpp_wave, "v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %9, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %10, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %11, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %12, %13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %13, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %14, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %15, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
KERNEL(k_dpp_row, "v_mov_b32_dpp %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %9, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %10, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %11, %12 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %13, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %14, %15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %15, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
KERNEL(k_mov32, "v_mov_b32 %8, %9\n v_mov_b32 %9, %10\n v_mov_b32 %10, %11\n v_mov_b32 %11, %12\n v_mov_b32 %12, %13\n v_mov_b32 %13, %14\n v_mov_b32 %14, %15\n v_mov_b32 %15, %8")
KERNEL(k_add32, "v_add_u32 %8, %9, %8\n v_add_u32 %9, %10, %9\n v_add_u32 %10, %11, %10\n v_add_u32 %11, %12, %11\n v_add_u32 %12, %13, %12\n v_add_u32 %13, %14, %13\n v_add_u32 %14, %15, %14\n v_add_u32 %15, %8, %15")
KERNEL(k_bperm, "ds_bpermute_b32 %8, %18, %8\n ds_bpermute_b32 %9, %18, %9\n ds_bpermute_b32 %10, %18, %10\n ds_bpermute_b32 %11, %18, %11\n ds_bpermute_b32 %12, %18, %12\n ds_bpermute_b32 %13, %18, %13\n ds_bpermute_b32 %14, %18, %14\n ds_bpermute_b32 %15, %18, %15\n s_waitcnt lgkmcnt(0)")
KERNEL(k_swizzle, "ds_swizzle_b32 %8, %8 offset:0x8001\n ds_swizzle_b32 %9, %9 offset:0x8001\n ds_swizzle_b32 %10, %10 offset:0x8001\n ds_swizzle_b32 %11, %11 offset:0x8001\n ds_swizzle_b32 %12, %12 offset:0x8001\n ds_swizzle_b32 %13, %13 offset:0x8001\n ds_swizzle_b32 %14, %14 offset:0x8001\n ds_swizzle_b32 %15, %15 offset:0x8001\n s_waitcnt lgkmcnt(0)")
// mixes: 4 fp64 + 4 of the other class, to see whether the two overlap (co-issue from different waves) or add
KERNEL(k_mix_fma_dpp, "v_fma_f64 %0, %0, %16, %17\n v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f64 %1, %1, %16, %17\n v_mov_b32_dpp %9, %10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f64 %2, %2, %16, %17\n v_mov_b32_dpp %10, %11 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f64 %3, %3, %16, %17\n v_mov_b32_dpp %11, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
KERNEL(k_mix_fma_bperm, "v_fma_f64 %0, %0, %16, %17\n ds_bpermute_b32 %8, %18, %8\n v_fma_f64 %1, %1, %16, %17\n ds_bpermute_b32 %9, %18, %9\n v_fma_f64 %2, %2, %16, %17\n ds_bpermute_b32 %10, %18, %10\n v_fma_f64 %3, %3, %16, %17\n ds_bpermute_b32 %11, %18, %11\n s_waitcnt lgkmcnt(0)")
KERNEL(k_mix_fma_mov, "v_fma_f64 %0, %0, %16, %17\n v_mov_b32 %8, %9\n v_fma_f64 %1, %1, %16, %17\n v_mov_b32 %9, %10\n v_fma_f64 %2, %2, %16, %17\n v_mov_b32 %10, %11\n v_fma_f64 %3, %3, %16, %17\n v_mov_b32 %11, %8")
// 6 fp64 + 2 bpermute (the strip kernel's ratio if the exchanges moved to the LDS crossbar): 36 fp64 : 8 moves
KERNEL(k_mix_6fma_2bperm, "v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n ds_bpermute_b32 %8, %18, %8\n v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n ds_bpermute_b32 %9, %18, %9\n s_waitcnt lgkmcnt(0)")
KERNEL(k_mix_6fma_2dpp, "v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_mov_b32_dpp %9, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")

typedef void (*kfn)(double*, int);
int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  const int cus = p.multiProcessorCount, blocks = cus * 2;
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 512));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* name; kfn f; int per_body; } ks[] = {
    {"v_fma_f64", k_fma64, 8}, {"v_add_f64", k_add64, 8}, {"v_mul_f64", k_mul64, 8}, {"v_min/max_f64", k_max64, 8},
    {"v_mov_b32_dpp wave_sh", k_dpp_wave, 8}, {"v_mov_b32_dpp row_sh", k_dpp_row, 8}, {"v_mov_b32", k_mov32, 8}, {"v_add_u32", k_add32, 8},
    {"ds_bpermute_b32", k_bperm, 8}, {"ds_swizzle_b32", k_swizzle, 8},
    {"mix 4 fma64 + 4 dpp", k_mix_fma_dpp, 8}, {"mix 4 fma64 + 4 bperm", k_mix_fma_bperm, 8}, {"mix 4 fma64 + 4 mov", k_mix_fma_mov, 8},
    {"mix 6 fma64 + 2 bperm", k_mix_6fma_2bperm, 8}, {"mix 6 fma64 + 2 dpp", k_mix_6fma_2dpp, 8}};
  const int n = REP;
  printf("device %s, %d CUs, clock %d MHz; 2 x 512-thread blocks per CU = 4 waves/SIMD\n", p.name, cus, p.clockRate / 1000);
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(512), 0, 0, out, 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(512), 0, 0, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 waves x n x 8 bodies x per_body instructions
    const double insts = 4.0 * n * 8 * k.per_body;
    printf("%-26s %8.3f ms  -> %6.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", k.name, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
  }
  return 0;
}
