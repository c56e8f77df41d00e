// Microbenchmark 2: issue cost of the select / compare / 64-bit helper instructions that
// surround the NTT butterflies (the first table showed v_cndmask_b32 far slower than the
// arithmetic instructions; this one separates the variants).
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates2.hip -o gpurun_out/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define KERNEL32(NAME, PRE, ASMSTR, NINSTR)                                      \
__global__ void NAME(uint32_t* out, uint32_t seed) {                             \
  uint32_t x[CHAINS];                                                            \
  uint32_t m = seed | 1u, c = seed * 3u + 7u;                                    \
  unsigned long long mask = 0x5555555555555555ull + seed;                        \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 7u + i + seed; \
  asm volatile(PRE ::"s"(mask) : "vcc");                                         \
  for (int it = 0; it < ITERS; ++it) {                                           \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                           \
      asm volatile(ASMSTR : "+v"(x[i]) : "v"(m), "v"(c), "s"(mask) : "vcc");     \
  }                                                                              \
  uint32_t s = 0;                                                                \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s ^= x[i];                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
}                                                                                \
constexpr int NAME##_n = NINSTR;

KERNEL32(k_add,           "", "v_add_u32 %0, %0, %1", 1)
KERNEL32(k_cnd_vcc,       "", "v_cndmask_b32 %0, %0, %1, vcc", 1)
KERNEL32(k_cnd_vcc_init,  "s_mov_b64 vcc, %0", "v_cndmask_b32 %0, %0, %1, vcc", 1)
KERNEL32(k_cnd_sgpr,      "", "v_cndmask_b32_e64 %0, %0, %1, %3", 1)
KERNEL32(k_cmp_only,      "", "v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %2", 2)
KERNEL32(k_cmp_cnd,       "", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc", 2)
KERNEL32(k_cmp_e64_cnd,   "", "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]", 2)
KERNEL32(k_min_u32,       "", "v_min_u32 %0, %0, %1", 1)
KERNEL32(k_max_i32,       "", "v_max_i32 %0, %0, %1", 1)
KERNEL32(k_med3_i32,      "", "v_med3_i32 %0, %0, %1, %2", 1)
KERNEL32(k_bfi,           "", "v_bfi_b32 %0, %1, %0, %2", 1)
KERNEL32(k_and_or,        "", "v_and_or_b32 %0, %0, %1, %2", 1)
KERNEL32(k_ashr,          "", "v_ashrrev_i32 %0, 1, %0", 1)
KERNEL32(k_sub,           "", "v_sub_u32 %0, %0, %1", 1)
KERNEL32(k_add3,          "", "v_add3_u32 %0, %0, %1, %2", 1)
KERNEL32(k_addco,         "", "v_add_co_u32 %0, vcc, %0, %1", 1)
KERNEL32(k_addco_addc,    "", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc", 2)
KERNEL32(k_mul_lo,        "", "v_mul_lo_u32 %0, %0, %1", 1)
KERNEL32(k_mul_i32_i24,   "", "v_mul_i32_i24 %0, %0, %1", 1)
KERNEL32(k_mad_i32_i24,   "", "v_mad_i32_i24 %0, %0, %1, %2", 1)
KERNEL32(k_perm,          "", "v_perm_b32 %0, %0, %1, %2", 1)
KERNEL32(k_alignbit,      "", "v_alignbit_b32 %0, %0, %1, 7", 1)
KERNEL32(k_bfe_i32,       "", "v_bfe_i32 %0, %0, 3, 7", 1)
KERNEL32(k_readlane_free, "", "v_mov_b32 %0, %0", 1)

#define KERNEL64(NAME, ASMSTR, NINSTR)                                           \
__global__ void NAME(uint32_t* out, uint32_t seed) {                             \
  unsigned long long x[CHAINS];                                                  \
  uint32_t m = seed | 1u, c = seed * 3u + 7u;                                    \
  unsigned long long w = 0x123456789ull + seed;                                  \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 7u + i + seed; \
  for (int it = 0; it < ITERS; ++it) {                                           \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                           \
      asm volatile(ASMSTR : "+v"(x[i]) : "v"(m), "v"(c), "v"(w) : "vcc");        \
  }                                                                              \
  unsigned long long s = 0;                                                      \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s ^= x[i];                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;                      \
}                                                                                \
constexpr int NAME##_n = NINSTR;

KERNEL64(k_mad_i64_i32,   "v_mad_i64_i32 %0, vcc, %1, %2, %0", 1)
KERNEL64(k_mad_u64_u32,   "v_mad_u64_u32 %0, vcc, %1, %2, %0", 1)
KERNEL64(k_mad_i64_sgprco,"v_mad_i64_i32 %0, s[20:21], %1, %2, %0", 1)
KERNEL64(k_lshl_add_u64,  "v_lshl_add_u64 %0, %0, 0, %3", 1)
KERNEL64(k_cmp_u64_add,   "v_cmp_lt_u64 vcc, %0, %3\n v_lshl_add_u64 %0, %0, 0, %3", 2)
KERNEL64(k_ashr_i64,      "v_ashrrev_i64 %0, 1, %0", 1)
KERNEL64(k_mov_b64,       "v_mov_b64 %0, %3", 1)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int n; };
#define E(NAME) {#NAME, NAME, NAME##_n}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint32_t* out;
  CK(hipMalloc(&out, sizeof(uint32_t) * cus * 16 * 256));
  std::vector<Entry> es = {E(k_add), E(k_cnd_vcc), E(k_cnd_vcc_init), E(k_cnd_sgpr), E(k_cmp_only), E(k_cmp_cnd),
                           E(k_cmp_e64_cnd), E(k_min_u32), E(k_max_i32), E(k_med3_i32), E(k_bfi), E(k_and_or), E(k_ashr),
                           E(k_sub), E(k_add3), E(k_addco), E(k_addco_addc), E(k_mul_lo), E(k_mul_i32_i24), E(k_mad_i32_i24),
                           E(k_perm), E(k_alignbit), E(k_bfe_i32), E(k_readlane_free), E(k_mad_i64_i32), E(k_mad_u64_u32),
                           E(k_mad_i64_sgprco), E(k_lshl_add_u64), E(k_cmp_u64_add), E(k_ashr_i64), E(k_mov_b64)};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bpcs[] = {1, 2, 4};
  printf("%-18s (cycles per wave-instruction GROUP at 2.4 GHz nominal; group size in [])\n", "instr");
  for (auto& e : es) {
    printf("%-18s [%d]", e.name, e.n);
    for (int bpc : bpcs) {
      int grid = cus * bpc;
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double cycles = ms * 1e-3 * 2.4e9;
      printf("   w/SIMD=%d %7.2f", bpc, cycles / ((double)ITERS * CHAINS * bpc));
    }
    printf("\n");
  }
  CK(hipFree(out));
  return 0;
}
