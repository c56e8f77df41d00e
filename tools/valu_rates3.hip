// Microbenchmark 3: what a (gate, digit position) of the index form of the key switch costs (kernels.hip
// keyswitch_index_kernel) -- four v_sub_u32 whose second source is taken relative to M0 (VGPR index mode), preceded by the
// scalar instructions that pick the row -- against the same four subtractions without index mode.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates3.hip -o gpurun_out/valu_rates3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// eight accumulators (v8..v39) and four rows (v40..v55: zeros, row 1, 2, 3), pinned like the kernel's
#define ACCS "+{v[8:11]}"(a0), "+{v[12:15]}"(a1), "+{v[16:19]}"(a2), "+{v[20:23]}"(a3), "+{v[24:27]}"(a4), "+{v[28:31]}"(a5), "+{v[32:35]}"(a6), "+{v[36:39]}"(a7)
#define ROWS "{v[40:43]}"(r0), "{v[44:47]}"(r1), "{v[48:51]}"(r2), "{v[52:55]}"(r3)
#define SUB4(A, B) "v_sub_u32 v" #A ", v" #A ", v" #B "\n\t"
#define GROUP(K, B0) SUB4(K, B0) "v_sub_u32 v%=, v%=, v0\n\t"
#define G4(A0, A1, A2, A3) "v_sub_u32 v" #A0 ", v" #A0 ", v40\n\tv_sub_u32 v" #A1 ", v" #A1 ", v41\n\tv_sub_u32 v" #A2 ", v" #A2 ", v42\n\tv_sub_u32 v" #A3 ", v" #A3 ", v43\n\t"
#define ALL8(PREFIX) \
    PREFIX(0) G4(8, 9, 10, 11) PREFIX(1) G4(12, 13, 14, 15) PREFIX(2) G4(16, 17, 18, 19) PREFIX(3) G4(20, 21, 22, 23) \
    PREFIX(4) G4(24, 25, 26, 27) PREFIX(5) G4(28, 29, 30, 31) PREFIX(6) G4(32, 33, 34, 35) PREFIX(7) G4(36, 37, 38, 39)

#define P_NONE(k) ""
#define P_IDX(k) "s_bfe_u32 %[t], %[y], " #k " * 4 + 0x40000\n\ts_set_gpr_idx_idx %[t]\n\t"
#define P_IDX_ONLY(k) "s_set_gpr_idx_idx %[y2]\n\t"
#define P_SKIP(k) "s_bfe_u32 %[t], %[y], " #k " * 4 + 0x40000\n\ts_cbranch_scc0 .Lskip%=_" #k "\n\ts_set_gpr_idx_idx %[t]\n\t"
#define L_SKIP(k) ".Lskip%=_" #k ":\n\t"

#define HEAD                                                                      \
  u4 a0, a1, a2, a3, a4, a5, a6, a7, r0 = (u4)(0u), r1, r2, r3;                   \
  a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = (u4)(threadIdx.x + seed);               \
  r1 = (u4)(seed | 1u); r2 = (u4)(seed * 3u + 7u); r3 = (u4)(seed * 5u + 11u);    \
  uint32_t t = 0, m = 0;                                                          \
  const uint32_t y = __builtin_amdgcn_readfirstlane(ymask), y2 = __builtin_amdgcn_readfirstlane(ymask & 12u);
#define TAIL                                                                      \
  u4 s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                   \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x ^ s.y ^ s.z ^ s.w ^ t ^ m;

// 1. plain: 8 x 4 subtractions, no index mode
__global__ void k_plain(uint32_t *out, uint32_t seed, uint32_t ymask) {
  HEAD
  for (int it = 0; it < ITERS; ++it)
    asm volatile(ALL8(P_NONE) : ACCS, [t] "+s"(t), [m] "+s"(m) : [y] "s"(y), [y2] "s"(y2), ROWS : "scc");
  TAIL
}
// 2. index mode on (set once per statement), the same 32 subtractions, index constant
__global__ void k_mode_on(uint32_t *out, uint32_t seed, uint32_t ymask) {
  HEAD
  for (int it = 0; it < ITERS; ++it)
    asm volatile("s_mov_b32 %[m], m0\n\ts_set_gpr_idx_on %[y2], 0x2\n\t" ALL8(P_NONE) "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[m]"
                 : ACCS, [t] "+s"(t), [m] "+s"(m) : [y] "s"(y), [y2] "s"(y2), ROWS : "scc");
  TAIL
}
// 3. a new index in front of every group of four (no s_bfe: the index is ready in a register)
__global__ void k_idx_only(uint32_t *out, uint32_t seed, uint32_t ymask) {
  HEAD
  for (int it = 0; it < ITERS; ++it)
    asm volatile("s_mov_b32 %[m], m0\n\ts_set_gpr_idx_on %[y2], 0x2\n\t" ALL8(P_IDX_ONLY) "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[m]"
                 : ACCS, [t] "+s"(t), [m] "+s"(m) : [y] "s"(y), [y2] "s"(y2), ROWS : "scc");
  TAIL
}
// 4. s_bfe_u32 + s_set_gpr_idx_idx in front of every group (the kernel's form before the zero-digit branch)
__global__ void k_bfe_idx(uint32_t *out, uint32_t seed, uint32_t ymask) {
  HEAD
  for (int it = 0; it < ITERS; ++it)
    asm volatile("s_mov_b32 %[m], m0\n\ts_set_gpr_idx_on %[y2], 0x2\n\t" ALL8(P_IDX) "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[m]"
                 : ACCS, [t] "+s"(t), [m] "+s"(m) : [y] "s"(y), [y2] "s"(y2), ROWS : "scc");
  TAIL
}
// 5. the kernel's form: s_bfe_u32, branch over the group for digit 0, s_set_gpr_idx_idx (ymask decides how many are skipped)
__global__ void k_skip(uint32_t *out, uint32_t seed, uint32_t ymask) {
  HEAD
  for (int it = 0; it < ITERS; ++it)
    asm volatile("s_mov_b32 %[m], m0\n\ts_set_gpr_idx_on %[y2], 0x2\n\t"
                 P_SKIP(0) G4(8, 9, 10, 11) L_SKIP(0) P_SKIP(1) G4(12, 13, 14, 15) L_SKIP(1) P_SKIP(2) G4(16, 17, 18, 19) L_SKIP(2)
                 P_SKIP(3) G4(20, 21, 22, 23) L_SKIP(3) P_SKIP(4) G4(24, 25, 26, 27) L_SKIP(4) P_SKIP(5) G4(28, 29, 30, 31) L_SKIP(5)
                 P_SKIP(6) G4(32, 33, 34, 35) L_SKIP(6) P_SKIP(7) G4(36, 37, 38, 39) L_SKIP(7)
                 "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[m]"
                 : ACCS, [t] "+s"(t), [m] "+s"(m) : [y] "s"(y), [y2] "s"(y2), ROWS : "scc");
  TAIL
}

typedef void (*kern_t)(uint32_t *, uint32_t, uint32_t);
struct Entry { const char *name; kern_t k; uint32_t ymask; const char *what; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d\n", prop.name, cus);
  uint32_t *out;
  CK(hipMalloc(&out, sizeof(uint32_t) * cus * 16 * 256));
  // index nibbles: 0x4C84C84C = digits 3,1,2,3,... never 0; 0x4C80C840: two of eight digits 0 (a quarter skipped)
  std::vector<Entry> es = {{"plain", k_plain, 0, "8 x 4 v_sub_u32, no index mode"},
                           {"mode_on", k_mode_on, 4, "index mode on, one index per 32 subtractions"},
                           {"idx_only", k_idx_only, 4, "s_set_gpr_idx_idx in front of every 4"},
                           {"bfe_idx", k_bfe_idx, 0x4C84C84Cu, "s_bfe_u32 + s_set_gpr_idx_idx in front of every 4"},
                           {"skip_none", k_skip, 0x4C84C84Cu, "+ branch for digit 0, never taken"},
                           {"skip_quarter", k_skip, 0x4C80C840u, "+ branch for digit 0, taken for 2 of 8"}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-13s SIMD cycles per group of four subtractions at 2.4 GHz nominal, by waves per SIMD\n", "form");
  for (auto &e : es) {
    printf("%-13s", e.name);
    for (int bpc : {1, 2, 4, 8}) {
      int grid = cus * bpc;
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u, e.ymask);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u, e.ymask);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double cycles = ms * 1e-3 * 2.4e9;
      printf("   w=%d %6.2f", bpc, cycles / ((double)ITERS * 8 * bpc));
    }
    printf("   %s\n", e.what);
  }
  CK(hipFree(out));
  return 0;
}
