// Microbenchmark: VALU issue rates on gfx950 for the instructions an exact
// modular NTT could be built from (32-bit integer multiplies vs FP64 FMA).
// The CDNA guides give no integer-multiply rates, so the NTT's modular
// arithmetic was chosen from this measurement.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;   // independent dependency chains per lane

// Each kernel runs ITERS * CHAINS instances of one instruction per lane.
#define KERNEL32(NAME, ASMSTR)                                                   \
__global__ void NAME(uint32_t* out, uint32_t seed) {                             \
  uint32_t x[CHAINS];                                                            \
  uint32_t m = seed | 1u, c = seed * 3u + 7u;                                    \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 7u + i + seed; \
  for (int it = 0; it < ITERS; ++it) {                                           \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                           \
      asm volatile(ASMSTR : "+v"(x[i]) : "v"(m), "v"(c));                        \
  }                                                                              \
  uint32_t s = 0;                                                                \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s ^= x[i];                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
}

KERNEL32(k_add_u32,     "v_add_u32 %0, %0, %1")
KERNEL32(k_mul_lo_u32,  "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32,  "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_fma_f32,     "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_xor_b32,     "v_xor_b32 %0, %0, %1")
KERNEL32(k_lshl_add,    "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_cndmask,     "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_mov_dpp,     "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp,     "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

#define KERNEL64(NAME, ASMSTR)                                                   \
__global__ void NAME(uint32_t* out, uint32_t seed) {                             \
  double x[CHAINS];                                                              \
  double m = 1.0000001 + seed * 1e-9, c = 0.25 + seed * 1e-9;                    \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) x[i] = 1.0 + (threadIdx.x * 7u + i) * 1e-6; \
  for (int it = 0; it < ITERS; ++it) {                                           \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i)                           \
      asm volatile(ASMSTR : "+v"(x[i]) : "v"(m), "v"(c));                        \
  }                                                                              \
  double s = 0;                                                                  \
  _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += x[i];                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(s); \
}

KERNEL64(k_fma_f64,   "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul_f64,   "v_mul_f64 %0, %0, %1")
KERNEL64(k_add_f64,   "v_add_f64 %0, %0, %2")
KERNEL64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL64(k_floor_f64, "v_floor_f64 %0, %0")
KERNEL64(k_pk_fma_f32,"v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_add_f32,"v_pk_add_f32 %0, %0, %2")
KERNEL64(k_lshl_b64,  "v_lshlrev_b64 %0, 1, %0")
KERNEL64(k_mov_b64,   "v_mov_b64 %0, %1")

// 32x32+64 -> 64 multiply-add (the building block of 64-bit modular products)
__global__ void k_mad_u64_u32(uint32_t* out, uint32_t seed) {
  unsigned long long x[CHAINS];
  uint32_t m = seed | 1u, c = seed * 3u + 7u;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 7u + i + seed;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(m), "v"(c) : "vcc");
  }
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

// conversions used at the NTT's integer<->double boundary
__global__ void k_cvt_f64_i32(uint32_t* out, uint32_t seed) {
  double x[CHAINS]; uint32_t y[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) y[i] = threadIdx.x * 7u + i + seed;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i)
      asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(s);
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint32_t* out;
  CK(hipMalloc(&out, sizeof(uint32_t) * cus * 16 * 256));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor_b32}, {"v_lshl_add_u32", k_lshl_add},
    {"v_cndmask_b32", k_cndmask}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_u32_dpp", k_add_dpp},
    {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
    {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24},
    {"v_mad_u64_u32", k_mad_u64_u32}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
    {"v_pk_add_f32", k_pk_add_f32},
    {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
    {"v_rndne_f64", k_rndne_f64}, {"v_floor_f64", k_floor_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32},
    {"v_lshlrev_b64", k_lshl_b64}, {"v_mov_b64", k_mov_b64},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // waves per SIMD swept by blocks-per-CU of 256 threads (1 block = 1 wave on each of 4 SIMDs)
  int bpcs[] = {1, 2, 4, 8};
  printf("%-16s", "instr");
  for (int b : bpcs) printf("  w/SIMD=%d cyc/instr(wave) lanes/clk/CU", b);
  printf("\n");
  for (auto& e : es) {
    printf("%-16s", e.name);
    for (int bpc : bpcs) {
      int grid = cus * bpc;
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u);  // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 12345u);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double instr_per_wave = (double)ITERS * CHAINS;
      double clk = 2.4e9;  // nominal; real clock may be lower under load
      double cycles = ms * 1e-3 * clk;
      // per SIMD: bpc waves, each instr_per_wave instructions
      double cyc_per_instr = cycles / (instr_per_wave * bpc);
      double lanes_per_clk_cu = 64.0 * 4 / cyc_per_instr;
      printf("  %8.3f ms %6.2f %7.1f", ms, cyc_per_instr, lanes_per_clk_cu);
    }
    printf("\n");
  }
  CK(hipFree(out));
  return 0;
}
