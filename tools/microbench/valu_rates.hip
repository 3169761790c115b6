// VALU instruction-rate microbenchmark for gfx950 (MI355X).
// Purpose: decide which multiplier path the 256-bit Montgomery field arithmetic
// of the MSM hot path should be built on (v_mad_u64_u32 vs 24-bit vs FP64 FMA).
// Each kernel issues ITER x 16 independent copies of one instruction per lane;
// the host reports wave-instructions per ns per CU and cycles per wave-instr per SIMD
// (assuming the clock printed by hipDeviceProp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 4096;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// ---- 32-bit dst, pattern: op d, a, b  (d accumulates through itself)
#define K_U32_2SRC(NAME, ASMSTR)                                                      \
__global__ void NAME(uint32_t* out, uint32_t s) {                                     \
  uint32_t r[16]; uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9e3779b9u;    \
  for (int i = 0; i < 16; ++i) r[i] = a + i;                                         \
  for (int it = 0; it < ITER; ++it) {                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                    \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b));                             \
  }                                                                                   \
  uint32_t acc = 0; for (int i = 0; i < 16; ++i) acc ^= r[i];                         \
  if (acc == 0x12345) out[0] = acc;                                                   \
}

K_U32_2SRC(k_add_u32,        "v_add_u32 %0, %0, %1")
K_U32_2SRC(k_add3_u32,       "v_add3_u32 %0, %0, %1, %2")
K_U32_2SRC(k_mul_lo_u32,     "v_mul_lo_u32 %0, %0, %1")
K_U32_2SRC(k_mul_hi_u32,     "v_mul_hi_u32 %0, %0, %1")
K_U32_2SRC(k_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
K_U32_2SRC(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
K_U32_2SRC(k_mad_u32_u24,    "v_mad_u32_u24 %0, %1, %2, %0")
K_U32_2SRC(k_dot4_u32_u8,    "v_dot4_u32_u8 %0, %1, %2, %0")
K_U32_2SRC(k_alignbit,       "v_alignbit_b32 %0, %0, %1, 13")
K_U32_2SRC(k_perm_b32,       "v_perm_b32 %0, %0, %1, %2")
K_U32_2SRC(k_lshl_add,       "v_lshl_add_u32 %0, %0, 3, %1")
K_U32_2SRC(k_fma_f32,        "v_fma_f32 %0, %1, %2, %0")
K_U32_2SRC(k_addc_chain,     "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
K_U32_2SRC(k_mad_u16,        "v_mad_u16 %0, %1, %2, %0")
K_U32_2SRC(k_pk_mad_u16,     "v_pk_mad_u16 %0, %1, %2, %0")
K_U32_2SRC(k_cvt_f32_u32,    "v_cvt_f32_u32 %0, %0")

// ---- 64-bit dst
#define K_U64(NAME, ASMSTR)                                                           \
__global__ void NAME(uint32_t* out, uint32_t s) {                                     \
  uint64_t r[16]; uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9e3779b9u;    \
  uint64_t a64 = ((uint64_t)a << 20) | b;                                             \
  for (int i = 0; i < 16; ++i) r[i] = a64 + i;                                        \
  for (int it = 0; it < ITER; ++it) {                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                    \
      asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b), "v"(a64) : "vcc");           \
  }                                                                                   \
  uint64_t acc = 0; for (int i = 0; i < 16; ++i) acc ^= r[i];                         \
  if (acc == 0x12345) out[0] = (uint32_t)acc;                                         \
}

K_U64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
K_U64(k_mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %1, %2, %0")
K_U64(k_fma_f64,      "v_fma_f64 %0, %3, %3, %0")
K_U64(k_mul_f64,      "v_mul_f64 %0, %0, %3")
K_U64(k_add_f64,      "v_add_f64 %0, %0, %3")
K_U64(k_pk_fma_f32,   "v_pk_fma_f32 %0, %3, %3, %0")
K_U64(k_pk_add_f32,   "v_pk_add_f32 %0, %0, %3")
K_U64(k_lshlrev_b64,  "v_lshlrev_b64 %0, 3, %0")
K_U64(k_cvt_f64_u32,  "v_cvt_f64_u32 %0, %1")

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); int instr_per_slot; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double ghz = prop.clockRate / 1e6;
  printf("device %s  CUs=%d  clock=%.3f GHz\n", prop.name, cus, ghz);
  uint32_t* d; CHECK(hipMalloc(&d, 4096));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
    {"v_add_co+v_addc_co (pair)", k_addc_chain, 2},
    {"v_lshl_add_u32", k_lshl_add, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_perm_b32", k_perm_b32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
    {"v_mad_u16", k_mad_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
    {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_lshlrev_b64", k_lshlrev_b64, 1},
    {"v_cvt_f32_u32", k_cvt_f32_u32, 1}, {"v_cvt_f64_u32", k_cvt_f64_u32, 1},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-28s %6s %12s %14s %16s\n", "instr", "w/SIMD", "ms", "winstr/ns/CU", "cyc/winstr/SIMD");
  for (auto& e : es) {
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;           // 256 threads per block = 4 waves = 1 per SIMD
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);  // warmup
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 2u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      double winstr_per_simd = (double)ITER * 16 * e.instr_per_slot * wps;   // wave-instrs issued on one SIMD
      double ns = ms * 1e6;
      double per_cu = winstr_per_simd * 4 / ns;
      double cyc = ns * ghz / winstr_per_simd;
      printf("%-28s %6d %12.4f %14.4f %16.3f\n", e.name, wps, ms, per_cu, cyc);
    }
  }
  CHECK(hipFree(d));
  return 0;
}
