// Issue intervals of the accumulate kernel's instruction kinds on gfx950, measured IN CYCLES by the wave itself.
//
// valu_rates.hip / valu_mix.hip time one 0.05-0.3 ms launch with HIP events and convert with an assumed 2.4 GHz: launch
// overhead and a clock that has not ramped up inflate their figures (their sum for the mixed addition, 1549 x 5.3 +
// 685 x 3.0 = 10 265 cycles, is MORE than the 9 520 cycles the real mixed addition takes at the same occupancy).  Here
// every wave stamps s_memtime (the shader clock) around a loop of >= 5 ms after the device has been kept busy for a
// second, so the result does not depend on what the clock was; s_memrealtime (100 MHz) gives the clock itself.
//
// Output: cycles per instruction per SIMD (= loop cycles / (instructions per trip x waves per SIMD)) for 1, 2, 3, 4
// waves per SIMD.  bench.py's roofline.secondary.peak is the v_mad_u64_u32 row of this table at the occupancy of the
// shipped accumulate kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 40000;

#define MADV(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[i]) : "v"(a), "v"(b) : "vcc");
#define MADS(i) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(m[i]), "=s"(sink) : "v"(a), "v"(b));
#define MADK(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[i]) : "v"(a), "s"(k) : "vcc");   // constant operand in an SGPR
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define DFMA(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(m[i]) : "v"(a64));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(m[i]));
#define CHAIN(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[0]) : "v"(a), "v"(r[i]) : "vcc");
#define CHAIN2(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[(i) & 1]) : "v"(a), "v"(r[i]) : "vcc");
#define CHAIN4(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[(i) & 3]) : "v"(a), "v"(r[i]) : "vcc");

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define MIX_1_1(A, B) A(0) B(0) A(1) B(1) A(2) B(2) A(3) B(3) A(4) B(4) A(5) B(5) A(6) B(6) A(7) B(7) A(8) B(8) A(9) B(9) A(10) B(10) A(11) B(11) A(12) B(12) A(13) B(13) A(14) B(14) A(15) B(15)
// 16 multiply-adds + 8 others: the accumulate kernel's ratio (1549 : 685) is 16 : 7
#define MIX_2_1(A, B) A(0) A(1) B(0) A(2) A(3) B(1) A(4) A(5) B(2) A(6) A(7) B(3) A(8) A(9) B(4) A(10) A(11) B(5) A(12) A(13) B(6) A(14) A(15) B(7)

#define KERNEL(NAME, BODY)                                                                              \
__global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t s, int iters) {                     \
  uint64_t m[16]; uint32_t r[16];                                                                       \
  uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9e3779b9u;                                      \
  uint64_t a64 = ((uint64_t)a << 20) | b; uint32_t k = s * 77u + 5u; uint64_t sink = 0;                      \
  for (int i = 0; i < 16; ++i) { m[i] = a64 + i; r[i] = a + i; }                                        \
  asm volatile("" : "+s"(k));                                                                           \
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();              \
  for (int it = 0; it < iters; ++it) { BODY }                                                           \
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();              \
  uint64_t acc = 0; for (int i = 0; i < 16; ++i) acc ^= m[i] ^ r[i];                                    \
  if (acc == 0x12345) out[0] = acc + sink;                                                                     \
  if ((threadIdx.x & 63) == 0) {                                                                        \
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);                                             \
    out[1 + 2 * w] = t1 - t0; out[2 + 2 * w] = q1 - q0;                                                 \
  }                                                                                                     \
}
KERNEL(k_mad_vcc, R16(MADV))
KERNEL(k_mad_sgpr, R16(MADS))
KERNEL(k_mad_const, R16(MADK))
KERNEL(k_add, R16(ADD))
KERNEL(k_and, R16(AND))
KERNEL(k_fma32, R16(FMA))
KERNEL(k_fma64, R16(DFMA))
KERNEL(k_mullo, R16(MULLO))
KERNEL(k_shr64, R16(SHR64))
KERNEL(k_chain1, R16(CHAIN))
KERNEL(k_chain2, R16(CHAIN2))
KERNEL(k_chain4, R16(CHAIN4))
KERNEL(k_mad_add_1_1, MIX_1_1(MADV, ADD))
KERNEL(k_mad_add_2_1, MIX_2_1(MADV, ADD))
KERNEL(k_mad_and_2_1, MIX_2_1(MADV, AND))
KERNEL(k_mad_shr_2_1, MIX_2_1(MADV, SHR64))
KERNEL(k_chain1_add, MIX_2_1(CHAIN, ADD))

struct Entry { const char* name; void (*fn)(uint64_t*, uint32_t, int); int instr; };

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint64_t* d; CHECK(hipMalloc(&d, (1 + 2 * cus * 4 * 8) * sizeof(uint64_t)));
  std::vector<uint64_t> h(1 + 2 * cus * 4 * 8);
  Entry es[] = {{"v_mad_u64_u32 (carry -> vcc)", k_mad_vcc, 16}, {"v_mad_u64_u32 (carry -> SGPR pair)", k_mad_sgpr, 16},
                {"v_mad_u64_u32 (SGPR multiplicand)", k_mad_const, 16},
                {"v_add_u32", k_add, 16}, {"v_and_b32", k_and, 16}, {"v_fma_f32", k_fma32, 16}, {"v_fma_f64", k_fma64, 16},
                {"v_mul_lo_u32", k_mullo, 16}, {"v_lshrrev_b64", k_shr64, 16},
                {"mad, ONE dependent chain", k_chain1, 16}, {"mad, two chains", k_chain2, 16}, {"mad, four chains", k_chain4, 16},
                {"16 mad + 16 add", k_mad_add_1_1, 32}, {"16 mad + 8 add", k_mad_add_2_1, 24}, {"16 mad + 8 and", k_mad_and_2_1, 24},
                {"16 mad + 8 lshrrev_b64", k_mad_shr_2_1, 24}, {"16 chained mad + 8 add", k_chain1_add, 24}};
  // keep the device busy for about a second first
  for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(k_mad_add_2_1, dim3(cus * 2), dim3(256), 0, 0, d, 1u, ITER);
  CHECK(hipDeviceSynchronize());
  printf("device %s  CUs=%d\n", prop.name, cus);
  printf("%-36s %s\n", "cycles per instruction per SIMD", "at 1, 2, 3, 4, 6, 8 waves per SIMD: median wave (shader clock GHz; ns per instruction per SIMD)");
  for (auto& e : es) {
    printf("%-36s", e.name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
      const int blocks = cus * wps;   // 256 threads = 4 waves per block = 1 wave per SIMD per block
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u, ITER);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 2u, ITER);
      CHECK(hipMemcpy(h.data(), d, (1 + 2 * blocks * 4) * sizeof(uint64_t), hipMemcpyDeviceToHost));
      std::vector<double> cyc, ghz;
      for (int w = 0; w < blocks * 4; ++w) {
        cyc.push_back((double)h[1 + 2 * w]);
        ghz.push_back((double)h[1 + 2 * w] / ((double)h[2 + 2 * w] * 10.0));   // 100 MHz real-time ticks
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(ghz.begin(), ghz.end());
      const double med = cyc[cyc.size() / 2];
      const double cpi = med / ((double)ITER * e.instr * wps), g = ghz[ghz.size() / 2];
      printf(" %7.3f (%5.3f; %5.3f)", cpi, g, cpi / g);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
