// Do VALU instruction costs ADD UP on gfx950, or do cheap instructions hide under the 64-bit multiply-adds?
// The accumulate kernel's mixed addition is 1549 v_mad_u64_u32 / v_mul_lo_u32 plus ~700 simple instructions; removing
// 180 of the simple ones (lockstep product scanning) or trading 27 multiplies for ~80 simple ones (Karatsuba) both
// left the time unchanged -- this benchmark measures the mix directly: per loop trip 8 independent v_mad_u64_u32
// and K independent instructions of another kind, K = 0, 8, 16.  If costs add, the time grows by K x the other
// instruction's own issue interval; if they overlap, it stays near the 8-multiply time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2048;

#define MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[i]) : "v"(a), "v"(b) : "vcc");
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(a64));
#define ADDC(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R8B(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// interleave: one multiply, then K/8 others
#define MIX1(X) MAD(0) X(0) MAD(1) X(1) MAD(2) X(2) MAD(3) X(3) MAD(4) X(4) MAD(5) X(5) MAD(6) X(6) MAD(7) X(7)
#define MIX2(X) MAD(0) X(0) X(8) MAD(1) X(1) X(9) MAD(2) X(2) X(10) MAD(3) X(3) X(11) MAD(4) X(4) X(12) MAD(5) X(5) X(13) MAD(6) X(6) X(14) MAD(7) X(7) X(15)

#define KERNEL(NAME, BODY)                                                              \
__global__ void NAME(uint32_t* out, uint32_t s) {                                       \
  uint64_t m[8], q[16]; uint32_t r[16];                                                 \
  uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9e3779b9u;                      \
  uint64_t a64 = ((uint64_t)a << 20) | b;                                               \
  for (int i = 0; i < 8; ++i) m[i] = a64 + i;                                           \
  for (int i = 0; i < 16; ++i) { q[i] = a64 * (i + 3); r[i] = a + i; }                  \
  for (int it = 0; it < ITER; ++it) { BODY }                                            \
  uint64_t acc = 0; for (int i = 0; i < 8; ++i) acc ^= m[i];                            \
  for (int i = 0; i < 16; ++i) acc ^= q[i] ^ r[i];                                      \
  if (acc == 0x12345) out[0] = (uint32_t)acc;                                           \
}
KERNEL(k_mad8, R8(MAD))
KERNEL(k_add8, R8(ADD))
KERNEL(k_add16, R8(ADD) R8B(ADD))
KERNEL(k_mad8_add8, MIX1(ADD))
KERNEL(k_mad8_add16, MIX2(ADD))
KERNEL(k_mad8_and16, MIX2(AND))
KERNEL(k_shr8, R8(SHR64))
KERNEL(k_mad8_shr8, MIX1(SHR64))
KERNEL(k_mad8_shr16, MIX2(SHR64))
KERNEL(k_lshladd8, R8(LSHLADD64))
KERNEL(k_mad8_lshladd8, MIX1(LSHLADD64))
KERNEL(k_mad8_lshladd16, MIX2(LSHLADD64))
KERNEL(k_mad8_addc8, MIX1(ADDC))
KERNEL(k_mullo8, R8(MULLO))
KERNEL(k_mad8_mullo8, MIX1(MULLO))

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); };
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* d; CHECK(hipMalloc(&d, 4096));
  Entry es[] = {{"8 mad", k_mad8}, {"8 add", k_add8}, {"16 add", k_add16}, {"8 mad + 8 add", k_mad8_add8},
                {"8 mad + 16 add", k_mad8_add16}, {"8 mad + 16 and", k_mad8_and16}, {"8 lshrrev_b64", k_shr8},
                {"8 mad + 8 lshrrev_b64", k_mad8_shr8}, {"8 mad + 16 lshrrev_b64", k_mad8_shr16},
                {"8 lshl_add_u64", k_lshladd8}, {"8 mad + 8 lshl_add_u64", k_mad8_lshladd8},
                {"8 mad + 16 lshl_add_u64", k_mad8_lshladd16}, {"8 mad + 8 (add_co,addc)", k_mad8_addc8},
                {"8 mul_lo", k_mullo8}, {"8 mad + 8 mul_lo", k_mad8_mullo8}};
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-28s %10s %10s %10s   (cycles @2.4 GHz per loop trip per SIMD, per wave resident)\n", "mix", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (auto& e : es) {
    printf("%-28s", e.name);
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;   // 256 threads = 4 waves per block = 1 wave per SIMD per block
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %10.1f", ms * 1e6 * 2.4 / ((double)ITER * wps));
    }
    printf("\n");
  }
  return 0;
}
