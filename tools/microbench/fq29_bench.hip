// Cycles per operation of the 29-bit-limb internal field / group law on gfx950 (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_ec29.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 1000;

template <int VARIANT>
__global__ void __launch_bounds__(256) k_op(const u256* in, u256* out) {
  const u256 xe = in[threadIdx.x & 63], ye = in[(threadIdx.x + 7) & 63];
  if (VARIANT <= 3 || VARIANT == 10) {
    fe29 x = Fq29::from_ext(xe), y = Fq29::from_ext(ye);
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
      if (VARIANT == 0) x = Fq29::mul(x, y);
      if (VARIANT == 1) x = Fq29::sqr(x);
      if (VARIANT == 2) x = Fq29::norm(Fq29::sub<K16E30>(y, x));
      if (VARIANT == 3) x = Fq29::norm(Fq29::add(x, y));
      if (VARIANT == 10) x = Fq29::mul_karatsuba(x, y);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Fq29::to_ext(x);
  } else if (VARIANT >= 6 && VARIANT <= 9) {
    // two multiplications per trip: one after the other, or side by side in lockstep product-scanning chains
    fe29 x = Fq29::from_ext(xe), y = Fq29::from_ext(ye), z = Fq29::from_ext(in[(threadIdx.x + 13) & 63]);
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
      fe29 a, b;
      if (VARIANT == 6) { a = Fq29::mul(x, y); b = Fq29::mul(z, y); }
      if (VARIANT == 7) Fq29::mul_pair(x, y, z, y, a, b);
      if (VARIANT == 8) { a = Fq29::sqr(x); b = Fq29::sqr(z); }
      if (VARIANT == 9) Fq29::sqr_pair(x, z, a, b);
      x = a;
      z = b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Fq29::to_ext(Fq29::norm(Fq29::add(x, z)));
  } else {
    Affine qa; qa.x = xe; qa.y = ye;           // not a curve point: timing only (no exceptional paths taken)
    const AffI q = affi_from_ext(qa);
    PtI acc = pti_from_affi(q);
    acc.x = Fq29::from_ext(ye);
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
      if (VARIANT == 4) acc = pti_madd(acc, q);
      if (VARIANT == 5) acc = pti_add_nz(acc, acc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Fq29::to_ext(acc.x);
  }
}

template <int VARIANT>
void run(const char* name, const u256* din, u256* dout, int cus) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL((k_op<VARIANT>), dim3(blocks), dim3(256), 0, 0, din, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_op<VARIANT>), dim3(blocks), dim3(256), 0, 0, din, dout);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops_per_simd = (double)ITER * wps;
    printf("%-14s waves/SIMD=%d  %8.3f ms  %9.1f cyc@2.4GHz per wave-op\n", name, wps, ms, ms * 1e6 * 2.4 / ops_per_simd);
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u256 h[64];
  for (int i = 0; i < 64; ++i) for (int l = 0; l < 8; ++l) h[i].v[l] = (l == 7) ? (0x1234567u + i) : (0x9E3779B9u * (i * 8 + l + 1));
  u256 *din, *dout; CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(u256) * 256 * cus * 4));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>("Fq29::mul", din, dout, cus);
  run<1>("Fq29::sqr", din, dout, cus);
  run<2>("Fq29::sub+norm", din, dout, cus);
  run<3>("Fq29::add+norm", din, dout, cus);
  run<10>("mul_karatsuba", din, dout, cus);
  run<6>("2 x mul", din, dout, cus);
  run<7>("mul_pair", din, dout, cus);
  run<8>("2 x sqr", din, dout, cus);
  run<9>("sqr_pair", din, dout, cus);
  run<4>("pti_madd", din, dout, cus);
  run<5>("pti_add_nz", din, dout, cus);
  return 0;
}
