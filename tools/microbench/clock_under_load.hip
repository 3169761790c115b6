// What clock does the chip hold under the accumulate kernel's instruction mix?  (DESIGN.md section 4 derives
// ~1.6-1.7 GHz from SQ_WAVE_CYCLES; this measures it directly, the way MI355X_MICROARCH.md prescribes: in-kernel
// clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped once around the loop after >= 2 s of back-to-back
// launches on random data.)  The loop is the production mixed addition pti_madd, 2 waves per SIMD on every SIMD.
// The stamps go to a buffer of their own; no output value depends on them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_ec29.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 400;

__global__ void __launch_bounds__(256) k_madd_loop(const u256* in, u256* out, uint64_t* stamps, int mode) {
  const u256 xe = in[threadIdx.x & 63], ye = in[(threadIdx.x + 7) & 63];
  Affine qa; qa.x = xe; qa.y = ye;
  const AffI q = affi_from_ext(qa);
  PtI acc = pti_from_affi(q);
  acc.x = Fq29::from_ext(ye);
  uint64_t t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  if (mode == 0) {
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) acc = pti_madd(acc, q);
  } else {   // a memory-only loop for comparison: the same trip count of dependent loads
    uint32_t v = threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < ITER * 40; ++i) v = in[v & 63].v[i & 7] + i;
    acc.x.l[0] ^= v;
  }
  if (threadIdx.x == 0) {
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = Fq29::to_ext(acc.x);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 2;   // 256 threads = 1 wave per SIMD; x2 = 2 waves per SIMD
  u256 h[64];
  for (int i = 0; i < 64; ++i) for (int l = 0; l < 8; ++l) h[i].v[l] = (l == 7) ? (0x1234567u + i) : (0x9E3779B9u * (i * 8 + l + 1));
  u256 *din, *dout; uint64_t* dst;
  CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(u256) * 256 * blocks)); CHECK(hipMalloc(&dst, 16 * blocks));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {   // heat soak
      for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_madd_loop, dim3(blocks), dim3(256), 0, 0, din, dout, dst, mode);
      CHECK(hipDeviceSynchronize());
      launches += 20;
    }
    hipLaunchKernelGGL(k_madd_loop, dim3(blocks), dim3(256), 0, 0, din, dout, dst, mode);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> st(2 * blocks);
    CHECK(hipMemcpy(st.data(), dst, 16 * blocks, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b)
      if (st[2 * b + 1]) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("%-34s after %4d launches (2.5 s): in-kernel clock median %.3f GHz (min %.3f, max %.3f), nominal %.3f GHz\n",
           mode == 0 ? "pti_madd loop, 2 waves/SIMD" : "dependent-load loop (no VALU work)", launches, ghz[ghz.size() / 2],
           ghz.front(), ghz.back(), prop.clockRate / 1e6);
  }
  return 0;
}
