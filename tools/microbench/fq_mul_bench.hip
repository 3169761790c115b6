// Cycles per 256-bit Montgomery multiplication (Fq::mul) on gfx950: dependent chains, 1 or 2 chains per
// lane, 1..4 waves per SIMD.  Development aid for tuning bn254_fq.hip.h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_fq.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2000;

template <int CHAINS, int VARIANT>
__global__ void __launch_bounds__(256) k_mul(const u256* in, u256* out) {
  u256 x[CHAINS], y;
  y = in[threadIdx.x & 63];
  for (int c = 0; c < CHAINS; ++c) { x[c] = in[(threadIdx.x + c + 1) & 63]; }
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (VARIANT == 0) x[c] = Fq::mul(x[c], y);
      else if (VARIANT == 1) x[c] = Fq::sqr(x[c]);
      else if (VARIANT == 2) x[c] = Fq::add(x[c], y);
      else if (VARIANT == 3) x[c] = Fq::sub(x[c], y);
    }
  }
  u256 r = x[0];
  for (int c = 1; c < CHAINS; ++c) r = Fq::add(r, x[c]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CHAINS, int VARIANT>
void run(const char* name, const u256* din, u256* dout, int cus, double ghz) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL((k_mul<CHAINS, VARIANT>), dim3(blocks), dim3(256), 0, 0, din, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mul<CHAINS, VARIANT>), dim3(blocks), dim3(256), 0, 0, din, dout);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops_per_simd = (double)ITER * CHAINS * wps;       // wave-level field ops issued on one SIMD
    double ns = ms * 1e6;
    printf("%-22s chains=%d waves/SIMD=%d  %8.3f ms  %8.1f cyc@2.4GHz per wave-op  (%.2f G lane-ops/s chip)\n", name,
           CHAINS, wps, ms, ns * ghz / ops_per_simd, ops_per_simd * 4 * cus * 64 / ns);
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double ghz = 2.4;
  u256 h[64];
  for (int i = 0; i < 64; ++i) for (int l = 0; l < 8; ++l) h[i].v[l] = (l == 7) ? (0x1234567u + i) : (0x9E3779B9u * (i * 8 + l + 1));
  u256 *din, *dout; CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(u256) * 256 * cus * 4));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  run<1, 0>("Fq::mul", din, dout, cus, ghz);
  run<2, 0>("Fq::mul", din, dout, cus, ghz);
  run<1, 1>("Fq::sqr", din, dout, cus, ghz);
  run<1, 2>("Fq::add", din, dout, cus, ghz);
  run<1, 3>("Fq::sub", din, dout, cus, ghz);
  return 0;
}
