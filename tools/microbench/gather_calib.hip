// Calibration of rocprofv3 FETCH_SIZE for the accumulate kernel's access pattern on gfx950: every lane gathers
// one REC-byte record (as 16-byte loads) at a random index of a table.  Known byte count = lanes * REC.
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int REC>
__global__ void __launch_bounds__(64) gather(const uint4* __restrict__ table, uint32_t nrec, uint32_t per_lane, uint32_t* out) {
  uint32_t x = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per_lane; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32_t r = (x >> 4) % nrec;
    const uint4* p = table + (size_t)r * (REC / 16);
#pragma unroll
    for (int k = 0; k < REC / 16; ++k) { const uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int REC>
void run(size_t table_bytes, uint32_t lanes, uint32_t per_lane) {
  const uint32_t nrec = (uint32_t)(table_bytes / REC);
  uint4* t; uint32_t* o;
  CHECK(hipMalloc(&t, (size_t)nrec * REC)); CHECK(hipMalloc(&o, 64));
  CHECK(hipMemset(t, 1, (size_t)nrec * REC));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((gather<REC>), dim3(lanes / 64), dim3(64), 0, 0, t, nrec, per_lane, o);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)lanes * per_lane * REC;
  printf("gather<%d> table=%zu MB  gathered=%.1f MB  %.3f ms  %.1f GB/s\n", REC, table_bytes >> 20, bytes / 1e6, ms, bytes / ms / 1e6);
  CHECK(hipFree(t)); CHECK(hipFree(o));
}

int main() {
  const uint32_t lanes = 1u << 20, per = 16;           // 16.8 M records per launch, like 2^20 x 17 windows
  run<80>((size_t)84 << 20, lanes, per);                // 80-byte internal bases, table fits the Infinity Cache
  run<64>((size_t)64 << 20, lanes, per);
  run<128>((size_t)128 << 20, lanes, per);
  run<80>((size_t)1344 << 20, lanes, per);              // table far beyond the Infinity Cache
  run<64>((size_t)1024 << 20, lanes, per);
  return 0;
}
