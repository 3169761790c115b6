// contention.hip -- why do the streaming front-end kernels slow down 4-19x next to a resident accumulate grid?
// A: VALU hog shaped like accumulate_kernel (64-thread workgroups, 2 waves/SIMD, v_mad_u64_u32 chains), with an
//    optional random 64-byte gather every G iterations (accumulate does one per ~12 k cycles).
// B: streaming kernel shaped like digits_kernel (reads 32 B, writes 34 B per thread) on a high-priority stream.
// Prints B's time alone and while A variants are resident.
//   hipcc -O3 --offload-arch=gfx950 -o contention contention.hip && ./contention
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

template <int N, int CH = 8>
__device__ __forceinline__ void rounds(uint64_t (&acc)[8], uint32_t (&a)[8], uint32_t& b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (CH == 8) acc[i] = (uint64_t)a[i] * b + acc[i];
    else acc[0] = (uint64_t)((uint32_t)acc[0] ^ a[i]) * b + acc[0];   // one dependent chain: the VALU idles between mads
  }
  a[N & 7] ^= (uint32_t)acc[(N + 3) & 7];   // data dependence: no algebraic merging of rounds
  b += 2 * N + 1;   // a different literal per round keeps the rounds from being re-rolled
  if constexpr (N > 1) rounds<N - 1, CH>(acc, a, b);
}

// R = unrolled rounds per loop iteration (code size: ~0.27 KB per round); PIN = allocate 192 VGPRs like the real kernel
template <int R, bool PIN, int CH = 8>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
hog_kernel(const uint4* __restrict__ table, uint32_t table_mask, uint32_t iters, uint32_t gather_every,
           uint64_t* __restrict__ sink) {
  if (PIN) asm volatile("v_mov_b32 v190, 0" ::: "v190");
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  uint64_t acc[8];
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = t + i;
    a[i] = t * 2654435761u + i;
  }
  uint32_t b = t | 1u, h = t * 747796405u;
  for (uint32_t it = 0; it < iters; ++it) {
    rounds<R, CH>(acc, a, b);
    if (gather_every && (it % gather_every) == 0) {
      h = h * 1664525u + 1013904223u;
      const uint4* rec = table + (size_t)((h >> 4) & table_mask) * 4;   // one 64-byte record
      uint4 v0 = rec[0], v1 = rec[1], v2 = rec[2], v3 = rec[3];
      a[0] ^= v0.x; a[1] ^= v1.y; a[2] ^= v2.z; a[3] ^= v3.w;
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + a[i];
  sink[t] = s;
}

// coarse_hist-like: 1024 threads, LDS atomics into 128 counters, u16 reads
__global__ void __launch_bounds__(1024)
hist_kernel(const uint16_t* __restrict__ in, uint32_t n, uint32_t chunk, uint32_t* __restrict__ out, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  __shared__ uint32_t h[128];
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  const uint16_t* src = in + (size_t)blockIdx.y * n;
  const uint32_t lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) atomicAdd(&h[(src[i] >> 7) & 127u], 1u);
  __syncthreads();
  if (threadIdx.x < 128) out[(blockIdx.y * gridDim.x + blockIdx.x) * 128 + threadIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256)
stream_kernel(const uint4* __restrict__ in, uint32_t n, uint16_t* __restrict__ out, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const uint4 lo = in[2 * (size_t)t], hi = in[2 * (size_t)t + 1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int k = 0; k < 17; ++k) out[(size_t)k * n + t] = (uint16_t)(w[k & 7] >> (k & 15));
}

int main() {
  const uint32_t n = 1u << 20, recs = 1u << 20;
  uint4 *d_in, *d_table;
  uint16_t* d_out;
  uint64_t* d_sink;
  const uint32_t hog_wgs = 6200;
  CK(hipMalloc(&d_in, (size_t)n * 32));
  CK(hipMalloc(&d_out, (size_t)n * 34));
  CK(hipMalloc(&d_table, (size_t)recs * 64));
  CK(hipMalloc(&d_sink, (size_t)hog_wgs * 64 * 8));
  CK(hipMemset(d_in, 1, (size_t)n * 32));
  CK(hipMemset(d_table, 2, (size_t)recs * 64));
  int lo_prio, hi_prio;
  CK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo_prio));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi_prio));
  hipEvent_t a0, a1, b0, b1;
  CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));

  uint32_t* d_hist;
  CK(hipMalloc(&d_hist, 17 * 30 * 128 * 4));
  auto run_b = [&](int which, int reps, float* avg_ms) -> int {
    float tot = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(b0, sb));
      if ((which & 1) == 0)
        hipLaunchKernelGGL(stream_kernel, dim3(n / 256), dim3(256), 0, sb, (const uint4*)d_in, n, d_out, which >> 1);
      else
        hipLaunchKernelGGL(hist_kernel, dim3(30, 17), dim3(1024), 0, sb, (const uint16_t*)d_out, n, (n + 29) / 30, d_hist, which >> 1);
      CK(hipEventRecord(b1, sb));
      CK(hipEventSynchronize(b1));
      float ms;
      CK(hipEventElapsedTime(&ms, b0, b1));
      tot += ms;
    }
    *avg_ms = tot / reps;
    return 0;
  };
  float alone[4];
  for (int w = 0; w < 4; ++w) {
    if (run_b(w, 3, &alone[w])) return 1;
    if (run_b(w, 10, &alone[w])) return 1;
  }
  std::printf("alone: stream kernel %.1f us, hist kernel %.1f us (with s_setprio: %.1f, %.1f)\n", alone[0] * 1e3, alone[1] * 1e3, alone[2] * 1e3, alone[3] * 1e3);

  using Hog = void (*)(const uint4*, uint32_t, uint32_t, uint32_t, uint64_t*);
  struct Variant { const char* name; Hog fn; uint32_t iters, gather_every; };
  const Variant vs[] = {
      {"0.4 KB loop, 56 VGPRs, VALU only", hog_kernel<4, false>, 6000, 0},
      {"0.4 KB loop, 192 VGPRs, gather/70 it", hog_kernel<4, true>, 6000, 70},
      {"8 KB loop", hog_kernel<93, true>, 258, 0},
      {"12 KB loop", hog_kernel<140, true>, 171, 0},
      {"16 KB loop", hog_kernel<186, true>, 129, 0},
      {"20 KB loop", hog_kernel<233, true>, 103, 0},
      {"24 KB loop", hog_kernel<280, true>, 86, 0},
      {"28 KB loop", hog_kernel<326, true>, 74, 0},
      {"32 KB loop", hog_kernel<373, true>, 64, 0},
      {"40 KB loop", hog_kernel<466, true>, 52, 0},
      {"48 KB loop", hog_kernel<560, true>, 43, 0},
      {"48 KB loop, gather/2 it", hog_kernel<560, true>, 43, 2},
      {"0.6 KB loop, dependent chain", hog_kernel<4, true, 1>, 3000, 0},
      {"24 KB loop, dependent chain", hog_kernel<200, true, 1>, 60, 0},
      {"48 KB loop, dependent chain", hog_kernel<400, true, 1>, 30, 0},
  };
  for (const Variant& v : vs) {
    CK(hipEventRecord(a0, sa));
    hipLaunchKernelGGL(v.fn, dim3(hog_wgs), dim3(64), 0, sa, (const uint4*)d_table, recs - 1, v.iters, v.gather_every,
                       d_sink);
    CK(hipEventRecord(a1, sa));
    CK(hipEventSynchronize(a1));
    float hog_alone;
    CK(hipEventElapsedTime(&hog_alone, a0, a1));
    float with[4];
    bool running = true;
    for (int w = 0; w < 4; ++w) {
      CK(hipEventRecord(a0, sa));
      for (int k = 0; k < 6; ++k)
        hipLaunchKernelGGL(v.fn, dim3(hog_wgs), dim3(64), 0, sa, (const uint4*)d_table, recs - 1, v.iters,
                           v.gather_every, d_sink);
      CK(hipEventRecord(a1, sa));
      if (run_b(w, 8, &with[w])) return 1;
      running = running && hipEventQuery(a1) == hipErrorNotReady;
      CK(hipEventSynchronize(a1));
    }
    std::printf("%-40s hog alone %.2f ms | stream %.1f us (x%.1f) | hist %.1f us (x%.1f) | with s_setprio 3: stream %.1f us, "
                "hist %.1f us%s\n", v.name, hog_alone, with[0] * 1e3, with[0] / alone[0], with[1] * 1e3, with[1] / alone[1],
                with[2] * 1e3, with[3] * 1e3, running ? "" : "  [hog finished early!]");
  }
  return 0;
}
