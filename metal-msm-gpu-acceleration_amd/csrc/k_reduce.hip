// Stage 4: window reduction kernels.  See device_common.hip.h for the pipeline overview.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// ------------------------------------------------------------------------------------------------
// Stage 4a: per-segment running sums.  Slot i of a window holds the bucket of digit magnitude i + 1.  For
// segment s (slots 8s .. 8s+7):   S[w][s] = sum_j X[8s+j]     T[w][s] = sum_j j * X[8s+j]
// so that  sum_i (i+1) X[i] = sum_s (T[s] + S[s]) + 8 * sum_s s*S[s].   Replaces sum_reduction_partial
// (msm.h.metal:319-461), whose combine step needs a scalar multiplication per merge.
__global__ void __launch_bounds__(64)
reduce_seg_kernel(const PtI* __restrict__ buckets, uint32_t total_segs,
                  PtI* __restrict__ S, PtI* __restrict__ T) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total_segs) return;
  // Two accumulators plus an operand and the temporaries of an addition need ~195 VGPRs; `sos` is parked in LDS
  // while `sum` is updated, so the kernel stays at <= 160 and its waves fit beside two accumulate waves (176 VGPRs
  // each) of the next instance instead of waiting for that grid to drain.
  __shared__ PtI park[64];
  const PtI* X = buckets + (size_t)s * kSeg;
  PtI sum = pti_identity();
  store_pti(&park[threadIdx.x], pti_identity());
#pragma unroll 1
  for (int j = kSeg - 1; j >= 1; --j) {
    sum = pti_add(sum, load_pti(&X[j]));
    asm volatile("" ::: "memory");   // keep the LDS reload below the addition above (it is the point of parking)
    const PtI sos = pti_add(load_pti(&park[threadIdx.x]), sum);
    store_pti(&park[threadIdx.x], sos);
    asm volatile("" ::: "memory");
  }
  sum = pti_add(sum, load_pti(&X[0]));
  store_pti(&S[s], sum);
  store_pti(&T[s], load_pti(&park[threadIdx.x]));
}

// Stage 4b: tree sums.  grid = (K + 2, W, parts) with K = lb - 3 bits of segment index; block = tree_threads
// (power of two, 64..512); dynamic LDS = tree_threads * 144 bytes.  A window with many segments (the single
// 2^15 .. 2^20-slot window of the table pipeline) is cut into `parts` slices whose sums the host adds at the same
// bit position; the per-call pipeline has parts = 1.
//   blockIdx.x == K : partial[w][K] = sum_s T[w][s]        blockIdx.x == K + 1 : partial[w][K+1] = sum_s S[w][s]
//   blockIdx.x  < K : partial[w][k] = sum over segments s with bit k set of S[w][s]
// The host then evaluates  W_w = partial[w][K] + 8 * sum_k 2^k partial[w][k]  inside one Horner pass
// over all bit positions (replaces sum_reduction_final msm.h.metal:463-562 and the doublings of
// final_accumulation.rs:19-39).
__global__ void __launch_bounds__(512)
reduce_tree_kernel(const PtI* __restrict__ S, const PtI* __restrict__ T, uint32_t nseg,
                   uint32_t K, Jacobian* __restrict__ partial) {
  extern __shared__ uint32_t lds_u32[];
  PtI* sh = reinterpret_cast<PtI*>(lds_u32);
  const uint32_t k = blockIdx.x, w = blockIdx.y;
  const PtI* Sw = S + (size_t)w * nseg;
  const PtI* Tw = T + (size_t)w * nseg;
  PtI acc = pti_identity();
  if (k >= K) {
    // slot i carries weight i + 1:  sum_i (i+1) X[i] = sum_s T[s] + sum_s S[s] + 8 sum_s s S[s];
    // the two plain sums get a workgroup each (k == K: T, k == K + 1: S) to keep the critical path short
    const PtI* src = (k == K) ? Tw : Sw;
    const uint32_t len = nseg / gridDim.z, first = blockIdx.z * len;
#pragma unroll 1
    for (uint32_t s = first + threadIdx.x; s < first + len; s += blockDim.x) acc = pti_add(acc, load_pti(&src[s]));
  } else {
    const uint32_t half = nseg >> 1;
    const uint32_t lowmask = (1u << k) - 1u;
    const uint32_t len = half / gridDim.z, first = blockIdx.z * len;
#pragma unroll 1
    for (uint32_t j = first + threadIdx.x; j < first + len; j += blockDim.x) {
      const uint32_t s = ((j & ~lowmask) << 1) | (1u << k) | (j & lowmask);
      acc = pti_add(acc, load_pti(&Sw[s]));
    }
  }
  store_pti(&sh[threadIdx.x], acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t stride = blockDim.x >> 1; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      const PtI a = load_pti(&sh[threadIdx.x]);
      const PtI b2 = load_pti(&sh[threadIdx.x + stride]);
      store_pti(&sh[threadIdx.x], pti_add(a, b2));
    }
    __syncthreads();
  }
  // the host Horner pass works on the external 32-bit-limb form
  if (threadIdx.x == 0)
    store_jac(&partial[((size_t)w * (K + 2) + k) * gridDim.z + blockIdx.z], pti_to_ext(load_pti(&sh[0])));
}

int reduce_set_attributes(const char** failed) {
  if (hipFuncSetAttribute((const void*)reduce_tree_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess) {
    (void)hipGetLastError();
    *failed = "reduce_tree_kernel";
    return 1;
  }
  return 0;
}

void launch_reduce(hipStream_t st, const Plan& p, const PtI* buckets, PtI* S, PtI* T, Jacobian* partial) {
  hipLaunchKernelGGL(reduce_seg_kernel, dim3((unsigned)((p.total_segs + 63) / 64)), dim3(64), 0, st, buckets,
                     (uint32_t)p.total_segs, S, T);
  hipLaunchKernelGGL(reduce_tree_kernel, dim3(p.K + 2, p.W, p.tree_parts), dim3(p.tree_threads),
                     p.tree_threads * sizeof(PtI), st, (const PtI*)S, (const PtI*)T, p.nseg, p.K,
                     partial);
}

}  // namespace msm_amd
