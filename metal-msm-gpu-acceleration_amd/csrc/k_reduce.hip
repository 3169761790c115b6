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
reduce_seg_kernel(const PtI* __restrict__ buckets, const uint32_t* __restrict__ bucket_size, uint32_t total_segs,
                  PtI* __restrict__ S, PtI* __restrict__ T) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total_segs) return;
  // bucket_size != nullptr: an empty bucket (no work item ever wrote it) holds stale memory and counts as the
  // identity -- the bucket matrix is not cleared per MSM.  The size is re-read per slot (an L2 hit) rather than
  // kept as a mask: one more live register would push the kernel over the next allocation granule.
  const bool sized = bucket_size != nullptr;   // wave-uniform
  // Two accumulators plus an operand and the temporaries of an addition need ~195 VGPRs.  Both running values are
  // parked in LDS between additions, so every addition has two operands that die in it (like the tree kernels:
  // <= 160 VGPRs) and a wave fits into the registers two resident accumulate waves (2 x 176 of 512) leave free.
  __shared__ PtI park_sum[64];
  __shared__ PtI park_sos[64];
  store_pti(&park_sum[threadIdx.x], pti_identity());
  store_pti(&park_sos[threadIdx.x], pti_identity());
#pragma unroll 1
  for (int j = kSeg - 1; j >= 0; --j) {
    if (!sized || bucket_size[s * kSeg + j] != 0) {
      const PtI sum = pti_add(load_pti(&park_sum[threadIdx.x]), load_pti(&buckets[s * kSeg + j]));
      store_pti(&park_sum[threadIdx.x], sum);
    }
    asm volatile("" ::: "memory");   // keep the LDS reloads below the addition above (it is the point of parking)
    if (j != 0) {
      const PtI sos = pti_add(load_pti(&park_sos[threadIdx.x]), load_pti(&park_sum[threadIdx.x]));
      store_pti(&park_sos[threadIdx.x], sos);
    }
    asm volatile("" ::: "memory");
  }
  store_pti(&S[s], load_pti(&park_sum[threadIdx.x]));
  store_pti(&T[s], load_pti(&park_sos[threadIdx.x]));
}

// Stage 4b: tree sums, in two levels of ONE-WAVE workgroups (64 lanes, <= 160 VGPRs): a workgroup of several
// waves can only be placed on a CU where every SIMD has room for its waves at once, which next to a resident
// accumulate grid (2 x 176 VGPRs per SIMD) never happens before that grid drains; single waves slot in anywhere.
//   level 1  grid = (K + 2, W, parts): slice `part` of sum k of window w -> tree_tmp[w][k][part]
//            k == K : sum_s T[w][s]      k == K + 1 : sum_s S[w][s]      k < K : sum over segments s with bit k set
//            of S[w][s]   (K = lb - 3 bits of segment index)
//   level 2  grid = (K + 2, W): partial[w][k] = sum_part tree_tmp[w][k][part], converted to the external form
// The host then evaluates  W_w = partial[w][K] + partial[w][K+1] + 8 * sum_k 2^k partial[w][k]  inside one Horner
// pass over all bit positions (replaces sum_reduction_final msm.h.metal:463-562 and the doublings of
// final_accumulation.rs:19-39).
__device__ __forceinline__ PtI wave_tree_sum(PtI acc, PtI* sh) {
  store_pti(&sh[threadIdx.x], acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      const PtI a = load_pti(&sh[threadIdx.x]);
      const PtI b2 = load_pti(&sh[threadIdx.x + stride]);
      store_pti(&sh[threadIdx.x], pti_add(a, b2));
    }
    __syncthreads();
  }
  return load_pti(&sh[0]);
}

__global__ void __launch_bounds__(64)
reduce_tree_kernel(const PtI* __restrict__ S, const PtI* __restrict__ T, uint32_t nseg, uint32_t K,
                   PtI* __restrict__ tree_tmp) {
  __shared__ PtI sh[64];
  const uint32_t k = blockIdx.x, w = blockIdx.y, parts = gridDim.z;
  const PtI* Sw = S + (size_t)w * nseg;
  const PtI* Tw = T + (size_t)w * nseg;
  PtI acc = pti_identity();
  if (k >= K) {
    // slot i carries weight i + 1:  sum_i (i+1) X[i] = sum_s T[s] + sum_s S[s] + 8 sum_s s S[s]
    const PtI* src = (k == K) ? Tw : Sw;
    const uint32_t len = nseg / parts, first = blockIdx.z * len;
#pragma unroll 1
    for (uint32_t s = first + threadIdx.x; s < first + len; s += 64) acc = pti_add(acc, load_pti(&src[s]));
  } else {
    const uint32_t half = nseg >> 1;
    const uint32_t lowmask = (1u << k) - 1u;
    const uint32_t len = half / parts, first = blockIdx.z * len;
#pragma unroll 1
    for (uint32_t j = first + threadIdx.x; j < first + len; j += 64) {
      const uint32_t s = ((j & ~lowmask) << 1) | (1u << k) | (j & lowmask);
      acc = pti_add(acc, load_pti(&Sw[s]));
    }
  }
  const PtI sum = wave_tree_sum(acc, sh);
  if (threadIdx.x == 0) store_pti(&tree_tmp[((size_t)w * (K + 2) + k) * parts + blockIdx.z], sum);
}

__global__ void __launch_bounds__(64)
reduce_tree_final_kernel(const PtI* __restrict__ tree_tmp, uint32_t parts, uint32_t K, Jacobian* __restrict__ partial) {
  __shared__ PtI sh[64];
  const uint32_t k = blockIdx.x, w = blockIdx.y;
  const PtI* src = tree_tmp + ((size_t)w * (K + 2) + k) * parts;
  PtI acc = pti_identity();
#pragma unroll 1
  for (uint32_t i = threadIdx.x; i < parts; i += 64) acc = pti_add(acc, load_pti(&src[i]));
  const PtI sum = wave_tree_sum(acc, sh);
  // the host Horner pass works on the external 32-bit-limb form
  if (threadIdx.x == 0) store_jac(&partial[(size_t)w * (K + 2) + k], pti_to_ext(sum));
}

// Per-call pipeline (windows of <= 2^14 slots, at most 4 segments per thread): one 512-thread workgroup per
// (sum, window) does both levels at once and writes the external form directly -- one launch less and, measured,
// 1.5 % more MSM/s than the two-level form above, whose purpose is the long single window of the table pipeline.
// grid = (K + 2, W), dynamic LDS = blockDim.x * 144 bytes.
__global__ void __launch_bounds__(512)
reduce_tree_wide_kernel(const PtI* __restrict__ S, const PtI* __restrict__ T, uint32_t nseg,
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
  if (hipFuncSetAttribute((const void*)reduce_tree_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess) {
    (void)hipGetLastError();
    *failed = "reduce_tree_wide_kernel";
    return 1;
  }
  return 0;
}

void launch_reduce(hipStream_t st, const Plan& p, const PtI* buckets, const uint32_t* bucket_size, PtI* S, PtI* T,
                   PtI* tree_tmp, Jacobian* partial) {
  hipLaunchKernelGGL(reduce_seg_kernel, dim3((unsigned)((p.total_segs + 63) / 64)), dim3(64), 0, st, buckets,
                     bucket_size, (uint32_t)p.total_segs, S, T);
  if (p.tree_wide_threads) {
    hipLaunchKernelGGL(reduce_tree_wide_kernel, dim3(p.K + 2, p.W, 1), dim3(p.tree_wide_threads),
                       p.tree_wide_threads * sizeof(PtI), st, (const PtI*)S, (const PtI*)T, p.nseg, p.K, partial);
    return;
  }
  hipLaunchKernelGGL(reduce_tree_kernel, dim3(p.K + 2, p.W, p.tree_parts), dim3(64), 0, st, (const PtI*)S,
                     (const PtI*)T, p.nseg, p.K, tree_tmp);
  hipLaunchKernelGGL(reduce_tree_final_kernel, dim3(p.K + 2, p.W), dim3(64), 0, st, (const PtI*)tree_tmp,
                     p.tree_parts, p.K, partial);
}

}  // namespace msm_amd
