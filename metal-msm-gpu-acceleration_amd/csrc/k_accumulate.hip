// Stage 3: bucket accumulation -- the dominant kernel -- and the combine pass for split buckets.
// See device_common.hip.h for the pipeline overview.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// One lane per work item.  A work item is (bucket b, chunk j): points [j*CH, min(size, (j+1)*CH)) of the
// bucket's slice of `sorted`.  The lane gathers each 64-byte affine base (software-prefetched one point
// ahead) and performs a mixed Jacobian+affine addition (madd-2007-bl, 7M+4S).  Items arrive sorted by
// descending length (`order`), so the 64 lanes of a wave run the same number of iterations and the
// longest items start first.  Replaces kernel bucket_wise_accumulation (msm.h.metal:75-315), which
// splits pairs evenly over threads and merges bucket boundaries through threadgroup memory.
//
// A bucket made of one item is written straight to buckets[b]; a split bucket writes its partial sums
// to partials[window_base + item_start[b] + j] and combine_kernel adds them up.
__global__ void __launch_bounds__(64)
accumulate_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ sorted,
                  const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                  const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,
                  const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,
                  uint32_t c, uint32_t CH, Jacobian* __restrict__ buckets, Jacobian* __restrict__ partials) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= counters->total_items) return;
  const uint2 it = order[slot];
  const uint32_t b = it.x, j = it.y;
  const uint32_t w = b >> c;
  const uint32_t size = bucket_size[b];
  const uint32_t lo = j * CH;
  const uint32_t cnt = min(size - lo, CH);
  const uint32_t* idx = sorted + (size_t)w * n + bucket_start[b] + lo;
  Jacobian acc = jac_identity();
  Affine nxt = load_affine(&bases[idx[0]]);
#pragma unroll 1
  for (uint32_t i = 0; i < cnt; ++i) {
    const Affine cur = nxt;
    if (i + 1 < cnt) nxt = load_affine(&bases[idx[i + 1]]);
    if (!affine_is_identity(cur)) acc = jac_madd(acc, cur);
  }
  if (size <= CH) {
    store_jac(&buckets[b], acc);
  } else {
    store_jac(&partials[(size_t)win_base[w] + item_start[b] + j], acc);
  }
}

// Buckets that were split into several items: one 64-lane workgroup per listed bucket (grid-stride over
// multi_list), lanes sum the partials strided, then a 6-level LDS tree.  Only runs into work for skewed
// digit distributions (equal scalars, the narrow top window of small window sizes).
__global__ void __launch_bounds__(64)
combine_kernel(const uint32_t* __restrict__ multi_list, const PlanCounters* __restrict__ counters,
               const uint32_t* __restrict__ bucket_size, const uint32_t* __restrict__ item_start,
               const uint32_t* __restrict__ win_base, uint32_t c, uint32_t CH,
               const Jacobian* __restrict__ partials, Jacobian* __restrict__ buckets) {
  __shared__ Jacobian sh[64];
  const uint32_t count = counters->multi_count;
  for (uint32_t m = blockIdx.x; m < count; m += gridDim.x) {
    const uint32_t b = multi_list[m];
    const uint32_t w = b >> c;
    const uint32_t nitems = (bucket_size[b] + CH - 1) / CH;
    const Jacobian* src = partials + (size_t)win_base[w] + item_start[b];
    Jacobian acc = jac_identity();
#pragma unroll 1
    for (uint32_t i = threadIdx.x; i < nitems; i += 64) acc = jac_add(acc, load_jac(&src[i]));
    store_jac(&sh[threadIdx.x], acc);
    __syncthreads();
#pragma unroll 1
    for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
      if (threadIdx.x < stride) {
        const Jacobian a = load_jac(&sh[threadIdx.x]);
        const Jacobian b2 = load_jac(&sh[threadIdx.x + stride]);
        store_jac(&sh[threadIdx.x], jac_add(a, b2));
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) store_jac(&buckets[b], load_jac(&sh[0]));
    __syncthreads();
  }
}

void launch_accumulate(hipStream_t st, const Plan& p, const Affine* bases, const SortBuffers& b, Jacobian* buckets,
                       Jacobian* partials) {
  // empty buckets produce no work item: all-zero memory is the identity (Z = 0)
  (void)hipMemsetAsync(buckets, 0, p.total_buckets * sizeof(Jacobian), st);
  hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((p.max_items + 63) / 64)), dim3(64), 0, st, bases,
                     (const uint32_t*)b.sorted, (const uint32_t*)b.bucket_start, (const uint32_t*)b.bucket_size,
                     (const uint32_t*)b.item_start, (const uint32_t*)b.win_items, (const uint2*)b.order,
                     (const PlanCounters*)b.counters, p.n, p.c, p.CH, buckets, partials);
  hipLaunchKernelGGL(combine_kernel, dim3(1024), dim3(64), 0, st, (const uint32_t*)b.multi_list,
                     (const PlanCounters*)b.counters, (const uint32_t*)b.bucket_size, (const uint32_t*)b.item_start,
                     (const uint32_t*)b.win_items, p.c, p.CH, (const Jacobian*)partials, buckets);
}

}  // namespace msm_amd
