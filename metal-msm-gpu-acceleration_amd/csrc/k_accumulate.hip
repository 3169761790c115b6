// Stage 3: bucket accumulation -- the dominant kernel -- and the combine pass for split buckets.
// See device_common.hip.h for the pipeline overview.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// One lane per work item.  A work item is (bucket b, chunk j): points [j*CH, min(size, (j+1)*CH)) of the
// bucket's slice of `sorted`.  The lane gathers each 80-byte internal-form affine base (software-prefetched
// one point ahead) and performs a mixed Jacobian+affine addition on 29-bit limbs (jaci_madd, 8M+3S).  Items arrive sorted by
// descending length (`order`), so the 64 lanes of a wave run the same number of iterations and the
// longest items start first.  Replaces kernel bucket_wise_accumulation (msm.h.metal:75-315), which
// splits pairs evenly over threads and merges bucket boundaries through threadgroup memory.
//
// A bucket made of one item is written straight to buckets[b]; a split bucket writes its partial sums
// to partials[window_base + item_start[b] + j] and combine_kernel adds them up.
__global__ void __launch_bounds__(64)
accumulate_kernel(const AffI* __restrict__ bases, const uint32_t* __restrict__ sorted,
                  const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                  const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,
                  const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,
                  uint32_t c, uint32_t CH, JacI* __restrict__ buckets, JacI* __restrict__ partials) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= counters->total_items) return;
  const uint2 it = order[slot];
  const uint32_t b = it.x, j = it.y;
  const uint32_t w = b >> c;
  const uint32_t size = bucket_size[b];
  const uint32_t lo = j * CH;
  const uint32_t cnt = min(size - lo, CH);
  const uint32_t* idx = sorted + (size_t)w * n + bucket_start[b] + lo;
  JacI acc = jaci_identity();
  uint32_t next_idx = idx[0];
#pragma unroll 1
  for (uint32_t i = 0; i < cnt; ++i) {
    // the gather of this point is issued here and first consumed after the Z1^2 squaring inside jaci_madd,
    // which hides most of its latency; only the next index is prefetched (one register, not a whole point)
    const AffI cur = load_affi(&bases[next_idx]);
    if (i + 1 < cnt) next_idx = idx[i + 1];
    if (jaci_is_identity(acc)) {
      if (!affi_is_identity(cur)) acc = jaci_from_affi(cur);
    } else {
      const JacI sum = jaci_madd(acc, cur);
      if (!affi_is_identity(cur)) acc = sum;
    }
  }
  if (size <= CH) {
    store_jaci(&buckets[b], acc);
  } else {
    store_jaci(&partials[(size_t)win_base[w] + item_start[b] + j], acc);
  }
}

// Buckets that were split into several items (only skewed digit distributions produce them: equal scalars,
// the narrow top window of small window sizes).  Each 64-lane workgroup takes 64 listed buckets at a time:
// a lane sums its bucket's partials serially when there are at most kSerialItems of them; buckets with more
// partials are then handled one by one by the whole workgroup (strided partial sums + 6-level LDS tree).
constexpr uint32_t kSerialItems = 8;

__global__ void __launch_bounds__(64)
combine_kernel(const uint32_t* __restrict__ multi_list, const PlanCounters* __restrict__ counters,
               const uint32_t* __restrict__ bucket_size, const uint32_t* __restrict__ item_start,
               const uint32_t* __restrict__ win_base, uint32_t c, uint32_t CH,
               const JacI* __restrict__ partials, JacI* __restrict__ buckets) {
  __shared__ JacI sh[64];
  __shared__ uint32_t big_b[64];
  __shared__ uint32_t big_n;
  const uint32_t count = counters->multi_count;
  for (uint32_t base = blockIdx.x * 64; base < count; base += gridDim.x * 64) {
    if (threadIdx.x == 0) big_n = 0;
    __syncthreads();
    const uint32_t m = base + threadIdx.x;
    if (m < count) {
      const uint32_t b = multi_list[m];
      const uint32_t nitems = (bucket_size[b] + CH - 1) / CH;
      if (nitems <= kSerialItems) {
        const JacI* src = partials + (size_t)win_base[b >> c] + item_start[b];
        JacI acc = load_jaci(&src[0]);
#pragma unroll 1
        for (uint32_t i = 1; i < nitems; ++i) acc = jaci_add(acc, load_jaci(&src[i]));
        store_jaci(&buckets[b], acc);
      } else {
        big_b[atomicAdd(&big_n, 1u)] = b;
      }
    }
    __syncthreads();
    const uint32_t nbig = big_n;
#pragma unroll 1
    for (uint32_t k = 0; k < nbig; ++k) {
      const uint32_t b = big_b[k];
      const uint32_t nitems = (bucket_size[b] + CH - 1) / CH;
      const JacI* src = partials + (size_t)win_base[b >> c] + item_start[b];
      JacI acc = jaci_identity();
#pragma unroll 1
      for (uint32_t i = threadIdx.x; i < nitems; i += 64) acc = jaci_add(acc, load_jaci(&src[i]));
      store_jaci(&sh[threadIdx.x], acc);
      __syncthreads();
#pragma unroll 1
      for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
        if (threadIdx.x < stride) {
          const JacI x = load_jaci(&sh[threadIdx.x]);
          const JacI y = load_jaci(&sh[threadIdx.x + stride]);
          store_jaci(&sh[threadIdx.x], jaci_add(x, y));
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) store_jaci(&buckets[b], load_jaci(&sh[0]));
      __syncthreads();
    }
    __syncthreads();
  }
}

void launch_accumulate(hipStream_t st, const Plan& p, const AffI* bases, const SortBuffers& b, JacI* buckets,
                       JacI* partials) {
  // empty buckets produce no work item: all-zero memory is the identity (Z = 0)
  (void)hipMemsetAsync(buckets, 0, p.total_buckets * sizeof(JacI), st);
  hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((p.max_items + 63) / 64)), dim3(64), 0, st, bases,
                     (const uint32_t*)b.sorted, (const uint32_t*)b.bucket_start, (const uint32_t*)b.bucket_size,
                     (const uint32_t*)b.item_start, (const uint32_t*)b.win_items, (const uint2*)b.order,
                     (const PlanCounters*)b.counters, p.n, p.c, p.CH, buckets, partials);
  hipLaunchKernelGGL(combine_kernel, dim3(256), dim3(64), 0, st, (const uint32_t*)b.multi_list,
                     (const PlanCounters*)b.counters, (const uint32_t*)b.bucket_size, (const uint32_t*)b.item_start,
                     (const uint32_t*)b.win_items, p.c, p.CH, (const JacI*)partials, buckets);
}

}  // namespace msm_amd
