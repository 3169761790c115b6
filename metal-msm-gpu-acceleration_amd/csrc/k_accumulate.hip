// Stage 3: bucket accumulation -- the dominant kernel -- and the combine pass for split buckets.
// See device_common.hip.h for the pipeline overview.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// One lane per work item.  A work item is (bucket b, chunk j): points [j*CH, min(size, (j+1)*CH)) of the
// bucket's slice of `sorted`.  The lane gathers each 64-byte packed affine base and performs a mixed XYZZ+affine addition on 29-bit limbs (pti_madd, madd-2008-s, 8M+2S).  Items arrive sorted by
// descending length (`order`), so the 64 lanes of a wave run the same number of iterations and the
// longest items start first.  Replaces kernel bucket_wise_accumulation (msm.h.metal:75-315), which
// splits pairs evenly over threads and merges bucket boundaries through threadgroup memory.
//
// A bucket made of one item is written straight to buckets[b]; a split bucket writes its partial sums
// to partials[window_base + item_start[b] + j] and combine_kernel adds them up.
// LOW_OCC = true pins a high VGPR so that the kernel runs at 2 instead of 3 waves per SIMD (about 3 % slower
// alone): with several streams in flight this leaves register file and wave slots for the sort / reduce
// kernels of the neighbouring instance, which otherwise cannot be placed until the whole accumulate grid has
// drained (measured: a 1024-thread plan_kernel workgroup waited 1.4 ms behind 3-wave accumulate waves).
// PREFETCH: gather the packed record of point i + 1 while point i is added (16 more live registers).
// PIN: touch v175 so that the kernel allocates 176 VGPRs and runs at two waves per SIMD whatever it needs itself.
template <bool PREFETCH, bool PIN, int WHATIF = 0>
__device__ __forceinline__ void
accumulate_item(const uint32_t slot, const AffPacked* __restrict__ bases, const uint32_t* __restrict__ sorted,
                const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,
                const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,
                uint32_t lb, uint32_t CH, PtI* __restrict__ buckets, PtI* __restrict__ partials) {
  constexpr bool LOW_OCC = PREFETCH;
  // WHATIF (timing experiments, -DMSM_AMD_EXPERIMENTS builds only): 1 = gathers without arithmetic, 2 = arithmetic
  // on a 1 MB slice of the bases (every gather hits the L2)
  constexpr uint32_t IDX_MASK = WHATIF == 2 ? 0x3FFFu : 0x7FFFFFFFu;
  if (PIN) asm volatile("v_mov_b32 v175, 0" ::: "v175");
  if (slot >= counters->total_items) return;
  const uint2 it = order[slot];
  const uint32_t b = it.x, j = it.y;
  const uint32_t w = b >> lb;
  const uint32_t size = bucket_size[b];
  const uint32_t lo = j * CH;
  const uint32_t cnt = min(size - lo, CH);
  const uint32_t* idx = sorted + (size_t)w * n + bucket_start[b] + lo;
  PtI acc = pti_identity();
  // What acc holds is tracked in a lane register instead of being read off its limbs every trip:
  //   kEmpty  the identity (nothing added yet, or a sum that cancelled)
  //   kOne    exactly one base, still affine (ZZ = ZZZ = 1): the next addition is affine + affine, 4M + 2S
  //   kMany   a general XYZZ point: mixed additions, 8M + 2S
  enum : uint32_t { kEmpty = 0, kOne = 1, kMany = 2 };
  uint32_t state = kEmpty;
  // Software pipeline.  LOW_OCC (2 waves/SIMD) has ~20 spare VGPRs: the packed 64-byte record of point i + 1 is
  // gathered while point i is added, so a whole mixed addition (~5 us) hides the gather.  The 3-wave variant has
  // no registers to spare: it issues the gather at the top of the iteration and first consumes it after the
  // Z1^2 squaring inside pti_madd, prefetching only the next index.
  uint32_t cur_idx = idx[0];
  uint32_t next_idx = cnt > 1 ? idx[1] : 0u;
  AffPacked pre;
  if (LOW_OCC) {
    pre.x = load_u256(&bases[cur_idx & IDX_MASK].x);
    pre.y = load_u256(&bases[cur_idx & IDX_MASK].y);
  }
#pragma unroll 1
  for (uint32_t i = 0; i < cnt; ++i) {
    AffPacked rec;
    if (LOW_OCC) {
      rec = pre;
      if (i + 1 < cnt) {
        pre.x = load_u256(&bases[next_idx & IDX_MASK].x);
        pre.y = load_u256(&bases[next_idx & IDX_MASK].y);
      }
    } else {
      rec.x = load_u256(&bases[cur_idx & IDX_MASK].x);
      rec.y = load_u256(&bases[cur_idx & IDX_MASK].y);
    }
    const bool negate = (cur_idx >> 31) != 0;   // negative digit: add -P (signed digits, see digits_kernel)
    cur_idx = next_idx;
    if (i + 2 < cnt) next_idx = idx[i + 2];
    if (WHATIF == 1) {
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        acc.x.l[l] ^= rec.x.v[l];
        acc.y.l[l] ^= rec.y.v[l];
      }
      continue;
    }
    if (affpacked_is_identity(rec)) continue;   // an identity base adds nothing (one word tells: affi_pack)
    AffI cur = affi_unpack_finite(rec);
    {
      // -y without the carry round (limbs < 2^30.5): y only ever multiplies the normalised ZZZ1, or enters the
      // lifted subtraction of pti_mmadd (bounds: tools/fq29_bounds.py)
      const fe29 ny = Fq29::neg_wide(cur.y);
#pragma unroll
      for (int l = 0; l < 9; ++l) cur.y.l[l] = negate ? ny.l[l] : cur.y.l[l];
    }
    if (state == kMany) {
      MSM_ISA_MARK("begin mixed_addition");
      bool vanished = false;
      acc = pti_madd(acc, cur, vanished);
      if (vanished) state = kEmpty;
      MSM_ISA_MARK("end");
    } else if (state == kOne) {   // second point of the item (wave-uniform in practice): affine + affine
      MSM_ISA_MARK("begin affine_start");
      bool vanished = false;
      acc = pti_mmadd(acc.x, acc.y, cur, vanished);
      state = vanished ? (uint32_t)kEmpty : (uint32_t)kMany;
      MSM_ISA_MARK("end");
    } else {
      acc = pti_from_affi(cur);
      state = kOne;
    }
  }
  if (size <= CH) {
    store_pti(&buckets[b], acc);
  } else {
    store_pti(&partials[(size_t)win_base[w] + item_start[b] + j], acc);
  }
}

template <bool PREFETCH, bool PIN, int WHATIF = 0>
__device__ __forceinline__ void
accumulate_body(const AffPacked* __restrict__ bases, const uint32_t* __restrict__ sorted,
                const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,
                const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,
                uint32_t lb, uint32_t CH, PtI* __restrict__ buckets, PtI* __restrict__ partials) {
  accumulate_item<PREFETCH, PIN, WHATIF>(blockIdx.x * blockDim.x + threadIdx.x, bases, sorted, bucket_start, bucket_size,
                                         item_start, win_base, order, counters, n, lb, CH, buckets, partials);
}

#define MSM_ACC_PARAMS const AffPacked* __restrict__ bases, const uint32_t* __restrict__ sorted,                       \
                       const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,           \
                       const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,                \
                       const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,        \
                       uint32_t lb, uint32_t CH, PtI* __restrict__ buckets, PtI* __restrict__ partials
#define MSM_ACC_FWD bases, sorted, bucket_start, bucket_size, item_start, win_base, order, counters, n, lb, CH, buckets, partials
// accumulate_kernel<true>: two waves per SIMD (pinned), prefetch.  accumulate_kernel<false>: the same body without
// prefetch and pin (a build for A/B runs; it needs 173 VGPRs, so it is a two-wave kernel too).
template <bool LOW_OCC>
__global__ void __launch_bounds__(64) accumulate_kernel(MSM_ACC_PARAMS) {
  accumulate_body<LOW_OCC, LOW_OCC>(MSM_ACC_FWD);
}
// Other builds of this kernel -- three waves per SIMD, the register-lean product-scanning form at four, the compiler's
// column form at four with Y / ZZ / ZZZ parked in LDS, the hand-allocated five-wave statement of
// tools/gen_accumulate_asm.py, and the what-if timing kernels -- were built, checked bit for bit and measured in round
// 4 (profiles/r04_ab_accumulate_occupancy.txt); none is faster than the kernel above, so they only exist in
// -DMSM_AMD_EXPERIMENTS builds (MSM_AMD_ACC_VARIANT selects one there).
#if defined(MSM_AMD_EXPERIMENTS)
#include "experiments/k_accumulate_variants.inc"
#endif

// Buckets that were split into several items (only skewed digit distributions produce them: equal scalars,
// the narrow top window of small window sizes).  Two passes over multi_list:
//   combine_small_kernel  one lane per listed bucket; sums up to kSerialItems partials serially, defers
//                         larger buckets to big_list
//   combine_big_kernel    one 64-lane workgroup per deferred bucket (grid-stride): strided partial sums +
//                         6-level LDS tree
constexpr uint32_t kSerialItems = 8;

__global__ void __launch_bounds__(64)
combine_small_kernel(const uint32_t* __restrict__ multi_list, PlanCounters* __restrict__ counters,
                     const uint32_t* __restrict__ bucket_size, const uint32_t* __restrict__ item_start,
                     const uint32_t* __restrict__ win_base, uint32_t lb, uint32_t CH,
                     const PtI* __restrict__ partials, PtI* __restrict__ buckets, uint32_t* __restrict__ big_list) {
  // the grid covers every possible split bucket (one lane each, launch_combine): no grid-stride loop, fewer live
  // registers (164 VGPRs)
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= counters->multi_count) return;
  const uint32_t b = multi_list[m];
  const uint32_t nitems = (bucket_size[b] + CH - 1) / CH;
  if (nitems > kSerialItems) {
    big_list[atomicAdd(&counters->pad[0], 1u)] = b;   // pad[0] = number of deferred buckets
    return;
  }
  const PtI* src = partials + (size_t)win_base[b >> lb] + item_start[b];
  const PtI* const end = src + nitems;
  PtI acc = load_pti(src);
#pragma unroll 1
  for (++src; src != end; ++src) acc = pti_add(acc, load_pti(src));
  store_pti(&buckets[multi_list[m]], acc);   // b is re-read: one live register less across the loop
}

__global__ void __launch_bounds__(64)
combine_big_kernel(const uint32_t* __restrict__ big_list, const PlanCounters* __restrict__ counters,
                   const uint32_t* __restrict__ bucket_size, const uint32_t* __restrict__ item_start,
                   const uint32_t* __restrict__ win_base, uint32_t lb, uint32_t CH,
                   const PtI* __restrict__ partials, PtI* __restrict__ buckets) {
  __shared__ PtI sh[64];
  const uint32_t count = counters->pad[0];
  for (uint32_t m = blockIdx.x; m < count; m += gridDim.x) {
    const uint32_t b = big_list[m];
    const uint32_t nitems = (bucket_size[b] + CH - 1) / CH;
    const PtI* src = partials + (size_t)win_base[b >> lb] + item_start[b];
    PtI acc = pti_identity();
#pragma unroll 1
    for (uint32_t i = threadIdx.x; i < nitems; i += 64) acc = pti_add(acc, load_pti(&src[i]));
    store_pti(&sh[threadIdx.x], acc);
    __syncthreads();
#pragma unroll 1
    for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
      if (threadIdx.x < stride) {
        const PtI x = load_pti(&sh[threadIdx.x]);
        const PtI y = load_pti(&sh[threadIdx.x + stride]);
        store_pti(&sh[threadIdx.x], pti_add(x, y));
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) store_pti(&buckets[b], load_pti(&sh[0]));
    __syncthreads();
  }
}

// variant: 0 = three waves per SIMD (no prefetch), 1 = two waves per SIMD (register pin, prefetch), 2 = register-lean
// (four waves per SIMD).  lds_bytes > 0 caps the resident workgroups per CU through the LDS allocation (160 KiB per
// CU: 13 KiB per 64-lane workgroup = 12 waves per CU = 3 per SIMD), leaving register file for the other streams.
void launch_accumulate(hipStream_t st, const Plan& p, const AffPacked* bases, const SortBuffers& b, PtI* buckets,
                       PtI* partials, int variant, uint32_t lds_bytes, hipEvent_t before_kernel, hipEvent_t after_kernel) {
  if (before_kernel) (void)hipEventRecord(before_kernel, st);
  const dim3 grid((unsigned)((p.max_items + 63) / 64)), block(64);
#define MSM_ACC_ARGS bases, (const uint32_t*)b.sorted, (const uint32_t*)b.bucket_start, (const uint32_t*)b.bucket_size, \
                     (const uint32_t*)b.item_start, (const uint32_t*)b.win_items, (const uint2*)b.order,                \
                     (const PlanCounters*)b.counters, p.n, p.lb, p.CH, buckets, partials
#if defined(MSM_AMD_EXPERIMENTS)
  if (variant == 4) {
    // the hand-allocated kernel (5 waves per SIMD) + the redo pass over the items it flagged (none for ordinary inputs)
    uint32_t* redo_list = b.redo_list;
    uint32_t* redo_count = &b.counters->pad[1];
    hipLaunchKernelGGL(accumulate_kernel_asm, grid, block, 0, st, MSM_ACC_ARGS, redo_list, redo_count);
    hipLaunchKernelGGL(accumulate_redo_kernel, dim3(1024), block, 0, st, MSM_ACC_ARGS, (const uint32_t*)redo_list,
                       (const uint32_t*)redo_count);
  } else if (variant == 2) {
    hipLaunchKernelGGL(accumulate_kernel_lean, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else if (variant == 3) {
    hipLaunchKernelGGL(accumulate_kernel_w3, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else if (variant == 5) {
    hipLaunchKernelGGL(accumulate_kernel_park, grid, block, 0, st, MSM_ACC_ARGS);
  } else if (variant == 10) {
    hipLaunchKernelGGL(accumulate_whatif_gathers, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else if (variant == 11) {
    hipLaunchKernelGGL(accumulate_whatif_math, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else if (variant == 12) {
    hipLaunchKernelGGL(accumulate_whatif_math_w3, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else
#endif
  if (variant == 1) {
    hipLaunchKernelGGL(accumulate_kernel<true>, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  } else {
    hipLaunchKernelGGL(accumulate_kernel<false>, grid, block, lds_bytes, st, MSM_ACC_ARGS);
  }
#undef MSM_ACC_ARGS
  if (after_kernel) (void)hipEventRecord(after_kernel, st);
}

// Sums the partial results of split buckets (no-op launches when nothing was split).
void launch_combine(hipStream_t st, const Plan& p, const SortBuffers& b, PtI* buckets, PtI* partials) {
  // multi_list doubles as big_list storage: its second half (entries max_items/2 ..) is free because a split
  // bucket accounts for at least two items
  uint32_t* big_list = b.multi_list + p.max_items / 2 + 1;
  // one lane per possibly-split bucket: at most one split bucket per two items
  hipLaunchKernelGGL(combine_small_kernel, dim3((unsigned)((p.max_items / 2 + 63) / 64)), dim3(64), 0, st, (const uint32_t*)b.multi_list, b.counters,
                     (const uint32_t*)b.bucket_size, (const uint32_t*)b.item_start, (const uint32_t*)b.win_items, p.lb,
                     p.CH, (const PtI*)partials, buckets, big_list);
  hipLaunchKernelGGL(combine_big_kernel, dim3(512), dim3(64), 0, st, (const uint32_t*)big_list,
                     (const PlanCounters*)b.counters, (const uint32_t*)b.bucket_size, (const uint32_t*)b.item_start,
                     (const uint32_t*)b.win_items, p.lb, p.CH, (const PtI*)partials, buckets);
}

}  // namespace msm_amd
