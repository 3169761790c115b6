// Stage 3: bucket accumulation -- the dominant kernel -- and the combine pass for split buckets.
// See device_common.hip.h for the pipeline overview.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// One lane per work item.  A work item is (bucket b, chunk j): points [j*CH, min(size, (j+1)*CH)) of the
// bucket's slice of `sorted`.  The lane gathers each 64-byte packed affine base and performs a mixed XYZZ+affine addition
// on 29-bit limbs (madd-2008-s, 8M+2S).  Items arrive sorted by descending length (`order`), so the 64 lanes of a wave
// run the same number of iterations and the longest items start first.  Replaces kernel bucket_wise_accumulation
// (msm.h.metal:75-315), which splits pairs evenly over threads and merges bucket boundaries through threadgroup memory.
//
// A bucket made of one item is written straight to buckets[b]; a split bucket writes its partial sums
// to partials[window_base + item_start[b] + j] and combine_kernel adds them up.
//
// What the accumulator holds is tracked in a lane register instead of being read off its limbs every trip:
//   kEmpty  the identity (nothing added yet, or a sum that cancelled)
//   kOne    exactly one base, still affine (ZZ = ZZZ = 1): the next addition is affine + affine, 4M + 2S
//   kMany   a general XYZZ point: mixed additions, 8M + 2S
// The loop is shaped for the register allocator (round 4, late; profiles/r04_accumulate_loop_shape.txt):
//   * phase A handles the first points of the item (kEmpty / kOne), phase B is a loop that holds nothing but the mixed
//     addition, so the accumulator stays in ONE set of registers (the three-way merge of a single loop cost 18 v_mov
//     per trip);
//   * software pipeline: the packed record of point i + 1 is gathered while point i is added (a whole mixed addition,
//     ~7 us at two waves per SIMD, hides the gather); the record is unpacked BEFORE the next gather is issued into the
//     SAME registers (loading first and unpacking later cost 2 x 8 v_mov_b64 per trip);
//   * the addition runs as head + tail (bn254_ec29.hip.h) with operands pinned once per basic block: the base dies in
//     the head, and the exceptional case q == p, which needs it again, gathers it again.
// PIN: touch v175 so that the kernel allocates at least 176 VGPRs and runs at TWO waves per SIMD whatever it needs itself
// (it needs 180): with several streams in flight this leaves register file and wave slots for the sort / reduce kernels
// of the neighbouring instances, which otherwise cannot be placed until the whole accumulate grid has drained (measured
// in round 1: a 1024-thread plan_kernel workgroup waited 1.4 ms behind 3-wave accumulate waves).
template <bool PIN>
__device__ __forceinline__ void
accumulate_item(const uint32_t slot, const AffPacked* __restrict__ bases, const uint32_t* __restrict__ sorted,
                   const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                   const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,
                   const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,
                   uint32_t lb, uint32_t CH, PtI* __restrict__ buckets, PtI* __restrict__ partials) {
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
  enum : uint32_t { kEmpty = 0, kOne = 1, kMany = 2 };
  uint32_t state = kEmpty;
  uint32_t cur_idx = idx[0];
  uint32_t next_idx = cnt > 1 ? idx[1] : 0u;
  AffPacked pre;
  pre.x = load_u256(&bases[cur_idx & 0x7FFFFFFFu].x);
  pre.y = load_u256(&bases[cur_idx & 0x7FFFFFFFu].y);
  uint32_t i = 0;
  // take(): point i in register form (negated for a negative digit), whether it is the identity; starts the gather of
  // point i + 1 into the registers it has just emptied and the load of index i + 2
  // register form of a gathered record: unpacked, y negated for a negative digit (-y without the carry round, limbs
  // < 2^30.5: y only ever multiplies the normalised ZZZ1 or enters the lifted subtraction of pti_mmadd; bounds:
  // tools/fq29_bounds.py)
  auto signed_point = [](const AffPacked& rec, uint32_t entry) {
    AffI r = affi_unpack_finite(rec);
    const bool negate = (entry >> 31) != 0;
    const fe29 ny = Fq29::neg_wide(r.y);
#pragma unroll
    for (int l = 0; l < 9; ++l) r.y.l[l] = negate ? ny.l[l] : r.y.l[l];
    return r;
  };
  // the exceptional case q == p (a doubling) needs q after its registers have been given away: gathered again
  auto regather = [&](uint32_t entry) {
    AffPacked rec;
    rec.x = load_u256(&bases[entry & 0x7FFFFFFFu].x);
    rec.y = load_u256(&bases[entry & 0x7FFFFFFFu].y);
    return signed_point(rec, entry);
  };
  auto take = [&](AffI& cur) -> bool {
    const bool ident = affpacked_is_identity(pre);
    cur = signed_point(pre, cur_idx);
    // the unpacking above is done before the registers of `pre` are loaded again (the statement ties the 18 limbs)
    asm volatile("" : "+v"(cur.x.l[0]), "+v"(cur.x.l[1]), "+v"(cur.x.l[2]), "+v"(cur.x.l[3]), "+v"(cur.x.l[4]),
                      "+v"(cur.x.l[5]), "+v"(cur.x.l[6]), "+v"(cur.x.l[7]), "+v"(cur.x.l[8]), "+v"(cur.y.l[0]),
                      "+v"(cur.y.l[1]), "+v"(cur.y.l[2]), "+v"(cur.y.l[3]), "+v"(cur.y.l[4]), "+v"(cur.y.l[5]),
                      "+v"(cur.y.l[6]), "+v"(cur.y.l[7]), "+v"(cur.y.l[8]) :: "memory");
    cur_idx = next_idx;
    if (i + 1 < cnt) {
      pre.x = load_u256(&bases[cur_idx & 0x7FFFFFFFu].x);
      pre.y = load_u256(&bases[cur_idx & 0x7FFFFFFFu].y);
    }
    if (i + 2 < cnt) next_idx = idx[i + 2];
    ++i;
    return ident;
  };
  while (i < cnt) {
    // phase A: until the accumulator is a general point
#pragma unroll 1
    while (i < cnt && state != kMany) {
      AffI cur;
      const uint32_t this_idx = cur_idx;
      if (take(cur)) continue;
      if (state == kOne) {
        MSM_ISA_MARK("begin affine_start");
        bool vanished = false;
        fe29 P, R;
        pti_mmadd_head(acc.x, acc.y, cur.x, cur.y, P, R);
        acc = pti_mmadd_tail(acc.x, acc.y, P, R, [&]() { return regather(this_idx); }, vanished);
        state = vanished ? (uint32_t)kEmpty : (uint32_t)kMany;
        MSM_ISA_MARK("end");
      } else {
        acc = pti_from_affi(cur);
        state = kOne;
      }
    }
    // phase B: mixed additions only
#pragma unroll 1
    while (i < cnt) {
      AffI cur;
      const uint32_t this_idx = cur_idx;
      if (take(cur)) continue;
      MSM_ISA_MARK("begin mixed_addition");
      bool vanished = false;
      fe29 U2, S2;
      pti_madd_head(acc, cur.x, cur.y, U2, S2);   // cur dies here: its pins cost no copies
      acc = pti_madd_tail(acc, U2, S2, [&]() { return regather(this_idx); }, vanished);
      MSM_ISA_MARK("end");
      if (vanished) {
        state = kEmpty;
        break;
      }
    }
  }
  if (size <= CH) {
    store_pti(&buckets[b], acc);
  } else {
    store_pti(&partials[(size_t)win_base[w] + item_start[b] + j], acc);
  }
}

#define MSM_ACC_PARAMS const AffPacked* __restrict__ bases, const uint32_t* __restrict__ sorted,                       \
                       const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,           \
                       const uint32_t* __restrict__ item_start, const uint32_t* __restrict__ win_base,                \
                       const uint2* __restrict__ order, const PlanCounters* __restrict__ counters, uint32_t n,        \
                       uint32_t lb, uint32_t CH, PtI* __restrict__ buckets, PtI* __restrict__ partials
#define MSM_ACC_FWD bases, sorted, bucket_start, bucket_size, item_start, win_base, order, counters, n, lb, CH, buckets, partials
// accumulate_kernel<true>: the shipped kernel (two waves per SIMD, pinned).  accumulate_kernel<false>: the same body
// without the pin (MSM_AMD_LOW_OCC=0; it needs 180 VGPRs, so it is a two-wave kernel too).
template <bool PIN>
__global__ void __launch_bounds__(64) accumulate_kernel(MSM_ACC_PARAMS) {
  accumulate_item<PIN>(blockIdx.x * blockDim.x + threadIdx.x, MSM_ACC_FWD);
}
// Other builds of this kernel -- the single-loop shape of rounds 1-4, three waves per SIMD, the register-lean
// product-scanning form at four, the compiler's column form at four with Y / ZZ / ZZZ parked in LDS, the hand-allocated
// five-wave statement of tools/gen_accumulate_asm.py, the kernel on 128-byte wide records, and the what-if timing
// kernels -- were built, checked bit for bit and measured in round 4 (profiles/r04_ab_accumulate_*.txt); none is faster
// inside the pipeline than the kernel above, so they only exist in -DMSM_AMD_EXPERIMENTS builds (MSM_AMD_ACC_VARIANT
// selects one there).
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
void launch_accumulate(hipStream_t st, const Plan& p, const void* bases_any, int wide, const SortBuffers& b, PtI* buckets,
                       PtI* partials, int variant, uint32_t lds_bytes, hipEvent_t before_kernel, hipEvent_t after_kernel) {
  if (before_kernel) (void)hipEventRecord(before_kernel, st);
  const dim3 grid((unsigned)((p.max_items + 63) / 64)), block(64);
  const AffPacked* bases = (const AffPacked*)bases_any;
#if defined(MSM_AMD_EXPERIMENTS)
  if (wide) {
    hipLaunchKernelGGL(accumulate_kernel_wide, grid, block, lds_bytes, st, (const AffWide*)bases_any,
                       (const uint32_t*)b.sorted, (const uint32_t*)b.bucket_start, (const uint32_t*)b.bucket_size,
                       (const uint32_t*)b.item_start, (const uint32_t*)b.win_items, (const uint2*)b.order,
                       (const PlanCounters*)b.counters, p.n, p.lb, p.CH, buckets, partials);
    if (after_kernel) (void)hipEventRecord(after_kernel, st);
    return;
  }
#else
  (void)wide;
#endif
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
  } else if (variant == 7) {
    hipLaunchKernelGGL(accumulate_kernel_r4, grid, block, lds_bytes, st, MSM_ACC_ARGS);
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
