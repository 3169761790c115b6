// Launch wrappers exported by the kernel translation units (k_*.hip) to the host driver (msm_host.hip).
// Every wrapper only enqueues work on `st`; none synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "bn254_ec29.hip.h"

namespace msm_amd {

constexpr int kSegLog = 3;   // a window has at least 2^3 bucket slots (c = 3 is padded)

// Geometry of one MSM (see make_plan in msm_host.hip).
struct Plan {
  uint32_t n, c, W;           // sorted entries per window, window bits, windows that own buckets
                              // (per-call pipeline: n points, W = floor(254/c) + 1 signed-digit windows;
                              //  precomputed tables: n = W_digits * n_scalars entries in ONE window)
  uint32_t n_scalars, W_digits;   // what digits_kernel sees: scalars and digit windows per scalar
  bool wide_digits;           // digits are u32 (c up to 24) instead of u16 (c <= 15)
  uint32_t lb, nb;            // bucket slots per window nb = 2^lb = max(2^(c-1), 8); slot i holds |digit| = i + 1
  uint32_t Q, chunk;          // sort: chunks per window, points per chunk
  uint32_t hb, mb, fb;        // sort: coarse / middle / fine bits of the slot (hb + mb + fb = lb; mb = 0: two passes)
  uint32_t Q2;                // sort: chunks per coarse region in the middle pass
  uint32_t tile_threads;      // workgroup size of the tiled scatter kernels
  uint32_t tiled;             // sort: coarse / middle scatter stage a tile in LDS and write whole runs (tile_scatter)
  uint32_t ballot;            // sort ranking: bit 0 coarse / middle passes, bit 1 pass 2 use the wave multisplit
  uint32_t front_threads;     // workgroup size of the sort / planning kernels (256..1024)
  uint32_t CH;                // accumulate: max points per work item (bucket chunk)
  uint32_t red_L, red_H;      // reduce: column / row bits of the slot index (L = ceil(lb / 2), H = lb - L)
  uint32_t rb_threads;        // reduce: threads per bit-subset sum (0 = by the sum's length; 64 for pipelined instances)
  uint32_t red_group;         // reduce: additions per lane and level of the row / column sums (kReduceGroupMin..16)
  size_t total_buckets, total_segs, partial_count, max_items;
};

// Device-side bookkeeping words written by the planning kernels.
struct PlanCounters {
  uint32_t total_items;       // number of accumulate work items
  uint32_t multi_count;       // number of buckets split into more than one item
  uint32_t pad[2];
};

struct SortBuffers {
  void* digits;               // [W_digits][n_scalars] u16 or u32 (Plan::wide_digits) = [W][n] entries
  uint32_t* coarse_cnt;       // [W][Q][2^hb]  per-chunk region counts, then write positions
  uint32_t* region_start;     // [W][2^hb + 1]
  uint32_t* tmp_idx;          // [W][n]        pass-1 output: index | sign << 31, grouped by coarse region
  uint16_t* tmp_fine;         // [W][n]        pass-1 output: fine digit
  uint32_t* tmp_idx2;         // [W][n]        middle-pass output (three-level sort only)
  uint16_t* tmp_fine2;        // [W][n]
  uint32_t* mid_cnt;          // [W * 2^hb][Q2][2^mb]
  uint32_t* region_start2;    // [W][2^(hb+mb) + 1]
  uint32_t* bucket_size;      // [W][nb]
  uint32_t* bucket_start;     // [W][nb]   offset inside the window's slice of `sorted`
  uint32_t* item_start;       // [W][nb]   first item id of the bucket inside its window
  uint32_t* win_items;        // [W]       items per window, then exclusive prefix (window base)
  uint2* tile_sums;           // [W][tiles] (points, items) per planning tile (windows of more than 16384 slots)
  uint32_t* size_bins;        // [CH + 1][ceil(total_buckets / front_threads)] item-size counts, then positions
  uint32_t* sorted;           // [W][n]
  uint2* order;               // [max_items] (bucket, chunk) by descending size
  uint32_t* multi_list;       // [max_items] buckets with more than one item
  uint32_t* redo_list;        // [max_items] work items accumulate_kernel_asm leaves to accumulate_redo_kernel
  PlanCounters* counters;
};

// k_sort.hip
int sort_set_attributes(const char** failed);
void launch_build_tables(hipStream_t st, const Affine* in, uint32_t n, uint32_t c, uint32_t W, AffPacked* tables);
void launch_digits(hipStream_t st, const Plan& p, const u256* scalars, int scalars_mont, void* digits);
void launch_sort(hipStream_t st, const Plan& p, const SortBuffers& b);
void launch_be32_to_le(hipStream_t st, const uint32_t* in, size_t words, uint32_t* out);
void launch_ark_affine_to_affine(hipStream_t st, const uint8_t* in, uint32_t n, Affine* out);

// k_accumulate.hip
// bases: AffPacked records; wide != 0 (experiments build, variant 8): AffWide records
void launch_accumulate(hipStream_t st, const Plan& p, const void* bases, int wide, const SortBuffers& b, PtI* buckets,
                       PtI* partials, int variant, uint32_t lds_bytes, hipEvent_t before_kernel, hipEvent_t after_kernel);

void launch_combine(hipStream_t st, const Plan& p, const SortBuffers& b, PtI* buckets, PtI* partials);

// k_reduce.hip
int reduce_set_attributes(const char** failed);
constexpr uint32_t kReduceGroup = 16;      // pipelined instances: chains of 15 additions, two levels at lb = 16
constexpr uint32_t kReduceGroupMin = 4;    // a lone call trades launches for shorter chains (launch_reduce picks per level)
constexpr size_t kReduceResidentLanes = 160 * 1024;   // sum_groups_kernel: 184 VGPRs, 2 waves/SIMD = 131 k lanes per round (a lone call's level picks the smallest group whose outputs stay near that)
size_t reduce_scratch_elems(uint32_t lb);   // PtI elements of S and of T per window (sized for kReduceGroupMin)
// bucket_size: [W][nb] point counts (zero = the bucket was never written and counts as the identity), or nullptr
// when every bucket holds a valid point (stage entry point sum_reduction)
// S, T: scratch of the row-sum / column-sum family, W * reduce_scratch_elems(lb) elements each
void launch_reduce(hipStream_t st, const Plan& p, const PtI* buckets, const uint32_t* bucket_size, PtI* S, PtI* T,
                   Jacobian* partial);

// k_misc.hip
void launch_projective_to_affine(hipStream_t st, const Jacobian* in, uint32_t n, Affine* out);
void launch_convert_bases(hipStream_t st, const Affine* in, uint32_t n, AffPacked* out);
#if defined(MSM_AMD_EXPERIMENTS)
void launch_convert_bases_wide(hipStream_t st, const Affine* in, uint32_t n, AffWide* out);
#endif
void launch_gen_instance(hipStream_t st, uint64_t seed, uint32_t n, int scalars_mont, Affine* bases, u256* scalars);

// k_stage.hip
// Wave priority of the front-end kernels (s_setprio 3 at their top): they share SIMDs with the accumulate grid of
// the previous instance, whose long straight-line VALU code starves ordinary-priority waves of issue slots
// (tools/microbench/contention.hip: 6-40x slower; with priority 3: no slowdown).  DESIGN.md section 4.
constexpr int kFrontPriority = 3;

constexpr int kRadixItems = 16;
constexpr int kRadixTile = 256 * kRadixItems;
void launch_ref_prepare(hipStream_t st, const u256* scalars, uint32_t n, uint32_t c, uint32_t W, uint2* pairs);
void launch_radix_sort_pairs(hipStream_t st, uint2* a, uint2* b, size_t n, uint32_t* tile_hist, uint2** result,
                             uint32_t key_bits = 32);
void launch_ref_accumulate(hipStream_t st, const uint2* pairs, size_t n_pairs, const Jacobian* points,
                           uint32_t n_points, uint32_t total_buckets, Jacobian* buckets);
void launch_pad_buckets(hipStream_t st, const Jacobian* in, uint32_t bs, uint32_t W, uint32_t lb, PtI* out);
void launch_hold(hipStream_t st, const uint32_t* release, uint64_t max_ticks);
void launch_filter_count(hipStream_t st, const u256* scalars, uint32_t n, uint32_t* block_counts);
void launch_filter_scatter(hipStream_t st, const u256* scalars, const Affine* points, uint32_t n,
                           const uint32_t* block_counts, u256* out_scalars, Affine* out_points);
void launch_test_op(hipStream_t st, int op, const u256* a, const u256* b, u256* out, uint32_t count);

// host_msm.hip (host code only)
Jacobian host_msm(const u256* scalars, int scalars_mont, const Affine* points, size_t n, int threads, bool* ok = nullptr);   // *ok = false: host allocation failed

}  // namespace msm_amd
