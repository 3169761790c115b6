// Kernels behind the per-stage C-ABI entry points (reference wire semantics) and the single-op test
// kernels.  Launched through launch.h.
//
//   ref_prepare_kernel      == kernel prepare_buckets_indices (msm.h.metal:17-59) output format
//   radix_*_kernel          stand-alone device sort of (u32 key, u32 value) pairs: the stage the
//                           reference left on the CPU (sort_buckets.rs:15-34).  LSD radix, 8 bits per
//                           pass, wave64 ballot ranking + LDS prefix sums, stable.
//   ref_accumulate_kernel   == kernel bucket_wise_accumulation (msm.h.metal:75-315) on Jacobian inputs
//   pad_buckets_kernel      feeds the production window reduction from a reference-layout bucket matrix
//   test_op_kernel          == the 12 single-thread test kernels of shader/tests/*.h.metal, batched
#include "device_common.hip.h"
#include "launch.h"
#include "test_ops.hip.h"

namespace msm_amd {


// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ref_prepare_kernel(const u256* __restrict__ scalars, uint32_t n, uint32_t c, uint32_t W,
                   uint2* __restrict__ pairs) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const u256 k = load_u256(&scalars[t]);
  const uint32_t bl = (c >= 32) ? 0xFFFFFFFFu : ((1u << c) - 1u);
  for (uint32_t i = 0; i < W; ++i) {
    const uint32_t start = i * c;
    const uint32_t m = (start < 256) ? u256_extract_bits(k, start, c) : 0u;
    uint2 pr;
    if (m != 0) {
      pr.x = i * bl + m - 1;
      pr.y = t;
    } else {
      pr.x = 0xFFFFFFFFu;
      pr.y = 0xFFFFFFFFu;
    }
    pairs[(size_t)t * W + i] = pr;
  }
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort, one pass = hist + scan + scatter.  Tile = 256 threads x kRadixItems items; each
// wave owns a contiguous quarter of the tile so that wave order == memory order (stability).

__global__ void __launch_bounds__(256)
radix_hist_kernel(const uint2* __restrict__ in, size_t n, uint32_t shift, uint32_t num_tiles,
                  uint32_t* __restrict__ tile_hist /* [256][num_tiles] */) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kRadixTile;
  for (int r = 0; r < kRadixItems; ++r) {
    const size_t i = base + (size_t)r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(in[i].x >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  tile_hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = h[threadIdx.x];
}

// Exclusive scan of a linear u32 array by ONE workgroup of 1024 threads (in place).
__global__ void __launch_bounds__(1024)
linear_scan_kernel(uint32_t* __restrict__ data, size_t len) {
  __shared__ uint32_t scratch[17];
  const size_t per = (len + blockDim.x - 1) / blockDim.x;
  const size_t first = (size_t)threadIdx.x * per;
  uint32_t local = 0;
  for (size_t j = 0; j < per; ++j)
    if (first + j < len) local += data[first + j];
  uint32_t total;
  uint32_t run = block_exclusive_scan(local, scratch, &total);
  for (size_t j = 0; j < per; ++j) {
    if (first + j < len) {
      const uint32_t v = data[first + j];
      data[first + j] = run;
      run += v;
    }
  }
}

// Exclusive scan of each digit's row of the [256][num_tiles] histogram (one workgroup per digit, coalesced
// chunks of blockDim tiles) + the digit totals; the scatter kernel adds the digit bases itself.
__global__ void __launch_bounds__(1024)
radix_row_scan_kernel(uint32_t* __restrict__ tile_hist, uint32_t num_tiles, uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t scratch[17];
  uint32_t* row = tile_hist + (size_t)blockIdx.x * num_tiles;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < num_tiles; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < num_tiles ? row[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan(v, scratch, &total);
    if (i < num_tiles) row[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(256)
radix_scatter_kernel(const uint2* __restrict__ in, uint2* __restrict__ out, size_t n, uint32_t shift,
                     uint32_t num_tiles, const uint32_t* __restrict__ tile_offset /* [256][num_tiles] */,
                     const uint32_t* __restrict__ digit_total /* [256] */) {
  __shared__ uint32_t wh[4][256];   // per-wave digit counts, then per-wave running write positions
  __shared__ uint32_t scratch[17];
  uint32_t all;
  const uint32_t digit_base = block_exclusive_scan(digit_total[threadIdx.x], scratch, &all);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int v = 0; v < 4; ++v) wh[v][threadIdx.x] = 0;
  __syncthreads();
  const size_t wbase = (size_t)blockIdx.x * kRadixTile + (size_t)wave * (kRadixTile / 4);
  for (int r = 0; r < kRadixItems; ++r) {
    const size_t i = wbase + (size_t)r * 64 + lane;
    if (i < n) atomicAdd(&wh[wave][(in[i].x >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  {
    // thread d turns the four per-wave counts of digit d into write positions
    uint32_t run = digit_base + tile_offset[(size_t)threadIdx.x * num_tiles + blockIdx.x];
    for (int v = 0; v < 4; ++v) {
      const uint32_t cnt = wh[v][threadIdx.x];
      wh[v][threadIdx.x] = run;
      run += cnt;
    }
  }
  __syncthreads();
  for (int r = 0; r < kRadixItems; ++r) {
    const size_t i = wbase + (size_t)r * 64 + lane;
    const bool valid = i < n;
    uint2 item = make_uint2(0, 0);
    if (valid) item = in[i];
    const uint32_t d = (item.x >> shift) & 0xFFu;
    // lanes holding the same digit: intersect 8 ballots (wavefront ballot ranking)
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long bal = __ballot((d >> bit) & 1u);
      same &= ((d >> bit) & 1u) ? bal : ~bal;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t rank = __popcll(same & lt);
    uint32_t pos = 0;
    if (valid) pos = wh[wave][d] + rank;
    // the lowest lane of each digit group advances the wave's cursor (same-wave LDS ops are ordered)
    if (valid && rank == 0) wh[wave][d] += __popcll(same);
    if (valid) out[pos] = item;
  }
}

// ------------------------------------------------------------------------------------------------
// Segmented sum of Jacobian points by sorted bucket key.  One thread per pair; the thread holding the
// first pair of a run accumulates the run.  Sentinel keys (0xFFFFFFFF) and keys >= total_buckets are
// skipped like the reference's CPU mirror (bucket_wise_accumulation.rs:671-678).
__global__ void __launch_bounds__(64)
ref_accumulate_kernel(const uint2* __restrict__ pairs, size_t n_pairs, const Jacobian* __restrict__ points,
                      uint32_t n_points, uint32_t total_buckets, Jacobian* __restrict__ buckets) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const uint32_t key = pairs[i].x;
  if (key == 0xFFFFFFFFu || key >= total_buckets) return;
  if (i > 0 && pairs[i - 1].x == key) return;
  Jacobian acc = jac_identity();
#pragma unroll 1
  for (size_t j = i; j < n_pairs && pairs[j].x == key; ++j) {
    const uint32_t pi = pairs[j].y;
    if (pi < n_points) acc = jac_add(acc, load_jac(&points[pi]));
  }
  store_jac(&buckets[key], acc);
}

// Reference bucket matrix [W][bs] (column b has weight b+1) -> production layout [W][nb], nb = 2^lb >= bs,
// slot i has weight i + 1: a plain copy with identity padding and conversion to the internal limbs.
__global__ void __launch_bounds__(256)
pad_buckets_kernel(const Jacobian* __restrict__ in, uint32_t bs, uint32_t W, uint32_t lb,
                   PtI* __restrict__ out) {
  const uint32_t nb = 1u << lb;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= W * nb) return;
  const uint32_t w = t >> lb, d = t & (nb - 1);
  Jacobian v = jac_identity();
  if (d < bs) v = load_jac(&in[(size_t)w * bs + d]);
  store_pti(&out[t], pti_from_ext(v));   // production window reduction works on internal limbs
}

// ------------------------------------------------------------------------------------------------
// filter_zeros (src/metal/msm.rs:448-507): drop (scalar, point) pairs whose scalar is zero.  The reference does
// it on the CPU with rayon when at least 30 % of the scalars are zero.  Three kernels: per-block count of
// non-zero scalars, exclusive scan of the block counts (linear_scan_kernel), order-preserving scatter.
constexpr int kFilterThreads = 1024;

__global__ void __launch_bounds__(kFilterThreads)
filter_count_kernel(const u256* __restrict__ scalars, uint32_t n, uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t scratch[17];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nz = 0;
  if (t < n) nz = u256_is_zero(load_u256(&scalars[t])) ? 0u : 1u;
  uint32_t total;
  (void)block_exclusive_scan(nz, scratch, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kFilterThreads)
filter_scatter_kernel(const u256* __restrict__ scalars, const Affine* __restrict__ points, uint32_t n,
                      const uint32_t* __restrict__ block_offsets, u256* __restrict__ out_scalars,
                      Affine* __restrict__ out_points) {
  __shared__ uint32_t scratch[17];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  u256 k = u256_zero();
  uint32_t nz = 0;
  if (t < n) {
    k = load_u256(&scalars[t]);
    nz = u256_is_zero(k) ? 0u : 1u;
  }
  uint32_t total;
  const uint32_t rank = block_exclusive_scan(nz, scratch, &total);
  if (nz) {
    const uint32_t pos = block_offsets[blockIdx.x] + rank;
    store_u256(&out_scalars[pos], k);
    store_affine(&out_points[pos], load_affine(&points[t]));
  }
}

// ------------------------------------------------------------------------------------------------
// Batched single-op kernel (bodies in test_ops.hip.h).
__global__ void __launch_bounds__(64)
test_op_kernel(int op, const u256* __restrict__ a, const u256* __restrict__ b, u256* __restrict__ out,
               uint32_t count) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  run_test_op(op, a, b, out, t);
}

// ------------------------------------------------------------------------------------------------
void launch_ref_prepare(hipStream_t st, const u256* scalars, uint32_t n, uint32_t c, uint32_t W, uint2* pairs) {
  hipLaunchKernelGGL(ref_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scalars, n, c, W, pairs);
}

// Sorts n pairs by the low key_bits bits of the key; a and b are ping-pong buffers of n pairs, tile_hist holds
// 256 * (tiles + 1) words.  *result points to whichever buffer holds the sorted output.
void launch_radix_sort_pairs(hipStream_t st, uint2* a, uint2* b, size_t n, uint32_t* tile_hist, uint2** result,
                             uint32_t key_bits) {
  const uint32_t tiles = (uint32_t)((n + kRadixTile - 1) / kRadixTile);
  uint32_t* digit_total = tile_hist + (size_t)tiles * 256;
  uint2* src = a;
  uint2* dst = b;
  for (uint32_t shift = 0; shift < key_bits; shift += 8) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(256), 0, st, (const uint2*)src, n, shift, tiles,
                       tile_hist);
    hipLaunchKernelGGL(radix_row_scan_kernel, dim3(256), dim3(1024), 0, st, tile_hist, tiles, digit_total);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(256), 0, st, (const uint2*)src, dst, n, shift, tiles,
                       (const uint32_t*)tile_hist, (const uint32_t*)digit_total);
    uint2* t = src;
    src = dst;
    dst = t;
  }
  *result = src;
}

void launch_ref_accumulate(hipStream_t st, const uint2* pairs, size_t n_pairs, const Jacobian* points,
                           uint32_t n_points, uint32_t total_buckets, Jacobian* buckets) {
  hipLaunchKernelGGL(ref_accumulate_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, st, pairs, n_pairs,
                     points, n_points, total_buckets, buckets);
}

void launch_pad_buckets(hipStream_t st, const Jacobian* in, uint32_t bs, uint32_t W, uint32_t lb, PtI* out) {
  const size_t total = (size_t)W << lb;
  hipLaunchKernelGGL(pad_buckets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, bs, W, lb, out);
}

// Two steps, so that the caller can skip the compaction when too few scalars are zero (msm.rs:470: 30 %).
// block_counts: ceil(n / 1024) + 1 words; after launch_filter_count block_counts[nblocks] holds the number of
// survivors and block_counts[b] the first output position of block b.
void launch_filter_count(hipStream_t st, const u256* scalars, uint32_t n, uint32_t* block_counts) {
  const uint32_t nblocks = (n + kFilterThreads - 1) / kFilterThreads;
  hipLaunchKernelGGL(filter_count_kernel, dim3(nblocks), dim3(kFilterThreads), 0, st, scalars, n, block_counts);
  (void)hipMemsetAsync(block_counts + nblocks, 0, 4, st);
  hipLaunchKernelGGL(linear_scan_kernel, dim3(1), dim3(1024), 0, st, block_counts, (size_t)nblocks + 1);
}

void launch_filter_scatter(hipStream_t st, const u256* scalars, const Affine* points, uint32_t n,
                           const uint32_t* block_counts, u256* out_scalars, Affine* out_points) {
  const uint32_t nblocks = (n + kFilterThreads - 1) / kFilterThreads;
  hipLaunchKernelGGL(filter_scatter_kernel, dim3(nblocks), dim3(kFilterThreads), 0, st, scalars, points, n, block_counts,
                     out_scalars, out_points);
}

void launch_test_op(hipStream_t st, int op, const u256* a, const u256* b, u256* out, uint32_t count) {
  hipLaunchKernelGGL(test_op_kernel, dim3((count + 63) / 64), dim3(64), 0, st, op, a, b, out, count);
}

}  // namespace msm_amd
