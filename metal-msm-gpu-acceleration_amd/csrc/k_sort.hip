// Sort-stage kernels: digit extraction, per-window two-pass LDS counting sort, work-item planning.
// See device_common.hip.h for the pipeline overview.
#include <algorithm>

#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

constexpr int kSortThreads = 1024;       // hist / plan / scatter workgroup size

// ------------------------------------------------------------------------------------------------
// Stage 1: signed digit extraction.  k = sum_w d_w 2^(c w) with d_w in (-2^(c-1), 2^(c-1)]: a raw window value
// v (plus the carry of the window below) above 2^(c-1) becomes v - 2^c with a carry into the next window, so
// only 2^(c-1) bucket magnitudes exist per window -- half the buckets of the reference's unsigned digits
// (kernel prepare_buckets_indices, msm.h.metal:17-59) for the window reduction to fold.  A negative digit adds
// -P, which costs one field negation of y.  Output per (window, point): u16 = sign << 15 | magnitude
// (magnitude 0 = no contribution).  scalars_mont: 1 = host Montgomery form (bn256::Fr / ark Fr memory; the
// reference de-Montgomerys on the CPU, limbs_conversion.rs:282-288), 0 = canonical integer.
// D = uint16_t (c <= 15: the per-call pipeline) or uint32_t (c <= 24: precomputed window tables, where the
// [W][n] digit matrix is consumed as ONE window of W * n entries, see make_table_plan in msm_host.hip).
// C > 0: window size and window count are compile-time constants (c = C, W = 254 / C + 1): the window loop unrolls
// and every digit is one v_alignbit + v_and on fixed words of the scalar instead of a 16-way select chain over a
// run-time word index (the common window sizes of the pipelined policy; 375 -> ~140 instructions per scalar).
// C = 0: c and W from the arguments.
template <typename D, int C = 0>
__global__ void __launch_bounds__(256)
digits_kernel(const u256* __restrict__ scalars, uint32_t n, uint32_t c_arg, uint32_t W_arg, int scalars_mont,
              D* __restrict__ digits) {
  constexpr uint32_t kSignShift = 8 * sizeof(D) - 1;
  const uint32_t c = C > 0 ? (uint32_t)C : c_arg;
  const uint32_t W = C > 0 ? (uint32_t)(254 / (C > 0 ? C : 1) + 1) : W_arg;
  __builtin_amdgcn_s_setprio(kFrontPriority);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u256 k = load_u256(&scalars[t]);
  if (scalars_mont) {
    k = Fr::from_mont(k);
  } else {
    // canonical layouts carry raw 256-bit integers (instance files, FFI callers): bring them below r first, so
    // that every window size sees the same scalar (2^256 / r < 6: at most five subtractions)
#pragma unroll 1
    for (int i = 0; i < 5; ++i) k = Fr::reduce_once(k);
  }
  const uint32_t half = 1u << (c - 1);
  uint32_t carry = 0;
#pragma unroll
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t start = w * c;
    uint32_t v = (start < 256 ? u256_extract_bits(k, start, c) : 0u) + carry;
    uint32_t neg = 0;
    carry = 0;
    if (v > half) {
      v = (1u << c) - v;
      neg = 1;
      carry = 1;
    }
    digits[(size_t)w * n + t] = (D)(v | (neg << kSignShift));
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 2: per-window sort of the point indices by bucket slot -- a two-pass MSD counting sort.
//
// A one-pass scatter (every (chunk, window) workgroup writing 4-byte indices at random positions of the
// window's 4 MB slice) measured 545 MB of WRITE_SIZE for 71 MB of payload at 2^20: the eight XCD-private L2s
// evict partially written lines.  So the slot (lb bits) is split into hb coarse + fb fine bits:
//   pass 1  coarse_hist_kernel / coarse_prefix_kernel / coarse_scatter_kernel
//           every (chunk q, window w) workgroup moves its points into 2^hb coarse regions of the window.
//           A workgroup owns one contiguous run per region (positions from returning LDS atomics on 2^hb
//           region cursors), so its stores extend the same few lines and one L2 assembles them whole.
//           Payload: index|sign (u32) + fine digit (u16).
//   pass 2  fine_sort_kernel: one workgroup per (region, window) counting-sorts its ~16 k points over the
//           2^fb fine slots entirely in LDS (histogram, scan, scatter into an LDS staging buffer) and writes
//           the region back fully coalesced, together with the bucket sizes.  Regions larger than the staging
//           buffer (skewed digits) fall back to scattering inside their own slice.
// The reference sorts 8-byte (bucket, point) pairs of ALL windows on the CPU (sort_buckets.rs:15-34).
constexpr uint32_t kFineCap = 28672;     // LDS staging entries of pass 2 (112 KB)

template <typename D>
__global__ void __launch_bounds__(kSortThreads)
coarse_hist_kernel(const D* __restrict__ digits, uint32_t n, uint32_t fb, uint32_t nhi, uint32_t chunk,
                   uint32_t* __restrict__ coarse_cnt /* [W][Q][nhi] */) {
  constexpr uint32_t kMagMask = (1u << (8 * sizeof(D) - 1)) - 1u;
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  const uint32_t q = blockIdx.x, Q = gridDim.x, w = blockIdx.y;
  for (uint32_t i = threadIdx.x; i < nhi; i += blockDim.x) lds_u32[i] = 0;
  __syncthreads();
  const uint32_t lo = q * chunk;
  const uint32_t hi = min(n, lo + chunk);
  const D* dw = digits + (size_t)w * n;
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
    const uint32_t m = dw[t] & kMagMask;
    if (m) atomicAdd(&lds_u32[(m - 1) >> fb], 1u);
  }
  __syncthreads();
  uint32_t* out = coarse_cnt + ((size_t)w * Q + q) * nhi;
  for (uint32_t i = threadIdx.x; i < nhi; i += blockDim.x) out[i] = lds_u32[i];
}

// grid = W, block = 1024 (>= nhi).  coarse_cnt[w][q][hi] -> first position (inside the window slice) of chunk
// q's run in region hi; region_start[w][hi] (nhi + 1 entries per window).
__global__ void __launch_bounds__(1024)
coarse_prefix_kernel(uint32_t* __restrict__ coarse_cnt, uint32_t Q, uint32_t nhi,
                     uint32_t* __restrict__ region_start) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  __shared__ uint32_t scratch[17];
  const uint32_t w = blockIdx.x, hi = threadIdx.x;
  uint32_t* cw = coarse_cnt + (size_t)w * Q * nhi;
  uint32_t tot = 0;
  if (hi < nhi) {
#pragma unroll 8
    for (uint32_t q = 0; q < Q; ++q) tot += cw[(size_t)q * nhi + hi];   // independent loads: keep 8 in flight
  }
  uint32_t total;
  uint32_t start = block_exclusive_scan(tot, scratch, &total);
  if (hi < nhi) {
    region_start[(size_t)w * (nhi + 1) + hi] = start;
    uint32_t run = start;
#pragma unroll 8
    for (uint32_t q = 0; q < Q; ++q) {
      const uint32_t c = cw[(size_t)q * nhi + hi];
      cw[(size_t)q * nhi + hi] = run;
      run += c;
    }
  }
  if (hi == 0) region_start[(size_t)w * (nhi + 1) + nhi] = total;
}


// Wave-level multisplit: the 64 lanes of a wave claim consecutive positions per key with ONE LDS atomic per distinct
// key in the wave instead of one per lane (ballot per key bit -> mask of the lanes holding the same key; rank =
// population count of that mask below the lane; the lowest lane of each group adds the group size to the key's
// cursor and hands the base to its peers).  Every lane of the wave must call it (`valid` = lane has an element).
// Same-key lanes are ranked in lane order, so the pass is stable.  north_star: "wavefront ballot/prefix-sum for
// bucket index sorting"; A/B against the plain returning-atomic ranking: profiles/r02_sort_ranking_ab.txt.
__device__ __forceinline__ uint32_t wave_claim(uint32_t key, uint32_t bits, bool valid, uint32_t* cursors) {
  // lanes that agree with this lane on every key bit: s = -bit (one v_bfe_i32); ~(ballot ^ s) keeps the lanes whose
  // bit equals this lane's (the compiler fuses the and-xnor into one v_bitop3_b32 per half: 4 VALU instructions per
  // key bit; the select  peers &= bit ? bal : ~bal  on a 64-bit mask took 8)
  const uint64_t peers0 = __ballot(valid);
  uint32_t lo = (uint32_t)peers0, hi = (uint32_t)(peers0 >> 32);
#pragma unroll 1
  for (uint32_t b = 0; b < bits; ++b) {
    const int32_t s = __builtin_amdgcn_sbfe((int32_t)key, b, 1u);   // 0 or -1
    const uint64_t bal = __ballot(s != 0);
    lo &= ~((uint32_t)bal ^ (uint32_t)s);
    hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)s);
  }
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
  const uint32_t count = __popc(lo) + __popc(hi);
  const uint32_t leader = lo ? (uint32_t)__builtin_ctz(lo) : 32u + (uint32_t)__builtin_ctz(hi | 0x80000000u);
  uint32_t base = 0;
  if (valid && rank == 0) base = atomicAdd(&cursors[key], count);
  base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(leader << 2), (int)base);
  return base + rank;
}

// grid = (Q, W), block = 1024, dynamic LDS = nhi * 4 bytes.  The workgroup owns ONE contiguous run per region
// (coarse_base[w][q][hi] ..): lanes take positions with a returning LDS atomic on the region cursor, so the
// 2^hb open output lines of a workgroup are written by its own 16 waves only and complete inside one L2.
template <typename D, bool BALLOT>
__global__ void __launch_bounds__(kSortThreads)
coarse_scatter_kernel(const D* __restrict__ digits, uint32_t n, uint32_t hb, uint32_t fb, uint32_t chunk,
                      const uint32_t* __restrict__ coarse_base /* [W][Q][nhi] */, uint32_t* __restrict__ tmp_idx,
                      uint16_t* __restrict__ tmp_fine) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];   // [nhi] region cursors
  const uint32_t nhi = 1u << hb;
  const uint32_t q = blockIdx.x, Q = gridDim.x, w = blockIdx.y;
  const uint32_t* base = coarse_base + ((size_t)w * Q + q) * nhi;
  for (uint32_t i = threadIdx.x; i < nhi; i += blockDim.x) lds_u32[i] = base[i];
  __syncthreads();
  const uint32_t lo = q * chunk;
  const uint32_t hi_end = min(n, lo + chunk);
  constexpr uint32_t kSignShift = 8 * sizeof(D) - 1;
  constexpr uint32_t kMagMask = (1u << kSignShift) - 1u;
  const D* dw = digits + (size_t)w * n;
  uint32_t* ti = tmp_idx + (size_t)w * n;
  uint16_t* tf = tmp_fine + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  if (BALLOT) {
    for (uint32_t t0 = lo; t0 < hi_end; t0 += blockDim.x) {   // wave-uniform trip count
      const uint32_t t = t0 + threadIdx.x;
      const uint32_t v = t < hi_end ? (uint32_t)dw[t] : 0u;
      const uint32_t m = v & kMagMask;
      const uint32_t slot = m - 1;
      const uint32_t pos = wave_claim(m ? slot >> fb : 0u, hb, m != 0, lds_u32);
      if (m) {
        ti[pos] = t | ((v >> kSignShift) << 31);
        tf[pos] = (uint16_t)(slot & fmask);
      }
    }
    return;
  }
  for (uint32_t t = lo + threadIdx.x; t < hi_end; t += blockDim.x) {
    const uint32_t v = dw[t];
    const uint32_t m = v & kMagMask;
    if (m) {
      const uint32_t slot = m - 1;
      const uint32_t pos = atomicAdd(&lds_u32[slot >> fb], 1u);
      ti[pos] = t | ((v >> kSignShift) << 31);   // bit 31: the digit is negative, add -P
      tf[pos] = (uint16_t)(slot & fmask);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Stage 2, optional middle pass (three-level sort).  When a window holds so many entries that 2^hb <= 128 coarse
// regions are still far larger than the LDS staging of pass 2 (2^23+ points, or the single 13-window-long window of
// the table pipeline), every coarse region is split once more by the next mb bits of the slot, with exactly the
// structure of pass 1 (histogram per chunk, prefix, scatter with LDS region cursors) applied INSIDE each region:
//   mid_hist_kernel / mid_prefix_kernel / mid_scatter_kernel, grid.y = window * 2^hb + coarse region.
// Both coarse passes keep at most 32..64 output lines open per workgroup (the 1024-region single pass measured
// 10.2 ms at 2^24 points, 128 regions 6.0 ms with a slow scattering pass 2); pass 2 then always finds regions of
// ~16 k entries that it sorts inside LDS.  region_start2[w][(r << mb) + m] has the layout pass 2 expects.
__global__ void __launch_bounds__(kSortThreads)
mid_hist_kernel(const uint16_t* __restrict__ tmp_fine, uint32_t n, uint32_t nhi, uint32_t fb, uint32_t nmid,
                const uint32_t* __restrict__ region_start, uint32_t* __restrict__ mid_cnt /* [W*nhi][Q2][nmid] */) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  const uint32_t q = blockIdx.x, Q2 = gridDim.x, R = blockIdx.y, w = R / nhi, r = R % nhi;
  for (uint32_t i = threadIdx.x; i < nmid; i += blockDim.x) lds_u32[i] = 0;
  __syncthreads();
  const uint32_t rs = region_start[(size_t)w * (nhi + 1) + r], re = region_start[(size_t)w * (nhi + 1) + r + 1];
  const uint32_t chunk = (((re - rs) + Q2 - 1) / Q2 + 63u) & ~63u;
  const uint32_t lo = min(re, rs + q * chunk), hi = min(re, lo + chunk);
  const uint16_t* tf = tmp_fine + (size_t)w * n;
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) atomicAdd(&lds_u32[tf[t] >> fb], 1u);
  __syncthreads();
  uint32_t* out = mid_cnt + ((size_t)R * Q2 + q) * nmid;
  for (uint32_t i = threadIdx.x; i < nmid; i += blockDim.x) out[i] = lds_u32[i];
}

// grid = W * nhi, block = 1024 (>= nmid): counts -> first position (inside the window slice) of chunk q's run in
// sub-region m of region r; region_start2[w][(r << mb) + m], terminal entry written by the window's last region.
__global__ void __launch_bounds__(1024)
mid_prefix_kernel(uint32_t* __restrict__ mid_cnt, uint32_t Q2, uint32_t nhi, uint32_t nmid,
                  const uint32_t* __restrict__ region_start, uint32_t* __restrict__ region_start2) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  __shared__ uint32_t scratch[17];
  const uint32_t R = blockIdx.x, w = R / nhi, r = R % nhi, m = threadIdx.x;
  uint32_t* cw = mid_cnt + (size_t)R * Q2 * nmid;
  const uint32_t rs = region_start[(size_t)w * (nhi + 1) + r];
  uint32_t tot = 0;
  if (m < nmid) {
#pragma unroll 4
    for (uint32_t q = 0; q < Q2; ++q) tot += cw[(size_t)q * nmid + m];
  }
  uint32_t total;
  const uint32_t start = rs + block_exclusive_scan(tot, scratch, &total);
  const size_t nhi2 = (size_t)nhi * nmid;
  if (m < nmid) {
    region_start2[(size_t)w * (nhi2 + 1) + (size_t)r * nmid + m] = start;
    uint32_t run = start;
#pragma unroll 4
    for (uint32_t q = 0; q < Q2; ++q) {
      const uint32_t c = cw[(size_t)q * nmid + m];
      cw[(size_t)q * nmid + m] = run;
      run += c;
    }
  }
  if (m == 0 && r == nhi - 1) region_start2[(size_t)w * (nhi2 + 1) + nhi2] = rs + total;
}

template <bool BALLOT>
__global__ void __launch_bounds__(kSortThreads)
mid_scatter_kernel(const uint32_t* __restrict__ tmp_idx, const uint16_t* __restrict__ tmp_fine, uint32_t n, uint32_t nhi,
                   uint32_t fb, uint32_t nmid, const uint32_t* __restrict__ region_start,
                   const uint32_t* __restrict__ mid_base, uint32_t* __restrict__ tmp_idx2,
                   uint16_t* __restrict__ tmp_fine2) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];   // [nmid] sub-region cursors
  const uint32_t q = blockIdx.x, Q2 = gridDim.x, R = blockIdx.y, w = R / nhi, r = R % nhi;
  const uint32_t* base = mid_base + ((size_t)R * Q2 + q) * nmid;
  for (uint32_t i = threadIdx.x; i < nmid; i += blockDim.x) lds_u32[i] = base[i];
  __syncthreads();
  const uint32_t rs = region_start[(size_t)w * (nhi + 1) + r], re = region_start[(size_t)w * (nhi + 1) + r + 1];
  const uint32_t chunk = (((re - rs) + Q2 - 1) / Q2 + 63u) & ~63u;
  const uint32_t lo = min(re, rs + q * chunk), hi = min(re, lo + chunk);
  const uint32_t* ti = tmp_idx + (size_t)w * n;
  const uint16_t* tf = tmp_fine + (size_t)w * n;
  uint32_t* ti2 = tmp_idx2 + (size_t)w * n;
  uint16_t* tf2 = tmp_fine2 + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  if (BALLOT) {
    uint32_t mbits = 0;
    while ((1u << mbits) < nmid) ++mbits;
    for (uint32_t t0 = lo; t0 < hi; t0 += blockDim.x) {
      const uint32_t t = t0 + threadIdx.x;
      const bool valid = t < hi;
      const uint32_t f = valid ? (uint32_t)tf[t] : 0u;
      const uint32_t pos = wave_claim(f >> fb, mbits, valid, lds_u32);
      if (valid) {
        ti2[pos] = ti[t];
        tf2[pos] = (uint16_t)(f & fmask);
      }
    }
    return;
  }
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
    const uint32_t f = tf[t];
    const uint32_t pos = atomicAdd(&lds_u32[f >> fb], 1u);
    ti2[pos] = ti[t];
    tf2[pos] = (uint16_t)(f & fmask);
  }
}


// Tile-staged scatter of the coarse / middle passes.  Writing every entry straight to its region's run costs one
// L2 transaction per ~2 entries (64 lanes of a store hit ~32 different lines, 4 + 2 bytes each): 251 M entries per
// pass at 2^24 points took 1.8-2.2 ms, the L2 transaction rate, not HBM (profiles/r02_sort_ranking_ab.txt).  Here a
// workgroup first groups a tile of kTileEntries entries by region inside LDS (ranks from wave_claim on per-tile
// counters, exclusive scan over the <= 128 regions), reserves the tile's share of every region's run, and then
// writes the tile out in staging order: consecutive lanes write consecutive addresses of one run (runs of
// tile / regions ~ 128..256 entries), i.e. whole lines.
constexpr uint32_t kTilePerThread = 8;
constexpr uint32_t kTileEntries = kSortThreads * kTilePerThread;   // 8192 entries: 56 KB of LDS staging
constexpr uint32_t kTileMaxRegions = 128;
constexpr size_t kTileLdsBytes = (4 * kTileMaxRegions + 40) * 4 + kTileEntries * 4 + kTileEntries * 2 + kTileEntries;

// load(t, &payload, &fine, &region) -> valid.  cursors_src: first global position of this workgroup's run per region.
template <typename Load>
__device__ __forceinline__ void tile_scatter(uint32_t lo, uint32_t hi, uint32_t nreg, uint32_t bits, Load load,
                                             const uint32_t* __restrict__ cursors_src, uint32_t* __restrict__ out_idx,
                                             uint16_t* __restrict__ out_fine, uint32_t* lds) {
  uint32_t* cursor = lds;                          // [128] next global position per region
  uint32_t* cnt = lds + kTileMaxRegions;           // [128] entries of the tile per region
  uint32_t* off = lds + 2 * kTileMaxRegions;       // [128] first staging index per region
  uint32_t* gbase = lds + 3 * kTileMaxRegions;     // [128] global position of the tile's first entry per region
  uint32_t* scratch = lds + 4 * kTileMaxRegions;   // 17 words (block scan) + [32] tile total
  uint32_t* st_idx = lds + 4 * kTileMaxRegions + 40;
  uint16_t* st_fine = reinterpret_cast<uint16_t*>(st_idx + kTileEntries);
  uint8_t* st_reg = reinterpret_cast<uint8_t*>(st_fine + kTileEntries);
  for (uint32_t i = threadIdx.x; i < nreg; i += blockDim.x) cursor[i] = cursors_src[i];
  const uint32_t tile_span = blockDim.x * kTilePerThread;
  for (uint32_t t0 = lo; t0 < hi; t0 += tile_span) {
    for (uint32_t i = threadIdx.x; i < nreg; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    uint32_t pay[kTilePerThread], fin[kTilePerThread], reg[kTilePerThread], rank[kTilePerThread];
#pragma unroll
    for (uint32_t e = 0; e < kTilePerThread; ++e) {
      const uint32_t t = t0 + e * blockDim.x + threadIdx.x;
      pay[e] = 0; fin[e] = 0; reg[e] = 0;
      const bool valid = t < hi && load(t, &pay[e], &fin[e], &reg[e]);
      rank[e] = wave_claim(reg[e], bits, valid, cnt);
      if (!valid) reg[e] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t total;
    const uint32_t c = threadIdx.x < nreg ? cnt[threadIdx.x] : 0u;
    const uint32_t ex = block_exclusive_scan(c, scratch, &total);
    if (threadIdx.x < nreg) {
      off[threadIdx.x] = ex;
      gbase[threadIdx.x] = cursor[threadIdx.x];
      cursor[threadIdx.x] += c;
    }
    if (threadIdx.x == 0) scratch[32] = total;
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < kTilePerThread; ++e) {
      if (reg[e] != 0xFFFFFFFFu) {
        const uint32_t p = off[reg[e]] + rank[e];
        st_idx[p] = pay[e];
        st_fine[p] = (uint16_t)fin[e];
        st_reg[p] = (uint8_t)reg[e];
      }
    }
    __syncthreads();
    const uint32_t tile_total = scratch[32];
    for (uint32_t j = threadIdx.x; j < tile_total; j += blockDim.x) {
      const uint32_t r = st_reg[j];
      const uint32_t gp = gbase[r] + (j - off[r]);
      out_idx[gp] = st_idx[j];
      out_fine[gp] = st_fine[j];
    }
    __syncthreads();
  }
}

template <typename D>
__global__ void __launch_bounds__(kSortThreads)
coarse_scatter_tiled_kernel(const D* __restrict__ digits, uint32_t n, uint32_t hb, uint32_t fb, uint32_t chunk,
                            const uint32_t* __restrict__ coarse_base, uint32_t* __restrict__ tmp_idx,
                            uint16_t* __restrict__ tmp_fine) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  constexpr uint32_t kSignShift = 8 * sizeof(D) - 1;
  constexpr uint32_t kMagMask = (1u << kSignShift) - 1u;
  const uint32_t nhi = 1u << hb;
  const uint32_t q = blockIdx.x, Q = gridDim.x, w = blockIdx.y;
  const uint32_t lo = q * chunk, hi = min(n, lo + chunk);
  const D* dw = digits + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  auto load = [&](uint32_t t, uint32_t* pay, uint32_t* fin, uint32_t* reg) {
    const uint32_t v = dw[t];
    const uint32_t m = v & kMagMask;
    if (!m) return false;
    const uint32_t slot = m - 1;
    *pay = t | ((v >> kSignShift) << 31);   // bit 31: the digit is negative, add -P
    *fin = slot & fmask;
    *reg = slot >> fb;
    return true;
  };
  tile_scatter(lo, hi, nhi, hb, load, coarse_base + ((size_t)w * Q + q) * nhi, tmp_idx + (size_t)w * n,
               tmp_fine + (size_t)w * n, lds_u32);
}

__global__ void __launch_bounds__(kSortThreads)
mid_scatter_tiled_kernel(const uint32_t* __restrict__ tmp_idx, const uint16_t* __restrict__ tmp_fine, uint32_t n,
                         uint32_t nhi, uint32_t fb, uint32_t mb, const uint32_t* __restrict__ region_start,
                         const uint32_t* __restrict__ mid_base, uint32_t* __restrict__ tmp_idx2,
                         uint16_t* __restrict__ tmp_fine2) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nmid = 1u << mb;
  const uint32_t q = blockIdx.x, Q2 = gridDim.x, R = blockIdx.y, w = R / nhi, r = R % nhi;
  const uint32_t rs = region_start[(size_t)w * (nhi + 1) + r], re = region_start[(size_t)w * (nhi + 1) + r + 1];
  const uint32_t chunk = (((re - rs) + Q2 - 1) / Q2 + 63u) & ~63u;
  const uint32_t lo = min(re, rs + q * chunk), hi = min(re, lo + chunk);
  const uint32_t* ti = tmp_idx + (size_t)w * n;
  const uint16_t* tf = tmp_fine + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  auto load = [&](uint32_t t, uint32_t* pay, uint32_t* fin, uint32_t* reg) {
    const uint32_t f = tf[t];
    *pay = ti[t];
    *fin = f & fmask;
    *reg = f >> fb;
    return true;
  };
  tile_scatter(lo, hi, nmid, mb, load, mid_base + ((size_t)R * Q2 + q) * nmid, tmp_idx2 + (size_t)w * n,
               tmp_fine2 + (size_t)w * n, lds_u32);
}

// grid = (nhi, W), block = 1024, dynamic LDS = (kFineCap + 2 * nfine + 32) * 4 bytes.
// Loops are kept rolled (<= 32 VGPRs): the 4 waves/SIMD of a 1024-thread workgroup must fit into the 128
// VGPRs that two resident accumulate waves leave free on a SIMD, or the workgroup waits for the accumulate tail.
template <bool BALLOT>
__global__ void __launch_bounds__(kSortThreads)
fine_sort_kernel(const uint32_t* __restrict__ tmp_idx, const uint16_t* __restrict__ tmp_fine, uint32_t n,
                 uint32_t lb, uint32_t fb, const uint32_t* __restrict__ region_start,
                 uint32_t* __restrict__ sorted, uint32_t* __restrict__ bucket_size) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nfine = 1u << fb;
  const uint32_t nhi = gridDim.x;
  const uint32_t hi = blockIdx.x, w = blockIdx.y;
  uint32_t* bins = lds_u32;                 // [nfine] counts, then cursors
  uint32_t* scratch = lds_u32 + nfine;      // 17 words (block scan)
  uint32_t* staging = lds_u32 + nfine + 32; // [kFineCap]
  const uint32_t rs = region_start[(size_t)w * (nhi + 1) + hi];
  const uint32_t re = region_start[(size_t)w * (nhi + 1) + hi + 1];
  const uint32_t size = re - rs;
  const uint32_t* ti = tmp_idx + (size_t)w * n + rs;
  const uint16_t* tf = tmp_fine + (size_t)w * n + rs;
  uint32_t* out = sorted + (size_t)w * n + rs;
  for (uint32_t i = threadIdx.x; i < nfine; i += blockDim.x) bins[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) atomicAdd(&bins[tf[i]], 1u);
  __syncthreads();
  // exclusive scan of the fine histogram (nfine <= 1024: one bin per thread)
  const uint32_t cnt = (threadIdx.x < nfine) ? bins[threadIdx.x] : 0u;
  uint32_t total;
  const uint32_t start = block_exclusive_scan(cnt, scratch, &total);
  if (threadIdx.x < nfine) {
    bins[threadIdx.x] = start;
    const uint32_t slot = (hi << fb) | threadIdx.x;
    if (slot < (1u << lb)) bucket_size[((size_t)w << lb) + slot] = cnt;
  }
  __syncthreads();
  if (size <= kFineCap) {
    if (BALLOT) {
#pragma unroll 1
      for (uint32_t i0 = 0; i0 < size; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const bool valid = i < size;
        const uint32_t pos = wave_claim(valid ? (uint32_t)tf[i] : 0u, fb, valid, bins);
        if (valid) staging[pos] = ti[i];
      }
    } else {
#pragma unroll 1
      for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) {
        const uint32_t pos = atomicAdd(&bins[tf[i]], 1u);
        staging[pos] = ti[i];
      }
    }
    __syncthreads();
  #pragma unroll 1
  for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) out[i] = staging[i];
  } else {
  #pragma unroll 1
  for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) {
      const uint32_t pos = atomicAdd(&bins[tf[i]], 1u);
      out[pos] = ti[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 2 (end): per-window planning.  grid = W, block = kSortThreads, dynamic LDS = (nb + nb/32 + 33) * 4 bytes.
// Two exclusive scans over the window's buckets (one LDS array, used twice):
//   bucket_start[w][d] = offset of bucket d inside the window's slice of `sorted`
//   item_start[w][d]   = first work-item id of bucket d inside the window, where a bucket of s points
//                        is cut into ceil(s / CH) items of at most CH points (empty buckets: none)
// and win_items[w] = number of items of the window.
__device__ __forceinline__ uint32_t window_scan_lds(uint32_t* tot, uint32_t* scratch, uint32_t nb) {
  // in: tot[skew(d)] = value of bucket d ; out: tot[skew(d)] = exclusive prefix ; returns the total
  const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;   // consecutive entries per thread
  const uint32_t first = threadIdx.x * per;
  uint32_t local = 0;
#pragma unroll 1
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t d = first + j;
    if (d < nb) local += tot[d + (d >> 5)];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(local, scratch, &total);
#pragma unroll 1
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t d = first + j;
    if (d < nb) {
      const uint32_t s = tot[d + (d >> 5)];
      tot[d + (d >> 5)] = run;
      run += s;
    }
  }
  __syncthreads();
  return total;
}

// Windows of more than kPlanTile slots (precomputed-table mode: one window of 2^15 .. 2^20 slots) are scanned by one
// workgroup per tile: plan_tile_sums_kernel first writes every tile's totals, plan_kernel then adds the totals of
// the tiles before its own.  The per-call pipeline (<= 2^14 slots per window) is a single tile and skips the
// first kernel.
constexpr uint32_t kPlanTile = 16384;

// grid = (tiles, W): tile_sums[w][tile] = (points, work items) of the tile
__global__ void __launch_bounds__(kSortThreads)
plan_tile_sums_kernel(const uint32_t* __restrict__ bucket_size, uint32_t lb, uint32_t CH, uint2* __restrict__ tile_sums) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  __shared__ uint32_t scratch[17];
  const uint32_t nb = 1u << lb, tile = min(nb, kPlanTile);
  const uint32_t* bsz = bucket_size + (size_t)blockIdx.y * nb + (size_t)blockIdx.x * tile;
  uint32_t pts = 0, items = 0;
#pragma unroll 4
  for (uint32_t d = threadIdx.x; d < tile; d += blockDim.x) {
    const uint32_t v = bsz[d];
    pts += v;
    items += (v + CH - 1) / CH;
  }
  uint32_t tp, ti;
  (void)block_exclusive_scan(pts, scratch, &tp);
  (void)block_exclusive_scan(items, scratch, &ti);
  if (threadIdx.x == 0) tile_sums[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = make_uint2(tp, ti);
}

// grid = (tiles, W), block = front_threads, dynamic LDS = lds_plan_bytes(lb)
__global__ void __launch_bounds__(kSortThreads)
plan_kernel(const uint32_t* __restrict__ bucket_size, uint32_t lb, uint32_t CH, const uint2* __restrict__ tile_sums,
            uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ item_start,
            uint32_t* __restrict__ win_items) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nb = 1u << lb;
  const uint32_t tile = min(nb, kPlanTile);
  const uint32_t w = blockIdx.y, tiles = gridDim.x;
  uint32_t* tot = lds_u32;                                   // skewed: index i lives at i + (i >> 5)
  uint32_t* scratch = lds_u32 + tile + (tile >> 5) + 1;      // 17 words
  uint32_t carry_start = 0, carry_items = 0;
  if (tiles > 1) {   // a few dozen tiles at most: every thread adds them up itself
#pragma unroll 1
    for (uint32_t t = 0; t < blockIdx.x; ++t) {
      const uint2 v = tile_sums[(size_t)w * tiles + t];
      carry_start += v.x;
      carry_items += v.y;
    }
  }
  const size_t base = (size_t)w * nb + (size_t)blockIdx.x * tile;
  const uint32_t* bsz = bucket_size + base;
#pragma unroll 1
  for (uint32_t d = threadIdx.x; d < tile; d += blockDim.x) tot[d + (d >> 5)] = bsz[d];
  __syncthreads();
  window_scan_lds(tot, scratch, tile);
#pragma unroll 1
  for (uint32_t d = threadIdx.x; d < tile; d += blockDim.x) bucket_start[base + d] = carry_start + tot[d + (d >> 5)];
  __syncthreads();
#pragma unroll 1
  for (uint32_t d = threadIdx.x; d < tile; d += blockDim.x) tot[d + (d >> 5)] = (bsz[d] + CH - 1) / CH;
  __syncthreads();
  const uint32_t tile_items = window_scan_lds(tot, scratch, tile);
#pragma unroll 1
  for (uint32_t d = threadIdx.x; d < tile; d += blockDim.x) item_start[base + d] = carry_items + tot[d + (d >> 5)];
  if (threadIdx.x == 0 && blockIdx.x == tiles - 1) win_items[w] = carry_items + tile_items;
}

// ------------------------------------------------------------------------------------------------
// Work-item ordering.  The accumulate kernel gives one lane to one work item (a bucket, or a CH-point
// chunk of a large bucket).  Lanes of a wave finish together only if their items have the same length,
// and the longest items must start first, so items are counting-sorted by DESCENDING size
// (CH + 1 size classes).  Three small kernels over the buckets:
//   size_hist_kernel     LDS histogram of item sizes per workgroup -> one column of a (size class x workgroup) table
//   size_scan_kernel     one workgroup: exclusive scan of the table in (descending size, workgroup) order,
//                        exclusive scan of win_items (-> window base of item ids), total item count
//   size_scatter_kernel  every bucket takes its positions with LDS atomics relative to its workgroup's bases;
//                        no global atomics except one per workgroup for the list of split buckets
constexpr int kSizeThreads = 1024;
// buckets per workgroup = kSizeChunks * workgroup size: the (size class x workgroup) table that ONE workgroup scans
// (size_scan_kernel) shrinks by this factor -- at 2^24 points it had 513 x 960 entries and the scan took 1.0 ms
constexpr uint32_t kSizeChunks = 8;

__device__ __forceinline__ void bucket_items(uint32_t s, uint32_t CH, uint32_t* nfull, uint32_t* last) {
  // s points -> nfull items of CH points + (last ? one item of `last` points : none)
  *nfull = s / CH;
  *last = s - *nfull * CH;
}

__global__ void __launch_bounds__(kSizeThreads)
size_hist_kernel(const uint32_t* __restrict__ bucket_size, uint32_t total_buckets, uint32_t CH,
                 uint32_t* __restrict__ wg_bins /* [CH + 1 rows, row r = size class CH - r][gridDim.x] */) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  for (uint32_t i = threadIdx.x; i <= CH; i += blockDim.x) lds_u32[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (uint32_t k = 0; k < kSizeChunks; ++k) {   // a workgroup owns kSizeChunks * blockDim.x consecutive buckets
    const uint32_t b = (blockIdx.x * kSizeChunks + k) * blockDim.x + threadIdx.x;
    if (b < total_buckets) {
      uint32_t nfull, last;
      bucket_items(bucket_size[b], CH, &nfull, &last);
      if (nfull) atomicAdd(&lds_u32[CH], nfull);
      if (last) atomicAdd(&lds_u32[last], 1u);
    }
  }
  __syncthreads();
  // no global atomics: every workgroup owns one column of the table
  for (uint32_t i = threadIdx.x; i <= CH; i += blockDim.x)
    wg_bins[(size_t)(CH - i) * gridDim.x + blockIdx.x] = lds_u32[i];
}

// One workgroup of 1024 threads: exclusive scan of the flattened (descending size class, workgroup) table ->
// first order[] position of every (size class, workgroup); exclusive scan of win_items -> window bases.
__global__ void __launch_bounds__(1024)
size_scan_kernel(uint32_t* __restrict__ wg_bins, uint32_t table_len, uint32_t* __restrict__ win_items, uint32_t W,
                 PlanCounters* __restrict__ counters) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  __shared__ uint32_t scratch[17];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (table_len + blockDim.x - 1) / blockDim.x;
  const uint32_t first = t * per;
  uint32_t local = 0;
  for (uint32_t k = 0; k < per; ++k)
    if (first + k < table_len) local += wg_bins[first + k];
  uint32_t total;
  uint32_t run = block_exclusive_scan(local, scratch, &total);
  for (uint32_t k = 0; k < per; ++k) {
    if (first + k < table_len) {
      const uint32_t v = wg_bins[first + k];
      wg_bins[first + k] = run;
      run += v;
    }
  }
  if (t == 0) {
    counters->total_items = total;
    counters->multi_count = 0;
    counters->pad[0] = 0;   // deferred (big) split buckets, see combine_small_kernel
    counters->pad[1] = 0;   // work items accumulate_kernel_asm hands to accumulate_redo_kernel
  }
  uint32_t wv = (t < W) ? win_items[t] : 0u;
  uint32_t wtotal;
  uint32_t wex = block_exclusive_scan(wv, scratch, &wtotal);
  if (t < W) win_items[t] = wex;
}

__global__ void __launch_bounds__(kSizeThreads)
size_scatter_kernel(const uint32_t* __restrict__ bucket_size, uint32_t total_buckets, uint32_t CH,
                    const uint32_t* __restrict__ wg_base, uint2* __restrict__ order,
                    uint32_t* __restrict__ multi_list, PlanCounters* __restrict__ counters) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  extern __shared__ uint32_t lds_u32[];
  uint32_t* cnt = lds_u32;              // [CH + 1] local ranks
  uint32_t* base = lds_u32 + CH + 1;    // [CH + 1] first position of this workgroup per size class
  uint32_t* multi = lds_u32 + 2 * (CH + 1);   // [2]: local count of split buckets, reserved global base
  for (uint32_t i = threadIdx.x; i <= CH; i += blockDim.x) {
    cnt[i] = 0;
    base[i] = wg_base[(size_t)(CH - i) * gridDim.x + blockIdx.x];
  }
  if (threadIdx.x == 0) multi[0] = 0;
  __syncthreads();
  uint32_t split_b[kSizeChunks], split_slot[kSizeChunks];
#pragma unroll
  for (uint32_t k = 0; k < kSizeChunks; ++k) {
    const uint32_t b = (blockIdx.x * kSizeChunks + k) * blockDim.x + threadIdx.x;
    split_b[k] = 0xFFFFFFFFu;
    split_slot[k] = 0;
    if (b < total_buckets) {
      uint32_t nfull = 0, last = 0;
      bucket_items(bucket_size[b], CH, &nfull, &last);
      if (nfull) {
        const uint32_t pos = base[CH] + atomicAdd(&cnt[CH], nfull);
        for (uint32_t j = 0; j < nfull; ++j) order[pos + j] = make_uint2(b, j);
      }
      if (last) {
        const uint32_t pos = base[last] + atomicAdd(&cnt[last], 1u);
        order[pos] = make_uint2(b, nfull);
      }
      if (nfull + (last ? 1u : 0u) > 1u) {
        split_b[k] = b;
        split_slot[k] = atomicAdd(&multi[0], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) multi[1] = multi[0] ? atomicAdd(&counters->multi_count, multi[0]) : 0u;   // one per workgroup
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < kSizeChunks; ++k)
    if (split_b[k] != 0xFFFFFFFFu) multi_list[multi[1] + split_slot[k]] = split_b[k];
}

// ark_bn254::G1Affine {x: Fq, y: Fq, infinity: bool} = 72 bytes (8-byte aligned).
__global__ void __launch_bounds__(256)
ark_affine_to_affine_kernel(const uint8_t* __restrict__ in, uint32_t n, Affine* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)t * 72);
  Affine a;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a.x.v[i] = src[i];
    a.y.v[i] = src[8 + i];
  }
  if (src[16] & 0xFFu) {
    a.x = u256_zero();
    a.y = u256_zero();
  }
  store_affine(&out[t], a);
}

// Reference wire layout (8 x u32, most significant limb first) -> little-endian.
// words = number of 256-bit values.
__global__ void __launch_bounds__(256)
be32_to_le_kernel(const uint32_t* __restrict__ in, size_t words, uint32_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= words * 8) return;
  const size_t e = t >> 3;
  const uint32_t l = (uint32_t)(t & 7);
  out[e * 8 + l] = in[e * 8 + (7 - l)];
}


// ------------------------------------------------------------------------------------------------
static size_t lds_plan_bytes(uint32_t lb) {
  const size_t tile = std::min<size_t>((size_t)1 << lb, kPlanTile);
  return (tile + (tile >> 5) + 33) * 4;
}

int sort_set_attributes(const char** failed) {
  const int max_lds = 160 * 1024;
  struct { const void* fn; const char* name; } ks[] = {
      {(const void*)plan_kernel, "plan_kernel"},
      {(const void*)coarse_scatter_kernel<uint16_t, false>, "coarse_scatter_kernel<u16>"},
      {(const void*)coarse_scatter_kernel<uint32_t, false>, "coarse_scatter_kernel<u32>"},
      {(const void*)coarse_scatter_kernel<uint16_t, true>, "coarse_scatter_kernel<u16, ballot>"},
      {(const void*)coarse_scatter_kernel<uint32_t, true>, "coarse_scatter_kernel<u32, ballot>"},
      {(const void*)coarse_scatter_tiled_kernel<uint16_t>, "coarse_scatter_tiled_kernel<u16>"},
      {(const void*)coarse_scatter_tiled_kernel<uint32_t>, "coarse_scatter_tiled_kernel<u32>"},
      {(const void*)mid_scatter_tiled_kernel, "mid_scatter_tiled_kernel"},
      {(const void*)fine_sort_kernel<false>, "fine_sort_kernel"},
      {(const void*)fine_sort_kernel<true>, "fine_sort_kernel<ballot>"}};
  for (auto& k : ks) {
    if (hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess) {
      (void)hipGetLastError();
      *failed = k.name;
      return 1;
    }
  }
  return 0;
}

void launch_digits(hipStream_t st, const Plan& p, const u256* scalars, int scalars_mont, void* digits) {
  const dim3 grid((p.n_scalars + 255) / 256), block(256);
#define LAUNCH_DIGITS_C(D, CC)                                                                                      \
  hipLaunchKernelGGL((digits_kernel<D, CC>), grid, block, 0, st, scalars, p.n_scalars, p.c, p.W_digits, scalars_mont, \
                     (D*)digits)
  if (p.W_digits == 254 / p.c + 1) {   // (always so; the specialisations compute W from C)
    if (p.wide_digits && p.c == 17) { LAUNCH_DIGITS_C(uint32_t, 17); return; }
    if (p.wide_digits && p.c == 16) { LAUNCH_DIGITS_C(uint32_t, 16); return; }
    if (!p.wide_digits && p.c == 15) { LAUNCH_DIGITS_C(uint16_t, 15); return; }
    if (!p.wide_digits && p.c == 13) { LAUNCH_DIGITS_C(uint16_t, 13); return; }
  }
#undef LAUNCH_DIGITS_C
  if (p.wide_digits)
    hipLaunchKernelGGL(digits_kernel<uint32_t>, grid, block, 0, st, scalars, p.n_scalars, p.c, p.W_digits, scalars_mont,
                       (uint32_t*)digits);
  else
    hipLaunchKernelGGL(digits_kernel<uint16_t>, grid, block, 0, st, scalars, p.n_scalars, p.c, p.W_digits, scalars_mont,
                       (uint16_t*)digits);
}

void launch_sort(hipStream_t st, const Plan& p, const SortBuffers& b) {
  const uint32_t nhi = 1u << p.hb, nmid = 1u << p.mb, nfine = 1u << p.fb;
  const uint32_t fb1 = p.mb + p.fb;   // bits of the slot that pass 1 leaves in tmp_fine
  if (p.wide_digits)
    hipLaunchKernelGGL(coarse_hist_kernel<uint32_t>, dim3(p.Q, p.W), dim3(p.front_threads), nhi * 4, st,
                       (const uint32_t*)b.digits, p.n, fb1, nhi, p.chunk, b.coarse_cnt);
  else
    hipLaunchKernelGGL(coarse_hist_kernel<uint16_t>, dim3(p.Q, p.W), dim3(p.front_threads), nhi * 4, st,
                       (const uint16_t*)b.digits, p.n, fb1, nhi, p.chunk, b.coarse_cnt);
  hipLaunchKernelGGL(coarse_prefix_kernel, dim3(p.W), dim3(1024), 0, st, b.coarse_cnt, p.Q, nhi, b.region_start);
  // ranking: bit 0 of p.ballot = coarse / middle scatter passes, bit 1 = pass 2 (see wave_claim)
  const bool bc = (p.ballot & 1u) != 0, bf = (p.ballot & 2u) != 0;
#define LAUNCH_COARSE_SCATTER(D, B)                                                                                 \
  hipLaunchKernelGGL((coarse_scatter_kernel<D, B>), dim3(p.Q, p.W), dim3(p.front_threads), nhi * 4, st,             \
                     (const D*)b.digits, p.n, p.hb, fb1, p.chunk, (const uint32_t*)b.coarse_cnt, b.tmp_idx, b.tmp_fine)
  const bool tiled = p.tiled && p.front_threads == kSortThreads && nhi <= kTileMaxRegions && nmid <= kTileMaxRegions;
  if (tiled) {
    if (p.wide_digits)
      hipLaunchKernelGGL(coarse_scatter_tiled_kernel<uint32_t>, dim3(p.Q, p.W), dim3(p.tile_threads), kTileLdsBytes, st,
                         (const uint32_t*)b.digits, p.n, p.hb, fb1, p.chunk, (const uint32_t*)b.coarse_cnt, b.tmp_idx,
                         b.tmp_fine);
    else
      hipLaunchKernelGGL(coarse_scatter_tiled_kernel<uint16_t>, dim3(p.Q, p.W), dim3(p.tile_threads), kTileLdsBytes, st,
                         (const uint16_t*)b.digits, p.n, p.hb, fb1, p.chunk, (const uint32_t*)b.coarse_cnt, b.tmp_idx,
                         b.tmp_fine);
  } else if (p.wide_digits) {
    if (bc) LAUNCH_COARSE_SCATTER(uint32_t, true); else LAUNCH_COARSE_SCATTER(uint32_t, false);
  } else {
    if (bc) LAUNCH_COARSE_SCATTER(uint16_t, true); else LAUNCH_COARSE_SCATTER(uint16_t, false);
  }
#undef LAUNCH_COARSE_SCATTER
  const uint32_t* fine_idx = b.tmp_idx;
  const uint16_t* fine_fine = b.tmp_fine;
  const uint32_t* fine_regions = b.region_start;
  uint32_t fine_nhi = nhi;
  if (p.mb) {   // three-level sort: split every coarse region once more
    hipLaunchKernelGGL(mid_hist_kernel, dim3(p.Q2, nhi * p.W), dim3(p.front_threads), nmid * 4, st,
                       (const uint16_t*)b.tmp_fine, p.n, nhi, p.fb, nmid, (const uint32_t*)b.region_start, b.mid_cnt);
    hipLaunchKernelGGL(mid_prefix_kernel, dim3(nhi * p.W), dim3(1024), 0, st, b.mid_cnt, p.Q2, nhi, nmid,
                       (const uint32_t*)b.region_start, b.region_start2);
    if (tiled)
      hipLaunchKernelGGL(mid_scatter_tiled_kernel, dim3(p.Q2, nhi * p.W), dim3(p.tile_threads), kTileLdsBytes, st,
                         (const uint32_t*)b.tmp_idx, (const uint16_t*)b.tmp_fine, p.n, nhi, p.fb, p.mb,
                         (const uint32_t*)b.region_start, (const uint32_t*)b.mid_cnt, b.tmp_idx2, b.tmp_fine2);
    else if (bc)
      hipLaunchKernelGGL(mid_scatter_kernel<true>, dim3(p.Q2, nhi * p.W), dim3(p.front_threads), nmid * 4, st,
                         (const uint32_t*)b.tmp_idx, (const uint16_t*)b.tmp_fine, p.n, nhi, p.fb, nmid,
                         (const uint32_t*)b.region_start, (const uint32_t*)b.mid_cnt, b.tmp_idx2, b.tmp_fine2);
    else
      hipLaunchKernelGGL(mid_scatter_kernel<false>, dim3(p.Q2, nhi * p.W), dim3(p.front_threads), nmid * 4, st,
                         (const uint32_t*)b.tmp_idx, (const uint16_t*)b.tmp_fine, p.n, nhi, p.fb, nmid,
                         (const uint32_t*)b.region_start, (const uint32_t*)b.mid_cnt, b.tmp_idx2, b.tmp_fine2);
    fine_idx = b.tmp_idx2;
    fine_fine = b.tmp_fine2;
    fine_regions = b.region_start2;
    fine_nhi = nhi * nmid;
  }
  if (bf)
    hipLaunchKernelGGL(fine_sort_kernel<true>, dim3(fine_nhi, p.W), dim3(std::max(p.front_threads, nfine)),
                       (kFineCap + nfine + 32) * 4, st, fine_idx, fine_fine, p.n, p.lb, p.fb, fine_regions, b.sorted,
                       b.bucket_size);
  else
    hipLaunchKernelGGL(fine_sort_kernel<false>, dim3(fine_nhi, p.W), dim3(std::max(p.front_threads, nfine)),
                       (kFineCap + nfine + 32) * 4, st, fine_idx, fine_fine, p.n, p.lb, p.fb, fine_regions, b.sorted,
                       b.bucket_size);
  const unsigned tiles = (unsigned)((p.nb + kPlanTile - 1) / kPlanTile);
  if (tiles > 1)
    hipLaunchKernelGGL(plan_tile_sums_kernel, dim3(tiles, p.W), dim3(p.front_threads), 0, st,
                       (const uint32_t*)b.bucket_size, p.lb, p.CH, b.tile_sums);
  hipLaunchKernelGGL(plan_kernel, dim3(tiles, p.W), dim3(p.front_threads), lds_plan_bytes(p.lb), st,
                     (const uint32_t*)b.bucket_size, p.lb, p.CH, (const uint2*)b.tile_sums, b.bucket_start,
                     b.item_start, b.win_items);
  const unsigned size_threads = p.front_threads;
  const unsigned gb = (unsigned)((p.total_buckets + size_threads * kSizeChunks - 1) / (size_threads * kSizeChunks));
  hipLaunchKernelGGL(size_hist_kernel, dim3(gb), dim3(size_threads), (p.CH + 1) * 4, st,
                     (const uint32_t*)b.bucket_size, (uint32_t)p.total_buckets, p.CH, b.size_bins);
  hipLaunchKernelGGL(size_scan_kernel, dim3(1), dim3(1024), 0, st, b.size_bins, (p.CH + 1) * gb, b.win_items, p.W,
                     b.counters);
  hipLaunchKernelGGL(size_scatter_kernel, dim3(gb), dim3(size_threads), (2 * (p.CH + 1) + 2) * 4, st,
                     (const uint32_t*)b.bucket_size, (uint32_t)p.total_buckets, p.CH, (const uint32_t*)b.size_bins, b.order,
                     b.multi_list, b.counters);
}

void launch_be32_to_le(hipStream_t st, const uint32_t* in, size_t words, uint32_t* out) {
  hipLaunchKernelGGL(be32_to_le_kernel, dim3((unsigned)((words * 8 + 255) / 256)), dim3(256), 0, st, in, words, out);
}

void launch_ark_affine_to_affine(hipStream_t st, const uint8_t* in, uint32_t n, Affine* out) {
  hipLaunchKernelGGL(ark_affine_to_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, n, out);
}

}  // namespace msm_amd
