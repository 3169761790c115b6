// Pippenger MSM pipeline for BN254 G1 on gfx950 (MI355X): overview + helpers shared by the kernel TUs
// (k_sort.hip, k_accumulate.hip, k_reduce.hip, k_misc.hip, k_stage.hip) and by the host driver.
//
// Pipeline (one MSM of n points, window c bits, W = floor(254/c) + 1 windows of SIGNED digits, nb = 2^(c-1)
// slots per window; slot i holds the points whose digit magnitude is i + 1):
//
//   front stream (k_sort.hip, k_misc.hip)
//   convert_bases_kernel  bases (affine 64 B, Montgomery R = 2^256) -> bases29 (64 B packed, internal domain 2^261)
//   digits_kernel         scalars (32 B, Montgomery or canonical) -> digits[W][n] (u16 = sign << 15 | magnitude)
//   coarse_hist_kernel    digits -> coarse_cnt[W][Q][2^hb]   LDS histogram of the high slot bits per (chunk, window)
//   coarse_prefix_kernel  coarse_cnt -> region_start[W][2^hb + 1], coarse_cnt := first position per (chunk, region)
//   coarse_scatter_kernel digits -> tmp_idx / tmp_fine, grouped by coarse region (per-workgroup LDS cursors)
//   fine_sort_kernel      one region per workgroup, counting sort by the low slot bits inside LDS
//                         -> sorted[W][n] (point index | sign << 31), bucket_size[W][nb]
//   plan_kernel           bucket_size -> bucket_start, item_start (work items of <= CH points), win_items
//   size_hist/scan/scatter  work items counting-sorted by descending length -> order[], list of split buckets
//
//   main stream (k_accumulate.hip)
//   accumulate_kernel     one lane per work item: sorted + bases29 -> buckets[W][nb] (XYZZ, 144 B) or
//                         item_partials for split buckets                                      <- dominant
//
//   reduce stream (k_accumulate.hip, k_reduce.hip)
//   combine_small/big     sum the partials of split buckets into buckets
//   sum_groups_kernel     buckets -> row sums R[W][2^H] and column sums C[W][2^L] of the slot matrix (slot = hi * 2^L
//                         + lo), in levels of groups of 16 (4..16 for a lone call): plain sums, every bucket added exactly twice
//   reduce_bits_kernel    R, C -> partial[W][lb + 1]  (bit-subset sums of C and of R + the window total, external
//                         Jacobian): window value = total + sum_k 2^k CB_k + 2^L sum_k 2^k RB_k
//   host                  Horner over the bit positions of the (lb + 1) * W partial points (msm_host.hip host_combine)
//
// This replaces the reference's prepare_buckets_indices / sort_buckets (CPU rayon sort!) /
// bucket_wise_accumulation / sum_reduction_partial+final kernels (src/metal/shader/msm.h.metal:17-562,
// src/metal/msm/sort_buckets.rs:15-34) with a design derived for wave64 + 160 KB LDS + eight non-coherent L2s:
//   * the sort is a two-pass MSD counting sort per window: pass 1 keeps every workgroup's stores on a few
//     open lines, pass 2 sorts a region inside LDS, so both passes write about their payload (a one-pass
//     scatter of 4-byte indices wrote 8x its payload as partial-line evictions);
//   * sorted output is 4 B point indices plus per-bucket offsets (the reference sorts 8 B pairs and
//     then binary-searches bucket boundaries per threadgroup, msm.h.metal:61-73,130-131);
//   * work is balanced by ordering work items by length, not by splitting pairs evenly over threads and
//     merging bucket boundaries through threadgroup memory (msm.h.metal:229-314);
//   * window sums use plain row / column sums of the slot matrix followed by bit-subset sums over them, which
//     needs no scalar multiplications (the reference multiplies sums by counts with double-and-add in
//     every combine, msm.h.metal:429-430);
//   * the three streams work on three different instances at any time (msm_host.hip enqueue_msm).
#pragma once
#include <hip/hip_runtime.h>
#include "bn254_ec29.hip.h"
#include "launch.h"

namespace msm_amd {


// ------------------------------------------------------------------------------------------------
// 16-byte vector loads/stores of field elements and points (coalescing unit is 16 B/lane).
__device__ __forceinline__ u256 load_u256(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  u256 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}

__device__ __forceinline__ void store_u256(void* p, const u256& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

__device__ __forceinline__ Affine load_affine(const Affine* p) {
  Affine r;
  r.x = load_u256(&p->x);
  r.y = load_u256(&p->y);
  return r;
}

__device__ __forceinline__ void store_affine(Affine* p, const Affine& a) {
  store_u256(&p->x, a.x);
  store_u256(&p->y, a.y);
}

#if defined(MSM_AMD_EXPERIMENTS)
// The wide record of a base (experiments/ec29_variants.inc, accumulate variant 8): stored as eight 16-byte pieces; read as x (9 words at the start of the
// 128-byte line) and ONE of y / -y (9 words at byte 36 or 72, picked by the sign bit of the sorted entry).
__device__ __forceinline__ void store_wide(AffWide* p, const AffWide& a) {
  const uint32_t* w = a.x;   // x, y, ny, pad are contiguous (static_assert on the size)
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
typedef uint32_t u32x4_dword_aligned __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load_wide(const AffWide* __restrict__ bases, uint32_t entry, AffI& q) {
  const uint32_t* r = reinterpret_cast<const uint32_t*>(bases + (entry & 0x7FFFFFFFu));
  const uint4 a = *reinterpret_cast<const uint4*>(r), b = *reinterpret_cast<const uint4*>(r + 4);
  q.x.l[0] = a.x; q.x.l[1] = a.y; q.x.l[2] = a.z; q.x.l[3] = a.w;
  q.x.l[4] = b.x; q.x.l[5] = b.y; q.x.l[6] = b.z; q.x.l[7] = b.w;
  q.x.l[8] = r[8];
  const uint32_t* ry = r + 9 + 9 * (entry >> 31);   // negative digit: the stored -y
  const u32x4_dword_aligned c = *reinterpret_cast<const u32x4_dword_aligned*>(ry);
  const u32x4_dword_aligned d = *reinterpret_cast<const u32x4_dword_aligned*>(ry + 4);
  q.y.l[0] = c.x; q.y.l[1] = c.y; q.y.l[2] = c.z; q.y.l[3] = c.w;
  q.y.l[4] = d.x; q.y.l[5] = d.y; q.y.l[6] = d.z; q.y.l[7] = d.w;
  q.y.l[8] = ry[8];
}
__device__ __forceinline__ bool wide_is_identity(const AffI& q) { return q.x.l[8] == 0xFFFFFFFFu; }
#endif

__device__ __forceinline__ Jacobian load_jac(const Jacobian* p) {
  Jacobian r;
  r.x = load_u256(&p->x);
  r.y = load_u256(&p->y);
  r.z = load_u256(&p->z);
  return r;
}

__device__ __forceinline__ void store_jac(Jacobian* p, const Jacobian& a) {
  store_u256(&p->x, a.x);
  store_u256(&p->y, a.y);
  store_u256(&p->z, a.z);
}

// Block-wide exclusive scan of one value per thread (blockDim.x <= 1024, multiple of 64).
// Returns the exclusive prefix; *total receives the block sum.  scratch: >= 17 words of LDS.
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= (uint32_t)off) incl += up;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t ws = lane < nwaves ? scratch[lane] : 0u;
    uint32_t wi = ws;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t up = __shfl_up(wi, off, 64);
      if (lane >= (uint32_t)off) wi += up;
    }
    if (lane < nwaves) scratch[lane] = wi - ws;   // exclusive wave offsets
    if (lane == nwaves - 1) scratch[16] = wi;     // block total
  }
  __syncthreads();
  const uint32_t res = scratch[wave] + incl - v;
  *total = scratch[16];
  __syncthreads();
  return res;
}

#endif

#if defined(__HIPCC__)
// 16-byte vector loads/stores of the internal-representation points (64 B packed affine, 144 B XYZZ).
template <int QUADS>
__device__ __forceinline__ void load_quads(const void* p, uint32_t* dst) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < QUADS; ++i) {
    const uint4 v = q[i];
    dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
  }
}
template <int QUADS>
__device__ __forceinline__ void store_quads(void* p, const uint32_t* src) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < QUADS; ++i) q[i] = make_uint4(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}
__device__ __forceinline__ AffI load_affi(const AffPacked* p) {
  AffPacked q;
  q.x = load_u256(&p->x);
  q.y = load_u256(&p->y);
  return affi_unpack(q);
}
__device__ __forceinline__ void store_affi(AffPacked* p, const AffI& a) {
  const AffPacked q = affi_pack(a);
  store_u256(&p->x, q.x);
  store_u256(&p->y, q.y);
}
__device__ __forceinline__ PtI load_pti(const PtI* p) {
  uint32_t w[36];
  load_quads<9>(p, w);
  PtI r;
#pragma unroll
  for (int i = 0; i < 9; ++i) { r.x.l[i] = w[i]; r.y.l[i] = w[9 + i]; r.zz.l[i] = w[18 + i]; r.zzz.l[i] = w[27 + i]; }
  return r;
}
__device__ __forceinline__ void store_pti(PtI* p, const PtI& a) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; ++i) { w[i] = a.x.l[i]; w[9 + i] = a.y.l[i]; w[18 + i] = a.zz.l[i]; w[27 + i] = a.zzz.l[i]; }
  store_quads<9>(p, w);
}
#endif

// ------------------------------------------------------------------------------------------------
// Inversion / normalisation (host + device).
MSM_HD u256 fq_inverse(const u256& a) {
  // a^(p-2): exponent limbs of p - 2
  u256 e = Fq::modulus();
  e.v[0] -= 2u;   // p is odd and its low limb > 2
  u256 r = Fq::one();
  for (int i = 255; i >= 0; --i) {
    r = Fq::sqr(r);
    if ((e.v[i >> 5] >> (i & 31)) & 1u) r = Fq::mul(r, a);
  }
  return r;
}

MSM_HD Affine jac_to_affine(const Jacobian& p) {
  Affine r;
  if (jac_is_identity(p)) {
    r.x = u256_zero();
    r.y = u256_zero();
    return r;
  }
  if (u256_eq(p.z, Fq::one())) {
    r.x = p.x;
    r.y = p.y;
    return r;
  }
  const u256 zi = fq_inverse(p.z);
  const u256 zi2 = Fq::sqr(zi);
  r.x = Fq::mul(p.x, zi2);
  r.y = Fq::mul(p.y, Fq::mul(zi2, zi));
  return r;
}

// ------------------------------------------------------------------------------------------------
// Deterministic synthetic instance generator; bit-for-bit the generator of oracle/bn254_ref.py
// (gen_point / gen_scalar) and oracle/msm_oracle.c.  Plays the role of the reference's random
// instance generation (src/utils/preprocess.rs:113-138) for benchmarks and large parity tests.
MSM_HD uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

MSM_HD uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t ctr) {
  return splitmix64(splitmix64(seed ^ (stream << 56)) + ctr);
}

MSM_HD u256 rnd256(uint64_t seed, uint64_t stream, uint64_t ctr4) {
  u256 r;
  MSM_UNROLL for (int k = 0; k < 4; ++k) {
    const uint64_t w = rnd64(seed, stream, ctr4 * 4 + k);
    r.v[2 * k] = (uint32_t)w;
    r.v[2 * k + 1] = (uint32_t)(w >> 32);
  }
  return r;
}

MSM_HD u256 fq_sqrt_candidate(const u256& a) {
  // a^((p+1)/4), p = 3 mod 4.  (p+1)/4 little-endian limbs:
  const uint32_t e[8] = {0xB61F3F52u, 0x4F082305u, 0x5A1C72A3u, 0x65E05AA4u,
                         0xA0605617u, 0x6E14116Du, 0xB84C680Au, 0x0C19139Cu};
  u256 r = Fq::one();
  for (int i = 251; i >= 0; --i) {
    r = Fq::sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = Fq::mul(r, a);
  }
  return r;
}

// Returns false if this attempt does not yield a point.
MSM_HD bool gen_point_attempt(uint64_t seed, uint64_t i, uint32_t attempt, Affine& out) {
  u256 raw = rnd256(seed, 0, i * 64 + attempt);
  const uint32_t sign = raw.v[7] >> 31;
  raw.v[7] &= 0x3FFFFFFFu;                         // 254 bits
  u256 d;
  if (u256_sub(d, raw, Fq::modulus()) == 0) return false;   // x >= p
  const u256 x = Fq::to_mont(raw);
  u256 three = u256_zero();
  three.v[0] = 3;
  const u256 rhs = Fq::add(Fq::mul(Fq::sqr(x), x), Fq::to_mont(three));
  u256 y = fq_sqrt_candidate(rhs);
  if (!u256_eq(Fq::sqr(y), rhs)) return false;
  if (sign) y = Fq::neg(y);
  out.x = x;
  out.y = y;
  return true;
}

MSM_HD u256 gen_scalar_canonical(uint64_t seed, uint64_t i) {
  // rejection sampling: uniform mod r (254 random bits per attempt, accept if < r)
  u256 raw = u256_zero();
  for (uint32_t attempt = 0; attempt < 16; ++attempt) {
    raw = rnd256(seed, 1, i * 16 + attempt);
    raw.v[7] &= 0x3FFFFFFFu;
    u256 d;
    if (u256_sub(d, raw, Fr::modulus()) != 0) return raw;   // borrow: raw < r
  }
  return Fr::reduce_once(raw);
}

}  // namespace msm_amd
