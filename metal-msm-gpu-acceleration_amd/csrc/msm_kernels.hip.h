// Pippenger MSM pipeline kernels for BN254 G1 on gfx950 (MI355X).  Included by msm_host.hip.
//
// Pipeline (one MSM of n points, window c bits, W = ceil(254/c) windows, nb = 2^c digit values):
//
//   digits_kernel        scalars (32 B, Montgomery or canonical)  -> digits[W][n] (u16, SoA)
//   hist_kernel          digits -> counts[W][Q][nb]     LDS histogram per (chunk q, window w)
//   chunk_prefix_kernel  counts -> bucket_size[W][nb], counts := exclusive prefix over chunks
//   scan_kernel          bucket_size -> bucket_start[W][nb]  (per-window exclusive scan in LDS)
//   scatter_kernel       digits + cursors -> sorted[W][n]  (point indices grouped by digit)
//   accumulate_kernel    sorted + bases(affine 64 B) -> buckets[W][nb]  (Jacobian 96 B)   <- dominant
//   reduce_seg_kernel    buckets -> S[W][nseg], T[W][nseg]   (segments of 8 buckets)
//   reduce_tree_kernel   S, T -> partial[W][K+1]  (one plain sum + K bit-subset sums per window)
//   host                 Horner over bit positions of the (K+1)*W partial points
//
// This replaces the reference's prepare_buckets_indices / sort_buckets (CPU rayon sort!) /
// bucket_wise_accumulation / sum_reduction_partial+final kernels (src/metal/shader/msm.h.metal:17-562,
// src/metal/msm/sort_buckets.rs:15-34) with a design derived for wave64 + 160 KB LDS:
//   * the sort is a per-window counting sort whose whole digit histogram (2^15 x u32 = 128 KB) lives in
//     LDS, so ranking is LDS atomics and the only global traffic is digits in / indices out;
//   * sorted output is 4 B point indices plus per-bucket offsets (the reference sorts 8 B pairs and
//     then binary-searches bucket boundaries per threadgroup, msm.h.metal:61-73,130-131);
//   * window sums use running sums over 8-bucket segments followed by bit-subset tree sums, which
//     needs no scalar multiplications (the reference multiplies sums by counts with double-and-add in
//     every combine, msm.h.metal:429-430).
#pragma once
#include <hip/hip_runtime.h>
#include "bn254_ec.hip.h"

namespace msm_amd {

constexpr int kSegLog = 3;               // window reduction: segments of 2^3 buckets
constexpr int kSeg = 1 << kSegLog;
constexpr int kSortThreads = 1024;       // hist / scan / scatter workgroup size

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

// ------------------------------------------------------------------------------------------------
// Stage 1: digit extraction.  Replaces kernel prepare_buckets_indices (msm.h.metal:17-59, one thread
// per threadgroup and a generic 256-bit shift per window) and the CPU de-Montgomery of scalars
// (limbs_conversion.rs:282-288).  scalars_mont: 1 = host Montgomery form (bn256::Fr / ark Fr memory),
// 0 = canonical integer.
__global__ void __launch_bounds__(256)
digits_kernel(const u256* __restrict__ scalars, uint32_t n, uint32_t c, uint32_t W, int scalars_mont,
              uint16_t* __restrict__ digits) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u256 k = load_u256(&scalars[t]);
  if (scalars_mont) k = Fr::from_mont(k);
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t d = u256_extract_bits(k, w * c, c);
    digits[(size_t)w * n + t] = (uint16_t)d;
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 2a: LDS histogram of one chunk of one window.  grid = (Q, W), block = kSortThreads,
// dynamic LDS = nb * 4 bytes.
__global__ void __launch_bounds__(kSortThreads)
hist_kernel(const uint16_t* __restrict__ digits, uint32_t n, uint32_t c, uint32_t chunk,
            uint32_t* __restrict__ counts) {
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nb = 1u << c;
  const uint32_t q = blockIdx.x, Q = gridDim.x, w = blockIdx.y;
  for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lds_u32[i] = 0;
  __syncthreads();
  const uint32_t lo = q * chunk;
  const uint32_t hi = min(n, lo + chunk);
  const uint16_t* dw = digits + (size_t)w * n;
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
    const uint32_t d = dw[t];
    if (d) atomicAdd(&lds_u32[d], 1u);
  }
  __syncthreads();
  uint32_t* out = counts + ((size_t)w * Q + q) * nb;
  for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) out[i] = lds_u32[i];
}

// Block-wide exclusive scan of one value per thread (blockDim.x <= 1024, multiple of 64).
// Returns the exclusive prefix; *total receives the block sum.  scratch: >= 17 words of LDS.
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

// Stage 2b: per-bucket totals.  One thread per (window, digit): turns counts[w][q][d] into the exclusive
// prefix over chunks q (position of chunk q's first element inside the bucket) and writes the bucket size.
__global__ void __launch_bounds__(256)
chunk_prefix_kernel(uint32_t* __restrict__ counts, uint32_t c, uint32_t Q, uint32_t W,
                    uint32_t* __restrict__ bucket_size) {
  const uint32_t nb = 1u << c;
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= W * nb) return;
  const uint32_t w = b >> c, d = b & (nb - 1);
  uint32_t* cw = counts + (size_t)w * Q * nb + d;
  uint32_t run = 0;
  for (uint32_t q = 0; q < Q; ++q) {
    const uint32_t cnt = cw[(size_t)q * nb];
    cw[(size_t)q * nb] = run;
    run += cnt;
  }
  bucket_size[b] = run;
}

// Stage 2c: per-window exclusive scan of bucket sizes -> bucket_start[w][d] (offset inside the window's
// slice of `sorted`).  grid = W, block = kSortThreads, dynamic LDS = (nb + nb/32 + 32) * 4 bytes.
__global__ void __launch_bounds__(kSortThreads)
scan_kernel(const uint32_t* __restrict__ bucket_size, uint32_t c, uint32_t* __restrict__ bucket_start) {
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nb = 1u << c;
  const uint32_t w = blockIdx.x;
  uint32_t* tot = lds_u32;                               // skewed: index i lives at i + (i >> 5)
  uint32_t* scratch = lds_u32 + nb + (nb >> 5) + 1;      // 17 words
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) tot[d + (d >> 5)] = bucket_size[(size_t)w * nb + d];
  __syncthreads();
  const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;   // consecutive entries per thread
  const uint32_t first = threadIdx.x * per;
  uint32_t local = 0;
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t d = first + j;
    if (d < nb) local += tot[d + (d >> 5)];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(local, scratch, &total);
  for (uint32_t j = 0; j < per; ++j) {
    const uint32_t d = first + j;
    if (d < nb) {
      const uint32_t s = tot[d + (d >> 5)];
      tot[d + (d >> 5)] = run;
      run += s;
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nb; d += blockDim.x) bucket_start[(size_t)w * nb + d] = tot[d + (d >> 5)];
}

// Stage 2d: scatter point indices to their bucket slots.  grid = (Q, W), dynamic LDS = nb * 4 bytes.
// Order inside a bucket is unspecified (LDS atomic arrival order), exactly as the reference allows
// (sort_buckets.rs:111-125 checks only multiset + non-decreasing keys).
__global__ void __launch_bounds__(kSortThreads)
scatter_kernel(const uint16_t* __restrict__ digits, uint32_t n, uint32_t c, uint32_t chunk,
               const uint32_t* __restrict__ chunk_prefix, const uint32_t* __restrict__ bucket_start,
               uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds_u32[];
  const uint32_t nb = 1u << c;
  const uint32_t q = blockIdx.x, Q = gridDim.x, w = blockIdx.y;
  const uint32_t* rel = chunk_prefix + ((size_t)w * Q + q) * nb;
  const uint32_t* bs = bucket_start + (size_t)w * nb;
  for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lds_u32[i] = bs[i] + rel[i];
  __syncthreads();
  const uint32_t lo = q * chunk;
  const uint32_t hi = min(n, lo + chunk);
  const uint16_t* dw = digits + (size_t)w * n;
  uint32_t* sw = sorted + (size_t)w * n;
  for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
    const uint32_t d = dw[t];
    if (d) {
      const uint32_t pos = atomicAdd(&lds_u32[d], 1u);
      sw[pos] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stage 3: bucket accumulation -- the dominant kernel.  One lane per bucket; each lane walks its
// bucket's slice of `sorted`, gathers the 64-byte affine base and performs a mixed Jacobian+affine
// addition (7M+4S).  Replaces kernel bucket_wise_accumulation (msm.h.metal:75-315).
// `order` (optional) maps launch slot -> bucket id so that lanes of one wave get buckets of similar
// size (see bucket_order kernels); nullptr = identity mapping.
__global__ void __launch_bounds__(64)
accumulate_kernel(const Affine* __restrict__ bases, const uint32_t* __restrict__ sorted,
                  const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ bucket_size,
                  const uint32_t* __restrict__ order, uint32_t n, uint32_t c, uint32_t total_buckets,
                  Jacobian* __restrict__ buckets) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= total_buckets) return;
  const uint32_t b = order ? order[slot] : slot;
  const uint32_t w = b >> c;
  const uint32_t cnt = bucket_size[b];
  const uint32_t* idx = sorted + (size_t)w * n + bucket_start[b];
  Jacobian acc = jac_identity();
  if (cnt) {
    Affine nxt = load_affine(&bases[idx[0]]);
#pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) {
      const Affine cur = nxt;
      if (i + 1 < cnt) nxt = load_affine(&bases[idx[i + 1]]);
      if (!affine_is_identity(cur)) acc = jac_madd(acc, cur);
    }
  }
  store_jac(&buckets[b], acc);
}

// ------------------------------------------------------------------------------------------------
// Stage 4a: per-segment running sums.  For segment s of window w (buckets d = 8s .. 8s+7):
//   S[w][s] = sum_j X[8s+j]          T[w][s] = sum_j j * X[8s+j]
// so that  sum_d d*X[d] = sum_s T[s] + 8 * sum_s s*S[s].   Replaces sum_reduction_partial
// (msm.h.metal:319-461), whose combine step needs a scalar multiplication per merge.
__global__ void __launch_bounds__(64)
reduce_seg_kernel(const Jacobian* __restrict__ buckets, uint32_t total_segs,
                  Jacobian* __restrict__ S, Jacobian* __restrict__ T) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total_segs) return;
  const Jacobian* X = buckets + (size_t)s * kSeg;
  Jacobian sum = jac_identity(), sos = jac_identity();
#pragma unroll 1
  for (int j = kSeg - 1; j >= 1; --j) {
    sum = jac_add(sum, load_jac(&X[j]));
    sos = jac_add(sos, sum);
  }
  sum = jac_add(sum, load_jac(&X[0]));
  store_jac(&S[s], sum);
  store_jac(&T[s], sos);
}

// Stage 4b: tree sums.  grid = (K + 1, W) with K = c - 3 bits of segment index; block = tree_threads
// (power of two, 64..1024); dynamic LDS = tree_threads * 96 bytes.
//   blockIdx.x == K : partial[w][K] = sum_s T[w][s]
//   blockIdx.x  < K : partial[w][k] = sum over segments s with bit k set of S[w][s]
// The host then evaluates  W_w = partial[w][K] + 8 * sum_k 2^k partial[w][k]  inside one Horner pass
// over all bit positions (replaces sum_reduction_final msm.h.metal:463-562 and the doublings of
// final_accumulation.rs:19-39).
__global__ void __launch_bounds__(1024)
reduce_tree_kernel(const Jacobian* __restrict__ S, const Jacobian* __restrict__ T, uint32_t nseg,
                   uint32_t K, Jacobian* __restrict__ partial) {
  extern __shared__ uint32_t lds_u32[];
  Jacobian* sh = reinterpret_cast<Jacobian*>(lds_u32);
  const uint32_t k = blockIdx.x, w = blockIdx.y;
  const Jacobian* Sw = S + (size_t)w * nseg;
  const Jacobian* Tw = T + (size_t)w * nseg;
  Jacobian acc = jac_identity();
  if (k == K) {
#pragma unroll 1
    for (uint32_t s = threadIdx.x; s < nseg; s += blockDim.x) acc = jac_add(acc, load_jac(&Tw[s]));
  } else {
    const uint32_t half = nseg >> 1;
    const uint32_t lowmask = (1u << k) - 1u;
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < half; j += blockDim.x) {
      const uint32_t s = ((j & ~lowmask) << 1) | (1u << k) | (j & lowmask);
      acc = jac_add(acc, load_jac(&Sw[s]));
    }
  }
  store_jac(&sh[threadIdx.x], acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t stride = blockDim.x >> 1; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      const Jacobian a = load_jac(&sh[threadIdx.x]);
      const Jacobian b2 = load_jac(&sh[threadIdx.x + stride]);
      store_jac(&sh[threadIdx.x], jac_add(a, b2));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_jac(&partial[(size_t)w * (K + 1) + k], load_jac(&sh[0]));
}

// ------------------------------------------------------------------------------------------------
// Input conversion kernels (the reference converts on the CPU with rayon, state.rs:88-109).

// ark_bn254::G1Projective (x, y, z Montgomery LE, 96 B) -> affine 64 B.  One thread per point with a
// Fermat inversion only when z is neither 0 nor one.
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

__global__ void __launch_bounds__(64)
projective_to_affine_kernel(const Jacobian* __restrict__ in, uint32_t n, Affine* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  store_affine(&out[t], jac_to_affine(load_jac(&in[t])));
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
  u256 raw = rnd256(seed, 1, i);
  raw.v[7] &= 0x3FFFFFFFu;
  return Fr::reduce_once(raw);
}

// scalars_mont: write scalars in Montgomery form (what bn256::Fr / ark Fr hold in memory) or canonical.
__global__ void __launch_bounds__(64)
gen_instance_kernel(uint64_t seed, uint32_t n, int scalars_mont, Affine* __restrict__ bases,
                    u256* __restrict__ scalars) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Affine pt;
  pt.x = u256_zero();
  pt.y = u256_zero();
#pragma unroll 1
  for (uint32_t attempt = 0; attempt < 64; ++attempt) {
    if (gen_point_attempt(seed, t, attempt, pt)) break;
  }
  store_affine(&bases[t], pt);
  u256 k = gen_scalar_canonical(seed, t);
  if (scalars_mont) k = Fr::to_mont(k);
  store_u256(&scalars[t], k);
}

}  // namespace msm_amd
