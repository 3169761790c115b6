// 256-bit Montgomery field arithmetic for BN254 (Fq and Fr), written for gfx950 (CDNA4).
//
// Replaces the reference's UnsignedInteger<8> (src/metal/shader/arithmetics/unsigned_int.h.metal:6-310)
// and FpBN254 (src/metal/shader/fields/fp_bn254.h.metal:48-291).  Differences by design:
//   * limbs are little-endian u32 (limb 0 least significant) so a field element in memory is
//     byte-identical to the host libraries' [u64;4] little-endian representation -- no limb
//     reordering at the boundary (the reference stores limb 0 = most significant);
//   * the multiplier is product-scanning Montgomery (FIPS): every 32x32 product is one
//     v_mad_u64_u32 (64-bit accumulate for free) + one v_addc_co_u32 for the third accumulator
//     word.  Measured on MI355X (profiles/r01_valu_rates_microbench.txt): v_mad_u64_u32 issues at
//     ~5.3 cycles per wave-instruction per SIMD, the same rate as v_fma_f64 and v_mul_lo_u32, so
//     the 32-bit-limb integer path is the right one on this chip (no FP64 limb tricks needed);
//   * R = 2^256 as in the reference, so Montgomery residues are bit-identical to the host libraries'.
//
// The file also compiles as plain C++ (g++) so the same arithmetic can be exercised by host-side
// unit tests and by the host final accumulation; the inline asm is only used in device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MSM_HD __host__ __device__ __forceinline__
#define MSM_UNROLL _Pragma("unroll")
#else
#define MSM_HD inline __attribute__((always_inline))
#define MSM_UNROLL
#endif

namespace msm_amd {

// ------------------------------------------------------------------------------------------------
// Field parameters.  mod(i) / one(i) / r2(i) are constexpr so that fully unrolled code sees literals.
struct FqParams {   // base field; constants match fp_bn254.h.metal:25-46 (N, R_SQUARED, MU)
  static constexpr uint32_t INV = 0xE4866389u;   // -p^-1 mod 2^32  (= MU 3834012553)
  MSM_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xD87CFD47u, 0x3C208C16u, 0x6871CA8Du, 0x97816A91u,
                               0x8181585Du, 0xB85045B6u, 0xE131A029u, 0x30644E72u};
    return m[i];
  }
  MSM_HD static constexpr uint32_t one(int i) {   // R mod p
    constexpr uint32_t m[8] = {0xC58F0D9Du, 0xD35D438Du, 0xF5C70B3Du, 0x0A78EB28u,
                               0x7879462Cu, 0x666EA36Fu, 0x9A07DF2Fu, 0x0E0A77C1u};
    return m[i];
  }
  MSM_HD static constexpr uint32_t r2(int i) {    // R^2 mod p
    constexpr uint32_t m[8] = {0x538AFA89u, 0xF32CFC5Bu, 0xD44501FBu, 0xB5E71911u,
                               0x0A417FF6u, 0x47AB1EFFu, 0xCAB8351Fu, 0x06D89F71u};
    return m[i];
  }
};

struct FrParams {   // scalar field
  static constexpr uint32_t INV = 0xEFFFFFFFu;
  MSM_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t m[8] = {0xF0000001u, 0x43E1F593u, 0x79B97091u, 0x2833E848u,
                               0x8181585Du, 0xB85045B6u, 0xE131A029u, 0x30644E72u};
    return m[i];
  }
  MSM_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t m[8] = {0x4FFFFFFBu, 0xAC96341Cu, 0x9F60CD29u, 0x36FC7695u,
                               0x7879462Eu, 0x666EA36Fu, 0x9A07DF2Fu, 0x0E0A77C1u};
    return m[i];
  }
  MSM_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t m[8] = {0xAE216DA7u, 0x1BB8E645u, 0xE35C59E3u, 0x53FE3AB1u,
                               0x53BB8085u, 0x8C49833Du, 0x7F4E44A5u, 0x0216D0B1u};
    return m[i];
  }
};

// ------------------------------------------------------------------------------------------------
// 256-bit unsigned integer, little-endian limbs.
struct u256 {
  uint32_t v[8];
};

MSM_HD u256 u256_zero() {
  u256 r;
  MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}

MSM_HD bool u256_is_zero(const u256& a) {
  uint32_t o = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}

MSM_HD bool u256_eq(const u256& a, const u256& b) {
  uint32_t o = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a + b, returns carry-out.
MSM_HD uint32_t u256_add(u256& r, const u256& a, const u256& b) {
  uint64_t c = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}

// r = a - b, returns borrow-out (1 if a < b).
MSM_HD uint32_t u256_sub(u256& r, const u256& a, const u256& b) {
  int64_t c = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) {
    c += (int64_t)a.v[i] - (int64_t)b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;   // arithmetic shift: 0 or -1
  }
  return (uint32_t)(c & 1);
}

// Low 256 bits of a * b (b is a 32-bit word).  Mirrors the operand shapes the reference's
// test_uint_prod kernel exercises (src/metal/tests/test_bn254.rs:128-141).
MSM_HD u256 u256_mul_u32(const u256& a, uint32_t b) {
  u256 r;
  uint64_t c = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] * b;
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

// Logical shifts by 0..255 bits (unsigned_int.h.metal operator<< / operator>>).
MSM_HD u256 u256_shl(const u256& a, uint32_t s) {
  u256 r;
  const uint32_t w = s >> 5, b = s & 31;
  MSM_UNROLL for (int i = 0; i < 8; ++i) {
    uint32_t lo = 0, hi = 0;
    MSM_UNROLL for (int j = 0; j < 8; ++j) {
      if ((uint32_t)j + w == (uint32_t)i) hi = a.v[j];
      if ((uint32_t)j + w + 1 == (uint32_t)i) lo = a.v[j];
    }
    r.v[i] = b ? ((hi << b) | (lo >> (32 - b))) : hi;
  }
  return r;
}

MSM_HD u256 u256_shr(const u256& a, uint32_t s) {
  u256 r;
  const uint32_t w = s >> 5, b = s & 31;
  MSM_UNROLL for (int i = 0; i < 8; ++i) {
    uint32_t lo = 0, hi = 0;
    MSM_UNROLL for (int j = 0; j < 8; ++j) {
      if ((uint32_t)i + w == (uint32_t)j) lo = a.v[j];
      if ((uint32_t)i + w + 1 == (uint32_t)j) hi = a.v[j];
    }
    r.v[i] = b ? ((lo >> b) | (hi << (32 - b))) : lo;
  }
  return r;
}

// Bits [start, start+width) of a, width <= 32, start+width may run past bit 255 (zero filled).
// This is the digit extraction of prepare_buckets_indices (msm.h.metal:38-48) without the generic
// 256-bit shift the reference performs per window.
MSM_HD uint32_t u256_extract_bits(const u256& a, uint32_t start, uint32_t width) {
  const uint32_t w = start >> 5, b = start & 31;
  uint32_t lo = 0, hi = 0;
  MSM_UNROLL for (int j = 0; j < 8; ++j) {
    if ((uint32_t)j == w) lo = a.v[j];
    if ((uint32_t)j == w + 1) hi = a.v[j];
  }
  uint64_t both = ((uint64_t)hi << 32) | lo;
  uint32_t x = (uint32_t)(both >> b);
  return width >= 32 ? x : (x & ((1u << width) - 1u));
}

// ------------------------------------------------------------------------------------------------
// Multiply-accumulate into a 96-bit column accumulator (lo: 64 bits, hi: 32 bits).
MSM_HD void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi)
      : "v"(a), "v"(b)
      : "vcc");
#else
  const uint64_t p = (uint64_t)a * b;
  lo += p;
  hi += (lo < p) ? 1u : 0u;
#endif
}

// Same, second factor is a compile-time constant kept in an SGPR (modulus limbs).
MSM_HD void mac96_k(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi)
      : "v"(a), "s"(k)
      : "vcc");
#else
  mac96(lo, hi, a, k);
#endif
}

// ------------------------------------------------------------------------------------------------
// Montgomery field over parameters F.  Elements are fully reduced: 0 <= x < p.
template <class F>
struct Field {
  MSM_HD static u256 modulus() {
    u256 r;
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = F::mod(i);
    return r;
  }
  MSM_HD static u256 one() {
    u256 r;
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = F::one(i);
    return r;
  }
  MSM_HD static u256 r2() {
    u256 r;
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = F::r2(i);
    return r;
  }

  // x >= p ? x - p : x      (x < 2p)
  MSM_HD static u256 reduce_once(const u256& x) {
    u256 d;
    const uint32_t borrow = u256_sub(d, x, modulus());
    u256 r;
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = borrow ? x.v[i] : d.v[i];
    return r;
  }

  MSM_HD static u256 add(const u256& a, const u256& b) {
    u256 s;
    u256_add(s, a, b);          // a + b < 2p < 2^255: no carry out
    return reduce_once(s);
  }

  MSM_HD static u256 sub(const u256& a, const u256& b) {
    u256 d, e;
    const uint32_t borrow = u256_sub(d, a, b);
    u256_add(e, d, modulus());
    u256 r;
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = borrow ? e.v[i] : d.v[i];
    return r;
  }

  MSM_HD static u256 dbl(const u256& a) { return add(a, a); }

  MSM_HD static u256 neg(const u256& a) {
    u256 d;
    u256_sub(d, modulus(), a);
    u256 r;
    const bool z = u256_is_zero(a);
    MSM_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = z ? 0u : d.v[i];
    return r;
  }

  // Montgomery product a*b*R^-1 mod p.
  // Device: product scanning with interleaved reduction (FIPS) on 32-bit limbs; replaces FpBN254::mul
  // (fp_bn254.h.metal:237-290, operand-scanning CIOS on MS-first limbs).
  // Host (final Horner pass, input conversion checks): the same value computed with 4 x 64-bit limbs and
  // unsigned __int128 (CIOS), about 3x faster on a CPU core; u256 is little-endian so the limbs alias.
  MSM_HD static u256 mul(const u256& a, const u256& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned __int128 u128;
    uint64_t x[4], y[4], pm[4];
    for (int i = 0; i < 4; ++i) {
      x[i] = ((uint64_t)a.v[2 * i + 1] << 32) | a.v[2 * i];
      y[i] = ((uint64_t)b.v[2 * i + 1] << 32) | b.v[2 * i];
      pm[i] = ((uint64_t)F::mod(2 * i + 1) << 32) | F::mod(2 * i);
    }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: inv64 = inv32 * (2 + p0 * inv32)
    // (with inv = -p^-1: x' = x * (2 + p * x) doubles the number of correct bits)
    const uint64_t inv32 = F::INV;
    const uint64_t inv64 = inv32 * (2 + pm[0] * inv32);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      u128 c = 0;
      for (int j = 0; j < 4; ++j) {
        c += (u128)x[j] * y[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * inv64;
      c = ((u128)m * pm[0] + t[0]) >> 64;
      for (int j = 1; j < 4; ++j) {
        c += (u128)m * pm[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    u256 r;
    for (int i = 0; i < 4; ++i) {
      r.v[2 * i] = (uint32_t)t[i];
      r.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once(r);   // t < 2p fits 256 bits (t[4] == 0)
#else
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    u256 r;
    MSM_UNROLL for (int k = 0; k < 8; ++k) {
      MSM_UNROLL for (int i = 0; i <= k; ++i) mac96(lo, hi, a.v[i], b.v[k - i]);
      MSM_UNROLL for (int i = 0; i < k; ++i) mac96_k(lo, hi, m[i], F::mod(k - i));
      m[k] = (uint32_t)lo * F::INV;
      mac96_k(lo, hi, m[k], F::mod(0));
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    MSM_UNROLL for (int k = 8; k < 15; ++k) {
      MSM_UNROLL for (int i = k - 7; i < 8; ++i) mac96(lo, hi, a.v[i], b.v[k - i]);
      MSM_UNROLL for (int i = k - 7; i < 8; ++i) mac96_k(lo, hi, m[i], F::mod(k - i));
      r.v[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    r.v[7] = (uint32_t)lo;
    return reduce_once(r);
#endif
  }

  MSM_HD static u256 sqr(const u256& a) { return mul(a, a); }

  MSM_HD static u256 to_mont(const u256& a) { return mul(a, r2()); }

  // a * R^-1 mod p: the Montgomery reduction alone (64 limb products instead of the 128 of mul(a, 1)); the
  // de-Montgomery of every scalar in digits_kernel.
  MSM_HD static u256 from_mont(const u256& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
    u256 o = u256_zero();
    o.v[0] = 1;
    return mul(a, o);
#else
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    u256 r;
    MSM_UNROLL for (int k = 0; k < 8; ++k) {
      mac96(lo, hi, a.v[k], 1u);
      MSM_UNROLL for (int i = 0; i < k; ++i) mac96_k(lo, hi, m[i], F::mod(k - i));
      m[k] = (uint32_t)lo * F::INV;
      mac96_k(lo, hi, m[k], F::mod(0));
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    MSM_UNROLL for (int k = 8; k < 15; ++k) {
      MSM_UNROLL for (int i = k - 7; i < 8; ++i) mac96_k(lo, hi, m[i], F::mod(k - i));
      r.v[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    r.v[7] = (uint32_t)lo;
    return reduce_once(r);
#endif
  }

  // a^e for a 32-bit exponent (fp_bn254.h.metal `pow`), square-and-multiply MSB first.
  MSM_HD static u256 pow_u32(const u256& a, uint32_t e) {
    u256 r = one();
    for (int i = 31; i >= 0; --i) {
      r = sqr(r);
      if ((e >> i) & 1u) r = mul(r, a);
    }
    return r;
  }
};

using Fq = Field<FqParams>;
using Fr = Field<FrParams>;

}  // namespace msm_amd
