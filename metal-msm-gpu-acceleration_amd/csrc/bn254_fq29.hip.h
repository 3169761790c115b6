// Carry-free internal representation of BN254 Fq for the hot kernels: 9 limbs x 29 bits, lazily reduced.
//
// Why (measured on MI355X, tools/microbench/fq_mul_bench.hip): with 8 x 32-bit limbs every limb product
// needs v_mad_u64_u32 + v_addc_co_u32 chained through VCC, and field add/sub are 8-long VCC carry chains
// with a conditional correction.  At the 3 waves/SIMD the accumulate kernel gets, those dependent chains
// cost ~1.4 k cycles per multiplication and ~245 cycles per addition.  With 29-bit limbs
//   * a column of 9 + 9 products (< 2^64) is accumulated by plain v_mad_u64_u32 -- no carry instruction,
//     17 independent column accumulators, so the compiler can interleave freely;
//   * add is 9 independent v_add_u32; sub is "a + K - b" with K a multiple of p whose limbs were lifted
//     by 2^30 / 2^31 so that no limb goes negative -- no borrow chain, no conditional;
//   * carries are only propagated inside the multiplication (which re-normalises its output anyway) and
//     by `norm`, one parallel round of shift/mask/add.
// The price: 81 + 81 instead of 64 + 64 limb products, and values are only bounded (not canonical), so
// zero tests use a one-limb filter with an exact slow path.
//
// Internal Montgomery radix is rho = 2^261.  The external form (host libraries, reference wire format,
// bn254_fq.hip.h) is Montgomery with R = 2^256 on 8 x u32; from_ext / to_ext convert (one internal
// multiplication each).  Constants come from tools/gen_fq29_constants.py.
//
// Bounds contract (p/rho = 0.0059):
//   mul/sqr operands : every limb <= 2^30 + 2^8, value <= ~40 p   (then all column sums stay < 2^64)
//   mul/sqr result   : limbs 0..7 < 2^29 exactly, limb 8 = carry; value < p * (alpha*beta*0.0059 + 1)
//   sub<K>(a, b)     : b limbs <= lift(K) - 2^(e-29), b value within the top-limb headroom of K;
//                      result = a - b + k*p, limbs < 2^32 -- NOT a valid mul operand until norm()
//   norm(a)          : limbs 0..7 < 2^29 + 8, value unchanged
#pragma once
#include "bn254_fq.hip.h"

namespace msm_amd {

struct fe29 {
  uint32_t l[9];
};

// Pins a limb as an opaque 32-bit VGPR value (no instruction is emitted).  LLVM hoists the zero-extension of a
// limb towards its definition; when definition and multiplication end up in different basic blocks (every
// multiplication after the exceptional-case branch of an addition, every loop-carried accumulator limb) the
// instruction selector no longer knows that the upper half of the 64-bit operand is zero and multiplies
// 64 x 32 bits: two v_mad_u64_u32 and two v_mov per limb product instead of one.  Measured in the accumulate
// kernel's ISA before this pin: 1243 instead of 1143 multiplier instructions and 200 moves per mixed addition.
MSM_HD uint32_t limb32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MSM_FQ29_NOPIN)   // NOPIN: A/B builds of tools/microbench only
  asm("" : "+v"(x));
#endif
  return x;
}
MSM_HD fe29 pin_limbs(const fe29& a) {
  fe29 r;
#if defined(__HIP_DEVICE_COMPILE__)
  _Pragma("unroll") for (int i = 0; i < 9; ++i) r.l[i] = limb32(a.l[i]);
#else
  r = a;
#endif
  return r;
}

// A comment line in the generated assembly (no instruction): tools/isa_counts.py tallies the instructions between
// marks to get the per-addition instruction counts of the shipped kernel (bench.py's second roofline).  Marks sit at
// the first / last statement of a basic block, where they cannot hold back the scheduler.
// (A mark tied to data -- an in/out operand of the empty asm -- would stay put, but it also costs the accumulate kernel
// 18 VGPRs: measured 207 instead of 189.  Plain comments leave the code as it is; tools/isa_counts.py copes with them
// drifting inside their block.)
#if defined(__HIP_DEVICE_COMPILE__)
#define MSM_ISA_MARK(name) asm volatile("; MSM_MARK " name)
#else
#define MSM_ISA_MARK(name) ((void)0)
#endif

enum KSel { K4E30 = 0, K8E30 = 1, K8E31 = 2, K16E30 = 3, K16E31 = 4 };

struct Fq29 {
  static constexpr uint32_t MASK = 0x1FFFFFFFu;
  static constexpr uint32_t INV = 0x04866389u;    // -p^-1 mod 2^29
  static constexpr uint32_t PINV = 0x1B799C77u;   //  p^-1 mod 2^29

  MSM_HD static constexpr uint32_t p(int i) {
    constexpr uint32_t c[9] = {0x187CFD47u, 0x010460B6u, 0x1C72A34Fu, 0x02D522D0u, 0x1585D978u,
                               0x02DB40C0u, 0x00A6E141u, 0x0E5C2634u, 0x0030644Eu};
    return c[i];
  }
  MSM_HD static constexpr uint32_t one_c(int i) {   // rho mod p
    constexpr uint32_t c[9] = {0x157CCC21u, 0x141C2758u, 0x185230D3u, 0x014C0419u, 0x0AA36FB9u,
                               0x1D4240CEu, 0x11D54C07u, 0x052AC7A8u, 0x000DC836u};
    return c[i];
  }
  MSM_HD static constexpr uint32_t cin_c(int i) {   // 2^(2*261-256) mod p : external -> internal
    constexpr uint32_t c[9] = {0x13349CA1u, 0x1A5D84A8u, 0x0A3E5CACu, 0x100249E0u, 0x12B951E8u,
                               0x0E92D304u, 0x14CB95B3u, 0x041B9D3Du, 0x00058003u};
    return c[i];
  }
  MSM_HD static constexpr uint32_t dout_c(int i) {  // 2^256 mod p : internal -> external
    constexpr uint32_t c[9] = {0x058F0D9Du, 0x1AEA1C6Eu, 0x11C2CF74u, 0x11D651EBu, 0x1462C0A7u,
                               0x11B7BC3Cu, 0x1CBD99BAu, 0x183340FBu, 0x000E0A77u};
    return c[i];
  }
  // k*p with limbs lifted by 2^e (see header): K4E30, K8E30, K8E31, K16E30, K16E31
  MSM_HD static constexpr uint32_t kc(int sel, int i) {
    constexpr uint32_t c[5][9] = {
        {0x41F3F51Cu, 0x441182D9u, 0x51CA8D3Au, 0x4B548B41u, 0x561765DEu, 0x4B6D0300u, 0x429B8502u, 0x597098CEu, 0x00C19137u},
        {0x43E7EA38u, 0x482305B4u, 0x43951A76u, 0x56A91685u, 0x4C2ECBBEu, 0x56DA0603u, 0x45370A06u, 0x52E1319Eu, 0x01832271u},
        {0x83E7EA38u, 0x882305B2u, 0x83951A74u, 0x96A91683u, 0x8C2ECBBCu, 0x96DA0601u, 0x85370A04u, 0x92E1319Cu, 0x0183226Fu},
        {0x47CFD470u, 0x50460B6Au, 0x472A34EEu, 0x4D522D0Cu, 0x585D977Fu, 0x4DB40C08u, 0x4A6E140Fu, 0x45C2633Eu, 0x030644E5u},
        {0x87CFD470u, 0x90460B68u, 0x872A34ECu, 0x8D522D0Au, 0x985D977Du, 0x8DB40C06u, 0x8A6E140Du, 0x85C2633Cu, 0x030644E3u}};
    return c[sel][i];
  }

  MSM_HD static fe29 zero() {
    fe29 r;
    MSM_UNROLL for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
  }
  MSM_HD static fe29 one() {
    fe29 r;
    MSM_UNROLL for (int i = 0; i < 9; ++i) r.l[i] = one_c(i);
    return r;
  }
  MSM_HD static bool is_zero_limbs(const fe29& a) {   // exact zero limbs (identity marker in memory)
    uint32_t o = 0;
    MSM_UNROLL for (int i = 0; i < 9; ++i) o |= a.l[i];
    return o == 0;
  }

  MSM_HD static fe29 add(const fe29& a, const fe29& b) {
    fe29 r;
    MSM_UNROLL for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
  }

  template <int SEL>
  MSM_HD static fe29 sub(const fe29& a, const fe29& b) {
    fe29 r;
    MSM_UNROLL for (int i = 0; i < 9; ++i) r.l[i] = (a.l[i] + kc(SEL, i)) - b.l[i];
    return r;
  }

  // -a (mod p) for a with limbs <= 2^30 - 2 and value < ~3.9 p; result < 4 p, limbs 0..7 < 2^29 + 8.
  MSM_HD static fe29 neg(const fe29& a) { return norm(sub<K4E30>(zero(), a)); }
  // The same without the carry round: limbs up to 2^30.5.  Only as ONE operand of a multiplication whose other
  // operand is normalised (every use is checked by tools/fq29_bounds.py).
  MSM_HD static fe29 neg_wide(const fe29& a) { return sub<K4E30>(zero(), a); }

  // One parallel carry round: limbs 0..7 < 2^29 + 8 afterwards, limb 8 absorbs the top carry.
  MSM_HD static fe29 norm(const fe29& a) {
    fe29 r;
    r.l[0] = a.l[0] & MASK;
    MSM_UNROLL for (int i = 1; i < 8; ++i) r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
  }

  // Montgomery reduction of 17 column sums (columns of weight 2^(29k)) modulo p with radix 2^261.
  MSM_HD static fe29 reduce_columns(uint64_t (&A)[17]) {
    uint64_t carry = 0;
    MSM_UNROLL for (int k = 0; k < 9; ++k) {
      A[k] += carry;
      const uint32_t m = ((uint32_t)A[k] * INV) & MASK;
      MSM_UNROLL for (int j = 0; j < 9; ++j) A[k + j] += (uint64_t)m * p(j);
      carry = A[k] >> 29;
    }
    fe29 r;
    MSM_UNROLL for (int k = 9; k < 17; ++k) {
      A[k] += carry;
      r.l[k - 9] = (uint32_t)A[k] & MASK;
      carry = A[k] >> 29;
    }
    r.l[8] = (uint32_t)carry;
    return r;
  }


  // The multiplication forms that were built, measured and NOT shipped (product scanning, lockstep chains, one
  // Karatsuba level: DESIGN.md / HISTORY.md) live in experiments/fq29_variants.inc; they are members of this struct only
  // in -DMSM_AMD_EXPERIMENTS builds (tools/build_variant.sh exp, the microbenchmarks, tests/test_gpu_experiments.py).
#if defined(MSM_AMD_EXPERIMENTS)
#include "experiments/fq29_variants.inc"
#endif

  // a*b*rho^-1 mod p (lazily reduced).  81 + 81 limb products, no carry instructions.
  // PIN = false (mul_np / mul2_np / sqr_np below): the caller has pinned the operands itself, ONCE per basic block
  // (pin_limbs on a value that is used again later costs a v_mov per limb: the empty asm's output is a new value, the
  // old one must survive beside it.  Pinning inside every multiplication cost the mixed addition 81 v_mov.)
  template <bool PIN = true>
  MSM_HD static fe29 mul(const fe29& a_in, const fe29& b_in) {
    const fe29 a = PIN ? pin_limbs(a_in) : a_in, b = PIN ? pin_limbs(b_in) : b_in;
#if defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_FIPS)
    {
      const fe29* const xs[1] = {&a};
      const fe29* const ys[1] = {&b};
      return fips<1>(xs, ys);
    }
#endif
    uint64_t A[17];
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      uint64_t s = 0;
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        if (j >= 0 && j < 9) s += (uint64_t)a.l[i] * b.l[j];
      }
      A[k] = s;
    }
    return reduce_columns(A);
  }

  // (a*b + c*d)*rho^-1 with ONE Montgomery reduction: 81 + 81 + 81 limb products instead of 2 x (81 + 81).
  // All four operands must be normalised (limbs <= 2^29 + 8) so that a column of 18 + 9 products stays
  // below 2^64.  Used for Y3 = R*T - Y1*PPP with d = -PPP.
  template <bool PIN = true>
  MSM_HD static fe29 mul2(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in) {
    const fe29 a = PIN ? pin_limbs(a_in) : a_in, b = PIN ? pin_limbs(b_in) : b_in, c = PIN ? pin_limbs(c_in) : c_in,
               d = PIN ? pin_limbs(d_in) : d_in;
#if defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_FIPS)
    {
      const fe29* const xs[2] = {&a, &c};
      const fe29* const ys[2] = {&b, &d};
      return fips<2>(xs, ys);
    }
#endif
    uint64_t A[17];
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      uint64_t s = 0;
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        if (j >= 0 && j < 9) {
          s += (uint64_t)a.l[i] * b.l[j];
          s += (uint64_t)c.l[i] * d.l[j];
        }
      }
      A[k] = s;
    }
    return reduce_columns(A);
  }

  // a*a*rho^-1: 45 + 81 limb products (cross products use the doubled operand).
  template <bool PIN = true>
  MSM_HD static fe29 sqr(const fe29& a_in) {
    const fe29 a = PIN ? pin_limbs(a_in) : a_in;
#if defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_FIPS)
    {
      fe29 d2;
      MSM_UNROLL for (int i = 0; i < 9; ++i) d2.l[i] = a.l[i] << 1;
      const fe29* const xs[1] = {&d2};
      const fe29* const ys[1] = {&a};
      return fips<1, true>(xs, ys);
    }
#endif
    uint32_t d[9];
    MSM_UNROLL for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;   // operand limbs <= 2^30 + 2^8 -> < 2^32
    uint64_t A[17];
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      uint64_t s = 0;
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        if (j >= 0 && j < 9 && i < j) s += (uint64_t)d[i] * a.l[j];
      }
      if ((k & 1) == 0) s += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
      A[k] = s;
    }
    return reduce_columns(A);
  }
  MSM_HD static fe29 mul_np(const fe29& a, const fe29& b) { return mul<false>(a, b); }
  MSM_HD static fe29 sqr_np(const fe29& a) { return sqr<false>(a); }
  MSM_HD static fe29 mul2_np(const fe29& a, const fe29& b, const fe29& c, const fe29& d) { return mul2<false>(a, b, c, d); }

  // 256-bit little-endian integer -> 9 x 29-bit limbs (pure bit slicing, value unchanged).
  MSM_HD static fe29 unpack256(const u256& x) {
    fe29 t;
    MSM_UNROLL for (int i = 0; i < 9; ++i) {
      const int bit = 29 * i;
      const int w = bit >> 5, s = bit & 31;
      uint32_t v = x.v[w] >> s;
      if (s > 3 && w + 1 < 8) v |= x.v[w + 1] << (32 - s);
      t.l[i] = (i < 8) ? (v & MASK) : v;
    }
    return t;
  }

  // Canonical limbs (each < 2^29, value < 2^256) -> 256-bit little-endian integer.
  MSM_HD static u256 pack256(const fe29& t) {
    u256 r;
    MSM_UNROLL for (int w = 0; w < 8; ++w) {
      const int bit = 32 * w;
      const int i = bit / 29, s = bit % 29;
      uint32_t v = t.l[i] >> s;
      if (i + 1 < 9) v |= t.l[i + 1] << (29 - s);
      if (29 - s + 29 < 32 && i + 2 < 9) v |= t.l[i + 2] << (58 - s);
      r.v[w] = v;
    }
    return r;
  }

  // External (8 x u32 little-endian, Montgomery R = 2^256, canonical) -> internal.
  MSM_HD static fe29 from_ext(const u256& x) {
    fe29 c;
    MSM_UNROLL for (int i = 0; i < 9; ++i) c.l[i] = cin_c(i);
    return mul(unpack256(x), c);
  }

  // Exact carry propagation + at most `rounds` conditional subtractions of p; input limbs arbitrary u32
  // with value < (rounds + 1) * p.  Returns canonical limbs (each < 2^29, value < p).
  MSM_HD static fe29 canonical(const fe29& a, int rounds) {
    fe29 r;
    uint32_t carry = 0;
    MSM_UNROLL for (int i = 0; i < 8; ++i) {
      const uint64_t s = (uint64_t)a.l[i] + carry;
      r.l[i] = (uint32_t)s & MASK;
      carry = (uint32_t)(s >> 29);
    }
    r.l[8] = a.l[8] + carry;
    for (int it = 0; it < rounds; ++it) {
      fe29 d;
      int32_t borrow = 0;
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int32_t s = (int32_t)r.l[i] - (int32_t)p(i) + borrow;
        d.l[i] = (i < 8) ? ((uint32_t)s & MASK) : (uint32_t)s;
        borrow = (i < 8) ? (s >> 29) : 0;   // arithmetic shift: 0 or -1
        if (i == 8) borrow = (s < 0) ? -1 : 0;
      }
      if (borrow == 0) r = d;
    }
    return r;
  }

  // Internal (lazy; limbs must be valid mul operands) -> external canonical Montgomery (R = 2^256).
  MSM_HD static u256 to_ext(const fe29& a) {
    fe29 d;
    MSM_UNROLL for (int i = 0; i < 9; ++i) d.l[i] = dout_c(i);
    return pack256(canonical(mul(a, d), 1));   // mul output < 1.3 p
  }

  // Internal (a multiplication result: exact limbs, value < 2 p) -> canonical value packed into 32 bytes, still
  // in the internal Montgomery domain; unpack256 restores limbs.  Storage format of the gathered bases.
  MSM_HD static u256 pack_canonical(const fe29& a) { return pack256(canonical(a, 1)); }

  // Cheap necessary condition for a == 0 (mod p) given value(a) < bound * p: if a = j*p then the low 29
  // bits satisfy j = a0 * p^-1 mod 2^29 < bound.  Low limb bits are exact even for lazy limbs.
  MSM_HD static bool maybe_zero(const fe29& a, uint32_t bound) {
    return (((a.l[0] & MASK) * PINV) & MASK) < bound;
  }

  // Exact test (slow path): squash through one multiplication by rho mod p, canonicalise, compare.
  MSM_HD static bool is_zero_exact(const fe29& a) {
    const fe29 t = canonical(mul(a, one()), 1);
    return is_zero_limbs(t);
  }
};

}  // namespace msm_amd
