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


  // ---- product-scanning ("FIPS") forms: one running 64-bit accumulator per column, whose chain STARTS from the
  // carry of the column below (the free 64-bit addend of the first v_mad_u64_u32), so no 64-bit additions are
  // needed to propagate carries, and no 17 column sums are live at once.  Same values as mul / sqr / mul2.  The
  // multiply-adds are written as inline assembly because LLVM re-associates a chain of 64-bit additions back
  // into independent partial sums plus 64-bit adds.
  MSM_HD static void mad64(uint64_t& t, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t co;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=s"(co) : "v"(x), "v"(y));
#else
    t += (uint64_t)x * y;
#endif
  }
  MSM_HD static void mad64c(uint64_t& t, uint32_t x, uint32_t c) {   // c: a constant limb of p (scalar register)
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t co;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=s"(co) : "v"(x), "s"(c));
#else
    t += (uint64_t)x * c;
#endif
  }
  // SQUARE: one product, x = 2a (doubled limbs), y = a: cross products i < j only, plus a_(k/2)^2 in even columns
  template <int NPROD, bool SQUARE = false>
  MSM_HD static fe29 fips(const fe29* const (&x)[NPROD], const fe29* const (&y)[NPROD]) {
    uint32_t m[9];
    fe29 r;
    uint64_t t = 0;
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      MSM_UNROLL for (int q = 0; q < NPROD; ++q) {
        MSM_UNROLL for (int i = 0; i < 9; ++i) {
          const int j = k - i;
          if (j >= 0 && j < 9 && (!SQUARE || i < j)) mad64(t, x[q]->l[i], y[q]->l[j]);
        }
        if (SQUARE && (k & 1) == 0) mad64(t, y[q]->l[k >> 1], y[q]->l[k >> 1]);
      }
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        if (i < k && j >= 1 && j < 9) mad64c(t, m[i], p(j));
      }
      if (k < 9) {
        m[k] = ((uint32_t)t * INV) & MASK;
        mad64c(t, m[k], p(0));
      } else {
        r.l[k - 9] = (uint32_t)t & MASK;
      }
      t >>= 29;
    }
    r.l[8] = (uint32_t)t;
    return r;
  }

  // Two (or three) INDEPENDENT products in lockstep, product-scanning form: column k of every job is accumulated
  // before column k + 1 of any, with the multiply-adds of the jobs alternating instruction by instruction.  One
  // product-scanning chain is a string of dependent v_mad_u64_u32 (8.5 cycles from one to the next when a wave is
  // alone: fq29_bench, 1741 cycles per multiplication at 1 wave/SIMD against 1107 at 4); two chains per wave and two
  // waves per SIMD give the issue logic four independent streams -- what fips<> alone only gets at 4 waves/SIMD,
  // which the accumulate kernel cannot have.  Against the column-parallel mul() this saves the 64-bit carry addition
  // of every column (the chain starts FROM the carry) and the 17 x 2 live column registers.
  // Job q: value = sum over its NPROD products x[q][r] * y[q][r]; SQUARE jobs pass x = 2a, y = a.
  struct FipsJob {
    const fe29* x[2];
    const fe29* y[2];
  };
  template <int NJOBS, int NP0, bool SQ0, int NP1, bool SQ1, int NP2 = 0, bool SQ2 = false>
  MSM_HD static void fips_multi(const FipsJob (&job)[NJOBS], fe29 (&r)[NJOBS]) {
    constexpr int np[3] = {NP0, NP1, NP2};
    constexpr bool sq[3] = {SQ0, SQ1, SQ2};
    uint32_t m[NJOBS][9];
    uint64_t t[NJOBS];
    MSM_UNROLL for (int q = 0; q < NJOBS; ++q) t[q] = 0;
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        MSM_UNROLL for (int q = 0; q < NJOBS; ++q) {
          MSM_UNROLL for (int pr = 0; pr < 2; ++pr) {
            if (pr < np[q] && j >= 0 && j < 9 && (!sq[q] || i < j)) mad64(t[q], job[q].x[pr]->l[i], job[q].y[pr]->l[j]);
          }
        }
      }
      MSM_UNROLL for (int q = 0; q < NJOBS; ++q) {
        MSM_UNROLL for (int pr = 0; pr < 2; ++pr) {
          if (pr < np[q] && sq[q] && (k & 1) == 0) mad64(t[q], job[q].y[pr]->l[k >> 1], job[q].y[pr]->l[k >> 1]);
        }
      }
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        MSM_UNROLL for (int q = 0; q < NJOBS; ++q) {
          if (i < k && j >= 1 && j < 9) mad64c(t[q], m[q][i], p(j));
        }
      }
      MSM_UNROLL for (int q = 0; q < NJOBS; ++q) {
        if (k < 9) {
          m[q][k] = ((uint32_t)t[q] * INV) & MASK;
          mad64c(t[q], m[q][k], p(0));
        } else {
          r[q].l[k - 9] = (uint32_t)t[q] & MASK;
        }
        t[q] >>= 29;
      }
    }
    MSM_UNROLL for (int q = 0; q < NJOBS; ++q) r[q].l[8] = (uint32_t)t[q];
  }
  // (a * b, c * d) -- two multiplications side by side
  MSM_HD static void mul_pair(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in, fe29& ab, fe29& cd) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in), c = pin_limbs(c_in), d = pin_limbs(d_in);
    const FipsJob jobs[2] = {{{&a, nullptr}, {&b, nullptr}}, {{&c, nullptr}, {&d, nullptr}}};
    fe29 r[2];
    fips_multi<2, 1, false, 1, false>(jobs, r);
    ab = r[0];
    cd = r[1];
  }
  // (a^2, b^2)
  MSM_HD static void sqr_pair(const fe29& a_in, const fe29& b_in, fe29& aa, fe29& bb) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in);
    fe29 a2, b2;
    MSM_UNROLL for (int i = 0; i < 9; ++i) {
      a2.l[i] = a.l[i] << 1;
      b2.l[i] = b.l[i] << 1;
    }
    const FipsJob jobs[2] = {{{&a2, nullptr}, {&a, nullptr}}, {{&b2, nullptr}, {&b, nullptr}}};
    fe29 r[2];
    fips_multi<2, 1, true, 1, true>(jobs, r);
    aa = r[0];
    bb = r[1];
  }
  // (a * b + c * d, e * f) -- the shared-reduction double product next to a plain multiplication
  MSM_HD static void mul2_mul_pair(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in, const fe29& e_in,
                                   const fe29& f_in, fe29& abcd, fe29& ef) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in), c = pin_limbs(c_in), d = pin_limbs(d_in), e = pin_limbs(e_in),
               f = pin_limbs(f_in);
    const FipsJob jobs[2] = {{{&a, &c}, {&b, &d}}, {{&e, nullptr}, {&f, nullptr}}};
    fe29 r[2];
    fips_multi<2, 2, false, 1, false>(jobs, r);
    abcd = r[0];
    ef = r[1];
  }
  // (a * b, c * d, e * f) -- three multiplications side by side
  MSM_HD static void mul_triple(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in, const fe29& e_in,
                                const fe29& f_in, fe29& ab, fe29& cd, fe29& ef) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in), c = pin_limbs(c_in), d = pin_limbs(d_in), e = pin_limbs(e_in),
               f = pin_limbs(f_in);
    const FipsJob jobs[3] = {{{&a, nullptr}, {&b, nullptr}}, {{&c, nullptr}, {&d, nullptr}}, {{&e, nullptr}, {&f, nullptr}}};
    fe29 r[3];
    fips_multi<3, 1, false, 1, false, 1, false>(jobs, r);
    ab = r[0];
    cd = r[1];
    ef = r[2];
  }

  // a*b*rho^-1 mod p (lazily reduced).  81 + 81 limb products, no carry instructions.
  MSM_HD static fe29 mul(const fe29& a_in, const fe29& b_in) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in);
#if defined(MSM_FQ29_FIPS)
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

  // One Karatsuba level over 3-limb blocks (a = A0 + A1 B + A2 B^2, B = 2^87): six 3 x 3 block products (54
  // multiply-adds) instead of nine (81), paid for with 18 limb additions and 30 64-bit column subtractions
  //   A0B0, A1B1, A2B2,  (A0+A1)(B0+B1) - A0B0 - A1B1,  (A0+A2)(B0+B2) - A0B0 - A2B2,  (A1+A2)(B1+B2) - A1B1 - A2B2
  // Every column of a bracketed product dominates the same column of what is subtracted from it (the difference is
  // the column of the cross terms), so the subtractions never borrow.  Operands must be normalised (limbs <= 2^29 + 8:
  // a block sum then stays below 2^30 + 16 and a 3-term column below 2^62).
  MSM_HD static void karatsuba_columns(const fe29& a, const fe29& b, uint64_t (&A)[17]) {
    uint64_t P[3][5];   // A_i * B_i
    MSM_UNROLL for (int blk = 0; blk < 3; ++blk) {
      MSM_UNROLL for (int c = 0; c < 5; ++c) {
        uint64_t s = 0;
        MSM_UNROLL for (int i = 0; i < 3; ++i) {
          const int j = c - i;
          if (j >= 0 && j < 3) s += (uint64_t)a.l[3 * blk + i] * b.l[3 * blk + j];
        }
        P[blk][c] = s;
      }
    }
    MSM_UNROLL for (int c = 0; c < 5; ++c) {   // A is ACCUMULATED into: the caller zeroes it or has another product there
      A[c] += P[0][c];
      A[12 + c] += P[2][c];
    }
    // cross terms of blocks (x, y) land at columns 3 (x + y) ..; the middle one starts from A1B1's columns
    MSM_UNROLL for (int pr = 0; pr < 3; ++pr) {
      const int x = pr == 2 ? 1 : 0, y = pr == 0 ? 1 : 2;
      uint32_t sa[3], sb[3];
      MSM_UNROLL for (int i = 0; i < 3; ++i) {
        sa[i] = limb32(a.l[3 * x + i] + a.l[3 * y + i]);
        sb[i] = limb32(b.l[3 * x + i] + b.l[3 * y + i]);
      }
      MSM_UNROLL for (int c = 0; c < 5; ++c) {
        uint64_t s = (pr == 1) ? P[1][c] : 0;
        MSM_UNROLL for (int i = 0; i < 3; ++i) {
          const int j = c - i;
          if (j >= 0 && j < 3) s += (uint64_t)sa[i] * sb[j];
        }
        s -= P[x][c];
        s -= P[y][c];
        A[3 * (x + y) + c] += s;
      }
    }
  }
  MSM_HD static fe29 mul_karatsuba(const fe29& a_in, const fe29& b_in) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in);
    uint64_t A[17];
    MSM_UNROLL for (int k = 0; k < 17; ++k) A[k] = 0;
    karatsuba_columns(a, b, A);
    return reduce_columns(A);
  }
  // a * b (schoolbook: b may be an un-normalised difference with limbs < 2^31) + c * d (Karatsuba), one reduction
  MSM_HD static fe29 mul2_karatsuba_second(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in), c = pin_limbs(c_in), d = pin_limbs(d_in);
    uint64_t A[17];
    MSM_UNROLL for (int k = 0; k < 17; ++k) {
      uint64_t s = 0;
      MSM_UNROLL for (int i = 0; i < 9; ++i) {
        const int j = k - i;
        if (j >= 0 && j < 9) s += (uint64_t)a.l[i] * b.l[j];
      }
      A[k] = s;
    }
    karatsuba_columns(c, d, A);
    return reduce_columns(A);
  }

  // (a*b + c*d)*rho^-1 with ONE Montgomery reduction: 81 + 81 + 81 limb products instead of 2 x (81 + 81).
  // All four operands must be normalised (limbs <= 2^29 + 8) so that a column of 18 + 9 products stays
  // below 2^64.  Used for Y3 = R*T - Y1*PPP with d = -PPP.
  MSM_HD static fe29 mul2(const fe29& a_in, const fe29& b_in, const fe29& c_in, const fe29& d_in) {
    const fe29 a = pin_limbs(a_in), b = pin_limbs(b_in), c = pin_limbs(c_in), d = pin_limbs(d_in);
#if defined(MSM_FQ29_FIPS)
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
  MSM_HD static fe29 sqr(const fe29& a_in) {
    const fe29 a = pin_limbs(a_in);
#if defined(MSM_FQ29_FIPS)
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
