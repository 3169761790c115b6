// Host-only BN254 Fq / G1 arithmetic on 4 x 64-bit limbs (unsigned __int128 products) for the CPU tail of every
// MSM: the Horner pass over the window partial sums (host_combine, ~254 doublings + ~255 additions) and the final
// normalisation.  The portable 8 x 32-bit code of bn254_fq.hip.h / bn254_ec.hip.h -- written for GPU lanes and
// shared with the host for the unit tests -- spends ~28 ns per field multiplication on a Zen core; this one ~9 ns.
// For a LONE 2^18-point call the Horner pass was 0.17 ms of 0.93 ms.
//
// Same value representation as the external form everywhere else (Montgomery, R = 2^256, canonical, little-endian),
// so u256 / Jacobian records are reinterpreted in place (little-endian hosts only, checked below).
// Replaces the arithmetic under final_accumulation.rs:19-39 (host-side window Horner of the reference).
#pragma once
#include <cstdlib>
#include <cstdint>
#include <cstring>

#include "bn254_ec.hip.h"

#if !defined(__BYTE_ORDER__) || __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "host_fq64.h reinterprets 8 x u32 little-endian limbs as 4 x u64: little-endian hosts only"
#endif

namespace msm_amd {
namespace h64 {

typedef unsigned __int128 u128;

struct Fe {
  uint64_t v[4];
};
struct Jac {   // same 96-byte layout as Jacobian
  Fe x, y, z;
};
static_assert(sizeof(Jac) == sizeof(Jacobian), "layout");

constexpr uint64_t P[4] = {0x3C208C16D87CFD47ull, 0x97816A916871CA8Dull, 0xB85045B68181585Dull, 0x30644E72E131A029ull};
constexpr uint64_t NINV = 0x87D20782E4866389ull;   // -p^-1 mod 2^64

inline bool is_zero(const Fe& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }

// t - p if t >= p else t, branch-free (t < 2^255)
inline void reduce_once(uint64_t t[4]) {
  uint64_t d[4];
  u128 b = (u128)t[0] - P[0];
  d[0] = (uint64_t)b;
  b = (u128)t[1] - P[1] - (uint64_t)((b >> 64) & 1u);
  d[1] = (uint64_t)b;
  b = (u128)t[2] - P[2] - (uint64_t)((b >> 64) & 1u);
  d[2] = (uint64_t)b;
  b = (u128)t[3] - P[3] - (uint64_t)((b >> 64) & 1u);
  d[3] = (uint64_t)b;
  const uint64_t keep = (uint64_t)0 - (uint64_t)((b >> 64) & 1u);   // all ones when t < p
  t[0] = (t[0] & keep) | (d[0] & ~keep);
  t[1] = (t[1] & keep) | (d[1] & ~keep);
  t[2] = (t[2] & keep) | (d[2] & ~keep);
  t[3] = (t[3] & keep) | (d[3] & ~keep);
}

inline Fe add(const Fe& a, const Fe& b) {   // a, b < p < 2^254: the sum fits 255 bits
  Fe r;
  u128 c = (u128)a.v[0] + b.v[0];
  r.v[0] = (uint64_t)c;
  c = (u128)a.v[1] + b.v[1] + (uint64_t)(c >> 64);
  r.v[1] = (uint64_t)c;
  c = (u128)a.v[2] + b.v[2] + (uint64_t)(c >> 64);
  r.v[2] = (uint64_t)c;
  r.v[3] = a.v[3] + b.v[3] + (uint64_t)(c >> 64);
  reduce_once(r.v);
  return r;
}
inline Fe dbl(const Fe& a) { return add(a, a); }

inline Fe sub(const Fe& a, const Fe& b) {
  Fe r;
  u128 d = (u128)a.v[0] - b.v[0];
  r.v[0] = (uint64_t)d;
  d = (u128)a.v[1] - b.v[1] - (uint64_t)((d >> 64) & 1u);
  r.v[1] = (uint64_t)d;
  d = (u128)a.v[2] - b.v[2] - (uint64_t)((d >> 64) & 1u);
  r.v[2] = (uint64_t)d;
  d = (u128)a.v[3] - b.v[3] - (uint64_t)((d >> 64) & 1u);
  r.v[3] = (uint64_t)d;
  const uint64_t m = (uint64_t)0 - (uint64_t)((d >> 64) & 1u);   // all ones when a < b: add p back
  u128 c = (u128)r.v[0] + (P[0] & m);
  r.v[0] = (uint64_t)c;
  c = (u128)r.v[1] + (P[1] & m) + (uint64_t)(c >> 64);
  r.v[1] = (uint64_t)c;
  c = (u128)r.v[2] + (P[2] & m) + (uint64_t)(c >> 64);
  r.v[2] = (uint64_t)c;
  r.v[3] = r.v[3] + (P[3] & m) + (uint64_t)(c >> 64);
  return r;
}

// Montgomery product a * b / 2^256 mod p, operand-scanning with the reduction interleaved (CIOS), fully unrolled.
// p < 2^254 keeps the running value below 2 p < 2^255: four words and one carry word suffice.
#define MSM_H64_ROUND(bi)                                   \
  {                                                         \
    u128 c = (u128)a.v[0] * (bi) + t0;                      \
    t0 = (uint64_t)c;                                       \
    c = (u128)a.v[1] * (bi) + t1 + (uint64_t)(c >> 64);     \
    t1 = (uint64_t)c;                                       \
    c = (u128)a.v[2] * (bi) + t2 + (uint64_t)(c >> 64);     \
    t2 = (uint64_t)c;                                       \
    c = (u128)a.v[3] * (bi) + t3 + (uint64_t)(c >> 64);     \
    t3 = (uint64_t)c;                                       \
    const uint64_t t4 = (uint64_t)(c >> 64);                \
    const uint64_t m = t0 * NINV;                           \
    c = ((u128)m * P[0] + t0) >> 64;                        \
    c = (u128)m * P[1] + t1 + (uint64_t)c;                  \
    t0 = (uint64_t)c;                                       \
    c = (u128)m * P[2] + t2 + (uint64_t)(c >> 64);          \
    t1 = (uint64_t)c;                                       \
    c = (u128)m * P[3] + t3 + (uint64_t)(c >> 64);          \
    t2 = (uint64_t)c;                                       \
    t3 = t4 + (uint64_t)(c >> 64);                          \
  }
inline Fe mul_portable(const Fe& a, const Fe& b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  MSM_H64_ROUND(b.v[0])
  MSM_H64_ROUND(b.v[1])
  MSM_H64_ROUND(b.v[2])
  MSM_H64_ROUND(b.v[3])
  Fe r;
  r.v[0] = t0;
  r.v[1] = t1;
  r.v[2] = t2;
  r.v[3] = t3;
  reduce_once(r.v);
  return r;
}
#undef MSM_H64_ROUND

#if defined(__x86_64__) && !defined(MSM_H64_NO_ASM)
// The same CIOS rounds with MULX and the two independent carry chains of ADCX (CF) / ADOX (OF): the low halves of
// a_j * b_i ride one chain, the high halves the other, so no carry is ever materialised in a register.  The
// running value lives in five registers whose roles rotate from round to round (the word that the reduction zeroes
// becomes the next round's top word).  ~1.45x the rate of the portable form above on Zen 5 / Sapphire Rapids.
// Only this ONE function is compiled for BMI2 + ADX (target attribute, no -m flags on the translation unit: the
// compiler must not use either extension anywhere else), and mul() asks CPUID once before using it -- the library
// loads and runs on an x86-64 host without them (the portable form), like the AVX-512 IFMA path of host_ifma.cpp.
#define MSM_H64_ASM_ROUND(i, T0, T1, T2, T3, T4)                                                    \
  "movq " #i "*8(%[b]), %%rdx\n\t"                                                                  \
  "xorl %k[" #T4 "], %k[" #T4 "]\n\t" /* top word = 0; clears CF and OF */                           \
  "mulx 0(%[a]), %[lo], %[hi]\n\t  adcx %[lo], %[" #T0 "]\n\t  adox %[hi], %[" #T1 "]\n\t"          \
  "mulx 8(%[a]), %[lo], %[hi]\n\t  adcx %[lo], %[" #T1 "]\n\t  adox %[hi], %[" #T2 "]\n\t"          \
  "mulx 16(%[a]), %[lo], %[hi]\n\t adcx %[lo], %[" #T2 "]\n\t  adox %[hi], %[" #T3 "]\n\t"          \
  "mulx 24(%[a]), %[lo], %[hi]\n\t adcx %[lo], %[" #T3 "]\n\t  adox %[hi], %[" #T4 "]\n\t"          \
  "movl $0, %k[lo]\n\t             adcx %[lo], %[" #T4 "]\n\t"                                      \
  "movq %[" #T0 "], %%rdx\n\t      imulq %[ninv], %%rdx\n\t"                                        \
  "xorl %k[hi], %k[hi]\n\t" /* clears CF and OF again (imul set them) */                            \
  "mulx 0(%[p]), %[lo], %[hi]\n\t  adcx %[lo], %[" #T0 "]\n\t  adox %[hi], %[" #T1 "]\n\t"          \
  "mulx 8(%[p]), %[lo], %[hi]\n\t  adcx %[lo], %[" #T1 "]\n\t  adox %[hi], %[" #T2 "]\n\t"          \
  "mulx 16(%[p]), %[lo], %[hi]\n\t adcx %[lo], %[" #T2 "]\n\t  adox %[hi], %[" #T3 "]\n\t"          \
  "mulx 24(%[p]), %[lo], %[hi]\n\t adcx %[lo], %[" #T3 "]\n\t  adox %[hi], %[" #T4 "]\n\t"          \
  "movl $0, %k[lo]\n\t             adcx %[lo], %[" #T4 "]\n\t"
__attribute__((target("bmi2,adx"))) inline Fe mul_mulx(const Fe& a, const Fe& b) {
  static const uint64_t kP[4] = {P[0], P[1], P[2], P[3]};
  uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, lo, hi;
  __asm__(MSM_H64_ASM_ROUND(0, r0, r1, r2, r3, r4)   /* leaves the value in r1 r2 r3 r4 (r0 = 0) */
          MSM_H64_ASM_ROUND(1, r1, r2, r3, r4, r0)
          MSM_H64_ASM_ROUND(2, r2, r3, r4, r0, r1)
          MSM_H64_ASM_ROUND(3, r3, r4, r0, r1, r2)   /* result: r4 r0 r1 r2 */
          : [r0] "+&r"(r0), [r1] "+&r"(r1), [r2] "+&r"(r2), [r3] "+&r"(r3), [r4] "+&r"(r4), [lo] "=&r"(lo), [hi] "=&r"(hi)
          : [a] "r"(a.v), [b] "r"(b.v), [p] "r"(kP), [ninv] "r"(NINV), "m"(a), "m"(b), "m"(kP)
          : "rdx", "cc");
  Fe r;
  r.v[0] = r4;
  r.v[1] = r0;
  r.v[2] = r1;
  r.v[3] = r2;
  reduce_once(r.v);
  return r;
}
#undef MSM_H64_ASM_ROUND
inline bool have_mulx_adx() {   // CPUID.(EAX=7,ECX=0):EBX bit 8 = BMI2, bit 19 = ADX
  static const bool ok = [] {
    unsigned a = 0, b = 0, c = 0, d = 0;
    __asm__("cpuid" : "=a"(a), "=b"(b), "=c"(c), "=d"(d) : "a"(0), "c"(0));
    if (a < 7) return false;
    __asm__("cpuid" : "=a"(a), "=b"(b), "=c"(c), "=d"(d) : "a"(7), "c"(0));
    return ((b >> 8) & 1u) && ((b >> 19) & 1u) && !std::getenv("MSM_AMD_HOST_NO_MULX");
  }();
  return ok;
}
inline Fe mul(const Fe& a, const Fe& b) { return have_mulx_adx() ? mul_mulx(a, b) : mul_portable(a, b); }
#else
inline Fe mul(const Fe& a, const Fe& b) { return mul_portable(a, b); }
#endif
inline Fe sqr(const Fe& a) { return mul(a, a); }

inline Fe one() {   // 2^256 mod p
  Fe r;
  const u256 o = Fq::one();
  std::memcpy(&r, &o, sizeof r);
  return r;
}

inline Fe inv_fermat(const Fe& a) {   // a^(p - 2): 254 squarings + 127 multiplications; kept as the checker of inv()
  uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
  Fe acc = one();
  for (int i = 253; i >= 0; --i) {
    acc = sqr(acc);
    if ((e[i >> 6] >> (i & 63)) & 1u) acc = mul(acc, a);
  }
  return acc;
}

// ---- inversion by the binary GCD with 31-step inner rounds on 64-bit approximations (T. Pornin, "Optimized Binary
// GCD for Modular Inversion", 2020).  Invariants a = u y, b = v y (mod p) with a = y, b = p, u = 1, v = 0 at the start;
// a round takes 31 steps of the binary GCD on xa = [top 33 bits | low 31 bits] of a and xb likewise (both aligned
// on the longer of the two), recording them as factors (f0, g0, f1, g1) with |f| + |g| <= 2^31, then applies
//     (a, b) <- ((f0 a + g0 b) / 2^31, (f1 a + g1 b) / 2^31)         exact: the low 31 bits were followed exactly
//     (u, v) <- ((f0 u + g0 v) / 2^31, (f1 u + g1 v) / 2^31) mod p   one Montgomery step each
// A wrong comparison on the approximations can only make a or b negative: it is negated together with its factors.
// 2 * 254 - 1 = 507 steps suffice for a 254-bit modulus; 17 rounds = 527 (steps on a = 0 change nothing).  Then b = 1
// and v = y^-1.  About a fifth of the cost of the exponentiation; the batched-affine CPU MSM pays one inversion per
// 512-1024 additions and every GPU MSM one for its normalisation.
namespace bingcd {
typedef __int128 i128;
constexpr uint32_t kPinv31 = (uint32_t)(NINV & 0x7FFFFFFFu);   // -p^-1 mod 2^31

struct S5 {   // 320-bit two's complement
  uint64_t w[5];
};
inline S5 lin(const uint64_t x[4], const uint64_t y[4], int64_t f, int64_t g) {   // f x + g y
  S5 r;
  i128 acc = 0;
  for (int i = 0; i < 4; ++i) {
    acc += (i128)f * (i128)x[i] + (i128)g * (i128)y[i];
    r.w[i] = (uint64_t)acc;
    acc >>= 64;
  }
  r.w[4] = (uint64_t)acc;
  return r;
}
inline void negate(S5& t) {
  u128 c = 1;
  for (int i = 0; i < 5; ++i) {
    c += (uint64_t)~t.w[i];
    t.w[i] = (uint64_t)c;
    c >>= 64;
  }
}
inline void shr31(S5& t) {   // arithmetic
  for (int i = 0; i < 4; ++i) t.w[i] = (t.w[i] >> 31) | (t.w[i + 1] << 33);
  t.w[4] = (uint64_t)((int64_t)t.w[4] >> 31);
}
// (f x + g y) / 2^31 >= 0 after a possible negation (reported: the caller negates the same row of factors)
inline bool lin_div_abs(const uint64_t x[4], const uint64_t y[4], int64_t f, int64_t g, uint64_t out[4]) {
  S5 t = lin(x, y, f, g);
  const bool negative = (int64_t)t.w[4] < 0;
  if (negative) negate(t);
  shr31(t);
  for (int i = 0; i < 4; ++i) out[i] = t.w[i];
  return negative;
}
// (f x + g y) / 2^31 mod p for x, y < p
inline void lin_div_mod(const uint64_t x[4], const uint64_t y[4], int64_t f, int64_t g, uint64_t out[4]) {
  S5 t = lin(x, y, f, g);                                               // |t| < 2^31 p
  const uint64_t k = ((uint32_t)t.w[0] * kPinv31) & 0x7FFFFFFFu;        // t + k p = 0 mod 2^31
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)k * P[i] + t.w[i];
    t.w[i] = (uint64_t)c;
    c >>= 64;
  }
  t.w[4] += (uint64_t)c;
  shr31(t);                                                             // in (-p, 2p)
  if ((int64_t)t.w[4] < 0) {
    u128 a = 0;
    for (int i = 0; i < 4; ++i) {
      a += (u128)t.w[i] + P[i];
      out[i] = (uint64_t)a;
      a >>= 64;
    }
  } else {
    for (int i = 0; i < 4; ++i) out[i] = t.w[i];
    reduce_once(out);
  }
}
}  // namespace bingcd

inline Fe inv(const Fe& y_mont) {   // y^-1 in Montgomery form; inv(0) = 0
  using namespace bingcd;
  uint64_t a[4] = {y_mont.v[0], y_mont.v[1], y_mont.v[2], y_mont.v[3]};
  uint64_t b[4] = {P[0], P[1], P[2], P[3]};
  uint64_t u[4] = {1, 0, 0, 0}, v[4] = {0, 0, 0, 0};
  for (int round = 0; round < 17; ++round) {
    uint64_t xa, xb;
    int j = 3;
    while (j > 0 && (a[j] | b[j]) == 0) --j;
    if (j == 0) {
      xa = a[0];
      xb = b[0];
    } else {
      const int s = __builtin_clzll(a[j] | b[j]);
      const uint64_t ta = s ? (a[j] << s) | (a[j - 1] >> (64 - s)) : a[j];
      const uint64_t tb = s ? (b[j] << s) | (b[j - 1] >> (64 - s)) : b[j];
      xa = ((ta >> 31) << 31) | (a[0] & 0x7FFFFFFFu);
      xb = ((tb >> 31) << 31) | (b[0] & 0x7FFFFFFFu);
    }
    int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    for (int i = 0; i < 31; ++i) {
      const uint64_t odd = (uint64_t)0 - (xa & 1u);
      const uint64_t swap = odd & ((uint64_t)0 - (uint64_t)(xa < xb));
      uint64_t t = (xa ^ xb) & swap;
      xa ^= t;
      xb ^= t;
      t = (uint64_t)(f0 ^ f1) & swap;
      f0 ^= (int64_t)t;
      f1 ^= (int64_t)t;
      t = (uint64_t)(g0 ^ g1) & swap;
      g0 ^= (int64_t)t;
      g1 ^= (int64_t)t;
      xa -= xb & odd;
      f0 -= f1 & (int64_t)odd;
      g0 -= g1 & (int64_t)odd;
      xa >>= 1;
      f1 = (int64_t)((uint64_t)f1 << 1);   // (a left shift of a negative value is undefined before C++20)
      g1 = (int64_t)((uint64_t)g1 << 1);
    }
    uint64_t na[4], nb[4], nu[4], nv[4];
    if (lin_div_abs(a, b, f0, g0, na)) {
      f0 = -f0;
      g0 = -g0;
    }
    if (lin_div_abs(a, b, f1, g1, nb)) {
      f1 = -f1;
      g1 = -g1;
    }
    lin_div_mod(u, v, f0, g0, nu);
    lin_div_mod(u, v, f1, g1, nv);
    for (int i = 0; i < 4; ++i) {
      a[i] = na[i];
      b[i] = nb[i];
      u[i] = nu[i];
      v[i] = nv[i];
    }
  }
  // v = (y R)^-1 as a plain integer; y^-1 R = v R^2 = mul(v, R^3)
  static const Fe r3 = [] {
    Fe r2 = one();
    for (int i = 0; i < 256; ++i) r2 = add(r2, r2);   // R * 2^256 = R^2
    return mul(r2, r2);                               // R^4 / R
  }();
  Fe vv;
  for (int i = 0; i < 4; ++i) vv.v[i] = v[i];
  return mul(vv, r3);
}

inline bool is_identity(const Jac& p) { return is_zero(p.z); }
inline Jac identity() {
  Jac r;
  r.x = one();
  r.y = one();
  std::memset(&r.z, 0, sizeof r.z);
  return r;
}

// dbl-2009-l, a = 0 (2M + 5S) -- the formulas of jac_double (bn254_ec.hip.h).
inline Jac jdouble(const Jac& p) {
  if (is_identity(p)) return p;
  const Fe A = sqr(p.x);
  const Fe B = sqr(p.y);
  const Fe C = sqr(B);
  const Fe D = dbl(sub(sub(sqr(add(p.x, B)), A), C));
  const Fe E = add(dbl(A), A);
  const Fe F = sqr(E);
  Jac r;
  r.x = sub(F, dbl(D));
  r.y = sub(mul(E, sub(D, r.x)), dbl(dbl(dbl(C))));
  r.z = dbl(mul(p.y, p.z));
  return r;
}

// add-2007-bl (11M + 5S) with the case analysis of jac_add (bn254_ec.hip.h).
inline Jac jadd(const Jac& p, const Jac& q) {
  if (is_identity(p)) return q;
  if (is_identity(q)) return p;
  const Fe Z1Z1 = sqr(p.z);
  const Fe Z2Z2 = sqr(q.z);
  const Fe U1 = mul(p.x, Z2Z2);
  const Fe U2 = mul(q.x, Z1Z1);
  const Fe S1 = mul(mul(p.y, q.z), Z2Z2);
  const Fe S2 = mul(mul(q.y, p.z), Z1Z1);
  const Fe H = sub(U2, U1);
  const Fe rr = sub(S2, S1);
  if (is_zero(H)) {
    if (is_zero(rr)) return jdouble(p);
    return identity();
  }
  const Fe I = sqr(dbl(H));
  const Fe J = mul(H, I);
  const Fe r2 = dbl(rr);
  const Fe V = mul(U1, I);
  Jac r;
  r.x = sub(sub(sqr(r2), J), dbl(V));
  r.y = sub(mul(r2, sub(V, r.x)), dbl(mul(S1, J)));
  r.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
  return r;
}

inline Jac load(const Jacobian& p) {
  Jac r;
  std::memcpy(&r, &p, sizeof r);
  return r;
}
inline Jacobian store(const Jac& p) {
  Jacobian r;
  std::memcpy(&r, &p, sizeof r);
  return r;
}

// (X, Y, Z) -> (X / Z^2, Y / Z^3, R mod p), or the canonical identity (1, 1, 0) in Montgomery form.
inline Jac normalise(const Jac& p) {
  if (is_identity(p)) return identity();
  const Fe zi = inv(p.z);
  const Fe zi2 = sqr(zi);
  Jac r;
  r.x = mul(p.x, zi2);
  r.y = mul(p.y, mul(zi2, zi));
  r.z = one();
  return r;
}

}  // namespace h64
}  // namespace msm_amd
