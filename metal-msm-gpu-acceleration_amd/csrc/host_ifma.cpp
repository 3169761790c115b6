// AVX-512 IFMA arithmetic for the batched-affine phase of the product's CPU MSM (host_msm.hip): eight bucket additions
// per vector.  Plain C++ (no HIP), compiled without global -mavx512* flags: every function carries a target attribute
// and host_msm.hip asks ifma::available() (CPUID) before calling in, so the library still loads on hosts without IFMA.
//
// Representation: eight field elements side by side, radix 2^52, five limbs each (V5), Montgomery radix Q = 2^260.
// One vpmadd52luq / vpmadd52huq pair multiplies eight 52-bit limbs and ACCUMULATES into 64-bit lanes, so a Montgomery
// multiplication is 105 IFMA instructions for eight products (about 1.8 ns per product on Zen 5 against 10 ns for the
// MULX / ADCX / ADOX code of host_fq64.h).  Values between operations live in [0, 2p) with limbs < 2^52 (IFMA reads only
// the low 52 bits of a lane); what is stored in a bucket is canonical (< p), so equal x-coordinates are equal words.
//
// The caller keeps bucket and point coordinates as 4 x u64 little-endian integers in the Q domain (value * 2^260 mod p);
// to_q / from_q move whole arrays between that and the R = 2^256 domain of the rest of the library.
#include "host_ifma.h"

#include <immintrin.h>

#include <cstring>

#define MSM_IFMA __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl,avx512bw")))

namespace msm_amd {
namespace ifma {

namespace {

constexpr uint64_t kP64[4] = {0x3C208C16D87CFD47ull, 0x97816A916871CA8Dull, 0xB85045B68181585Dull, 0x30644E72E131A029ull};
constexpr uint64_t kM52 = (1ull << 52) - 1;

struct V5 {
  __m512i l[5];
};

struct Consts {
  uint64_t p52[5], two_p52[5], ninv52;
  uint64_t to_q52[5];     // 2^264 mod p : mont(x R, 2^264) = x Q
  uint64_t from_q52[5];   // 2^256 mod p : mont(x Q, 2^256) = x R
  uint64_t one_q52[5];    // 2^260 mod p : the field's one in the Q domain
};

void split52(const uint64_t a[4], uint64_t o[5]) {
  o[0] = a[0] & kM52;
  o[1] = ((a[0] >> 52) | (a[1] << 12)) & kM52;
  o[2] = ((a[1] >> 40) | (a[2] << 24)) & kM52;
  o[3] = ((a[2] >> 28) | (a[3] << 36)) & kM52;
  o[4] = a[3] >> 16;
}

// 2^k mod p by doubling on 4 x u64 (plain integers below p)
void pow2_mod_p(unsigned k, uint64_t out[4]) {
  typedef unsigned __int128 u128;
  uint64_t v[4] = {1, 0, 0, 0};
  for (unsigned i = 0; i < k; ++i) {
    uint64_t d[4], c = 0;
    for (int j = 0; j < 4; ++j) {
      const u128 s = ((u128)v[j] << 1) | c;
      d[j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    // d < 2p < 2^255: subtract p if d >= p
    uint64_t e[4];
    u128 b = 0;
    for (int j = 0; j < 4; ++j) {
      const u128 s = (u128)d[j] - kP64[j] - (uint64_t)b;
      e[j] = (uint64_t)s;
      b = (s >> 64) & 1;
    }
    std::memcpy(v, b ? d : e, sizeof v);
  }
  std::memcpy(out, v, 32);
}

const Consts& consts() {
  static const Consts c = [] {
    Consts k;
    split52(kP64, k.p52);
    uint64_t two_p[4], carry = 0;
    for (int j = 0; j < 4; ++j) {
      two_p[j] = (kP64[j] << 1) | carry;
      carry = kP64[j] >> 63;
    }
    split52(two_p, k.two_p52);
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - kP64[0] * inv;   // p^-1 mod 2^64
    k.ninv52 = (0 - inv) & kM52;
    uint64_t t[4];
    pow2_mod_p(264, t);
    split52(t, k.to_q52);
    pow2_mod_p(256, t);
    split52(t, k.from_q52);
    pow2_mod_p(260, t);
    split52(t, k.one_q52);
    return k;
  }();
  return c;
}

MSM_IFMA inline V5 bcast(const uint64_t v[5]) {
  V5 r;
  for (int i = 0; i < 5; ++i) r.l[i] = _mm512_set1_epi64((long long)v[i]);
  return r;
}

// a * b / 2^260 mod p, result in [0, 2p) for a, b < 2^258, limbs normalised
MSM_IFMA inline V5 mont(const V5& a, const V5& b, const Consts& K) {
  const __m512i zero = _mm512_setzero_si512();
  const __m512i ninv = _mm512_set1_epi64((long long)K.ninv52);
  __m512i p[5];
  for (int i = 0; i < 5; ++i) p[i] = _mm512_set1_epi64((long long)K.p52[i]);
  __m512i t[11];
  for (int i = 0; i < 11; ++i) t[i] = zero;
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 5; ++j) {
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], a.l[i], b.l[j]);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], a.l[i], b.l[j]);
    }
    const __m512i m = _mm512_madd52lo_epu64(zero, t[i], ninv);
    for (int j = 0; j < 5; ++j) {
      t[i + j] = _mm512_madd52lo_epu64(t[i + j], m, p[j]);
      t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], m, p[j]);
    }
    t[i + 1] = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(t[i], 52));
  }
  const __m512i mask = _mm512_set1_epi64((long long)kM52);
  V5 r;
  __m512i c = zero;
  for (int i = 0; i < 5; ++i) {
    const __m512i s = _mm512_add_epi64(t[5 + i], c);
    r.l[i] = i < 4 ? _mm512_and_si512(s, mask) : s;
    c = _mm512_srli_epi64(s, 52);
  }
  return r;
}

// a - b for a, b in [0, 2p): exact, + 2p where the difference is negative; result in [0, 2p), limbs normalised
MSM_IFMA inline V5 sub2p(const V5& a, const V5& b, const Consts& K) {
  const __m512i mask = _mm512_set1_epi64((long long)kM52);
  V5 r;
  __m512i borrow = _mm512_setzero_si512();   // 0 or -1 per lane
  for (int i = 0; i < 5; ++i) {
    __m512i d = _mm512_add_epi64(_mm512_sub_epi64(a.l[i], b.l[i]), borrow);
    borrow = _mm512_srai_epi64(d, 63);
    r.l[i] = i < 4 ? _mm512_and_si512(d, mask) : d;
  }
  // lanes whose top limb went negative get 2p added (the low limbs already wrapped modulo 2^52 each)
  const __m512i neg = _mm512_srai_epi64(r.l[4], 63);
  __m512i carry = _mm512_setzero_si512();
  for (int i = 0; i < 5; ++i) {
    const __m512i add = _mm512_and_si512(neg, _mm512_set1_epi64((long long)K.two_p52[i]));
    __m512i s = _mm512_add_epi64(_mm512_add_epi64(r.l[i], add), carry);
    if (i < 4) {
      carry = _mm512_srli_epi64(s, 52);
      s = _mm512_and_si512(s, mask);
    } else {
      // top limb: the borrow left it as (value - 2^48-ish wrap) in two's complement; adding 2p's top limb and the carry
      // brings it back into [0, 2^51)
      s = _mm512_and_si512(s, _mm512_set1_epi64((long long)((1ull << 52) - 1)));
    }
    r.l[i] = s;
  }
  return r;
}

// a in [0, 2p) -> [0, p)
MSM_IFMA inline V5 canon(const V5& a, const Consts& K) {
  const __m512i mask = _mm512_set1_epi64((long long)kM52);
  V5 d;
  __m512i borrow = _mm512_setzero_si512();
  for (int i = 0; i < 5; ++i) {
    __m512i t = _mm512_add_epi64(_mm512_sub_epi64(a.l[i], _mm512_set1_epi64((long long)K.p52[i])), borrow);
    borrow = _mm512_srai_epi64(t, 63);
    d.l[i] = _mm512_and_si512(t, mask);
  }
  // borrow == -1: a < p, keep a
  const __mmask8 keep = _mm512_cmpneq_epi64_mask(borrow, _mm512_setzero_si512());
  V5 r;
  for (int i = 0; i < 5; ++i) r.l[i] = _mm512_mask_blend_epi64(keep, d.l[i], a.l[i]);
  return r;
}

// 4 x u64 words (per lane) -> 5 x 52-bit limbs
MSM_IFMA inline V5 from_words(const __m512i w[4]) {
  const __m512i mask = _mm512_set1_epi64((long long)kM52);
  V5 r;
  r.l[0] = _mm512_and_si512(w[0], mask);
  r.l[1] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[0], 52), _mm512_slli_epi64(w[1], 12)), mask);
  r.l[2] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[1], 40), _mm512_slli_epi64(w[2], 24)), mask);
  r.l[3] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[2], 28), _mm512_slli_epi64(w[3], 36)), mask);
  r.l[4] = _mm512_srli_epi64(w[3], 16);
  return r;
}
MSM_IFMA inline void to_words(const V5& a, __m512i w[4]) {   // a < 2^256, limbs normalised
  w[0] = _mm512_or_si512(a.l[0], _mm512_slli_epi64(a.l[1], 52));
  w[1] = _mm512_or_si512(_mm512_srli_epi64(a.l[1], 12), _mm512_slli_epi64(a.l[2], 40));
  w[2] = _mm512_or_si512(_mm512_srli_epi64(a.l[2], 24), _mm512_slli_epi64(a.l[3], 28));
  w[3] = _mm512_or_si512(_mm512_srli_epi64(a.l[3], 36), _mm512_slli_epi64(a.l[4], 16));
}

// 8 x 8 transpose of u64: in[k] = record k (8 words)  ->  out[j] = word j of the eight records (and back: the
// transpose is its own inverse).  8 unpacks + 16 lane shuffles; the hardware gather of the same 64 words costs about twice as
// much on Zen 4 / 5, a scatter more.
MSM_IFMA inline void transpose8(const __m512i in[8], __m512i out[8]) {
  __m512i a[8], b[8];
  for (int i = 0; i < 8; i += 2) {
    a[i] = _mm512_unpacklo_epi64(in[i], in[i + 1]);
    a[i + 1] = _mm512_unpackhi_epi64(in[i], in[i + 1]);
  }
  b[0] = _mm512_shuffle_i64x2(a[0], a[2], 0x88);
  b[1] = _mm512_shuffle_i64x2(a[0], a[2], 0xDD);
  b[2] = _mm512_shuffle_i64x2(a[4], a[6], 0x88);
  b[3] = _mm512_shuffle_i64x2(a[4], a[6], 0xDD);
  b[4] = _mm512_shuffle_i64x2(a[1], a[3], 0x88);
  b[5] = _mm512_shuffle_i64x2(a[1], a[3], 0xDD);
  b[6] = _mm512_shuffle_i64x2(a[5], a[7], 0x88);
  b[7] = _mm512_shuffle_i64x2(a[5], a[7], 0xDD);
  out[0] = _mm512_shuffle_i64x2(b[0], b[2], 0x88);
  out[4] = _mm512_shuffle_i64x2(b[0], b[2], 0xDD);
  out[2] = _mm512_shuffle_i64x2(b[1], b[3], 0x88);
  out[6] = _mm512_shuffle_i64x2(b[1], b[3], 0xDD);
  out[1] = _mm512_shuffle_i64x2(b[4], b[6], 0x88);
  out[5] = _mm512_shuffle_i64x2(b[4], b[6], 0xDD);
  out[3] = _mm512_shuffle_i64x2(b[5], b[7], 0x88);
  out[7] = _mm512_shuffle_i64x2(b[5], b[7], 0xDD);
}

// eight records of 8 u64 (x words 0..3, y words 4..7), record k at base + 8 * idx[k]  ->  x, y
MSM_IFMA inline void gather_xy(const uint64_t* base, const uint64_t idx[8], V5& x, V5& y) {
  __m512i rec[8], w[8];
  for (int k = 0; k < 8; ++k) rec[k] = _mm512_loadu_si512(base + 8 * idx[k]);
  transpose8(rec, w);
  x = from_words(w);
  y = from_words(w + 4);
}

}  // namespace

bool available() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") &&
                         __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                         __builtin_cpu_supports("avx512bw");
  return ok;
}

// in / out: n field elements of 4 x u64; dir 0: R domain -> Q domain, 1: Q -> R.  Canonical results.
MSM_IFMA void convert(const uint64_t* in, uint64_t* out, size_t n, int dir) {
  const Consts& K = consts();
  const V5 c = bcast(dir == 0 ? K.to_q52 : K.from_q52);
  alignas(64) uint64_t buf[4][8];
  for (size_t at = 0; at < n; at += 8) {
    const size_t m = n - at < 8 ? n - at : 8;
    for (int j = 0; j < 4; ++j)
      for (size_t k = 0; k < 8; ++k) buf[j][k] = k < m ? in[(at + k) * 4 + j] : 0;
    __m512i w[4];
    for (int j = 0; j < 4; ++j) w[j] = _mm512_load_si512(buf[j]);
    const V5 r = canon(mont(from_words(w), c, K), K);
    to_words(r, w);
    for (int j = 0; j < 4; ++j) _mm512_store_si512(buf[j], w[j]);
    for (int j = 0; j < 4; ++j)
      for (size_t k = 0; k < m; ++k) out[(at + k) * 4 + j] = buf[j][k];
  }
}

// -a for canonical a != 0 (a y-coordinate of a finite point): p - a, limbs normalised
MSM_IFMA inline V5 neg_canon(const V5& a, const Consts& K) {
  const __m512i mask = _mm512_set1_epi64((long long)kM52);
  V5 r;
  __m512i borrow = _mm512_setzero_si512();
  for (int i = 0; i < 5; ++i) {
    const __m512i d = _mm512_add_epi64(_mm512_sub_epi64(_mm512_set1_epi64((long long)K.p52[i]), a.l[i]), borrow);
    borrow = _mm512_srai_epi64(d, 63);
    r.l[i] = _mm512_and_si512(d, mask);
  }
  return r;
}

// Single operations for the unit tests (tests/test_host_ifma.py): a, b, out = count field elements of 4 x u64,
// canonical in and out.  op 0: a * b / 2^260 mod p   1: a - b mod p   2: -a mod p (a != 0)
MSM_IFMA void test_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count) {
  const Consts& K = consts();
  alignas(64) uint64_t ba[4][8], bb[4][8];
  for (size_t at = 0; at < count; at += 8) {
    const size_t m = count - at < 8 ? count - at : 8;
    for (int j = 0; j < 4; ++j)
      for (size_t k = 0; k < 8; ++k) {
        ba[j][k] = k < m ? a[(at + k) * 4 + j] : (j == 0 ? 1 : 0);
        bb[j][k] = k < m ? b[(at + k) * 4 + j] : (j == 0 ? 1 : 0);
      }
    __m512i wa[4], wb[4];
    for (int j = 0; j < 4; ++j) {
      wa[j] = _mm512_load_si512(ba[j]);
      wb[j] = _mm512_load_si512(bb[j]);
    }
    const V5 x = from_words(wa), y = from_words(wb);
    V5 r;
    if (op == 0) r = canon(mont(x, y, K), K);
    else if (op == 1) r = canon(sub2p(x, y, K), K);
    else r = neg_canon(x, K);
    to_words(r, wa);
    for (int j = 0; j < 4; ++j) _mm512_store_si512(ba[j], wa[j]);
    for (int j = 0; j < 4; ++j)
      for (size_t k = 0; k < m; ++k) out[(at + k) * 4 + j] = ba[j][k];
  }
}

// Forward pass of one batch: element k adds point number (pt_idx[k] & 0x7FFFFFFF) of `pts` (8 u64 per point: x, y),
// negated when bit 31 of pt_idx[k] is set, to bucket buckets[8 * bucket_idx[k] ..].
// Computes d = x2 - x1 and the running products of eight interleaved chains (lane = k mod 8); returns the eight chain
// totals (Q domain, canonical, 4 x u64 each).  All coordinates Q domain, canonical; x2 != x1 for every element.
MSM_IFMA int forward(const uint64_t* buckets, const uint32_t* bucket_idx, const uint64_t* pts, const uint32_t* pt_idx,
                     int count, Scratch& ws, uint64_t totals[8][4]) {
  const Consts& K = consts();
  const int rows = (count + 7) / 8;
  ws.rows = rows;
  ws.n_special = 0;
  std::memset(ws.is_special, 0, (size_t)count);
  const V5 one = bcast(K.one_q52);
  V5* X1 = (V5*)ws.x1;
  V5* Y1 = (V5*)ws.y1;
  V5* X2 = (V5*)ws.x2;
  V5* Y2 = (V5*)ws.y2;
  V5* D = (V5*)ws.d;
  V5* PRE = (V5*)ws.pre;
  for (int r = 0; r < rows; ++r) {
    const int base = 8 * r, live = count - base < 8 ? count - base : 8;
    const __mmask8 m = (__mmask8)((1u << live) - 1u);
    // pad lanes re-read element `base` (valid addresses) and get d = 1 below
    alignas(64) uint64_t bi[8], pi[8], sg[8];
    for (int k = 0; k < 8; ++k) {
      const int e = k < live ? base + k : base;
      bi[k] = (uint64_t)bucket_idx[e];
      pi[k] = (uint64_t)(pt_idx[e] & 0x7FFFFFFFu);
      sg[k] = (uint64_t)0 - (uint64_t)(pt_idx[e] >> 31);
    }
    gather_xy(buckets, bi, X1[r], Y1[r]);
    V5 y2;
    gather_xy(pts, pi, X2[r], y2);
    const __mmask8 negate = _mm512_cmpneq_epi64_mask(_mm512_load_si512(sg), _mm512_setzero_si512());
    const V5 ny = neg_canon(y2, K);
    for (int i = 0; i < 5; ++i) Y2[r].l[i] = _mm512_mask_blend_epi64(negate, y2.l[i], ny.l[i]);
    V5 d = sub2p(X2[r], X1[r], K);
    // x2 = x1: the same point again or its negative -- not a chord addition.  The lane computes with d = 1 and the
    // element is reported back (the caller routes the point to the bucket's Jacobian side accumulator).
    const __m512i any = _mm512_or_si512(_mm512_or_si512(_mm512_or_si512(d.l[0], d.l[1]), _mm512_or_si512(d.l[2], d.l[3])), d.l[4]);
    const __mmask8 zero_d = (__mmask8)(_mm512_cmpeq_epi64_mask(any, _mm512_setzero_si512()) & m);
    if (zero_d) {
      for (int k = 0; k < live; ++k)
        if ((zero_d >> k) & 1) {
          ws.special[ws.n_special++] = base + k;
          ws.is_special[base + k] = 1;
        }
    }
    const __mmask8 keep = (__mmask8)(m & ~zero_d);
    for (int i = 0; i < 5; ++i) d.l[i] = _mm512_mask_blend_epi64(keep, one.l[i], d.l[i]);
    D[r] = d;
    PRE[r] = r ? mont(PRE[r - 1], d, K) : d;
  }
  const V5 t = canon(PRE[rows - 1], K);
  __m512i w[4];
  to_words(t, w);
  alignas(64) uint64_t buf[4][8];
  for (int j = 0; j < 4; ++j) _mm512_store_si512(buf[j], w[j]);
  for (int k = 0; k < 8; ++k)
    for (int j = 0; j < 4; ++j) totals[k][j] = buf[j][k];
  return ws.n_special;
}

// Backward pass: inv[k] = 1 / totals[k] (Q domain).  Writes the sums back into the buckets.
MSM_IFMA void backward(uint64_t* buckets, const uint32_t* bucket_idx, int count, Scratch& ws, const uint64_t inv[8][4]) {
  const Consts& K = consts();
  const int rows = ws.rows;
  V5* X1 = (V5*)ws.x1;
  V5* Y1 = (V5*)ws.y1;
  V5* X2 = (V5*)ws.x2;
  V5* Y2 = (V5*)ws.y2;
  V5* D = (V5*)ws.d;
  V5* PRE = (V5*)ws.pre;
  alignas(64) uint64_t buf[8][8];
  for (int k = 0; k < 8; ++k)
    for (int j = 0; j < 4; ++j) buf[j][k] = inv[k][j];
  __m512i w[8];
  for (int j = 0; j < 4; ++j) w[j] = _mm512_load_si512(buf[j]);
  V5 I = from_words(w);
  for (int r = rows - 1; r >= 0; --r) {
    const V5 dinv = r ? mont(I, PRE[r - 1], K) : I;
    if (r) I = mont(I, D[r], K);
    const V5 num = sub2p(Y2[r], Y1[r], K);
    const V5 lam = mont(num, dinv, K);
    const V5 l2 = mont(lam, lam, K);
    const V5 x3 = sub2p(sub2p(l2, X1[r], K), X2[r], K);
    const V5 e = sub2p(X1[r], x3, K);
    const V5 y3 = sub2p(mont(lam, e, K), Y1[r], K);
    to_words(canon(x3, K), w);
    to_words(canon(y3, K), w + 4);
    __m512i rec[8];
    transpose8(w, rec);
    const int base = 8 * r, live = count - base < 8 ? count - base : 8;
    for (int k = 0; k < live; ++k) {
      // (an element with x2 = x1 computed with d = 1: its bucket keeps its value)
      if (!ws.is_special[base + k]) _mm512_storeu_si512(buckets + (size_t)bucket_idx[base + k] * 8, rec[k]);
    }
  }
}

}  // namespace ifma
}  // namespace msm_amd
