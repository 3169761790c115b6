/* CPU oracle (TEST INFRASTRUCTURE, not product code) -- plain C restatement.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (metal-msm-gpu-acceleration_amd/libmsm_amd.so) never links or calls it.
 *
 * What it restates
 *   - BN254 Fq/Fr Montgomery arithmetic with R = 2^256 (fp_bn254.h.metal:225-290 does the same with
 *     8 x u32 limbs; here 4 x u64 + unsigned __int128);
 *   - the Jacobian group law of src/metal/shader/curves/ec_point.h.metal:13-69,141-175;
 *   - the reference MSM pipeline msm.rs:189-217 (digits c=15/3 -> sort -> bucket sums -> weighted
 *     window sums -> Horner), function oracle_msm_reference_pipeline;
 *   - the CPU MSM the reference is compared with and hybridised with, halo2curves 0.7.0
 *     `msm::msm_best` (Cargo.toml:48; called at msm.rs:412,443,600 and gpu_profiler.rs:158).  That crate
 *     is NOT vendored and cannot be built here (no cargo/rustc).  oracle_msm_best restates its
 *     published algorithm from memory of the 0.7 line (unverified offline): window c = ceil(ln n)
 *     (3 below 32 points), one task per window, Booth digits, affine buckets updated in batches of 64
 *     with one shared inversion, Jacobian side buckets for colliding updates, summation by parts,
 *     window values shifted and added (see oracle_msm_best_ex).  It is the timed CPU baseline
 *     ("port"), labelled as a restatement, never as halo2curves itself.  oracle_msm_chunked (points
 *     split over threads, serial Jacobian bucket method per slice) is kept as a second checker.
 *
 * PARITY PINNING: see oracle/bn254_ref.py -- the reference holds no golden vectors for this path, so
 * at the literal-fixture level this oracle is "parity unpinned"; it is pinned mathematically (unique
 * group element, canonical affine coordinates) and cross-checked against the independent Python
 * big-int oracle and against known answers (2G, r*G = O) in tests/test_oracle.py.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;            /* field element, little-endian limbs */
typedef struct { fe x, y; } aff_t;               /* (0,0) = identity (halo2curves convention) */
typedef struct { fe x, y, z; } jac_t;            /* z == 0 = identity */

typedef struct { fe mod, one, r2; uint64_t inv; } field_t;

static const field_t FQ = {
  {{0x3C208C16D87CFD47ull, 0x97816A916871CA8Dull, 0xB85045B68181585Dull, 0x30644E72E131A029ull}},
  {{0xD35D438DC58F0D9Dull, 0x0A78EB28F5C70B3Dull, 0x666EA36F7879462Cull, 0x0E0A77C19A07DF2Full}},
  {{0xF32CFC5B538AFA89ull, 0xB5E71911D44501FBull, 0x47AB1EFF0A417FF6ull, 0x06D89F71CAB8351Full}},
  0x87D20782E4866389ull
};
static const field_t FR = {
  {{0x43E1F593F0000001ull, 0x2833E84879B97091ull, 0xB85045B68181585Dull, 0x30644E72E131A029ull}},
  {{0xAC96341C4FFFFFFBull, 0x36FC76959F60CD29ull, 0x666EA36F7879462Eull, 0x0E0A77C19A07DF2Full}},
  {{0x1BB8E645AE216DA7ull, 0x53FE3AB1E35C59E3ull, 0x8C49833D53BB8085ull, 0x0216D0B17F4E44A5ull}},
  0xC2E1F593EFFFFFFFull
};

static inline int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
static inline int fe_geq(const fe* a, const fe* b) {
  for (int i = 3; i >= 0; --i) { if (a->v[i] != b->v[i]) return a->v[i] > b->v[i]; }
  return 1;
}
static inline uint64_t raw_add(fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static inline uint64_t raw_sub(fe* r, const fe* a, const fe* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a->v[i] - b->v[i] - borrow;
    r->v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
static inline void f_add(const field_t* F, fe* r, const fe* a, const fe* b) {
  fe s; raw_add(&s, a, b);
  if (fe_geq(&s, &F->mod)) raw_sub(r, &s, &F->mod); else *r = s;
}
static inline void f_sub(const field_t* F, fe* r, const fe* a, const fe* b) {
  fe d; if (raw_sub(&d, a, b)) raw_add(r, &d, &F->mod); else *r = d;
}
static inline void f_neg(const field_t* F, fe* r, const fe* a) {
  if (fe_is_zero(a)) { *r = *a; return; }
  raw_sub(r, &F->mod, a);
}
/* Montgomery product a*b*R^-1 mod p, CIOS (fp_bn254.h.metal:237-290 is the 32-bit-limb CIOS). */
static inline void f_mul(const field_t* F, fe* r, const fe* a, const fe* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->inv;
    c = ((u128)m * F->mod.v[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * F->mod.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe s = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || fe_geq(&s, &F->mod)) raw_sub(r, &s, &F->mod); else *r = s;
}
static inline void f_sqr(const field_t* F, fe* r, const fe* a) { f_mul(F, r, a, a); }
static void f_pow(const field_t* F, fe* r, const fe* a, const fe* e) {
  fe acc = F->one;
  for (int i = 255; i >= 0; --i) {
    f_sqr(F, &acc, &acc);
    if ((e->v[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &acc, a);
  }
  *r = acc;
}
static void f_inv(const field_t* F, fe* r, const fe* a) {
  fe e = F->mod; e.v[0] -= 2; f_pow(F, r, a, &e);
}
static inline void f_to_mont(const field_t* F, fe* r, const fe* a) { f_mul(F, r, a, &F->r2); }
static inline void f_from_mont(const field_t* F, fe* r, const fe* a) {
  fe o = {{1, 0, 0, 0}}; f_mul(F, r, a, &o);
}

#define QADD(r, a, b) f_add(&FQ, r, a, b)
#define QSUB(r, a, b) f_sub(&FQ, r, a, b)
#define QMUL(r, a, b) f_mul(&FQ, r, a, b)
#define QSQR(r, a) f_sqr(&FQ, r, a)

static inline int aff_is_id(const aff_t* p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int jac_is_id(const jac_t* p) { return fe_is_zero(&p->z); }
static inline void jac_set_id(jac_t* p) { p->x = FQ.one; p->y = FQ.one; memset(&p->z, 0, sizeof(fe)); }

/* dbl-2009-l (a = 0); same group element as the reference's dbl-2007-bl (ec_point.h.metal:141-175). */
static void jac_double(jac_t* r, const jac_t* p) {
  if (jac_is_id(p)) { *r = *p; return; }
  fe A, B, C, D, E, F, t, c8;
  QSQR(&A, &p->x); QSQR(&B, &p->y); QSQR(&C, &B);
  QADD(&t, &p->x, &B); QSQR(&t, &t); QSUB(&t, &t, &A); QSUB(&t, &t, &C); QADD(&D, &t, &t);
  QADD(&E, &A, &A); QADD(&E, &E, &A); QSQR(&F, &E);
  jac_t o;
  QADD(&t, &D, &D); QSUB(&o.x, &F, &t);
  QADD(&c8, &C, &C); QADD(&c8, &c8, &c8); QADD(&c8, &c8, &c8);
  QSUB(&t, &D, &o.x); QMUL(&t, &E, &t); QSUB(&o.y, &t, &c8);
  QMUL(&t, &p->y, &p->z); QADD(&o.z, &t, &t);
  *r = o;
}
/* add-2007-bl with the reference's case analysis (ec_point.h.metal:13-69). */
static void jac_add(jac_t* r, const jac_t* p, const jac_t* q) {
  if (jac_is_id(p)) { *r = *q; return; }
  if (jac_is_id(q)) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, rr, i, j, v, t;
  QSQR(&z1z1, &p->z); QSQR(&z2z2, &q->z);
  QMUL(&u1, &p->x, &z2z2); QMUL(&u2, &q->x, &z1z1);
  QMUL(&s1, &p->y, &q->z); QMUL(&s1, &s1, &z2z2);
  QMUL(&s2, &q->y, &p->z); QMUL(&s2, &s2, &z1z1);
  QSUB(&h, &u2, &u1); QSUB(&rr, &s2, &s1);
  if (fe_is_zero(&h)) { if (fe_is_zero(&rr)) jac_double(r, p); else jac_set_id(r); return; }
  QADD(&i, &h, &h); QSQR(&i, &i); QMUL(&j, &h, &i); QADD(&rr, &rr, &rr); QMUL(&v, &u1, &i);
  jac_t o;
  QSQR(&o.x, &rr); QSUB(&o.x, &o.x, &j); QSUB(&o.x, &o.x, &v); QSUB(&o.x, &o.x, &v);
  QSUB(&t, &v, &o.x); QMUL(&t, &rr, &t); QMUL(&s1, &s1, &j); QADD(&s1, &s1, &s1); QSUB(&o.y, &t, &s1);
  QADD(&t, &p->z, &q->z); QSQR(&t, &t); QSUB(&t, &t, &z1z1); QSUB(&t, &t, &z2z2); QMUL(&o.z, &t, &h);
  *r = o;
}
/* madd-2007-bl: Jacobian + affine (affine must not be the identity). */
static void jac_madd(jac_t* r, const jac_t* p, const aff_t* q) {
  if (jac_is_id(p)) { r->x = q->x; r->y = q->y; r->z = FQ.one; return; }
  fe z1z1, u2, s2, h, hh, rr, i, j, v, t, yj;
  QSQR(&z1z1, &p->z); QMUL(&u2, &q->x, &z1z1);
  QMUL(&s2, &q->y, &p->z); QMUL(&s2, &s2, &z1z1);
  QSUB(&h, &u2, &p->x); QSUB(&rr, &s2, &p->y);
  if (fe_is_zero(&h)) { if (fe_is_zero(&rr)) jac_double(r, p); else jac_set_id(r); return; }
  QSQR(&hh, &h); QADD(&i, &hh, &hh); QADD(&i, &i, &i); QMUL(&j, &h, &i); QADD(&rr, &rr, &rr); QMUL(&v, &p->x, &i);
  jac_t o;
  QSQR(&o.x, &rr); QSUB(&o.x, &o.x, &j); QSUB(&o.x, &o.x, &v); QSUB(&o.x, &o.x, &v);
  QSUB(&t, &v, &o.x); QMUL(&t, &rr, &t); QMUL(&yj, &p->y, &j); QADD(&yj, &yj, &yj); QSUB(&o.y, &t, &yj);
  QADD(&t, &p->z, &h); QSQR(&t, &t); QSUB(&t, &t, &z1z1); QSUB(&o.z, &t, &hh);
  *r = o;
}
static void jac_madd_signed(jac_t* r, const jac_t* p, const aff_t* q, int negate) {
  if (!negate) { jac_madd(r, p, q); return; }
  aff_t n = *q; f_neg(&FQ, &n.y, &q->y); jac_madd(r, p, &n);
}
static void jac_normalise(jac_t* r, const jac_t* p) {
  if (jac_is_id(p)) { jac_set_id(r); return; }
  fe zi, zi2, zi3; f_inv(&FQ, &zi, &p->z); QSQR(&zi2, &zi); QMUL(&zi3, &zi2, &zi);
  jac_t o; QMUL(&o.x, &p->x, &zi2); QMUL(&o.y, &p->y, &zi3); o.z = FQ.one; *r = o;
}

/* ------------------------------------------------------------------------------------------------ */
/* Exposed single operations (for pinning this file against oracle/bn254_ref.py). Values are 32-byte LE. */
void oracle_fq_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); QMUL(&z, &x, &y); memcpy(out, &z, 32); }
void oracle_fq_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); QADD(&z, &x, &y); memcpy(out, &z, 32); }
void oracle_fq_sub(const uint8_t* a, const uint8_t* b, uint8_t* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); QSUB(&z, &x, &y); memcpy(out, &z, 32); }
void oracle_fr_from_mont(const uint8_t* a, uint8_t* out) { fe x, z; memcpy(&x, a, 32); f_from_mont(&FR, &z, &x); memcpy(out, &z, 32); }
void oracle_jac_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { jac_t x, y, z; memcpy(&x, a, 96); memcpy(&y, b, 96); jac_add(&z, &x, &y); jac_normalise(&z, &z); memcpy(out, &z, 96); }
void oracle_jac_double(const uint8_t* a, uint8_t* out) { jac_t x, z; memcpy(&x, a, 96); jac_double(&z, &x); jac_normalise(&z, &z); memcpy(out, &z, 96); }

/* ------------------------------------------------------------------------------------------------ */
/* Deterministic generator: identical to oracle/bn254_ref.py gen_point / gen_scalar. */
static inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
static inline uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t ctr) {
  return splitmix64(splitmix64(seed ^ (stream << 56)) + ctr);
}
static void gen_point(uint64_t seed, uint64_t i, aff_t* out) {
  static const fe SQRT_E = {{0x4F082305B61F3F52ull, 0x65E05AA45A1C72A3ull, 0x6E14116DA0605617ull, 0x0C19139CB84C680Aull}};
  for (uint64_t attempt = 0; attempt < 64; ++attempt) {
    fe raw; for (int k = 0; k < 4; ++k) raw.v[k] = rnd64(seed, 0, (i * 64 + attempt) * 4 + k);
    int sign = (int)(raw.v[3] >> 63);
    raw.v[3] &= 0x3FFFFFFFFFFFFFFFull;
    if (fe_geq(&raw, &FQ.mod)) continue;
    fe x, rhs, y, y2, three = {{3, 0, 0, 0}}, t3;
    f_to_mont(&FQ, &x, &raw); f_to_mont(&FQ, &t3, &three);
    QSQR(&rhs, &x); QMUL(&rhs, &rhs, &x); QADD(&rhs, &rhs, &t3);
    f_pow(&FQ, &y, &rhs, &SQRT_E); QSQR(&y2, &y);
    if (!fe_eq(&y2, &rhs)) continue;
    if (sign) f_neg(&FQ, &y, &y);
    out->x = x; out->y = y; return;
  }
  memset(out, 0, sizeof(*out));
}
static void gen_scalar(uint64_t seed, uint64_t i, int mont, fe* out) {
  fe raw;
  int ok = 0;
  for (uint64_t attempt = 0; attempt < 16 && !ok; ++attempt) {   /* rejection sampling: uniform mod r */
    for (int k = 0; k < 4; ++k) raw.v[k] = rnd64(seed, 1, (i * 16 + attempt) * 4 + k);
    raw.v[3] &= 0x3FFFFFFFFFFFFFFFull;
    ok = !fe_geq(&raw, &FR.mod);
  }
  if (!ok) raw_sub(&raw, &raw, &FR.mod);
  if (mont) f_to_mont(&FR, out, &raw); else *out = raw;
}

typedef struct { uint64_t seed; size_t lo, hi; int mont; aff_t* pts; fe* sc; } gen_job;
static void* gen_worker(void* arg) {
  gen_job* j = (gen_job*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) { gen_point(j->seed, i, &j->pts[i]); gen_scalar(j->seed, i, j->mont, &j->sc[i]); }
  return NULL;
}
void oracle_gen_instance(uint64_t seed, size_t n, int scalars_mont, uint8_t* points64, uint8_t* scalars32, int threads) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)(n ? n : 1);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  gen_job* jobs = (gen_job*)malloc(sizeof(gen_job) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (gen_job){seed, n * t / threads, n * (t + 1) / threads, scalars_mont, (aff_t*)points64, (fe*)scalars32};
    pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}

/* ------------------------------------------------------------------------------------------------ */
static inline uint32_t get_bits(const fe* k, uint32_t start, uint32_t width) {
  if (start >= 256) return 0;
  uint32_t w = start >> 6, b = start & 63;
  uint64_t lo = k->v[w] >> b;
  if (b && w + 1 < 4) lo |= k->v[w + 1] << (64 - b);
  return (uint32_t)(lo & ((width >= 32) ? 0xFFFFFFFFull : ((1ull << width) - 1)));
}

/* The reference pipeline restated serially (msm.rs:189-217): digits (prepare_buckets_indices.rs:92-118),
 * bucket sums (bucket_wise_accumulation.rs:662-681, here by direct bucket indexing instead of sort +
 * segmented sum -- same bucket contents), weighted window sums (sum_reduction.rs:358-378, via running
 * sums), Horner (final_accumulation.rs:19-39).  window_size 0 = reference policy (3 if n < 32 else 15). */
int oracle_msm_reference_pipeline(const uint8_t* scalars32_mont, const uint8_t* points64, size_t n,
                                  uint32_t window_size, uint8_t* out96) {
  if (n == 0) return 1;
  const uint32_t c = window_size ? window_size : (n < 32 ? 3 : 15);
  const uint32_t W = (254 + c - 1) / c, bl = (1u << c) - 1;
  const aff_t* pts = (const aff_t*)points64;
  jac_t* buckets = (jac_t*)malloc(sizeof(jac_t) * (size_t)bl);
  jac_t total; jac_set_id(&total);
  fe* ks = (fe*)malloc(sizeof(fe) * n);
  for (size_t i = 0; i < n; ++i) { fe m; memcpy(&m, scalars32_mont + 32 * i, 32); f_from_mont(&FR, &ks[i], &m); }
  for (int w = (int)W - 1; w >= 0; --w) {
    for (uint32_t b = 0; b < bl; ++b) jac_set_id(&buckets[b]);
    for (size_t i = 0; i < n; ++i) {
      uint32_t m = get_bits(&ks[i], (uint32_t)w * c, c);
      if (m && !aff_is_id(&pts[i])) jac_madd(&buckets[m - 1], &buckets[m - 1], &pts[i]);
    }
    jac_t run, acc; jac_set_id(&run); jac_set_id(&acc);
    for (int b = (int)bl - 1; b >= 0; --b) { jac_add(&run, &run, &buckets[b]); jac_add(&acc, &acc, &run); }
    for (uint32_t i = 0; i < c; ++i) jac_double(&total, &total);
    jac_add(&total, &total, &acc);
  }
  jac_normalise(&total, &total);
  memcpy(out96, &total, 96);
  free(buckets); free(ks);
  return 0;
}

/* halo2curves-style serial bucket method on one slice (signed Booth digits). */
static void msm_serial(const fe* ks, const aff_t* pts, size_t n, jac_t* out) {
  jac_set_id(out);
  if (n == 0) return;
  uint32_t c;
  if (n < 4) c = 1; else if (n < 32) c = 3; else c = (uint32_t)ceil(log((double)n));
  const uint32_t W = (256 + c - 1) / c + 1;        /* one extra window absorbs the Booth carry */
  const uint32_t nbk = 1u << (c - 1);
  jac_t* buckets = (jac_t*)malloc(sizeof(jac_t) * nbk);
  for (int w = (int)W - 1; w >= 0; --w) {
    for (uint32_t i = 0; i < c; ++i) jac_double(out, out);
    for (uint32_t b = 0; b < nbk; ++b) jac_set_id(&buckets[b]);
    for (size_t i = 0; i < n; ++i) {
      /* Booth recoding: digit = bits[w*c-1 .. w*c+c-1] -> signed value in [-2^(c-1), 2^(c-1)] */
      const uint32_t start = (uint32_t)w * c;
      uint32_t raw;
      if (start == 0) raw = get_bits(&ks[i], 0, c) << 1; else raw = get_bits(&ks[i], start - 1, c + 1);
      const uint32_t sign = (raw >> c) & 1;
      int32_t d = (int32_t)((raw + 1) >> 1);
      if (sign) d = d - (int32_t)(1u << c);         /* now in [-2^(c-1), 2^(c-1)] */
      if (d == 0 || aff_is_id(&pts[i])) continue;
      const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
      jac_madd_signed(&buckets[mag - 1], &buckets[mag - 1], &pts[i], d < 0);
    }
    jac_t run, acc; jac_set_id(&run); jac_set_id(&acc);
    for (int b = (int)nbk - 1; b >= 0; --b) { jac_add(&run, &run, &buckets[b]); jac_add(&acc, &acc, &run); }
    jac_add(out, out, &acc);
  }
  free(buckets);
}

typedef struct { const fe* ks; const aff_t* pts; size_t n; jac_t res; } msm_job;
static void* msm_worker(void* arg) { msm_job* j = (msm_job*)arg; msm_serial(j->ks, j->pts, j->n, &j->res); return NULL; }
typedef struct { const uint8_t* src; fe* dst; size_t lo, hi; } dm_job;
static void* dm_worker(void* arg) {
  dm_job* j = (dm_job*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) { fe m; memcpy(&m, j->src + 32 * i, 32); f_from_mont(&FR, &j->dst[i], &m); }
  return NULL;
}

/* The round-1 baseline, kept as a second, independent checker: the points are split over the threads and every
 * thread runs the serial bucket method above on its slice (the shape of halo2curves' older best_multiexp). */
int oracle_msm_chunked(const uint8_t* scalars32_mont, const uint8_t* points64, size_t n, int threads, uint8_t* out96) {
  if (n == 0) return 1;
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)n;
  fe* ks = (fe*)malloc(sizeof(fe) * n);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  dm_job* dj = (dm_job*)malloc(sizeof(dm_job) * threads);
  for (int t = 0; t < threads; ++t) {
    dj[t] = (dm_job){scalars32_mont, ks, n * t / threads, n * (t + 1) / threads};
    pthread_create(&th[t], NULL, dm_worker, &dj[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  msm_job* jobs = (msm_job*)malloc(sizeof(msm_job) * threads);
  const aff_t* pts = (const aff_t*)points64;
  for (int t = 0; t < threads; ++t) {
    size_t lo = n * t / threads, hi = n * (t + 1) / threads;
    jobs[t].ks = ks + lo; jobs[t].pts = pts + lo; jobs[t].n = hi - lo;
    pthread_create(&th[t], NULL, msm_worker, &jobs[t]);
  }
  jac_t total; jac_set_id(&total);
  for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); jac_add(&total, &total, &jobs[t].res); }
  jac_normalise(&total, &total);
  memcpy(out96, &total, 96);
  free(jobs); free(dj); free(th); free(ks);
  return 0;
}


/* ------------------------------------------------------------------------------------------------
 * halo2curves 0.7 `msm::msm_best`, restated from memory of its published source (the crate is not in the
 * container; see the header): one rayon task per WINDOW; Booth-recoded digits; buckets are kept AFFINE and up
 * to 64 pending (bucket += +/-base) updates are executed together with one shared field inversion (batch_add:
 * prefix products of the denominators, one inversion, back-substitution -- 6 multiplications per addition); an
 * update whose bucket already has a pending entry in the schedule goes to a Jacobian side bucket instead
 * ("greedy accumulation"); window value by summation by parts over (Jacobian side bucket + affine bucket);
 * the window is shifted into place by c*w doublings; window values are added.  Beyond halo2curves (which can
 * keep only `windows` = 19 cores busy at 2^20 points), `groups` > 1 splits the points into groups that run the
 * same window tasks side by side, so that every core of the node has work; the per-group results are added.
 * Identity bases (0,0) are skipped (halo2curves assumes there are none). */
#define SCHED_BATCH 64

/* modular inverse of a Montgomery residue, result a Montgomery residue: binary extended Euclid on the integer
 * (a R)^-1 = a^-1 R^-1, then two Montgomery products by R^2 bring it to a^-1 R. */
static inline void fe_shr1(fe* a, uint64_t top) {
  a->v[0] = (a->v[0] >> 1) | (a->v[1] << 63); a->v[1] = (a->v[1] >> 1) | (a->v[2] << 63);
  a->v[2] = (a->v[2] >> 1) | (a->v[3] << 63); a->v[3] = (a->v[3] >> 1) | (top << 63);
}
static inline void halve_mod(const field_t* F, fe* x) {   /* x/2 mod p for x < p */
  if (x->v[0] & 1) { uint64_t c = raw_add(x, x, &F->mod); fe_shr1(x, c); } else fe_shr1(x, 0);
}
static void f_inv_fast(const field_t* F, fe* r, const fe* a) {
  fe u = *a, v = F->mod, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  const fe one = {{1, 0, 0, 0}};
  if (fe_is_zero(a)) { *r = *a; return; }
  while (!fe_eq(&u, &one) && !fe_eq(&v, &one)) {
    while (!(u.v[0] & 1)) { fe_shr1(&u, 0); halve_mod(F, &x1); }
    while (!(v.v[0] & 1)) { fe_shr1(&v, 0); halve_mod(F, &x2); }
    if (fe_geq(&u, &v)) { raw_sub(&u, &u, &v); f_sub(F, &x1, &x1, &x2); }
    else { raw_sub(&v, &v, &u); f_sub(F, &x2, &x2, &x1); }
  }
  fe t = fe_eq(&u, &one) ? x1 : x2;
  f_mul(F, &t, &t, &F->r2);      /* a^-1 R^-1 * R^2 * R^-1 = a^-1        */
  f_mul(F, r, &t, &F->r2);       /* a^-1 * R^2 * R^-1     = a^-1 R      */
}

static inline int32_t booth_digit(const fe* k, uint32_t w, uint32_t c) {
  const uint32_t start = w * c;
  uint32_t raw;
  if (start == 0) raw = get_bits(k, 0, c) << 1; else raw = get_bits(k, start - 1, c + 1);
  const uint32_t sign = (raw >> c) & 1;
  int32_t d = (int32_t)((raw + 1) >> 1);
  if (sign) d = d - (int32_t)(1u << c);
  return d;                                         /* in [-2^(c-1), 2^(c-1)] */
}

typedef struct { uint32_t base, buck; int neg; } sched_pt;
typedef struct {
  aff_t* a_bucks;          /* affine buckets, (0,0) = empty */
  jac_t* j_bucks;          /* Jacobian side buckets for updates that collide with a pending one */
  uint8_t* pending;        /* bucket has an entry in the current batch */
  sched_pt set[SCHED_BATCH];
  int ptr;
} schedule_t;

/* executes the pending updates bucket[b] += +/- base with ONE inversion */
static void sched_execute(schedule_t* S, const aff_t* bases) {
  fe t[SCHED_BATCH], z[SCHED_BATCH];
  uint8_t kind[SCHED_BATCH];                        /* 0 = plain store / nothing, 1 = add, 2 = double */
  fe acc = FQ.one;
  const int size = S->ptr;
  for (int i = 0; i < size; ++i) {
    const sched_pt sp = S->set[i];
    aff_t* b = &S->a_bucks[sp.buck];
    const aff_t* q = &bases[sp.base];
    kind[i] = 0;
    if (aff_is_id(b)) continue;                     /* empty bucket: set it in the second pass */
    fe qy = q->y; if (sp.neg) f_neg(&FQ, &qy, &q->y);
    if (fe_eq(&b->x, &q->x)) {
      if (fe_eq(&b->y, &qy)) {                      /* doubling: lambda = 3 x^2 / 2 y */
        fe xx, n3; QSQR(&xx, &q->x); QADD(&n3, &xx, &xx); QADD(&n3, &n3, &xx);
        QADD(&z[i], &qy, &qy); QMUL(&t[i], &acc, &n3); QMUL(&acc, &acc, &z[i]); kind[i] = 2;
      } else { memset(b, 0, sizeof *b); kind[i] = 3; }   /* P + (-P): bucket becomes empty, skip in pass 2 */
      continue;
    }
    fe dy; QSUB(&z[i], &q->x, &b->x); QSUB(&dy, &qy, &b->y);
    QMUL(&t[i], &acc, &dy); QMUL(&acc, &acc, &z[i]); kind[i] = 1;
  }
  fe inv; f_inv_fast(&FQ, &inv, &acc);
  for (int i = size - 1; i >= 0; --i) {
    const sched_pt sp = S->set[i];
    aff_t* b = &S->a_bucks[sp.buck];
    const aff_t* q = &bases[sp.base];
    S->pending[sp.buck] = 0;
    if (kind[i] == 3) continue;
    if (kind[i] == 0) { b->x = q->x; b->y = q->y; if (sp.neg) f_neg(&FQ, &b->y, &q->y); continue; }
    fe lambda, x3, y3, tt;
    QMUL(&lambda, &inv, &t[i]); QMUL(&inv, &inv, &z[i]);
    QSQR(&x3, &lambda); QSUB(&x3, &x3, &b->x); QSUB(&x3, &x3, &q->x);     /* doubling: q.x == b.x */
    QSUB(&tt, &b->x, &x3); QMUL(&y3, &lambda, &tt); QSUB(&y3, &y3, &b->y);
    b->x = x3; b->y = y3;
  }
  S->ptr = 0;
}

typedef struct {
  const fe* ks; const aff_t* pts; size_t n;   /* the task's point group */
  uint32_t w, c;
  jac_t res;
} win_task;

static void run_window_task(win_task* T) {
  const uint32_t c = T->c, nbk = 1u << (c - 1);
  schedule_t S;
  S.a_bucks = (aff_t*)calloc(nbk, sizeof(aff_t));
  S.j_bucks = (jac_t*)malloc(sizeof(jac_t) * nbk);
  S.pending = (uint8_t*)calloc(nbk, 1);
  S.ptr = 0;
  for (uint32_t b = 0; b < nbk; ++b) jac_set_id(&S.j_bucks[b]);
  for (size_t i = 0; i < T->n; ++i) {
    const int32_t d = booth_digit(&T->ks[i], T->w, c);
    if (d == 0 || aff_is_id(&T->pts[i])) continue;
    const uint32_t buck = (uint32_t)(d < 0 ? -d : d) - 1;
    if (S.pending[buck]) {                                     /* greedy accumulation on the side bucket */
      jac_madd_signed(&S.j_bucks[buck], &S.j_bucks[buck], &T->pts[i], d < 0);
    } else {
      S.set[S.ptr++] = (sched_pt){(uint32_t)i, buck, d < 0};
      S.pending[buck] = 1;
      if (S.ptr == SCHED_BATCH) sched_execute(&S, T->pts);
    }
  }
  sched_execute(&S, T->pts);
  jac_t run, acc; jac_set_id(&run); jac_set_id(&acc);
  for (int b = (int)nbk - 1; b >= 0; --b) {                    /* summation by parts */
    if (!jac_is_id(&S.j_bucks[b])) jac_add(&run, &run, &S.j_bucks[b]);
    if (!aff_is_id(&S.a_bucks[b])) jac_madd(&run, &run, &S.a_bucks[b]);
    jac_add(&acc, &acc, &run);
  }
  for (uint32_t i = 0; i < c * T->w; ++i) jac_double(&acc, &acc);
  T->res = acc;
  free(S.a_bucks); free(S.j_bucks); free(S.pending);
}

typedef struct { win_task* tasks; size_t count; size_t next; pthread_mutex_t mu; } task_pool;
static void* pool_worker(void* arg) {
  task_pool* P = (task_pool*)arg;
  for (;;) {
    pthread_mutex_lock(&P->mu);
    const size_t k = P->next < P->count ? P->next++ : (size_t)-1;
    pthread_mutex_unlock(&P->mu);
    if (k == (size_t)-1) return NULL;
    run_window_task(&P->tasks[k]);
  }
}

/* groups = 0: as many point groups as keep `threads` busy (threads / windows, at least 1); groups = 1 is the
 * halo2curves shape exactly (window tasks only). */
int oracle_msm_best_ex(const uint8_t* scalars32_mont, const uint8_t* points64, size_t n, int threads, int groups,
                       uint8_t* out96, uint32_t* info /* [c, windows, groups, threads used] or NULL */) {
  if (n == 0) return 1;
  if (threads < 1) threads = 1;
  fe* ks = (fe*)malloc(sizeof(fe) * n);
  {
    int dt = threads; if ((size_t)dt > n) dt = (int)n;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * dt);
    dm_job* dj = (dm_job*)malloc(sizeof(dm_job) * dt);
    for (int t = 0; t < dt; ++t) {
      dj[t] = (dm_job){scalars32_mont, ks, n * t / dt, n * (t + 1) / dt};
      pthread_create(&th[t], NULL, dm_worker, &dj[t]);
    }
    for (int t = 0; t < dt; ++t) pthread_join(th[t], NULL);
    free(th); free(dj);
  }
  uint32_t c;
  if (groups < 1) groups = 0;
  /* the window follows the size of a GROUP (each group is an MSM of its own) */
  uint32_t W0 = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t gn = groups > 0 ? (n + groups - 1) / groups : n;
    if (gn < 4) c = 1; else if (gn < 32) c = 3; else c = (uint32_t)ceil(log((double)gn));
    W0 = 254 / c + 1;
    if (groups > 0) break;
    /* enough (group, window) tasks that the last round of the thread pool is well filled: >= 3 per thread */
    groups = (3 * threads + (int)W0 - 1) / (int)W0; if (groups < 1) groups = 1;
    if (threads == 1) groups = 1;
    if ((size_t)groups > n) groups = (int)n;
  }
  const uint32_t W = W0;
  const size_t count = (size_t)groups * W;
  win_task* tasks = (win_task*)malloc(sizeof(win_task) * count);
  const aff_t* pts = (const aff_t*)points64;
  for (int g = 0; g < groups; ++g) {
    const size_t lo = n * g / groups, hi = n * (g + 1) / groups;
    for (uint32_t w = 0; w < W; ++w)                           /* high windows first: they cost the most doublings */
      tasks[(size_t)g * W + w] = (win_task){ks + lo, pts + lo, hi - lo, W - 1 - w, c, {{{0}}}};
  }
  task_pool P = {tasks, count, 0, PTHREAD_MUTEX_INITIALIZER};
  int used = threads; if ((size_t)used > count) used = (int)count;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * used);
  for (int t = 0; t < used; ++t) pthread_create(&th[t], NULL, pool_worker, &P);
  for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  jac_t total; jac_set_id(&total);
  for (size_t k = 0; k < count; ++k) jac_add(&total, &total, &tasks[k].res);
  jac_normalise(&total, &total);
  memcpy(out96, &total, 96);
  if (info) { info[0] = c; info[1] = W; info[2] = (uint32_t)groups; info[3] = (uint32_t)used; }
  free(th); free(tasks); free(ks);
  return 0;
}

int oracle_msm_best(const uint8_t* scalars32_mont, const uint8_t* points64, size_t n, int threads, uint8_t* out96) {
  return oracle_msm_best_ex(scalars32_mont, points64, n, threads, 0, out96, NULL);
}

/* sum_i k_i * P_i by plain double-and-add (the definition); for small n only. */
int oracle_msm_naive(const uint8_t* scalars32_mont, const uint8_t* points64, size_t n, uint8_t* out96) {
  const aff_t* pts = (const aff_t*)points64;
  jac_t total; jac_set_id(&total);
  for (size_t i = 0; i < n; ++i) {
    fe m, k; memcpy(&m, scalars32_mont + 32 * i, 32); f_from_mont(&FR, &k, &m);
    jac_t acc; jac_set_id(&acc);
    if (!aff_is_id(&pts[i])) {
      for (int b = 255; b >= 0; --b) {
        jac_double(&acc, &acc);
        if ((k.v[b >> 6] >> (b & 63)) & 1) jac_madd(&acc, &acc, &pts[i]);
      }
    }
    jac_add(&total, &total, &acc);
  }
  jac_normalise(&total, &total);
  memcpy(out96, &total, 96);
  return 0;
}

/* Size-independent check for big instances: points P_i = (a0 + i*d) * G built by repeated addition
 * (affine, via one inversion per point block), so that MSM(k, P) must equal (sum k_i (a0 + i d) mod r) G.
 * Writes the points (64 B affine Montgomery) and returns the expected result in out96. */
typedef struct { size_t lo, hi; const fe* a0; const fe* d; aff_t* pts; } dl_job;
static void scalar_mul_gen(const fe* k_canon, jac_t* out) {
  aff_t g; fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
  f_to_mont(&FQ, &g.x, &one); f_to_mont(&FQ, &g.y, &two);
  jac_set_id(out);
  for (int b = 255; b >= 0; --b) {
    jac_double(out, out);
    if ((k_canon->v[b >> 6] >> (b & 63)) & 1) jac_madd(out, out, &g);
  }
}
static void fr_mul_canon(fe* r, const fe* a, const fe* b) {   /* canonical in/out */
  fe am, bm, pm; f_to_mont(&FR, &am, a); f_to_mont(&FR, &bm, b); f_mul(&FR, &pm, &am, &bm); f_from_mont(&FR, r, &pm);
}
static void* dl_worker(void* arg) {
  dl_job* j = (dl_job*)arg;
  /* start = (a0 + lo*d) G, step = d G */
  fe lo_fe = {{(uint64_t)j->lo, 0, 0, 0}}, t, s;
  fr_mul_canon(&t, &lo_fe, j->d); f_add(&FR, &s, j->a0, &t);
  jac_t cur, step; scalar_mul_gen(&s, &cur); scalar_mul_gen(j->d, &step);
  jac_t stepn; jac_normalise(&stepn, &step);
  aff_t stepa = {stepn.x, stepn.y};
  enum { BLK = 256 };
  jac_t blk[BLK]; fe pref[BLK];
  for (size_t base = j->lo; base < j->hi; base += BLK) {
    size_t m = j->hi - base < BLK ? j->hi - base : BLK;
    for (size_t i = 0; i < m; ++i) { blk[i] = cur; if (jac_is_id(&cur)) jac_set_id(&cur); jac_madd(&cur, &cur, &stepa); }
    /* batch-normalise the block (Montgomery's trick); identity entries keep z = 0 */
    fe acc = FQ.one;
    for (size_t i = 0; i < m; ++i) { pref[i] = acc; if (!jac_is_id(&blk[i])) QMUL(&acc, &acc, &blk[i].z); }
    fe inv; f_inv(&FQ, &inv, &acc);
    for (size_t i = m; i-- > 0;) {
      if (jac_is_id(&blk[i])) { memset(&j->pts[base + i], 0, sizeof(aff_t)); continue; }
      fe zi, zi2, zi3; QMUL(&zi, &inv, &pref[i]); QMUL(&inv, &inv, &blk[i].z);
      QSQR(&zi2, &zi); QMUL(&zi3, &zi2, &zi);
      QMUL(&j->pts[base + i].x, &blk[i].x, &zi2); QMUL(&j->pts[base + i].y, &blk[i].y, &zi3);
    }
  }
  return NULL;
}
int oracle_dlog_instance(const uint8_t* a0_canon32, const uint8_t* d_canon32, const uint8_t* scalars32_mont, size_t n,
                         int threads, uint8_t* points64_out, uint8_t* expected96) {
  if (n == 0) return 1;
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)n;
  fe a0, d; memcpy(&a0, a0_canon32, 32); memcpy(&d, d_canon32, 32);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  dl_job* jobs = (dl_job*)malloc(sizeof(dl_job) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (dl_job){n * t / threads, n * (t + 1) / threads, &a0, &d, (aff_t*)points64_out};
    pthread_create(&th[t], NULL, dl_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  /* expected scalar: sum k_i (a0 + i d) mod r, all in Montgomery form */
  fe a0m, dm, im, sum; f_to_mont(&FR, &a0m, &a0); f_to_mont(&FR, &dm, &d);
  memset(&sum, 0, sizeof(sum));
  fe cur = a0m;                                   /* (a0 + i d) in Montgomery form */
  for (size_t i = 0; i < n; ++i) {
    fe k, t; memcpy(&k, scalars32_mont + 32 * i, 32);
    f_mul(&FR, &t, &k, &cur); f_add(&FR, &sum, &sum, &t);
    f_add(&FR, &cur, &cur, &dm);
  }
  (void)im;
  fe e; f_from_mont(&FR, &e, &sum);
  jac_t res; scalar_mul_gen(&e, &res); jac_normalise(&res, &res);
  memcpy(expected96, &res, 96);
  free(th); free(jobs);
  return 0;
}
