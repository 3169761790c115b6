// BN254 G1 group law (y^2 = x^3 + 3) on Montgomery coordinates, for gfx950 kernels and host code.
//
// Replaces the reference's ECPoint (src/metal/shader/curves/ec_point.h.metal:3-176) and the
// serialised mirrors of src/metal/shader/curves/ser_point.h.metal:14-44.  Differences by design:
//   * bases are kept AFFINE (64 B) and added with the mixed formula madd-2007-bl (7M+4S) instead of
//     the reference's full Jacobian add-2007-bl (11M+5S) on points whose Z is known to be one
//     (SURVEY Appendix B item 9);
//   * doubling uses dbl-2009-l (2M+5S, a = 0) instead of dbl-2007-bl with per-call constant
//     conversion (ec_point.h.metal:141-175);
//   * identity: Jacobian Z == 0 (as the reference, ec_point.h.metal:106-108); affine (0,0), which is
//     what halo2curves stores for the identity -- the reference mishandles it (Appendix B item 1).
#pragma once
#include "bn254_fq.hip.h"

namespace msm_amd {

struct Affine {     // 64 bytes: x, y in Montgomery form, little-endian limbs == bn256::G1Affine
  u256 x, y;
};

struct Jacobian {   // 96 bytes: X, Y, Z Montgomery, (X/Z^2, Y/Z^3); Z == 0 is the identity
  u256 x, y, z;
};

MSM_HD bool affine_is_identity(const Affine& p) {
  uint32_t o = 0;
  MSM_UNROLL for (int i = 0; i < 8; ++i) o |= p.x.v[i] | p.y.v[i];
  return o == 0;
}

MSM_HD bool jac_is_identity(const Jacobian& p) { return u256_is_zero(p.z); }

MSM_HD Jacobian jac_identity() {
  Jacobian r;
  r.x = Fq::one();
  r.y = Fq::one();
  r.z = u256_zero();
  return r;
}

MSM_HD Jacobian jac_from_affine(const Affine& p) {
  Jacobian r;
  if (affine_is_identity(p)) return jac_identity();
  r.x = p.x;
  r.y = p.y;
  r.z = Fq::one();
  return r;
}

// dbl-2009-l, a = 0.  2M + 5S.
MSM_HD Jacobian jac_double(const Jacobian& p) {
  if (jac_is_identity(p)) return p;
  const u256 A = Fq::sqr(p.x);
  const u256 B = Fq::sqr(p.y);
  const u256 C = Fq::sqr(B);
  u256 t = Fq::add(p.x, B);
  t = Fq::sqr(t);
  t = Fq::sub(t, A);
  t = Fq::sub(t, C);
  const u256 D = Fq::dbl(t);
  const u256 E = Fq::add(Fq::dbl(A), A);
  const u256 F = Fq::sqr(E);
  Jacobian r;
  r.x = Fq::sub(F, Fq::dbl(D));
  const u256 C8 = Fq::dbl(Fq::dbl(Fq::dbl(C)));
  r.y = Fq::sub(Fq::mul(E, Fq::sub(D, r.x)), C8);
  r.z = Fq::dbl(Fq::mul(p.y, p.z));
  return r;
}

// Full Jacobian addition add-2007-bl (11M + 5S) with the same case analysis as the reference's
// operator+ (ec_point.h.metal:13-69): identity operands, equal points -> doubling; P + (-P) gives
// Z3 = 0 through H = 0.
MSM_HD Jacobian jac_add(const Jacobian& p, const Jacobian& q) {
  if (jac_is_identity(p)) return q;
  if (jac_is_identity(q)) return p;
  const u256 Z1Z1 = Fq::sqr(p.z);
  const u256 Z2Z2 = Fq::sqr(q.z);
  const u256 U1 = Fq::mul(p.x, Z2Z2);
  const u256 U2 = Fq::mul(q.x, Z1Z1);
  const u256 S1 = Fq::mul(Fq::mul(p.y, q.z), Z2Z2);
  const u256 S2 = Fq::mul(Fq::mul(q.y, p.z), Z1Z1);
  const u256 H = Fq::sub(U2, U1);
  const u256 rr = Fq::sub(S2, S1);
  if (u256_is_zero(H)) {
    if (u256_is_zero(rr)) return jac_double(p);
    return jac_identity();
  }
  const u256 I = Fq::sqr(Fq::dbl(H));
  const u256 J = Fq::mul(H, I);
  const u256 r2 = Fq::dbl(rr);
  const u256 V = Fq::mul(U1, I);
  Jacobian r;
  r.x = Fq::sub(Fq::sub(Fq::sqr(r2), J), Fq::dbl(V));
  r.y = Fq::sub(Fq::mul(r2, Fq::sub(V, r.x)), Fq::dbl(Fq::mul(S1, J)));
  u256 zz = Fq::sqr(Fq::add(p.z, q.z));
  zz = Fq::sub(Fq::sub(zz, Z1Z1), Z2Z2);
  r.z = Fq::mul(zz, H);
  return r;
}

// Mixed addition madd-2007-bl: Jacobian + affine (Z2 = 1).  7M + 4S.
// `q` must not be the affine identity (callers filter (0,0)).
MSM_HD Jacobian jac_madd(const Jacobian& p, const Affine& q) {
  if (jac_is_identity(p)) {
    Jacobian r;
    r.x = q.x;
    r.y = q.y;
    r.z = Fq::one();
    return r;
  }
  const u256 Z1Z1 = Fq::sqr(p.z);
  const u256 U2 = Fq::mul(q.x, Z1Z1);
  const u256 S2 = Fq::mul(Fq::mul(q.y, p.z), Z1Z1);
  const u256 H = Fq::sub(U2, p.x);
  const u256 rr = Fq::sub(S2, p.y);
  if (u256_is_zero(H)) {
    if (u256_is_zero(rr)) return jac_double(p);
    return jac_identity();
  }
  const u256 HH = Fq::sqr(H);
  const u256 I = Fq::dbl(Fq::dbl(HH));
  const u256 J = Fq::mul(H, I);
  const u256 r2 = Fq::dbl(rr);
  const u256 V = Fq::mul(p.x, I);
  Jacobian r;
  r.x = Fq::sub(Fq::sub(Fq::sqr(r2), J), Fq::dbl(V));
  r.y = Fq::sub(Fq::mul(r2, Fq::sub(V, r.x)), Fq::dbl(Fq::mul(p.y, J)));
  u256 zz = Fq::sqr(Fq::add(p.z, H));
  r.z = Fq::sub(Fq::sub(zz, Z1Z1), HH);
  return r;
}

// k * P, double-and-add MSB first over a 256-bit canonical scalar (reference: operate_with_self,
// ec_point.h.metal:110-131).
MSM_HD Jacobian jac_scalar_mul(const Jacobian& p, const u256& k) {
  Jacobian acc = jac_identity();
  for (int i = 255; i >= 0; --i) {
    acc = jac_double(acc);
    if ((k.v[i >> 5] >> (i & 31)) & 1u) acc = jac_add(acc, p);
  }
  return acc;
}

// k * P for a small (<= 32 bit) multiplier (reference: operate_with_self(uint64), ec_point.h.metal:79-92).
MSM_HD Jacobian jac_mul_u32(const Jacobian& p, uint32_t k) {
  Jacobian acc = jac_identity();
  for (int i = 31; i >= 0; --i) {
    acc = jac_double(acc);
    if ((k >> i) & 1u) acc = jac_add(acc, p);
  }
  return acc;
}

}  // namespace msm_amd
