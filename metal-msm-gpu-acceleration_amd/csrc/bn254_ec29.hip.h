// BN254 G1 group law on the 29-bit-limb internal representation (bn254_fq29.hip.h) -- the arithmetic of
// the hot kernels (accumulate, combine, window reduction).  Points in HBM:
//   AffI  80 bytes: x, y (9 limbs each) + 2 pad words -> five 16-byte loads per gathered base
//   JacI 112 bytes: X, Y, Z (9 limbs each) + 1 pad word; Z limbs all zero = identity
// Formulas: mixed addition madd-2004-hmv (8M + 3S), full addition add-1998-cmo-2 shape (12M + 4S); both
// avoid the small-constant multiples of the 2007-bl variants, which would cost extra normalisations on
// lazily reduced limbs.  The reference's operator+ (ec_point.h.metal:13-69) is add-2007-bl on 32-bit limbs.
//
// Value bounds maintained for every point that is stored or carried in a register between additions
// (multiples of p; see the bounds contract in bn254_fq29.hip.h):
//      X < 9.5 p      Y < 6 p      Z < 2 p      limbs 0..7 < 2^29 + 8
// The exceptional cases (equal points -> doubling, opposite points -> identity) are detected with the
// one-limb filter Fq29::maybe_zero, confirmed exactly (Fq29::is_zero_exact) and resolved by jaci_double / the
// identity, all on the same limbs.
#pragma once
#include "bn254_ec.hip.h"
#include "bn254_fq29.hip.h"

namespace msm_amd {

struct AffI {
  fe29 x, y;
  uint32_t pad[2];
};
struct JacI {
  fe29 x, y, z;
  uint32_t pad;
};
static_assert(sizeof(AffI) == 80, "AffI must be 80 bytes");
static_assert(sizeof(JacI) == 112, "JacI must be 112 bytes");

MSM_HD bool affi_is_identity(const AffI& p) {
  uint32_t o = 0;
  MSM_UNROLL for (int i = 0; i < 9; ++i) o |= p.x.l[i] | p.y.l[i];
  return o == 0;
}
MSM_HD bool jaci_is_identity(const JacI& p) { return Fq29::is_zero_limbs(p.z); }

MSM_HD JacI jaci_identity() {
  JacI r;
  r.x = Fq29::one();
  r.y = Fq29::one();
  r.z = Fq29::zero();
  r.pad = 0;
  return r;
}

MSM_HD JacI jaci_from_affi(const AffI& q) {   // q must not be the identity
  JacI r;
  r.x = q.x;
  r.y = q.y;
  r.z = Fq29::one();
  r.pad = 0;
  return r;
}

// ---- conversions to / from the external (8 x u32, R = 2^256) representation ---------------------------
MSM_HD AffI affi_from_ext(const Affine& p) {
  AffI r;
  if (affine_is_identity(p)) {
    r.x = Fq29::zero();
    r.y = Fq29::zero();
  } else {
    r.x = Fq29::from_ext(p.x);
    r.y = Fq29::from_ext(p.y);
  }
  r.pad[0] = r.pad[1] = 0;
  return r;
}

MSM_HD JacI jaci_from_ext(const Jacobian& p) {
  if (jac_is_identity(p)) return jaci_identity();
  JacI r;
  r.x = Fq29::from_ext(p.x);
  r.y = Fq29::from_ext(p.y);
  r.z = Fq29::from_ext(p.z);
  r.pad = 0;
  return r;
}

MSM_HD Jacobian jaci_to_ext(const JacI& p) {
  if (jaci_is_identity(p)) return jac_identity();
  Jacobian r;
  r.x = Fq29::to_ext(p.x);
  r.y = Fq29::to_ext(p.y);
  r.z = Fq29::to_ext(p.z);
  return r;
}

// ---- rare path: doubling (reached only when an addition meets two equal points) ----------------------
// dbl-2009-l (a = 0), 2M + 5S, plus three "squash" multiplications by rho mod p that bring lazily grown
// values back under the stored-point bounds.  Cost is irrelevant (exceptional case); it is written on the
// same 29-bit limbs so that the hot kernels contain no second field implementation and no function call.
MSM_HD fe29 fq29_squash(const fe29& a) { return Fq29::mul(a, Fq29::one()); }   // same value, < 1.3 p

MSM_HD JacI jaci_double(const JacI& p) {   // p not the identity
  const fe29 A = Fq29::sqr(p.x);
  const fe29 B = Fq29::sqr(p.y);
  const fe29 C = Fq29::sqr(B);
  const fe29 tt = Fq29::sqr(Fq29::add(p.x, B));
  const fe29 Dh = Fq29::norm(Fq29::sub<K8E31>(tt, Fq29::add(A, C)));       // (X+B)^2 - A - C   < 9.8 p
  const fe29 D = fq29_squash(Fq29::add(Dh, Dh));                            // D = 2 (...)        < 1.2 p
  const fe29 E = Fq29::norm(Fq29::add(A, Fq29::add(A, A)));                 // 3 A               < 4.7 p
  const fe29 F = Fq29::sqr(E);
  JacI r;
  r.x = Fq29::norm(Fq29::sub<K4E30>(F, Fq29::add(D, D)));                   // F - 2D            < 5.2 p
  const fe29 T = Fq29::norm(Fq29::sub<K8E30>(D, r.x));                      // D - X3            < 9.3 p
  const fe29 C2 = Fq29::add(C, C);
  const fe29 C4 = Fq29::norm(Fq29::add(C2, C2));
  const fe29 C8 = Fq29::norm(Fq29::add(C4, C4));                            // 8 C               < 8.2 p
  r.y = fq29_squash(Fq29::norm(Fq29::sub<K16E30>(Fq29::mul(E, T), C8)));    // E (D - X3) - 8C   < 1.2 p
  const fe29 YZ = Fq29::mul(p.y, p.z);
  r.z = fq29_squash(Fq29::add(YZ, YZ));                                     // 2 Y Z
  r.pad = 0;
  return r;
}

// ---- fast paths -------------------------------------------------------------------------------------
// p + q, p Jacobian (not identity), q affine (not identity).  8M + 3S.
MSM_HD JacI jaci_madd(const JacI& p, const AffI& q) {
  const fe29 Z1Z1 = Fq29::sqr(p.z);
  const fe29 U2 = Fq29::mul(q.x, Z1Z1);
  const fe29 S2 = Fq29::mul(Fq29::mul(q.y, p.z), Z1Z1);
  const fe29 H = Fq29::norm(Fq29::sub<K16E30>(U2, p.x));   // < 17.1 p
  const fe29 R = Fq29::norm(Fq29::sub<K8E30>(S2, p.y));    // <  9.1 p
  if (Fq29::maybe_zero(H, 18)) {
    if (Fq29::is_zero_exact(H)) {   // same x: either q == p (double) or q == -p (identity)
      if (Fq29::is_zero_exact(R)) return jaci_double(jaci_from_affi(q));
      return jaci_identity();
    }
  }
  const fe29 HH = Fq29::sqr(H);
  const fe29 HHH = Fq29::mul(H, HH);
  const fe29 V = Fq29::mul(p.x, HH);
  const fe29 RR = Fq29::sqr(R);
  JacI r;
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(HHH, Fq29::add(V, V))));              // < 9.5 p
  const fe29 T1 = Fq29::norm(Fq29::sub<K16E30>(V, r.x));                                  // < 17.2 p
  r.y = Fq29::norm(Fq29::sub<K4E30>(Fq29::mul(R, T1), Fq29::mul(p.y, HHH)));             // < 6 p
  r.z = Fq29::mul(p.z, H);
  r.pad = 0;
  return r;
}

// p + q, both Jacobian, neither the identity.  12M + 4S.
MSM_HD JacI jaci_add_nz(const JacI& p, const JacI& q) {
  const fe29 Z1Z1 = Fq29::sqr(p.z);
  const fe29 Z2Z2 = Fq29::sqr(q.z);
  const fe29 U1 = Fq29::mul(p.x, Z2Z2);
  const fe29 U2 = Fq29::mul(q.x, Z1Z1);
  const fe29 S1 = Fq29::mul(Fq29::mul(p.y, q.z), Z2Z2);
  const fe29 S2 = Fq29::mul(Fq29::mul(q.y, p.z), Z1Z1);
  const fe29 H = Fq29::norm(Fq29::sub<K4E30>(U2, U1));     // < 5.1 p
  const fe29 R = Fq29::norm(Fq29::sub<K4E30>(S2, S1));     // < 5.1 p
  if (Fq29::maybe_zero(H, 6)) {
    if (Fq29::is_zero_exact(H)) {
      if (Fq29::is_zero_exact(R)) return jaci_double(p);
      return jaci_identity();
    }
  }
  const fe29 HH = Fq29::sqr(H);
  const fe29 HHH = Fq29::mul(H, HH);
  const fe29 V = Fq29::mul(U1, HH);
  const fe29 RR = Fq29::sqr(R);
  JacI r;
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(HHH, Fq29::add(V, V))));              // < 9.2 p
  const fe29 T1 = Fq29::norm(Fq29::sub<K16E30>(V, r.x));                                  // < 17.1 p
  r.y = Fq29::norm(Fq29::sub<K4E30>(Fq29::mul(R, T1), Fq29::mul(S1, HHH)));              // < 5.6 p
  r.z = Fq29::mul(Fq29::mul(p.z, q.z), H);
  r.pad = 0;
  return r;
}

// General addition with identity operands allowed.
MSM_HD JacI jaci_add(const JacI& p, const JacI& q) {
  if (jaci_is_identity(p)) return q;
  if (jaci_is_identity(q)) return p;
  return jaci_add_nz(p, q);
}

}  // namespace msm_amd
