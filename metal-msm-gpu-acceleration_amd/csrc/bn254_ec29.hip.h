// BN254 G1 group law on the 29-bit-limb internal representation (bn254_fq29.hip.h) -- the arithmetic of
// the hot kernels (accumulate, combine, window reduction).  Points in HBM:
//   AffPacked 64 bytes: canonical x, y of the internal Montgomery domain, bit-packed -> four 16-byte loads per
//        gathered base, never straddling a 64-byte boundary (80-byte records measured 2.3x slower to gather:
//        tools/microbench/gather_calib.hip); unpacked to 2 x 9 limbs (AffI) in registers
//   PtI  144 bytes: X, Y, ZZ, ZZZ (9 limbs each), extended Jacobian ("XYZZ": x = X/ZZ, y = Y/ZZZ,
//        ZZ^3 = ZZZ^2); ZZ limbs all zero = identity
// Formulas (EFD, short Weierstrass a = 0, XYZZ): mixed addition madd-2008-s (8M + 2S), addition add-2008-s
// (12M + 2S), doubling dbl-2008-s-1.  One squaring less per mixed addition than Jacobian madd (8M + 3S) and
// two less per full addition; no small-constant multiples, which would cost extra normalisations on lazily
// reduced limbs.  The reference's operator+ (ec_point.h.metal:13-69) is Jacobian add-2007-bl (11M + 5S) on
// 32-bit limbs for every pair.
//
// Value bounds maintained for every point that is stored or carried in a register between additions
// (multiples of p; see the bounds contract in bn254_fq29.hip.h):
//      X < 10 p      Y < 6 p      ZZ < 2.8 p      ZZZ < 2 p      limbs 0..7 < 2^29 + 8
// (pti_mmadd sets the X and ZZ figures: its R < 12.1 p and ZZ3 = P^2 with P < 17.1 p).  tools/fq29_bounds.py
// re-derives every bound below by interval arithmetic and checks that no 64-bit column sum can overflow.
// The exceptional cases (equal points -> doubling, opposite points -> identity) are detected with the
// one-limb filter Fq29::maybe_zero, confirmed exactly (Fq29::is_zero_exact) and resolved by pti_double / the
// identity, all on the same limbs.
#pragma once
#include "bn254_ec.hip.h"
#include "bn254_fq29.hip.h"

namespace msm_amd {

struct AffI {   // register form of a base point
  fe29 x, y;
};
struct AffPacked {   // memory form: canonical x, y in the internal Montgomery domain, 2 x 32 bytes
  u256 x, y;
};
struct PtI {
  fe29 x, y, zz, zzz;
};
static_assert(sizeof(AffPacked) == 64, "AffPacked must be 64 bytes");
static_assert(sizeof(PtI) == 144, "PtI must be 144 bytes");

MSM_HD bool affi_is_identity(const AffI& p) {
  uint32_t o = 0;
  MSM_UNROLL for (int i = 0; i < 9; ++i) o |= p.x.l[i] | p.y.l[i];
  return o == 0;
}
MSM_HD bool pti_is_identity(const PtI& p) { return Fq29::is_zero_limbs(p.zz); }

MSM_HD PtI pti_identity() {
  PtI r;
  r.x = Fq29::one();
  r.y = Fq29::one();
  r.zz = Fq29::zero();
  r.zzz = Fq29::zero();
  return r;
}

MSM_HD PtI pti_from_affi(const AffI& q) {   // q must not be the identity
  PtI r;
  r.x = q.x;
  r.y = Fq29::norm(q.y);   // q.y may be an un-normalised negation (accumulate_kernel); a stored point is normalised
  r.zz = Fq29::one();
  r.zzz = Fq29::one();
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
  return r;
}

// The identity is stored as x = 2^256 - 1, y = 0: a canonical coordinate is < p < 2^254, so ONE word tells the two
// apart (affpacked_is_identity) where the unpacked form needs all 18 limbs (affi_is_identity).
MSM_HD AffPacked affi_pack(const AffI& p) {   // p from affi_from_ext (multiplication outputs or exact zeros)
  AffPacked r;
  if (affi_is_identity(p)) {
    MSM_UNROLL for (int i = 0; i < 8; ++i) {
      r.x.v[i] = 0xFFFFFFFFu;
      r.y.v[i] = 0u;
    }
    return r;
  }
  r.x = Fq29::pack_canonical(p.x);
  r.y = Fq29::pack_canonical(p.y);
  return r;
}
MSM_HD bool affpacked_is_identity(const AffPacked& p) { return p.x.v[7] == 0xFFFFFFFFu; }

MSM_HD AffI affi_unpack(const AffPacked& p) {   // the identity comes back as exact zero limbs
  AffI r;
  if (affpacked_is_identity(p)) {
    r.x = Fq29::zero();
    r.y = Fq29::zero();
    return r;
  }
  r.x = Fq29::unpack256(p.x);
  r.y = Fq29::unpack256(p.y);
  return r;
}
// The same without the identity test, for a caller that has made it (accumulate_kernel).
MSM_HD AffI affi_unpack_finite(const AffPacked& p) {
  AffI r;
  r.x = Fq29::unpack256(p.x);
  r.y = Fq29::unpack256(p.y);
  return r;
}

// Jacobian (X, Y, Z) -> (X, Y, Z^2, Z^3)
MSM_HD PtI pti_from_ext(const Jacobian& p) {
  if (jac_is_identity(p)) return pti_identity();
  PtI r;
  r.x = Fq29::from_ext(p.x);
  r.y = Fq29::from_ext(p.y);
  const fe29 z = Fq29::from_ext(p.z);
  r.zz = Fq29::sqr(z);
  r.zzz = Fq29::mul(r.zz, z);
  return r;
}

// (X, Y, ZZ, ZZZ) -> Jacobian (X*ZZ, Y*ZZZ, ZZ): x = X*ZZ/ZZ^2, y = Y*ZZZ/ZZ^3 (ZZ^3 = ZZZ^2).
MSM_HD Jacobian pti_to_ext(const PtI& p) {
  if (pti_is_identity(p)) return jac_identity();
  Jacobian r;
  r.x = Fq29::to_ext(Fq29::mul(p.x, p.zz));
  r.y = Fq29::to_ext(Fq29::mul(p.y, p.zzz));
  r.z = Fq29::to_ext(p.zz);
  return r;
}

// ---- rare path: doubling (reached only when an addition meets two equal points) ----------------------
// dbl-2008-s-1 (a = 0): 6M + 3S.
MSM_HD PtI pti_double(const PtI& p) {   // p not the identity
  const fe29 U = Fq29::add(p.y, p.y);                                        // 2 Y1          < 12 p
  const fe29 V = Fq29::sqr(U);
  const fe29 W = Fq29::mul(U, V);
  const fe29 S = Fq29::mul(p.x, V);
  const fe29 XX = Fq29::sqr(p.x);
  const fe29 M = Fq29::norm(Fq29::add(XX, Fq29::add(XX, XX)));              // 3 X1^2        < 4.7 p
  const fe29 MM = Fq29::sqr(M);
  PtI r;
  r.x = Fq29::norm(Fq29::sub<K4E30>(MM, Fq29::add(S, S)));                  // M^2 - 2S      < 5.2 p
  const fe29 T = Fq29::norm(Fq29::sub<K8E30>(S, r.x));                      // S - X3        < 9.2 p
  r.y = Fq29::norm(Fq29::sub<K4E30>(Fq29::mul(M, T), Fq29::mul(W, p.y)));   // M(S-X3) - W Y1 < 5.3 p
  r.zz = Fq29::mul(V, p.zz);
  r.zzz = Fq29::mul(W, p.zzz);
  return r;
}

// ---- fast paths -------------------------------------------------------------------------------------
// p + q, p XYZZ (not identity), q affine (not identity).  madd-2008-s, 8M + 2S with the two products of Y3
// sharing one Montgomery reduction (7 full multiplications + 1 product-only + 2 squarings).
// `vanished` is set when the sum is the identity (q == -p); the caller tracks that state instead of testing limbs.
// MSM_FQ29_LOCKSTEP: the independent products of the formula run side by side as lockstep product-scanning chains
// (Fq29::fips_multi): (U2, S2), (PP, RR), (PPP, Q, ZZ3), (Y3, ZZZ3).
// The mixed addition in two steps, for a caller that re-uses q's registers in between (accumulate_kernel gathers the
// next base into them): the head consumes q, the tail needs it again only in the exceptional case q == p and asks
// `reload_q` for it.  pti_madd below is head + tail.
// Pins (pin_limbs, bn254_fq29.hip.h) are set ONCE per basic block and the pinned value replaces the old one -- p.zz and
// p.zzz in the head (hence the non-const p), everything the long block of the tail reads at its top -- and the
// multiplications inside run without pins of their own.
MSM_HD void pti_madd_head(PtI& p, const fe29& qx, const fe29& qy, fe29& U2, fe29& S2) {
  p.zz = pin_limbs(p.zz);
  p.zzz = pin_limbs(p.zzz);
  U2 = Fq29::mul_np(pin_limbs(qx), p.zz);
  S2 = Fq29::mul_np(pin_limbs(qy), p.zzz);
}
template <class ReloadQ>
MSM_HD PtI pti_madd_tail(const PtI& p, const fe29& U2, const fe29& S2, ReloadQ&& reload_q, bool& vanished) {
  const fe29 P0 = Fq29::norm(Fq29::sub<K16E30>(U2, p.x));   // < 17.1 p
  const fe29 R0 = Fq29::norm(Fq29::sub<K8E30>(S2, p.y));    // <  9.1 p
  if (Fq29::maybe_zero(P0, 18)) {
    MSM_ISA_MARK("rare mixed_addition");
    if (Fq29::is_zero_exact(P0)) {   // same x: either q == p (double) or q == -p (identity)
      if (Fq29::is_zero_exact(R0)) return pti_double(pti_from_affi(reload_q()));
      vanished = true;
      return pti_identity();
    }
  }
  MSM_ISA_MARK("resume mixed_addition");
  const fe29 P = pin_limbs(P0), R = pin_limbs(R0), X1 = pin_limbs(p.x), Y1 = pin_limbs(p.y), ZZ1 = pin_limbs(p.zz),
             ZZZ1 = pin_limbs(p.zzz);   // one block from here on: no pins inside
  PtI r;
  const fe29 PP = Fq29::sqr_np(P);
  const fe29 PPP = Fq29::mul_np(P, PP);
  const fe29 Q = Fq29::mul_np(X1, PP);
  const fe29 RR = Fq29::sqr_np(R);
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(PPP, Fq29::add(Q, Q))));               // < 9.5 p
  // T and -PPP stay un-normalised (limbs < 2^31 / 2^30.5): their partners R and Y1 in the double product are
  // normalised, and tools/fq29_bounds.py checks that no column of R*T + Y1*(-PPP) can reach 2^64
  const fe29 T = Fq29::sub<K16E30>(Q, r.x);                                               // < 17.2 p
  r.y = Fq29::mul2_np(R, T, Y1, Fq29::neg_wide(PPP));   // R*T - Y1*PPP in one reduction       // < 1.2 p
  r.zz = Fq29::mul_np(ZZ1, PP);
  r.zzz = Fq29::mul_np(ZZZ1, PPP);
  return r;
}
// The affine + affine start likewise: the head consumes q.
MSM_HD void pti_mmadd_head(const fe29& px, const fe29& py, const fe29& qx, const fe29& qy, fe29& P, fe29& R) {
  P = Fq29::norm(Fq29::sub<K16E30>(qx, px));   // < 17.1 p
  R = Fq29::norm(Fq29::sub<K8E30>(qy, py));    // < 12.1 p (q.y may be an un-normalised negation, py not)
}
template <class ReloadQ>
MSM_HD PtI pti_mmadd_tail(const fe29& px, const fe29& py, const fe29& P, const fe29& R, ReloadQ&& reload_q,
                          bool& vanished) {
  if (Fq29::maybe_zero(P, 18)) {
    MSM_ISA_MARK("rare affine_start");
    if (Fq29::is_zero_exact(P)) {   // same x: either q == p (double) or q == -p (identity)
      if (Fq29::is_zero_exact(R)) return pti_double(pti_from_affi(reload_q()));
      vanished = true;
      return pti_identity();
    }
  }
  MSM_ISA_MARK("resume affine_start");
  const fe29 Pp = pin_limbs(P), Rp = pin_limbs(R), X1 = pin_limbs(px), Y1 = pin_limbs(py);   // one block from here on
  const fe29 PP = Fq29::sqr_np(Pp);
  const fe29 PPP = Fq29::mul_np(Pp, PP);
  const fe29 Q = Fq29::mul_np(X1, PP);
  const fe29 RR = Fq29::sqr_np(Rp);
  PtI r;
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(PPP, Fq29::add(Q, Q))));               // < 9.9 p
  const fe29 T = Fq29::norm(Fq29::sub<K16E30>(Q, r.x));                                   // < 17.2 p
  r.y = Fq29::mul2_np(Rp, T, Y1, Fq29::neg_wide(PPP));   // R*T - Y1*PPP in one reduction      // < 1.2 p
  r.zz = PP;                                                                              // < 2.8 p
  r.zzz = PPP;
  return r;
}

MSM_HD PtI pti_madd(const PtI& p, const AffI& q, bool& vanished) {
#if defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_LOCKSTEP)
  fe29 U2, S2;
  Fq29::mul_pair(q.x, p.zz, q.y, p.zzz, U2, S2);
#elif defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_KARATSUBA)
  const fe29 U2 = Fq29::mul_karatsuba(q.x, p.zz);
  const fe29 S2 = Fq29::mul_karatsuba(q.y, p.zzz);
#else
  fe29 U2, S2;
  PtI pp = p;
  pti_madd_head(pp, q.x, q.y, U2, S2);
  return pti_madd_tail(pp, U2, S2, [&]() { return q; }, vanished);
#endif
#if defined(MSM_AMD_EXPERIMENTS) && (defined(MSM_FQ29_LOCKSTEP) || defined(MSM_FQ29_KARATSUBA))
  const fe29 P = Fq29::norm(Fq29::sub<K16E30>(U2, p.x));   // < 17.1 p
  const fe29 R = Fq29::norm(Fq29::sub<K8E30>(S2, p.y));    // <  9.1 p
  if (Fq29::maybe_zero(P, 18)) {
    MSM_ISA_MARK("rare mixed_addition");
    if (Fq29::is_zero_exact(P)) {   // same x: either q == p (double) or q == -p (identity)
      if (Fq29::is_zero_exact(R)) return pti_double(pti_from_affi(q));
      vanished = true;
      return pti_identity();
    }
  }
  MSM_ISA_MARK("resume mixed_addition");
  PtI r;
#if defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_LOCKSTEP)
  fe29 PP, RR, PPP, Q;
  Fq29::sqr_pair(P, R, PP, RR);
  Fq29::mul_triple(P, PP, p.x, PP, p.zz, PP, PPP, Q, r.zz);
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(PPP, Fq29::add(Q, Q))));               // < 9.5 p
  const fe29 T = Fq29::sub<K16E30>(Q, r.x);                                               // < 17.2 p
  Fq29::mul2_mul_pair(R, T, p.y, Fq29::neg_wide(PPP), p.zzz, PPP, r.y, r.zzz);
#elif defined(MSM_AMD_EXPERIMENTS) && defined(MSM_FQ29_KARATSUBA)
  const fe29 PP = Fq29::sqr(P);
  const fe29 PPP = Fq29::mul_karatsuba(P, PP);
  const fe29 Q = Fq29::mul_karatsuba(p.x, PP);
  const fe29 RR = Fq29::sqr(R);
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(PPP, Fq29::add(Q, Q))));               // < 9.5 p
  const fe29 T = Fq29::sub<K16E30>(Q, r.x);                                               // < 17.2 p
  r.y = Fq29::mul2_karatsuba_second(R, T, p.y, Fq29::neg_wide(PPP));
  r.zz = Fq29::mul_karatsuba(p.zz, PP);
  r.zzz = Fq29::mul_karatsuba(p.zzz, PPP);
#endif
  return r;
#endif
}

// p + q, BOTH affine, neither the identity (the second point of every work item: the accumulator was just set
// from an affine base, ZZ = ZZZ = 1).  mmadd-2008-s: pti_madd without the four multiplications by ZZ1 / ZZZ1,
// 4M + 2S.  (px, py) = p with px < 9.5 p, py < 6 p as in pti_madd; q.y may be a lazily negated value < 4 p.
// The result obeys the same bounds as pti_madd's.
MSM_HD PtI pti_mmadd(const fe29& px, const fe29& py, const AffI& q, bool& vanished) {
  fe29 P, R;
  pti_mmadd_head(px, py, q.x, q.y, P, R);
  return pti_mmadd_tail(px, py, P, R, [&]() { return q; }, vanished);
}


// The register-lean (product-scanning) and LDS-parked forms of the mixed addition behind the experimental accumulate
// kernels: experiments/ec29_variants.inc, -DMSM_AMD_EXPERIMENTS builds only.
#if defined(MSM_AMD_EXPERIMENTS)
#include "experiments/ec29_variants.inc"
#endif

MSM_HD PtI pti_madd(const PtI& p, const AffI& q) {
  bool vanished = false;
  return pti_madd(p, q, vanished);
}
MSM_HD PtI pti_mmadd(const fe29& px, const fe29& py, const AffI& q) {
  bool vanished = false;
  return pti_mmadd(px, py, q, vanished);
}

// p + q, both XYZZ, neither the identity.  add-2008-s, 12M + 2S.
MSM_HD PtI pti_add_nz(const PtI& p_in, const PtI& q_in, bool& vanished) {
  // pins once per basic block, multiplications without pins of their own (see pti_madd_head)
  PtI p, q;
  p.x = pin_limbs(p_in.x); p.y = pin_limbs(p_in.y); p.zz = pin_limbs(p_in.zz); p.zzz = pin_limbs(p_in.zzz);
  q.x = pin_limbs(q_in.x); q.y = pin_limbs(q_in.y); q.zz = pin_limbs(q_in.zz); q.zzz = pin_limbs(q_in.zzz);
  const fe29 U1a = Fq29::mul_np(p.x, q.zz);
  const fe29 U2 = Fq29::mul_np(q.x, p.zz);
  const fe29 S1a = Fq29::mul_np(p.y, q.zzz);
  const fe29 S2 = Fq29::mul_np(q.y, p.zzz);
  const fe29 P0 = Fq29::norm(Fq29::sub<K4E30>(U2, U1a));     // < 5.2 p
  const fe29 R0 = Fq29::norm(Fq29::sub<K4E30>(S2, S1a));     // < 5.1 p
  if (Fq29::maybe_zero(P0, 6)) {
    if (Fq29::is_zero_exact(P0)) {
      if (Fq29::is_zero_exact(R0)) return pti_double(p);
      vanished = true;   // q == -p
      return pti_identity();
    }
  }
  const fe29 P = pin_limbs(P0), R = pin_limbs(R0), U1 = pin_limbs(U1a), S1 = pin_limbs(S1a), ZZ1 = pin_limbs(p.zz),
             ZZ2 = pin_limbs(q.zz), ZZZ1 = pin_limbs(p.zzz), ZZZ2 = pin_limbs(q.zzz);
  const fe29 PP = Fq29::sqr_np(P);
  const fe29 PPP = Fq29::mul_np(P, PP);
  const fe29 Q = Fq29::mul_np(U1, PP);
  const fe29 RR = Fq29::sqr_np(R);
  PtI r;
  r.x = Fq29::norm(Fq29::sub<K8E31>(RR, Fq29::add(PPP, Fq29::add(Q, Q))));               // < 9.2 p
  const fe29 T = Fq29::sub<K16E30>(Q, r.x);      // un-normalised, like -PPP (see pti_madd)   // < 17.1 p
  r.y = Fq29::mul2_np(R, T, S1, Fq29::neg_wide(PPP));    // R*T - S1*PPP in one reduction      // < 1.2 p
  r.zz = Fq29::mul_np(Fq29::mul_np(ZZ1, ZZ2), PP);
  r.zzz = Fq29::mul_np(Fq29::mul_np(ZZZ1, ZZZ2), PPP);
  return r;
}
MSM_HD PtI pti_add_nz(const PtI& p, const PtI& q) {
  bool vanished = false;
  return pti_add_nz(p, q, vanished);
}

// General addition with identity operands allowed.
MSM_HD PtI pti_add(const PtI& p, const PtI& q) {
  if (pti_is_identity(p)) return q;
  if (pti_is_identity(q)) return p;
  return pti_add_nz(p, q);
}

}  // namespace msm_amd
