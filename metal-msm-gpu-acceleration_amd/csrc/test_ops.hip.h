// Single-operation test bodies shared by the device test kernel (k_stage.hip) and the host-side test entry
// point msm_amd_test_op_host: the same source runs on a GPU lane and on a CPU core.  Mirrors the reference's
// one-thread test kernels (src/metal/shader/tests/*.h.metal, curves/bn254.h.metal) and adds the ops of the
// 29-bit-limb internal representation used by the hot kernels.
#pragma once
#include "bn254_ec29.hip.h"

namespace msm_amd {

// a, b, out: little-endian u256 arrays; element t uses 1 (integer/field ops) or 3 (points) consecutive u256.
MSM_HD void run_test_op(int op, const u256* a, const u256* b, u256* out, uint32_t t) {
  if (op <= 9 || (op >= 14 && op <= 21) || op >= 32) {
    const u256 x = a[t];
    const u256 y = b[t];
    u256 r = u256_zero();
    switch (op) {
      case 0: u256_add(r, x, y); break;
      case 1: u256_sub(r, x, y); break;
      case 2: r = u256_mul_u32(x, y.v[0]); break;
      case 3: r = u256_shl(x, y.v[0] & 255u); break;
      case 4: r = u256_shr(x, y.v[0] & 255u); break;
      case 5: r = Fq::add(x, y); break;
      case 6: r = Fq::sub(x, y); break;
      case 7: r = Fq::mul(x, y); break;
      case 8: r = Fq::neg(x); break;
      case 9: r = Fq::pow_u32(x, y.v[0]); break;
      default: {
        const fe29 xi = Fq29::from_ext(x);
        const fe29 yi = Fq29::from_ext(y);
        const fe29 y3 = Fq29::add(yi, Fq29::add(yi, yi));
        fe29 ri = Fq29::zero();
        switch (op) {
          case 14: ri = Fq29::mul(xi, yi); break;
          case 15: ri = Fq29::sqr(xi); break;
          case 16: ri = Fq29::norm(Fq29::sub<K4E30>(xi, yi)); break;
          case 17: ri = Fq29::norm(Fq29::sub<K8E30>(xi, yi)); break;
          case 18: ri = Fq29::norm(Fq29::sub<K8E31>(xi, y3)); break;
          case 19: ri = Fq29::norm(Fq29::sub<K16E30>(xi, yi)); break;
          case 20: ri = Fq29::norm(Fq29::sub<K16E31>(xi, y3)); break;
          case 21: ri = Fq29::unpack256(Fq29::pack_canonical(xi)); break;   // incl. the 32-byte storage form
#if defined(MSM_AMD_EXPERIMENTS)
          // build options measured and not shipped (HISTORY.md; -DMSM_AMD_EXPERIMENTS builds only): same values as mul / mul2 / sqr
          case 32: ri = Fq29::mul_karatsuba(xi, yi); break;
          case 33: {   // lockstep product-scanning chains: (x*y, y*y) side by side, their sum checked
            fe29 u, v;
            Fq29::mul_pair(xi, yi, yi, yi, u, v);
            ri = Fq29::norm(Fq29::add(u, v));
            break;
          }
          case 34: {   // (x*y + y*x, x*x) and the squaring pair
            fe29 u, v, w, z;
            Fq29::mul2_mul_pair(xi, yi, yi, xi, xi, xi, u, v);
            Fq29::sqr_pair(xi, yi, w, z);
            ri = Fq29::norm(Fq29::add(Fq29::add(u, v), Fq29::add(w, z)));
            break;
          }
          case 35: {   // three products side by side
            fe29 u, v, w;
            Fq29::mul_triple(xi, yi, xi, xi, yi, yi, u, v, w);
            ri = Fq29::norm(Fq29::add(u, Fq29::add(v, w)));
            break;
          }
          case 36: ri = Fq29::mul2_karatsuba_second(xi, yi, yi, xi); break;   // 2 x y
#endif
        }
        r = Fq29::to_ext(ri);
      }
    }
    out[t] = r;
    return;
  }
  const Jacobian* pa = reinterpret_cast<const Jacobian*>(a);
  const Jacobian* pb = reinterpret_cast<const Jacobian*>(b);
  Jacobian* po = reinterpret_cast<Jacobian*>(out);
  const Jacobian p = pa[t];
  Jacobian r = jac_identity();
  if (op == 10) {
    r = jac_add(p, pb[t]);
  } else if (op == 11) {
    r = jac_scalar_mul(p, b[t]);
  } else if (op == 12) {
    const Jacobian q = pb[t];
    if (jac_is_identity(q)) {
      r = p;
    } else {
      Affine qa;
      qa.x = q.x;
      qa.y = q.y;
      r = jac_madd(p, qa);
    }
  } else if (op == 13) {
    r = jac_double(p);
  } else if (op == 22) {   // internal representation: Jacobian + affine (b given with z = one or z = 0)
    const Jacobian q = pb[t];
    const PtI pi = pti_from_ext(p);
    if (jac_is_identity(q)) {
      r = pti_to_ext(pi);
    } else {
      Affine qa;
      qa.x = q.x;
      qa.y = q.y;
      const AffI qi = affi_from_ext(qa);
      r = pti_to_ext(pti_is_identity(pi) ? pti_from_affi(qi) : pti_madd(pi, qi));
    }
  } else if (op == 23) {   // internal representation: Jacobian + Jacobian
    r = pti_to_ext(pti_add(pti_from_ext(p), pti_from_ext(pb[t])));
  } else if (op == 24) {   // 64 chained mixed additions without leaving the lazy internal form: p + 64 q
    const Jacobian q = pb[t];
    PtI acc = pti_from_ext(p);
    if (!jac_is_identity(q)) {
      Affine qa;
      qa.x = q.x;
      qa.y = q.y;
      const AffI qi = affi_from_ext(qa);
      for (int i = 0; i < 64; ++i) acc = pti_is_identity(acc) ? pti_from_affi(qi) : pti_madd(acc, qi);
    }
    r = pti_to_ext(acc);
  } else if (op == 25) {   // 16 chained full additions: p + 16 q with q Jacobian
    const PtI qi = pti_from_ext(pb[t]);
    PtI acc = pti_from_ext(p);
    for (int i = 0; i < 16; ++i) acc = pti_add(acc, qi);
    r = pti_to_ext(acc);
  } else if (op == 26) {   // the start of every work item: a (affine, z = one) + b (affine) by pti_mmadd, with both
                           // operands lazily negated first (the bound-critical case), then 3 more mixed additions
                           // of -b: the result is -(a + b) - 3 b = -a - 4 b
    const Jacobian q = pb[t];
    if (jac_is_identity(p) || jac_is_identity(q)) {
      r = jac_is_identity(p) ? q : p;   // callers pass finite points; identities are returned unchanged
    } else {
      Affine pa, qa;
      pa.x = p.x;
      pa.y = p.y;
      qa.x = q.x;
      qa.y = q.y;
      AffI pi = affi_from_ext(pa), qi = affi_from_ext(qa);
      pi.y = Fq29::neg(pi.y);         // the accumulator's y: normalised when the point was stored (pti_from_affi)
      qi.y = Fq29::neg_wide(qi.y);    // the incoming base's y: un-normalised negation, as in accumulate_kernel
      PtI acc = pti_mmadd(pi.x, pi.y, qi);
      for (int i = 0; i < 3; ++i) acc = pti_is_identity(acc) ? pti_from_affi(qi) : pti_madd(acc, qi);
      r = pti_to_ext(acc);
    }
  }
  po[t] = r;
}

#if defined(MSM_AMD_EXPERIMENTS)
constexpr int kTestOpMax = 36;   // 27..31 exist on the host only (msm_host.hip), 32..36: the experimental multiplication forms
#else
constexpr int kTestOpMax = 31;
#endif
MSM_HD bool test_op_is_point(int op) { return (op >= 10 && op <= 13) || (op >= 22 && op <= 26); }

}  // namespace msm_amd
