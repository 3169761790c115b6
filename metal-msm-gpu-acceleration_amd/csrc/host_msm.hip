// Host-CPU bucket-method MSM used ONLY as the CPU half of gpu_with_cpu (src/metal/msm.rs:366-421), where the
// reference calls the third-party halo2curves::msm::msm_best on its share of the points (msm.rs:412).  This is
// product code (it ships in libmsm_amd.so); it is not the test oracle and the GPU path never routes through it.
// Multi-threaded over point slices: every worker runs a serial signed-digit bucket method on its slice
// (window c = ln(n) rounded up), the slice results are added.  Field arithmetic: bn254_fq.hip.h's host path
// (4 x 64-bit limbs, unsigned __int128).
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "launch.h"

namespace msm_amd {

namespace {

Affine affine_neg(const Affine& p) {
  Affine r;
  r.x = p.x;
  r.y = Fq::neg(p.y);
  return r;
}

void msm_slice(const u256* scalars, int scalars_mont, const Affine* points, size_t n, Jacobian* out) {
  Jacobian acc = jac_identity();
  if (n == 0) {
    *out = acc;
    return;
  }
  uint32_t c = n < 32 ? 3u : (uint32_t)std::ceil(std::log((double)n));
  c = std::min(16u, std::max(3u, c));
  const uint32_t W = 254 / c + 1;
  const uint32_t half = 1u << (c - 1);
  std::vector<u256> ks(n);
  for (size_t i = 0; i < n; ++i) {
    if (scalars_mont) {
      ks[i] = Fr::from_mont(scalars[i]);
    } else {   // raw canonical integers may exceed r: reduce (2^256 / r < 6), as digits_kernel does
      ks[i] = scalars[i];
      for (int k = 0; k < 5; ++k) ks[i] = Fr::reduce_once(ks[i]);
    }
  }
  // signed digits, window-major: digit[w][i]
  std::vector<int32_t> digits((size_t)W * n);
  for (size_t i = 0; i < n; ++i) {
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; ++w) {
      const uint32_t start = w * c;
      uint32_t v = (start < 256 ? u256_extract_bits(ks[i], start, c) : 0u) + carry;
      carry = 0;
      int32_t d = (int32_t)v;
      if (v > half) {
        d = (int32_t)v - (int32_t)(1u << c);
        carry = 1;
      }
      digits[(size_t)w * n + i] = d;
    }
  }
  std::vector<Jacobian> buckets(half);
  for (int w = (int)W - 1; w >= 0; --w) {
    for (uint32_t i = 0; i < c; ++i) acc = jac_double(acc);
    for (auto& b : buckets) b = jac_identity();
    const int32_t* dw = &digits[(size_t)w * n];
    for (size_t i = 0; i < n; ++i) {
      const int32_t d = dw[i];
      if (d == 0 || affine_is_identity(points[i])) continue;
      const uint32_t m = (uint32_t)(d < 0 ? -d : d);
      buckets[m - 1] = jac_madd(buckets[m - 1], d < 0 ? affine_neg(points[i]) : points[i]);
    }
    Jacobian run = jac_identity(), sum = jac_identity();
    for (int b = (int)half - 1; b >= 0; --b) {
      run = jac_add(run, buckets[b]);
      sum = jac_add(sum, run);
    }
    acc = jac_add(acc, sum);
  }
  *out = acc;
}

}  // namespace

// sum_i k_i * P_i on `threads` host threads.  scalars: 32-byte LE (Montgomery if scalars_mont), points: 64-byte
// affine Montgomery LE with (0,0) = identity.
Jacobian host_msm(const u256* scalars, int scalars_mont, const Affine* points, size_t n, int threads) {
  if (n == 0) return jac_identity();
  threads = std::max(1, std::min<int>(threads, (int)std::min<size_t>(n, 256)));
  std::vector<Jacobian> res(threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
    pool.emplace_back(msm_slice, scalars + lo, scalars_mont, points + lo, hi - lo, &res[t]);
  }
  Jacobian total = jac_identity();
  for (int t = 0; t < threads; ++t) {
    pool[t].join();
    total = jac_add(total, res[t]);
  }
  return total;
}

}  // namespace msm_amd
