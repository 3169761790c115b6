// The product's CPU MSM: where the reference calls the third-party halo2curves::msm::msm_best -- the `cpu` mode of
// gpu_profiler (src/bin/gpu_profiler.rs:157-159, BASELINE config 1), the CPU half of gpu_with_cpu
// (src/metal/msm.rs:412) and msm_best's size dispatch (msm.rs:440-444).  Product code: it ships in libmsm_amd.so, it is
// not the test oracle (nothing here includes or links oracle/), and the GPU entry points never route through it.
//
// Algorithm (written for this library, on the 4 x 64-bit host arithmetic of host_fq64.h):
//   phase 0  scalars -> canonical -> signed c-bit digits, [window][point] int16, threads over point ranges
//   phase 1  tasks (window, point group): affine buckets filled by BATCHED-AFFINE additions -- up to 512 pending
//            (bucket, point) pairs with distinct buckets share one field inversion (Montgomery's trick, four
//            interleaved product chains), ~8 field multiplications per addition instead of 11 for a Jacobian mixed
//            addition; a point whose bucket is already pending in the batch goes to that bucket's Jacobian side
//            accumulator instead (uniform scalars: a few per cent; all-equal scalars: all of them, still 11
//            multiplications each -- no pathological case).  (Tasks over bucket RANGES instead of point groups --
//            disjoint bucket sets, nothing to merge -- were measured too: 18-24 ms against 15.7-17.5 ms at 2^16
//            points on 16 threads, the strided reads of the points cost more than the merge.)
//            On hosts with AVX-512 IFMA (Zen 4 / 5, Ice Lake and later: CPUID-checked, MSM_AMD_HOST_NO_IFMA=1 turns it
//            off) the batched additions run eight per vector in radix 2^52 (host_ifma.cpp: 1.7 ns per field
//            multiplication against 10 ns), a collision waits for the next batch instead of paying a Jacobian
//            addition, and the batch is half the bucket count: 9.2 instead of 13.3 ms at 2^16 points on 16 threads.
//   phase 2  tasks (window, bucket segment): running sums over the segment across all groups, two interleaved chains,
//            sum_b (b + 1) B_b = sum_seg [ sum_{b in seg} (b - lo + 1) B_b  +  lo * sum_{b in seg} B_b ]
//   phase 3  Horner over the windows (one thread; 254 doublings)
// Tasks are handed out by an atomic counter; the team of threads lives for one call and meets at spin barriers.
// The window c minimises a cost model in field multiplications (window_for).
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/msm_amd.h"
#include "device_common.hip.h"
#include "launch.h"
#include "host_fq64.h"
#include "host_ifma.h"

namespace msm_amd {

namespace {

using h64::Fe;
using h64::Jac;

struct Aff {   // same 64 bytes as Affine: x, y Montgomery LE
  Fe x, y;
};
static_assert(sizeof(Aff) == sizeof(Affine), "layout");

inline bool fe_eq(const Fe& a, const Fe& b) {
  return ((a.v[0] ^ b.v[0]) | (a.v[1] ^ b.v[1]) | (a.v[2] ^ b.v[2]) | (a.v[3] ^ b.v[3])) == 0;
}
inline Fe fe_neg(const Fe& a) {
  Fe z;
  std::memset(&z, 0, sizeof z);
  return h64::is_zero(a) ? a : h64::sub(z, a);
}

// Jacobian + affine (finite), 8M + 3S, with the exceptional cases.
inline Jac jmadd(const Jac& p, const Aff& q) {
  if (h64::is_identity(p)) {
    Jac r;
    r.x = q.x;
    r.y = q.y;
    r.z = h64::one();
    return r;
  }
  const Fe Z1Z1 = h64::sqr(p.z);
  const Fe U2 = h64::mul(q.x, Z1Z1);
  const Fe S2 = h64::mul(q.y, h64::mul(p.z, Z1Z1));
  const Fe H = h64::sub(U2, p.x);
  const Fe R = h64::sub(S2, p.y);
  if (h64::is_zero(H)) {
    if (h64::is_zero(R)) return h64::jdouble(p);
    return h64::identity();
  }
  const Fe HH = h64::sqr(H);
  const Fe HHH = h64::mul(H, HH);
  const Fe V = h64::mul(p.x, HH);
  Jac r;
  r.x = h64::sub(h64::sub(h64::sqr(R), HHH), h64::dbl(V));
  r.y = h64::sub(h64::mul(R, h64::sub(V, r.x)), h64::mul(p.y, HHH));
  r.z = h64::mul(p.z, H);
  return r;
}

// k * p for a small k (the segment offset of phase 2), double-and-add from the top bit.
inline Jac jmul_small(const Jac& p, uint32_t k) {
  Jac acc = h64::identity();
  for (int bit = 31; bit >= 0; --bit) {
    acc = h64::jdouble(acc);
    if ((k >> bit) & 1u) acc = h64::jadd(acc, p);
  }
  return acc;
}

struct Barrier {   // the team meets here between phases; short waits, so spin (with yields) rather than sleep
  explicit Barrier(int n) : total(n) {}
  void wait() {
    const int gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == total) {
      arrived.store(0, std::memory_order_relaxed);
      generation.store(gen + 1, std::memory_order_release);
      return;
    }
    int spins = 0;
    while (generation.load(std::memory_order_acquire) == gen)
      if (++spins > 2000) std::this_thread::yield();
  }
  const int total;
  std::atomic<int> arrived{0}, generation{0};
};

// Buckets of one phase-1 task: affine value + occupancy + Jacobian side accumulator (z = 0: empty).
struct BucketSet {   // aff / stamp / full: nbuckets entries each inside the call's scratch block (run())
  Aff* aff = nullptr;
  uint32_t* stamp = nullptr;   // id of the batch in which the bucket has a pending addition
  uint8_t* full = nullptr;
  std::vector<Jac> side;       // allocated on the first collision only
  bool any_side = false;
};
constexpr size_t bucket_set_bytes(size_t nbuckets) {
  return ((nbuckets * (sizeof(Aff) + sizeof(uint32_t) + 1)) + 63) & ~(size_t)63;
}

constexpr int kBatch = 1024;
static_assert(kBatch <= ifma::kMaxBatch, "the vector path's scratch holds a whole batch");

struct Pending {
  uint32_t bucket[kBatch];
  alignas(64) Aff pt[kBatch];
  Fe den[kBatch], pre[kBatch];
  uint8_t kind[kBatch];   // 0 = generic addition, 1 = doubling, 2 = P + (-P)
  int count = 0;
};

// One shared inversion for the whole batch (Montgomery's trick), run as kLanes INTERLEAVED product chains (element k
// belongs to chain k mod kLanes): a single chain is a string of dependent multiplications, i.e. bound by the LATENCY of
// the field multiplication (~1.5x its reciprocal throughput); four independent chains keep the multiplier busy.
constexpr int kLanes = 4;

void flush(BucketSet& B, Pending& q) {
  if (q.count == 0) return;
  // denominators: x2 - x1, or 2 y for a doubling; one for P + (-P) so that the product stays invertible
  for (int k = 0; k < q.count; ++k) {
    const Aff& a = B.aff[q.bucket[k]];
    Fe d = h64::sub(q.pt[k].x, a.x);
    q.kind[k] = 0;
    if (h64::is_zero(d)) {
      if (fe_eq(q.pt[k].y, a.y)) {   // BN254 G1 has prime order: y != 0 for every finite point
        q.kind[k] = 1;
        d = h64::dbl(a.y);
      } else {
        q.kind[k] = 2;
        d = h64::one();
      }
    }
    q.den[k] = d;
    q.pre[k] = k >= kLanes ? h64::mul(q.pre[k - kLanes], d) : d;
  }
  // chain totals -> inverse of every chain total from ONE inversion
  Fe tot[kLanes], inv[kLanes];
  for (int j = 0; j < kLanes; ++j) {
    if (j >= q.count) {
      tot[j] = h64::one();
      continue;
    }
    int last = q.count - 1;
    last -= ((last - j) % kLanes + kLanes) % kLanes;   // largest k <= count - 1 with k mod kLanes == j
    tot[j] = q.pre[last];
  }
  {
    const Fe p01 = h64::mul(tot[0], tot[1]), p23 = h64::mul(tot[2], tot[3]);
    const Fe all = h64::inv(h64::mul(p01, p23));
    const Fe i01 = h64::mul(all, p23), i23 = h64::mul(all, p01);
    inv[0] = h64::mul(i01, tot[1]);
    inv[1] = h64::mul(i01, tot[0]);
    inv[2] = h64::mul(i23, tot[3]);
    inv[3] = h64::mul(i23, tot[2]);
  }
  for (int k = q.count - 1; k >= 0; --k) {
    Fe& run = inv[k % kLanes];   // 1 / (product of this chain's denominators up to and including k)
    const Fe dinv = k >= kLanes ? h64::mul(run, q.pre[k - kLanes]) : run;
    if (k >= kLanes) run = h64::mul(run, q.den[k]);
    Aff& a = B.aff[q.bucket[k]];
    if (q.kind[k] == 2) {
      B.full[q.bucket[k]] = 0;
      continue;
    }
    Fe num;
    if (q.kind[k] == 1) {
      const Fe xx = h64::sqr(a.x);
      num = h64::add(h64::dbl(xx), xx);
    } else {
      num = h64::sub(q.pt[k].y, a.y);
    }
    const Fe lambda = h64::mul(num, dinv);
    const Fe x3 = h64::sub(h64::sub(h64::sqr(lambda), a.x), q.pt[k].x);
    const Fe y3 = h64::sub(h64::mul(lambda, h64::sub(a.x, x3)), a.y);
    a.x = x3;
    a.y = y3;
  }
  q.count = 0;
}

// One phase-1 task: the points [lo, hi) of one window into B (all 2^(c-1) buckets of the window).
// `batch` = pending additions per shared inversion; 0 = no batched-affine additions at all (every point goes to its
// bucket's Jacobian accumulator): windows with few buckets or tasks with few points cannot fill a batch, and an
// inversion costs ~380 multiplications.
void fill_buckets(BucketSet& B, Pending& q, const int16_t* digits, const Aff* points, size_t lo, size_t hi,
                  uint32_t nbuckets, int batch) {
  std::memset(B.full, 0, nbuckets);
  std::fill(B.stamp, B.stamp + nbuckets, batch ? 0u : 1u);   // batch == 0: every bucket looks "pending" -> Jacobian path
  B.side.clear();
  B.any_side = false;
  uint32_t batch_id = 1;
  q.count = 0;
  for (size_t i = lo; i < hi; ++i) {
    const int32_t d = digits[i];
    if (d == 0) continue;   // (also every digit of an identity point: phase 0)
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
    Aff p = points[i];
    if (d < 0) p.y = fe_neg(p.y);
    if (B.stamp[b] == batch_id) {   // its bucket already has an addition pending in this batch
      if (!B.any_side) {
        B.side.resize(nbuckets);
        std::memset((void*)B.side.data(), 0, nbuckets * sizeof(Jac));   // z = 0: identity
        B.any_side = true;
      }
      B.side[b] = jmadd(B.side[b], p);
      continue;
    }
    if (!B.full[b]) {
      B.aff[b] = p;
      B.full[b] = 1;
      continue;
    }
    B.stamp[b] = batch_id;
    q.bucket[q.count] = b;
    q.pt[q.count] = p;
    if (++q.count == batch) {
      flush(B, q);
      ++batch_id;
    }
  }
  flush(B, q);
}

// ---- the same phase on AVX-512 IFMA (host_ifma.cpp): eight additions per vector -------------------------------------
// Bucket and point coordinates live in the vector code's Montgomery domain (Q = 2^260 instead of R = 2^256) as
// canonical 4 x u64 integers while a task fills its buckets; `points_q` is the whole point array converted once per
// MSM, the task's buckets are converted back before phase 2 reads them.  The shared inversion stays scalar: the eight
// chain totals of a batch are inverted together by Montgomery's trick on host_fq64.h.
struct IfmaState {
  ifma::Scratch scratch;
  uint32_t pt_idx[kBatch];              // the batch's points: index into points_q, bit 31 = negated
  std::vector<uint32_t> retry, again;   // points waiting for the next batch (fill_buckets_ifma)
  Fe to_q;   // 2^264 mod p as an R-domain operand: mul(R^2 / X, to_q) = Q^2 / X
};

Fe pow2_mod_p(unsigned k) {   // 2^k mod p as a plain integer below p
  Fe v;
  std::memset(&v, 0, sizeof v);
  v.v[0] = 1;
  for (unsigned i = 0; i < k; ++i) v = h64::add(v, v);
  return v;
}

thread_local uint64_t tl_cycles[4];   // MSM_AMD_HOST_TRACE: forward / inversion / backward / whole flush, in TSC ticks
// Returns the number of elements the vector code did NOT add because the point has its bucket's x (st.scratch.special[]).
int flush_ifma(BucketSet& B, Pending& q, IfmaState& st, const Aff* points_q) {
  if (q.count == 0) return 0;
  uint64_t totals[8][4], inv[8][4];
  const uint64_t c0 = __builtin_ia32_rdtsc();
  const int specials = ifma::forward((const uint64_t*)B.aff, q.bucket, (const uint64_t*)points_q, st.pt_idx, q.count, st.scratch, totals);
  // 1 / totals[k] in the Q domain: with X = totals[k] read as an R-domain element, inv(X) = R^2 / X and
  // mul(R^2 / X, 2^264) = Q^2 / X = (T Q)^-1 Q^2 = T^-1 Q.  One inversion for the eight of them.
  const uint64_t c1 = __builtin_ia32_rdtsc();
  Fe x[8], pre[8];
  for (int k = 0; k < 8; ++k) {
    std::memcpy(&x[k], totals[k], 32);
    pre[k] = k ? h64::mul(pre[k - 1], x[k]) : x[k];
  }
  static const bool fermat = std::getenv("MSM_AMD_HOST_INV_FERMAT") != nullptr;   // A/B aid
  Fe run = fermat ? h64::inv_fermat(pre[7]) : h64::inv(pre[7]);
  for (int k = 7; k >= 0; --k) {
    const Fe xi = k ? h64::mul(run, pre[k - 1]) : run;
    if (k) run = h64::mul(run, x[k]);
    const Fe r = h64::mul(xi, st.to_q);
    std::memcpy(inv[k], &r, 32);
  }
  const uint64_t c2 = __builtin_ia32_rdtsc();
  ifma::backward((uint64_t*)B.aff, q.bucket, q.count, st.scratch, inv);
  const uint64_t c3 = __builtin_ia32_rdtsc();
  tl_cycles[0] += c1 - c0;
  tl_cycles[1] += c2 - c1;
  tl_cycles[2] += c3 - c2;
  q.count = 0;
  return specials;
}

// fill_buckets with the batched additions on the vector unit.  `points` (R domain) still feeds the Jacobian side
// accumulators; `points_q` (Q domain) feeds the affine buckets.  A point whose x equals its bucket's current x (the
// same point again, or its negative) goes to the side accumulator too -- jmadd resolves doubling and cancellation --
// so every addition in a batch is a generic one.
void fill_buckets_ifma(BucketSet& B, Pending& q, IfmaState& st, const int16_t* digits, const Aff* points, const Aff* points_q,
                       size_t lo, size_t hi, uint32_t nbuckets, int batch) {
  std::memset((void*)B.aff, 0, nbuckets * sizeof(Aff));
  std::memset(B.full, 0, nbuckets);
  std::memset(B.stamp, 0, nbuckets * sizeof(uint32_t));
  B.side.clear();
  B.any_side = false;
  uint32_t batch_id = 1;
  q.count = 0;
  auto to_side = [&](uint32_t b, const Aff& src, bool negate) {
    if (!B.any_side) {
      B.side.resize(nbuckets);
      std::memset((void*)B.side.data(), 0, nbuckets * sizeof(Jac));   // z = 0: identity
      B.any_side = true;
    }
    Aff p = src;
    if (negate) p.y = fe_neg(p.y);
    B.side[b] = jmadd(B.side[b], p);
  };
  // A point whose bucket already has an addition pending in this batch WAITS for the next batch (st.retry) instead of
  // paying a Jacobian addition; when the waiting list is full, or a batch drew less than an eighth of its additions
  // from a non-empty list (skewed digits: everything wants the same few buckets), the rest goes to the Jacobian side
  // accumulators after all.
  std::vector<uint32_t>& retry = st.retry;
  std::vector<uint32_t>& again = st.again;
  retry.clear();
  again.clear();
  const size_t retry_cap = (size_t)batch;
  static const bool prefetch = std::getenv("MSM_AMD_HOST_NO_PREFETCH") == nullptr;   // A/B aid
  auto add_one = [&](uint32_t i, bool may_wait) {
    const int32_t d = digits[i];
    const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
    if (B.stamp[b] == batch_id) {
      if (may_wait && retry.size() < retry_cap) retry.push_back(i);
      else to_side(b, points[i], d < 0);
      return;
    }
    if (!B.full[b]) {
      const Aff& pq = points_q[i];
      B.aff[b] = pq;
      if (d < 0) B.aff[b].y = fe_neg(pq.y);
      B.full[b] = 1;
      return;
    }
    // (a point with its bucket's x -- doubling or cancellation, not a chord addition -- is found by the vector code
    // from its zero denominator and comes back through flush_ifma's return value)
    B.stamp[b] = batch_id;
    q.bucket[q.count] = b;
    st.pt_idx[q.count] = i | (d < 0 ? 0x80000000u : 0u);   // the vector code gathers and negates the point itself
    ++q.count;
    if (prefetch) {   // both records are first read by the vector forward pass, a batch later
      __builtin_prefetch(&B.aff[b], 0, 3);
      __builtin_prefetch(&points_q[i], 0, 3);
    }
  };
  size_t i = lo;
  for (;;) {
    const size_t waiting = again.size();
    int from_waiting = 0;
    while (q.count < batch) {
      if (!again.empty()) {
        const int before = q.count;
        const uint32_t idx = again.back();
        again.pop_back();
        add_one(idx, true);
        from_waiting += q.count - before;
      } else if (i < hi) {
        const uint32_t idx = (uint32_t)i++;
        if (digits[idx] == 0) continue;   // (also every digit of an identity point: phase 0)
        add_one(idx, true);
      } else {
        break;
      }
    }
    if (waiting >= 8 && (size_t)from_waiting * 8 < waiting) {   // the waiting list is not draining: skew
      for (uint32_t idx : again) add_one(idx, false);
      again.clear();
      for (uint32_t idx : retry) {
        const int32_t d = digits[idx];
        to_side((uint32_t)(d < 0 ? -d : d) - 1, points[idx], d < 0);
      }
      retry.clear();
    }
    if (q.count == 0 && again.empty() && retry.empty() && i >= hi) break;
    const int specials = flush_ifma(B, q, st, points_q);
    for (int e = 0; e < specials; ++e) {   // q.bucket / st.pt_idx still hold the batch
      const int at = st.scratch.special[e];
      to_side(q.bucket[at], points[st.pt_idx[at] & 0x7FFFFFFFu], (st.pt_idx[at] >> 31) != 0);
    }
    ++batch_id;
    for (uint32_t idx : retry) again.push_back(idx);   // `again` may still hold entries when the batch filled up first
    retry.clear();
  }
  // back to the R domain for phase 2 (empty slots are converted along; nobody reads them)
  ifma::convert((const uint64_t*)B.aff, (uint64_t*)B.aff, (size_t)nbuckets * 2, 1);
}

// One phase-2 task: sum_{b in [lo, hi)} (b + 1) B_b where B_b is the sum over the window's point groups, by running
// sums -- as TWO interleaved chains (upper and lower half of the segment): a running sum is a string of dependent
// field multiplications (bound by the multiplier's latency), two independent ones in one loop overlap in the
// out-of-order core.  With m = lo + (hi - lo) / 2:
//   sum = [ sum_{b >= m} (b - m + 1) B_b + m * sum_{b >= m} B_b ]  +  [ sum_{b < m} (b - lo + 1) B_b + lo * sum_{b < m} B_b ]
Jac reduce_segment(const BucketSet* groups, uint32_t n_groups, uint32_t lo, uint32_t hi) {
  auto step = [&](uint32_t b, Jac& running, Jac& acc) {
    for (uint32_t g = 0; g < n_groups; ++g) {
      const BucketSet& S = groups[g];
      if (S.full[b]) running = jmadd(running, S.aff[b]);
      if (S.any_side && !h64::is_identity(S.side[b])) running = h64::jadd(running, S.side[b]);
    }
    acc = h64::jadd(acc, running);
  };
  const uint32_t m = lo + (hi - lo) / 2;
  Jac run_hi = h64::identity(), acc_hi = h64::identity(), run_lo = h64::identity(), acc_lo = h64::identity();
  uint32_t bh = hi, bl = m;
  while (bl > lo) {   // the upper half [m, hi) has as many buckets as the lower [lo, m), or one more
    step(--bh, run_hi, acc_hi);
    step(--bl, run_lo, acc_lo);
  }
  while (bh > m) step(--bh, run_hi, acc_hi);
  Jac total = h64::jadd(acc_hi, acc_lo);
  if (m) total = h64::jadd(total, jmul_small(run_hi, m));
  if (lo) total = h64::jadd(total, jmul_small(run_lo, lo));
  return total;
}

// Cost model in field multiplications, as the MAKESPAN of the two task phases on `threads` threads (tasks are equal
// in size, so a phase takes ceil(tasks / threads) task times: 40 tasks on 16 threads cost as much as 48).
//   one addition into a bucket: Jacobian mixed addition 11; batched-affine 6 + its share of the inversion
//   (~384 / batch) + the collisions that fall back to the Jacobian accumulator (about batch / (2 buckets) of the
//   points, 11 each), with batch = a quarter of the buckets, at most kBatch;
//   phase 2 per window and bucket: `groups` mixed additions (11) and one full addition (16), weighted 1.3 because the
//   running sums are bound by latency rather than throughput.
// Measured on an EPYC 9575F, 16 threads, 2^16 points (`gpu_profiler 16 1 cpu 5`, profiles/r03_cpu_msm_sweep.txt):
// c = 11 with 2 groups (48 tasks) 13.4 ms, c = 12 / 2 groups (44 tasks) 13.9, c = 13 / 4 groups (80 tasks) 14.3,
// c = 13 / 2 groups (40 tasks) 15.6 -- the order this model gives.
// One batched-affine addition on the vector unit costs about as much as 1.8 scalar multiplications (6 vector
// multiplications + 6 exact subtractions per 8 additions, gathers and stores).
constexpr double kVecAdd = 1.8;

struct Choice {
  uint32_t c;
  uint32_t groups;   // point groups per window (phase-1 tasks = windows x groups)
  int batch;         // 0 = Jacobian accumulators only
};
int batch_for(uint32_t c, size_t points_per_task, bool vec = false) {
  // scalar path: a quarter of the buckets (a collision costs a Jacobian addition); vector path: half of them (a
  // collision only waits for the next batch)
  const size_t nbk = (size_t)1 << (c - 1);
  // (vector path: 1024 per inversion measured no better than 512 once the inversion is the binary GCD -- 2^18 points 32.0
  // vs 29.2 ms, 2^20 88.4 vs 87.5 -- and the batch's scratch is half the size: profiles/r03_cpu_msm_steps.txt)
  const int batch = (int)std::min<size_t>(kBatch / 2, vec ? nbk / 2 : nbk / 4);
  if (batch < 8 || points_per_task < (size_t)4 * batch) return 0;
  const double nb = (double)nbk;
  const double cost = vec ? kVecAdd + 384.0 / batch : 6.0 + 384.0 / batch + 11.0 * batch / (2.0 * nb);
  return cost < 11.0 ? batch : 0;
}
uint32_t segments_for(uint32_t c, uint32_t groups, int threads) {   // phase-2 tasks per window, >= 32 buckets each
  const uint32_t half = 1u << (c - 1);
  return threads <= 1 ? 1u : std::max(1u, std::min(groups, half / 32 ? half / 32 : 1u));
}
Choice window_for(size_t n, int threads, bool vec = false) {
  if (n < 32) return {3, 1, 0};   // the reference's policy for tiny instances (msm.rs:137-138)
  Choice best{4, 1, 0};
  double best_cost = 1e300;
  const uint32_t max_groups = threads <= 1 ? 1u : 8u;
  for (uint32_t c = 4; c <= 15; ++c) {
    const uint32_t W = 254 / c + 1;
    const double nb = (double)(1u << (c - 1));
    for (uint32_t groups = 1; groups <= max_groups; ++groups) {
      if (groups > 1 && n / groups < 256) break;
      const int batch = batch_for(c, n / groups, vec);
      const double add = !batch ? 11.0 : (vec ? kVecAdd + 384.0 / batch : 6.0 + 384.0 / batch + 11.0 * batch / (2.0 * nb));
      const uint32_t segs = segments_for(c, groups, threads);
      const double rounds1 = std::ceil((double)W * groups / threads), rounds2 = std::ceil((double)W * segs / threads);
      const double cost = rounds1 * ((double)n / groups) * add + rounds2 * (nb / segs) * 1.3 * (11.0 * groups + 16.0);
      if (cost < best_cost) {
        best_cost = cost;
        best = {c, groups, batch};
      }
    }
  }
  return best;
}

// The two arrays of n entries (digit matrix, points in the vector code's domain) live in ONE process-wide scratch
// block that outlives the call: a std::vector would zero 106 MB on the calling thread at 2^20 points -- fresh pages from
// mmap every call, 36 ms of page faults next to 87 ms of arithmetic -- where the workers of phase 0 overwrite every
// entry anyway.  A concurrent second call (the block is leased under try_lock) takes plain uninitialised memory of its
// own; blocks above kScratchKeep are returned to the system after the call.
constexpr size_t kScratchKeep = (size_t)1 << 30;
struct ScratchBlock {
  std::mutex m;
  void* p = nullptr;
  size_t cap = 0;
};
ScratchBlock g_scratch;
struct ScratchLease {
  void* p = nullptr;
  bool shared = false;
  explicit ScratchLease(size_t bytes) {
    if (g_scratch.m.try_lock()) {
      shared = true;
      if (g_scratch.cap < bytes) {
        std::free(g_scratch.p);
        g_scratch.p = std::aligned_alloc(64, (bytes + 63) & ~(size_t)63);
        g_scratch.cap = g_scratch.p ? bytes : 0;
      }
      p = g_scratch.p;
    } else {
      p = std::aligned_alloc(64, (bytes + 63) & ~(size_t)63);
    }
  }
  ~ScratchLease() {
    if (!shared) {
      std::free(p);
      return;
    }
    if (g_scratch.cap > kScratchKeep) {
      std::free(g_scratch.p);
      g_scratch.p = nullptr;
      g_scratch.cap = 0;
    }
    g_scratch.m.unlock();
  }
};

Jacobian run(const u256* scalars, int scalars_mont, const Aff* points, size_t n, int threads, bool& ok) {
  const int T = std::max(1, std::min<int>(threads, (int)std::min<size_t>((n + 63) / 64, 256)));
  const bool vec = ifma::available() && std::getenv("MSM_AMD_HOST_NO_IFMA") == nullptr;
  Choice choice = window_for(n, T, vec);
  if (const char* e = std::getenv("MSM_AMD_HOST_WINDOW")) {   // experiments (tools/dbg/cpu_sweep.sh)
    const int v = std::atoi(e);
    if (v >= 3 && v <= 15) choice.c = (uint32_t)v;
  }
  if (const char* e = std::getenv("MSM_AMD_HOST_GROUPS")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) choice.groups = (uint32_t)v;
  }
  if (std::getenv("MSM_AMD_HOST_WINDOW") || std::getenv("MSM_AMD_HOST_GROUPS"))
    choice.batch = batch_for(choice.c, n / choice.groups, vec);
  if (const char* e = std::getenv("MSM_AMD_HOST_BATCH")) {
    const int v = std::atoi(e);
    if (v >= 8 && v <= kBatch && choice.batch) choice.batch = std::min(choice.batch, v);
  }
  const bool use_vec = vec && choice.batch != 0;
  const uint32_t c = choice.c, groups = choice.groups;
  const uint32_t W = 254 / c + 1;
  const uint32_t half = 1u << (c - 1);   // buckets per window: slot b <-> digit magnitude b + 1
  const uint32_t segs = segments_for(c, groups, T);   // phase-2 tasks per window
  const uint32_t seg_len = (half + segs - 1) / segs;
  const size_t digit_bytes = ((size_t)W * n * sizeof(int16_t) + 63) & ~(size_t)63;
  const size_t pointq_bytes = use_vec ? n * sizeof(Aff) : 0, set_bytes = bucket_set_bytes(half);
  ScratchLease lease(digit_bytes + pointq_bytes + (size_t)W * groups * set_bytes);
  if (!lease.p) {   // out of host memory: reported, not thrown across the C ABI
    ok = false;
    return jac_identity();
  }
  int16_t* const digits = (int16_t*)lease.p;                       // [W][n], written by phase 0
  Aff* const points_q = (Aff*)((uint8_t*)lease.p + digit_bytes);   // the points in the vector code's Montgomery domain
  const Fe to_q = pow2_mod_p(264);
  std::vector<BucketSet> sets((size_t)W * groups);
  for (size_t t = 0; t < sets.size(); ++t) {   // (first written by the worker that takes task t)
    uint8_t* at = (uint8_t*)lease.p + digit_bytes + pointq_bytes + t * set_bytes;
    sets[t].aff = (Aff*)at;
    sets[t].stamp = (uint32_t*)(at + (size_t)half * sizeof(Aff));
    sets[t].full = at + (size_t)half * (sizeof(Aff) + sizeof(uint32_t));
  }
  std::vector<Jac> part((size_t)W * segs);
  std::atomic<size_t> next0{0}, next1{0}, next2{0};
  Barrier barrier(T);
  const size_t chunk0 = 1024;
  const bool trace = std::getenv("MSM_AMD_HOST_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what) {
    if (trace)
      std::fprintf(stderr, "host_msm: %-8s %8.3f ms (n=%zu c=%u W=%u groups=%u segs=%u batch=%d T=%d ifma=%d)\n", what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), n, c, W,
                   groups, segs, choice.batch, T, (int)use_vec);
  };
  auto worker = [&](int tid) {
    // ---- phase 0: digits
    for (;;) {
      const size_t lo = next0.fetch_add(chunk0, std::memory_order_relaxed);
      if (lo >= n) break;
      const size_t hi = std::min(n, lo + chunk0);
      for (size_t i = lo; i < hi; ++i) {
        u256 k;
        if (scalars_mont) {
          k = Fr::from_mont(scalars[i]);
        } else {   // raw canonical integers may exceed r: reduce (2^256 / r < 6), as digits_kernel does
          k = scalars[i];
          for (int t = 0; t < 5; ++t) k = Fr::reduce_once(k);
        }
        if (h64::is_zero(points[i].x) && h64::is_zero(points[i].y)) k = u256_zero();   // affine identity (0, 0): no digits
        uint32_t carry = 0;
        for (uint32_t w = 0; w < W; ++w) {
          const uint32_t start = w * c;
          const uint32_t v = (start < 256 ? u256_extract_bits(k, start, c) : 0u) + carry;
          carry = 0;
          int32_t d = (int32_t)v;
          if (v > half) {
            d = (int32_t)v - (int32_t)(1u << c);
            carry = 1;
          }
          digits[(size_t)w * n + i] = (int16_t)d;
        }
      }
      if (use_vec) ifma::convert((const uint64_t*)(points + lo), (uint64_t*)(points_q + lo), (hi - lo) * 2, 0);
    }
    barrier.wait();
    if (tid == 0) stamp("digits");
    // ---- phase 1: tasks (window, point group)
    tl_cycles[0] = tl_cycles[1] = tl_cycles[2] = 0;
    const uint64_t phase1_begin = __builtin_ia32_rdtsc();
    {
      Pending* q = new Pending();
      IfmaState* vs = use_vec ? new IfmaState() : nullptr;
      if (vs) vs->to_q = to_q;
      for (;;) {
        const size_t t = next1.fetch_add(1, std::memory_order_relaxed);
        if (t >= (size_t)W * groups) break;
        const uint32_t w = (uint32_t)(t / groups), g = (uint32_t)(t % groups);
        const size_t p_lo = n * g / groups, p_hi = n * (g + 1) / groups;
        if (use_vec)
          fill_buckets_ifma(sets[t], *q, *vs, &digits[(size_t)w * n], points, points_q, p_lo, p_hi, half, choice.batch);
        else
          fill_buckets(sets[t], *q, &digits[(size_t)w * n], points, p_lo, p_hi, half, choice.batch);
      }
      delete vs;
      delete q;
    }
    const uint64_t phase1_ticks = __builtin_ia32_rdtsc() - phase1_begin;
    barrier.wait();
    if (tid == 0) stamp("buckets");
    if (tid == 0 && trace && use_vec)
      std::fprintf(stderr, "host_msm: thread 0 of phase 1: %.1f %% vector forward, %.1f %% inversion, %.1f %% vector backward, "
                   "%.1f %% scheduling and the rest\n", 100.0 * tl_cycles[0] / phase1_ticks, 100.0 * tl_cycles[1] / phase1_ticks,
                   100.0 * tl_cycles[2] / phase1_ticks,
                   100.0 * (phase1_ticks - tl_cycles[0] - tl_cycles[1] - tl_cycles[2]) / phase1_ticks);
    // ---- phase 2: tasks (window, bucket segment)
    for (;;) {
      const size_t t = next2.fetch_add(1, std::memory_order_relaxed);
      if (t >= (size_t)W * segs) break;
      const uint32_t w = (uint32_t)(t / segs), s = (uint32_t)(t % segs);
      const uint32_t lo = std::min(half, s * seg_len), hi = std::min(half, lo + seg_len);
      part[t] = lo < hi ? reduce_segment(&sets[(size_t)w * groups], groups, lo, hi) : h64::identity();
    }
  };
  std::vector<std::thread> team;
  for (int t = 1; t < T; ++t) team.emplace_back(worker, t);
  worker(0);
  for (std::thread& th : team) th.join();
  stamp("segments");
  // ---- phase 3: Horner over the windows (final_accumulation.rs:19-39)
  Jac total = h64::identity();
  for (int w = (int)W - 1; w >= 0; --w) {
    for (uint32_t i = 0; i < c; ++i) total = h64::jdouble(total);
    for (uint32_t s = 0; s < segs; ++s) total = h64::jadd(total, part[(size_t)w * segs + s]);
  }
  stamp("horner");
  return h64::store(total);
}

}  // namespace

// sum_i k_i * P_i on `threads` host threads.  scalars: 32-byte LE (Montgomery if scalars_mont), points: 64-byte
// affine Montgomery LE with (0,0) = identity.  The result is NOT normalised (callers add it to a GPU partial first).
Jacobian host_msm(const u256* scalars, int scalars_mont, const Affine* points, size_t n, int threads, bool* ok) {
  bool fine = true;
  const Jacobian r = n == 0 ? jac_identity() : run(scalars, scalars_mont, (const Aff*)points, n, threads, fine);
  if (ok) *ok = fine;
  return r;
}

}  // namespace msm_amd

extern "C" {

// The CPU MSM as an entry point of its own: no ctx, no GPU (`gpu_profiler <log> <inst> cpu`).  h2c affine points;
// scalars MONT_LE or CANON_LE.  threads <= 0: every CPU the process may run on.
int msm_amd_host_msm(int scalar_layout, int point_layout, const void* scalars, const void* points, size_t n, int threads,
                     void* out96) {
  using namespace msm_amd;
  if (!scalars || !points || !out96 || n == 0) return MSM_AMD_INPUT_ERROR;
  if (point_layout != MSM_AMD_POINT_H2C_AFFINE) return MSM_AMD_INPUT_ERROR;
  if (scalar_layout != MSM_AMD_SCALAR_MONT_LE && scalar_layout != MSM_AMD_SCALAR_CANON_LE) return MSM_AMD_INPUT_ERROR;
  if (threads <= 0) threads = msm_amd_host_threads();
  bool ok = true;
  const Jacobian r = host_msm((const u256*)scalars, scalar_layout == MSM_AMD_SCALAR_MONT_LE, (const Affine*)points, n, threads, &ok);
  if (!ok) return MSM_AMD_PIPELINE_ERROR;   // host allocation failed
  const h64::Jac nrm = h64::normalise(h64::load(r));
  std::memcpy(out96, &nrm, 96);
  return MSM_AMD_OK;
}

// Unit-test hook of the AVX-512 IFMA arithmetic (host_ifma.cpp).  Returns MSM_AMD_FUNCTION_ERROR on a host without
// IFMA (the CPU MSM then takes its scalar path and there is nothing to test).  op 0: a b / 2^260 mod p, 1: a - b mod p,
// 2: -a mod p (a != 0), 3: R-domain -> Q-domain (a 2^4), 4: back; operands canonical 4 x u64 little-endian.
int msm_amd_test_op_ifma(int op, const void* a, const void* b, void* out, size_t count) {
  using namespace msm_amd;
  if (!a || !b || !out || count == 0 || op < 0 || op > 4) return MSM_AMD_INPUT_ERROR;
  if (!ifma::available()) return MSM_AMD_FUNCTION_ERROR;
  if (op <= 2) ifma::test_op(op, (const uint64_t*)a, (const uint64_t*)b, (uint64_t*)out, count);
  else ifma::convert((const uint64_t*)a, (uint64_t*)out, count, op == 3 ? 0 : 1);
  return MSM_AMD_OK;
}

// The deterministic synthetic instance of msm_amd_generate_instance, generated on the host (same generator code,
// identical bytes): `gpu_profiler ... cpu` needs no GPU at all, like the reference's `cpu` mode.
int msm_amd_generate_instance_host(uint64_t seed, size_t n, int scalars_mont, void* points, void* scalars, int threads) {
  using namespace msm_amd;
  if (!points || !scalars || n == 0) return MSM_AMD_INPUT_ERROR;
  if (threads <= 0) threads = msm_amd_host_threads();
  threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, (n + 255) / 256));
  Affine* P = (Affine*)points;
  u256* K = (u256*)scalars;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      const size_t lo = next.fetch_add(256, std::memory_order_relaxed);
      if (lo >= n) break;
      for (size_t t = lo; t < std::min(n, lo + 256); ++t) {
        Affine pt;
        pt.x = u256_zero();
        pt.y = u256_zero();
        for (uint32_t attempt = 0; attempt < 64; ++attempt)
          if (gen_point_attempt(seed, t, attempt, pt)) break;
        P[t] = pt;
        u256 k = gen_scalar_canonical(seed, t);
        if (scalars_mont) k = Fr::to_mont(k);
        K[t] = k;
      }
    }
  };
  std::vector<std::thread> team;
  for (int t = 1; t < threads; ++t) team.emplace_back(worker);
  worker();
  for (std::thread& th : team) th.join();
  return MSM_AMD_OK;
}

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU box
// shows 256 logical CPUs and grants 16: one thread per VISIBLE CPU runs 16 threads' worth of quota on 256 threads).
int msm_amd_host_threads(void) {
  static const int cached = [] {
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
    long long quota = -1, period = 100000;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
      char q[64] = {0};
      if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
      std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
      std::fclose(g);
      if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(h, "%lld", &period) != 1) period = 100000;
        std::fclose(h);
      }
    }
    if (quota > 0 && period > 0) cpus = std::min<int>(cpus, (int)std::max<long long>(1, (quota + period - 1) / period));
    return std::max(1, cpus);
  }();
  return cached;
}

}  // extern "C"
