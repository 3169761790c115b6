// Batched-affine bucket accumulation, GLOBAL variant: what would one round of a pair-tree accumulation cost on gfx950
// when the shared inversion is amortised over the whole round instead of one wave (batched_affine_bench.hip measured
// the lane-local variant and found the per-wave inversion and the LDS product tree too expensive)?
//
// One round adds N independent pairs of affine points (x1,y1) + (x2,y2) -> (x3,y3):
//   forward   lane t walks K1 pairs: d = x2 - x1, running product, prefix products to HBM (36 B each), lane product out
//   up        the same one level higher (K2 lane products per lane), twice  -> N / (K1 K2 K3) values
//   invert    Fermat inversion of those (64 per wave-operation, every lane busy)
//   down      back-substitution through the two upper levels (2 multiplications per element)
//   backward  lane t walks its K1 pairs in reverse: 1/d (2 multiplications), lambda, x3, y3 (2M + 1S), result packed into
//             64 bytes (value < 2 p, exact limbs) and stored
// Per addition: 1 + 2 + 3 = 6 multiplications (5M + 1S) + 3/K1 + ... against 8M + 2S of the mixed XYZZ addition, plus
// 36 B written + 36 B read of prefix products, the pair read twice and 64 B of result.
// Operands are random field elements (the addition law never checks the curve equation); the result is verified on the
// device through  (y3 + y1)(x2 - x1) = (y2 - y1)(x1 - x3)  and  (x3 + x1 + x2)(x2 - x1)^2 = (y2 - y1)^2.
// Two access patterns: "gather" (pairs of random records of a 2^20-record table: round 0 of an MSM reads the bases through
// the sorted index) and "stream" (pairs (2t, 2t+1) of a 2N-record array: later rounds).
// Output: ms per kernel, picoseconds of GPU time per addition; the accumulate kernel of the 2^20 pipeline spends
// 1.04 ms / 15.7 M = 66 ps per addition when it runs alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../metal-msm-gpu-acceleration_amd/csrc/device_common.hip.h"
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_ec29.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// p - 2, little-endian 32-bit words
__device__ __constant__ uint32_t kPm2[8] = {0xD87CFD45u, 0x3C208C16u, 0x6871CA8Du, 0x97816A91u,
                                            0x8181585Du, 0xB85045B6u, 0xE131A029u, 0x30644E72u};

__device__ __forceinline__ fe29 inv_fermat(const fe29& a) {
  fe29 tab[16];
  tab[0] = Fq29::one();
  tab[1] = a;
#pragma unroll 1
  for (int i = 2; i < 16; ++i) tab[i] = Fq29::mul(tab[i - 1], a);
  fe29 acc = Fq29::one();
#pragma unroll 1
  for (int nib = 63; nib >= 0; --nib) {
#pragma unroll 1
    for (int s = 0; s < 4; ++s) acc = Fq29::sqr(acc);
    const uint32_t d = (kPm2[nib >> 3] >> ((nib & 7) * 4)) & 15u;
    if (d) acc = Fq29::mul(acc, tab[d]);   // wave-uniform
  }
  return acc;
}

// fe29 arrays in HBM: chunks of 64 elements, limb-major inside a chunk, so that the 64 lanes of a wave store / load one
// limb with one fully coalesced 256-byte access.
__device__ __forceinline__ void store_fe(uint32_t* base, size_t e, const fe29& v) {
  uint32_t* p = base + (e >> 6) * (9 * 64) + (e & 63);
#pragma unroll
  for (int l = 0; l < 9; ++l) p[l * 64] = v.l[l];
}
__device__ __forceinline__ fe29 load_fe(const uint32_t* base, size_t e) {
  const uint32_t* p = base + (e >> 6) * (9 * 64) + (e & 63);
  fe29 v;
#pragma unroll
  for (int l = 0; l < 9; ++l) v.l[l] = p[l * 64];
  return v;
}

// value < 8 p, arbitrary u32 limbs -> exact limbs (< 2^29), value < 2 p (< 2^256: packs into 32 bytes).
// Quotient estimate from the top limb: q = floor(l8 / (p8 + 1)) <= floor(v / p) <= q + 1.
__device__ __forceinline__ fe29 semi_canonical(const fe29& a) {
  fe29 r;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t s = a.l[i] + carry;   // limbs < 2^32 - 2^4: no overflow for norm()ed or sub<> values below 2^31.6
    r.l[i] = s & Fq29::MASK;
    carry = s >> 29;
  }
  r.l[8] = a.l[8] + carry;
  constexpr uint32_t P8 = Fq29::p(8) + 1;
  uint32_t q = 0;
#pragma unroll
  for (uint32_t j = 1; j <= 7; ++j) q += (r.l[8] >= j * P8) ? 1u : 0u;
  int64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int64_t s = (int64_t)r.l[i] - (int64_t)((uint64_t)q * Fq29::p(i)) + borrow;
    r.l[i] = (i < 8) ? ((uint32_t)s & Fq29::MASK) : (uint32_t)s;
    borrow = s >> 29;
  }
  return r;
}

template <int K>
__global__ void __launch_bounds__(64)
ba_forward(const AffPacked* __restrict__ pts, const uint32_t* __restrict__ ia, const uint32_t* __restrict__ ib,
           uint32_t* __restrict__ prefix, uint32_t* __restrict__ prod) {
  const size_t wave = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  fe29 acc;
#pragma unroll 2
  for (int k = 0; k < K; ++k) {
    const size_t t = (wave * K + k) * 64 + lane;
    const fe29 x1 = Fq29::unpack256(load_u256(&pts[ia[t]].x));
    const fe29 x2 = Fq29::unpack256(load_u256(&pts[ib[t]].x));
    const fe29 d = Fq29::norm(Fq29::sub<K4E30>(x2, x1));
    acc = k ? Fq29::mul(acc, d) : d;
    if (k < K - 1) store_fe(prefix, t, acc);
  }
  store_fe(prod, wave * 64 + lane, acc);
}

template <int K>
__global__ void __launch_bounds__(64)
ba_up(const uint32_t* __restrict__ in, uint32_t* __restrict__ prefix, uint32_t* __restrict__ out) {
  const size_t wave = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  fe29 acc;
#pragma unroll 2
  for (int k = 0; k < K; ++k) {
    const size_t e = (wave * K + k) * 64 + lane;
    const fe29 d = load_fe(in, e);
    acc = k ? Fq29::mul(acc, d) : d;
    if (k < K - 1) store_fe(prefix, e, acc);
  }
  store_fe(out, wave * 64 + lane, acc);
}

__global__ void __launch_bounds__(64) ba_invert(uint32_t* __restrict__ x) {
  const size_t e = (size_t)blockIdx.x * 64 + threadIdx.x;
  store_fe(x, e, inv_fermat(load_fe(x, e)));
}

// inv_out[e'] = 1 / (product of the K inputs of lane e')  ->  inv_in[e] = 1 / in[e]
template <int K>
__global__ void __launch_bounds__(64)
ba_down(const uint32_t* __restrict__ inv_out, const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ in,
        uint32_t* __restrict__ inv_in) {
  const size_t wave = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  fe29 run = load_fe(inv_out, wave * 64 + lane);
#pragma unroll 2
  for (int k = K - 1; k >= 1; --k) {
    const size_t e = (wave * K + k) * 64 + lane;
    store_fe(inv_in, e, Fq29::mul(run, load_fe(prefix, e - 64)));
    run = Fq29::mul(run, load_fe(in, e));
  }
  store_fe(inv_in, wave * K * 64 + lane, run);
}

template <int K>
__global__ void __launch_bounds__(64)
ba_backward(const AffPacked* __restrict__ pts, const uint32_t* __restrict__ ia, const uint32_t* __restrict__ ib,
            const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ inv, AffPacked* __restrict__ out) {
  const size_t wave = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  fe29 run = load_fe(inv, wave * 64 + lane);
#pragma unroll 1
  for (int k = K - 1; k >= 0; --k) {
    const size_t t = (wave * K + k) * 64 + lane;
    const uint32_t a = ia[t], b = ib[t];
    const fe29 x1 = Fq29::unpack256(load_u256(&pts[a].x));
    const fe29 y1 = Fq29::unpack256(load_u256(&pts[a].y));
    const fe29 x2 = Fq29::unpack256(load_u256(&pts[b].x));
    const fe29 y2 = Fq29::unpack256(load_u256(&pts[b].y));
    const fe29 d = Fq29::norm(Fq29::sub<K4E30>(x2, x1));
    fe29 inv_d = run;
    if (k) {
      inv_d = Fq29::mul(run, load_fe(prefix, t - 64));
      run = Fq29::mul(run, d);
    }
    const fe29 dy = Fq29::norm(Fq29::sub<K4E30>(y2, y1));
    const fe29 lam = Fq29::mul(dy, inv_d);
    const fe29 x3 = semi_canonical(Fq29::sub<K4E30>(Fq29::sqr(lam), Fq29::add(x1, x2)));   // < 5.1 p -> < 2 p
    const fe29 tt = Fq29::norm(Fq29::sub<K4E30>(x1, x3));                                   // < 6 p
    const fe29 y3 = semi_canonical(Fq29::sub<K4E30>(Fq29::mul(lam, tt), y1));               // < 5.1 p -> < 2 p
    AffPacked r;
    r.x = Fq29::pack256(x3);
    r.y = Fq29::pack256(y3);
    store_u256(&out[t].x, r.x);
    store_u256(&out[t].y, r.y);
  }
}

__global__ void check_kernel(const AffPacked* __restrict__ pts, const uint32_t* __restrict__ ia,
                             const uint32_t* __restrict__ ib, const AffPacked* __restrict__ out, size_t n,
                             uint32_t* __restrict__ bad) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const fe29 x1 = Fq29::unpack256(load_u256(&pts[ia[t]].x)), y1 = Fq29::unpack256(load_u256(&pts[ia[t]].y));
  const fe29 x2 = Fq29::unpack256(load_u256(&pts[ib[t]].x)), y2 = Fq29::unpack256(load_u256(&pts[ib[t]].y));
  const fe29 x3 = Fq29::unpack256(load_u256(&out[t].x)), y3 = Fq29::unpack256(load_u256(&out[t].y));
  const fe29 d = Fq29::norm(Fq29::sub<K4E30>(x2, x1));
  const fe29 dy = Fq29::norm(Fq29::sub<K4E30>(y2, y1));
  const fe29 l1 = Fq29::mul(Fq29::norm(Fq29::add(y3, y1)), d);
  const fe29 r1 = Fq29::mul(dy, Fq29::norm(Fq29::sub<K4E30>(x1, x3)));
  const fe29 l2 = Fq29::mul(Fq29::norm(Fq29::add(x3, Fq29::add(x1, x2))), Fq29::sqr(d));
  const fe29 r2 = Fq29::sqr(dy);
  const bool ok = Fq29::is_zero_exact(Fq29::norm(Fq29::sub<K4E30>(l1, r1))) &&
                  Fq29::is_zero_exact(Fq29::norm(Fq29::sub<K4E30>(l2, r2)));
  if (!ok) atomicAdd(bad, 1u);
}

__global__ void fill_points(AffPacked* pts, size_t n, uint64_t seed) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint64_t s = seed + t * 0x9E3779B97F4A7C15ull;
  uint32_t* w = reinterpret_cast<uint32_t*>(&pts[t]);
  for (int i = 0; i < 16; ++i) {
    s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
    w[i] = (uint32_t)s;
  }
  w[7] &= 0x1FFFFFFFu;    // x, y < 2^253 < p
  w[15] &= 0x1FFFFFFFu;
}
__global__ void fill_index(uint32_t* ia, uint32_t* ib, size_t n, uint32_t mask, int gather, uint64_t seed) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (gather) {
    uint64_t s = seed + t * 0x9E3779B97F4A7C15ull;
    s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
    ia[t] = (uint32_t)s & mask;
    ib[t] = (uint32_t)(s >> 32) & mask;
    if (ia[t] == ib[t]) ib[t] ^= 1u;
  } else {
    ia[t] = (uint32_t)(2 * t);
    ib[t] = (uint32_t)(2 * t + 1);
  }
}

constexpr int K1 = 8, K2 = 16, K3 = 16;

int main(int argc, char** argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 22;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const size_t N = (size_t)1 << logn;
  const size_t M1 = N / K1, M2 = M1 / K2, M3 = M2 / K3;
  if (M3 < 64) { fprintf(stderr, "log2(N) must be at least 17\n"); return 1; }
  printf("# N = 2^%d pair additions per round, K1 = %d, K2 = %d, K3 = %d -> %zu inversions (%zu wave-operations)\n", logn,
         K1, K2, K3, M3, M3 / 64);
  for (int gather = 1; gather >= 0; --gather) {
    const size_t P = gather ? ((size_t)1 << 20) : 2 * N;
    AffPacked *pts, *out;
    uint32_t *ia, *ib, *pre1, *prod1, *pre2, *prod2, *pre3, *prod3, *inv2, *inv1, *bad;
    CHECK(hipMalloc(&pts, P * sizeof(AffPacked)));
    CHECK(hipMalloc(&out, N * sizeof(AffPacked)));
    CHECK(hipMalloc(&ia, N * 4));
    CHECK(hipMalloc(&ib, N * 4));
    CHECK(hipMalloc(&pre1, N * 36));
    CHECK(hipMalloc(&prod1, M1 * 36));
    CHECK(hipMalloc(&pre2, M1 * 36));
    CHECK(hipMalloc(&prod2, M2 * 36));
    CHECK(hipMalloc(&pre3, M2 * 36));
    CHECK(hipMalloc(&prod3, M3 * 36));
    CHECK(hipMalloc(&inv2, M2 * 36));
    CHECK(hipMalloc(&inv1, M1 * 36));
    CHECK(hipMalloc(&bad, 4));
    CHECK(hipMemset(bad, 0, 4));
    fill_points<<<(unsigned)((P + 255) / 256), 256>>>(pts, P, 0x1234);
    fill_index<<<(unsigned)((N + 255) / 256), 256>>>(ia, ib, N, (uint32_t)(P - 1), gather, 0x77);
    CHECK(hipDeviceSynchronize());
    hipEvent_t ev[8];
    for (auto& evt : ev) CHECK(hipEventCreate(&evt));
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < reps + 1; ++r) {
      CHECK(hipEventRecord(ev[0]));
      ba_forward<K1><<<(unsigned)(M1 / 64), 64>>>(pts, ia, ib, pre1, prod1);
      CHECK(hipEventRecord(ev[1]));
      ba_up<K2><<<(unsigned)(M2 / 64), 64>>>(prod1, pre2, prod2);
      CHECK(hipEventRecord(ev[2]));
      ba_up<K3><<<(unsigned)(M3 / 64), 64>>>(prod2, pre3, prod3);
      CHECK(hipEventRecord(ev[3]));
      ba_invert<<<(unsigned)(M3 / 64), 64>>>(prod3);
      CHECK(hipEventRecord(ev[4]));
      ba_down<K3><<<(unsigned)(M3 / 64), 64>>>(prod3, pre3, prod2, inv2);
      CHECK(hipEventRecord(ev[5]));
      ba_down<K2><<<(unsigned)(M2 / 64), 64>>>(inv2, pre2, prod1, inv1);
      CHECK(hipEventRecord(ev[6]));
      ba_backward<K1><<<(unsigned)(M1 / 64), 64>>>(pts, ia, ib, pre1, inv1, out);
      CHECK(hipEventRecord(ev[7]));
      CHECK(hipDeviceSynchronize());
      if (r == 0) continue;   // warm-up
      for (int i = 0; i < 7; ++i) {
        float ms;
        CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        sum[i] += ms;
      }
    }
    check_kernel<<<(unsigned)((N + 255) / 256), 256>>>(pts, ia, ib, out, N, bad);
    uint32_t hbad = 0;
    CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    const char* names[7] = {"forward (K1)", "up (K2)", "up (K3)", "invert", "down (K3)", "down (K2)", "backward (K1)"};
    printf("%s pairs: %u of %zu results fail the addition-law check\n", gather ? "gathered" : "streamed", hbad, N);
    double total = 0, busy = 0;
    for (int i = 0; i < 7; ++i) {
      const double ms = sum[i] / reps;
      printf("  %-14s %8.3f ms\n", names[i], ms);
      total += ms;
      if (i != 3) busy += ms;
    }
    // the inversion kernel occupies M3/64 of 1024 SIMD slots for its whole (latency-bound) duration
    const double inv_share = (sum[3] / reps) * (double)(M3 / 64) / 1024.0;
    printf("  total          %8.3f ms = %6.1f ps per addition; counting the inversion by the SIMDs it occupies: %6.1f ps\n",
           total, total * 1e9 / (double)N, (busy + inv_share) * 1e9 / (double)N);
    for (auto& evt : ev) CHECK(hipEventDestroy(evt));
    for (void* p : {(void*)pts, (void*)out, (void*)ia, (void*)ib, (void*)pre1, (void*)prod1, (void*)pre2, (void*)prod2,
                    (void*)pre3, (void*)prod3, (void*)inv2, (void*)inv1, (void*)bad})
      CHECK(hipFree(p));
  }
  return 0;
}
