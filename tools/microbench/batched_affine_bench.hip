// Is batched-affine bucket accumulation worth it on gfx950?  (measure, don't argue.)
//
// Measures, on the production field arithmetic (bn254_fq29.hip.h), with NO memory traffic at all (operands in
// registers / LDS -- the most favourable setting for the batched-affine side):
//   1. one mixed XYZZ addition (pti_madd, 8M + 2S)                      -- what the accumulate kernel does per point
//   2. one field inversion by Fermat (fixed 4-bit windows: 254 S + 78 M)
//   3. one field inversion by a constant-trip binary extended Euclid (2 x 261 shift/subtract steps on 9 limbs),
//      an upper bound for what a divstep ("safegcd") inversion costs without its 2x2-matrix batching
//   4. lane-local batched-affine additions, K pending additions per lane, ONE inversion shared by the 64 lanes of
//      the wave through an LDS product tree (6 levels up, 6 down) and -- variant 5 -- by all waves of a 256-thread
//      workgroup: forward prefix products, tree, inversion, back-substitution, 3 multiplications per addition.
// Output: cycles@2.4GHz per wave-operation (per SIMD throughput, like fq29_bench) and, for 4/5, per ADDITION, to be
// compared with line 1.  K = 8 is what fits the LDS at 2 waves/SIMD (7 prefix products x 36 B per lane x 512 lanes
// = 129 KB of 160 KB); K = 16 is shown for the trend (it only fits at 1 wave/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_ec29.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 200;

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

// constant-trip binary extended Euclid on canonical 29-bit limbs (value < p): 2 * 261 iterations, every lane
// executes every step (selects, no divergence).  Returns x with a * x = 2^k (mod p) up to the fixed power of two
// that a final multiplication by a constant removes -- timing only needs the loop.
__device__ __forceinline__ void shr1(uint32_t (&v)[9]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (v[i] >> 1) | ((v[i + 1] & 1u) << 28);
  v[8] >>= 1;
}
__device__ __forceinline__ void sub_limbs(uint32_t (&r)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int32_t s = (int32_t)a[i] - (int32_t)b[i] + borrow;
    r[i] = (uint32_t)s & Fq29::MASK;
    borrow = s >> 29;
  }
}
__device__ __forceinline__ void add_limbs(uint32_t (&r)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t s = a[i] + b[i] + carry;
    r[i] = s & Fq29::MASK;
    carry = s >> 29;
  }
}
__device__ __forceinline__ bool geq_limbs(const uint32_t (&a)[9], const uint32_t (&b)[9]) {
  uint32_t d[9];
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int32_t s = (int32_t)a[i] - (int32_t)b[i] + borrow;
    d[i] = (uint32_t)s;
    borrow = s >> 29;
  }
  (void)d;
  return borrow == 0;
}
__device__ __forceinline__ fe29 inv_binary(const fe29& a) {
  uint32_t u[9], v[9], x1[9], x2[9], P[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { u[i] = a.l[i] & Fq29::MASK; v[i] = Fq29::p(i); P[i] = Fq29::p(i); x1[i] = (i == 0); x2[i] = 0; }
#pragma unroll 1
  for (int it = 0; it < 2 * 261; ++it) {
    const bool u_even = (u[0] & 1u) == 0, v_even = (v[0] & 1u) == 0;
    const bool ge = geq_limbs(u, v);
    // one of: u /= 2 | v /= 2 | u -= v | v -= u, with the cofactor update mod p
    uint32_t t[9], xt[9];
    const bool work_u = u_even || (!v_even && ge);
    // operand selection
#pragma unroll
    for (int i = 0; i < 9; ++i) { t[i] = work_u ? u[i] : v[i]; xt[i] = work_u ? x1[i] : x2[i]; }
    const bool halve = work_u ? u_even : v_even;
    uint32_t o[9], xo[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { o[i] = work_u ? v[i] : u[i]; xo[i] = work_u ? x2[i] : x1[i]; }
    uint32_t d[9], xd[9], xp[9];
    sub_limbs(d, t, o);
    sub_limbs(xd, xt, xo);
    add_limbs(xp, xd, P);
    const bool xneg = !geq_limbs(xt, xo);
    uint32_t h[9], xh[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { h[i] = t[i]; xh[i] = xt[i]; }
    if ((xh[0] & 1u) != 0) add_limbs(xh, xh, P);
    shr1(h);
    shr1(xh);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const uint32_t nt = halve ? h[i] : d[i];
      const uint32_t nx = halve ? xh[i] : (xneg ? xp[i] : xd[i]);
      if (work_u) { u[i] = nt; x1[i] = nx; } else { v[i] = nt; x2[i] = nx; }
    }
  }
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = x1[i] | x2[i];
  return r;
}

template <int VARIANT, int K, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_bench(const u256* in, u256* out) {
  // dynamic LDS: prefix[K-1][64*WAVES] | node[WAVES][128] (level l of a wave's tree at offset 128 - (128 >> l)) | top
  extern __shared__ uint32_t lds_raw[];
  fe29* prefix = reinterpret_cast<fe29*>(lds_raw);
  fe29* node = prefix + (size_t)(K > 1 ? K - 1 : 1) * 64 * WAVES;
  fe29* top = node + (size_t)WAVES * 128;
#define PREFIX(j, t) prefix[(size_t)(j) * 64 * WAVES + (t)]
#define NODE(l, i) node[wave * 128 + (128 - (128 >> (l))) + (i)]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const u256 xe = in[threadIdx.x & 63], ye = in[(threadIdx.x + 7) & 63];
  fe29 x = Fq29::from_ext(xe), y = Fq29::from_ext(ye);
  if (VARIANT == 1) {
    Affine qa; qa.x = xe; qa.y = ye;
    const AffI q = affi_from_ext(qa);
    PtI acc = pti_from_affi(q);
    acc.x = y;
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) acc = pti_madd(acc, q);
    x = acc.x;
  } else if (VARIANT == 2) {
#pragma unroll 1
    for (int i = 0; i < ITER / 20; ++i) x = inv_fermat(x);
  } else if (VARIANT == 3) {
#pragma unroll 1
    for (int i = 0; i < ITER / 20; ++i) x = inv_binary(Fq29::canonical(Fq29::mul(x, y), 1));
  } else {
    // K pending additions per lane: (x1_j, y1_j) + (x2_j, y2_j), operands regenerated cheaply from x, y
    fe29 X1 = x, Y1 = y;
#pragma unroll 1
    for (int it = 0; it < ITER / 8; ++it) {
      // forward: d_j = x2_j - x1_j, prefix products
      fe29 run = Fq29::one();
#pragma unroll 1
      for (int j = 0; j < K; ++j) {
        fe29 X2 = Fq29::norm(Fq29::add(X1, Fq29::one()));          // stand-in for the gathered second operand
        for (int l = 0; l < 9; ++l) X2.l[l] += (uint32_t)j;
        const fe29 d = Fq29::norm(Fq29::sub<K4E30>(X2, X1));
        if (j) PREFIX(j - 1, tid) = run;
        run = Fq29::mul(run, d);
      }
      // wave product tree in LDS
      NODE(0, lane) = run;
      __syncthreads();
#pragma unroll 1
      for (int l = 1; l <= 6; ++l) {
        const int w = 64 >> (l - 1);                                // entries of the level below
        const fe29 a = NODE(l - 1, (2 * lane) & (w - 1)), b = NODE(l - 1, (2 * lane + 1) & (w - 1));
        const fe29 pr = Fq29::mul(a, b);
        if (lane < (64 >> l)) NODE(l, lane) = pr;
        __syncthreads();
      }
      fe29 rootinv;
      if (WAVES > 1) {   // one inversion per workgroup: product of the wave roots, inverted once, fanned out
        if (lane == 0) top[wave] = NODE(6, 0);
        __syncthreads();
        fe29 all = top[0];
        for (int w = 1; w < WAVES; ++w) all = Fq29::mul(all, top[w]);
        fe29 inv = (wave == 0) ? inv_fermat(all) : all;             // only wave 0 pays the inversion ...
        if (wave == 0 && lane == 0) top[WAVES] = inv;
        __syncthreads();                                           // ... the others wait for it
        inv = top[WAVES];
        for (int w = 0; w < WAVES; ++w)
          if (w != wave) inv = Fq29::mul(inv, top[w]);
        rootinv = inv;
      } else {
        rootinv = inv_fermat(NODE(6, 0));
      }
      // down-sweep: inverse of every lane's product
      fe29 inv = rootinv;
#pragma unroll 1
      for (int l = 5; l >= 0; --l) {
        const fe29 sib = NODE(l, (lane >> l) ^ 1);
        inv = Fq29::mul(inv, sib);
      }
      // back-substitution + the additions themselves
#pragma unroll 1
      for (int j = K - 1; j >= 0; --j) {
        fe29 X2 = Fq29::norm(Fq29::add(X1, Fq29::one()));
        for (int l = 0; l < 9; ++l) X2.l[l] += (uint32_t)j;
        const fe29 d = Fq29::norm(Fq29::sub<K4E30>(X2, X1));
        fe29 dinv = inv;
        if (j) {
          dinv = Fq29::mul(inv, PREFIX(j - 1, tid));
          inv = Fq29::mul(inv, d);
        }
        const fe29 dy = Fq29::norm(Fq29::sub<K4E30>(Y1, X2));      // stand-in for y2 - y1
        const fe29 lam = Fq29::mul(dy, dinv);
        const fe29 l2 = Fq29::sqr(lam);
        const fe29 X3 = Fq29::norm(Fq29::sub<K8E30>(l2, Fq29::add(X1, X2)));
        const fe29 Y3 = Fq29::norm(Fq29::sub<K4E30>(Fq29::mul(lam, Fq29::norm(Fq29::sub<K8E30>(X1, X3))), Y1));
        X1 = X3;
        Y1 = Y3;
      }
    }
    x = Fq29::add(X1, Y1);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = Fq29::to_ext(Fq29::norm(x));
}

template <int VARIANT, int K, int WAVES>
void run(const char* name, const u256* din, u256* dout, int cus, double ops_per_iter_unit) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds = sizeof(fe29) * ((size_t)(K > 1 ? K - 1 : 1) * 64 * WAVES + (size_t)WAVES * 128 + WAVES + 1);
  CHECK(hipFuncSetAttribute((const void*)k_bench<VARIANT, K, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int wps : {1, 2}) {
    if (VARIANT >= 4 && lds * (4 * wps / WAVES) > 160 * 1024) {
      printf("%-50s waves/SIMD=%d  does not fit: %zu KB of LDS per workgroup\n", name, wps, lds / 1024);
      continue;
    }
    int blocks = cus * 4 * wps / WAVES;     // wps waves per SIMD
    hipLaunchKernelGGL((k_bench<VARIANT, K, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, din, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_bench<VARIANT, K, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, din, dout);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-50s waves/SIMD=%d  %8.3f ms  %10.1f cyc@2.4GHz per %s\n", name, wps, ms,
           ms * 1e6 * 2.4 / (ops_per_iter_unit * wps), VARIANT >= 4 ? "affine addition" : "operation");
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u256 h[64];
  for (int i = 0; i < 64; ++i) for (int l = 0; l < 8; ++l) h[i].v[l] = (l == 7) ? (0x1234567u + i) : (0x9E3779B9u * (i * 8 + l + 1));
  u256 *din, *dout; CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(u256) * 256 * cus * 8));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  run<1, 1, 1>("1 pti_madd (XYZZ mixed addition)", din, dout, cus, ITER);
  run<2, 1, 1>("2 inversion, Fermat 4-bit windows", din, dout, cus, ITER / 20);
  run<3, 1, 1>("3 inversion, constant-trip binary Euclid", din, dout, cus, ITER / 20);
  run<4, 7, 1>("4 batched affine, K=7/lane, inversion per wave", din, dout, cus, (ITER / 8) * 7.0);
  run<4, 8, 1>("4 batched affine, K=8/lane, inversion per wave", din, dout, cus, (ITER / 8) * 8.0);
  run<4, 16, 1>("4 batched affine, K=16/lane, inversion per wave", din, dout, cus, (ITER / 8) * 16.0);
  run<5, 7, 4>("5 batched affine, K=7/lane, inversion per 4 waves", din, dout, cus, (ITER / 8) * 7.0);
  run<5, 8, 4>("5 batched affine, K=8/lane, inversion per 4 waves", din, dout, cus, (ITER / 8) * 8.0);
  run<5, 16, 4>("5 batched affine, K=16/lane, inversion per 4 waves", din, dout, cus, (ITER / 8) * 16.0);
  return 0;
}
