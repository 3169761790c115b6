// Does the 29-bit-limb field arithmetic get faster with more waves per SIMD?  Measured in shader cycles by the
// waves themselves (s_memtime) at occupancies the launch really has: one 64-lane workgroup per wave, the waves per SIMD
// capped through the REGISTER allocation (the kernel touches VGPR 512 / k - 1, so exactly k waves fit a SIMD's 512; an
// LDS cap limits the waves per CU but lets the dispatcher put eight on one SIMD and none on the next -- the first
// version of this benchmark did that, and its odd "4 waves are slower than 3" rows were placement).  Every wave also
// reports HW_ID, and the host prints how many waves shared a SIMD.
// (fq29_bench.hip launches "waves per SIMD" x CUs workgroups of 256 lanes whatever the kernel's register count allows:
// its 3- and 4-wave rows of pti_madd ran two waves at a time.)
//   mul        Fq29::mul, 17 column sums live (the shipped form)
//   fips       Fq29::fips, one running column (product scanning)
//   madd       pti_madd (the mixed addition of the accumulate kernel), register-only loop
//   madd_lean  pti_madd_lean
//   straight   16 independent v_mad_u64_u32 repeated 128 times in a row (a 16 KiB loop body of nothing but multiply-adds)
//   tight      the same 16 multiply-adds as a 128-byte loop body
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "../../metal-msm-gpu-acceleration_amd/csrc/bn254_ec29.hip.h"
using namespace msm_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define MADV(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[i]) : "v"(a), "v"(b) : "vcc");
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define R8X(X) X X X X X X X X
#define STRAIGHT R8X(R8X(R16(MADV) R16(MADV)))   // 8 * 8 * 32 = 2048 multiply-adds

template <int K> __device__ __forceinline__ void pin_registers() {
  if (K == 1) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
  if (K == 2) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (K == 3) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (K == 4) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (K == 5) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (K == 6) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if (K == 8) asm volatile("v_mov_b32 v63, 0" ::: "v63");
}

template <int V, int K>
__global__ void __launch_bounds__(64) k_op(const u256* in, uint64_t* out, int iters) {
  pin_registers<K>();
  const u256 xe = in[threadIdx.x & 63], ye = in[(threadIdx.x + 7) & 63];
  uint64_t t0 = 0, t1 = 0;
  uint32_t sinkv = 0;
  if (V == 0 || V == 1) {
    fe29 x = Fq29::from_ext(xe), y = Fq29::from_ext(ye);
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
      if (V == 0) x = Fq29::mul(x, y);
      if (V == 1) x = fips_mul(x, y);
    }
    t1 = __builtin_amdgcn_s_memtime();
    sinkv = x.l[0] ^ x.l[8];
  } else if (V == 2 || V == 3) {
    Affine qa; qa.x = xe; qa.y = ye;           // not a curve point: timing only (no exceptional path is taken)
    const AffI q = affi_from_ext(qa);
    PtI acc = pti_from_affi(q);
    acc.x = Fq29::from_ext(ye);
    const auto again = [&]() { return q; };
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
      if (V == 2) acc = pti_madd(acc, q);
      if (V == 3) pti_madd_lean(acc, q, again, [] {});
    }
    t1 = __builtin_amdgcn_s_memtime();
    sinkv = acc.x.l[0] ^ acc.y.l[3] ^ acc.zz.l[1] ^ acc.zzz.l[2];
  } else {
    uint64_t m[16];
    uint32_t a = xe.v[0], b = ye.v[1];
    for (int i = 0; i < 16; ++i) m[i] = xe.v[i & 7] + i;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
      if (V == 4) { STRAIGHT }
      if (V == 5) { R16(MADV) }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) sinkv ^= (uint32_t)m[i];
  }
  if (sinkv == 0x12345u) out[0] = sinkv;
  if (threadIdx.x == 0) {
    out[1 + 2 * blockIdx.x] = t1 - t0;
    // HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID register 20, bits [3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    out[2 + 2 * blockIdx.x] = ((uint64_t)xcc << 32) | hw;
  }
}

template <int V, int K>
void run_one(const u256* din, uint64_t* dout, int cus, int iters, double ops_per_iter) {
  const int blocks = cus * 4 * K;
  hipLaunchKernelGGL((k_op<V, K>), dim3(blocks), dim3(64), 0, 0, din, dout, iters);
  hipLaunchKernelGGL((k_op<V, K>), dim3(blocks), dim3(64), 0, 0, din, dout, iters);
  std::vector<uint64_t> h(1 + 2 * blocks);
  CHECK(hipMemcpy(h.data(), dout, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<uint64_t> t;
  std::map<uint64_t, int> per_simd;
  for (int b = 0; b < blocks; ++b) {
    t.push_back(h[1 + 2 * b]);
    const uint64_t id = h[2 + 2 * b];
    per_simd[((id >> 32) << 16) | ((id & 0xFFFF) >> 4)]++;   // xcc, se, sh, cu, simd
  }
  std::sort(t.begin(), t.end());
  int lo = 1 << 30, hi = 0;
  for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
  // cycles per operation per SIMD = wave cycles / (operations per wave x waves per SIMD)
  printf("  %dw %8.1f..%8.1f [%zu SIMDs, %d..%d waves each]", K, (double)t[0] / (iters * ops_per_iter * K),
         (double)t.back() / (iters * ops_per_iter * K), per_simd.size(), lo, hi);
}
template <int V>
void run(const char* name, const u256* din, uint64_t* dout, int cus, int iters, double ops_per_iter, int max_k) {
  printf("%-10s", name);
  run_one<V, 1>(din, dout, cus, iters, ops_per_iter);
  run_one<V, 2>(din, dout, cus, iters, ops_per_iter);
  if (max_k >= 3) run_one<V, 3>(din, dout, cus, iters, ops_per_iter);
  if (max_k >= 4) run_one<V, 4>(din, dout, cus, iters, ops_per_iter);
  if (max_k >= 5) run_one<V, 5>(din, dout, cus, iters, ops_per_iter);
  if (max_k >= 6) run_one<V, 6>(din, dout, cus, iters, ops_per_iter);
  if (max_k >= 8) run_one<V, 8>(din, dout, cus, iters, ops_per_iter);
  printf("\n");
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u256 h[64];
  for (int i = 0; i < 64; ++i) for (int l = 0; l < 8; ++l) h[i].v[l] = (l == 7) ? (0x1234567u + i) : (0x9E3779B9u * (i * 8 + l + 1));
  u256* din; uint64_t* dout;
  CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(uint64_t) * (1 + 2 * cus * 4 * 8)));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k_op<0, 2>), dim3(cus * 8), dim3(64), 0, 0, din, dout, 20000);   // ~1 s of load first
  CHECK(hipDeviceSynchronize());
  printf("cycles per operation per SIMD (fastest .. slowest wave) at k waves per SIMD\n");
  run<0>("mul", din, dout, cus, 6000, 1, 6);
  run<1>("fips", din, dout, cus, 6000, 1, 6);
  run<2>("madd", din, dout, cus, 800, 1, 3);
  run<3>("madd_lean", din, dout, cus, 800, 1, 4);
  run<4>("straight", din, dout, cus, 40, 2048, 8);     // per multiply-add
  run<5>("tight", din, dout, cus, 5120, 16, 8);        // per multiply-add
  return 0;
}
