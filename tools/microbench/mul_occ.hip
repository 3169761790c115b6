// Does the 29-bit-limb field arithmetic get faster with more waves per SIMD?  No: read the LAUNCH column (HIP-event time of
// the whole launch per operation per SIMD).  The per-wave columns (a wave's own s_memtime cycles / k) overstate the
// throughput beyond two waves, because the waves of a SIMD are not resident for the same span.  Occupancies are real: one 64-lane workgroup per wave, the waves per SIMD
// capped through the REGISTER allocation (the kernel touches VGPR 512 / k - 1, so exactly k waves fit a SIMD's 512; an
// LDS cap limits the waves per CU but lets the dispatcher put eight on one SIMD and none on the next -- the first
// version of this benchmark did that, and its odd "4 waves are slower than 3" rows were placement).  Every wave also
// reports HW_ID, and the host prints how many waves shared a SIMD.
// (fq29_bench.hip launches "waves per SIMD" x CUs workgroups of 256 lanes whatever the kernel's register count allows:
// its 3- and 4-wave rows of pti_madd ran two waves at a time.)
//   mul        Fq29::mul, 17 column sums live (the shipped form)
//   mul x10    the same multiplication unrolled ten times (the loop body grows from ~1.8 KiB to ~18 KiB), x40: ~72 KiB
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
#include "asm_bench.inc"
#if __has_include("cmul_bench.inc")
#include "cmul_bench.inc"   // experiment: the COMPILER's multiplication loop body transplanted into a statement
#endif   // tools/gen_accumulate_asm.py with MSM_ASM_BENCH=1: the kernel statement + timing-only statements
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
  const uint64_t q0 = __builtin_amdgcn_s_memrealtime();
  uint32_t sinkv = 0;
  if (V == 8) {   // the compiler's multiplication chain in a kernel that allocates the asm kernel's 6912 B of LDS
    __shared__ uint32_t lds_dummy[MSM_ACC_ASM_LDS_BYTES / 4];
    lds_dummy[threadIdx.x] = xe.v[0];
    sinkv ^= lds_dummy[(threadIdx.x + 1) & 63];
  }
  if (V == 0 || V == 1 || V == 6 || V == 7 || V == 8) {
    fe29 x = Fq29::from_ext(xe), y = Fq29::from_ext(ye);
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
      if (V == 0 || V == 8) x = Fq29::mul(x, y);
      if (V == 1) x = fips_mul(x, y);
      if (V == 6) {   // the same multiplication ten times in a row: a loop body of ~18 KiB instead of ~1.8 KiB
#define MUL1 x = Fq29::mul(x, y); asm volatile("" : "+v"(x.l[0]), "+v"(x.l[8]));
#define MUL10 MUL1 MUL1 MUL1 MUL1 MUL1 MUL1 MUL1 MUL1 MUL1 MUL1
        MUL10
      }
      if (V == 7) {   // forty times: ~72 KiB, more than the 64 KiB instruction cache two CUs share
        MUL10 MUL10 MUL10 MUL10
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    sinkv = x.l[0] ^ x.l[8];
  } else if (V == 9) {   // the LDS-parked column-form addition of accumulate_kernel_park (experiments/ec29_variants.inc)
    __shared__ uint32_t parkmem[27 * 64];
    struct Pk {
      uint32_t* base;
      __device__ fe29 load(int c) const { fe29 r; for (int l = 0; l < 9; ++l) r.l[l] = base[(9 * c + l) * 64]; return r; }
      __device__ void store(int c, const fe29& v) const { for (int l = 0; l < 9; ++l) base[(9 * c + l) * 64] = v.l[l]; }
    } pk{parkmem + threadIdx.x};
    Affine qa; qa.x = xe; qa.y = ye;
    const AffI q = affi_from_ext(qa);
    fe29 X = Fq29::from_ext(ye);
    pk.store(0, q.y); pk.store(1, Fq29::one()); pk.store(2, Fq29::one());
    const auto again = [&]() { return q; };
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) pti_madd_park(X, pk, q, again, [] {});
    t1 = __builtin_amdgcn_s_memtime();
    sinkv = X.l[0] ^ pk.load(0).l[3];
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
    const uint64_t q1 = __builtin_amdgcn_s_memrealtime();
    out[1 + 2 * blockIdx.x] = ((t1 - t0) & 0xFFFFFFFFFFull) | ((q1 - q0) << 40);   // cycles | 100 MHz ticks
    out[40000 + 2 * blockIdx.x] = q0;
    out[40001 + 2 * blockIdx.x] = q1;
    // HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID register 20, bits [3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    out[2 + 2 * blockIdx.x] = ((uint64_t)xcc << 32) | hw;
  }
}

// The generated statement of accumulate_kernel_asm (tools/gen_accumulate_asm.py) on one work item of `iters` points
// per lane that all come from a 2-record table: the cost of its point addition without the memory system behind it.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_asm(const AffPacked* bases, const uint32_t* idx, uint32_t iters, PtI* sink, uint32_t* redo_list, uint32_t* redo_count, uint64_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t park[MSM_ACC_ASM_LDS_BYTES / 4];
  uint64_t idxp = (uint64_t)idx;
  const uint64_t outp = (uint64_t)&sink[blockIdx.x * 64 + threadIdx.x];
  const uint32_t slot = blockIdx.x * 64 + threadIdx.x, cnt = iters;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)park;
  const uint32_t lds128 = lds0 + threadIdx.x * 16u, lds32 = lds0 + 6u * 1024u + threadIdx.x * 4u;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  asm volatile(MSM_ACC_ASM_TEXT
               : [idxp] "+v"(idxp)
               : [cnt] "v"(cnt), [outp] "v"(outp), [slot] "v"(slot), [lds128] "v"(lds128), [lds32] "v"(lds32),
                 [bases] "s"(bases), [redo_list] "s"(redo_list), [redo_count] "s"(redo_count),
                 [p0] "s"(Fq29::p(0)), [p1] "s"(Fq29::p(1)), [p2] "s"(Fq29::p(2)), [p3] "s"(Fq29::p(3)),
                 [p4] "s"(Fq29::p(4)), [p5] "s"(Fq29::p(5)), [p6] "s"(Fq29::p(6)), [p7] "s"(Fq29::p(7)),
                 [p8] "s"(Fq29::p(8)), [inv] "s"(Fq29::INV), [pinv] "s"(Fq29::PINV)
               : MSM_ACC_ASM_CLOBBERS);
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[1 + 2 * blockIdx.x] = ((t1 - t0) & 0xFFFFFFFFFFull) | ((q1 - q0) << 40);
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    out[2 + 2 * blockIdx.x] = ((uint64_t)xcc << 32) | hw;
  }
}

// Timing-only statements of the generator: B = 0 the row-form multiplication in a chain, 1 with the simple-instruction
// tail of a subtraction + carry round, 2 with the operand parked in LDS and fetched back every trip.
template <int B, int K>
__global__ void __launch_bounds__(64) k_asmb(uint32_t iters, uint64_t* out) {
  pin_registers<K>();
  uint32_t lds0 = 0;
  if (B != 6) {   // B == 6: the multiplication statement in a kernel WITHOUT an LDS allocation
    __shared__ __attribute__((aligned(16))) uint32_t park[MSM_ACC_ASM_LDS_BYTES / 4];
    lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)park;
  }
  const uint32_t lds128 = lds0 + threadIdx.x * 16u, lds32 = lds0 + 6u * 1024u + threadIdx.x * 4u, cnt = iters;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
#define BENCH_OPERANDS : : [cnt] "v"(cnt), [scnt] "s"(iters), [seed] "v"(threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u), [lds128] "v"(lds128), [lds32] "v"(lds32), \
                 [p0] "s"(Fq29::p(0)), [p1] "s"(Fq29::p(1)), [p2] "s"(Fq29::p(2)), [p3] "s"(Fq29::p(3)), \
                 [p4] "s"(Fq29::p(4)), [p5] "s"(Fq29::p(5)), [p6] "s"(Fq29::p(6)), [p7] "s"(Fq29::p(7)), \
                 [p8] "s"(Fq29::p(8)), [inv] "s"(Fq29::INV) : MSM_ACC_ASM_CLOBBERS
  if (B == 0 || B == 6) asm volatile(MSM_ACC_ASM_BENCH_MUL BENCH_OPERANDS);
  if (B == 1) asm volatile(MSM_ACC_ASM_BENCH_MULSUB BENCH_OPERANDS);
  if (B == 2) asm volatile(MSM_ACC_ASM_BENCH_MULPARK BENCH_OPERANDS);
  if (B == 3) asm volatile(MSM_ACC_ASM_BENCH_MUL_SGPR BENCH_OPERANDS);
  if (B == 4) asm volatile(MSM_ACC_ASM_BENCH_MUL_BANKS BENCH_OPERANDS);
  if (B == 5) asm volatile(MSM_ACC_ASM_BENCH_MUL_NONOP BENCH_OPERANDS);
  if (B == 7) asm volatile(MSM_ACC_ASM_BENCH_MUL_IL BENCH_OPERANDS);
  if (B == 8) asm volatile(MSM_ACC_ASM_BENCH_MUL_PV BENCH_OPERANDS);
#if defined(MSM_ACC_ASM_BENCH_CMUL)
  if (B == 9) asm volatile(MSM_ACC_ASM_BENCH_CMUL BENCH_OPERANDS);
  if (B == 10) asm volatile(MSM_ACC_ASM_BENCH_CMUL_S BENCH_OPERANDS);
  if (B == 11) asm volatile(MSM_ACC_ASM_BENCH_CMUL_D BENCH_OPERANDS);
#endif
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[40000 + 2 * blockIdx.x] = q0;   // absolute 100 MHz ticks: were the waves of a SIMD resident TOGETHER?
    out[40001 + 2 * blockIdx.x] = q1;
    out[1 + 2 * blockIdx.x] = ((t1 - t0) & 0xFFFFFFFFFFull) | ((q1 - q0) << 40);
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    out[2 + 2 * blockIdx.x] = ((uint64_t)xcc << 32) | hw;
  }
}

template <int B, int K>
void run_asmb_one(uint64_t* dout, int cus, uint32_t iters) {
  const int blocks = cus * 4 * K;
  hipLaunchKernelGGL((k_asmb<B, K>), dim3(blocks), dim3(64), 0, 0, iters, dout);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_asmb<B, K>), dim3(blocks), dim3(64), 0, 0, iters, dout);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float launch_ms = 0; CHECK(hipEventElapsedTime(&launch_ms, e0, e1));
  std::vector<uint64_t> h2(40000 + 2 * blocks + 2);
  CHECK(hipMemcpy(h2.data(), dout, h2.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<uint64_t> t; std::vector<double> ghz; std::map<uint64_t, int> per_simd;
  std::map<uint64_t, std::vector<std::pair<uint64_t, uint64_t>>> spans;
  for (int b = 0; b < blocks; ++b) {
    t.push_back(h2[1 + 2 * b] & 0xFFFFFFFFFFull);
    ghz.push_back((double)(h2[1 + 2 * b] & 0xFFFFFFFFFFull) / ((double)(h2[1 + 2 * b] >> 40) * 10.0));
    const uint64_t id = h2[2 + 2 * b];
    const uint64_t key = ((id >> 32) << 16) | ((id & 0xFFFF) >> 4);
    per_simd[key]++;
    spans[key].push_back({h2[40000 + 2 * b], h2[40001 + 2 * b]});
  }
  std::sort(t.begin(), t.end()); std::sort(ghz.begin(), ghz.end());
  int lo = 1 << 30, hi = 0;
  for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
  const double g = ghz[ghz.size() / 2], slow = (double)t.back() / ((double)iters * K);
  printf("  %dw %7.1f..%7.1f cyc @%.2f GHz = %6.1f ns [%d..%d w/SIMD] LAUNCH %6.1f ns", K, (double)t[0] / ((double)iters * K), slow, g, slow / g, lo, hi,
         launch_ms * 1e6 / ((double)iters * K));
  if (K == 5 && B == 0) {   // residency of the waves of one SIMD (us from the first start)
    auto& v = spans.begin()->second;
    std::sort(v.begin(), v.end());
    printf("\n      one SIMD's waves, start..end in us:");
    for (auto& se : v) printf(" %.0f..%.0f", (se.first - v[0].first) / 100.0, (se.second - v[0].first) / 100.0);
  }
}
template <int B>
void run_asmb(const char* name, uint64_t* dout, int cus) {
  printf("%-10s", name);
  run_asmb_one<B, 1>(dout, cus, 4000);
  run_asmb_one<B, 2>(dout, cus, 4000);
  run_asmb_one<B, 3>(dout, cus, 4000);
  run_asmb_one<B, 4>(dout, cus, 4000);
  run_asmb_one<B, 5>(dout, cus, 4000);
  printf("\n");
  fflush(stdout);
}

void run_asm(uint64_t* dout, int cus) {
  const uint32_t iters = 600;
  AffPacked h[2];
  for (int k = 0; k < 2; ++k) for (int l = 0; l < 8; ++l) { h[k].x.v[l] = 0x01234567u * (l + 3 * k + 1) >> (l == 7 ? 4 : 0); h[k].y.v[l] = 0x089ABCDEu * (l + 5 * k + 2) >> (l == 7 ? 4 : 0); }
  std::vector<uint32_t> hidx(iters + 4);
  for (uint32_t i = 0; i < hidx.size(); ++i) hidx[i] = i & 1;   // P0 + P1 + P0 + ...: never cancels (a cancelling pattern flags every lane)
  AffPacked* db; uint32_t* didx; PtI* sink; uint32_t* redo;
  CHECK(hipMalloc(&db, sizeof h)); CHECK(hipMalloc(&didx, hidx.size() * 4)); CHECK(hipMalloc(&sink, sizeof(PtI) * cus * 4 * 5 * 64));
  CHECK(hipMalloc(&redo, 4 * (cus * 4 * 5 * 64 + 16)));
  CHECK(hipMemcpy(db, h, sizeof h, hipMemcpyHostToDevice)); CHECK(hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
  printf("%-10s", "asm madd");
  for (int K = 1; K <= 5; ++K) {
    const int blocks = cus * 4 * K;
    // 5 fit a SIMD by registers; fewer resident ones by launching fewer (the dispatcher spreads 64-lane workgroups evenly
    // when the CUs are otherwise empty -- the HW_ID census below shows what it did)
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(redo, 0, 64));   // the redo list holds one entry per lane of ONE launch
      hipLaunchKernelGGL(k_asm, dim3(blocks), dim3(64), 0, 0, db, didx, iters, sink, redo + 16, redo, dout);
    }
    std::vector<uint64_t> h2(1 + 2 * blocks);
    CHECK(hipMemcpy(h2.data(), dout, h2.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> t; std::vector<double> ghz; std::map<uint64_t, int> per_simd;
    for (int b = 0; b < blocks; ++b) {
      t.push_back(h2[1 + 2 * b] & 0xFFFFFFFFFFull);
      ghz.push_back((double)(h2[1 + 2 * b] & 0xFFFFFFFFFFull) / ((double)(h2[1 + 2 * b] >> 40) * 10.0));
      const uint64_t id = h2[2 + 2 * b];
      per_simd[((id >> 32) << 16) | ((id & 0xFFFF) >> 4)]++;
    }
    std::sort(t.begin(), t.end()); std::sort(ghz.begin(), ghz.end());
    int lo = 1 << 30, hi = 0;
    for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
    const double g = ghz[ghz.size() / 2], slow = (double)t.back() / ((double)iters * K);
    printf("  %dw %7.1f..%7.1f cyc @%.2f GHz = %6.1f ns [%d..%d w/SIMD]", K, (double)t[0] / ((double)iters * K), slow, g, slow / g, lo, hi);
  }
  printf("\n");
}

template <int V, int K>
void run_one(const u256* din, uint64_t* dout, int cus, int iters, double ops_per_iter) {
  const int blocks = cus * 4 * K;
  hipLaunchKernelGGL((k_op<V, K>), dim3(blocks), dim3(64), 0, 0, din, dout, iters);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_op<V, K>), dim3(blocks), dim3(64), 0, 0, din, dout, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float launch_ms = 0; CHECK(hipEventElapsedTime(&launch_ms, e0, e1));
  std::vector<uint64_t> h(40000 + 2 * blocks + 2);
  CHECK(hipMemcpy(h.data(), dout, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<uint64_t> t;
  std::map<uint64_t, int> per_simd;
  std::map<uint64_t, std::vector<std::pair<uint64_t, uint64_t>>> spans;
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) {
    spans[((h[2 + 2 * b] >> 32) << 16) | ((h[2 + 2 * b] & 0xFFFF) >> 4)].push_back({h[40000 + 2 * b], h[40001 + 2 * b]});
    t.push_back(h[1 + 2 * b] & 0xFFFFFFFFFFull);
    ghz.push_back((double)(h[1 + 2 * b] & 0xFFFFFFFFFFull) / ((double)(h[1 + 2 * b] >> 40) * 10.0));
    const uint64_t id = h[2 + 2 * b];
    per_simd[((id >> 32) << 16) | ((id & 0xFFFF) >> 4)]++;   // xcc, se, sh, cu, simd
  }
  std::sort(t.begin(), t.end());
  int lo = 1 << 30, hi = 0;
  for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
  // cycles per operation per SIMD = wave cycles / (operations per wave x waves per SIMD)
  std::sort(ghz.begin(), ghz.end());
  const double g = ghz[ghz.size() / 2], slow = (double)t.back() / (iters * ops_per_iter * K);
  // cycles: fastest .. slowest wave; shader clock (median wave, from s_memrealtime); ns per operation per SIMD at that clock
  printf("  %dw %7.1f..%7.1f cyc @%.2f GHz = %6.1f ns [%d..%d w/SIMD] LAUNCH %6.1f ns", K, (double)t[0] / (iters * ops_per_iter * K), slow, g,
         slow / g, lo, hi, launch_ms * 1e6 / (iters * ops_per_iter * K));
  if (false) {
    auto& v = spans.begin()->second;
    std::sort(v.begin(), v.end());
    printf("\n      one SIMD's waves, start..end in us:");
    for (auto& se : v) printf(" %.0f..%.0f", (se.first - v[0].first) / 100.0, (se.second - v[0].first) / 100.0);
  }
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
  CHECK(hipMalloc(&din, sizeof(h))); CHECK(hipMalloc(&dout, sizeof(uint64_t) * (40000 + 2 * cus * 4 * 8 + 16)));
  CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k_op<0, 2>), dim3(cus * 8), dim3(64), 0, 0, din, dout, 20000);   // ~1 s of load first
  CHECK(hipDeviceSynchronize());
  printf("cycles per operation per SIMD (fastest .. slowest wave), the shader clock during the run, ns per operation per SIMD\n");
  run<0>("mul", din, dout, cus, 6000, 1, 6);
  run<8>("mul +lds", din, dout, cus, 6000, 1, 6);
  run<6>("mul x10", din, dout, cus, 600, 10, 6);     // per multiplication
  run<7>("mul x40", din, dout, cus, 150, 40, 6);
  run<1>("fips", din, dout, cus, 6000, 1, 6);
  run<2>("madd", din, dout, cus, 800, 1, 3);
  run<3>("madd_lean", din, dout, cus, 800, 1, 4);
  run<9>("madd_park", din, dout, cus, 800, 1, 4);
  run<4>("straight", din, dout, cus, 40, 2048, 8);     // per multiply-add
  run<5>("tight", din, dout, cus, 5120, 16, 8);        // per multiply-add
  run_asmb<0>("asm mul", dout, cus);
  run_asmb<6>("asm mul-l", dout, cus);     // the same statement in a kernel without an LDS allocation
#if defined(MSM_ACC_ASM_BENCH_CMUL)
  run_asmb<9>("cmul asm", dout, cus);      // the compiler's loop body as a statement
  run_asmb<10>("cmul asm s", dout, cus);   // ... with a scalar loop counter (no VALU-written vcc branch)
  run_asmb<11>("cmul asm d", dout, cus);   // ... and every register initialised with lane-dependent 29-bit values
#endif
  run_asmb<7>("asm mul i", dout, cus);     // reduction of row i interleaved with the products of row i + 1
  run_asmb<8>("asm mul p", dout, cus);     // the limbs of p in VGPRs instead of SGPRs
  run_asmb<3>("asm mul s", dout, cus);     // carry-out to an SGPR pair instead of vcc
  run_asmb<4>("asm mul b", dout, cus);     // operand registers not 4-aligned
  run_asmb<5>("asm mul n", dout, cus);     // without the s_nop wait states (timing only)
  run_asmb<1>("asm mulsub", dout, cus);
  run_asmb<2>("asm mulpark", dout, cus);
  run_asm(dout, cus);
  return 0;
}
