// Does the issue interval of v_mad_u64_u32 on gfx950 depend on WHICH registers its operands sit in (VGPR bank conflicts
// in the operand fetch: bank = register number mod 4)?  The accumulate kernel's multiply-adds issue every ~5.3 cycles
// per SIMD with two waves resident; if an operand placement existed that brought this to 4, a register-aware
// multiplication would be worth writing by hand.  Every variant runs 8 independent multiply-adds per loop trip with
// PHYSICAL registers named in one inline-assembly block (the loop included, so the compiler cannot move anything).
//   acc pairs v[8:9], v[12:13], ..., v[36:37]  (banks 0,1) unless stated
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define ITER_S "4096"
constexpr int ITER = 4096;

#define CLOBBERS "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
  "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
  "v40", "v41", "v42", "v43", "s20", "s21", "s22", "s23", "vcc", "scc"

#define INIT \
  "v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %1\n v_mov_b32 v5, %2\n v_mov_b32 v6, %1\n v_mov_b32 v7, %2\n" \
  "v_mov_b32 v8, %1\n v_mov_b32 v9, %2\n v_mov_b32 v10, %1\n v_mov_b32 v11, %2\n v_mov_b32 v12, %1\n v_mov_b32 v13, %2\n" \
  "v_mov_b32 v14, %1\n v_mov_b32 v15, %2\n v_mov_b32 v16, %1\n v_mov_b32 v17, %2\n v_mov_b32 v18, %1\n v_mov_b32 v19, %2\n" \
  "v_mov_b32 v20, %1\n v_mov_b32 v21, %2\n v_mov_b32 v22, %1\n v_mov_b32 v23, %2\n v_mov_b32 v24, %1\n v_mov_b32 v25, %2\n" \
  "v_mov_b32 v26, %1\n v_mov_b32 v27, %2\n v_mov_b32 v28, %1\n v_mov_b32 v29, %2\n v_mov_b32 v30, %1\n v_mov_b32 v31, %2\n" \
  "v_mov_b32 v32, %1\n v_mov_b32 v33, %2\n v_mov_b32 v34, %1\n v_mov_b32 v35, %2\n v_mov_b32 v36, %1\n v_mov_b32 v37, %2\n" \
  "v_mov_b32 v38, %1\n v_mov_b32 v39, %2\n v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %1\n v_mov_b32 v43, %2\n" \
  "v_readfirstlane_b32 s22, %1\n v_readfirstlane_b32 s23, %2\n s_mov_b32 s20, " ITER_S "\n"
#define LOOP_END "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
#define FINISH \
  "v_xor_b32 %0, v8, v12\n v_xor_b32 %0, %0, v16\n v_xor_b32 %0, %0, v20\n v_xor_b32 %0, %0, v24\n v_xor_b32 %0, %0, v28\n" \
  "v_xor_b32 %0, %0, v32\n v_xor_b32 %0, %0, v36\n v_xor_b32 %0, %0, v9\n v_xor_b32 %0, %0, v13\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v11\n"

#define KERNEL(NAME, BODY)                                                                 \
__global__ void NAME(uint32_t* out, uint32_t s) {                                          \
  uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9e3779b9u, res;                    \
  asm volatile(INIT "1:\n" BODY LOOP_END FINISH : "=&v"(res) : "v"(a), "v"(b) : CLOBBERS); \
  if (res == 0x12345) out[0] = res;                                                        \
}

// 8 accumulators at v[8:9] + 4k (banks 0,1); sources given per variant
#define M8(S0, S1) \
  "v_mad_u64_u32 v[8:9], vcc, " S0 ", " S1 ", v[8:9]\n v_mad_u64_u32 v[12:13], vcc, " S0 ", " S1 ", v[12:13]\n" \
  "v_mad_u64_u32 v[16:17], vcc, " S0 ", " S1 ", v[16:17]\n v_mad_u64_u32 v[20:21], vcc, " S0 ", " S1 ", v[20:21]\n" \
  "v_mad_u64_u32 v[24:25], vcc, " S0 ", " S1 ", v[24:25]\n v_mad_u64_u32 v[28:29], vcc, " S0 ", " S1 ", v[28:29]\n" \
  "v_mad_u64_u32 v[32:33], vcc, " S0 ", " S1 ", v[32:33]\n v_mad_u64_u32 v[36:37], vcc, " S0 ", " S1 ", v[36:37]\n"
// accumulators at v[10:11] + 4k (banks 2,3)  (gfx950 wants 64-bit tuples at even registers: an accumulator is (0,1) or (2,3))
#define M8HI(S0, S1) \
  "v_mad_u64_u32 v[10:11], vcc, " S0 ", " S1 ", v[10:11]\n v_mad_u64_u32 v[14:15], vcc, " S0 ", " S1 ", v[14:15]\n" \
  "v_mad_u64_u32 v[18:19], vcc, " S0 ", " S1 ", v[18:19]\n v_mad_u64_u32 v[22:23], vcc, " S0 ", " S1 ", v[22:23]\n" \
  "v_mad_u64_u32 v[26:27], vcc, " S0 ", " S1 ", v[26:27]\n v_mad_u64_u32 v[30:31], vcc, " S0 ", " S1 ", v[30:31]\n" \
  "v_mad_u64_u32 v[34:35], vcc, " S0 ", " S1 ", v[34:35]\n v_mad_u64_u32 v[38:39], vcc, " S0 ", " S1 ", v[38:39]\n"
// a column of a schoolbook product: ONE accumulator, 8 different source pairs (the dependent form the kernel has per column,
// here 2 chains of 4 to keep two accumulators busy)
#define COL(A, B, C, D) \
  "v_mad_u64_u32 v[8:9], vcc, " A ", " B ", v[8:9]\n v_mad_u64_u32 v[12:13], vcc, " C ", " D ", v[12:13]\n"

KERNEL(k_distinct, M8("v2", "v3"))            // src banks 2,3; acc banks 0,1: no two operands share a bank
KERNEL(k_src_same_bank, M8("v2", "v6"))       // both sources in bank 2
KERNEL(k_src_acc_lo, M8("v4", "v3"))          // src0 in bank 0 = bank of the accumulator's low half
KERNEL(k_src_acc_both, M8("v4", "v5"))        // src0 bank 0, src1 bank 1: both collide with the accumulator pair
KERNEL(k_all_bank0, M8("v4", "v40"))          // src0, src1 and acc.lo all in bank 0
KERNEL(k_hi_acc, M8HI("v4", "v5"))            // accumulator pairs in banks 2,3, sources banks 0,1
KERNEL(k_sgpr_src, M8("s22", "v3"))           // one source a scalar register (the reduction's constant limbs of p)
KERNEL(k_sgpr_both, M8("s22", "s22"))         // (one SGPR read twice; two different ones exceed the constant bus)
KERNEL(k_same_src, M8("v2", "v2"))            // squaring: one register read twice
KERNEL(k_dependent2, COL("v2", "v3", "v6", "v7") COL("v10", "v11", "v14", "v15") COL("v18", "v19", "v22", "v23") COL("v26", "v27", "v30", "v31"))

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t); };
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* d; CHECK(hipMalloc(&d, 4096));
  Entry es[] = {{"sources b2,b3  acc b0,b1", k_distinct}, {"sources b2,b2", k_src_same_bank}, {"src0 b0 (= acc.lo)", k_src_acc_lo},
                {"sources b0,b1 (= acc pair)", k_src_acc_both}, {"src0, src1, acc.lo all b0", k_all_bank0},
                {"acc b2,b3  sources b0,b1", k_hi_acc}, {"src0 SGPR", k_sgpr_src}, {"src0 = src1 = one SGPR", k_sgpr_both},
                {"src0 = src1 (square)", k_same_src}, {"2 chains of 4 dependent", k_dependent2}};
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-30s %10s %10s %10s   (cycles @2.4 GHz per v_mad_u64_u32 per SIMD)\n", "operand placement", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (auto& e : es) {
    printf("%-30s", e.name);
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;   // 256 threads = 4 waves per block = 1 wave per SIMD per block
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %10.2f", ms * 1e6 * 2.4 / ((double)ITER * wps * 8));
    }
    printf("\n");
  }
  return 0;
}
