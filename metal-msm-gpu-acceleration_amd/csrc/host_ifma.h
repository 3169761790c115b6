// Interface of host_ifma.cpp: the AVX-512 IFMA kernels of the product's CPU MSM (eight batched-affine bucket additions
// per vector).  Plain C++; see host_ifma.cpp for the representation.
#pragma once
#include <cstddef>
#include <cstdint>

namespace msm_amd {
namespace ifma {

constexpr int kMaxBatch = 1024;              // additions per shared inversion, at most
constexpr int kMaxRows = kMaxBatch / 8;

// Per-thread scratch of one batch: six arrays of kMaxRows vectors of 5 limbs x 8 lanes.
struct Scratch {
  alignas(64) uint64_t x1[kMaxRows * 40], y1[kMaxRows * 40], x2[kMaxRows * 40], y2[kMaxRows * 40], d[kMaxRows * 40],
      pre[kMaxRows * 40];
  int rows = 0;
  int n_special = 0;            // elements of the batch with x2 = x1 (forward reports them, backward leaves their buckets alone)
  int special[kMaxBatch];
  uint8_t is_special[kMaxBatch];
};

bool available();
// unit-test hook: count field elements of 4 x u64, canonical; op 0: a b / 2^260, 1: a - b, 2: -a (all mod p)
void test_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count);   // CPUID: AVX-512 F / IFMA / DQ / VL / BW
void convert(const uint64_t* in, uint64_t* out, size_t n, int dir);   // dir 0: R (2^256) -> Q (2^260) domain, 1: back
// pt_idx[k]: index of the point in `pts` (8 u64 per point), bit 31 set = add the negative
// returns the number of elements whose point has the x of its bucket (ws.special[]: not added, see host_msm.hip)
int forward(const uint64_t* buckets, const uint32_t* bucket_idx, const uint64_t* pts, const uint32_t* pt_idx, int count,
            Scratch& ws, uint64_t totals[8][4]);
void backward(uint64_t* buckets, const uint32_t* bucket_idx, int count, Scratch& ws, const uint64_t inv[8][4]);

}  // namespace ifma
}  // namespace msm_amd
