// Stage 4: window reduction kernels.  See device_common.hip.h for the pipeline overview.
//
// Slot i of a window holds the bucket of digit magnitude i + 1, so the window value is  sum_i (i + 1) X[i].  With
// the slot index written as  i = hi * 2^L + lo  (L = ceil(lb / 2) column bits, H = lb - L row bits):
//     sum_i (i + 1) X[i] = sum_i X[i]  +  sum_lo lo * C[lo]  +  2^L * sum_hi hi * R[hi]
//     R[hi] = sum_lo X[hi, lo]  (row sums)        C[lo] = sum_hi X[hi, lo]  (column sums)
// Row and column sums are PLAIN sums: every bucket is added exactly twice, in independent chains of at most 7
// additions (sum_groups_kernel, groups of 16 -- per level 4..16 for a lone call --), instead of the running-sum pair "sum += X; sos += sum"
// (two dependent additions per bucket) followed by bit-subset tree sums over the segment sums (another 0.9 per
// bucket) that round 1 used: 2.0 instead of 2.9 full additions per bucket, and chains half as long.  The weights
// lo and hi are applied the same way as before: bit-subset sums over the 2^L column sums and the 2^H row sums
// (reduce_bits_kernel: tiny), whose powers of two the host Horner pass supplies with its doublings.
// Replaces sum_reduction_partial / sum_reduction_final (msm.h.metal:319-562), whose combine step needs a scalar
// multiplication per merge.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// One job of a group-sum launch:  dst[row][q] = sum_{j < group, q*group + j < len} src[row_base(row) + (q*group + j) * elem_stride]
//   row_base(row) = (row / rows_per_window) * window_stride + (row % rows_per_window) * row_stride
// valid (level 1 only; same indexing as src): 0 = the slot was never written (the bucket matrix is not cleared per
// MSM) and counts as the identity.
struct GroupJob {
  const PtI* src;
  const uint32_t* valid;
  PtI* dst;
  size_t window_stride;
  uint32_t total_rows, rows_per_window, row_stride, elem_stride, len, group, out_len;
  uint32_t outputs;   // total_rows * out_len
};

// The row-sum job and the column-sum job of one level in ONE launch (they are independent; a launch costs more
// queueing behind the resident accumulate grid than the additions themselves).  One lane per output; both operands
// of every addition die in it (184 VGPRs, two waves per SIMD: it takes the place of an accumulate wave, it does not fit beside two).
__global__ void __launch_bounds__(64)
sum_groups_kernel(GroupJob j0, GroupJob j1) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = t >= j0.outputs;
  const GroupJob& J = second ? j1 : j0;
  if (second) t -= j0.outputs;
  if (t >= J.outputs) return;
  const uint32_t row = t / J.out_len, q = t - row * J.out_len;
  const size_t base = (size_t)(row / J.rows_per_window) * J.window_stride + (size_t)(row % J.rows_per_window) * J.row_stride;
  const uint32_t first = q * J.group;
  const uint32_t cnt = min(J.group, J.len - first);
  PtI acc = pti_identity();
#pragma unroll 1
  for (uint32_t j = 0; j < cnt; ++j) {
    const size_t at = base + (size_t)(first + j) * J.elem_stride;
    if (J.valid == nullptr || J.valid[at] != 0) acc = pti_add(acc, load_pti(&J.src[at]));
  }
  store_pti(&J.dst[t], acc);
}

// grid = (lb + 1, W), one workgroup (a power of two of threads, 64..512) per sum:
//   k < L      : sum of the column sums C[w][i] with bit k of i set
//   L <= k < lb: sum of the row sums R[w][i] with bit k - L of i set
//   k == lb    : sum of all row sums = the window total
// written in the external Jacobian form to out[w * (lb + 1) + k].
__global__ void __launch_bounds__(512)
reduce_bits_kernel(const PtI* __restrict__ C, const PtI* __restrict__ R, uint32_t L, uint32_t H,
                   Jacobian* __restrict__ out) {
  extern __shared__ uint32_t lds_u32[];
  PtI* sh = reinterpret_cast<PtI*>(lds_u32);
  const uint32_t k = blockIdx.x, w = blockIdx.y, lb = L + H;
  const bool cols = k < L;
  const uint32_t len = cols ? (1u << L) : (1u << H);
  const PtI* Vw = (cols ? C : R) + (size_t)w * len;
  PtI acc = pti_identity();
  if (k == lb) {
#pragma unroll 1
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) acc = pti_add(acc, load_pti(&Vw[i]));
  } else {
    const uint32_t bit = cols ? k : k - L;
    const uint32_t half = len >> 1;
    const uint32_t lowmask = (1u << bit) - 1u;
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < half; j += blockDim.x) {
      const uint32_t i = ((j & ~lowmask) << 1) | (1u << bit) | (j & lowmask);
      acc = pti_add(acc, load_pti(&Vw[i]));
    }
  }
  store_pti(&sh[threadIdx.x], acc);
  __syncthreads();
#pragma unroll 1
  for (uint32_t stride = blockDim.x >> 1; stride >= 1; stride >>= 1) {
    if (threadIdx.x < stride) {
      const PtI a = load_pti(&sh[threadIdx.x]);
      const PtI b2 = load_pti(&sh[threadIdx.x + stride]);
      store_pti(&sh[threadIdx.x], pti_add(a, b2));
    }
    __syncthreads();
  }
  // the host Horner pass works on the external 32-bit-limb form
  if (threadIdx.x == 0) store_jac(&out[(size_t)w * (lb + 1) + k], pti_to_ext(load_pti(&sh[0])));
}

// Threads of one bit-subset sum.  A lone call wants the shortest dependency chain (one summand per thread, then the LDS
// tree).  A pipelined instance (Plan::rb_threads = 64) wants ONE wave per sum: a workgroup of two or more 182-VGPR waves
// needs that many free wave slots on one CU at once while the accumulate grid of the next instance owns the machine --
// with one wave the reduce span of an instance drops from 1.39 to 0.88 ms at the same throughput
// (profiles/r04_reduce_bits_one_wave.txt).
static uint32_t reduce_bits_threads(const Plan& p) {
  if (p.rb_threads) return p.rb_threads;
  const uint32_t lb = p.lb;
  const uint32_t longest = 1u << ((lb + 1) / 2);
  uint32_t t = 64;
  while (t < 512 && t < longest / 2) t <<= 1;
  return t;
}

int reduce_set_attributes(const char** failed) {
  if (hipFuncSetAttribute((const void*)reduce_bits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess) {
    (void)hipGetLastError();
    *failed = "reduce_bits_kernel";
    return 1;
  }
  return 0;
}


// Elements of scratch one family (rows or columns) needs per window: all levels of the group sums.
size_t reduce_scratch_elems(uint32_t lb) {
  const uint32_t L = (lb + 1) / 2, H = lb - L;
  size_t need = 0;
  for (int fam = 0; fam < 2; ++fam) {
    const size_t rows = (size_t)1 << (fam ? L : H);
    uint32_t len = 1u << (fam ? H : L);
    size_t s = 0;
    while (len > 1) {   // the smallest group has the most levels and the most intermediate sums
      len = (len + kReduceGroupMin - 1) / kReduceGroupMin;
      s += rows * len;
    }
    if (s == 0) s = rows;
    need = std::max(need, s);
  }
  return need;
}

// partial[w][0 .. L-1]      bit sums of the column sums (weights 2^k)
// partial[w][L .. L+H-1]    bit sums of the row sums    (weights 2^(L + k))
// partial[w][lb]            sum of all buckets of the window (weight 1)
void launch_reduce(hipStream_t st, const Plan& p, const PtI* buckets, const uint32_t* bucket_size, PtI* S, PtI* T,
                   Jacobian* partial) {
  const uint32_t L = p.red_L, H = p.red_H;
  const uint32_t ncols = 1u << L, nrows = 1u << H;
  const uint32_t min_group = std::min(std::max(p.red_group, kReduceGroupMin), kReduceGroup);
  // family 0: row sums R[w][hi] (scratch S), family 1: column sums C[w][lo] (scratch T)
  GroupJob job[2];
  PtI* next_dst[2] = {S, T};
  for (int fam = 0; fam < 2; ++fam) {
    GroupJob& J = job[fam];
    J.src = buckets;
    J.valid = bucket_size;
    J.window_stride = p.nb;
    J.rows_per_window = fam ? ncols : nrows;
    J.total_rows = p.W * J.rows_per_window;
    J.row_stride = fam ? 1u : ncols;
    J.elem_stride = fam ? ncols : 1u;
    J.len = fam ? nrows : ncols;
  }
  while (job[0].len > 1 || job[1].len > 1) {
    // per level: the smallest group (shortest chains) whose outputs still fit the lanes one launch can have resident;
    // red_group = kReduceGroup (pipelined instances) pins 16
    uint32_t group = min_group;
    while (group < kReduceGroup) {
      size_t outs = 0;
      for (int fam = 0; fam < 2; ++fam)
        if (job[fam].len > 1) outs += (size_t)job[fam].total_rows * ((job[fam].len + group - 1) / group);
      if (outs <= kReduceResidentLanes) break;
      group <<= 1;
    }
    for (int fam = 0; fam < 2; ++fam) {
      GroupJob& J = job[fam];
      if (J.len > 1) {
        J.group = std::min(J.len, group);
        J.out_len = (J.len + J.group - 1) / J.group;
        J.outputs = J.total_rows * J.out_len;
        J.dst = next_dst[fam];
      } else {
        J.outputs = 0;   // this family is done
      }
    }
    const size_t outputs = (size_t)job[0].outputs + job[1].outputs;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)((outputs + 63) / 64)), dim3(64), 0, st, job[0], job[1]);
    for (int fam = 0; fam < 2; ++fam) {   // the next level reads what this one wrote: contiguous [W * rows][out_len]
      GroupJob& J = job[fam];
      if (J.outputs == 0) continue;
      J.src = J.dst;
      J.valid = nullptr;
      J.window_stride = (size_t)J.rows_per_window * J.out_len;
      J.row_stride = J.out_len;
      J.elem_stride = 1;
      J.len = J.out_len;
      next_dst[fam] = J.dst + J.outputs;
    }
  }
  // job[fam].src now points at [W][rows] sums
  hipLaunchKernelGGL(reduce_bits_kernel, dim3(p.lb + 1, p.W), dim3(reduce_bits_threads(p)),
                     reduce_bits_threads(p) * sizeof(PtI), st, job[1].src, job[0].src, L, H, partial);
}

}  // namespace msm_amd
