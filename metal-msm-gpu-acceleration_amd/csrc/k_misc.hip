// Conversion of projective inputs and the device-side synthetic instance generator.
#include "device_common.hip.h"
#include "launch.h"

namespace msm_amd {

// ark_bn254::G1Projective (x, y, z Montgomery LE, 96 B) -> affine 64 B.  One thread per point with a
// Fermat inversion only when z is neither 0 nor one (the reference converts on the CPU, state.rs:88-109).
__global__ void __launch_bounds__(64)
projective_to_affine_kernel(const Jacobian* __restrict__ in, uint32_t n, Affine* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  store_affine(&out[t], jac_to_affine(load_jac(&in[t])));
}

// scalars_mont: write scalars in Montgomery form (what bn256::Fr / ark Fr hold in memory) or canonical.
__global__ void __launch_bounds__(64)
gen_instance_kernel(uint64_t seed, uint32_t n, int scalars_mont, Affine* __restrict__ bases,
                    u256* __restrict__ scalars) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Affine pt;
  pt.x = u256_zero();
  pt.y = u256_zero();
#pragma unroll 1
  for (uint32_t attempt = 0; attempt < 64; ++attempt) {
    if (gen_point_attempt(seed, t, attempt, pt)) break;
  }
  store_affine(&bases[t], pt);
  u256 k = gen_scalar_canonical(seed, t);
  if (scalars_mont) k = Fr::to_mont(k);
  store_u256(&scalars[t], k);
}


// External affine bases (64 B, 8 x u32 Montgomery R = 2^256) -> packed internal form (64 B, see bn254_ec29.hip.h).  One pass
// per MSM: 2 internal multiplications per point, ~1 % of the accumulation work.
__global__ void __launch_bounds__(128)
convert_bases_kernel(const Affine* __restrict__ in, uint32_t n, AffPacked* __restrict__ out) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  store_affi(&out[t], affi_from_ext(load_affine(&in[t])));
}

#if defined(MSM_AMD_EXPERIMENTS)
// The same into the wide record (one 128-byte line per base: x, y, -y as limbs; bn254_ec29.hip.h AffWide) -- accumulate
// variant 8 of the experiments build.
__global__ void __launch_bounds__(128)
convert_bases_wide_kernel(const Affine* __restrict__ in, uint32_t n, AffWide* __restrict__ out) {
  __builtin_amdgcn_s_setprio(kFrontPriority);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  store_wide(&out[t], affi_widen(affi_from_ext(load_affine(&in[t]))));
}
#endif

// Precomputed window tables: tables[w * n + i] = 2^(c w) P_i for w = 0 .. W-1, in the packed internal
// form, so that window w of scalar i adds into the SAME bucket set as window 0: one set of 2^(c-1) buckets serves
// all windows and c can grow to 18..20 (about 18 % fewer additions at 2^20 points).  One thread per point walks the
// windows: c doublings on the external Jacobian form, one inversion per table entry.  A set-up cost (~0.1 s per
// 2^20 points), paid once per SRS.
__global__ void __launch_bounds__(64)
build_tables_kernel(const Affine* __restrict__ in, uint32_t n, uint32_t c, uint32_t W, AffPacked* __restrict__ tables) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Affine a = load_affine(&in[t]);
  store_affi(&tables[t], affi_from_ext(a));
  Jacobian p = affine_is_identity(a) ? jac_identity() : jac_from_affine(a);
#pragma unroll 1
  for (uint32_t w = 1; w < W; ++w) {
#pragma unroll 1
    for (uint32_t i = 0; i < c; ++i) p = jac_double(p);
    a = jac_to_affine(p);                    // identity stays (0, 0)
    p = affine_is_identity(a) ? jac_identity() : jac_from_affine(a);   // z back to one: cheaper doublings
    store_affi(&tables[(size_t)w * n + t], affi_from_ext(a));
  }
}

// Test aid for the bounded host waits (msm_amd_test_hold): ONE lane that keeps its stream busy until the host sets
// *release or `max_ticks` of the 100 MHz wall clock have passed -- an exit condition the wave reaches whatever the
// host does.
__global__ void __launch_bounds__(64)
hold_kernel(const volatile uint32_t* release, uint64_t max_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();
  while (*release == 0u && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(127);
}

void launch_hold(hipStream_t st, const uint32_t* release, uint64_t max_ticks) {
  hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, st, (const volatile uint32_t*)release, max_ticks);
}

void launch_build_tables(hipStream_t st, const Affine* in, uint32_t n, uint32_t c, uint32_t W, AffPacked* tables) {
  hipLaunchKernelGGL(build_tables_kernel, dim3((n + 63) / 64), dim3(64), 0, st, in, n, c, W, tables);
}

void launch_convert_bases(hipStream_t st, const Affine* in, uint32_t n, AffPacked* out) {
  hipLaunchKernelGGL(convert_bases_kernel, dim3((n + 127) / 128), dim3(128), 0, st, in, n, out);
}

#if defined(MSM_AMD_EXPERIMENTS)
void launch_convert_bases_wide(hipStream_t st, const Affine* in, uint32_t n, AffWide* out) {
  hipLaunchKernelGGL(convert_bases_wide_kernel, dim3((n + 127) / 128), dim3(128), 0, st, in, n, out);
}
#endif

void launch_projective_to_affine(hipStream_t st, const Jacobian* in, uint32_t n, Affine* out) {
  hipLaunchKernelGGL(projective_to_affine_kernel, dim3((n + 63) / 64), dim3(64), 0, st, in, n, out);
}

void launch_gen_instance(hipStream_t st, uint64_t seed, uint32_t n, int scalars_mont, Affine* bases, u256* scalars) {
  hipLaunchKernelGGL(gen_instance_kernel, dim3((n + 63) / 64), dim3(64), 0, st, seed, n, scalars_mont, bases, scalars);
}

}  // namespace msm_amd
