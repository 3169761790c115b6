/* msm_amd.h -- C ABI of the MI355X-native BN254 G1 multi-scalar multiplication (libmsm_amd.so).
 *
 * Drop-in boundary for the Apple-Metal MSM path of ElusAegis/metal-msm-gpu-acceleration (crate
 * `mopro-msm`).  The reference has no C ABI of its own (its callers are generic Rust functions,
 * src/metal/msm.rs); every entry point below cites the reference interface it replaces, and
 * INTEGRATION.md shows the Rust `extern "C"` shim a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types;
 *   - every function returns an int status (0 = OK), never aborts;
 *   - a ctx owns one HIP device, four HIP streams (front end, accumulate, two reduce streams: consecutive
 *     instances overlap their phases) plus a copy stream, and grow-on-demand device workspaces; calls on one ctx are
 *     serialised internally, different ctxs (one per GPU, or several on one GPU) run concurrently -- see
 *     msm_amd_msm_batch_multi;
 *   - all 256-bit values are little-endian (least significant byte first) unless a *_BE32 layout
 *     is named; field coordinates are in Montgomery form with R = 2^256 exactly as halo2curves and
 *     arkworks hold them in memory.
 */
#ifndef MSM_AMD_H
#define MSM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msm_amd_ctx msm_amd_ctx;

/* Status codes; 1..5 mirror MetalError (src/metal/abstraction/errors.rs:4-19). */
enum {
  MSM_AMD_OK = 0,
  MSM_AMD_DEVICE_NOT_FOUND = 1, /* MetalError::DeviceNotFound */
  MSM_AMD_LIBRARY_ERROR = 2,    /* MetalError::LibraryError   (code object missing / not gfx950) */
  MSM_AMD_FUNCTION_ERROR = 3,   /* MetalError::FunctionError  (kernel attribute / symbol) */
  MSM_AMD_PIPELINE_ERROR = 4,   /* MetalError::PipelineError  (launch / runtime failure) */
  MSM_AMD_INPUT_ERROR = 5,      /* MetalError::InputError     (null, n == 0, bad layout/size) */
  /* 6..8 mirror HarnessError of the instance-file harness (src/utils/preprocess.rs:11-21). */
  MSM_AMD_FILE_OPEN_ERROR = 6,        /* HarnessError::FileOpenError        (open/read/write failed) */
  MSM_AMD_DESERIALIZATION_ERROR = 7,  /* HarnessError::DeserializationError (truncated or malformed bincode) */
  MSM_AMD_INVALID_DATA = 8            /* HarnessError::InvalidData          (instance count / size mismatch) */
};

/* Scalar layouts (32 bytes each). */
enum {
  MSM_AMD_SCALAR_MONT_LE = 0,   /* halo2curves bn256::Fr / ark_bn254::Fr in memory: [u64;4] LE, Montgomery
                                   (msm.rs:258-270 reinterprets &[C::Scalar]) */
  MSM_AMD_SCALAR_CANON_LE = 1,  /* canonical integer, little-endian */
  MSM_AMD_SCALAR_CANON_BE32 = 2 /* reference wire layout: 8 x u32, most significant limb first, canonical
                                   (limbs_conversion.rs:116-121, :282-288) */
};

/* Point layouts. */
enum {
  MSM_AMD_POINT_H2C_AFFINE = 0,     /* bn256::G1Affine {x,y}: 64 B, Montgomery LE, identity = (0,0) */
  MSM_AMD_POINT_ARK_PROJECTIVE = 1, /* ark_bn254::G1Projective {x,y,z}: 96 B Jacobian, Montgomery LE
                                       (limbs_conversion.rs:123-130) */
  MSM_AMD_POINT_ARK_AFFINE = 2,     /* ark_bn254::G1Affine {x,y,infinity:bool}: 72 B (limbs_conversion.rs:132-137) */
  MSM_AMD_POINT_JAC_BE32 = 3,       /* reference wire layout: 24 x u32 (x,y,z each MS-limb first), Montgomery */
  MSM_AMD_POINT_PREPARED = 4,       /* device-only: 64-byte records written by msm_amd_bases_upload /
                                       msm_amd_bases_prepare_device (opaque internal form of affine points) */
  MSM_AMD_POINT_TABLES = 5          /* the "points" pointer is a msm_amd_tables handle (precomputed window tables) */
};

/* Per-stage device times of the last MSM on this ctx, milliseconds, from hipEvents on the ctx stream
 * (replaces the log::debug! Instant timers of msm.rs:193-214, 288-328). */
typedef struct msm_amd_timings {
  float convert_ms;     /* input layout conversion (0 when inputs are already native) */
  float digits_ms;      /* prepare_buckets_indices */
  float sort_ms;        /* sort_buckets: hist + prefix + scan + scatter (+ bucket ordering) */
  float accumulate_ms;  /* bucket_wise_accumulation (dominant kernel) */
  float reduce_ms;      /* sum_reduction: segment + tree kernels */
  float final_ms;       /* final_accumulation on the host (wall clock) */
  float total_gpu_ms;   /* SUM of the stage spans above (stages of neighbouring instances overlap on other streams,
                           so this is not a wall interval) */
  uint32_t n;
  uint32_t window_size;
  uint32_t num_windows;
  uint32_t reserved;            /* number of instances the averages were taken over; for ONE large instance that ran
                                   as pipelined point ranges (lone calls from 2^23 device-resident / 2^19 host points,
                                   MSM_AMD_SPLIT): the number of ranges, and the stage fields are SUMS over them */
  float accumulate_kernel_ms;   /* accumulate_kernel alone (events directly around its launch) */
  float reserved2[3];           /* [0] = work items of the last instance's accumulate grid (exact below 2^24)
                                   [1] = 1 if bucket accumulation had not finished yet when the after_sort
                                         callback of msm_amd_gpu_msm_h2c_sync fired, 0 if it had, -1 if no callback ran
                                   [2] = device time (ms) from that callback to the end of bucket accumulation */
} msm_amd_timings;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* setup_metal_state (msm.rs:77-94): pick `device` (ordinal; -1 = current HIP device), create the
 * stream, check the kernels are loadable. */
int msm_amd_init(int device, msm_amd_ctx** out);
/* setup_metal_state_reusable (msm.rs:96-109): process-global cached ctx on the current device. */
int msm_amd_init_reusable(msm_amd_ctx** out);
/* get_global_metal_config (msm.rs:114-119): the cached ctx, MSM_AMD_INPUT_ERROR if never initialised. */
int msm_amd_get_global(msm_amd_ctx** out);
void msm_amd_destroy(msm_amd_ctx* ctx);
const char* msm_amd_strerror(int status);
/* Human-readable detail of the last failure on this ctx (empty string if none). */
const char* msm_amd_last_error(const msm_amd_ctx* ctx);

/* encode_instances' `window_size: Option<u32>` (msm.rs:130-141): 0 = automatic, else 3..17. */
int msm_amd_set_window_size(msm_amd_ctx* ctx, uint32_t window_size);
/* The automatic choice for n points.  Reference policy: 3 if n < 32 else 15 (msm.rs:135-141).  This library: 3 below
 * 32 points, then the window measured fastest on MI355X per size class (5 up to 2^14 points, 15 up
 * to 2^18, 16 at 2^19, 17 beyond) for instances that run PIPELINED (batches, or a call submitted while others are
 * in flight); results never depend on it. */
uint32_t msm_amd_auto_window_size(size_t n);
/* The automatic choice for ONE instance submitted while nothing else is in flight (a blocking gpu_msm_h2c call):
 * such a call is a chain of launches and dependent additions, not a throughput problem, and wider windows with a
 * full top digit are faster (5 up to 2^6 points, 8 up to 2^12, 15 up to 2^18, 17 beyond). */
uint32_t msm_amd_auto_window_size_lone(size_t n);

/* ---- whole-MSM entry points: host buffers ---------------------------------------------------- */
/* gpu_msm_h2c::<G1Affine, .., Fr>(scalars, points) -> G1 (msm.rs:352-364).
 * scalars: n x 32 B MSM_AMD_SCALAR_MONT_LE; points: n x 64 B MSM_AMD_POINT_H2C_AFFINE;
 * out: 96 B Jacobian (x, y, z) Montgomery LE, normalised to z = R mod p, or z = 0 for the identity:
 * memcpy-compatible with bn256::G1 / G1Projective.
 * A call of 2^19 points or more with nothing else in flight is executed as a pipelined batch of 2 / 4 / 8 point
 * ranges (upload and sort of one range under the bucket accumulation of the previous one) whose results are added
 * on the host -- same result, 17-38 % less wall time; every single-instance entry point does this (device-resident
 * inputs from 2^23 points). */
int msm_amd_gpu_msm_h2c(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, void* out96);
/* gpu_msm_h2c_sync(scalars, points, sync_pair: Arc<(Mutex<bool>, Condvar)>) (msm.rs:237-349): the same MSM, and
 * `after_sort(user)` is called once, on the calling thread, as soon as the sort stage of this MSM has finished on
 * the GPU -- the point where the reference sets the flag and notifies the condvar (msm.rs:306-312) so that a
 * hybrid caller may start its CPU half (gpu_with_cpu, msm.rs:403-415).  after_sort may be NULL.  The callback must
 * not call into the same ctx. */
typedef void (*msm_amd_after_sort_fn)(void* user);
int msm_amd_gpu_msm_h2c_sync(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n,
                             msm_amd_after_sort_fn after_sort, void* user, void* out96);
/* metal_msm::<ArkG, ArkFr>(points, scalars, &mut config) -> Result<ArkG, MetalError> (msm.rs:220-234).
 * points: n x 96 B MSM_AMD_POINT_ARK_PROJECTIVE; scalars: n x 32 B MSM_AMD_SCALAR_MONT_LE. */
int msm_amd_metal_msm_ark(msm_amd_ctx* ctx, const void* points, const void* scalars, size_t n, void* out96);
/* Generic form of the two above with explicit layouts. */
int msm_amd_msm(msm_amd_ctx* ctx, int scalar_layout, int point_layout, const void* scalars, const void* points,
                size_t n, void* out96);
/* The instance loop of gpu_profiler / benches (gpu_profiler.rs:104-106, msm_benchmark.rs:29-34):
 * n_inst independent MSMs; out = n_inst x 96 B.  Instance i + 1 is uploaded while instance i computes.  With
 * MSM_AMD_POINT_PREPARED / MSM_AMD_POINT_TABLES the points array holds device pointers / table handles (see
 * below) and only the scalars are uploaded. */
int msm_amd_msm_batch(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                      const void* const* scalars, const void* const* points, const size_t* n, void* out);

/* Page-lock a caller buffer so that the host-buffer entry points above upload it by DMA at PCIe rate and without
 * blocking the calling thread (pageable memory is staged by the runtime at about half that rate).  Meant for
 * long-lived inputs such as the bases of an SRS, which the reference re-uploads on every call (msm.rs:152-153).
 * The memory must stay valid and in place until msm_amd_host_unregister (or msm_amd_destroy). */
int msm_amd_host_register(msm_amd_ctx* ctx, const void* ptr, size_t bytes);
int msm_amd_host_unregister(msm_amd_ctx* ctx, const void* ptr);

/* Resident-bases speed for drop-in callers, without an API change (opt-in).  The reference's callers hand over the
 * SAME bases slice on every call (benches/msm_benchmark.rs:116-121) and the reference re-uploads and re-converts it
 * every time (msm.rs:152-153).  With a cache budget of max_bytes > 0 (or MSM_AMD_BASES_CACHE_MB at msm_amd_init) the
 * host-slice entry points -- msm_amd_gpu_msm_h2c, msm_amd_msm, msm_amd_msm_batch, msm_amd_metal_msm_ark,
 * msm_amd_msm_best -- keep the converted device copy of every points array they see (64 B per point, least recently
 * used arrays make way) and on the next call with the same (pointer, n, layout) upload the scalars only.
 * Contract: an array handed over at an unchanged address holds unchanged bases.  As a safety net every hit re-hashes
 * ~2 k sampled records of the caller's memory (record i belongs to phase i mod (n / 1024); phase 0 and one rotating
 * phase are checked per call): a changed array is detected at once if the change touches phase 0, within n / 1024
 * calls otherwise, and is then re-uploaded.  max_bytes = 0 switches the cache off and frees it.
 * stats: [0] hits, [1] misses (entries filled), [2] invalidations (checksum mismatch), [3] bytes held, [4] entries. */
int msm_amd_set_bases_cache(msm_amd_ctx* ctx, size_t max_bytes);
int msm_amd_bases_cache_stats(msm_amd_ctx* ctx, uint64_t stats[5]);
/* A caller that DOES change bases in place says so: the entries of the array at host_points (NULL: all entries) are
 * dropped, the next call uploads again.  This is the contract's other half; the sampling above is only a net. */
int msm_amd_bases_cache_invalidate(msm_amd_ctx* ctx, const void* host_points);
/* full != 0 (or MSM_AMD_BASES_CACHE_VERIFY=full at msm_amd_init): every hit re-hashes EVERY record of the caller's
 * array (64 contiguous slices on four host threads, ~1.5 ms per 2^20 points) before the cached copy is used -- no
 * stale window at all, at about the cost of the upload the cache saves (the conversion is still saved).  0: sampling. */
int msm_amd_set_bases_cache_verify(msm_amd_ctx* ctx, int full);

/* ---- hybrid front-end ----------------------------------------------------------------------- */
/* msm_best::<G1Affine, ..>(scalars, points) -> G1 (msm.rs:424-445): filter_zeros (drop zero scalars when at
 * least 30 % of them are zero, msm.rs:448-507, done here by a device compaction) and then the MSM.  The
 * reference sends n < 2^17 to halo2curves on the CPU because its Metal path is slower there (msm.rs:440-444);
 * the same dispatch exists here with the threshold measured on MI355X: below msm_amd_cpu_dispatch_below()
 * points one blocking GPU call costs more than the product's host bucket method (host_msm, the CPU half of
 * gpu_with_cpu), so those sizes are computed on the host -- an explicit size dispatch as in the reference,
 * never a fallback: without a GPU this entry point fails like every other.  h2c layouts. */
int msm_amd_msm_best(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, void* out96);
/* msm_best's size threshold (the reference's is 2^17, msm.rs:440): sizes below it go to the host bucket method. */
size_t msm_amd_cpu_dispatch_below(void);
/* gpu_with_cpu (msm.rs:366-421): the first split_at points go to the GPU, the rest to a multi-threaded host
 * bucket method (cpu_threads <= 0: all hardware threads); the two results are added.  h2c layouts. */
int msm_amd_gpu_with_cpu(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, size_t split_at,
                         int cpu_threads, void* out96);
/* The reference's split policy (msm.rs:377-383): n/3 below 2^18, n/2 below 2^20, else 2n/3 go to the GPU.
 * On MI355X the throughput-optimal split is split_at = n; the policy is kept for parity. */
size_t msm_amd_reference_split(size_t n);
/* The split measured on MI355X: the whole instance goes to the GPU (split_at = n) unless it is smaller than
 * msm_amd_cpu_dispatch_below() -- a CPU share only lengthens the call on this hardware. */
size_t msm_amd_tuned_split(size_t n);
/* The CPU MSM of the library by itself (no ctx, no GPU): what `gpu_profiler <log> <n> cpu` runs where the reference
 * runs halo2curves::msm::msm_best (gpu_profiler.rs:157-159).  Multi-threaded signed-digit Pippenger with
 * batched-affine bucket additions on 4 x 64-bit limbs; h2c affine points, scalars MONT_LE or CANON_LE;
 * threads <= 0: msm_amd_host_threads().  out96 as every other entry point (normalised Jacobian). */
int msm_amd_host_msm(int scalar_layout, int point_layout, const void* scalars, const void* points, size_t n, int threads,
                     void* out96);
/* CPUs the process may really use: affinity mask capped by the cgroup CPU quota. */
int msm_amd_host_threads(void);

/* ---- whole-MSM entry points: inputs already resident in device memory ------------------------ */
/* Same as msm_amd_msm / msm_amd_msm_batch but scalars/points are device pointers on ctx's device
 * (the reference re-uploads and re-converts per call, msm.rs:152-153; an SRS is uploaded once here). */
int msm_amd_msm_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, const void* d_scalars,
                       const void* d_points, size_t n, void* out96_host);
int msm_amd_msm_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                             const void* const* d_scalars, const void* const* d_points, const size_t* n,
                             void* out_host);

/* Pipelined form of msm_amd_msm_batch_device: submit enqueues all GPU work of the batch and returns a ticket;
 * wait finishes it (host Horner pass) and fills out_host (n_inst x 96 B, must stay valid until then).  Up to 4
 * batches may be in flight, so the front end of batch k+1 runs under the accumulation of batch k and the host
 * work of batch k under the GPU work of batch k+1.  The d_scalars / d_points / n arrays are read at submit. */
int msm_amd_submit_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                                const void* const* d_scalars, const void* const* d_points, const size_t* n,
                                void* out_host, int* ticket);
int msm_amd_wait_batch(msm_amd_ctx* ctx, int ticket);

/* Upper bound of every host wait for the GPU inside the library, in milliseconds (default 60 000, or the environment
 * variable MSM_AMD_WAIT_TIMEOUT_MS at msm_amd_init; 0 = wait without bound).  The reference's gpu_msm_h2c_sync is one
 * blocking call that always returns (msm.rs:237-349); so is every call here: a wait that reaches the bound returns
 * MSM_AMD_PIPELINE_ERROR and msm_amd_last_error names the stage event and the instance the device did not reach.
 * The work stays in flight: a ticket of msm_amd_submit_batch_device stays valid and may be waited for again; after a
 * blocking entry point timed out, the next call first checks whether the device has caught up (and fails the same
 * way if not); msm_amd_synchronize waits once more; msm_amd_destroy gives the device resources up rather than
 * freeing memory under running kernels.
 * The same bound covers memory management: the library never calls hipMalloc / hipHostMalloc / hipFree / hipHostFree
 * while the ctx has work in flight (they may wait for the device, hipFree always does).  A call that has to GROW a
 * workspace, a page-locked result slot or the staging ring first waits -- bounded -- for the ctx's streams to run
 * empty and returns MSM_AMD_PIPELINE_ERROR ("device busy ... cannot grow") if they do not; buffers that are outgrown
 * or dropped (bases cache entries, msm_amd_device_free on a busy ctx) are released when the ctx is idle again.
 * After a timeout, host buffers the caller registered with msm_amd_host_register may still be read by DMA: keep
 * them alive until msm_amd_synchronize has returned MSM_AMD_OK. */
int msm_amd_set_wait_timeout_ms(msm_amd_ctx* ctx, uint32_t timeout_ms);

/* ---- several GPUs ---------------------------------------------------------
 * The instance loop of the reference (gpu_profiler.rs:101-106, benches/msm_benchmark.rs:29-34) sharded over
 * several ctxs: instance j runs on ctxs[j mod n_ctx], one host thread per ctx (pinned to the CPUs local to the
 * ctx's GPU when sysfs exposes them), no data-path collective; results land at out + 96 j.  The ctxs may sit on
 * different GPUs (the intended use) or on one (they then share it).  _multi takes host buffers like
 * msm_amd_msm_batch; _multi_device takes device buffers, those of instance j on the device of ctxs[j mod n_ctx].
 * On failure the first failing ctx's status is returned and msm_amd_last_error(that ctx) has the detail. */
int msm_amd_msm_batch_multi(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout, size_t n_inst,
                            const void* const* scalars, const void* const* points, const size_t* n, void* out);
int msm_amd_msm_batch_multi_device(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                                   size_t n_inst, const void* const* d_scalars, const void* const* d_points,
                                   const size_t* n, void* out_host);
/* Pipelined form of msm_amd_msm_batch_multi_device (the multi-ctx twin of msm_amd_submit_batch_device /
 * msm_amd_wait_batch): submit enqueues every ctx's share and returns a ticket, wait finishes all shares and writes
 * out_host + 96 j (out_host must stay valid until then).  Up to 4 batches may be in flight per ctx, so a caller
 * looping over batches -- benches/msm_benchmark.rs:29-34 -- keeps every GPU busy across calls.  wait frees the ticket
 * on success; after a bounded-wait timeout (MSM_AMD_PIPELINE_ERROR, work still in flight) the ticket stays valid. */
typedef struct msm_amd_multi_ticket msm_amd_multi_ticket;
int msm_amd_submit_batch_multi_device(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                                      size_t n_inst, const void* const* d_scalars, const void* const* d_points,
                                      const size_t* n, void* out_host, msm_amd_multi_ticket** ticket);
int msm_amd_wait_batch_multi(msm_amd_multi_ticket* ticket);
/* ONE instance of n points over several ctxs, split by point range ("single huge instance"):
 * ctx g uploads and runs the MSM of points [begin_g, end_g) (msm_amd_shard_range), the partial results are added with
 * msm_amd_sum_points -- the algebra of the reference's GPU + CPU split, src/metal/msm.rs:385-419.  Host buffers in a
 * host layout (not MSM_AMD_POINT_PREPARED / _TABLES, which belong to one ctx); out96 as msm_amd_msm. */
int msm_amd_msm_range_multi(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                            const void* scalars, const void* points, size_t n, void* out96);
void msm_amd_shard_range(size_t n, size_t n_ctx, size_t k, size_t* begin, size_t* end);
/* Bytes per element of a layout (0 = unknown layout). */
size_t msm_amd_scalar_bytes(int scalar_layout);
size_t msm_amd_point_bytes(int point_layout);
/* The sharding arithmetic: owner of instance j, and how many instances ctx k of n_ctx gets. */
size_t msm_amd_shard_owner(size_t instance, size_t n_ctx);
size_t msm_amd_shard_count(size_t n_inst, size_t n_ctx, size_t k);
/* HIP device ordinal of a ctx. */
int msm_amd_ctx_device(const msm_amd_ctx* ctx);
/* Restrict the calling thread to the CPUs local to `device` (sysfs local_cpulist of its PCI function, intersected
 * with the thread's current mask).  0 = pinned, 1 = no NUMA information or fewer than 8 CPUs to intersect (not an
 * error; the mask is left alone). */
int msm_amd_pin_thread_to_device(int device);

/* RCCL all-gather of per-rank result blocks (one communicator per listed device, created in this process with
 * ncclCommInitAll; librccl is loaded on first use, not linked).  send_host[k] = rank k's bytes_per_rank bytes
 * (its ceil(I / G) x 96 B of results), recv_host[k] = n_devices x bytes_per_rank bytes, every rank's block in rank
 * order, moved through rank k's GPU over xGMI.  A deployment with one PROCESS per GPU does the same with its own
 * communicator (bench.py: torch.distributed, backend nccl = RCCL). */
typedef struct msm_amd_gather msm_amd_gather;
int msm_amd_gather_init(const int* devices, int n_devices, msm_amd_gather** out);
int msm_amd_gather_size(const msm_amd_gather* g);
int msm_amd_gather_all(msm_amd_gather* g, const void* const* send_host, size_t bytes_per_rank, void* const* recv_host);
const char* msm_amd_gather_last_error(const msm_amd_gather* g);
void msm_amd_gather_destroy(msm_amd_gather* g);

/* ---- persistent bases -----------------------------------------------------------------------
 * The reference converts and re-uploads the bases on every call (msm.rs:152-153); provers reuse one SRS for
 * many MSMs.  These calls convert a point array once into the library's internal 64-byte form, resident on the
 * device; pass the result as d_points with MSM_AMD_POINT_PREPARED to the *_device entry points, or use
 * msm_amd_msm_prepared with host scalars.  Prepared arrays are freed with msm_amd_device_free. */
int msm_amd_bases_upload(msm_amd_ctx* ctx, int point_layout, const void* points, size_t n, void** d_prepared);
int msm_amd_bases_prepare_device(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n,
                                 void* d_prepared /* n x 64 bytes */);
int msm_amd_msm_prepared(msm_amd_ctx* ctx, int scalar_layout, const void* scalars, const void* d_prepared, size_t n,
                         void* out96);

/* ---- partial results ------------------------------------------------------------------------
 * Host-side sum of `count` results (96 B Jacobian Montgomery LE each, as every entry point above returns
 * them), normalised like them.  It is the final addition of gpu_with_cpu (msm.rs:418-419) made available to
 * callers that split ONE instance by point range across several GPUs: every rank runs the MSM of its range,
 * the 96-byte partial results are all-gathered, and every rank adds them. */
int msm_amd_sum_points(const void* points96, size_t count, void* out96);

/* ---- precomputed window tables (fixed bases) ---------------------------------------------------
 * Beyond the reference: for a fixed set of bases (an SRS) the library can store 2^(c w) P_i for every window w
 * (W x n x 64 bytes; 0.9 GB for 2^20 points -- HBM is 288 GB).  All windows then share ONE set of 2^(c-1)
 * buckets, so the window grows to c = log2(n) - 1 and an MSM needs ~18 % fewer point additions at 2^20 points,
 * with the same results bit for bit.  Build once (about 0.1 s per 2^20 points), then pass the handle as the points
 * pointer with MSM_AMD_POINT_TABLES to the *_device entry points (n must equal the table's n), or call
 * msm_amd_msm_tables with host scalars.  window_size 0 = automatic. */
typedef struct msm_amd_tables msm_amd_tables;
int msm_amd_tables_build(msm_amd_ctx* ctx, int point_layout, const void* points, size_t n, uint32_t window_size,
                         msm_amd_tables** out);
int msm_amd_tables_build_device(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n,
                                uint32_t window_size, msm_amd_tables** out);
int msm_amd_tables_info(msm_amd_ctx* ctx, const msm_amd_tables* tables, size_t* n, uint32_t* window_size,
                        uint32_t* num_windows, size_t* device_bytes);
int msm_amd_tables_free(msm_amd_ctx* ctx, msm_amd_tables* tables);
int msm_amd_msm_tables(msm_amd_ctx* ctx, const msm_amd_tables* tables, int scalar_layout, const void* scalars,
                       void* out96);

/* ---- device memory helpers (so callers without a HIP binding can stage data) ------------------ */
int msm_amd_device_alloc(msm_amd_ctx* ctx, size_t bytes, void** d_ptr);
int msm_amd_device_free(msm_amd_ctx* ctx, void* d_ptr);
int msm_amd_copy_to_device(msm_amd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int msm_amd_copy_to_host(msm_amd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* Raw hipStream_t of the ctx (for event/stream interop from torch or HIP callers). */
void* msm_amd_stream(msm_amd_ctx* ctx);
int msm_amd_synchronize(msm_amd_ctx* ctx);

/* Deterministic synthetic instance on the device (role of preprocess.rs:113-138): n uniform random
 * G1 points (64 B affine, Montgomery) and n uniform scalars (Montgomery if scalars_mont else canonical LE). */
int msm_amd_generate_instance(msm_amd_ctx* ctx, uint64_t seed, size_t n, int scalars_mont, void* d_points,
                              void* d_scalars);

/* Unit-test hook of the CPU MSM's AVX-512 IFMA arithmetic (csrc/host_ifma.cpp): count canonical field elements of
 * 4 x u64 LE each.  op 0: a b / 2^260 mod p; 1: a - b mod p; 2: -a mod p (a != 0); 3: a 2^4 mod p (R -> Q domain);
 * 4: a / 2^4 mod p (Q -> R).  MSM_AMD_FUNCTION_ERROR on a host without IFMA. */
int msm_amd_test_op_ifma(int op, const void* a, const void* b, void* out, size_t count);
/* The same instance generated on the host (identical bytes; no ctx, no GPU): inputs of `gpu_profiler ... cpu`. */
int msm_amd_generate_instance_host(uint64_t seed, size_t n, int scalars_mont, void* points, void* scalars, int threads);

/* ---- instance files (src/utils/preprocess.rs:30-111, 143-212) -------------------------------------
 * The reference caches benchmark inputs as bincode 1.3 `Vec<(Vec<Vec<u32>>, Vec<Vec<u32>>)>`:
 *   u64 n_instances; per instance { u64 n_points; n_points x { u64 24; u32 x 24 };
 *                                   u64 n_scalars; n_scalars x { u64 8; u32 x 8 } }     (all little-endian)
 * with points in MSM_AMD_POINT_JAC_BE32 and scalars in MSM_AMD_SCALAR_CANON_BE32, under the name
 * msm_{log_size}x{num_instances}.bin.  These host-only functions read and write that format, so that
 * files interchange with the reference's ~/.msm_gpu_acceleration/msm_vecs cache.  No ctx and no GPU needed. */
typedef struct msm_amd_instance_file msm_amd_instance_file;

/* save_msm_instances (preprocess.rs:84-97): points[j] = n[j] x 96 B (JAC_BE32), scalars[j] = n[j] x 32 B
 * (CANON_BE32). */
int msm_amd_instances_save(const char* path, size_t n_inst, const size_t* n, const void* const* points,
                           const void* const* scalars);
/* load_msm_instances (preprocess.rs:99-111), streaming: open scans the record structure only. */
int msm_amd_instances_open(const char* path, msm_amd_instance_file** out);
size_t msm_amd_instances_count(const msm_amd_instance_file* f);
size_t msm_amd_instances_size(const msm_amd_instance_file* f, size_t j);   /* points (= scalars) of instance j */
int msm_amd_instances_read(msm_amd_instance_file* f, size_t j, void* points_out, void* scalars_out);
void msm_amd_instances_close(msm_amd_instance_file* f);
/* msm_{log}x{n}.bin under dir, or under $HOME/.msm_gpu_acceleration/msm_vecs when dir is NULL
 * (preprocess.rs:165, 204-212).  Returns the length written (without the NUL), 0 if buf is too small. */
size_t msm_amd_instances_default_path(const char* dir, uint32_t log_size, uint32_t num_instances, char* buf,
                                      size_t buf_len);
/* Host layout conversion between a caller layout and the wire layout (role of the ToLimbs / FromLimbs
 * impls, limbs_conversion.rs:87-195, 282-389).  to_wire accepts every scalar and point layout; an affine
 * identity becomes z = 0.  from_wire produces MSM_AMD_SCALAR_{MONT_LE,CANON_LE} and
 * MSM_AMD_POINT_ARK_PROJECTIVE (a limb reorder: coordinates are not normalised). */
int msm_amd_to_wire(int scalar_layout, int point_layout, const void* scalars, const void* points, size_t n,
                    void* scalars_be32_out, void* points_be32_out);
int msm_amd_from_wire(int scalar_layout, int point_layout, const void* scalars_be32, const void* points_be32,
                      size_t n, void* scalars_out, void* points_out);

/* ---- per-stage entry points in the reference's wire layout (host buffers) ---------------------
 * These are the `pub` stage functions the reference's own stage tests drive through
 * create_test_instance (sort_buckets.rs:38-69, bucket_wise_accumulation.rs:154-224,
 * sum_reduction.rs:210-258). u32 limbs are most-significant-first (MSM_AMD_*_BE32). */
/* prepare_buckets_indices (prepare_buckets_indices.rs:15-38 + msm.h.metal:17-59): scalars n x 8 u32
 * canonical BE32 -> pairs n*W x 2 u32 at [t*W + i] = (i*(2^c-1) + m - 1, t) or (0xFFFFFFFF,0xFFFFFFFF). */
int msm_amd_prepare_buckets_indices(msm_amd_ctx* ctx, const uint32_t* scalars_be32, size_t n, uint32_t window_size,
                                    uint32_t num_windows, uint32_t* pairs_out);
/* sort_buckets_indices (sort_buckets.rs:15-34): sort n_pairs (u32,u32) pairs by .0 ascending, in place. */
int msm_amd_sort_buckets_indices(msm_amd_ctx* ctx, uint32_t* pairs, size_t n_pairs);
/* The same sort on a device-resident buffer, in place (the measured form of the stage: what
 * benches/sort_buckets_indices_benchmark.rs:10-32 times around the reference's CPU sort).  Only the low
 * key_bits bits of the key are sorted on (32 = full keys incl. the 0xFFFFFFFF sentinels; 8 bits per pass).
 * kernel_ms (optional) receives the device time of the sort. */
int msm_amd_sort_pairs_device(msm_amd_ctx* ctx, void* d_pairs, size_t n_pairs, uint32_t key_bits, float* kernel_ms);
/* bucket_wise_accumulation (bucket_wise_accumulation.rs:26-107): pairs sorted by bucket; points
 * n_points x 24 u32 Jacobian BE32; buckets_out total_buckets x 24 u32 (untouched buckets = all zero,
 * i.e. z = 0, as Metal's zero-filled buffers give the reference). */
int msm_amd_bucket_wise_accumulation(msm_amd_ctx* ctx, const uint32_t* sorted_pairs, size_t n_pairs,
                                     const uint32_t* points_be32, size_t n_points, uint32_t total_buckets,
                                     uint32_t* buckets_out);
/* sum_reduction (sum_reduction.rs:161-181): res[j] = sum_b (b+1) * buckets[j*buckets_size + b]. */
int msm_amd_sum_reduction(msm_amd_ctx* ctx, const uint32_t* buckets_be32, uint32_t buckets_size,
                          uint32_t num_windows, uint32_t* res_out);
/* final_accumulation (final_accumulation.rs:5-40): Horner over the window sums, on the host. */
int msm_amd_final_accumulation(const uint32_t* res_be32, uint32_t num_windows, uint32_t window_size,
                               uint32_t* point_out);

/* ---- single-op test kernels (role of the kernels in shader/tests/ and curves/bn254.h.metal) ------------
 * Batched: `count` independent operations, one lane each.  Operands are 8 x u32 BE32 limbs per 256-bit
 * value; points are 24 x u32. */
enum {
  MSM_AMD_OP_UINT_ADD = 0,  /* test_uint_add   a + b mod 2^256 */
  MSM_AMD_OP_UINT_SUB = 1,  /* test_uint_sub */
  MSM_AMD_OP_UINT_PROD = 2, /* test_uint_prod  a * b[low 32 bits] mod 2^256 */
  MSM_AMD_OP_UINT_SHL = 3,  /* test_uint_shl   a << (b mod 256) */
  MSM_AMD_OP_UINT_SHR = 4,  /* test_uint_shr */
  MSM_AMD_OP_FP_ADD = 5,    /* fp_bn254_add (Montgomery residues in and out) */
  MSM_AMD_OP_FP_SUB = 6,
  MSM_AMD_OP_FP_MUL = 7,
  MSM_AMD_OP_FP_NEG = 8,
  MSM_AMD_OP_FP_POW = 9,    /* a ^ (b low 32 bits) */
  MSM_AMD_OP_EC_ADD = 10,   /* bn254_add: Jacobian + Jacobian; a, b, out are 24 limbs */
  MSM_AMD_OP_EC_MUL = 11,   /* bn254_scalar_mul: a = point (24 limbs), b = scalar (8 limbs, canonical) */
  MSM_AMD_OP_EC_MADD = 12,  /* Jacobian a + affine b (b given as 24 limbs with z = one or z = 0) */
  MSM_AMD_OP_EC_DBL = 13,   /* 2 * a */
  /* ops of the 29-bit-limb internal representation the hot kernels compute in (csrc/bn254_fq29.hip.h);
   * operands and results cross the boundary in the same external form as above */
  MSM_AMD_OP_FP29_MUL = 14,
  MSM_AMD_OP_FP29_SQR = 15,
  MSM_AMD_OP_FP29_SUB_K4E30 = 16,  /* a - b through the lifted constant 4p */
  MSM_AMD_OP_FP29_SUB_K8E30 = 17,
  MSM_AMD_OP_FP29_SUB_K8E31 = 18,  /* a - 3b (lazy three-term subtrahend) */
  MSM_AMD_OP_FP29_SUB_K16E30 = 19,
  MSM_AMD_OP_FP29_SUB_K16E31 = 20, /* a - 3b */
  MSM_AMD_OP_FP29_ROUNDTRIP = 21,  /* external -> internal -> external */
  MSM_AMD_OP_EC29_MADD = 22,       /* Jacobian a + affine b on internal limbs */
  MSM_AMD_OP_EC29_ADD = 23,        /* Jacobian a + Jacobian b on internal limbs */
  MSM_AMD_OP_EC29_MADD_CHAIN = 24, /* a + 64 b: 64 chained mixed additions kept in the lazy internal form */
  MSM_AMD_OP_EC29_ADD_CHAIN = 25,  /* a + 16 b: 16 chained full additions */
  MSM_AMD_OP_EC29_MMADD = 26,      /* -a - 4b: affine + affine (4M + 2S start of a work item) on lazily negated
                                      operands, then three mixed additions; a, b finite with z = one */
  /* host only (msm_amd_test_op_host; msm_amd_test_op rejects them): the 4 x 64-bit arithmetic of the CPU tail of
     every MSM -- window Horner pass and normalisation (csrc/host_fq64.h) */
  MSM_AMD_OP_H64_FP_MUL = 27,
  MSM_AMD_OP_H64_FP_ADD = 28,
  MSM_AMD_OP_H64_FP_SUB = 29,
  MSM_AMD_OP_H64_EC_ADD = 30,      /* Jacobian + Jacobian, 24 limbs each */
  MSM_AMD_OP_H64_EC_DBL = 31,
  /* field ops on internal limbs again (device and host): build options of the multiplication that were measured and
     not shipped -- one Karatsuba level, lockstep product-scanning chains; operands / results as for op 14 */
  MSM_AMD_OP_FP29_MUL_KARATSUBA = 32, /* a * b */
  MSM_AMD_OP_FP29_LOCKSTEP_PAIR = 33, /* a * b + b * b        (two products side by side) */
  MSM_AMD_OP_FP29_LOCKSTEP_MIX = 34,  /* 2 a b + 2 a^2 + b^2  (double product next to a product, squaring pair) */
  MSM_AMD_OP_FP29_LOCKSTEP_TRIPLE = 35, /* a b + a^2 + b^2    (three products side by side) */
  MSM_AMD_OP_FP29_MUL2_KARATSUBA = 36, /* 2 a b               (schoolbook + Karatsuba product, one reduction) */
  /* host only again: the inversion of the CPU tail (normalisation of every result, batched-affine CPU MSM) */
  MSM_AMD_OP_H64_FP_INV = 37,        /* a^-1 by the binary GCD with 31-step rounds; 0 -> 0; b ignored */
  MSM_AMD_OP_H64_FP_INV_FERMAT = 38  /* a^(p-2), its checker */
};
int msm_amd_test_op(msm_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t count);
/* The same operation bodies executed on the host CPU (no GPU needed): host-logic tests. */
int msm_amd_test_op_host(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t count);

/* Test aid for the bounded waits: keeps the ctx's main stream busy for at most max_ms (<= 5000) or until
 * msm_amd_test_release(handle).  The kernel carries its own time limit. */
int msm_amd_test_hold(msm_amd_ctx* ctx, uint32_t max_ms, void** handle);
int msm_amd_test_release(msm_amd_ctx* ctx, void* handle);

/* ---- introspection --------------------------------------------------------------------------- */
int msm_amd_last_timings(const msm_amd_ctx* ctx, msm_amd_timings* out);
/* Algorithmic HBM bytes of one MSM: whole pipeline and accumulation only. */
uint64_t msm_amd_algorithmic_bytes(size_t n, uint32_t window_size, int accumulate_only);
const char* msm_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MSM_AMD_H */
