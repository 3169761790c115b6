// Host driver + C ABI of libmsm_amd.so (see include/msm_amd.h).
//
// Replaces the reference's L2-L4 layers for the MSM path: MetalState (src/metal/abstraction/state.rs),
// MetalMsmConfig/Instance + encode_instances + exec_metal_commands + gpu_msm_h2c_sync
// (src/metal/msm.rs:28-349) and the stage drivers (src/metal/msm/*.rs).  The stages of one MSM are enqueued on
// FOUR HIP streams (front end / accumulate / two alternating reduce streams, ordered by events) with no host
// synchronisation in between (the reference blocks after every stage: prepare_buckets_indices.rs:35-36,
// bucket_wise_accumulation.rs:104-105, sum_reduction.rs:80-81); the only device->host hand-off is the
// (c-2)*W partial window points for the host Horner pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msm_amd.h"
#include "device_common.hip.h"
#include "launch.h"
#include "host_fq64.h"
#include "test_ops.hip.h"

using namespace msm_amd;

namespace {

constexpr uint32_t kModulusBits = 254;   // limbs_conversion.rs:172, :344
constexpr uint32_t kMinWindow = 3, kMaxWindow = 17;   // u16 digits up to 15, u32 digits for 16 and 17
constexpr uint32_t kDefaultBallot = 1;                // sort ranking (Plan::ballot), chosen by profiles/r02_sort_ranking_ab.txt
constexpr size_t kCpuDispatchBelow = 17;              // msm_best: see cpu_dispatch_below()

struct DeviceBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// One hipEventRecord costs ~6 us of stream idle time on this stack (profiles/r02_lone_call_2p18_timeline.txt: 5.7 us
// between two kernels with one event between them, none without), so the accumulate kernel is bracketed by ONE pair
// of events that serves as stage span, kernel span (the roofline figure of bench.py) and start of the reduce span.
enum {
  EV_START = 0, EV_CONVERT, EV_DIGITS, EV_SORT, EV_ACC_S, EV_ACC, EV_REDUCE, EV_COUNT,
  EV_ACC_K0 = EV_ACC_S, EV_ACC_K1 = EV_ACC, EV_RED_S = EV_ACC
};

struct InstanceSlot {
  hipEvent_t ev[EV_COUNT];
  Jacobian* h_partial = nullptr;   // pinned
  size_t h_partial_cap = 0;
  bool has_events = false;
};

}  // namespace

// Device buffers of one in-flight MSM.  A ctx owns kWorkspaces (4) of them and uses them round-robin: the window
// reduction of instance i runs on a side stream while the main stream accumulates instance i+1 and the front
// stream already sorts instance i+2, each in its own workspace.  The reduction is latency-bound (about 25 dependent point additions on a few hundred
// workgroups) and so is the sort front-end; running them side by side hides the shorter of the two.  (Running
// whole instances on parallel streams was tried and gained little: a resident accumulate grid keeps the
// 1024-thread sort workgroups of the other stream from being placed at all.)
struct Workspace {
  DeviceBuf digits, coarse_cnt, region_start, tmp_idx, tmp_fine, tmp_idx2, tmp_fine2, mid_cnt, region_start2, bsize, bstart, istart, win_items, size_bins, sorted,
      order, multi_list, redo_list, counters, bases29, buckets, item_partials, S, T, partial, conv_scalars, conv_points,
      conv_tmp;
  hipEvent_t front_done = nullptr;    // front stream: sorted indices / work items of this workspace are ready
  hipEvent_t acc_done = nullptr;      // main stream: buckets of this workspace are complete (incl. combine)
  hipEvent_t reduce_done = nullptr;   // reduce stream: buckets / partial of this workspace are free again
  bool acc_pending = false, reduce_pending = false;
};

// One set of bases a host-slice caller keeps handing over (msm_amd_set_bases_cache): the converted device copy, keyed
// by (host pointer, n, layout) and guarded by sampled checksums of the host records.  The records are cut into
// S = n / 1024 interleaved phases (record i belongs to phase i mod S); every phase is hashed when the entry is filled,
// and every later call re-hashes phase 0 and one rotating phase of the caller's memory (~2 k records, tens of
// microseconds): a changed array at an unchanged address is caught at once if it touches phase 0 and within S calls
// otherwise.  The caller's contract stays "same pointer, same bases"; the checksums are the safety net.
struct BasesCacheEntry {
  const void* host = nullptr;
  size_t n = 0;
  int layout = 0;
  std::vector<uint64_t> phase_hash;
  std::vector<uint64_t> chunk_hash;   // kCacheChunks contiguous slices: the FULL verification (bases_cache_verify)
  void* d_prepared = nullptr;   // n x AffPacked
  uint64_t last_use = 0;        // bases_cache_call of the last call that used it
  uint64_t wanted_by = 0;       // bases_cache_call of the last call whose arguments name it (bases_cache_reserve)
  uint32_t next_phase = 1;
};

// Upload of PAGEABLE host memory through the library's own page-locked ring: helper threads copy 4 MiB chunks into
// pinned slots, each chunk goes on by DMA as soon as it is there.  The runtime's staged copy of pageable memory blocks
// the calling thread either way; on the HIP 7.0 runtime (the one inside the PyTorch wheel) it also serialises with
// the kernels of the other streams, this path does not (a page-locked source never did).
struct Stager {
  static constexpr size_t kChunk = (size_t)4 << 20;
  static constexpr int kSlots = 4, kMaxHelpers = 7;
  int n_helpers = 3;   // MSM_AMD_STAGE_HELPERS = 1 .. 7 overrides (measured: profiles/r03_staged_upload_helpers.txt)
  void* slot[kSlots] = {};
  hipEvent_t sent[kSlots] = {};
  bool busy[kSlots] = {};
  int next = 0;
  bool ready = false;
  // helper threads: copy parts 1 .. n_helpers of the current chunk (the caller copies part 0)
  std::vector<std::thread> helpers;
  std::mutex m;
  std::condition_variable wake, finished;
  const uint8_t* src = nullptr;
  uint8_t* dst = nullptr;
  size_t bytes = 0;
  std::atomic<uint64_t> generation{0};
  int pending = 0;
  bool quit = false;
};

// The uploads of a host-slice batch run on a helper thread, one instance ahead of the thread that submits kernels: a
// copy from pageable memory blocks whoever issues it (the runtime stages it, or staged_upload does), and on the
// submitting thread that was 1.3 ms per 2^20-point instance during which nothing else got enqueued.
struct UploadJob {
  void* d_scalars = nullptr;
  const void* h_scalars = nullptr;
  size_t scalar_bytes = 0;
  void* d_points = nullptr;
  const void* h_points = nullptr;
  size_t point_bytes = 0;          // 0: the points are resident (prepared bases, tables, a bases-cache hit)
  hipEvent_t wait_for = nullptr;   // GPU-side: the previous reader of the staging set (nullptr: none)
  hipEvent_t done = nullptr;       // recorded on the copy stream behind the copies
};
struct Uploader {
  std::thread th;
  std::mutex m;
  std::condition_variable cv, cv_idle;
  UploadJob job;
  bool has_job = false, busy = false, quit = false, started = false;
  int status = 0;
  std::string error;
};

constexpr int kWorkspaces = 4;
constexpr int kReduceStreams = 2;
constexpr int kMaxBatches = 4;   // batches that may be in flight between submit and wait

struct Batch {
  std::vector<InstanceSlot> slots;
  std::vector<Plan> plans;
  void* out = nullptr;
  size_t n_inst = 0;
  bool active = false;
  bool abandoned = false;   // a blocking entry point gave up waiting for it (wait timeout): released once the device is idle
};

// Precomputed window tables of one set of bases (msm_amd_tables_*): tables[w * n + i] = 2^(c w) P_i, packed form.
// Handles are validated by membership in msm_amd_ctx::live_tables, never by dereferencing the caller's pointer.
struct msm_amd_tables {
  size_t n = 0;
  uint32_t c = 0, W = 0;
  void* d_tables = nullptr;   // W * n AffPacked
};

struct msm_amd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;          // main stream: conversion, digits, sort, accumulate; stage entry points
  hipStream_t reduce_streams[kReduceStreams] = {};   // side streams: window reduction + copy of the partial points.
                                         // Consecutive instances alternate between them: the tail of an instance is a
                                         // chain of latency-bound kernels that is as long as one accumulate, so two
                                         // of them must be able to overlap.  (More streams than the four hardware
                                         // queues HIP gives a process -- main, front, two reduce -- share a queue
                                         // with the accumulate grid and wait behind it: measured, 1.6x slower.)
  bool alt_reduce = true;                // MSM_AMD_ALT_REDUCE=0: one reduce stream
  int acc_variant = 1;                   // accumulate kernel build (launch_accumulate): 1 shipped, 0 without the register pin; experiments build: 2..12
  uint32_t acc_lds = 0;                  // LDS bytes per accumulate workgroup: caps its waves per CU (MSM_AMD_ACC_LDS)
  uint32_t seq = 0;
  hipStream_t front_stream = nullptr;    // side stream: conversion, digits, sort, work-item planning
  hipStream_t copy_stream = nullptr;     // host-buffer entry points: uploads (DMA when the caller registered its buffers)
  hipEvent_t uploaded[2] = {nullptr, nullptr};   // per staging set: the upload on copy_stream has finished
  std::vector<std::pair<const void*, size_t>> host_regs;   // msm_amd_host_register
  bool overlap_reduce = true;            // MSM_AMD_OVERLAP_REDUCE=0 puts the reduction on the main stream
  bool overlap_front = true;             // MSM_AMD_OVERLAP_FRONT=0 puts the front end on the main stream
  bool lone_single_stream = true;        // MSM_AMD_LONE_SINGLE_STREAM=0: a lone instance uses the stream split too
  Workspace ws[kWorkspaces];
  std::mutex mu;
  std::string last_error;
  uint32_t forced_window = 0;
  std::vector<msm_amd_tables*> live_tables;
  int next_ws = 0;
  DeviceBuf scratch_a, scratch_b, scratch_c, scratch_b2, scratch_c2;
  Batch batches[kMaxBatches];
  msm_amd_timings timings{};
  float after_sort_state = -1.0f;   // timings.reserved2[1] of the next wait (see msm_amd_gpu_msm_h2c_sync)
  float after_sort_lead_ms = 0.0f;  // timings.reserved2[2]: device time from the callback to the end of accumulation
  hipEvent_t after_sort_mark = nullptr;   // recorded on the idle copy stream the moment the callback is about to run
  // Scalars staged by a single-instance entry point (msm_prepared, msm_tables, gpu_msm_h2c_sync): the upload is on
  // `upload_stream`; whichever stream enqueue_msm then picks for the front end waits for `upload_done` unless it IS
  // that stream (the entry point cannot know: run_batch_device may turn the call into pipelined point ranges).
  hipEvent_t upload_done = nullptr;
  hipStream_t upload_stream = nullptr;
  bool upload_pending = false;
  // Upper bound of every host wait for a GPU event (MSM_AMD_WAIT_TIMEOUT_MS, msm_amd_set_wait_timeout_ms; 0 = none).
  // A wait that runs into it returns MSM_AMD_PIPELINE_ERROR naming the event and instance not reached; the batch
  // stays in flight (`stalled`), and the next entry point first checks whether the device has caught up.
  uint32_t wait_timeout_ms = 60000;
  bool stalled = false;
  // Opt-in cache of converted bases for the host-slice entry points (msm_amd_set_bases_cache, MSM_AMD_BASES_CACHE_MB)
  std::vector<BasesCacheEntry> bases_cache;
  size_t bases_cache_budget = 0, bases_cache_bytes = 0;
  // 0 = sampled verification of a hit (phase 0 + one rotating phase of the caller's records), 1 = every record
  // re-hashed on every hit (msm_amd_set_bases_cache_verify, MSM_AMD_BASES_CACHE_VERIFY=full)
  int bases_cache_verify = 0;
  uint64_t bases_cache_call = 0, bases_cache_hits = 0, bases_cache_misses = 0, bases_cache_invalidations = 0;
  // device room for cache entries, allocated at the START of a host-slice call while the ctx is idle (nothing is ever
  // allocated under work in flight) and taken by bases_cache_insert when the instance is dispatched
  struct CacheReserve {
    const void* host;
    size_t n;
    int layout;
    void* d;
  };
  std::vector<CacheReserve> cache_reserve;
  AffPacked* convert_into = nullptr;   // enqueue_msm writes the converted bases of the next instance here (a cache fill)
  // Device buffers replaced by bigger ones while work was in flight.  hipFree synchronises the whole device: inside
  // a submit it would stall the pipeline and, on a device that does not answer, block without bound.  They are
  // freed when the ctx has nothing in flight (reap_graveyard) or at msm_amd_destroy.
  std::vector<void*> graveyard;
  Stager stager;
  Uploader uploader;
  int staged_uploads = -1;   // -1 = decide by the runtime version (stage_pageable()), 0 / 1 = MSM_AMD_STAGED_UPLOAD
};

namespace {

std::mutex g_global_mu;
msm_amd_ctx* g_global_ctx = nullptr;

// A helper thread of the library (the uploader) reports through its own string: ctx->last_error belongs to the thread
// that holds ctx->mu.
thread_local std::string* tl_error_sink = nullptr;

int fail(msm_amd_ctx* ctx, int status, const std::string& msg) {
  if (tl_error_sink) *tl_error_sink = msg;
  else if (ctx) ctx->last_error = msg;
  return status;
}

// Wait for everything this ctx has enqueued (error paths and set-up steps that reuse workspace 0).
void drain_streams(msm_amd_ctx* ctx) {
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  (void)hipStreamSynchronize(ctx->front_stream);
  (void)hipStreamSynchronize(ctx->stream);
  for (hipStream_t rs : ctx->reduce_streams) (void)hipStreamSynchronize(rs);
  (void)hipGetLastError();
}

// All streams of the ctx idle?  (hipStreamQuery never blocks)
bool streams_idle(msm_amd_ctx* ctx) {
  hipStream_t all[] = {ctx->copy_stream, ctx->front_stream, ctx->stream, ctx->reduce_streams[0], ctx->reduce_streams[1]};
  bool idle = true;
  for (hipStream_t s : all)
    if (s && hipStreamQuery(s) != hipSuccess) idle = false;
  (void)hipGetLastError();
  return idle;
}

// drain_streams with the ctx's wait bound: false if the device did not get there in time.
bool drain_streams_bounded(msm_amd_ctx* ctx) {
  if (ctx->wait_timeout_ms == 0) {
    drain_streams(ctx);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (!streams_idle(ctx)) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(ctx->wait_timeout_ms)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  return true;
}

// Error paths and set-up steps: the bounded drain; a device that does not get there leaves the ctx marked stalled (the
// next entry point checks whether it has caught up) instead of holding the caller.
bool drain_or_mark_stalled(msm_amd_ctx* ctx);
int sync_stream_bounded(msm_amd_ctx* ctx, hipStream_t st, const char* what);

// hipEventSynchronize may park the thread (it did, for more than a millisecond, inside a process that also hosts
// torch's thread pools) and it waits without bound: a device that stalls would block the caller of a blocking MSM
// for ever, where the reference's call always returns (msm.rs:237-349).  The waits on the critical path therefore
// poll the event -- spinning for the first few milliseconds, then with short sleeps -- up to `timeout_ms`
// (0 = unbounded) and report hipErrorNotReady when the bound is reached.
hipError_t wait_event(hipEvent_t ev, uint32_t timeout_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      return q;
    }
    (void)hipGetLastError();
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (timeout_ms && waited > std::chrono::milliseconds(timeout_ms)) return hipErrorNotReady;
    if (waited > std::chrono::milliseconds(4))
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    else
      __builtin_ia32_pause();
  }
}

uint32_t default_wait_timeout_ms() {
  if (const char* e = std::getenv("MSM_AMD_WAIT_TIMEOUT_MS")) return (uint32_t)std::strtoul(e, nullptr, 10);
  return 60000;
}

const char* const kEventNames[] = {"start", "convert", "digits", "sort", "accumulate-start", "accumulate", "reduce"};

#define HIP_TRY(ctx, expr)                                                                        \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      (void)hipGetLastError();                                                                    \
      return fail(ctx, MSM_AMD_PIPELINE_ERROR,                                                    \
                  std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +         \
                      std::to_string(__LINE__) + ")");                                            \
    }                                                                                             \
  } while (0)

// A wait on this ctx ran into its bound earlier.  Before anything new is enqueued: has the device caught up?  If every
// stream is idle the batches a blocking entry point abandoned are released; otherwise the call fails like the wait did.
int recover_if_stalled(msm_amd_ctx* ctx) {
  if (!ctx->stalled) return MSM_AMD_OK;
  if (!streams_idle(ctx))
    return fail(ctx, MSM_AMD_PIPELINE_ERROR,
                "the device is still busy with work whose wait timed out earlier (msm_amd_synchronize waits again; "
                "msm_amd_destroy gives the ctx up)");
  for (Batch& b : ctx->batches)
    if (b.abandoned) b.active = b.abandoned = false;
  ctx->stalled = false;
  return MSM_AMD_OK;
}

// Nothing of this ctx is in flight (every submitted batch has been waited for): outgrown buffers can go.
void reap_graveyard(msm_amd_ctx* ctx) {
  if (ctx->graveyard.empty() || ctx->stalled) return;
  for (const Batch& b : ctx->batches)
    if (b.active) return;
  for (void* p : ctx->graveyard) (void)hipFree(p);
  (void)hipGetLastError();
  ctx->graveyard.clear();
}

// hipMalloc / hipHostMalloc / hipFree / hipHostFree may wait for the device (hipFree always does).  With work in flight
// on a device that does not answer they would block without bound, before any bounded wait is reached -- so nothing
// is allocated or released while the ctx has work in flight: a buffer that must grow first waits, WITH the ctx's wait
// bound, for the ctx's streams to run empty, and the call fails with MSM_AMD_PIPELINE_ERROR if they do not.  (On a
// healthy device this costs a pipeline drain on the first use of a workspace, never in steady state.)
int quiesce_for_allocation(msm_amd_ctx* ctx, const char* what) {
  if (streams_idle(ctx)) return MSM_AMD_OK;
  if (drain_streams_bounded(ctx)) return MSM_AMD_OK;
  ctx->stalled = true;
  return fail(ctx, MSM_AMD_PIPELINE_ERROR,
              std::string("device busy past the wait bound of ") + std::to_string(ctx->wait_timeout_ms) +
                  " ms: cannot grow " + what + " while earlier work is in flight (msm_amd_synchronize waits again)");
}

// hipStreamSynchronize with the ctx's wait bound (the stream is polled): PIPELINE_ERROR, ctx marked stalled, when the
// device does not get there.
int sync_stream_bounded(msm_amd_ctx* ctx, hipStream_t st, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return MSM_AMD_OK;
    (void)hipGetLastError();
    if (q != hipErrorNotReady)
      return fail(ctx, MSM_AMD_PIPELINE_ERROR, std::string("hipStreamQuery (") + what + "): " + hipGetErrorString(q));
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (ctx->wait_timeout_ms && waited > std::chrono::milliseconds(ctx->wait_timeout_ms)) {
      ctx->stalled = true;
      return fail(ctx, MSM_AMD_PIPELINE_ERROR, "timed out after " + std::to_string(ctx->wait_timeout_ms) +
                                                   " ms waiting for the stream (" + what + ")");
    }
    if (waited > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else __builtin_ia32_pause();
  }
}

bool drain_or_mark_stalled(msm_amd_ctx* ctx) {
  if (drain_streams_bounded(ctx)) return true;
  ctx->stalled = true;
  return false;
}

int ensure(msm_amd_ctx* ctx, DeviceBuf& b, size_t bytes) {
  if (bytes <= b.cap) return MSM_AMD_OK;
  if (int rc = quiesce_for_allocation(ctx, "a device workspace")) return rc;
  if (b.p) ctx->graveyard.push_back(b.p);   // freed later, see msm_amd_ctx::graveyard
  b.p = nullptr;
  b.cap = 0;
  // grow by 25% to avoid re-allocating for every slightly larger instance
  const size_t want = bytes + bytes / 4;
  HIP_TRY(ctx, hipMalloc(&b.p, want));
  b.cap = want;
  return MSM_AMD_OK;
}

uint32_t floor_log2(size_t n) {
  uint32_t l = 0;
  while ((n >> (l + 1)) != 0) ++l;
  return l;
}

uint32_t auto_window(size_t n) {
  // The reference uses 3 below 32 points and 15 from there on (msm.rs:137-141).  Measured on MI355X in pipelined
  // batches (ms per MSM, round 2 -- the row / column-sum reduction made buckets cheaper, which moved every
  // boundary one size down): 2^12 c=5 0.23 | 2^14 c=5 0.24 | 2^16 c=13 0.33 | 2^18 c=15 0.47 (16: 0.50) |
  // 2^19 c=16 0.79 (15: 0.80, 17: 0.84) | 2^20 c=17 1.41 (16: 1.43, 15: 1.54) | 2^21 c=17 2.72 (16: 2.83) |
  // 2^22 c=17 5.82 (16: 6.21).  Windows whose TOP digit is narrow are traps: the n entries of the top window then
  // share a handful of buckets (c = 14 leaves 2 bits for the top window of a 254-bit scalar, c = 11 one bit: 2^17
  // points cost 1.2 ms with c = 10 or 11, 0.38 with 15).
  if (n < 32) return 3;   // msm.rs:137-138
  const uint32_t l = floor_log2(n);
  if (l <= 14) return 5;
  if (l <= 18) return 15;   // 2^16 in batches of 40 after the 64-bit host pass: c=15 0.207, c=13 0.233, c=16 0.246 ms per MSM
  if (l == 19) return 16;   // u32 digits from here on
  return kMaxWindow;
}

// Window size of a LONE call (one instance, nothing else in flight: lone_call()).  The pipelined policy above minimises
// GPU work per MSM; a lone call is a chain of launches and of dependent point additions, where a wide window costs
// little (the window reduction is a fixed number of levels, the accumulate kernel is far from filling the machine)
// and a narrow top window costs a lot (split buckets -> the combine pass).  Median ms of one blocking device-resident
// call on MI355X (tools/lone_latency.py, profiles/r02_lone_call_window_sweep.txt), best | pipelined policy:
//   2^8 c=8 0.346 | c=5 0.357     2^12 c=8 0.446 | c=5 0.538     2^14 c=15 0.487 | c=5 0.596
//   2^16 c=15 0.543 | c=13 0.661  2^18 c=15 0.810 (16: 0.796)    2^19 c=17 1.201 | c=16 1.253
uint32_t auto_window_lone(size_t n) {
  if (n < 32) return 3;
  const uint32_t l = floor_log2(n);
  if (l <= 6) return 5;
  if (l <= 12) return 8;
  if (l <= 18) return 15;
  return kMaxWindow;
}

// `windows` = 0: the per-call pipeline (every signed-digit window owns a bucket set).  `windows` = W_digits > 0: the
// precomputed-table pipeline, where the [W_digits][n_scalars] digit matrix is sorted as ONE window of
// W_digits * n_scalars entries whose "point index" addresses the table entry 2^(c w) P_i directly.
Plan make_plan(size_t n_scalars, uint32_t c, uint32_t windows = 0) {
  Plan p{};
  const size_t n = windows ? n_scalars * windows : n_scalars;
  p.n = (uint32_t)n;
  p.c = c;
  p.W_digits = kModulusBits / c + 1;  // signed digits: the top window absorbs the last carry (= ceil(255 / c))
  p.W = windows ? 1u : p.W_digits;
  p.n_scalars = (uint32_t)n_scalars;
  p.wide_digits = c > 15;
  p.lb = std::max(c - 1, (uint32_t)kSegLog);
  p.nb = 1u << p.lb;                  // slot i <-> digit magnitude i + 1 (c = 3 is padded to 8 slots)
  // sort / planning workgroup size: big workgroups are the fastest alone, but they can hardly be placed while
  // an accumulate grid of another instance is resident (they need 16 free wave slots on one CU at once)
  p.front_threads = 1024;
  if (const char* e = std::getenv("MSM_AMD_FRONT_THREADS")) {
    const int v = std::atoi(e);
    if (v == 256 || v == 512 || v == 1024) p.front_threads = (uint32_t)v;
  }
  // sort pass 1: one workgroup per (chunk, window); about 2 x 1024 threads per CU in total
  uint32_t Q = std::max(1u, (512u * (1024u / p.front_threads)) / p.W);
  const uint32_t max_q = (uint32_t)((n + 4095) / 4096);
  Q = std::max(1u, std::min(Q, max_q));
  p.Q = Q;
  p.chunk = (uint32_t)((((n + Q - 1) / Q) + 63) & ~(size_t)63);
  // Sort geometry.  Pass 2 sorts a region inside LDS, so the coarse passes must cut a window into regions of
  // ~16 k entries: T = ceil(log2(n / 16384)) coarse bits.  Up to 7 bits (128 regions) one coarse pass does it
  // (every per-call plan up to 2^21 points); beyond that the bits are split over TWO coarse passes of <= 6 bits
  // each (hb + mb, three-level sort): one pass with 1024 regions keeps too many partially written lines open
  // (2^24 points: 10.2 ms), and 128 regions of 131 k entries force pass 2 to scatter in global memory (6.0 ms).
  uint32_t T = 0;
  while (T + 2 < p.lb && (n >> T) > 16384) ++T;
  uint32_t hb = T, mb = 0;
  if (T > 7) {
    hb = (T + 1) / 2;
    mb = T - hb;
  }
  if (const char* e = std::getenv("MSM_AMD_HB")) {   // experiments: coarse bits of the two-pass sort
    const int v = std::atoi(e);
    if (v >= 0 && (uint32_t)v + 2 <= p.lb) {
      hb = (uint32_t)v;
      mb = 0;
    }
  }
  if (const char* e = std::getenv("MSM_AMD_MB")) {   // experiments: middle-pass bits
    const int v = std::atoi(e);
    if (v >= 0 && hb + (uint32_t)v + 2 <= p.lb) mb = (uint32_t)v;
  }
  p.hb = hb;
  p.mb = mb;
  p.fb = p.lb - hb - mb;
  if (p.fb > 10) {   // the fine histogram is scanned with one bin per thread (<= 1024 bins)
    const uint32_t extra = p.fb - 10;
    p.fb = 10;
    if (p.mb || p.hb + extra > 7) p.mb += extra; else p.hb += extra;
  }
  // what pass 1 leaves in the u16 tmp_fine is mb + fb bits
  while (p.mb + p.fb > 16) {
    ++p.hb;
    --p.mb;
  }
  // Tile-staged scatter (k_sort.hip tile_scatter): 26 % less sort time for ONE huge instance (2^24 points: 7.0 ->
  // 5.2 ms with 1024-thread workgroups), but its 60 VGPRs x 1024 threads cannot be placed beside a resident
  // accumulate grid, so inside a pipeline of instances it loses (headline 695 -> 643 MSM/s; with 512 threads: no
  // difference to the direct scatter).  Used where the sort is not hidden behind another instance's accumulation:
  // per-call plans that need the three-level sort (2^22 points and more).
  p.tiled = (p.mb != 0 && windows == 0) ? 1u : 0u;
  if (const char* e = std::getenv("MSM_AMD_TILED")) p.tiled = std::atoi(e) != 0;
  p.tile_threads = 1024;
  if (const char* e = std::getenv("MSM_AMD_TILE_THREADS")) {
    const int v = std::atoi(e);
    if (v == 256 || v == 512 || v == 1024) p.tile_threads = (uint32_t)v;
  }
  p.ballot = kDefaultBallot;
  if (const char* e = std::getenv("MSM_AMD_BALLOT")) p.ballot = (uint32_t)std::atoi(e) & 3u;
  p.Q2 = 1;
  if (p.mb) {
    const uint32_t regions = p.W << p.hb;
    p.Q2 = std::max(1u, 1024u / regions);
    while (p.Q2 > 1 && (n >> p.hb) / p.Q2 < 4096) p.Q2 >>= 1;
  }
  // accumulate work items: at most CH points each; buckets longer than CH are cut into several items whose partial
  // sums the combine kernels add up (one extra full addition per cut, and a second affine+affine start).  Uniform
  // scalars must split (almost) nothing: CH >= mean + 5 sigma of the Poisson bucket size (2^20 points, c = 16: mean
  // 32, CH = 64; the round-1 rule picked 32 there and cut 228 k of 524 k buckets in two).  Only when the buckets
  // alone cannot fill the chip twice (2 waves/SIMD x 1024 SIMDs x 64 lanes = 131 k lanes per round) are items made
  // shorter on purpose.  Skewed inputs (equal scalars, narrow top windows) still get cut at CH.
  const double mean_len = (double)n / (double)p.nb;
  uint32_t ch = 16;
  while (ch < 512 && (double)ch < mean_len + 5.0 * std::sqrt(mean_len)) ch <<= 1;
  while (ch > 16 && (size_t)p.W * p.nb + ((size_t)p.W * n) / ch < 262144) ch >>= 1;
  if (const char* e = std::getenv("MSM_AMD_CH")) {
    const int v = std::atoi(e);
    if (v >= 16 && v <= 512 && (v & (v - 1)) == 0) ch = (uint32_t)v;
  }
  p.CH = ch;
  p.red_L = (p.lb + 1) / 2;
  p.red_H = p.lb - p.red_L;
  p.red_group = kReduceGroup;
  p.total_buckets = (size_t)p.W * p.nb;
  p.total_segs = (size_t)p.W * reduce_scratch_elems(p.lb);   // scratch elements of each of the two sum families
  p.max_items = p.total_buckets + ((size_t)p.W * n) / ch + 1;
  p.partial_count = (size_t)p.W * (p.lb + 1);
  return p;
}

// Geometry of the window reduction alone (stage entry point sum_reduction).
Plan make_reduce_plan(uint32_t lb, uint32_t W) {
  Plan p{};
  p.c = lb + 1;   // only used to space bit positions in host_combine (one window at a time)
  p.W = W;
  p.lb = lb;
  p.nb = 1u << lb;
  p.red_L = (lb + 1) / 2;
  p.red_H = lb - p.red_L;
  p.red_group = kReduceGroup;
  p.total_buckets = (size_t)p.W * p.nb;
  p.total_segs = (size_t)p.W * reduce_scratch_elems(lb);
  p.partial_count = (size_t)p.W * (lb + 1);
  return p;
}

// ---- host-side big-endian-limb helpers (reference wire layout) ----------------------------------
u256 be32_to_u256(const uint32_t* l) {
  u256 r;
  for (int i = 0; i < 8; ++i) r.v[i] = l[7 - i];
  return r;
}
void u256_to_be32(const u256& a, uint32_t* l) {
  for (int i = 0; i < 8; ++i) l[i] = a.v[7 - i];
}
Jacobian be32_to_jac(const uint32_t* l) {
  Jacobian r;
  r.x = be32_to_u256(l);
  r.y = be32_to_u256(l + 8);
  r.z = be32_to_u256(l + 16);
  return r;
}
void jac_to_be32(const Jacobian& p, uint32_t* l) {
  u256_to_be32(p.x, l);
  u256_to_be32(p.y, l + 8);
  u256_to_be32(p.z, l + 16);
}

// Normalise to z = R mod p (or the canonical identity (1,1,0) in Montgomery form).
Jacobian normalise(const Jacobian& p) { return h64::store(h64::normalise(h64::load(p))); }

// Window value  W_w = partial[w][lb] + sum_k 2^k * partial[w][k]  (total; bit sums of the column sums for k < L, of the
// row sums -- already offset by L -- for L <= k < lb; see k_reduce.hip) and the final Horner sum_w 2^(c*w) W_w, fused
// into ONE pass over bit positions: term partial[w][k] sits at bit c*w + k, the total at bit c*w.
// Replaces sum_reduction_final + final_accumulation.rs:19-39.
Jacobian host_combine(const Jacobian* partial, const Plan& p) {
  const uint32_t top = p.c * (p.W - 1) + p.lb;   // highest bit position in use
  // 4 x 64-bit host arithmetic (host_fq64.h): this pass is the whole CPU tail of an MSM
  h64::Jac acc = h64::identity();
  for (int pos = (int)top; pos >= 0; --pos) {
    acc = h64::jdouble(acc);
    // the terms at this bit: partial[w][k] with c w + k = pos and k < lb, and the window total partial[w][lb] at k = 0.
    // Production plans have lb = c - 1 (one window per position); tiny windows keep lb >= kSegLog > c - 1, where the
    // upper bit sums of window w share positions with window w + 1
    for (uint32_t w = std::min((uint32_t)pos / p.c, p.W - 1);; --w) {
      const uint32_t k = (uint32_t)pos - p.c * w;
      if (k > p.lb) break;
      const Jacobian* pw = partial + (size_t)w * (p.lb + 1);
      if (k == 0) acc = h64::jadd(acc, h64::load(pw[p.lb]));
      if (k < p.lb) acc = h64::jadd(acc, h64::load(pw[k]));
      if (w == 0) break;
    }
  }
  return h64::store(acc);
}

int set_kernel_attributes(msm_amd_ctx* ctx) {
  const char* failed = nullptr;
  if (sort_set_attributes(&failed) || reduce_set_attributes(&failed))
    return fail(ctx, MSM_AMD_FUNCTION_ERROR, std::string("hipFuncSetAttribute(") + (failed ? failed : "?") + ")");
  return MSM_AMD_OK;
}

int slot_prepare(msm_amd_ctx* ctx, InstanceSlot& s, size_t partial_count) {
  if (!s.has_events) {
    for (int i = 0; i < EV_COUNT; ++i) HIP_TRY(ctx, hipEventCreate(&s.ev[i]));
    s.has_events = true;
  }
  if (partial_count > s.h_partial_cap) {
    if (int rc = quiesce_for_allocation(ctx, "a page-locked result slot")) return rc;
    if (s.h_partial) HIP_TRY(ctx, hipHostFree(s.h_partial));
    s.h_partial = nullptr;
    // one extra record at the end receives the plan counters of the instance (work-item statistics for timings)
    HIP_TRY(ctx, hipHostMalloc((void**)&s.h_partial, (partial_count + 1) * sizeof(Jacobian), hipHostMallocDefault));
    s.h_partial_cap = partial_count;
  }
  return MSM_AMD_OK;
}

// ONE instance submitted while nothing else is in flight (what a blocking gpu_msm_h2c call is) runs on ONE stream:
// the front / accumulate / reduce split exists to overlap neighbouring instances, and with no neighbour its three
// cross-stream event hand-offs are pure latency -- 29 + 28 + 18 us of GPU idle time inside a 742 us call at 2^18
// points (profiles/r02_lone_call_2p18_timeline.txt).
bool lone_call(const msm_amd_ctx* ctx, size_t n_inst) {
  if (n_inst != 1 || !ctx->lone_single_stream) return false;
  for (const Batch& b : ctx->batches)
    if (b.active) return false;
  return true;
}
hipStream_t front_stream_of(const msm_amd_ctx* ctx, bool lone) {
  return (lone || !ctx->overlap_front) ? ctx->stream : ctx->front_stream;
}

// Bring inputs to the native device layout (affine 64 B Montgomery LE; scalars 32 B LE).
// On return *scalars_native / *points_native point to device memory valid until the next call.
int convert_inputs(msm_amd_ctx* ctx, Workspace& w, hipStream_t st, int scalar_layout, int point_layout, const void* d_scalars,
                   const void* d_points, size_t n, const u256** scalars_native, int* scalars_mont,
                   const Affine** points_native) {
  switch (scalar_layout) {
    case MSM_AMD_SCALAR_MONT_LE:
      *scalars_native = (const u256*)d_scalars;
      *scalars_mont = 1;
      break;
    case MSM_AMD_SCALAR_CANON_LE:
      *scalars_native = (const u256*)d_scalars;
      *scalars_mont = 0;
      break;
    case MSM_AMD_SCALAR_CANON_BE32: {
      int rc = ensure(ctx, w.conv_scalars, n * 32);
      if (rc) return rc;
      launch_be32_to_le(st, (const uint32_t*)d_scalars, n, (uint32_t*)w.conv_scalars.p);
      *scalars_native = (const u256*)w.conv_scalars.p;
      *scalars_mont = 0;
      break;
    }
    default:
      return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown scalar layout");
  }
  switch (point_layout) {
    case MSM_AMD_POINT_PREPARED:   // already in the internal packed form (msm_amd_bases_*): nothing to convert
      *points_native = nullptr;
      break;
    case MSM_AMD_POINT_H2C_AFFINE:
      *points_native = (const Affine*)d_points;
      break;
    case MSM_AMD_POINT_ARK_PROJECTIVE: {
      int rc = ensure(ctx, w.conv_points, n * sizeof(Affine));
      if (rc) return rc;
      launch_projective_to_affine(st, (const Jacobian*)d_points, (uint32_t)n, (Affine*)w.conv_points.p);
      *points_native = (const Affine*)w.conv_points.p;
      break;
    }
    case MSM_AMD_POINT_ARK_AFFINE: {
      int rc = ensure(ctx, w.conv_points, n * sizeof(Affine));
      if (rc) return rc;
      launch_ark_affine_to_affine(st, (const uint8_t*)d_points, (uint32_t)n, (Affine*)w.conv_points.p);
      *points_native = (const Affine*)w.conv_points.p;
      break;
    }
    case MSM_AMD_POINT_JAC_BE32: {
      int rc = ensure(ctx, w.conv_tmp, n * sizeof(Jacobian));
      if (rc) return rc;
      rc = ensure(ctx, w.conv_points, n * sizeof(Affine));
      if (rc) return rc;
      launch_be32_to_le(st, (const uint32_t*)d_points, n * 3, (uint32_t*)w.conv_tmp.p);
      launch_projective_to_affine(st, (const Jacobian*)w.conv_tmp.p, (uint32_t)n, (Affine*)w.conv_points.p);
      *points_native = (const Affine*)w.conv_points.p;
      break;
    }
    default:
      return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown point layout");
  }
  HIP_TRY(ctx, hipGetLastError());
  return MSM_AMD_OK;
}

size_t scalar_bytes(int) { return 32; }
size_t point_bytes(int layout) {
  switch (layout) {
    case MSM_AMD_POINT_H2C_AFFINE: return 64;
    case MSM_AMD_POINT_PREPARED: return sizeof(AffPacked);
    case MSM_AMD_POINT_TABLES: return sizeof(AffPacked);
    case MSM_AMD_POINT_ARK_PROJECTIVE: return 96;
    case MSM_AMD_POINT_ARK_AFFINE: return 72;
    case MSM_AMD_POINT_JAC_BE32: return 96;
  }
  return 0;
}

// Window reduction of a production-layout bucket matrix: buckets [W][nb] -> partial [W][K+1] on device.
int enqueue_reduce(msm_amd_ctx* ctx, Workspace& w, hipStream_t st, const Plan& p, const PtI* buckets,
                   const uint32_t* bucket_size) {
  int rc;
  if ((rc = ensure(ctx, w.S, p.total_segs * sizeof(PtI)))) return rc;
  if ((rc = ensure(ctx, w.T, p.total_segs * sizeof(PtI)))) return rc;
  if ((rc = ensure(ctx, w.partial, p.partial_count * sizeof(Jacobian)))) return rc;
  launch_reduce(st, p, buckets, bucket_size, (PtI*)w.S.p, (PtI*)w.T.p, (Jacobian*)w.partial.p);
  HIP_TRY(ctx, hipGetLastError());
  return MSM_AMD_OK;
}

const msm_amd_tables* find_tables(const msm_amd_ctx* ctx, const void* handle) {
  for (const msm_amd_tables* t : ctx->live_tables)
    if ((const void*)t == handle) return t;
  return nullptr;
}

// Row / column sums of the window reduction for a LONE call: with no neighbouring instance to fill the machine, the
// 15-addition chains of the pipelined setting (one lane per 16 buckets: 35 k lanes at 2^18 points, 7.7 k in the second
// level at 2^20) are pure latency.  launch_reduce then takes, level by level, the smallest group (>= this minimum) whose
// outputs still fit the lanes one launch can have resident; a further level costs one launch (~5 us), one addition
// in a chain about 6 us.
uint32_t pick_reduce_group(const Plan&) {
  if (const char* e = std::getenv("MSM_AMD_REDUCE_GROUP")) return (uint32_t)std::atoi(e);
  return kReduceGroupMin;
}

// Enqueue one whole MSM on the ctx stream; results land in slot.h_partial after slot.ev[EV_REDUCE].
int enqueue_msm(msm_amd_ctx* ctx, Workspace& w, InstanceSlot& slot, int scalar_layout, int point_layout, const void* d_scalars,
                const void* d_points, size_t n, Plan* plan_out, bool lone) {
  hipStream_t st = ctx->stream;
  const msm_amd_tables* tb = nullptr;
  if (point_layout == MSM_AMD_POINT_TABLES) {   // d_points is the handle of msm_amd_tables_build*
    tb = find_tables(ctx, d_points);
    if (!tb) return fail(ctx, MSM_AMD_INPUT_ERROR, "not a table handle of this ctx");
    if (n != tb->n) return fail(ctx, MSM_AMD_INPUT_ERROR, "n differs from the number of points the tables hold");
  }
  const uint32_t c = tb ? tb->c : (ctx->forced_window ? ctx->forced_window : (lone ? auto_window_lone(n) : auto_window(n)));
  Plan p = tb ? make_plan(n, c, tb->W) : make_plan(n, c);
  if (lone) p.red_group = pick_reduce_group(p);   // (pipelined small instances: 2^16 -6 %, 2^18 +3 % -- not taken)
  p.rb_threads = lone ? 0u : 64u;                 // one wave per bit-subset sum beside a resident accumulate grid (k_reduce.hip)
  // the tile-staged scatter (1024-thread workgroups, 60 VGPRs, 64 KB of LDS) is for a sort that has the machine to
  // itself; beside a resident accumulate grid it is slower than the plain scatter (4 x 2^22 points: 6.15 vs 5.65 ms per MSM)
  if (!lone && !std::getenv("MSM_AMD_TILED")) p.tiled = 0;
  *plan_out = p;
  int rc;
  if ((rc = slot_prepare(ctx, slot, p.partial_count))) return rc;
  n = p.n;   // from here on: sorted entries per window (= points, or W_digits * points with tables)
  if ((rc = ensure(ctx, w.digits, (size_t)p.W * n * (p.wide_digits ? sizeof(uint32_t) : sizeof(uint16_t))))) return rc;
  if ((rc = ensure(ctx, w.coarse_cnt, (size_t)p.W * p.Q * (1u << p.hb) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.region_start, (size_t)p.W * ((1u << p.hb) + 1) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.tmp_idx, (size_t)p.W * n * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.tmp_fine, (size_t)p.W * n * sizeof(uint16_t)))) return rc;
  if (p.mb) {
    if ((rc = ensure(ctx, w.tmp_idx2, (size_t)p.W * n * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, w.tmp_fine2, (size_t)p.W * n * sizeof(uint16_t)))) return rc;
    if ((rc = ensure(ctx, w.mid_cnt, ((size_t)p.W << p.hb) * p.Q2 * (1u << p.mb) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, w.region_start2, (size_t)p.W * ((1u << (p.hb + p.mb)) + 1) * sizeof(uint32_t)))) return rc;
  }
  if ((rc = ensure(ctx, w.bsize, p.total_buckets * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.bstart, p.total_buckets * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.istart, p.total_buckets * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.win_items, (1024 + 2 * 1024) * sizeof(uint32_t)))) return rc;   // + tile_sums (<= 1024 tiles)
  if ((rc = ensure(ctx, w.size_bins, (size_t)(p.CH + 1) * ((p.total_buckets + p.front_threads - 1) / p.front_threads) *
                                         sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.sorted, (size_t)p.W * n * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.order, p.max_items * sizeof(uint2)))) return rc;
  if ((rc = ensure(ctx, w.multi_list, p.max_items * sizeof(uint32_t)))) return rc;
  if (ctx->acc_variant == 4 && (rc = ensure(ctx, w.redo_list, p.max_items * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, w.counters, sizeof(PlanCounters)))) return rc;
  const bool prepared = point_layout == MSM_AMD_POINT_PREPARED || tb != nullptr;
  AffPacked* const fill = prepared ? nullptr : ctx->convert_into;   // bases cache fill: convert straight into the entry
#if defined(MSM_AMD_EXPERIMENTS)
  // variant 8: bases converted inside this call go into WIDE records (128 B: x, y, -y as limbs)
  const bool wide = !prepared && !fill && ctx->acc_variant == 8;
  const size_t base_record = wide ? sizeof(AffWide) : sizeof(AffPacked);
#else
  const bool wide = false;
  const size_t base_record = sizeof(AffPacked);
#endif
  if (!prepared && !fill && (rc = ensure(ctx, w.bases29, n * base_record))) return rc;
  if ((rc = ensure(ctx, w.buckets, p.total_buckets * sizeof(PtI)))) return rc;
  if ((rc = ensure(ctx, w.item_partials, p.max_items * sizeof(PtI)))) return rc;
  SortBuffers sb{};
  sb.digits = w.digits.p;
  sb.coarse_cnt = (uint32_t*)w.coarse_cnt.p;
  sb.region_start = (uint32_t*)w.region_start.p;
  sb.tmp_idx = (uint32_t*)w.tmp_idx.p;
  sb.tmp_fine = (uint16_t*)w.tmp_fine.p;
  sb.tmp_idx2 = (uint32_t*)w.tmp_idx2.p;
  sb.tmp_fine2 = (uint16_t*)w.tmp_fine2.p;
  sb.mid_cnt = (uint32_t*)w.mid_cnt.p;
  sb.region_start2 = (uint32_t*)w.region_start2.p;
  sb.bucket_size = (uint32_t*)w.bsize.p;
  sb.bucket_start = (uint32_t*)w.bstart.p;
  sb.item_start = (uint32_t*)w.istart.p;
  sb.win_items = (uint32_t*)w.win_items.p;
  sb.tile_sums = (uint2*)((uint32_t*)w.win_items.p + 1024);   // second half of the same small buffer
  sb.size_bins = (uint32_t*)w.size_bins.p;
  sb.sorted = (uint32_t*)w.sorted.p;
  sb.order = (uint2*)w.order.p;
  sb.multi_list = (uint32_t*)w.multi_list.p;
  sb.redo_list = (uint32_t*)w.redo_list.p;
  sb.counters = (PlanCounters*)w.counters.p;

  // Four streams (front, main, two alternating reduce streams), kWorkspaces workspaces (consecutive instances take
  // consecutive workspaces):
  //   front  : conversion, digits, sort, planning, bucket clear of instance i -- needs the workspace's previous
  //            accumulate and reduction done; runs while instance i-1 accumulates and i-2 reduces
  //   main   : accumulate of instance i                                      -- needs front(i)
  //   reduce : combine + window reduction + copy of instance i               -- needs main(i)
  hipStream_t fs = front_stream_of(ctx, lone);
  hipStream_t rs = (lone || !ctx->overlap_reduce)
                       ? st
                       : ctx->reduce_streams[ctx->alt_reduce ? ctx->seq++ % kReduceStreams : 0];
  if (w.acc_pending && (fs != st || lone)) {   // the previous accumulate in this workspace still reads its plan ...
    HIP_TRY(ctx, hipStreamWaitEvent(fs, w.acc_done, 0));
  }
  if (w.reduce_pending && (fs != rs || lone)) {   // ... and so does its combine pass (on whichever reduce stream
    HIP_TRY(ctx, hipStreamWaitEvent(fs, w.reduce_done, 0));   // the previous user of the workspace had)
  }
  w.acc_pending = false;
  if (ctx->upload_pending) {   // scalars staged by the entry point (stage_upload): order them before this front end
    if (fs != ctx->upload_stream) {
      HIP_TRY(ctx, hipEventRecord(ctx->upload_done, ctx->upload_stream));
      HIP_TRY(ctx, hipStreamWaitEvent(fs, ctx->upload_done, 0));
      ctx->upload_stream = fs;   // later point ranges of the same call run their front ends on this stream too
    }
  }
  HIP_TRY(ctx, hipEventRecord(slot.ev[EV_START], fs));
  const u256* sc = nullptr;
  const Affine* pts = nullptr;
  int sc_mont = 0;
  if ((rc = convert_inputs(ctx, w, fs, scalar_layout, tb ? MSM_AMD_POINT_PREPARED : point_layout, d_scalars, d_points,
                           p.n_scalars, &sc, &sc_mont, &pts)))
    return rc;
  const AffPacked* bases = tb ? (const AffPacked*)tb->d_tables
                              : (prepared ? (const AffPacked*)d_points
                                          : (fill ? (const AffPacked*)fill : (const AffPacked*)w.bases29.p));
#if defined(MSM_AMD_EXPERIMENTS)
  if (wide) launch_convert_bases_wide(fs, pts, p.n, (AffWide*)w.bases29.p);
#endif
  if (!prepared && !wide)   // external 8 x u32 -> packed internal domain
    launch_convert_bases(fs, pts, p.n, fill ? fill : (AffPacked*)w.bases29.p);
  HIP_TRY(ctx, hipEventRecord(slot.ev[EV_CONVERT], fs));
  launch_digits(fs, p, sc, sc_mont, sb.digits);
  HIP_TRY(ctx, hipEventRecord(slot.ev[EV_DIGITS], fs));
  launch_sort(fs, p, sb);
  HIP_TRY(ctx, hipEventRecord(slot.ev[EV_SORT], fs));
  // the bucket matrix is NOT cleared: a bucket without points gets no work item and is never written; the window
  // reduction reads bucket_size and takes such a slot as the identity (the reference relies on Metal's zero-filled
  // fresh buffers instead, msm.rs:154-156)
  if (fs != st) {
    HIP_TRY(ctx, hipEventRecord(w.front_done, fs));
    HIP_TRY(ctx, hipStreamWaitEvent(st, w.front_done, 0));
  }
  if (w.reduce_pending) {   // the previous user of this workspace may still be reducing its buckets
    if (rs != st) HIP_TRY(ctx, hipStreamWaitEvent(st, w.reduce_done, 0));
    w.reduce_pending = false;
  }
  launch_accumulate(st, p, bases, wide ? 1 : 0, sb, (PtI*)w.buckets.p, (PtI*)w.item_partials.p,
                    ctx->acc_variant, ctx->acc_lds, slot.ev[EV_ACC_K0], slot.ev[EV_ACC_K1]);
  HIP_TRY(ctx, hipEventRecord(w.acc_done, st));
  w.acc_pending = true;

  // combine (split buckets) + window reduction + copy: off the main stream, which goes straight to the next
  // accumulate.  front(i+2) reuses this workspace's plan buffers, which combine still reads: it waits for
  // reduce_done as well (see the top of this function).
  if (rs != st) HIP_TRY(ctx, hipStreamWaitEvent(rs, w.acc_done, 0));
  launch_combine(rs, p, sb, (PtI*)w.buckets.p, (PtI*)w.item_partials.p);
  if ((rc = enqueue_reduce(ctx, w, rs, p, (const PtI*)w.buckets.p, (const uint32_t*)w.bsize.p))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(slot.h_partial, w.partial.p, p.partial_count * sizeof(Jacobian),
                              hipMemcpyDeviceToHost, rs));
  HIP_TRY(ctx, hipMemcpyAsync(slot.h_partial + slot.h_partial_cap, w.counters.p, sizeof(PlanCounters),
                              hipMemcpyDeviceToHost, rs));
  HIP_TRY(ctx, hipEventRecord(slot.ev[EV_REDUCE], rs));
  HIP_TRY(ctx, hipEventRecord(w.reduce_done, rs));
  w.reduce_pending = true;
  HIP_TRY(ctx, hipGetLastError());
  return MSM_AMD_OK;
}

void accumulate_timings(msm_amd_ctx* ctx, InstanceSlot& s, const Plan& p, float final_ms, size_t n_inst) {
  auto span = [&](int a, int b) {
    float t = 0;
    if (hipEventElapsedTime(&t, s.ev[a], s.ev[b]) != hipSuccess) {
      (void)hipGetLastError();
      t = 0;
    }
    return t;
  };
  float ms[EV_COUNT] = {0};
  ms[EV_CONVERT] = span(EV_START, EV_CONVERT);
  ms[EV_DIGITS] = span(EV_CONVERT, EV_DIGITS);
  ms[EV_SORT] = span(EV_DIGITS, EV_SORT);
  ms[EV_ACC] = span(EV_ACC_S, EV_ACC);
  ms[EV_REDUCE] = span(EV_RED_S, EV_REDUCE);
  msm_amd_timings& T = ctx->timings;
  const float inv = 1.0f / (float)n_inst;
  T.convert_ms += ms[EV_CONVERT] * inv;
  T.digits_ms += ms[EV_DIGITS] * inv;
  T.sort_ms += ms[EV_SORT] * inv;
  T.accumulate_ms += ms[EV_ACC] * inv;
  T.reduce_ms += ms[EV_REDUCE] * inv;
  T.accumulate_kernel_ms += span(EV_ACC_K0, EV_ACC_K1) * inv;
  T.final_ms += final_ms * inv;
  T.total_gpu_ms += (ms[EV_CONVERT] + ms[EV_DIGITS] + ms[EV_SORT] + ms[EV_ACC] + ms[EV_REDUCE]) * inv;
  T.n = p.n_scalars;
  T.window_size = p.c;
  T.num_windows = p.W_digits;
  T.reserved = (uint32_t)n_inst;
  PlanCounters pc;
  std::memcpy(&pc, s.h_partial + s.h_partial_cap, sizeof pc);
  T.reserved2[0] = (float)pc.total_items;   // work items (= lanes with work) of the last instance's accumulate grid
  T.reserved2[1] = ctx->after_sort_state;
  T.reserved2[2] = ctx->after_sort_lead_ms;
}

// Batch of MSMs with device-resident inputs, in two halves so that callers can pipeline batches:
//   submit_batch_device  enqueues every kernel of every instance (four streams, see enqueue_msm) and returns
//   wait_batch           finishes each instance on the host as soon as its partial points have landed (the host
//                        Horner pass of instance i overlaps the GPU work of i+1.. and of later batches)
// lone_hint: -1 = decide here (one instance, nothing in flight), 0 / 1 = the caller (run_batch_host, which submits
// the instances of ITS batch one by one and has already put the upload waits on a stream) decided
int submit_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                        const void* const* d_scalars, const void* const* d_points, const size_t* n, void* out_host,
                        int* ticket, int lone_hint = -1) {
  if (!ctx || !d_scalars || !d_points || !n || !out_host || !ticket || n_inst == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "null argument or empty batch");
  if (point_bytes(point_layout) == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown point layout");
  for (size_t i = 0; i < n_inst; ++i) {
    if (n[i] == 0 || n[i] > 0x7FFFFFFFull || !d_scalars[i] || !d_points[i])
      return fail(ctx, MSM_AMD_INPUT_ERROR, "instance with n == 0, n >= 2^31 or null pointer");
  }
  if (int rc = recover_if_stalled(ctx)) return rc;
  int id = -1;
  for (int b = 0; b < kMaxBatches; ++b)
    if (!ctx->batches[b].active) {
      id = b;
      break;
    }
  if (id < 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "too many batches in flight: wait for one first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const bool lone = lone_hint < 0 ? lone_call(ctx, n_inst) : lone_hint != 0;
  Batch& B = ctx->batches[id];
  if (B.slots.size() < n_inst) B.slots.resize(n_inst);
  B.plans.assign(n_inst, Plan{});
  B.out = out_host;
  B.n_inst = n_inst;
  for (size_t i = 0; i < n_inst; ++i) {
    Workspace& w = ctx->ws[ctx->next_ws];
    ctx->next_ws = (ctx->next_ws + 1) % kWorkspaces;
    int rc = enqueue_msm(ctx, w, B.slots[i], scalar_layout, point_layout, d_scalars[i], d_points[i], n[i], &B.plans[i],
                         lone);
    if (rc) {   // nothing of a failed submit stays in flight (or the ctx is marked stalled); the slot was never active
      (void)drain_or_mark_stalled(ctx);
      return rc;
    }
  }
  B.active = true;
  *ticket = id;
  return MSM_AMD_OK;
}

// Host threads for the CPU tail (window Horner pass + normalisation, ~0.11 ms per instance) of one batch.  Batches of
// small MSMs are bound by it: 40 x 2^12 points take 0.16 ms per MSM on one thread, of which 0.11 ms is this pass and
// 0.05 ms the enqueue.  One thread per 4 instances, at most 4 and at most half of the cores; MSM_AMD_FINISH_THREADS
// overrides (1 = the calling thread only).
unsigned finish_threads(size_t n_inst) {
  if (const char* e = std::getenv("MSM_AMD_FINISH_THREADS")) return (unsigned)std::max(1, std::atoi(e));
  const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
  return (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)4, n_inst / 4, (size_t)hw / 2}));
}

// `internal`: the caller is one of the library's blocking entry points, which cannot hand the ticket back to its own
// caller -- on a timed-out wait the batch is marked abandoned (released by recover_if_stalled once the device is idle).
// Through msm_amd_wait_batch the ticket stays valid and the caller may wait again.
int wait_batch(msm_amd_ctx* ctx, int ticket, bool internal = true) {
  if (!ctx || ticket < 0 || ticket >= kMaxBatches || !ctx->batches[ticket].active || ctx->batches[ticket].abandoned)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown batch ticket");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Batch& B = ctx->batches[ticket];
  ctx->timings = msm_amd_timings{};
  struct Done {
    hipError_t err = hipSuccess;
    float final_ms = 0;
  };
  std::vector<Done> done(B.n_inst);
  const uint32_t timeout_ms = ctx->wait_timeout_ms;
  // instance i: wait for its partial sums, Horner pass, result; instances are independent, each writes its own slot
  auto finish_from = [&](size_t first, size_t stride, bool set_device) {
    if (set_device && hipSetDevice(ctx->device) != hipSuccess) {
      for (size_t i = first; i < B.n_inst; i += stride) done[i].err = hipErrorInvalidDevice;
      return;
    }
    for (size_t i = first; i < B.n_inst; i += stride) {
      InstanceSlot& s = B.slots[i];
      done[i].err = wait_event(s.ev[EV_REDUCE], timeout_ms);
      if (done[i].err != hipSuccess) {
        if (done[i].err == hipErrorNotReady) {   // the later instances of this thread are behind the same stall
          for (size_t k = i + stride; k < B.n_inst; k += stride) done[k].err = hipErrorNotReady;
          return;
        }
        continue;
      }
      const auto t0 = std::chrono::steady_clock::now();
      const Jacobian res = normalise(host_combine(s.h_partial, B.plans[i]));
      done[i].final_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::memcpy((uint8_t*)B.out + i * 96, &res, 96);
    }
  };
  const unsigned T = std::min<size_t>(finish_threads(B.n_inst), B.n_inst);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < T; ++t) pool.emplace_back(finish_from, (size_t)t, (size_t)T, true);
  finish_from(0, T, false);
  for (std::thread& th : pool) th.join();
  for (size_t i = 0; i < B.n_inst; ++i) {
    if (done[i].err == hipErrorNotReady) {   // the bound of the wait was reached: the work is still in flight
      ctx->stalled = true;
      if (internal) B.abandoned = true;
      const InstanceSlot& s = B.slots[i];
      int reached = -1;   // last stage event of this instance the device did get to
      for (int e = 0; e < EV_COUNT; ++e) {
        if (hipEventQuery(s.ev[e]) == hipSuccess) reached = e;
      }
      (void)hipGetLastError();
      return fail(ctx, MSM_AMD_PIPELINE_ERROR,
                  "timed out after " + std::to_string(timeout_ms) + " ms waiting for event 'reduce' of instance " +
                      std::to_string(i) + " of " + std::to_string(B.n_inst) + " (ticket " + std::to_string(ticket) +
                      "); last event reached: " + (reached < 0 ? "none" : kEventNames[reached]) +
                      (internal ? "" : "; the ticket stays valid"));
    }
    if (done[i].err != hipSuccess) {   // release the ticket on every exit: a failed wait must not block later submits
      if (!drain_or_mark_stalled(ctx)) B.abandoned = true;
      else B.active = false;
      return fail(ctx, MSM_AMD_PIPELINE_ERROR, std::string("hipEventQuery: ") + hipGetErrorString(done[i].err));
    }
    accumulate_timings(ctx, B.slots[i], B.plans[i], done[i].final_ms, B.n_inst);
  }
  B.active = false;
  reap_graveyard(ctx);
  return MSM_AMD_OK;
}

// Is this host pointer page-locked (msm_amd_host_register, hipHostMalloc, hipHostRegister)?
bool host_pinned(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

// Pageable uploads of one range overlap the kernels of the previous range only if the runtime's staged copy does not
// serialise with the device.  Measured on this image: the system runtime (HIP 7.2) overlaps them (2^22 host points:
// 14.0 -> 9.6 ms with 4 ranges); the HIP 7.0 runtime bundled with the PyTorch wheel -- which the process binds to when
// torch is imported first, as in bench.py -- does not (14.0 -> 14.9 ms).  Page-locked buffers overlap on both.
bool pageable_uploads_overlap() {
  static const bool ok = [] {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    return v >= 70200000;
  }();
  return ok;
}

void stager_worker(Stager* S, int part) {
  uint64_t seen = 0;
  for (;;) {
    const uint8_t* src;
    uint8_t* dst;
    size_t bytes;
    // chunks of one upload follow each other within ~100 us: spin that long before going to sleep on the condvar
    for (int spin = 0; spin < 20000 && S->generation.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
    {
      std::unique_lock<std::mutex> lk(S->m);
      S->wake.wait(lk, [&] { return S->quit || S->generation.load() != seen; });
      if (S->quit) return;
      seen = S->generation.load();
      src = S->src;
      dst = S->dst;
      bytes = S->bytes;
    }
    const size_t parts = (size_t)S->n_helpers + 1, lo = bytes * part / parts, hi = bytes * (part + 1) / parts;
    std::memcpy(dst + lo, src + lo, hi - lo);
    {
      std::lock_guard<std::mutex> lk(S->m);
      if (--S->pending == 0) S->finished.notify_one();
    }
  }
}

int stager_prepare(msm_amd_ctx* ctx) {
  Stager& S = ctx->stager;
  if (S.ready) return MSM_AMD_OK;
  if (int rc = quiesce_for_allocation(ctx, "the page-locked staging ring")) return rc;
  for (int k = 0; k < Stager::kSlots; ++k) {
    HIP_TRY(ctx, hipHostMalloc(&S.slot[k], Stager::kChunk, hipHostMallocDefault));
    HIP_TRY(ctx, hipEventCreateWithFlags(&S.sent[k], hipEventDisableTiming));
  }
  if (const char* e = std::getenv("MSM_AMD_STAGE_HELPERS")) S.n_helpers = std::max(1, std::min(Stager::kMaxHelpers, std::atoi(e)));
  for (int h = 1; h <= S.n_helpers; ++h) S.helpers.emplace_back(stager_worker, &S, h);
  S.ready = true;
  return MSM_AMD_OK;
}

void stager_shutdown(msm_amd_ctx* ctx) {
  Stager& S = ctx->stager;
  if (!S.ready) return;
  {
    std::lock_guard<std::mutex> lk(S.m);
    S.quit = true;
  }
  S.wake.notify_all();
  for (std::thread& t : S.helpers) t.join();
  S.helpers.clear();
  for (int k = 0; k < Stager::kSlots; ++k) {
    if (S.sent[k]) (void)hipEventDestroy(S.sent[k]);
    if (S.slot[k]) (void)hipHostFree(S.slot[k]);
    S.sent[k] = nullptr;
    S.slot[k] = nullptr;
  }
  S.ready = false;
}

// Should a pageable upload of this size go through the ring?  By default where the runtime's own staged copy does not
// overlap kernels (HIP < 7.2); MSM_AMD_STAGED_UPLOAD=0 / 1 overrides.
bool stage_pageable(msm_amd_ctx* ctx, const void* src, size_t bytes) {
  if (bytes < 2 * Stager::kChunk) return false;
  if (ctx->staged_uploads < 0) {
    const char* e = std::getenv("MSM_AMD_STAGED_UPLOAD");
    ctx->staged_uploads = e ? (std::atoi(e) != 0) : (pageable_uploads_overlap() ? 0 : 1);
  }
  return ctx->staged_uploads == 1 && !host_pinned(src);
}

// hipMemcpyAsync(dst, src, bytes, H2D, stream) for a pageable src, through the ring.  Returns when the last chunk has
// been handed to the DMA engine (the caller's buffer is no longer needed), not when it has arrived.
int staged_upload(msm_amd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream) {
  int rc;
  if ((rc = stager_prepare(ctx))) return rc;
  Stager& S = ctx->stager;
  for (size_t off = 0; off < bytes; off += Stager::kChunk) {
    const size_t len = std::min(Stager::kChunk, bytes - off);
    const int k = S.next;
    S.next = (S.next + 1) % Stager::kSlots;
    if (S.busy[k]) {
      if (wait_event(S.sent[k], ctx->wait_timeout_ms) != hipSuccess)
        return fail(ctx, MSM_AMD_PIPELINE_ERROR, "timed out waiting for a staged upload chunk to leave its slot");
      S.busy[k] = false;
    }
    {
      std::lock_guard<std::mutex> lk(S.m);
      S.src = (const uint8_t*)h_src + off;
      S.dst = (uint8_t*)S.slot[k];
      S.bytes = len;
      S.pending = S.n_helpers;
      ++S.generation;
    }
    S.wake.notify_all();
    std::memcpy(S.slot[k], (const uint8_t*)h_src + off, len / ((size_t)S.n_helpers + 1));   // part 0 on this thread
    {
      std::unique_lock<std::mutex> lk(S.m);
      S.finished.wait(lk, [&] { return S.pending == 0; });
    }
    HIP_TRY(ctx, hipMemcpyAsync((uint8_t*)d_dst + off, S.slot[k], len, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipEventRecord(S.sent[k], stream));
    S.busy[k] = true;
  }
  return MSM_AMD_OK;
}

// A LONE call of many points runs as ONE PIPELINED BATCH of sub-instances over point ranges (the algebra of the
// reference's GPU + CPU split, msm.rs:385-419: the MSM of a union of point ranges is the sum of the MSMs).  Alone, an
// instance is a serial chain upload -> conversion / digits / sort -> accumulate -> reduction; as a batch, the front end
// (and, from host buffers, the upload) of range k + 1 overlaps the accumulate kernel of range k.  Costs: one window
// reduction and one host Horner pass per range (overlapped, except the last) and a final addition of `parts` points.
// Thresholds measured on MI355X (profiles/r02_lone_call_split.txt); MSM_AMD_SPLIT=<parts> forces a count, 1 disables.
unsigned split_parts(const msm_amd_ctx* ctx, int point_layout, size_t n, bool host_buffers, const void* scalars = nullptr,
                     const void* points = nullptr) {
  if (point_layout == MSM_AMD_POINT_TABLES || !lone_call(ctx, 1)) return 1;
  unsigned parts = 1;
  if (const char* e = std::getenv("MSM_AMD_SPLIT")) {
    parts = (unsigned)std::max(1, std::min(8, std::atoi(e)));
  } else {
    const uint32_t l = floor_log2(n);
    if (host_buffers) {   // the upload overlaps too
      const bool dev_points = point_layout == MSM_AMD_POINT_PREPARED;
      const bool pinned = scalars && host_pinned(scalars) && (dev_points || (points && host_pinned(points)));
      if (pinned || pageable_uploads_overlap()) parts = l >= 22 ? 8 : (l >= 20 ? 4 : (l >= 19 ? 2 : 1));
    } else {
      parts = l >= 24 ? 8 : (l == 23 ? 4 : 1);
    }
  }
  while (parts > 1 && n / parts < 4096) parts >>= 1;
  return parts;
}

// ---- bases cache ------------------------------------------------------------------------------------------------
inline uint64_t mix64(uint64_t h, uint64_t v) {
  h = (h ^ v) * 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}
inline uint64_t hash_record(uint64_t h, const uint8_t* rec, size_t pb) {
  for (size_t o = 0; o + 8 <= pb; o += 8) {
    uint64_t v;
    std::memcpy(&v, rec + o, 8);
    h = mix64(h, v);
  }
  return h;
}
size_t cache_phases(size_t n) { return std::max<size_t>(1, n / 1024); }
uint64_t hash_phase(const void* host, size_t n, size_t pb, size_t S, size_t phase) {
  uint64_t h = 0x243F6A8885A308D3ull ^ (uint64_t)n ^ ((uint64_t)phase << 40);
  const uint8_t* base = (const uint8_t*)host;
  for (size_t i = phase; i < n; i += S) h = hash_record(h, base + i * pb, pb);
  return h;
}

// Full verification: the array as kCacheChunks contiguous slices, each hashed sequentially, the slices spread over a
// few short-lived threads (a 64 MiB array: ~1.5 ms on four threads against ~12 ms on one).
constexpr size_t kCacheChunks = 64;
void hash_chunks(const void* host, size_t n, size_t pb, uint64_t* out) {
  const uint8_t* base = (const uint8_t*)host;
  auto one = [&](size_t c) {
    const size_t lo = n * c / kCacheChunks, hi = n * (c + 1) / kCacheChunks;
    uint64_t h = 0x13198A2E03707344ull ^ (uint64_t)n ^ ((uint64_t)c << 48);
    for (size_t i = lo; i < hi; ++i) h = hash_record(h, base + i * pb, pb);
    out[c] = h;
  };
  const size_t T = n * pb >= ((size_t)4 << 20) ? 4 : 1;
  std::vector<std::thread> th;
  try {
    for (size_t t = 1; t < T; ++t)
      th.emplace_back([&, t] { for (size_t c = t; c < kCacheChunks; c += T) one(c); });
  } catch (...) {
    for (std::thread& x : th) x.join();
    th.clear();
    for (size_t c = 0; c < kCacheChunks; ++c) one(c);
    return;
  }
  const size_t started = th.size() + 1;
  for (size_t c = 0; c < kCacheChunks; c += started) one(c);
  for (std::thread& x : th) x.join();
}

void bases_cache_drop(msm_amd_ctx* ctx, size_t k) {
  BasesCacheEntry& e = ctx->bases_cache[k];
  if (e.d_prepared) ctx->graveyard.push_back(e.d_prepared);   // hipFree waits for the whole device: later (reap_graveyard)
  ctx->bases_cache_bytes -= e.n * sizeof(AffPacked);
  ctx->bases_cache.erase(ctx->bases_cache.begin() + (long)k);
}

void bases_cache_clear(msm_amd_ctx* ctx) {
  while (!ctx->bases_cache.empty()) bases_cache_drop(ctx, ctx->bases_cache.size() - 1);
}

// Hit: the entry (checksums verified against the caller's memory).  A stale entry is dropped and counts as a miss.
BasesCacheEntry* bases_cache_lookup(msm_amd_ctx* ctx, const void* host, size_t n, int layout) {
  for (size_t k = 0; k < ctx->bases_cache.size(); ++k) {
    BasesCacheEntry& e = ctx->bases_cache[k];
    if (e.host != host || e.n != n || e.layout != layout) continue;
    const size_t pb = point_bytes(layout), S = e.phase_hash.size();
    bool same = hash_phase(host, n, pb, S, 0) == e.phase_hash[0];
    if (same && ctx->bases_cache_verify) {   // every record, every hit
      uint64_t now[kCacheChunks];
      hash_chunks(host, n, pb, now);
      same = std::memcmp(now, e.chunk_hash.data(), sizeof now) == 0;
    } else if (same && S > 1) {
      const uint32_t ph = e.next_phase;
      e.next_phase = ph + 1 >= S ? 1 : ph + 1;
      same = hash_phase(host, n, pb, S, ph) == e.phase_hash[ph];
    }
    if (!same) {
      if (e.last_use == ctx->bases_cache_call) return nullptr;   // in use by this very call: leave it, just do not hit
      ++ctx->bases_cache_invalidations;
      bases_cache_drop(ctx, k);
      return nullptr;
    }
    e.last_use = ctx->bases_cache_call;
    ++ctx->bases_cache_hits;
    return &e;
  }
  return nullptr;
}

// Start of a host-slice call: room on the device for those of its arrays that have no entry yet (least recently used
// entries that this call will not hit make way).  The ctx is idle here or is waited for, bounded; an array that gets
// no room runs uncached.
void bases_cache_reserve(msm_amd_ctx* ctx, size_t n_inst, const void* const* points, const size_t* n, int layout) {
  for (const msm_amd_ctx::CacheReserve& r : ctx->cache_reserve) ctx->graveyard.push_back(r.d);   // an earlier call's leftovers
  ctx->cache_reserve.clear();
  if (!ctx->bases_cache_budget) return;
  auto has_key = [&](size_t j) {
    for (const BasesCacheEntry& e : ctx->bases_cache)
      if (e.host == points[j] && e.n == n[j] && e.layout == layout) return true;
    for (const msm_amd_ctx::CacheReserve& r : ctx->cache_reserve)
      if (r.host == points[j] && r.n == n[j] && r.layout == layout) return true;
    return false;
  };
  for (size_t j = 0; j < n_inst; ++j)   // what this call is about to hit stays
    for (BasesCacheEntry& e : ctx->bases_cache)
      if (e.host == points[j] && e.n == n[j] && e.layout == layout) e.wanted_by = ctx->bases_cache_call;
  bool idle = false;
  size_t reserved_bytes = 0;
  for (size_t j = 0; j < n_inst; ++j) {
    if (!points[j] || has_key(j)) continue;
    const size_t bytes = n[j] * sizeof(AffPacked);
    if (bytes > ctx->bases_cache_budget) continue;
    bool room = true;
    while (ctx->bases_cache_bytes + reserved_bytes + bytes > ctx->bases_cache_budget) {
      size_t victim = ctx->bases_cache.size();
      for (size_t k = 0; k < ctx->bases_cache.size(); ++k)
        if (ctx->bases_cache[k].last_use != ctx->bases_cache_call && ctx->bases_cache[k].wanted_by != ctx->bases_cache_call &&
            (victim == ctx->bases_cache.size() || ctx->bases_cache[k].last_use < ctx->bases_cache[victim].last_use))
          victim = k;
      if (victim == ctx->bases_cache.size()) {
        room = false;
        break;
      }
      bases_cache_drop(ctx, victim);
    }
    if (!room) continue;
    if (!idle) {
      if (!streams_idle(ctx) && !drain_streams_bounded(ctx)) return;   // busy past the bound: this call runs uncached
      idle = true;
    }
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    ctx->cache_reserve.push_back({points[j], n[j], layout, d});
    reserved_bytes += bytes;
  }
}

// Miss: a new entry on the room bases_cache_reserve made for it, or nullptr -- the instance then runs uncached.
BasesCacheEntry* bases_cache_insert(msm_amd_ctx* ctx, const void* host, size_t n, int layout) {
  const size_t bytes = n * sizeof(AffPacked);
  for (size_t k = 0; k < ctx->bases_cache.size(); ++k)   // a stale twin the lookup had to leave alone
    if (ctx->bases_cache[k].host == host && ctx->bases_cache[k].n == n && ctx->bases_cache[k].layout == layout)
      return nullptr;
  BasesCacheEntry e;
  e.host = host;
  e.n = n;
  e.layout = layout;
  for (size_t k = 0; k < ctx->cache_reserve.size(); ++k)
    if (ctx->cache_reserve[k].host == host && ctx->cache_reserve[k].n == n && ctx->cache_reserve[k].layout == layout) {
      e.d_prepared = ctx->cache_reserve[k].d;
      ctx->cache_reserve.erase(ctx->cache_reserve.begin() + (long)k);
      break;
    }
  if (!e.d_prepared) return nullptr;
  const size_t pb = point_bytes(layout), S = cache_phases(n);
  e.phase_hash.assign(S, 0);
  for (size_t ph = 0; ph < S; ++ph) e.phase_hash[ph] = 0x243F6A8885A308D3ull ^ (uint64_t)n ^ ((uint64_t)ph << 40);
  const uint8_t* base = (const uint8_t*)host;
  size_t ph = 0;
  for (size_t i = 0; i < n; ++i) {   // one sequential pass fills every phase
    e.phase_hash[ph] = hash_record(e.phase_hash[ph], base + i * pb, pb);
    if (++ph == S) ph = 0;
  }
  e.chunk_hash.assign(kCacheChunks, 0);
  hash_chunks(host, n, pb, e.chunk_hash.data());
  e.last_use = ctx->bases_cache_call;
  ctx->bases_cache_bytes += bytes;
  ++ctx->bases_cache_misses;
  ctx->bases_cache.push_back(std::move(e));
  return &ctx->bases_cache.back();
}

int run_batch_host(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst, const void* const* scalars,
                   const void* const* points, const size_t* n, void* out);
int run_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                     const void* const* d_scalars, const void* const* d_points, const size_t* n, void* out_host);

int run_split(msm_amd_ctx* ctx, int scalar_layout, int point_layout, const void* scalars, const void* points, size_t n,
              unsigned parts, void* out, bool host_buffers) {
  const size_t sb = scalar_bytes(scalar_layout), pb = point_bytes(point_layout);
  std::vector<const void*> sp(parts), pp(parts);
  std::vector<size_t> nn(parts);
  std::vector<uint8_t> outs((size_t)parts * 96);
  for (unsigned k = 0; k < parts; ++k) {
    const size_t b = n * k / parts, e = n * (k + 1) / parts;
    sp[k] = (const uint8_t*)scalars + sb * b;
    pp[k] = (const uint8_t*)points + pb * b;
    nn[k] = e - b;
  }
  int rc;
  if (host_buffers) {
    rc = run_batch_host(ctx, scalar_layout, point_layout, parts, sp.data(), pp.data(), nn.data(), outs.data());
  } else {
    int ticket = -1;
    rc = submit_batch_device(ctx, scalar_layout, point_layout, parts, sp.data(), pp.data(), nn.data(), outs.data(),
                             &ticket, 0);
    if (!rc) rc = wait_batch(ctx, ticket);
  }
  if (rc) return rc;
  h64::Jac acc = h64::identity();
  for (unsigned k = 0; k < parts; ++k) {
    h64::Jac p;
    std::memcpy(&p, outs.data() + (size_t)k * 96, 96);
    acc = h64::jadd(acc, p);
  }
  const h64::Jac res = h64::normalise(acc);
  std::memcpy(out, &res, 96);
  // stage spans of the call = sums over its ranges (wait_batch / run_batch_host report per-instance averages)
  msm_amd_timings& T = ctx->timings;
  const float P = (float)parts;
  T.convert_ms *= P; T.digits_ms *= P; T.sort_ms *= P; T.accumulate_ms *= P; T.accumulate_kernel_ms *= P;
  T.reduce_ms *= P; T.final_ms *= P; T.total_gpu_ms *= P;
  T.n = (uint32_t)n;
  T.reserved = parts;
  return MSM_AMD_OK;
}

int run_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                     const void* const* d_scalars, const void* const* d_points, const size_t* n, void* out_host) {
  if (n_inst == 1 && ctx && d_scalars && d_points && n && d_scalars[0] && d_points[0]) {
    const unsigned parts = split_parts(ctx, point_layout, n[0], false);
    if (parts > 1)
      return run_split(ctx, scalar_layout, point_layout, d_scalars[0], d_points[0], n[0], parts, out_host, false);
  }
  int ticket = -1;
  int rc = submit_batch_device(ctx, scalar_layout, point_layout, n_inst, d_scalars, d_points, n, out_host, &ticket);
  if (rc) return rc;
  return wait_batch(ctx, ticket);
}

int upload_one(msm_amd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, bool ring = true) {
  if (ring && stage_pageable(ctx, h_src, bytes)) return staged_upload(ctx, d_dst, h_src, bytes, ctx->copy_stream);
  HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  return MSM_AMD_OK;
}

void uploader_main(msm_amd_ctx* ctx) {
  Uploader& U = ctx->uploader;
  (void)hipSetDevice(ctx->device);
  std::string my_error;
  tl_error_sink = &my_error;
  for (;;) {
    UploadJob j;
    {
      std::unique_lock<std::mutex> lk(U.m);
      U.cv.wait(lk, [&] { return U.quit || U.has_job; });
      if (U.quit) return;
      j = U.job;
      U.has_job = false;
    }
    int rc = MSM_AMD_OK;
    hipError_t e = hipSuccess;
    if (j.wait_for) e = hipStreamWaitEvent(ctx->copy_stream, j.wait_for, 0);
    if (e == hipSuccess) rc = upload_one(ctx, j.d_scalars, j.h_scalars, j.scalar_bytes);
    if (e == hipSuccess && !rc && j.point_bytes) rc = upload_one(ctx, j.d_points, j.h_points, j.point_bytes);
    if (e == hipSuccess && !rc) e = hipEventRecord(j.done, ctx->copy_stream);
    {
      std::lock_guard<std::mutex> lk(U.m);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        rc = MSM_AMD_PIPELINE_ERROR;
        U.error = std::string("upload: ") + hipGetErrorString(e);
      } else if (rc) {
        U.error = my_error;
      }
      U.status = rc;
      U.busy = false;
    }
    U.cv_idle.notify_all();
  }
}

void uploader_dispatch(msm_amd_ctx* ctx, const UploadJob& j) {
  Uploader& U = ctx->uploader;
  if (!U.started) {
    U.th = std::thread(uploader_main, ctx);
    U.started = true;
  }
  {
    std::lock_guard<std::mutex> lk(U.m);
    U.job = j;
    U.has_job = true;
    U.busy = true;
  }
  U.cv.notify_one();
}

int uploader_await(msm_amd_ctx* ctx) {   // the dispatched job has been enqueued completely (or has failed)
  Uploader& U = ctx->uploader;
  if (!U.started) return MSM_AMD_OK;
  std::unique_lock<std::mutex> lk(U.m);
  U.cv_idle.wait(lk, [&] { return !U.busy; });
  const int rc = U.status;
  U.status = MSM_AMD_OK;
  if (rc) ctx->last_error = U.error;
  return rc;
}

void uploader_shutdown(msm_amd_ctx* ctx) {
  Uploader& U = ctx->uploader;
  if (!U.started) return;
  (void)uploader_await(ctx);
  {
    std::lock_guard<std::mutex> lk(U.m);
    U.quit = true;
  }
  U.cv.notify_all();
  U.th.join();
  U.started = false;
  U.quit = false;
}

// Host-buffer variant: stage inputs into device scratch, then run the device path per instance.
int run_batch_host(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst, const void* const* scalars,
                   const void* const* points, const size_t* n, void* out) {
  if (!ctx || !scalars || !points || !n || !out || n_inst == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "null argument or empty batch");
  const size_t pb = point_bytes(point_layout);
  if (pb == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown point layout");
  // prepared bases / window tables live on the device already: points[i] is the device pointer / table handle and
  // only the scalars (32 bytes per point instead of 96) cross PCIe -- the repeated-SRS case
  const bool dev_points = point_layout == MSM_AMD_POINT_PREPARED || point_layout == MSM_AMD_POINT_TABLES;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = recover_if_stalled(ctx)) return rc;
  if (n_inst == 1 && scalars[0] && points[0] && n[0]) {   // a lone call of many points: pipelined point ranges
    const unsigned parts = split_parts(ctx, point_layout, n[0], true, scalars[0], points[0]);
    if (parts > 1) return run_split(ctx, scalar_layout, point_layout, scalars[0], points[0], n[0], parts, out, true);
  }
  for (size_t i = 0; i < n_inst; ++i) {
    if (n[i] == 0 || !scalars[i] || !points[i]) return fail(ctx, MSM_AMD_INPUT_ERROR, "n == 0 or null pointer");
    if (point_layout == MSM_AMD_POINT_PREPARED) {   // must really be device memory: a host array here would fault
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, points[i]) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(ctx, MSM_AMD_INPUT_ERROR,
                    "MSM_AMD_POINT_PREPARED takes the device array written by msm_amd_bases_upload / "
                    "msm_amd_bases_prepare_device, not a host buffer");
      }
    }
  }
  // Two staging sets and up to three instances in flight.  Uploads run on their own stream (DMA for buffers the
  // caller registered, msm_amd_host_register; the runtime's staged copy, which blocks this thread, for pageable
  // memory): the upload of instance i waits -- on the GPU, not on the host -- until the front end of instance
  // i - 2 has read the staging set (its EV_DIGITS event: conversion, convert_bases and digits are done), and the
  // front-end stream of instance i waits for the upload.  The host only blocks to collect instance i - 3.
  constexpr size_t kInflight = 3;
  static_assert(kInflight <= (size_t)kMaxBatches, "one ticket per instance in flight");
  std::vector<int> tickets(n_inst, -1);
  msm_amd_timings avg{};
  size_t collected = 0;
  auto collect = [&](size_t i) {   // wait for instance i and fold its stage times into the running average
    int rc = wait_batch(ctx, tickets[i]);
    tickets[i] = -1;
    if (rc) return rc;
    const msm_amd_timings& T = ctx->timings;
    const float a = (float)collected / (float)(collected + 1), b = 1.0f / (float)(collected + 1);
    msm_amd_timings k = T;
    k.convert_ms = avg.convert_ms * a + T.convert_ms * b;
    k.digits_ms = avg.digits_ms * a + T.digits_ms * b;
    k.sort_ms = avg.sort_ms * a + T.sort_ms * b;
    k.accumulate_ms = avg.accumulate_ms * a + T.accumulate_ms * b;
    k.accumulate_kernel_ms = avg.accumulate_kernel_ms * a + T.accumulate_kernel_ms * b;
    k.reduce_ms = avg.reduce_ms * a + T.reduce_ms * b;
    k.final_ms = avg.final_ms * a + T.final_ms * b;
    k.total_gpu_ms = avg.total_gpu_ms * a + T.total_gpu_ms * b;
    k.reserved = (uint32_t)(++collected);
    avg = k;
    return (int)MSM_AMD_OK;
  };
  auto bail = [&](int rc) {   // error exit: nothing of this call may stay in flight or hold a ticket
    ctx->convert_into = nullptr;
    for (BasesCacheEntry& c : ctx->bases_cache)   // an entry filled by this call may never have been written: no hits
      if (c.last_use == ctx->bases_cache_call) c.host = nullptr;
    if (ctx->stalled) {       // ... unless the device is not answering: leave the rest to recover_if_stalled
      for (int t : tickets)
        if (t >= 0) ctx->batches[t].abandoned = true;
      return rc;
    }
    if (!drain_or_mark_stalled(ctx)) {
      for (int t : tickets)
        if (t >= 0) ctx->batches[t].abandoned = true;
      return rc;
    }
    for (int t : tickets)
      if (t >= 0) ctx->batches[t].active = false;
    return rc;
  };
  {   // staging buffers are sized once, before anything is in flight (growing one would free memory in use)
    size_t max_n = 0;
    for (size_t i = 0; i < n_inst; ++i) max_n = std::max(max_n, n[i]);
    int rc;
    for (DeviceBuf* b : {&ctx->scratch_b, &ctx->scratch_b2})
      if ((rc = ensure(ctx, *b, max_n * scalar_bytes(scalar_layout)))) return rc;
    if (!dev_points)
      for (DeviceBuf* b : {&ctx->scratch_c, &ctx->scratch_c2})
        if ((rc = ensure(ctx, *b, max_n * pb))) return rc;
  }
  const bool lone = lone_call(ctx, n_inst);
  hipStream_t fs = front_stream_of(ctx, lone);
  const bool use_cache = ctx->bases_cache_budget != 0 && !dev_points;
  if (use_cache) bases_cache_reserve(ctx, n_inst, points, n, point_layout);
  // bases cache (opt-in): a hit replaces the 64..96 B per point upload and the conversion by the resident converted
  // copy; a miss uploads as usual and converts straight into the new entry (the reference re-uploads and re-converts
  // the bases on every call, msm.rs:152-153; its callers pass the same SRS slice again and again,
  // benches/msm_benchmark.rs:116-121)
  std::vector<const void*> cached(n_inst, nullptr);
  std::vector<AffPacked*> fill(n_inst, nullptr);
  // upload of instance j on the helper thread (uploader_main), decided and dispatched by this one
  auto dispatch = [&](size_t j) -> int {
    if (use_cache) {
      if (BasesCacheEntry* e = bases_cache_lookup(ctx, points[j], n[j], point_layout)) cached[j] = e->d_prepared;
      else if (BasesCacheEntry* f = bases_cache_insert(ctx, points[j], n[j], point_layout)) fill[j] = (AffPacked*)f->d_prepared;
    }
    UploadJob job;
    job.d_scalars = ((j & 1) ? ctx->scratch_b2 : ctx->scratch_b).p;
    job.h_scalars = scalars[j];
    job.scalar_bytes = n[j] * scalar_bytes(scalar_layout);
    if (!dev_points && !cached[j]) {
      job.d_points = ((j & 1) ? ctx->scratch_c2 : ctx->scratch_c).p;
      job.h_points = points[j];
      job.point_bytes = n[j] * pb;
    }
    // the staging set's previous reader is instance j - 2: its front end has read the set once its digits are done
    if (j >= 2) job.wait_for = ctx->batches[tickets[j - 2]].slots[0].ev[EV_DIGITS];
    job.done = ctx->uploaded[j & 1];
    if (n_inst > 1) {
      uploader_dispatch(ctx, job);
      return (int)MSM_AMD_OK;
    }
    // ONE instance: nothing to overlap the upload with -- the plain copy on this thread is the shortest path (the
    // helper thread's hand-offs and the ring cost a lone 2^18 call 0.2 ms, measured)
    int rc = upload_one(ctx, job.d_scalars, job.h_scalars, job.scalar_bytes, false);
    if (!rc && job.point_bytes) rc = upload_one(ctx, job.d_points, job.h_points, job.point_bytes, false);
    if (!rc && hipEventRecord(job.done, ctx->copy_stream) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(ctx, MSM_AMD_PIPELINE_ERROR, "hipEventRecord(uploaded)");
    }
    return rc;
  };
  auto bail_upload = [&](int rc) {   // the helper may still be reading the caller's buffers
    (void)uploader_await(ctx);
    return bail(rc);
  };
  if (int rc0 = dispatch(0)) return bail(rc0);
  for (size_t i = 0; i < n_inst; ++i) {
    int rc;
    if (i >= kInflight && (rc = collect(i - kInflight))) return bail_upload(rc);
    if (n_inst > 1 && (rc = uploader_await(ctx))) return bail(rc);   // upload i is enqueued, uploaded[i & 1] recorded
    // upload i + 1 goes on while this thread enqueues the kernels of instance i (its staging set was last read by
    // instance i - 1, submitted in the previous iteration)
    if (i + 1 < n_inst) (void)dispatch(i + 1);
    const hipError_t e = hipStreamWaitEvent(fs, ctx->uploaded[i & 1], 0);
    if (e != hipSuccess) return bail_upload(fail(ctx, MSM_AMD_PIPELINE_ERROR, hipGetErrorString(e)));
    const void* ds = ((i & 1) ? ctx->scratch_b2 : ctx->scratch_b).p;
    const void* dp = dev_points ? points[i] : (cached[i] ? cached[i] : ((i & 1) ? ctx->scratch_c2 : ctx->scratch_c).p);
    int ticket = -1;
    ctx->convert_into = fill[i];
    rc = submit_batch_device(ctx, scalar_layout, cached[i] ? (int)MSM_AMD_POINT_PREPARED : point_layout, 1, &ds, &dp, &n[i],
                             (uint8_t*)out + i * 96, &ticket, lone ? 1 : 0);
    ctx->convert_into = nullptr;
    if (rc) return bail_upload(rc);
    tickets[i] = ticket;
  }
  for (size_t i = n_inst > kInflight ? n_inst - kInflight : 0; i < n_inst; ++i) {
    int rc = collect(i);
    if (rc) return bail(rc);
  }
  ctx->timings = avg;
  return MSM_AMD_OK;
}

// Host -> device copy of a single-instance entry point's inputs into ctx scratch.  The copy goes on the stream a lone
// call's front end uses (no cross-stream hand-off in the common case); enqueue_msm orders it before the front end
// if that ends up on another stream -- a call of many points becomes pipelined point ranges (run_split), whose front
// ends run on the front stream.  With pageable memory the runtime's staged copy blocks the host and hid the missing
// edge; with page-locked scalars (msm_amd_host_register) digits_kernel could read the scratch before the DMA landed.
int stage_upload(msm_amd_ctx* ctx, void* dst, const void* src, size_t bytes) {
  hipStream_t up = front_stream_of(ctx, lone_call(ctx, 1));
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, up));
  ctx->upload_stream = up;
  ctx->upload_pending = true;
  return MSM_AMD_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* msm_amd_version(void) {
#if defined(MSM_AMD_EXPERIMENTS)
  return "msm_amd 0.4 (gfx950) +experiments";
#else
  return "msm_amd 0.4 (gfx950)";
#endif
}

const char* msm_amd_strerror(int status) {
  switch (status) {
    case MSM_AMD_OK: return "ok";
    case MSM_AMD_DEVICE_NOT_FOUND: return "Couldn't find a HIP device (MetalError::DeviceNotFound)";
    case MSM_AMD_LIBRARY_ERROR: return "Couldn't load the gfx950 code object (MetalError::LibraryError)";
    case MSM_AMD_FUNCTION_ERROR: return "Couldn't set up a kernel function (MetalError::FunctionError)";
    case MSM_AMD_PIPELINE_ERROR: return "HIP launch/runtime failure (MetalError::PipelineError)";
    case MSM_AMD_INPUT_ERROR: return "Invalid input (MetalError::InputError)";
    case MSM_AMD_FILE_OPEN_ERROR: return "I/O error (HarnessError::FileOpenError)";
    case MSM_AMD_DESERIALIZATION_ERROR: return "Data in file is invalid or incomplete (HarnessError::DeserializationError)";
    case MSM_AMD_INVALID_DATA: return "Invalid data (HarnessError::InvalidData)";
  }
  return "unknown status";
}

const char* msm_amd_last_error(const msm_amd_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int msm_amd_init(int device, msm_amd_ctx** out) {
  if (!out) return MSM_AMD_INPUT_ERROR;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return MSM_AMD_DEVICE_NOT_FOUND;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return MSM_AMD_DEVICE_NOT_FOUND;
  }
  if (device >= count) return MSM_AMD_DEVICE_NOT_FOUND;
  if (hipSetDevice(device) != hipSuccess) return MSM_AMD_DEVICE_NOT_FOUND;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MSM_AMD_DEVICE_NOT_FOUND;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    // the code object is built for gfx950 only: fail loudly instead of falling back
    std::fprintf(stderr, "msm_amd: device %d is %s, this library targets gfx950 (MI355X)\n", device,
                 prop.gcnArchName);
    return MSM_AMD_LIBRARY_ERROR;
  }
  msm_amd_ctx* ctx = new msm_amd_ctx();
  ctx->device = device;
  if (const char* e = std::getenv("MSM_AMD_OVERLAP_REDUCE")) ctx->overlap_reduce = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSM_AMD_OVERLAP_FRONT")) ctx->overlap_front = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSM_AMD_ALT_REDUCE")) ctx->alt_reduce = std::atoi(e) != 0;
  if (const char* e = std::getenv("MSM_AMD_LONE_SINGLE_STREAM")) ctx->lone_single_stream = std::atoi(e) != 0;
  ctx->acc_variant = ctx->overlap_front ? 1 : 0;
  if (const char* e = std::getenv("MSM_AMD_LOW_OCC")) ctx->acc_variant = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("MSM_AMD_ACC_VARIANT")) ctx->acc_variant = std::atoi(e);
  if (const char* e = std::getenv("MSM_AMD_ACC_LDS")) ctx->acc_lds = (uint32_t)std::atoi(e);
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  // the short front-end / reduction kernels get priority over the long accumulate grid
  // CREATION ORDER MATTERS: HIP maps streams onto four hardware queues in creation order, so the fifth stream (copy,
  // idle in device-resident runs) shares the queue of the first (main).  With the reduce streams created last, one
  // of them shared the accumulate grid's queue and every second window reduction waited behind it: 760 -> 720 MSM/s
  // on the same box (profiles/r02_stream_creation_order_ab.txt).  The copy stream is best off on the main stream's
  // queue as well: on a reduce stream's queue the host-slice batches lose 4 %, on the front stream's 18 %.
  bool ok = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_least) == hipSuccess;
  for (int k = 0; ok && k < kReduceStreams; ++k)
    ok = hipStreamCreateWithPriority(&ctx->reduce_streams[k], hipStreamNonBlocking, prio_greatest) == hipSuccess;
  ok = ok &&
            hipStreamCreateWithPriority(&ctx->front_stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->uploaded[0], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->uploaded[1], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming) == hipSuccess;
  ctx->wait_timeout_ms = default_wait_timeout_ms();
  if (const char* e = std::getenv("MSM_AMD_BASES_CACHE_VERIFY")) ctx->bases_cache_verify = std::strcmp(e, "full") == 0;
  if (const char* e = std::getenv("MSM_AMD_BASES_CACHE_MB")) {
    ctx->bases_cache_budget = (size_t)std::strtoull(e, nullptr, 10) << 20;
    if (ctx->bases_cache_budget)   // switched on from OUTSIDE the calling code: say so, once per ctx
      std::fprintf(stderr,
                   "libmsm_amd: MSM_AMD_BASES_CACHE_MB=%s: host-slice entry points now reuse converted bases by (pointer, "
                   "n, layout); a hit is verified %s -- callers that change bases in place must call "
                   "msm_amd_bases_cache_invalidate\n",
                   e, ctx->bases_cache_verify ? "in full (every record)" : "by sampling (MSM_AMD_BASES_CACHE_VERIFY=full re-hashes every record)");
  }
  for (int k = 0; ok && k < kWorkspaces; ++k)
    ok = hipEventCreateWithFlags(&ctx->ws[k].front_done, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ctx->ws[k].acc_done, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ctx->ws[k].reduce_done, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    delete ctx;
    return MSM_AMD_PIPELINE_ERROR;
  }
  int rc = set_kernel_attributes(ctx);
  if (rc) {
    std::fprintf(stderr, "msm_amd: %s\n", ctx->last_error.c_str());
    (void)hipStreamDestroy(ctx->stream);
    for (hipStream_t rs : ctx->reduce_streams)
      if (rs) (void)hipStreamDestroy(rs);
    (void)hipStreamDestroy(ctx->front_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
    (void)hipEventDestroy(ctx->uploaded[0]);
    (void)hipEventDestroy(ctx->uploaded[1]);
    (void)hipEventDestroy(ctx->upload_done);
    delete ctx;
    return rc;
  }
  *out = ctx;
  return MSM_AMD_OK;
}

int msm_amd_init_reusable(msm_amd_ctx** out) {
  if (!out) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(g_global_mu);
  if (!g_global_ctx) {
    int rc = msm_amd_init(-1, &g_global_ctx);
    if (rc) return rc;
  }
  *out = g_global_ctx;
  return MSM_AMD_OK;
}

int msm_amd_get_global(msm_amd_ctx** out) {
  if (!out) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(g_global_mu);
  if (!g_global_ctx) return MSM_AMD_INPUT_ERROR;
  *out = g_global_ctx;
  return MSM_AMD_OK;
}

// Teardown order: wait until the DEVICE is idle (every stream of the ctx, the copy stream included, and whatever the
// runtime still has queued: a profiler's own work), then events, then memory, then streams, every handle nulled.
// (Round 2 freed buffers and events stream by stream without ever synchronising the copy stream; under rocprofv3 a
// profiled 2^24-point run once stalled and once aborted AFTER its last kernel had completed, i.e. in here / at exit.)
void msm_amd_destroy(msm_amd_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> g(g_global_mu);
    if (ctx == g_global_ctx) g_global_ctx = nullptr;
  }
  std::unique_lock<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  if (!drain_streams_bounded(ctx)) {
    // the device does not answer: freeing memory under work in flight would turn a stall into a fault.  Give the
    // device resources up (they go with the process) and release the host side only.
    std::fprintf(stderr, "msm_amd_destroy: device %d still busy after %u ms; leaking the ctx's device resources\n",
                 ctx->device, ctx->wait_timeout_ms);
    lk.unlock();
    return;
  }
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  auto kill_event = [](hipEvent_t& e) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  };
  auto kill_buf = [](DeviceBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  };
  auto kill_stream = [](hipStream_t& s) {
    if (s) (void)hipStreamDestroy(s);
    s = nullptr;
  };
  // 1. events
  for (Batch& B : ctx->batches)
    for (InstanceSlot& s : B.slots) {
      if (s.has_events)
        for (int i = 0; i < EV_COUNT; ++i) kill_event(s.ev[i]);
      s.has_events = false;
    }
  for (int k = 0; k < kWorkspaces; ++k) {
    kill_event(ctx->ws[k].front_done);
    kill_event(ctx->ws[k].acc_done);
    kill_event(ctx->ws[k].reduce_done);
  }
  for (hipEvent_t& e : ctx->uploaded) kill_event(e);
  kill_event(ctx->upload_done);
  kill_event(ctx->after_sort_mark);
  uploader_shutdown(ctx);
  stager_shutdown(ctx);
  // 2. memory
  for (int k = 0; k < kWorkspaces; ++k) {
    Workspace& w = ctx->ws[k];
    DeviceBuf* bufs[] = {&w.digits, &w.coarse_cnt, &w.region_start, &w.tmp_idx, &w.tmp_fine, &w.tmp_idx2, &w.tmp_fine2,
                         &w.mid_cnt, &w.region_start2, &w.bsize, &w.bstart, &w.istart, &w.win_items, &w.size_bins,
                         &w.sorted, &w.order, &w.multi_list, &w.redo_list, &w.counters, &w.bases29, &w.buckets, &w.item_partials,
                         &w.S, &w.T, &w.partial, &w.conv_scalars, &w.conv_points, &w.conv_tmp};
    for (DeviceBuf* b : bufs) kill_buf(*b);
  }
  for (msm_amd_tables* t : ctx->live_tables) {   // tables the caller did not free
    (void)hipFree(t->d_tables);
    delete t;
  }
  ctx->live_tables.clear();
  for (DeviceBuf* b : {&ctx->scratch_a, &ctx->scratch_b, &ctx->scratch_c, &ctx->scratch_b2, &ctx->scratch_c2}) kill_buf(*b);
  bases_cache_clear(ctx);   // (entries and unused reserves go through the graveyard)
  for (const msm_amd_ctx::CacheReserve& r : ctx->cache_reserve) ctx->graveyard.push_back(r.d);
  ctx->cache_reserve.clear();
  for (void* p : ctx->graveyard) (void)hipFree(p);
  ctx->graveyard.clear();
  for (Batch& B : ctx->batches) {
    for (InstanceSlot& s : B.slots) {
      if (s.h_partial) (void)hipHostFree(s.h_partial);
      s.h_partial = nullptr;
      s.h_partial_cap = 0;
    }
    B.slots.clear();
    B.active = B.abandoned = false;
  }
  for (auto& r : ctx->host_regs) (void)hipHostUnregister(const_cast<void*>(r.first));   // left registered by the caller
  ctx->host_regs.clear();
  (void)hipGetLastError();
  // 3. streams
  kill_stream(ctx->stream);
  for (hipStream_t& rs : ctx->reduce_streams) kill_stream(rs);
  kill_stream(ctx->front_stream);
  kill_stream(ctx->copy_stream);
  (void)hipGetLastError();
  lk.unlock();
  delete ctx;
}

// Page-locks a caller buffer for the life of the registration, so that the host-buffer entry points upload it by
// DMA at PCIe rate and without blocking the calling thread (pageable memory is staged by the runtime at roughly
// half that rate, on the calling thread).  Meant for long-lived inputs -- the bases of an SRS -- exactly the data
// the reference re-uploads on every call (msm.rs:152-153).  The caller keeps the memory valid and in place until
// msm_amd_host_unregister (or msm_amd_destroy).
int msm_amd_host_register(msm_amd_ctx* ctx, const void* ptr, size_t bytes) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  if (!ptr || bytes == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "null pointer or empty range");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (auto& r : ctx->host_regs)
    if (r.first == ptr) return fail(ctx, MSM_AMD_INPUT_ERROR, "range already registered");
  HIP_TRY(ctx, hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault));
  ctx->host_regs.emplace_back(ptr, bytes);
  return MSM_AMD_OK;
}

int msm_amd_host_unregister(msm_amd_ctx* ctx, const void* ptr) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  for (size_t i = 0; i < ctx->host_regs.size(); ++i)
    if (ctx->host_regs[i].first == ptr) {
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      if (!drain_or_mark_stalled(ctx))   // DMA from this buffer may still be in flight
        return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy past the wait bound: the buffer stays registered");
      HIP_TRY(ctx, hipHostUnregister(const_cast<void*>(ptr)));
      ctx->host_regs.erase(ctx->host_regs.begin() + (long)i);
      return MSM_AMD_OK;
    }
  return fail(ctx, MSM_AMD_INPUT_ERROR, "range was not registered on this ctx");
}

int msm_amd_set_window_size(msm_amd_ctx* ctx, uint32_t window_size) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  if (window_size != 0 && (window_size < kMinWindow || window_size > kMaxWindow))
    return fail(ctx, MSM_AMD_INPUT_ERROR, "window_size must be 0 (auto) or 3..17");
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->forced_window = window_size;
  return MSM_AMD_OK;
}

uint32_t msm_amd_auto_window_size(size_t n) { return auto_window(n); }
uint32_t msm_amd_auto_window_size_lone(size_t n) { return auto_window_lone(n); }

int msm_amd_msm_batch(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                      const void* const* scalars, const void* const* points, const size_t* n, void* out) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  ++ctx->bases_cache_call;
  return run_batch_host(ctx, scalar_layout, point_layout, n_inst, scalars, points, n, out);
}

int msm_amd_set_bases_cache(msm_amd_ctx* ctx, size_t max_bytes) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_streams_bounded(ctx)) return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy: bases cache left as it was");
  ctx->bases_cache_budget = max_bytes;
  ++ctx->bases_cache_call;
  while (ctx->bases_cache_bytes > max_bytes && !ctx->bases_cache.empty()) {   // least recently used first
    size_t victim = 0;
    for (size_t k = 1; k < ctx->bases_cache.size(); ++k)
      if (ctx->bases_cache[k].last_use < ctx->bases_cache[victim].last_use) victim = k;
    bases_cache_drop(ctx, victim);
  }
  return MSM_AMD_OK;
}

int msm_amd_set_bases_cache_verify(msm_amd_ctx* ctx, int full) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->bases_cache_verify = full ? 1 : 0;
  return MSM_AMD_OK;
}

// The caller's word that the array at host_points changed (or NULL: everything): its entries are dropped.
int msm_amd_bases_cache_invalidate(msm_amd_ctx* ctx, const void* host_points) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  for (size_t k = ctx->bases_cache.size(); k-- > 0;)
    if (!host_points || ctx->bases_cache[k].host == host_points) {
      ++ctx->bases_cache_invalidations;
      bases_cache_drop(ctx, k);   // the device copy goes to the graveyard: safe under work in flight
    }
  return MSM_AMD_OK;
}

int msm_amd_bases_cache_stats(msm_amd_ctx* ctx, uint64_t out[5]) {
  if (!ctx || !out) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  out[0] = ctx->bases_cache_hits;
  out[1] = ctx->bases_cache_misses;
  out[2] = ctx->bases_cache_invalidations;
  out[3] = ctx->bases_cache_bytes;
  out[4] = ctx->bases_cache.size();
  return MSM_AMD_OK;
}

int msm_amd_msm(msm_amd_ctx* ctx, int scalar_layout, int point_layout, const void* scalars, const void* points,
                size_t n, void* out96) {
  return msm_amd_msm_batch(ctx, scalar_layout, point_layout, 1, &scalars, &points, &n, out96);
}

int msm_amd_gpu_msm_h2c(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, void* out96) {
  return msm_amd_msm(ctx, MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, scalars, points, n, out96);
}

int msm_amd_gpu_msm_h2c_sync(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n,
                             msm_amd_after_sort_fn after_sort, void* user, void* out96) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  if (!scalars || !points || !out96 || n == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "n == 0 or null pointer");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->scratch_b, n * 32))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, n * 64))) return rc;
  if ((rc = recover_if_stalled(ctx))) return rc;
  if ((rc = stage_upload(ctx, ctx->scratch_b.p, scalars, n * 32))) return rc;
  if ((rc = stage_upload(ctx, ctx->scratch_c.p, points, n * 64))) return rc;
  const void* ds = ctx->scratch_b.p;
  const void* dp = ctx->scratch_c.p;
  int ticket = -1;
  rc = submit_batch_device(ctx, MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, 1, &ds, &dp, &n, out96, &ticket);
  ctx->upload_pending = false;
  if (rc) return rc;
  ctx->after_sort_state = -1.0f;
  if (after_sort) {
    InstanceSlot& s = ctx->batches[ticket].slots[0];
    const hipError_t e = wait_event(s.ev[EV_SORT], ctx->wait_timeout_ms);   // sorted indices of this MSM exist (msm.rs:306-312)
    if (e == hipErrorNotReady) {
      ctx->stalled = true;
      ctx->batches[ticket].abandoned = true;
      return fail(ctx, MSM_AMD_PIPELINE_ERROR,
                  "timed out after " + std::to_string(ctx->wait_timeout_ms) + " ms waiting for event 'sort' of the MSM");
    }
    if (e != hipSuccess) {
      if (!drain_or_mark_stalled(ctx)) ctx->batches[ticket].abandoned = true;
      else ctx->batches[ticket].active = false;
      return fail(ctx, MSM_AMD_PIPELINE_ERROR, std::string("hipEventSynchronize(sort): ") + hipGetErrorString(e));
    }
    // a GPU-clock mark of "now" on a stream that has nothing queued: after the MSM has finished, the device time
    // from this mark to the end of bucket accumulation tells whether the callback ran while the GPU was still
    // accumulating (positive) -- host-side event queries race with the runtime's own polling
    if (!ctx->after_sort_mark) HIP_TRY(ctx, hipEventCreate(&ctx->after_sort_mark));
    HIP_TRY(ctx, hipEventRecord(ctx->after_sort_mark, ctx->copy_stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->after_sort_mark));   // the runtime submits lazily: make the mark real now
    after_sort(user);
    if (wait_event(s.ev[EV_ACC], ctx->wait_timeout_ms) == hipErrorNotReady) {
      ctx->stalled = true;
      ctx->batches[ticket].abandoned = true;
      return fail(ctx, MSM_AMD_PIPELINE_ERROR, "timed out after " + std::to_string(ctx->wait_timeout_ms) +
                                                   " ms waiting for event 'accumulate' of the MSM");
    }
    float lead = 0.0f;
    if (hipEventElapsedTime(&lead, ctx->after_sort_mark, s.ev[EV_ACC]) != hipSuccess) {
      (void)hipGetLastError();
      lead = 0.0f;
    }
    ctx->after_sort_state = lead > 0.0f ? 1.0f : 0.0f;
    ctx->after_sort_lead_ms = lead;
  }
  rc = wait_batch(ctx, ticket);
  ctx->after_sort_state = -1.0f;
  ctx->after_sort_lead_ms = 0.0f;
  return rc;
}

int msm_amd_metal_msm_ark(msm_amd_ctx* ctx, const void* points, const void* scalars, size_t n, void* out96) {
  return msm_amd_msm(ctx, MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_ARK_PROJECTIVE, scalars, points, n, out96);
}

// ---- hybrid front-end (src/metal/msm.rs:366-507) ---------------------------------------------------------
static size_t cpu_dispatch_below_fwd();
size_t msm_amd_reference_split(size_t n) {
  // gpu_with_cpu's split_at (msm.rs:377-383): the GPU gets the first n/3, n/2 or 2n/3 points
  if (n < ((size_t)1 << 18)) return n / 3;
  if (n < ((size_t)1 << 20)) return n / 2;
  return n * 2 / 3;
}

// The split measured on MI355X (tools/crossover.py, profiles/r03_crossover.txt): one GPU runs 2^20 points from host
// slices in ~3 ms, the host cores of a 16-thread grant need that long for ~2^14 points, so any share given to the CPU
// beyond msm_best's tiny-instance dispatch makes the call slower: everything goes to the GPU.
size_t msm_amd_tuned_split(size_t n) { return n < cpu_dispatch_below_fwd() ? 0 : n; }

int msm_amd_gpu_with_cpu(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, size_t split_at,
                         int cpu_threads, void* out96) {
  if (!ctx || !scalars || !points || !out96 || n == 0 || split_at > n)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad gpu_with_cpu arguments");
  if (cpu_threads <= 0) cpu_threads = msm_amd_host_threads();
  const u256* sc = (const u256*)scalars;
  const Affine* pt = (const Affine*)points;
  // CPU share on host threads (the reference waits for the GPU sort first because its sort borrows the CPU,
  // msm.rs:403-415; here the host cores are free from the start)
  Jacobian cpu_res = jac_identity();
  std::thread cpu_thread;
  const size_t n_cpu = n - split_at;
  bool cpu_ok = true;
  if (n_cpu) cpu_thread = std::thread([&] { cpu_res = host_msm(sc + split_at, 1, pt + split_at, n_cpu, cpu_threads, &cpu_ok); });
  Jacobian gpu_res = jac_identity();
  int rc = MSM_AMD_OK;
  if (split_at) {
    uint8_t buf[96];
    rc = msm_amd_msm(ctx, MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, scalars, points, split_at, buf);
    if (rc == MSM_AMD_OK) std::memcpy(&gpu_res, buf, 96);
  }
  if (n_cpu) cpu_thread.join();
  if (rc) return rc;
  if (!cpu_ok) return fail(ctx, MSM_AMD_PIPELINE_ERROR, "gpu_with_cpu: host allocation failed in the CPU half");
  const Jacobian sum = normalise(jac_add(gpu_res, cpu_res));   // msm.rs:418-419
  std::memcpy(out96, &sum, 96);
  return MSM_AMD_OK;
}

// msm_best's size dispatch (msm.rs:440-444: `if n < 2^17 { cpu } else { gpu }`), threshold measured on MI355X with
// tools/crossover.py (profiles/r03_crossover.txt; round 2: r02_crossover.txt, 5 points): one blocking GPU call costs
// 0.28-0.38 ms of launches, event waits and host Horner whatever n is; the single-threaded host bucket method is faster
// than that up to 16 points (0.34 against 0.38 ms).
static size_t cpu_dispatch_below() {
  static const size_t v = [] {
    if (const char* e = std::getenv("MSM_AMD_CPU_BELOW")) return (size_t)std::strtoull(e, nullptr, 10);
    return (size_t)kCpuDispatchBelow;
  }();
  return v;
}

static size_t cpu_dispatch_below_fwd() { return cpu_dispatch_below(); }
size_t msm_amd_cpu_dispatch_below(void) { return cpu_dispatch_below(); }

int msm_amd_msm_best(msm_amd_ctx* ctx, const void* scalars, const void* points, size_t n, void* out96) {
  if (!ctx || !scalars || !points || !out96 || n == 0 || n > 0x7FFFFFFFull)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad msm_best arguments");
  if (n < cpu_dispatch_below()) {   // explicit size dispatch as in the reference, on a ctx that owns a GPU
    bool ok = true;
    const Jacobian r = normalise(host_msm((const u256*)scalars, 1, (const Affine*)points, n, 1, &ok));
    if (!ok) return fail(ctx, MSM_AMD_PIPELINE_ERROR, "msm_best: host allocation failed");
    std::memcpy(out96, &r, 96);
    return MSM_AMD_OK;
  }
  std::unique_lock<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = recover_if_stalled(ctx))) return rc;
  const uint32_t nblocks = (uint32_t)((n + 1023) / 1024);
  if ((rc = ensure(ctx, ctx->scratch_b, n * 32))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, n * 64))) return rc;
  // filter_zeros (msm.rs:448-507): count the zero scalars on the device as soon as the scalars have arrived (the
  // points upload runs behind it), compact only if >= 30 % were zero (msm.rs:470)
  if ((rc = ensure(ctx, ctx->scratch_a, (((size_t)nblocks + 1) * 4 + 63) / 64 * 64 + n * 96))) return rc;
  uint32_t* counts = (uint32_t*)ctx->scratch_a.p;
  u256* f_sc = (u256*)((uint8_t*)ctx->scratch_a.p + (((size_t)nblocks + 1) * 4 + 63) / 64 * 64);
  Affine* f_pt = (Affine*)(f_sc + n);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_b.p, scalars, n * 32, hipMemcpyHostToDevice, st));
  launch_filter_count(st, (const u256*)ctx->scratch_b.p, (uint32_t)n, counts);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t survivors = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&survivors, counts + nblocks, 4, hipMemcpyDeviceToHost, st));
  // bases cache (opt-in, msm_amd_set_bases_cache): the converted resident copy stands in for upload + conversion; the
  // compaction below moves 64-byte records whatever their content, so it works on the converted form as well
  ++ctx->bases_cache_call;
  const void* dp = ctx->scratch_c.p;
  int pt_layout = MSM_AMD_POINT_H2C_AFFINE;
  BasesCacheEntry* hit = nullptr;
  AffPacked* fill = nullptr;
  if (ctx->bases_cache_budget) {
    bases_cache_reserve(ctx, 1, &points, &n, MSM_AMD_POINT_H2C_AFFINE);
    if ((hit = bases_cache_lookup(ctx, points, n, MSM_AMD_POINT_H2C_AFFINE)) == nullptr)
      if (BasesCacheEntry* f = bases_cache_insert(ctx, points, n, MSM_AMD_POINT_H2C_AFFINE)) fill = (AffPacked*)f->d_prepared;
  }
  if (hit) {
    dp = hit->d_prepared;
    pt_layout = MSM_AMD_POINT_PREPARED;
  } else {
    // an entry that was inserted for this call but never (provably) written must not be hit later: on any failure
    // from here to the end of the fill its key is cleared
    auto unkey = [&]() {
      if (fill)
        for (BasesCacheEntry& c : ctx->bases_cache)
          if (c.d_prepared == (void*)fill) c.host = nullptr;
    };
    hipError_t he = hipMemcpyAsync(ctx->scratch_c.p, points, n * 64, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && fill) {
      launch_convert_bases(st, (const Affine*)ctx->scratch_c.p, (uint32_t)n, fill);
      he = hipGetLastError();
      dp = fill;
      pt_layout = MSM_AMD_POINT_PREPARED;
    }
    if (he == hipSuccess) {
      if (int src_ = sync_stream_bounded(ctx, st, "msm_best upload")) {
        unkey();
        return src_;
      }
    }
    if (he != hipSuccess) {
      unkey();
      HIP_TRY(ctx, he);
    }
  }
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  const double zero_ratio = (double)(n - survivors) / (double)n;
  const void* ds = ctx->scratch_b.p;
  size_t m = n;
  if (zero_ratio >= 0.30) {   // msm.rs:470
    launch_filter_scatter(st, (const u256*)ctx->scratch_b.p, (const Affine*)dp, (uint32_t)n, counts, f_sc, f_pt);
    HIP_TRY(ctx, hipGetLastError());
    if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;   // run_batch_device starts on the front stream
    ds = f_sc;
    dp = f_pt;
    m = survivors;
  }
  if (m == 0) {   // every scalar is zero: the sum is the identity
    const Jacobian id = jac_identity();
    std::memcpy(out96, &id, 96);
    return MSM_AMD_OK;
  }
  return run_batch_device(ctx, MSM_AMD_SCALAR_MONT_LE, pt_layout, 1, &ds, &dp, &m, out96);
}

int msm_amd_msm_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                             const void* const* d_scalars, const void* const* d_points, const size_t* n,
                             void* out_host) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  return run_batch_device(ctx, scalar_layout, point_layout, n_inst, d_scalars, d_points, n, out_host);
}

int msm_amd_submit_batch_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, size_t n_inst,
                                const void* const* d_scalars, const void* const* d_points, const size_t* n,
                                void* out_host, int* ticket) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  return submit_batch_device(ctx, scalar_layout, point_layout, n_inst, d_scalars, d_points, n, out_host, ticket);
}

int msm_amd_wait_batch(msm_amd_ctx* ctx, int ticket) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  return wait_batch(ctx, ticket, false);
}

int msm_amd_set_wait_timeout_ms(msm_amd_ctx* ctx, uint32_t timeout_ms) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->wait_timeout_ms = timeout_ms;
  return MSM_AMD_OK;
}

int msm_amd_msm_device(msm_amd_ctx* ctx, int scalar_layout, int point_layout, const void* d_scalars,
                       const void* d_points, size_t n, void* out96_host) {
  return msm_amd_msm_batch_device(ctx, scalar_layout, point_layout, 1, &d_scalars, &d_points, &n, out96_host);
}

// Conversion of a resident point array to the internal packed form; ctx->mu held by the caller.
static int prepare_bases_locked(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n, void* d_prepared) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_or_mark_stalled(ctx)) return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy past the wait bound");   // a set-up step: the conversion scratch of workspace 0 must be idle
  hipStream_t st = ctx->stream;
  const u256* sc = nullptr;
  const Affine* pts = nullptr;
  int sc_mont = 0, rc;
  if ((rc = convert_inputs(ctx, ctx->ws[0], st, MSM_AMD_SCALAR_CANON_LE, point_layout, d_points, d_points, n, &sc,
                           &sc_mont, &pts)))
    return rc;
  launch_convert_bases(st, pts, (uint32_t)n, (AffPacked*)d_prepared);
  HIP_TRY(ctx, hipGetLastError());
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_bases_prepare_device(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n, void* d_prepared) {
  if (!ctx || !d_points || !d_prepared || n == 0 || n > 0xFFFFFFFFull)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad bases_prepare arguments");
  if (point_layout == MSM_AMD_POINT_PREPARED || point_bytes(point_layout) == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad point layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  return prepare_bases_locked(ctx, point_layout, d_points, n, d_prepared);
}

int msm_amd_bases_upload(msm_amd_ctx* ctx, int point_layout, const void* points, size_t n, void** d_prepared) {
  if (!ctx || !points || !d_prepared || n == 0 || n > 0xFFFFFFFFull)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad bases_upload arguments");
  const size_t pb = point_bytes(point_layout);
  if (pb == 0 || point_layout == MSM_AMD_POINT_PREPARED) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad point layout");
  *d_prepared = nullptr;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->scratch_c, n * pb))) return rc;
  void* d_out = nullptr;
  if (int qrc = quiesce_for_allocation(ctx, "the resident copy of the bases")) return qrc;
  HIP_TRY(ctx, hipMalloc(&d_out, n * sizeof(AffPacked)));
  hipError_t e = hipMemcpy(ctx->scratch_c.p, points, n * pb, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d_out);
    HIP_TRY(ctx, e);
  }
  if ((rc = prepare_bases_locked(ctx, point_layout, ctx->scratch_c.p, n, d_out))) {
    (void)hipFree(d_out);
    return rc;
  }
  *d_prepared = d_out;
  return MSM_AMD_OK;
}

// ---- precomputed window tables ------------------------------------------------------------
static uint32_t auto_table_window(size_t n) {
  // One bucket set serves all windows, so the window can grow until the 2^(c-1) buckets of the window reduction
  // (2 full additions each) cost as much as the mixed additions they save.  Windows whose top digit is narrow are
  // excluded: the n entries of the top window would all fall into its few slots (c = 19 leaves 7 bits for the top
  // window of a 254-bit scalar: 2^20 entries in 128 buckets and in ONE region of the sort).
  uint32_t best = 4;
  double best_cost = 1e300;
  for (uint32_t c = 4; c <= 21; ++c) {
    const uint32_t W = kModulusBits / c + 1;
    const uint32_t top = kModulusBits - (W - 1) * c;          // bits of the top window
    if (top + 6 < c && c > 6) continue;                        // top window concentrated > 64-fold
    if ((size_t)W * n > 0x7FFFFFFFull) continue;
    const double cost = 9.5 * (double)W * (double)n + 27.0 * (double)((size_t)1 << (c - 1)) + 14.0 * (double)W * (double)n / 32.0;
    if (cost < best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  return best;
}

static int tables_build_locked(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n, uint32_t window_size,
                               msm_amd_tables** out) {
  const uint32_t c = window_size ? window_size : auto_table_window(n);
  if (c < 4 || c > 21) return fail(ctx, MSM_AMD_INPUT_ERROR, "table window_size must be 0 (auto) or 4..21");
  const uint32_t W = kModulusBits / c + 1;
  if ((size_t)W * n > 0x7FFFFFFFull) return fail(ctx, MSM_AMD_INPUT_ERROR, "windows * n must stay below 2^31");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_or_mark_stalled(ctx)) return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy past the wait bound");   // a set-up step: the conversion scratch of workspace 0 must be idle
  hipStream_t st = ctx->stream;
  const u256* sc = nullptr;
  const Affine* pts = nullptr;
  int sc_mont = 0, rc;
  if (point_layout == MSM_AMD_POINT_PREPARED || point_layout == MSM_AMD_POINT_TABLES)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "tables are built from one of the host point layouts");
  if ((rc = convert_inputs(ctx, ctx->ws[0], st, MSM_AMD_SCALAR_CANON_LE, point_layout, d_points, d_points, n, &sc,
                           &sc_mont, &pts)))
    return rc;
  void* d_tab = nullptr;
  if (int qrc = quiesce_for_allocation(ctx, "the window tables")) return qrc;
  HIP_TRY(ctx, hipMalloc(&d_tab, (size_t)W * n * sizeof(AffPacked)));
  launch_build_tables(st, pts, (uint32_t)n, c, W, (AffPacked*)d_tab);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && sync_stream_bounded(ctx, st, "table build")) {
    ctx->graveyard.push_back(d_tab);   // the build may still be running: released when the ctx is idle
    return MSM_AMD_PIPELINE_ERROR;
  }
  if (e != hipSuccess) {
    (void)hipFree(d_tab);
    HIP_TRY(ctx, e);
  }
  auto* t = new msm_amd_tables();
  t->n = n;
  t->c = c;
  t->W = W;
  t->d_tables = d_tab;
  ctx->live_tables.push_back(t);
  *out = t;
  return MSM_AMD_OK;
}

int msm_amd_tables_build_device(msm_amd_ctx* ctx, int point_layout, const void* d_points, size_t n, uint32_t window_size,
                                msm_amd_tables** out) {
  if (!ctx || !d_points || !out || n == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad tables_build arguments");
  *out = nullptr;
  if (point_bytes(point_layout) == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown point layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  return tables_build_locked(ctx, point_layout, d_points, n, window_size, out);
}

int msm_amd_tables_build(msm_amd_ctx* ctx, int point_layout, const void* points, size_t n, uint32_t window_size,
                         msm_amd_tables** out) {
  if (!ctx || !points || !out || n == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad tables_build arguments");
  *out = nullptr;
  const size_t pb = point_bytes(point_layout);
  if (pb == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown point layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->scratch_c, n * pb))) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->scratch_c.p, points, n * pb, hipMemcpyHostToDevice));
  return tables_build_locked(ctx, point_layout, ctx->scratch_c.p, n, window_size, out);
}

int msm_amd_tables_info(msm_amd_ctx* ctx, const msm_amd_tables* tables, size_t* n, uint32_t* window_size,
                        uint32_t* num_windows, size_t* device_bytes) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  const msm_amd_tables* t = find_tables(ctx, tables);
  if (!t) return fail(ctx, MSM_AMD_INPUT_ERROR, "not a table handle of this ctx");
  if (n) *n = t->n;
  if (window_size) *window_size = t->c;
  if (num_windows) *num_windows = t->W;
  if (device_bytes) *device_bytes = (size_t)t->W * t->n * sizeof(AffPacked);
  return MSM_AMD_OK;
}

int msm_amd_tables_free(msm_amd_ctx* ctx, msm_amd_tables* tables) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = std::find(ctx->live_tables.begin(), ctx->live_tables.end(), tables);
  if (it == ctx->live_tables.end()) return fail(ctx, MSM_AMD_INPUT_ERROR, "not a table handle of this ctx");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->live_tables.erase(it);
  if (drain_or_mark_stalled(ctx)) (void)hipFree(tables->d_tables);
  else ctx->graveyard.push_back(tables->d_tables);   // hipFree would wait for the device without bound
  delete tables;
  return MSM_AMD_OK;
}

int msm_amd_msm_tables(msm_amd_ctx* ctx, const msm_amd_tables* tables, int scalar_layout, const void* scalars,
                       void* out96) {
  if (!ctx || !tables || !scalars || !out96) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad msm_tables arguments");
  if (scalar_layout < MSM_AMD_SCALAR_MONT_LE || scalar_layout > MSM_AMD_SCALAR_CANON_BE32)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown scalar layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  const msm_amd_tables* t = find_tables(ctx, tables);
  if (!t) return fail(ctx, MSM_AMD_INPUT_ERROR, "not a table handle of this ctx");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  const size_t n = t->n;
  if ((rc = recover_if_stalled(ctx))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, n * 32))) return rc;
  if ((rc = stage_upload(ctx, ctx->scratch_b.p, scalars, n * 32))) return rc;
  const void* ds = ctx->scratch_b.p;
  const void* dp = tables;
  rc = run_batch_device(ctx, scalar_layout, MSM_AMD_POINT_TABLES, 1, &ds, &dp, &n, out96);
  ctx->upload_pending = false;
  return rc;
}

int msm_amd_msm_prepared(msm_amd_ctx* ctx, int scalar_layout, const void* scalars, const void* d_prepared, size_t n,
                         void* out96) {
  if (!ctx || !scalars || !d_prepared || !out96 || n == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad msm_prepared arguments");
  if (scalar_layout < MSM_AMD_SCALAR_MONT_LE || scalar_layout > MSM_AMD_SCALAR_CANON_BE32)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "unknown scalar layout");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = recover_if_stalled(ctx))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, n * 32))) return rc;
  if ((rc = stage_upload(ctx, ctx->scratch_b.p, scalars, n * 32))) return rc;
  const void* ds = ctx->scratch_b.p;
  rc = run_batch_device(ctx, scalar_layout, MSM_AMD_POINT_PREPARED, 1, &ds, &d_prepared, &n, out96);
  ctx->upload_pending = false;
  return rc;
}

int msm_amd_device_alloc(msm_amd_ctx* ctx, size_t bytes, void** d_ptr) {
  if (!ctx || !d_ptr || bytes == 0) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad device_alloc arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = quiesce_for_allocation(ctx, "device memory (msm_amd_device_alloc)")) return rc;
  HIP_TRY(ctx, hipMalloc(d_ptr, bytes));
  return MSM_AMD_OK;
}

int msm_amd_device_free(msm_amd_ctx* ctx, void* d_ptr) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!d_ptr) return MSM_AMD_OK;
  if (!drain_streams_bounded(ctx)) {   // the device does not answer: hipFree would wait for it without bound
    ctx->graveyard.push_back(d_ptr);   // released once the ctx is idle again, or at msm_amd_destroy
    return MSM_AMD_OK;
  }
  HIP_TRY(ctx, hipFree(d_ptr));
  return MSM_AMD_OK;
}

int msm_amd_copy_to_device(msm_amd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx || !d_dst || !h_src) return fail(ctx, MSM_AMD_INPUT_ERROR, "null pointer");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (int src_ = sync_stream_bounded(ctx, ctx->stream, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_copy_to_host(msm_amd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx || !h_dst || !d_src) return fail(ctx, MSM_AMD_INPUT_ERROR, "null pointer");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (int src_ = sync_stream_bounded(ctx, ctx->stream, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_ctx_device(const msm_amd_ctx* ctx) { return ctx ? ctx->device : -1; }

void* msm_amd_stream(msm_amd_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int msm_amd_synchronize(msm_amd_ctx* ctx) {
  if (!ctx) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_streams_bounded(ctx)) {
    ctx->stalled = true;
    return fail(ctx, MSM_AMD_PIPELINE_ERROR,
                "timed out after " + std::to_string(ctx->wait_timeout_ms) + " ms waiting for the ctx's streams to drain");
  }
  HIP_TRY(ctx, hipGetLastError());
  return recover_if_stalled(ctx);
}

int msm_amd_generate_instance(msm_amd_ctx* ctx, uint64_t seed, size_t n, int scalars_mont, void* d_points,
                              void* d_scalars) {
  if (!ctx || !d_points || !d_scalars || n == 0 || n > 0x7FFFFFFFull)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad generate_instance arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  launch_gen_instance(ctx->stream, seed, (uint32_t)n, scalars_mont, (Affine*)d_points, (u256*)d_scalars);
  HIP_TRY(ctx, hipGetLastError());
  if (int src_ = sync_stream_bounded(ctx, ctx->stream, __func__)) return src_;
  return MSM_AMD_OK;
}

// ---- per-stage entry points ------------------------------------------------------------------------
int msm_amd_prepare_buckets_indices(msm_amd_ctx* ctx, const uint32_t* scalars_be32, size_t n, uint32_t window_size,
                                    uint32_t num_windows, uint32_t* pairs_out) {
  if (!ctx || !scalars_be32 || !pairs_out || n == 0 || window_size == 0 || window_size > 32 || num_windows == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad prepare_buckets_indices arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const size_t pair_bytes = n * num_windows * 8;
  if ((rc = ensure(ctx, ctx->scratch_a, n * 32))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, n * 32))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, pair_bytes))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.p, scalars_be32, n * 32, hipMemcpyHostToDevice, st));
  launch_be32_to_le(st, (const uint32_t*)ctx->scratch_a.p, n, (uint32_t*)ctx->scratch_b.p);
  launch_ref_prepare(st, (const u256*)ctx->scratch_b.p, (uint32_t)n, window_size, num_windows,
                     (uint2*)ctx->scratch_c.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(pairs_out, ctx->scratch_c.p, pair_bytes, hipMemcpyDeviceToHost, st));
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_sort_buckets_indices(msm_amd_ctx* ctx, uint32_t* pairs, size_t n_pairs) {
  if (!ctx || !pairs) return fail(ctx, MSM_AMD_INPUT_ERROR, "null pointer");
  if (n_pairs == 0) return MSM_AMD_OK;
  if (n_pairs > 0xFFFFFFFFull) return fail(ctx, MSM_AMD_INPUT_ERROR, "too many pairs");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const uint32_t tiles = (uint32_t)((n_pairs + kRadixTile - 1) / kRadixTile);
  if ((rc = ensure(ctx, ctx->scratch_a, n_pairs * 8))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, n_pairs * 8))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, ((size_t)tiles + 1) * 256 * 4))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.p, pairs, n_pairs * 8, hipMemcpyHostToDevice, st));
  uint2* src = nullptr;
  launch_radix_sort_pairs(st, (uint2*)ctx->scratch_a.p, (uint2*)ctx->scratch_b.p, n_pairs,
                          (uint32_t*)ctx->scratch_c.p, &src);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(pairs, src, n_pairs * 8, hipMemcpyDeviceToHost, st));
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_sort_pairs_device(msm_amd_ctx* ctx, void* d_pairs, size_t n_pairs, uint32_t key_bits, float* kernel_ms) {
  if (!ctx || !d_pairs || key_bits == 0 || key_bits > 32) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad sort arguments");
  if (kernel_ms) *kernel_ms = 0.f;
  if (n_pairs == 0) return MSM_AMD_OK;
  if (n_pairs > 0xFFFFFFFFull) return fail(ctx, MSM_AMD_INPUT_ERROR, "too many pairs");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  const uint32_t tiles = (uint32_t)((n_pairs + kRadixTile - 1) / kRadixTile);
  if ((rc = ensure(ctx, ctx->scratch_b, n_pairs * 8))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, ((size_t)tiles + 1) * 256 * 4))) return rc;
  hipEvent_t e0, e1;
  HIP_TRY(ctx, hipEventCreate(&e0));
  HIP_TRY(ctx, hipEventCreate(&e1));
  HIP_TRY(ctx, hipEventRecord(e0, st));
  uint2* src = nullptr;
  launch_radix_sort_pairs(st, (uint2*)d_pairs, (uint2*)ctx->scratch_b.p, n_pairs, (uint32_t*)ctx->scratch_c.p, &src,
                          key_bits);
  hipError_t le = hipGetLastError();
  if (le == hipSuccess && src != (uint2*)d_pairs)   // odd number of passes: the result sits in the scratch buffer
    le = hipMemcpyAsync(d_pairs, src, n_pairs * 8, hipMemcpyDeviceToDevice, st);
  if (le == hipSuccess) le = hipEventRecord(e1, st);
  if (le == hipSuccess && sync_stream_bounded(ctx, st, "pair sort")) le = hipErrorNotReady;
  float ms = 0.f;
  if (le == hipSuccess) le = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIP_TRY(ctx, le);
  if (kernel_ms) *kernel_ms = ms;
  return MSM_AMD_OK;
}

int msm_amd_bucket_wise_accumulation(msm_amd_ctx* ctx, const uint32_t* sorted_pairs, size_t n_pairs,
                                     const uint32_t* points_be32, size_t n_points, uint32_t total_buckets,
                                     uint32_t* buckets_out) {
  if (!ctx || !sorted_pairs || !points_be32 || !buckets_out || n_points == 0 || total_buckets == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad bucket_wise_accumulation arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_or_mark_stalled(ctx))   // workspace 0 may belong to an MSM between submit_batch_device and wait_batch
    return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy past the wait bound");
  hipStream_t st = ctx->stream;
  int rc;
  const size_t pts_bytes = n_points * 96, bkt_bytes = (size_t)total_buckets * 96;
  if ((rc = ensure(ctx, ctx->scratch_a, std::max(pts_bytes, bkt_bytes)))) return rc;   // BE32 staging
  if ((rc = ensure(ctx, ctx->scratch_b, pts_bytes))) return rc;                        // points LE
  if ((rc = ensure(ctx, ctx->scratch_c, std::max<size_t>(n_pairs * 8, 8)))) return rc;
  if ((rc = ensure(ctx, ctx->ws[0].buckets, bkt_bytes))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.p, points_be32, pts_bytes, hipMemcpyHostToDevice, st));
  launch_be32_to_le(st, (const uint32_t*)ctx->scratch_a.p, n_points * 3, (uint32_t*)ctx->scratch_b.p);
  HIP_TRY(ctx, hipMemsetAsync(ctx->ws[0].buckets.p, 0, bkt_bytes, st));   // Appendix B item 8: explicit zero fill
  if (n_pairs) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_c.p, sorted_pairs, n_pairs * 8, hipMemcpyHostToDevice, st));
    launch_ref_accumulate(st, (const uint2*)ctx->scratch_c.p, n_pairs, (const Jacobian*)ctx->scratch_b.p,
                          (uint32_t)n_points, total_buckets, (Jacobian*)ctx->ws[0].buckets.p);
  }
  // LE -> BE32 is the same limb reversal
  launch_be32_to_le(st, (const uint32_t*)ctx->ws[0].buckets.p, (size_t)total_buckets * 3, (uint32_t*)ctx->scratch_a.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(buckets_out, ctx->scratch_a.p, bkt_bytes, hipMemcpyDeviceToHost, st));
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  return MSM_AMD_OK;
}

int msm_amd_sum_reduction(msm_amd_ctx* ctx, const uint32_t* buckets_be32, uint32_t buckets_size,
                          uint32_t num_windows, uint32_t* res_out) {
  if (!ctx || !buckets_be32 || !res_out || buckets_size == 0 || num_windows == 0)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad sum_reduction arguments");
  // production layout: nb = 2^lb >= buckets_size slots per window, slot i carries weight i + 1
  uint32_t c = kSegLog;
  while ((1u << c) < buckets_size) ++c;
  if (c > 24) return fail(ctx, MSM_AMD_INPUT_ERROR, "buckets_size too large");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!drain_or_mark_stalled(ctx))   // workspace 0 may belong to an MSM between submit_batch_device and wait_batch
    return fail(ctx, MSM_AMD_PIPELINE_ERROR, "device busy past the wait bound");
  hipStream_t st = ctx->stream;
  const Plan p = make_reduce_plan(c, num_windows);
  int rc;
  const size_t in_bytes = (size_t)buckets_size * num_windows * 96;
  if ((rc = ensure(ctx, ctx->scratch_a, in_bytes))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, in_bytes))) return rc;
  if ((rc = ensure(ctx, ctx->ws[0].buckets, p.total_buckets * sizeof(PtI)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.p, buckets_be32, in_bytes, hipMemcpyHostToDevice, st));
  const size_t words = (size_t)buckets_size * num_windows * 3;
  launch_be32_to_le(st, (const uint32_t*)ctx->scratch_a.p, words, (uint32_t*)ctx->scratch_b.p);
  launch_pad_buckets(st, (const Jacobian*)ctx->scratch_b.p, buckets_size, p.W, p.lb, (PtI*)ctx->ws[0].buckets.p);
  if ((rc = enqueue_reduce(ctx, ctx->ws[0], st, p, (const PtI*)ctx->ws[0].buckets.p, nullptr))) return rc;
  std::vector<Jacobian> partial(p.partial_count);
  HIP_TRY(ctx, hipMemcpyAsync(partial.data(), ctx->ws[0].partial.p, p.partial_count * sizeof(Jacobian),
                              hipMemcpyDeviceToHost, st));
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  // per-window value: reuse the fused Horner with a single window
  Plan one = p;
  one.W = 1;
  for (uint32_t w = 0; w < num_windows; ++w) {
    const Jacobian r = host_combine(partial.data() + (size_t)w * (p.lb + 1), one);
    jac_to_be32(r, res_out + (size_t)w * 24);
  }
  return MSM_AMD_OK;
}

size_t msm_amd_scalar_bytes(int scalar_layout) {
  return (scalar_layout == MSM_AMD_SCALAR_MONT_LE || scalar_layout == MSM_AMD_SCALAR_CANON_LE ||
          scalar_layout == MSM_AMD_SCALAR_CANON_BE32) ? 32 : 0;
}
size_t msm_amd_point_bytes(int point_layout) { return point_bytes(point_layout); }

int msm_amd_sum_points(const void* points96, size_t count, void* out96) {
  if ((!points96 && count) || !out96) return MSM_AMD_INPUT_ERROR;
  h64::Jac acc = h64::identity();
  for (size_t i = 0; i < count; ++i) {
    h64::Jac p;
    std::memcpy(&p, (const uint8_t*)points96 + i * 96, 96);
    acc = h64::jadd(acc, p);
  }
  const h64::Jac res = h64::normalise(acc);
  std::memcpy(out96, &res, 96);
  return MSM_AMD_OK;
}

int msm_amd_final_accumulation(const uint32_t* res_be32, uint32_t num_windows, uint32_t window_size,
                               uint32_t* point_out) {
  if (!res_be32 || !point_out || num_windows == 0) return MSM_AMD_INPUT_ERROR;
  h64::Jac acc = h64::identity();   // same formulas as jac_double / jac_add: identical coordinates, not only the same point
  for (int w = (int)num_windows - 1; w >= 0; --w) {
    for (uint32_t i = 0; i < window_size; ++i) acc = h64::jdouble(acc);
    acc = h64::jadd(acc, h64::load(be32_to_jac(res_be32 + (size_t)w * 24)));
  }
  jac_to_be32(h64::store(acc), point_out);
  return MSM_AMD_OK;
}

// Layout helper for the test-op entry points: limb reversal between the BE32 wire format and LE u256.
static void test_op_widths(int op, size_t* wa, size_t* wb) {
  const bool pt = test_op_is_point(op);
  *wa = pt ? 3 : 1;
  *wb = (op == MSM_AMD_OP_EC_MUL) ? 1 : *wa;
}

int msm_amd_test_op(msm_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t count) {
  if (!ctx || !a || !b || !out || count == 0 || op < 0 || op > kTestOpMax ||
      (op >= MSM_AMD_OP_H64_FP_MUL && op <= MSM_AMD_OP_H64_EC_DBL))   // (host-only ops 37, 38 are above kTestOpMax)
    return fail(ctx, MSM_AMD_INPUT_ERROR, "bad test_op arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  size_t wa, wb;
  test_op_widths(op, &wa, &wb);
  const size_t wo = wa;
  std::vector<uint32_t> la(count * wa * 8), lb(count * wb * 8), lo(count * wo * 8);
  for (size_t i = 0; i < count * wa; ++i)
    for (int l = 0; l < 8; ++l) la[i * 8 + l] = a[i * 8 + 7 - l];
  for (size_t i = 0; i < count * wb; ++i)
    for (int l = 0; l < 8; ++l) lb[i * 8 + l] = b[i * 8 + 7 - l];
  int rc;
  if ((rc = ensure(ctx, ctx->scratch_a, la.size() * 4))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_b, lb.size() * 4))) return rc;
  if ((rc = ensure(ctx, ctx->scratch_c, lo.size() * 4))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.p, la.data(), la.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_b.p, lb.data(), lb.size() * 4, hipMemcpyHostToDevice, st));
  launch_test_op(st, op, (const u256*)ctx->scratch_a.p, (const u256*)ctx->scratch_b.p, (u256*)ctx->scratch_c.p,
                 (uint32_t)count);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(lo.data(), ctx->scratch_c.p, lo.size() * 4, hipMemcpyDeviceToHost, st));
  if (int src_ = sync_stream_bounded(ctx, st, __func__)) return src_;
  for (size_t i = 0; i < count * wo; ++i)
    for (int l = 0; l < 8; ++l) out[i * 8 + l] = lo[i * 8 + 7 - l];
  return MSM_AMD_OK;
}

// ops 27..31, 37, 38 exist on the host only: the 4 x 64-bit arithmetic of the CPU tail (host_fq64.h)
static bool host64_test_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t count) {
  const bool inverse = op == MSM_AMD_OP_H64_FP_INV || op == MSM_AMD_OP_H64_FP_INV_FERMAT;
  if ((op < MSM_AMD_OP_H64_FP_MUL || op > MSM_AMD_OP_H64_EC_DBL) && !inverse) return false;
  const bool pt = op == MSM_AMD_OP_H64_EC_ADD || op == MSM_AMD_OP_H64_EC_DBL;
  for (size_t t = 0; t < count; ++t) {
    if (!pt) {
      const u256 xa = be32_to_u256(a + t * 8), xb = be32_to_u256(b + t * 8);
      h64::Fe x, y, r;
      std::memcpy(&x, &xa, 32);
      std::memcpy(&y, &xb, 32);
      if (inverse) r = op == MSM_AMD_OP_H64_FP_INV ? h64::inv(x) : h64::inv_fermat(x);
      else r = op == MSM_AMD_OP_H64_FP_MUL ? h64::mul(x, y) : (op == MSM_AMD_OP_H64_FP_ADD ? h64::add(x, y) : h64::sub(x, y));
      u256 ro;
      std::memcpy(&ro, &r, 32);
      u256_to_be32(ro, out + t * 8);
    } else {
      const h64::Jac p = h64::load(be32_to_jac(a + t * 24)), q = h64::load(be32_to_jac(b + t * 24));
      jac_to_be32(h64::store(op == MSM_AMD_OP_H64_EC_ADD ? h64::jadd(p, q) : h64::jdouble(p)), out + t * 24);
    }
  }
  return true;
}

int msm_amd_test_op_host(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t count) {
  if (!a || !b || !out || count == 0 || op < 0) return MSM_AMD_INPUT_ERROR;
  if (host64_test_op(op, a, b, out, count)) return MSM_AMD_OK;
  if (op > kTestOpMax) return MSM_AMD_INPUT_ERROR;
  size_t wa, wb;
  test_op_widths(op, &wa, &wb);
  const size_t wo = wa;
  // the op bodies index points as 3 consecutive u256 per element, scalars of EC_MUL as 1 per element
  std::vector<u256> la(count * wa), lb(count * std::max(wb, wa)), lo(count * wo);
  for (size_t i = 0; i < count * wa; ++i) la[i] = be32_to_u256(a + i * 8);
  for (size_t i = 0; i < count * wb; ++i) lb[i] = be32_to_u256(b + i * 8);
  for (size_t t = 0; t < count; ++t) run_test_op(op, la.data(), lb.data(), lo.data(), (uint32_t)t);
  for (size_t i = 0; i < count * wo; ++i) u256_to_be32(lo[i], out + i * 8);
  return MSM_AMD_OK;
}

// Test aid: occupy the ctx's main stream for at most max_ms (<= 5000) or until msm_amd_test_release -- what a stalled
// device looks like to the host waits.  The kernel has its own time bound, so nothing can stay blocked.
int msm_amd_test_hold(msm_amd_ctx* ctx, uint32_t max_ms, void** handle) {
  if (!ctx || !handle || max_ms == 0 || max_ms > 5000) return fail(ctx, MSM_AMD_INPUT_ERROR, "bad test_hold arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint32_t* flag = nullptr;
  HIP_TRY(ctx, hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  *flag = 0;
  launch_hold(ctx->stream, flag, (uint64_t)max_ms * 100000ull);   // wall_clock64 ticks at 100 MHz
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    (void)hipHostFree(flag);
    HIP_TRY(ctx, e);
  }
  *handle = flag;
  return MSM_AMD_OK;
}

int msm_amd_test_release(msm_amd_ctx* ctx, void* handle) {
  if (!ctx || !handle) return MSM_AMD_INPUT_ERROR;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  *(volatile uint32_t*)handle = 1u;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // bounded by the kernel's own limit
  HIP_TRY(ctx, hipHostFree(handle));
  return MSM_AMD_OK;
}

int msm_amd_last_timings(const msm_amd_ctx* ctx, msm_amd_timings* out) {
  if (!ctx || !out) return MSM_AMD_INPUT_ERROR;
  *out = ctx->timings;
  return MSM_AMD_OK;
}

uint64_t msm_amd_algorithmic_bytes(size_t n, uint32_t window_size, int accumulate_only) {
  // A(n) = 32n + 72nW + 2*96*totB ; A3(n) = 72nW + 96*totB, with
  // totB = W * (2^c - 1), 64-byte affine bases, 8-byte (bucket, point) pairs, 96-byte buckets.
  const uint64_t c = window_size ? window_size : auto_window(n);
  const uint64_t W = (kModulusBits + c - 1) / c;
  const uint64_t totB = W * ((1ull << c) - 1);
  if (accumulate_only) return 72ull * n * W + 96ull * totB;
  return 32ull * n + 72ull * n * W + 2ull * 96ull * totB;
}

}  // extern "C"
