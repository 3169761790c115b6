// Instance sharding over several contexts (GPUs) below the C ABI, and the RCCL gather of the 96-byte results.
//
// The loop being sharded is the reference's instance loop (src/bin/gpu_profiler.rs:101-106,
// benches/msm_benchmark.rs:29-34): MSM instances are independent, so instance j goes to ctx j mod G, every ctx is
// driven by its own host thread (pinned to the NUMA node of its GPU when sysfs tells), and no data-path collective
// exists -- the only exchange is the gather of ceil(I / G) x 96 bytes per rank at the end.
// In ONE process the results of all ctxs already land in the caller's buffer; the RCCL all-gather
// (msm_amd_gather_*) is what a rank-per-GPU deployment uses, and what `gpu_profiler --gpus N` runs so that every
// rank ends up with every result, over xGMI.
//
// RCCL is loaded with dlopen at msm_amd_gather_init, not linked: a process that already hosts an RCCL (PyTorch's
// wheel ships its own) keeps using that one, and libmsm_amd.so has no hard dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msm_amd.h"

namespace {

// "0-7,64-71" -> cpu_set_t
bool parse_cpulist(const char* s, cpu_set_t* set) {
  CPU_ZERO(set);
  bool any = false;
  while (*s) {
    char* end = nullptr;
    const long a = std::strtol(s, &end, 10);
    if (end == s) break;
    long b = a;
    s = end;
    if (*s == '-') {
      b = std::strtol(s + 1, &end, 10);
      if (end == s + 1) break;
      s = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
      if (c >= 0) {
        CPU_SET((int)c, set);
        any = true;
      }
    }
    while (*s == ',' || *s == ' ' || *s == '\n') ++s;
  }
  return any;
}

bool read_line(const std::string& path, char* buf, size_t len) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) return false;
  const bool ok = std::fgets(buf, (int)len, f) != nullptr;
  std::fclose(f);
  return ok;
}

// body(k) for every ctx k < G: the caller's thread takes ctx 0, one short-lived thread each for the others (pinned to
// the CPUs local to their GPU).  false = a thread could not be started (the ctxs without a thread did not run).
template <class F>
bool for_each_ctx_threaded(size_t G, bool pin, msm_amd_ctx* const* ctxs, F&& body) {
  std::vector<std::thread> threads;
  bool ok = true;
  auto run = [&](size_t k, bool pin_this) {
    cpu_set_t before;
    const bool restore = pin_this && sched_getaffinity(0, sizeof before, &before) == 0;
    if (pin_this) (void)msm_amd_pin_thread_to_device(msm_amd_ctx_device(ctxs[k]));
    body(k);
    if (restore) (void)sched_setaffinity(0, sizeof before, &before);
  };
  try {
    for (size_t k = 1; k < G; ++k) threads.emplace_back(run, k, pin);
  } catch (...) {
    ok = false;
  }
  if (ok) run(0, pin && G > 1);
  for (std::thread& t : threads) t.join();
  return ok;
}

}  // namespace

extern "C" {

size_t msm_amd_shard_owner(size_t instance, size_t n_ctx) { return n_ctx ? instance % n_ctx : 0; }

size_t msm_amd_shard_count(size_t n_inst, size_t n_ctx, size_t k) {
  if (n_ctx == 0 || k >= n_ctx) return 0;
  return n_inst / n_ctx + (k < n_inst % n_ctx ? 1 : 0);
}

// Restrict the calling thread to the CPUs local to `device` (its PCI function's local_cpulist in sysfs), intersected
// with the CPUs the thread may use now.  0 = pinned; 1 = nothing done (no NUMA information, or the intersection has
// fewer than 8 CPUs -- a container that was granted CPUs of another node); never an error for the MSM itself.
int msm_amd_pin_thread_to_device(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
    (void)hipGetLastError();
    return 1;
  }
  for (char* c = bus; *c; ++c) *c = (char)std::tolower(*c);
  const std::string dir = std::string("/sys/bus/pci/devices/") + bus;
  char line[4096];
  if (!read_line(dir + "/numa_node", line, sizeof line) || std::atoi(line) < 0) return 1;
  if (!read_line(dir + "/local_cpulist", line, sizeof line)) return 1;
  cpu_set_t local, now, both;
  if (!parse_cpulist(line, &local)) return 1;
  if (sched_getaffinity(0, sizeof now, &now) != 0) return 1;
  CPU_AND(&both, &local, &now);
  // never squeeze a rank's threads (submission, Horner passes, its parity check) onto a handful of CPUs: a cpuset that
  // barely touches the GPU's node is worse than no pinning
  if (CPU_COUNT(&both) < 8 && CPU_COUNT(&both) < CPU_COUNT(&now)) return 1;
  return sched_setaffinity(0, sizeof both, &both) == 0 ? 0 : 1;
}

// msm_amd_msm_batch over several ctxs: instance j runs on ctxs[j mod n_ctx]; one host thread per ctx (the caller's
// thread drives ctxs[0]); results land at out + 96 j.  With host buffers (device = 0) every ctx uploads its own
// instances; with device = 1 the buffers of instance j must live on the device of ctxs[j mod n_ctx].
static int batch_multi(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout, size_t n_inst,
                       const void* const* scalars, const void* const* points, const size_t* n, void* out, int device) {
  if (!ctxs || n_ctx == 0 || !scalars || !points || !n || !out || n_inst == 0) return MSM_AMD_INPUT_ERROR;
  for (size_t k = 0; k < n_ctx; ++k) {
    if (!ctxs[k]) return MSM_AMD_INPUT_ERROR;
    for (size_t l = 0; l < k; ++l)
      if (ctxs[l] == ctxs[k]) return MSM_AMD_INPUT_ERROR;   // one thread per ctx: a ctx listed twice would serialise
  }
  const size_t G = std::min(n_ctx, n_inst);
  // nothing below may throw across the C boundary: allocation failures and std::system_error from thread creation
  // become MSM_AMD_PIPELINE_ERROR, with every thread that did start joined first
  try {
    std::vector<int> rc(G, MSM_AMD_OK);
    auto run = [&](size_t k, bool pin) {
      try {
        cpu_set_t before;
        const bool restore = pin && sched_getaffinity(0, sizeof before, &before) == 0;
        if (pin) (void)msm_amd_pin_thread_to_device(msm_amd_ctx_device(ctxs[k]));
        const size_t cnt = msm_amd_shard_count(n_inst, G, k);
        std::vector<const void*> sp(cnt), pp(cnt);
        std::vector<size_t> nn(cnt);
        std::vector<uint8_t> res(cnt * 96);
        for (size_t i = 0; i < cnt; ++i) {
          const size_t j = k + i * G;
          sp[i] = scalars[j];
          pp[i] = points[j];
          nn[i] = n[j];
        }
        rc[k] = device ? msm_amd_msm_batch_device(ctxs[k], scalar_layout, point_layout, cnt, sp.data(), pp.data(),
                                                  nn.data(), res.data())
                       : msm_amd_msm_batch(ctxs[k], scalar_layout, point_layout, cnt, sp.data(), pp.data(), nn.data(),
                                           res.data());
        if (rc[k] == MSM_AMD_OK)
          for (size_t i = 0; i < cnt; ++i) std::memcpy((uint8_t*)out + (k + i * G) * 96, res.data() + i * 96, 96);
        if (restore) (void)sched_setaffinity(0, sizeof before, &before);   // the caller's thread gets its mask back
      } catch (...) {
        rc[k] = MSM_AMD_PIPELINE_ERROR;
      }
    };
    std::vector<std::thread> threads;
    bool spawn_failed = false;
    try {
      for (size_t k = 1; k < G; ++k) threads.emplace_back(run, k, true);
    } catch (...) {
      spawn_failed = true;   // the ctxs whose thread did not start simply do not run
    }
    run(0, G > 1);
    for (std::thread& t : threads) t.join();
    if (spawn_failed) return MSM_AMD_PIPELINE_ERROR;
    for (size_t k = 0; k < G; ++k)
      if (rc[k]) return rc[k];
    return MSM_AMD_OK;
  } catch (...) {
    return MSM_AMD_PIPELINE_ERROR;
  }
}

int msm_amd_msm_batch_multi(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout, size_t n_inst,
                            const void* const* scalars, const void* const* points, const size_t* n, void* out) {
  return batch_multi(ctxs, n_ctx, scalar_layout, point_layout, n_inst, scalars, points, n, out, 0);
}

int msm_amd_msm_batch_multi_device(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                                   size_t n_inst, const void* const* d_scalars, const void* const* d_points,
                                   const size_t* n, void* out_host) {
  return batch_multi(ctxs, n_ctx, scalar_layout, point_layout, n_inst, d_scalars, d_points, n, out_host, 1);
}

// Pipelined form of msm_amd_msm_batch_multi_device: submit enqueues the share of every ctx (msm_amd_submit_batch_device,
// one short-lived host thread per ctx so that eight GPUs are fed side by side) and returns; wait finishes every share
// (the host Horner passes, again one thread per ctx) and scatters the results to out_host + 96 j.  Up to four such
// batches may be in flight per ctx, so a caller that loops -- the reference's benchmark loop,
// benches/msm_benchmark.rs:29-34 -- keeps every GPU's pipeline full across calls instead of paying the call-boundary
// bubble of the blocking form once per batch.
struct msm_amd_multi_ticket {
  std::vector<msm_amd_ctx*> ctxs;
  std::vector<int> tickets;                 // per ctx, -1 = nothing submitted
  std::vector<std::vector<uint8_t>> res;    // per ctx: its results, contiguous
  size_t n_inst = 0;
  uint8_t* out = nullptr;
};

int msm_amd_submit_batch_multi_device(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                                      size_t n_inst, const void* const* d_scalars, const void* const* d_points,
                                      const size_t* n, void* out_host, msm_amd_multi_ticket** ticket) {
  if (ticket) *ticket = nullptr;
  if (!ctxs || n_ctx == 0 || !d_scalars || !d_points || !n || !out_host || !ticket || n_inst == 0)
    return MSM_AMD_INPUT_ERROR;
  for (size_t k = 0; k < n_ctx; ++k) {
    if (!ctxs[k]) return MSM_AMD_INPUT_ERROR;
    for (size_t l = 0; l < k; ++l)
      if (ctxs[l] == ctxs[k]) return MSM_AMD_INPUT_ERROR;
  }
  try {
    const size_t G = std::min(n_ctx, n_inst);
    std::unique_ptr<msm_amd_multi_ticket> t(new msm_amd_multi_ticket());
    t->ctxs.assign(ctxs, ctxs + G);
    t->tickets.assign(G, -1);
    t->res.resize(G);
    t->n_inst = n_inst;
    t->out = (uint8_t*)out_host;
    for (size_t k = 0; k < G; ++k) t->res[k].resize(msm_amd_shard_count(n_inst, G, k) * 96);
    std::vector<int> rc(G, MSM_AMD_OK);
    const bool spawned = for_each_ctx_threaded(G, true, ctxs, [&](size_t k) {
      try {
        const size_t cnt = msm_amd_shard_count(n_inst, G, k);
        std::vector<const void*> sp(cnt), pp(cnt);
        std::vector<size_t> nn(cnt);
        for (size_t i = 0; i < cnt; ++i) {
          const size_t j = k + i * G;
          sp[i] = d_scalars[j];
          pp[i] = d_points[j];
          nn[i] = n[j];
        }
        rc[k] = msm_amd_submit_batch_device(ctxs[k], scalar_layout, point_layout, cnt, sp.data(), pp.data(), nn.data(),
                                            t->res[k].data(), &t->tickets[k]);
      } catch (...) {
        rc[k] = MSM_AMD_PIPELINE_ERROR;
      }
    });
    int first = spawned ? MSM_AMD_OK : MSM_AMD_PIPELINE_ERROR;
    for (size_t k = 0; k < G && !first; ++k) first = rc[k];
    if (first) {   // finish what did get submitted: its result buffers die with the ticket
      for (size_t k = 0; k < G; ++k)
        if (t->tickets[k] >= 0 && rc[k] == MSM_AMD_OK) (void)msm_amd_wait_batch(ctxs[k], t->tickets[k]);
      return first;
    }
    *ticket = t.release();
    return MSM_AMD_OK;
  } catch (...) {
    return MSM_AMD_PIPELINE_ERROR;
  }
}

// Finishes a batch of msm_amd_submit_batch_multi_device and frees the ticket -- unless a ctx's wait ran into the
// bounded-wait limit (MSM_AMD_PIPELINE_ERROR with the work still in flight, msm_amd_set_wait_timeout_ms): the ticket
// then stays valid and may be waited for again, like a ticket of msm_amd_submit_batch_device.
int msm_amd_wait_batch_multi(msm_amd_multi_ticket* t) {
  if (!t) return MSM_AMD_INPUT_ERROR;
  try {
    const size_t G = t->ctxs.size();
    std::vector<int> rc(G, MSM_AMD_OK);
    const bool spawned = for_each_ctx_threaded(G, true, t->ctxs.data(), [&](size_t k) {
      if (t->tickets[k] < 0) return;
      rc[k] = msm_amd_wait_batch(t->ctxs[k], t->tickets[k]);
      if (rc[k] == MSM_AMD_OK) {
        t->tickets[k] = -1;
        const size_t cnt = t->res[k].size() / 96;
        for (size_t i = 0; i < cnt; ++i) std::memcpy(t->out + (k + i * G) * 96, t->res[k].data() + i * 96, 96);
      }
    });
    if (!spawned) return MSM_AMD_PIPELINE_ERROR;   // nothing lost: the tickets not waited for are still in t
    for (size_t k = 0; k < G; ++k)
      if (rc[k]) return rc[k];
    delete t;
    return MSM_AMD_OK;
  } catch (...) {
    return MSM_AMD_PIPELINE_ERROR;
  }
}

// ONE instance over several ctxs, split by point range: ctx g runs the MSM of points [begin_g, end_g) over all
// windows and the partial results are added -- the MSM of a union of point ranges is the sum of the MSMs, the algebra
// of the reference's GPU + CPU split (src/metal/msm.rs:385-419; SURVEY.md section 8e, "single huge instance").  Host
// buffers only: every ctx uploads its own range.  Ranges are the n / G split with the remainder on the first ranges.
void msm_amd_shard_range(size_t n, size_t n_ctx, size_t k, size_t* begin, size_t* end) {
  size_t b = 0, e = 0;
  if (n_ctx && k < n_ctx) {
    const size_t base = n / n_ctx, extra = n % n_ctx;
    b = k * base + std::min(k, extra);
    e = b + base + (k < extra ? 1 : 0);
  }
  if (begin) *begin = b;
  if (end) *end = e;
}

int msm_amd_msm_range_multi(msm_amd_ctx* const* ctxs, size_t n_ctx, int scalar_layout, int point_layout,
                            const void* scalars, const void* points, size_t n, void* out96) {
  if (!ctxs || n_ctx == 0 || !scalars || !points || !out96 || n == 0) return MSM_AMD_INPUT_ERROR;
  const size_t sb = msm_amd_scalar_bytes(scalar_layout), pb = msm_amd_point_bytes(point_layout);
  if (sb == 0 || pb == 0 || point_layout == MSM_AMD_POINT_PREPARED || point_layout == MSM_AMD_POINT_TABLES)
    return MSM_AMD_INPUT_ERROR;   // (device-side layouts belong to one ctx)
  const size_t G = std::min(n_ctx, n);
  std::vector<const void*> sp(G), pp(G);
  std::vector<size_t> nn(G);
  for (size_t g = 0; g < G; ++g) {
    size_t b, e;
    msm_amd_shard_range(n, G, g, &b, &e);
    sp[g] = (const uint8_t*)scalars + b * sb;
    pp[g] = (const uint8_t*)points + b * pb;
    nn[g] = e - b;
  }
  std::vector<uint8_t> partial(G * 96);
  const int rc = batch_multi(ctxs, G, scalar_layout, point_layout, G, sp.data(), pp.data(), nn.data(), partial.data(), 0);
  if (rc) return rc;
  return msm_amd_sum_points(partial.data(), G, out96);
}

// ---- RCCL gather ---------------------------------------------------------------------------------------------------
struct msm_amd_gather {
  void* lib = nullptr;
  int n = 0;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<void*> send, recv;
  size_t cap = 0;   // bytes per rank the device buffers hold
  std::string error;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const char* msm_amd_gather_last_error(const msm_amd_gather* g) { return g ? g->error.c_str() : ""; }

void msm_amd_gather_destroy(msm_amd_gather* g) {
  if (!g) return;
  for (int k = 0; k < (int)g->streams.size(); ++k) {
    (void)hipSetDevice(g->devices[k]);
    if (g->streams[k]) (void)hipStreamSynchronize(g->streams[k]);
  }
  for (int k = 0; k < (int)g->comms.size(); ++k)
    if (g->comms[k] && g->CommDestroy) (void)g->CommDestroy(g->comms[k]);
  for (int k = 0; k < (int)g->streams.size(); ++k) {
    (void)hipSetDevice(g->devices[k]);
    if (k < (int)g->send.size() && g->send[k]) (void)hipFree(g->send[k]);
    if (k < (int)g->recv.size() && g->recv[k]) (void)hipFree(g->recv[k]);
    if (g->streams[k]) (void)hipStreamDestroy(g->streams[k]);
  }
  (void)hipGetLastError();
  // the library stays loaded: unloading an RCCL whose proxy threads are winding down is not worth the risk
  delete g;
}

// One communicator per listed device, all in this process (ncclCommInitAll).  RCCL refuses a device listed twice.
int msm_amd_gather_init(const int* devices, int n_devices, msm_amd_gather** out) {
  if (!devices || n_devices <= 0 || !out) return MSM_AMD_INPUT_ERROR;
  *out = nullptr;
  msm_amd_gather* g = new msm_amd_gather();
  g->n = n_devices;
  g->devices.assign(devices, devices + n_devices);
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  g->lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);   // the RCCL this process already hosts, if any
  for (const char* nm : names) {
    if (g->lib) break;
    g->lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
  }
  if (!g->lib) {
    std::fprintf(stderr, "msm_amd_gather_init: cannot load librccl: %s\n", dlerror());
    delete g;
    return MSM_AMD_LIBRARY_ERROR;
  }
  g->CommInitAll = (decltype(g->CommInitAll))dlsym(g->lib, "ncclCommInitAll");
  g->CommDestroy = (decltype(g->CommDestroy))dlsym(g->lib, "ncclCommDestroy");
  g->AllGather = (decltype(g->AllGather))dlsym(g->lib, "ncclAllGather");
  g->GroupStart = (decltype(g->GroupStart))dlsym(g->lib, "ncclGroupStart");
  g->GroupEnd = (decltype(g->GroupEnd))dlsym(g->lib, "ncclGroupEnd");
  g->GetErrorString = (decltype(g->GetErrorString))dlsym(g->lib, "ncclGetErrorString");
  if (!g->CommInitAll || !g->CommDestroy || !g->AllGather || !g->GroupStart || !g->GroupEnd || !g->GetErrorString) {
    std::fprintf(stderr, "msm_amd_gather_init: librccl lacks a collective entry point\n");
    delete g;
    return MSM_AMD_FUNCTION_ERROR;
  }
  g->comms.assign(n_devices, nullptr);
  const ncclResult_t r = g->CommInitAll(g->comms.data(), n_devices, devices);
  if (r != ncclSuccess) {
    std::fprintf(stderr, "msm_amd_gather_init: ncclCommInitAll: %s\n", g->GetErrorString(r));
    g->comms.clear();
    msm_amd_gather_destroy(g);
    return MSM_AMD_PIPELINE_ERROR;
  }
  g->streams.assign(n_devices, nullptr);
  g->send.assign(n_devices, nullptr);
  g->recv.assign(n_devices, nullptr);
  for (int k = 0; k < n_devices; ++k) {
    if (hipSetDevice(devices[k]) != hipSuccess ||
        hipStreamCreateWithFlags(&g->streams[k], hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      msm_amd_gather_destroy(g);
      return MSM_AMD_PIPELINE_ERROR;
    }
  }
  *out = g;
  return MSM_AMD_OK;
}

int msm_amd_gather_size(const msm_amd_gather* g) { return g ? g->n : 0; }

// All-gather of bytes_per_rank bytes from every rank: send_host[k] is rank k's contribution, recv_host[k] (n x
// bytes_per_rank) receives everybody's, in rank order, through device buffers on rank k's GPU.
int msm_amd_gather_all(msm_amd_gather* g, const void* const* send_host, size_t bytes_per_rank, void* const* recv_host) {
  if (!g || !send_host || !recv_host || bytes_per_rank == 0) return MSM_AMD_INPUT_ERROR;
  auto hip_fail = [&](hipError_t e, const char* what) {
    (void)hipGetLastError();
    g->error = std::string(what) + ": " + hipGetErrorString(e);
    return (int)MSM_AMD_PIPELINE_ERROR;
  };
  hipError_t e;
  if (bytes_per_rank > g->cap) {
    for (int k = 0; k < g->n; ++k) {
      if ((e = hipSetDevice(g->devices[k])) != hipSuccess) return hip_fail(e, "hipSetDevice");
      if (g->send[k]) (void)hipFree(g->send[k]);
      if (g->recv[k]) (void)hipFree(g->recv[k]);
      g->send[k] = g->recv[k] = nullptr;
      if ((e = hipMalloc(&g->send[k], bytes_per_rank)) != hipSuccess) return hip_fail(e, "hipMalloc");
      if ((e = hipMalloc(&g->recv[k], bytes_per_rank * (size_t)g->n)) != hipSuccess) return hip_fail(e, "hipMalloc");
    }
    g->cap = bytes_per_rank;
  }
  for (int k = 0; k < g->n; ++k) {
    if (!send_host[k] || !recv_host[k]) return MSM_AMD_INPUT_ERROR;
    if ((e = hipSetDevice(g->devices[k])) != hipSuccess) return hip_fail(e, "hipSetDevice");
    if ((e = hipMemcpyAsync(g->send[k], send_host[k], bytes_per_rank, hipMemcpyHostToDevice, g->streams[k])) != hipSuccess)
      return hip_fail(e, "hipMemcpyAsync");
  }
  ncclResult_t r = g->GroupStart();
  for (int k = 0; r == ncclSuccess && k < g->n; ++k)
    r = g->AllGather(g->send[k], g->recv[k], bytes_per_rank, ncclUint8, g->comms[k], g->streams[k]);
  const ncclResult_t r2 = g->GroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) {
    g->error = std::string("ncclAllGather: ") + g->GetErrorString(r);
    return MSM_AMD_PIPELINE_ERROR;
  }
  for (int k = 0; k < g->n; ++k) {
    if ((e = hipSetDevice(g->devices[k])) != hipSuccess) return hip_fail(e, "hipSetDevice");
    if ((e = hipMemcpyAsync(recv_host[k], g->recv[k], bytes_per_rank * (size_t)g->n, hipMemcpyDeviceToHost,
                            g->streams[k])) != hipSuccess)
      return hip_fail(e, "hipMemcpyAsync");
  }
  for (int k = 0; k < g->n; ++k) {
    if ((e = hipSetDevice(g->devices[k])) != hipSuccess) return hip_fail(e, "hipSetDevice");
    if ((e = hipStreamSynchronize(g->streams[k])) != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return MSM_AMD_OK;
}

}  // extern "C"
