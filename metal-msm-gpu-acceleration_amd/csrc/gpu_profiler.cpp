// gpu_profiler -- C++ counterpart of the reference CLI (src/bin/gpu_profiler.rs:17-172), same positional
// arguments and the same two report lines:
//
//   gpu_profiler [log_size=16] [num_instances=1] [mode=gpu] [retries=3] [parallel=false]
//                [--seed S] [--device D] [--window C] [--layout h2c|ark] [--json] [--vec-dir DIR | --vec-cache]
//                [--warmup N]   N untimed passes first (default 0, like the reference: its first timed pass then
//                               includes the one-off growth of the device workspaces)
//                [--gpus N]     shard the instance loop (gpu_profiler.rs:101-106) over N GPUs: instance j runs on
//                               GPU j mod N, one context and one host thread per GPU (msm_amd_msm_batch_multi);
//                               afterwards the per-GPU result blocks are all-gathered over RCCL (xGMI) and compared
//                [--devices a,b,..]  the device ordinals to use instead of 0..N-1 (a device may repeat: contexts then
//                               share it and the RCCL gather, which refuses duplicates, is skipped)
//                [--range-split]  with --gpus / --devices in gpu mode: every instance is split by POINT RANGE over all
//                               contexts instead (msm_amd_msm_range_multi: the "single huge instance" sharding)
//                [--no-rccl]    skip the RCCL gather (results are gathered through host memory anyway)
//                [--threads T]  host threads of the cpu / gpu_cpu modes (default: every CPU the process may use)
//                [--reference-split]  gpu_cpu: the reference's split policy instead of the one measured on MI355X
//                [--bases-cache MB]   gpu / best_gpu: opt-in cache of converted bases (msm_amd_set_bases_cache)
//
// Modes (gpu_profiler.rs:143-172)
//   gpu       metal::msm::gpu_msm_h2c      -> msm_amd_gpu_msm_h2c (host buffers, upload included)
//   gpu_cpu   metal::msm::gpu_with_cpu     -> msm_amd_gpu_with_cpu; split measured on MI355X (msm_amd_tuned_split)
//                                             unless --reference-split; the split used is printed
//   best_gpu  metal::msm_best              -> msm_amd_msm_best (device filter_zeros + MSM)
//   cpu       halo2curves::msm::msm_best   -> msm_amd_host_msm: the library's own multi-threaded CPU MSM (signed
//                                             digits, batched-affine buckets); needs NO GPU, like the reference's
//                                             cpu mode.  NOT halo2curves and not the test oracle.
//   check     gpu_with_cpu vs cpu equality -> byte comparison of the two normalised 96-byte results
// Extra mode
//   gpu_resident   inputs generated once on the device (msm_amd_generate_instance) and kept resident: the
//                  configuration the headline metric is quoted on.
//
// Instances come from the deterministic generator (role of preprocess.rs:143-202's instance cache): on the device
// for the GPU modes (copied to the host for the host-buffer modes), on the host for the cpu mode -- same bytes.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msm_amd.h"

static std::vector<size_t> ns_of(unsigned count, size_t n) { return std::vector<size_t>(count, n); }

static std::vector<msm_amd_ctx*> g_ctxs;
static msm_amd_gather* g_gather = nullptr;

// Explicit teardown before the process returns into static destruction: contexts (device-wide sync, events, memory,
// streams), the RCCL communicators, then both output streams flushed.
static void teardown() {
  if (g_gather) msm_amd_gather_destroy(g_gather);
  g_gather = nullptr;
  for (msm_amd_ctx* c : g_ctxs) msm_amd_destroy(c);
  g_ctxs.clear();
  std::fflush(stdout);
  std::fflush(stderr);
}

static void die(msm_amd_ctx* ctx, int st, const char* what) {
  std::fprintf(stderr, "[ERROR] %s: %s %s\n", what, msm_amd_strerror(st), ctx ? msm_amd_last_error(ctx) : "");
  teardown();
  std::exit(1);
}

static int fail(const char* msg) {
  std::fprintf(stderr, "[ERROR] %s\n", msg);
  teardown();
  return 1;
}

int main(int argc, char** argv) {
  std::vector<std::string> pos;
  uint64_t seed = 0xB2540000ull;
  int device = -1, window = 0, gpus = 1, threads = 0;
  unsigned warmup = 0;
  size_t cache_mb = 0;
  bool json = false, no_rccl = false, reference_split = false, range_split = false;
  bool ark = false;   // --layout ark: ark_bn254 G1Projective points (96 B, z = one), config 5 of BASELINE.json
  bool use_vecs = false;   // --vec-dir DIR | --vec-cache: inputs come from / go to the reference's instance file
  std::string vec_dir;
  std::vector<int> devices;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--window" && i + 1 < argc) window = std::atoi(argv[++i]);
    else if (a == "--layout" && i + 1 < argc) ark = std::string(argv[++i]) == "ark";
    else if (a == "--json") json = true;
    else if (a == "--warmup" && i + 1 < argc) warmup = (unsigned)std::strtoul(argv[++i], nullptr, 10);
    else if (a == "--vec-dir" && i + 1 < argc) { use_vecs = true; vec_dir = argv[++i]; }
    else if (a == "--vec-cache") use_vecs = true;   // $HOME/.msm_gpu_acceleration/msm_vecs, like the reference
    else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
    else if (a == "--devices" && i + 1 < argc) {
      for (const char* s = argv[++i]; *s;) {
        char* end = nullptr;
        devices.push_back((int)std::strtol(s, &end, 10));
        if (end == s) break;
        s = *end == ',' ? end + 1 : end;
      }
    }
    else if (a == "--no-rccl") no_rccl = true;
    else if (a == "--range-split") range_split = true;
    else if (a == "--threads" && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (a == "--reference-split") reference_split = true;
    else if (a == "--bases-cache" && i + 1 < argc) cache_mb = (size_t)std::strtoull(argv[++i], nullptr, 10);
    else pos.push_back(a);
  }
  // positional parsing with the reference's defaults (gpu_profiler.rs:24-63)
  const unsigned log_size = pos.size() > 0 ? (unsigned)std::strtoul(pos[0].c_str(), nullptr, 10) : 16;
  const unsigned num_instances = pos.size() > 1 ? (unsigned)std::strtoul(pos[1].c_str(), nullptr, 10) : 1;
  std::string mode = pos.size() > 2 ? pos[2] : "gpu";
  for (auto& ch : mode) ch = (char)std::tolower(ch);
  const unsigned retries = pos.size() > 3 ? (unsigned)std::strtoul(pos[3].c_str(), nullptr, 10) : 3;
  const bool parallel = pos.size() > 4 && pos[4] == "true";
  std::fprintf(stderr, "[INFO] Log instance size: %u\n[INFO] Number of instances: %u\n[INFO] Run mode: %s\n"
                       "[INFO] Retries: %u\n[INFO] Parallel runs: %s\n",
               log_size, num_instances, mode.c_str(), retries, parallel ? "true" : "false");
  if (log_size == 0 || log_size > 28 || num_instances == 0 || retries == 0) return fail("bad arguments");
  const bool host_inputs = mode == "gpu" || mode == "gpu_cpu" || mode == "best_gpu" || mode == "cpu" || mode == "check";
  if (!host_inputs && mode != "gpu_resident") {
    std::fprintf(stderr, "[ERROR] Invalid RUN_MODE: %s\n", mode.c_str());   // gpu_profiler.rs:167-170
    return 1;
  }
  if (parallel)
    std::fprintf(stderr, "[INFO] parallel=true: instances go through ONE batched call (the reference's random "
                         "chunk/sleep harness, gpu_profiler.rs:107-131, is not reproduced)\n");
  if (!devices.empty()) gpus = (int)devices.size();
  if (gpus < 1 || gpus > 64) return fail("--gpus must be 1..64");
  const bool multi = gpus > 1 || !devices.empty();
  if (multi && mode != "gpu" && mode != "gpu_resident") return fail("--gpus / --devices shard the gpu and gpu_resident modes");
  if (multi && (ark || use_vecs)) return fail("--gpus / --devices take the h2c layout and generated instances");
  if (range_split && (!multi || mode != "gpu")) return fail("--range-split needs --gpus / --devices and the gpu mode (host buffers)");
  if (range_split) no_rccl = true;   // the partial results are added inside msm_amd_msm_range_multi; nothing to gather
  if (threads <= 0) threads = msm_amd_host_threads();

  const size_t n = (size_t)1 << log_size;
  std::vector<uint8_t> out((size_t)num_instances * 96);
  std::vector<std::vector<uint8_t>> h_pts, h_sc;
  int st = 0;

  // ---- cpu mode: no GPU anywhere (BASELINE config 1: `gpu_profiler 16 1 cpu 5`) --------------------------------
  if (mode == "cpu") {
    h_pts.resize(num_instances);
    h_sc.resize(num_instances);
    for (unsigned j = 0; j < num_instances; ++j) {
      h_pts[j].resize(n * 64);
      h_sc[j].resize(n * 32);
      if ((st = msm_amd_generate_instance_host(seed + j, n, 1, h_pts[j].data(), h_sc[j].data(), threads)))
        die(nullptr, st, "generate_instance_host");
    }
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned r = 0; r < retries + warmup; ++r) {
      if (r == warmup) t0 = std::chrono::steady_clock::now();
      for (unsigned j = 0; j < num_instances; ++j)
        if ((st = msm_amd_host_msm(MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, h_sc[j].data(), h_pts[j].data(), n,
                                   threads, out.data() + (size_t)j * 96)))
          die(nullptr, st, "host_msm");
    }
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "[INFO] CPU MSM of this library on %d host threads (no GPU used)\n", threads);
    std::fprintf(stderr, "[INFO] Total Execution Time: %.3fms\n", total_ms);
    std::fprintf(stderr, "[INFO] Average Instance Execution Time: %.3fms\n", total_ms / num_instances / retries);
    if (json) {
      std::printf("{\"log_size\": %u, \"num_instances\": %u, \"mode\": \"cpu\", \"retries\": %u, \"total_ms\": %.4f, "
                  "\"avg_instance_ms\": %.4f, \"host_threads\": %d, \"result0_x_le_hex\": \"",
                  log_size, num_instances, retries, total_ms, total_ms / num_instances / retries, threads);
      for (int i = 0; i < 32; ++i) std::printf("%02x", out[i]);
      std::printf("\"}\n");
    }
    teardown();
    return 0;
  }

  // ---- contexts: one per GPU ------------------------------------------------------------------------------------
  if (devices.empty()) {
    if (gpus == 1) devices.push_back(device);
    else for (int g = 0; g < gpus; ++g) devices.push_back(g);
  }
  const size_t G = devices.size();
  for (size_t g = 0; g < G; ++g) {
    msm_amd_ctx* c = nullptr;
    if ((st = msm_amd_init(devices[g], &c))) die(nullptr, st, "msm_amd_init");
    g_ctxs.push_back(c);
    if (window && (st = msm_amd_set_window_size(c, (uint32_t)window))) die(c, st, "set_window_size");
    if (cache_mb && (st = msm_amd_set_bases_cache(c, cache_mb << 20))) die(c, st, "set_bases_cache");
  }
  msm_amd_ctx* ctx = g_ctxs[0];
  auto owner = [&](unsigned j) { return g_ctxs[msm_amd_shard_owner(j, G)]; };
  if (multi) {
    std::fprintf(stderr, "[INFO] Sharding %u instances over %zu contexts (devices:", num_instances, G);
    for (size_t g = 0; g < G; ++g) std::fprintf(stderr, " %d", msm_amd_ctx_device(g_ctxs[g]));
    std::fprintf(stderr, "), instance j -> context j mod %zu\n", G);
  }

  std::vector<void*> d_pts(num_instances), d_sc(num_instances);
  for (unsigned j = 0; j < num_instances; ++j) {   // instance j lives on the GPU that will run it
    msm_amd_ctx* c = owner(j);
    if ((st = msm_amd_device_alloc(c, n * 64, &d_pts[j]))) die(c, st, "device_alloc");
    if ((st = msm_amd_device_alloc(c, n * 32, &d_sc[j]))) die(c, st, "device_alloc");
    if ((st = msm_amd_generate_instance(c, seed + j, n, 1, d_pts[j], d_sc[j]))) die(c, st, "generate_instance");
  }
  if (host_inputs) {
    h_pts.resize(num_instances);
    h_sc.resize(num_instances);
    for (unsigned j = 0; j < num_instances; ++j) {
      h_pts[j].resize(n * 64);
      h_sc[j].resize(n * 32);
      if ((st = msm_amd_copy_to_host(owner(j), h_pts[j].data(), d_pts[j], n * 64))) die(owner(j), st, "copy_to_host");
      if ((st = msm_amd_copy_to_host(owner(j), h_sc[j].data(), d_sc[j], n * 32))) die(owner(j), st, "copy_to_host");
    }
  }
  if (ark) {
    if (mode != "gpu_resident") return fail("--layout ark is measured in gpu_resident mode only");
    // build G1Projective {x, y, z = R mod p} on the host from the generated affine points and keep it resident
    static const uint8_t kMontOne[32] = {0x9d, 0x0d, 0x8f, 0xc5, 0x8d, 0x43, 0x5d, 0xd3, 0x3d, 0x0b, 0xc7,
                                         0xf5, 0x28, 0xeb, 0x78, 0x0a, 0x2c, 0x46, 0x79, 0x78, 0x6f, 0xa3,
                                         0x6e, 0x66, 0x2f, 0xdf, 0x07, 0x9a, 0xc1, 0x77, 0x0a, 0x0e};
    std::vector<uint8_t> aff(n * 64), proj(n * 96);
    for (unsigned j = 0; j < num_instances; ++j) {
      if ((st = msm_amd_copy_to_host(ctx, aff.data(), d_pts[j], n * 64))) die(ctx, st, "copy_to_host");
      for (size_t i = 0; i < n; ++i) {
        std::memcpy(&proj[i * 96], &aff[i * 64], 64);
        std::memcpy(&proj[i * 96 + 64], kMontOne, 32);
      }
      msm_amd_device_free(ctx, d_pts[j]);
      if ((st = msm_amd_device_alloc(ctx, n * 96, &d_pts[j]))) die(ctx, st, "device_alloc");
      if ((st = msm_amd_copy_to_device(ctx, d_pts[j], proj.data(), n * 96))) die(ctx, st, "copy_to_device");
    }
  }
  // get_or_create_msm_instances (preprocess.rs:143-202): wire-layout inputs from msm_{log}x{n}.bin, written
  // from the generated instances when the file does not exist yet
  int sc_layout = MSM_AMD_SCALAR_MONT_LE;
  int pt_layout = ark ? MSM_AMD_POINT_ARK_PROJECTIVE : MSM_AMD_POINT_H2C_AFFINE;
  if (use_vecs) {
    if (ark || (mode != "gpu" && mode != "gpu_resident")) return fail("instance files feed the gpu / gpu_resident modes (wire layout)");
    char path[4096];
    if (!msm_amd_instances_default_path(vec_dir.empty() ? nullptr : vec_dir.c_str(), log_size, num_instances, path,
                                        sizeof path))
      return fail("instance path too long");
    std::vector<std::vector<uint8_t>> w_pts(num_instances), w_sc(num_instances);
    for (unsigned j = 0; j < num_instances; ++j) {
      w_pts[j].resize(n * 96);
      w_sc[j].resize(n * 32);
    }
    msm_amd_instance_file* file = nullptr;
    st = msm_amd_instances_open(path, &file);
    if (st == MSM_AMD_OK) {
      std::fprintf(stderr, "[INFO] Loading MSM instances from file: %s\n", path);
      if (msm_amd_instances_count(file) != num_instances || msm_amd_instances_size(file, 0) != n) {
        std::fprintf(stderr, "[ERROR] Invalid data: File mismatch: has instance_size=%zu and num_instances=%zu, "
                             "need %u & %u\n",
                     msm_amd_instances_size(file, 0), msm_amd_instances_count(file), log_size, num_instances);
        msm_amd_instances_close(file);
        teardown();
        return 1;
      }
      for (unsigned j = 0; j < num_instances; ++j) {
        if (msm_amd_instances_size(file, j) != n) die(ctx, MSM_AMD_INVALID_DATA, "instance size");
        if ((st = msm_amd_instances_read(file, j, w_pts[j].data(), w_sc[j].data()))) die(ctx, st, "instances_read");
      }
      msm_amd_instances_close(file);
    } else if (st == MSM_AMD_FILE_OPEN_ERROR) {
      std::fprintf(stderr, "[INFO] Saving MSM instances to file: %s\n", path);
      std::vector<uint8_t> hp(n * 64), hs(n * 32);
      std::vector<const void*> wp(num_instances), wsp(num_instances);
      for (unsigned j = 0; j < num_instances; ++j) {
        if ((st = msm_amd_copy_to_host(ctx, hp.data(), d_pts[j], n * 64))) die(ctx, st, "copy_to_host");
        if ((st = msm_amd_copy_to_host(ctx, hs.data(), d_sc[j], n * 32))) die(ctx, st, "copy_to_host");
        if ((st = msm_amd_to_wire(MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, hs.data(), hp.data(), n,
                                  w_sc[j].data(), w_pts[j].data())))
          die(ctx, st, "to_wire");
        wp[j] = w_pts[j].data();
        wsp[j] = w_sc[j].data();
      }
      if ((st = msm_amd_instances_save(path, num_instances, ns_of(num_instances, n).data(), wp.data(), wsp.data())))
        die(ctx, st, "instances_save (does the directory exist?)");
    } else {
      die(ctx, st, "instances_open");
    }
    sc_layout = MSM_AMD_SCALAR_CANON_BE32;
    pt_layout = MSM_AMD_POINT_JAC_BE32;
    for (unsigned j = 0; j < num_instances; ++j) {
      msm_amd_device_free(ctx, d_pts[j]);
      if ((st = msm_amd_device_alloc(ctx, n * 96, &d_pts[j]))) die(ctx, st, "device_alloc");
      if ((st = msm_amd_copy_to_device(ctx, d_pts[j], w_pts[j].data(), n * 96))) die(ctx, st, "copy_to_device");
      if ((st = msm_amd_copy_to_device(ctx, d_sc[j], w_sc[j].data(), n * 32))) die(ctx, st, "copy_to_device");
    }
    h_pts = std::move(w_pts);
    h_sc = std::move(w_sc);
  }
  std::vector<const void*> sp(num_instances), pp(num_instances);
  std::vector<size_t> ns(num_instances, n);
  const size_t split_at = reference_split ? msm_amd_reference_split(n) : msm_amd_tuned_split(n);
  if (mode == "gpu_cpu" || mode == "check")
    std::fprintf(stderr, "[INFO] gpu_with_cpu split: %zu of %zu points to the GPU, %zu to %d host threads (%s)\n", split_at, n,
                 n - split_at, threads,
                 reference_split ? "the reference's policy, msm.rs:377-383" : "measured on MI355X; --reference-split for the reference's policy");

  std::vector<uint8_t> out_odd(out.size());
  msm_amd_multi_ticket* pending = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned r = 0; r < retries + warmup; ++r) {
    if (r == warmup) t0 = std::chrono::steady_clock::now();
    if (multi && range_split) {   // every instance over all contexts, by point range
      for (unsigned j = 0; j < num_instances && !st; ++j)
        st = msm_amd_msm_range_multi(g_ctxs.data(), G, sc_layout, pt_layout, h_sc[j].data(), h_pts[j].data(), n,
                                     out.data() + (size_t)j * 96);
      if (st) {
        for (msm_amd_ctx* c : g_ctxs)
          if (*msm_amd_last_error(c)) std::fprintf(stderr, "[ERROR] device %d: %s\n", msm_amd_ctx_device(c), msm_amd_last_error(c));
        die(nullptr, st, "msm_range_multi");
      }
    } else if (multi) {   // the sharded instance loop
      for (unsigned j = 0; j < num_instances; ++j) {
        sp[j] = host_inputs ? (const void*)h_sc[j].data() : d_sc[j];
        pp[j] = host_inputs ? (const void*)h_pts[j].data() : d_pts[j];
      }
      if (host_inputs) {
        st = msm_amd_msm_batch_multi(g_ctxs.data(), G, sc_layout, pt_layout, num_instances, sp.data(), pp.data(), ns.data(),
                                     out.data());
      } else {
        // resident inputs: retry r + 1 is submitted before retry r is waited for, so that every GPU's pipeline stays
        // full across the retries (msm_amd_submit_batch_multi_device); the last retry's results end up in `out`
        std::vector<uint8_t>& dst = r % 2 ? out_odd : out;
        msm_amd_multi_ticket* t = nullptr;
        st = msm_amd_submit_batch_multi_device(g_ctxs.data(), G, sc_layout, pt_layout, num_instances, sp.data(), pp.data(),
                                               ns.data(), dst.data(), &t);
        if (!st && pending) st = msm_amd_wait_batch_multi(pending);
        pending = t;
        if (!st && r + 1 == retries + warmup) {
          st = msm_amd_wait_batch_multi(pending);
          pending = nullptr;
          if (!st && r % 2) out = out_odd;
        }
      }
      if (st) {
        for (msm_amd_ctx* c : g_ctxs)
          if (*msm_amd_last_error(c)) std::fprintf(stderr, "[ERROR] device %d: %s\n", msm_amd_ctx_device(c), msm_amd_last_error(c));
        die(nullptr, st, "msm_batch_multi");
      }
    } else if (mode == "gpu") {
      if (parallel) {
        for (unsigned j = 0; j < num_instances; ++j) { sp[j] = h_sc[j].data(); pp[j] = h_pts[j].data(); }
        st = msm_amd_msm_batch(ctx, sc_layout, pt_layout, num_instances, sp.data(), pp.data(), ns.data(), out.data());
        if (st) die(ctx, st, "msm_batch");
      } else {
        for (unsigned j = 0; j < num_instances; ++j) {   // sequential runs (gpu_profiler.rs:104-106)
          st = use_vecs ? msm_amd_msm(ctx, sc_layout, pt_layout, h_sc[j].data(), h_pts[j].data(), n,
                                      out.data() + (size_t)j * 96)
                        : msm_amd_gpu_msm_h2c(ctx, h_sc[j].data(), h_pts[j].data(), n, out.data() + (size_t)j * 96);
          if (st) die(ctx, st, "gpu_msm_h2c");
        }
      }
    } else if (mode == "gpu_cpu" || mode == "best_gpu" || mode == "check") {
      for (unsigned j = 0; j < num_instances; ++j) {   // run_selected_msm (gpu_profiler.rs:143-172)
        uint8_t* o = out.data() + (size_t)j * 96;
        if (mode == "gpu_cpu") {
          st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, split_at, threads, o);
        } else if (mode == "best_gpu") {
          st = msm_amd_msm_best(ctx, h_sc[j].data(), h_pts[j].data(), n, o);
        } else {   // check always exercises BOTH halves: the reference's split
          uint8_t ref[96];
          st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, msm_amd_reference_split(n), threads, o);
          if (!st) st = msm_amd_host_msm(MSM_AMD_SCALAR_MONT_LE, MSM_AMD_POINT_H2C_AFFINE, h_sc[j].data(), h_pts[j].data(),
                                         n, threads, ref);
          if (!st && std::memcmp(o, ref, 96) != 0) {
            std::fprintf(stderr, "[ERROR] check failed: gpu_with_cpu != cpu for instance %u\n", j);   // :161-165
            teardown();
            return 1;
          }
        }
        if (st) die(ctx, st, mode.c_str());
      }
    } else {
      for (unsigned j = 0; j < num_instances; ++j) { sp[j] = d_sc[j]; pp[j] = d_pts[j]; }
      st = msm_amd_msm_batch_device(ctx, sc_layout, pt_layout, num_instances, sp.data(), pp.data(), ns.data(),
                                    out.data());
      if (st) die(ctx, st, "msm_batch_device");
    }
  }
  const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // the reference's two report lines (gpu_profiler.rs:135-140)
  std::fprintf(stderr, "[INFO] Total Execution Time: %.3fms\n", total_ms);
  std::fprintf(stderr, "[INFO] Average Instance Execution Time: %.3fms\n", total_ms / num_instances / retries);

  // ---- RCCL gather of the per-GPU result blocks: rank g contributes its ceil(I / G) x 96 B
  bool rccl_done = false, rccl_ok = false;
  double rccl_ms = 0;
  if (multi && !no_rccl) {
    bool distinct = true;
    for (size_t a = 0; a < G; ++a)
      for (size_t b = 0; b < a; ++b)
        if (msm_amd_ctx_device(g_ctxs[a]) == msm_amd_ctx_device(g_ctxs[b])) distinct = false;
    if (!distinct) {
      std::fprintf(stderr, "[INFO] RCCL gather skipped: contexts share a device (RCCL wants one rank per GPU)\n");
    } else {
      std::vector<int> devs(G);
      for (size_t g = 0; g < G; ++g) devs[g] = msm_amd_ctx_device(g_ctxs[g]);
      if ((st = msm_amd_gather_init(devs.data(), (int)G, &g_gather))) die(nullptr, st, "msm_amd_gather_init (RCCL)");
      const size_t per = msm_amd_shard_count(num_instances, G, 0);   // the largest block; shorter ones are zero-padded
      std::vector<std::vector<uint8_t>> send(G, std::vector<uint8_t>(per * 96, 0)), recv(G, std::vector<uint8_t>(per * 96 * G));
      std::vector<const void*> sendp(G);
      std::vector<void*> recvp(G);
      for (size_t g = 0; g < G; ++g) {
        for (size_t i = 0; i < msm_amd_shard_count(num_instances, G, g); ++i)
          std::memcpy(&send[g][i * 96], &out[(g + i * G) * 96], 96);
        sendp[g] = send[g].data();
        recvp[g] = recv[g].data();
      }
      const auto g0 = std::chrono::steady_clock::now();
      if ((st = msm_amd_gather_all(g_gather, sendp.data(), per * 96, recvp.data()))) {
        std::fprintf(stderr, "[ERROR] RCCL gather: %s\n", msm_amd_gather_last_error(g_gather));
        die(nullptr, st, "msm_amd_gather_all");
      }
      rccl_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();
      rccl_done = rccl_ok = true;
      for (size_t r = 0; r < G && rccl_ok; ++r)     // every rank must now hold every instance's result
        for (unsigned j = 0; j < num_instances && rccl_ok; ++j) {
          const size_t g = j % G, i = j / G;
          if (std::memcmp(&recv[r][(g * per + i) * 96], &out[(size_t)j * 96], 96) != 0) rccl_ok = false;
        }
      std::fprintf(stderr, "[INFO] RCCL all-gather of %zu x 96 B per rank over %zu ranks: %.3f ms (first call, incl. "
                           "connection setup), every rank holds all %u results: %s\n",
                   per, G, rccl_ms, num_instances, rccl_ok ? "yes" : "NO");
      if (!rccl_ok) {
        teardown();
        return 1;
      }
    }
  }
  msm_amd_timings t;
  msm_amd_last_timings(ctx, &t);
  if (json) {
    std::printf("{\"log_size\": %u, \"num_instances\": %u, \"mode\": \"%s\", \"retries\": %u, \"total_ms\": %.4f, "
                "\"avg_instance_ms\": %.4f, \"window_size\": %u, \"gpus\": %zu, \"rccl_gather\": %s, \"rccl_gather_ms\": %.3f, "
                "\"stage_ms\": {\"convert\": %.4f, \"digits\": %.4f, "
                "\"sort\": %.4f, \"accumulate\": %.4f, \"reduce\": %.4f, \"host_final\": %.4f, \"gpu_total\": %.4f}, "
                "\"result0_x_le_hex\": \"",
                log_size, num_instances, mode.c_str(), retries, total_ms, total_ms / num_instances / retries,
                t.window_size, G, rccl_done ? "true" : "false", rccl_ms, t.convert_ms, t.digits_ms, t.sort_ms,
                t.accumulate_ms, t.reduce_ms, t.final_ms, t.total_gpu_ms);
    for (int i = 0; i < 32; ++i) std::printf("%02x", out[i]);
    std::printf("\", \"results_fnv1a64\": \"");
    uint64_t h = 0xcbf29ce484222325ull;   // one checksum over all results: runs with different --gpus must agree
    for (uint8_t b : out) h = (h ^ b) * 0x100000001b3ull;
    std::printf("%016llx\"}\n", (unsigned long long)h);
  }
  for (unsigned j = 0; j < num_instances; ++j) {
    msm_amd_device_free(owner(j), d_pts[j]);
    msm_amd_device_free(owner(j), d_sc[j]);
  }
  teardown();
  return 0;
}
