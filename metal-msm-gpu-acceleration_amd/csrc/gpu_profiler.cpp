// gpu_profiler -- C++ counterpart of the reference CLI (src/bin/gpu_profiler.rs:17-172), same positional
// arguments and the same two report lines:
//
//   gpu_profiler [log_size=16] [num_instances=1] [mode=gpu] [retries=3] [parallel=false]
//                [--seed S] [--device D] [--window C] [--layout h2c|ark] [--json] [--vec-dir DIR | --vec-cache]
//                [--warmup N]   N untimed passes first (default 0, like the reference: its first timed pass then
//                               includes the one-off growth of the device workspaces)
//
// Modes (gpu_profiler.rs:143-172)
//   gpu       metal::msm::gpu_msm_h2c      -> msm_amd_gpu_msm_h2c (host buffers, upload included)
//   gpu_cpu   metal::msm::gpu_with_cpu     -> msm_amd_gpu_with_cpu with the reference's split policy
//   best_gpu  metal::msm_best              -> msm_amd_msm_best (device filter_zeros + MSM)
//   cpu       halo2curves::msm::msm_best   -> the library's own host bucket method (the CPU half of
//                                             gpu_with_cpu with split_at = 0); NOT halo2curves and not the
//                                             test oracle -- bench.py's cpu_baseline leg times the oracle
//   check     gpu_with_cpu vs cpu equality -> byte comparison of the two normalised 96-byte results
// Extra mode
//   gpu_resident   inputs generated once on the device (msm_amd_generate_instance) and kept resident: the
//                  configuration the headline metric is quoted on.
//
// Instances are generated on the device with the deterministic generator (role of
// preprocess.rs:143-202's instance cache) and copied to the host for the host-buffer modes.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msm_amd.h"

static std::vector<size_t> ns_of(unsigned count, size_t n) { return std::vector<size_t>(count, n); }

static void die(msm_amd_ctx* ctx, int st, const char* what) {
  std::fprintf(stderr, "[ERROR] %s: %s %s\n", what, msm_amd_strerror(st), ctx ? msm_amd_last_error(ctx) : "");
  std::exit(1);
}

int main(int argc, char** argv) {
  std::vector<std::string> pos;
  uint64_t seed = 0xB2540000ull;
  int device = -1, window = 0;
  unsigned warmup = 0;
  bool json = false;
  bool ark = false;   // --layout ark: ark_bn254 G1Projective points (96 B, z = one), config 5 of BASELINE.json
  bool use_vecs = false;   // --vec-dir DIR | --vec-cache: inputs come from / go to the reference's instance file
  std::string vec_dir;
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
  if (log_size == 0 || log_size > 28 || num_instances == 0 || retries == 0) {
    std::fprintf(stderr, "[ERROR] bad arguments\n");
    return 1;
  }
  const bool host_inputs = mode == "gpu" || mode == "gpu_cpu" || mode == "best_gpu" || mode == "cpu" || mode == "check";
  if (!host_inputs && mode != "gpu_resident") {
    std::fprintf(stderr, "[ERROR] Invalid RUN_MODE: %s\n", mode.c_str());   // gpu_profiler.rs:167-170
    return 1;
  }
  if (parallel)
    std::fprintf(stderr, "[INFO] parallel=true: instances go through ONE batched call (the reference's random "
                         "chunk/sleep harness, gpu_profiler.rs:107-131, is not reproduced)\n");

  msm_amd_ctx* ctx = nullptr;
  int st = msm_amd_init(device, &ctx);
  if (st) die(nullptr, st, "msm_amd_init");
  if (window && (st = msm_amd_set_window_size(ctx, (uint32_t)window))) die(ctx, st, "set_window_size");

  const size_t n = (size_t)1 << log_size;
  std::vector<void*> d_pts(num_instances), d_sc(num_instances);
  std::vector<std::vector<uint8_t>> h_pts, h_sc;
  for (unsigned j = 0; j < num_instances; ++j) {
    if ((st = msm_amd_device_alloc(ctx, n * 64, &d_pts[j]))) die(ctx, st, "device_alloc");
    if ((st = msm_amd_device_alloc(ctx, n * 32, &d_sc[j]))) die(ctx, st, "device_alloc");
    if ((st = msm_amd_generate_instance(ctx, seed + j, n, 1, d_pts[j], d_sc[j]))) die(ctx, st, "generate_instance");
  }
  if (host_inputs) {
    h_pts.resize(num_instances);
    h_sc.resize(num_instances);
    for (unsigned j = 0; j < num_instances; ++j) {
      h_pts[j].resize(n * 64);
      h_sc[j].resize(n * 32);
      if ((st = msm_amd_copy_to_host(ctx, h_pts[j].data(), d_pts[j], n * 64))) die(ctx, st, "copy_to_host");
      if ((st = msm_amd_copy_to_host(ctx, h_sc[j].data(), d_sc[j], n * 32))) die(ctx, st, "copy_to_host");
    }
  }
  if (ark) {
    if (mode != "gpu_resident") {
      std::fprintf(stderr, "[ERROR] --layout ark is measured in gpu_resident mode only\n");
      return 1;
    }
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
    if (ark || (mode != "gpu" && mode != "gpu_resident")) {
      std::fprintf(stderr, "[ERROR] instance files feed the gpu / gpu_resident modes (wire layout)\n");
      return 1;
    }
    char path[4096];
    if (!msm_amd_instances_default_path(vec_dir.empty() ? nullptr : vec_dir.c_str(), log_size, num_instances, path,
                                        sizeof path)) {
      std::fprintf(stderr, "[ERROR] instance path too long\n");
      return 1;
    }
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
  std::vector<uint8_t> out((size_t)num_instances * 96);
  std::vector<const void*> sp(num_instances), pp(num_instances);
  std::vector<size_t> ns(num_instances, n);

  auto t0 = std::chrono::steady_clock::now();
  for (unsigned r = 0; r < retries + warmup; ++r) {
    if (r == warmup) t0 = std::chrono::steady_clock::now();
    if (mode == "gpu") {
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
    } else if (mode == "gpu_cpu" || mode == "best_gpu" || mode == "cpu" || mode == "check") {
      for (unsigned j = 0; j < num_instances; ++j) {   // run_selected_msm (gpu_profiler.rs:143-172)
        uint8_t* o = out.data() + (size_t)j * 96;
        if (mode == "gpu_cpu") {
          st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, msm_amd_reference_split(n), 0, o);
        } else if (mode == "best_gpu") {
          st = msm_amd_msm_best(ctx, h_sc[j].data(), h_pts[j].data(), n, o);
        } else if (mode == "cpu") {
          st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, 0, 0, o);
        } else {
          uint8_t ref[96];
          st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, msm_amd_reference_split(n), 0, o);
          if (!st) st = msm_amd_gpu_with_cpu(ctx, h_sc[j].data(), h_pts[j].data(), n, 0, 0, ref);
          if (!st && std::memcmp(o, ref, 96) != 0) {
            std::fprintf(stderr, "[ERROR] check failed: gpu_with_cpu != cpu for instance %u\n", j);   // :161-165
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
  msm_amd_timings t;
  msm_amd_last_timings(ctx, &t);
  if (json) {
    std::printf("{\"log_size\": %u, \"num_instances\": %u, \"mode\": \"%s\", \"retries\": %u, \"total_ms\": %.4f, "
                "\"avg_instance_ms\": %.4f, \"window_size\": %u, \"stage_ms\": {\"convert\": %.4f, \"digits\": %.4f, "
                "\"sort\": %.4f, \"accumulate\": %.4f, \"reduce\": %.4f, \"host_final\": %.4f, \"gpu_total\": %.4f}, "
                "\"result0_x_le_hex\": \"",
                log_size, num_instances, mode.c_str(), retries, total_ms, total_ms / num_instances / retries,
                t.window_size, t.convert_ms, t.digits_ms, t.sort_ms, t.accumulate_ms, t.reduce_ms, t.final_ms,
                t.total_gpu_ms);
    for (int i = 0; i < 32; ++i) std::printf("%02x", out[i]);
    std::printf("\"}\n");
  }
  for (unsigned j = 0; j < num_instances; ++j) {
    msm_amd_device_free(ctx, d_pts[j]);
    msm_amd_device_free(ctx, d_sc[j]);
  }
  msm_amd_destroy(ctx);
  return 0;
}
