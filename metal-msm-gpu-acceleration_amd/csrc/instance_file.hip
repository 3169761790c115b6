// instance_file.hip -- reader / writer of the reference's cached-instance files and the host-side
// wire-layout conversions.  Host code only (no kernels): it lives in the library so that the Python
// mirror, gpu_profiler and any FFI caller share one implementation.
//
// Format (src/utils/preprocess.rs:30-111): bincode 1.3 with default options = fixed-width little-endian
// integers, u64 sequence lengths, no framing for tuples:
//   Vec<MsmInstance>                      u64 n_instances
//   MsmInstance = (points, scalars)       points: u64 n_points, then n_points x Vec<u32>{u64 24, 24 x u32}
//                                         scalars: u64 n_scalars, then n_scalars x Vec<u32>{u64 8, 8 x u32}
// A point is (x, y, z), each 8 x u32 most-significant limb first, Montgomery R = 2^256; a scalar is the
// canonical integer, 8 x u32 most-significant limb first (limbs_conversion.rs:87-137, 282-327).
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msm_amd.h"
#include "bn254_fq.hip.h"

namespace {

using msm_amd::Fr;
using msm_amd::u256;

constexpr size_t kPointLimbs = 24, kScalarLimbs = 8;
constexpr size_t kPointRec = 8 + 4 * kPointLimbs;     // 104 bytes on disk
constexpr size_t kScalarRec = 8 + 4 * kScalarLimbs;   //  40 bytes on disk
constexpr size_t kChunk = 1 << 14;                    // records per buffered read / write

bool read_u64(FILE* f, uint64_t* v) { return std::fread(v, 8, 1, f) == 1; }   // hosts are little-endian

// Vec<Vec<u32>> with inner length `limbs`: strips the per-record length prefix into a dense array.
int read_records(FILE* f, size_t count, size_t limbs, uint8_t* out) {
  const size_t rec = 8 + 4 * limbs;
  std::vector<uint8_t> buf(kChunk * rec);
  for (size_t done = 0; done < count;) {
    const size_t m = count - done < kChunk ? count - done : kChunk;
    if (std::fread(buf.data(), rec, m, f) != m) return MSM_AMD_DESERIALIZATION_ERROR;
    for (size_t i = 0; i < m; ++i) {
      uint64_t len;
      std::memcpy(&len, &buf[i * rec], 8);
      if (len != limbs) return MSM_AMD_DESERIALIZATION_ERROR;
      std::memcpy(out + (done + i) * 4 * limbs, &buf[i * rec + 8], 4 * limbs);
    }
    done += m;
  }
  return MSM_AMD_OK;
}

int write_records(FILE* f, size_t count, size_t limbs, const uint8_t* in) {
  const size_t rec = 8 + 4 * limbs;
  std::vector<uint8_t> buf(kChunk * rec);
  const uint64_t len = limbs;
  for (size_t done = 0; done < count;) {
    const size_t m = count - done < kChunk ? count - done : kChunk;
    for (size_t i = 0; i < m; ++i) {
      std::memcpy(&buf[i * rec], &len, 8);
      std::memcpy(&buf[i * rec + 8], in + (done + i) * 4 * limbs, 4 * limbs);
    }
    if (std::fwrite(buf.data(), rec, m, f) != m) return MSM_AMD_FILE_OPEN_ERROR;
    done += m;
  }
  return MSM_AMD_OK;
}

// 32-byte little-endian value <-> 8 x u32 most-significant limb first
inline void le_to_be32(const uint8_t* le, uint8_t* be) {
  for (int i = 0; i < 8; ++i) std::memcpy(be + 4 * i, le + 4 * (7 - i), 4);
}
inline u256 load_le(const uint8_t* p) {
  u256 r;
  std::memcpy(r.v, p, 32);
  return r;
}
inline bool is_zero32(const uint8_t* p) {
  for (int i = 0; i < 32; ++i)
    if (p[i]) return false;
  return true;
}

}  // namespace

struct msm_amd_instance_file {
  FILE* f = nullptr;
  std::vector<uint64_t> n;        // points per instance
  std::vector<long long> offset;  // file offset of the instance's n_points field
};

extern "C" {

int msm_amd_instances_save(const char* path, size_t n_inst, const size_t* n, const void* const* points,
                           const void* const* scalars) {
  if (!path || (n_inst && (!n || !points || !scalars))) return MSM_AMD_INPUT_ERROR;
  for (size_t j = 0; j < n_inst; ++j)
    if (n[j] && (!points[j] || !scalars[j])) return MSM_AMD_INPUT_ERROR;
  FILE* f = std::fopen(path, "wb");
  if (!f) return MSM_AMD_FILE_OPEN_ERROR;
  int rc = MSM_AMD_OK;
  uint64_t v = n_inst;
  if (std::fwrite(&v, 8, 1, f) != 1) rc = MSM_AMD_FILE_OPEN_ERROR;
  for (size_t j = 0; j < n_inst && !rc; ++j) {
    v = n[j];
    if (std::fwrite(&v, 8, 1, f) != 1) rc = MSM_AMD_FILE_OPEN_ERROR;
    if (!rc) rc = write_records(f, n[j], kPointLimbs, (const uint8_t*)points[j]);
    if (!rc && std::fwrite(&v, 8, 1, f) != 1) rc = MSM_AMD_FILE_OPEN_ERROR;
    if (!rc) rc = write_records(f, n[j], kScalarLimbs, (const uint8_t*)scalars[j]);
  }
  if (std::fclose(f) != 0 && !rc) rc = MSM_AMD_FILE_OPEN_ERROR;
  return rc;
}

int msm_amd_instances_open(const char* path, msm_amd_instance_file** out) {
  if (!path || !out) return MSM_AMD_INPUT_ERROR;
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return MSM_AMD_FILE_OPEN_ERROR;
  auto fail = [&](int rc) {
    std::fclose(f);
    return rc;
  };
  if (fseeko(f, 0, SEEK_END) != 0) return fail(MSM_AMD_FILE_OPEN_ERROR);
  const long long file_size = ftello(f);
  if (fseeko(f, 0, SEEK_SET) != 0) return fail(MSM_AMD_FILE_OPEN_ERROR);
  uint64_t n_inst;
  if (!read_u64(f, &n_inst)) return fail(MSM_AMD_DESERIALIZATION_ERROR);
  // every instance needs at least its two length fields: bounds n_inst before anything is allocated
  if (n_inst > (uint64_t)(file_size - 8) / 16) return fail(MSM_AMD_DESERIALIZATION_ERROR);
  auto* h = new msm_amd_instance_file;
  h->f = f;
  long long pos = 8;
  for (uint64_t j = 0; j < n_inst; ++j) {
    uint64_t np, ns;
    if (fseeko(f, pos, SEEK_SET) != 0 || !read_u64(f, &np) ||
        np > (uint64_t)(file_size - pos - 8) / kPointRec) {
      delete h;
      return fail(MSM_AMD_DESERIALIZATION_ERROR);
    }
    const long long spos = pos + 8 + (long long)(np * kPointRec);
    if (fseeko(f, spos, SEEK_SET) != 0 || !read_u64(f, &ns) ||
        ns > (uint64_t)(file_size - spos - 8) / kScalarRec) {
      delete h;
      return fail(MSM_AMD_DESERIALIZATION_ERROR);
    }
    if (ns != np) {   // the reference asserts points.len() == scalars.len() (preprocess.rs:78)
      delete h;
      return fail(MSM_AMD_INVALID_DATA);
    }
    h->n.push_back(np);
    h->offset.push_back(pos);
    pos = spos + 8 + (long long)(ns * kScalarRec);
  }
  *out = h;
  return MSM_AMD_OK;
}

size_t msm_amd_instances_count(const msm_amd_instance_file* f) { return f ? f->n.size() : 0; }

size_t msm_amd_instances_size(const msm_amd_instance_file* f, size_t j) {
  return (f && j < f->n.size()) ? (size_t)f->n[j] : 0;
}

int msm_amd_instances_read(msm_amd_instance_file* f, size_t j, void* points_out, void* scalars_out) {
  if (!f || j >= f->n.size() || (f->n[j] && (!points_out || !scalars_out))) return MSM_AMD_INPUT_ERROR;
  const size_t n = (size_t)f->n[j];
  if (fseeko(f->f, f->offset[j] + 8, SEEK_SET) != 0) return MSM_AMD_FILE_OPEN_ERROR;
  int rc = read_records(f->f, n, kPointLimbs, (uint8_t*)points_out);
  if (rc) return rc;
  uint64_t ns;
  if (!read_u64(f->f, &ns) || ns != n) return MSM_AMD_DESERIALIZATION_ERROR;
  return read_records(f->f, n, kScalarLimbs, (uint8_t*)scalars_out);
}

void msm_amd_instances_close(msm_amd_instance_file* f) {
  if (!f) return;
  if (f->f) std::fclose(f->f);
  delete f;
}

size_t msm_amd_instances_default_path(const char* dir, uint32_t log_size, uint32_t num_instances, char* buf,
                                      size_t buf_len) {
  std::string d;
  if (dir) {
    d = dir;
  } else {   // default_msm_vec_repo (preprocess.rs:204-212)
    const char* home = std::getenv("HOME");
    if (!home) home = std::getenv("USERPROFILE");
    d = std::string(home ? home : "/tmp") + "/.msm_gpu_acceleration/msm_vecs";
  }
  const std::string p = d + "/msm_" + std::to_string(log_size) + "x" + std::to_string(num_instances) + ".bin";
  if (!buf || p.size() + 1 > buf_len) return 0;
  std::memcpy(buf, p.c_str(), p.size() + 1);
  return p.size();
}

int msm_amd_to_wire(int scalar_layout, int point_layout, const void* scalars, const void* points, size_t n,
                    void* scalars_be32_out, void* points_be32_out) {
  if (n && ((scalars && !scalars_be32_out) || (points && !points_be32_out))) return MSM_AMD_INPUT_ERROR;
  if (scalars) {
    const uint8_t* in = (const uint8_t*)scalars;
    uint8_t* out = (uint8_t*)scalars_be32_out;
    switch (scalar_layout) {
      case MSM_AMD_SCALAR_MONT_LE:   // Fr::to_bytes() of the h2c path, into_bigint() of the ark path
        for (size_t i = 0; i < n; ++i) {
          const u256 k = Fr::from_mont(load_le(in + 32 * i));
          le_to_be32((const uint8_t*)k.v, out + 32 * i);
        }
        break;
      case MSM_AMD_SCALAR_CANON_LE:
        for (size_t i = 0; i < n; ++i) le_to_be32(in + 32 * i, out + 32 * i);
        break;
      case MSM_AMD_SCALAR_CANON_BE32:
        std::memmove(out, in, 32 * n);
        break;
      default:
        return MSM_AMD_INPUT_ERROR;
    }
  }
  if (points) {
    const uint8_t* in = (const uint8_t*)points;
    uint8_t* out = (uint8_t*)points_be32_out;
    uint8_t one_be[32], zero_be[32] = {0};
    {
      const u256 one = msm_amd::Fq::one();
      le_to_be32((const uint8_t*)one.v, one_be);
    }
    switch (point_layout) {
      case MSM_AMD_POINT_H2C_AFFINE:   // (x, y, Mont(1)); the identity (0, 0) becomes z = 0
        for (size_t i = 0; i < n; ++i) {
          const uint8_t* p = in + 64 * i;
          le_to_be32(p, out + 96 * i);
          le_to_be32(p + 32, out + 96 * i + 32);
          std::memcpy(out + 96 * i + 64, (is_zero32(p) && is_zero32(p + 32)) ? zero_be : one_be, 32);
        }
        break;
      case MSM_AMD_POINT_ARK_AFFINE:   // {x, y, infinity: bool} padded to 72 B -> into_group()
        for (size_t i = 0; i < n; ++i) {
          const uint8_t* p = in + 72 * i;
          le_to_be32(p, out + 96 * i);
          le_to_be32(p + 32, out + 96 * i + 32);
          std::memcpy(out + 96 * i + 64, p[64] ? zero_be : one_be, 32);
        }
        break;
      case MSM_AMD_POINT_ARK_PROJECTIVE:
        for (size_t i = 0; i < 3 * n; ++i) le_to_be32(in + 32 * i, out + 32 * i);
        break;
      case MSM_AMD_POINT_JAC_BE32:
        std::memmove(out, in, 96 * n);
        break;
      default:
        return MSM_AMD_INPUT_ERROR;
    }
  }
  return MSM_AMD_OK;
}

int msm_amd_from_wire(int scalar_layout, int point_layout, const void* scalars_be32, const void* points_be32,
                      size_t n, void* scalars_out, void* points_out) {
  if (n && ((scalars_be32 && !scalars_out) || (points_be32 && !points_out))) return MSM_AMD_INPUT_ERROR;
  if (scalars_be32) {
    const uint8_t* in = (const uint8_t*)scalars_be32;
    uint8_t* out = (uint8_t*)scalars_out;
    switch (scalar_layout) {
      case MSM_AMD_SCALAR_MONT_LE:
        for (size_t i = 0; i < n; ++i) {
          u256 k;
          le_to_be32(in + 32 * i, (uint8_t*)k.v);   // the limb reversal is its own inverse
          k = Fr::to_mont(k);
          std::memcpy(out + 32 * i, k.v, 32);
        }
        break;
      case MSM_AMD_SCALAR_CANON_LE:
        for (size_t i = 0; i < n; ++i) le_to_be32(in + 32 * i, out + 32 * i);
        break;
      case MSM_AMD_SCALAR_CANON_BE32:
        std::memmove(out, in, 32 * n);
        break;
      default:
        return MSM_AMD_INPUT_ERROR;
    }
  }
  if (points_be32) {
    const uint8_t* in = (const uint8_t*)points_be32;
    uint8_t* out = (uint8_t*)points_out;
    switch (point_layout) {
      case MSM_AMD_POINT_ARK_PROJECTIVE:
        for (size_t i = 0; i < 3 * n; ++i) le_to_be32(in + 32 * i, out + 32 * i);
        break;
      case MSM_AMD_POINT_JAC_BE32:
        std::memmove(out, in, 96 * n);
        break;
      default:   // affine outputs would need a normalisation; callers feed JAC_BE32 to the MSM directly
        return MSM_AMD_INPUT_ERROR;
    }
  }
  return MSM_AMD_OK;
}

}  // extern "C"
