// K0: multi-sweep point-cloud assembly on the device.  See include/fsf_hip.h.
//
// Replaces, for the test-time pipeline of configs/_base_/datasets/nuscenes_dataloader.py:96-137, the host numpy / torch passes of
//   LoadPointsFromMultiSweeps  (projects/mmdet3d_plugin/datasets/pipelines/loading.py:825-877: close-point removal in the
//                               sensor frame, sweep -> key-frame transform, time lag, concatenation),
//   SaveNoAugPoints            (:341-354: the un-augmented xyz appended as three more columns),
//   PointsRangeFilter          (mmdet3d: strict in-range test on xyz),
//   NormalizePoints            (:537-563: (x - mean) / std on one column)
// with ONE host -> device copy of the raw sweep files and one count -> scan -> fill pass: rows keep the reference's order
// (key frame first, sweeps in list order, file order inside a sweep).  Arithmetic follows the reference's promotions so
// the result is bit-identical: the rotation is a float64 product (numpy promotes fp32 points @ float64 matrix) rounded to
// fp32 on assignment, the translation a float64 add rounded to fp32, the normalisation an fp32 subtract and divide.
#include "common.h"
#include "scan.h"

namespace fsf {

constexpr int AS_MAX_SWEEPS = 16;

struct AsArgs {
  const float* raw;
  float* out;
  int64_t n;
  int load_dim, out_dim, nsweeps, norm_col, use_range;
  float close_radius, norm_mean, norm_std;
  float range[6];
  int64_t offsets[AS_MAX_SWEEPS + 1];
  double params[AS_MAX_SWEEPS][13];  // R (row-major 3x3) | t | time lag
  unsigned char transform[AS_MAX_SWEEPS];     // 0: rows pass unchanged (the key frame, padded copies)
  unsigned char remove_close[AS_MAX_SWEEPS];
};

__device__ __forceinline__ int as_sweep_of(const AsArgs& a, int64_t i) {
  int s = 0;
  for (int k = 1; k < a.nsweeps; ++k) s += i >= a.offsets[k] ? 1 : 0;
  return s;
}

// xyz of row i in the key frame; returns false if the row is dropped
__device__ __forceinline__ bool as_row(const AsArgs& a, int64_t i, int s, float (&xyz)[3]) {
  const float* p = a.raw + i * a.load_dim;
  const float x = p[0], y = p[1], z = p[2];
  if (a.remove_close[s] && fabsf(x) < a.close_radius && fabsf(y) < a.close_radius) return false;  // sensor frame (:803-823)
  if (a.transform[s]) {
    const double* m = a.params[s];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double d = __dadd_rn(__dadd_rn(__dmul_rn((double)x, m[3 * r]), __dmul_rn((double)y, m[3 * r + 1])), __dmul_rn((double)z, m[3 * r + 2]));
      xyz[r] = (float)__dadd_rn((double)(float)d, m[9 + r]);
    }
  } else {
    xyz[0] = x; xyz[1] = y; xyz[2] = z;
  }
  if (a.use_range && !(xyz[0] > a.range[0] && xyz[1] > a.range[1] && xyz[2] > a.range[2] && xyz[0] < a.range[3] &&
                       xyz[1] < a.range[4] && xyz[2] < a.range[5]))
    return false;
  return true;
}

// The functors hold a POINTER to the argument block (a copy in the workspace): by value the block (~1.9 KB: 16 sweeps x 13
// doubles) rode in the kernel arguments twice (In and Out), ~3.9 KB against the 4 KB kernarg limit.
struct AsIn {
  const AsArgs* ap;
  __device__ uint32_t operator()(int64_t i) const {
    const AsArgs& a = *ap;
    float xyz[3];
    return as_row(a, i, as_sweep_of(a, i), xyz) ? 1u : 0u;
  }
};

struct AsOut {
  const AsArgs* ap;
  __device__ void operator()(int64_t i, uint32_t pos, uint32_t keep) const {
    if (!keep) return;
    const AsArgs& a = *ap;
    const int s = as_sweep_of(a, i);
    float xyz[3];
    as_row(a, i, s, xyz);
    const float* p = a.raw + i * a.load_dim;
    float* o = a.out + (int64_t)pos * a.out_dim;
    for (int c = 0; c < a.load_dim; ++c) {
      float v = c < 3 ? xyz[c] : p[c];
      if (c == 4) v = (float)a.params[s][12];  // time lag: 0 for the key frame (:843) and for padded copies of it
      if (c == a.norm_col) v = __fdiv_rn(__fsub_rn(v, a.norm_mean), a.norm_std);
      o[c] = v;
    }
    o[a.load_dim] = xyz[0];  // SaveNoAugPoints: the xyz before any augmentation
    o[a.load_dim + 1] = xyz[1];
    o[a.load_dim + 2] = xyz[2];
  }
};

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_assemble_sweeps_workspace_bytes(int64_t n_rows) {
  return fsf_align_up(scan_num_tiles(n_rows) * 4, 256) + 256 + fsf_align_up((int64_t)sizeof(AsArgs), 256);
}

extern "C" int fsf_assemble_sweeps(const float* raw, int64_t n_rows, int32_t load_dim, const int64_t* sweep_offsets, int32_t num_sweeps,
                                   const double* sweep_params, const uint8_t* sweep_transform, const uint8_t* sweep_remove_close,
                                   float close_radius, const float* pc_range, int32_t norm_col, float norm_mean, float norm_std,
                                   float* out, int64_t* count_dev, int64_t* count_host, void* workspace, int64_t workspace_bytes,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || load_dim < 4 || load_dim > 16 || num_sweeps < 1 || num_sweeps > AS_MAX_SWEEPS || !sweep_offsets || !sweep_params ||
      !sweep_transform || !sweep_remove_close || norm_col >= load_dim || (norm_col >= 0 && norm_std == 0.0f) ||
      (n_rows > 0 && (!raw || !out)) || (!count_dev && !count_host))
    return FSF_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < fsf_assemble_sweeps_workspace_bytes(n_rows)) return FSF_ERR_WORKSPACE;
  AsArgs a;
  a.raw = raw; a.out = out; a.n = n_rows;
  a.load_dim = load_dim; a.out_dim = load_dim + 3; a.nsweeps = num_sweeps; a.norm_col = norm_col;
  a.use_range = pc_range ? 1 : 0;
  a.close_radius = close_radius; a.norm_mean = norm_mean; a.norm_std = norm_std;
  for (int k = 0; k < 6; ++k) a.range[k] = pc_range ? pc_range[k] : 0.0f;
  for (int s = 0; s <= num_sweeps; ++s) a.offsets[s] = sweep_offsets[s];
  if (a.offsets[0] != 0 || a.offsets[num_sweeps] != n_rows) return FSF_ERR_INVALID_ARG;
  for (int s = 0; s < num_sweeps; ++s) {
    if (a.offsets[s + 1] < a.offsets[s]) return FSF_ERR_INVALID_ARG;
    for (int k = 0; k < 13; ++k) a.params[s][k] = sweep_params[s * 13 + k];
    a.transform[s] = sweep_transform[s];
    a.remove_close[s] = sweep_remove_close[s];
  }
  FsfArena arena(workspace, workspace_bytes);
  uint32_t* tile_sums = arena.take<uint32_t>(scan_num_tiles(n_rows));
  int64_t* total = arena.take<int64_t>(1);
  AsArgs* a_dev = reinterpret_cast<AsArgs*>(arena.take<char>((int64_t)sizeof(AsArgs)));
  if (!arena.ok()) return FSF_ERR_WORKSPACE;
  int64_t* tot = count_dev ? count_dev : total;
  FSF_HIP_TRY(hipMemcpyAsync(a_dev, &a, sizeof(AsArgs), hipMemcpyHostToDevice, stream));  // (pageable source: staged before the call returns)
  int rc = exclusive_scan_u32(AsIn{a_dev}, AsOut{a_dev}, n_rows, tile_sums, nullptr, tot, stream);
  if (rc != FSF_OK) return rc;
  if (count_host) {
    FSF_READ_BACK(count_host, tot, sizeof(int64_t), stream);
  }
  return FSF_OK;
}
