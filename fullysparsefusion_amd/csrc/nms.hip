// K20: greedy BEV non-maximum suppression over rotated (or axis-aligned) boxes.
// Replaces: mmdet3d.ops.iou3d nms_gpu / nms_normal_gpu [UNVENDORED mmdet3d 0.x iou3d_kernel.cu], reached through
//   box3d_multiclass_nms from FrustumClusterHead._get_bboxes_single
//   (projects/mmdet3d_plugin/models/dense_heads/frustum_cluster_head.py:636-667) and
//   SparseClusterHeadV2 (sparse_cluster_head_v2.py:538-609).
// Input boxes are (x1, y1, x2, y2, ry): the axis-aligned extent of the box BEFORE rotation plus the yaw, already
//   sorted by descending score (mmdet3d.core.xywhr2xyxyr of LiDARInstance3DBoxes.bev).  IoU = overlap / max(sa + sb -
//   overlap, 1e-8); box j is suppressed by an earlier kept box i iff IoU(i, j) > thresh.
// Upstream builds a 64x64-blocked suppression bitmask on the GPU, copies it to the host and runs the greedy scan on
//   the CPU.  Here the scan stays on the device (one workgroup, the `removed` bitset in LDS), so a head's NMS is two
//   launches and no host round trip.  The overlap of two rotated rectangles is computed by clipping A's corners
//   (moved into B's frame) against B's four sides (Sutherland-Hodgman) — no intersection-point sort, no degenerate
//   cases — which gives the same area as upstream's intersect-and-sort polygon up to fp32 rounding.
#include "common.h"

namespace fsf {

struct NmsArgs {
  const float* boxes;  // [n,5]
  int64_t n;
  float thresh;
  int rotated;
  uint64_t* mask;  // [n][words]
  int words;
  int64_t* keep;
  int64_t* num_keep;
};

__device__ __forceinline__ float rect_overlap_rotated(const float* a, const float* b) {
  // B frame: origin at B's centre, axes along B's sides
  const float bcx = 0.5f * (b[0] + b[2]), bcy = 0.5f * (b[1] + b[3]);
  const float bhx = 0.5f * (b[2] - b[0]), bhy = 0.5f * (b[3] - b[1]);
  const float acx = 0.5f * (a[0] + a[2]), acy = 0.5f * (a[1] + a[3]);
  const float ahx = 0.5f * (a[2] - a[0]), ahy = 0.5f * (a[3] - a[1]);
  if (!(bhx > 0.f) || !(bhy > 0.f) || !(ahx > 0.f) || !(ahy > 0.f)) return 0.f;
  const float ca = cosf(a[4]), sa = sinf(a[4]), cb = cosf(b[4]), sb = sinf(b[4]);
  // Upstream's corner rotation (iou3d rotate_around_center, the mmdet3d 0.x yaw sense, the same one K17's
  // lidar_to_local_coords implies): corner = centre + M(yaw) * offset with M = [[cos, sin], [-sin, cos]].
  // Polygon = A's corners in the world, then into B's frame with M(yaw_b)^T.
  float px[8], py[8], qx[8], qy[8];
  const float ox[4] = {-ahx, ahx, ahx, -ahx}, oy[4] = {-ahy, -ahy, ahy, ahy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float wx = acx + ca * ox[k] + sa * oy[k] - bcx;
    const float wy = acy - sa * ox[k] + ca * oy[k] - bcy;
    px[k] = cb * wx - sb * wy;
    py[k] = sb * wx + cb * wy;
  }
  int n = 4;
  // clip against x <= bhx, x >= -bhx, y <= bhy, y >= -bhy
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    const float lim = (side < 2) ? bhx : bhy;
    const float sgn = (side & 1) ? -1.f : 1.f;
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const int k2 = (k + 1 == n) ? 0 : k + 1;
      const float c0 = sgn * ((side < 2) ? px[k] : py[k]);
      const float c1 = sgn * ((side < 2) ? px[k2] : py[k2]);
      const bool in0 = c0 <= lim, in1 = c1 <= lim;
      if (in0) {
        qx[m] = px[k];
        qy[m] = py[k];
        ++m;
      }
      if (in0 != in1) {
        const float t = (lim - c0) / (c1 - c0);
        qx[m] = px[k] + t * (px[k2] - px[k]);
        qy[m] = py[k] + t * (py[k2] - py[k]);
        ++m;
      }
    }
    n = m;
    for (int k = 0; k < n; ++k) {
      px[k] = qx[k];
      py[k] = qy[k];
    }
    if (n < 3) return 0.f;
  }
  float area = 0.f;
  for (int k = 1; k + 1 < n; ++k)
    area += (px[k] - px[0]) * (py[k + 1] - py[0]) - (px[k + 1] - px[0]) * (py[k] - py[0]);
  return 0.5f * fabsf(area);
}

__device__ __forceinline__ float rect_overlap_normal(const float* a, const float* b) {
  const float l = fmaxf(a[0], b[0]), r = fminf(a[2], b[2]);
  const float t = fmaxf(a[1], b[1]), d = fminf(a[3], b[3]);
  return fmaxf(r - l, 0.f) * fmaxf(d - t, 0.f);
}

__device__ __forceinline__ float iou_bev(const float* a, const float* b, int rotated) {
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float ov = rotated ? rect_overlap_rotated(a, b) : rect_overlap_normal(a, b);
  return ov / fmaxf(sa + sb - ov, 1e-8f);
}

// block (col word, row block): thread = row box, tests it against the 64 boxes of the column word
__global__ void __launch_bounds__(64) nms_mask_kernel(NmsArgs a) {
  const int col_blk = blockIdx.x, row_blk = blockIdx.y;
  if (col_blk < row_blk) return;  // a box is only suppressed by an earlier (higher-score) one
  __shared__ float cb[64 * 5];
  const int64_t c0 = (int64_t)col_blk * 64;
  const int ncol = (int)min((int64_t)64, a.n - c0);
  for (int t = threadIdx.x; t < ncol * 5; t += 64) cb[t] = a.boxes[c0 * 5 + t];
  __syncthreads();
  const int64_t i = (int64_t)row_blk * 64 + threadIdx.x;
  if (i >= a.n) return;
  float mine[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) mine[t] = a.boxes[i * 5 + t];
  uint64_t bits = 0;
  const int start = (row_blk == col_blk) ? threadIdx.x + 1 : 0;
  for (int j = start; j < ncol; ++j)
    if (iou_bev(mine, cb + j * 5, a.rotated) > a.thresh) bits |= 1ull << j;
  a.mask[i * a.words + col_blk] = bits;
}

// one workgroup: greedy scan in score order
__global__ void __launch_bounds__(256) nms_scan_kernel(NmsArgs a) {
  extern __shared__ uint64_t removed[];
  __shared__ int64_t nkeep;
  for (int w = threadIdx.x; w < a.words; w += 256) removed[w] = 0;
  if (threadIdx.x == 0) nkeep = 0;
  __syncthreads();
  for (int64_t i = 0; i < a.n; ++i) {
    const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;  // same value in every thread (read after a barrier)
    if (dead) continue;
    __syncthreads();  // everyone has read bit i before anyone updates the words
    if (threadIdx.x == 0) a.keep[nkeep++] = i;
    const uint64_t* row = a.mask + i * a.words;
    for (int w = (int)(i >> 6) + threadIdx.x; w < a.words; w += 256) removed[w] |= row[w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *a.num_keep = nkeep;
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_nms_bev_workspace_bytes(int64_t n) {
  const int64_t words = (n + 63) / 64;
  return fsf_align_up((n > 0 ? n : 1) * (words > 0 ? words : 1) * 8, 256) + 512;
}

extern "C" int fsf_nms_bev(const float* boxes, int64_t n, float thresh, int32_t rotated, int64_t* keep,
                           int64_t* num_keep_dev, int64_t* num_keep_host, void* workspace, int64_t workspace_bytes,
                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || (n > 0 && (!boxes || !keep)) || (!num_keep_dev && !num_keep_host)) return FSF_ERR_INVALID_ARG;
  const int64_t words = (n + 63) / 64;
  if (words * 8 > 60 * 1024) return FSF_ERR_UNSUPPORTED;  // `removed` bitset lives in LDS (n <= 491520)
  if (workspace_bytes < fsf_nms_bev_workspace_bytes(n) || !workspace) return FSF_ERR_WORKSPACE;
  FsfArena arena(workspace, workspace_bytes);
  uint64_t* mask = arena.take<uint64_t>((n > 0 ? n : 1) * (words > 0 ? words : 1));
  int64_t* tmp = arena.take<int64_t>(1);
  if (!arena.ok()) return FSF_ERR_WORKSPACE;
  int64_t* ndev = num_keep_dev ? num_keep_dev : tmp;
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(ndev, 0, sizeof(int64_t), stream));
  } else {
    NmsArgs a{boxes, n, thresh, (int)rotated, mask, (int)words, keep, ndev};
    // words below the diagonal are never written by the mask kernel and never read by the scan (w starts at i / 64)
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), (size_t)words * 8, stream, a);
    FSF_LAUNCH_CHECK();
  }
  if (num_keep_host) {
    FSF_HIP_TRY(hipMemcpyAsync(num_keep_host, ndev, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    FSF_HIP_TRY(hipStreamSynchronize(stream));
  }
  return FSF_OK;
}
