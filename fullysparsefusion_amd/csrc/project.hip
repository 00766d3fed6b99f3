// K13-K16: LiDAR -> camera projection with per-point instance-mask gather, camera select and 2-D score
// lookup.  See include/fsf_hip.h.  One thread per (point, camera); the mask is read in its stored integer
// type (no float copy).  Every fp32 operation is an explicit round-to-nearest op in the order the
// reference's elementwise PyTorch ops apply them (FSF.py:169-200), so pixel indices are bit-exact.
#include "common.h"

namespace fsf {

struct ProjArgs {
  const float* xyz;
  const float* lidar2img;
  const void* mask;
  int64_t* obj_id;
  float* pts_2d;
  int64_t n;
  int stride;
  int ncam, ncls, H, W;
};

// pts_4d @ lidar2img^T for one output row: ((x*m0 + y*m1) + z*m2) + m3 as an fma chain in k order,
// which is what the CPU sgemm micro-kernel of the in-container reference evaluates (see oracle/project.py).
__device__ __forceinline__ float proj_row(const float* m, float x, float y, float z) {
  float acc = __fmul_rn(x, m[0]);
  acc = __fmaf_rn(y, m[1], acc);
  acc = __fmaf_rn(z, m[2], acc);
  acc = __fmaf_rn(1.0f, m[3], acc);
  return acc;
}

template <typename MaskT>
__global__ void __launch_bounds__(256) project_gather_kernel(ProjArgs a) {
  extern __shared__ float s_mat[];  // ncam * 12 floats (rows 0..2 of each 4x4)
  for (int t = threadIdx.x; t < a.ncam * 12; t += blockDim.x) {
    const int cam = t / 12, r = t % 12;
    s_mat[t] = a.lidar2img[cam * 16 + r];
  }
  __syncthreads();
  const MaskT* mask = reinterpret_cast<const MaskT*>(a.mask);
  const float fw = (float)a.W, fh = (float)a.H;
  const int64_t total = a.n * a.ncam;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / a.ncam;
    const int cam = (int)(t - i * a.ncam);
    const float* p = a.xyz + i * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    const float* m = s_mat + cam * 12;
    float px = proj_row(m, x, y, z);
    float py = proj_row(m + 4, x, y, z);
    float pz = proj_row(m + 8, x, y, z);
    const bool depth_valid = pz > 1e-3f;
    pz = fminf(fmaxf(pz, 1e-5f), 1e5f);
    px = __fdiv_rn(px, pz);
    py = __fdiv_rn(py, pz);
    px = __fdiv_rn(px, fw);
    py = __fdiv_rn(py, fh);
    float gx = __fmul_rn(__fsub_rn(px, 0.5f), 2.0f);
    float gy = __fmul_rn(__fsub_rn(py, 0.5f), 2.0f);
    const bool valid = depth_valid && (gx > -1.0f) && (gx < 1.0f) && (gy > -1.0f) && (gy < 1.0f);
    if (!valid) {
      gx = -2.0f;
      gy = -2.0f;
    }
    if (a.pts_2d) {
      float2 g = make_float2(gx, gy);
      *reinterpret_cast<float2*>(a.pts_2d + ((int64_t)cam * a.n + i) * 2) = g;
    }
    int64_t* o = a.obj_id + t * a.ncls;
    // grid_sample(mode='nearest', align_corners=False, padding_mode='zeros'):
    // ix = (g + 1) * (W / 2) - 0.5, nearest = nearbyint (half-to-even), out of bounds -> 0
    const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), __fdiv_rn(fw, 2.0f)), 0.5f);
    const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), __fdiv_rn(fh, 2.0f)), 0.5f);
    const float rx = rintf(ix), ry = rintf(iy);
    const bool inb = valid && rx >= 0.0f && rx < fw && ry >= 0.0f && ry < fh;
    if (inb) {
      const int64_t pix = (int64_t)ry * a.W + (int64_t)rx;
      const int64_t plane = (int64_t)a.H * a.W;
      const MaskT* mc = mask + (int64_t)cam * a.ncls * plane + pix;
      for (int k = 0; k < a.ncls; ++k) o[k] = (int64_t)mc[(int64_t)k * plane];
    } else {
      for (int k = 0; k < a.ncls; ++k) o[k] = 0;
    }
  }
}

__global__ void __launch_bounds__(256)
    cam_select_score_kernel(const int64_t* __restrict__ obj_id, int64_t n, int ncam, int ncls,
                            const float* __restrict__ anno, int num_anno, int anno_dim, int score_col,
                            int64_t* __restrict__ out_ids, float* __restrict__ out_score) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t* ids = obj_id + i * ncam * ncls;
    int best = 0;
    int64_t best_sum = INT64_MIN;
    for (int c = 0; c < ncam; ++c) {
      int64_t s = 0;
      for (int k = 0; k < ncls; ++k) s += ids[c * ncls + k];
      if (s > best_sum) {  // strict: first maximum wins, like torch.max(dim)[1] on CPU
        best_sum = s;
        best = c;
      }
    }
    for (int k = 0; k < ncls; ++k) {
      const int64_t id = ids[best * ncls + k];
      if (out_ids) out_ids[i * ncls + k] = id;
      float sc = 0.0f;
      if (id > 0 && id <= num_anno) sc = anno[(id - 1) * anno_dim + score_col];
      out_score[i * ncls + k] = sc;
    }
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_project_gather_mask(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img,
                                       int32_t ncam, const void* mask, int32_t elem_bytes, int32_t ncls, int32_t img_h,
                                       int32_t img_w, int64_t* obj_id, float* pts_2d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || xyz_stride < 3 || ncam < 1 || ncam > 64 || ncls < 1 || img_h < 1 || img_w < 1 || !lidar2img || !mask ||
      (elem_bytes != 1 && elem_bytes != 4) || (n > 0 && (!xyz || !obj_id)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  ProjArgs a{xyz, lidar2img, mask, obj_id, pts_2d, n, (int)xyz_stride, (int)ncam, (int)ncls, (int)img_h, (int)img_w};
  const int grid = fsf_stream_grid(n * ncam, 256) * 4 > 8192 ? 8192 : fsf_stream_grid(n * ncam, 256) * 4;
  const size_t shmem = (size_t)ncam * 12 * sizeof(float);
  if (elem_bytes == 1)
    hipLaunchKernelGGL((project_gather_kernel<uint8_t>), dim3(grid), dim3(256), shmem, stream, a);
  else
    hipLaunchKernelGGL((project_gather_kernel<int32_t>), dim3(grid), dim3(256), shmem, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_cam_select_score(const int64_t* obj_id, int64_t n, int32_t ncam, int32_t ncls,
                                    const float* mask_anno, int32_t num_anno, int32_t anno_dim, int32_t score_col,
                                    int64_t* out_ids, float* out_score, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || ncam < 1 || ncls < 1 || num_anno < 0 || anno_dim < 1 || score_col < 0 || score_col >= anno_dim ||
      (n > 0 && (!obj_id || !out_score)) || (num_anno > 0 && !mask_anno))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  hipLaunchKernelGGL(cam_select_score_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, obj_id, n, (int)ncam,
                     (int)ncls, mask_anno, (int)num_anno, (int)anno_dim, (int)score_col, out_ids, out_score);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// The k largest ids of every row, descending — `obj_id_tensor[mask].topk(k, dim=-1)[0]` of FSF.double_overlap_pts
// (projects/mmdet3d_plugin/models/detectors/FSF.py:284-286): rows are the 60 (camera, class) slots of a point, at most
// a handful non-zero.  ATen's generic radix top-k spends ~0.2 ms per call on this shape; a team of 16 lanes per row with
// a k-step "take the max, knock it out" loop is one coalesced read of the row.
namespace fsf {
__global__ void __launch_bounds__(256) row_topk_kernel(const int64_t* __restrict__ x, int64_t n, int w, int k,
                                                       int64_t* __restrict__ out) {
  const int tl = threadIdx.x & 15;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < n; row += (int64_t)gridDim.x * 16) {
    constexpr int PER = 8;  // w <= 128
    int64_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c = tl + 16 * j;
      v[j] = c < w ? x[row * w + c] : INT64_MIN;
    }
    for (int t = 0; t < k; ++t) {
      int64_t best = INT64_MIN;
      int where = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (v[j] > best) {
          best = v[j];
          where = tl + 16 * j;  // column of this lane's candidate
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {  // team arg-max; ties -> the lower column (torch.topk's value order is unaffected)
        const int64_t ob = __shfl_xor(best, o, 16);
        const int ow = __shfl_xor(where, o, 16);
        if (ob > best || (ob == best && ow < where)) {
          best = ob;
          where = ow;
        }
      }
      if (tl == 0) out[row * k + t] = best;
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (tl + 16 * j == where) v[j] = INT64_MIN;
    }
  }
}
}  // namespace fsf

extern "C" int fsf_row_topk_desc(const int64_t* x, int64_t n, int32_t w, int32_t k, int64_t* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || w < 1 || k < 1 || k > w || (n > 0 && (!x || !out))) return FSF_ERR_INVALID_ARG;
  if (w > 128) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  int64_t g = (n + 15) / 16;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(fsf::row_topk_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, n, (int)w, (int)k, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
