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
// K13-K16 fused: xyz + integer id planes -> the per-point class scores of the argmax camera, straight.
// FSF.img_cross_attn (FSF.py:694-728) needs, per point, only the ids of ONE camera (the one with the largest id sum,
// :716-718) turned into scores (get_all_cls_preds_2d :506-535, encode_preds_2d :472-473); the [n, ncam, ncls] int64 tensor
// the reference materialises (480 B per point, written by one kernel and re-read by the next) never exists here.  Also
// emits the "inside any mask" flag (obj_id.sum((-2, -1)) > 0, FSF.py:299-308) so that the camera-query branch gathers ids
// for the foreground points only.  One thread per point, the cameras in order, the best camera's ids in registers.
namespace fsf {
constexpr int PS_MAX_CLS = 16;

struct ProjScoreArgs {
  const float* xyz;
  const float* lidar2img;
  const void* mask;
  const float* anno;
  float* score;
  int64_t* ids;
  unsigned char* fg;
  int64_t n;
  int stride, ncam, ncls, H, W, num_anno, anno_dim, score_col;
};

template <typename MaskT>
__global__ void __launch_bounds__(256) project_score_kernel(ProjScoreArgs a) {
  extern __shared__ float s_mat[];
  for (int t = threadIdx.x; t < a.ncam * 12; t += blockDim.x) s_mat[t] = a.lidar2img[(t / 12) * 16 + t % 12];
  __syncthreads();
  const MaskT* mask = reinterpret_cast<const MaskT*>(a.mask);
  const float fw = (float)a.W, fh = (float)a.H;
  const int64_t plane = (int64_t)a.H * a.W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = a.xyz + i * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    int best[PS_MAX_CLS];
#pragma unroll
    for (int k = 0; k < PS_MAX_CLS; ++k) best[k] = 0;
    int64_t best_sum = INT64_MIN;
    bool any = false;
    for (int cam = 0; cam < a.ncam; ++cam) {
      const float* m = s_mat + cam * 12;
      float px = proj_row(m, x, y, z), py = proj_row(m + 4, x, y, z), pz = proj_row(m + 8, x, y, z);
      const bool depth_valid = pz > 1e-3f;
      pz = fminf(fmaxf(pz, 1e-5f), 1e5f);
      px = __fdiv_rn(__fdiv_rn(px, pz), fw);
      py = __fdiv_rn(__fdiv_rn(py, pz), fh);
      float gx = __fmul_rn(__fsub_rn(px, 0.5f), 2.0f), gy = __fmul_rn(__fsub_rn(py, 0.5f), 2.0f);
      const bool valid = depth_valid && gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;
      if (!valid) {
        gx = -2.0f;
        gy = -2.0f;
      }
      const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), __fdiv_rn(fw, 2.0f)), 0.5f);
      const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), __fdiv_rn(fh, 2.0f)), 0.5f);
      const float rx = rintf(ix), ry = rintf(iy);
      const bool inb = valid && rx >= 0.0f && rx < fw && ry >= 0.0f && ry < fh;
      int cur[PS_MAX_CLS];
      int64_t sum = 0;
      const MaskT* mc = mask + (int64_t)cam * a.ncls * plane + (inb ? (int64_t)ry * a.W + (int64_t)rx : 0);
#pragma unroll
      for (int k = 0; k < PS_MAX_CLS; ++k) {
        cur[k] = (inb && k < a.ncls) ? (int)mc[(int64_t)k * plane] : 0;
        sum += cur[k];
      }
      any |= sum > 0;
      if (sum > best_sum) {  // strict: the first maximum wins, like torch.max(dim)[1]
        best_sum = sum;
#pragma unroll
        for (int k = 0; k < PS_MAX_CLS; ++k) best[k] = cur[k];
      }
    }
#pragma unroll
    for (int k = 0; k < PS_MAX_CLS; ++k) {
      if (k < a.ncls) {
        const int id = best[k];
        if (a.ids) a.ids[i * a.ncls + k] = id;
        a.score[i * a.ncls + k] = (id > 0 && id <= a.num_anno) ? a.anno[(int64_t)(id - 1) * a.anno_dim + a.score_col] : 0.0f;
      }
    }
    if (a.fg) a.fg[i] = any ? 1 : 0;
  }
}
}  // namespace fsf

extern "C" int fsf_project_score(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam,
                                 const void* mask, int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w,
                                 const float* mask_anno, int32_t num_anno, int32_t anno_dim, int32_t score_col, float* out_score,
                                 int64_t* out_ids, uint8_t* out_fg, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || xyz_stride < 3 || ncam < 1 || ncam > 64 || ncls < 1 || img_h < 1 || img_w < 1 || !lidar2img || !mask ||
      (elem_bytes != 1 && elem_bytes != 4) || num_anno < 0 || anno_dim < 1 || score_col < 0 || score_col >= anno_dim ||
      (num_anno > 0 && !mask_anno) || (n > 0 && (!xyz || !out_score)))
    return FSF_ERR_INVALID_ARG;
  if (ncls > fsf::PS_MAX_CLS) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  fsf::ProjScoreArgs a{xyz, lidar2img, mask, mask_anno, out_score, out_ids, out_fg, n, (int)xyz_stride, (int)ncam, (int)ncls,
                       (int)img_h, (int)img_w, (int)num_anno, (int)anno_dim, (int)score_col};
  const int grid = fsf_stream_grid(n, 256);
  const size_t shmem = (size_t)ncam * 12 * sizeof(float);
  if (elem_bytes == 1)
    hipLaunchKernelGGL((fsf::project_score_kernel<uint8_t>), dim3(grid), dim3(256), shmem, stream, a);
  else
    hipLaunchKernelGGL((fsf::project_score_kernel<int32_t>), dim3(grid), dim3(256), shmem, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// K13b: LiDAR -> camera projection + per-point BILINEAR image-feature gather (BASELINE.json north_star: "LiDAR->camera
// projection with per-point bilinear image-feature gather").  The reference itself only gathers instance ids with
// grid_sample(nearest) (FSF.py:216-225); this is the same projection (prj_points_2d, FSF.py:169-200: fma chain, depth /
// image validity, -2 for invalid) followed by F.grid_sample(feat, grid, mode='bilinear', align_corners=False,
// padding_mode='zeros') semantics: ix = ((g + 1) * W - 1) / 2, the four corners weighted, corners outside the map
// contribute zero, an invalid projection samples nothing.
//   feature map f32 [ncam, C, Hf, Wf] (channels_last = 0, what a conv backbone emits) or [ncam, Hf, Wf, C] (channels_last = 1:
//   a pixel's channels contiguous, the coalesced layout); out f32 [n, ncam, C] (reduce = 0) or [n, C] = sum over the cameras that
//   see the point (reduce = 1); count u8 [n] = cameras that see the point (optional).
// One 16-lane team per (point, camera): lanes walk the channels four at a time (float4 in the channels-last layout).
namespace fsf {
struct BilinArgs {
  const float* xyz;
  const float* lidar2img;
  const float* feat;
  float* out;
  unsigned char* count;
  int64_t n;
  int stride, ncam, C, Hf, Wf, img_h, img_w, channels_last, reduce;
};

__global__ void __launch_bounds__(256) project_bilinear_kernel(BilinArgs a) {
  extern __shared__ float s_mat[];
  for (int t = threadIdx.x; t < a.ncam * 12; t += blockDim.x) s_mat[t] = a.lidar2img[(t / 12) * 16 + t % 12];
  __syncthreads();
  const int tl = threadIdx.x & 15;
  const float fw = (float)a.img_w, fh = (float)a.img_h;
  const int64_t plane = (int64_t)a.Hf * a.Wf;
  for (int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < a.n; i += (int64_t)gridDim.x * 16) {
    const float* p = a.xyz + i * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    int seen = 0;
    for (int cam = 0; cam < a.ncam; ++cam) {
      const float* m = s_mat + cam * 12;
      float px = proj_row(m, x, y, z), py = proj_row(m + 4, x, y, z), pz = proj_row(m + 8, x, y, z);
      const bool depth_valid = pz > 1e-3f;
      pz = fminf(fmaxf(pz, 1e-5f), 1e5f);
      px = __fdiv_rn(__fdiv_rn(px, pz), fw);
      py = __fdiv_rn(__fdiv_rn(py, pz), fh);
      const float gx = __fmul_rn(__fsub_rn(px, 0.5f), 2.0f), gy = __fmul_rn(__fsub_rn(py, 0.5f), 2.0f);
      const bool valid = depth_valid && gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;
      seen += valid ? 1 : 0;
      // grid_sample, align_corners = False
      const float ix = ((gx + 1.0f) * (float)a.Wf - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)a.Hf - 1.0f) * 0.5f;
      const float x0f = floorf(ix), y0f = floorf(iy);
      const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
      const int x0 = (int)x0f, y0 = (int)y0f;
      const bool inx0 = x0 >= 0 && x0 < a.Wf, inx1 = x0 + 1 >= 0 && x0 + 1 < a.Wf;
      const bool iny0 = y0 >= 0 && y0 < a.Hf, iny1 = y0 + 1 >= 0 && y0 + 1 < a.Hf;
      const float w00 = valid && inx0 && iny0 ? wx0 * wy0 : 0.0f, w01 = valid && inx1 && iny0 ? wx1 * wy0 : 0.0f;
      const float w10 = valid && inx0 && iny1 ? wx0 * wy1 : 0.0f, w11 = valid && inx1 && iny1 ? wx1 * wy1 : 0.0f;
      const int xa = min(max(x0, 0), a.Wf - 1), xb = min(max(x0 + 1, 0), a.Wf - 1);
      const int ya = min(max(y0, 0), a.Hf - 1), yb = min(max(y0 + 1, 0), a.Hf - 1);
      float* o = a.reduce ? a.out + i * a.C : a.out + (i * a.ncam + cam) * a.C;
      const bool touch = w00 != 0.0f || w01 != 0.0f || w10 != 0.0f || w11 != 0.0f;
      if (a.channels_last && (a.C & 3) == 0) {
        const float* f = a.feat + (int64_t)cam * plane * a.C;
        for (int c = tl * 4; c < a.C; c += 64) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (touch) {
            const float4 v00 = *reinterpret_cast<const float4*>(f + ((int64_t)ya * a.Wf + xa) * a.C + c);
            const float4 v01 = *reinterpret_cast<const float4*>(f + ((int64_t)ya * a.Wf + xb) * a.C + c);
            const float4 v10 = *reinterpret_cast<const float4*>(f + ((int64_t)yb * a.Wf + xa) * a.C + c);
            const float4 v11 = *reinterpret_cast<const float4*>(f + ((int64_t)yb * a.Wf + xb) * a.C + c);
            v.x = v00.x * w00 + v01.x * w01 + v10.x * w10 + v11.x * w11;
            v.y = v00.y * w00 + v01.y * w01 + v10.y * w10 + v11.y * w11;
            v.z = v00.z * w00 + v01.z * w01 + v10.z * w10 + v11.z * w11;
            v.w = v00.w * w00 + v01.w * w01 + v10.w * w10 + v11.w * w11;
          }
          float4* dst = reinterpret_cast<float4*>(o + c);
          if (a.reduce && cam > 0) {
            const float4 prev = *dst;
            v.x += prev.x; v.y += prev.y; v.z += prev.z; v.w += prev.w;
          }
          *dst = v;
        }
      } else {
        for (int c = tl; c < a.C; c += 16) {
          float v = 0.0f;
          if (touch) {
            const int64_t cs = a.channels_last ? 1 : plane, ps = a.channels_last ? a.C : 1;
            const float* f = a.feat + (int64_t)cam * plane * a.C + (int64_t)c * cs;
            v = f[((int64_t)ya * a.Wf + xa) * ps] * w00 + f[((int64_t)ya * a.Wf + xb) * ps] * w01 +
                f[((int64_t)yb * a.Wf + xa) * ps] * w10 + f[((int64_t)yb * a.Wf + xb) * ps] * w11;
          }
          if (a.reduce && cam > 0) v += o[c];
          o[c] = v;
        }
      }
    }
    if (a.count && tl == 0) a.count[i] = (unsigned char)seen;
  }
}
}  // namespace fsf

extern "C" int fsf_project_gather_bilinear(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam,
                                           const float* feat, int32_t channels, int32_t feat_h, int32_t feat_w,
                                           int32_t channels_last, int32_t img_h, int32_t img_w, int32_t reduce_cams, float* out,
                                           uint8_t* count, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || xyz_stride < 3 || ncam < 1 || ncam > 64 || channels < 1 || feat_h < 1 || feat_w < 1 || img_h < 1 || img_w < 1 ||
      !lidar2img || !feat || (n > 0 && (!xyz || !out)))
    return FSF_ERR_INVALID_ARG;
  if (channels_last && (channels % 4) == 0 && (((uintptr_t)feat % 16) != 0 || ((uintptr_t)out % 16) != 0)) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  fsf::BilinArgs a{xyz, lidar2img, feat, out, count, n, (int)xyz_stride, (int)ncam, (int)channels, (int)feat_h, (int)feat_w,
                   (int)img_h, (int)img_w, (int)(channels_last != 0), (int)(reduce_cams != 0)};
  int64_t g = (n + 15) / 16;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(fsf::project_bilinear_kernel, dim3((unsigned)g), dim3(256), (size_t)ncam * 12 * sizeof(float), stream, a);
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
