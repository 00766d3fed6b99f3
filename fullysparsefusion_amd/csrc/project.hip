// K13-K16: LiDAR -> camera projection with per-point instance-mask gather, camera select and 2-D score
// lookup.  See include/fsf_hip.h.  One thread per (point, camera); the mask is read in its stored integer
// type (no float copy).  Every fp32 operation is an explicit round-to-nearest op in the order the
// reference's elementwise PyTorch ops apply them (FSF.py:169-200), so pixel indices are bit-exact.
#include "common.h"
#include "radix_sort.h"
#include "scan.h"

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
  unsigned char* count;
  int32_t* max_id;
  int64_t n;
  int stride, ncam, ncls, H, W, num_anno, anno_dim, score_col;
};


// One camera of FSF.prj_points_2d (FSF.py:169-200) + the nearest-pixel rule of grid_sample(mode='nearest', align_corners=False,
// padding_mode='zeros') that points_in_mask (:202-226) applies: true and *pix = row * W + column if the point lands inside the image.
__device__ __forceinline__ bool proj_pixel(const float* m, float x, float y, float z, float fw, float fh, int W, int64_t* pix) {
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
  *pix = inb ? (int64_t)ry * W + (int64_t)rx : 0;
  return inb;
}

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
    int positive = 0, top = 0;  // cells with an id > 0 over ALL cameras, and the largest id (obj_id_tensor.max(-1), FSF.py:263)
    for (int cam = 0; cam < a.ncam; ++cam) {
      int64_t pix;
      const bool inb = proj_pixel(s_mat + cam * 12, x, y, z, fw, fh, a.W, &pix);
      int cur[PS_MAX_CLS];
      int64_t sum = 0;
      const MaskT* mc = mask + (int64_t)cam * a.ncls * plane + (inb ? pix : 0);
#pragma unroll
      for (int k = 0; k < PS_MAX_CLS; ++k) {
        cur[k] = (inb && k < a.ncls) ? (int)mc[(int64_t)k * plane] : 0;
        sum += cur[k];
        positive += cur[k] > 0 ? 1 : 0;
        top = cur[k] > top ? cur[k] : top;
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
    if (a.count) a.count[i] = (unsigned char)(positive > 255 ? 255 : positive);
    if (a.max_id) a.max_id[i] = top;
  }
}
}  // namespace fsf

extern "C" int fsf_project_score(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam,
                                 const void* mask, int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w,
                                 const float* mask_anno, int32_t num_anno, int32_t anno_dim, int32_t score_col, float* out_score,
                                 int64_t* out_ids, uint8_t* out_fg, uint8_t* out_count, int32_t* out_max_id, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || xyz_stride < 3 || ncam < 1 || ncam > 64 || ncls < 1 || img_h < 1 || img_w < 1 || !lidar2img || !mask ||
      (elem_bytes != 1 && elem_bytes != 4) || num_anno < 0 || anno_dim < 1 || score_col < 0 || score_col >= anno_dim ||
      (num_anno > 0 && !mask_anno) || (n > 0 && (!xyz || !out_score)))
    return FSF_ERR_INVALID_ARG;
  if (ncls > fsf::PS_MAX_CLS) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  fsf::ProjScoreArgs a{xyz, lidar2img, mask, mask_anno, out_score, out_ids, out_fg, out_count, out_max_id, n, (int)xyz_stride, (int)ncam, (int)ncls,
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

// ----------------------------------------------------------------------------------------------------------------
// K26: the row list of the camera-query branch in two calls and ONE read-back.
// FSF.frustum_pooling (FSF.py:388-437) keeps the points inside any mask (extract_fg_pts :299-308), then double_overlap_pts
// (:260-297) appends, for k = 2, 3, ... and within k for j = 1 .. k - 1, one more row per point that lies inside k masks, carrying the
// j-th largest of its ids (row 0 of topk went to the point's own row), and get_sir_coors (:373-376) makes the (batch, 0, id) keys.
// Upstream: a boolean-mask compaction of five tensors per k (a host sync each).  The plugin's ATen form of it: nonzero, the ids of the
// foreground points as an [F, ncam * ncls] int64 tensor, ~45 small launches and three host syncs.  Here:
//   fsf_overlap_plan:  from K13-K16's per-point cell count: scan of the foreground flag (-> the ascending foreground list), ONE 8-bit
//                      stable radix pass over (k if k >= 2 else 0) (-> the points of each k in ascending order), a 256-thread kernel that
//                      turns the sorted keys into per-k starts / counts / row bases; reads back (F, M, T);
//   fsf_overlap_rows:  row r < F: point fg[r], id = its largest id; the rows of a point with k cells, rank q among the points of its k:
//                      F + base[k] + (j - 1) * count[k] + q, id = its j-th largest (the mask cells are re-read: 1-byte L2 hits).
namespace fsf {

struct OvIn {
  const uint8_t* fg;
  const uint8_t* count;
  uint64_t* keys;
  uint32_t* vals;
  __device__ uint32_t operator()(int64_t i) const {
    const bool f = fg[i] != 0;
    const uint32_t k = count[i];
    keys[i] = (f && k >= 2u) ? (uint64_t)k : 0ull;  // (idempotent side effect: the 3-launch scan calls this twice)
    vals[i] = (uint32_t)i;
    return f ? 1u : 0u;
  }
};
struct OvOut {
  uint32_t* fg_idx;
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t v) const {
    if (v) fg_idx[excl] = (uint32_t)i;
  }
};

// table: int32 [3][256] = start / count / base per k; ret: int64 [4] = F (already there), M, T, saturated-count flag
__global__ void __launch_bounds__(256) ov_plan_kernel(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ table, int64_t* __restrict__ ret) {
  __shared__ int64_t s_start[257];
  __shared__ int64_t s_rows[256];
  const int k = threadIdx.x;
  int64_t lo = 0, hi = n;  // first position with key >= k
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (uint64_t)k) lo = mid + 1; else hi = mid;
  }
  s_start[k] = lo;
  if (k == 0) s_start[256] = n;
  __syncthreads();
  const int64_t cnt = k >= 2 ? s_start[k + 1] - s_start[k] : 0;
  s_rows[k] = cnt * (k - 1);
  __syncthreads();
  int64_t base = 0;
  for (int q = 2; q < k; ++q) base += s_rows[q];
  table[k] = (int32_t)s_start[k];
  table[256 + k] = (int32_t)cnt;
  table[512 + k] = (int32_t)base;
  if (k == 255) {
    ret[1] = n - s_start[2];
    ret[2] = base + s_rows[255];
    ret[3] = cnt > 0 ? 1 : 0;  // a count of 255 may be a saturated one: the caller must not trust the plan
  }
}

struct OvRowsArgs {
  const float* xyz;
  const float* lidar2img;
  const void* mask;
  const int32_t* max_id;
  const int64_t* batch_idx;  // or NULL: batch 0
  const uint32_t* fg_idx;
  const uint64_t* keys;
  const uint32_t* vals;
  const int32_t* table;
  int64_t* src_pt;
  int64_t* sir_coors;
  int64_t n, F, M;
  int stride, ncam, ncls, H, W;
};

__global__ void __launch_bounds__(256) ov_own_rows_kernel(OvRowsArgs a) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.F; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = a.fg_idx[r];
    a.src_pt[r] = i;
    a.sir_coors[r * 3 + 0] = a.batch_idx ? a.batch_idx[i] : 0;
    a.sir_coors[r * 3 + 1] = 0;
    a.sir_coors[r * 3 + 2] = a.max_id[i];
  }
}

template <typename MaskT>
__global__ void __launch_bounds__(256) ov_extra_rows_kernel(OvRowsArgs a) {
  extern __shared__ float s_mat[];
  for (int t = threadIdx.x; t < a.ncam * 12; t += blockDim.x) s_mat[t] = a.lidar2img[(t / 12) * 16 + t % 12];
  __syncthreads();
  const MaskT* mask = reinterpret_cast<const MaskT*>(a.mask);
  const float fw = (float)a.W, fh = (float)a.H;
  const int64_t plane = (int64_t)a.H * a.W;
  const int64_t first = a.n - a.M;  // the sorted list holds the points with k < 2 (key 0) in front
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.M; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = first + t;
    const int64_t i = a.vals[s];
    const int k = (int)a.keys[s];
    const int64_t per_j = a.table[256 + k];
    const int64_t row0 = a.F + a.table[512 + k] + (s - a.table[k]);
    const int64_t b = a.batch_idx ? a.batch_idx[i] : 0;
    const float* p = a.xyz + i * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    int64_t prev = INT64_MAX;
    int emitted = 0;
    while (emitted < k) {
      // the largest id below `prev` and how many cells hold it (topk keeps duplicates as separate entries)
      int64_t cur = 0;
      int same = 0;
      for (int cam = 0; cam < a.ncam; ++cam) {
        int64_t pix;
        if (!proj_pixel(s_mat + cam * 12, x, y, z, fw, fh, a.W, &pix)) continue;
        const MaskT* mc = mask + (int64_t)cam * a.ncls * plane + pix;
        for (int c = 0; c < a.ncls; ++c) {
          const int64_t v = (int64_t)mc[(int64_t)c * plane];
          if (v <= 0 || v >= prev) continue;
          if (v > cur) {
            cur = v;
            same = 1;
          } else if (v == cur) {
            ++same;
          }
        }
      }
      if (same == 0) break;  // (cannot happen when `k` is this point's cell count)
      for (int e = 0; e < same; ++e) {
        const int j = emitted + e;
        if (j >= 1 && j < k) {
          const int64_t r = row0 + (int64_t)(j - 1) * per_j;
          a.src_pt[r] = i;
          a.sir_coors[r * 3 + 0] = b;
          a.sir_coors[r * 3 + 1] = 0;
          a.sir_coors[r * 3 + 2] = cur;
        }
      }
      emitted += same;
      prev = cur;
    }
  }
}

struct OvLayout {
  uint32_t* fg_idx;
  uint64_t *keys_a, *keys_b;
  uint32_t *vals_a, *vals_b;
  uint32_t* hist;
  uint32_t* tile_sums;
  int64_t* ret;
  int32_t* table;
  size_t zero_bytes;
  bool ok;
};
static OvLayout ov_layout(void* ws, int64_t ws_bytes, int64_t n) {
  FsfArena ar(ws, ws_bytes);
  OvLayout l;
  l.fg_idx = ar.take<uint32_t>(n);
  l.keys_a = ar.take<uint64_t>(n);
  l.keys_b = ar.take<uint64_t>(n);
  l.vals_a = ar.take<uint32_t>(n);
  l.vals_b = ar.take<uint32_t>(n);
  l.hist = ar.take<uint32_t>((radix_num_tiles(n) + 1) * RS_BINS);  // [hist | tile_sums | ret] are consecutive: one memset
  l.tile_sums = ar.take<uint32_t>(scan_num_tiles(n));
  l.ret = ar.take<int64_t>(4);
  l.table = ar.take<int32_t>(3 * 256);
  l.zero_bytes = (size_t)((char*)l.table - (char*)l.hist);
  l.ok = ar.ok();
  return l;
}
}  // namespace fsf

extern "C" int64_t fsf_overlap_plan_workspace_bytes(int64_t n) {
  const int64_t m = n > 0 ? n : 1;
  return fsf_align_up(m * 4, 256) * 3 + fsf_align_up(m * 8, 256) * 2 + fsf_align_up((radix_num_tiles(n) + 1) * RS_BINS * 4, 256) +
         fsf_align_up(scan_num_tiles(n) * 4, 256) + 256 + fsf_align_up(3 * 256 * 4, 256);
}

extern "C" int fsf_overlap_plan(const uint8_t* fg, const uint8_t* count, int64_t n, int32_t max_cells, int64_t* counts_host, void* workspace,
                                int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || !counts_host || (n > 0 && (!fg || !count))) return FSF_ERR_INVALID_ARG;
  if (max_cells < 1 || max_cells > 254 || n >= ((int64_t)1 << 31)) return FSF_ERR_UNSUPPORTED;  // (a u8 count must not saturate)
  counts_host[0] = counts_host[1] = counts_host[2] = 0;
  if (n == 0) return FSF_OK;
  if (!workspace || workspace_bytes < fsf_overlap_plan_workspace_bytes(n)) return FSF_ERR_WORKSPACE;
  const OvLayout l = ov_layout(workspace, workspace_bytes, n);
  if (!l.ok) return FSF_ERR_WORKSPACE;
  FSF_HIP_TRY(hipMemsetAsync(l.hist, 0, l.zero_bytes, stream));
  int rc = exclusive_scan_u32(OvIn{fg, count, l.keys_a, l.vals_a}, OvOut{l.fg_idx}, n, l.tile_sums, nullptr, l.ret, stream, 1, true);
  if (rc != FSF_OK) return rc;
  uint64_t* keys_s;
  uint32_t* vals_s;
  rc = radix_sort_pairs(l.keys_a, l.vals_a, l.keys_b, l.vals_b, l.hist, n, 8, &keys_s, &vals_s, stream, true);
  if (rc != FSF_OK) return rc;
  if (keys_s != l.keys_b || vals_s != l.vals_b) return FSF_ERR_UNSUPPORTED;  // (one pass: fsf_overlap_rows reads the alternate buffers)
  hipLaunchKernelGGL(ov_plan_kernel, dim3(1), dim3(256), 0, stream, keys_s, n, l.table, l.ret);
  FSF_LAUNCH_CHECK();
  int64_t ret_h[4] = {0, 0, 0, 0};
  FSF_READ_BACK(ret_h, l.ret, sizeof(ret_h), stream);
  if (ret_h[3]) return FSF_ERR_UNSUPPORTED;
  counts_host[0] = ret_h[0];
  counts_host[1] = ret_h[1];
  counts_host[2] = ret_h[2];
  return FSF_OK;
}

extern "C" int fsf_overlap_rows(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam, const void* mask,
                                int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w, const int32_t* max_id,
                                const int64_t* batch_idx, const void* workspace, int64_t workspace_bytes, int64_t num_fg, int64_t num_multi,
                                int64_t num_extra, int64_t* src_pt, int64_t* sir_coors, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || xyz_stride < 3 || ncam < 1 || ncam > 64 || ncls < 1 || img_h < 1 || img_w < 1 || !lidar2img || !mask ||
      (elem_bytes != 1 && elem_bytes != 4) || num_fg < 0 || num_multi < 0 || num_extra < 0 || num_fg > n || num_multi > num_fg ||
      (num_fg > 0 && (!xyz || !max_id || !src_pt || !sir_coors)))
    return FSF_ERR_INVALID_ARG;
  if (num_fg == 0) return FSF_OK;
  if (!workspace || workspace_bytes < fsf_overlap_plan_workspace_bytes(n)) return FSF_ERR_WORKSPACE;
  const OvLayout l = ov_layout(const_cast<void*>(workspace), workspace_bytes, n);
  if (!l.ok) return FSF_ERR_WORKSPACE;
  (void)num_extra;
  OvRowsArgs a{xyz, lidar2img, mask, max_id, batch_idx, l.fg_idx, l.keys_b, l.vals_b, l.table, src_pt, sir_coors, n, num_fg, num_multi,
               (int)xyz_stride, (int)ncam, (int)ncls, (int)img_h, (int)img_w};
  hipLaunchKernelGGL(ov_own_rows_kernel, dim3(fsf_stream_grid(num_fg, 256)), dim3(256), 0, stream, a);
  if (num_multi > 0) {
    const size_t shmem = (size_t)ncam * 12 * sizeof(float);
    if (elem_bytes == 1)
      hipLaunchKernelGGL((ov_extra_rows_kernel<uint8_t>), dim3(fsf_stream_grid(num_multi, 256)), dim3(256), shmem, stream, a);
    else
      hipLaunchKernelGGL((ov_extra_rows_kernel<int32_t>), dim3(fsf_stream_grid(num_multi, 256)), dim3(256), shmem, stream, a);
  }
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
