// K29: the row-shuffling glue between the query stages, one launch per cluster of what were 3-18 ATen launches each (round 6).
// Every kernel is a pure gather / scatter / elementwise pass whose arithmetic is the ATen expression's, operation for operation
// (single IEEE adds / subtracts / multiplies; expf / atan2f exactly as K24's decoder, which is pinned bit-for-bit to the ATen chain),
// so the results are bit-identical to the chains they replace (tests/test_query_glue_gpu.py).
#include "common.h"

namespace fsf {

// ---- K29a: the inputs of a SIR stack on rows sorted by group (SIR._forward_sorted, models/backbones/sir.py:65-85) ---------------------
struct SortedRowsArgs {
  const int32_t* order;   // [n] sorted position -> source row
  const int64_t* inv;     // [n] source row -> group
  int64_t n;
  const float* points;    // [n, pts_cols] (row stride pts_stride)
  int64_t pts_stride;
  int pts_cols;
  const float* f_cluster; // [n, 3] (row stride fcl_stride) or NULL: then points[:, :3] - centers[inv]
  int64_t fcl_stride;
  const float* centers;   // [m, 3] (row stride centers_stride)
  int64_t centers_stride;
  const int64_t* index;   // [n] source row -> feature row, or NULL (identity)
  int64_t* seg_ids;       // [n]
  float* pts_sorted;      // [n, pts_cols]
  float* fcl_sorted;      // [n, 3]
  int64_t* idx_sorted;    // [n]
  float* fill;            // [fill_count] <- fill_value (the group table's -inf), or NULL
  int64_t fill_count;
  float fill_value;
};

__global__ void __launch_bounds__(256) sorted_rows_kernel(SortedRowsArgs a) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < a.n; i += step) {
    const int64_t src = a.order[i];
    const int64_t g = a.inv[src];
    a.seg_ids[i] = g;
    a.idx_sorted[i] = a.index ? a.index[src] : src;
    const float* p = a.points + src * a.pts_stride;
    float* q = a.pts_sorted + i * a.pts_cols;
    for (int c = 0; c < a.pts_cols; ++c) q[c] = p[c];
    float* f = a.fcl_sorted + i * 3;
    if (a.f_cluster) {
      const float* s = a.f_cluster + src * a.fcl_stride;
      f[0] = s[0]; f[1] = s[1]; f[2] = s[2];
    } else {
      const float* c = a.centers + g * a.centers_stride;
      f[0] = __fsub_rn(p[0], c[0]); f[1] = __fsub_rn(p[1], c[1]); f[2] = __fsub_rn(p[2], c[2]);
    }
  }
  for (int64_t i = tid; i < a.fill_count; i += step) a.fill[i] = a.fill_value;
}

// ---- K29b: the survivors of ClusterAssigner's density filter, every per-pair / per-key tensor in one pass --------------------------
__global__ void __launch_bounds__(256)
    compact_pairs_kernel(const float* __restrict__ means, int64_t means_stride, const int64_t* __restrict__ k_idx, int64_t nk,
                         float* __restrict__ vox_centers, const int64_t* __restrict__ g_ids, const int64_t* __restrict__ p_ids,
                         const int64_t* __restrict__ b_pts, const float* __restrict__ centers, const int64_t* __restrict__ v_idx, int64_t nv,
                         int64_t* __restrict__ g_out, int64_t* __restrict__ p_out, int64_t* __restrict__ b_out,
                         float* __restrict__ centers_out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < nv; i += step) {
    const int64_t s = v_idx[i];
    g_out[i] = g_ids[s];
    p_out[i] = p_ids[s];
    b_out[i] = b_pts[s];
    centers_out[i * 3 + 0] = centers[s * 3 + 0];
    centers_out[i * 3 + 1] = centers[s * 3 + 1];
    centers_out[i * 3 + 2] = centers[s * 3 + 2];
  }
  for (int64_t i = tid; i < nk; i += step) {
    const float* m = means + k_idx[i] * means_stride;
    vox_centers[i * 3 + 0] = m[0];
    vox_centers[i * 3 + 1] = m[1];
    vox_centers[i * 3 + 2] = m[2];
  }
}

// ---- K29c: FSF.combine_frustum_and_fsd's index / centre / 2-D prediction rows (FSF.py:657-692) -------------------------------------
__global__ void __launch_bounds__(256)
    combine_queries_kernel(const float* __restrict__ f_centers, int64_t mf, const float* __restrict__ l_centers, int64_t ml,
                           const int64_t* __restrict__ f_coors, const int64_t* __restrict__ l_coors, const float* __restrict__ f_preds,
                           int d, int64_t begin_idx, float* __restrict__ centers, int64_t* __restrict__ coors, float* __restrict__ preds) {
  const int64_t m = mf + ml;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < mf) {
      for (int c = 0; c < 3; ++c) centers[i * 3 + c] = f_centers[i * 3 + c], coors[i * 3 + c] = f_coors[i * 3 + c];
      for (int c = 0; c < d; ++c) preds[i * d + c] = f_preds[i * d + c];
    } else {
      const int64_t j = i - mf;
      for (int c = 0; c < 3; ++c) centers[i * 3 + c] = l_centers[j * 3 + c];
      coors[i * 3 + 0] = l_coors[j * 3 + 1];  // (class, batch, id) -> (batch, class, id + begin)
      coors[i * 3 + 1] = l_coors[j * 3 + 0];
      coors[i * 3 + 2] = l_coors[j * 3 + 2] + begin_idx;
      for (int c = 0; c < d; ++c) preds[i * d + c] = 0.0f;
    }
  }
}

// ---- K29d: FSF.decode_stage_bboxes (FSF.py:1085-1094): BasePointBBoxCoder.decode + the batch column ---------------------------------
__global__ void __launch_bounds__(256)
    decode_rois_kernel(const float* __restrict__ reg, int64_t reg_stride, int code, const float* __restrict__ centers, int64_t c_stride,
                       const int64_t* __restrict__ batch, int64_t b_stride, int64_t m, float eps, float* __restrict__ rois) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const float* r = reg + i * reg_stride;
    const float* p = centers + i * c_stride;
    float* o = rois + i * code;
    o[0] = (float)batch[i * b_stride];
    o[1] = __fadd_rn(r[0], p[0]); o[2] = __fadd_rn(r[1], p[1]); o[3] = __fadd_rn(r[2], p[2]);
    o[4] = __fsub_rn(expf(r[3]), eps); o[5] = __fsub_rn(expf(r[4]), eps); o[6] = __fsub_rn(expf(r[5]), eps);
    o[7] = atan2f(r[6], r[7]);
    for (int c = 8; c < code; ++c) o[c] = r[c];  // velocity rides along (code 10)
  }
}

// ---- K29e: the rows FSF.query_feat_refine / FullySparseBboxHead.forward build from the pooling result (FSF.py:961-1010,
// fsd_bbox_head.py:96-112): points[ext_pts_inds] and f_cluster = cat(local_xyz, boundary_offset, is_in_margin, xyz - roi centre) -----
__global__ void __launch_bounds__(256)
    refine_rows_kernel(const float* __restrict__ info, const float* __restrict__ points, int64_t p_stride, int p_cols,
                       const int64_t* __restrict__ pts_idx, const int64_t* __restrict__ roi_idx, const float* __restrict__ rois,
                       int64_t roi_stride, int64_t k, float* __restrict__ points_out, float* __restrict__ f_cluster) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = points + pts_idx[i] * p_stride;
    float* q = points_out + i * p_cols;
    for (int c = 0; c < p_cols; ++c) q[c] = p[c];
    const float* s = info + i * 13;
    float* f = f_cluster + i * 13;
    for (int c = 0; c < 10; ++c) f[c] = s[3 + c];
    const float* r = rois + roi_idx[i] * roi_stride;
    f[10] = __fsub_rn(p[0], r[0]); f[11] = __fsub_rn(p[1], r[1]); f[12] = __fsub_rn(p[2], r[2]);
  }
}

// ---- K29f: FSF.get_single_cls_preds_2d + encode_preds_2d (FSF.py:476-504, :449-474) for the camera queries ----------------------------
__global__ void __launch_bounds__(256)
    encode_preds_kernel(const float* __restrict__ anno, int64_t num_anno, int d, const int64_t* __restrict__ coors, int64_t m,
                        int num_classes, float inv_w, float inv_h, float* __restrict__ preds, float* __restrict__ enc, int64_t enc_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = coors[i * 3 + 2];
    const bool valid = id > 0;
    const int64_t row = valid ? id - 1 : 0;  // (obj_id - 1).clamp(min=0)
    const float* a = anno + (row < num_anno ? row : num_anno - 1) * d;
    float* p = preds + i * d;
    // mask_anno[b, safe] * valid: a product with 0.0 / 1.0 (keeps the sign of zero and NaN exactly as the ATen multiply does)
    const float v = valid ? 1.0f : 0.0f;
    for (int c = 0; c < d; ++c) p[c] = __fmul_rn(a[c], v);
    if (!valid) p[5] = (float)num_classes;
    float* e = enc + i * enc_stride;
    e[0] = __fmul_rn(p[0], inv_w); e[1] = __fmul_rn(p[1], inv_h); e[2] = __fmul_rn(p[2], inv_w); e[3] = __fmul_rn(p[3], inv_h);
    e[4] = p[4];
    const int64_t cat = (int64_t)p[5];  // category.long()
    for (int c = 0; c <= num_classes; ++c) e[5 + c] = c == cat ? 1.0f : 0.0f;
  }
}

// ---- K29g: FSF.get_cluster_delta_weighted's operands (FSF.py:313-329): cat(xyz * w, w) with w = clamp(weight, 1e-5); and the
// weighted centres mean[:, :3] / mean[:, 3:4] -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    weighted_xyz_kernel(const float* __restrict__ points, int64_t p_stride, const float* __restrict__ w, int64_t n, float wmin,
                        float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = points + i * p_stride;
    const float wi = w[i];
    const float c = wi != wi ? wi : (wi < wmin ? wmin : wi);  // clamp(min): NaN stays NaN
    float4 o;
    o.x = __fmul_rn(p[0], c); o.y = __fmul_rn(p[1], c); o.z = __fmul_rn(p[2], c); o.w = c;
    ((float4*)out)[i] = o;
  }
}

__global__ void __launch_bounds__(256) centroid_divide_kernel(const float* __restrict__ mean, int64_t m, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)mean)[i];
    out[i * 3 + 0] = __fdiv_rn(v.x, v.w); out[i * 3 + 1] = __fdiv_rn(v.y, v.w); out[i * 3 + 2] = __fdiv_rn(v.z, v.w);
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_sorted_rows(const int32_t* order, const int64_t* inv, int64_t n, const float* points, int64_t pts_stride,
                               int32_t pts_cols, const float* f_cluster, int64_t fcl_stride, const float* centers, int64_t centers_stride,
                               const int64_t* index, int64_t* seg_ids, float* pts_sorted, float* fcl_sorted, int64_t* idx_sorted,
                               float* fill, int64_t fill_count, float fill_value, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || fill_count < 0 || pts_cols < 3 || pts_cols > 64 || pts_stride < pts_cols || (fill_count > 0 && !fill)) return FSF_ERR_INVALID_ARG;
  if (n > 0 && (!order || !inv || !points || !seg_ids || !pts_sorted || !fcl_sorted || !idx_sorted || (!f_cluster && !centers)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0 && fill_count == 0) return FSF_OK;
  SortedRowsArgs a{order, inv, n, points, pts_stride, (int)pts_cols, f_cluster, fcl_stride, centers, centers_stride, index, seg_ids,
                   pts_sorted, fcl_sorted, idx_sorted, fill, fill_count, fill_value};
  const int64_t work = n > fill_count ? n : fill_count;
  hipLaunchKernelGGL(sorted_rows_kernel, dim3(fsf_stream_grid(work, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_compact_pairs(const float* means, int64_t means_stride, const int64_t* k_idx, int64_t nk, float* vox_centers,
                                 const int64_t* g_ids, const int64_t* p_ids, const int64_t* b_pts, const float* centers,
                                 const int64_t* v_idx, int64_t nv, int64_t* g_out, int64_t* p_out, int64_t* b_out, float* centers_out,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nk < 0 || nv < 0 || means_stride < 3) return FSF_ERR_INVALID_ARG;
  if (nk > 0 && (!means || !k_idx || !vox_centers)) return FSF_ERR_INVALID_ARG;
  if (nv > 0 && (!g_ids || !p_ids || !b_pts || !centers || !v_idx || !g_out || !p_out || !b_out || !centers_out)) return FSF_ERR_INVALID_ARG;
  if (nk == 0 && nv == 0) return FSF_OK;
  hipLaunchKernelGGL(compact_pairs_kernel, dim3(fsf_stream_grid(nk > nv ? nk : nv, 256)), dim3(256), 0, stream, means, means_stride, k_idx,
                     nk, vox_centers, g_ids, p_ids, b_pts, centers, v_idx, nv, g_out, p_out, b_out, centers_out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_combine_queries(const float* f_centers, int64_t mf, const float* l_centers, int64_t ml, const int64_t* f_coors,
                                   const int64_t* l_coors, const float* f_preds_2d, int32_t d, int64_t begin_idx, float* centers,
                                   int64_t* coors, float* preds_2d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (mf < 0 || ml < 0 || d < 0 || d > 64) return FSF_ERR_INVALID_ARG;
  if (mf > 0 && (!f_centers || !f_coors || (d > 0 && !f_preds_2d))) return FSF_ERR_INVALID_ARG;
  if (ml > 0 && (!l_centers || !l_coors)) return FSF_ERR_INVALID_ARG;
  if (mf + ml == 0) return FSF_OK;
  if (!centers || !coors || (d > 0 && !preds_2d)) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(combine_queries_kernel, dim3(fsf_stream_grid(mf + ml, 256)), dim3(256), 0, stream, f_centers, mf, l_centers, ml, f_coors,
                     l_coors, f_preds_2d, (int)d, begin_idx, centers, coors, preds_2d);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_decode_rois(const float* reg_preds, int64_t reg_stride, int32_t code_size, const float* centers, int64_t centers_stride,
                               const int64_t* batch, int64_t batch_stride, int64_t m, float eps, float* rois, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || (code_size != 8 && code_size != 10) || reg_stride < code_size || centers_stride < 3 || batch_stride < 1) return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  if (!reg_preds || !centers || !batch || !rois) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(decode_rois_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, reg_preds, reg_stride, (int)code_size, centers,
                     centers_stride, batch, batch_stride, m, eps, rois);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_refine_rows(const float* info, const float* points, int64_t points_stride, int32_t points_cols, const int64_t* pts_idx,
                               const int64_t* roi_idx, const float* roi_xyz, int64_t roi_stride, int64_t k, float* points_out,
                               float* f_cluster, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (k < 0 || points_cols < 3 || points_cols > 64 || points_stride < points_cols || roi_stride < 3) return FSF_ERR_INVALID_ARG;
  if (k == 0) return FSF_OK;
  if (!info || !points || !pts_idx || !roi_idx || !roi_xyz || !points_out || !f_cluster) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(refine_rows_kernel, dim3(fsf_stream_grid(k, 256)), dim3(256), 0, stream, info, points, points_stride, (int)points_cols,
                     pts_idx, roi_idx, roi_xyz, roi_stride, k, points_out, f_cluster);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_encode_preds_2d(const float* mask_anno, int64_t num_anno, int32_t d, const int64_t* obj_coors, int64_t m,
                                   int32_t num_classes, float img_w, float img_h, float* preds_2d, float* encoded, int64_t enc_stride,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || num_anno < 1 || d < 6 || d > 64 || num_classes < 1 || enc_stride < 6 + num_classes || !(img_w > 0.0f) || !(img_h > 0.0f))
    return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  if (!mask_anno || !obj_coors || !preds_2d || !encoded) return FSF_ERR_INVALID_ARG;
  // (ATen divides a tensor by a host scalar as a product with its fp32 reciprocal: BinaryDivTrueKernel's scalar path)
  hipLaunchKernelGGL(encode_preds_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, mask_anno, num_anno, (int)d, obj_coors, m,
                     (int)num_classes, 1.0f / img_w, 1.0f / img_h, preds_2d, encoded, enc_stride);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_weighted_xyz(const float* points, int64_t points_stride, const float* weights, int64_t n, float weight_min, float* out,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || points_stride < 3) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (!points || !weights || !out || ((uintptr_t)out & 15)) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(weighted_xyz_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, points, points_stride, weights, n, weight_min, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_centroid_divide(const float* mean, int64_t m, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0) return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  if (!mean || !out || ((uintptr_t)mean & 15)) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(centroid_divide_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, mean, m, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
