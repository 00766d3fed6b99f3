// K24: the box tail of a cluster head around the multi-class NMS (K20) — see include/fsf_hip.h.
//   fsf_decode_cluster_boxes : sigmoid scores (class-major) + box decode + the NMS form of the BEV boxes, one pass over the queries
//   fsf_class_rank_desc      : per class, the stable descending score order of the boxes above the threshold (ONE packed-key radix
//                              sort over all classes) -> order / rank / count, the inputs of fsf_nms_bev_multiclass[_capped]
//   fsf_nms_select           : the kept boxes of all classes -> the best max_num rows (box | score | label), count and flags in
//                              one buffer the caller brings to the host with a single copy
// Replaces ~95 ATen launches of FrustumClusterHead._get_bboxes_single / box3d_multiclass_nms / bbox3d2result per frame.
#include "radix_sort.h"

namespace fsf {

// order-preserving map float -> u32 (ascending), then inverted: ascending key = DESCENDING score
__device__ __forceinline__ uint32_t bt_desc_key(float s) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

__global__ void __launch_bounds__(256)
    bt_decode_kernel(const float* __restrict__ cls, int64_t cls_stride, const float* __restrict__ reg, int64_t reg_stride,
                     const float* __restrict__ xyz, int64_t xyz_stride, int64_t n, int C, int code, float eps, float* __restrict__ boxes,
                     float* __restrict__ boxes_nms, float* __restrict__ scores_t) {
  const int D = code - 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* r = reg + i * reg_stride;
    const float* p = xyz + i * xyz_stride;
    const float x = __fadd_rn(r[0], p[0]), y = __fadd_rn(r[1], p[1]), z = __fadd_rn(r[2], p[2]);
    const float dx = __fsub_rn(expf(r[3]), eps), dy = __fsub_rn(expf(r[4]), eps), dz = __fsub_rn(expf(r[5]), eps);
    const float yaw = atan2f(r[6], r[7]);
    float* b = boxes + i * D;
    b[0] = x; b[1] = y; b[2] = z; b[3] = dx; b[4] = dy; b[5] = dz; b[6] = yaw;
    for (int d = 7; d < D; ++d) b[d] = r[d + 1];  // velocity rides along (code 10)
    const float hw = __fmul_rn(dx, 0.5f), hh = __fmul_rn(dy, 0.5f);
    float* q = boxes_nms + i * 5;
    q[0] = __fsub_rn(x, hw); q[1] = __fsub_rn(y, hh); q[2] = __fadd_rn(x, hw); q[3] = __fadd_rn(y, hh); q[4] = yaw;
    for (int c = 0; c < C; ++c)
      scores_t[(int64_t)c * n + i] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-cls[i * cls_stride + c])));
  }
}

__global__ void __launch_bounds__(256)
    bt_keys_kernel(const float* __restrict__ scores_t, int64_t total, int64_t n, float thr, uint64_t* __restrict__ keys,
                   uint32_t* __restrict__ vals) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = p / n;
    const float s = scores_t[p];
    keys[p] = ((uint64_t)c << 32) | (uint64_t)(s > thr ? bt_desc_key(s) : 0xFFFFFFFFu);  // (a valid score never maps to all ones)
    vals[p] = (uint32_t)(p - c * n);
  }
}

__global__ void __launch_bounds__(256)
    bt_rank_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t total, int64_t n,
                   int32_t* __restrict__ order, int32_t* __restrict__ rank, int32_t* __restrict__ count) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = keys[p];
    const int64_t c = (int64_t)(k >> 32);  // (= p / n: every class has exactly n keys)
    const int64_t pos = p - c * n;
    const uint32_t i = vals[p];
    const bool valid = (uint32_t)k != 0xFFFFFFFFu;
    order[p] = (int32_t)i;
    rank[c * n + i] = valid ? (int32_t)pos : -1;
    if (valid && (pos + 1 == n || (uint32_t)keys[p + 1] == 0xFFFFFFFFu)) count[c] = (int32_t)(pos + 1);  // the class's last valid box
  }
}

constexpr int BT_SEL_THREADS = 1024;
constexpr int BT_SEL_CAP = 16384;  // kept boxes over all classes the selection can take (dynamic LDS: 8 B of key per box, 128 KB at most)
constexpr int BT_MAX_CLASSES = 32;

// one workgroup: class-major list of the kept boxes -> (if more than max_num) the best max_num by descending score, ties in class-major order
__global__ void __launch_bounds__(BT_SEL_THREADS)
    bt_select_kernel(const float* __restrict__ boxes, const float* __restrict__ scores_t, const int32_t* __restrict__ order,
                     const int64_t* __restrict__ keep, int64_t keep_stride, const int64_t* __restrict__ num, int C, int64_t n, int D,
                     int max_num, int cap, const int64_t* __restrict__ lut, const int32_t* __restrict__ incomplete,
                     float* __restrict__ out, int32_t* __restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) char bt_smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(bt_smem);  // [next power of two >= num_classes * max_keep]
  __shared__ int offs[BT_MAX_CLASSES + 1];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int acc = 0;
    for (int c = 0; c < C; ++c) {
      offs[c] = acc;
      int64_t k = num[c];
      k = k < 0 ? 0 : (k > n ? n : k);
      acc += (int)(k > cap - acc ? cap - acc : k);  // (never clipped when num[c] <= max_keep, as K20 guarantees: cap >= C * max_keep)
    }
    offs[C] = acc;
  }
  __syncthreads();
  const int T = offs[C];
  auto locate = [&](int t, int& c, int64_t& i) {
    c = 0;
    while (c + 1 < C && offs[c + 1] <= t) ++c;
    i = order[(int64_t)c * n + keep[(int64_t)c * keep_stride + (t - offs[c])]];
  };
  // More than max_num kept: the best max_num by descending score, ties in class-major order.  Every class's kept boxes already come in
  // descending score order (K20 settles a class in that order), so the global rank of an element is a sum of lower bounds — one binary
  // search per class over keys that sit in LDS — instead of a sort: ~C log2(max_keep) LDS reads per element (the bitonic sort of the
  // 8 192-slot list this replaces took 100 us of the frame's serial tail in one workgroup).
  const bool select = T > max_num;
  if (select) {
    for (int t = tid; t < T; t += BT_SEL_THREADS) {
      int c;
      int64_t i;
      locate(t, c, i);
      keys[t] = ((uint64_t)bt_desc_key(scores_t[(int64_t)c * n + i]) << 32) | (uint32_t)t;  // ascending key = descending score, then position
    }
    __syncthreads();
  }
  const int K = T < max_num ? T : max_num;
  const int W = D + 2;
  for (int t = tid; t < T; t += BT_SEL_THREADS) {
    int o = t;
    if (select) {
      const uint64_t key = keys[t];
      o = 0;
      for (int c2 = 0; c2 < C && o < K; ++c2) {  // elements of class c2 with a smaller key (keys are distinct: the position is in them)
        int lo = offs[c2], hi = offs[c2 + 1];
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (keys[mid] < key) lo = mid + 1;
          else hi = mid;
        }
        o += lo - offs[c2];
      }
      if (o >= K) continue;
    }
    int c;
    int64_t i;
    locate(t, c, i);
    float* dst = out + (int64_t)o * W;
    for (int d = 0; d < D; ++d) dst[d] = boxes[i * D + d];
    dst[D] = scores_t[(int64_t)c * n + i];
    dst[D + 1] = (float)(lut ? lut[c] : (int64_t)c);
  }
  if (tid == 0) {
    meta[0] = K;
    meta[1] = T;
    meta[2] = incomplete ? *incomplete : 0;
    meta[3] = 0;
  }
}

}  // namespace fsf

using namespace fsf;

static int bt_class_bits(int32_t c) {
  int b = 0;
  while ((1 << b) < c) ++b;
  return b;
}

extern "C" int fsf_decode_cluster_boxes(const float* cls_logits, int64_t cls_stride, const float* reg_preds, int64_t reg_stride,
                                        const float* cluster_xyz, int64_t xyz_stride, int64_t n, int32_t num_classes, int32_t code_size,
                                        float eps, float* boxes, float* boxes_nms, float* scores_t, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_classes < 1 || (n > 0 && (!cls_logits || !reg_preds || !cluster_xyz || !boxes || !boxes_nms || !scores_t)))
    return FSF_ERR_INVALID_ARG;
  if (cls_stride < num_classes || reg_stride < code_size || xyz_stride < 3) return FSF_ERR_INVALID_ARG;
  if (code_size != 8 && code_size != 10) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  hipLaunchKernelGGL(bt_decode_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, cls_logits, cls_stride, reg_preds,
                     reg_stride, cluster_xyz, xyz_stride, n, (int)num_classes, (int)code_size, eps, boxes, boxes_nms, scores_t);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_class_rank_desc_workspace_bytes(int64_t n, int32_t num_classes) {
  if (n < 0 || num_classes < 1) return 0;
  const int64_t total = n * num_classes;
  return 2 * fsf_align_up(total * 8, 256) + 2 * fsf_align_up(total * 4, 256) +
         fsf_align_up((int64_t)RS_BINS * (radix_num_tiles(total) + 1) * 4, 256) + 1024;
}

extern "C" int fsf_class_rank_desc(const float* scores_t, int64_t n, int32_t num_classes, float score_thr, int32_t* order,
                                   int32_t* rank, int32_t* count, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_classes < 1 || !count || (n > 0 && (!scores_t || !order || !rank))) return FSF_ERR_INVALID_ARG;
  if (num_classes > BT_MAX_CLASSES || n * (int64_t)num_classes >= ((int64_t)1 << 31)) return FSF_ERR_UNSUPPORTED;
  FSF_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t) * num_classes, stream));
  if (n == 0) return FSF_OK;
  const int64_t total = n * num_classes;
  FsfArena arena(workspace, workspace_bytes);
  uint64_t* ka = arena.take<uint64_t>(total);
  uint64_t* kb = arena.take<uint64_t>(total);
  uint32_t* va = arena.take<uint32_t>(total);
  uint32_t* vb = arena.take<uint32_t>(total);
  uint32_t* hist = arena.take<uint32_t>((int64_t)RS_BINS * (radix_num_tiles(total) + 1));
  if (!ka || !kb || !va || !vb || !hist) return FSF_ERR_WORKSPACE;
  hipLaunchKernelGGL(bt_keys_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, scores_t, total, n, score_thr, ka, va);
  uint64_t* sk;
  uint32_t* sv;
  const int rc = radix_sort_pairs(ka, va, kb, vb, hist, total, 32 + bt_class_bits(num_classes), &sk, &sv, stream);
  if (rc != FSF_OK) return rc;
  hipLaunchKernelGGL(bt_rank_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, sk, sv, total, n, order, rank, count);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_nms_select_capacity(void) { return BT_SEL_CAP; }
extern "C" int32_t fsf_box_tail_max_classes(void) { return BT_MAX_CLASSES; }

extern "C" int fsf_nms_select(const float* boxes, int32_t box_dim, const float* scores_t, const int32_t* order, const int64_t* keep,
                              int64_t keep_stride, const int64_t* num_keep, int64_t n, int32_t num_classes, int64_t max_keep,
                              int32_t max_num, const int64_t* label_lut, const int32_t* incomplete, float* out, int32_t* meta,
                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_classes < 1 || box_dim < 1 || max_num < 1 || !out || !meta || !num_keep ||
      (n > 0 && (!boxes || !scores_t || !order || !keep)))
    return FSF_ERR_INVALID_ARG;
  // every class contributes at most max_keep boxes: the list must fit the selection's LDS
  if (num_classes > BT_MAX_CLASSES || max_keep < 1 || max_keep * (int64_t)num_classes > BT_SEL_CAP || n >= ((int64_t)1 << 31))
    return FSF_ERR_UNSUPPORTED;
  int64_t cap = 1;
  while (cap < max_keep * (int64_t)num_classes) cap <<= 1;
  static std::atomic<uint64_t> attr_done{0};
  FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)bt_select_kernel, BT_SEL_CAP * 8, attr_done));
  hipLaunchKernelGGL(bt_select_kernel, dim3(1), dim3(BT_SEL_THREADS), (size_t)cap * 8, stream, boxes, scores_t, order, keep, keep_stride, num_keep,
                     (int)num_classes, n, (int)box_dim, (int)max_num, (int)cap, label_lut, incomplete, out, meta);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
