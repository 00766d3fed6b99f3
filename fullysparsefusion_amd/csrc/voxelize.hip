// K1+K2: dynamic voxelization, both float->index formulas of the reference path.  See include/fsf_hip.h.
// HBM-bound streaming kernels: 12 B/pt read (x,y,z of a point_stride-float row) + 12 B (i32 zyx) and/or
// 32 B (i64 b,z,y,x) written.  All arithmetic is explicit round-to-nearest fp32 (no contraction, no
// reciprocal multiply) so the integer results are bit-exact with the CPU formulas.
#include "common.h"

namespace fsf {

struct VoxParams {
  float vx, vy, vz;
  float xmin, ymin, zmin;
  int gx, gy, gz;
};

__global__ void __launch_bounds__(256)
    voxelize_dynamic_kernel(const float* __restrict__ points, int64_t n, int stride, int batch_idx, VoxParams p,
                            int32_t* __restrict__ coors_zyx, int64_t* __restrict__ coors_bzyx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* pt = points + i * stride;
    const float x = pt[0], y = pt[1], z = pt[2];
    // upstream early-out order x -> y -> z; untouched slots stay at the zero the reference allocates
    int c0 = 0, c1 = 0, c2 = 0;  // (z, y, x) slots
    const int cx = (int)floorf(__fdiv_rn(__fsub_rn(x, p.xmin), p.vx));
    if (cx < 0 || cx >= p.gx) {
      c0 = -1;
    } else {
      const int cy = (int)floorf(__fdiv_rn(__fsub_rn(y, p.ymin), p.vy));
      if (cy < 0 || cy >= p.gy) {
        c0 = -1;
        c1 = -1;
      } else {
        const int cz = (int)floorf(__fdiv_rn(__fsub_rn(z, p.zmin), p.vz));
        if (cz < 0 || cz >= p.gz) {
          c0 = -1;
          c1 = -1;
          c2 = -1;
        } else {
          c0 = cz;
          c1 = cy;
          c2 = cx;
        }
      }
    }
    if (coors_zyx) {
      coors_zyx[i * 3 + 0] = c0;
      coors_zyx[i * 3 + 1] = c1;
      coors_zyx[i * 3 + 2] = c2;
    }
    if (coors_bzyx) {
      longlong4 v;
      v.x = batch_idx;
      v.y = c0;
      v.z = c1;
      v.w = c2;
      *reinterpret_cast<longlong4*>(coors_bzyx + i * 4) = v;
    }
  }
}

// c10::div_floor_floating<float> (torch.div(..., rounding_mode='floor')), restated for fp32.
__device__ __forceinline__ float div_floor_f32(float a, float b) {
  if (b == 0.0f) return __fdiv_rn(a, b);
  const float mod = fmodf(a, b);
  float div = __fdiv_rn(__fsub_rn(a, mod), b);
  if ((mod != 0.0f) && ((b < 0.0f) != (mod < 0.0f))) div = __fsub_rn(div, 1.0f);
  float fd;
  if (div != 0.0f) {
    fd = floorf(div);
    if (__fsub_rn(div, fd) > 0.5f) fd = __fadd_rn(fd, 1.0f);
  } else {
    fd = copysignf(0.0f, __fdiv_rn(a, b));
  }
  return fd;
}

__global__ void __launch_bounds__(256)
    voxelize_divfloor_kernel(const float* __restrict__ points, int64_t n, int stride, VoxParams p, int order,
                             const int64_t* __restrict__ batch_idx_in, int64_t* __restrict__ coors) {
  const int kc = batch_idx_in ? 4 : 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* pt = points + i * stride;
    const int64_t cx = (int64_t)div_floor_f32(__fsub_rn(pt[0], p.xmin), p.vx);
    const int64_t cy = (int64_t)div_floor_f32(__fsub_rn(pt[1], p.ymin), p.vy);
    const int64_t cz = (int64_t)div_floor_f32(__fsub_rn(pt[2], p.zmin), p.vz);
    int64_t* o = coors + i * kc;
    int j = 0;
    if (batch_idx_in) o[j++] = batch_idx_in[i];
    if (order == 0) {
      o[j] = cx; o[j + 1] = cy; o[j + 2] = cz;
    } else {
      o[j] = cz; o[j + 1] = cy; o[j + 2] = cx;
    }
  }
}

// ---- vote centres + cluster-voxel keys of the grouped cluster assignment (one pass instead of ~22 torch launches) --------
// Replaces, for every (class group g, point p) pair of the group-sampled foreground (single_stage_fsd.py:802-865 `group_sample`
// + ClusterAssigner.forward :903-982, all groups at once):
//   w = softmax-free "is an arg-max class of the group" weights (ties within 1e-6 split evenly),
//   centre = xyz[p] + sum_c offsets[p, c, :] * w_c,   vox = floor_div(centre - range_min, cluster_voxel_size[g]),
//   key = (g * batch_size + batch[p], vox_x, vox_y, vox_z).
// Same fp32 operations as the torch expressions (sub, abs, compare, div by the tie count, mul, adds in class order,
// c10::div_floor_floating); with one or two tied classes — every row in practice — the class sum has a single possible value.
constexpr int VC_MAX_GROUPS = 16;
struct VoteArgs {
  const float* logits; int logit_stride;
  const float* offsets; int offset_stride;
  const float* points; int point_stride;
  const int64_t* batch_idx;
  const int64_t* g_ids; const int64_t* p_ids;
  int64_t n;
  int nc, ng, bsz;
  uint32_t mask[VC_MAX_GROUPS];
  float vs[VC_MAX_GROUPS][3];
  float rmin[3];
  float* centers; int64_t* keys; int64_t* batch_out;
};

// (round 6) The rows of a pair — nc logits, 3 nc vote offsets — are read with 16-byte loads (4-byte aligned: rows of 11 / 33 floats)
// into registers and walked with compile-time indices; as runtime-bounded scalar loops every one of the ~75 loads of a pair was a wave
// instruction of its own touching ~50 cache lines (144 us for 510 k pairs, 0.4 TB/s).  The arithmetic and its order are unchanged.
struct __attribute__((packed, aligned(4))) vc_quad { float v[4]; };

__global__ void __launch_bounds__(256) vote_centers_keys_kernel(VoteArgs a) {
  const int nc = a.nc;  // <= 32 (the class masks are 32 bits wide)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)a.g_ids[i];
    const int64_t p = a.p_ids[i];
    const uint32_t mem = a.mask[g];
    const float* lgp = a.logits + p * a.logit_stride;
    float lg[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (4 * q + 4 <= nc) {  // (uniform) a whole quad of the row
        const vc_quad v = *reinterpret_cast<const vc_quad*>(lgp + 4 * q);
        lg[4 * q] = v.v[0]; lg[4 * q + 1] = v.v[1]; lg[4 * q + 2] = v.v[2]; lg[4 * q + 3] = v.v[3];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[4 * q + r] = 4 * q + r < nc ? lgp[4 * q + r] : 0.0f;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < nc && ((mem >> c) & 1u)) mx = fmaxf(mx, lg[c]);
    uint32_t tie = 0;
    float cnt = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < nc && ((mem >> c) & 1u) && fabsf(__fsub_rn(lg[c], mx)) < 1e-6f) {
        tie |= 1u << c;
        cnt = __fadd_rn(cnt, 1.0f);
      }
    const float wv = __fdiv_rn(1.0f, cnt);
    const float* off = a.offsets + p * a.offset_stride;
    float sacc[3] = {0.0f, 0.0f, 0.0f};
    const int ne = 3 * nc;
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      if (4 * q >= ne) break;  // (uniform)
      float o4[4];
      if (4 * q + 4 <= ne) {
        const vc_quad v = *reinterpret_cast<const vc_quad*>(off + 4 * q);
        o4[0] = v.v[0]; o4[1] = v.v[1]; o4[2] = v.v[2]; o4[3] = v.v[3];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = 4 * q + r < ne ? off[4 * q + r] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r, c = e / 3, jx = e - 3 * c;  // (compile-time)
        if (e < ne) {
          const float w = ((tie >> c) & 1u) ? wv : 0.0f;
          sacc[jx] = __fadd_rn(sacc[jx], __fmul_rn(o4[r], w));
        }
      }
    }
    const float sx = sacc[0], sy = sacc[1], sz = sacc[2];
    const float* pt = a.points + p * a.point_stride;
    const float cx = __fadd_rn(pt[0], sx), cy = __fadd_rn(pt[1], sy), cz = __fadd_rn(pt[2], sz);
    a.centers[3 * i + 0] = cx;
    a.centers[3 * i + 1] = cy;
    a.centers[3 * i + 2] = cz;
    const int64_t b = a.batch_idx ? a.batch_idx[p] : 0;
    a.keys[4 * i + 0] = (int64_t)g * a.bsz + b;
    a.keys[4 * i + 1] = (int64_t)div_floor_f32(__fsub_rn(cx, a.rmin[0]), a.vs[g][0]);
    a.keys[4 * i + 2] = (int64_t)div_floor_f32(__fsub_rn(cy, a.rmin[1]), a.vs[g][1]);
    a.keys[4 * i + 3] = (int64_t)div_floor_f32(__fsub_rn(cz, a.rmin[2]), a.vs[g][2]);
    if (a.batch_out) a.batch_out[i] = b;
  }
}

// ---- DynamicScatterVFE input decoration in one pass (the reference builds it from ~20 elementwise launches) ------------------
// out[i] = [ features[i, :P] | features[i, :3] - voxel_mean[inv[i], :3] (with_cluster_center) | features[i, :3] - centre of the
// point's voxel = coor * voxel_size + offset, coordinate order x <- coors[:, 3], y <- coors[:, 2], z <- coors[:, 1]
// (with_voxel_center) ], the fp32 operations of the torch expressions in their order (int -> float, mul, add, sub).
struct VfeDecoArgs {
  const float* feat; int feat_stride; int p;
  const float* vmean; int vmean_stride; const int64_t* inv;
  const int64_t* coors;
  float vx, vy, vz, ox, oy, oz;
  int with_cluster, with_center;
  float* out; int out_stride;
  int64_t n;
};

__global__ void __launch_bounds__(256) vfe_decorate_kernel(VfeDecoArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* f = a.feat + i * a.feat_stride;
    float* o = a.out + i * a.out_stride;
    for (int c = 0; c < a.p; ++c) o[c] = f[c];
    int col = a.p;
    const float x = f[0], y = f[1], z = f[2];
    if (a.with_cluster) {
      const float* m = a.vmean + a.inv[i] * a.vmean_stride;
      o[col + 0] = __fsub_rn(x, m[0]);
      o[col + 1] = __fsub_rn(y, m[1]);
      o[col + 2] = __fsub_rn(z, m[2]);
      col += 3;
    }
    if (a.with_center) {
      const int64_t* c = a.coors + 4 * i;
      o[col + 0] = __fsub_rn(x, __fadd_rn(__fmul_rn((float)c[3], a.vx), a.ox));
      o[col + 1] = __fsub_rn(y, __fadd_rn(__fmul_rn((float)c[2], a.vy), a.oy));
      o[col + 2] = __fsub_rn(z, __fadd_rn(__fmul_rn((float)c[1], a.vz), a.oz));
      col += 3;
    }
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_voxelize_dynamic(const float* points, int64_t n, int32_t point_stride, int32_t batch_idx,
                                    const float voxel_size[3], const float pc_range[6], const int32_t grid[3],
                                    int32_t* coors_zyx, int64_t* coors_bzyx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || point_stride < 3 || !voxel_size || !pc_range || !grid || (!coors_zyx && !coors_bzyx) ||
      (n > 0 && !points))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  VoxParams p{voxel_size[0], voxel_size[1], voxel_size[2], pc_range[0], pc_range[1], pc_range[2],
              grid[0],       grid[1],       grid[2]};
  hipLaunchKernelGGL(voxelize_dynamic_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, points, n,
                     (int)point_stride, (int)batch_idx, p, coors_zyx, coors_bzyx);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_voxelize_divfloor(const float* points, int64_t n, int32_t point_stride, const float voxel_size[3],
                                     const float range_min[3], int32_t order, const int64_t* batch_idx_in,
                                     int64_t* coors, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || point_stride < 3 || !voxel_size || !range_min || (order != 0 && order != 1) ||
      (n > 0 && (!points || !coors)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  VoxParams p{voxel_size[0], voxel_size[1], voxel_size[2], range_min[0], range_min[1], range_min[2], 0, 0, 0};
  hipLaunchKernelGGL(voxelize_divfloor_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, points, n,
                     (int)point_stride, p, (int)order, batch_idx_in, coors);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_vote_centers_keys(const float* logits, int32_t logit_stride, const float* offsets, int32_t offset_stride,
                                     const float* points, int32_t point_stride, const int64_t* batch_idx, const int64_t* g_ids,
                                     const int64_t* p_ids, int64_t n, int32_t num_classes, int32_t num_groups,
                                     const uint32_t* group_class_mask, const float* group_voxel_size, const float range_min[3],
                                     int32_t batch_size, float* centers, int64_t* keys, int64_t* batch_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_classes < 1 || num_classes > 32 || num_groups < 1 || num_groups > VC_MAX_GROUPS || batch_size < 1 ||
      logit_stride < num_classes || offset_stride < 3 * num_classes || point_stride < 3 || !group_class_mask || !group_voxel_size ||
      !range_min || (n > 0 && (!logits || !offsets || !points || !g_ids || !p_ids || !centers || !keys)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  VoteArgs a;
  a.logits = logits; a.logit_stride = logit_stride; a.offsets = offsets; a.offset_stride = offset_stride;
  a.points = points; a.point_stride = point_stride; a.batch_idx = batch_idx; a.g_ids = g_ids; a.p_ids = p_ids; a.n = n;
  a.nc = num_classes; a.ng = num_groups; a.bsz = batch_size;
  for (int g = 0; g < VC_MAX_GROUPS; ++g) {
    a.mask[g] = g < num_groups ? group_class_mask[g] : 0u;
    for (int j = 0; j < 3; ++j) a.vs[g][j] = g < num_groups ? group_voxel_size[3 * g + j] : 1.0f;
  }
  for (int j = 0; j < 3; ++j) a.rmin[j] = range_min[j];
  a.centers = centers; a.keys = keys; a.batch_out = batch_out;
  hipLaunchKernelGGL(vote_centers_keys_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_vfe_decorate(const float* features, int64_t n, int32_t feat_stride, int32_t p, const float* voxel_mean,
                                int32_t vmean_stride, const int64_t* inv, const int64_t* coors_bzyx, const float voxel_size[3],
                                const float offset[3], int32_t with_cluster_center, int32_t with_voxel_center, float* out,
                                int32_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int width = p + (with_cluster_center ? 3 : 0) + (with_voxel_center ? 3 : 0);
  if (n < 0 || p < 3 || feat_stride < p || out_stride < width || (with_cluster_center && (!voxel_mean || !inv || vmean_stride < 3)) ||
      (with_voxel_center && (!coors_bzyx || !voxel_size || !offset)) || (n > 0 && (!features || !out)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  VfeDecoArgs a{features, (int)feat_stride, (int)p, voxel_mean, (int)vmean_stride, inv, coors_bzyx,
                with_voxel_center ? voxel_size[0] : 0.f, with_voxel_center ? voxel_size[1] : 0.f, with_voxel_center ? voxel_size[2] : 0.f,
                with_voxel_center ? offset[0] : 0.f, with_voxel_center ? offset[1] : 0.f, with_voxel_center ? offset[2] : 0.f,
                (int)with_cluster_center, (int)with_voxel_center, out, (int)out_stride, n};
  hipLaunchKernelGGL(vfe_decorate_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
