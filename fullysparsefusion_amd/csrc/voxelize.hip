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
