// K19: connected components of the "xy-distance < dist" graph over cluster-voxel centres, on the device.
// Replaces the GPU -> CPU -> scipy.sparse.csgraph.connected_components -> GPU round trip of
// find_connected_componets_single_batch / find_connected_componets
// (projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:45-82), six times per frame.
// Lock-free union-find (hook the larger root under the smaller with atomicMin), then labels = rank of each
// component's smallest member index — exactly the labelling scipy produces (components numbered in order
// of their first node), so the labels are bit-exact, not just the partition.
#include "common.h"
#include "scan.h"

namespace fsf {

__device__ __forceinline__ int uf_find(int* __restrict__ parent, int x) {
  int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    x = p;
    p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}

__device__ __forceinline__ void uf_union(int* __restrict__ parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      int t = a;
      a = b;
      b = t;
    }
    // a > b: hook root a under b
    const int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;  // somebody re-rooted a meanwhile; retry from there
  }
}

__global__ void __launch_bounds__(256) ccl_init_kernel(int* parent, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    parent[i] = (int)i;
}

// upper-triangular 256 x 256 tiles of the pair matrix
__global__ void __launch_bounds__(256)
    ccl_pairs_kernel(const float* __restrict__ pts, int stride, const int32_t* __restrict__ batch, int64_t n, float dist,
                     int* __restrict__ parent, int tiles, const float* __restrict__ dist_table,
                     const int2* __restrict__ tile_range) {
  __shared__ float sx[256], sy[256];
  __shared__ int sb[256];
  // linear block id -> (ti, tj) with tj >= ti
  int rem = blockIdx.x, ti = 0;
  while (rem >= tiles - ti) {
    rem -= tiles - ti;
    ++ti;
  }
  const int tj = ti + rem;
  if (tile_range) {  // grouped form: two tiles whose group ranges do not meet have no pair to test
    const int2 ri = tile_range[ti], rj = tile_range[tj];
    if (ri.y < rj.x || rj.y < ri.x) return;
  }
  const int64_t j0 = (int64_t)tj * 256;
  {
    const int64_t j = j0 + threadIdx.x;
    sx[threadIdx.x] = (j < n) ? pts[j * stride + 0] : 0.f;
    sy[threadIdx.x] = (j < n) ? pts[j * stride + 1] : 0.f;
    sb[threadIdx.x] = (j < n && batch) ? batch[j] : 0;
  }
  __syncthreads();
  const int64_t i = (int64_t)ti * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pts[i * stride + 0], y = pts[i * stride + 1];
  const int b = batch ? batch[i] : 0;
  if (dist_table) dist = dist_table[b];
  const int jn = (int)((n - j0 < 256) ? (n - j0) : 256);
  for (int jj = 0; jj < jn; ++jj) {
    const int64_t j = j0 + jj;
    if (j <= i) continue;
    if (sb[jj] != b) continue;
    const float dx = __fsub_rn(x, sx[jj]), dy = __fsub_rn(y, sy[jj]);
    const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    if (d < dist) uf_union(parent, (int)i, (int)j);
  }
}

// (min, max) group id of every 256-point tile
__global__ void __launch_bounds__(256) ccl_tile_range_kernel(const int32_t* __restrict__ group, int64_t n, int2* __restrict__ range) {
  __shared__ int smin[4], smax[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lo = i < n ? group[i] : 0x7fffffff, hi = i < n ? group[i] : (int)0x80000000;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    range[blockIdx.x] = make_int2(min(min(smin[0], smin[1]), min(smin[2], smin[3])), max(max(smax[0], smax[1]), max(smax[2], smax[3])));
}

__global__ void __launch_bounds__(256) ccl_flatten_kernel(int* parent, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    parent[i] = uf_find(parent, (int)i);
}

struct RootIn {
  const int* parent;
  __device__ uint32_t operator()(int64_t i) const { return parent[i] == (int)i ? 1u : 0u; }
};
struct RootOut {
  int* rank;
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t is_root) const {
    if (is_root) rank[i] = (int)excl;
  }
};

__global__ void __launch_bounds__(256)
    ccl_label_kernel(const int* __restrict__ parent, const int* __restrict__ rank, int64_t n, int32_t* __restrict__ labels) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    labels[i] = rank[parent[i]];
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_connected_components_workspace_bytes(int64_t n) {
  const int64_t nn = n > 0 ? n : 1;
  return fsf_align_up(nn * 4, 256) * 2 + fsf_align_up(scan_num_tiles(n) * 4, 256) + fsf_align_up((nn + 255) / 256 * 8, 256) + 256;
}

static int ccl_run(const float* points, int64_t n, int32_t point_stride, const int32_t* batch_idx, float dist,
                   const float* dist_table, int32_t* labels, int64_t* num_components_dev, void* workspace,
                   int64_t workspace_bytes, hipStream_t stream) {
  if (n < 0 || point_stride < 2 || (n > 0 && (!points || !labels))) return FSF_ERR_INVALID_ARG;
  if (n >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  if (n == 0) {
    if (num_components_dev) FSF_HIP_TRY(hipMemsetAsync(num_components_dev, 0, sizeof(int64_t), stream));
    return FSF_OK;
  }
  if (workspace_bytes < fsf_connected_components_workspace_bytes(n)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  int* parent = ar.take<int>(n);
  int* rank = ar.take<int>(n);
  uint32_t* tile_sums = ar.take<uint32_t>(scan_num_tiles(n));
  const int64_t tiles = (n + 255) / 256;
  int2* tile_range = dist_table ? ar.take<int2>(tiles) : nullptr;
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  const int grid = fsf_stream_grid(n, 256);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(grid), dim3(256), 0, stream, parent, n);
  const int64_t blocks = tiles * (tiles + 1) / 2;
  if (blocks >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  if (tile_range) hipLaunchKernelGGL(ccl_tile_range_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, batch_idx, n, tile_range);
  hipLaunchKernelGGL(ccl_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, points, (int)point_stride, batch_idx,
                     n, dist, parent, (int)tiles, dist_table, (const int2*)tile_range);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(grid), dim3(256), 0, stream, parent, n);
  RootIn rin{parent};
  RootOut rout{rank};
  int rc = exclusive_scan_u32(rin, rout, n, tile_sums, nullptr, num_components_dev, stream);
  if (rc != FSF_OK) return rc;
  hipLaunchKernelGGL(ccl_label_kernel, dim3(grid), dim3(256), 0, stream, parent, rank, n, labels);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_connected_components(const float* points, int64_t n, int32_t point_stride, const int32_t* batch_idx,
                                        float dist, int32_t* labels, int64_t* num_components_dev, void* workspace,
                                        int64_t workspace_bytes, void* stream_) {
  return ccl_run(points, n, point_stride, batch_idx, dist, nullptr, labels, num_components_dev, workspace, workspace_bytes,
                 (hipStream_t)stream_);
}

extern "C" int fsf_connected_components_grouped(const float* points, int64_t n, int32_t point_stride, const int32_t* group_idx,
                                                const float* dist_table, int32_t num_groups, int32_t* labels,
                                                int64_t* num_components_dev, void* workspace, int64_t workspace_bytes,
                                                void* stream_) {
  if (!group_idx || !dist_table || num_groups < 1) return FSF_ERR_INVALID_ARG;
  return ccl_run(points, n, point_stride, group_idx, 0.0f, dist_table, labels, num_components_dev, workspace, workspace_bytes,
                 (hipStream_t)stream_);
}
