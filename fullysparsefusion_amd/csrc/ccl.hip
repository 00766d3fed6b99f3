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

// Find with path halving.  Roots are hooked larger-under-smaller, so parent[x] <= x always and a pointer only ever
// moves towards the root; the shortcut parent[x] <- grandparent is written with atomicMin (fire and forget): it can
// never undo a concurrent hook, which also only lowers the value.  Without it the sorted input builds chains —
// parent[k] = k - 1 along a ground-plane component of 10^4 centres — and every find walks them hop by hop
// (231 k unions on the 10-sweep frame: 650 us; with halving: see DESIGN.md).
// The loads are ordinary cached loads (workgroup scope): a stale pointer still names a node of x's component with a
// smaller index, so the walk ends at a node that WAS a root; the hook's atomicMin (device scope) returns the current
// value and uf_union retries from there if that node has been hooked since.  Device-scope loads on every hop made a
// union ~1.3 us.
__device__ __forceinline__ int uf_find(int* __restrict__ parent, int x) {
  int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (p != x) {
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gp != p) atomicMin(&parent[x], gp);
    x = p;
    p = gp;
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

// Pair tests over the upper-triangular 256 x 256 tiles of the pair matrix.  Block (bj, ti) walks tiles
// tj = max(ti, 16 bj) .. 16 bj + 15 of row ti and skips the ones that cannot hold a link: group ranges that do not meet
// (grouped form), or xy boxes further apart than any linking distance both tiles use — the points arrive in voxel-key
// order, a 256-point tile is a thin strip, and 9 of 10 tile pairs of a 10-sweep frame fall to the box test.  (One block
// per tile pair spent its time on dispatch: 54 k blocks that mostly exit at once cost 0.4 ms.)
constexpr int CCL_TJ = 4;

// smallest float T with sqrt(T) >= dist (round-to-nearest sqrt is monotone), so that  s < T  <=>  sqrtf(s) < dist :
// the pair loop compares squared distances and still takes exactly the reference's decisions
__device__ __forceinline__ float ccl_sq_threshold(float dist) {
  if (!(dist > 0.0f)) return 0.0f;
  float t = __fmul_rn(dist, dist);
  while (t > 0.0f && __fsqrt_rn(__uint_as_float(__float_as_uint(t) - 1u)) >= dist) t = __uint_as_float(__float_as_uint(t) - 1u);
  while (__fsqrt_rn(t) < dist) t = __uint_as_float(__float_as_uint(t) + 1u);
  return t;
}

__global__ void __launch_bounds__(256)
    ccl_pairs_kernel(const float* __restrict__ pts, int stride, const int32_t* __restrict__ batch, int64_t n, float dist,
                     int* __restrict__ parent, int tiles, const float* __restrict__ dist_table,
                     const int2* __restrict__ tile_range, const float4* __restrict__ tile_box,
                     const float* __restrict__ tile_dmax) {
  __shared__ __attribute__((aligned(16))) float sx[256], sy[256];
  __shared__ __attribute__((aligned(16))) int sb[256];
  // Links found by the scan are queued and united afterwards, one per thread: a union is a chain of dependent memory
  // round trips, and called from inside the scan it ran with one or two live lanes per wave (a thread has ~3 links at
  // 3 different j) — 190 serial unions per wave and tile pair, 0.65 ms for the 231 k links of a 10-sweep frame.
  constexpr int QCAP = 2048;
  __shared__ int2 q[QCAP];
  __shared__ int qn;
  const int ti = blockIdx.y;
  const int tj_end = min(tiles, ((int)blockIdx.x + 1) * CCL_TJ);
  int tj = max(ti, (int)blockIdx.x * CCL_TJ);
  if (tj >= tj_end) return;
  const int64_t i = (int64_t)ti * 256 + threadIdx.x;
  const bool live = i < n;
  const float x = live ? pts[i * stride + 0] : 0.f, y = live ? pts[i * stride + 1] : 0.f;
  const int b = (live && batch) ? batch[i] : 0;
  if (dist_table) dist = live ? dist_table[b] : 0.f;
  const float thr = live ? ccl_sq_threshold(dist) : 0.0f;
  const float4 bi = tile_box[ti];  // (xmin, xmax, ymin, ymax)
  const float di = tile_dmax[ti];
  int2 ri = make_int2(0, 0);
  if (tile_range) ri = tile_range[ti];
  for (; tj < tj_end; ++tj) {  // (block-uniform loop and tests)
    if (tile_range) {
      const int2 rj = tile_range[tj];
      if (ri.y < rj.x || rj.y < ri.x) continue;
    }
    // d >= |dx| >= the box gap up to one rounding of the square root, hence the 1e-4 slack
    const float4 bj = tile_box[tj];
    const float gap = fmaxf(fmaxf(bj.x - bi.y, bi.x - bj.y), fmaxf(bj.z - bi.w, bi.z - bj.w));
    if (gap > fminf(di, tile_dmax[tj]) * 1.0001f) continue;
    const int64_t j0 = (int64_t)tj * 256;
    __syncthreads();  // the previous tile's LDS copy and queue are no longer read
    if (threadIdx.x == 0) qn = 0;
    {
      const int64_t j = j0 + threadIdx.x;  // rows past n: another group id, never linked
      sx[threadIdx.x] = (j < n) ? pts[j * stride + 0] : 0.f;
      sy[threadIdx.x] = (j < n) ? pts[j * stride + 1] : 0.f;
      sb[threadIdx.x] = (j < n) ? (batch ? batch[j] : 0) : -0x7fffffff;
    }
    __syncthreads();
    // same tile: only j > i; the four-wide loop starts at the first group of four that holds such a j
    const int jbeg = !live ? 256 : (tj == ti) ? ((int)threadIdx.x + 1) & ~3 : 0;
    for (int jj = jbeg; jj < 256; jj += 4) {
      const float4 xs = *reinterpret_cast<const float4*>(sx + jj), ys = *reinterpret_cast<const float4*>(sy + jj);
      const int4 bs = *reinterpret_cast<const int4*>(sb + jj);
      const float dx0 = __fsub_rn(x, xs.x), dy0 = __fsub_rn(y, ys.x), dx1 = __fsub_rn(x, xs.y), dy1 = __fsub_rn(y, ys.y);
      const float dx2 = __fsub_rn(x, xs.z), dy2 = __fsub_rn(y, ys.z), dx3 = __fsub_rn(x, xs.w), dy3 = __fsub_rn(y, ys.w);
      const float s0 = __fadd_rn(__fmul_rn(dx0, dx0), __fmul_rn(dy0, dy0)), s1 = __fadd_rn(__fmul_rn(dx1, dx1), __fmul_rn(dy1, dy1));
      const float s2 = __fadd_rn(__fmul_rn(dx2, dx2), __fmul_rn(dy2, dy2)), s3 = __fadd_rn(__fmul_rn(dx3, dx3), __fmul_rn(dy3, dy3));
      const int64_t jb = j0 + jj;
      const bool l0 = s0 < thr && bs.x == b && jb > i, l1 = s1 < thr && bs.y == b && jb + 1 > i;
      const bool l2 = s2 < thr && bs.z == b && jb + 2 > i, l3 = s3 < thr && bs.w == b && jb + 3 > i;
      if (l0 | l1 | l2 | l3) {
        const int cnt = (int)l0 + (int)l1 + (int)l2 + (int)l3;
        int pos = atomicAdd(&qn, cnt);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool l = u == 0 ? l0 : u == 1 ? l1 : u == 2 ? l2 : l3;
          if (l) {
            if (pos < QCAP) q[pos] = make_int2((int)i, (int)(jb + u));
            else uf_union(parent, (int)i, (int)(jb + u));  // (queue full: unite on the spot)
            ++pos;
          }
        }
      }
    }
    __syncthreads();
    const int nq = min(qn, QCAP);
    for (int t = threadIdx.x; t < nq; t += 256) uf_union(parent, q[t].x, q[t].y);
  }
}

// per 256-point tile: (min, max) group id, xy bounding box, largest linking distance of the groups it holds
__global__ void __launch_bounds__(256)
    ccl_tile_range_kernel(const float* __restrict__ pts, int stride, const int32_t* __restrict__ group, int64_t n, float dist,
                          const float* __restrict__ dist_table, int2* __restrict__ range, float4* __restrict__ box,
                          float* __restrict__ dmax) {
  __shared__ int smin[4], smax[4];
  __shared__ float sbox[4][4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  int lo = (live && group) ? group[i] : 0x7fffffff, hi = (live && group) ? group[i] : (int)0x80000000;
  const float x = live ? pts[i * stride + 0] : 0.f, y = live ? pts[i * stride + 1] : 0.f;
  float x0 = live ? x : INFINITY, x1 = live ? x : -INFINITY, y0 = live ? y : INFINITY, y1 = live ? y : -INFINITY;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
    x0 = fminf(x0, __shfl_xor(x0, o));
    x1 = fmaxf(x1, __shfl_xor(x1, o));
    y0 = fminf(y0, __shfl_xor(y0, o));
    y1 = fmaxf(y1, __shfl_xor(y1, o));
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    smin[w] = lo;
    smax[w] = hi;
    sbox[w][0] = x0; sbox[w][1] = x1; sbox[w][2] = y0; sbox[w][3] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
    hi = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    if (range) range[blockIdx.x] = make_int2(lo, hi);
    box[blockIdx.x] = make_float4(fminf(fminf(sbox[0][0], sbox[1][0]), fminf(sbox[2][0], sbox[3][0])),
                                  fmaxf(fmaxf(sbox[0][1], sbox[1][1]), fmaxf(sbox[2][1], sbox[3][1])),
                                  fminf(fminf(sbox[0][2], sbox[1][2]), fminf(sbox[2][2], sbox[3][2])),
                                  fmaxf(fmaxf(sbox[0][3], sbox[1][3]), fmaxf(sbox[2][3], sbox[3][3])));
    float d = dist;
    if (dist_table) {
      d = 0.f;
      for (int g = lo; g <= hi; ++g) d = fmaxf(d, dist_table[g]);
    }
    dmax[blockIdx.x] = d;
  }
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
  const int64_t tiles = (nn + 255) / 256;
  return fsf_align_up(nn * 4, 256) * 2 + fsf_align_up(scan_num_tiles(n) * 4, 256) + fsf_align_up(tiles * 8, 256) +
         fsf_align_up(tiles * 16, 256) + fsf_align_up(tiles * 4, 256) + 256;
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
  float4* tile_box = ar.take<float4>(tiles);
  float* tile_dmax = ar.take<float>(tiles);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  const int grid = fsf_stream_grid(n, 256);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(grid), dim3(256), 0, stream, parent, n);
  if (tiles > 65535) return FSF_ERR_UNSUPPORTED;  // (gridDim.y; 16.7 M points)
  hipLaunchKernelGGL(ccl_tile_range_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, points, (int)point_stride,
                     dist_table ? batch_idx : nullptr, n, dist, dist_table, tile_range, tile_box, tile_dmax);
  hipLaunchKernelGGL(ccl_pairs_kernel, dim3((unsigned)((tiles + CCL_TJ - 1) / CCL_TJ), (unsigned)tiles), dim3(256), 0, stream,
                     points, (int)point_stride, batch_idx,
                     n, dist, parent, (int)tiles, dist_table, (const int2*)tile_range, (const float4*)tile_box,
                     (const float*)tile_dmax);
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
