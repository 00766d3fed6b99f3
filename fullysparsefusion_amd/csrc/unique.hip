// K3: unique rows (+inverse, counts, CSR segment plan) by packed-key radix sort.  See include/fsf_hip.h.
//
// Pipeline: column min/max (device) -> pack each row into one u64 key that preserves lexicographic row
// order -> stable LSD radix sort over the significant bits only -> head flags + exclusive scan ->
// inverse / order / CSR offsets / decoded unique rows in one apply pass.
// Algorithmic HBM bytes: 8k B/row read + 8 B/row inverse + 8k B/unique row.
#include "common.h"
#include "radix_sort.h"
#include "scan.h"

namespace fsf {

struct ColRange {
  int64_t mn[4];
  int64_t mx[4];
};

struct PackSpec {
  int64_t mn[4];
  int64_t mx[4];
  int shift[4];
  uint64_t mask[4];
  int k;
};

__global__ void uq_range_init_kernel(ColRange* r) {
  int j = threadIdx.x;
  if (j < 4) {
    r->mn[j] = INT64_MAX;
    r->mx[j] = INT64_MIN;
  }
}

__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int64_t t = __shfl_xor(v, o);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int64_t t = __shfl_xor(v, o);
    v = t > v ? t : v;
  }
  return v;
}

template <int K>
__global__ void __launch_bounds__(256) uq_range_kernel(const int64_t* __restrict__ coors, int64_t n, ColRange* r) {
  int64_t mn[K], mx[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    mn[j] = INT64_MAX;
    mx[j] = INT64_MIN;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      int64_t v = coors[i * K + j];
      mn[j] = v < mn[j] ? v : mn[j];
      mx[j] = v > mx[j] ? v : mx[j];
    }
  }
  __shared__ int64_t smn[4][K], smx[4][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    int64_t a = wave_min_i64(mn[j]);
    int64_t b = wave_max_i64(mx[j]);
    if (lane == 0) {
      smn[wave][j] = a;
      smx[wave][j] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int j = threadIdx.x;
    int64_t a = smn[0][j], b = smx[0][j];
    for (int w = 1; w < 4; ++w) {
      a = smn[w][j] < a ? smn[w][j] : a;
      b = smx[w][j] > b ? smx[w][j] : b;
    }
    __hip_atomic_fetch_min(&r->mn[j], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(&r->mx[j], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int K>
__global__ void __launch_bounds__(256)
    uq_pack_kernel(const int64_t* __restrict__ coors, int64_t n, PackSpec spec, uint64_t* __restrict__ keys,
                   uint32_t* __restrict__ vals, int32_t* __restrict__ err_flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t key = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      int64_t v = coors[i * K + j];
      bad |= (v < spec.mn[j]) | (v > spec.mx[j]);
      key |= ((uint64_t)(v - spec.mn[j]) & spec.mask[j]) << spec.shift[j];
    }
    if (bad) *err_flag = 1;
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
}

struct HeadIn {
  const uint64_t* keys;
  __device__ uint32_t operator()(int64_t i) const { return (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; }
};

struct HeadOut {
  const uint64_t* keys;
  const uint32_t* vals;
  PackSpec spec;
  int64_t* new_coors;
  int64_t* inv;
  int32_t* order;
  int32_t* seg_offsets;
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t head) const {
    const uint32_t seg = excl + head - 1u;
    const uint32_t p = vals[i];
    inv[p] = (int64_t)seg;
    order[i] = (int32_t)p;
    if (head) {
      seg_offsets[seg] = (int32_t)i;
      const uint64_t key = keys[i];
      for (int j = 0; j < spec.k; ++j)
        new_coors[(int64_t)seg * spec.k + j] = (int64_t)((key >> spec.shift[j]) & spec.mask[j]) + spec.mn[j];
    }
  }
};

// `ret` (optional): [0] = m, [1] = the key-range error flag — what the host reads back, in ONE 16-byte copy
__global__ void uq_finish_kernel(const int64_t* __restrict__ m_dev, int64_t n, int32_t* __restrict__ seg_offsets,
                                 int64_t* __restrict__ cnt, const int32_t* __restrict__ err_flag = nullptr, int64_t* __restrict__ ret = nullptr) {
  const int64_t m = *m_dev;
  if (ret && blockIdx.x == 0 && threadIdx.x == 0) {
    ret[0] = m;
    ret[1] = *err_flag;
  }
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < m; s += (int64_t)gridDim.x * blockDim.x) {
    const int32_t lo = seg_offsets[s];
    const int32_t hi = (s + 1 < m) ? seg_offsets[s + 1] : (int32_t)n;
    if (cnt) cnt[s] = (int64_t)(hi - lo);
    if (s + 1 == m) seg_offsets[m] = (int32_t)n;
  }
  if (m == 0 && blockIdx.x == 0 && threadIdx.x == 0) seg_offsets[0] = 0;
}

static int bit_width_u64(uint64_t v) {
  int b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

// ---- segment plan from a caller-supplied inverse -------------------------------------------------
__global__ void __launch_bounds__(256)
    sp_pack_kernel(const int64_t* __restrict__ inv, int64_t n, int64_t m, uint64_t* __restrict__ keys,
                   uint32_t* __restrict__ vals, int32_t* __restrict__ err_flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = inv[i];
    if (v < 0 || v >= m) {
      *err_flag = 1;
      v = 0;
    }
    keys[i] = (uint64_t)v;
    vals[i] = (uint32_t)i;
  }
}

// one thread per sorted position (plus one past the end): fills seg_offsets for every segment id in
// (key[i-1], key[i]], which also covers empty segments.
__global__ void __launch_bounds__(256)
    sp_offsets_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n, int64_t m,
                      int32_t* __restrict__ order, int32_t* __restrict__ seg_offsets) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = (i == 0) ? -1 : (int64_t)keys[i - 1];
    const int64_t cur = (i == n) ? m : (int64_t)keys[i];
    for (int64_t s = prev + 1; s <= cur; ++s) seg_offsets[s] = (int32_t)i;
    if (i < n) order[i] = (int32_t)vals[i];
  }
}

__global__ void sp_counts_kernel(const int32_t* __restrict__ seg_offsets, int64_t m, int64_t* __restrict__ cnt) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < m; s += (int64_t)gridDim.x * blockDim.x)
    cnt[s] = (int64_t)(seg_offsets[s + 1] - seg_offsets[s]);
}

// ---- in-group rank (K18) --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    ig_rank_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                   const int32_t* __restrict__ head_pos, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[vals[i]] = (int64_t)(i - head_pos[i]);
}

struct IgIn {
  const uint64_t* keys;
  __device__ uint32_t operator()(int64_t i) const { return (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; }
};


// ---- K25: which cluster voxels (and which (group, point) pairs) survive ClusterAssigner's density filter ------------------------
// ClusterAssigner.forward_single_class (single_stage_fsd.py:951-956) per class group: a voxel key with fewer than `min_points` pairs
// is dropped — unless NONE of the group's keys is dense enough, then the group keeps everything.  With all groups in one key list
// (group = key[0] / batch size, keys sorted by it) that is: flag per key, OR per group, two stable compactions (keys, pairs) and the
// pair -> surviving-voxel map.  Upstream does it with boolean masks and a second torch.unique on the survivors (two host syncs per
// group); the plugin had ~22 small ATen launches and two nonzero() syncs here.
__global__ void __launch_bounds__(256)
    ks_group_valid_kernel(const int64_t* __restrict__ new_keys, int key_cols, const int64_t* __restrict__ cnt, int64_t m, int64_t bsz,
                          int64_t min_points, int ng, volatile int* has_valid) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    if (cnt[i] >= min_points) {
      const int64_t g = new_keys[i * key_cols] / bsz;
      // a plain store of 1 (idempotent, so the race is benign): a wave's lanes mostly share g and their stores merge into one
      // write, where one atomicOr per dense key serialised ~100k same-address atomics on one L2 channel (0.95 ms per 10-sweep frame)
      if (g >= 0 && g < ng && has_valid[g] == 0) has_valid[g] = 1;
    }
  }
}

struct KsKeyIn {
  const int64_t* new_keys; int key_cols; const int64_t* cnt; int64_t bsz, min_points; int ng; const int* has_valid;
  __device__ uint32_t operator()(int64_t i) const {
    const int64_t g = new_keys[i * key_cols] / bsz;
    const bool group_has = g >= 0 && g < ng && has_valid[g] != 0;
    return (cnt[i] >= min_points || !group_has) ? 1u : 0u;
  }
};
struct KsKeyOut {
  int64_t* k_idx; int32_t* kpos; int32_t* k_group; const int64_t* new_keys; int key_cols; int64_t bsz;
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t keep) const {
    kpos[i] = keep ? (int32_t)excl : -1;
    if (keep) {
      k_idx[excl] = i;
      if (k_group) k_group[excl] = (int32_t)(new_keys[i * key_cols] / bsz);
    }
  }
};

// cluster id of every surviving pair: the connected-component label of its voxel, renumbered from 0 inside its class group
// (labels are numbered by first member over ALL voxels and the voxels are group-sorted: a group's labels start at its first voxel's)
__global__ void __launch_bounds__(256)
    ks_group_base_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ vox_group, int64_t m, int ng, int32_t* __restrict__ base) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = vox_group[i];
    if ((i == 0 || vox_group[i - 1] != g) && g >= 0 && g < ng) base[g] = labels[i];
  }
}
__global__ void __launch_bounds__(256)
    ks_point_ids_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ vox_group, const int32_t* __restrict__ base,
                        const int64_t* __restrict__ vox_inv, const int64_t* __restrict__ g_ids, const int64_t* __restrict__ b_pts, int64_t nv,
                        int64_t* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nv; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = vox_inv[j];
    out[3 * j] = g_ids[j];
    out[3 * j + 1] = b_pts[j];
    out[3 * j + 2] = (int64_t)(labels[v] - base[vox_group[v]]);
  }
}
struct KsPairIn {
  const int64_t* inv; const int32_t* kpos;
  __device__ uint32_t operator()(int64_t i) const { return kpos[inv[i]] >= 0 ? 1u : 0u; }
};
struct KsPairOut {
  const int64_t* inv; const int32_t* kpos; int64_t* v_idx; int64_t* vox_inv;
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t keep) const {
    if (keep) {
      v_idx[excl] = i;
      vox_inv[excl] = (int64_t)kpos[inv[i]];
    }
  }
};

// ---- K27: the (group, point) pairs of SingleStageFSD's grouped sampling -----------------------------------------------------------
// `fg = grouped_score > thresh[None, :]; fg[0] |= ~fg.any(0); gp = fg.t().nonzero()` (single_stage_fsd.py:826-838 for one sample: every
// class group keeps the points whose summed class scores pass the group's threshold, a group nobody passes keeps point 0) as: one pass
// that ORs the groups somebody passes into a word, one scan over the (group, point) grid in group-major order that writes the pairs.
// Replaces a compare, an any-reduction, three boolean elementwise ops, a transposing copy and ATen's nonzero (~12 launches and its
// blocking count read-back).
// `groups.use`: `score` holds CLASS scores and group g's score is the sum of its one or two member columns (lo, hi) — what
// `scores @ member.t()` / `scores[:, cols].sum(1)` give for such groups, bit for bit (one add of two floats has one result)
struct GpGroups {
  int use;
  unsigned char lo[32], hi[32];  // hi == 255: one member
};
__device__ __forceinline__ float gp_score(const float* row, int g, const GpGroups& groups) {
  if (!groups.use) return row[g];
  const float a = row[groups.lo[g]];
  return groups.hi[g] == 255 ? a : __fadd_rn(a, row[groups.hi[g]]);
}
__global__ void __launch_bounds__(256)
    gp_any_kernel(const float* __restrict__ score, int64_t n, int ng, int64_t stride, const float* __restrict__ thresh, GpGroups groups,
                  uint32_t* __restrict__ any_mask) {
  uint32_t mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    for (int g = 0; g < ng; ++g)
      if (gp_score(score + i * stride, g, groups) > thresh[g]) mine |= 1u << g;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= (uint32_t)__shfl_xor((int)mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicOr(any_mask, mine);
}
struct GpIn {
  const float* score; int64_t n; int64_t stride; const float* thresh; const uint32_t* any_mask; int keep_one; GpGroups groups;
  __device__ uint32_t operator()(int64_t t) const {
    const int g = (int)(t / n);
    const int64_t i = t - (int64_t)g * n;
    if (gp_score(score + i * stride, g, groups) > thresh[g]) return 1u;
    return (keep_one && i == 0 && !((*any_mask >> g) & 1u)) ? 1u : 0u;
  }
};
struct GpOut {
  int64_t n; int64_t* g_ids; int64_t* p_ids;
  __device__ void operator()(int64_t t, uint32_t excl, uint32_t keep) const {
    if (keep) {
      const int64_t g = t / n;
      g_ids[excl] = g;
      p_ids[excl] = t - g * n;
    }
  }
};
}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_unique_rows_workspace_bytes(int64_t n, int32_t k) {
  (void)k;
  return radix_sort_scratch_bytes(n) + fsf_align_up(scan_num_tiles(n) * 4, 256) + 4 * 256;
}

extern "C" int fsf_unique_rows(const int64_t* coors, int64_t n, int32_t k, const int64_t* col_min,
                               const int64_t* col_max, int64_t* new_coors, int64_t* inv, int64_t* cnt,
                               int32_t* order, int32_t* seg_offsets, int64_t* m_dev, int64_t* m_host, void* workspace,
                               int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || k < 1 || k > 4 || !m_dev || (n > 0 && (!coors || !new_coors || !inv || !order)) || !seg_offsets)
    return FSF_ERR_INVALID_ARG;
  if (n >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  if (workspace_bytes < fsf_unique_rows_workspace_bytes(n, k)) return FSF_ERR_WORKSPACE;
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(m_dev, 0, sizeof(int64_t), stream));
    FSF_HIP_TRY(hipMemsetAsync(seg_offsets, 0, sizeof(int32_t), stream));
    if (m_host) {
      FSF_STREAM_WAIT(stream);
      *m_host = 0;
    }
    return FSF_OK;
  }
  FsfArena ar(workspace, workspace_bytes);
  uint64_t* keys_a = ar.take<uint64_t>(n);
  uint64_t* keys_b = ar.take<uint64_t>(n);
  uint32_t* vals_a = ar.take<uint32_t>(n);
  uint32_t* vals_b = ar.take<uint32_t>(n);
  // [hist | tile_sums | err_flag (+ the read-back pair)] are consecutive: ONE memset clears what the sort, the scan and the packing
  // kernel each used to clear for themselves
  uint32_t* hist = ar.take<uint32_t>((radix_num_tiles(n) + 1) * RS_BINS);
  uint32_t* tile_sums = ar.take<uint32_t>(scan_num_tiles(n));
  int32_t* err_flag = ar.take<int32_t>(1);
  int64_t* ret_dev = ar.take<int64_t>(2);
  const size_t zero_bytes = (size_t)((char*)ret_dev - (char*)hist);
  ColRange* range_dev = ar.take<ColRange>(1);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;

  const int grid = fsf_stream_grid(n, 256);
  const int rgrid = grid > 512 ? 512 : grid;  // one atomic pair per column per workgroup
  ColRange range;
  if (col_min && col_max) {
    for (int j = 0; j < k; ++j) {
      range.mn[j] = col_min[j];
      range.mx[j] = col_max[j];
      if (range.mx[j] < range.mn[j]) return FSF_ERR_INVALID_ARG;
    }
  } else {
    // data-dependent bounds: one small D2H copy + sync (torch.unique in the reference syncs as well)
    hipLaunchKernelGGL(uq_range_init_kernel, dim3(1), dim3(64), 0, stream, range_dev);
    switch (k) {
      case 1: hipLaunchKernelGGL((uq_range_kernel<1>), dim3(rgrid), dim3(256), 0, stream, coors, n, range_dev); break;
      case 2: hipLaunchKernelGGL((uq_range_kernel<2>), dim3(rgrid), dim3(256), 0, stream, coors, n, range_dev); break;
      case 3: hipLaunchKernelGGL((uq_range_kernel<3>), dim3(rgrid), dim3(256), 0, stream, coors, n, range_dev); break;
      default: hipLaunchKernelGGL((uq_range_kernel<4>), dim3(rgrid), dim3(256), 0, stream, coors, n, range_dev); break;
    }
    FSF_READ_BACK(&range, range_dev, sizeof(ColRange), stream);
  }
  PackSpec spec;
  spec.k = k;
  int total_bits = 0;
  int bits[4] = {0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) {
    spec.mn[j] = 0;
    spec.mx[j] = 0;
    spec.shift[j] = 0;
    spec.mask[j] = 0;
  }
  for (int j = 0; j < k; ++j) {
    spec.mn[j] = range.mn[j];
    spec.mx[j] = range.mx[j];
    const uint64_t span = (uint64_t)range.mx[j] - (uint64_t)range.mn[j];
    bits[j] = bit_width_u64(span);
    total_bits += bits[j];
  }
  if (total_bits > 64) return FSF_ERR_KEY_RANGE;
  int sh = 0;
  for (int j = k - 1; j >= 0; --j) {
    spec.shift[j] = sh;
    spec.mask[j] = bits[j] >= 64 ? ~0ull : ((1ull << bits[j]) - 1ull);
    sh += bits[j];
  }
  FSF_HIP_TRY(hipMemsetAsync(hist, 0, zero_bytes, stream));
  switch (k) {
    case 1: hipLaunchKernelGGL((uq_pack_kernel<1>), dim3(grid), dim3(256), 0, stream, coors, n, spec, keys_a, vals_a, err_flag); break;
    case 2: hipLaunchKernelGGL((uq_pack_kernel<2>), dim3(grid), dim3(256), 0, stream, coors, n, spec, keys_a, vals_a, err_flag); break;
    case 3: hipLaunchKernelGGL((uq_pack_kernel<3>), dim3(grid), dim3(256), 0, stream, coors, n, spec, keys_a, vals_a, err_flag); break;
    default: hipLaunchKernelGGL((uq_pack_kernel<4>), dim3(grid), dim3(256), 0, stream, coors, n, spec, keys_a, vals_a, err_flag); break;
  }
  uint64_t* keys_s;
  uint32_t* vals_s;
  int rc = radix_sort_pairs(keys_a, vals_a, keys_b, vals_b, hist, n, total_bits, &keys_s, &vals_s, stream, true);
  if (rc != FSF_OK) return rc;
  HeadIn hin{keys_s};
  HeadOut hout{keys_s, vals_s, spec, new_coors, inv, order, seg_offsets};
  rc = exclusive_scan_u32(hin, hout, n, tile_sums, nullptr, m_dev, stream, 1, true);
  if (rc != FSF_OK) return rc;
  hipLaunchKernelGGL(uq_finish_kernel, dim3(grid), dim3(256), 0, stream, m_dev, n, seg_offsets, cnt, err_flag, m_host ? ret_dev : nullptr);
  FSF_LAUNCH_CHECK();
  if (m_host) {
    int64_t ret_h[2] = {0, 0};
    FSF_READ_BACK(ret_h, ret_dev, sizeof(ret_h), stream);  // (count + error flag in one copy)
    *m_host = ret_h[0];
    if (ret_h[1]) return FSF_ERR_KEY_RANGE;
  }
  return FSF_OK;
}

extern "C" int64_t fsf_segment_plan_workspace_bytes(int64_t n, int64_t m) {
  (void)m;
  return radix_sort_scratch_bytes(n) + 2 * 256;
}

extern "C" int fsf_segment_plan_from_inverse(const int64_t* inv, int64_t n, int64_t m, int32_t* order,
                                             int32_t* seg_offsets, int64_t* cnt, void* workspace,
                                             int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || !seg_offsets || (n > 0 && (!inv || !order))) return FSF_ERR_INVALID_ARG;
  if (n >= (int64_t)1 << 31 || m >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  if (workspace_bytes < fsf_segment_plan_workspace_bytes(n, m)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  uint64_t* keys_a = ar.take<uint64_t>(n);
  uint64_t* keys_b = ar.take<uint64_t>(n);
  uint32_t* vals_a = ar.take<uint32_t>(n);
  uint32_t* vals_b = ar.take<uint32_t>(n);
  uint32_t* hist = ar.take<uint32_t>((radix_num_tiles(n) + 1) * RS_BINS);
  int32_t* err_flag = ar.take<int32_t>(1);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  const int grid = fsf_stream_grid(n + 1, 256);
  FSF_HIP_TRY(hipMemsetAsync(err_flag, 0, sizeof(int32_t), stream));
  uint64_t* keys_s = keys_a;
  uint32_t* vals_s = vals_a;
  if (n > 0) {
    hipLaunchKernelGGL(sp_pack_kernel, dim3(grid), dim3(256), 0, stream, inv, n, m, keys_a, vals_a, err_flag);
    int rc = radix_sort_pairs(keys_a, vals_a, keys_b, vals_b, hist, n, bit_width_u64(m > 0 ? (uint64_t)(m - 1) : 0),
                              &keys_s, &vals_s, stream);
    if (rc != FSF_OK) return rc;
  }
  hipLaunchKernelGGL(sp_offsets_kernel, dim3(grid), dim3(256), 0, stream, keys_s, vals_s, n, m, order, seg_offsets);
  if (cnt && m > 0)
    hipLaunchKernelGGL(sp_counts_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, seg_offsets, m, cnt);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_ingroup_rank_workspace_bytes(int64_t n) {
  return radix_sort_scratch_bytes(n) + fsf_align_up(scan_num_tiles(n) * 4, 256) + fsf_align_up((n > 0 ? n : 1) * 4, 256) +
         4 * 256;
}

namespace fsf {
// head position of each sorted element = running max of (head ? i : 0): computed by a scan of head flags
// (segment id) followed by a lookup of the segment's start written at the heads.
struct IgOut {
  int32_t* seg_start;  // [n] scratch: start position per segment id
  int32_t* seg_of;     // [n] scratch reused as head_pos afterwards
  __device__ void operator()(int64_t i, uint32_t excl, uint32_t head) const {
    const uint32_t seg = excl + head - 1u;
    if (head) seg_start[seg] = (int32_t)i;
    seg_of[i] = (int32_t)seg;
  }
};
__global__ void __launch_bounds__(256)
    ig_final_kernel(const uint32_t* __restrict__ vals, const int32_t* __restrict__ seg_of,
                    const int32_t* __restrict__ seg_start, int64_t n, int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[vals[i]] = (int64_t)i - (int64_t)seg_start[seg_of[i]];
}
}  // namespace fsf

extern "C" int fsf_ingroup_rank(const int64_t* group_inds, int64_t n, int64_t* out_inds, void* workspace,
                                int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || (n > 0 && (!group_inds || !out_inds))) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (n >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  if (workspace_bytes < fsf_ingroup_rank_workspace_bytes(n)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  uint64_t* keys_a = ar.take<uint64_t>(n);
  uint64_t* keys_b = ar.take<uint64_t>(n);
  uint32_t* vals_a = ar.take<uint32_t>(n);
  uint32_t* vals_b = ar.take<uint32_t>(n);
  uint32_t* hist = ar.take<uint32_t>((radix_num_tiles(n) + 1) * RS_BINS);
  uint32_t* tile_sums = ar.take<uint32_t>(scan_num_tiles(n));
  int32_t* seg_start = ar.take<int32_t>(n);
  ColRange* range_dev = ar.take<ColRange>(1);
  int32_t* err_flag = ar.take<int32_t>(1);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  const int grid = fsf_stream_grid(n, 256);
  ColRange range;
  hipLaunchKernelGGL(uq_range_init_kernel, dim3(1), dim3(64), 0, stream, range_dev);
  hipLaunchKernelGGL((uq_range_kernel<1>), dim3(grid > 512 ? 512 : grid), dim3(256), 0, stream, group_inds, n, range_dev);
  FSF_READ_BACK(&range, range_dev, sizeof(ColRange), stream);
  PackSpec spec;
  for (int j = 0; j < 4; ++j) {
    spec.mn[j] = 0; spec.mx[j] = 0; spec.shift[j] = 0; spec.mask[j] = 0;
  }
  spec.k = 1;
  spec.mn[0] = range.mn[0];
  spec.mx[0] = range.mx[0];
  const int bits = bit_width_u64((uint64_t)range.mx[0] - (uint64_t)range.mn[0]);
  spec.mask[0] = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
  FSF_HIP_TRY(hipMemsetAsync(err_flag, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL((uq_pack_kernel<1>), dim3(grid), dim3(256), 0, stream, group_inds, n, spec, keys_a, vals_a, err_flag);
  uint64_t* keys_s;
  uint32_t* vals_s;
  int rc = radix_sort_pairs(keys_a, vals_a, keys_b, vals_b, hist, n, bits, &keys_s, &vals_s, stream);
  if (rc != FSF_OK) return rc;
  // seg_of aliases the no-longer-needed alternate value buffer
  int32_t* seg_of = (int32_t*)(vals_s == vals_a ? vals_b : vals_a);
  IgIn iin{keys_s};
  IgOut iout{seg_start, seg_of};
  rc = exclusive_scan_u32(iin, iout, n, tile_sums, nullptr, nullptr, stream);
  if (rc != FSF_OK) return rc;
  hipLaunchKernelGGL(ig_final_kernel, dim3(grid), dim3(256), 0, stream, vals_s, seg_of, seg_start, n, out_inds);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_cluster_key_survival_workspace_bytes(int64_t m, int64_t n) {
  return fsf_align_up((m > 0 ? m : 1) * 4, 256) + fsf_align_up(scan_num_tiles(m) * 4, 256) + fsf_align_up(scan_num_tiles(n) * 4, 256) + 6 * 256;
}

extern "C" int fsf_cluster_key_survival(const int64_t* new_keys, int32_t key_cols, const int64_t* cnt, int64_t m, const int64_t* inv,
                                        int64_t n, int64_t batch_size, int64_t min_points, int32_t num_groups, int64_t* k_idx,
                                        int32_t* k_group, int64_t* v_idx, int64_t* vox_inv, int64_t* counts_host, void* workspace,
                                        int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || n < 0 || key_cols < 1 || batch_size < 1 || num_groups < 1 || num_groups > 64 || !counts_host ||
      (m > 0 && (!new_keys || !cnt || !k_idx)) || (n > 0 && (!inv || !v_idx || !vox_inv)))
    return FSF_ERR_INVALID_ARG;
  if (m >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31)) return FSF_ERR_UNSUPPORTED;
  if (workspace_bytes < fsf_cluster_key_survival_workspace_bytes(m, n) || !workspace) return FSF_ERR_WORKSPACE;
  counts_host[0] = counts_host[1] = 0;
  if (m == 0 || n == 0) return FSF_OK;
  FsfArena ar(workspace, workspace_bytes);
  int32_t* kpos = ar.take<int32_t>(m);
  // [has_valid | tile sums of the two scans] are consecutive: one memset
  int* has_valid = ar.take<int>(64);
  uint32_t* tiles_k = ar.take<uint32_t>(scan_num_tiles(m));
  uint32_t* tiles_p = ar.take<uint32_t>(scan_num_tiles(n));
  int64_t* totals = ar.take<int64_t>(2);
  const size_t zero_bytes = (size_t)((char*)totals - (char*)has_valid);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  FSF_HIP_TRY(hipMemsetAsync(has_valid, 0, zero_bytes, stream));
  hipLaunchKernelGGL(ks_group_valid_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, new_keys, (int)key_cols, cnt, m,
                     batch_size, min_points, (int)num_groups, has_valid);
  int rc = exclusive_scan_u32(KsKeyIn{new_keys, (int)key_cols, cnt, batch_size, min_points, (int)num_groups, has_valid},
                              KsKeyOut{k_idx, kpos, k_group, new_keys, (int)key_cols, batch_size}, m, tiles_k, nullptr, totals, stream, 1, true);
  if (rc != FSF_OK) return rc;
  rc = exclusive_scan_u32(KsPairIn{inv, kpos}, KsPairOut{inv, kpos, v_idx, vox_inv}, n, tiles_p, nullptr, totals + 1, stream, 1, true);
  if (rc != FSF_OK) return rc;
  int64_t ret_h[2] = {0, 0};
  FSF_READ_BACK(ret_h, totals, sizeof(ret_h), stream);  // (both counts in one copy)
  counts_host[0] = ret_h[0];
  counts_host[1] = ret_h[1];
  return FSF_OK;
}

extern "C" int fsf_cluster_point_ids(const int32_t* labels, const int32_t* vox_group, int64_t m, const int64_t* vox_inv,
                                     const int64_t* g_ids, const int64_t* b_pts, int64_t nv, int32_t num_groups, int64_t* out,
                                     void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || nv < 0 || num_groups < 1 || num_groups > 64 || (nv > 0 && (!labels || !vox_group || !vox_inv || !g_ids || !b_pts || !out || m < 1)))
    return FSF_ERR_INVALID_ARG;
  if (workspace_bytes < 256 || !workspace) return FSF_ERR_WORKSPACE;
  if (nv == 0) return FSF_OK;
  int32_t* base = (int32_t*)workspace;
  FSF_HIP_TRY(hipMemsetAsync(base, 0, 64 * sizeof(int32_t), stream));
  hipLaunchKernelGGL(ks_group_base_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, labels, vox_group, m, (int)num_groups, base);
  hipLaunchKernelGGL(ks_point_ids_kernel, dim3(fsf_stream_grid(nv, 256)), dim3(256), 0, stream, labels, vox_group, base, vox_inv, g_ids, b_pts,
                     nv, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_group_pairs_workspace_bytes(int64_t n, int32_t ng) {
  return fsf_align_up(scan_num_tiles((n > 0 ? n : 1) * (ng > 0 ? ng : 1)) * 4, 256) + 2 * 256;
}

extern "C" int fsf_group_pairs(const float* score, int64_t n, int32_t ng, int64_t score_stride, const float* thresh, int32_t keep_one,
                               const uint32_t* group_class_masks, int32_t num_classes, int64_t* g_ids, int64_t* p_ids, int64_t capacity,
                               int64_t* count_host, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int cols = group_class_masks ? num_classes : ng;
  if (n < 0 || ng < 1 || !thresh || !count_host || score_stride < cols || (n > 0 && (!score || !g_ids || !p_ids))) return FSF_ERR_INVALID_ARG;
  if (ng > 32 || n * (int64_t)ng >= ((int64_t)1 << 30)) return FSF_ERR_UNSUPPORTED;
  GpGroups groups;
  groups.use = group_class_masks ? 1 : 0;
  for (int g = 0; g < 32; ++g) groups.lo[g] = 0, groups.hi[g] = 255;
  if (group_class_masks) {  // (a HOST array: one or two member classes per group, else the caller sums the columns itself)
    if (num_classes < 1 || num_classes > 32) return FSF_ERR_UNSUPPORTED;
    for (int g = 0; g < ng; ++g) {
      const uint32_t m = group_class_masks[g];
      const int members = __builtin_popcount(m);
      if (members < 1 || members > 2 || (num_classes < 32 && (m >> num_classes))) return FSF_ERR_UNSUPPORTED;
      groups.lo[g] = (unsigned char)__builtin_ctz(m);
      if (members == 2) groups.hi[g] = (unsigned char)(31 - __builtin_clz(m));
    }
  }
  if (capacity < n * ng) return FSF_ERR_CAPACITY;  // (the caller allocates the upper bound: the count is only known afterwards)
  *count_host = 0;
  if (n == 0) return FSF_OK;
  if (!workspace || workspace_bytes < fsf_group_pairs_workspace_bytes(n, ng)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  uint32_t* any_mask = ar.take<uint32_t>(1);  // [any_mask | the scan's tile words]: one memset
  uint32_t* tiles = ar.take<uint32_t>(scan_num_tiles(n * ng));
  int64_t* total = ar.take<int64_t>(1);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  FSF_HIP_TRY(hipMemsetAsync(any_mask, 0, (size_t)((char*)total - (char*)any_mask), stream));
  if (keep_one)
    hipLaunchKernelGGL(gp_any_kernel, dim3(fsf_stream_grid(n, 256)), dim3(256), 0, stream, score, n, (int)ng, score_stride, thresh, groups,
                       any_mask);
  const int rc = exclusive_scan_u32(GpIn{score, n, score_stride, thresh, any_mask, (int)keep_one, groups}, GpOut{n, g_ids, p_ids}, n * ng, tiles, nullptr,
                                    total, stream, 1, true);
  if (rc != FSF_OK) return rc;
  int64_t total_h = 0;
  FSF_READ_BACK(&total_h, total, sizeof(int64_t), stream);
  *count_host = total_h;
  return FSF_OK;
}
