// K7/K8: sparse-conv rulebooks through an open-addressing hash table in HBM (spconv v1 uses a dense
// B*Z*Y*X int grid: 42 MB for nuScenes, 537 MB per sample for Argoverse 2).  See include/fsf_hip.h.
// Output-major neighbour tables: nbr[o*kvol + k] = input row or -1.
// Algorithmic HBM bytes: 16 B/active voxel read, 4*kvol B/out voxel written
// (the dense table; 8 B/pair in spconv's pair-list form) + 16 B/out voxel.
#include "common.h"
#include "radix_sort.h"

namespace fsf {

constexpr uint64_t HASH_EMPTY = ~0ull;

struct ConvGeom {
  int B, Z, Y, X;     // input spatial shape
  int OZ, OY, OX;     // output spatial shape
  int kz, ky, kx;     // kernel size
  int sz, sy, sx;     // stride
  int pz, py, px;     // padding
  int dz, dy, dx;     // dilation
  int kvol;
};

__device__ __forceinline__ uint64_t hash_mix(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

__device__ __forceinline__ void hash_insert(uint64_t* __restrict__ keys, int32_t* __restrict__ vals, uint64_t mask,
                                            uint64_t key, int32_t val) {
  uint64_t slot = hash_mix(key) & mask;
  while (true) {
    const uint64_t prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)HASH_EMPTY, (unsigned long long)key);
    if (prev == HASH_EMPTY || prev == key) {
      vals[slot] = val;
      return;
    }
    slot = (slot + 1) & mask;
  }
}

// returns true when `key` was not present before (first inserter)
__device__ __forceinline__ bool hash_insert_set(uint64_t* __restrict__ keys, uint64_t mask, uint64_t key) {
  uint64_t slot = hash_mix(key) & mask;
  while (true) {
    const uint64_t prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)HASH_EMPTY, (unsigned long long)key);
    if (prev == HASH_EMPTY) return true;
    if (prev == key) return false;
    slot = (slot + 1) & mask;
  }
}

__device__ __forceinline__ int32_t hash_lookup(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                               uint64_t mask, uint64_t key) {
  uint64_t slot = hash_mix(key) & mask;
  while (true) {
    const uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == HASH_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
}

// (`zero_word`: a counter the same call needs cleared — one launch instead of a fill and a memset)
__global__ void __launch_bounds__(256) hash_fill_empty_kernel(uint64_t* keys, int64_t cap, uint32_t* zero_word = nullptr) {
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x)
    keys[i] = HASH_EMPTY;
}

__device__ __forceinline__ uint64_t lin_in(const ConvGeom& g, int b, int z, int y, int x) {
  return (((uint64_t)b * g.Z + z) * g.Y + y) * g.X + x;
}
__device__ __forceinline__ uint64_t lin_out(const ConvGeom& g, int b, int z, int y, int x) {
  return (((uint64_t)b * g.OZ + z) * g.OY + y) * g.OX + x;
}

__global__ void __launch_bounds__(256)
    rb_insert_inputs_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g, uint64_t* keys, int32_t* vals,
                            uint64_t mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = *reinterpret_cast<const int4*>(indices + i * 4);
    hash_insert(keys, vals, mask, lin_in(g, c.x, c.y, c.z, c.w), (int32_t)i);
  }
}

// SubM: out coords == in coords; nbr[o][k] = row at o + k*dil - pad
__global__ void __launch_bounds__(256)
    rb_subm_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g, const uint64_t* __restrict__ keys,
                   const int32_t* __restrict__ vals, uint64_t mask, int32_t* __restrict__ nbr) {
  const int64_t total = m * g.kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / g.kvol;
    const int k = (int)(t - o * g.kvol);
    const int kxi = k % g.kx, kyi = (k / g.kx) % g.ky, kzi = k / (g.kx * g.ky);
    const int4 c = *reinterpret_cast<const int4*>(indices + o * 4);
    const int z = c.y + kzi * g.dz - g.pz;
    const int y = c.z + kyi * g.dy - g.py;
    const int x = c.w + kxi * g.dx - g.px;
    int32_t r = -1;
    if (z >= 0 && z < g.Z && y >= 0 && y < g.Y && x >= 0 && x < g.X)
      r = hash_lookup(keys, vals, mask, lin_in(g, c.x, z, y, x));
    nbr[t] = r;
  }
}

// strided conv, pass 1: every (input, offset) proposes an output site; first proposer appends it.  The append
// position comes from a workgroup-level count (LDS atomics) and ONE device atomic per workgroup and round: a single
// device-wide counter bumped by every wave saturates at ~90 increments / us and was the whole cost of this kernel.
__global__ void __launch_bounds__(256)
    rb_propose_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g, uint64_t* set_keys, uint64_t set_mask,
                      uint64_t* __restrict__ list, uint32_t* __restrict__ list_count) {
  __shared__ uint32_t s_cnt, s_base;
  const int64_t total = m * g.kvol;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (total + step - 1) / step;  // block-uniform trip count (barriers inside)
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t t = r * step + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    bool mine = false;
    uint64_t key = 0;
    if (t < total) {
      const int64_t i = t / g.kvol;
      const int k = (int)(t - i * g.kvol);
      const int kxi = k % g.kx, kyi = (k / g.kx) % g.ky, kzi = k / (g.kx * g.ky);
      const int4 c = *reinterpret_cast<const int4*>(indices + i * 4);
      const int nz = c.y + g.pz - kzi * g.dz;
      const int ny = c.z + g.py - kyi * g.dy;
      const int nx = c.w + g.px - kxi * g.dx;
      if (nz >= 0 && ny >= 0 && nx >= 0 && !(nz % g.sz || ny % g.sy || nx % g.sx)) {
        const int oz = nz / g.sz, oy = ny / g.sy, ox = nx / g.sx;
        if (oz < g.OZ && oy < g.OY && ox < g.OX) {
          key = lin_out(g, c.x, oz, oy, ox);
          mine = hash_insert_set(set_keys, set_mask, key);
        }
      }
    }
    uint32_t local = 0;
    if (mine) local = atomicAdd(&s_cnt, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(list_count, s_cnt);
    __syncthreads();
    if (mine) list[s_base + local] = key;
  }
}

// The same pass with a thread per INPUT voxel (round 3): a voxel walks its kvol offsets itself — most fail the stride test at once (a
// stride-2 3 x 3 x 3 convolution admits at most 8 of 27) — and parks the sites it was first to propose in an LDS list of the
// workgroup (at most `per_in` per thread); one device atomic and one coalesced copy per workgroup.  m * kvol threads, two barriers per
// 524 k of them, took 90 us on the 92 k-voxel level; the order of `list` differs, the sorted result does not.
__global__ void __launch_bounds__(256)
    rb_propose_rows_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g, uint64_t* set_keys, uint64_t set_mask,
                           uint64_t* __restrict__ list, uint32_t* __restrict__ list_count) {
  extern __shared__ __attribute__((aligned(16))) char rbp_smem[];
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(rbp_smem);  // [256 * per_in]
  __shared__ uint32_t s_cnt, s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) {
    const int4 c = *reinterpret_cast<const int4*>(indices + i * 4);
    for (int kzi = 0; kzi < g.kz; ++kzi) {
      const int nz = c.y + g.pz - kzi * g.dz;
      if (nz < 0 || nz % g.sz || nz / g.sz >= g.OZ) continue;
      for (int kyi = 0; kyi < g.ky; ++kyi) {
        const int ny = c.z + g.py - kyi * g.dy;
        if (ny < 0 || ny % g.sy || ny / g.sy >= g.OY) continue;
        for (int kxi = 0; kxi < g.kx; ++kxi) {
          const int nx = c.w + g.px - kxi * g.dx;
          if (nx < 0 || nx % g.sx || nx / g.sx >= g.OX) continue;
          const uint64_t key = lin_out(g, c.x, nz / g.sz, ny / g.sy, nx / g.sx);
          if (hash_insert_set(set_keys, set_mask, key)) s_keys[atomicAdd(&s_cnt, 1u)] = key;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n_new = s_cnt;
  if (threadIdx.x == 0 && n_new) s_base = atomicAdd(list_count, n_new);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < n_new; t += blockDim.x) list[s_base + t] = s_keys[t];
}

__global__ void __launch_bounds__(256) rb_iota_kernel(uint32_t* v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v[i] = (uint32_t)i;
}

// pass 2: sorted output sites -> out_indices rows, out-site hash map
__global__ void __launch_bounds__(256)
    rb_assign_out_kernel(const uint64_t* __restrict__ sorted, int64_t m_out, ConvGeom g, int32_t* __restrict__ out_indices,
                         uint64_t* out_keys, int32_t* out_vals, uint64_t out_mask) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m_out; r += (int64_t)gridDim.x * blockDim.x) {
    uint64_t key = sorted[r];
    hash_insert(out_keys, out_vals, out_mask, key, (int32_t)r);
    const int x = (int)(key % g.OX);
    key /= g.OX;
    const int y = (int)(key % g.OY);
    key /= g.OY;
    const int z = (int)(key % g.OZ);
    key /= g.OZ;
    *reinterpret_cast<int4*>(out_indices + r * 4) = make_int4((int)key, z, y, x);
  }
}

// pass 3a: nbr[o][k] = input row at o*stride - pad + k*dil
__global__ void __launch_bounds__(256)
    rb_strided_nbr_kernel(const int32_t* __restrict__ out_indices, int64_t m_out, ConvGeom g,
                          const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, uint64_t mask,
                          int32_t* __restrict__ nbr) {
  const int64_t total = m_out * g.kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / g.kvol;
    const int k = (int)(t - o * g.kvol);
    const int kxi = k % g.kx, kyi = (k / g.kx) % g.ky, kzi = k / (g.kx * g.ky);
    const int4 c = *reinterpret_cast<const int4*>(out_indices + o * 4);
    const int z = c.y * g.sz - g.pz + kzi * g.dz;
    const int y = c.z * g.sy - g.py + kyi * g.dy;
    const int x = c.w * g.sx - g.px + kxi * g.dx;
    int32_t r = -1;
    if (z >= 0 && z < g.Z && y >= 0 && y < g.Y && x >= 0 && x < g.X)
      r = hash_lookup(keys, vals, mask, lin_in(g, c.x, z, y, x));
    nbr[t] = r;
  }
}

// pass 3b: nbr_inv[i][k] = output row (i + pad - k*dil)/stride — the table SparseInverseConv3d runs on
__global__ void __launch_bounds__(256)
    rb_strided_inv_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g,
                          const uint64_t* __restrict__ out_keys, const int32_t* __restrict__ out_vals, uint64_t out_mask,
                          int32_t* __restrict__ nbr_inv) {
  const int64_t total = m * g.kvol;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / g.kvol;
    const int k = (int)(t - i * g.kvol);
    const int kxi = k % g.kx, kyi = (k / g.kx) % g.ky, kzi = k / (g.kx * g.ky);
    const int4 c = *reinterpret_cast<const int4*>(indices + i * 4);
    const int nz = c.y + g.pz - kzi * g.dz;
    const int ny = c.z + g.py - kyi * g.dy;
    const int nx = c.w + g.px - kxi * g.dx;
    int32_t r = -1;
    if (nz >= 0 && ny >= 0 && nx >= 0 && !(nz % g.sz) && !(ny % g.sy) && !(nx % g.sx)) {
      const int oz = nz / g.sz, oy = ny / g.sy, ox = nx / g.sx;
      if (oz < g.OZ && oy < g.OY && ox < g.OX) r = hash_lookup(out_keys, out_vals, out_mask, lin_out(g, c.x, oz, oy, ox));
    }
    nbr_inv[t] = r;
  }
}

// nbr table -> spconv v1 pair lists: for each offset k, the (in,out) pairs in ascending out row.
// Two launches: (1) every RBP_SEG-row segment counts its live entries per offset (coalesced walk of the table, LDS counters),
// (2) workgroup (segment, offset) sums the counts of the segments before it and compacts its own rows in order.
// (One workgroup per offset walking all rows in 256-row steps with three barriers each was 160 us on a 1e5-row level.)
constexpr int RBP_SEG = 256;  // rows per counting segment (2048: 50 workgroups on the 101 k-row table, 95 us per layer)

__global__ void __launch_bounds__(256)
    rb_pairs_count_kernel(const int32_t* __restrict__ nbr, int64_t m_out, int kvol, int32_t* __restrict__ cnt) {
  extern __shared__ int32_t c_s[];  // [kvol]
  for (int k = threadIdx.x; k < kvol; k += 256) c_s[k] = 0;
  __syncthreads();
  const int64_t e0 = (int64_t)blockIdx.x * RBP_SEG * kvol;
  const int64_t rows = m_out - (int64_t)blockIdx.x * RBP_SEG < RBP_SEG ? m_out - (int64_t)blockIdx.x * RBP_SEG : RBP_SEG;
  const int64_t n = rows * kvol;
  for (int64_t t = threadIdx.x; t < n; t += 256)
    if (nbr[e0 + t] >= 0) atomicAdd(&c_s[(int)(t % kvol)], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < kvol; k += 256) cnt[(int64_t)blockIdx.x * kvol + k] = c_s[k];
}

__global__ void __launch_bounds__(256)
    rb_pairs_fill_kernel(const int32_t* __restrict__ nbr, int64_t m_out, int kvol, const int32_t* __restrict__ cnt,
                         int32_t* __restrict__ pairs, int64_t cap, int32_t* __restrict__ num) {
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t base_s;
  const int seg = blockIdx.x, k = blockIdx.y, nseg = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // pairs before this segment (and, for the last segment, the total)
  uint32_t before = 0;
  for (int s = threadIdx.x; s < seg; s += 256) before += (uint32_t)cnt[(int64_t)s * kvol + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
  if (lane == 0) wtot[wave] = before;
  __syncthreads();
  if (threadIdx.x == 0) {
    base_s = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (seg == nseg - 1) num[k] = (int32_t)(base_s + (uint32_t)cnt[(int64_t)seg * kvol + k]);
  }
  __syncthreads();
  const int64_t o_begin = (int64_t)seg * RBP_SEG;
  const int64_t o_end = o_begin + RBP_SEG < m_out ? o_begin + RBP_SEG : m_out;
  for (int64_t o0 = o_begin; o0 < o_end; o0 += 256) {
    const int64_t o = o0 + threadIdx.x;
    const int32_t in = (o < o_end) ? nbr[o * kvol + k] : -1;
    const bool has = in >= 0;
    const uint64_t bal = __ballot(has);
    const uint32_t below = (uint32_t)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wtot[w];
    const uint32_t pos = base_s + wbase + below;
    if (has && pos < cap) {
      pairs[((int64_t)k * 2 + 0) * cap + pos] = in;
      pairs[((int64_t)k * 2 + 1) * cap + pos] = (int32_t)o;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  }
}

static uint64_t pow2_at_least(uint64_t v) {
  uint64_t p = 1024;
  while (p < v) p <<= 1;
  return p;
}

static int bit_width64(uint64_t v) {
  int b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_rulebook_workspace_bytes(int64_t m_in, int32_t kvol) {
  const int64_t m = m_in > 0 ? m_in : 1;
  const int64_t in_cap = (int64_t)pow2_at_least((uint64_t)m * 2);
  const int64_t cand = m * kvol;
  const int64_t set_cap = (int64_t)pow2_at_least((uint64_t)cand * 2);
  // input table + candidate set/out table + candidate list (x2 for sort) + sort scratch
  return fsf_align_up(in_cap * 12, 256) + fsf_align_up(set_cap * 12, 256) + radix_sort_scratch_bytes(cand) + 8 * 256;
}

static int make_geom(ConvGeom* g, int32_t batch_size, const int32_t shape[3], const int32_t ksize[3],
                     const int32_t stride[3], const int32_t padding[3], const int32_t dilation[3]) {
  g->B = batch_size;
  g->Z = shape[0]; g->Y = shape[1]; g->X = shape[2];
  g->kz = ksize[0]; g->ky = ksize[1]; g->kx = ksize[2];
  g->sz = stride[0]; g->sy = stride[1]; g->sx = stride[2];
  g->pz = padding[0]; g->py = padding[1]; g->px = padding[2];
  g->dz = dilation[0]; g->dy = dilation[1]; g->dx = dilation[2];
  g->kvol = g->kz * g->ky * g->kx;
  if (g->B < 1 || g->Z < 1 || g->Y < 1 || g->X < 1 || g->kvol < 1 || g->sz < 1 || g->sy < 1 || g->sx < 1 ||
      g->dz < 1 || g->dy < 1 || g->dx < 1 || g->pz < 0 || g->py < 0 || g->px < 0)
    return FSF_ERR_INVALID_ARG;
  g->OZ = (g->Z + 2 * g->pz - g->dz * (g->kz - 1) - 1) / g->sz + 1;
  g->OY = (g->Y + 2 * g->py - g->dy * (g->ky - 1) - 1) / g->sy + 1;
  g->OX = (g->X + 2 * g->px - g->dx * (g->kx - 1) - 1) / g->sx + 1;
  if (g->OZ < 1 || g->OY < 1 || g->OX < 1) return FSF_ERR_INVALID_ARG;
  return FSF_OK;
}

extern "C" int fsf_rulebook_subm(const int32_t* indices, int64_t m, int32_t batch_size, const int32_t spatial_shape[3],
                                 const int32_t ksize[3], const int32_t dilation[3], int32_t* nbr, void* workspace,
                                 int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || !spatial_shape || !ksize || !dilation || (m > 0 && (!indices || !nbr))) return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  const int32_t one[3] = {1, 1, 1};
  // SubM: stride 1, padding = dil*(k-1)/2 (odd kernels) so that out coords == in coords
  int32_t pad[3];
  for (int j = 0; j < 3; ++j) {
    if (ksize[j] % 2 == 0) return FSF_ERR_UNSUPPORTED;
    pad[j] = dilation[j] * (ksize[j] - 1) / 2;
  }
  ConvGeom g;
  int rc = make_geom(&g, batch_size, spatial_shape, ksize, one, pad, dilation);
  if (rc != FSF_OK) return rc;
  if (workspace_bytes < fsf_rulebook_workspace_bytes(m, g.kvol)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  const int64_t cap = (int64_t)pow2_at_least((uint64_t)m * 2);
  uint64_t* keys = ar.take<uint64_t>(cap);
  int32_t* vals = ar.take<int32_t>(cap);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  hipLaunchKernelGGL(hash_fill_empty_kernel, dim3(fsf_stream_grid(cap, 256)), dim3(256), 0, stream, keys, cap);
  hipLaunchKernelGGL(rb_insert_inputs_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, indices, m, g, keys,
                     vals, (uint64_t)(cap - 1));
  hipLaunchKernelGGL(rb_subm_kernel, dim3(fsf_stream_grid(m * g.kvol, 256)), dim3(256), 0, stream, indices, m, g, keys,
                     vals, (uint64_t)(cap - 1), nbr);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_rulebook_strided(const int32_t* indices, int64_t m, int32_t batch_size,
                                    const int32_t spatial_shape[3], const int32_t ksize[3], const int32_t stride[3],
                                    const int32_t padding[3], const int32_t dilation[3], int32_t* out_indices,
                                    int64_t cap_out, int32_t* nbr, int32_t* nbr_inv, int64_t* m_out_dev,
                                    int64_t* m_out_host, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || !spatial_shape || !ksize || !stride || !padding || !dilation || !m_out_host ||
      (m > 0 && (!indices || !out_indices || !nbr)))
    return FSF_ERR_INVALID_ARG;
  ConvGeom g;
  int rc = make_geom(&g, batch_size, spatial_shape, ksize, stride, padding, dilation);
  if (rc != FSF_OK) return rc;
  if (m == 0) {
    *m_out_host = 0;
    if (m_out_dev) FSF_HIP_TRY(hipMemsetAsync(m_out_dev, 0, sizeof(int64_t), stream));
    return FSF_OK;
  }
  if (workspace_bytes < fsf_rulebook_workspace_bytes(m, g.kvol)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  const int64_t in_cap = (int64_t)pow2_at_least((uint64_t)m * 2);
  const int64_t cand = m * g.kvol;
  // distinct output sites one input can propose: per dim, the offsets k with (in + pad - k*dil) % stride == 0
  auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
  const int pz_ = g.sz / gcd(g.sz, g.dz), py_ = g.sy / gcd(g.sy, g.dy), px_ = g.sx / gcd(g.sx, g.dx);
  const int64_t per_in = (int64_t)((g.kz + pz_ - 1) / pz_) * ((g.ky + py_ - 1) / py_) * ((g.kx + px_ - 1) / px_);
  const int64_t set_cap = (int64_t)pow2_at_least((uint64_t)(m * per_in) * 2);
  // (the two key tables back to back — capacities are powers of two >= 2, so both stay 256-byte aligned: ONE fill launch)
  uint64_t* in_keys = ar.take<uint64_t>(in_cap + set_cap);
  uint64_t* set_keys = in_keys + in_cap;
  int32_t* in_vals = ar.take<int32_t>(in_cap);
  int32_t* set_vals = ar.take<int32_t>(set_cap);
  uint64_t* list_a = ar.take<uint64_t>(cand);
  uint64_t* list_b = ar.take<uint64_t>(cand);
  uint32_t* lv_a = ar.take<uint32_t>(cand);
  uint32_t* lv_b = ar.take<uint32_t>(cand);
  uint32_t* hist = ar.take<uint32_t>((radix_num_tiles(cand) + 1) * RS_BINS);
  uint32_t* count_dev = ar.take<uint32_t>(1);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;

  hipLaunchKernelGGL(hash_fill_empty_kernel, dim3(fsf_stream_grid(in_cap + set_cap, 256)), dim3(256), 0, stream, in_keys,
                     in_cap + set_cap, count_dev);
  hipLaunchKernelGGL(rb_insert_inputs_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, indices, m, g, in_keys,
                     in_vals, (uint64_t)(in_cap - 1));
  const bool propose_by_pair = false;  // (the round-1 kernel: only where the per-workgroup list of the rows kernel does not fit)
  // (the per-workgroup list of the rows kernel is 256 * per_in * 8 bytes of dynamic LDS + 8 static: 64 KB without an attribute)
  if (propose_by_pair || 256 * per_in * 8 + 8 > 64 * 1024 || (m + 255) / 256 > 0x7FFFFFFF)
    hipLaunchKernelGGL(rb_propose_kernel, dim3(fsf_stream_grid(cand, 256)), dim3(256), 0, stream, indices, m, g, set_keys,
                       (uint64_t)(set_cap - 1), list_a, count_dev);
  else
    hipLaunchKernelGGL(rb_propose_rows_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), (size_t)256 * per_in * 8, stream, indices, m,
                       g, set_keys, (uint64_t)(set_cap - 1), list_a, count_dev);
  FSF_LAUNCH_CHECK();
  uint32_t count_h = 0;
  FSF_READ_BACK(&count_h, count_dev, sizeof(uint32_t), stream);
  const int64_t m_out = (int64_t)count_h;
  *m_out_host = m_out;
  if (m_out > cap_out) return FSF_ERR_CAPACITY;
  if (m_out_dev) {
    int64_t tmp = m_out;
    FSF_HIP_TRY(hipMemcpyAsync(m_out_dev, &tmp, sizeof(int64_t), hipMemcpyHostToDevice, stream));
    FSF_STREAM_WAIT(stream);
  }
  if (m_out == 0) return FSF_OK;
  // ascending linear (b,z,y,x) order of the output sites
  hipLaunchKernelGGL(rb_iota_kernel, dim3(fsf_stream_grid(m_out, 256)), dim3(256), 0, stream, lv_a, m_out);
  const uint64_t total_cells = (uint64_t)g.B * g.OZ * g.OY * g.OX;
  uint64_t* sorted;
  uint32_t* sorted_v;
  rc = radix_sort_pairs(list_a, lv_a, list_b, lv_b, hist, m_out, bit_width64(total_cells - 1), &sorted, &sorted_v, stream);
  if (rc != FSF_OK) return rc;
  // the candidate set's key array already holds exactly the output sites: reuse it as the out-site map
  hipLaunchKernelGGL(rb_assign_out_kernel, dim3(fsf_stream_grid(m_out, 256)), dim3(256), 0, stream, sorted, m_out, g,
                     out_indices, set_keys, set_vals, (uint64_t)(set_cap - 1));
  hipLaunchKernelGGL(rb_strided_nbr_kernel, dim3(fsf_stream_grid(m_out * g.kvol, 256)), dim3(256), 0, stream, out_indices,
                     m_out, g, in_keys, in_vals, (uint64_t)(in_cap - 1), nbr);
  if (nbr_inv)
    hipLaunchKernelGGL(rb_strided_inv_kernel, dim3(fsf_stream_grid(cand, 256)), dim3(256), 0, stream, indices, m, g,
                       set_keys, set_vals, (uint64_t)(set_cap - 1), nbr_inv);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// neighbour-mask row order (include/fsf_hip.h: fsf_order_by_neighbor_mask)
namespace fsf {
// half a wave per row: lane k < 27 of the half probes offset k, the half's ballot IS the row's neighbour mask
__global__ void __launch_bounds__(256)
    mo_keys_kernel(const int32_t* __restrict__ indices, int64_t m, ConvGeom g, const uint64_t* __restrict__ hkeys,
                   const int32_t* __restrict__ hvals, uint64_t hmask, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int lane = threadIdx.x & 63, half = lane >> 5, k = lane & 31;
  const int64_t halves = ((int64_t)gridDim.x * blockDim.x) >> 5;
  const int64_t rounds = (m + halves - 1) / halves;  // (wave-uniform trip count: the ballot needs both halves inside the loop)
  int64_t r = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5);
  for (int64_t it = 0; it < rounds; ++it, r += halves) {
    bool hit = false;
    if (r < m && k < 27) {
      const int4 c = *reinterpret_cast<const int4*>(indices + r * 4);
      const int z = c.y + k / 9 - 1, y = c.z + (k / 3) % 3 - 1, x = c.w + k % 3 - 1;
      hit = z >= 0 && z < g.Z && y >= 0 && y < g.Y && x >= 0 && x < g.X && hash_lookup(hkeys, hvals, hmask, lin_in(g, c.x, z, y, x)) >= 0;
    }
    const unsigned long long bal = __ballot(hit);
    if (r < m && k == 0) {
      const uint32_t mask = (uint32_t)(bal >> (32 * half)) & 0x7ffffffu;
      // 16-bit key (two radix passes instead of four): the 9 in-plane neighbours exactly, the planes below / above by their counts.
      // On the 10-sweep frame it groups as well as the full 27-bit mask (0.4 m level: 27.9 k (block, offset) steps against 28.0 k;
      // 0.2 m level 13.2 k against 11.6 k; lexicographic order 39.1 k / 24.3 k).  Ascending sort of the complement: full rows first.
      const uint32_t mid = (mask >> 9) & 511u;
      const uint32_t lo = min(__popc(mask & 511u), 7), hi = min(__popc(mask >> 18), 15);
      // ... and the parity of (z, y, x) as the LEAST significant digit (a third pass): a stride-2 inverse convolution reaches a fine
      // row only through the kernel offsets its parity admits, so rows of one neighbourhood key are kept together by parity — the
      // inverse convolutions into the level then walk 17.5 k / 21.9 k steps instead of 33.5 k / 35.4 k (lexicographic 20.2 k / 19.6 k)
      // and the submanifold layers are unchanged (13.3 k / 28.0 k)
      const int4 c0 = *reinterpret_cast<const int4*>(indices + r * 4);
      const uint32_t parity = (uint32_t)(((c0.y & 1) << 2) | ((c0.z & 1) << 1) | (c0.w & 1));
      keys[r] = ((uint64_t)(0xffffu & ~((mid << 7) | (lo << 4) | hi)) << 3) | parity;
      vals[r] = (uint32_t)r;
    }
  }
}
__global__ void __launch_bounds__(256) mo_finish_kernel(const uint32_t* __restrict__ order, int64_t m, int32_t* __restrict__ perm,
                                                        int32_t* __restrict__ inv_perm) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = order[i];
    perm[i] = (int32_t)r;
    inv_perm[r] = (int32_t)i;
  }
}
__global__ void __launch_bounds__(256) remap_indices_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ map,
                                                            int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = in[i];
    out[i] = v >= 0 ? map[v] : -1;
  }
}
}  // namespace fsf

extern "C" int64_t fsf_order_by_neighbor_mask_workspace_bytes(int64_t m) {
  if (m < 0) return 0;
  const int64_t n = m > 0 ? m : 1;
  const int64_t cap = (int64_t)pow2_at_least((uint64_t)n * 2);
  return fsf_align_up(cap * 8, 256) + fsf_align_up(cap * 4, 256) + 2 * fsf_align_up(n * 8, 256) + 2 * fsf_align_up(n * 4, 256) +
         fsf_align_up((radix_num_tiles(n) + 1) * RS_BINS * 4, 256) + 256;
}

extern "C" int fsf_order_by_neighbor_mask(const int32_t* indices, int64_t m, int32_t batch_size, const int32_t spatial_shape[3],
                                          int32_t* perm, int32_t* inv_perm, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 0 || !spatial_shape || (m > 0 && (!indices || !perm || !inv_perm))) return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  const int32_t three[3] = {3, 3, 3}, one[3] = {1, 1, 1};
  ConvGeom g;
  int rc = make_geom(&g, batch_size, spatial_shape, three, one, one, one);
  if (rc != FSF_OK) return rc;
  if (!workspace || workspace_bytes < fsf_order_by_neighbor_mask_workspace_bytes(m)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  const int64_t cap = (int64_t)pow2_at_least((uint64_t)m * 2);
  uint64_t* hkeys = ar.take<uint64_t>(cap);
  int32_t* hvals = ar.take<int32_t>(cap);
  uint64_t* keys_a = ar.take<uint64_t>(m);
  uint64_t* keys_b = ar.take<uint64_t>(m);
  uint32_t* vals_a = ar.take<uint32_t>(m);
  uint32_t* vals_b = ar.take<uint32_t>(m);
  uint32_t* hist = ar.take<uint32_t>((radix_num_tiles(m) + 1) * RS_BINS);
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  hipLaunchKernelGGL(hash_fill_empty_kernel, dim3(fsf_stream_grid(cap, 256)), dim3(256), 0, stream, hkeys, cap);
  hipLaunchKernelGGL(rb_insert_inputs_kernel, dim3(fsf_stream_grid(m, 256)), dim3(256), 0, stream, indices, m, g, hkeys, hvals,
                     (uint64_t)(cap - 1));
  hipLaunchKernelGGL(mo_keys_kernel, dim3((unsigned)fsf_stream_grid(m * 32, 256)), dim3(256), 0, stream, indices, m, g, hkeys, hvals,
                     (uint64_t)(cap - 1), keys_a, vals_a);
  uint64_t* keys = nullptr;
  uint32_t* order = nullptr;
  rc = radix_sort_pairs(keys_a, vals_a, keys_b, vals_b, hist, m, 19, &keys, &order, stream);
  if (rc != FSF_OK) return rc;
  hipLaunchKernelGGL(mo_finish_kernel, dim3((unsigned)fsf_stream_grid(m, 256)), dim3(256), 0, stream, order, m, perm, inv_perm);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_remap_indices(const int32_t* in, int64_t n, const int32_t* map, int32_t* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || (n > 0 && (!in || !map || !out))) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  hipLaunchKernelGGL(remap_indices_kernel, dim3((unsigned)fsf_stream_grid(n, 256)), dim3(256), 0, stream, in, n, map, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_rulebook_to_pairs_workspace_bytes(int64_t m_out, int32_t kvol) {
  const int64_t nseg = (m_out > 0 ? m_out + RBP_SEG - 1 : RBP_SEG) / RBP_SEG;
  return fsf_align_up(nseg * (kvol > 0 ? kvol : 1) * 4, 256) + 256;
}

extern "C" int fsf_rulebook_to_pairs(const int32_t* nbr, int64_t m_out, int32_t kvol, int32_t* indice_pairs, int64_t cap,
                                     int32_t* indice_num, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m_out < 0 || kvol < 1 || cap < 0 || !indice_num || (m_out > 0 && (!nbr || !indice_pairs))) return FSF_ERR_INVALID_ARG;
  if (m_out == 0) {
    FSF_HIP_TRY(hipMemsetAsync(indice_num, 0, sizeof(int32_t) * kvol, stream));
    return FSF_OK;
  }
  if (!workspace || workspace_bytes < fsf_rulebook_to_pairs_workspace_bytes(m_out, kvol)) return FSF_ERR_WORKSPACE;
  const int64_t nseg = (m_out + RBP_SEG - 1) / RBP_SEG;
  if (nseg > 65535 * 32) return FSF_ERR_UNSUPPORTED;
  int32_t* cnt = (int32_t*)workspace;
  hipLaunchKernelGGL(rb_pairs_count_kernel, dim3((unsigned)nseg), dim3(256), sizeof(int32_t) * kvol, stream, nbr, m_out, (int)kvol,
                     cnt);
  hipLaunchKernelGGL(rb_pairs_fill_kernel, dim3((unsigned)nseg, (unsigned)kvol), dim3(256), 0, stream, nbr, m_out, (int)kvol, cnt,
                     indice_pairs, cap, indice_num);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
