// Device-wide exclusive scan of u32 values produced by a functor, consumed by a functor.
// ONE launch (+ one memset of the tile status words): single-pass chained scan with decoupled look-back — a tile publishes
// its aggregate, walks back over its predecessors' status words (aggregate / inclusive prefix) until it meets a prefix,
// publishes its own prefix and applies.  Tile ids come from an atomic ticket, so a tile only ever waits for tiles that
// started before it (no dispatch-order assumption); status words are 32-bit (2 flag bits + 30-bit value: totals < 2^30)
// read and written with agent-scope atomics, which is what crosses the non-coherent per-XCD L2s.  The sums are integers:
// the result does not depend on timing.  (The previous three-launch form — tile reduce, scan of tile sums, apply — is
// kept below for totals >= 2^30.)
//   In : __device__ uint32_t operator()(int64_t i) const
//   Out: __device__ void operator()(int64_t i, uint32_t exclusive_prefix, uint32_t value) const
#pragma once
#include "common.h"

namespace fsf {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;

static inline int64_t scan_grid_tiles(int64_t n) { return n > 0 ? (n + SC_TILE - 1) / SC_TILE : 1; }
// u32 words of scratch a scan over n items needs (tile status words + the ticket counter)
static inline int64_t scan_num_tiles(int64_t n) { return scan_grid_tiles(n) + 64; }

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total_out) {
  __shared__ uint32_t wtot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = fsf_wave_inclusive_scan(v);
  __syncthreads();  // protect wtot reuse across calls
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wave) base += wtot[w];
  if (total_out) *total_out = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  return base + incl - v;
}

template <class In>
__global__ void __launch_bounds__(SC_THREADS) scan_reduce_kernel(In in, int64_t n, uint32_t* __restrict__ tile_sums) {
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j)
    if (base + j < n) s += in(base + j);
  uint32_t tot;
  block_exclusive_scan_256(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// One workgroup: exclusive scan of tile sums in place; total -> *total_u32 and (optionally) *total_i64.
template <int UNUSED>
__global__ void __launch_bounds__(1024)
    scan_tilesums_kernel(uint32_t* __restrict__ tile_sums, int64_t tiles, uint32_t* total_u32, int64_t* total_i64) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x;
  const int64_t per = (tiles + 1023) / 1024;
  const int64_t lo = tid * per;
  const int64_t hi = (lo + per < tiles) ? lo + per : tiles;
  uint32_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += tile_sums[i];
  uint32_t incl = fsf_wave_inclusive_scan(sum);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0, all = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) wbase += wave_tot[w];
    all += wave_tot[w];
  }
  uint32_t run = wbase + incl - sum;
  for (int64_t i = lo; i < hi; ++i) {
    uint32_t v = tile_sums[i];
    tile_sums[i] = run;
    run += v;
  }
  if (tid == 0) {
    if (total_u32) *total_u32 = all;
    if (total_i64) *total_i64 = (int64_t)all;
  }
}

template <class In, class Out>
__global__ void __launch_bounds__(SC_THREADS)
    scan_apply_kernel(In in, Out out, int64_t n, const uint32_t* __restrict__ tile_sums) {
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    v[j] = (base + j < n) ? in(base + j) : 0u;
    s += v[j];
  }
  uint32_t excl = block_exclusive_scan_256(s, nullptr) + tile_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    if (base + j < n) out(base + j, excl, v[j]);
    excl += v[j];
  }
}

constexpr uint32_t SC_FLAG_AGG = 1u << 30, SC_FLAG_PREFIX = 2u << 30, SC_VALUE_MASK = (1u << 30) - 1u;

__device__ __forceinline__ void sc_publish(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t sc_peek(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exclusive prefix of tile `tile` from the status words of its predecessors (one lane; the word carries flag AND value)
__device__ __forceinline__ uint32_t sc_look_back(const uint32_t* status, int64_t tile, int64_t stride) {
  uint32_t excl = 0;
  for (int64_t t = tile - 1; t >= 0; --t) {
    uint32_t w;
    while (((w = sc_peek(status + t * stride)) >> 30) == 0u) __builtin_amdgcn_s_sleep(1);
    excl += w & SC_VALUE_MASK;
    if (w & SC_FLAG_PREFIX) break;
  }
  return excl;
}

// status: u32[tiles] zeroed, ticket: u32 zeroed
template <class In, class Out>
__global__ void __launch_bounds__(SC_THREADS)
    scan_lookback_kernel(In in, Out out, int64_t n, uint32_t* __restrict__ status, uint32_t* __restrict__ ticket, int64_t tiles,
                         uint32_t* total_u32, int64_t* total_i64) {
  __shared__ uint32_t tile_s, excl_s;
  if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);
  __syncthreads();
  const int64_t tile = tile_s;
  const int64_t base = tile * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    v[j] = (base + j < n) ? in(base + j) : 0u;
    s += v[j];
  }
  uint32_t tot;
  uint32_t excl = block_exclusive_scan_256(s, &tot);
  if (threadIdx.x == 0) {
    if (tile > 0) sc_publish(status + tile, tot | SC_FLAG_AGG);
    const uint32_t before = tile > 0 ? sc_look_back(status, tile, 1) : 0u;
    sc_publish(status + tile, (before + tot) | SC_FLAG_PREFIX);
    excl_s = before;
    if (tile == tiles - 1) {
      if (total_u32) *total_u32 = before + tot;
      if (total_i64) *total_i64 = (int64_t)(before + tot);
    }
  }
  __syncthreads();
  excl += excl_s;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    if (base + j < n) out(base + j, excl, v[j]);
    excl += v[j];
  }
}

// tile_sums: u32[scan_num_tiles(n)] scratch.
// max_item: an upper bound of every scanned value (1 for flag scans).  The one-launch look-back form carries running totals in 30
// bits next to the 2 flag bits of a status word, so it is taken only when n * max_item cannot reach 2^30; beyond that the
// three-launch form (32-bit sums) runs.
template <class In, class Out>
static inline int exclusive_scan_u32(In in, Out out, int64_t n, uint32_t* tile_sums, uint32_t* total_u32,
                                     int64_t* total_i64, hipStream_t stream, int64_t max_item = 1, bool tile_sums_zeroed = false) {
  if (max_item < 1) max_item = 1;
  if (n < (int64_t)SC_VALUE_MASK / max_item) {
    const int64_t tiles = scan_grid_tiles(n);
    if (!tile_sums_zeroed && hipMemsetAsync(tile_sums, 0, (size_t)(tiles + 1) * 4, stream) != hipSuccess) return FSF_ERR_HIP;
    hipLaunchKernelGGL((scan_lookback_kernel<In, Out>), dim3((unsigned)tiles), dim3(SC_THREADS), 0, stream, in, out, n, tile_sums,
                       tile_sums + tiles, tiles, total_u32, total_i64);
    FSF_LAUNCH_CHECK();
    return FSF_OK;
  }
  const int64_t tiles = scan_grid_tiles(n);
  hipLaunchKernelGGL((scan_reduce_kernel<In>), dim3((unsigned)tiles), dim3(SC_THREADS), 0, stream, in, n, tile_sums);
  hipLaunchKernelGGL((scan_tilesums_kernel<0>), dim3(1), dim3(1024), 0, stream, tile_sums, tiles, total_u32, total_i64);
  hipLaunchKernelGGL((scan_apply_kernel<In, Out>), dim3((unsigned)tiles), dim3(SC_THREADS), 0, stream, in, out, n,
                     tile_sums);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

}  // namespace fsf
