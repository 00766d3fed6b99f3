// Device-wide exclusive scan of u32 values produced by a functor, consumed by a functor.
// Three launches: tile reduce -> scan of tile sums (one workgroup) -> tile scan + apply.
//   In : __device__ uint32_t operator()(int64_t i) const
//   Out: __device__ void operator()(int64_t i, uint32_t exclusive_prefix, uint32_t value) const
#pragma once
#include "common.h"

namespace fsf {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;

static inline int64_t scan_num_tiles(int64_t n) { return n > 0 ? (n + SC_TILE - 1) / SC_TILE : 1; }

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

// tile_sums: u32[scan_num_tiles(n)] scratch.
template <class In, class Out>
static inline int exclusive_scan_u32(In in, Out out, int64_t n, uint32_t* tile_sums, uint32_t* total_u32,
                                     int64_t* total_i64, hipStream_t stream) {
  const int64_t tiles = scan_num_tiles(n);
  hipLaunchKernelGGL((scan_reduce_kernel<In>), dim3((unsigned)tiles), dim3(SC_THREADS), 0, stream, in, n, tile_sums);
  hipLaunchKernelGGL((scan_tilesums_kernel<0>), dim3(1), dim3(1024), 0, stream, tile_sums, tiles, total_u32, total_i64);
  hipLaunchKernelGGL((scan_apply_kernel<In, Out>), dim3((unsigned)tiles), dim3(SC_THREADS), 0, stream, in, out, n,
                     tile_sums);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

}  // namespace fsf
