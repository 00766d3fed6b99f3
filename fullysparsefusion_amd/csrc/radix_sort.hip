// Stable LSD radix sort for wave64 (see radix_sort.h).  HBM traffic per pass: 12 B read (histogram: 8 B)
// + 12 B written per pair; with <= 32 significant key bits that is <= 4 passes.
#include "radix_sort.h"

namespace fsf {

__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                            uint32_t* __restrict__ hist, int tiles) {
  __shared__ uint32_t lh[RS_BINS];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int64_t e = base + it * RS_THREADS + threadIdx.x;
    if (e < n) atomicAdd(&lh[(uint32_t)(keys[e] >> shift) & 0xffu], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * tiles + blockIdx.x] = lh[threadIdx.x];
}

// Digit-major histogram hist[digit][tile] -> exclusive offsets, in two small launches:
//  (1) one workgroup per digit scans its row over the tiles (in place) and records the digit total,
//  (2) one wave scans the 256 digit totals; the scatter kernel adds digit_base[d] + hist[d][tile].
__global__ void __launch_bounds__(256) rs_scan_rows_kernel(uint32_t* __restrict__ hist, int tiles, uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t carry_s;
  uint32_t* row = hist + (int64_t)blockIdx.x * tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const uint32_t v = (t < tiles) ? row[t] : 0u;
    const uint32_t incl = fsf_wave_inclusive_scan(v);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t base = carry_s;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    if (t < tiles) row[t] = base + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) digit_total[blockIdx.x] = carry_s;
}

__global__ void __launch_bounds__(256) rs_scan_digits_kernel(uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wtot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t v = digit_total[threadIdx.x];
  const uint32_t incl = fsf_wave_inclusive_scan(v);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  digit_total[threadIdx.x] = base + incl - v;
}

__global__ void __launch_bounds__(RS_THREADS)
    rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int shift,
                      const uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_base, int tiles) {
  // slot = it*4 + wave enumerates the tile's 64-key groups in key order
  __shared__ uint32_t cnt[RS_ITEMS * 4][RS_BINS];
  __shared__ uint32_t gbase[RS_BINS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;

  uint64_t key[RS_ITEMS];
  uint32_t val[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int64_t e = base + it * RS_THREADS + tid;
    bool valid = e < n;
    key[it] = valid ? keys_in[e] : 0ull;
    val[it] = valid ? vals_in[e] : 0u;
  }
#pragma unroll
  for (int s = 0; s < RS_ITEMS * 4; ++s) cnt[s][tid] = 0;
  __syncthreads();

  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int64_t e = base + it * RS_THREADS + tid;
    bool valid = e < n;
    uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      bool bit = (d >> b) & 1u;
      uint64_t bm = __ballot(valid && bit);
      peers &= bit ? bm : ~bm;
    }
    uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    rank[it] = r;
    if (valid && r == 0) cnt[it * 4 + wave][d] = (uint32_t)__popcll(peers);
  }
  __syncthreads();
  {
    uint32_t run = 0;
#pragma unroll
    for (int s = 0; s < RS_ITEMS * 4; ++s) {
      uint32_t c = cnt[s][tid];
      cnt[s][tid] = run;
      run += c;
    }
    gbase[tid] = digit_base[tid] + hist[(int64_t)tid * tiles + blockIdx.x];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int64_t e = base + it * RS_THREADS + tid;
    if (e < n) {
      uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
      uint32_t pos = gbase[d] + cnt[it * 4 + wave][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-sweep form (n < 2^30): ONE launch per 8-bit pass instead of four.  The digit counts of every pass are global
// properties of the key multiset (a permutation does not change them), so one upfront kernel histograms all passes; a pass
// then needs, per tile, only the number of keys of each digit in the tiles before it — obtained by decoupled look-back over
// per-(tile, digit) status words (thread d walks digit d's chain), tile ids from an atomic ticket (a tile only waits for
// tiles that started earlier), status words written / read with agent-scope atomics (flag and count in one 32-bit word).
// Same stable order as the four-launch form (integer prefix sums: no dependence on timing).
constexpr uint32_t RS_FLAG_AGG = 1u << 30, RS_FLAG_PREFIX = 2u << 30, RS_VALUE_MASK = (1u << 30) - 1u;
constexpr int RS_MAX_PASSES = 8;

__global__ void __launch_bounds__(RS_THREADS)
    rs_hist_all_kernel(const uint64_t* __restrict__ keys, int64_t n, int passes, uint32_t* __restrict__ ghist) {
  __shared__ uint32_t lh[RS_MAX_PASSES][RS_BINS];
  for (int p = 0; p < passes; ++p) lh[p][threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t e = base + it * RS_THREADS + threadIdx.x;
    if (e < n) {
      const uint64_t k = keys[e];
      for (int p = 0; p < passes; ++p) atomicAdd(&lh[p][(uint32_t)(k >> (8 * p)) & 0xffu], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = lh[p][threadIdx.x];
    if (c) atomicAdd(&ghist[p * RS_BINS + threadIdx.x], c);
  }
}

__global__ void __launch_bounds__(RS_THREADS)
    rs_onesweep_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
                       uint32_t* __restrict__ vals_out, int64_t n, int shift, const uint32_t* __restrict__ ghist,
                       uint32_t* __restrict__ status, uint32_t* __restrict__ ticket) {
  __shared__ uint32_t cnt[RS_ITEMS * 4][RS_BINS];
  __shared__ uint32_t gbase[RS_BINS];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) tile_s = atomicAdd(ticket, 1u);
  __syncthreads();
  const int64_t tile = tile_s;
  const int64_t base = tile * RS_TILE;

  uint64_t key[RS_ITEMS];
  uint32_t val[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t e = base + it * RS_THREADS + tid;
    const bool valid = e < n;
    key[it] = valid ? keys_in[e] : 0ull;
    val[it] = valid ? vals_in[e] : 0u;
  }
#pragma unroll
  for (int s = 0; s < RS_ITEMS * 4; ++s) cnt[s][tid] = 0;
  __syncthreads();
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t e = base + it * RS_THREADS + tid;
    const bool valid = e < n;
    const uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bm = __ballot(valid && bit);
      peers &= bit ? bm : ~bm;
    }
    const uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    rank[it] = r;
    if (valid && r == 0) cnt[it * 4 + wave][d] = (uint32_t)__popcll(peers);
  }
  __syncthreads();
  uint32_t run = 0;  // thread tid owns digit tid: the tile's keys of that digit, slot by slot
#pragma unroll
  for (int s = 0; s < RS_ITEMS * 4; ++s) {
    const uint32_t c = cnt[s][tid];
    cnt[s][tid] = run;
    run += c;
  }
  uint32_t* st = status + tile * RS_BINS + tid;
  uint32_t before = 0;
  if (tile > 0) {
    __hip_atomic_store(st, run | RS_FLAG_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int64_t t = tile - 1; t >= 0; --t) {
      uint32_t w;
      while (((w = __hip_atomic_load(status + t * RS_BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u)
        __builtin_amdgcn_s_sleep(1);
      before += w & RS_VALUE_MASK;
      if (w & RS_FLAG_PREFIX) break;
    }
  }
  __hip_atomic_store(st, (before + run) | RS_FLAG_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // exclusive scan of the global digit counts (256 values, one per thread)
  const uint32_t g = ghist[tid];
  const uint32_t incl = fsf_wave_inclusive_scan(g);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t dbase = incl - g;
  for (int w = 0; w < wave; ++w) dbase += wtot[w];
  gbase[tid] = dbase + before;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t e = base + it * RS_THREADS + tid;
    if (e < n) {
      const uint32_t d = (uint32_t)(key[it] >> shift) & 0xffu;
      const uint32_t pos = gbase[d] + cnt[it * 4 + wave][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

int64_t radix_sort_scratch_bytes(int64_t n) {
  int64_t nn = n > 0 ? n : 1;
  return fsf_align_up(nn * 8, 256) * 2 + fsf_align_up(nn * 4, 256) * 2 +
         fsf_align_up((radix_num_tiles(n) + 1) * RS_BINS * 4, 256);  // (radix_num_tiles counts the one-sweep status words in)
}

int radix_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t* hist,
                     int64_t n, int key_bits, uint64_t** keys_out, uint32_t** vals_out, hipStream_t stream, bool hist_zeroed) {
  uint64_t* kin = keys_a;
  uint32_t* vin = vals_a;
  uint64_t* kout = keys_b;
  uint32_t* vout = vals_b;
  const int passes_all = (key_bits + 7) / 8;
  if (n > 0 && n < (int64_t)RS_VALUE_MASK && passes_all >= 1 && passes_all <= RS_MAX_PASSES) {
    // one-sweep: [ghist: passes x 256][ticket: passes (padded to 256)][status: passes x tiles x 256], zeroed by ONE memset
    const int64_t tiles = radix_grid_tiles(n);
    uint32_t* ghist = hist;
    uint32_t* ticket = ghist + (int64_t)passes_all * RS_BINS;
    uint32_t* status = ticket + RS_BINS;
    const size_t zero_bytes = ((size_t)passes_all * RS_BINS + RS_BINS + (size_t)passes_all * tiles * RS_BINS) * 4;
    if (!hist_zeroed && hipMemsetAsync(hist, 0, zero_bytes, stream) != hipSuccess) return FSF_ERR_HIP;
    hipLaunchKernelGGL(rs_hist_all_kernel, dim3((unsigned)tiles), dim3(RS_THREADS), 0, stream, kin, n, passes_all, ghist);
    for (int p = 0; p < passes_all; ++p) {
      hipLaunchKernelGGL(rs_onesweep_kernel, dim3((unsigned)tiles), dim3(RS_THREADS), 0, stream, kin, vin, kout, vout, n, p * 8,
                         ghist + p * RS_BINS, status + (int64_t)p * tiles * RS_BINS, ticket + p);
      uint64_t* tk = kin; kin = kout; kout = tk;
      uint32_t* tv = vin; vin = vout; vout = tv;
    }
    FSF_LAUNCH_CHECK();
  } else if (n > 0) {
    const int tiles = (int)radix_grid_tiles(n);
    const int passes = (key_bits + 7) / 8;
    for (int p = 0; p < passes; ++p) {
      const int shift = p * 8;
      hipLaunchKernelGGL(rs_hist_kernel, dim3(tiles), dim3(RS_THREADS), 0, stream, kin, n, shift, hist, tiles);
      uint32_t* digit_total = hist + (int64_t)tiles * RS_BINS;
      hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(RS_BINS), dim3(256), 0, stream, hist, tiles, digit_total);
      hipLaunchKernelGGL(rs_scan_digits_kernel, dim3(1), dim3(256), 0, stream, digit_total);
      hipLaunchKernelGGL(rs_scatter_kernel, dim3(tiles), dim3(RS_THREADS), 0, stream, kin, vin, kout, vout, n, shift,
                         hist, digit_total, tiles);
      uint64_t* tk = kin; kin = kout; kout = tk;
      uint32_t* tv = vin; vin = vout; vout = tv;
    }
    FSF_LAUNCH_CHECK();
  }
  *keys_out = kin;
  *vals_out = vin;
  return FSF_OK;
}

}  // namespace fsf
