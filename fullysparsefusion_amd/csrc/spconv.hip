// K9/K11: sparse convolution forward as an output-stationary implicit GEMM on the fp32 matrix cores.
// See include/fsf_hip.h.
//
// One workgroup (4 waves) owns a tile of TM=64 (or 128, big layers) output rows x TN<=128 output channels whose accumulator
// lives in LDS for the whole kernel-offset loop, so every output row is written exactly once (no
// scatter-add atomics, deterministic).  For each kernel offset k the rows of the tile that actually have
// a neighbour are COMPACTED (ballot prefix, precomputed per tile) and v_mfma_f32_16x16x4_f32 runs over
// ceil(cnt/16) row blocks only — on LiDAR data 20-55 % of (row, offset) pairs exist, so the dense 27-offset
// product would waste most of the matrix-core time.  Each wave owns a TN/4-column slice and keeps its B
// (weight) fragments for the current stage in registers (weights pre-transposed to [k][cout][cin] so a lane's
// four consecutive K values arrive as one 16-byte load).
//
// Stages = (offset k, 64-wide cin chunk).  The compacted input rows of stage s+1 are gathered by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging) into the second half of a double-buffered A tile while the
// matrix cores work on stage s, and the B fragments of s+1 stream into a second register set; ONE barrier per
// stage.  The DMA writes LDS linearly (wave base + lane*16), so the bank-conflict swizzle is applied to the
// per-lane SOURCE address and undone on the ds_read_b128 side (16-byte chunk c of row j lives at c ^ (j & 15)).
// Tiles are mapped to workgroups so that each XCD walks a contiguous range of output rows (neighbouring tiles
// re-gather the same input rows -> L2 hits).  Layers with too few tiles to fill 256 CUs split the offset loop
// over gridDim.z and a second kernel folds the partial tiles in fixed order.  Eval-mode BN affine + residual +
// ReLU are the fused epilogue.
//
// Roofline: flops = 2*P*Cin*Cout on the fp32 MFMA (157.3 TF/s peak),
// bytes = P*(Cin+Cout)*4 + kvol*Cin*Cout*4 + 8P.
#include <stdlib.h>

#include "common.h"

namespace fsf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SC_TM = 64;
constexpr int SC_KC = 64;                 // cin chunk per stage (floats)
constexpr int SC_NSTEPS = SC_KC / 16;     // 16-wide K steps per stage
constexpr int SC_AROW = SC_KC;            // A tile row stride in floats (linear: the DMA cannot pad)
constexpr int SC_ABUF = SC_TM * SC_AROW;  // floats per A buffer
constexpr int SC_MAXK = 27;
constexpr int SC_NXCD = 8;

struct SpconvArgs {
  const float* feat;
  const float* wt;  // [kvol][cout][cin]
  const int32_t* nbr;
  const float* scale;
  const float* shift;
  const float* residual;
  float* out;
  float* partial;  // [ksplit][m_out][cout] when ksplit > 1
  int64_t m_in, m_out;
  int cin, cout, kvol, relu, ksplit;
  int* queue;  // [SC_NXCD] work counters of the persistent fast path, then [ntiles * cout_blocks] arrival counters of
               // the offset splits (all zeroed before the launch)
  int ntiles, cout_blocks;
};

__device__ __forceinline__ float epilogue_one(const SpconvArgs& a, float x, int64_t o, int col) {
  if (a.scale) x = __fmaf_rn(x, a.scale[col], a.shift[col]);
  else if (a.shift) x = __fadd_rn(x, a.shift[col]);
  if (a.residual) x = __fadd_rn(x, a.residual[o * a.cout + col]);
  if (a.relu) x = fmaxf(x, 0.0f);
  return x;
}

// LDS carve shared by both kernels (TM = output rows per workgroup)
template <int TN, int TM = SC_TM>
struct SpconvSmem {
  static constexpr int CS_STRIDE = TN + 4;
  static constexpr int CS_FLOATS = (TM + 1) * CS_STRIDE;  // + dump row
  static constexpr int A_FLOATS = 2 * TM * SC_AROW > TM * (SC_KC + 4) ? 2 * TM * SC_AROW : TM * (SC_KC + 4);
  static constexpr size_t bytes() {
    return (size_t)(CS_FLOATS + A_FLOATS) * 4 + (size_t)SC_MAXK * TM * 4 + (size_t)SC_MAXK * TM + (size_t)SC_MAXK * 8 + 64 + 16;
  }
};

// XCD-aware tile mapping (bijective): workgroup b runs on XCD b % 8; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
  const int q = ntiles / SC_NXCD, r = ntiles % SC_NXCD, xcd = b % SC_NXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / SC_NXCD;
}

// Work queue of the persistent fast path.  The tiles are cut into SC_NXCD contiguous ranges, one per XCD (neighbouring
// tiles re-gather the same input rows -> hits in that XCD's L2); inside a range the items run tile-major, i.e. the cout
// blocks and offset splits of one tile are handed out back to back and gather the same rows at about the same time.
// A workgroup asks its own XCD's counter first and, once that range is drained, the other ranges in ring order — the
// ranges differ in work (ground-plane tiles have 3x the pairs of the tiles above them), a static split leaves whole
// XCDs idle for the last ~20 % of the kernel.
__device__ __forceinline__ void queue_range(const SpconvArgs& a, int x, int& start, int& len) {
  const int q = a.ntiles / SC_NXCD, r = a.ntiles % SC_NXCD;
  start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  len = x < r ? q + 1 : q;
}

__device__ __forceinline__ void fetch_item(const SpconvArgs& a, int xcd, unsigned& drained, int32_t* item) {
  const int per_tile = a.cout_blocks * a.ksplit;
  for (int v = 0; v < SC_NXCD; ++v) {
    const int x = (xcd + v) & (SC_NXCD - 1);
    if (drained & (1u << x)) continue;
    int start, len;
    queue_range(a, x, start, len);
    const int i = __hip_atomic_fetch_add(a.queue + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i < len * per_tile) {
      const int rem = i % per_tile;
      item[0] = start + i / per_tile;
      item[1] = rem / a.ksplit;
      item[2] = rem % a.ksplit;
      return;
    }
    drained |= 1u << x;
  }
  item[0] = -1;
}

// per-tile compaction lists for every offset + the active-offset list of this z-split.
// The tile's [64][kvol] block of the neighbour table is contiguous in HBM: it is staged through LDS with
// coalesced loads (`stage`, >= 64*kvol ints), then each wave compacts offsets k = wave, wave+4, ...
// rl_in holds ELEMENT offsets (input row * cin) so the gather needs no 64-bit multiply per row.
template <int TM, int NT = 256>
__device__ __forceinline__ void build_row_lists(const SpconvArgs& a, int64_t o0, int zsplit, int32_t* stage, int32_t* rl_in,
                                                uint8_t* rl_loc, int32_t* rl_cnt, int32_t* act_k, int32_t* act_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t rows_left = a.m_out - o0;
  const int nvalid = (int)((rows_left < TM ? rows_left : TM) * a.kvol);
  const int32_t* src = a.nbr + o0 * a.kvol;
  for (int t = tid; t < TM * a.kvol; t += NT) stage[t] = (t < nvalid) ? src[t] : -1;
  __syncthreads();
  for (int k = wave; k < a.kvol; k += NT / 64) {
    int base = 0;
#pragma unroll
    for (int h = 0; h < TM / 64; ++h) {  // 64 tile rows per ballot, appended in row order
      const int32_t in = stage[(h * 64 + lane) * a.kvol + k];
      const bool has = in >= 0;
      const uint64_t bal = __ballot(has);
      if (has) {
        const int pos = base + (int)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        rl_in[k * TM + pos] = in * a.cin;
        rl_loc[k * TM + pos] = (uint8_t)(h * 64 + lane);
      }
      base += (int)__popcll(bal);
    }
    if (lane == 0) rl_cnt[k] = base;
  }
  __syncthreads();
  if (wave == 0) {  // active offsets of this z-split, in ascending k (ballot prefix instead of a serial loop)
    const bool on = lane < a.kvol && rl_cnt[lane < a.kvol ? lane : 0] > 0;
    const uint64_t bal = __ballot(on);
    const int rank = (int)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    const bool mine = on && (rank % a.ksplit == zsplit);
    const uint64_t bal2 = __ballot(mine);
    if (mine) act_k[(int)__popcll(bal2 & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = lane;
    if (lane == 0) act_n[0] = (int)__popcll(bal2);
  }
  __syncthreads();
}

// 16-byte write-through (sc1) store: the partial tiles of an offset split are read by a workgroup that may sit on
// another XCD, whose L2 never sees this one's dirty lines.  Written through, they need no release fence (a
// buffer_wbl2 per work item costs more than the separate fold launch it replaces).  The compiler does not count
// this store: the publisher drains with an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void store_f4_write_through(float* p, float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

template <int TN, int TM = SC_TM, bool WT = false, int NT = 256>
__device__ __forceinline__ void write_tile(const SpconvArgs& a, const float* Cs, int64_t o0, int n0, int zsplit) {
  constexpr int CS_STRIDE = TN + 4;
  constexpr int F4_PER_ROW = TN / 4;
  constexpr int ROWS_PER_PASS = NT / F4_PER_ROW;
  // a thread keeps the same 4 columns for all of its rows: the BN affine is loaded once
  const int c4 = (threadIdx.x % F4_PER_ROW) * 4;
  const int r0 = threadIdx.x / F4_PER_ROW;
  const int col = n0 + c4;
  if (col >= a.cout) return;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool fin = a.ksplit == 1;
  if (fin && a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
  if (fin && a.shift) sh = *reinterpret_cast<const float4*>(a.shift + col);
#pragma unroll 4
  for (int r = r0; r < TM; r += ROWS_PER_PASS) {
    const int64_t o = o0 + r;
    if (o >= a.m_out) break;
    float4 v = *reinterpret_cast<const float4*>(Cs + r * CS_STRIDE + c4);
    if (!fin) {
      float* pp = a.partial + ((int64_t)zsplit * a.m_out + o) * a.cout + col;
      if constexpr (WT) store_f4_write_through(pp, v);
      else *reinterpret_cast<float4*>(pp) = v;
      continue;
    }
    if (a.scale) {
      v.x = __fmaf_rn(v.x, sc.x, sh.x); v.y = __fmaf_rn(v.y, sc.y, sh.y);
      v.z = __fmaf_rn(v.z, sc.z, sh.z); v.w = __fmaf_rn(v.w, sc.w, sh.w);
    } else if (a.shift) {
      v.x = __fadd_rn(v.x, sh.x); v.y = __fadd_rn(v.y, sh.y); v.z = __fadd_rn(v.z, sh.z); v.w = __fadd_rn(v.w, sh.w);
    }
    if (a.residual) {
      const float4 rs = *reinterpret_cast<const float4*>(a.residual + o * a.cout + col);
      v.x = __fadd_rn(v.x, rs.x); v.y = __fadd_rn(v.y, rs.y); v.z = __fadd_rn(v.z, rs.z); v.w = __fadd_rn(v.w, rs.w);
    }
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(a.out + o * a.cout + col) = v;
  }
}

// Offset-split layers, persistent path: the workgroup that finishes a (tile, cout block) LAST folds the partial tiles
// of the other splits into the result — in split order, its own contribution taken from LDS, so the sum is the same
// fixed-order sum the stand-alone fold kernel computes — and applies the epilogue.  Release/acquire at agent scope:
// the partial tiles cross XCDs, whose L2s are not coherent with each other.
template <int TN, int TM, int NT>
__device__ __forceinline__ void fold_tile(const SpconvArgs& a, const float* Cs, int64_t o0, int n0, int zsplit) {
  constexpr int CS_STRIDE = TN + 4;
  constexpr int F4_PER_ROW = TN / 4;
  constexpr int ROWS_PER_PASS = NT / F4_PER_ROW;
  const int c4 = (threadIdx.x % F4_PER_ROW) * 4;
  const int r0 = threadIdx.x / F4_PER_ROW;
  const int col = n0 + c4;
  if (col >= a.cout) return;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
  if (a.shift) sh = *reinterpret_cast<const float4*>(a.shift + col);
  for (int r = r0; r < TM; r += ROWS_PER_PASS) {
    const int64_t o = o0 + r;
    if (o >= a.m_out) break;
    const float4 own = *reinterpret_cast<const float4*>(Cs + r * CS_STRIDE + c4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < a.ksplit; ++z) {
      float4 p = own;
      if (z != zsplit) p = *reinterpret_cast<const float4*>(a.partial + ((int64_t)z * a.m_out + o) * a.cout + col);
      if (z == 0) v = p;
      else { v.x = __fadd_rn(v.x, p.x); v.y = __fadd_rn(v.y, p.y); v.z = __fadd_rn(v.z, p.z); v.w = __fadd_rn(v.w, p.w); }
    }
    if (a.scale) {
      v.x = __fmaf_rn(v.x, sc.x, sh.x); v.y = __fmaf_rn(v.y, sc.y, sh.y);
      v.z = __fmaf_rn(v.z, sc.z, sh.z); v.w = __fmaf_rn(v.w, sc.w, sh.w);
    } else if (a.shift) {
      v.x = __fadd_rn(v.x, sh.x); v.y = __fadd_rn(v.y, sh.y); v.z = __fadd_rn(v.z, sh.z); v.w = __fadd_rn(v.w, sh.w);
    }
    if (a.residual) {
      const float4 rs = *reinterpret_cast<const float4*>(a.residual + o * a.cout + col);
      v.x = __fadd_rn(v.x, rs.x); v.y = __fadd_rn(v.y, rs.y); v.z = __fadd_rn(v.z, rs.z); v.w = __fadd_rn(v.w, rs.w);
    }
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(a.out + o * a.cout + col) = v;
  }
}

#ifdef FSF_ABL_TIMING
__device__ long long fsf_dbg[4096 * 8];
#define FSF_BID ((item[0] * a.cout_blocks + item[1]) * a.ksplit + item[2])
#define FSF_STAMP(i) do { if (threadIdx.x == 0 && FSF_BID < 4096) fsf_dbg[FSF_BID * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FSF_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------
// Fast path: cin % 64 == 0.  LDS-DMA double-buffered A tile, register double-buffered B, one barrier per stage.
// NW waves per workgroup: 4 (each wave owns TN/4 columns) or 8 (TN/8 columns: twice the waves per SIMD to hide the
// per-stage barrier, LDS and load latency behind, half the B and accumulator registers per wave).
template <int TN, int TM, int NW>
__global__ void __launch_bounds__(NW * 64, (TM == 64 ? 2 : 1) * NW / 4) spconv_fwd_dma_kernel(SpconvArgs a) {
  using SM = SpconvSmem<TN, TM>;
  constexpr int NT = NW * 64;
  constexpr int ABUF = TM * SC_AROW;  // floats per A buffer
  constexpr int CS_STRIDE = SM::CS_STRIDE;
  constexpr int WCOLS = TN / NW;   // columns per wave
  constexpr int NCT = WCOLS / 16;  // 16-column MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Cs = reinterpret_cast<float*>(smem);                         // [TM + 1][CS_STRIDE], row TM = dump row
  float* As = Cs + SM::CS_FLOATS;                                     // [2][TM][SC_AROW], swizzled 16-B chunks
  int32_t* rl_in = reinterpret_cast<int32_t*>(As + SM::A_FLOATS);     // [SC_MAXK][TM]
  uint8_t* rl_loc = reinterpret_cast<uint8_t*>(rl_in + SC_MAXK * TM);
  int32_t* rl_cnt = reinterpret_cast<int32_t*>(rl_loc + SC_MAXK * TM);
  int32_t* act_k = rl_cnt + SC_MAXK;
  int32_t* act_n = act_k + SC_MAXK;
  int32_t* item = act_n + 4;  // [3] tile, cout block, offset split of the current work item

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15;  // A row / B column inside a 16x16 tile
  const int kgrp = lane >> 4;  // which 4-float K group this lane feeds
  const int wcol0 = wave * WCOLS;
  const int nchunks = a.cin / SC_KC;
  const int64_t wk_stride = (int64_t)a.cout * a.cin;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & (SC_NXCD - 1));
  unsigned drained = 0;  // (thread 0) ranges known to be empty

  // Persistent workgroup: 2 per CU, each pulls (tile, cout block, offset split) items until the queues are dry.
  for (;;) {
  if (tid == 0) fetch_item(a, xcd, drained, item);
  __syncthreads();
  if (item[0] < 0) break;
  const int64_t o0 = (int64_t)item[0] * TM;
  const int n0 = item[1] * TN;
  const int zsplit = item[2];

  FSF_STAMP(0);
  // the A buffers double as the staging area of the neighbour-table block
  build_row_lists<TM, NT>(a, o0, zsplit, reinterpret_cast<int32_t*>(As), rl_in, rl_loc, rl_cnt, act_k, act_n);
  FSF_STAMP(1);
  const int nstages = act_n[0] * nchunks;

  f32x4 bcur[NCT][SC_NSTEPS], bnext[NCT][SC_NSTEPS];

  // LDS-DMA gather of a stage into A buffer `buf`: one wave instruction moves 4 rows (64 lanes x 16 B); wave w takes
  // row groups w, w+NW, ...; a group is issued iff its first row is live (wave-uniform branch).  The lane's source rows
  // (goff, element offsets) are read from the row list once per OFFSET and reused for all of its cin chunks, and the
  // (offset, count) pair of the next stage is carried in registers: a stage starts with no LDS round trip in front
  // of its DMA and B loads.
  constexpr int NIT = TM / (4 * NW);
  int goff[NIT];
  auto load_goff = [&](int k, int cnt) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int j = 4 * (wave + NW * it) + (lane >> 4);
      j = j < cnt ? j : cnt - 1;  // rows past cnt re-read the last live row (finite filler)
      goff[it] = rl_in[k * TM + (j < 0 ? 0 : j)];
    }
  };
  auto issue_gather = [&](int cnt, int cin0, int buf) {
    float* abuf = As + buf * ABUF;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int g = wave + NW * it;  // row group: rows 4g .. 4g+3
      if (4 * g < cnt) {
        const int j = 4 * g + (lane >> 4);
        const int chunk = (lane & 15) ^ (j & 15);  // logical 16-B chunk that must land at physical slot lane&15 of row j
        const float* src = a.feat + goff[it] + cin0 + 4 * chunk;
#ifndef FSF_ABL_NO_GATHER
        __builtin_amdgcn_global_load_lds(src, abuf + 4 * g * SC_AROW, 16, 0, 0);
#else
        asm volatile("" ::"v"(src));
#endif
      }
    }
  };
  // lane owns an adjacent column pair (see the C accesses); columns >= cout read column 0 and are never written out
  const float* wlane[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int col = n0 + wcol0 + NCT * lrow + ct;
    wlane[ct] = a.wt + (int64_t)(col < a.cout ? col : 0) * a.cin + 4 * kgrp;
  }
  auto load_b = [&](int k, int cin0, f32x4 (&bf)[NCT][SC_NSTEPS]) {
    const int64_t koff = k * wk_stride + cin0;  // wave-uniform
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float* wp = wlane[ct] + koff;
#pragma unroll
      for (int st = 0; st < SC_NSTEPS; ++st)  // columns >= cout read column 0 (finite) and are never written out;
        bf[ct][st] = *reinterpret_cast<const f32x4*>(wp + 16 * st);  // a select here would wait for the load at once
    }
  };

  int k = 0, cnt = 0;  // offset and live-row count of the current stage
  if (nstages > 0) {
    k = act_k[0];
    cnt = rl_cnt[k];
    load_goff(k, cnt);
    issue_gather(cnt, 0, 0);
    load_b(k, 0, bcur);
  }
  FSF_STAMP(2);
  // C (incl. the dump row) is zeroed while the first stage is in flight.  The A buffers are NOT cleared: rows past an
  // offset's live count hold stale data, but an MFMA output row depends on its own A row only and those rows land in
  // the dump row, which is never read back.
  for (int t = tid; t < SM::CS_FLOATS / 4; t += NT) reinterpret_cast<float4*>(Cs)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();  // (the compiler drains the DMA before the barrier) A[0] complete and visible
  FSF_STAMP(3);

  // Accumulators of ALL row blocks of the current offset stay in registers across its cin chunks: the LDS C tile is
  // read once and written once per (offset, row block) instead of once per 64-wide chunk (the C read-modify-write
  // through LDS is what competes with the MFMA issue otherwise).  Lane (lrow, kgrp) owns columns wcol0 + 2*lrow + ct
  // (adjacent pair -> one 8-byte LDS access) of compact rows rb*16 + kgrp*4 + r.
  constexpr int MAXRB = TM / 16;
  f32x4 acc[MAXRB][NCT];
  uint32_t locp[MAXRB];  // 4 local output rows (one byte each) of this lane's compact rows; dead rows -> dump row TM
  int chunk_c = 0, ki = 0;
  for (int s = 0; s < nstages; ++s) {
    const int cur = s & 1;
    int nk = k, ncnt = cnt;
    if (s + 1 < nstages) {
      const bool same_k = chunk_c + 1 < nchunks;
      const int ncin0 = same_k ? (chunk_c + 1) * SC_KC : 0;
      if (!same_k) {
        nk = act_k[ki + 1];
        ncnt = rl_cnt[nk];
        load_goff(nk, ncnt);
      }
      issue_gather(ncnt, ncin0, cur ^ 1);  // lands in the other A buffer while the matrix cores work on stage s
#ifndef FSF_ABL_NO_BLOAD
      load_b(nk, ncin0, bnext);
#else
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int st = 0; st < SC_NSTEPS; ++st) bnext[ct][st] = bcur[ct][st];
#endif
    }
    {
      const int nrb = (cnt + 15) >> 4;
      const float* abuf = As + cur * ABUF;
#ifdef FSF_ABL_NO_CRMW
      if (s == 0) {
#else
      if (chunk_c == 0) {
#endif
#pragma unroll
        for (int rb = 0; rb < MAXRB; ++rb) {
          if (rb < nrb) {
            // compacted row -> local output row (4 consecutive bytes); rows past cnt go to the dump row
            const uint32_t packed = *reinterpret_cast<const uint32_t*>(rl_loc + k * TM + rb * 16 + kgrp * 4);
            const int live = cnt - (rb * 16 + kgrp * 4);  // this lane's rows r < live exist
            const uint32_t keep = live >= 4 ? 0xffffffffu : (live <= 0 ? 0u : (1u << (8 * live)) - 1u);
            locp[rb] = (packed & keep) | ((uint32_t)TM * 0x01010101u & ~keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int l = (int)((locp[rb] >> (8 * r)) & 0xffu);
              const float* cp = Cs + l * CS_STRIDE + wcol0 + NCT * lrow;
              if constexpr (NCT == 2) {
                const float2 c2 = *reinterpret_cast<const float2*>(cp);
                acc[rb][0][r] = (l < TM) ? c2.x : 0.0f;
                acc[rb][1][r] = (l < TM) ? c2.y : 0.0f;
              } else {
                acc[rb][0][r] = (l < TM) ? cp[0] : 0.0f;
              }
            }
          }
        }
      }
#pragma unroll
      for (int rb = 0; rb < MAXRB; ++rb) {
        if (rb < nrb) {
          f32x4 af[SC_NSTEPS];
          const int arow_idx = rb * 16 + lrow;
          const float* arow = abuf + arow_idx * SC_AROW;
#pragma unroll
          for (int st = 0; st < SC_NSTEPS; ++st)  // logical chunk 4*st + kgrp lives at physical (chunk ^ (row & 15))
#ifndef FSF_ABL_NO_AREAD
            af[st] = *reinterpret_cast<const f32x4*>(arow + 4 * ((4 * st + kgrp) ^ (arow_idx & 15)));
#else
            af[st] = f32x4{(float)arow_idx, 1.f, 2.f, (float)st};
#endif
#pragma unroll
          for (int st = 0; st < SC_NSTEPS; ++st) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
#ifndef FSF_ABL_NO_MFMA
                acc[rb][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st][t], bcur[ct][st][t], acc[rb][ct], 0, 0, 0);
#else
                acc[rb][ct][t] += af[st][t] * bcur[ct][st][t];
#endif
            }
          }
        }
      }
#ifdef FSF_ABL_NO_CRMW
      if (s == nstages - 1) {
#else
      if (chunk_c == nchunks - 1) {
#endif
#pragma unroll
        for (int rb = 0; rb < MAXRB; ++rb) {
          if (rb < nrb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* cp = Cs + (int)((locp[rb] >> (8 * r)) & 0xffu) * CS_STRIDE + wcol0 + NCT * lrow;
              if constexpr (NCT == 2) *reinterpret_cast<float2*>(cp) = make_float2(acc[rb][0][r], acc[rb][1][r]);
              else cp[0] = acc[rb][0][r];
            }
          }
        }
      }
    }
    if (++chunk_c == nchunks) {
      chunk_c = 0;
      ++ki;
    }
    k = nk;
    cnt = ncnt;
    if (s + 1 < nstages) {
      // Pin the first use of the prefetched B fragments HERE (after the MFMA loop): without this hipcc hoists the
      // bcur <- bnext copies above the loop and with them the vmcnt(0) wait, which serialises the prefetch.
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int st = 0; st < SC_NSTEPS; ++st) asm volatile("" : "+v"(bnext[ct][st]) : : "memory");
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int st = 0; st < SC_NSTEPS; ++st) bcur[ct][st] = bnext[ct][st];
    }
#ifndef FSF_ABL_NO_BARRIER
    __syncthreads();  // A[cur^1] landed + visible, everyone is done reading A[cur]
#endif
  }
  FSF_STAMP(4);
  write_tile<TN, TM, true, NT>(a, Cs, o0, n0, zsplit);
  if (a.ksplit > 1) {
    // publish the partial tile (write-through stores), then take an arrival ticket: every wave drains its stores,
    // barrier, ONE lane draws the relaxed agent-scope ticket; the last arriver acquires once (drops this CU's stale L1
    // lines) and folds.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(a.queue + SC_NXCD + item[0] * a.cout_blocks + item[1], 1, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == a.ksplit - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      item[3] = ticket;
    }
    __syncthreads();
    if (item[3] == a.ksplit - 1) fold_tile<TN, TM, NT>(a, Cs, o0, n0, zsplit);
  }
  FSF_STAMP(5);
#ifdef FSF_ABL_TIMING
  if (threadIdx.x == 0 && FSF_BID < 4096) {
    fsf_dbg[FSF_BID * 8 + 6] = nstages;
    unsigned hwid;  // HW_ID: cu_id [11:8], sh [12], se [15:13]; XCC_ID is a separate register
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    fsf_dbg[FSF_BID * 8 + 7] = (long long)hwid | ((long long)(xcc & 0xf) << 32);
  }
#endif
  }  // work-item loop
}

// ------------------------------------------------------------------------------------------------------
// Generic path (cin % 16 == 0, any cin): synchronous register-staged gather, used for the odd channel counts
// only the unit tests exercise.
template <int TN>
__global__ void __launch_bounds__(256, 2) spconv_fwd_generic_kernel(SpconvArgs a) {
  using SM = SpconvSmem<TN>;
  constexpr int CS_STRIDE = SM::CS_STRIDE;
  constexpr int WCOLS = TN / 4;
  constexpr int NCT = WCOLS / 16;
  constexpr int ASTRIDE = SC_KC + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Cs = reinterpret_cast<float*>(smem);
  float* As = Cs + SM::CS_FLOATS;  // [SC_TM][ASTRIDE]
  int32_t* rl_in = reinterpret_cast<int32_t*>(As + SM::A_FLOATS);
  uint8_t* rl_loc = reinterpret_cast<uint8_t*>(rl_in + SC_MAXK * SC_TM);
  int32_t* rl_cnt = reinterpret_cast<int32_t*>(rl_loc + SC_MAXK * SC_TM);
  int32_t* act_k = rl_cnt + SC_MAXK;
  int32_t* act_n = act_k + SC_MAXK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int64_t o0 = (int64_t)tile * SC_TM;
  const int n0 = blockIdx.y * TN;
  const int zsplit = blockIdx.z;
  build_row_lists<SC_TM>(a, o0, zsplit, reinterpret_cast<int32_t*>(As), rl_in, rl_loc, rl_cnt, act_k, act_n);
  for (int t = tid; t < SM::CS_FLOATS + SM::A_FLOATS; t += 256) Cs[t] = 0.0f;
  __syncthreads();

  const int lrow = lane & 15, kgrp = lane >> 4;
  const int wcol0 = wave * WCOLS;
  const int nact = act_n[0];
  for (int ai = 0; ai < nact; ++ai) {
    const int k = act_k[ai];
    const int cnt = rl_cnt[k];
    const int nrb = (cnt + 15) >> 4;
    for (int cin0 = 0; cin0 < a.cin; cin0 += SC_KC) {
      const int kc = (a.cin - cin0 < SC_KC) ? (a.cin - cin0) : SC_KC;  // multiple of 16
      const int f4_per_row = kc >> 2;
      for (int t = tid; t < cnt * f4_per_row; t += 256) {
        const int j = t / f4_per_row, v = t % f4_per_row;
        const int64_t in_off = rl_in[k * SC_TM + j];
        *reinterpret_cast<float4*>(As + j * ASTRIDE + 4 * v) =
            *reinterpret_cast<const float4*>(a.feat + in_off + cin0 + 4 * v);
      }
      __syncthreads();
      const int nsteps = kc >> 4;
      for (int rb = 0; rb < nrb; ++rb) {
        f32x4 acc[NCT];
        int loc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = rb * 16 + kgrp * 4 + r;
          const int l = (int)rl_loc[k * SC_TM + (j < SC_TM ? j : SC_TM - 1)];
          loc[r] = (j < cnt) ? l : SC_TM;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float c = Cs[loc[r] * CS_STRIDE + wcol0 + ct * 16 + lrow];
            acc[ct][r] = (loc[r] < SC_TM) ? c : 0.0f;
          }
        const float* arow = As + (rb * 16 + lrow) * ASTRIDE + 4 * kgrp;
        for (int st = 0; st < nsteps; ++st) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * st);
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const int col = n0 + wcol0 + ct * 16 + lrow;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (col < a.cout)
              w = *reinterpret_cast<const f32x4*>(a.wt + ((int64_t)k * a.cout + col) * a.cin + cin0 + 4 * kgrp + 16 * st);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], w[t], acc[ct], 0, 0, 0);
          }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[loc[r] * CS_STRIDE + wcol0 + ct * 16 + lrow] = acc[ct][r];
      }
      __syncthreads();
    }
  }
  write_tile<TN>(a, Cs, o0, n0, zsplit);
}

// folds the ksplit partial tiles in fixed order and applies the epilogue
__global__ void __launch_bounds__(256) spconv_reduce_kernel(SpconvArgs a) {
  const int64_t total4 = a.m_out * (a.cout / 4);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / (a.cout / 4);
    const int col = (int)(t - o * (a.cout / 4)) * 4;
    float4 acc = *reinterpret_cast<const float4*>(a.partial + o * a.cout + col);
    for (int z = 1; z < a.ksplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(a.partial + ((int64_t)z * a.m_out + o) * a.cout + col);
      acc.x = __fadd_rn(acc.x, v.x);
      acc.y = __fadd_rn(acc.y, v.y);
      acc.z = __fadd_rn(acc.z, v.z);
      acc.w = __fadd_rn(acc.w, v.w);
    }
    acc.x = epilogue_one(a, acc.x, o, col);
    acc.y = epilogue_one(a, acc.y, o, col + 1);
    acc.z = epilogue_one(a, acc.z, o, col + 2);
    acc.w = epilogue_one(a, acc.w, o, col + 3);
    *reinterpret_cast<float4*>(a.out + o * a.cout + col) = acc;
  }
}

// weight [kvol][cin][cout] -> [kvol][cout][cin]
__global__ void __launch_bounds__(256)
    transpose_weight_kernel(const float* __restrict__ w, int kvol, int cin, int cout, float* __restrict__ wt) {
  const int64_t total = (int64_t)kvol * cin * cout;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(t % cin);
    const int64_t r = t / cin;
    const int co = (int)(r % cout);
    const int k = (int)(r / cout);
    wt[t] = w[((int64_t)k * cin + ci) * cout + co];
  }
}

// How many ways to split the offset loop of a layer (work items = tiles x cout blocks x splits, handed to `slots`
// resident workgroups).  A small layer needs the split to fill the chip at all; a mid-size one to avoid a last round
// that is mostly empty (864 items on 512 slots run as long as 1024).  Cost model in microseconds, constants measured
// on the 10-sweep frame: a work item costs ~8 us of fixed work (neighbour-table block, compaction, first gather, tile
// write) + ~3.8 us per (offset, 64-channel chunk) stage when two workgroups share a CU's matrix pipe (x0.65 when one
// has the CU to itself); every extra split adds a partial tile written and re-read by the last-arriving workgroup.
static int pick_ksplit(int64_t tiles, int cout_blocks, int kvol, int nchunks, int64_t m_out, int cout, int64_t slots) {
  if (kvol < 3) return 1;
  int gmax = kvol / 3 < 9 ? kvol / 3 : 9;
  const double t_stage = cout <= 64 ? 2.6 : 3.8, fixed = 8.0;
  const double tile_mb = (double)m_out * cout * 4 * 1e-6;
  double best = 0;
  int best_g = 1;
  for (int g = 1; g <= gmax; ++g) {
    const double n = (double)tiles * cout_blocks * g;
    // items differ in length (p90 / mean ~ 1.25): a single round ends with its slowest item; many rounds of
    // dynamically scheduled items end about 0.8 item lengths after the mean load per slot
    const double rounds = n <= slots ? 1.3 : n / slots + 0.8;
    const int offsets = (kvol + g - 1) / g;
    double item = fixed + offsets * nchunks * t_stage;
    if (2 * n <= slots) item *= 0.65;
    const double fold = g > 1 ? 0.5 * (2 * g - 1) * tile_mb / 3.0 : 0.0;  // MB / (3 TB/s) = us; half of it hides under other items
    const double t = rounds * item + fold;
    if (g == 1 || t < best) best = t, best_g = g;
  }
  return best_g;
}

// Launch shape of one layer.  The 128-row tile (FSF_SPCONV_TM=128) loads each weight fragment once per 128 rows and rounds
// the compacted row counts to 16 over twice as many rows (row-block utilisation 0.85 -> 0.91 on the level-2 layers); it
// needs ~150 KB of LDS, i.e. one workgroup per CU.
struct SpconvPlan {
  int tm, ksplit, cout_blocks;
  int64_t tiles;
};

static SpconvPlan spconv_plan(int64_t m_out, int cin, int cout, int kvol) {
  const int tm_env = 0;
  SpconvPlan p;
  p.cout_blocks = cout <= 64 ? 1 : (cout + 127) / 128;
  const bool fast = (cin % SC_KC) == 0;
  // Measured on the 10-sweep frame (34 layers): the 128-row tile is 5-30 % SLOWER on every layer (U-Net 16.1 ms vs 14.3 ms)
  // — one workgroup per CU leaves a single wave per SIMD and nothing to hide the per-stage barrier and LDS latency
  // behind — so it stays opt-in until it gets a second wave group.
  const bool big = tm_env == 128 && fast;
  p.tm = big ? 128 : SC_TM;
  p.tiles = (m_out + p.tm - 1) / p.tm;
  p.ksplit = pick_ksplit(p.tiles, p.cout_blocks, kvol, (cin + SC_KC - 1) / SC_KC, m_out, cout, big ? 256 : 512);
  return p;
}

static int64_t spconv_queue_bytes(int64_t tiles, int cout_blocks, int ksplit) {
  return fsf_align_up((SC_NXCD + (ksplit > 1 ? tiles * cout_blocks : 0)) * (int64_t)sizeof(int), 256);
}

}  // namespace fsf

using namespace fsf;

#ifdef FSF_ABL_TIMING
extern "C" int fsf_debug_read(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fsf::fsf_dbg), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int fsf_spconv_transpose_weight(const float* weight, int32_t kvol, int32_t cin, int32_t cout, float* weight_t,
                                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !weight_t || kvol < 1 || cin < 1 || cout < 1) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(transpose_weight_kernel, dim3(fsf_stream_grid((int64_t)kvol * cin * cout, 256)), dim3(256), 0, stream,
                     weight, (int)kvol, (int)cin, (int)cout, weight_t);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_spconv_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout, int32_t kvol) {
  if (m_out <= 0 || cin < 1 || cout < 1 || kvol < 1) return 256;
  const SpconvPlan p = spconv_plan(m_out, cin, cout, kvol);
  return (p.ksplit > 1 ? fsf_align_up((int64_t)p.ksplit * m_out * cout * 4, 256) : 0) +
         spconv_queue_bytes(p.tiles, p.cout_blocks, p.ksplit);
}

extern "C" int fsf_spconv_forward(const float* feat, int64_t m_in, int32_t cin, const float* weight_t, int32_t kvol,
                                  int32_t cout, const int32_t* nbr, int64_t m_out, const float* scale, const float* shift,
                                  const float* residual, int32_t relu, float* out, void* workspace, int64_t workspace_bytes,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m_in < 0 || m_out < 0 || cin < 1 || cout < 1 || kvol < 1 || !weight_t || (scale && !shift) ||
      (m_out > 0 && (!nbr || !out)) || (m_in > 0 && !feat))
    return FSF_ERR_INVALID_ARG;
  if (kvol > SC_MAXK || (cin % 16) != 0 || (cout % 4) != 0) return FSF_ERR_UNSUPPORTED;
  if (m_in * cin >= ((int64_t)1 << 31)) return FSF_ERR_UNSUPPORTED;  // row lists hold 32-bit element offsets
  if (m_out == 0) return FSF_OK;
  const SpconvPlan plan = spconv_plan(m_out, cin, cout, kvol);
  if (plan.tiles >= (int64_t)1 << 31) return FSF_ERR_UNSUPPORTED;
  const int ksplit = plan.ksplit;
  if (workspace_bytes < fsf_spconv_workspace_bytes(m_out, cin, cout, kvol)) return FSF_ERR_WORKSPACE;
  if (!workspace) return FSF_ERR_WORKSPACE;
  const bool fast = (cin % SC_KC) == 0;
  // workspace = [ksplit partial tiles][work-queue counters]
  int* queue = reinterpret_cast<int*>((char*)workspace + (ksplit > 1 ? fsf_align_up((int64_t)ksplit * m_out * cout * 4, 256) : 0));
  SpconvArgs a{feat, weight_t, nbr, scale, shift, residual, out, (float*)workspace, m_in, m_out,
               (int)cin, (int)cout, (int)kvol, (int)relu, ksplit, queue, (int)plan.tiles, plan.cout_blocks};
  dim3 grid((unsigned)plan.tiles, plan.cout_blocks, ksplit);
  if (fast) {  // persistent: one workgroup per resident slot
    const int64_t items = plan.tiles * plan.cout_blocks * ksplit;
    const int64_t slots = plan.tm == 128 ? 256 : 512;
    grid = dim3((unsigned)(items < slots ? items : slots), 1, 1);
    FSF_HIP_TRY(hipMemsetAsync(queue, 0, spconv_queue_bytes(plan.tiles, plan.cout_blocks, ksplit), stream));
  }
#define FSF_SPCONV_LAUNCH(KERNEL, SMEM_T, NTHREADS)                                                                            \
  do {                                                                                                               \
    static std::atomic<uint64_t> attr_done{0};                                                                                    \
    const size_t smem_bytes = SMEM_T::bytes();                                                                       \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)KERNEL, (int)smem_bytes, attr_done));                                                                                                                \
    hipLaunchKernelGGL(KERNEL, grid, dim3(NTHREADS), smem_bytes, stream, a);                                         \
  } while (0)
  using S64_64 = SpconvSmem<64, 64>;
  using S128_64 = SpconvSmem<128, 64>;
  using S64_128 = SpconvSmem<64, 128>;
  using S128_128 = SpconvSmem<128, 128>;
  const bool wide = true;  // (8 waves per workgroup for the 128-column kernel)
  if (cout <= 64) {
    if (fast && plan.tm == 128) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<64, 128, 4>), S64_128, 256);
    else if (fast) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<64, 64, 4>), S64_64, 256);
    else FSF_SPCONV_LAUNCH((spconv_fwd_generic_kernel<64>), S64_64, 256);
  } else {
    if (fast && plan.tm == 128 && wide) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<128, 128, 8>), S128_128, 512);
    else if (fast && plan.tm == 128) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<128, 128, 4>), S128_128, 256);
    else if (fast && wide) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<128, 64, 8>), S128_64, 512);
    else if (fast) FSF_SPCONV_LAUNCH((spconv_fwd_dma_kernel<128, 64, 4>), S128_64, 256);
    else FSF_SPCONV_LAUNCH((spconv_fwd_generic_kernel<128>), S128_64, 256);
  }
#undef FSF_SPCONV_LAUNCH
  if (ksplit > 1 && !fast)
    hipLaunchKernelGGL(spconv_reduce_kernel, dim3(fsf_stream_grid(m_out * (cout / 4), 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
