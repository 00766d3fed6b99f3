// K9b: sparse convolution forward, ROW-stationary, on the bf16 matrix cores with the exact 3-way bf16 split (fp32-accurate).
// See include/fsf_hip.h (fsf_spconv_forward_split) and spconv.hip for the fp32-pipe kernel it complements.
//
// out[o, :] = act(scale * (sum_k feat[nbr[o, k], :] @ W[k]) + shift + residual).  The fp32-pipe kernel keeps a
// [64 rows x 128 channels] tile in LDS and compacts, per offset, the rows that have a neighbour; its matrix phase is
// bound by the fp32 MFMA rate (1/16 of bf16) and the compaction costs an LDS read-modify-write of the tile per offset,
// row lists, an A tile in LDS and a workgroup barrier per stage.  Here a wave OWNS 32 output rows for all 27 offsets:
//   * transposed product out^T[channel, row] = W_k^T[channel, cin] x X_k^T[cin, row] — the weights are the A operand
//     (split once per layer into hi/mid/lo bf16 planes in fragment order, streamed through LDS per (offset, 32-cin chunk),
//     double-buffered by LDS-DMA), the wave's 2 x 16 rows are the B operand: lane (row, g) reads the 32 bytes of ITS
//     neighbour row straight from HBM/L2 (no A tile in LDS, no compaction, no row lists) and splits them in registers —
//     each input value is split by exactly one wave;
//   * the accumulators (2 row groups x 8 channel tiles) stay in registers over the whole offset loop: no LDS C tile, no
//     scatter; rows without a neighbour at an offset contribute zeros (their lanes are wasted: 46 % on the 0.4 m level,
//     80 % on the 0.2 m level — paid for by the 2.5x cheaper six-term bf16 product; a wave skips an offset none of its
//     32 rows has);
//   * six leading cross terms of the exact split, fp32 accumulation: fp32 accuracy (see linear_norm_act.hip).
// Deterministic: fixed (offset, cin) order, no atomics.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace fsf {

typedef __bf16 scs_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 scs_f16x8 __attribute__((ext_vector_type(8)));
typedef float scs_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned scs_u32x4 __attribute__((ext_vector_type(4)));

constexpr int SCS_KC = 32;   // cin per LDS weight chunk (one MFMA k step)
constexpr int SCS_NW = 4;    // waves per workgroup
constexpr int SCS_RG = 2;    // 16-row groups per wave
constexpr int SCS_ROWS = SCS_NW * SCS_RG * 16;
#ifndef SCS_EPI_TB
#define SCS_EPI_TB 1  // (4: the whole residual row before the stores — 42 spilled VGPRs, 1.5 % slower)
#endif
#ifndef SCS_WPS
#define SCS_WPS 3            // workgroups per CU the register budget is set for
#endif

struct ScsArgs {
  const float* feat;
  const uint4* planes;  // [slice][kvol][cin/32][T][3][64 lanes] x 16 B
  const int32_t* nbr;
  const float *scale, *shift, *residual;
  float* out;
  float* partial;  // [ksplit][m_out][cout] raw sums when the (offset, cin chunk) sequence is split over gridDim.z
  int64_t m_in, m_out;
  int cin, cout, kvol, relu, ksplit;
  // xcd_lanes > 0: a 1-D grid of 8 * xcd_lanes * ceil(nslice * ksplit / 8) workgroups in which the xcd_lanes workgroups that stream
  // the SAME weight chunks (one (slice, k range), different row blocks) are congruent modulo 8, i.e. share an XCD and its L2
  int xcd_lanes, nslice;
  // K9b-XP: `feat` holds the input rows in PLANE form ([row][cin / 8][2][8] f16 hi | lo of x * s_row, fsf_rows_to_planes: the same
  // 4 cin bytes per row, so every address of the fp32 form stands) with x_inv_scale[row] = 1 / s_row, `planes` = f16 hi | lo weight
  // fragments behind a 256-byte header (1 / s_w first); three v_mfma_f32_16x16x32_f16 per fp32-equivalent product.  A lane's rows
  // change scale from offset to offset: the accumulators are kept in the UNIT of the row being multiplied (K9e's scheme) — before an
  // offset's first MFMA they are multiplied by old unit / new unit, a power of two (exact) — and brought back once at the end.
  const float* x_inv_scale;
};

__device__ __forceinline__ void scs_split8(const float (&v)[8], scs_u32x4& hi, scs_u32x4& mid, scs_u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    const float ah = __uint_as_float(__float_as_uint(a) & 0xffff0000u), bh = __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float ar = __fsub_rn(a, ah), br = __fsub_rn(b, bh);
    const float am = __uint_as_float(__float_as_uint(ar) & 0xffff0000u), bm = __uint_as_float(__float_as_uint(br) & 0xffff0000u);
    const float al = __fsub_rn(ar, am), bl = __fsub_rn(br, bm);
    hi[j] = __builtin_amdgcn_perm(__float_as_uint(bh), __float_as_uint(ah), 0x07060302u);
    mid[j] = __builtin_amdgcn_perm(__float_as_uint(bm), __float_as_uint(am), 0x07060302u);
    lo[j] = __builtin_amdgcn_perm(__float_as_uint(bl), __float_as_uint(al), 0x07060302u);
  }
}

// weight [kvol][cin][cout] fp32 (the spconv v1 layout) -> fragment-ordered bf16 planes of W_k^T
__global__ void __launch_bounds__(256)
    scs_prepare_kernel(const float* __restrict__ w, int kvol, int cin, int cout, int T, int nkc, int nslice, uint4* planes) {
  const int64_t total = (int64_t)nslice * kvol * nkc * T * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t r = idx >> 6;
    const int t = (int)(r % T); r /= T;
    const int kc = (int)(r % nkc); r /= nkc;
    const int k = (int)(r % kvol);
    const int slice = (int)(r / kvol);
    const int col = 128 * slice + 16 * t + (lane & 15), c0 = kc * SCS_KC + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (col < cout && c0 + e < cin) ? w[((int64_t)k * cin + c0 + e) * cout + col] : 0.0f;
    scs_u32x4 hi, mid, lo;
    scs_split8(v, hi, mid, lo);
    uint4* dst = planes + ((((int64_t)slice * kvol + k) * nkc + kc) * T + t) * 3 * 64 + lane;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
    dst[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// ---- K9b-XP weights: one power-of-two scale per layer (s * max |w| in [2^13, 2^14)), hi = rn_f16(w s), lo = rn_f16(w s - hi)
__device__ __forceinline__ void scs_pick_scale(float amax, float& s, float& inv) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  e = amax > 0.0f ? (e < -113 ? -113 : e) : 13;
  s = __uint_as_float((unsigned)(13 - e + 127) << 23);
  inv = __uint_as_float((unsigned)(e - 13 + 127) << 23);
}

__global__ void __launch_bounds__(256) scs_weight_absmax_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ hdr) {
  __shared__ float wave_max[4];
  float amax = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) amax = fmaxf(amax, fabsf(w[i]));
  amax = fsf_wave_max(amax);
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(hdr + 2, __float_as_uint(fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]))));
}

__global__ void __launch_bounds__(256)
    scs_prepare_f16_kernel(const float* __restrict__ w, int kvol, int cin, int cout, int T, int nkc, int nslice, float* __restrict__ hdr,
                           uint4* __restrict__ planes) {
  float s_w, inv_w;
  scs_pick_scale(__uint_as_float(reinterpret_cast<const unsigned*>(hdr)[2]), s_w, inv_w);
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = inv_w; hdr[1] = s_w; }
  const int64_t total = (int64_t)nslice * kvol * nkc * T * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t r = idx >> 6;
    const int t = (int)(r % T); r /= T;
    const int kc = (int)(r % nkc); r /= nkc;
    const int k = (int)(r % kvol);
    const int slice = (int)(r / kvol);
    const int col = 128 * slice + 16 * t + (lane & 15), c0 = kc * SCS_KC + 8 * (lane >> 4);
    scs_f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (col < cout && c0 + e < cin) ? w[((int64_t)k * cin + c0 + e) * cout + col] : 0.0f;
      const float xs = __fmul_rn(v, s_w);
      h[e] = (_Float16)xs;
      l[e] = (_Float16)__fsub_rn(xs, (float)h[e]);
    }
    const scs_u32x4 hi = __builtin_bit_cast(scs_u32x4, h), lo = __builtin_bit_cast(scs_u32x4, l);
    uint4* dst = planes + ((((int64_t)slice * kvol + k) * nkc + kc) * T + t) * 2 * 64 + lane;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

template <int T, bool XP = false>
__global__ void __launch_bounds__(SCS_NW * 64, SCS_WPS) spconv_fwd_split_kernel(ScsArgs a) {
  constexpr int NPL = XP ? 2 : 3;
  constexpr int CHUNK_U4 = T * NPL * 64;
  extern __shared__ __attribute__((aligned(16))) char scs_smem[];
  uint4* wbuf = reinterpret_cast<uint4*>(scs_smem);  // [2][CHUNK_U4], then scale | shift of this 128-channel slice
  float* vec = reinterpret_cast<float*>(wbuf + 2 * CHUNK_U4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = lane & 15, grp = lane >> 4;
  const int nkc = (a.cin + SCS_KC - 1) / SCS_KC;
  const int nchunks = a.kvol * nkc;
  // a small layer (too few 128-row workgroups to fill the chip) splits the chunk sequence over gridDim.z; the slices are
  // folded in order by scs_fold_kernel
  // (slice, k range, first row block, row-block step) of this workgroup.  The deep U-Net levels have a dozen row blocks and 42 MB of
  // split weights per layer: every row block streams all of them, and with the row block on blockIdx.x the twelve workgroups that
  // read the same chunks were dealt to eight different XCDs — each L2 pulled the whole weight set through the fabric (630 MB per
  // launch measured against 184 MB algorithmic).  In the XCD-aware layout they sit on ONE XCD and meet in its L2.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z, bstep = (int)gridDim.x;
  if (a.xcd_lanes > 0) {
    const int wg = (int)blockIdx.x, j = wg >> 3;
    const int group = (j / a.xcd_lanes) * 8 + (wg & 7);
    if (group >= a.nslice * a.ksplit) return;  // (uniform: the last round of groups may be partial)
    bx = j % a.xcd_lanes; bstep = a.xcd_lanes;
    by = group % a.nslice; bz = group / a.nslice;
  }
  const int ci_begin = (int)((int64_t)nchunks * bz / a.ksplit), ci_end = (int)((int64_t)nchunks * (bz + 1) / a.ksplit);
  const int64_t nblk = (a.m_out + SCS_ROWS - 1) / SCS_ROWS;
  const int ch_base = 128 * by;
  const uint4* planes = a.planes + (XP ? 16 : 0) + (int64_t)by * nchunks * CHUNK_U4;  // (XP: behind the 256-byte header)
  const int last_quad = a.cin - 4;
  // The epilogue's per-channel vectors go through LDS, and its residual row is loaded before the first store: as plain
  // global loads inside the tile loop each one sat behind the previous tile's store (the pointers may alias) — ~48
  // serialized round trips at the end of every workgroup's chain.
  {
    const int t = threadIdx.x;  // 256 threads: scale[128] | shift[128]
    const int ch = 128 * by + (t & 127);
    const float* src = t < 128 ? a.scale : a.shift;
    vec[t] = (src && ch < a.cout) ? src[ch] : (t < 128 ? 1.0f : 0.0f);
  }
  __syncthreads();

  auto stage_w = [&](int ci, int buf) {
    const float* src = reinterpret_cast<const float*>(planes + (int64_t)ci * CHUNK_U4);
    float* dst = reinterpret_cast<float*>(wbuf + buf * CHUNK_U4);
    for (int u = wave * 64; u < CHUNK_U4; u += SCS_NW * 64)
      __builtin_amdgcn_global_load_lds(src + 4 * (u + lane), dst + 4 * u, 16, 0, 0);
  };

  for (int64_t blk = bx; blk < nblk; blk += bstep) {
    const int64_t row0 = blk * SCS_ROWS + (int64_t)wave * (SCS_RG * 16);
    scs_f32x4 acc[SCS_RG][T];
#pragma unroll
    for (int rg = 0; rg < SCS_RG; ++rg)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[rg][t] = scs_f32x4{0.f, 0.f, 0.f, 0.f};
    const int32_t* nrow[SCS_RG];  // this lane's row of the neighbour table
#pragma unroll
    for (int rg = 0; rg < SCS_RG; ++rg) {
      int64_t r = row0 + 16 * rg + rowl;
      if (r >= a.m_out) r = a.m_out - 1;  // rows past m_out repeat the last one (never stored)
      nrow[rg] = a.nbr + r * a.kvol;
    }
    // neighbour ids two offsets ahead of their use, raw x one chunk ahead (a missing neighbour reads row 0 and is zeroed)
    int k = ci_begin / nkc, kc = ci_begin - k * nkc;
    int idx_cur[SCS_RG], idx_nxt[SCS_RG];
#pragma unroll
    for (int rg = 0; rg < SCS_RG; ++rg) {
      idx_cur[rg] = nrow[rg][k];
      idx_nxt[rg] = k + 1 < a.kvol ? nrow[rg][k + 1] : -1;
    }
    auto load_x = [&](const int (&idx)[SCS_RG], int kc, float (&v)[SCS_RG][8]) {
#pragma unroll
      for (int rg = 0; rg < SCS_RG; ++rg) {
        const float* base = a.feat + (int64_t)(idx[rg] < 0 ? 0 : idx[rg]) * a.cin;
        const int cq = kc * SCS_KC + 8 * grp;
        const float4 p = *reinterpret_cast<const float4*>(base + min(cq, last_quad));
        const float4 q = *reinterpret_cast<const float4*>(base + min(cq + 4, last_quad));
        v[rg][0] = p.x; v[rg][1] = p.y; v[rg][2] = p.z; v[rg][3] = p.w;
        v[rg][4] = q.x; v[rg][5] = q.y; v[rg][6] = q.z; v[rg][7] = q.w;
      }
    };
    float xc[SCS_RG][8];
    load_x(idx_cur, kc, xc);
    // XP: inverse row scale of the rows in `xc` (the chunk multiplied next) and the unit the accumulators are currently kept in
    // A row's unit is CAPPED at 2^60: the accumulators move between the units of the rows they meet, and a neighbour row
    // whose maximum is tiny but not zero (1e-30 behind a ReLU) would otherwise multiply accumulators of ~2^42 by ~2^100 — inf, and inf
    // again after the way back.  Such a row's f16 fragments are scaled down to the capped unit instead (`dn` = 2^60 / s_row, a power of
    // two: exact until it flushes what lies 2^-22 below 7e-15); rows of ordinary magnitude never see the branch.
    float sc_x[SCS_RG], inv_cur[SCS_RG], dn[SCS_RG];
    auto row_unit = [&](int rg) {
      const float raw = a.x_inv_scale[idx_cur[rg] < 0 ? 0 : idx_cur[rg]];
      sc_x[rg] = fmaxf(raw, 0x1p-60f);
      dn[rg] = raw * __uint_as_float(0x7f000000u - __float_as_uint(sc_x[rg]));  // 1 unless capped
    };
#pragma unroll
    for (int rg = 0; rg < SCS_RG; ++rg) {
      inv_cur[rg] = 1.0f;
      sc_x[rg] = 1.0f;
      dn[rg] = 1.0f;
      if constexpr (XP) row_unit(rg);
    }
    const bool tail_chunks = (a.cin % SCS_KC) != 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with both weight buffers of the previous row block
    asm volatile("" ::: "memory");
    stage_w(ci_begin, 0);
    for (int ci = ci_begin; ci < ci_end; ++ci) {
      const int buf = (ci - ci_begin) & 1;
      // ---- split the chunk that arrived while the previous one was multiplied
      scs_u32x4 xh[SCS_RG], xm[SCS_RG], xl[SCS_RG];
      bool any_live = false;
      if constexpr (XP) {  // first chunk of an offset (or of this workgroup's k range): the accumulators move to the new rows' unit
        if (ci == ci_begin || kc == 0) {
#pragma unroll
          for (int rg = 0; rg < SCS_RG; ++rg) {
            if (idx_cur[rg] >= 0) {
              const float ratio = inv_cur[rg] * __uint_as_float(0x7f000000u - __float_as_uint(sc_x[rg]));  // old unit / new unit: a power of two
#pragma unroll
              for (int t = 0; t < T; ++t) acc[rg][t] = acc[rg][t] * ratio;
              inv_cur[rg] = sc_x[rg];
            }
          }
        }
      }
#pragma unroll
      for (int rg = 0; rg < SCS_RG; ++rg) {
        const bool live = idx_cur[rg] >= 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) xc[rg][e] = live ? xc[rg][e] : 0.0f;  // a missing neighbour contributes zeros
        if constexpr (XP) {  // the planes ARE the operands (cin is a multiple of 32 here)
#pragma unroll
          for (int e = 0; e < 4; ++e) { xh[rg][e] = __float_as_uint(xc[rg][e]); xl[rg][e] = __float_as_uint(xc[rg][4 + e]); }
          if (dn[rg] != 1.0f) {  // (a row beyond the unit cap: its fragments move to the capped unit)
            typedef _Float16 scs_h2 __attribute__((ext_vector_type(2)));
            const _Float16 d = (_Float16)dn[rg];
            const scs_h2 d2 = {d, d};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xh[rg][e] = __builtin_bit_cast(unsigned, (scs_h2)(__builtin_bit_cast(scs_h2, xh[rg][e]) * d2));
              xl[rg][e] = __builtin_bit_cast(unsigned, (scs_h2)(__builtin_bit_cast(scs_h2, xl[rg][e]) * d2));
            }
          }
          xm[rg] = xh[rg];
          any_live |= live;
          continue;
        }
        if (tail_chunks) {  // (uniform) a chunk can reach past cin only when cin is not a multiple of 32
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kc * SCS_KC + 8 * grp + e >= a.cin) xc[rg][e] = 0.0f;
        }
        scs_split8(xc[rg], xh[rg], xm[rg], xl[rg]);
        any_live |= live;
      }
      const bool wave_live = __ballot(any_live) != 0ull;  // does any of this wave's 32 rows have a neighbour at offset k?
      // ---- advance (k, kc) to the next chunk; the neighbour ids of its offset are already in registers
      const int k_this = k;
      int nk = k, nkci = kc + 1;
      if (nkci == nkc) { nkci = 0; nk = k + 1; }
      if (nk != k) {
#pragma unroll
        for (int rg = 0; rg < SCS_RG; ++rg) idx_cur[rg] = idx_nxt[rg];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // chunk ci's weights have landed everywhere; buffer buf^1 is free
      asm volatile("" ::: "memory");
      if (ci + 1 < ci_end) {
        stage_w(ci + 1, buf ^ 1);
        load_x(idx_cur, nkci, xc);
        if constexpr (XP) {
          if (nk != k) {
#pragma unroll
            for (int rg = 0; rg < SCS_RG; ++rg) row_unit(rg);
          }
        }
        if (nk != k) {  // first chunk of a new offset: fetch the ids of the offset after it
#pragma unroll
          for (int rg = 0; rg < SCS_RG; ++rg) idx_nxt[rg] = nk + 1 < a.kvol ? nrow[rg][nk + 1] : -1;
        }
      }
      k = nk;
      kc = nkci;
      (void)k_this;
      if (wave_live) {
        const uint4* wc = wbuf + buf * CHUNK_U4;
        if constexpr (XP) {
#pragma unroll
          for (int t = 0; t < T; t += 2) {
            scs_f16x8 wfr[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              const uint4* wf = wc + ((t + tt) * 2) * 64 + lane;
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) wfr[tt][pl] = __builtin_bit_cast(scs_f16x8, wf[64 * pl]);
            }
            constexpr int TERM_W[3] = {1, 0, 0};  // (weight plane, x plane): lo hi, hi lo, hi hi — small terms first
            constexpr int TERM_X[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int rg = 0; rg < SCS_RG; ++rg) {
                  const scs_u32x4 xb = TERM_X[term] == 0 ? xh[rg] : xl[rg];
                  acc[rg][t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfr[tt][TERM_W[term]], __builtin_bit_cast(scs_f16x8, xb),
                                                                           acc[rg][t + tt], 0, 0, 0);
                }
          }
        } else {
#pragma unroll
        for (int t = 0; t < T; t += 2) {
          scs_bf16x8 wfr[2][3];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const uint4* wf = wc + ((t + tt) * 3) * 64 + lane;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wfr[tt][pl] = __builtin_bit_cast(scs_bf16x8, wf[64 * pl]);
          }
          constexpr int TERM_W[6] = {2, 0, 1, 1, 0, 0};  // (weight plane, x plane) of the six leading cross terms,
          constexpr int TERM_X[6] = {0, 2, 1, 0, 1, 0};  // small ones first
#pragma unroll
          for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int rg = 0; rg < SCS_RG; ++rg) {
                const scs_u32x4 xb = TERM_X[term] == 0 ? xh[rg] : TERM_X[term] == 1 ? xm[rg] : xl[rg];
#ifndef FSF_ABL_SCS_NO_MFMA
                acc[rg][t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[tt][TERM_W[term]], __builtin_bit_cast(scs_bf16x8, xb),
                                                                          acc[rg][t + tt], 0, 0, 0);
#else  // ablation: one VALU op per product term keeps the operand loads alive without the matrix pipe
                acc[rg][t + tt][term & 3] += __uint_as_float(xb[term & 3] ^ __builtin_bit_cast(scs_u32x4, wfr[tt][TERM_W[term]])[term & 3]);
#endif
              }
        }
        }
      }
    }
    if constexpr (XP) {  // back from the last rows' unit (and the weights') to the plain product
      const float w_inv = *reinterpret_cast<const float*>(a.planes);
#pragma unroll
      for (int rg = 0; rg < SCS_RG; ++rg) {
        const float sc = inv_cur[rg] * w_inv;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[rg][t] = acc[rg][t] * sc;
      }
    }
    // ---- epilogue: lane (row, g) holds channels ch_base + 16 t + 4 g + r of its row
#pragma unroll
    for (int rg = 0; rg < SCS_RG; ++rg) {
      const int64_t row = row0 + 16 * rg + rowl;
      if (row < a.m_out) {
        if (a.ksplit > 1) {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int ch0 = ch_base + 16 * t + 4 * grp;
            if (ch0 < a.cout)
              *reinterpret_cast<float4*>(a.partial + ((int64_t)bz * a.m_out + row) * a.cout + ch0) =
                  make_float4(acc[rg][t][0], acc[rg][t][1], acc[rg][t][2], acc[rg][t][3]);
          }
          continue;
        }
        const bool affine = a.scale || a.shift;
        constexpr int TB = SCS_EPI_TB < T ? SCS_EPI_TB : T;  // the residual row TB tiles at a time: loads first, then the stores
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += TB) {
          float4 rs[TB];
          if (a.residual) {
#pragma unroll
            for (int t = 0; t < TB; ++t) {
              const int ch0 = ch_base + 16 * (t0 + t) + 4 * grp;
              rs[t] = *reinterpret_cast<const float4*>(a.residual + row * a.cout + (ch0 < a.cout ? ch0 : 0));
            }
          }
#pragma unroll
          for (int tt = 0; tt < TB; ++tt) {
            const int t = t0 + tt;
            const int ch0 = ch_base + 16 * t + 4 * grp;
            if (ch0 < a.cout) {
              float4 y = make_float4(acc[rg][t][0], acc[rg][t][1], acc[rg][t][2], acc[rg][t][3]);
              if (affine) {  // (scale defaults to 1: fma(y, 1, shift) == y + shift exactly)
                const float4 sc = *reinterpret_cast<const float4*>(vec + 16 * t + 4 * grp);
                const float4 sh = *reinterpret_cast<const float4*>(vec + 128 + 16 * t + 4 * grp);
                y.x = __fmaf_rn(y.x, sc.x, sh.x); y.y = __fmaf_rn(y.y, sc.y, sh.y);
                y.z = __fmaf_rn(y.z, sc.z, sh.z); y.w = __fmaf_rn(y.w, sc.w, sh.w);
              }
              if (a.residual) {
                y.x = __fadd_rn(y.x, rs[tt].x); y.y = __fadd_rn(y.y, rs[tt].y);
                y.z = __fadd_rn(y.z, rs[tt].z); y.w = __fadd_rn(y.w, rs[tt].w);
              }
              if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
              *reinterpret_cast<float4*>(a.out + row * a.cout + ch0) = y;
            }
          }
        }
      }
    }
  }
}

// folds the offset-split slices in order and applies the epilogue
__global__ void __launch_bounds__(256) scs_fold_kernel(ScsArgs a) {
  const int64_t total4 = a.m_out * (a.cout / 4);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / (a.cout / 4);
    const int col = (int)(t - o * (a.cout / 4)) * 4;
    float4 y = *reinterpret_cast<const float4*>(a.partial + o * a.cout + col);
    for (int z = 1; z < a.ksplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(a.partial + ((int64_t)z * a.m_out + o) * a.cout + col);
      y.x = __fadd_rn(y.x, v.x); y.y = __fadd_rn(y.y, v.y); y.z = __fadd_rn(y.z, v.z); y.w = __fadd_rn(y.w, v.w);
    }
    if (a.scale) {
      const float4 sc = *reinterpret_cast<const float4*>(a.scale + col), sh = *reinterpret_cast<const float4*>(a.shift + col);
      y.x = __fmaf_rn(y.x, sc.x, sh.x); y.y = __fmaf_rn(y.y, sc.y, sh.y); y.z = __fmaf_rn(y.z, sc.z, sh.z); y.w = __fmaf_rn(y.w, sc.w, sh.w);
    } else if (a.shift) {
      const float4 sh = *reinterpret_cast<const float4*>(a.shift + col);
      y.x = __fadd_rn(y.x, sh.x); y.y = __fadd_rn(y.y, sh.y); y.z = __fadd_rn(y.z, sh.z); y.w = __fadd_rn(y.w, sh.w);
    }
    if (a.residual) {
      const float4 rs = *reinterpret_cast<const float4*>(a.residual + o * a.cout + col);
      y.x = __fadd_rn(y.x, rs.x); y.y = __fadd_rn(y.y, rs.y); y.z = __fadd_rn(y.z, rs.z); y.w = __fadd_rn(y.w, rs.w);
    }
    if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    *reinterpret_cast<float4*>(a.out + o * a.cout + col) = y;
  }
}

}  // namespace fsf

using namespace fsf;

// how many ways to split the chunk sequence so that ~all 768 workgroup slots (3 per CU) have work
static int scs_ksplit(int64_t m_out, int cin, int cout, int kvol) {
  const int64_t wgs = ((m_out + SCS_ROWS - 1) / SCS_ROWS) * ((cout + 127) / 128);
  const int nchunks = kvol * ((cin + SCS_KC - 1) / SCS_KC);
  int64_t s = (256 * SCS_WPS + wgs - 1) / wgs;
  if (wgs * 2 > 256 * SCS_WPS) s = 1;  // more than half the slots busy already: the fold pass would cost more than it gains
  if (s > nchunks / 4) s = nchunks / 4;
  if (s > 32) s = 32;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t fsf_spconv_split_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout, int32_t kvol) {
  if (m_out <= 0 || cin < 1 || cout < 1 || kvol < 1) return 256;
  const int s = scs_ksplit(m_out, cin, cout, kvol);
  return s > 1 ? fsf_align_up((int64_t)s * m_out * cout * 4, 256) : 256;
}

static int scs_tiles(int cout) { return cout <= 64 ? 4 : 8; }
static int scs_slices(int cout) { return (cout + 127) / 128; }

extern "C" int64_t fsf_spconv_split_weight_bytes(int32_t kvol, int32_t cin, int32_t cout) {
  if (kvol < 1 || cin < 1 || cout < 1) return 0;
  const int64_t nkc = (cin + SCS_KC - 1) / SCS_KC;
  return (int64_t)scs_slices(cout) * kvol * nkc * scs_tiles(cout) * 3 * 64 * 16;
}

extern "C" int fsf_spconv_prepare_weight_split(const float* weight, int32_t kvol, int32_t cin, int32_t cout, void* planes,
                                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !planes || kvol < 1 || cin < 1 || cout < 1) return FSF_ERR_INVALID_ARG;
  const int T = scs_tiles(cout), nkc = (cin + SCS_KC - 1) / SCS_KC, nslice = scs_slices(cout);
  const int64_t total = (int64_t)nslice * kvol * nkc * T * 64;
  hipLaunchKernelGGL(scs_prepare_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, weight, (int)kvol, (int)cin,
                     (int)cout, T, nkc, nslice, (uint4*)planes);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_spconv_forward_split(const float* feat, int64_t m_in, int32_t cin, const void* planes, int32_t kvol,
                                        int32_t cout, const int32_t* nbr, int64_t m_out, const float* scale,
                                        const float* shift, const float* residual, int32_t relu, float* out, void* workspace,
                                        int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m_in < 0 || m_out < 0 || cin < 1 || cout < 1 || kvol < 1 || !planes || (scale && !shift) ||
      (m_out > 0 && (!nbr || !out)) || (m_in > 0 && !feat))
    return FSF_ERR_INVALID_ARG;
  if ((cin % 4) != 0 || (cout % 4) != 0 || ((uintptr_t)feat % 16) != 0 || ((uintptr_t)out % 16) != 0) return FSF_ERR_UNSUPPORTED;
  if (m_out == 0) return FSF_OK;
  if (m_in == 0) return FSF_ERR_INVALID_ARG;  // (a missing neighbour reads row 0)
  const int ksplit = scs_ksplit(m_out, cin, cout, kvol);
  if (ksplit > 1 && (!workspace || workspace_bytes < fsf_spconv_split_workspace_bytes(m_out, cin, cout, kvol))) return FSF_ERR_WORKSPACE;
  ScsArgs a{feat, (const uint4*)planes, nbr, scale, shift, residual, out, (float*)workspace, m_in, m_out,
            (int)cin, (int)cout, (int)kvol, (int)relu, ksplit, 0, 0, nullptr};
  const int64_t nblk = (m_out + SCS_ROWS - 1) / SCS_ROWS;
  const int nslice = scs_slices(cout);
  int64_t gx = (256 * SCS_WPS + nslice * ksplit - 1) / (nslice * ksplit);
  if (gx > nblk) gx = nblk;
  dim3 grid((unsigned)gx, (unsigned)nslice, (unsigned)ksplit);
  if (gx > 1 && nslice * ksplit >= 8) {  // several row blocks stream the same chunks and there are groups for every XCD
    a.xcd_lanes = (int)gx;
    a.nslice = nslice;
    grid = dim3((unsigned)(8 * gx * ((nslice * ksplit + 7) / 8)), 1, 1);
  }
#define FSF_SCS(T_)                                                                                                     \
  do {                                                                                                                 \
    constexpr size_t smem = (size_t)2 * T_ * 3 * 64 * 16 + 1024;                                                       \
    static std::atomic<uint64_t> attr_done{0};                                                                                      \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)spconv_fwd_split_kernel<T_>, (int)smem, attr_done));                                                                                                                  \
    hipLaunchKernelGGL((spconv_fwd_split_kernel<T_>), grid, dim3(SCS_NW * 64), smem, stream, a);                       \
  } while (0)
  if (scs_tiles(cout) == 4) FSF_SCS(4);
  else FSF_SCS(8);
#undef FSF_SCS
  if (ksplit > 1)
    hipLaunchKernelGGL(scs_fold_kernel, dim3(fsf_stream_grid(m_out * (cout / 4), 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ---- K9b-XP entry points -----------------------------------------------------------------------------------------------------
extern "C" int64_t fsf_spconv_split_weight_f16_bytes(int32_t kvol, int32_t cin, int32_t cout) {
  if (kvol < 1 || cin < 1 || cout < 1) return 0;
  const int64_t nkc = (cin + SCS_KC - 1) / SCS_KC;
  return 256 + (int64_t)scs_slices(cout) * kvol * nkc * scs_tiles(cout) * 2 * 64 * 16;
}

extern "C" int fsf_spconv_prepare_weight_split_f16(const float* weight, int32_t kvol, int32_t cin, int32_t cout, void* planes,
                                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !planes || kvol < 1 || cin < 1 || cout < 1) return FSF_ERR_INVALID_ARG;
  if (((uintptr_t)planes % 16) != 0) return FSF_ERR_UNSUPPORTED;
  const int T = scs_tiles(cout), nkc = (cin + SCS_KC - 1) / SCS_KC, nslice = scs_slices(cout);
  FSF_HIP_TRY(hipMemsetAsync(planes, 0, 256, stream));
  const int64_t nw = (int64_t)kvol * cin * cout;
  // (at most 256 workgroups: each ends in ONE atomic on the same word — 2 048 of them took 25 us for a 1 M-element weight)
  hipLaunchKernelGGL(scs_weight_absmax_kernel, dim3(std::min(fsf_stream_grid(nw, 256), 256)), dim3(256), 0, stream, weight, nw, (unsigned*)planes);
  FSF_LAUNCH_CHECK();
  const int64_t total = (int64_t)nslice * kvol * nkc * T * 64;
  hipLaunchKernelGGL(scs_prepare_f16_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, weight, (int)kvol, (int)cin,
                     (int)cout, T, nkc, nslice, (float*)planes, (uint4*)planes + 16);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_spconv_forward_split_planes(const void* feat_planes, const float* feat_inv_scales, int64_t m_in, int32_t cin,
                                               const void* planes, int32_t kvol, int32_t cout, const int32_t* nbr, int64_t m_out,
                                               const float* scale, const float* shift, const float* residual, int32_t relu,
                                               float* out, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m_in < 0 || m_out < 0 || cin < 1 || cout < 1 || kvol < 1 || !planes || (scale && !shift) ||
      (m_out > 0 && (!nbr || !out)) || (m_in > 0 && (!feat_planes || !feat_inv_scales)))
    return FSF_ERR_INVALID_ARG;
  // whole 32-cin chunks (the plane rows are cin * 4 bytes: every address of the fp32 form stands), 128-channel tiles
  if ((cin % 32) != 0 || (cout % 4) != 0 || scs_tiles(cout) != 8 || ((uintptr_t)feat_planes % 16) != 0 || ((uintptr_t)out % 16) != 0 ||
      ((uintptr_t)planes % 16) != 0)
    return FSF_ERR_UNSUPPORTED;
  if (m_out == 0) return FSF_OK;
  if (m_in == 0) return FSF_ERR_INVALID_ARG;
  const int ksplit = scs_ksplit(m_out, cin, cout, kvol);
  if (ksplit > 1 && (!workspace || workspace_bytes < fsf_spconv_split_workspace_bytes(m_out, cin, cout, kvol))) return FSF_ERR_WORKSPACE;
  ScsArgs a{(const float*)feat_planes, (const uint4*)planes, nbr, scale, shift, residual, out, (float*)workspace, m_in, m_out,
            (int)cin, (int)cout, (int)kvol, (int)relu, ksplit, 0, 0, feat_inv_scales};
  const int64_t nblk = (m_out + SCS_ROWS - 1) / SCS_ROWS;
  const int nslice = scs_slices(cout);
  int64_t gx = (256 * SCS_WPS + nslice * ksplit - 1) / (nslice * ksplit);
  if (gx > nblk) gx = nblk;
  dim3 grid((unsigned)gx, (unsigned)nslice, (unsigned)ksplit);
  if (gx > 1 && nslice * ksplit >= 8) {
    a.xcd_lanes = (int)gx;
    a.nslice = nslice;
    grid = dim3((unsigned)(8 * gx * ((nslice * ksplit + 7) / 8)), 1, 1);
  }
  constexpr size_t smem = (size_t)2 * 8 * 2 * 64 * 16 + 1024;
  static std::atomic<uint64_t> attr_done{0};
  FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)spconv_fwd_split_kernel<8, true>, (int)smem, attr_done));
  hipLaunchKernelGGL((spconv_fwd_split_kernel<8, true>), grid, dim3(SCS_NW * 64), smem, stream, a);
  if (ksplit > 1) hipLaunchKernelGGL(scs_fold_kernel, dim3(fsf_stream_grid(m_out * (cout / 4), 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
