// K22: per-point Linear (+ bias) -> LayerNorm | eval-BatchNorm affine -> ReLU | GELU in one pass, fp32 in / fp32 out.
// See include/fsf_hip.h.
//
// Replaces (inference): the [Linear, norm, act] blocks that build_mlp (projects/mmdet3d_plugin/ops/sst_ops.py:808-833) and
// DynamicVFELayer [UNVENDORED] put on every point: a dense fp32 GEMM on the library (70 % of the fp32 matrix-pipe peak
// at best) + one more pass over the [n, C] activations for norm + act.
//
// The fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 157 TFLOP/s) is 1/16 of the bf16 one, so the product is formed on
// v_mfma_f32_16x16x32_bf16 from an EXACT three-way split of both operands: x = hi + mid + lo with hi = x truncated to
// its top 8 significant bits (a bf16), mid = the same of x - hi, lo = x - hi - mid (8 bits left: a bf16, exactly) —
// three 8-bit pieces cover the 24-bit fp32 significand, nothing is rounded away.  a*b is then the sum of nine bf16
// products (each exact in the fp32 accumulator's input); the six leading ones are issued, the dropped mid*lo, lo*mid,
// lo*lo terms are < 2^-23 of |a*b| together — the size of ONE fp32 rounding — and the accumulation is fp32 as on the
// fp32 pipe.  6 x ~17 cycles per 16x16x32 block instead of 8 x 32: the matrix phase is 2.5x shorter at fp32 accuracy
// (tests: error vs float64 no larger than torch's fp32 F.linear).  This is not a reduced-precision mode.
//
// Transposed formulation (as K21): out^T[channel, row] = W[channel, k] x X^T[k, row]; the weights are the A operand
// (pre-split once per layer into fragment-ordered bf16 planes by fsf_linear_prepare_weight, streamed through LDS in
// 64-wide k chunks, double-buffered by LDS-DMA), 16 point rows are the B operand (loaded straight from HBM in operand
// layout — lane (row, g) reads 32 contiguous bytes per k step — and split in registers).  A lane ends up with channels
// 16 t + 4 g + r of its row: LayerNorm is an in-lane sum + two shuffles, the result leaves as 16-byte stores.
#include <stdlib.h>
#include <algorithm>

#include "common.h"

namespace fsf {

typedef __bf16 lna_bf16x8 __attribute__((ext_vector_type(8)));
typedef float lna_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lna_u32x4 __attribute__((ext_vector_type(4)));

constexpr int LNA_KC = 32;        // k per LDS chunk (one MFMA k step)
constexpr int LNA_NW = 4;         // waves per workgroup (two workgroups per CU: one wave of each per SIMD)
constexpr int LNA_RG = 2;         // 16-row groups per wave and iteration
#ifndef LNA_WPS
#define LNA_WPS 3                 // waves per SIMD the register budget is set for (workgroups per CU)
#endif
constexpr int LNA_ROWS = LNA_NW * LNA_RG * 16;  // rows per workgroup iteration (4-wave workgroups)

struct LnaArgs {
  const float* x; int64_t x_stride; int k;
  const uint4* planes;  // [slice][KP/32][T][3][64 lanes] x 16 B (slice = 128 output channels)
  const float *bias, *gamma, *beta;
  float eps; int norm, act;  // norm 0 none / 1 LayerNorm / 2 affine (y * gamma + beta); act 0 / 1 ReLU / 2 GELU(erf)
  float* out; int64_t out_stride;
  int64_t n; int c;
  // optional per-row addend before the norm: row_add[row_add_index[row]][c] — the right half of a
  // `cat([point_feats, group_feats[inv]], 1) @ W^T` product, applied to the groups once instead of to every point
  const float* row_add; const int64_t* row_add_index; int64_t row_add_stride;
  // output channels per blockIdx.y slice (128 unless "sliced": independent layers side by side, one per slice), the width a
  // LayerNorm spans (c, or the slice width), and the column offset between the inputs of consecutive slices
  int slice_w, norm_w; int64_t x_slice_off;
};

__device__ __forceinline__ float lna_row_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// exact three-way split of 8 floats into bf16 planes (two bf16 per dword, element 2j in the low half)
__device__ __forceinline__ void lna_split8(const float (&v)[8], lna_u32x4& hi, lna_u32x4& mid, lna_u32x4& lo) {
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

// GELU with erf by Abramowitz & Stegun 7.1.26, branch-free (|error| <= 1.5e-7 absolute — float epsilon; the same form as
// K21's, see sir_input.hip)
__device__ __forceinline__ float lna_gelu(float y) {
  const float u = fabsf(y) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float pe = p * t * __builtin_amdgcn_exp2f(u * u * -1.4426950408889634f);
  return 0.5f * y * (y < 0.0f ? pe : 2.0f - pe);
}

__device__ __forceinline__ float lna_act(float y, int act) {
  if (act == 1) return fmaxf(y, 0.0f);
  if (act == 2) return lna_gelu(y);
  return y;
}

// weight [c, k] fp32 -> fragment-ordered bf16 planes (zero padded to T tiles x KP)
__global__ void __launch_bounds__(256)
    lna_prepare_kernel(const float* __restrict__ w, int k, int c, int T, int nkc, int nslice, int slice_w, uint4* planes) {
  const int64_t total = (int64_t)nslice * nkc * T * 64;  // (slice, chunk, tile, lane): three 16-byte fragments each
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const int t = (int)((idx >> 6) % T);
    const int kc = (int)(((idx >> 6) / T) % nkc);
    const int slice = (int)((idx >> 6) / ((int64_t)T * nkc));
    const int lc = 16 * t + (lane & 15), col = slice_w * slice + lc, k0 = kc * LNA_KC + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (lc < slice_w && col < c && k0 + e < k) ? w[(int64_t)col * k + k0 + e] : 0.0f;
    lna_u32x4 hi, mid, lo;
    lna_split8(v, hi, mid, lo);
    uint4* dst = planes + (((int64_t)slice * nkc + kc) * T + t) * 3 * 64 + lane;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
    dst[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// epilogue of one row block: lane (row, g) holds channels ch_base + 16 t + 4 g + r of its row
// bias | gamma | beta of the 128-channel slice at ch_base -> LDS (defaults 0 | 1 | 0 where absent or beyond c)
__device__ __forceinline__ void lna_stage_vectors(const LnaArgs& a, int ch_base, float* vec) {
  for (int t = threadIdx.x; t < 384; t += blockDim.x) {
    const int which = t >> 7, ch = ch_base + (t & 127);
    const float* src = which == 0 ? a.bias : (a.norm != 0 ? (which == 1 ? a.gamma : a.beta) : nullptr);
    vec[t] = (src && (t & 127) < a.slice_w && ch < a.c) ? src[ch] : (which == 1 ? 1.0f : 0.0f);
  }
  __syncthreads();
}

// `vec` = this slice's bias | gamma | beta, 128 floats each, staged in LDS once per workgroup: as ordinary global loads in
// here every one of them was followed by the `vmcnt(0)` hipcc emits at the first use of a load beside an LDS-DMA — 24-48
// serialized L2 round trips per row block (and a drain of the next block's prefetch each time).
template <int T>
__device__ __forceinline__ void lna_epilogue(const LnaArgs& a, lna_f32x4 (&acc)[LNA_RG][T], int64_t row0, int ch_base, int rowl,
                                             int grp, const float* vec) {
  const float inv_c = 1.0f / (float)a.norm_w;
#pragma unroll
  for (int rg = 0; rg < LNA_RG; ++rg) {
    const int64_t row = row0 + 16 * rg + rowl;
    float mean = 0.0f, rstd = 1.0f;
    if (a.bias) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(vec + 16 * t + 4 * grp);  // (0 beyond c)
        acc[rg][t][0] += b.x; acc[rg][t][1] += b.y; acc[rg][t][2] += b.z; acc[rg][t][3] += b.w;
      }
    }
    if (a.row_add) {  // all loads of the row first, then the adds: one wait instead of one per tile
      const int64_t row_c = row0 + 16 * rg + rowl;
      const float* add = a.row_add + a.row_add_index[row_c < a.n ? row_c : a.n - 1] * a.row_add_stride;
      constexpr int TB = T < 4 ? T : 4;  // four loads in flight per wait
#pragma unroll
      for (int t0 = 0; t0 < T; t0 += TB) {
        float4 b[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int lc0 = 16 * (t0 + t) + 4 * grp, ch0 = ch_base + lc0;
          b[t] = *reinterpret_cast<const float4*>(add + (lc0 < a.slice_w && ch0 < a.c ? ch0 : 0));
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          if (16 * (t0 + t) + 4 * grp < a.slice_w && ch_base + 16 * (t0 + t) + 4 * grp < a.c) {
            acc[rg][t0 + t][0] += b[t].x; acc[rg][t0 + t][1] += b[t].y; acc[rg][t0 + t][2] += b[t].z; acc[rg][t0 + t][3] += b[t].w;
          }
        }
      }
    }
    if (a.norm == 1) {  // LayerNorm over the c channels (channels >= c are exactly 0: zero weights, no bias)
      float s = 0.0f;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[rg][t][r];
      mean = lna_row_sum(s) * inv_c;
      float q = 0.0f;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = (16 * t + 4 * grp + r < a.slice_w && ch_base + 16 * t + 4 * grp + r < a.c) ? acc[rg][t][r] - mean : 0.0f;
          q += d * d;
        }
      rstd = rsqrtf(lna_row_sum(q) * inv_c + a.eps);
    }
    if (row < a.n) {
      float* orow = a.out + row * a.out_stride;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int ch0 = ch_base + 16 * t + 4 * grp;
        if (16 * t + 4 * grp < a.slice_w && ch0 < a.c) {
          const float4 g = *reinterpret_cast<const float4*>(vec + 128 + 16 * t + 4 * grp);  // (1 / 0 without a norm)
          const float4 b = *reinterpret_cast<const float4*>(vec + 256 + 16 * t + 4 * grp);
          float4 y;
          y.x = lna_act((acc[rg][t][0] - mean) * rstd * g.x + b.x, a.act);
          y.y = lna_act((acc[rg][t][1] - mean) * rstd * g.y + b.y, a.act);
          y.z = lna_act((acc[rg][t][2] - mean) * rstd * g.z + b.z, a.act);
          y.w = lna_act((acc[rg][t][3] - mean) * rstd * g.w + b.w, a.act);
#ifndef FSF_ABL_LNA_NO_STORE
          *reinterpret_cast<float4*>(orow + ch0) = y;
#else
          if (y.x == 123.456f) *reinterpret_cast<float4*>(orow + ch0) = y;
#endif
        }
      }
    }
  }
}

// NW waves per workgroup.  A weight chunk enters the CU once per WORKGROUP and chunk (LDS-DMA), so three 4-wave workgroups per CU
// take the same 24 KB in three times per 128 rows each; ONE 12-wave workgroup per CU (same 12 waves, same registers) takes it in
// once per 384 rows.  Measured (round 3, same box): no faster in isolation (510 k x 256 -> 128: 277 vs 279 us; k = 128 .. 180: 5-15 %
// SLOWER — a barrier over twelve waves per chunk) and 7 % slower in the frame (a 768-thread workgroup shuts the other stream's kernels
// out of its CU) — so the weight stream is not what this kernel waits for.  Kept behind FSF_K22_WIDE_MIN_ROWS=<rows> (default: never).
template <int T, int NW>  // 16-channel tiles (c <= 16 T)
__global__ void __launch_bounds__(NW * 64, NW == 4 ? LNA_WPS : 3) linear_norm_act_kernel(LnaArgs a) {
  constexpr int LNA_NW = NW;
  constexpr int LNA_ROWS = NW * LNA_RG * 16;
  constexpr int CHUNK_U4 = T * 3 * 64;  // uint4 per weight chunk
  extern __shared__ __attribute__((aligned(16))) char lna_smem[];
  uint4* wbuf = reinterpret_cast<uint4*>(lna_smem);  // [2][CHUNK_U4], then 384 floats of per-channel vectors
  float* vec = reinterpret_cast<float*>(wbuf + 2 * CHUNK_U4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = lane & 15, grp = lane >> 4;
  const int nkc = (a.k + LNA_KC - 1) / LNA_KC;
  const int64_t nblk = (a.n + LNA_ROWS - 1) / LNA_ROWS;
  // more than 128 output channels: gridDim.y slices of 128, each an independent [rows, 128] product (no LayerNorm then:
  // its statistics span the slices; the caller runs fsf_norm_act on the result)
  const int ch_base = a.slice_w * (int)blockIdx.y;
  const uint4* planes = a.planes + (int64_t)blockIdx.y * nkc * CHUNK_U4;

  // weight chunk kc -> LDS buffer by LDS-DMA: fragment order in HBM == fragment order in LDS, 1 KB per wave instruction
  auto stage_w = [&](int kc, int buf) {
#ifdef FSF_ABL_LNA_NO_W  // ablation: the weight chunks are staged once (chunk 0 into both buffers), never again
    if (kc > 1) return;
#endif
    const float* src = reinterpret_cast<const float*>(planes + (int64_t)kc * CHUNK_U4);
    float* dst = reinterpret_cast<float*>(wbuf + buf * CHUNK_U4);
    for (int u = wave * 64; u < CHUNK_U4; u += LNA_NW * 64)
      __builtin_amdgcn_global_load_lds(src + 4 * (u + lane), dst + 4 * u, 16, 0, 0);
  };

  // The (row block, k chunk) sequence of a workgroup is ONE software pipeline: during the last chunk of a block the first
  // weight chunk and the first x chunk of the NEXT block are already requested, so the epilogue (norm, activation,
  // stores) runs with them in flight instead of every block starting with an exposed HBM round trip.
  const float* xrow[LNA_RG];
  auto set_rows = [&](int64_t blk) {
#pragma unroll
    for (int rg = 0; rg < LNA_RG; ++rg) {
      int64_t r = blk * LNA_ROWS + (int64_t)wave * (LNA_RG * 16) + 16 * rg + rowl;
      if (r >= a.n) r = a.n - 1;  // rows past n repeat the last one (finite, never stored)
#ifdef FSF_ABL_LNA_X_HOT  // ablation: every x load hits one of 4096 cache-resident rows
      r &= 4095;
#endif
      xrow[rg] = a.x + r * a.x_stride + (int64_t)blockIdx.y * a.x_slice_off;
    }
  };
  // raw x of one chunk: [row group][8 floats]; the NEXT chunk is requested while this one is split and multiplied.
  // Always exactly two 16-byte loads per row group: offsets past the row are clamped into it (x_stride is a multiple
  // of 4 and >= k, so a quad that holds any column < k is never clamped) and the columns >= k are zeroed afterwards
  // (what follows the row in memory may be NaN).
  const int last_quad = (int)a.x_stride - 4 - (int)((int64_t)blockIdx.y * a.x_slice_off);  // (relative to this slice's first column)
  auto load_x = [&](int kc, float (&v)[LNA_RG][8]) {
#pragma unroll
    for (int rg = 0; rg < LNA_RG; ++rg) {
      const int kq = kc * LNA_KC + 8 * grp;  // this lane's 8 k values
      const float4 p = *reinterpret_cast<const float4*>(xrow[rg] + min(kq, last_quad));
      const float4 q = *reinterpret_cast<const float4*>(xrow[rg] + min(kq + 4, last_quad));
      v[rg][0] = p.x; v[rg][1] = p.y; v[rg][2] = p.z; v[rg][3] = p.w;
      v[rg][4] = q.x; v[rg][5] = q.y; v[rg][6] = q.z; v[rg][7] = q.w;
    }
  };
  lna_stage_vectors(a, ch_base, vec);
  float xc[LNA_RG][8];
  int buf = 0;
  if ((int64_t)blockIdx.x < nblk) {
    set_rows(blockIdx.x);
    load_x(0, xc);
    stage_w(0, 0);
  }
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * LNA_ROWS + (int64_t)wave * (LNA_RG * 16);
    lna_f32x4 acc[LNA_RG][T];
#pragma unroll
    for (int rg = 0; rg < LNA_RG; ++rg)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[rg][t] = lna_f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef FSF_ABL_LNA_NO_XBLK  // ablation: every row block starts with an exposed load (the kernel before the cross-block pipeline)
    if (blk != (int64_t)blockIdx.x) {
      set_rows(blk);
      load_x(0, xc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage_w(0, buf);
    }
#endif
    for (int kc = 0; kc < nkc; ++kc, buf ^= 1) {
      // split the chunk that arrived while the previous one was multiplied; its registers then take the next prefetch
      lna_u32x4 xh[LNA_RG], xm[LNA_RG], xl[LNA_RG];
      if ((kc + 1) * LNA_KC > a.k) {  // (uniform: the last chunk of a k that is not a multiple of 32)
#pragma unroll
        for (int rg = 0; rg < LNA_RG; ++rg)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kc * LNA_KC + 8 * grp + e >= a.k) xc[rg][e] = 0.0f;
      }
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) lna_split8(xc[rg], xh[rg], xm[rg], xl[rg]);
      // this chunk's weights (DMA issued one iteration ago, before that iteration's MFMAs) have landed; the raw barrier
      // carries no fence, so nothing else is drained with them
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // + every wave is done reading buffer buf^1
      asm volatile("" ::: "memory");
      if (kc + 1 < nkc) {
        stage_w(kc + 1, buf ^ 1);
        load_x(kc + 1, xc);
      }
#ifndef FSF_ABL_LNA_NO_XBLK
      else if (blk + gridDim.x < nblk) {  // first chunk of the next row block
        stage_w(0, buf ^ 1);
        set_rows(blk + gridDim.x);
        load_x(0, xc);
      }
#endif
      const uint4* wc = wbuf + buf * CHUNK_U4;
      // Two channel tiles x LNA_RG row groups = 4 independent accumulators per product term: consecutive MFMAs never hit
      // the same accumulator
#pragma unroll
      for (int t = 0; t < T; t += 2) {
        lna_bf16x8 wfr[2][3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const uint4* wf = wc + ((t + tt) * 3) * 64 + lane;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) wfr[tt][pl] = __builtin_bit_cast(lna_bf16x8, wf[64 * pl]);
        }
        // (weight plane, x plane) of the six leading cross terms, small ones first
        constexpr int TERM_W[6] = {2, 0, 1, 1, 0, 0};
        constexpr int TERM_X[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int rg = 0; rg < LNA_RG; ++rg) {
              const lna_u32x4 xb = TERM_X[term] == 0 ? xh[rg] : TERM_X[term] == 1 ? xm[rg] : xl[rg];
#ifndef FSF_ABL_LNA_NO_MFMA
              acc[rg][t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[tt][TERM_W[term]], __builtin_bit_cast(lna_bf16x8, xb),
                                                                        acc[rg][t + tt], 0, 0, 0);
#else  // ablation: one VALU op per product term keeps the operand loads alive without the matrix pipe
              acc[rg][t + tt][term & 3] += __uint_as_float(xb[term & 3] ^ __builtin_bit_cast(lna_u32x4, wfr[tt][TERM_W[term]])[term & 3]);
#endif
            }
      }
    }
    lna_epilogue<T>(a, acc, row0, ch_base, rowl, grp, vec);
  }
}

}  // namespace fsf

using namespace fsf;

// 16-channel tiles of the launched kernel variant (the prepared weight is laid out for exactly this count)
static int lna_tiles(int c) {
  const int t = (c + 15) / 16;
  return t <= 2 ? 2 : (t <= 4 ? 4 : 8);
}

static int lna_slices(int c) { return (c + 127) / 128; }

extern "C" int64_t fsf_linear_prepared_weight_bytes(int32_t k, int32_t c) {
  if (k < 1 || c < 1) return 0;
  const int64_t nkc = (k + LNA_KC - 1) / LNA_KC;
  return lna_slices(c) * nkc * lna_tiles(c) * 3 * 64 * 16;
}

extern "C" int fsf_linear_prepare_weight(const float* weight, int32_t k, int32_t c, void* planes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !planes || k < 1 || c < 1) return FSF_ERR_INVALID_ARG;
  const int T = lna_tiles(c), nkc = (k + LNA_KC - 1) / LNA_KC, nslice = lna_slices(c);
  const int64_t total = (int64_t)nslice * nkc * T * 64;
  hipLaunchKernelGGL(lna_prepare_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, weight, (int)k, (int)c, T, nkc,
                     nslice, 128, (uint4*)planes);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// "sliced": nslice independent layers of slice_c output channels each, side by side in one launch (weight rows
// [s * slice_c, (s + 1) * slice_c) belong to layer s)
extern "C" int64_t fsf_linear_prepared_weight_sliced_bytes(int32_t k, int32_t nslice, int32_t slice_c) {
  if (k < 1 || nslice < 1 || slice_c < 1 || slice_c > 128) return 0;
  const int64_t nkc = (k + LNA_KC - 1) / LNA_KC;
  return (int64_t)nslice * nkc * lna_tiles(slice_c) * 3 * 64 * 16;
}

extern "C" int fsf_linear_prepare_weight_sliced(const float* weight, int32_t k, int32_t nslice, int32_t slice_c, void* planes,
                                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !planes || k < 1 || nslice < 1 || slice_c < 1 || slice_c > 128) return FSF_ERR_INVALID_ARG;
  const int T = lna_tiles(slice_c), nkc = (k + LNA_KC - 1) / LNA_KC;
  const int64_t total = (int64_t)nslice * nkc * T * 64;
  hipLaunchKernelGGL(lna_prepare_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, weight, (int)k,
                     (int)(nslice * slice_c), T, nkc, (int)nslice, (int)slice_c, (uint4*)planes);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

static int lna_launch(const LnaArgs& a_in, int nslice, hipStream_t stream) {
  LnaArgs a = a_in;
  const int T = lna_tiles(a.slice_w < a.c ? a.slice_w : a.c);
  const int rows = LNA_NW * LNA_RG * 16;
  const int64_t nblk = (a.n + rows - 1) / rows;
  int64_t gx = (256 * LNA_WPS + nslice - 1) / nslice;
  if (gx > nblk) gx = nblk;
  const dim3 grid((unsigned)gx, (unsigned)nslice);
#define FSF_LNA(T_, NW_)                                                                                                \
  do {                                                                                                                 \
    constexpr size_t smem = (size_t)2 * T_ * 3 * 64 * 16 + 384 * 4;                                                    \
    static std::atomic<uint64_t> attr_done{0};                                                                         \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)linear_norm_act_kernel<T_, NW_>, (int)smem, attr_done));          \
    hipLaunchKernelGGL((linear_norm_act_kernel<T_, NW_>), grid, dim3(NW_ * 64), smem, stream, a);                      \
  } while (0)
  if (T == 2) FSF_LNA(2, 4);
  else if (T == 4) FSF_LNA(4, 4);
  else FSF_LNA(8, 4);
#undef FSF_LNA
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_linear_norm_act_sliced(const float* x, int64_t n, int32_t k, int64_t x_stride, int64_t x_slice_offset,
                                          const void* planes, int32_t nslice, int32_t slice_c, const float* bias, int32_t norm,
                                          const float* gamma, const float* beta, float eps, int32_t act, float* out,
                                          int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || k < 1 || nslice < 1 || slice_c < 1 || !planes || norm < 0 || norm > 2 || act < 0 || act > 2 ||
      (norm != 0 && (!gamma || !beta)) || (n > 0 && (!x || !out)) || x_slice_offset < 0)
    return FSF_ERR_INVALID_ARG;
  if (slice_c > 128 || (slice_c % 4) != 0 || (x_stride % 4) != 0 || (x_slice_offset % 4) != 0 || (out_stride % 4) != 0 ||
      ((uintptr_t)x % 16) != 0 || ((uintptr_t)out % 16) != 0)
    return FSF_ERR_UNSUPPORTED;
  if (x_stride < (int64_t)(nslice - 1) * x_slice_offset + k || out_stride < (int64_t)nslice * slice_c) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  LnaArgs a{x, x_stride, (int)k, (const uint4*)planes, bias, gamma, beta, eps, (int)norm, (int)act, out, out_stride, n,
            (int)(nslice * slice_c), nullptr, nullptr, 0, (int)slice_c, (int)slice_c, x_slice_offset};
  return lna_launch(a, nslice, stream);
}

extern "C" int fsf_linear_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                           const float* bias, const float* row_add, const int64_t* row_add_index,
                                           int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta,
                                           float eps, int32_t act, float* out, int64_t out_stride, void* stream_);

extern "C" int fsf_linear_norm_act(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                   const float* bias, int32_t norm, const float* gamma, const float* beta, float eps,
                                   int32_t act, float* out, int64_t out_stride, void* stream_) {
  return fsf_linear_norm_act_grouped(x, n, k, x_stride, planes, c, bias, nullptr, nullptr, 0, norm, gamma, beta, eps, act, out,
                                     out_stride, stream_);
}

extern "C" int fsf_linear_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                           const float* bias, const float* row_add, const int64_t* row_add_index,
                                           int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta,
                                           float eps, int32_t act, float* out, int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((row_add == nullptr) != (row_add_index == nullptr)) return FSF_ERR_INVALID_ARG;
  if (row_add && ((row_add_stride % 4) != 0 || row_add_stride < c || ((uintptr_t)row_add % 16) != 0)) return FSF_ERR_UNSUPPORTED;
  if (n < 0 || k < 1 || c < 1 || !planes || norm < 0 || norm > 2 || act < 0 || act > 2 || (norm != 0 && (!gamma || !beta)) ||
      (n > 0 && (!x || !out)))
    return FSF_ERR_INVALID_ARG;
  // 16-byte row accesses: x rows and out rows must be 16-byte aligned, c a multiple of 4
  if ((c > 128 && norm == 1) || (c % 4) != 0 || (x_stride % 4) != 0 || (out_stride % 4) != 0 || ((uintptr_t)x % 16) != 0 ||
      ((uintptr_t)out % 16) != 0)
    return FSF_ERR_UNSUPPORTED;
  if (x_stride < k || out_stride < c) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  LnaArgs a{x, x_stride, (int)k, (const uint4*)planes, bias, gamma, beta, eps, (int)norm, (int)act, out, out_stride, n, (int)c,
            row_add, row_add_index, row_add_stride, 128, (int)c, 0};
  return lna_launch(a, lna_slices(c), stream);
}
