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
typedef _Float16 lna_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lna_f16x4 __attribute__((ext_vector_type(4)));
typedef float lna_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lna_u32x4 __attribute__((ext_vector_type(4)));

#ifdef FSF_LNA_TIMELINE  // profiling build only (tools/profiling/lna_timeline.sh): where a workgroup's wave 0 spends its clocks
__device__ unsigned long long lna_tl[16];
struct LnaTl {
  unsigned long long last, acc[8];
  __device__ __forceinline__ void start() { last = __builtin_readcyclecounter(); for (int i = 0; i < 8; ++i) acc[i] = 0; }
  __device__ __forceinline__ void mark(int i) { const unsigned long long now = __builtin_readcyclecounter(); acc[i] += now - last; last = now; }
};
#define LNA_TL_MARK(tl, i) (tl).mark(i)
#else
struct LnaTl {};
#define LNA_TL_MARK(tl, i) ((void)0)
#endif

constexpr int LNA_KC = 32;        // k per LDS chunk (one MFMA k step)
constexpr int LNA_NW = 4;         // waves per workgroup (two workgroups per CU: one wave of each per SIMD)
constexpr int LNA_RG = 2;         // 16-row groups per wave and iteration
#ifndef LNA_WPS
#define LNA_WPS 3                 // waves per SIMD the register budget is set for (workgroups per CU)
#endif
#ifndef LNA_SEG_WPS
#define LNA_SEG_WPS 3             // ... of the K22s variants (2 = no spills at 256 registers, a third fewer waves: measured, see DESIGN)
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
  // fused segmented max (K22s): rows arrive SORTED by segment (seg_ids nondecreasing); seg_out[s, ch] = max over the rows of
  // segment s of the activated output.  seg_out must hold -inf on entry (a segment that reaches beyond one 128-row block is
  // combined with atomic max).  `out` may then be null.
  const int64_t* seg_ids; float* seg_out; int64_t seg_out_stride;
  // K22h (XP): `x` is the input in PLANE form — [row][k / 8][2][8] f16 hi | lo of x * s_row (rows_to_planes_kernel below; the layout
  // of fsf_to_planes with ONE power-of-two scale per row) —, x_inv_scale[row] = 1 / s_row, `planes` = f16 hi | lo fragments of
  // W * s_w behind a 256-byte header whose first float is 1 / s_w.  The product runs as three v_mfma_f32_16x16x32_f16 per
  // fp32-equivalent one (hi hi + hi lo + lo hi, as K9d), no split in the main loop.
  const float* x_inv_scale;
};

// max of two floats into memory, any signs, by integer atomics on the IEEE bit patterns (target initialised to -inf): a value
// >= 0 orders like a signed int above every negative pattern, a value < 0 orders inversely as an unsigned int below every
// non-negative pattern's... (min over unsigned: non-negative patterns are the smallest, so a stored non-negative survives).
__device__ __forceinline__ void lna_atomic_max(float* p, float v) {
  // (the branch is on the SIGN BIT: -0.0f compares >= 0 but its pattern is INT_MIN, which a signed max never stores)
  if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(p), __float_as_uint(v));
}

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

typedef float lna_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ lna_f32x2 lna_pk_fma(lna_f32x2 a, lna_f32x2 b, lna_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ lna_f32x2 lna_pk(float v) { return lna_f32x2{v, v}; }

// GELU: the library's one form (common.h: max(y, 0) - t 2^P(t), one transcendental per value), on TWO values per lane — gfx950 issues
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the rate of their scalar forms, and the kernel is bound by VALU issue, not by the
// matrix pipe (docs/kernels/K21_K22_linear_family.md).
__device__ __forceinline__ lna_f32x2 lna_gelu2(lna_f32x2 y) { return fsf_gelu2(y); }

__device__ __forceinline__ lna_f32x2 lna_act2(lna_f32x2 y, int act) {
  if (act == 1) return lna_f32x2{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)};
  if (act == 2) return lna_gelu2(y);
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

// ---- K22h: both operands as f16 hi | lo planes --------------------------------------------------------------------------
// power of two s with s * amax in [2^13, 2^14) and inv = 1 / s (exact; the scheme of K9c / K9d, csrc/spconv_planes.hip):
// hi = rn_f16(x s), lo = rn_f16(x s - hi): |x s - hi - lo| <= max(2^-22 |x s|, 2^-25), no f16 range hazard for any finite input
__device__ __forceinline__ void lna_pick_scale(float amax, float& s, float& inv) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  e = amax > 0.0f ? (e < -113 ? -113 : e) : 13;
  s = __uint_as_float((unsigned)(13 - e + 127) << 23);
  inv = __uint_as_float((unsigned)(e - 13 + 127) << 23);
}

__device__ __forceinline__ void lna_split8_f16(const float (&v)[8], float s, lna_u32x4& hi, lna_u32x4& lo) {
  lna_f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xs = __fmul_rn(v[e], s);
    h[e] = (_Float16)xs;
    l[e] = (_Float16)__fsub_rn(xs, (float)h[e]);
  }
  hi = __builtin_bit_cast(lna_u32x4, h);
  lo = __builtin_bit_cast(lna_u32x4, l);
}

constexpr unsigned LNA_F16_TAG = 0x4B323266u;  // "K22f": word 3 of the f16 weight planes' 256-byte header

// max |w| over the layer -> hdr[2] (bit pattern; cleared by the caller), one atomic per workgroup
__global__ void __launch_bounds__(256) lna_weight_absmax_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ hdr) {
  __shared__ float wave_max[4];
  float amax = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) amax = fmaxf(amax, fabsf(w[i]));
  amax = fsf_wave_max(amax);
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(hdr + 2, __float_as_uint(fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]))));
}

// weight [c, k] fp32 -> header (inverse scale, scale, max |w| bits) + fragment-ordered f16 hi | lo planes of W * s_w
__global__ void __launch_bounds__(256)
    lna_prepare_f16_kernel(const float* __restrict__ w, int k, int c, int T, int nkc, int nslice, int slice_w, float* __restrict__ hdr,
                           uint4* __restrict__ planes) {
  float s_w, inv_w;
  lna_pick_scale(__uint_as_float(reinterpret_cast<const unsigned*>(hdr)[2]), s_w, inv_w);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[0] = inv_w;
    hdr[1] = s_w;
    reinterpret_cast<unsigned*>(hdr)[3] = LNA_F16_TAG;  // the format's tag word: the f16-weight kernels refuse a buffer without it
  }
  const int64_t total = (int64_t)nslice * nkc * T * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const int t = (int)((idx >> 6) % T);
    const int kc = (int)(((idx >> 6) / T) % nkc);
    const int slice = (int)((idx >> 6) / ((int64_t)T * nkc));
    const int lc = 16 * t + (lane & 15), col = slice_w * slice + lc, k0 = kc * LNA_KC + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (lc < slice_w && col < c && k0 + e < k) ? w[(int64_t)col * k + k0 + e] : 0.0f;
    lna_u32x4 hi, lo;
    lna_split8_f16(v, s_w, hi, lo);
    uint4* dst = planes + (((int64_t)slice * nkc + kc) * T + t) * 2 * 64 + lane;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// fp32 rows [n, c] (optionally through LayerNorm + ReLU / GELU first: the norm pass that follows a Linear wider than 128 channels)
// -> row planes [n][c / 8][2][8] f16 + one inverse scale per row (+ the fp32 rows themselves when `out` is given).  One wave per row;
// lane l owns the 8-channel blocks l, l + 64, ... (BL of them: c <= 512 BL).
template <int BL, int NORM, int ACT>
__global__ void __launch_bounds__(256)
    rows_to_planes_kernel(const float* __restrict__ x, int64_t n, int c, int64_t x_stride, const float* __restrict__ gamma,
                          const float* __restrict__ beta, float eps, uint4* __restrict__ planes, float* __restrict__ inv_scales,
                          float* __restrict__ out, int64_t out_stride) {
  const int lane = threadIdx.x & 63;
  const int nb = c >> 3;
  const float inv_c = 1.0f / (float)c;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
    float v[BL][8];
#pragma unroll
    for (int b = 0; b < BL; ++b) {
      const int blk = lane + 64 * b;
      if (blk < nb) {
        const float4 p = *reinterpret_cast<const float4*>(x + row * x_stride + 8 * blk);
        const float4 q = *reinterpret_cast<const float4*>(x + row * x_stride + 8 * blk + 4);
        v[b][0] = p.x; v[b][1] = p.y; v[b][2] = p.z; v[b][3] = p.w; v[b][4] = q.x; v[b][5] = q.y; v[b][6] = q.z; v[b][7] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[b][e] = 0.0f;
      }
    }
    if (NORM == 1) {
      float s = 0.0f;
#pragma unroll
      for (int b = 0; b < BL; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[b][e];
      const float mean = fsf_wave_sum(s) * inv_c;
      float q = 0.0f;
#pragma unroll
      for (int b = 0; b < BL; ++b)
        if (lane + 64 * b < nb) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[b][e] - mean; q = fmaf(d, d, q); }
        }
      const float rstd = rsqrtf(fsf_wave_sum(q) * inv_c + eps);
#pragma unroll
      for (int b = 0; b < BL; ++b) {
        const int blk = lane + 64 * b;
        if (blk < nb) {
          const float4 g0 = *reinterpret_cast<const float4*>(gamma + 8 * blk), g1 = *reinterpret_cast<const float4*>(gamma + 8 * blk + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(beta + 8 * blk), b1 = *reinterpret_cast<const float4*>(beta + 8 * blk + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            lna_f32x2 y = (lna_f32x2{v[b][e], v[b][e + 1]} - lna_pk(mean)) * lna_pk(rstd) * lna_f32x2{gg[e], gg[e + 1]} + lna_f32x2{bb[e], bb[e + 1]};
            y = lna_act2(y, ACT);
            v[b][e] = y.x; v[b][e + 1] = y.y;
          }
        }
      }
    } else if (ACT != 0) {
#pragma unroll
      for (int b = 0; b < BL; ++b)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const lna_f32x2 y = lna_act2(lna_f32x2{v[b][e], v[b][e + 1]}, ACT);
          v[b][e] = y.x; v[b][e + 1] = y.y;
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int b = 0; b < BL; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[b][e]));
    amax = fsf_wave_max(amax);
    float s_row, inv_row;
    lna_pick_scale(amax, s_row, inv_row);
    if (lane == 0) inv_scales[row] = inv_row;
#pragma unroll
    for (int b = 0; b < BL; ++b) {
      const int blk = lane + 64 * b;
      if (blk < nb) {
        lna_u32x4 hi, lo;
        lna_split8_f16(v[b], s_row, hi, lo);
        uint4* dst = planes + (row * nb + blk) * 2;
        dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        dst[1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        if (out) {
          *reinterpret_cast<float4*>(out + row * out_stride + 8 * blk) = make_float4(v[b][0], v[b][1], v[b][2], v[b][3]);
          *reinterpret_cast<float4*>(out + row * out_stride + 8 * blk + 4) = make_float4(v[b][4], v[b][5], v[b][6], v[b][7]);
        }
      }
    }
  }
}

// ---- K22s: segmented max of the activated tile, rows sorted by segment ------------------------------------------------
// Per 16-row group a segmented max-scan along the rows (16 lanes of a DPP row per channel quad), then per run:
//   closed (the segment starts and ends inside the group)  -> its last lane stores the maximum,
//   open at the head and / or the tail                      -> one of the group's two LDS slots (in the weight buffer the last
//                                                              chunk just left free), merged in row order by 128 threads.
// Only a segment that reaches beyond its 128-row block (<= 2 per block) ends in an atomic max; every other segment is stored
// once.  max is exact: the result does not depend on any order.  (A contiguous range of blocks per workgroup with the open
// maximum carried from block to block — atomics only at the range ends — was built first and measured 10 % slower for the
// whole kernel: 768 workgroups each streaming its own region lose to 768 workgroups sweeping one window.)
// The scan runs inside the epilogue's tile loop (four values at a time, right after they are activated): nothing but the
// per-group flags below outlives a tile.
constexpr int LNA_DPP_ROW_SHR = 0x110, LNA_DPP_ROW_SHL = 0x100;

template <int CTRL>
__device__ __forceinline__ float lna_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int lna_dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }

constexpr int LNA_SLOT_HEAD_OPEN = 1 << 29, LNA_SLOT_TAIL_OPEN = 1 << 30, LNA_SLOT_SID = (1 << 29) - 1;  // (-1 = empty slot)

struct LnaSegSmem {      // persistent part (behind the per-channel vectors)
  int slot_sid[16];      // segment | LNA_SLOT_HEAD_OPEN | LNA_SLOT_TAIL_OPEN, or -1
};

struct LnaSegCtx {       // per lane, for ONE 16-row group of its wave (built right before the group's tiles: half the live masks)
  bool one_seg;          // (wave-uniform) the whole group is one run
  uint64_t m1, m2, m4, m8;  // lane masks (SGPR pairs): the row 1 / 2 / 4 / 8 above belongs to the same segment
  bool write, to_slot;   // this lane ends a run of a group that holds rows < n; the run is open (its maxima go to an LDS slot)
  int dst;               // float offset of the run's row in seg_out (a closed run) or in the slots — 32 bits: a 64-bit pointer per
                         // lane was spilled and re-read from scratch for every tile
};

struct LnaSegBlock {     // what the epilogue needs to build the groups' contexts
  int sid[LNA_RG];       // segment of this lane's row in either group (requested at the top of the block: no exposed latency here)
  int sid_before, sid_after;  // (wave-uniform) segment of the row above the wave's 32 rows / below them, -1 where there is none
  int64_t blk_row0;
  int wave;
  float* slots;
  LnaSegSmem* sm;
};

// The scan of one tile's four values, hand-scheduled: a step is ONE instruction per value — `v_max_f32_dpp x, x(row_shr:k), x`, a lane
// whose source lies outside its 16-lane row keeps x (bound_ctrl off) — plus, when the group holds more than one segment, a select on the
// step's lane mask.  As compiled from `fmaxf(x, update_dpp(x, x))` a step was five to six (a copy for the tied old value, the hazard nop,
// v_mov_dpp, a canonicalising v_max of the shuffled operand, the v_max, v_cndmask): the scan was ~1 400 of a wave's ~3 700 VALU
// instructions per row block in a kernel that is bound by VALU issue (profiles/r5_pmc_k22s.txt).  The DPP read-after-VALU-write hazard
// (two wait states; the assembler does not see into the block) is covered by the interleaving: a value's next step comes four
// instructions after its last write, and the block opens with a nop for whatever produced the inputs.
__device__ __forceinline__ float4 lna_seg_scan(const LnaSegCtx& sc, float4 y) {
  float a = y.x, b = y.y, c = y.z, d = y.w;
#define LNA_SCAN_MAX1(K)                                               \
  "v_max_f32_dpp %0, %0, %0 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %1, %1, %1 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %2, %2, %2 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %3, %3, %3 row_shr:" #K " row_mask:0xf bank_mask:0xf\n"
#define LNA_SCAN_MAXSEL(K, M)                                          \
  "v_max_f32_dpp %4, %0, %0 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %5, %1, %1 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %6, %2, %2 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_max_f32_dpp %7, %3, %3 row_shr:" #K " row_mask:0xf bank_mask:0xf\n" \
  "v_cndmask_b32_e64 %0, %0, %4, " M "\n"                              \
  "v_cndmask_b32_e64 %1, %1, %5, " M "\n"                              \
  "v_cndmask_b32_e64 %2, %2, %6, " M "\n"                              \
  "v_cndmask_b32_e64 %3, %3, %7, " M "\n"
  if (sc.one_seg) {  // (wave-uniform) plain prefix maxima, no selects
    asm volatile("s_nop 1\n" LNA_SCAN_MAX1(1) LNA_SCAN_MAX1(2) LNA_SCAN_MAX1(4) LNA_SCAN_MAX1(8)
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  } else {  // (lanes without a source keep garbage in the scratch values: their mask bits are clear)
    float oa, ob, oc, od;
    asm volatile("s_nop 1\n" LNA_SCAN_MAXSEL(1, "%8") LNA_SCAN_MAXSEL(2, "%9") LNA_SCAN_MAXSEL(4, "%10") LNA_SCAN_MAXSEL(8, "%11")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(oa), "=&v"(ob), "=&v"(oc), "=&v"(od)
                 : "s"(sc.m1), "s"(sc.m2), "s"(sc.m4), "s"(sc.m8));
  }
#undef LNA_SCAN_MAX1
#undef LNA_SCAN_MAXSEL
  return make_float4(a, b, c, d);
}

// one group's segment ids -> scan flags, run ends and their destinations; called after the barrier that frees the slot buffer
__device__ __forceinline__ LnaSegCtx lna_seg_prepare(const LnaArgs& a, const LnaSegBlock& sb, int rg, int rowl, int grp) {
  LnaSegCtx sc;
  const int g = 2 * sb.wave + rg;
  const int64_t grow0 = sb.blk_row0 + 16 * g;
  const int sid = sb.sid[rg];  // (rows past n repeat row n - 1 in every respect: the maxima are unchanged)
  const int up1 = lna_dpp_i<LNA_DPP_ROW_SHR + 1>(-1, sid), up2 = lna_dpp_i<LNA_DPP_ROW_SHR + 2>(-1, sid);
  const int up4 = lna_dpp_i<LNA_DPP_ROW_SHR + 4>(-1, sid), up8 = lna_dpp_i<LNA_DPP_ROW_SHR + 8>(-1, sid);
  const int dn1 = lna_dpp_i<LNA_DPP_ROW_SHL + 1>(-1, sid);
  sc.m1 = __builtin_amdgcn_ballot_w64(up1 == sid); sc.m2 = __builtin_amdgcn_ballot_w64(up2 == sid);  // (ids >= 0: -1 = no such lane)
  sc.m4 = __builtin_amdgcn_ballot_w64(up4 == sid); sc.m8 = __builtin_amdgcn_ballot_w64(up8 == sid);
  sc.one_seg = __builtin_amdgcn_readfirstlane(sid) == __builtin_amdgcn_readlane(sid, 15);  // sorted: first == last
  sc.write = sc.to_slot = false;
  sc.dst = 0;
  if (grow0 < a.n && (rowl == 15 || dn1 != sid)) {  // (a group past the last row forms no run: its slots stay empty)
    sc.write = true;
    // a run is open at the head iff it is the group's first run and the row above the group belongs to the same segment (ids are
    // sorted), open at the tail iff it is the last run and the row below does: no look-up of the segment's bounds
    const int first = __builtin_amdgcn_readfirstlane(sid), last = __builtin_amdgcn_readlane(sid, 15);
    const int above = rg == 0 ? sb.sid_before : __builtin_amdgcn_readlane(sb.sid[0], 15);
    const int below = rg == LNA_RG - 1 ? sb.sid_after : __builtin_amdgcn_readfirstlane(sb.sid[LNA_RG - 1]);
    const bool head_open = sid == first && above == first, tail_open = sid == last && below == last;
    if (!head_open && !tail_open) {
      sc.dst = sid * (int)a.seg_out_stride;
    } else {
      const int slot = 2 * g + (head_open ? 0 : 1);
      if (grp == 0) sb.sm->slot_sid[slot] = sid | (head_open ? LNA_SLOT_HEAD_OPEN : 0) | (tail_open ? LNA_SLOT_TAIL_OPEN : 0);
      sc.to_slot = true;
      sc.dst = slot * 128;
    }
  }
  return sc;
}

// after every wave has parked its open runs: merge them in row order (128 threads, one per channel).  A segment whose parts all
// lie in this 128-row block (its first slot is closed at the head, its last at the tail) is stored plainly; one that reaches into
// a neighbouring block — some other workgroup's — is combined by atomic max: at most two per block, fire-and-forget.
__device__ __forceinline__ void lna_seg_merge(const LnaArgs& a, const float* slots, const LnaSegSmem* sm) {
  __syncthreads();
  if (threadIdx.x < 128) {
    const int ch = threadIdx.x;
    int cs = -1;  // segment | LNA_SLOT_HEAD_OPEN (it began above this block)
    float cv = -INFINITY;
    auto flush = [&](int tag, float v, bool complete) {
      if (ch >= a.c) return;
      float* p = a.seg_out + (int64_t)(tag & LNA_SLOT_SID) * a.seg_out_stride + ch;
      if (complete && !(tag & LNA_SLOT_HEAD_OPEN)) *p = v;
      else lna_atomic_max(p, v);
    };
    for (int s = 0; s < 16; ++s) {
      const int ss = sm->slot_sid[s];
      if (ss < 0) continue;
      const float v = slots[s * 128 + ch];
      if (cs >= 0 && (ss & LNA_SLOT_SID) == (cs & LNA_SLOT_SID)) cv = fmaxf(cv, v);
      else {  // (an open segment always continues in the next occupied slot; kept general)
        if (cs >= 0) flush(cs, cv, false);
        cs = ss & (LNA_SLOT_SID | LNA_SLOT_HEAD_OPEN);
        cv = v;
      }
      if (!(ss & LNA_SLOT_TAIL_OPEN)) {  // the segment ends in this group
        flush(cs, cv, true);
        cs = -1;
      }
    }
    if (cs >= 0) flush(cs, cv, false);  // continues below this block
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
template <int T, bool SEG = false, int NORM_CT = -1, int ACT_CT = -1>  // SEG: the activated values also go through the segmented
// max-scan; NORM_CT / ACT_CT >= 0: norm and activation fixed at compile time (the K22s variants: their epilogue is already twice
// the code, and the run-time switches of the plain kernel would double it again)
__device__ __forceinline__ void lna_epilogue(const LnaArgs& a, lna_f32x4 (&acc)[LNA_RG][T], int64_t row0, int ch_base, int rowl,
                                             int grp, const float* vec, LnaTl& tl, const LnaSegBlock* sb = nullptr) {
  const float inv_c = 1.0f / (float)a.norm_w;
  // the arithmetic below runs on pairs (v_pk_*_f32): the same IEEE operations per value as the scalar form, half the instructions
  auto lo = [](const lna_f32x4& v) { return lna_f32x2{v[0], v[1]}; };
  auto hi = [](const lna_f32x4& v) { return lna_f32x2{v[2], v[3]}; };
  auto put = [](lna_f32x4& v, lna_f32x2 l, lna_f32x2 h) { v[0] = l.x; v[1] = l.y; v[2] = h.x; v[3] = h.y; };
#pragma unroll
  for (int rg = 0; rg < LNA_RG; ++rg) {
    const int64_t row = row0 + 16 * rg + rowl;
    float mean = 0.0f, rstd = 1.0f;
    LnaSegCtx sc;
    int g4 = 4 * grp;  // this lane's channel offset inside a tile
    if constexpr (SEG) {
      sc = lna_seg_prepare(a, *sb, rg, rowl, grp);
      // (opaque: with the scan's live state on top, the compiler otherwise hoists the 64-bit per-lane store offsets of all tiles out
      // of the block loop, spills them, and re-reads one from scratch in front of every tile's stores)
      asm volatile("" : "+v"(g4));
    }
    if (a.bias) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(vec + 16 * t + 4 * grp);  // (0 beyond c)
        put(acc[rg][t], lo(acc[rg][t]) + lna_f32x2{b.x, b.y}, hi(acc[rg][t]) + lna_f32x2{b.z, b.w});
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
          if (16 * (t0 + t) + 4 * grp < a.slice_w && ch_base + 16 * (t0 + t) + 4 * grp < a.c)
            put(acc[rg][t0 + t], lo(acc[rg][t0 + t]) + lna_f32x2{b[t].x, b[t].y}, hi(acc[rg][t0 + t]) + lna_f32x2{b[t].z, b[t].w});
        }
      }
    }
    LNA_TL_MARK(tl, 2);  // segment context + bias + the per-row addend (its gather is waited for here)
    if ((NORM_CT >= 0 ? NORM_CT : a.norm) == 1) {  // LayerNorm over the c channels (channels >= c are exactly 0: zero weights, no bias)
      lna_f32x2 s2 = lna_pk(0.0f);
#pragma unroll
      for (int t = 0; t < T; ++t) s2 = (s2 + lo(acc[rg][t])) + hi(acc[rg][t]);
      mean = lna_row_sum(s2.x + s2.y) * inv_c;
      const lna_f32x2 m2 = lna_pk(mean);
      lna_f32x2 q2 = lna_pk(0.0f);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const bool live = 16 * t + 4 * grp < a.slice_w && ch_base + 16 * t + 4 * grp < a.c;  // (widths are multiples of 4)
        const lna_f32x2 dl = live ? lo(acc[rg][t]) - m2 : lna_pk(0.0f), dh = live ? hi(acc[rg][t]) - m2 : lna_pk(0.0f);
        q2 = lna_pk_fma(dl, dl, q2);
        q2 = lna_pk_fma(dh, dh, q2);
      }
      rstd = rsqrtf(lna_row_sum(q2.x + q2.y) * inv_c + a.eps);
    }
    LNA_TL_MARK(tl, 3);  // LayerNorm statistics
    if (row < a.n || SEG) {
      float* orow = a.out + row * a.out_stride;
      const lna_f32x2 m2 = lna_pk(mean), r2 = lna_pk(rstd);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int ch0 = ch_base + 16 * t + g4;
        if (16 * t + 4 * grp < a.slice_w && ch_base + 16 * t + 4 * grp < a.c) {
#ifdef FSF_ABL_LNA_NO_VEC  // ablation: no LDS reads of gamma / beta in the tile loop (constants: WRONG results)
          const float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
#else
          const float4 g = *reinterpret_cast<const float4*>(vec + 128 + 16 * t + 4 * grp);  // (1 / 0 without a norm)
          const float4 b = *reinterpret_cast<const float4*>(vec + 256 + 16 * t + 4 * grp);
#endif
          const lna_f32x2 yl = lna_act2((lo(acc[rg][t]) - m2) * r2 * lna_f32x2{g.x, g.y} + lna_f32x2{b.x, b.y}, ACT_CT >= 0 ? ACT_CT : a.act);
          const lna_f32x2 yh = lna_act2((hi(acc[rg][t]) - m2) * r2 * lna_f32x2{g.z, g.w} + lna_f32x2{b.z, b.w}, ACT_CT >= 0 ? ACT_CT : a.act);
          const float4 y = make_float4(yl.x, yl.y, yh.x, yh.y);
#if defined(FSF_ABL_LNA_COAL_ST)  // ablation (WRONG places, same bytes): every store instruction writes 1 KB of consecutive addresses
          if (!SEG || (a.out && row < a.n)) *reinterpret_cast<float4*>(orow - (int64_t)rowl * a.out_stride + t * 256 + (rowl + 16 * grp) * 4) = y;
#elif !defined(FSF_ABL_LNA_NO_STORE)
          if (!SEG || (a.out && row < a.n)) *reinterpret_cast<float4*>(orow + ch0) = y;
#else
          if (y.x == 123.456f) *reinterpret_cast<float4*>(orow + ch0) = y;
#endif
          if constexpr (SEG) {
            const float4 mx = lna_seg_scan(sc, y);  // (a group past the last row scans copies of row n - 1 and writes nothing)
            if (sc.write) {
              if (sc.to_slot) *reinterpret_cast<float4*>(sb->slots + (sc.dst + 16 * t + g4)) = mx;
              else *reinterpret_cast<float4*>(a.seg_out + (sc.dst + 16 * t + g4)) = mx;
            }
          }
        }
      }
    }
    LNA_TL_MARK(tl, 4);  // affine + activation + stores (+ the segmented scan and its run stores)
  }
}


// NW waves per workgroup.  A weight chunk enters the CU once per WORKGROUP and chunk (LDS-DMA), so three 4-wave workgroups per CU
// take the same 24 KB in three times per 128 rows each; ONE 12-wave workgroup per CU (same 12 waves, same registers) takes it in
// once per 384 rows.  Measured (round 3, same box): no faster in isolation (510 k x 256 -> 128: 277 vs 279 us; k = 128 .. 180: 5-15 %
// SLOWER — a barrier over twelve waves per chunk) and 7 % slower in the frame (a 768-thread workgroup shuts the other stream's kernels
// out of its CU) — so the weight stream is not what this kernel waits for.  Kept behind FSF_K22_WIDE_MIN_ROWS=<rows> (default: never).
template <int T, int NW, bool SEG = false, int NORM_CT = -1, int ACT_CT = -1, int XM = 0>  // 16-channel tiles (c <= 16 T); SEG: +
// segmented max of the output (rows sorted by segment), norm / act fixed at compile time; XM = 1 (XP): x and W arrive as f16 hi | lo
// planes (K22h); XM = 2 (XF, K22f): fp32 x split IN the kernel into f16 hi | lo of x * s_row (s_row: the running power-of-two unit of the
// row, below), W as f16 planes — three MFMA passes per product instead of the six of the exact bf16 split (XM = 0)
__global__ void __launch_bounds__(NW * 64, NW == 4 ? (SEG ? LNA_SEG_WPS : LNA_WPS) : 3) linear_norm_act_kernel(LnaArgs a) {
  constexpr int LNA_NW = NW;
  constexpr int LNA_ROWS = NW * LNA_RG * 16;
  constexpr bool XP = XM == 1, XF = XM == 2;
  constexpr int NPL = XM != 0 ? 2 : 3;        // weight planes per tile
  constexpr int CHUNK_U4 = T * NPL * 64;      // uint4 per weight chunk
  static_assert(!SEG || (NW == 4 && CHUNK_U4 * 16 >= 16 * 128 * 4), "the segmented max parks 16 x 128 floats in a weight buffer");
  extern __shared__ __attribute__((aligned(16))) char lna_smem[];
  uint4* wbuf = reinterpret_cast<uint4*>(lna_smem);  // [2][CHUNK_U4], then 384 floats of per-channel vectors (, then LnaSegSmem)
  float* vec = reinterpret_cast<float*>(wbuf + 2 * CHUNK_U4);
  LnaSegSmem* segsm = reinterpret_cast<LnaSegSmem*>(vec + 384);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = lane & 15, grp = lane >> 4;
  const int nkc = (a.k + LNA_KC - 1) / LNA_KC;
  const int64_t nblk = (a.n + LNA_ROWS - 1) / LNA_ROWS;
  // more than 128 output channels: gridDim.y slices of 128, each an independent [rows, 128] product (no LayerNorm then:
  // its statistics span the slices; the caller runs fsf_norm_act on the result)
  // XP launches come as a 1-D grid laid out so that the workgroups that share a row block (one per 128-channel slice) have the same
  // id modulo 8, i.e. land on the same XCD and meet their x rows in its L2: id = xcd + 8 (slice + nslice j), row-block lane 8 j + xcd
  int slice_id = (int)blockIdx.y;
  int64_t blk_first_ = blockIdx.x, blk_step_ = gridDim.x;
  if constexpr (XP) {
    const int nslice = (a.c + a.slice_w - 1) / a.slice_w;
    const int wg = (int)blockIdx.x, q = wg >> 3;
    slice_id = q % nslice;
    blk_first_ = (int64_t)(q / nslice) * 8 + (wg & 7);
    blk_step_ = gridDim.x / nslice;
  }
  const int ch_base = a.slice_w * slice_id;
  const uint4* planes = a.planes + (XM != 0 ? 16 : 0) + (int64_t)slice_id * nkc * CHUNK_U4;  // (f16 planes: behind the 256-byte header)
  if (XM != 0 && reinterpret_cast<const unsigned*>(a.planes)[3] != LNA_F16_TAG)
    __builtin_trap();  // a buffer that fsf_linear_prepare_weight_f16 did not write: the launch fails loudly instead of multiplying garbage

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
      if constexpr (XP) xrow[rg] = a.x + r * (int64_t)a.k;  // (plane rows: k / 8 blocks x 2 planes x 16 B = 4 k bytes, like fp32 rows)
      else xrow[rg] = a.x + r * a.x_stride + (int64_t)slice_id * a.x_slice_off;
    }
  };
  // raw x of one chunk: [row group][8 floats]; the NEXT chunk is requested while this one is split and multiplied.
  // Always exactly two 16-byte loads per row group: offsets past the row are clamped into it (x_stride is a multiple
  // of 4 and >= k, so a quad that holds any column < k is never clamped) and the columns >= k are zeroed afterwards
  // (what follows the row in memory may be NaN).
  const int last_quad = XP ? 0 : (int)a.x_stride - 4 - (int)((int64_t)slice_id * a.x_slice_off);  // (relative to this slice's first column)
  auto load_x = [&](int kc, float (&v)[LNA_RG][8]) {
#pragma unroll
    for (int rg = 0; rg < LNA_RG; ++rg) {
      if constexpr (XP) {  // k block 4 kc + grp of the row: hi plane | lo plane, 32 contiguous bytes (k is a multiple of 32 here)
        const float4* pb = reinterpret_cast<const float4*>(xrow[rg]) + 2 * (4 * kc + grp);
        const float4 p = pb[0], q = pb[1];
        v[rg][0] = p.x; v[rg][1] = p.y; v[rg][2] = p.z; v[rg][3] = p.w;
        v[rg][4] = q.x; v[rg][5] = q.y; v[rg][6] = q.z; v[rg][7] = q.w;
        continue;
      }
      const int kq = kc * LNA_KC + 8 * grp;  // this lane's 8 k values
#ifdef FSF_ABL_LNA_COAL_X  // ablation (WRONG values, same bytes): every load instruction reads 1 KB of consecutive addresses, lane by lane
      const float* cb = xrow[rg] - (int64_t)rowl * a.x_stride + (lane & 63) * 4;
      const float4 p = *reinterpret_cast<const float4*>(cb + ((2 * kc) & 7) * 256);
      const float4 q = *reinterpret_cast<const float4*>(cb + ((2 * kc + 1) & 7) * 256);
#else
      const float4 p = *reinterpret_cast<const float4*>(xrow[rg] + min(kq, last_quad));
      const float4 q = *reinterpret_cast<const float4*>(xrow[rg] + min(kq + 4, last_quad));
#endif
      v[rg][0] = p.x; v[rg][1] = p.y; v[rg][2] = p.z; v[rg][3] = p.w;
      v[rg][4] = q.x; v[rg][5] = q.y; v[rg][6] = q.z; v[rg][7] = q.w;
    }
  };
  // row blocks of this workgroup: strided over the grid, or (SEG) one contiguous range, so that a segment's rows meet in one
  // workgroup wherever they can
  const int64_t blk_first = blk_first_;
#define LNA_BLK_END nblk
#define LNA_BLK_STEP blk_step_
  lna_stage_vectors(a, ch_base, vec);
  // (XF) the weight scale s_w, and the cap of a row's FIRST scale: with a per-row addend in the accumulators — it enters them multiplied
  // by s * s_w — s * s_w stays <= 2^40 (an addend below 2^87 cannot overflow; a row whose values all lie below 2^-27 / s_w loses bits
  // it could not contribute next to an addend anyway); without one, any scale a finite row asks for
  float w_scale = 1.0f, xs_cap = 0x1p126f, xs_cap_inv = 0x1p-126f;
  if constexpr (XF) {
    w_scale = reinterpret_cast<const float*>(a.planes)[1];
    if (a.row_add) {
      int e = (int)((__float_as_uint(reinterpret_cast<const float*>(a.planes)[0]) >> 23) & 0xffu) - 127 + 40;
      e = e < -126 ? -126 : (e > 126 ? 126 : e);
      xs_cap = __uint_as_float((unsigned)(e + 127) << 23);
      xs_cap_inv = __uint_as_float((unsigned)(127 - e) << 23);
    }
  }
  LnaTl tl;
#ifdef FSF_LNA_TIMELINE
  tl.start();
#endif
  // The per-row addend (`cat([point, group[inv]]) W^T` = point W_left^T + (group W_right^T)[inv]) enters through the ACCUMULATORS: they
  // start a row block as the gathered addend rows instead of zeros, the MFMAs accumulate on top.  In the epilogue (round 4) the gather
  // was four exposed round trips per block — index, then 4 + 4 tiles per row group, twice — 20 % of the grouped K22s kernel
  // (profiles/r5_lna_timeline.txt); now the index of the NEXT block's rows is fetched under the chunk loop and the 16 row loads of a
  // block are in flight together, into registers that are free at that point, behind the first chunk's split.
  int64_t radd_idx[LNA_RG];
  auto fetch_addend_index = [&](int64_t blk) {
    if (a.row_add) {
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) {
        const int64_t r = blk * LNA_ROWS + (int64_t)wave * (LNA_RG * 16) + 16 * rg + rowl;
        radd_idx[rg] = a.row_add_index[r < a.n ? r : a.n - 1];
      }
    }
  };
  LnaArgs a_epi = a;  // (the epilogue's copy: the addend is already in the accumulators)
  a_epi.row_add = nullptr;
  float xc[LNA_RG][8];
  int buf = 0;
  if (blk_first < LNA_BLK_END) {
    fetch_addend_index(blk_first);
    set_rows(blk_first);
    load_x(0, xc);
    stage_w(0, 0);
  }
  for (int64_t blk = blk_first; blk < LNA_BLK_END; blk += LNA_BLK_STEP) {
    const int64_t row0 = blk * LNA_ROWS + (int64_t)wave * (LNA_RG * 16);
    LnaSegBlock sb;
    if constexpr (SEG) {  // the rows' segment ids (and those of the rows just above / below the wave's 32) arrive under the chunk loop
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) {
        const int64_t r = row0 + 16 * rg + rowl;
        sb.sid[rg] = (int)a.seg_ids[r < a.n ? r : a.n - 1];
      }
      sb.sid_before = row0 > 0 ? (int)a.seg_ids[row0 - 1 < a.n ? row0 - 1 : a.n - 1] : -1;
      sb.sid_after = row0 + 16 * LNA_RG < a.n ? (int)a.seg_ids[row0 + 16 * LNA_RG] : -1;
    }
    lna_f32x4 acc[LNA_RG][T];
#pragma unroll
    for (int rg = 0; rg < LNA_RG; ++rg)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[rg][t] = lna_f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.row_add) {
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) {
        const float* add = a.row_add + radd_idx[rg] * a.row_add_stride;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int lc0 = 16 * t + 4 * grp, ch0 = ch_base + lc0;
          if (lc0 < a.slice_w && ch0 < a.c) {
            const float4 b = *reinterpret_cast<const float4*>(add + ch0);
            acc[rg][t] = lna_f32x4{b.x, b.y, b.z, b.w};
          }
        }
      }
    }
    float xinv[LNA_RG];
    float xs_cur[LNA_RG];  // (XF) the unit the row's accumulators are kept in: acc = (true sum) * xs_cur * s_w; xinv = 1 / xs_cur
    if constexpr (XP) {  // the rows' inverse scales (requested here, used after the chunk loop)
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) {
        const int64_t r = row0 + 16 * rg + rowl;
        xinv[rg] = a.x_inv_scale[r < a.n ? r : a.n - 1];
      }
    }
#ifdef FSF_ABL_LNA_NO_XBLK  // ablation: every row block starts with an exposed load (the kernel before the cross-block pipeline)
    if (blk != blk_first) {
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
      if constexpr (XP) {  // the planes ARE the operands: nothing to split
#pragma unroll
        for (int rg = 0; rg < LNA_RG; ++rg) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xh[rg][e] = __float_as_uint(xc[rg][e]); xl[rg][e] = __float_as_uint(xc[rg][4 + e]); }
          xm[rg] = xh[rg];
        }
      } else
      if ((kc + 1) * LNA_KC > a.k) {  // (uniform: the last chunk of a k that is not a multiple of 32)
#pragma unroll
        for (int rg = 0; rg < LNA_RG; ++rg)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kc * LNA_KC + 8 * grp + e >= a.k) xc[rg][e] = 0.0f;
      }
      if constexpr (XM == 0) {
#pragma unroll
        for (int rg = 0; rg < LNA_RG; ++rg) lna_split8(xc[rg], xh[rg], xm[rg], xl[rg]);
      }
      if constexpr (XF) {
        // K22f: a row's chunk is scaled by a power of two s with s * max|x| in [2^13, 2^14) and split into f16 hi + lo (22 bits relative
        // to the maximum — the arithmetic of K9d / K22h).  The scale may only FALL from chunk to chunk (it follows the running maximum of
        // the row, so a row is held to 22 bits of ITS maximum, like a whole-row scale would); when it falls, the row's accumulators,
        // which are kept in the unit s * s_w, are multiplied by new / old — a power of two, exact.  The first chunk's scale is capped
        // (xs_cap) where a per-row addend sits in the accumulators, which it enters multiplied by s * s_w.
        float ratio[LNA_RG];
        bool changed = false;
#pragma unroll
        for (int rg = 0; rg < LNA_RG; ++rg) {
          unsigned mb = __float_as_uint(xc[rg][0]) & 0x7fffffffu;  // (bit patterns of |x| order like the values; NaN / inf end up largest)
#pragma unroll
          for (int e = 1; e < 8; ++e) mb = max(mb, __float_as_uint(xc[rg][e]) & 0x7fffffffu);
          const auto r16 = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);  // the four lanes of a row: lane ^ 16, lane ^ 32
          mb = max(r16[0], r16[1]);
          const auto r32 = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
          mb = max(r32[0], r32[1]);
          float s_new, inv_new;
          lna_pick_scale(__uint_as_float(mb), s_new, inv_new);
          if (kc == 0) {
            if (mb == 0u || s_new > xs_cap) { s_new = xs_cap; inv_new = xs_cap_inv; }
            ratio[rg] = s_new * w_scale;
            changed |= a.row_add != nullptr;  // (the accumulators hold the addend, unit 1, or zeros)
          } else {
            if (mb == 0u || s_new > xs_cur[rg]) { s_new = xs_cur[rg]; inv_new = xinv[rg]; }
            ratio[rg] = s_new * xinv[rg];  // (1 where nothing changed)
            changed |= s_new != xs_cur[rg];
          }
          xs_cur[rg] = s_new;
          xinv[rg] = inv_new;
          lna_split8_f16(xc[rg], s_new, xh[rg], xl[rg]);
          xm[rg] = xh[rg];
        }
        if (__builtin_amdgcn_ballot_w64(changed) != 0) {  // (wave-uniform: rare after the first chunk)
#pragma unroll
          for (int rg = 0; rg < LNA_RG; ++rg)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[rg][t] = acc[rg][t] * ratio[rg];
        }
      }
      // this chunk's weights (DMA issued one iteration ago, before that iteration's MFMAs) have landed; the raw barrier
      // carries no fence, so nothing else is drained with them
      LNA_TL_MARK(tl, 0);  // chunk work: the previous chunk's MFMAs (issue), this chunk's split
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // + every wave is done reading buffer buf^1
      asm volatile("" ::: "memory");
      LNA_TL_MARK(tl, 1);  // waiting: this chunk's x and weights, the LDS reads, the barrier
      if (kc + 1 < nkc) {
        stage_w(kc + 1, buf ^ 1);
        load_x(kc + 1, xc);
      }
#ifndef FSF_ABL_LNA_NO_XBLK
      else if (blk + LNA_BLK_STEP < LNA_BLK_END) {  // first chunk of the next row block
        stage_w(0, buf ^ 1);
        set_rows(blk + LNA_BLK_STEP);
        load_x(0, xc);
        fetch_addend_index(blk + LNA_BLK_STEP);
      }
#endif
      const uint4* wc = wbuf + buf * CHUNK_U4;
      // Two channel tiles x LNA_RG row groups = 4 independent accumulators per product term: consecutive MFMAs never hit
      // the same accumulator
      if constexpr (XM != 0) {
#pragma unroll
        for (int t = 0; t < T; t += 2) {
          lna_f16x8 wfr[2][2];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const uint4* wf = wc + ((t + tt) * 2) * 64 + lane;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wfr[tt][pl] = __builtin_bit_cast(lna_f16x8, wf[64 * pl]);
          }
          // (weight plane, x plane): lo hi, hi lo, hi hi — small terms first
          constexpr int TERM_W[3] = {1, 0, 0};
          constexpr int TERM_X[3] = {0, 1, 0};
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int rg = 0; rg < LNA_RG; ++rg) {
                const lna_u32x4 xb = TERM_X[term] == 0 ? xh[rg] : xl[rg];
                acc[rg][t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfr[tt][TERM_W[term]], __builtin_bit_cast(lna_f16x8, xb),
                                                                         acc[rg][t + tt], 0, 0, 0);
              }
        }
      } else {
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
    }
    if constexpr (XM != 0) {  // back to the unscaled product: both scales are powers of two (exact)
      const float w_inv = *reinterpret_cast<const float*>(a.planes);
#pragma unroll
      for (int rg = 0; rg < LNA_RG; ++rg) {
        const float sc = xinv[rg] * w_inv;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[rg][t] = acc[rg][t] * sc;
      }
    }
    if constexpr (SEG) {  // (`buf` was flipped by the loop: the chunk loop's last buffer, free now, is buf ^ 1)
      float* slots = reinterpret_cast<float*>(wbuf + (buf ^ 1) * CHUNK_U4);
      sb.blk_row0 = blk * LNA_ROWS; sb.wave = wave; sb.slots = slots; sb.sm = segsm;
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS reads of the last chunk have returned ...
      __builtin_amdgcn_s_barrier();        // ... and so have every other wave's: the slots may overlay that buffer
      if (lane < 4) segsm->slot_sid[4 * wave + lane] = -1;
      LNA_TL_MARK(tl, 0);  // (the last chunk's MFMAs + the barrier that frees the slot buffer)
      lna_epilogue<T, true, NORM_CT, ACT_CT>(a_epi, acc, row0, ch_base, rowl, grp, vec, tl, &sb);
      lna_seg_merge(a, slots, segsm);
      LNA_TL_MARK(tl, 5);  // the block's slot merge (a barrier + 128 threads)
    } else {
      LNA_TL_MARK(tl, 0);
      lna_epilogue<T, false>(a_epi, acc, row0, ch_base, rowl, grp, vec, tl);
    }
  }
#undef LNA_BLK_END
#undef LNA_BLK_STEP
#ifdef FSF_LNA_TIMELINE
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&lna_tl[i], tl.acc[i]);
    atomicAdd(&lna_tl[8], 1ull);
  }
#endif
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

static int lna_launch(const LnaArgs& a_in, int nslice, hipStream_t stream, bool xp = false, bool xf = false) {
  LnaArgs a = a_in;
  const int T = lna_tiles(a.slice_w < a.c ? a.slice_w : a.c);
  const int rows = LNA_NW * LNA_RG * 16;
  const int64_t nblk = (a.n + rows - 1) / rows;
  int64_t gx = (256 * (a.seg_out ? LNA_SEG_WPS : LNA_WPS) + nslice - 1) / nslice;
  if (gx > nblk) gx = nblk;
  if (xp) {  // K22h: 1-D grid, `gx` row-block lanes (a multiple of 8: one XCD per lane) x nslice workgroups each
    if (T != 8 || a.seg_out) return FSF_ERR_UNSUPPORTED;
    gx = (gx + 7) / 8 * 8;
    constexpr size_t smem = (size_t)2 * 8 * 2 * 64 * 16 + 384 * 4;
    static std::atomic<uint64_t> attr_done{0};
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)linear_norm_act_kernel<8, 4, false, -1, -1, true>, (int)smem, attr_done));
    hipLaunchKernelGGL((linear_norm_act_kernel<8, 4, false, -1, -1, true>), dim3((unsigned)(gx * nslice)), dim3(256), smem, stream, a);
    FSF_LAUNCH_CHECK();
    return FSF_OK;
  }
  const dim3 grid((unsigned)gx, (unsigned)nslice);
#define FSF_LNA_X(T_, NW_, SEG_, NORM_, ACT_, XM_)                                                                                   \
  do {                                                                                                                              \
    constexpr size_t smem = (size_t)2 * T_ * (XM_ ? 2 : 3) * 64 * 16 + 384 * 4 + (SEG_ ? sizeof(LnaSegSmem) : 0);                    \
    static std::atomic<uint64_t> attr_done{0};                                                                                      \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)linear_norm_act_kernel<T_, NW_, SEG_, NORM_, ACT_, XM_>, (int)smem, attr_done)); \
    hipLaunchKernelGGL((linear_norm_act_kernel<T_, NW_, SEG_, NORM_, ACT_, XM_>), grid, dim3(NW_ * 64), smem, stream, a);           \
  } while (0)
#define FSF_LNA(T_, NW_, SEG_, NORM_, ACT_) FSF_LNA_X(T_, NW_, SEG_, NORM_, ACT_, 0)
  if (xf) {  // K22f: the 64- and 128-channel-tile forms (what the SIR / VFE / segmentation-head layers are)
    if (a.seg_out) {
      if (a.norm != 1 || (a.act != 1 && a.act != 2)) return FSF_ERR_UNSUPPORTED;
      if (T == 4 && a.act == 2) FSF_LNA_X(4, 4, true, 1, 2, 2);
      else if (T == 4) FSF_LNA_X(4, 4, true, 1, 1, 2);
      else if (T == 8 && a.act == 2) FSF_LNA_X(8, 4, true, 1, 2, 2);
      else if (T == 8) FSF_LNA_X(8, 4, true, 1, 1, 2);
      else return FSF_ERR_UNSUPPORTED;
    } else if (T == 4) FSF_LNA_X(4, 4, false, -1, -1, 2);
    else if (T == 8) FSF_LNA_X(8, 4, false, -1, -1, 2);
    else return FSF_ERR_UNSUPPORTED;
    FSF_LAUNCH_CHECK();
    return FSF_OK;
  }
  if (a.seg_out) {  // K22s: LayerNorm + GELU / ReLU (the SIR layers), 36 .. 128 channels
    if (a.norm != 1 || (a.act != 1 && a.act != 2)) return FSF_ERR_UNSUPPORTED;
    if (T == 4 && a.act == 2) FSF_LNA(4, 4, true, 1, 2);
    else if (T == 4) FSF_LNA(4, 4, true, 1, 1);
    else if (T == 8 && a.act == 2) FSF_LNA(8, 4, true, 1, 2);
    else if (T == 8) FSF_LNA(8, 4, true, 1, 1);
    else return FSF_ERR_UNSUPPORTED;
  } else if (T == 2) FSF_LNA(2, 4, false, -1, -1);
  else if (T == 4) FSF_LNA(4, 4, false, -1, -1);
  else FSF_LNA(8, 4, false, -1, -1);
#undef FSF_LNA
#undef FSF_LNA_X
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
            (int)(nslice * slice_c), nullptr, nullptr, 0, (int)slice_c, (int)slice_c, x_slice_offset, nullptr, nullptr, 0, nullptr};
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

static int lna_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c, const float* bias,
                       const float* row_add, const int64_t* row_add_index, int64_t row_add_stride, int32_t norm, const float* gamma,
                       const float* beta, float eps, int32_t act, float* out, int64_t out_stride, void* stream_, bool xf) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((row_add == nullptr) != (row_add_index == nullptr)) return FSF_ERR_INVALID_ARG;
  if (xf && (c <= 32 || ((uintptr_t)planes % 16) != 0)) return FSF_ERR_UNSUPPORTED;
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
            row_add, row_add_index, row_add_stride, 128, (int)c, 0, nullptr, nullptr, 0, nullptr};
  return lna_launch(a, lna_slices(c), stream, false, xf);
}

extern "C" int fsf_linear_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                           const float* bias, const float* row_add, const int64_t* row_add_index,
                                           int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta,
                                           float eps, int32_t act, float* out, int64_t out_stride, void* stream_) {
  return lna_grouped(x, n, k, x_stride, planes, c, bias, row_add, row_add_index, row_add_stride, norm, gamma, beta, eps, act, out,
                     out_stride, stream_, false);
}

extern "C" int fsf_linear_f16w_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* w_planes, int32_t c,
                                                const float* bias, const float* row_add, const int64_t* row_add_index,
                                                int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta,
                                                float eps, int32_t act, float* out, int64_t out_stride, void* stream_) {
  return lna_grouped(x, n, k, x_stride, w_planes, c, bias, row_add, row_add_index, row_add_stride, norm, gamma, beta, eps, act, out,
                     out_stride, stream_, true);
}

static int lna_segmax(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c, const float* bias,
                      const float* row_add, const int64_t* row_add_index, int64_t row_add_stride, int32_t norm, const float* gamma,
                      const float* beta, float eps, int32_t act, const int64_t* seg_ids, int64_t num_segments, float* seg_out,
                      int64_t seg_out_stride, float* out, int64_t out_stride, void* stream_, bool xf) {
  hipStream_t stream = (hipStream_t)stream_;
  if (xf && planes && ((uintptr_t)planes % 16) != 0) return FSF_ERR_UNSUPPORTED;
  if ((row_add == nullptr) != (row_add_index == nullptr)) return FSF_ERR_INVALID_ARG;
  if (row_add && ((row_add_stride % 4) != 0 || row_add_stride < c || ((uintptr_t)row_add % 16) != 0)) return FSF_ERR_UNSUPPORTED;
  if (n < 0 || k < 1 || c < 1 || !planes || norm < 0 || norm > 2 || act < 0 || act > 2 || (norm != 0 && (!gamma || !beta)) ||
      num_segments < 0 || (n > 0 && (!x || !seg_ids || !seg_out || num_segments < 1)))
    return FSF_ERR_INVALID_ARG;
  // LayerNorm + ReLU / GELU (what SIRLayer puts in front of its max), one 128-channel slice at most, more than 32 channels (the
  // slots overlay a >= 12 KB weight buffer)
  if (norm != 1 || act == 0 || c > 128 || c <= 32 || (c % 4) != 0 || (x_stride % 4) != 0 || ((uintptr_t)x % 16) != 0 || (seg_out_stride % 4) != 0 ||
      seg_out_stride < c || ((uintptr_t)seg_out % 16) != 0 || (out && ((out_stride % 4) != 0 || ((uintptr_t)out % 16) != 0)) ||
      n >= ((int64_t)1 << 31) || num_segments >= LNA_SLOT_HEAD_OPEN || num_segments * seg_out_stride >= ((int64_t)1 << 31))
    return FSF_ERR_UNSUPPORTED;
  if (x_stride < k || (out && out_stride < c)) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  LnaArgs a{x, x_stride, (int)k, (const uint4*)planes, bias, gamma, beta, eps, (int)norm, (int)act, out, out_stride, n, (int)c,
            row_add, row_add_index, row_add_stride, 128, (int)c, 0, seg_ids, seg_out, seg_out_stride, nullptr};
  return lna_launch(a, 1, stream, false, xf);
}

extern "C" int fsf_linear_norm_act_segmax(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                          const float* bias, const float* row_add, const int64_t* row_add_index,
                                          int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta, float eps,
                                          int32_t act, const int64_t* seg_ids, int64_t num_segments,
                                          float* seg_out, int64_t seg_out_stride, float* out, int64_t out_stride, void* stream_) {
  return lna_segmax(x, n, k, x_stride, planes, c, bias, row_add, row_add_index, row_add_stride, norm, gamma, beta, eps, act, seg_ids,
                    num_segments, seg_out, seg_out_stride, out, out_stride, stream_, false);
}

extern "C" int fsf_linear_f16w_norm_act_segmax(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* w_planes, int32_t c,
                                               const float* bias, const float* row_add, const int64_t* row_add_index,
                                               int64_t row_add_stride, int32_t norm, const float* gamma, const float* beta, float eps,
                                               int32_t act, const int64_t* seg_ids, int64_t num_segments,
                                               float* seg_out, int64_t seg_out_stride, float* out, int64_t out_stride, void* stream_) {
  return lna_segmax(x, n, k, x_stride, w_planes, c, bias, row_add, row_add_index, row_add_stride, norm, gamma, beta, eps, act, seg_ids,
                    num_segments, seg_out, seg_out_stride, out, out_stride, stream_, true);
}

// ---- K22h entry points ---------------------------------------------------------------------------------------------------
extern "C" int64_t fsf_linear_prepared_weight_f16_bytes(int32_t k, int32_t c, int32_t slice_c) {
  if (k < 1 || c < 1 || slice_c < 1 || slice_c > 128) return 0;
  const int64_t nkc = (k + LNA_KC - 1) / LNA_KC, nslice = (c + slice_c - 1) / slice_c;
  return 256 + nslice * nkc * lna_tiles(slice_c) * 2 * 64 * 16;
}

extern "C" int fsf_linear_prepare_weight_f16(const float* weight, int32_t k, int32_t c, int32_t slice_c, void* planes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !planes || k < 1 || c < 1 || slice_c < 1 || slice_c > 128) return FSF_ERR_INVALID_ARG;
  if (((uintptr_t)planes % 16) != 0) return FSF_ERR_UNSUPPORTED;
  const int T = lna_tiles(slice_c), nkc = (k + LNA_KC - 1) / LNA_KC, nslice = (c + slice_c - 1) / slice_c;
  FSF_HIP_TRY(hipMemsetAsync(planes, 0, 256, stream));
  const int64_t nw = (int64_t)c * k;
  // (at most 256 workgroups: each ends in ONE atomic on the same word — 2 048 of them took 25 us for a 1 M-element weight)
  hipLaunchKernelGGL(lna_weight_absmax_kernel, dim3(std::min(fsf_stream_grid(nw, 256), 256)), dim3(256), 0, stream, weight, nw, (unsigned*)planes);
  FSF_LAUNCH_CHECK();
  const int64_t total = (int64_t)nslice * nkc * T * 64;
  hipLaunchKernelGGL(lna_prepare_f16_kernel, dim3(fsf_stream_grid(total, 256)), dim3(256), 0, stream, weight, (int)k, (int)c, T, nkc,
                     nslice, (int)slice_c, (float*)planes, (uint4*)planes + 16);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int64_t fsf_row_planes_bytes(int64_t n, int32_t c) { return (n < 0 || c < 8 || (c % 8) != 0) ? 0 : n * (int64_t)c * 4; }

extern "C" int fsf_rows_to_planes(const float* x, int64_t n, int32_t c, int64_t x_stride, int32_t norm, const float* gamma,
                                  const float* beta, float eps, int32_t act, void* planes, float* inv_scales, float* out,
                                  int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 8 || norm < 0 || norm > 1 || act < 0 || act > 2 || (norm == 1 && (!gamma || !beta)) ||
      (n > 0 && (!x || !planes || !inv_scales)))
    return FSF_ERR_INVALID_ARG;
  if ((c % 8) != 0 || c > 2048 || (x_stride % 4) != 0 || x_stride < c || ((uintptr_t)x % 16) != 0 || ((uintptr_t)planes % 16) != 0 ||
      (out && ((out_stride % 4) != 0 || out_stride < c || ((uintptr_t)out % 16) != 0)) ||
      (norm == 1 && (((uintptr_t)gamma % 16) != 0 || ((uintptr_t)beta % 16) != 0)))
    return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  const int bl = (c / 8 + 63) / 64;  // 8-channel blocks per lane
  int64_t g = (n + 3) / 4;
  if (g > 16384) g = 16384;
#define FSF_R2P(BL_, N_, A_)                                                                                                     \
  hipLaunchKernelGGL((rows_to_planes_kernel<BL_, N_, A_>), dim3((unsigned)g), dim3(256), 0, stream, x, n, (int)c, x_stride, gamma, \
                     beta, eps, (uint4*)planes, inv_scales, out, out_stride)
#define FSF_R2P_ACT(BL_, N_)                      \
  do {                                            \
    if (act == 0) FSF_R2P(BL_, N_, 0);            \
    else if (act == 1) FSF_R2P(BL_, N_, 1);       \
    else FSF_R2P(BL_, N_, 2);                     \
  } while (0)
  if (bl <= 1) { if (norm) FSF_R2P_ACT(1, 1); else FSF_R2P_ACT(1, 0); }
  else if (bl <= 2) { if (norm) FSF_R2P_ACT(2, 1); else FSF_R2P_ACT(2, 0); }
  else { if (norm) FSF_R2P_ACT(4, 1); else FSF_R2P_ACT(4, 0); }
#undef FSF_R2P_ACT
#undef FSF_R2P
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_linear_planes_norm_act(const void* x_planes, const float* x_inv_scales, int64_t n, int32_t k, const void* w_planes,
                                          int32_t c, int32_t slice_c, const float* bias, int32_t norm, const float* gamma,
                                          const float* beta, float eps, int32_t act, float* out, int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || k < 1 || c < 1 || slice_c < 1 || !w_planes || norm < 0 || norm > 2 || act < 0 || act > 2 ||
      (norm != 0 && (!gamma || !beta)) || (n > 0 && (!x_planes || !x_inv_scales || !out)))
    return FSF_ERR_INVALID_ARG;
  // whole 32-k chunks; 128-channel tiles (slice_c in 68..128 so that the launched variant is the 8-tile one); a LayerNorm spans ONE slice
  if ((k % 32) != 0 || slice_c > 128 || slice_c <= 64 || (slice_c % 4) != 0 || (c % 4) != 0 || (out_stride % 4) != 0 ||
      ((uintptr_t)x_planes % 16) != 0 || ((uintptr_t)w_planes % 16) != 0 || ((uintptr_t)out % 16) != 0)
    return FSF_ERR_UNSUPPORTED;
  if (out_stride < c) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  const int nslice = (c + slice_c - 1) / slice_c;
  LnaArgs a{(const float*)x_planes, (int64_t)k, (int)k, (const uint4*)w_planes, bias, gamma, beta, eps, (int)norm, (int)act, out, out_stride, n,
            (int)c, nullptr, nullptr, 0, (int)slice_c, (int)slice_c, 0, nullptr, nullptr, 0, x_inv_scales};
  return lna_launch(a, nslice, stream, true);
}

#ifdef FSF_LNA_TIMELINE
// profiling build only: clocks wave 0 of every workgroup spent per phase since the last reset ([8] = workgroups counted)
extern "C" int fsf_debug_lna_timeline(unsigned long long* host16, int reset) {
  if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(fsf::lna_tl), 16 * sizeof(unsigned long long)) != hipSuccess) return FSF_ERR_HIP;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(fsf::lna_tl), z, sizeof(z)) != hipSuccess) return FSF_ERR_HIP;
  }
  return FSF_OK;
}
#endif
