// K21: the input side of a SIR layer in one pass:
//   out = cat(points / xyz_normalizer, feats, extra / extra_div) * rel_mlp(f_cluster / rel_div)
// Replaces: the per-block `torch.cat([points, out_feats], 1)` of SIR.forward (projects/mmdet3d_plugin/models/backbones/
//   sir.py:72-74; fsd_bbox_head.py:129-132 for the refine head), and in SIRLayer / DynamicClusterVFE [UNVENDORED] the
//   xyz normalisation `cat([f[:, :3] / normalizer, f[:, 3:]])`, the position MLP `rel_mlp(f_cluster / rel_dist_scaler)`
//   = 3 x (Linear(no bias) -> LayerNorm -> GELU/ReLU) built by build_mlp (ops/sst_ops.py:808-833) and the
//   `features * rel` product: two concat copies, three skinny GEMMs (K = 3|13, 16, 32: far too thin for a GEMM library),
//   three norm/act passes and a multiply — about ten launches and ~12 trips of the [n, C] activations through HBM per
//   block, twelve blocks per frame — become one read of the sources and one write of the GEMM input.
// Bytes: 4(P + Cf + Ce + R) B/row read + 4C B/row written; the position MLP runs on the fp32 matrix cores.
#include "common.h"

namespace fsf {

constexpr int SI_MAX_R = 16;   // f_cluster columns
constexpr int SI_MAX_H1 = 16;
constexpr int SI_MAX_H2 = 32;

struct SirInputArgs {
  const float* points; int64_t points_stride; int p_cols;
  const float* feats;  int64_t feats_stride;  int f_cols;   // f_cols = all feature columns (the sum over the parts below)
  // the feature columns may come from up to three tensors side by side (parts 1, 2 follow part 0), and their rows may be taken
  // through an index (row i of the layer input = row feats_index[i] of every part): the gather of the group-sampled points and
  // the [n, 11 + 33 + 131] concat the reference materialises before its first SIR layer happen in this kernel's loads
  const float* feats1; int64_t feats1_stride; int f0_cols, f1_cols;
  const float* feats2; int64_t feats2_stride;
  const int64_t* feats_index;
  int direct_mask;  // bit p: part p is NOT read through feats_index (its rows are already the layer's rows)
  const float* extra;  int64_t extra_stride;  int e_cols; float extra_div;
  const float* fcl;    int64_t fcl_stride;    int r_cols; float rel_div;
  float norm[3];
  const float *w1, *g1, *b1; int h1;
  const float *w2, *g2, *b2; int h2;
  const float *w3, *g3, *b3;
  float eps; int act;
  float* out; int64_t out_stride;
  int64_t n; int c;
};

// GELU: the library's one form (common.h: max(y, 0) - t 2^P(t), branch-free, one transcendental per value); the kernel is
// instruction-bound and libm's two-branch erff is 35 VALU ops + divergence per element, 60 elements per lane per 16 rows.
__device__ __forceinline__ float si_gelu(float y) { return fsf_gelu(y); }

__device__ __forceinline__ float si_act(float y, int act) {
  if (act == 1) return fmaxf(y, 0.0f);
  if (act == 2) return si_gelu(y);
  return y;
}

typedef float si_f32x4 __attribute__((ext_vector_type(4)));
typedef float si_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ si_f32x2 si_pk(float v) { return si_f32x2{v, v}; }
__device__ __forceinline__ si_f32x2 si_pk_fma(si_f32x2 a, si_f32x2 b, si_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// The same GELU on TWO values per lane (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue at the rate of their scalar forms).
__device__ __forceinline__ si_f32x2 si_gelu2(si_f32x2 y) { return fsf_gelu2(y); }

__device__ __forceinline__ si_f32x2 si_act2(si_f32x2 y, int act) {
  if (act == 1) return si_f32x2{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)};
  if (act == 2) return si_gelu2(y);
  return y;
}

// sum over the 4 lanes that share a point row (lane = row + 16 * group)
__device__ __forceinline__ float si_row_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// The position MLP on the matrix cores, TRANSPOSED: out^T[channel, row] = W[channel, unit] x h^T[unit, row] with
// v_mfma_f32_16x16x4_f32, the weights as the A operand and 16 point rows as the B operand.  Lane (row = lane & 15,
// group = lane >> 4) then holds channels 16 t + 4 group + r (r = 0..3) of its row for every 16-channel tile t — which
// is exactly the B operand of the NEXT layer if that layer walks its units in the order (t, r): the three layers chain
// through registers with no cross-lane traffic, LayerNorm is an in-lane sum + two shuffles over the 4 lanes of a row,
// and the activation runs on 1/4 row per lane.  (Before: lane = row for the two thin layers and lane = channel for the
// wide one, all on the VALU — 9 k FMAs per row were ~40 % of its ~350 wave instructions per row.)
// The activated [16, C] tile goes through a per-wave LDS slice so that the product with the concatenated sources and
// the store run with lane = channel (coalesced rows).
template <int NT3, int ACT>
__global__ void __launch_bounds__(256, 2) sir_input_kernel(SirInputArgs a) {
  constexpr int TS = NT3 * 16 + 4;  // tile row stride (floats): 16-byte rows, 4-bank skew between rows
  extern __shared__ __attribute__((aligned(16))) char si_smem[];
  float* w3f = reinterpret_cast<float*>(si_smem);  // [NT3][2][64 lanes][4]: layer-3 weights in fragment order
  float* g3s = w3f + NT3 * 2 * 64 * 4;             // [NT3 * 16]
  float* b3s = g3s + NT3 * 16;
  float* tiles = b3s + NT3 * 16;                   // [4 waves][16 rows][TS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = lane & 15, grp = lane >> 4;

  for (int idx = threadIdx.x; idx < NT3 * 2 * 64; idx += 256) {
    const int t3 = idx >> 7, t2 = (idx >> 6) & 1, l = idx & 63;
    const int out = 16 * t3 + (l & 15), in0 = 16 * t2 + 4 * (l >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      w3f[idx * 4 + r] = (out < a.c && in0 + r < a.h2) ? a.w3[(int64_t)out * a.h2 + in0 + r] : 0.0f;
  }
  for (int t = threadIdx.x; t < NT3 * 16; t += 256) {
    g3s[t] = t < a.c ? a.g3[t] : 0.0f;
    b3s[t] = t < a.c ? a.b3[t] : 0.0f;
  }
  // layers 1 and 2: weight fragments and LayerNorm affine of this lane's channels, in registers for the whole kernel
  float w1f[4], g1r[4], b1r[4], w2f[2][4], g2r[2][4], b2r[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int in = 4 * grp + r;  // A operand: lane (m = rowl, group) supplies W[m][4 group + r]
    w1f[r] = (rowl < a.h1 && in < a.r_cols) ? a.w1[rowl * a.r_cols + in] : 0.0f;
    g1r[r] = in < a.h1 ? a.g1[in] : 0.0f;  // D: the same lane holds channel 4 group + r
    b1r[r] = in < a.h1 ? a.b1[in] : 0.0f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      w2f[t2][r] = (16 * t2 + rowl < a.h2 && in < a.h1) ? a.w2[(16 * t2 + rowl) * a.h1 + in] : 0.0f;
      g2r[t2][r] = 16 * t2 + in < a.h2 ? a.g2[16 * t2 + in] : 0.0f;
      b2r[t2][r] = 16 * t2 + in < a.h2 ? a.b2[16 * t2 + in] : 0.0f;
    }
  }
  __syncthreads();

  // lane = channel view of the concatenated sources: column lane + 64 t lives in ONE of the three tensors
  constexpr int T = (NT3 * 16 + 63) / 64;
  const float* xsrc[T];
  int64_t xstride[T];
  float xdiv[T];
  bool xgath[T];  // this lane's column of tile t is read through feats_index
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int c = lane + 64 * t;
    xgath[t] = false;
    xsrc[t] = a.points;  // (columns >= c read a valid address and are never stored)
    xstride[t] = a.points_stride;
    xdiv[t] = 1.0f;
    if (c < a.p_cols) {
      xsrc[t] = a.points + c;
      if (c < 3) xdiv[t] = a.norm[c];
    } else if (c < a.p_cols + a.f_cols) {
      const int fc = c - a.p_cols;
      int part = 0;
      if (fc < a.f0_cols) {
        xsrc[t] = a.feats + fc;
        xstride[t] = a.feats_stride;
      } else if (fc < a.f0_cols + a.f1_cols) {
        xsrc[t] = a.feats1 + (fc - a.f0_cols);
        xstride[t] = a.feats1_stride;
        part = 1;
      } else {
        xsrc[t] = a.feats2 + (fc - a.f0_cols - a.f1_cols);
        xstride[t] = a.feats2_stride;
        part = 2;
      }
      xgath[t] = a.feats_index != nullptr && !((a.direct_mask >> part) & 1);
    } else if (c < a.c) {
      xsrc[t] = a.extra + (c - a.p_cols - a.f_cols);
      xstride[t] = a.extra_stride;
      xdiv[t] = a.extra_div;
    }
  }
  bool xneed[T];  // wave-uniform: does 64-column tile t hold a column that is divided (xyz, or `extra`)?
#pragma unroll
  for (int t = 0; t < T; ++t) xneed[t] = t == 0 || (a.e_cols > 0 && 64 * t + 63 >= a.p_cols + a.f_cols && 64 * t < a.c);
  float* tile = tiles + wave * 16 * TS;
  const float inv_h1 = 1.0f / (float)a.h1, inv_h2 = 1.0f / (float)a.h2, inv_c = 1.0f / (float)a.c;
  const int64_t groups = (a.n + 15) / 16;
  // MFMA-layout load of the layer-1 input (f_cluster) of a group: lane (row, group) reads columns 4 group + r
  auto load_fcl = [&](int64_t gi, float (&v)[4]) {
    const int64_t r0 = gi * 16;
    const int nr = (int)min((int64_t)16, a.n - r0);
    const int64_t rc = r0 + (rowl < nr ? rowl : nr - 1);  // rows past n repeat the last one (finite, never stored)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 4 * grp + r < a.r_cols ? a.fcl[rc * a.fcl_stride + 4 * grp + r] : 0.0f;
  };
  // the group's feature-row indices, lane (row, .) holds its row's: requested one group ahead like the MLP input
  auto load_idx = [&](int64_t gi) -> int {
    if (!a.feats_index) return 0;
    const int64_t r0 = gi * 16;
    const int nr = (int)min((int64_t)16, a.n - r0);
    return (int)a.feats_index[r0 + (rowl < nr ? rowl : nr - 1)];
  };
  const int64_t gstep = (int64_t)gridDim.x * 4;
  float xnext[4] = {0.f, 0.f, 0.f, 0.f};
  int inext = 0;
  if ((int64_t)blockIdx.x * 4 + wave < groups) {
    load_fcl((int64_t)blockIdx.x * 4 + wave, xnext);
    inext = load_idx((int64_t)blockIdx.x * 4 + wave);
  }
  for (int64_t gi = (int64_t)blockIdx.x * 4 + wave; gi < groups; gi += gstep) {
    const int64_t row0 = gi * 16;
    const int nrow = (int)min((int64_t)16, a.n - row0);
    // ---- the group's sources are requested first, lane = channel: 16 rows x T loads per lane stay in flight under the
    // whole position MLP (a wave has only one other wave on its SIMD to hide HBM latency behind)
    // (the MLP input of the NEXT group is requested here too: it is the first thing a group needs, and waiting for it
    // with nothing else to do was 40 % of the wave cycles)
    float xin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xin[r] = xnext[r];
    const int icur = inext;
    if (gi + gstep < groups) {
      load_fcl(gi + gstep, xnext);
      inext = load_idx(gi + gstep);
    }
    float x[16][T];
    if (a.feats_index) {  // (wave-uniform) gathered feature rows: row i of the group reads row readlane(icur, i) of the feature parts
      // Address of (row i, tile t) = cur[t] + ri * gstr[t]: the lane's columns that are NOT gathered walk down the group by their
      // stride (cur[t], one 64-bit add per load), the gathered ones add the row index — a scalar — times their stride in ONE
      // v_mad_u64_u32 (32 x 32 + 64 bits: indices and strides fit 32 bits, checked at launch).  As `(gathered ? ri : rr) * stride` in
      // 64 bits every load cost two selects and a 64 x 64-bit multiply: ~190 quarter-rate integer ops per 16-row group.
      const float* cur[T];
      uint32_t gstr[T];
      int64_t step[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        cur[t] = xgath[t] ? xsrc[t] : xsrc[t] + row0 * xstride[t];
        gstr[t] = xgath[t] ? (uint32_t)xstride[t] : 0u;
        step[t] = xgath[t] ? 0 : xstride[t];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane(icur, i);   // (rows past n repeat the last one: see load_idx)
#pragma unroll
        for (int t = 0; t < T; ++t) {
#ifdef SI_ABL_NO_SRC  // ablation: no source loads (WRONG results)
          x[i][t] = __uint_as_float(ri + t);
#else
          x[i][t] = *(cur[t] + (uint64_t)ri * (uint64_t)gstr[t]);
#endif
          if (i + 1 < nrow) cur[t] += step[t];  // (wave-uniform)
        }
      }
    } else {
      const float* rp[T];  // row pointers walk down the group: one 64-bit add per load instead of a 64-bit multiply
#pragma unroll
      for (int t = 0; t < T; ++t) rp[t] = xsrc[t] + row0 * xstride[t];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#ifdef SI_ABL_NO_SRC
          x[i][t] = (float)(i + t);
#else
          x[i][t] = *rp[t];
#endif
          if (i + 1 < nrow) rp[t] += xstride[t];  // (wave-uniform; rows past n repeat the last one)
        }
      }
    }
    // ---- layer 1: K = r_cols (<= 16), one MFMA per r
    si_f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) xin[r] = 4 * grp + r < a.r_cols ? __fdiv_rn(xin[r], a.rel_div) : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[r], xin[r], acc1, 0, 0, 0);
    float h1v[4];
    {
      float s = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) s += 4 * grp + r < a.h1 ? acc1[r] : 0.0f;
      const float mean = si_row_sum(s) * inv_h1;
      float q = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = 4 * grp + r < a.h1 ? acc1[r] - mean : 0.0f;
        q += d * d;
      }
      const float rstd = rsqrtf(si_row_sum(q) * inv_h1 + a.eps);
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const si_f32x2 y = si_act2((si_f32x2{acc1[r], acc1[r + 1]} - si_pk(mean)) * si_pk(rstd) * si_f32x2{g1r[r], g1r[r + 1]} +
                                   si_f32x2{b1r[r], b1r[r + 1]}, ACT);
        h1v[r] = 4 * grp + r < a.h1 ? y.x : 0.0f;
        h1v[r + 1] = 4 * grp + r + 1 < a.h1 ? y.y : 0.0f;
      }
    }
    // ---- layer 2: two 16-channel tiles, K = h1
    si_f32x4 acc2[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      acc2[t2] = si_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc2[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[t2][r], h1v[r], acc2[t2], 0, 0, 0);
    }
    float h2v[2][4];
    {
      float s = 0.0f;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += 16 * t2 + 4 * grp + r < a.h2 ? acc2[t2][r] : 0.0f;
      const float mean = si_row_sum(s) * inv_h2;
      float q = 0.0f;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = 16 * t2 + 4 * grp + r < a.h2 ? acc2[t2][r] - mean : 0.0f;
          q += d * d;
        }
      const float rstd = rsqrtf(si_row_sum(q) * inv_h2 + a.eps);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const si_f32x2 y = si_act2((si_f32x2{acc2[t2][r], acc2[t2][r + 1]} - si_pk(mean)) * si_pk(rstd) *
                                     si_f32x2{g2r[t2][r], g2r[t2][r + 1]} + si_f32x2{b2r[t2][r], b2r[t2][r + 1]}, ACT);
          h2v[t2][r] = 16 * t2 + 4 * grp + r < a.h2 ? y.x : 0.0f;
          h2v[t2][r + 1] = 16 * t2 + 4 * grp + r + 1 < a.h2 ? y.y : 0.0f;
        }
    }
    // ---- layer 3: NT3 tiles, K = h2 walked in (t2, r) order = the order the lanes hold h2v
    si_f32x4 acc3[NT3];
#pragma unroll
    for (int t3 = 0; t3 < NT3; ++t3) {
      acc3[t3] = si_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const si_f32x4 wf = *reinterpret_cast<const si_f32x4*>(w3f + ((t3 * 2 + t2) * 64 + lane) * 4);
#pragma unroll
#ifdef SI_ABL_NO_MFMA3  // ablation: layer 3's 8 NT3 fp32 MFMAs replaced by one VALU op each (WRONG results)
        for (int r = 0; r < 4; ++r) acc3[t3][r] += wf[r] * h2v[t2][r];
#else
        for (int r = 0; r < 4; ++r) acc3[t3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[r], h2v[t2][r], acc3[t3], 0, 0, 0);
#endif
      }
    }
    {
      // channels >= c have zero weights: their accumulators are exactly 0 and drop out of the sum; for the squared
      // deviations they are masked by a per-lane limit (kept opaque so the 4 NT3 compares are redone per group instead
      // of living in 2 x 4 NT3 scalar registers for the whole kernel)
      int lim = a.c - 4 * grp;
      asm volatile("" : "+v"(lim));
      si_f32x2 s2 = si_pk(0.0f);
#pragma unroll
      for (int t3 = 0; t3 < NT3; ++t3) s2 = (s2 + si_f32x2{acc3[t3][0], acc3[t3][1]}) + si_f32x2{acc3[t3][2], acc3[t3][3]};
      const float mean = si_row_sum(s2.x + s2.y) * inv_c;
      si_f32x2 q2 = si_pk(0.0f);
#pragma unroll
      for (int t3 = 0; t3 < NT3; ++t3)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          si_f32x2 d = si_f32x2{acc3[t3][r], acc3[t3][r + 1]} - si_pk(mean);
          d.x = 16 * t3 + r < lim ? d.x : 0.0f;
          d.y = 16 * t3 + r + 1 < lim ? d.y : 0.0f;
          q2 = si_pk_fma(d, d, q2);
        }
      const float rstd = rsqrtf(si_row_sum(q2.x + q2.y) * inv_c + a.eps);
#pragma unroll
      for (int t3 = 0; t3 < NT3; ++t3) {
        const int ch0 = 16 * t3 + 4 * grp;
        const float4 gv = *reinterpret_cast<const float4*>(g3s + ch0), bv = *reinterpret_cast<const float4*>(b3s + ch0);
        const si_f32x2 yl = si_act2((si_f32x2{acc3[t3][0], acc3[t3][1]} - si_pk(mean)) * si_pk(rstd) * si_f32x2{gv.x, gv.y} +
                                    si_f32x2{bv.x, bv.y}, ACT);
        const si_f32x2 yh = si_act2((si_f32x2{acc3[t3][2], acc3[t3][3]} - si_pk(mean)) * si_pk(rstd) * si_f32x2{gv.z, gv.w} +
                                    si_f32x2{bv.z, bv.w}, ACT);
        *reinterpret_cast<float4*>(tile + rowl * TS + ch0) = make_float4(yl.x, yl.y, yh.x, yh.y);  // (channels >= c: affine 0 -> act(0) = 0, never read)
      }
    }
    // ---- product with the concatenated sources, lane = channel (the tile is private to this wave: its LDS writes are
    // ordered before these reads)
    float* orow = a.out + row0 * a.out_stride + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < nrow) {  // wave-uniform
#pragma unroll
        for (int t = 0; t < T; ++t) {
          // true divisions, as the reference's `/` (x / 1 is exact), only in the tiles that hold a divided column: a REAL branch
          // (the empty volatile statement keeps the compiler from turning the wave-uniform test into a select, which ran the
          // thirteen-instruction IEEE division for every tile of every row: 48 per 16-row group at c = 180 where 16 are needed)
          float xv = x[i][t];
          if (xneed[t]) {
            asm volatile("" ::: "memory");
            xv = __fdiv_rn(xv, xdiv[t]);
          }
#ifdef SI_ABL_NO_STORE  // ablation: no output stores
          if (xv * tile[i * TS + lane + 64 * t] == 123.456f) orow[64 * t] = 0.0f;
#else
          if (lane + 64 * t < a.c) orow[64 * t] = xv * tile[i * TS + lane + 64 * t];
#endif
        }
        orow += a.out_stride;
      }
    }
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_sir_input_gather(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                                    const float* const* feat_parts, const int64_t* feat_strides, const int32_t* feat_cols,
                                    int32_t num_parts, const int64_t* feats_index, int32_t direct_parts_mask, const float* extra,
                                    int64_t extra_stride,
                                    int32_t e_cols, float extra_div, const float* f_cluster, int64_t f_cluster_stride,
                                    int32_t r_cols, float rel_div, const float* w1, const float* g1, const float* b1, int32_t h1,
                                    const float* w2, const float* g2, const float* b2, int32_t h2, const float* w3,
                                    const float* g3, const float* b3, float eps, int32_t act, int64_t n, float* out,
                                    int64_t out_stride, void* stream_);

extern "C" int fsf_sir_input(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                             const float* feats, int64_t feats_stride, int32_t f_cols, const float* extra,
                             int64_t extra_stride, int32_t e_cols, float extra_div, const float* f_cluster,
                             int64_t f_cluster_stride, int32_t r_cols, float rel_div, const float* w1, const float* g1,
                             const float* b1, int32_t h1, const float* w2, const float* g2, const float* b2, int32_t h2,
                             const float* w3, const float* g3, const float* b3, float eps, int32_t act, int64_t n,
                             float* out, int64_t out_stride, void* stream_) {
  const float* parts[1] = {feats};
  const int64_t strides[1] = {feats_stride};
  const int32_t cols[1] = {f_cols};
  return fsf_sir_input_gather(points, points_stride, p_cols, xyz_normalizer, parts, strides, cols, f_cols > 0 ? 1 : 0, nullptr, 0, extra,
                              extra_stride, e_cols, extra_div, f_cluster, f_cluster_stride, r_cols, rel_div, w1, g1, b1, h1, w2, g2,
                              b2, h2, w3, g3, b3, eps, act, n, out, out_stride, stream_);
}

extern "C" int fsf_sir_input_gather(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                                    const float* const* feat_parts, const int64_t* feat_strides, const int32_t* feat_cols,
                                    int32_t num_parts, const int64_t* feats_index, int32_t direct_parts_mask, const float* extra,
                                    int64_t extra_stride,
                                    int32_t e_cols, float extra_div, const float* f_cluster, int64_t f_cluster_stride,
                                    int32_t r_cols, float rel_div, const float* w1, const float* g1, const float* b1, int32_t h1,
                                    const float* w2, const float* g2, const float* b2, int32_t h2, const float* w3,
                                    const float* g3, const float* b3, float eps, int32_t act, int64_t n, float* out,
                                    int64_t out_stride, void* stream_) {
  if (num_parts < 0 || num_parts > 3 || (num_parts > 0 && (!feat_parts || !feat_strides || !feat_cols))) return FSF_ERR_INVALID_ARG;
  int32_t f_cols = 0;
  for (int i = 0; i < num_parts; ++i) {
    if (feat_cols[i] < 1 || feat_strides[i] < feat_cols[i] || (n > 0 && !feat_parts[i])) return FSF_ERR_INVALID_ARG;
    if (feat_strides[i] > 0xffffffffLL) return FSF_ERR_UNSUPPORTED;  // (the gather multiplies a 32-bit row index by a 32-bit stride)
    f_cols += feat_cols[i];
  }
  const float* feats = num_parts > 0 ? feat_parts[0] : nullptr;
  const int64_t feats_stride = num_parts > 0 ? feat_strides[0] : 0;
  hipStream_t stream = (hipStream_t)stream_;
  const int c = p_cols + f_cols + e_cols;
  if (n < 0 || p_cols < 3 || f_cols < 0 || e_cols < 0 || r_cols < 1 || h1 < 1 || h2 < 1 || act < 0 || act > 2 ||
      !xyz_normalizer || !w1 || !g1 || !b1 || !w2 || !g2 || !b2 || !w3 || !g3 || !b3 ||
      (n > 0 && (!points || !f_cluster || !out || (f_cols > 0 && !feats) || (e_cols > 0 && !extra))))
    return FSF_ERR_INVALID_ARG;
  if (r_cols > SI_MAX_R || h1 > SI_MAX_H1 || h2 > SI_MAX_H2 || c > 256) return FSF_ERR_UNSUPPORTED;
  if (out_stride == 0) out_stride = c;
  if (out_stride < c || points_stride < p_cols || f_cluster_stride < r_cols) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  SirInputArgs a;
  a.points = points; a.points_stride = points_stride; a.p_cols = p_cols;
  a.feats = feats; a.feats_stride = feats_stride; a.f_cols = f_cols;
  a.f0_cols = num_parts > 0 ? feat_cols[0] : 0;
  a.feats1 = num_parts > 1 ? feat_parts[1] : nullptr; a.feats1_stride = num_parts > 1 ? feat_strides[1] : 0;
  a.f1_cols = num_parts > 1 ? feat_cols[1] : 0;
  a.feats2 = num_parts > 2 ? feat_parts[2] : nullptr; a.feats2_stride = num_parts > 2 ? feat_strides[2] : 0;
  a.feats_index = num_parts > 0 ? feats_index : nullptr;
  a.direct_mask = direct_parts_mask;
  a.extra = extra; a.extra_stride = extra_stride; a.e_cols = e_cols; a.extra_div = extra_div;
  a.fcl = f_cluster; a.fcl_stride = f_cluster_stride; a.r_cols = r_cols; a.rel_div = rel_div;
  for (int i = 0; i < 3; ++i) a.norm[i] = xyz_normalizer[i];
  a.w1 = w1; a.g1 = g1; a.b1 = b1; a.h1 = h1;
  a.w2 = w2; a.g2 = g2; a.b2 = b2; a.h2 = h2;
  a.w3 = w3; a.g3 = g3; a.b3 = b3;
  a.eps = eps; a.act = act; a.out = out; a.out_stride = out_stride; a.n = n; a.c = c;
  const int64_t groups = (n + 15) / 16;
  int64_t g = (groups + 3) / 4;
  if (g > 512) g = 512;  // 2 workgroups per CU, each walks its share of the 16-row groups (the weight staging is per workgroup)
#define FSF_SI2(NT3_, ACT_)                                                                                            \
  do {                                                                                                                 \
    static std::atomic<uint64_t> attr_done{0};                                                                                      \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)sir_input_kernel<NT3_, ACT_>, (int)smem, attr_done));                                                                                                                  \
    hipLaunchKernelGGL((sir_input_kernel<NT3_, ACT_>), dim3((unsigned)g), dim3(256), smem, stream, a);                 \
  } while (0)
#define FSF_SI(NT3_)                                                                                                   \
  do {                                                                                                                 \
    constexpr size_t smem = (size_t)(NT3_ * 2 * 64 * 4 + 2 * NT3_ * 16 + 4 * 16 * (NT3_ * 16 + 4)) * sizeof(float);    \
    if (act == 2) FSF_SI2(NT3_, 2);                                                                                    \
    else if (act == 1) FSF_SI2(NT3_, 1);                                                                               \
    else FSF_SI2(NT3_, 0);                                                                                             \
  } while (0)
  if (c <= 64) FSF_SI(4);
  else if (c <= 128) FSF_SI(8);
  else if (c <= 160) FSF_SI(10);
  else if (c <= 192) FSF_SI(12);
  else FSF_SI(16);
#undef FSF_SI
#undef FSF_SI2
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// K28 (training): y = cat([points[:, :3] / normalizer, points[:, 3:], feats, extra / extra_div], 1) * h  and its adjoint.
// SIRLayer.forward (sir_layer [UNVENDORED]) in training mode: the position MLP h = rel_mlp(f_cluster / scaler) stays
// in autograd (three thin layers), but around it ATen wrote the concatenation twice (SIR.forward's cat, then the copy with the
// normalised xyz), the product, and in the backward two more products, a zero-filled slice-gradient and its accumulation:
// ~1.8 GB written per 491 k-row block.  Here the forward writes y only and the backward writes grad_h and grad_feats
// (, grad_extra) only; x is re-formed from its sources in both.  The same IEEE operations per element as the ATen chain
// (division, not multiplication by a reciprocal): bit-identical results.
namespace fsf {
struct CatMulArgs {
  const float* points; int64_t points_stride; int p_cols;
  const float* feats; int64_t feats_stride; int f_cols;
  const float* extra; int64_t extra_stride; int e_cols; float extra_div;
  float norm[3];
  const float* h;       // [n, c] contiguous
  const float* g;       // backward: grad of y [n, c] contiguous
  float* out;           // forward: y [n, c];  backward: grad_h [n, c]
  float* g_feats;       // backward (optional): [n, f_cols] contiguous
  float* g_extra;       // backward (optional): [n, e_cols] contiguous
  int64_t n; int c;
};

__device__ __forceinline__ float cm_x(const CatMulArgs& a, int64_t i, int col) {
  if (col < a.p_cols) {
    const float v = a.points[i * a.points_stride + col];
    return col < 3 ? __fdiv_rn(v, a.norm[col]) : v;
  }
  col -= a.p_cols;
  if (col < a.f_cols) return a.feats[i * a.feats_stride + col];
  col -= a.f_cols;
  return __fdiv_rn(a.extra[i * a.extra_stride + col], a.extra_div);
}

__global__ void __launch_bounds__(256) cat_mul_kernel(CatMulArgs a) {
  const int64_t total = a.n * a.c;
  const bool small = total <= 0x7fffffff;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = small ? (int64_t)((uint32_t)t / (uint32_t)a.c) : t / a.c;
    const int col = (int)(t - i * a.c);
    a.out[t] = __fmul_rn(cm_x(a, i, col), a.h[t]);
  }
}

__global__ void __launch_bounds__(256) cat_mul_bwd_kernel(CatMulArgs a) {
  const int64_t total = a.n * a.c;
  const bool small = total <= 0x7fffffff;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = small ? (int64_t)((uint32_t)t / (uint32_t)a.c) : t / a.c;
    const int col = (int)(t - i * a.c);
    const float g = a.g[t];
    a.out[t] = __fmul_rn(g, cm_x(a, i, col));  // d/dh
    const int fc = col - a.p_cols;
    if (fc >= 0) {
      const float gx = __fmul_rn(g, a.h[t]);   // d/dx
      if (fc < a.f_cols) {
        if (a.g_feats) a.g_feats[i * a.f_cols + fc] = gx;
      } else if (a.g_extra) {
        a.g_extra[i * a.e_cols + (fc - a.f_cols)] = __fdiv_rn(gx, a.extra_div);
      }
    }
  }
}
}  // namespace fsf

static int cat_mul_check(const float* points, int64_t points_stride, int32_t p_cols, const float* xyz_normalizer, const float* feats,
                         int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride, int32_t e_cols, const float* h,
                         int64_t n) {
  if (n < 0 || p_cols < 3 || f_cols < 0 || e_cols < 0 || !xyz_normalizer || points_stride < p_cols || (f_cols > 0 && feats_stride < f_cols) ||
      (e_cols > 0 && extra_stride < e_cols) || (n > 0 && (!points || !h || (f_cols > 0 && !feats) || (e_cols > 0 && !extra))))
    return FSF_ERR_INVALID_ARG;
  return FSF_OK;
}

extern "C" int fsf_concat_mul(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3], const float* feats,
                              int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride, int32_t e_cols,
                              float extra_div, const float* h, int64_t n, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = cat_mul_check(points, points_stride, p_cols, xyz_normalizer, feats, feats_stride, f_cols, extra, extra_stride, e_cols, h, n);
  if (rc != FSF_OK || (n > 0 && !out)) return rc != FSF_OK ? rc : FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  fsf::CatMulArgs a{points, points_stride, (int)p_cols, feats, feats_stride, (int)f_cols, extra, extra_stride, (int)e_cols, extra_div,
                    {xyz_normalizer[0], xyz_normalizer[1], xyz_normalizer[2]}, h, nullptr, out, nullptr, nullptr, n,
                    (int)(p_cols + f_cols + e_cols)};
  hipLaunchKernelGGL(fsf::cat_mul_kernel, dim3(fsf_stream_grid(n * a.c, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_concat_mul_backward(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                                       const float* feats, int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride,
                                       int32_t e_cols, float extra_div, const float* h, const float* grad_out, int64_t n, float* grad_h,
                                       float* grad_feats, float* grad_extra, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = cat_mul_check(points, points_stride, p_cols, xyz_normalizer, feats, feats_stride, f_cols, extra, extra_stride, e_cols, h, n);
  if (rc != FSF_OK || (n > 0 && (!grad_out || !grad_h))) return rc != FSF_OK ? rc : FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  fsf::CatMulArgs a{points, points_stride, (int)p_cols, feats, feats_stride, (int)f_cols, extra, extra_stride, (int)e_cols, extra_div,
                    {xyz_normalizer[0], xyz_normalizer[1], xyz_normalizer[2]}, h, grad_out, grad_h, grad_feats, grad_extra, n,
                    (int)(p_cols + f_cols + e_cols)};
  hipLaunchKernelGGL(fsf::cat_mul_bwd_kernel, dim3(fsf_stream_grid(n * a.c, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
