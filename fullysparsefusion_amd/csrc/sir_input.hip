// K21: the input side of a SIR layer in one pass:
//   out = cat(points / xyz_normalizer, feats, extra / extra_div) * rel_mlp(f_cluster / rel_div)
// Replaces: the per-block `torch.cat([points, out_feats], 1)` of SIR.forward (projects/mmdet3d_plugin/models/backbones/
//   sir.py:72-74; fsd_bbox_head.py:129-132 for the refine head), and in SIRLayer / DynamicClusterVFE [UNVENDORED] the
//   xyz normalisation `cat([f[:, :3] / normalizer, f[:, 3:]])`, the position MLP `rel_mlp(f_cluster / rel_dist_scaler)`
//   = 3 x (Linear(no bias) -> LayerNorm -> GELU/ReLU) built by build_mlp (ops/sst_ops.py:808-833) and the
//   `features * rel` product: two concat copies, three skinny GEMMs (K = 3|13, 16, 32: far too thin for a GEMM library),
//   three norm/act passes and a multiply — about ten launches and ~12 trips of the [n, C] activations through HBM per
//   block, twelve blocks per frame — become one read of the sources and one write of the GEMM input.
// HBM-bound: 4(P + Cf + Ce + R) B/row read + 4C B/row written (see the kernel for the two-phase lane mapping).
#include "common.h"

namespace fsf {

constexpr int SI_MAX_R = 16;   // f_cluster columns
constexpr int SI_MAX_H1 = 16;
constexpr int SI_MAX_H2 = 32;

struct SirInputArgs {
  const float* points; int64_t points_stride; int p_cols;
  const float* feats;  int64_t feats_stride;  int f_cols;
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

__device__ __forceinline__ float si_act(float y, int act) {
  if (act == 1) return fmaxf(y, 0.0f);
  if (act == 2) return 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
  return y;
}

// A wave owns 64 consecutive rows.  Phase A, lane = row: the two thin layers of the position MLP entirely in registers
// (weights are wave-uniform LDS broadcasts, LayerNorm needs no cross-lane traffic); the 32 hidden values go to LDS.
// Phase B, lane = channel: row by row, the last layer (transposed weight in LDS), LayerNorm over the wave, activation,
// product with the concatenated sources, coalesced store.  Rows of phase B are independent: two are interleaved to hide
// the reduction latency.
template <int T>
__global__ void __launch_bounds__(256) sir_input_kernel(SirInputArgs a) {
  constexpr int CP = T * 64;
  __shared__ float w3t[SI_MAX_H2 * CP];  // [hidden unit][channel]: 64 consecutive channels per read, conflict-free
  __shared__ __attribute__((aligned(16))) float sw1[SI_MAX_H1 * SI_MAX_R], sw2[SI_MAX_H2 * SI_MAX_H1];
  __shared__ float sg1[SI_MAX_H1], sb1[SI_MAX_H1], sg2[SI_MAX_H2], sb2[SI_MAX_H2];
  __shared__ __attribute__((aligned(16))) float hs[4][SI_MAX_H2][64];  // [wave][hidden unit of layer 2][row of the wave's tile]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < SI_MAX_H1 * SI_MAX_R; t += 256) {
    const int j = t / SI_MAX_R, r = t % SI_MAX_R;
    sw1[t] = (j < a.h1 && r < a.r_cols) ? a.w1[j * a.r_cols + r] : 0.0f;
  }
  for (int t = threadIdx.x; t < SI_MAX_H2 * SI_MAX_H1; t += 256) {
    const int j = t / SI_MAX_H1, k = t % SI_MAX_H1;
    sw2[t] = (j < a.h2 && k < a.h1) ? a.w2[j * a.h1 + k] : 0.0f;
  }
  if (threadIdx.x < SI_MAX_H1) {
    sg1[threadIdx.x] = threadIdx.x < a.h1 ? a.g1[threadIdx.x] : 0.f;
    sb1[threadIdx.x] = threadIdx.x < a.h1 ? a.b1[threadIdx.x] : 0.f;
  }
  if (threadIdx.x < SI_MAX_H2) {
    sg2[threadIdx.x] = threadIdx.x < a.h2 ? a.g2[threadIdx.x] : 0.f;
    sb2[threadIdx.x] = threadIdx.x < a.h2 ? a.b2[threadIdx.x] : 0.f;
  }
  for (int t = threadIdx.x; t < SI_MAX_H2 * CP; t += 256) {
    const int k = t / CP, c = t - k * CP;
    w3t[t] = (c < a.c && k < a.h2) ? a.w3[(int64_t)c * a.h2 + k] : 0.0f;
  }
  float g3[T], b3[T];  // last layer: lane owns channels lane + 64 t
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int c = lane + 64 * t;
    g3[t] = c < a.c ? a.g3[c] : 0.f;
    b3[t] = c < a.c ? a.b3[c] : 0.f;
  }
  __syncthreads();
  const float inv_c = 1.0f / (float)a.c;
  const int64_t tiles = (a.n + 63) / 64;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row0 = tile * 64;
    const int nrow = (int)min((int64_t)64, a.n - row0);
    // ---- phase A: lane = row.  Hidden values live in this wave's LDS slice (column = lane: conflict-free), the loops
    // over hidden units stay rolled — unrolled, the compiler hoists all 768 weight reads into registers and spills.
    {
      const int64_t row = row0 + (lane < nrow ? lane : nrow - 1);
      float fc[SI_MAX_R];
#pragma unroll
      for (int r = 0; r < SI_MAX_R; ++r) fc[r] = r < a.r_cols ? __fdiv_rn(a.fcl[row * a.fcl_stride + r], a.rel_div) : 0.0f;
      float (*u2)[64] = hs[wave];
      float s = 0.0f;
#pragma unroll 1
      for (int j = 0; j < a.h1; ++j) {  // rolled (unrolled, the compiler hoists every weight read and runs out of registers);
        float acc = 0.0f;               // the raw values wait in this lane's LDS column, rows 0..h1-1 of the layer-2 slice
#pragma unroll
        for (int r4 = 0; r4 < SI_MAX_R / 4; ++r4) {
          const float4 w = reinterpret_cast<const float4*>(sw1)[j * (SI_MAX_R / 4) + r4];  // wave-uniform: LDS broadcast
          acc = fmaf(w.x, fc[4 * r4], acc);
          acc = fmaf(w.y, fc[4 * r4 + 1], acc);
          acc = fmaf(w.z, fc[4 * r4 + 2], acc);
          acc = fmaf(w.w, fc[4 * r4 + 3], acc);
        }
        u2[j][lane] = acc;
        s += acc;
      }
      float h1[SI_MAX_H1];
#pragma unroll
      for (int j = 0; j < SI_MAX_H1; ++j) h1[j] = j < a.h1 ? u2[j][lane] : 0.0f;
      float mean = s / (float)a.h1, q = 0.0f;
#pragma unroll
      for (int j = 0; j < SI_MAX_H1; ++j) {
        const float d = j < a.h1 ? h1[j] - mean : 0.0f;
        q += d * d;
      }
      float rstd = rsqrtf(q / (float)a.h1 + a.eps);
#pragma unroll
      for (int j = 0; j < SI_MAX_H1; ++j) h1[j] = j < a.h1 ? si_act((h1[j] - mean) * rstd * sg1[j] + sb1[j], a.act) : 0.0f;
      s = 0.0f;
#pragma unroll 2
      for (int j = 0; j < a.h2; ++j) {  // rolled: unrolled, the compiler hoists all 512 weight reads and spills
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < SI_MAX_H1 / 4; ++k4) {
          const float4 w = reinterpret_cast<const float4*>(sw2)[j * (SI_MAX_H1 / 4) + k4];
          acc = fmaf(w.x, h1[4 * k4], acc);  // k order = the GEMM's
          acc = fmaf(w.y, h1[4 * k4 + 1], acc);
          acc = fmaf(w.z, h1[4 * k4 + 2], acc);
          acc = fmaf(w.w, h1[4 * k4 + 3], acc);
        }
        u2[j][lane] = acc;
        s += acc;
      }
      mean = s / (float)a.h2;
      q = 0.0f;
#pragma unroll 4
      for (int j = 0; j < a.h2; ++j) {
        const float d = u2[j][lane] - mean;
        q += d * d;
      }
      rstd = rsqrtf(q / (float)a.h2 + a.eps);
#pragma unroll 2
      for (int j = 0; j < a.h2; ++j) u2[j][lane] = si_act((u2[j][lane] - mean) * rstd * sg2[j] + sb2[j], a.act);
    }
    // (hs[wave] is private to this wave: no workgroup barrier, the LDS writes are ordered before the reads below)
    // ---- phase B: lane = channel
    // eight rows of sources are requested before the first of them is needed: with ~0.7 KB per row and a handful of
    // waves per CU, fewer rows in flight leave the HBM pipe mostly empty
    constexpr int RB = 8;
    for (int i0 = 0; i0 < nrow; i0 += RB) {
      float x[RB][T];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int64_t row = row0 + min(i0 + rr, nrow - 1);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int c = lane + 64 * t;
          float v = 0.0f;
          if (c < a.p_cols) v = a.points[row * a.points_stride + c];
          else if (c < a.p_cols + a.f_cols) v = a.feats[row * a.feats_stride + (c - a.p_cols)];
          else if (c < a.c) v = a.extra[row * a.extra_stride + (c - a.p_cols - a.f_cols)];
          x[rr][t] = v;
        }
      }
      // last layer for the eight rows at once: one read of the weight column block feeds all of them.  (The kernel is
      // VALU-bound — a wave64 VALU op occupies its SIMD for 4 cycles, ~300 of them per row — and pairing rows into
      // v_pk_fma_f32 was measured 2x SLOWER on gfx950, so these stay scalar FMAs.)
      float y[RB][T];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr)
#pragma unroll
        for (int t = 0; t < T; ++t) y[rr][t] = 0.0f;
#pragma unroll 2
      for (int k = 0; k < a.h2; ++k) {
        float w[T];
#pragma unroll
        for (int t = 0; t < T; ++t) w[t] = w3t[k * CP + lane + 64 * t];
        const float4 ha = *reinterpret_cast<const float4*>(&hs[wave][k][i0]);  // rows i0..i0+7 (tile rows past nrow hold
        const float4 hb = *reinterpret_cast<const float4*>(&hs[wave][k][i0 + 4]);  // the clamped last row: finite)
        const float hk[RB] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int t = 0; t < T; ++t) y[rr][t] = fmaf(w[t], hk[rr], y[rr][t]);
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int i = i0 + rr;
        if (i < nrow) {  // wave-uniform
          float s = 0.0f;
#pragma unroll
          for (int t = 0; t < T; ++t) s += (lane + 64 * t < a.c) ? y[rr][t] : 0.0f;
          const float mean = fsf_wave_sum(s) * inv_c;
          float q = 0.0f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const float d = (lane + 64 * t < a.c) ? y[rr][t] - mean : 0.0f;
            q += d * d;
          }
          const float rstd = rsqrtf(fsf_wave_sum(q) * inv_c + a.eps);
          float* orow = a.out + (row0 + i) * a.out_stride;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int c = lane + 64 * t;
            float xv = x[rr][t];
            if (t == 0 && c < 3) xv = __fdiv_rn(xv, a.norm[c]);  // true divisions, as the reference's `/`
            if (a.e_cols > 0 && c >= a.p_cols + a.f_cols) xv = __fdiv_rn(xv, a.extra_div);  // (wave-uniform guard first)
            if (c < a.c) orow[c] = xv * si_act((y[rr][t] - mean) * rstd * g3[t] + b3[t], a.act);
          }
        }
      }
    }
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_sir_input(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                             const float* feats, int64_t feats_stride, int32_t f_cols, const float* extra,
                             int64_t extra_stride, int32_t e_cols, float extra_div, const float* f_cluster,
                             int64_t f_cluster_stride, int32_t r_cols, float rel_div, const float* w1, const float* g1,
                             const float* b1, int32_t h1, const float* w2, const float* g2, const float* b2, int32_t h2,
                             const float* w3, const float* g3, const float* b3, float eps, int32_t act, int64_t n,
                             float* out, int64_t out_stride, void* stream_) {
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
  a.extra = extra; a.extra_stride = extra_stride; a.e_cols = e_cols; a.extra_div = extra_div;
  a.fcl = f_cluster; a.fcl_stride = f_cluster_stride; a.r_cols = r_cols; a.rel_div = rel_div;
  for (int i = 0; i < 3; ++i) a.norm[i] = xyz_normalizer[i];
  a.w1 = w1; a.g1 = g1; a.b1 = b1; a.h1 = h1;
  a.w2 = w2; a.g2 = g2; a.b2 = b2; a.h2 = h2;
  a.w3 = w3; a.g3 = g3; a.b3 = b3;
  a.eps = eps; a.act = act; a.out = out; a.out_stride = out_stride; a.n = n; a.c = c;
  const int t = (c + 63) / 64;
  int64_t g = ((n + 63) / 64 + 3) / 4;
  if (g > 4096) g = 4096;
#define FSF_SI(T_) hipLaunchKernelGGL((sir_input_kernel<T_>), dim3((unsigned)g), dim3(256), 0, stream, a)
  if (t == 1) FSF_SI(1);
  else if (t == 2) FSF_SI(2);
  else if (t == 3) FSF_SI(3);
  else FSF_SI(4);
#undef FSF_SI
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
