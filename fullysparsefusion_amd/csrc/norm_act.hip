// K12: fused row normalisation + activation epilogue for the per-point MLPs of the VFE / SIR blocks.
// Replaces: the `norm -> act` tail of every `Linear -> norm -> act` block built by build_mlp
//   (projects/mmdet3d_plugin/ops/sst_ops.py:808-833) and DynamicVFELayer [UNVENDORED]: upstream runs
//   LayerNorm (or eval BatchNorm) and GELU/ReLU as separate ATen kernels, i.e. the [n, C] activations make
//   3 round trips through HBM after the GEMM; here it is one read + one write.
// HBM-bound: 4C B/row read + 4C B/row written.  One lane team per row, the row lives in registers, mean and
// variance are two-pass over the registers (no E[x^2]-E[x]^2 cancellation), var = biased (LayerNorm).
#include <stdlib.h>

#include "common.h"

namespace fsf {

enum { NORM_LN = 0, NORM_AFFINE = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

constexpr int NA_MAX_PER_LANE = 8;

// The BACKWARD's terms: erf by Abramowitz & Stegun 7.1.26, branch-free (|error| <= 1.5e-7 absolute): pe = 1 - erf(|y| / sqrt 2),
// e = exp(-y^2 / 2).  (The forward is common.h's fsf_gelu; libm's erff made these passes VALU-bound: ~80 of the 173 us of the
// [4.9e5, 128] forward.)
__device__ __forceinline__ void gelu_terms(float y, float& pe, float& e) {
  const float u = fabsf(y) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __builtin_amdgcn_exp2f(u * u * -1.4426950408889634f);
  pe = p * t * e;
}
__device__ __forceinline__ float gelu_erf(float y) { return fsf_gelu(y); }  // the forward form every kernel shares (common.h)

typedef float na_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ na_f32x2 na_pk(float v) { return na_f32x2{v, v}; }
// two values per lane on v_pk_*_f32 (full rate on gfx950: half the instructions per value outside v_rcp / v_exp)
__device__ __forceinline__ na_f32x2 gelu_erf2(na_f32x2 y) { return fsf_gelu2(y); }

template <int TEAM, int ACT, int NORM, int PL = NA_MAX_PER_LANE>  // PL channels per lane (16 for the 1024-wide query MLPs)
__global__ void __launch_bounds__(256)
    norm_act_kernel(const float* x, int64_t n, int c, const float* __restrict__ gamma,
                    const float* __restrict__ beta, float eps, float* out, int64_t out_stride) {
  const int tl = threadIdx.x % TEAM;
  const int teams_per_block = 256 / TEAM;
  const float inv_c = 1.0f / (float)c;
  for (int64_t row = (int64_t)blockIdx.x * teams_per_block + threadIdx.x / TEAM; row < n;
       row += (int64_t)gridDim.x * teams_per_block) {
    const float* xr = x + row * c;
    float v[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int ch = tl + k * TEAM;
      v[k] = (ch < c) ? xr[ch] : 0.0f;
    }
    float mean = 0.0f, rstd = 1.0f;
    if (NORM == NORM_LN) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < PL; ++k) s += v[k];
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      mean = s * inv_c;
      float q = 0.0f;
#pragma unroll
      for (int k = 0; k < PL; ++k) {
        const int ch = tl + k * TEAM;
        const float d = (ch < c) ? v[k] - mean : 0.0f;
        q += d * d;
      }
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
      rstd = rsqrtf(q * inv_c + eps);
    }
    float* orow = out + row * out_stride;
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int ch = tl + k * TEAM;
      if (ch < c) {
        float y;
        if (NORM == NORM_LN) {
          y = (v[k] - mean) * rstd;
          if (gamma) y = y * gamma[ch] + beta[ch];
        } else {
          y = v[k] * gamma[ch] + beta[ch];
        }
        if (ACT == ACT_RELU) y = fmaxf(y, 0.0f);
        if (ACT == ACT_GELU) y = gelu_erf(y);
        orow[ch] = y;
      }
    }
  }
}

template <int TEAM, int PL = NA_MAX_PER_LANE>
static int launch_norm_act(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps, int norm,
                           int act, float* out, int64_t out_stride, hipStream_t stream) {
  const int teams_per_block = 256 / TEAM;
  int64_t g = (n + teams_per_block - 1) / teams_per_block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
#define FSF_NA(N_, A_) \
  hipLaunchKernelGGL((norm_act_kernel<TEAM, A_, N_, PL>), dim3((unsigned)g), dim3(256), 0, stream, x, n, c, gamma, beta, eps, out, out_stride)
  if (norm == NORM_LN) {
    if (act == ACT_GELU) FSF_NA(NORM_LN, ACT_GELU);
    else if (act == ACT_RELU) FSF_NA(NORM_LN, ACT_RELU);
    else FSF_NA(NORM_LN, ACT_NONE);
  } else {
    if (act == ACT_GELU) FSF_NA(NORM_AFFINE, ACT_GELU);
    else if (act == ACT_RELU) FSF_NA(NORM_AFFINE, ACT_RELU);
    else FSF_NA(NORM_AFFINE, ACT_NONE);
  }
#undef FSF_NA
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 16-byte variant (c % 4 == 0, 16-byte aligned rows): lane tl of a team owns channels 4 * (tl + k * TEAM) .. + 3, k < PLV, RU rows of a
// team are in flight together.  Four-byte lanes moved 256 bytes per wave instruction and sustained 2.9 TB/s on [4.9e5, 128];
// the arithmetic per element is the same, only the order of the row sums differs.
template <int TEAM, int ACT, int NORM, int PLV, int RU>
__global__ void __launch_bounds__(256)
    norm_act_v4_kernel(const float* x, int64_t n, int c, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                       float* out, int64_t out_stride) {
  constexpr int TEAMS = 256 / TEAM;
  const int tl = threadIdx.x % TEAM, team = threadIdx.x / TEAM;
  const int c4 = c >> 2;
  const float inv_c = 1.0f / (float)c;
  float4 g[PLV], b[PLV];
#pragma unroll
  for (int k = 0; k < PLV; ++k) {
    const int ch4 = tl + k * TEAM;
    g[k] = (gamma && ch4 < c4) ? reinterpret_cast<const float4*>(gamma)[ch4] : make_float4(1.f, 1.f, 1.f, 1.f);
    b[k] = (beta && ch4 < c4) ? reinterpret_cast<const float4*>(beta)[ch4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row0 = ((int64_t)blockIdx.x * TEAMS + team) * RU; row0 < n; row0 += (int64_t)gridDim.x * TEAMS * RU) {
    float4 v[RU][PLV];
#pragma unroll
    for (int u = 0; u < RU; ++u)
#pragma unroll
      for (int k = 0; k < PLV; ++k) {
        const int ch4 = tl + k * TEAM;
        v[u][k] = (ch4 < c4 && row0 + u < n) ? reinterpret_cast<const float4*>(x + (row0 + u) * c)[ch4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      float mean = 0.0f, rstd = 1.0f;
      if (NORM == NORM_LN) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < PLV; ++k) s += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
#pragma unroll
        for (int o = TEAM >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        mean = s * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < PLV; ++k) {
          if (tl + k * TEAM < c4) {
            const float d0 = v[u][k].x - mean, d1 = v[u][k].y - mean, d2 = v[u][k].z - mean, d3 = v[u][k].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
        }
#pragma unroll
        for (int o = TEAM >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
        rstd = rsqrtf(q * inv_c + eps);
      }
      if (row0 + u < n) {
#pragma unroll
        for (int k = 0; k < PLV; ++k) {
          const int ch4 = tl + k * TEAM;
          if (ch4 < c4) {
            na_f32x2 y[2] = {na_f32x2{v[u][k].x, v[u][k].y}, na_f32x2{v[u][k].z, v[u][k].w}};
            const na_f32x2 gg[2] = {na_f32x2{g[k].x, g[k].y}, na_f32x2{g[k].z, g[k].w}};
            const na_f32x2 bb[2] = {na_f32x2{b[k].x, b[k].y}, na_f32x2{b[k].z, b[k].w}};
#pragma unroll
            for (int e = 0; e < 2; ++e) {  // (pairs: the same IEEE operations per value as the scalar form)
              if (NORM == NORM_LN) {
                y[e] = (y[e] - na_pk(mean)) * na_pk(rstd);
                if (gamma) y[e] = y[e] * gg[e] + bb[e];
              } else {
                y[e] = y[e] * gg[e] + bb[e];
              }
              if (ACT == ACT_RELU) y[e] = na_f32x2{fmaxf(y[e].x, 0.0f), fmaxf(y[e].y, 0.0f)};
              if (ACT == ACT_GELU) y[e] = gelu_erf2(y[e]);
            }
            reinterpret_cast<float4*>(out + (row0 + u) * out_stride)[ch4] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
          }
        }
      }
    }
  }
}

template <int TEAM, int PLV, int RU>
static int launch_norm_act_v4(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps, int norm,
                              int act, float* out, int64_t out_stride, hipStream_t stream) {
  const int rows_per_block = 256 / TEAM * RU;
  int64_t g = (n + rows_per_block - 1) / rows_per_block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
#define FSF_NA4(N_, A_) \
  hipLaunchKernelGGL((norm_act_v4_kernel<TEAM, A_, N_, PLV, RU>), dim3((unsigned)g), dim3(256), 0, stream, x, n, c, gamma, beta, eps, out, out_stride)
  if (norm == NORM_LN) {
    if (act == ACT_GELU) FSF_NA4(NORM_LN, ACT_GELU);
    else if (act == ACT_RELU) FSF_NA4(NORM_LN, ACT_RELU);
    else FSF_NA4(NORM_LN, ACT_NONE);
  } else {
    if (act == ACT_GELU) FSF_NA4(NORM_AFFINE, ACT_GELU);
    else if (act == ACT_RELU) FSF_NA4(NORM_AFFINE, ACT_RELU);
    else FSF_NA4(NORM_AFFINE, ACT_NONE);
  }
#undef FSF_NA4
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of y = act(LayerNorm(x)) in one pass over (x, grad_out): the row statistics and the pre-activation are
// recomputed from x in registers (nothing but x is kept from the forward), d(act) is applied, the two row means of the
// LayerNorm gradient are team reductions, grad_x is written once.  grad_gamma / grad_beta are column sums over all
// rows: every lane owns fixed channels, accumulates them over the rows its team walks, teams of a workgroup are folded
// through LDS in team order, workgroups write partials that a second kernel folds in workgroup order (deterministic).
// Replaces, in training, ATen's layer_norm_backward (2 kernels + a column reduction) and the separate GELU backward.
__device__ __forceinline__ float dgelu_erf(float y) {
  float pe, e;
  gelu_terms(y, pe, e);
  const float cdf = y < 0.0f ? 0.5f * pe : 1.0f - 0.5f * pe;
  return cdf + y * (0.39894228040143267794f * e);
}

template <int TEAM, int ACT>
__global__ void __launch_bounds__(256)
    norm_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, int64_t n, int c,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ gx,
                        float* __restrict__ part /* [gridDim.x][2][c] */) {
  constexpr int TEAMS = 256 / TEAM;
  __shared__ float red[TEAMS][2][TEAM * NA_MAX_PER_LANE + 1];
  const int tl = threadIdx.x % TEAM, team = threadIdx.x / TEAM;
  const float inv_c = 1.0f / (float)c;
  float g[NA_MAX_PER_LANE], b[NA_MAX_PER_LANE], dg[NA_MAX_PER_LANE], db[NA_MAX_PER_LANE];
#pragma unroll
  for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
    const int ch = tl + k * TEAM;
    g[k] = (ch < c && gamma) ? gamma[ch] : 1.0f;
    b[k] = (ch < c && beta) ? beta[ch] : 0.0f;
    dg[k] = db[k] = 0.0f;
  }
  for (int64_t row = (int64_t)blockIdx.x * TEAMS + team; row < n; row += (int64_t)gridDim.x * TEAMS) {
    const float* xr = x + row * c;
    const float* gr = gout + row * c;
    float v[NA_MAX_PER_LANE], go[NA_MAX_PER_LANE];
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
      const int ch = tl + k * TEAM;
      v[k] = ch < c ? xr[ch] : 0.0f;
      go[k] = ch < c ? gr[ch] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) s += v[k];
#pragma unroll
    for (int o = TEAM >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
      const float d = (tl + k * TEAM < c) ? v[k] - mean : 0.0f;
      q += d * d;
    }
#pragma unroll
    for (int o = TEAM >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_c + eps);
    float s1 = 0.0f, s2 = 0.0f;  // sum(g_xhat), sum(g_xhat * xhat)
    float xh[NA_MAX_PER_LANE], gh[NA_MAX_PER_LANE];
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
      const bool live = tl + k * TEAM < c;
      xh[k] = live ? (v[k] - mean) * rstd : 0.0f;
      const float y = xh[k] * g[k] + b[k];
      float gy = go[k];
      if (ACT == ACT_RELU) gy = y > 0.0f ? gy : 0.0f;
      if (ACT == ACT_GELU) gy *= dgelu_erf(y);
      gy = live ? gy : 0.0f;
      db[k] += gy;
      dg[k] += gy * xh[k];
      gh[k] = gy * g[k];
      s1 += gh[k];
      s2 += gh[k] * xh[k];
    }
#pragma unroll
    for (int o = TEAM >> 1; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float m1 = s1 * inv_c, m2 = s2 * inv_c;
    float* gxr = gx + row * c;
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
      const int ch = tl + k * TEAM;
      if (ch < c) gxr[ch] = rstd * (gh[k] - m1 - xh[k] * m2);
    }
  }
  // column sums: teams -> workgroup (team order), workgroup partial to memory
#pragma unroll
  for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
    red[team][0][tl + k * TEAM] = dg[k];
    red[team][1][tl + k * TEAM] = db[k];
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int t = 0; t < TEAMS; ++t) {
      a0 += red[t][0][ch];
      a1 += red[t][1][ch];
    }
    part[((int64_t)blockIdx.x * 2 + 0) * c + ch] = a0;
    part[((int64_t)blockIdx.x * 2 + 1) * c + ch] = a1;
  }
}

// 16-byte variant of the backward pass (see norm_act_v4_kernel): the same arithmetic per element, the same fixed fold order of
// the column sums (teams of a workgroup in team order, workgroups in the fold kernel).
template <int TEAM, int ACT, int PLV, int RU>
__global__ void __launch_bounds__(256)
    norm_act_bwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ gout, int64_t n, int c,
                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ gx,
                           float* __restrict__ part /* [gridDim.x][2][c] */) {
  constexpr int TEAMS = 256 / TEAM;
  __shared__ float red[TEAMS][2][TEAM * 4 * PLV + 1];
  const int tl = threadIdx.x % TEAM, team = threadIdx.x / TEAM;
  const int c4 = c >> 2;
  const float inv_c = 1.0f / (float)c;
  float g[PLV][4], b[PLV][4], dg[PLV][4], db[PLV][4];
#pragma unroll
  for (int k = 0; k < PLV; ++k) {
    const int ch4 = tl + k * TEAM;
    const float4 g4 = (gamma && ch4 < c4) ? reinterpret_cast<const float4*>(gamma)[ch4] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b4 = (beta && ch4 < c4) ? reinterpret_cast<const float4*>(beta)[ch4] : make_float4(0.f, 0.f, 0.f, 0.f);
    g[k][0] = g4.x; g[k][1] = g4.y; g[k][2] = g4.z; g[k][3] = g4.w;
    b[k][0] = b4.x; b[k][1] = b4.y; b[k][2] = b4.z; b[k][3] = b4.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[k][e] = db[k][e] = 0.0f;
  }
  for (int64_t row0 = ((int64_t)blockIdx.x * TEAMS + team) * RU; row0 < n; row0 += (int64_t)gridDim.x * TEAMS * RU) {
    float4 v4[RU][PLV], go4[RU][PLV];
#pragma unroll
    for (int u = 0; u < RU; ++u)
#pragma unroll
      for (int k = 0; k < PLV; ++k) {
        const int ch4 = tl + k * TEAM;
        const bool live = ch4 < c4 && row0 + u < n;
        v4[u][k] = live ? reinterpret_cast<const float4*>(x + (row0 + u) * c)[ch4] : make_float4(0.f, 0.f, 0.f, 0.f);
        go4[u][k] = live ? reinterpret_cast<const float4*>(gout + (row0 + u) * c)[ch4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      float v[PLV][4], go[PLV][4];
#pragma unroll
      for (int k = 0; k < PLV; ++k) {
        v[k][0] = v4[u][k].x; v[k][1] = v4[u][k].y; v[k][2] = v4[u][k].z; v[k][3] = v4[u][k].w;
        go[k][0] = go4[u][k].x; go[k][1] = go4[u][k].y; go[k][2] = go4[u][k].z; go[k][3] = go4[u][k].w;
      }
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < PLV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s * inv_c;
      float q = 0.0f;
#pragma unroll
      for (int k = 0; k < PLV; ++k)
        if (tl + k * TEAM < c4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[k][e] - mean;
            q += d * d;
          }
        }
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * inv_c + eps);
      float s1 = 0.0f, s2 = 0.0f;  // sum(g_xhat), sum(g_xhat * xhat)
      float xh[PLV][4], gh[PLV][4];
      const bool row_live = row0 + u < n;
#pragma unroll
      for (int k = 0; k < PLV; ++k) {
        const bool live = row_live && tl + k * TEAM < c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[k][e] = live ? (v[k][e] - mean) * rstd : 0.0f;
          const float y = xh[k][e] * g[k][e] + b[k][e];
          float gy = go[k][e];
          if (ACT == ACT_RELU) gy = y > 0.0f ? gy : 0.0f;
          if (ACT == ACT_GELU) gy *= dgelu_erf(y);
          gy = live ? gy : 0.0f;
          db[k][e] += gy;
          dg[k][e] += gy * xh[k][e];
          gh[k][e] = gy * g[k][e];
          s1 += gh[k][e];
          s2 += gh[k][e] * xh[k][e];
        }
      }
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      const float m1 = s1 * inv_c, m2 = s2 * inv_c;
      if (row_live) {
#pragma unroll
        for (int k = 0; k < PLV; ++k) {
          const int ch4 = tl + k * TEAM;
          if (ch4 < c4)
            reinterpret_cast<float4*>(gx + (row0 + u) * c)[ch4] =
                make_float4(rstd * (gh[k][0] - m1 - xh[k][0] * m2), rstd * (gh[k][1] - m1 - xh[k][1] * m2),
                            rstd * (gh[k][2] - m1 - xh[k][2] * m2), rstd * (gh[k][3] - m1 - xh[k][3] * m2));
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PLV; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[team][0][4 * (tl + k * TEAM) + e] = dg[k][e];
      red[team][1][4 * (tl + k * TEAM) + e] = db[k][e];
    }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int t = 0; t < TEAMS; ++t) {
      a0 += red[t][0][ch];
      a1 += red[t][1][ch];
    }
    part[((int64_t)blockIdx.x * 2 + 0) * c + ch] = a0;
    part[((int64_t)blockIdx.x * 2 + 1) * c + ch] = a1;
  }
}

// grad_gamma / grad_beta = column sums of the per-workgroup partials, in a fixed order: 16 channels x 16 slices per
// workgroup, slice s adds partials s, s + 16, ... (four independent chains in flight), then the 16 slice sums are added
// in slice order.  (One thread per channel walking all 1024 partials serially was a 230 us latency chain.)
__global__ void __launch_bounds__(256) norm_act_bwd_fold_kernel(const float* __restrict__ part, int blocks, int c,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][16][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    int b = sl;
    for (; b + 48 < blocks; b += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0[u] += part[((int64_t)(b + 16 * u) * 2 + 0) * c + ch];
        a1[u] += part[((int64_t)(b + 16 * u) * 2 + 1) * c + ch];
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (b + 16 * u < blocks) {
        a0[u] += part[((int64_t)(b + 16 * u) * 2 + 0) * c + ch];
        a1[u] += part[((int64_t)(b + 16 * u) * 2 + 1) * c + ch];
      }
    }
  }
  red[0][sl][cl] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
  red[1][sl][cl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4;
    const int oc = blockIdx.x * 16 + cl;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[which][i][cl];
    float* dst = which ? dbeta : dgamma;
    if (oc < c && dst) dst[oc] = s;
  }
}

constexpr int NA_BWD_BLOCKS = 1024;

template <int TEAM>
static int launch_norm_act_bwd(const float* x, const float* gout, int64_t n, int c, const float* gamma, const float* beta,
                               float eps, int act, float* gx, float* dgamma, float* dbeta, float* part, hipStream_t stream) {
  const int teams_per_block = 256 / TEAM;
  int64_t g = (n + teams_per_block - 1) / teams_per_block;
  if (g > NA_BWD_BLOCKS) g = NA_BWD_BLOCKS;
  if (g < 1) g = 1;
#define FSF_NAB(A_) \
  hipLaunchKernelGGL((norm_act_bwd_kernel<TEAM, A_>), dim3((unsigned)g), dim3(256), 0, stream, x, gout, n, c, gamma, beta, eps, gx, part)
  if (act == ACT_GELU) FSF_NAB(ACT_GELU);
  else if (act == ACT_RELU) FSF_NAB(ACT_RELU);
  else FSF_NAB(ACT_NONE);
#undef FSF_NAB
  hipLaunchKernelGGL(norm_act_bwd_fold_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, stream, part, (int)g, c, dgamma, dbeta);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

template <int TEAM, int PLV, int RU>
static int launch_norm_act_bwd_v4(const float* x, const float* gout, int64_t n, int c, const float* gamma, const float* beta,
                                  float eps, int act, float* gx, float* dgamma, float* dbeta, float* part, hipStream_t stream) {
  const int rows_per_block = 256 / TEAM * RU;
  int64_t g = (n + rows_per_block - 1) / rows_per_block;
  if (g > NA_BWD_BLOCKS) g = NA_BWD_BLOCKS;
  if (g < 1) g = 1;
#define FSF_NAB4(A_) \
  hipLaunchKernelGGL((norm_act_bwd_v4_kernel<TEAM, A_, PLV, RU>), dim3((unsigned)g), dim3(256), 0, stream, x, gout, n, c, gamma, beta, eps, gx, part)
  if (act == ACT_GELU) FSF_NAB4(ACT_GELU);
  else if (act == ACT_RELU) FSF_NAB4(ACT_RELU);
  else FSF_NAB4(ACT_NONE);
#undef FSF_NAB4
  hipLaunchKernelGGL(norm_act_bwd_fold_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, stream, part, (int)g, c, dgamma, dbeta);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

static bool na_v4_enabled() {
  return true;
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_norm_act(const float* x, int64_t n, int32_t c, const float* gamma, const float* beta, float eps,
                            int32_t norm, int32_t act, float* out, int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || norm < 0 || norm > 1 || act < 0 || act > 2 || (n > 0 && (!x || !out)) || ((gamma == nullptr) != (beta == nullptr)) ||
      (norm == NORM_AFFINE && !gamma))
    return FSF_ERR_INVALID_ARG;
  if (c > 64 * 16) return FSF_ERR_UNSUPPORTED;
  if (out_stride == 0) out_stride = c;
  if (out_stride < c) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (na_v4_enabled() && (c % 4) == 0 && (out_stride % 4) == 0 && (((uintptr_t)x | (uintptr_t)out) % 16) == 0 &&
      (!gamma || (((uintptr_t)gamma | (uintptr_t)beta) % 16) == 0)) {
    const int c4 = c / 4;
    if (c4 <= 16) return launch_norm_act_v4<16, 1, 2>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
    if (c4 <= 32) return launch_norm_act_v4<32, 1, 2>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
    if (c4 <= 64) return launch_norm_act_v4<64, 1, 2>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
    if (c4 <= 128) return launch_norm_act_v4<64, 2, 1>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
    return launch_norm_act_v4<64, 4, 1>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  }
  if (c > 64 * NA_MAX_PER_LANE) return launch_norm_act<64, 16>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  if (c <= 16 * NA_MAX_PER_LANE / 2) return launch_norm_act<16>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  if (c <= 32 * NA_MAX_PER_LANE / 2) return launch_norm_act<32>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  return launch_norm_act<64>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
}

extern "C" int64_t fsf_norm_act_backward_workspace_bytes(int32_t c) {
  return (int64_t)NA_BWD_BLOCKS * 2 * (c > 0 ? c : 1) * 4 + 256;
}

extern "C" int fsf_norm_act_backward(const float* x, const float* grad_out, int64_t n, int32_t c, const float* gamma,
                                     const float* beta, float eps, int32_t act, float* grad_x, float* grad_gamma,
                                     float* grad_beta, void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || act < 0 || act > 2 || (n > 0 && (!x || !grad_out || !grad_x)) || ((gamma == nullptr) != (beta == nullptr)))
    return FSF_ERR_INVALID_ARG;
  const bool v4_ok = na_v4_enabled() && (c % 4) == 0 && (((uintptr_t)x | (uintptr_t)grad_out | (uintptr_t)grad_x) % 16) == 0 &&
                     (!gamma || (((uintptr_t)gamma | (uintptr_t)beta) % 16) == 0);
  if (c > 64 * 16 || (c > 64 * NA_MAX_PER_LANE && !v4_ok)) return FSF_ERR_UNSUPPORTED;  // (513 .. 1024 channels: the float4 form only)
  if (workspace_bytes < fsf_norm_act_backward_workspace_bytes(c) || !workspace) return FSF_ERR_WORKSPACE;
  float* part = (float*)workspace;
  if (n == 0) {
    if (grad_gamma) FSF_HIP_TRY(hipMemsetAsync(grad_gamma, 0, sizeof(float) * c, stream));
    if (grad_beta) FSF_HIP_TRY(hipMemsetAsync(grad_beta, 0, sizeof(float) * c, stream));
    return FSF_OK;
  }
  if (v4_ok) {
    const int c4 = c / 4;
    if (c4 <= 16) return launch_norm_act_bwd_v4<16, 1, 2>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
    if (c4 <= 32) return launch_norm_act_bwd_v4<32, 1, 2>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
    if (c4 <= 64) return launch_norm_act_bwd_v4<64, 1, 1>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
    if (c4 <= 128) return launch_norm_act_bwd_v4<64, 2, 1>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
    // the 1024-wide LayerNorm + GELU of the query / refine heads (round 6: torch's layer_norm backward + a GELU backward pass before)
    return launch_norm_act_bwd_v4<64, 4, 1>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
  }
  if (c <= 16 * NA_MAX_PER_LANE / 2)
    return launch_norm_act_bwd<16>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
  if (c <= 32 * NA_MAX_PER_LANE / 2)
    return launch_norm_act_bwd<32>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
  return launch_norm_act_bwd<64>(x, grad_out, n, c, gamma, beta, eps, act, grad_x, grad_gamma, grad_beta, part, stream);
}
