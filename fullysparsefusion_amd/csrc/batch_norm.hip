// K23: column statistics of a [n, c] row-major matrix and training-mode BatchNorm1d (+ ReLU) forward / backward.
// Replaces (training only): ATen's batch_norm_collect_statistics / batch_norm_backward_reduce / batch_norm_backward_elemt
//   channels-last kernels behind every `conv -> BN -> ReLU` of the sparse U-Net (mmdet3d.ops.make_sparse_convmodule /
//   SparseBasicBlock [UNVENDORED]; norm_cfg naiveSyncBN1d, projects/configs/nuScenes/FSF_nuScenes_config.py:50,63,85)
//   and the bias gradient `grad.sum(0)` of the per-point Linear layers (build_mlp, ops/sst_ops.py:808-833), which
//   run at 2-5 % of the HBM rate on [1e5..5e5, 64..131] inputs.
// HBM-bound: statistics read x twice (mean, then squared deviations about it: no E[x^2] - E[x]^2 cancellation), the
// backward reads (x, grad) twice and writes grad_x once.  Every sum is a fixed-order two-stage reduction (rows ->
// workgroup partial -> fold), so results are bit-reproducible run to run.
#include "common.h"

namespace fsf {

constexpr int CS_BLOCKS = 1024;

enum { CS_SUM = 0, CS_SQDEV = 1, CS_BN_BWD = 2 };

struct CsArgs {
  const float* x;
  const float* g;
  int64_t n;
  int c;
  int cw_log2;             // columns walked per pass = 1 << cw_log2 (<= 256); rows side by side = 256 >> cw_log2
  int64_t rows_per_block;
  const float* mean;
  const float* invstd;
  const float* scale;      // gamma * invstd (or NULL = invstd) and beta - mean * scale (or NULL): y = fma(x, scale, shift)
  const float* shift;
  int relu;
  float* part;             // [blocks][2][c]
};

template <int MODE>
__device__ __forceinline__ void cs_accumulate(const CsArgs& a, int64_t row, int col, float mu, float is, float sc, float sh,
                                              float& s0, float& s1) {
  const float v = a.x[row * a.c + col];
  if (MODE == CS_SUM) {
    s0 += v;
  } else if (MODE == CS_SQDEV) {
    const float d = v - mu;
    s0 += d * d;
  } else {
    float g = a.g[row * a.c + col];
    if (a.relu && !(__fmaf_rn(v, sc, sh) > 0.0f)) g = 0.0f;
    s0 += g;
    s1 += g * ((v - mu) * is);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) column_stats_kernel(CsArgs a) {
  __shared__ float red[2][256];
  const int cw = 1 << a.cw_log2;
  const int tx = threadIdx.x & (cw - 1), ty = threadIdx.x >> a.cw_log2, ry = 256 >> a.cw_log2;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  const int64_t r1 = r0 + a.rows_per_block < a.n ? r0 + a.rows_per_block : a.n;
  for (int col0 = 0; col0 < a.c; col0 += cw) {
    const int col = col0 + tx;
    const bool live = col < a.c;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      float mu = 0.0f, is = 1.0f, sc = 1.0f, sh = 0.0f;
      if (MODE != CS_SUM) mu = a.mean[col];
      if (MODE == CS_BN_BWD) {
        is = a.invstd[col];
        sc = a.scale ? a.scale[col] : is;
        sh = a.shift ? a.shift[col] : -mu * sc;
      }
      int64_t r = r0 + ty;
      for (; r + 3 * ry < r1; r += 4 * ry) {
#pragma unroll
        for (int u = 0; u < 4; ++u) cs_accumulate<MODE>(a, r + (int64_t)u * ry, col, mu, is, sc, sh, s0[u], s1[u]);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (r + (int64_t)u * ry < r1) cs_accumulate<MODE>(a, r + (int64_t)u * ry, col, mu, is, sc, sh, s0[u], s1[u]);
    }
    red[0][threadIdx.x] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    red[1][threadIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    __syncthreads();
    if (ty == 0 && live) {
      float t0 = 0.0f, t1 = 0.0f;
      for (int i = 0; i < ry; ++i) {
        t0 += red[0][i * cw + tx];
        t1 += red[1][i * cw + tx];
      }
      a.part[((int64_t)blockIdx.x * 2 + 0) * a.c + col] = t0;
      a.part[((int64_t)blockIdx.x * 2 + 1) * a.c + col] = t1;
    }
    __syncthreads();
  }
}

// out0 / out1 [c] = (column sums of the workgroup partials) * mul, in a fixed order: 16 channels x 16 slices per workgroup,
// slice s adds partials s, s + 16, ... on four independent chains, then the 16 slice sums are added in slice order.
__global__ void __launch_bounds__(256) column_fold_kernel(const float* __restrict__ part, int blocks, int c, float mul,
                                                          float* __restrict__ out0, float* __restrict__ out1) {
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
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[which][i][cl];
    float* dst = which ? out1 : out0;
    if (ch < c && dst) dst[ch] = s * mul;
  }
}

// The second fold of a training-mode BatchNorm's statistics with everything that hangs on it: biased variance -> invstd,
// scale = weight * invstd, shift = bias - mean * scale, and the running-statistics update (momentum form) — one launch instead of
// the fold + ~10 tiny ATen kernels per layer (38 layers per training step).  Same fold order as column_fold_kernel.
struct BnFinalArgs {
  const float *mean, *weight, *bias;
  float eps, keep, momentum, var_alpha;  // running = running * keep + batch * momentum (variance: * var_alpha = momentum * n / (n - 1))
  float *running_mean, *running_var;     // nullable
  float *var, *invstd, *scale, *shift;   // var nullable
};

__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ part, int blocks, int c, float mul, BnFinalArgs f) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  float a0[4] = {0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    int b = sl;
    for (; b + 48 < blocks; b += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a0[u] += part[((int64_t)(b + 16 * u) * 2 + 0) * c + ch];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (b + 16 * u < blocks) a0[u] += part[((int64_t)(b + 16 * u) * 2 + 0) * c + ch];
  }
  red[sl][cl] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
  __syncthreads();
  if (threadIdx.x < 16 && ch < c) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][cl];
    const float var = s * mul, mean = f.mean[ch];
    const float invstd = (float)(1.0 / sqrt((double)var + (double)f.eps));  // (correctly rounded; ATen's rsqrt is within 1 ulp of it)
    const float scale = f.weight ? f.weight[ch] * invstd : invstd;
    const float shift = (f.bias ? f.bias[ch] : 0.0f) - mean * scale;
    if (f.var) f.var[ch] = var;
    f.invstd[ch] = invstd;
    f.scale[ch] = scale;
    f.shift[ch] = shift;
    if (f.running_mean) f.running_mean[ch] = __fmaf_rn(f.momentum, mean, f.running_mean[ch] * f.keep);
    if (f.running_var) f.running_var[ch] = __fmaf_rn(f.var_alpha, var, f.running_var[ch] * f.keep);
  }
}

struct BnArgs {
  const float* x;
  const float* g;
  int64_t n;
  int c;
  int cw_log2;
  int64_t rows_per_block;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  const float* sum_g;   // grad_beta
  const float* sum_gx;  // grad_gamma
  float inv_n;
  int relu;
  float* out;
};

// FWD: out = [relu] fma(x, scale, shift).  !FWD: grad_x = scale * (g' - (sum_g + xhat * sum_gx) / n), g' = grad masked by
// the sign of the SAME fma as the forward (so the mask never disagrees with the forward's ReLU).
template <bool FWD>
__global__ void __launch_bounds__(256) bn_rows_kernel(BnArgs a) {
  const int cw = 1 << a.cw_log2;
  const int tx = threadIdx.x & (cw - 1), ty = threadIdx.x >> a.cw_log2, ry = 256 >> a.cw_log2;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  const int64_t r1 = r0 + a.rows_per_block < a.n ? r0 + a.rows_per_block : a.n;
  for (int col = tx; col < a.c; col += cw) {
    float mu = 0.0f, is = 1.0f, sg = 0.0f, sgx = 0.0f;
    if (!FWD) {
      mu = a.mean[col];
      is = a.invstd[col];
      sg = a.sum_g[col] * a.inv_n;
      sgx = a.sum_gx[col] * a.inv_n;
    }
    const float sc = a.scale ? a.scale[col] : is;
    const float sh = a.shift ? a.shift[col] : -mu * sc;
#pragma unroll 4
    for (int64_t r = r0 + ty; r < r1; r += ry) {
      const float v = a.x[r * a.c + col];
      const float y = __fmaf_rn(v, sc, sh);
      if (FWD) {
        a.out[r * a.c + col] = (a.relu && !(y > 0.0f)) ? 0.0f : y;
      } else {
        float g = a.g[r * a.c + col];
        if (a.relu && !(y > 0.0f)) g = 0.0f;
        a.out[r * a.c + col] = sc * (g - (sg + (v - mu) * is * sgx));
      }
    }
  }
}

static int cs_cw_log2(int c) {
  int l = 0;
  while ((1 << l) < c && l < 8) ++l;
  return l;
}

static int cs_blocks(int64_t n, int cw_log2, int64_t* rows_per_block) {
  const int ry = 256 >> cw_log2;
  int64_t rpb = fsf_cdiv(n, (int64_t)CS_BLOCKS);
  const int64_t min_rows = (int64_t)ry * 8;  // at least two unrolled trips per workgroup
  if (rpb < min_rows) rpb = min_rows;
  *rows_per_block = rpb;
  return (int)fsf_cdiv(n, rpb);
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_column_stats_workspace_bytes(int32_t c) { return (int64_t)CS_BLOCKS * 2 * (c > 0 ? c : 1) * 4 + 256; }

extern "C" int fsf_column_stats(const float* x, int64_t n, int32_t c, float* mean, float* var, void* workspace,
                                int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || !mean || (n > 0 && !x)) return FSF_ERR_INVALID_ARG;
  if (workspace_bytes < fsf_column_stats_workspace_bytes(c) || !workspace) return FSF_ERR_WORKSPACE;
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(mean, 0, sizeof(float) * c, stream));
    if (var) FSF_HIP_TRY(hipMemsetAsync(var, 0, sizeof(float) * c, stream));
    return FSF_OK;
  }
  CsArgs a{};
  a.x = x; a.n = n; a.c = c; a.cw_log2 = cs_cw_log2(c); a.part = (float*)workspace;
  const int blocks = cs_blocks(n, a.cw_log2, &a.rows_per_block);
  const unsigned fold_grid = (unsigned)((c + 15) / 16);
  hipLaunchKernelGGL((column_stats_kernel<CS_SUM>), dim3(blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(column_fold_kernel, dim3(fold_grid), dim3(256), 0, stream, a.part, blocks, c, var ? 1.0f / (float)n : 1.0f,
                     mean, (float*)nullptr);
  if (var) {
    a.mean = mean;
    hipLaunchKernelGGL((column_stats_kernel<CS_SQDEV>), dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(column_fold_kernel, dim3(fold_grid), dim3(256), 0, stream, a.part, blocks, c, 1.0f / (float)n, var,
                       (float*)nullptr);
  }
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_batch_norm_train_stats(const float* x, int64_t n, int32_t c, const float* weight, const float* bias, float eps,
                                          float keep, float momentum, float var_alpha, float* running_mean, float* running_var,
                                          float* mean, float* var, float* invstd, float* scale, float* shift, void* workspace,
                                          int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 1 || c < 1 || !x || !mean || !invstd || !scale || !shift || ((weight == nullptr) != (bias == nullptr)))
    return FSF_ERR_INVALID_ARG;
  if (workspace_bytes < fsf_column_stats_workspace_bytes(c) || !workspace) return FSF_ERR_WORKSPACE;
  CsArgs a{};
  a.x = x; a.n = n; a.c = c; a.cw_log2 = cs_cw_log2(c); a.part = (float*)workspace;
  const int blocks = cs_blocks(n, a.cw_log2, &a.rows_per_block);
  const unsigned fold_grid = (unsigned)((c + 15) / 16);
  hipLaunchKernelGGL((column_stats_kernel<CS_SUM>), dim3(blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(column_fold_kernel, dim3(fold_grid), dim3(256), 0, stream, a.part, blocks, c, 1.0f / (float)n, mean,
                     (float*)nullptr);
  a.mean = mean;
  hipLaunchKernelGGL((column_stats_kernel<CS_SQDEV>), dim3(blocks), dim3(256), 0, stream, a);
  BnFinalArgs f{mean, weight, bias, eps, keep, momentum, var_alpha, running_mean, running_var, var, invstd, scale, shift};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(fold_grid), dim3(256), 0, stream, a.part, blocks, c, 1.0f / (float)n, f);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_batch_norm_act_forward(const float* x, int64_t n, int32_t c, const float* scale, const float* shift,
                                          int32_t relu, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || !scale || !shift || (n > 0 && (!x || !out))) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  BnArgs a{};
  a.x = x; a.n = n; a.c = c; a.cw_log2 = cs_cw_log2(c); a.scale = scale; a.shift = shift; a.relu = relu; a.out = out;
  const int ry = 256 >> a.cw_log2;
  a.rows_per_block = (int64_t)ry * 4;
  const int64_t blocks = fsf_cdiv(n, a.rows_per_block);
  hipLaunchKernelGGL((bn_rows_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_batch_norm_act_backward(const float* x, const float* grad_out, int64_t n, int32_t c, const float* mean,
                                           const float* invstd, const float* scale, const float* shift, int32_t relu,
                                           float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                                           int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || !mean || !invstd || !grad_gamma || !grad_beta || ((scale == nullptr) != (shift == nullptr)) ||
      (n > 0 && (!x || !grad_out || !grad_x)))
    return FSF_ERR_INVALID_ARG;
  if (workspace_bytes < fsf_column_stats_workspace_bytes(c) || !workspace) return FSF_ERR_WORKSPACE;
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(grad_gamma, 0, sizeof(float) * c, stream));
    FSF_HIP_TRY(hipMemsetAsync(grad_beta, 0, sizeof(float) * c, stream));
    return FSF_OK;
  }
  CsArgs a{};
  a.x = x; a.g = grad_out; a.n = n; a.c = c; a.cw_log2 = cs_cw_log2(c); a.part = (float*)workspace;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.relu = relu;
  const int blocks = cs_blocks(n, a.cw_log2, &a.rows_per_block);
  hipLaunchKernelGGL((column_stats_kernel<CS_BN_BWD>), dim3(blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(column_fold_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, stream, a.part, blocks, c, 1.0f, grad_beta,
                     grad_gamma);
  BnArgs b{};
  b.x = x; b.g = grad_out; b.n = n; b.c = c; b.cw_log2 = a.cw_log2; b.mean = mean; b.invstd = invstd; b.scale = scale;
  b.shift = shift; b.sum_g = grad_beta; b.sum_gx = grad_gamma; b.inv_n = 1.0f / (float)n; b.relu = relu; b.out = grad_x;
  const int ry = 256 >> b.cw_log2;
  b.rows_per_block = (int64_t)ry * 4;
  hipLaunchKernelGGL((bn_rows_kernel<false>), dim3((unsigned)fsf_cdiv(n, b.rows_per_block)), dim3(256), 0, stream, b);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
