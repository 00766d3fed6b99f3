// K12: fused row normalisation + activation epilogue for the per-point MLPs of the VFE / SIR blocks.
// Replaces: the `norm -> act` tail of every `Linear -> norm -> act` block built by build_mlp
//   (projects/mmdet3d_plugin/ops/sst_ops.py:808-833) and DynamicVFELayer [UNVENDORED]: upstream runs
//   LayerNorm (or eval BatchNorm) and GELU/ReLU as separate ATen kernels, i.e. the [n, C] activations make
//   3 round trips through HBM after the GEMM; here it is one read + one write.
// HBM-bound: 4C B/row read + 4C B/row written.  One lane team per row, the row lives in registers, mean and
// variance are two-pass over the registers (no E[x^2]-E[x]^2 cancellation), var = biased (LayerNorm).
#include "common.h"

namespace fsf {

enum { NORM_LN = 0, NORM_AFFINE = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

constexpr int NA_MAX_PER_LANE = 8;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int TEAM, int ACT, int NORM>
__global__ void __launch_bounds__(256)
    norm_act_kernel(const float* x, int64_t n, int c, const float* __restrict__ gamma,
                    const float* __restrict__ beta, float eps, float* out, int64_t out_stride) {
  const int tl = threadIdx.x % TEAM;
  const int teams_per_block = 256 / TEAM;
  const float inv_c = 1.0f / (float)c;
  for (int64_t row = (int64_t)blockIdx.x * teams_per_block + threadIdx.x / TEAM; row < n;
       row += (int64_t)gridDim.x * teams_per_block) {
    const float* xr = x + row * c;
    float v[NA_MAX_PER_LANE];
#pragma unroll
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
      const int ch = tl + k * TEAM;
      v[k] = (ch < c) ? xr[ch] : 0.0f;
    }
    float mean = 0.0f, rstd = 1.0f;
    if (NORM == NORM_LN) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < NA_MAX_PER_LANE; ++k) s += v[k];
#pragma unroll
      for (int o = TEAM >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      mean = s * inv_c;
      float q = 0.0f;
#pragma unroll
      for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
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
    for (int k = 0; k < NA_MAX_PER_LANE; ++k) {
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

template <int TEAM>
static int launch_norm_act(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps, int norm,
                           int act, float* out, int64_t out_stride, hipStream_t stream) {
  const int teams_per_block = 256 / TEAM;
  int64_t g = (n + teams_per_block - 1) / teams_per_block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
#define FSF_NA(N_, A_) \
  hipLaunchKernelGGL((norm_act_kernel<TEAM, A_, N_>), dim3((unsigned)g), dim3(256), 0, stream, x, n, c, gamma, beta, eps, out, out_stride)
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

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_norm_act(const float* x, int64_t n, int32_t c, const float* gamma, const float* beta, float eps,
                            int32_t norm, int32_t act, float* out, int64_t out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || c < 1 || norm < 0 || norm > 1 || act < 0 || act > 2 || (n > 0 && (!x || !out)) || ((gamma == nullptr) != (beta == nullptr)) ||
      (norm == NORM_AFFINE && !gamma))
    return FSF_ERR_INVALID_ARG;
  if (c > 64 * NA_MAX_PER_LANE) return FSF_ERR_UNSUPPORTED;
  if (out_stride == 0) out_stride = c;
  if (out_stride < c) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (c <= 16 * NA_MAX_PER_LANE / 2) return launch_norm_act<16>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  if (c <= 32 * NA_MAX_PER_LANE / 2) return launch_norm_act<32>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
  return launch_norm_act<64>(x, n, c, gamma, beta, eps, norm, act, out, out_stride, stream);
}
