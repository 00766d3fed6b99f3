// K9/K11: sparse convolution forward as an output-stationary implicit GEMM on the fp32 matrix cores.
// See include/fsf_hip.h.
//
// One workgroup (4 waves) owns a tile of TM=64 output rows x TN<=128 output channels whose accumulator
// lives in LDS for the whole kernel-offset loop, so every output row is written exactly once (no
// scatter-add atomics, deterministic).  For each kernel offset k the rows of the tile that actually have
// a neighbour are COMPACTED (ballot prefix, precomputed per tile), their input rows are gathered with
// coalesced 16-byte loads into an LDS A tile, and v_mfma_f32_16x16x4_f32 runs over ceil(cnt/16) row blocks
// only — on LiDAR data ~22 % of (row, offset) pairs exist, so the dense 27-offset product would waste 4/5
// of the matrix-core time.  Each wave owns a 32-column slice and keeps its B (weight) fragments for the
// current offset in registers (weights pre-transposed to [k][cout][cin] so a lane's four consecutive
// K values arrive as one 16-byte load).  Eval-mode BN affine + residual + ReLU are the fused epilogue.
//
// Roofline: flops = 2*P*Cin*Cout on the fp32 MFMA (157.3 TF/s peak),
// bytes = P*(Cin+Cout)*4 + kvol*Cin*Cout*4 + 8P.
#include "common.h"

namespace fsf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SC_TM = 64;
constexpr int SC_KC = 128;
constexpr int SC_ASTRIDE = SC_KC + 4;
constexpr int SC_MAXK = 27;

struct SpconvArgs {
  const float* feat;
  const float* wt;  // [kvol][cout][cin]
  const int32_t* nbr;
  const float* scale;
  const float* shift;
  const float* residual;
  float* out;
  int64_t m_in, m_out;
  int cin, cout, kvol, relu;
};

template <int TN>
__global__ void __launch_bounds__(256, 2) spconv_fwd_kernel(SpconvArgs a) {
  constexpr int CS_STRIDE = TN + 4;
  constexpr int WCOLS = TN / 4;    // columns per wave
  constexpr int NCT = WCOLS / 16;  // 16-column MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Cs = reinterpret_cast<float*>(smem);                          // [SC_TM][CS_STRIDE]
  float* As = Cs + SC_TM * CS_STRIDE;                                  // [SC_TM][SC_ASTRIDE]
  int32_t* rl_in = reinterpret_cast<int32_t*>(As + SC_TM * SC_ASTRIDE);  // [SC_MAXK][SC_TM]
  uint8_t* rl_loc = reinterpret_cast<uint8_t*>(rl_in + SC_MAXK * SC_TM);  // [SC_MAXK][SC_TM]
  int32_t* rl_cnt = reinterpret_cast<int32_t*>(rl_loc + SC_MAXK * SC_TM);  // [SC_MAXK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t o0 = (int64_t)blockIdx.x * SC_TM;
  const int n0 = blockIdx.y * TN;

  // ---- per-tile compaction lists for every offset (wave w handles k = w, w+4, ...) ----
  for (int k = wave; k < a.kvol; k += 4) {
    const int64_t o = o0 + lane;
    const int32_t in = (o < a.m_out) ? a.nbr[o * a.kvol + k] : -1;
    const bool has = in >= 0;
    const uint64_t bal = __ballot(has);
    if (has) {
      const int pos = (int)__popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      rl_in[k * SC_TM + pos] = in;
      rl_loc[k * SC_TM + pos] = (uint8_t)lane;
    }
    if (lane == 0) rl_cnt[k] = (int)__popcll(bal);
  }
  for (int t = tid; t < SC_TM * CS_STRIDE; t += 256) Cs[t] = 0.0f;
  __syncthreads();

  const int lrow = lane & 15;   // A row / B column inside a 16x16 tile
  const int kgrp = lane >> 4;   // which 4-float K group this lane feeds
  const int wcol0 = wave * WCOLS;

  for (int k = 0; k < a.kvol; ++k) {
    const int cnt = rl_cnt[k];
    if (cnt == 0) continue;
    const int nrb = (cnt + 15) >> 4;
    for (int cin0 = 0; cin0 < a.cin; cin0 += SC_KC) {
      const int kc = (a.cin - cin0 < SC_KC) ? (a.cin - cin0) : SC_KC;  // multiple of 16
      // ---- gather the compacted input rows into As ----
      {
        const int f4_per_row = kc >> 2;
        const int rows_per_pass = 256 / f4_per_row;
        const int v = tid % f4_per_row;
        const int r0 = tid / f4_per_row;
        for (int j = r0; j < cnt; j += rows_per_pass) {
          const int64_t in = rl_in[k * SC_TM + j];
          const float4 val = *reinterpret_cast<const float4*>(a.feat + in * a.cin + cin0 + 4 * v);
          *reinterpret_cast<float4*>(As + j * SC_ASTRIDE + 4 * v) = val;
        }
      }
      __syncthreads();
      // ---- B fragments of this offset / cin chunk, kept in registers across row blocks ----
      f32x4 bfrag[NCT][SC_KC / 16];
      const int nsteps = kc >> 4;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = n0 + wcol0 + ct * 16 + lrow;
        const float* wp = a.wt + ((int64_t)k * a.cout + (col < a.cout ? col : 0)) * a.cin + cin0 + 4 * kgrp;
#pragma unroll
        for (int s = 0; s < SC_KC / 16; ++s) {
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (s < nsteps && col < a.cout) w = *reinterpret_cast<const f32x4*>(wp + 16 * s);
          bfrag[ct][s] = w;
        }
      }
      for (int rb = 0; rb < nrb; ++rb) {
        f32x4 acc[NCT];
        int loc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = rb * 16 + kgrp * 4 + r;
          loc[r] = (j < cnt) ? (int)rl_loc[k * SC_TM + j] : -1;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[ct][r] = (loc[r] >= 0) ? Cs[loc[r] * CS_STRIDE + wcol0 + ct * 16 + lrow] : 0.0f;
        }
        const float* arow = As + (rb * 16 + lrow) * SC_ASTRIDE + 4 * kgrp;
#pragma unroll
        for (int s = 0; s < SC_KC / 16; ++s) {
          if (s < nsteps) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * s);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bfrag[ct][s][t], acc[ct], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (loc[r] >= 0) Cs[loc[r] * CS_STRIDE + wcol0 + ct * 16 + lrow] = acc[ct][r];
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: BN affine (+ residual) (+ ReLU), coalesced float4 stores ----
  constexpr int F4_PER_ROW = TN / 4;
  for (int t = tid; t < SC_TM * F4_PER_ROW; t += 256) {
    const int r = t / F4_PER_ROW;
    const int c4 = (t % F4_PER_ROW) * 4;
    const int64_t o = o0 + r;
    const int col = n0 + c4;
    if (o >= a.m_out || col >= a.cout) continue;
    float4 v = *reinterpret_cast<const float4*>(Cs + r * CS_STRIDE + c4);
    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = vv[q];
      if (a.scale) x = __fmaf_rn(x, a.scale[col + q], a.shift[col + q]);
      else if (a.shift) x = __fadd_rn(x, a.shift[col + q]);
      if (a.residual) x = __fadd_rn(x, a.residual[o * a.cout + col + q]);
      if (a.relu) x = fmaxf(x, 0.0f);
      vv[q] = x;
    }
    *reinterpret_cast<float4*>(a.out + o * a.cout + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
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

template <int TN>
static size_t spconv_smem_bytes() {
  return (size_t)SC_TM * (TN + 4) * 4 + (size_t)SC_TM * SC_ASTRIDE * 4 + (size_t)SC_MAXK * SC_TM * 4 +
         (size_t)SC_MAXK * SC_TM + (size_t)SC_MAXK * 4 + 64;
}

}  // namespace fsf

using namespace fsf;

extern "C" int fsf_spconv_transpose_weight(const float* weight, int32_t kvol, int32_t cin, int32_t cout, float* weight_t,
                                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!weight || !weight_t || kvol < 1 || cin < 1 || cout < 1) return FSF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(transpose_weight_kernel, dim3(fsf_stream_grid((int64_t)kvol * cin * cout, 256)), dim3(256), 0, stream,
                     weight, (int)kvol, (int)cin, (int)cout, weight_t);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_spconv_forward(const float* feat, int64_t m_in, int32_t cin, const float* weight_t, int32_t kvol,
                                  int32_t cout, const int32_t* nbr, int64_t m_out, const float* scale, const float* shift,
                                  const float* residual, int32_t relu, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (m_in < 0 || m_out < 0 || cin < 1 || cout < 1 || kvol < 1 || !weight_t || (scale && !shift) ||
      (m_out > 0 && (!nbr || !out)) || (m_in > 0 && !feat))
    return FSF_ERR_INVALID_ARG;
  if (kvol > SC_MAXK || (cin % 16) != 0 || (cout % 4) != 0) return FSF_ERR_UNSUPPORTED;
  if (m_out == 0) return FSF_OK;
  SpconvArgs a{feat, weight_t, nbr, scale, shift, residual, out, m_in, m_out, (int)cin, (int)cout, (int)kvol, (int)relu};
  const unsigned tiles = (unsigned)((m_out + SC_TM - 1) / SC_TM);
  if (cout <= 64) {
    static bool attr_set64 = false;
    if (!attr_set64) {
      FSF_HIP_TRY(hipFuncSetAttribute((const void*)spconv_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)spconv_smem_bytes<64>()));
      attr_set64 = true;
    }
    hipLaunchKernelGGL((spconv_fwd_kernel<64>), dim3(tiles, (cout + 63) / 64), dim3(256), spconv_smem_bytes<64>(), stream, a);
  } else {
    static bool attr_set128 = false;
    if (!attr_set128) {
      FSF_HIP_TRY(hipFuncSetAttribute((const void*)spconv_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)spconv_smem_bytes<128>()));
      attr_set128 = true;
    }
    hipLaunchKernelGGL((spconv_fwd_kernel<128>), dim3(tiles, (cout + 127) / 128), dim3(256), spconv_smem_bytes<128>(), stream, a);
  }
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
