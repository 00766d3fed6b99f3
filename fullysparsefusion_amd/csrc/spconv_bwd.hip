// K10: sparse-convolution weight gradient.
//   grad_W[k][ci][co] = sum over pairs p of offset k:  feat[in_k[p]][ci] * grad_out[out_k[p]][co]
// Replaces: spconv v1 indice_conv_backward's per-offset (gather, gather, cuBLAS mm^T) triple [UNVENDORED
//   mmdet3d.ops.spconv, used under SimpleSparseUNet in training, projects/configs/nuScenes/FSF_nuScenes_config.py:58-70].
//   (The data gradient needs no kernel of its own: it is fsf_spconv_forward over the transposed neighbour table
//   with the un-transposed weight, see include/fsf_hip.h.)
//
// Shape of the problem: 27 skinny GEMMs whose reduction dimension is the pair list (up to m_out long) and whose
// output is tiny ([Cin, Cout]).  So the grid is (pair-range split, offset, channel tile) and every workgroup keeps its
// [<=128 x <=128] slice of grad_W[k] in MFMA accumulators for its whole pair range; partial slices go to the workspace
// and a second kernel folds them in split order (deterministic, no float atomics).
//
// Both operands are "K-major" in memory (a row of feat / grad_out is one K index).  v_mfma_f32_16x16x4_f32 takes
// A[m][k] from lane (m = lane%16, k = lane/16): lane reads ONE float4 of row 4*step + lane/16 at channel 4*(lane%16)
// and its 4 components feed 4 different MFMAs (MFMA q owns channels {4*m + q}); the same for B.  One 16-byte LDS read
// of each operand feeds 16 MFMAs, and 16 consecutive lanes read 256 contiguous bytes (conflict-free, no swizzle).
//
// Stages of 32 pairs are gathered by LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per stage; the
// pair indices of stage s+2 are prefetched into registers while stage s computes.
#include <stdlib.h>

#include "common.h"

namespace fsf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BW_RT = 32;  // pairs per stage

struct SpconvBwdArgs {
  const float* feat;      // [m_in, cin]
  const float* gout;      // [m_out, cout]
  const int32_t* pairs;   // [kvol, 2, cap]
  const int32_t* num;     // [kvol]
  float* gw;              // [kvol, cin, cout]
  float* part;            // [kvol, nsplit, cin, cout] (nsplit > 1)
  int64_t cap;
  int cin, cout, kvol;
  int nsplit;             // pair-range splits per offset
  int range;              // pairs per split (multiple of BW_RT)
  int tiles_b;            // cout tiles
};

template <int TA, int TB>
struct BwdSmem {
  static constexpr int STAGE_FLOATS = BW_RT * (TA + TB);
  static constexpr int KS = 4 / ((TA / 64) * (TB / 64));  // in-block split of the k-steps
  static constexpr int RED_FLOATS = (KS - 1) * (TA / 64) * (TB / 64) * 4096;
  static constexpr int FLOATS = 2 * STAGE_FLOATS > RED_FLOATS ? 2 * STAGE_FLOATS : RED_FLOATS;
  static constexpr size_t bytes() { return sizeof(float) * FLOATS; }
};

// One operand tile [BW_RT rows][T floats] by LDS-DMA.  A wave instruction moves 1024 contiguous LDS bytes = RPI rows;
// wave w issues row groups w, w+4, ...
template <int T>
struct TileLoad {
  static constexpr int LPR = T / 4;          // lanes per row
  static constexpr int RPI = 64 / LPR;       // rows per DMA instruction
  static constexpr int NG = BW_RT / RPI;     // instructions per tile
  static constexpr int PER_WAVE = NG / 4;
};

template <int TA, int TB>
__global__ void __launch_bounds__(256, 2) spconv_bwd_weight_kernel(SpconvBwdArgs a) {
  using SM = BwdSmem<TA, TB>;
  using LA = TileLoad<TA>;
  using LB = TileLoad<TB>;
  constexpr int KS = SM::KS;
  constexpr int NWB = TB / 64;
  constexpr int NSTEP = BW_RT / 4 / KS;  // k-steps of a stage owned by one wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* S = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y;
  const int split = blockIdx.x;
  const int a0 = (blockIdx.z / a.tiles_b) * TA;
  const int b0 = (blockIdx.z % a.tiles_b) * TB;
  const int n = a.num ? a.num[k] : (int)a.cap;  // no pair lists: the identity pairing of a dense layer (kvol = 1)
  const int p_begin = split * a.range;
  if (p_begin >= n) return;  // uniform; the fold kernel only reads live splits
  const int p_end = min(n, p_begin + a.range);
  const int nstages = (p_end - p_begin + BW_RT - 1) / BW_RT;

  const bool ident = a.pairs == nullptr;
  const int32_t* pin = ident ? nullptr : a.pairs + (int64_t)k * 2 * a.cap;
  const int32_t* pout = ident ? nullptr : pin + a.cap;

  // wave tile: 64 x 64 channels; waves that share (wa, wb) split the k-steps of each stage
  const int wt = wave / KS, ks = wave % KS;
  const int wa = wt / NWB, wb = wt % NWB;
  const int l16 = lane & 15, kk = lane >> 4;

  // this lane's channel chunk inside the DMA rows (clamped: channels past cin/cout only feed accumulators that are
  // never written out, they just have to be readable)
  int ca = a0 + 4 * (lane % LA::LPR);
  ca = ca + 4 <= a.cin ? ca : a.cin - 4;
  int cb = b0 + 4 * (lane % LB::LPR);
  cb = cb + 4 <= a.cout ? cb : a.cout - 4;
  const int ra = lane / LA::LPR, rb = lane / LB::LPR;  // row inside the instruction's row group

  // Pair indices of the stage to be issued next.  The load address is clamped instead of the value being selected
  // after the load: a select on the loaded value would make the compiler wait for it (and for the DMA queued before
  // it) right at the load, ahead of the MFMA work it is meant to hide under.
  int32_t ia[LA::PER_WAVE], ib[LB::PER_WAVE];
  auto load_indices = [&](int stage) {
    const int p0 = p_begin + stage * BW_RT;
#pragma unroll
    for (int i = 0; i < LA::PER_WAVE; ++i) {
      const int p = min(p0 + (wave + 4 * i) * LA::RPI + ra, p_end - 1);
      ia[i] = ident ? p : pin[p];
    }
#pragma unroll
    for (int i = 0; i < LB::PER_WAVE; ++i) {
      const int p = min(p0 + (wave + 4 * i) * LB::RPI + rb, p_end - 1);
      ib[i] = ident ? p : pout[p];
    }
  };
  auto issue_stage = [&](int stage) {
    float* As = S + (stage & 1) * SM::STAGE_FLOATS;
    float* Bs = As + BW_RT * TA;
    const int p0 = p_begin + stage * BW_RT;
#pragma unroll
    for (int i = 0; i < LA::PER_WAVE; ++i) {
      float* dst = As + (wave + 4 * i) * (LA::RPI * TA);  // wave-uniform; the DMA adds lane * 16 B
      if (p0 + (wave + 4 * i) * LA::RPI + ra < p_end)
        __builtin_amdgcn_global_load_lds(a.feat + (int64_t)ia[i] * a.cin + ca, dst, 16, 0, 0);
      else  // past the end of the pair list: the lane zero-fills its 16 bytes
        reinterpret_cast<f32x4*>(dst)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < LB::PER_WAVE; ++i) {
      float* dst = Bs + (wave + 4 * i) * (LB::RPI * TB);
      if (p0 + (wave + 4 * i) * LB::RPI + rb < p_end)
        __builtin_amdgcn_global_load_lds(a.gout + (int64_t)ib[i] * a.cout + cb, dst, 16, 0, 0);
      else
        reinterpret_cast<f32x4*>(dst)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int qa = 0; qa < 4; ++qa)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) acc[qa][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_indices(0);
  issue_stage(0);
  if (nstages > 1) load_indices(1);

  const int aoff = wa * 64 + 4 * l16, boff = BW_RT * TA + wb * 64 + 4 * l16;
  for (int s = 0; s < nstages; ++s) {
    __syncthreads();  // stage s landed (each wave drains its own DMA before the barrier); buffer (s+1)&1 is free
#ifndef BW_ABL_NO_DMA  // (ablation builds: tools/profiling/bwd_weight_layers.py with FSF_EXTRA_HIPCC_FLAGS)
    if (s + 1 < nstages) {
      issue_stage(s + 1);
      if (s + 2 < nstages) load_indices(s + 2);
    }
#endif
    const float* As = S + (s & 1) * SM::STAGE_FLOATS;
    f32x4 av[NSTEP], bv[NSTEP];
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      const int row = 4 * (ks + KS * i) + kk;
      av[i] = *reinterpret_cast<const f32x4*>(As + row * TA + aoff);
      bv[i] = *reinterpret_cast<const f32x4*>(As + row * TB + boff);
    }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i)
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#ifdef BW_ABL_NO_MFMA
          acc[qa][qb][0] += av[i][qa] * bv[i][qb];
#else
          acc[qa][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][qa], bv[i][qb], acc[qa][qb], 0, 0, 0);
#endif
  }

  // fold the in-block k-step split through LDS in fixed order (ks = 1, 2, 3 onto ks = 0)
  if (KS > 1) {
    __syncthreads();
    if (ks > 0) {
      f32x4* red = reinterpret_cast<f32x4*>(S + ((ks - 1) * (4 / KS) + wt) * 4096);
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) red[(qa * 4 + qb) * 64 + lane] = acc[qa][qb];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll 1
    for (int r = 1; r < KS; ++r) {
      const f32x4* red = reinterpret_cast<const f32x4*>(S + ((r - 1) * (4 / KS) + wt) * 4096);
#pragma unroll
      for (int qa = 0; qa < 4; ++qa)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) acc[qa][qb] += red[(qa * 4 + qb) * 64 + lane];
    }
  }

  // D[m][n] of MFMA (qa, qb): m = 4 * (lane / 16) + j -> channel ci = 4 m + qa;  n = lane % 16 -> co = 4 n + qb
  float* dst = a.nsplit > 1 ? a.part + ((int64_t)k * a.nsplit + split) * a.cin * a.cout : a.gw + (int64_t)k * a.cin * a.cout;
  const int co = b0 + wb * 64 + 4 * l16;
  if (co < a.cout) {
#pragma unroll
    for (int qa = 0; qa < 4; ++qa)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = a0 + wa * 64 + 4 * (4 * kk + j) + qa;
        if (ci < a.cin)
          *reinterpret_cast<f32x4*>(dst + (int64_t)ci * a.cout + co) =
              f32x4{acc[qa][0][j], acc[qa][1][j], acc[qa][2][j], acc[qa][3][j]};
      }
  }
}

// grad_W[k] = sum of the live partial slices (or zero when the offset has no pairs), in a fixed order: a workgroup owns 16
// float4 elements x 16 slices; slice s adds partials s, s + 16, ... on four independent chains, then the 16 slice sums are
// added in slice order.  (One thread per element walking up to ~1000 partials serially was a 100-250 us latency chain.)
__global__ void __launch_bounds__(256) spconv_bwd_fold_kernel(SpconvBwdArgs a) {
  __shared__ f32x4 red[16][17];
  const int64_t per_k = (int64_t)a.cin * a.cout / 4;
  const int64_t total = per_k * a.kvol;
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < total; t0 += (int64_t)gridDim.x * 16) {
    const int64_t t = t0 + el;
    const bool ok = t < total;
    const int k = ok ? (int)(t / per_k) : 0;
    const int64_t e = t - (int64_t)k * per_k;
    const int n = a.num ? a.num[k] : (int)a.cap;
    const int live = ok ? (n + a.range - 1) / a.range : 0;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.nsplit > 1) {
      f32x4 acc[4] = {zero, zero, zero, zero};
      const f32x4* p = reinterpret_cast<const f32x4*>(a.part) + (int64_t)k * a.nsplit * per_k + e;
      int i = sl;
      for (; i + 48 < live; i += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += p[(int64_t)(i + 16 * u) * per_k];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (i + 16 * u < live) acc[u] += p[(int64_t)(i + 16 * u) * per_k];
      red[sl][el] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      __syncthreads();
      if (sl == 0 && ok) {
        f32x4 s = zero;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][el];
        reinterpret_cast<f32x4*>(a.gw)[t] = s;
      }
      __syncthreads();
    } else if (ok && live == 0 && sl == 0) {
      reinterpret_cast<f32x4*>(a.gw)[t] = zero;
    }
  }
}

static void bwd_plan(int64_t cap, int cin, int cout, int kvol, int* ta, int* tb, int* nsplit, int* range) {
  *ta = cin <= 64 ? 64 : 128;
  *tb = cout <= 64 ? 64 : 128;
  const int64_t tiles = (int64_t)fsf_cdiv(cin, *ta) * fsf_cdiv(cout, *tb);
  // enough workgroups to keep 512 resident slots busy despite the uneven pair counts per offset, but at least 8 stages
  // each so the accumulator write-out stays small next to the MFMA work
  // (a dense layer, kvol = 1, has one evenly divisible pair list: fewer, longer ranges keep the fold pass — which reads
  // nsplit x cin x cout floats — negligible)
  static const int target_env = getenv("FSF_BWD_TARGET_WGS") ? atoi(getenv("FSF_BWD_TARGET_WGS")) : 0;  // (A/B switch)
  int64_t s = fsf_cdiv(target_env > 0 ? target_env : (kvol >= 8 ? 6144 : 512), kvol * tiles);
  const int64_t max_s = fsf_cdiv(cap, 8 * BW_RT);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int64_t r = fsf_align_up(fsf_cdiv(cap, s), BW_RT);
  if (r < BW_RT) r = BW_RT;
  *range = (int)r;
  *nsplit = fsf_cdiv(cap > 0 ? cap : 1, r);
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_spconv_backward_weight_workspace_bytes(int64_t cap, int32_t cin, int32_t cout, int32_t kvol) {
  int ta, tb, nsplit, range;
  bwd_plan(cap, cin, cout, kvol, &ta, &tb, &nsplit, &range);
  return nsplit > 1 ? fsf_align_up((int64_t)kvol * nsplit * cin * cout * 4, 256) + 256 : 256;
}

extern "C" int fsf_spconv_backward_weight(const float* feat, int64_t m_in, int32_t cin, const float* grad_out, int64_t m_out,
                                          int32_t cout, const int32_t* indice_pairs, const int32_t* indice_num, int64_t cap,
                                          int32_t kvol, float* grad_weight, void* workspace, int64_t workspace_bytes,
                                          void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool identity = indice_pairs == nullptr && indice_num == nullptr;  // dense layer: pair p = (row p, row p), kvol 1
  if (m_in < 0 || m_out < 0 || cin < 1 || cout < 1 || kvol < 1 || cap < 0 || !grad_weight ||
      (!identity && (!indice_num || (cap > 0 && !indice_pairs))) || (identity && (kvol != 1 || cap > m_in || cap > m_out)) ||
      (cap > 0 && (!feat || !grad_out)))
    return FSF_ERR_INVALID_ARG;
  if ((cin % 4) != 0 || (cout % 4) != 0 || cap >= ((int64_t)1 << 31)) return FSF_ERR_UNSUPPORTED;
  int ta, tb, nsplit, range;
  bwd_plan(cap, cin, cout, kvol, &ta, &tb, &nsplit, &range);
  if (workspace_bytes < fsf_spconv_backward_weight_workspace_bytes(cap, cin, cout, kvol) || (nsplit > 1 && !workspace))
    return FSF_ERR_WORKSPACE;
  SpconvBwdArgs a{feat, grad_out, indice_pairs, indice_num, grad_weight, (float*)workspace, cap,
                  (int)cin, (int)cout, (int)kvol, nsplit, range, fsf_cdiv(cout, tb)};
  const dim3 grid((unsigned)nsplit, (unsigned)kvol, (unsigned)(fsf_cdiv(cin, ta) * fsf_cdiv(cout, tb)));
#define FSF_BWD_LAUNCH(TA_, TB_)                                                                                         \
  do {                                                                                                                   \
    static std::atomic<uint64_t> attr_done{0};                                                                                        \
    const size_t smem_bytes = BwdSmem<TA_, TB_>::bytes();                                                                \
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)spconv_bwd_weight_kernel<TA_, TB_>, (int)smem_bytes, attr_done));                                                                                                                    \
    hipLaunchKernelGGL((spconv_bwd_weight_kernel<TA_, TB_>), grid, dim3(256), smem_bytes, stream, a);                    \
  } while (0)
  if (cap > 0) {
    if (ta == 64 && tb == 64) FSF_BWD_LAUNCH(64, 64);
    else if (ta == 64) FSF_BWD_LAUNCH(64, 128);
    else if (tb == 64) FSF_BWD_LAUNCH(128, 64);
    else FSF_BWD_LAUNCH(128, 128);
  }
#undef FSF_BWD_LAUNCH
  hipLaunchKernelGGL(spconv_bwd_fold_kernel, dim3(fsf_stream_grid((int64_t)kvol * cin * cout / 4 * 16, 256)), dim3(256), 0, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
