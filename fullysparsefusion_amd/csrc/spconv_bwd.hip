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

// =====================================================================================================================
// K10p (round 4): the same weight gradient on the bf16 matrix cores from an EXACT three-way split of both operands.
//
// The kernel above sits on v_mfma_f32_16x16x4_f32 — 1/16 of the 16-bit rate — at 0.5 of that pipe's peak.  Here, as in K22 / K9b:
// x = hi + mid + lo with each piece the next 8 significant bits (truncation: exact, three bf16 cover the 24-bit significand), a * b =
// the six leading cross products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (dropped terms < 2^-23 of the product): fp32
// accuracy at 2.7x the fp32 pipe's ceiling.
// The reduction index of these GEMMs is the PAIR, and both operands arrive pair-major (a gathered feature row = one k index), while
// the 16-bit MFMA wants 8 consecutive k of one channel in a lane's register.  So a stage of 32 pairs takes a detour through registers:
//   * thread (channel quad cq = tid % 32, pair quad pq = tid / 32) loads a 4 pair x 4 channel micro-tile of either operand — for a
//     fixed pair the 32 lanes of a half-wave read one whole 512-byte row: line-coalesced;
//   * it splits the 16 values, and for each of its 4 channels packs the 4 pairs' hi (mid, lo) pieces into 8 bytes = 4 CONSECUTIVE k
//     of that channel, written to a channel-major LDS image [plane][row(channel)][32 k] of 64-byte rows.  Bank arithmetic decides the
//     rest: the bank period (256 B) is four rows, and at step c every thread writes a channel = c mod 4 — with rows in channel order
//     all 64 lanes of a write would share a quarter of the banks (8-way conflicts; measured: the first version of this kernel spent
//     as long in its LDS writes as in its MFMAs).  So channel ch lives in row (ch & 3) * 32 + (ch >> 2), a wave's lanes are
//     (channel quad cq & 7, pair quad pq) — 8 consecutive quads still cover a whole 128-byte line of a gathered row — and the
//     16-byte unit of a row is XOR-ed with ch & 3: a write instruction then touches every bank exactly twice (the minimum for
//     512 bytes) and a fragment read — lane (m, kg) takes unit kg of channel 16 t + m — every bank once;
//   * each wave owns a 64 x 64 corner of the 128 x 128 tile of grad_W[k]: its four A tiles' fragments (3 planes) stay in registers
//     for the stage, the B tiles' stream through; 96 MFMAs per wave and stage.
// One LDS image (48 KB: two workgroups per CU) and the NEXT stage's micro-tiles in flight in registers during the MFMAs; two barriers
// per stage (image free -> written -> read).  Partial tiles and the fold kernel are the fp32 kernel's.
typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned bw_u32x2 __attribute__((ext_vector_type(2)));

constexpr int BWS_ROW = 64;                       // bytes per channel row of a plane: 32 k x bf16
constexpr int BWS_PLANE = 128 * BWS_ROW;          // one plane of one operand
constexpr int BWS_OPER = 3 * BWS_PLANE;           // hi | mid | lo
constexpr int BWS_SMEM = 2 * BWS_OPER;            // A | B: 48 KB

// 4 values (the same channel of 4 consecutive pairs) -> three 8-byte groups of bf16 (element e in the low / high half of dword e / 2)
__device__ __forceinline__ void bws_split4(const float (&v)[4], bw_u32x2& hi, bw_u32x2& mid, bw_u32x2& lo) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
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

__global__ void __launch_bounds__(256, 2) spconv_bwd_weight_split_kernel(SpconvBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char bws_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.y;
  const int split = blockIdx.x;
  const int a0 = (blockIdx.z / a.tiles_b) * 128;
  const int b0 = (blockIdx.z % a.tiles_b) * 128;
  const int n = a.num ? a.num[k] : (int)a.cap;
  const int p_begin = split * a.range;
  if (p_begin >= n) return;
  const int p_end = min(n, p_begin + a.range);
  const int nstages = (p_end - p_begin + BW_RT - 1) / BW_RT;
  const bool ident = a.pairs == nullptr;
  const int32_t* pin = ident ? nullptr : a.pairs + (int64_t)k * 2 * a.cap;
  const int32_t* pout = ident ? nullptr : pin + a.cap;

  // ---- loader role: micro-tile (pairs 4 pq .. 4 pq + 3) x (channels 4 cq .. 4 cq + 3) of both operands
  const int cq = 8 * wave + (lane & 7), pq = lane >> 3;
  int ca = a0 + 4 * cq;
  ca = ca + 4 <= a.cin ? ca : a.cin - 4;   // (channels past cin / cout only feed accumulators that are never written out)
  int cb = b0 + 4 * cq;
  cb = cb + 4 <= a.cout ? cb : a.cout - 4;
  // LDS byte offset of this thread's 8-byte group for channel 4 cq + c: row c * 32 + cq, unit (pq >> 1) ^ c, half pq & 1
  const int wr_row = cq * BWS_ROW + (pq & 1) * 8, wr_unit = pq >> 1;

  int32_t ia[4], ib[4];
  f32x4 xa[4], xb[4];
  auto load_indices = [&](int stage) {
    const int p0 = p_begin + stage * BW_RT + 4 * pq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = min(p0 + r, p_end - 1);  // (clamped address; rows past the end are zeroed after the load)
      ia[r] = ident ? p : pin[p];
      ib[r] = ident ? p : pout[p];
    }
  };
  auto load_stage = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xa[r] = *reinterpret_cast<const f32x4*>(a.feat + (int64_t)ia[r] * a.cin + ca);
      xb[r] = *reinterpret_cast<const f32x4*>(a.gout + (int64_t)ib[r] * a.cout + cb);
    }
  };
  auto write_stage = [&](int stage) {
    const int p0 = p_begin + stage * BW_RT + 4 * pq;
#pragma unroll
    for (int oper = 0; oper < 2; ++oper) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        char* base = bws_smem + oper * BWS_OPER + wr_row + (c * 32) * BWS_ROW + ((wr_unit ^ c) * 16);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = p0 + r < p_end ? (oper == 0 ? xa[r][c] : xb[r][c]) : 0.0f;
        bw_u32x2 hi, mid, lo;
        bws_split4(v, hi, mid, lo);
        *reinterpret_cast<bw_u32x2*>(base) = hi;
        *reinterpret_cast<bw_u32x2*>(base + BWS_PLANE) = mid;
        *reinterpret_cast<bw_u32x2*>(base + 2 * BWS_PLANE) = lo;
      }
    }
  };

  // ---- MFMA role: wave (wa, wb) owns channels [64 wa, +64) x [64 wb, +64) of the tile
  const int wa = wave >> 1, wb = wave & 1;
  const int m = lane & 15, kg = lane >> 4;
  // channel 64 w + 16 t + m lives in row (m & 3) * 32 + 16 w + 4 t + (m >> 2), its unit kg at kg ^ (m & 3)
  const int rd_off = ((m & 3) * 32 + (m >> 2)) * BWS_ROW + ((kg ^ (m & 3)) * 16);
  f32x4 acc[4][4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_indices(0);
  load_stage();
  if (nstages > 1) load_indices(1);
  for (int s = 0; s < nstages; ++s) {
    if (s > 0) __syncthreads();  // every wave is done reading the image of stage s - 1
    write_stage(s);              // (waits for this stage's micro-tiles)
    __syncthreads();
    if (s + 1 < nstages) {       // the next stage's rows travel under this stage's MFMAs
      load_stage();
      if (s + 2 < nstages) load_indices(s + 2);
    }
    const char* As = bws_smem + (16 * wa) * BWS_ROW + rd_off;
    const char* Bs = bws_smem + BWS_OPER + (16 * wb) * BWS_ROW + rd_off;
    bw_bf16x8 af[4][3];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        af[ta][pl] = *reinterpret_cast<const bw_bf16x8*>(As + (4 * ta) * BWS_ROW + pl * BWS_PLANE);
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      bw_bf16x8 bf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bw_bf16x8*>(Bs + (4 * tb) * BWS_ROW + pl * BWS_PLANE);
      // (plane of A, plane of B) of the six leading cross terms, small ones first
      constexpr int TERM_A[6] = {2, 0, 1, 1, 0, 0};
      constexpr int TERM_B[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ta][TERM_A[term]], bf[TERM_B[term]], acc[ta][tb], 0, 0, 0);
    }
  }

  // D of tile (ta, tb): lane (n = lane % 16, g = lane / 16) holds rows ci = 16 ta + 4 g + r, column co = 16 tb + n
  float* dst = a.nsplit > 1 ? a.part + ((int64_t)k * a.nsplit + split) * a.cin * a.cout : a.gw + (int64_t)k * a.cin * a.cout;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      const int co = b0 + 64 * wb + 16 * tb + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = a0 + 64 * wa + 16 * ta + 4 * kg + r;
        if (ci < a.cin && co < a.cout) dst[(int64_t)ci * a.cout + co] = acc[ta][tb][r];
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
  int64_t s = fsf_cdiv(kvol >= 8 ? 6144 : 512, kvol * tiles);
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
  if (cap > 0 && ta == 128 && tb == 128) {
    static std::atomic<uint64_t> attr_done{0};
    FSF_HIP_TRY(fsf_set_max_dynamic_lds((const void*)spconv_bwd_weight_split_kernel, BWS_SMEM, attr_done));
    hipLaunchKernelGGL(spconv_bwd_weight_split_kernel, grid, dim3(256), BWS_SMEM, stream, a);
  } else if (cap > 0) {
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
