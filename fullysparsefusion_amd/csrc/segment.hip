// K4/K5/K6: segmented sum/mean/max over sort-once segment plans, row gather, the voxel->point neck.
// See include/fsf_hip.h.  No atomics: the sorted row list is cut into fixed 32-row chunks; a lane team
// (channels across lanes, float4 where the row width allows) walks one chunk, writes every segment that
// lies wholly inside it straight to `out`, and parks the (at most two) pieces that straddle a chunk
// boundary in a partial buffer which a second pass folds in chunk order.  The summation order is a pure
// function of the plan, so results are run-to-run deterministic.
// Algorithmic HBM bytes: 4C B/row + 12 B/row of indices read, 4C B/segment written
// (+8C B/segment when argmax is kept).
#include "common.h"

namespace fsf {

#ifndef SEG_CHUNK_ROWS
#define SEG_CHUNK_ROWS 32
#endif
constexpr int SEG_CHUNK = SEG_CHUNK_ROWS;
constexpr int SEG_BLOCK = 256;
constexpr int SEG_MIN_TEAM = 4;
constexpr int SEG_LONG_SPAN = 16;  // chunks; longer segments are folded by a whole workgroup

enum { MODE_SUM = 0, MODE_MEAN = 1, MODE_MAX = 2 };

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  float v[1];
};
template <>
struct Vec<4> {
  float v[4];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> load_vec(const float* p) {
  Vec<VEC> r;
  if constexpr (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const Vec<VEC>& r) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
    *p = r.v[0];
  }
}

struct SegArgs {
  const float* feat;
  const int32_t* order;
  const int64_t* inv;
  const int32_t* seg_offsets;
  float* out;
  int64_t* argmax;
  float* part_val;    // [nchunks, 2, c]
  int32_t* part_arg;  // [nchunks, 2, c] (max + argmax only)
  int64_t n;
  int64_t m;
  int c;
  int team;  // lanes per team (power of two, 4..64)
  int64_t feat_stride;  // row stride of feat in floats (>= c)
};

template <int VEC, int MODE>
__device__ __forceinline__ void seg_flush(const SegArgs& a, int64_t chunk, int seg, int piece_lo, int piece_hi, int ch,
                                          const Vec<VEC>& acc, const int32_t* arg) {
  const int S = a.seg_offsets[seg];
  const int E = a.seg_offsets[seg + 1];
  if (piece_lo == S && piece_hi == E) {
    Vec<VEC> r = acc;
    if constexpr (MODE == MODE_MEAN) {
      const float cntf = (float)(E - S);
#pragma unroll
      for (int q = 0; q < VEC; ++q) r.v[q] = __fdiv_rn(r.v[q], cntf);
    }
    store_vec<VEC>(a.out + (int64_t)seg * a.c + ch, r);
    if constexpr (MODE == MODE_MAX) {
      if (a.argmax) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) a.argmax[(int64_t)seg * a.c + ch + q] = (int64_t)arg[q];
      }
    }
  } else {
    const int which = (S < piece_lo) ? 0 : 1;  // 0: continues an earlier chunk, 1: continues into the next
    const int64_t slot = (chunk * 2 + which) * a.c + ch;
    store_vec<VEC>(a.part_val + slot, acc);
    if constexpr (MODE == MODE_MAX) {
      if (a.argmax) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) a.part_arg[slot + q] = arg[q];
      }
    }
  }
}

template <int VEC, int MODE>
__global__ void __launch_bounds__(SEG_BLOCK) seg_reduce_kernel(SegArgs a) {
  // per wave: (64/team) teams x 32 rows of (point, segment) pairs staged once through LDS
  __shared__ int2 rows[SEG_BLOCK / 64][(64 / SEG_MIN_TEAM) * SEG_CHUNK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int teams_per_wave = 64 / a.team;
  const int team_in_wave = lane / a.team;
  const int tl = lane % a.team;
  const int64_t nchunks = (a.n + SEG_CHUNK - 1) / SEG_CHUNK;
  const int64_t waves_total = (int64_t)gridDim.x * (SEG_BLOCK / 64);
  const int64_t wave_id = (int64_t)blockIdx.x * (SEG_BLOCK / 64) + wave;
  const int rows_per_wave = teams_per_wave * SEG_CHUNK;
  const float ident = (MODE == MODE_MAX) ? -INFINITY : 0.0f;

  for (int64_t wchunk0 = wave_id * teams_per_wave; wchunk0 < nchunks; wchunk0 += waves_total * teams_per_wave) {
    const int64_t row0 = wchunk0 * SEG_CHUNK;
    for (int r = lane; r < rows_per_wave; r += 64) {
      const int64_t j = row0 + r;
      int2 e = make_int2(-1, -1);
      if (j < a.n) {
        e.x = a.order[j];
        e.y = (int)a.inv[e.x];
      }
      rows[wave][r] = e;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int64_t chunk = wchunk0 + team_in_wave;
    if (chunk < nchunks) {
      const int lo = (int)(chunk * SEG_CHUNK);
      const int hi = (int)((lo + SEG_CHUNK < a.n) ? lo + SEG_CHUNK : a.n);
      const int2* my_rows = &rows[wave][team_in_wave * SEG_CHUNK];
      for (int ch = tl * VEC; ch < a.c; ch += a.team * VEC) {
        Vec<VEC> acc;
        int32_t arg[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          acc.v[q] = ident;
          arg[q] = -1;
        }
        int cur = -1, piece_lo = lo;
        for (int j0 = lo; j0 < hi; j0 += 4) {
          int2 e[4];
          Vec<VEC> v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            e[u] = (j0 + u < hi) ? my_rows[j0 + u - lo] : make_int2(-1, -1);
            if (e[u].x >= 0) v[u] = load_vec<VEC>(a.feat + (int64_t)e[u].x * a.feat_stride + ch);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e[u].x < 0) continue;
            if (e[u].y != cur) {
              if (cur >= 0) seg_flush<VEC, MODE>(a, chunk, cur, piece_lo, j0 + u, ch, acc, arg);
              cur = e[u].y;
              piece_lo = j0 + u;
#pragma unroll
              for (int q = 0; q < VEC; ++q) {
                acc.v[q] = ident;
                arg[q] = -1;
              }
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              if constexpr (MODE == MODE_MAX) {
                if (v[u].v[q] > acc.v[q] || arg[q] < 0) {
                  acc.v[q] = v[u].v[q];
                  arg[q] = e[u].x;
                }
              } else {
                acc.v[q] = __fadd_rn(acc.v[q], v[u].v[q]);
              }
            }
          }
        }
        if (cur >= 0) seg_flush<VEC, MODE>(a, chunk, cur, piece_lo, hi, ch, acc, arg);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// second pass: segments that straddle a chunk boundary (fold the partials in chunk order) and empty
// segments (torch_scatter: out = 0, arg = n).  `team_id` of `teams_total` lane teams walks the segments.
template <int VEC, int MODE>
__device__ __forceinline__ void seg_fixup_short(const SegArgs& a, int64_t team_id, int64_t teams_total, int tl) {
  for (int64_t s = team_id; s < a.m; s += teams_total) {
    const int S = a.seg_offsets[s];
    const int E = a.seg_offsets[s + 1];
    if (E == S) {
      for (int ch = tl * VEC; ch < a.c; ch += a.team * VEC) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          a.out[s * a.c + ch + q] = 0.0f;
          if (MODE == MODE_MAX && a.argmax) a.argmax[s * a.c + ch + q] = a.n;
        }
      }
      continue;
    }
    const int cs = S / SEG_CHUNK, ce = (E - 1) / SEG_CHUNK;
    if (cs == ce || ce - cs > SEG_LONG_SPAN) continue;  // long segments: the workgroup-wide fold
    for (int ch = tl * VEC; ch < a.c; ch += a.team * VEC) {
      Vec<VEC> acc = load_vec<VEC>(a.part_val + ((int64_t)cs * 2 + 1) * a.c + ch);
      int32_t arg[VEC];
      if (MODE == MODE_MAX && a.argmax) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) arg[q] = a.part_arg[((int64_t)cs * 2 + 1) * a.c + ch + q];
      }
      for (int cc = cs + 1; cc <= ce; ++cc) {
        const int64_t slot = ((int64_t)cc * 2 + 0) * a.c + ch;
        Vec<VEC> v = load_vec<VEC>(a.part_val + slot);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          if constexpr (MODE == MODE_MAX) {
            if (v.v[q] > acc.v[q]) {
              acc.v[q] = v.v[q];
              if (a.argmax) arg[q] = a.part_arg[slot + q];
            }
          } else {
            acc.v[q] = __fadd_rn(acc.v[q], v.v[q]);
          }
        }
      }
      if constexpr (MODE == MODE_MEAN) {
        const float cntf = (float)(E - S);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc.v[q] = __fdiv_rn(acc.v[q], cntf);
      }
      store_vec<VEC>(a.out + s * a.c + ch, acc);
      if (MODE == MODE_MAX && a.argmax) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) a.argmax[s * a.c + ch + q] = (int64_t)arg[q];
      }
    }
  }
}

template <int VEC, int MODE>
__global__ void __launch_bounds__(SEG_BLOCK) seg_fixup_kernel(SegArgs a) {
  const int lane = threadIdx.x & 63;
  const int teams_per_wave = 64 / a.team;
  seg_fixup_short<VEC, MODE>(a, ((int64_t)blockIdx.x * (SEG_BLOCK / 64) + (threadIdx.x >> 6)) * teams_per_wave + lane / a.team,
                             (int64_t)gridDim.x * (SEG_BLOCK / 64) * teams_per_wave, lane % a.team);
}

// segments spanning more than SEG_LONG_SPAN chunks (SIR groups with 1e3..1e5 points): one workgroup per
// segment; its lane teams stride over the chunk partials, then the team results are folded in team order
// through LDS.  Ties in max keep the smaller point index (= first in the stable sorted order).
template <int VEC, int MODE>
__global__ void __launch_bounds__(SEG_BLOCK) seg_fixup_long_kernel(SegArgs a) {
  __shared__ float s_val[SEG_BLOCK / SEG_MIN_TEAM][SEG_MIN_TEAM * 4];
  __shared__ int32_t s_arg[SEG_BLOCK / SEG_MIN_TEAM][SEG_MIN_TEAM * 4];
  // gridDim.y workgroups share a long segment by channel slice: the one workgroup per segment was the whole kernel's
  // critical path (a 1e5-point group = 3 000 partials of 128 channels); with narrower slices more lane teams stride the
  // partials.  (max / argmax do not depend on the fold order; sums keep a fixed one for a given launch shape.)
  {  // the short straddlers and the empty segments first (what seg_fixup_kernel does when no segment can be long): one launch
    const int lane = threadIdx.x & 63;
    const int teams_per_wave = 64 / a.team;
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nblk = (int64_t)gridDim.x * gridDim.y;
    seg_fixup_short<VEC, MODE>(a, (blk * (SEG_BLOCK / 64) + (threadIdx.x >> 6)) * teams_per_wave + lane / a.team,
                               nblk * (SEG_BLOCK / 64) * teams_per_wave, lane % a.team);
  }
  const int csl = (int)((a.c + (int)gridDim.y * VEC - 1) / ((int)gridDim.y * VEC)) * VEC;  // channels per slice
  const int c_lo = (int)blockIdx.y * csl, c_hi = c_lo + csl < a.c ? c_lo + csl : a.c;
  if (c_lo >= a.c) return;
  int team = SEG_MIN_TEAM;
  while (team * VEC < csl && team < 64) team <<= 1;
  const int nteams = SEG_BLOCK / team;
  const int team_id = threadIdx.x / team;
  const int tl = threadIdx.x % team;
  const float ident = (MODE == MODE_MAX) ? -INFINITY : 0.0f;
  for (int64_t s = blockIdx.x; s < a.m; s += gridDim.x) {
    const int S = a.seg_offsets[s];
    const int E = a.seg_offsets[s + 1];
    if (E == S) continue;
    const int cs = S / SEG_CHUNK, ce = (E - 1) / SEG_CHUNK;
    if (ce - cs <= SEG_LONG_SPAN) continue;
    for (int ch0 = c_lo; ch0 < c_hi; ch0 += team * VEC) {  // workgroup-uniform trip count (barriers inside)
      const int ch = ch0 + tl * VEC;
      const bool act = ch < c_hi;
      Vec<VEC> acc;
      int32_t arg[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        acc.v[q] = ident;
        arg[q] = INT32_MAX;
      }
      // virtual partial index p in [0, ce-cs]: p = 0 is the head piece (slot 1 of chunk cs), p > 0 slot 0 of cs+p
      // four partials are requested before the first is folded (same fold order as one by one, 4x the loads in
      // flight: a 30 k-point camera frustum is ~940 partials for its one workgroup)
      constexpr int UNR = 4;
      for (int p0 = team_id; act && p0 <= ce - cs; p0 += nteams * UNR) {
        Vec<VEC> v[UNR];
        int32_t va[UNR][VEC];
        bool live[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int p = p0 + u * nteams;
          live[u] = p <= ce - cs;
          const int pc = live[u] ? p : p0;
          const int64_t slot = ((int64_t)(cs + pc) * 2 + (pc == 0 ? 1 : 0)) * a.c + ch;
          v[u] = load_vec<VEC>(a.part_val + slot);
#pragma unroll
          for (int q = 0; q < VEC; ++q) va[u][q] = (MODE == MODE_MAX && a.argmax) ? a.part_arg[slot + q] : 0;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (!live[u]) continue;
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            if constexpr (MODE == MODE_MAX) {
              if (v[u].v[q] > acc.v[q] || (v[u].v[q] == acc.v[q] && va[u][q] < arg[q])) {
                acc.v[q] = v[u].v[q];
                arg[q] = va[u][q];
              }
            } else {
              acc.v[q] = __fadd_rn(acc.v[q], v[u].v[q]);
            }
          }
        }
      }
      // fold the team results in team order (fixed, deterministic)
      __syncthreads();
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        (&s_val[0][0])[(team_id * team + tl) * VEC + q] = acc.v[q];
        (&s_arg[0][0])[(team_id * team + tl) * VEC + q] = arg[q];
      }
      __syncthreads();
      if (team_id == 0 && act) {
        for (int t = 1; t < nteams; ++t) {
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const float v = (&s_val[0][0])[(t * team + tl) * VEC + q];
            if constexpr (MODE == MODE_MAX) {
              const int32_t va = (&s_arg[0][0])[(t * team + tl) * VEC + q];
              if (v > acc.v[q] || (v == acc.v[q] && va < arg[q])) {
                acc.v[q] = v;
                arg[q] = va;
              }
            } else {
              acc.v[q] = __fadd_rn(acc.v[q], v);
            }
          }
        }
        if constexpr (MODE == MODE_MEAN) {
          const float cntf = (float)(E - S);
#pragma unroll
          for (int q = 0; q < VEC; ++q) acc.v[q] = __fdiv_rn(acc.v[q], cntf);
        }
        store_vec<VEC>(a.out + s * a.c + ch, acc);
        if (MODE == MODE_MAX && a.argmax) {
#pragma unroll
          for (int q = 0; q < VEC; ++q) a.argmax[s * a.c + ch + q] = (int64_t)arg[q];
        }
      }
    }
  }
}

// ---- short segments (voxels: a few rows each, never more than a few hundred): thread = (segment, channel [quad]) walks the
// segment's rows in sorted order — no chunking, no partials, no fix-up launches; several tensors over the same plan in ONE launch
// (pre_voxelize reduces five).  Correct for any segment length (a long segment just serialises on its threads).
constexpr int SEG_SHORT_MAX = 8;
struct SegShortArgs {
  const float* feat[SEG_SHORT_MAX];
  float* out[SEG_SHORT_MAX];
  int64_t stride[SEG_SHORT_MAX];
  int c[SEG_SHORT_MAX];
  int goff[SEG_SHORT_MAX + 1];  // prefix sums of the tensors' channel groups (c / VEC each)
  int nt;
  const int32_t* order;
  const int32_t* seg_offsets;
  int64_t n, m;
  int64_t* argmax;  // max over ONE tensor only
};

template <int VEC, int MODE>
__global__ void __launch_bounds__(256) seg_short_kernel(SegShortArgs a) {
  const int gtot = a.goff[a.nt];
  const int64_t total = a.m * gtot;
  const bool small = total <= 0x7fffffff;  // (32-bit index arithmetic where it fits)
  // (four (segment, channel) items per thread with their first rows' loads issued together was tried: 40-60 % SLOWER —
  // the extra registers and the divergent folds cost more than the deeper load queue gained)
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t seg = small ? (int64_t)((uint32_t)idx / (uint32_t)gtot) : idx / gtot;
    const int gg = (int)(idx - seg * gtot);
    int t = 0;
#pragma unroll
    for (int u = 1; u < SEG_SHORT_MAX; ++u) t += (u < a.nt && gg >= a.goff[u]) ? 1 : 0;
    const int ch = (gg - a.goff[t]) * VEC;
    const float* __restrict__ f = a.feat[t] + ch;
    const int64_t stride = a.stride[t];
    const int S = a.seg_offsets[seg], E = a.seg_offsets[seg + 1];
    Vec<VEC> acc;
    int32_t arg[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      acc.v[q] = (MODE == MODE_MAX && E > S) ? -INFINITY : 0.0f;
      arg[q] = -1;
    }
    for (int j0 = S; j0 < E; j0 += 4) {
      int r[4];
      Vec<VEC> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r[u] = j0 + u < E ? a.order[j0 + u] : -1;
        if (r[u] >= 0) v[u] = load_vec<VEC>(f + (int64_t)r[u] * stride);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r[u] < 0) continue;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          if constexpr (MODE == MODE_MAX) {
            if (v[u].v[q] > acc.v[q] || arg[q] < 0) {
              acc.v[q] = v[u].v[q];
              arg[q] = r[u];
            }
          } else {
            acc.v[q] = __fadd_rn(acc.v[q], v[u].v[q]);
          }
        }
      }
    }
    if constexpr (MODE == MODE_MEAN) {
      if (E > S) {
        const float cntf = (float)(E - S);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc.v[q] = __fdiv_rn(acc.v[q], cntf);
      }
    }
    store_vec<VEC>(a.out[t] + seg * a.c[t] + ch, acc);
    if (MODE == MODE_MAX && a.argmax) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) a.argmax[seg * a.c[t] + ch + q] = E > S ? (int64_t)arg[q] : a.n;
    }
  }
}

static int pick_team(int c, int vec) {
  int groups = (c + vec - 1) / vec;
  int t = SEG_MIN_TEAM;
  while (t < groups && t < 64) t <<= 1;
  return t;
}

template <int VEC>
__global__ void __launch_bounds__(256)
    gather_rows_kernel(const float* __restrict__ src, int64_t src_stride, const int64_t* __restrict__ idx, int64_t n, int c,
                       float* __restrict__ out, int64_t out_stride, const float* __restrict__ add = nullptr, int64_t add_stride = 0) {
  const int cv = c / VEC;
  const int64_t total = n * cv;
  if (total <= 0x7fffffff) {  // (a 64-bit division per element made the 131-channel gather instruction-bound: 315 us for 0.53 GB)
    // four elements per thread and round: the row index, then the element — two dependent loads — are each in flight four deep
    const uint32_t tot = (uint32_t)total, step = gridDim.x * blockDim.x, ucv = (uint32_t)cv;
    for (uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x; t0 < tot; t0 += 4 * step) {
      uint32_t i[4];
      int col[4];
      int64_t row[4];
      Vec<VEC> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t t = t0 + u * step;
        i[u] = (t < tot ? t : t0) / ucv;
        col[u] = (int)((t < tot ? t : t0) - i[u] * ucv) * VEC;
        row[u] = idx[i[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load_vec<VEC>(src + row[u] * src_stride + col[u]);
      if (add) {  // out = add + src[idx]  (the per-group half of a Linear over cat([point, group[inv]]), training forward)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const Vec<VEC> a = load_vec<VEC>(add + (int64_t)i[u] * add_stride + col[u]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[u].v[e] = a.v[e] + v[u].v[e];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t0 + u * step < tot) store_vec<VEC>(out + (int64_t)i[u] * out_stride + col[u], v[u]);
    }
    return;
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / cv;
    const int col = (int)(t - i * cv) * VEC;
    Vec<VEC> v = load_vec<VEC>(src + idx[i] * src_stride + col);
    if (add) {
      const Vec<VEC> a = load_vec<VEC>(add + i * add_stride + col);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v.v[e] = a.v[e] + v.v[e];
    }
    store_vec<VEC>(out + i * out_stride + col, v);
  }
}

// decoder shortcut of SimpleSparseUNet: out[i, j] = add[i, j] + sum_{q < r} feat[i, j * r + q]   (r = cin / cout)
template <int R>
__global__ void __launch_bounds__(256)
    channel_group_sum_add_kernel(const float* __restrict__ feat, const float* __restrict__ add, int64_t n, int cout,
                                 float* __restrict__ out) {
  const int cv = cout / 4;
  const int64_t total = n * cv;
  const bool small = total <= 0x7fffffff;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = small ? (int64_t)((uint32_t)t / (uint32_t)cv) : t / cv;
    const int j = (int)(t - i * cv) * 4;
    const float* f = feat + (i * cout + j) * R;
    float4 o = add ? *reinterpret_cast<const float4*>(add + i * cout + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // torch's sum(dim=2) adds the r values of a group in index order
      float s = f[e * R];
#pragma unroll
      for (int q = 1; q < R; ++q) s = __fadd_rn(s, f[e * R + q]);
      acc[e] = s;
    }
    o.x = __fadd_rn(o.x, acc[0]); o.y = __fadd_rn(o.y, acc[1]); o.z = __fadd_rn(o.z, acc[2]); o.w = __fadd_rn(o.w, acc[3]);
    *reinterpret_cast<float4*>(out + i * cout + j) = o;
  }
}

// the same on the channel concatenation [a | b] WITHOUT materialising it (r = 2; widths multiples of 8, so the eight input columns of
// an output quad lie in one source): out[i, j] = add[i, j] + cat[i, 2 j] + cat[i, 2 j + 1]
__global__ void __launch_bounds__(256)
    channel_pair_sum_add2_kernel(const float* __restrict__ fa, int ca, const float* __restrict__ fb, int cb, const float* __restrict__ add,
                                 int64_t n, int cout, float* __restrict__ out) {
  const int cv = cout / 4;
  const int64_t total = n * cv;
  const bool small = total <= 0x7fffffff;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = small ? (int64_t)((uint32_t)t / (uint32_t)cv) : t / cv;
    const int j = (int)(t - i * cv) * 4;
    const float* f = 2 * j < ca ? fa + i * ca + 2 * j : fb + i * cb + (2 * j - ca);
    const float4 p = *reinterpret_cast<const float4*>(f), q = *reinterpret_cast<const float4*>(f + 4);
    float4 o = add ? *reinterpret_cast<const float4*>(add + i * cout + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x = __fadd_rn(o.x, __fadd_rn(p.x, p.y)); o.y = __fadd_rn(o.y, __fadd_rn(p.z, p.w));
    o.z = __fadd_rn(o.z, __fadd_rn(q.x, q.y)); o.w = __fadd_rn(o.w, __fadd_rn(q.z, q.w));
    *reinterpret_cast<float4*>(out + i * cout + j) = o;
  }
}

struct V2PParams {
  float vx, vy, vz, xmin, ymin, zmin, padding;
};

// one wave-row team per point: lanes stride the c voxel channels; lane 0..2 of the team add local xyz
__global__ void __launch_bounds__(256)
    voxel2point_kernel(const float* __restrict__ points, int stride, const int64_t* __restrict__ coors,
                       const float* __restrict__ vf, int c, const int64_t* __restrict__ inv, int64_t n, V2PParams p,
                       float* __restrict__ out, int64_t oc, uint8_t* __restrict__ valid, int team) {
  const int lane = threadIdx.x & 63;
  const int tl = lane % team;
  const int teams_per_block = 256 / team;
  for (int64_t i = (int64_t)blockIdx.x * teams_per_block + threadIdx.x / team; i < n;
       i += (int64_t)gridDim.x * teams_per_block) {
    const int64_t row = inv[i];
    bool all_pad = true;
    for (int ch = tl; ch < c; ch += team) {
      const float v = vf[row * c + ch];
      all_pad &= (v == p.padding);
      out[i * oc + ch] = v;
    }
    // team-wide AND of all_pad
    for (int o = team >> 1; o > 0; o >>= 1) all_pad &= (bool)__shfl_xor((int)all_pad, o);
    if (tl < 3) {
      // (coor + 0.5) * voxel + range_min, column order x<-3, y<-2, z<-1 (voxel2point_neck.py:51)
      const float cf = (float)coors[i * 4 + (3 - tl)];
      const float vs = tl == 0 ? p.vx : (tl == 1 ? p.vy : p.vz);
      const float mn = tl == 0 ? p.xmin : (tl == 1 ? p.ymin : p.zmin);
      const float center = __fadd_rn(__fmul_rn(__fadd_rn(cf, 0.5f), vs), mn);
      out[i * oc + c + tl] = __fsub_rn(points[i * stride + tl], center);
    }
    if (tl == 0 && valid) valid[i] = all_pad ? 0 : 1;
  }
}

// the same on 16-byte lanes (c % 4 == 0, voxel rows and output rows 16-byte aligned: the 10-sweep frame's [92 k, 128] -> [310 k, 128 + 3]):
// a team of c / 4 lanes per point moves its row in one load and one store per lane
__global__ void __launch_bounds__(256)
    voxel2point_v4_kernel(const float* __restrict__ points, int stride, const int64_t* __restrict__ coors,
                          const float* __restrict__ vf, int c, const int64_t* __restrict__ inv, int64_t n, V2PParams p,
                          float* __restrict__ out, int64_t oc, uint8_t* __restrict__ valid, int team) {
  const int lane = threadIdx.x & 63;
  const int tl = lane % team;
  const int teams_per_block = 256 / team;
  const int cv = c >> 2;
  for (int64_t i = (int64_t)blockIdx.x * teams_per_block + threadIdx.x / team; i < n;
       i += (int64_t)gridDim.x * teams_per_block) {
    const int64_t row = inv[i];
    bool all_pad = true;
    for (int q = tl; q < cv; q += team) {
      const float4 v = *reinterpret_cast<const float4*>(vf + row * c + 4 * q);
      all_pad &= (v.x == p.padding) & (v.y == p.padding) & (v.z == p.padding) & (v.w == p.padding);
      *reinterpret_cast<float4*>(out + i * oc + 4 * q) = v;
    }
    for (int o = team >> 1; o > 0; o >>= 1) all_pad &= (bool)__shfl_xor((int)all_pad, o);
    if (tl < 3) {
      const float cf = (float)coors[i * 4 + (3 - tl)];
      const float vs = tl == 0 ? p.vx : (tl == 1 ? p.vy : p.vz);
      const float mn = tl == 0 ? p.xmin : (tl == 1 ? p.ymin : p.zmin);
      const float center = __fadd_rn(__fmul_rn(__fadd_rn(cf, 0.5f), vs), mn);
      out[i * oc + c + tl] = __fsub_rn(points[i * stride + tl], center);
    }
    if (tl == 0 && valid) valid[i] = all_pad ? 0 : 1;
  }
}

template <int MODE>
__global__ void __launch_bounds__(256)
    seg_backward_dense_kernel(const float* __restrict__ grad_out, int64_t n, int c, const int64_t* __restrict__ inv,
                              const int32_t* __restrict__ seg_offsets, float* __restrict__ grad_feat) {
  const int64_t total = n * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    const int64_t s = inv[i];
    float g = grad_out[s * c + ch];
    if (MODE == MODE_MEAN) {
      const int cnt = seg_offsets[s + 1] - seg_offsets[s];
      g = __fdiv_rn(g, (float)(cnt > 1 ? cnt : 1));
    }
    grad_feat[t] = g;
  }
}

__global__ void __launch_bounds__(256)
    seg_backward_max_kernel(const float* __restrict__ grad_out, int64_t m, int64_t n, int c,
                            const int64_t* __restrict__ argmax, float* __restrict__ grad_feat) {
  const int64_t total = m * c;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(t % c);
    const int64_t p = argmax[t];
    if (p >= 0 && p < n) grad_feat[p * c + ch] = grad_out[t];
  }
}

}  // namespace fsf

using namespace fsf;

extern "C" int64_t fsf_segment_reduce_workspace_bytes(int64_t n, int64_t m, int32_t c) {
  (void)m;
  const int64_t nchunks = (n + SEG_CHUNK - 1) / SEG_CHUNK + 1;
  return fsf_align_up(nchunks * 2 * c * 4, 256) * 2 + 256;
}

template <int VEC>
static int seg_launch(const SegArgs& a, int mode, hipStream_t stream) {
  const int64_t nchunks = (a.n + SEG_CHUNK - 1) / SEG_CHUNK;
  const int teams_per_block = (SEG_BLOCK / 64) * (64 / a.team);
  int64_t g1 = (nchunks + teams_per_block - 1) / teams_per_block;
  if (g1 < 1) g1 = 1;
  if (g1 > 8192) g1 = 8192;
  int64_t g2 = (a.m + teams_per_block - 1) / teams_per_block;
  if (g2 < 1) g2 = 1;
  if (g2 > 4096) g2 = 4096;
  // a segment can only be "long" when the input has more rows than SEG_LONG_SPAN chunks
  const bool has_long = a.n > (int64_t)SEG_LONG_SPAN * SEG_CHUNK;
  int64_t g3x = a.m < 1024 ? a.m : 1024;
  if (g3x < 1) g3x = 1;
  const int groups = (a.c + VEC - 1) / VEC;
  // (float4 rows only: with one float per lane a 16-lane team moves 64 B per partial and the narrow-row kernel got slower)
  const dim3 g3((unsigned)g3x, (unsigned)(VEC == 4 ? (groups >= 32 ? 4 : (groups >= 16 ? 2 : 1)) : 1));
  switch (mode) {
    case MODE_SUM:
      if (a.n > 0) hipLaunchKernelGGL((seg_reduce_kernel<VEC, MODE_SUM>), dim3((unsigned)g1), dim3(SEG_BLOCK), 0, stream, a);
      if (has_long) hipLaunchKernelGGL((seg_fixup_long_kernel<VEC, MODE_SUM>), g3, dim3(SEG_BLOCK), 0, stream, a);
      else hipLaunchKernelGGL((seg_fixup_kernel<VEC, MODE_SUM>), dim3((unsigned)g2), dim3(SEG_BLOCK), 0, stream, a);
      break;
    case MODE_MEAN:
      if (a.n > 0) hipLaunchKernelGGL((seg_reduce_kernel<VEC, MODE_MEAN>), dim3((unsigned)g1), dim3(SEG_BLOCK), 0, stream, a);
      if (has_long) hipLaunchKernelGGL((seg_fixup_long_kernel<VEC, MODE_MEAN>), g3, dim3(SEG_BLOCK), 0, stream, a);
      else hipLaunchKernelGGL((seg_fixup_kernel<VEC, MODE_MEAN>), dim3((unsigned)g2), dim3(SEG_BLOCK), 0, stream, a);
      break;
    default:
      if (a.n > 0) hipLaunchKernelGGL((seg_reduce_kernel<VEC, MODE_MAX>), dim3((unsigned)g1), dim3(SEG_BLOCK), 0, stream, a);
      if (has_long) hipLaunchKernelGGL((seg_fixup_long_kernel<VEC, MODE_MAX>), g3, dim3(SEG_BLOCK), 0, stream, a);
      else hipLaunchKernelGGL((seg_fixup_kernel<VEC, MODE_MAX>), dim3((unsigned)g2), dim3(SEG_BLOCK), 0, stream, a);
      break;
  }
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_segment_reduce(const float* feat, int64_t feat_stride, int64_t n, int32_t c, const int32_t* order, const int64_t* inv,
                                  const int32_t* seg_offsets, int64_t m, int32_t mode, float* out, int64_t* argmax,
                                  void* workspace, int64_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || c < 1 || mode < 0 || mode > 2 || !seg_offsets || (m > 0 && !out) ||
      (n > 0 && (!feat || !order || !inv)))
    return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  if (workspace_bytes < fsf_segment_reduce_workspace_bytes(n, m, c)) return FSF_ERR_WORKSPACE;
  FsfArena ar(workspace, workspace_bytes);
  const int64_t nchunks = (n + SEG_CHUNK - 1) / SEG_CHUNK + 1;
  SegArgs a;
  a.feat = feat; a.order = order; a.inv = inv; a.seg_offsets = seg_offsets; a.out = out;
  a.argmax = (mode == MODE_MAX) ? argmax : nullptr;
  a.part_val = ar.take<float>(nchunks * 2 * c);
  a.part_arg = ar.take<int32_t>(nchunks * 2 * c);
  a.n = n; a.m = m; a.c = c;
  a.feat_stride = feat_stride > 0 ? feat_stride : c;
  if (a.feat_stride < c) return FSF_ERR_INVALID_ARG;
  if (!ar.ok()) return FSF_ERR_WORKSPACE;
  const bool vec4 = (c % 4 == 0) && (a.feat_stride % 4 == 0) && (((uintptr_t)feat | (uintptr_t)out) % 16 == 0);
  if (vec4) {
    a.team = pick_team(c, 4);
    return seg_launch<4>(a, mode, stream);
  }
  a.team = pick_team(c, 1);
  return seg_launch<1>(a, mode, stream);
}

extern "C" int fsf_segment_reduce_short(const float* const* feats, const int64_t* feat_strides, const int32_t* channels,
                                        int32_t ntensors, int64_t n, const int32_t* order, const int32_t* seg_offsets, int64_t m,
                                        int32_t mode, float* const* outs, int64_t* argmax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!feats || !feat_strides || !channels || !outs || ntensors < 1 || ntensors > SEG_SHORT_MAX || n < 0 || m < 0 || mode < 0 ||
      mode > 2 || !seg_offsets || (n > 0 && !order) || (argmax && (ntensors != 1 || mode != MODE_MAX)))
    return FSF_ERR_INVALID_ARG;
  if (m == 0) return FSF_OK;
  SegShortArgs a;
  bool vec4 = true;
  for (int t = 0; t < ntensors; ++t) {
    const int64_t st = feat_strides[t] > 0 ? feat_strides[t] : channels[t];
    if (channels[t] < 1 || st < channels[t] || !outs[t] || (n > 0 && !feats[t])) return FSF_ERR_INVALID_ARG;
    a.feat[t] = feats[t]; a.out[t] = outs[t]; a.stride[t] = st; a.c[t] = channels[t];
    vec4 = vec4 && (channels[t] % 4) == 0 && (st % 4) == 0 && (((uintptr_t)feats[t] | (uintptr_t)outs[t]) % 16) == 0;
  }
  a.goff[0] = 0;
  for (int t = 0; t < SEG_SHORT_MAX; ++t) {
    if (t >= ntensors) { a.feat[t] = nullptr; a.out[t] = nullptr; a.stride[t] = 0; a.c[t] = 0; }
    a.goff[t + 1] = a.goff[t] + (t < ntensors ? channels[t] / (vec4 ? 4 : 1) : 0);
  }
  a.nt = ntensors; a.order = order; a.seg_offsets = seg_offsets; a.n = n; a.m = m; a.argmax = argmax;
  const unsigned grid = (unsigned)fsf_stream_grid(m * a.goff[ntensors], 256);
#define FSF_SEG_SHORT(V_, M_) hipLaunchKernelGGL((seg_short_kernel<V_, M_>), dim3(grid), dim3(256), 0, stream, a)
  if (vec4) {
    if (mode == MODE_SUM) FSF_SEG_SHORT(4, MODE_SUM);
    else if (mode == MODE_MEAN) FSF_SEG_SHORT(4, MODE_MEAN);
    else FSF_SEG_SHORT(4, MODE_MAX);
  } else {
    if (mode == MODE_SUM) FSF_SEG_SHORT(1, MODE_SUM);
    else if (mode == MODE_MEAN) FSF_SEG_SHORT(1, MODE_MEAN);
    else FSF_SEG_SHORT(1, MODE_MAX);
  }
#undef FSF_SEG_SHORT
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_segment_reduce_backward(const float* grad_out, int64_t n, int32_t c, const int64_t* inv,
                                           const int32_t* seg_offsets, int64_t m, int32_t mode, const int64_t* argmax,
                                           float* grad_feat, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || m < 0 || c < 1 || mode < 0 || mode > 2 || (n > 0 && (!grad_out || !grad_feat))) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (mode == MODE_MAX) {
    if (!argmax) return FSF_ERR_INVALID_ARG;
    FSF_HIP_TRY(hipMemsetAsync(grad_feat, 0, (size_t)n * c * sizeof(float), stream));
    if (m > 0)
      hipLaunchKernelGGL(seg_backward_max_kernel, dim3(fsf_stream_grid(m * c, 256)), dim3(256), 0, stream, grad_out, m,
                         n, (int)c, argmax, grad_feat);
  } else {
    if (!inv || !seg_offsets) return FSF_ERR_INVALID_ARG;
    if (mode == MODE_MEAN)
      hipLaunchKernelGGL((seg_backward_dense_kernel<MODE_MEAN>), dim3(fsf_stream_grid(n * c, 256)), dim3(256), 0, stream,
                         grad_out, n, (int)c, inv, seg_offsets, grad_feat);
    else
      hipLaunchKernelGGL((seg_backward_dense_kernel<MODE_SUM>), dim3(fsf_stream_grid(n * c, 256)), dim3(256), 0, stream,
                         grad_out, n, (int)c, inv, seg_offsets, grad_feat);
  }
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

static int gather_rows_launch(const float* src, int64_t src_stride, int32_t c, const int64_t* idx, int64_t n, const float* add,
                              int64_t add_stride, float* out, int64_t out_stride, hipStream_t stream) {
  if (n < 0 || c < 1 || (n > 0 && (!src || !idx || !out))) return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  if (out_stride == 0) out_stride = c;
  if (src_stride == 0) src_stride = c;
  if (add_stride == 0) add_stride = c;
  if (out_stride < c || src_stride < c || add_stride < c) return FSF_ERR_INVALID_ARG;
  const bool vec4 = (c % 4 == 0) && (out_stride % 4 == 0) && (src_stride % 4 == 0) && (add_stride % 4 == 0) &&
                    (((uintptr_t)src | (uintptr_t)out | (uintptr_t)add) % 16 == 0);
  if (vec4)
    hipLaunchKernelGGL((gather_rows_kernel<4>), dim3(fsf_stream_grid(n * (c / 4), 256)), dim3(256), 0, stream, src, src_stride, idx,
                       n, (int)c, out, out_stride, add, add_stride);
  else
    hipLaunchKernelGGL((gather_rows_kernel<1>), dim3(fsf_stream_grid(n * c, 256)), dim3(256), 0, stream, src, src_stride, idx, n,
                       (int)c, out, out_stride, add, add_stride);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_gather_rows_strided(const float* src, int64_t src_stride, int64_t m, int32_t c, const int64_t* idx, int64_t n,
                                       float* out, int64_t out_stride, void* stream_) {
  (void)m;
  return gather_rows_launch(src, src_stride, c, idx, n, nullptr, 0, out, out_stride, (hipStream_t)stream_);
}

extern "C" int fsf_gather_rows_add(const float* src, int64_t src_stride, int64_t m, int32_t c, const int64_t* idx, int64_t n,
                                   const float* add, int64_t add_stride, float* out, int64_t out_stride, void* stream_) {
  (void)m;
  if (n > 0 && !add) return FSF_ERR_INVALID_ARG;
  return gather_rows_launch(src, src_stride, c, idx, n, add, add_stride, out, out_stride, (hipStream_t)stream_);
}

extern "C" int fsf_gather_rows(const float* src, int64_t m, int32_t c, const int64_t* idx, int64_t n, float* out,
                               int64_t out_stride, void* stream_) {
  return fsf_gather_rows_strided(src, c, m, c, idx, n, out, out_stride, stream_);
}

extern "C" int fsf_channel_group_sum_add(const float* feat, int64_t n, int32_t cin, int32_t cout, const float* add, float* out,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || cin < 1 || cout < 1 || (n > 0 && (!feat || !out))) return FSF_ERR_INVALID_ARG;
  // r = 2 only (every decoder level of the FSF configs): a two-term sum has one order, so the result is bit-identical to
  // torch's `sum(2)`; wider groups would have to copy ATen's reduction tree to stay so
  if (cin != 2 * cout || cout % 4 != 0) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  const dim3 grid(fsf_stream_grid(n * (cout / 4), 256));
  hipLaunchKernelGGL((channel_group_sum_add_kernel<2>), grid, dim3(256), 0, stream, feat, add, n, (int)cout, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_channel_pair_sum_add2(const float* feat_a, int32_t ca, const float* feat_b, int32_t cb, int64_t n, const float* add,
                                         float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || ca < 1 || cb < 1 || (n > 0 && (!feat_a || !feat_b || !out))) return FSF_ERR_INVALID_ARG;
  if (ca % 8 != 0 || cb % 8 != 0) return FSF_ERR_UNSUPPORTED;
  if (n == 0) return FSF_OK;
  const int cout = (ca + cb) / 2;
  const dim3 grid(fsf_stream_grid(n * (cout / 4), 256));
  hipLaunchKernelGGL(channel_pair_sum_add2_kernel, grid, dim3(256), 0, stream, feat_a, (int)ca, feat_b, (int)cb, add, n, cout, out);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_voxel2point_strided(const float* points, int32_t point_stride, const int64_t* coors_bzyx,
                                       const float* voxel_feats, int64_t m, int32_t c, const int64_t* inv, int64_t n,
                                       const float voxel_size[3], const float range_min[3], float padding, float* out,
                                       int64_t out_stride, uint8_t* valid, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)m;
  if (n < 0 || c < 1 || point_stride < 3 || !voxel_size || !range_min || (out_stride != 0 && out_stride < c + 3) ||
      (n > 0 && (!points || !coors_bzyx || !voxel_feats || !inv || !out)))
    return FSF_ERR_INVALID_ARG;
  if (n == 0) return FSF_OK;
  V2PParams p{voxel_size[0], voxel_size[1], voxel_size[2], range_min[0], range_min[1], range_min[2], padding};
  const int64_t oc = out_stride ? out_stride : (int64_t)c + 3;
  const bool v4 = (c % 4) == 0 && c >= 16 && (oc % 4) == 0 && ((uintptr_t)voxel_feats % 16) == 0 && ((uintptr_t)out % 16) == 0;
  int team = 4;
  while (team < (v4 ? c / 4 : c) && team < 64) team <<= 1;
  const int teams_per_block = 256 / team;
  int64_t g = (n + teams_per_block - 1) / teams_per_block;
  if (g > 16384) g = 16384;
  if (v4)
    hipLaunchKernelGGL(voxel2point_v4_kernel, dim3((unsigned)g), dim3(256), 0, stream, points, (int)point_stride, coors_bzyx,
                       voxel_feats, (int)c, inv, n, p, out, oc, valid, team);
  else
    hipLaunchKernelGGL(voxel2point_kernel, dim3((unsigned)g), dim3(256), 0, stream, points, (int)point_stride, coors_bzyx,
                       voxel_feats, (int)c, inv, n, p, out, oc, valid, team);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}

extern "C" int fsf_voxel2point(const float* points, int32_t point_stride, const int64_t* coors_bzyx,
                               const float* voxel_feats, int64_t m, int32_t c, const int64_t* inv, int64_t n,
                               const float voxel_size[3], const float range_min[3], float padding, float* out,
                               uint8_t* valid, void* stream_) {
  return fsf_voxel2point_strided(points, point_stride, coors_bzyx, voxel_feats, m, c, inv, n, voxel_size, range_min, padding, out, 0,
                                 valid, stream_);
}
