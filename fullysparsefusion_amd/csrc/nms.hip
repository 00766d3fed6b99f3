// K20: greedy BEV non-maximum suppression over rotated (or axis-aligned) boxes.
// Replaces: mmdet3d.ops.iou3d nms_gpu / nms_normal_gpu [UNVENDORED mmdet3d 0.x iou3d_kernel.cu], reached through
//   box3d_multiclass_nms from FrustumClusterHead._get_bboxes_single
//   (projects/mmdet3d_plugin/models/dense_heads/frustum_cluster_head.py:636-667) and
//   SparseClusterHeadV2 (sparse_cluster_head_v2.py:538-609).
// Input boxes are (x1, y1, x2, y2, ry): the axis-aligned extent of the box BEFORE rotation plus the yaw, already
//   sorted by descending score (mmdet3d.core.xywhr2xyxyr of LiDARInstance3DBoxes.bev).  IoU = overlap / max(sa + sb -
//   overlap, 1e-8); box j is suppressed by an earlier kept box i iff IoU(i, j) > thresh.
// Upstream builds a 64x64-blocked suppression bitmask on the GPU, copies it to the host and runs the greedy scan on
//   the CPU.  Here the scan stays on the device (one workgroup, the `removed` bitset in LDS), so a head's NMS is two
//   launches and no host round trip.  The overlap of two rotated rectangles is computed by clipping A's corners
//   (moved into B's frame) against B's four sides (Sutherland-Hodgman) — no intersection-point sort, no degenerate
//   cases — which gives the same area as upstream's intersect-and-sort polygon up to fp32 rounding.
#include "common.h"

namespace fsf {

struct NmsArgs {
  const float* boxes;  // [n,5]
  int64_t n;
  float thresh;
  int rotated;
  uint64_t* mask;    // [n][words]   bit j of word w of row i: box 64 w + j (> i) overlaps box i beyond the threshold
  uint64_t* rowsum;  // [n][sum_words]   bit w set iff mask[i][w] != 0 (lets the scan skip the zero words)
  int words, sum_words;
  int64_t* keep;
  int64_t* num_keep;
  // multi-class form: one scan workgroup per class over its own (mask, rowsum, keep) slice; boxes per class on the device
  const int32_t* count;
  int64_t mask_stride, sum_stride, keep_stride;
  int64_t max_keep;  // > 0: a scan stops once it has kept this many boxes (the caller only uses the best max_keep of a class)
  int64_t window;    // > 0: only the `window` best-scoring boxes of a class are in its mask (rows / bits >= window do not exist)
  int32_t* incomplete;  // set to 1 when a windowed class ran out of boxes before max_keep keeps (the caller reruns without a window)
};

struct NmsBuildArgs {
  const uint64_t* mask0;    // pair bits in the original box order (upper triangle)
  const uint64_t* rowsum0;
  const int32_t* rank;      // [C][n] position of box i in class c's descending-score order, -1 = below the threshold
  uint64_t* mask;           // [C][rows][cwords]      rows = window (or n), cwords = ceil(rows / 64)
  uint64_t* rowsum;         // [C][rows][csum_words]
  int64_t n;
  int words, sum_words;     // of mask0 / rowsum0 (all n boxes)
  int64_t rows;
  int cwords, csum_words;
};

__device__ __forceinline__ float rect_overlap_rotated(const float* a, const float* b) {
  // B frame: origin at B's centre, axes along B's sides
  const float bcx = 0.5f * (b[0] + b[2]), bcy = 0.5f * (b[1] + b[3]);
  const float bhx = 0.5f * (b[2] - b[0]), bhy = 0.5f * (b[3] - b[1]);
  const float acx = 0.5f * (a[0] + a[2]), acy = 0.5f * (a[1] + a[3]);
  const float ahx = 0.5f * (a[2] - a[0]), ahy = 0.5f * (a[3] - a[1]);
  if (!(bhx > 0.f) || !(bhy > 0.f) || !(ahx > 0.f) || !(ahy > 0.f)) return 0.f;
  const float ca = cosf(a[4]), sa = sinf(a[4]), cb = cosf(b[4]), sb = sinf(b[4]);
  // Upstream's corner rotation (iou3d rotate_around_center, the mmdet3d 0.x yaw sense, the same one K17's
  // lidar_to_local_coords implies): corner = centre + M(yaw) * offset with M = [[cos, sin], [-sin, cos]].
  // Polygon = A's corners in the world, then into B's frame with M(yaw_b)^T.
  float px[8], py[8], qx[8], qy[8];
  const float ox[4] = {-ahx, ahx, ahx, -ahx}, oy[4] = {-ahy, -ahy, ahy, ahy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float wx = acx + ca * ox[k] + sa * oy[k] - bcx;
    const float wy = acy - sa * ox[k] + ca * oy[k] - bcy;
    px[k] = cb * wx - sb * wy;
    py[k] = sb * wx + cb * wy;
  }
  int n = 4;
  // clip against x <= bhx, x >= -bhx, y <= bhy, y >= -bhy
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    const float lim = (side < 2) ? bhx : bhy;
    const float sgn = (side & 1) ? -1.f : 1.f;
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const int k2 = (k + 1 == n) ? 0 : k + 1;
      const float c0 = sgn * ((side < 2) ? px[k] : py[k]);
      const float c1 = sgn * ((side < 2) ? px[k2] : py[k2]);
      const bool in0 = c0 <= lim, in1 = c1 <= lim;
      if (in0) {
        qx[m] = px[k];
        qy[m] = py[k];
        ++m;
      }
      if (in0 != in1) {
        const float t = (lim - c0) / (c1 - c0);
        qx[m] = px[k] + t * (px[k2] - px[k]);
        qy[m] = py[k] + t * (py[k2] - py[k]);
        ++m;
      }
    }
    n = m;
    for (int k = 0; k < n; ++k) {
      px[k] = qx[k];
      py[k] = qy[k];
    }
    if (n < 3) return 0.f;
  }
  float area = 0.f;
  for (int k = 1; k + 1 < n; ++k)
    area += (px[k] - px[0]) * (py[k + 1] - py[0]) - (px[k + 1] - px[0]) * (py[k] - py[0]);
  return 0.5f * fabsf(area);
}

__device__ __forceinline__ float rect_overlap_normal(const float* a, const float* b) {
  const float l = fmaxf(a[0], b[0]), r = fminf(a[2], b[2]);
  const float t = fmaxf(a[1], b[1]), d = fminf(a[3], b[3]);
  return fmaxf(r - l, 0.f) * fmaxf(d - t, 0.f);
}

__device__ __forceinline__ float iou_bev(const float* a, const float* b, int rotated) {
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float ov = rotated ? rect_overlap_rotated(a, b) : rect_overlap_normal(a, b);
  return ov / fmaxf(sa + sb - ov, 1e-8f);
}

// block (col word, row block): thread = row box, tests it against the 64 boxes of the column word.  A pair whose
// circumscribed circles do not touch cannot overlap: that test (5 flops) removes nearly every pair of a real scene
// before the polygon clipping.
__global__ void __launch_bounds__(64) nms_mask_kernel(NmsArgs a) {
  const int col_blk = blockIdx.x, row_blk = blockIdx.y;
  if (col_blk < row_blk) return;  // a box is only suppressed by an earlier (higher-score) one
  __shared__ float cb[64 * 5];
  __shared__ float ccx[64], ccy[64], crad[64];
  const int64_t c0 = (int64_t)col_blk * 64;
  const int ncol = (int)min((int64_t)64, a.n - c0);
  for (int t = threadIdx.x; t < ncol * 5; t += 64) cb[t] = a.boxes[c0 * 5 + t];
  __syncthreads();
  if ((int)threadIdx.x < ncol) {
    const float* b = cb + threadIdx.x * 5;
    const float hx = 0.5f * (b[2] - b[0]), hy = 0.5f * (b[3] - b[1]);
    ccx[threadIdx.x] = 0.5f * (b[0] + b[2]);
    ccy[threadIdx.x] = 0.5f * (b[1] + b[3]);
    crad[threadIdx.x] = sqrtf(hx * hx + hy * hy);
  }
  __syncthreads();
  const int64_t i = (int64_t)row_blk * 64 + threadIdx.x;
  if (i >= a.n) return;
  float mine[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) mine[t] = a.boxes[i * 5 + t];
  const float mhx = 0.5f * (mine[2] - mine[0]), mhy = 0.5f * (mine[3] - mine[1]);
  const float mcx = 0.5f * (mine[0] + mine[2]), mcy = 0.5f * (mine[1] + mine[3]);
  const float mrad = sqrtf(mhx * mhx + mhy * mhy);
  uint64_t bits = 0;
  const int start = (row_blk == col_blk) ? threadIdx.x + 1 : 0;
  for (int j = start; j < ncol; ++j) {
    if (a.rotated) {
      const float dx = ccx[j] - mcx, dy = ccy[j] - mcy, rr = (crad[j] + mrad) * 1.0001f;
      if (dx * dx + dy * dy > rr * rr) continue;
    }
    if (iou_bev(mine, cb + j * 5, a.rotated) > a.thresh) bits |= 1ull << j;
  }
  a.mask[i * a.words + col_blk] = bits;
  if (bits) atomicOr((unsigned long long*)&a.rowsum[i * a.sum_words + (col_blk >> 6)], 1ull << (col_blk & 63));
}

// ---- pair bits through a BEV cell grid (n >= NMS_BIN_MIN boxes) ---------------------------------------------------
// Two boxes can only overlap when their circumscribed circles touch, i.e. when their centres are at most r_i + r_j
// apart.  Boxes with r <= NMS_CELL / 2 are binned by centre into NMS_CELL-sized cells: an overlapping pair of them lies in
// the same or in adjacent cells, so each box tests only the boxes of its 3 x 3 cell neighbourhood instead of all n
// (10.6 k boxes over a 100 m scene: 60 x fewer pair tests than the all-pairs mask kernel).  Larger (or non-finite) boxes
// go on a short list and are tested against everything.  The bits are a symmetric relation written with atomicOr, so the
// unordered cell lists do not make the result depend on the run; iou_bev is always called (lower index, higher index)
// as in the all-pairs kernel: the masks are bit-identical.
constexpr int NMS_BIN_MIN = 1024;
constexpr int NMS_GRID = 128;          // cells per axis (+-512 m); coordinates beyond are clamped to the border cells
constexpr float NMS_CELL = 8.0f;       // metres: half of it bounds the radius of a binned box (cars, most trucks)

struct NmsBins {
  int32_t* cell_cnt;    // [G*G]
  int32_t* cell_start;  // [G*G + 1]
  int32_t* box_cell;    // [n]  cell of a small box, -1 = on the big list
  int32_t* box_slot;    // [n]  position inside its cell
  int32_t* cell_list;   // [n]  box ids grouped by cell
  int32_t* big_list;    // [n]
  int32_t* nbig;        // [1]
};

__device__ __forceinline__ int nms_cell_coord(float v) {
  int c = (int)floorf(v * (1.0f / NMS_CELL)) + NMS_GRID / 2;
  return c < 0 ? 0 : (c >= NMS_GRID ? NMS_GRID - 1 : c);
}

__global__ void __launch_bounds__(256) nms_bin_count_kernel(NmsArgs a, NmsBins b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const float* bx = a.boxes + i * 5;
    const float hx = 0.5f * (bx[2] - bx[0]), hy = 0.5f * (bx[3] - bx[1]);
    const float cx = 0.5f * (bx[0] + bx[2]), cy = 0.5f * (bx[1] + bx[3]);
    const float r = sqrtf(hx * hx + hy * hy) * 1.0001f;
    const bool small = r <= 0.5f * NMS_CELL && fabsf(cx) < 1e6f && fabsf(cy) < 1e6f;  // (false for NaN / inf as well)
    if (small) {
      const int cell = nms_cell_coord(cy) * NMS_GRID + nms_cell_coord(cx);
      b.box_cell[i] = cell;
      b.box_slot[i] = atomicAdd(&b.cell_cnt[cell], 1);
    } else {
      b.box_cell[i] = -1;
      b.big_list[atomicAdd(b.nbig, 1)] = (int32_t)i;
    }
  }
}

// exclusive scan of the NMS_GRID^2 cell counts by one workgroup of 1024 threads (64 cells each)
__global__ void __launch_bounds__(1024) nms_bin_scan_kernel(NmsBins b) {
  __shared__ int32_t part[1024];
  constexpr int PER = NMS_GRID * NMS_GRID / 1024;
  const int t = threadIdx.x;
  int32_t c[PER];
  int32_t s = 0;
#pragma unroll
  for (int e = 0; e < PER; e += 4) {
    const int4 v = *reinterpret_cast<const int4*>(b.cell_cnt + t * PER + e);
    c[e] = v.x; c[e + 1] = v.y; c[e + 2] = v.z; c[e + 3] = v.w;
    s += v.x + v.y + v.z + v.w;
  }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int32_t run = part[t] - s;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    b.cell_start[t * PER + e] = run;
    run += c[e];
  }
  if (t == 1023) b.cell_start[NMS_GRID * NMS_GRID] = run;
}

__global__ void __launch_bounds__(256) nms_bin_fill_kernel(NmsArgs a, NmsBins b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const int cell = b.box_cell[i];
    if (cell >= 0) b.cell_list[b.cell_start[cell] + b.box_slot[i]] = (int32_t)i;
  }
}

__device__ __forceinline__ void nms_set_pair(const NmsArgs& a, int64_t lo, int64_t hi) {
  atomicOr((unsigned long long*)&a.mask[lo * a.words + (hi >> 6)], 1ull << (hi & 63));
  atomicOr((unsigned long long*)&a.rowsum[lo * a.sum_words + (hi >> 12)], 1ull << ((hi >> 6) & 63));
}

// one wave per small box: the later boxes of its 3 x 3 cell neighbourhood
__global__ void __launch_bounds__(256) nms_bin_pairs_kernel(NmsArgs a, NmsBins b) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= a.n) return;
  const int cell = b.box_cell[i];
  if (cell < 0) return;
  float mine[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) mine[t] = a.boxes[i * 5 + t];
  const float mhx = 0.5f * (mine[2] - mine[0]), mhy = 0.5f * (mine[3] - mine[1]);
  const float mcx = 0.5f * (mine[0] + mine[2]), mcy = 0.5f * (mine[1] + mine[3]);
  const float mrad = sqrtf(mhx * mhx + mhy * mhy);
  const int cy = cell / NMS_GRID, cx = cell - cy * NMS_GRID;
  for (int dy = -1; dy <= 1; ++dy) {
    const int y = cy + dy;
    if (y < 0 || y >= NMS_GRID) continue;
    const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx < NMS_GRID - 1 ? cx + 1 : NMS_GRID - 1;
    const int s = b.cell_start[y * NMS_GRID + x0], e = b.cell_start[y * NMS_GRID + x1 + 1];  // the row's cells are contiguous
    for (int p = s + lane; p < e; p += 64) {
      const int64_t j = b.cell_list[p];
      if (j <= i) continue;
      float other[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) other[t] = a.boxes[j * 5 + t];
      if (a.rotated) {
        const float ohx = 0.5f * (other[2] - other[0]), ohy = 0.5f * (other[3] - other[1]);
        const float dx = 0.5f * (other[0] + other[2]) - mcx, dyc = 0.5f * (other[1] + other[3]) - mcy;
        const float rr = (sqrtf(ohx * ohx + ohy * ohy) + mrad) * 1.0001f;
        if (dx * dx + dyc * dyc > rr * rr) continue;
      }
      if (iou_bev(mine, other, a.rotated) > a.thresh) nms_set_pair(a, i, j);
    }
  }
}

// one workgroup per big box: every other box (a pair of two big boxes is taken by the one with the lower index)
__global__ void __launch_bounds__(256) nms_big_pairs_kernel(NmsArgs a, NmsBins b) {
  const int nbig = *b.nbig;
  for (int t = blockIdx.x; t < nbig; t += gridDim.x) {
    const int64_t bi = b.big_list[t];
    float big[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) big[q] = a.boxes[bi * 5 + q];
    for (int64_t j = threadIdx.x; j < a.n; j += 256) {
      if (j == bi || (b.box_cell[j] < 0 && j < bi)) continue;
      float other[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) other[q] = a.boxes[j * 5 + q];
      const bool first = bi < j;
      if (iou_bev(first ? big : other, first ? other : big, a.rotated) > a.thresh) nms_set_pair(a, first ? bi : j, first ? j : bi);
    }
  }
}

// OR the mask words named by the set bits of `sbits` (summary word q of row `row`) into the removed bitset, four loads in
// flight at a time (one at a time, every word was a dependent L2 round trip of the scan's critical path)
__device__ __forceinline__ void nms_or_row_words(const uint64_t* __restrict__ row, int q, uint64_t sbits, uint64_t* removed) {
  while (sbits) {
    int w2[4];
    uint64_t v[4];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (sbits) {
        w2[u] = q * 64 + __builtin_ctzll(sbits);
        sbits &= sbits - 1;
        cnt = u + 1;
      } else {
        w2[u] = w2[0];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = row[w2[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < cnt) atomicOr((unsigned long long*)&removed[w2[u]], (unsigned long long)v[u]);
  }
}

// One workgroup: greedy scan in score order, one 64-box word at a time.  Inside a word the dependency chain (a kept box
// removes later boxes of the same word) is resolved by wave 0 on the diagonal mask words held one per lane; the rows
// of the kept boxes are then OR-ed into the `removed` bits of the later words — only the non-zero words, found through
// the per-row summary bits (a real scene has a handful of overlaps per box, so the dense row is never walked).
__global__ void __launch_bounds__(256) nms_scan_kernel(NmsArgs a) {
  extern __shared__ uint64_t removed[];
  __shared__ uint64_t kept_word;
  __shared__ int64_t nkeep;
  if (a.count) {  // class slice; the row strides (a.words, a.sum_words) are those of the full box count
    const int c = blockIdx.x;
    a.n = a.count[c];
    if (a.window > 0 && a.n > a.window) a.n = a.window;
    a.mask += c * a.mask_stride;
    a.rowsum += c * a.sum_stride;
    a.keep += c * a.keep_stride;
    a.num_keep += c;
  }
  const int row_words = a.words, row_sum = a.sum_words;
  a.words = (int)((a.n + 63) / 64);
  a.sum_words = (a.words + 63) / 64;
  for (int w = threadIdx.x; w < a.words; w += 256) removed[w] = 0;
  if (threadIdx.x == 0) nkeep = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // up to 16384 boxes the summary words of a 64-row block are one load per thread and are prefetched a step ahead
  // together with the diagonal words; beyond that they are loaded on demand
  const bool pre = 64 * a.sum_words <= 256;
  const int ub = threadIdx.x & 63, uq = threadIdx.x >> 6;
  uint64_t diag_next = (threadIdx.x < 64 && lane < a.n) ? a.mask[(int64_t)lane * row_words] : 0ull;
  uint64_t sum_next = (pre && uq < a.sum_words && ub < a.n) ? a.rowsum[(int64_t)ub * row_sum + uq] : 0ull;
  for (int w = 0; w < a.words; ++w) {
    const int64_t i0 = (int64_t)w * 64;
    const int nb = (int)min((int64_t)64, a.n - i0);
    const uint64_t sum_cur = sum_next;
    if (pre && w + 1 < a.words && uq < a.sum_words && i0 + 64 + ub < a.n) sum_next = a.rowsum[(i0 + 64 + ub) * row_sum + uq];
    else sum_next = 0ull;
    if (threadIdx.x < 64) {
      const uint64_t diag = diag_next;
      if (w + 1 < a.words && i0 + 64 + lane < a.n) diag_next = a.mask[(i0 + 64 + lane) * row_words + w + 1];  // prefetch
      else diag_next = 0ull;
      const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
      uint64_t alive = ~removed[w];
      if (nb < 64) alive &= (1ull << nb) - 1ull;
      uint64_t kept = 0;
      while (alive) {  // `alive` is the same in every lane.  A pass settles every box up to the first alive one whose
        // row still hits an alive box: the ones before it remove nobody and are kept in bulk (a sparse scene settles a
        // whole word in one pass instead of 64)
        const bool hits = ((alive >> lane) & 1ull) && (diag & alive) != 0ull;
        const uint64_t hb = __ballot(hits);
        if (hb == 0ull) {
          kept |= alive;
          break;
        }
        const int b = __builtin_ctzll(hb);  // (wave-uniform)
        const uint64_t upto = alive & ((2ull << b) - 1ull);
        const uint64_t row = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dlo, b) |
                             ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dhi, b) << 32);
        kept |= upto;
        alive &= ~upto;
        alive &= ~row;
      }
      if (lane == 0) {
        kept_word = kept;
        int64_t k = nkeep;
        const int64_t cap = a.max_keep > 0 ? a.max_keep : a.n;
        for (uint64_t m = kept; m && k < cap; m &= m - 1) a.keep[k++] = i0 + __builtin_ctzll(m);
        nkeep = k;
      }
    }
    __syncthreads();
    if (a.max_keep > 0 && nkeep >= a.max_keep) break;  // (uniform: nkeep was written before the barrier)
    const uint64_t kept = kept_word;
    const int q0 = w >> 6;
    if (pre) {
      if (uq >= q0 && uq < a.sum_words && ((kept >> ub) & 1ull)) {
        uint64_t sbits = sum_cur;
        if (uq == q0) sbits &= ~((2ull << (w & 63)) - 1ull);  // words <= w are already settled
        nms_or_row_words(a.mask + (i0 + ub) * row_words, uq, sbits, removed);
      }
    } else {
      for (int u = threadIdx.x; u < 64 * (a.sum_words - q0); u += 256) {
        const int b = u & 63, q = q0 + (u >> 6);
        if (!((kept >> b) & 1ull)) continue;
        const int64_t i = i0 + b;
        uint64_t sbits = a.rowsum[i * row_sum + q];
        if (q == q0) sbits &= ~((2ull << (w & 63)) - 1ull);
        nms_or_row_words(a.mask + i * row_words, q, sbits, removed);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *a.num_keep = nkeep;
    if (a.incomplete && a.window > 0 && a.count && nkeep < a.max_keep && (int64_t)a.count[blockIdx.x] > a.window) *a.incomplete = 1;
  }
}

// Multi-class: the pair bits were computed once in the original box order; class c's scan needs them in ITS score order.
// thread = (box i, summary word q): walk the few set bits, translate both ends through rank[c], set the bit (lower rank
// = row) in the class's mask and summary.
__global__ void __launch_bounds__(256) nms_build_kernel(NmsBuildArgs a) {
  const int c = blockIdx.y;
  const int32_t* rank = a.rank + (int64_t)c * a.n;
  uint64_t* mask = a.mask + (int64_t)c * a.rows * a.cwords;
  uint64_t* rowsum = a.rowsum + (int64_t)c * a.rows * a.csum_words;
  const int64_t total = a.n * a.sum_words;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / a.sum_words;
    const int q = (int)(t - i * a.sum_words);
    const int ri = rank[i];
    if (ri < 0 || ri >= a.rows) continue;
    for (uint64_t sb = a.rowsum0[t]; sb; sb &= sb - 1) {
      const int w = q * 64 + __builtin_ctzll(sb);
      for (uint64_t bits = a.mask0[i * a.words + w]; bits; bits &= bits - 1) {
        const int64_t j = (int64_t)w * 64 + __builtin_ctzll(bits);
        const int rj = rank[j];
        if (rj < 0 || rj >= a.rows) continue;
        const int lo = min(ri, rj), hi = max(ri, rj);
        atomicOr((unsigned long long*)&mask[(int64_t)lo * a.cwords + (hi >> 6)], 1ull << (hi & 63));
        atomicOr((unsigned long long*)&rowsum[(int64_t)lo * a.csum_words + (hi >> 12)], 1ull << ((hi >> 6) & 63));
      }
    }
  }
}

}  // namespace fsf

using namespace fsf;

static int64_t nms_bins_bytes(int64_t n) {
  if (n < NMS_BIN_MIN) return 0;
  return fsf_align_up((int64_t)(2 * NMS_GRID * NMS_GRID + 1) * 4, 256) + 4 * fsf_align_up(n * 4, 256) + 256;
}

// pair bits of all boxes into a.mask / a.rowsum (rowsum must be zero on entry); `prezeroed`: the caller has cleared a.mask and everything
// this function takes from the arena (one memset over the lot instead of three here)
static int nms_launch_mask(const NmsArgs& a, FsfArena& arena, hipStream_t stream, bool prezeroed = false) {
  if (a.n < NMS_BIN_MIN) {
    // words below the diagonal are never written by the mask kernel and never read by the scan / build (w starts at i / 64)
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)a.words, (unsigned)a.words), dim3(64), 0, stream, a);
    return FSF_OK;
  }
  NmsBins b{};
  int32_t* cells = arena.take<int32_t>(2 * NMS_GRID * NMS_GRID + 1);
  b.cell_cnt = cells;
  b.cell_start = cells + NMS_GRID * NMS_GRID;
  b.box_cell = arena.take<int32_t>(a.n);
  b.box_slot = arena.take<int32_t>(a.n);
  b.cell_list = arena.take<int32_t>(a.n);
  b.big_list = arena.take<int32_t>(a.n);
  b.nbig = arena.take<int32_t>(1);
  if (!arena.ok()) return FSF_ERR_WORKSPACE;
  if (!prezeroed) {
    FSF_HIP_TRY(hipMemsetAsync(a.mask, 0, (size_t)a.n * a.words * 8, stream));
    FSF_HIP_TRY(hipMemsetAsync(b.cell_cnt, 0, sizeof(int32_t) * NMS_GRID * NMS_GRID, stream));
    FSF_HIP_TRY(hipMemsetAsync(b.nbig, 0, sizeof(int32_t), stream));
  }
  const unsigned g = (unsigned)fsf_stream_grid(a.n, 256);
  hipLaunchKernelGGL(nms_bin_count_kernel, dim3(g), dim3(256), 0, stream, a, b);
  hipLaunchKernelGGL(nms_bin_scan_kernel, dim3(1), dim3(1024), 0, stream, b);
  hipLaunchKernelGGL(nms_bin_fill_kernel, dim3(g), dim3(256), 0, stream, a, b);
  hipLaunchKernelGGL(nms_bin_pairs_kernel, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, stream, a, b);
  hipLaunchKernelGGL(nms_big_pairs_kernel, dim3((unsigned)(a.n < 1024 ? a.n : 1024)), dim3(256), 0, stream, a, b);
  return FSF_OK;
}

extern "C" int64_t fsf_nms_bev_workspace_bytes(int64_t n) {
  const int64_t words = (n + 63) / 64, sum_words = (words + 63) / 64;
  const int64_t n1 = n > 0 ? n : 1;
  return fsf_align_up(n1 * (words > 0 ? words : 1) * 8, 256) + fsf_align_up(n1 * (sum_words > 0 ? sum_words : 1) * 8, 256) + 512 +
         nms_bins_bytes(n);
}

extern "C" int fsf_nms_bev(const float* boxes, int64_t n, float thresh, int32_t rotated, int64_t* keep,
                           int64_t* num_keep_dev, int64_t* num_keep_host, void* workspace, int64_t workspace_bytes,
                           void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || (n > 0 && (!boxes || !keep)) || (!num_keep_dev && !num_keep_host)) return FSF_ERR_INVALID_ARG;
  const int64_t words = (n + 63) / 64;
  if (words * 8 > 60 * 1024) return FSF_ERR_UNSUPPORTED;  // `removed` bitset lives in LDS (n <= 491520)
  if (workspace_bytes < fsf_nms_bev_workspace_bytes(n) || !workspace) return FSF_ERR_WORKSPACE;
  FsfArena arena(workspace, workspace_bytes);
  const int64_t sum_words = (words + 63) / 64;
  uint64_t* mask = arena.take<uint64_t>((n > 0 ? n : 1) * (words > 0 ? words : 1));
  uint64_t* rowsum = arena.take<uint64_t>((n > 0 ? n : 1) * (sum_words > 0 ? sum_words : 1));
  int64_t* tmp = arena.take<int64_t>(1);
  if (!arena.ok()) return FSF_ERR_WORKSPACE;
  int64_t* ndev = num_keep_dev ? num_keep_dev : tmp;
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(ndev, 0, sizeof(int64_t), stream));
  } else {
    NmsArgs a{boxes, n, thresh, (int)rotated, mask, rowsum, (int)words, (int)sum_words, keep, ndev, nullptr, 0, 0, 0, 0, 0, nullptr};
    FSF_HIP_TRY(hipMemsetAsync(rowsum, 0, (size_t)n * sum_words * 8, stream));
    const int rc = nms_launch_mask(a, arena, stream);
    if (rc != FSF_OK) return rc;
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), (size_t)words * 8, stream, a);
    FSF_LAUNCH_CHECK();
  }
  if (num_keep_host) {
    FSF_READ_BACK(num_keep_host, ndev, sizeof(int64_t), stream);
  }
  return FSF_OK;
}

static int64_t nms_window(int64_t n, int64_t max_keep, bool windowed) {
  if (!windowed || max_keep <= 0) return n;
  int64_t k = 4 * max_keep > 2048 ? 4 * max_keep : 2048;
  k = (k + 63) / 64 * 64;
  return k < n ? k : n;
}

static int64_t nms_multiclass_bytes(int64_t n, int32_t num_classes, int64_t rows) {
  const int64_t words = (n + 63) / 64, sum_words = (words + 63) / 64;
  const int64_t cw = (rows + 63) / 64, cs = (cw + 63) / 64;
  const int64_t n1 = n > 0 ? n : 1, w1 = words > 0 ? words : 1, s1 = sum_words > 0 ? sum_words : 1;
  const int64_t r1 = rows > 0 ? rows : 1, cw1 = cw > 0 ? cw : 1, cs1 = cs > 0 ? cs : 1;
  const int64_t c1 = num_classes > 0 ? num_classes : 1;
  return fsf_align_up(n1 * w1 * 8, 256) + fsf_align_up(n1 * s1 * 8, 256) + fsf_align_up(c1 * r1 * cw1 * 8, 256) +
         fsf_align_up(c1 * r1 * cs1 * 8, 256) + 512 + nms_bins_bytes(n);
}

extern "C" int64_t fsf_nms_bev_multiclass_workspace_bytes(int64_t n, int32_t num_classes) {
  return nms_multiclass_bytes(n, num_classes, n);
}

// with a cap AND an `incomplete` flag the per-class masks only hold each class's best window = max(4 max_keep, 2048) boxes:
// (1 + C (window / n)^2) n^2 / 8 bytes instead of (1 + C) n^2 / 8
extern "C" int64_t fsf_nms_bev_multiclass_capped_workspace_bytes(int64_t n, int32_t num_classes, int64_t max_keep) {
  return nms_multiclass_bytes(n, num_classes, nms_window(n, max_keep, true));
}

extern "C" int fsf_nms_bev_multiclass_capped(const float* boxes, int64_t n, int32_t num_classes, const int32_t* rank,
                                             const int32_t* count, float thresh, int32_t rotated, int64_t max_keep, int64_t* keep,
                                             int64_t* num_keep, int32_t* incomplete, void* workspace, int64_t workspace_bytes,
                                             void* stream_);

extern "C" int fsf_nms_bev_multiclass(const float* boxes, int64_t n, int32_t num_classes, const int32_t* rank,
                                      const int32_t* count, float thresh, int32_t rotated, int64_t* keep, int64_t* num_keep,
                                      void* workspace, int64_t workspace_bytes, void* stream_) {
  return fsf_nms_bev_multiclass_capped(boxes, n, num_classes, rank, count, thresh, rotated, 0, keep, num_keep, nullptr, workspace,
                                       workspace_bytes, stream_);
}

extern "C" int fsf_nms_bev_multiclass_capped(const float* boxes, int64_t n, int32_t num_classes, const int32_t* rank,
                                             const int32_t* count, float thresh, int32_t rotated, int64_t max_keep, int64_t* keep,
                                             int64_t* num_keep, int32_t* incomplete, void* workspace, int64_t workspace_bytes,
                                             void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || num_classes < 1 || !count || !num_keep || (n > 0 && (!boxes || !rank || !keep))) return FSF_ERR_INVALID_ARG;
  const int64_t words = (n + 63) / 64, sum_words = (words + 63) / 64;
  if (words * 8 > 60 * 1024) return FSF_ERR_UNSUPPORTED;
  const bool windowed = incomplete != nullptr && max_keep > 0;
  const int64_t rows = nms_window(n, max_keep, windowed);
  const int64_t cwords = (rows + 63) / 64, csum = (cwords + 63) / 64;
  if (workspace_bytes < nms_multiclass_bytes(n, num_classes, rows) || !workspace) return FSF_ERR_WORKSPACE;
  if (incomplete) FSF_HIP_TRY(hipMemsetAsync(incomplete, 0, sizeof(int32_t), stream));
  if (n == 0) {
    FSF_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int64_t) * num_classes, stream));
    return FSF_OK;
  }
  FsfArena arena(workspace, workspace_bytes);
  uint64_t* mask0 = arena.take<uint64_t>(n * words);
  uint64_t* rowsum0 = arena.take<uint64_t>(n * sum_words);
  uint64_t* mask = arena.take<uint64_t>((int64_t)num_classes * rows * cwords);
  uint64_t* rowsum = arena.take<uint64_t>((int64_t)num_classes * rows * csum);
  if (!arena.ok()) return FSF_ERR_WORKSPACE;
  // ONE clear for everything that must start at zero: the four arrays above (the build kernel walks mask0 through rowsum0, so the
  // lower-triangle words the mask kernel skips are never read) and, right behind them in the arena, the cell counters / lists
  // nms_launch_mask takes (six memsets before)
  const int64_t clear_bytes = std::min<int64_t>(arena.used + nms_bins_bytes(n), workspace_bytes);
  FSF_HIP_TRY(hipMemsetAsync(mask0, 0, (size_t)clear_bytes, stream));
  NmsArgs a0{boxes, n, thresh, (int)rotated, mask0, rowsum0, (int)words, (int)sum_words, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr};
  const int64_t used_before = arena.used;
  const int rc = nms_launch_mask(a0, arena, stream, true);
  if (rc != FSF_OK) return rc;
  // the single clear above covers what nms_launch_mask took only if it took it right behind the four arrays and no more
  // than nms_bins_bytes(n): a layout change there must fail here, not read uncleared counters
  if (arena.used - used_before > nms_bins_bytes(n) || arena.used > clear_bytes) return FSF_ERR_WORKSPACE;
  NmsBuildArgs b{mask0, rowsum0, rank, mask, rowsum, n, (int)words, (int)sum_words, rows, (int)cwords, (int)csum};
  hipLaunchKernelGGL(nms_build_kernel, dim3((unsigned)fsf_stream_grid(n * sum_words, 256), (unsigned)num_classes), dim3(256), 0,
                     stream, b);
  NmsArgs a{boxes, n, thresh, (int)rotated, mask, rowsum, (int)cwords, (int)csum, keep, num_keep, count,
            rows * cwords, rows * csum, n, max_keep, windowed ? rows : 0, incomplete};
  hipLaunchKernelGGL(nms_scan_kernel, dim3((unsigned)num_classes), dim3(256), (size_t)cwords * 8, stream, a);
  FSF_LAUNCH_CHECK();
  return FSF_OK;
}
