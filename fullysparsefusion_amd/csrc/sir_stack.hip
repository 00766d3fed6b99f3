// K31: a whole SIR stack on rows sorted by group as ONE native call (round 6; VERDICT r5 next-1 (ii): "the SIR stack (K21 + 2 x K22s x 3
// blocks + the group-feature writes) as one call").
//
// SIR.forward (projects/mmdet3d_plugin/models/backbones/sir.py:65-85) and FullySparseBboxHead.forward
// (models/roi_heads/bbox_heads/fsd_bbox_head.py:96-197) run `num_blocks` SIRLayer / DynamicClusterVFE blocks [UNVENDORED; SURVEY
// App. C]; on rows sorted by group (SIR._forward_sorted) a block is K21 (fsf_sir_input_gather: concat + xyz normalisation + position MLP
// + product) and, per DynamicVFELayer, one K22s launch (Linear + LayerNorm + activation + segmented max), preceded from the second layer
// on by the group half of the layer's weight applied to the previous layer's group maxima (fsf_linear_norm_act on [groups, c]).  From
// Python that is 4 C-ABI calls per two-layer block, 13 per stack, 39 per frame, each behind ~25 us of interpreter time (module
// attributes, format checks, six output allocations per block) — on the single-sweep frame, whose kernels here last 5-30 us, the
// interpreter is the bound (profiles/r6_host_gaps_config2_1sweep.txt).  Here the same entry points are called back to back from C++ with
// the arguments the Python path passes: bit-identical results (tests/test_sir_stack_gpu.py).
//
// Memory: ONE arena from the caller (fsf_sir_stack_arena_bytes): the K21 output of the current block, two row buffers the layers
// alternate between, two group tables.  The group features land in the caller's [groups, sum of layer widths] table (pre-filled with
// -inf), the last layer's rows in `rows_out` when the caller wants them.  Nothing is allocated or freed here.
#include <algorithm>

#include "common.h"

static inline int64_t sst_pad4(int64_t c) { return (c + 3) / 4 * 4; }

static int sst_widths(const FsfSirBlock* blocks, int32_t num_blocks, int32_t in_cols0, int32_t e_cols, int64_t* x_cols_max, int64_t* c_max) {
  // block b's K21 output is [n, in_cols_b] with in_cols_0 given and in_cols_b = p_cols + (last width of block b - 1) + e_cols afterwards:
  // the caller states in_cols per block (`in_cols`), checked against the position MLP's width by fsf_sir_input_gather itself
  int64_t xm = 0, cm = 0;
  for (int b = 0; b < num_blocks; ++b) {
    const FsfSirBlock& k = blocks[b];
    if (k.num_layers < 1 || k.num_layers > FSF_SIR_MAX_LAYERS || k.in_cols < 1) return FSF_ERR_INVALID_ARG;
    xm = std::max<int64_t>(xm, sst_pad4(k.in_cols));
    for (int i = 0; i < k.num_layers; ++i) {
      if (k.layer[i].c < 4 || (k.layer[i].c % 4) != 0 || !k.layer[i].planes_left || (i > 0 && !k.layer[i].planes_right)) return FSF_ERR_INVALID_ARG;
      cm = std::max<int64_t>(cm, k.layer[i].c);
    }
  }
  (void)in_cols0; (void)e_cols;
  *x_cols_max = xm;
  *c_max = cm;
  return FSF_OK;
}

extern "C" int64_t fsf_sir_stack_arena_bytes(const FsfSirBlock* blocks, int32_t num_blocks, int64_t n, int64_t num_groups) {
  int64_t xm = 0, cm = 0;
  if (!blocks || num_blocks < 1 || n < 0 || num_groups < 0 || sst_widths(blocks, num_blocks, 0, 0, &xm, &cm) != FSF_OK) return 0;
  const int64_t nn = n > 0 ? n : 1, gg = num_groups > 0 ? num_groups : 1;
  return fsf_align_up(nn * xm * 4, 256) + 2 * fsf_align_up(nn * cm * 4, 256) + 2 * fsf_align_up(gg * cm * 4, 256) + 256;
}

extern "C" int fsf_sir_stack_forward(const FsfSirBlock* blocks, int32_t num_blocks, const float* points, int64_t points_stride, int32_t p_cols,
                                     const float* const* feat_parts, const int64_t* feat_strides, const int32_t* feat_cols,
                                     int32_t num_parts, const int64_t* feats_index, int32_t direct_parts_mask, const float* extra,
                                     int64_t extra_stride, int32_t e_cols, float extra_div, const float* f_cluster,
                                     int64_t f_cluster_stride, int32_t r_cols, const int64_t* seg_ids, int64_t n, int64_t num_groups,
                                     float* groups, int64_t groups_stride, float* rows_out, void* arena, int64_t arena_bytes,
                                     void* stream) {
  if (!blocks || num_blocks < 1 || n < 1 || num_groups < 1 || !points || !f_cluster || !seg_ids || !groups || !arena) return FSF_ERR_INVALID_ARG;
  if (((uintptr_t)arena & 255) || arena_bytes < fsf_sir_stack_arena_bytes(blocks, num_blocks, n, num_groups)) return FSF_ERR_WORKSPACE;
  int64_t xm = 0, cm = 0;
  int rc = sst_widths(blocks, num_blocks, 0, 0, &xm, &cm);
  if (rc != FSF_OK) return rc;
  char* base = (char*)arena;
  float* xbuf = (float*)base;
  base += fsf_align_up(n * xm * 4, 256);
  float* rbuf[2];
  rbuf[0] = (float*)base;
  base += fsf_align_up(n * cm * 4, 256);
  rbuf[1] = (float*)base;
  base += fsf_align_up(n * cm * 4, 256);
  float* tbuf[2];
  tbuf[0] = (float*)base;
  base += fsf_align_up(num_groups * cm * 4, 256);
  tbuf[1] = (float*)base;

  int64_t col = 0;            // first column of the current layer's group maxima in `groups`
  const float* prev_rows = nullptr;  // the previous block's point rows [n, prev_c] (sorted order)
  int32_t prev_c = 0;
  int rsel = 0, tsel = 0;
  for (int b = 0; b < num_blocks; ++b) {
    const FsfSirBlock& k = blocks[b];
    const int64_t cpad = sst_pad4(k.in_cols);
    {  // the block's input width as its position MLP was built for it (hip_ops.sir_input asserts the same)
      int32_t fcols = 0;
      if (b == 0) {
        for (int i = 0; i < num_parts; ++i) fcols += feat_cols ? feat_cols[i] : 0;
      } else {
        fcols = prev_c;
      }
      if (p_cols + fcols + e_cols != k.in_cols) return FSF_ERR_INVALID_ARG;
    }
    // ---- K21: the block's input rows (SIRLayer.forward_sorted -> hip_ops.sir_input)
    if (b == 0) {
      rc = fsf_sir_input_gather(points, points_stride, p_cols, k.xyz_normalizer, feat_parts, feat_strides, feat_cols, num_parts, feats_index,
                                direct_parts_mask, extra, extra_stride, e_cols, extra_div, f_cluster, f_cluster_stride, r_cols, k.rel_div, k.w1,
                                k.g1, k.b1, k.h1, k.w2, k.g2, k.b2, k.h2, k.w3, k.g3, k.b3, k.mlp_eps, k.mlp_act, n, xbuf, cpad, stream);
    } else {
      const float* parts[1] = {prev_rows};
      const int64_t strides[1] = {prev_c};
      const int32_t cols[1] = {prev_c};
      rc = fsf_sir_input_gather(points, points_stride, p_cols, k.xyz_normalizer, parts, strides, cols, 1, nullptr, 0, extra, extra_stride, e_cols,
                                extra_div, f_cluster, f_cluster_stride, r_cols, k.rel_div, k.w1, k.g1, k.b1, k.h1, k.w2, k.g2, k.b2, k.h2, k.w3,
                                k.g3, k.b3, k.mlp_eps, k.mlp_act, n, xbuf, cpad, stream);
    }
    if (rc != FSF_OK) return rc;
    // ---- the block's DynamicVFELayers (sst_ops.sorted_stack_forward)
    const float* x = xbuf;
    int64_t x_stride = cpad;
    int32_t x_cols = k.in_cols;
    for (int i = 0; i < k.num_layers; ++i) {
      const FsfSirLayer& L = k.layer[i];
      const bool last_layer = i == k.num_layers - 1, last_block = b == num_blocks - 1;
      const bool want_rows = !last_layer || !last_block || rows_out != nullptr;
      float* out = !want_rows ? nullptr : (last_layer && last_block ? rows_out : rbuf[rsel]);
      float* seg_out = groups + col;
      const float* table = nullptr;
      if (i > 0) {  // (group W_right^T): the right half of cat([point, group[inv]], 1) W^T, once per group
        const float* g = groups + (col - x_cols);
        float* t = tbuf[tsel];
        tsel ^= 1;
        if (L.right_f16)
          rc = fsf_linear_f16w_norm_act_grouped(g, num_groups, x_cols, groups_stride, L.planes_right, L.c, nullptr, nullptr, nullptr, 0, 0, nullptr,
                                                nullptr, 0.0f, 0, t, L.c, stream);
        else
          rc = fsf_linear_norm_act(g, num_groups, x_cols, groups_stride, L.planes_right, L.c, nullptr, 0, nullptr, nullptr, 0.0f, 0, t, L.c, stream);
        if (rc != FSF_OK) return rc;
        table = t;
      }
      const int64_t so_stride = num_groups > 1 ? groups_stride : sst_pad4(L.c);
      if (L.left_f16)
        rc = fsf_linear_f16w_norm_act_segmax(x, n, x_cols, x_stride, L.planes_left, L.c, L.bias, table, table ? seg_ids : nullptr, table ? L.c : 0,
                                             L.norm, L.gamma, L.beta, L.eps, L.act, seg_ids, num_groups, seg_out, so_stride, out, L.c, stream);
      else
        rc = fsf_linear_norm_act_segmax(x, n, x_cols, x_stride, L.planes_left, L.c, L.bias, table, table ? seg_ids : nullptr, table ? L.c : 0, L.norm,
                                        L.gamma, L.beta, L.eps, L.act, seg_ids, num_groups, seg_out, so_stride, out, L.c, stream);
      if (rc != FSF_OK) return rc;
      col += L.c;
      if (out) {
        x = out;
        x_stride = L.c;
        x_cols = L.c;
        if (out != rows_out) rsel ^= 1;
      }
    }
    prev_rows = x;
    prev_c = x_cols;
    // (the row buffer that holds prev_rows must survive the next block's K21, which reads it: the next block's first layer writes the
    // OTHER buffer — rsel was flipped when prev_rows was written)
  }
  return FSF_OK;
}
