/*
 * fsf_hip.h — C ABI of libfsf_hip.so, the MI355X (gfx950) hot path of FullySparseFusion.
 *
 * Every entry point takes raw DEVICE pointers + sizes + a hipStream_t (passed as void*), writes into
 * caller-owned buffers and returns an int status (0 = FSF_OK, negative = error, see fsf_status_string).
 * No torch types cross this boundary.  Data-dependent output sizes use "capacity in, count out":
 * the caller allocates worst-case capacity, the callee writes the count to a device scalar and, when the
 * `*_host` out-pointer is non-NULL, also synchronises the stream and stores it on the host.
 *
 * Each declaration cites the reference interface it replaces.  Paths are relative to the reference repo
 * (BraveGroup/FullySparseFusion); "[UNVENDORED]" marks a symbol whose native source lives in a dependency
 * that is not in the reference tree (mmdet3d fork / spconv v1 / torch_scatter 2.0.2 / TorchEx), see SURVEY.md §2.2.
 */
#ifndef FSF_HIP_H_
#define FSF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSF_OK 0
#define FSF_ERR_INVALID_ARG (-1)   /* bad pointer / size / mode */
#define FSF_ERR_WORKSPACE (-2)     /* workspace too small */
#define FSF_ERR_KEY_RANGE (-3)     /* unique: packed row key does not fit in 64 bits */
#define FSF_ERR_HIP (-4)           /* a HIP runtime call failed (hipGetLastError has the detail) */
#define FSF_ERR_CAPACITY (-5)      /* data-dependent output exceeded the caller's capacity */
#define FSF_ERR_UNSUPPORTED (-6)   /* shape outside what the kernels are built for */

const char* fsf_status_string(int status);
/* ABI version, bumped whenever a signature changes or an entry point is added; a loader compares fsf_abi_version() of the
 * library it found with the FSF_ABI_VERSION of the header it was written against. */
#define FSF_ABI_VERSION 21
int fsf_abi_version(void);

/* Process-wide algorithm switches (A/B runs and tests that compare two device paths in one process); the defaults are the
 * fast paths.  Stored atomically inside the library: entry points read them instead of the environment, so a call never
 * touches getenv (the library is driven from several host threads).  Returns FSF_ERR_INVALID_ARG for an unknown option.
 *   FSF_OPT_POOL_BRUTE (1): fsf_dynamic_point_pool through the P x R brute-force passes instead of the cell-binned path
 *                           (initial value: the environment variable FSF_POOL_BRUTE at load time, else 0).
 *   FSF_OPT_HOST_WAITS (2), read-only: how many times the library has made the calling host thread wait for a stream since it was
 *                           loaded (every count / flag read-back of every entry point) — what bench.py reports as host waits. */
#define FSF_OPT_POOL_BRUTE 1
#define FSF_OPT_HOST_WAITS 2
int fsf_set_option(int32_t option, int64_t value);
int64_t fsf_get_option(int32_t option);

/* ------------------------------------------------------------------------------------------------
 * K1+K2  dynamic voxelization
 * Replaces: mmdet3d.ops.Voxelization(max_num_points=-1) [UNVENDORED dynamic_voxelize_kernel], constructed at
 *   projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:176, called :218, plus the batch-index pad,
 *   concat and .long() glue of VoteSegmentor.voxelize/extract_feat (:206-226, :231).
 * c = floor((p - range_min) / voxel) evaluated in fp32, subtract-then-divide; x, then y, then z early-out;
 * out-of-range: x OOB writes -1 to slot 0 only, y OOB slots 0,1, z OOB slots 0,1,2 (upstream semantics;
 * the untouched slots are written as 0, which is what the reference's zero-initialised output holds).
 *   points      f32 [n, point_stride] row-major, columns 0..2 = x,y,z
 *   grid        {gx, gy, gz} voxel counts
 *   coors_zyx   i32 [n,3] (z,y,x) or NULL         — what Voxelization returns
 *   coors_bzyx  i64 [n,4] (batch,z,y,x) or NULL   — what extract_feat consumes after pad+long
 */
int fsf_voxelize_dynamic(const float* points, int64_t n, int32_t point_stride, int32_t batch_idx,
                         const float voxel_size[3], const float pc_range[6], const int32_t grid[3],
                         int32_t* coors_zyx, int64_t* coors_bzyx, void* stream);

/* torch.div(p - min, v, rounding_mode='floor') voxel keys (c10::div_floor_floating semantics).
 * Replaces the pure-PyTorch sites single_stage_fsd.py:270 (voxel_downsample), :591-592 (pre_voxelize, zyx)
 * and :948-950 (ClusterAssigner.forward_single_class, xyz).  These sites DISAGREE with fsf_voxelize_dynamic
 * on some boundary values; each keeps its own formula.
 *   order: 0 = (x,y,z) columns, 1 = (z,y,x) columns
 *   batch_idx_in: i64 [n] per-point batch index to prepend, or NULL (then coors has 3 columns)
 *   coors: i64 [n, 3 or 4]
 */
int fsf_voxelize_divfloor(const float* points, int64_t n, int32_t point_stride, const float voxel_size[3],
                          const float range_min[3], int32_t order, const int64_t* batch_idx_in,
                          int64_t* coors, void* stream);

/* DynamicScatterVFE's input decoration [UNVENDORED mmdet3d DynamicVFE.forward: with_cluster_center / with_voxel_center] in one
 * pass: out[i] = [features[i, :p] | xyz - voxel_mean[inv[i], :3] | xyz - (coor * voxel_size + offset)] (x <- coors[:, 3],
 * y <- coors[:, 2], z <- coors[:, 1]); out rows may be padded (out_stride >= width) so that the result feeds
 * fsf_linear_norm_act in place.  features f32 [n, feat_stride >= p], voxel_mean f32 [m, vmean_stride >= 3], inv i64 [n],
 * coors_bzyx i64 [n, 4]. */
int fsf_vfe_decorate(const float* features, int64_t n, int32_t feat_stride, int32_t p, const float* voxel_mean, int32_t vmean_stride,
                     const int64_t* inv, const int64_t* coors_bzyx, const float voxel_size[3], const float offset[3],
                     int32_t with_cluster_center, int32_t with_voxel_center, float* out, int32_t out_stride, void* stream);

/* Vote centres + cluster-voxel keys for every (class group, point) pair of the group-sampled foreground, one pass.
 * Replaces the per-group body of SingleStageFSD.group_sample (single_stage_fsd.py:802-865: arg-max-class weights over the
 * group's classes, ties within 1e-6 split evenly; centre = xyz + sum_c offsets[:, c] * w_c) and the key computation of
 * ClusterAssigner.forward_single_class (:945-950: torch.div(centre - range_min, cluster_voxel_size, rounding_mode='floor')),
 * all groups at once (group id folded into the key's batch column: key = (g * batch_size + b, vx, vy, vz)).
 *   logits f32 [P, logit_stride >= num_classes]; offsets f32 [P, offset_stride >= 3 * num_classes] (class-major xyz triples);
 *   points f32 [P, point_stride >= 3]; batch_idx i64 [P] or NULL; g_ids / p_ids i64 [n] (the pairs);
 *   group_class_mask u32 [num_groups] HOST (bit c: class c belongs to the group; num_classes <= 32, num_groups <= 16);
 *   group_voxel_size f32 [num_groups, 3] HOST; centers f32 [n, 3]; keys i64 [n, 4]; batch_out i64 [n] or NULL. */
int fsf_vote_centers_keys(const float* logits, int32_t logit_stride, const float* offsets, int32_t offset_stride,
                          const float* points, int32_t point_stride, const int64_t* batch_idx, const int64_t* g_ids,
                          const int64_t* p_ids, int64_t n, int32_t num_classes, int32_t num_groups,
                          const uint32_t* group_class_mask, const float* group_voxel_size, const float range_min[3],
                          int32_t batch_size, float* centers, int64_t* keys, int64_t* batch_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  unique rows (+ inverse, counts, CSR segment plan)
 * Replaces: torch.unique(coors, return_inverse=True, return_counts=True, dim=0) at
 *   projects/mmdet3d_plugin/ops/sst_ops.py:156,165; models/backbones/sir.py:68; single_stage_fsd.py:32,595;
 *   roi_heads/bbox_heads/fsd_bbox_head.py:115.
 * new_coors are in ascending lexicographic row order (load-bearing).
 * Beyond what torch.unique returns, the call emits the sort-once segment plan reused by every segmented
 * reduction on the same key (`order` = point indices stably sorted by segment, `seg_offsets` = CSR).
 *   coors        i64 [n,k], k in 1..4
 *   new_coors    i64 [cap>=n rows, k]   (first m rows valid)
 *   inv          i64 [n]
 *   cnt          i64 [cap] or NULL
 *   order        i32 [n]
 *   seg_offsets  i32 [cap+1]            (first m+1 valid)
 *   m_dev        i64 device scalar  (count of unique rows)
 *   m_host       host pointer or NULL; when non-NULL the stream is synchronised
 *   col_min/max  HOST arrays [k] bounding every column (e.g. the voxel grid), or NULL: then the bounds are
 *                reduced on the device and read back (one extra sync).  Rows are packed into one u64 key
 *                (sum of per-column bit widths must be <= 64, else FSF_ERR_KEY_RANGE); a value outside the
 *                given bounds is reported as FSF_ERR_KEY_RANGE at the m_host sync.
 */
int64_t fsf_unique_rows_workspace_bytes(int64_t n, int32_t k);
int fsf_unique_rows(const int64_t* coors, int64_t n, int32_t k, const int64_t* col_min, const int64_t* col_max,
                    int64_t* new_coors, int64_t* inv, int64_t* cnt, int32_t* order, int32_t* seg_offsets,
                    int64_t* m_dev, int64_t* m_host, void* workspace, int64_t workspace_bytes, void* stream);

/* Segment plan from a caller-supplied inverse (values in [0,m)): the `unq_inv=`/`new_coors=` path of
 * scatter_v2 (sst_ops.py:157-158).  Writes order/seg_offsets (and cnt if non-NULL). */
int64_t fsf_segment_plan_workspace_bytes(int64_t n, int64_t m);
int fsf_segment_plan_from_inverse(const int64_t* inv, int64_t n, int64_t m, int32_t* order, int32_t* seg_offsets,
                                  int64_t* cnt, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4/K5  segmented reduce  (mode: 0 = sum, 1 = mean, 2 = max)
 * Replaces: torch_scatter.scatter(feat, inv, dim=0, reduce='sum'|'mean') and torch_scatter.scatter_max
 *   [UNVENDORED torch_scatter 2.0.2] at projects/mmdet3d_plugin/ops/sst_ops.py:168,170.
 * mean = sum / max(count,1) in fp32; max also returns the arg row (first row attaining the max in
 * ascending point index; torch_scatter's choice among ties is atomics-order dependent upstream).
 * Deterministic: no atomics, fixed summation order (ascending point index inside fixed 64-row chunks,
 * then chunks in order).
 *   feat f32 [n,c] with row stride feat_stride floats (0 = c: dense; > c: a column slice of a wider buffer);
 *   order/inv/seg_offsets: the plan from fsf_unique_rows or fsf_segment_plan_from_inverse;
 *   out f32 [m,c]; argmax i64 [m,c] or NULL (mode 2 only)
 */
int64_t fsf_segment_reduce_workspace_bytes(int64_t n, int64_t m, int32_t c);
int fsf_segment_reduce(const float* feat, int64_t feat_stride, int64_t n, int32_t c, const int32_t* order, const int64_t* inv,
                       const int32_t* seg_offsets, int64_t m, int32_t mode, float* out, int64_t* argmax,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Backward of the above w.r.t. feat (scatter_v2 is differentiable w.r.t. feat):
 *   sum : grad_feat[i,:] = grad_out[inv[i],:]
 *   mean: grad_feat[i,:] = grad_out[inv[i],:] / max(cnt[inv[i]],1)
 *   max : grad_feat[argmax[s,ch], ch] = grad_out[s,ch], zero elsewhere (torch_scatter scatter_max backward)
 */
/* The same reduction for plans whose segments are SHORT (voxels: a few rows each): thread per (segment, channel), rows walked
 * in sorted order, no workspace, no fix-up launches, and up to 8 tensors over the same plan in one launch (pre_voxelize,
 * single_stage_fsd.py:585-605, takes the mean of every float field of the point dict over one unique).  Any segment length
 * is handled (long ones serialise).  feats / feat_strides / channels / outs are HOST arrays of ntensors entries (device
 * pointers inside); argmax only with ntensors == 1 and mode max.  A sum's order is the sorted order start to end (the chunked
 * kernel folds chunk partials: the two differ by fp32 rounding on segments that straddle its 32-row chunks). */
int fsf_segment_reduce_short(const float* const* feats, const int64_t* feat_strides, const int32_t* channels, int32_t ntensors,
                             int64_t n, const int32_t* order, const int32_t* seg_offsets, int64_t m, int32_t mode,
                             float* const* outs, int64_t* argmax, void* stream);
int fsf_segment_reduce_backward(const float* grad_out, int64_t n, int32_t c, const int64_t* inv,
                                const int32_t* seg_offsets, int64_t m, int32_t mode, const int64_t* argmax,
                                float* grad_feat, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6  row gather  out[i,:] = src[idx[i],:]   (and its adjoint, a deterministic segmented sum).
 * out_stride (floats, 0 = c) lets the rows land in a column slice of a wider buffer — the
 * `cat([point_feats, voxel_feats[inv]], 1)` of the VFE / SIR layers then needs no concat copy.
 * Replaces: ATen advanced-index `voxel_feats[voxel2point_inds]` at
 *   projects/mmdet3d_plugin/models/necks/voxel2point_neck.py:42 and the "map back" gathers inside
 *   DynamicScatterVFE / SIRLayer [UNVENDORED], FSF.py:311.
 */
int fsf_gather_rows(const float* src, int64_t m, int32_t c, const int64_t* idx, int64_t n, float* out,
                    int64_t out_stride, void* stream);
/* The same with a row stride on the source (src rows may be a column block of a wider buffer, src_stride >= c floats). */
int fsf_gather_rows_strided(const float* src, int64_t src_stride, int64_t m, int32_t c, const int64_t* idx, int64_t n, float* out,
                            int64_t out_stride, void* stream);
/* out[i,:] = add[i,:] + src[idx[i],:].  The per-group half of a Linear over the VFE / SIR layers'
 * cat([point_feats, group_feats[inv]], 1) in TRAINING: cat(p, g[inv]) W^T = p W_left^T + (g W_right^T)[inv], so the [n, 2C]
 * concat and its gather are never written and the adjoint of this op is the identity for `add` plus the deterministic segmented
 * sum (fsf_segment_reduce, mode sum) for `src`. */
int fsf_gather_rows_add(const float* src, int64_t src_stride, int64_t m, int32_t c, const int64_t* idx, int64_t n, const float* add,
                        int64_t add_stride, float* out, int64_t out_stride, void* stream);

/* SimpleSparseUNet decoder shortcut [UNVENDORED; SURVEY App. C decoder_layer_forward]:
 *   `x.features.view(n, C_out, -1).sum(2) + m.features` in one pass: out[i,j] = add[i,j] + sum_q feat[i, j*r + q],
 *   r = cin / cout = 2 (every decoder level of the FSF configs; bit-identical to the two torch ops); add may be NULL;
 *   cout % 4 == 0. */
int fsf_channel_group_sum_add(const float* feat, int64_t n, int32_t cin, int32_t cout, const float* add, float* out,
                              void* stream);
/* The same shortcut read straight from the two tensors the decoder concatenates — `cat([x_bottom.features, x_lateral.features], 1)`
 * of decoder_layer_forward — so that the [n, ca + cb] concatenation is never written when the merge convolution reads its two
 * sources as planes: out[i,j] = add[i,j] + cat[i,2j] + cat[i,2j+1], cout = (ca + cb) / 2; ca, cb multiples of 8 (an output quad's eight
 * input columns then lie in one source; FSF_ERR_UNSUPPORTED otherwise); add may be NULL.  Bit-identical to fsf_channel_group_sum_add
 * on the concatenation. */
int fsf_channel_pair_sum_add2(const float* feat_a, int32_t ca, const float* feat_b, int32_t cb, int64_t n, const float* add, float* out,
                              void* stream);
/* The same sums leaving as PLANES (round 6): what fsf_to_planes makes of fsf_channel_pair_sum_add2's result (planes / scales as there,
 * bit for bit), in one launch and without the [n, cout] fp32 rows — for a decoder level whose upsampling convolution
 * (SimpleSparseUNet.decoder_layer_forward's `upsample_layer(x)`) reads planes only.  ca, cb multiples of 16, rows 16-byte aligned. */
int fsf_channel_pair_sum_add2_planes(const float* feat_a, int32_t ca, const float* feat_b, int32_t cb, int64_t n, const float* add,
                                     void* planes, float* scales, void* stream);


/* Fused a7: Voxel2PointScatterNeck.forward (voxel2point_neck.py:27-70) without the boolean compaction:
 *   out[i, 0:c]   = voxel_feats[inv[i], :]
 *   out[i, c:c+3] = xyz[i] - ((coors[i,[3,2,1]] + 0.5) * voxel + range_min)
 *   valid[i]      = !(all(out[i,0:c] == padding))
 */
int fsf_voxel2point(const float* points, int32_t point_stride, const int64_t* coors_bzyx, const float* voxel_feats,
                    int64_t m, int32_t c, const int64_t* inv, int64_t n, const float voxel_size[3],
                    const float range_min[3], float padding, float* out, uint8_t* valid, void* stream);
/* The same with a row stride on `out` (>= c + 3 floats): a stride that is a multiple of 4 keeps the [n, c + 3] result (131
 * columns in the FSF configs) a legal operand of fsf_linear_norm_act — the segmentation head's first layer reads it in place. */
int fsf_voxel2point_strided(const float* points, int32_t point_stride, const int64_t* coors_bzyx, const float* voxel_feats,
                            int64_t m, int32_t c, const int64_t* inv, int64_t n, const float voxel_size[3],
                            const float range_min[3], float padding, float* out, int64_t out_stride, uint8_t* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K12  fused row normalisation + activation (the tail of every `Linear -> norm -> act` MLP block)
 * Replaces: LayerNorm (or eval-mode BatchNorm1d) followed by GELU / ReLU as separate ATen kernels in the MLPs built
 *   by build_mlp (projects/mmdet3d_plugin/ops/sst_ops.py:808-833) and in DynamicVFELayer [UNVENDORED].
 *   norm: 0 = LayerNorm over the c channels of each row (biased variance, eps inside the sqrt; gamma/beta [c] or
 *             both NULL), 1 = per-channel affine y = x * gamma + beta (eval BatchNorm folded by the caller)
 *   act:  0 = none, 1 = ReLU, 2 = GELU (erf form)
 *   x f32 [n,c] -> out f32 [n,c] with row stride out_stride floats (0 = c; out may alias x when dense); c <= 1024 (fsf_norm_act_backward: c <= 512).
 */
int fsf_norm_act(const float* x, int64_t n, int32_t c, const float* gamma, const float* beta, float eps, int32_t norm,
                 int32_t act, float* out, int64_t out_stride, void* stream);
/* Backward of the LayerNorm form (norm = 0) for training: grad_x f32 [n,c], grad_gamma / grad_beta f32 [c] (NULL to skip);
 * x is the forward INPUT (statistics and pre-activation are recomputed).  Replaces ATen's layer_norm_backward + the
 * activation's own backward kernel behind the MLP blocks of build_mlp (ops/sst_ops.py:808-833). */
int64_t fsf_norm_act_backward_workspace_bytes(int32_t c);
int fsf_norm_act_backward(const float* x, const float* grad_out, int64_t n, int32_t c, const float* gamma, const float* beta,
                          float eps, int32_t act, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K23  column statistics of a [n, c] matrix + training-mode BatchNorm1d (+ ReLU) forward / backward
 * Replaces (training only): ATen's batch_norm_collect_statistics / _backward_reduce / _backward_elemt kernels behind the
 *   `conv -> BN -> ReLU` modules of the sparse U-Net (mmdet3d.ops.make_sparse_convmodule / SparseBasicBlock [UNVENDORED],
 *   norm_cfg naiveSyncBN1d: projects/configs/nuScenes/FSF_nuScenes_config.py:50,63,85) and the bias gradient
 *   `grad.sum(0)` of the per-point Linear layers (build_mlp, projects/mmdet3d_plugin/ops/sst_ops.py:808-833).
 *   fsf_column_stats: var == NULL: mean[c] <- column SUMS of x (a bias gradient); otherwise mean[c] <- column means and
 *     var[c] <- biased variances, two passes (squared deviations about the mean).  Fixed summation order.
 *   fsf_batch_norm_act_forward: out = [relu] fma(x, scale, shift) with scale = gamma * invstd, shift = beta - mean * scale.
 *   fsf_batch_norm_act_backward: batch-statistics backward; with relu != 0 grad_out is first masked by the sign of the
 *     same fma as the forward.  grad_beta[c] = sum g', grad_gamma[c] = sum g' * xhat,
 *     grad_x = scale * (g' - (grad_beta + xhat * grad_gamma) / n), xhat = (x - mean) * invstd.  scale / shift both NULL =
 *     no affine (gamma 1, beta 0).
 *   workspace: fsf_column_stats_workspace_bytes(c) for both.
 */
int64_t fsf_column_stats_workspace_bytes(int32_t c);
int fsf_column_stats(const float* x, int64_t n, int32_t c, float* mean, float* var, void* workspace, int64_t workspace_bytes,
                     void* stream);
/* Training-mode BatchNorm1d statistics of x f32 [n, c] with everything that hangs on them, four launches: column mean, biased
 * variance about it (two passes, fixed fold order), then in the last fold invstd = rsqrt(var + eps), scale = weight * invstd,
 * shift = bias - mean * scale and the running-statistics update running = running * keep + batch * momentum (variance with
 * var_alpha = momentum * n / (n - 1); running_* nullable).  Replaces the statistics + ~10 elementwise ATen kernels per layer of
 * nn.BatchNorm1d / naiveSyncBN1d on one rank [UNVENDORED mmdet3d norm layers; cfg norm_cfg, FSF_nuScenes_config.py:63]. */
int fsf_batch_norm_train_stats(const float* x, int64_t n, int32_t c, const float* weight, const float* bias, float eps, float keep,
                               float momentum, float var_alpha, float* running_mean, float* running_var, float* mean, float* var,
                               float* invstd, float* scale, float* shift, void* workspace, int64_t workspace_bytes, void* stream);
int fsf_batch_norm_act_forward(const float* x, int64_t n, int32_t c, const float* scale, const float* shift, int32_t relu,
                               float* out, void* stream);
int fsf_batch_norm_act_backward(const float* x, const float* grad_out, int64_t n, int32_t c, const float* mean,
                                const float* invstd, const float* scale, const float* shift, int32_t relu, float* grad_x,
                                float* grad_gamma, float* grad_beta, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K22  per-point Linear (+ bias) -> LayerNorm | affine -> ReLU | GELU(erf) in one pass (inference)
 * Replaces: the [nn.Linear, norm, act] blocks of build_mlp (projects/mmdet3d_plugin/ops/sst_ops.py:808-833) and of
 *   DynamicVFELayer [UNVENDORED] applied to every point / cluster row: a library fp32 GEMM plus fsf_norm_act.
 *   x f32 [n, k] (row stride x_stride floats, a multiple of 4; base 16-byte aligned), weight f32 [c, k] in torch Linear
 *   layout, c a multiple of 4 (LayerNorm: c <= 128; wider layers run as 128-channel slices with norm 0 or 2 and the
 *   caller applies fsf_norm_act), bias f32 [c] or NULL; norm 0 none / 1 LayerNorm(gamma, beta, eps) /
 *   2 affine y * gamma + beta (eval BatchNorm1d folded by the caller); act 0 none / 1 ReLU / 2 GELU(erf);
 *   out f32 [n, c] (row stride out_stride, a multiple of 4).
 * fsf_linear_prepare_weight splits the weight ONCE per layer into three bf16 planes (x = hi + mid + lo, an EXACT
 *   split of the fp32 significand) in matrix-core fragment order; fsf_linear_norm_act splits x the same way in
 *   registers and sums the six leading bf16 cross products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the
 *   dropped terms are < 2^-23 of each product, i.e. fp32 accuracy (not a reduced-precision mode), at 2.5x the rate of
 *   the fp32 matrix pipe.  Deterministic.
 */
int64_t fsf_linear_prepared_weight_bytes(int32_t k, int32_t c);
int fsf_linear_prepare_weight(const float* weight, int32_t k, int32_t c, void* planes, void* stream);
int fsf_linear_norm_act(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                        const float* bias, int32_t norm, const float* gamma, const float* beta, float eps, int32_t act,
                        float* out, int64_t out_stride, void* stream);
/* The same with a per-row addend before the norm: out = act(norm(x W^T + bias + row_add[row_add_index[row]])), row_add f32
 * [g, row_add_stride >= c] (16-byte aligned rows), row_add_index i64 [n].  With x = point_feats, W = the left half of a
 * layer's weight and row_add = group_feats W_right^T this is `Linear(cat([point_feats, group_feats[inv]], 1))` of
 * DynamicVFELayer / SIRLayer [UNVENDORED] without the [n, 2C] concat and with the right half applied once per group. */
int fsf_linear_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                                const float* bias, const float* row_add, const int64_t* row_add_index, int64_t row_add_stride,
                                int32_t norm, const float* gamma, const float* beta, float eps, int32_t act, float* out,
                                int64_t out_stride, void* stream);
/* K22s: the same layer PLUS the segmented max that follows it in DynamicVFELayer / SIRLayer [UNVENDORED] —
 * `scatter_v2(point_feats, coors, mode='max')` (ops/sst_ops.py:150-177 over torch_scatter.scatter_max) — in one pass, for rows
 * that arrive SORTED by segment: seg_ids i64 [n] nondecreasing, values in [0, num_segments) (the inverse of the rows' keys, in row
 * order).  seg_out f32 [num_segments, seg_out_stride >= c] receives max over each segment's rows of the activated output and MUST
 * hold -inf on entry: per 16-row group a segmented max-scan across lanes, a 128-row block's open runs merged through LDS, and only
 * a segment that reaches beyond its block is combined by atomic max — max is exact, so the result is bit-identical to
 * fsf_segment_reduce(mode max) over `out`.  out f32 [n, c] or NULL (the rows are then never written: the last layer of a stack
 * whose point features nobody reads).  norm 1 (LayerNorm), act 1 | 2, 32 < c <= 128, num_segments < 2^29,
 * num_segments * seg_out_stride < 2^31; else FSF_ERR_UNSUPPORTED (the caller runs the two kernels).  row_add / row_add_index as
 * fsf_linear_norm_act_grouped, or both NULL. */
int fsf_linear_norm_act_segmax(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* planes, int32_t c,
                               const float* bias, const float* row_add, const int64_t* row_add_index, int64_t row_add_stride,
                               int32_t norm, const float* gamma, const float* beta, float eps, int32_t act,
                               const int64_t* seg_ids, int64_t num_segments, float* seg_out, int64_t seg_out_stride, float* out,
                               int64_t out_stride, void* stream);

/* K22f — the same two operators (fsf_linear_norm_act_grouped, of which fsf_linear_norm_act is the row_add == NULL case, and
 * fsf_linear_norm_act_segmax: same arguments, same results up to the arithmetic below) on THREE matrix passes per product instead of six:
 * `w_planes` comes from fsf_linear_prepare_weight_f16(weight, k, c, slice_c = min(128, c)) — f16 hi | lo of W * s_w behind a 256-byte
 * header — and the kernel splits x the same way per row: a chunk of 32 columns is scaled by a power of two s with s * max|x| in
 * [2^13, 2^14), s following the running maximum of the row (it only falls along the row; the accumulators, kept in the unit s * s_w, are
 * multiplied by new / old — exact — when it does), hi = rn_f16(x s), lo = rn_f16(x s - hi); product = hi hi + hi lo + lo hi on
 * v_mfma_f32_16x16x32_f16, fp32 accumulation.  A row is held to 22 bits relative to ITS largest element (the bf16 form: 24 bits of every
 * element): error against float64 <= that of an fp32 GEMM on the tests' inputs (tests/test_hip_ops.py).  With a per-row addend the
 * row's first scale is capped so that s * s_w <= 2^40 (the addend sits in the accumulators in that unit: |addend| < 2^87).
 * 32 < c (more than two 16-channel tiles); otherwise the restrictions of the bf16 entry points.  FSF_ERR_UNSUPPORTED outside them. */
int fsf_linear_f16w_norm_act_grouped(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* w_planes, int32_t c,
                                     const float* bias, const float* row_add, const int64_t* row_add_index, int64_t row_add_stride,
                                     int32_t norm, const float* gamma, const float* beta, float eps, int32_t act, float* out,
                                     int64_t out_stride, void* stream);
int fsf_linear_f16w_norm_act_segmax(const float* x, int64_t n, int32_t k, int64_t x_stride, const void* w_planes, int32_t c,
                                    const float* bias, const float* row_add, const int64_t* row_add_index, int64_t row_add_stride,
                                    int32_t norm, const float* gamma, const float* beta, float eps, int32_t act, const int64_t* seg_ids,
                                    int64_t num_segments, float* seg_out, int64_t seg_out_stride, float* out, int64_t out_stride,
                                    void* stream);
/* "Sliced": nslice INDEPENDENT layers of slice_c (<= 128, % 4 == 0) output channels each in ONE launch — the per-attribute
 * MLPs of FSDSeparateHead (projects/mmdet3d_plugin/models/dense_heads/sparse_cluster_head_v2.py:18-50: center / dim / rot /
 * vel / score branches, every one `build_mlp(in, [hidden] * num_layer + [out_dim])` on the SAME query features) side by side:
 * out[:, s * slice_c : (s + 1) * slice_c] = act(norm_s(x[:, s * x_slice_offset : + k] W_s^T + bias_s)), weight rows
 * [s * slice_c, (s + 1) * slice_c) = W_s, LayerNorm statistics per slice.  x_slice_offset = 0: every layer reads the same k
 * columns (first layer of the branches); = slice_c of the previous call: layer s reads branch s's activations.  Per slice the
 * arithmetic is that of fsf_linear_norm_act (results identical to nslice separate calls). */
int64_t fsf_linear_prepared_weight_sliced_bytes(int32_t k, int32_t nslice, int32_t slice_c);
int fsf_linear_prepare_weight_sliced(const float* weight, int32_t k, int32_t nslice, int32_t slice_c, void* planes, void* stream);
int fsf_linear_norm_act_sliced(const float* x, int64_t n, int32_t k, int64_t x_stride, int64_t x_slice_offset, const void* planes,
                               int32_t nslice, int32_t slice_c, const float* bias, int32_t norm, const float* gamma,
                               const float* beta, float eps, int32_t act, float* out, int64_t out_stride, void* stream);

/* K22h: the same per-row Linear for the WIDE layers of the query / refine heads — `shared_mlp_dims=[1024, 1024]`, `embed_dims=1024`
 * (projects/configs/nuScenes/FSF_nuScenes_config.py: `mlp_cfg`, the cluster heads; FSF.py:120-164 builds them with build_mlp,
 * ops/sst_ops.py:808-833) and the first layer of FSDSeparateHead's branches (sparse_cluster_head_v2.py:18-50) — with BOTH operands
 * pre-split into f16 hi | lo planes, K9d's arithmetic (fsf_to_planes / fsf_spconv_forward_planes): x s_row = hi + lo with
 * |x s_row - hi - lo| <= max(2^-22 |x s_row|, 2^-25), s_row the power of two that puts the row's largest magnitude in
 * [2^13, 2^14), one power-of-two scale per layer for the weights; x w = hi hi + hi lo + lo hi: THREE v_mfma_f32_16x16x32_f16 per
 * fp32-equivalent product (K22 issues six bf16 ones), fp32 accumulation, nothing split inside the main loop.
 * fsf_rows_to_planes: x f32 [n, c] (row stride x_stride; c % 8 == 0, c <= 2048), optionally through LayerNorm(gamma, beta, eps)
 *   (norm 1) and ReLU / GELU(erf) (act 1 / 2) first — the norm pass that follows a Linear wider than 128 channels — ->
 *   planes [n][c / 8][2][8] f16 (fsf_row_planes_bytes) + inv_scales f32 [n] (1 / s_row) (+ the fp32 rows in `out` unless NULL).
 * fsf_linear_prepare_weight_f16: weight f32 [c, k] -> 256-byte header (1 / s_w, s_w, max |w|) + fragment-ordered planes; slice_c =
 *   output channels per 128-wide tile group: 128 for a plain layer, the per-branch width (65 .. 128) for independent layers stacked
 *   along c (weight rows [s * slice_c, (s + 1) * slice_c) = layer s).
 * fsf_linear_planes_norm_act: out f32 [n, c] = act(norm(x W^T + bias)); k % 32 == 0; a LayerNorm (norm 1) spans ONE slice of
 *   slice_c channels (stacked layers) — a plain layer wider than 128 channels takes norm 0 and the caller runs
 *   fsf_rows_to_planes(norm 1, ...) / fsf_norm_act on the result.  Workgroups of one row block (one per slice) share an XCD.
 *   Deterministic; FSF_ERR_UNSUPPORTED for shapes outside the above (the caller runs fsf_linear_norm_act). */
int64_t fsf_row_planes_bytes(int64_t n, int32_t c);
int fsf_rows_to_planes(const float* x, int64_t n, int32_t c, int64_t x_stride, int32_t norm, const float* gamma, const float* beta,
                       float eps, int32_t act, void* planes, float* inv_scales, float* out, int64_t out_stride, void* stream);
int64_t fsf_linear_prepared_weight_f16_bytes(int32_t k, int32_t c, int32_t slice_c);
int fsf_linear_prepare_weight_f16(const float* weight, int32_t k, int32_t c, int32_t slice_c, void* planes, void* stream);
int fsf_linear_planes_norm_act(const void* x_planes, const float* x_inv_scales, int64_t n, int32_t k, const void* w_planes, int32_t c,
                               int32_t slice_c, const float* bias, int32_t norm, const float* gamma, const float* beta, float eps,
                               int32_t act, float* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K0  multi-sweep point-cloud assembly on the device (input side of the path)
 * Replaces the host passes of the test pipeline (configs/_base_/datasets/nuscenes_dataloader.py:96-137):
 *   LoadPointsFromMultiSweeps (projects/mmdet3d_plugin/datasets/pipelines/loading.py:825-877), SaveNoAugPoints (:341-354),
 *   PointsRangeFilter, NormalizePoints (:537-563) — after ONE host->device copy of the raw sweep files.
 *   raw            f32 [n_rows, load_dim] device: the key frame's rows, then every sweep's, as read from the .bin files
 *   sweep_offsets  i64 [num_sweeps + 1] HOST: row ranges (sweep 0 = the key frame); num_sweeps <= 16
 *   sweep_params   f64 [num_sweeps, 13] HOST: sensor2lidar rotation (row-major 3x3) | translation | time lag (ts - sweep ts)
 *   sweep_transform / sweep_remove_close  u8 [num_sweeps] HOST: apply the transform + time lag / drop |x|,|y| < close_radius
 *   pc_range       f32 [6] HOST or NULL (strict in-range test on the transformed xyz); norm_col < 0: no normalisation
 *   out            f32 [n_rows, load_dim + 3] (capacity): surviving rows in input order, columns = the input's with xyz
 *                  transformed, column 4 = time lag, column norm_col = (v - mean) / std, then the transformed xyz again
 *   count          rows written (device and/or host; the host copy costs one stream sync)
 * Bit-identical to the reference's numpy / torch promotions (float64 rotation rounded to fp32, float64 translation add).
 */
int64_t fsf_assemble_sweeps_workspace_bytes(int64_t n_rows);
int fsf_assemble_sweeps(const float* raw, int64_t n_rows, int32_t load_dim, const int64_t* sweep_offsets, int32_t num_sweeps,
                        const double* sweep_params, const uint8_t* sweep_transform, const uint8_t* sweep_remove_close,
                        float close_radius, const float* pc_range, int32_t norm_col, float norm_mean, float norm_std, float* out,
                        int64_t* count_dev, int64_t* count_host, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K13-K15  LiDAR -> camera projection + per-point instance-mask gather
 * Replaces: FSF.prj_points_2d (projects/mmdet3d_plugin/models/detectors/FSF.py:169-200) and
 *   FSF.points_in_mask (:202-226) for one batch sample; the caller loops samples like frustum_gather (:228-258).
 * Gathers straight from the integer mask (no .float() copy, FSF.py:209); pixel =
 * nearbyint(((g+1)*S-1)/2) (round-half-even, grid_sample nearest, align_corners=False, zero padding).
 *   xyz        f32 [n, xyz_stride] (cols 0..2)
 *   lidar2img  f32 [ncam,4,4] row-major
 *   mask       u8 (elem_bytes=1, nuScenes) or i32 (elem_bytes=4, AV2) [ncam,ncls,H,W]
 *   obj_id     i64 [n,ncam,ncls]
 *   pts_2d     f32 [ncam,n,2] or NULL (the normalised grid coords, -2 for invalid)
 */
int fsf_project_gather_mask(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam,
                            const void* mask, int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w,
                            int64_t* obj_id, float* pts_2d, void* stream);

/* K13-K16 fused  xyz + id planes -> per-point class scores of the argmax camera (FSF.img_cross_attn, FSF.py:694-728:
 * frustum_gather :228-258 + cam-select :716-718 + get_all_cls_preds_2d :506-535 + encode_preds_2d :472-473) without the
 * [n, ncam, ncls] int64 tensor in between; same projection / pixel / tie rules as K13-K16 (results identical to
 * fsf_project_gather_mask followed by fsf_cam_select_score).  ncls <= 16.
 *   out_score f32 [n, ncls]; out_ids i64 [n, ncls] or NULL (ids of the selected camera);
 *   out_fg u8 [n] or NULL: 1 if the point is inside any mask of any camera (obj_id.sum((-2,-1)) > 0, FSF.py:299-308)
 *   out_count u8 [n] or NULL: (camera, class) cells with an id > 0 (`(obj_id_tensor > 0).sum(-1)`, FSF.py:262), saturated at 255
 *   out_max_id i32 [n] or NULL: the largest id over all cells (`obj_id_tensor.max(-1)[0]`, FSF.py:263)
 */
int fsf_project_score(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam, const void* mask,
                      int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w, const float* mask_anno, int32_t num_anno,
                      int32_t anno_dim, int32_t score_col, float* out_score, int64_t* out_ids, uint8_t* out_fg, uint8_t* out_count,
                      int32_t* out_max_id, void* stream);

/* K26  the camera-query branch's row list: extract_fg_pts (FSF.py:299-308) + double_overlap_pts (:260-297) + get_sir_coors (:373-376)
 * for ONE sample, from fsf_project_score's per-point outputs, in two calls and one 32-byte read-back.
 *   fsf_overlap_plan: fg u8 [n], count u8 [n] (max_cells = ncam * ncls <= 254, else FSF_ERR_UNSUPPORTED: the count must not saturate)
 *     -> counts_host i64 [3] = (F foreground points, M of them inside >= 2 masks, T appended rows = sum (k - 1)); the index lists stay in
 *     `workspace` (fsf_overlap_plan_workspace_bytes(n)), which the caller hands UNCHANGED to
 *   fsf_overlap_rows: -> src_pt i64 [F + T] (the point a row is a copy of), sir_coors i64 [F + T, 3] = (batch_idx[pt] or 0, 0, id).
 *     Rows 0 .. F-1: the foreground points ascending, id = the point's largest id.  Then, exactly in the order the reference's loop over
 *     `overlap_num = 2, 3, ...` appends them (k ascending; within k, j = 1 .. k-1; within (k, j) the points ascending): the row of point p
 *     with k cells carries its j-th largest id (0-based; `topk(k)[0][:, j]`, duplicates kept).  xyz / lidar2img / mask as fsf_project_score.
 */
int64_t fsf_overlap_plan_workspace_bytes(int64_t n);
int fsf_overlap_plan(const uint8_t* fg, const uint8_t* count, int64_t n, int32_t max_cells, int64_t* counts_host, void* workspace,
                     int64_t workspace_bytes, void* stream);
int fsf_overlap_rows(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam, const void* mask,
                     int32_t elem_bytes, int32_t ncls, int32_t img_h, int32_t img_w, const int32_t* max_id, const int64_t* batch_idx,
                     const void* workspace, int64_t workspace_bytes, int64_t num_fg, int64_t num_multi, int64_t num_extra,
                     int64_t* src_pt, int64_t* sir_coors, void* stream);

/* K13b  LiDAR -> camera projection + per-point BILINEAR image-feature gather (BASELINE.json north_star; the reference only
 * gathers instance ids, FSF.py:216-225 — this is the feature-map counterpart of the same call site: prj_points_2d
 * (FSF.py:169-200) followed by F.grid_sample(feat, grid, mode='bilinear', align_corners=False, padding_mode='zeros')).
 *   feat   f32 [ncam, channels, feat_h, feat_w] (channels_last = 0) or [ncam, feat_h, feat_w, channels] (channels_last = 1)
 *   img_h, img_w: the image size the projection normalises by (the feature map may be a strided version of it)
 *   out    f32 [n, ncam, channels] (reduce_cams = 0; zeros where the point is not inside the camera's image) or
 *          f32 [n, channels] = the sum over the cameras that see the point (reduce_cams = 1)
 *   count  u8 [n] or NULL: number of cameras that see the point
 */
int fsf_project_gather_bilinear(const float* xyz, int64_t n, int32_t xyz_stride, const float* lidar2img, int32_t ncam,
                                const float* feat, int32_t channels, int32_t feat_h, int32_t feat_w, int32_t channels_last,
                                int32_t img_h, int32_t img_w, int32_t reduce_cams, float* out, uint8_t* count, void* stream);

/* K16  camera select + 2-D prediction lookup for the per-point branch (nuScenes: score column only).
 * Replaces: FSF.img_cross_attn cam-select (FSF.py:716-718) + get_all_cls_preds_2d (:506-535) +
 *   encode_preds_2d(encode_single_cls=False) (:449-474) for one batch sample:
 *   cam* = argmax_c sum_k id[i,c,k] (first max); ids = id[i,cam*,:]; score[i,k] = ids>0 ? anno[ids-1, col] : 0
 *   obj_id i64 [n,ncam,ncls]; mask_anno f32 [num_anno, anno_dim]; out_ids i64 [n,ncls] or NULL;
 *   out_score f32 [n,ncls]
 */
int fsf_cam_select_score(const int64_t* obj_id, int64_t n, int32_t ncam, int32_t ncls, const float* mask_anno,
                         int32_t num_anno, int32_t anno_dim, int32_t score_col, int64_t* out_ids, float* out_score,
                         void* stream);

/* The k largest values of every row, descending: `obj_id_tensor[overlaps_mask].topk(overlap_num, dim=-1)[0]` of
 * FSF.double_overlap_pts (models/detectors/FSF.py:284-286).  x i64 [n, w] (w <= 128), out i64 [n, k]. */
int fsf_row_topk_desc(const int64_t* x, int64_t n, int32_t w, int32_t k, int64_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K7/K8  sparse-conv rulebooks (hash table instead of spconv v1's dense grid)
 * Replaces: mmdet3d.ops.spconv get_indice_pairs [UNVENDORED spconv v1 indice_cuda.cu] used inside
 *   SimpleSparseUNet [UNVENDORED], configured at projects/configs/nuScenes/FSF_nuScenes_config.py:58-70.
 * Rulebook layout is OUTPUT-MAJOR: nbr[o*kvol + k] = input row feeding output row o through kernel offset
 * k = (kz*KY + ky)*KX + kx, or -1.  fsf_rulebook_to_pairs converts to spconv v1's indicePairs[kvol,2,cap] /
 * indiceNum[kvol] (pairs in ascending output row — a canonical order; upstream order is atomics-dependent).
 *   indices i32 [m,4] (b,z,y,x), spatial_shape {Z,Y,X}
 */
int64_t fsf_rulebook_workspace_bytes(int64_t m_in, int32_t kvol);
int fsf_rulebook_subm(const int32_t* indices, int64_t m, int32_t batch_size, const int32_t spatial_shape[3],
                      const int32_t ksize[3], const int32_t dilation[3], int32_t* nbr, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* Strided SparseConv3d: out coords are the ascending-linear-index unique set of
 * (in + pad - k*dil)/stride (divisible, in-range); out_shape = (in + 2p - d(k-1) - 1)/s + 1.
 *   out_indices i32 [cap,4]; nbr i32 [cap,kvol]; m_out_dev/m_out_host as in fsf_unique_rows.
 * Also emits the transposed table for SparseInverseConv3d on the same indice_key:
 *   nbr_inv i32 [m, kvol] (fine row -> coarse row per offset) or NULL.
 */
int fsf_rulebook_strided(const int32_t* indices, int64_t m, int32_t batch_size, const int32_t spatial_shape[3],
                         const int32_t ksize[3], const int32_t stride[3], const int32_t padding[3],
                         const int32_t dilation[3], int32_t* out_indices, int64_t cap, int32_t* nbr,
                         int32_t* nbr_inv, int64_t* m_out_dev, int64_t* m_out_host, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* Row order of a level INSIDE the network (an optimisation, never visible at the reference's interfaces: the rulebooks above keep
 * spconv v1's orders bit for bit).  The plane kernels walk a (64-row block, kernel offset) pair only if one of the block's rows has
 * a neighbour at that offset; in the reference's lexicographic voxel order a block of the 0.2 m level touches 16.8 of 27 offsets with
 * its cells 39 % full, with rows of equal 3x3x3 neighbour mask adjacent 8.1 offsets and 80 % (0.4 m level: 24.8 -> 17.1, 61 -> 93 %).
 * fsf_order_by_neighbor_mask: from the level's coordinates (indices i32 [m, 4] b,z,y,x) — a hash of the sites, 27 probes per row —
 * perm[i] = the row that goes to position i under a 19-bit key (three radix passes): the 9 in-plane neighbours of the row exactly, the
 * planes below / above by their neighbour counts (clipped to 7 / 15), descending — then, inside one such neighbourhood key, the parity
 * of (z, y, x) ascending (a stride-2 inverse convolution reaches a fine row only through the kernel offsets its parity admits);
 * ties in ascending row order (stable: deterministic); inv_perm[perm[i]] = i.  The full 27-bit mask groups no better on LiDAR
 * occupancy (see rulebook.hip) and costs a fourth pass; parity as the MOST significant digit was measured: the inverse
 * convolutions gain 80 us, the submanifold layers lose 190.
 * fsf_remap_indices: out[j] = in[j] >= 0 ? map[in[j]] : -1 (a table whose VALUES are rows of the reordered level).
 * SimpleSparseUNet [UNVENDORED] applies the order to its fine levels at inference and restores the input order on the way out. */
int64_t fsf_order_by_neighbor_mask_workspace_bytes(int64_t m);
int fsf_order_by_neighbor_mask(const int32_t* indices, int64_t m, int32_t batch_size, const int32_t spatial_shape[3], int32_t* perm,
                               int32_t* inv_perm, void* workspace, int64_t workspace_bytes, void* stream);
int fsf_remap_indices(const int32_t* in, int64_t n, const int32_t* map, int32_t* out, void* stream);
int64_t fsf_rulebook_to_pairs_workspace_bytes(int64_t m_out, int32_t kvol);
int fsf_rulebook_to_pairs(const int32_t* nbr, int64_t m_out, int32_t kvol, int32_t* indice_pairs, int64_t cap,
                          int32_t* indice_num, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9/K11  sparse convolution forward: out[o,:] = act(scale * (sum_k feat[nbr[o,k],:] @ W[k]) + shift + residual)
 * Replaces: spconv v1 indice_conv (per-offset gather -> cuBLAS mm -> scatter-add, 27x3 launches per layer)
 *   [UNVENDORED] for SubMConv3d / SparseConv3d / SparseInverseConv3d inside SimpleSparseUNet; the eval-mode
 *   naiveSyncBN1d affine and ReLU that follow every conv (cfg :63 order conv-norm-act) are the fused epilogue.
 *   feat f32 [m_in,cin] (cin % 16 == 0); weight_t f32 [kvol,cout,cin] = the spconv v1 weight
 *   [kz,ky,kx,Cin,Cout] transposed once per layer by fsf_spconv_transpose_weight; scale/shift f32 [cout] or
 *   NULL (shift alone = bias); residual f32 [m_out,cout] or NULL (added before the ReLU); relu 0/1.
 * One launch per layer: persistent workgroups (2 per CU) pull (tile, cout block, offset split) work items from
 * per-XCD queues; a layer that splits its offset loop keeps the partial tiles in the workspace and the workgroup
 * that finishes a tile last folds them in split order (the workspace also holds the queue words and is always
 * required: fsf_spconv_workspace_bytes).  fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32 fma accumulation in
 * a fixed order, deterministic.
 */
int fsf_spconv_transpose_weight(const float* weight, int32_t kvol, int32_t cin, int32_t cout, float* weight_t,
                                void* stream);
int64_t fsf_spconv_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout, int32_t kvol);
int fsf_spconv_forward(const float* feat, int64_t m_in, int32_t cin, const float* weight_t, int32_t kvol,
                       int32_t cout, const int32_t* nbr, int64_t m_out, const float* scale, const float* shift,
                       const float* residual, int32_t relu, float* out, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* K9b  the same convolution, row-stationary on the bf16 matrix cores (inference).  The weight [kvol,cin,cout] is
 * split ONCE per layer (fsf_spconv_prepare_weight_split) into three bf16 planes — an exact split of the fp32
 * significand — in matrix-core fragment order; the kernel splits the gathered feature rows the same way in registers
 * and sums the six leading cross products with fp32 accumulation: fp32 accuracy at 2.5x the fp32 pipe's rate (see
 * K22).  A wave owns 32 output rows for the whole offset loop (accumulators in registers, no compaction, no LDS tile):
 * the better choice for submanifold layers (>= ~9 neighbours per output row); fsf_spconv_forward stays the choice for the
 * sparse neighbourhoods of strided / inverse convolutions.  A layer with too few 128-row workgroups splits the (offset,
 * cin chunk) sequence over more workgroups and folds the partial sums (workspace) in a fixed order.  cin % 4 == 0,
 * cout % 4 == 0.  Deterministic. */
int64_t fsf_spconv_split_weight_bytes(int32_t kvol, int32_t cin, int32_t cout);
int fsf_spconv_prepare_weight_split(const float* weight, int32_t kvol, int32_t cin, int32_t cout, void* planes, void* stream);
int64_t fsf_spconv_split_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout, int32_t kvol);
int fsf_spconv_forward_split(const float* feat, int64_t m_in, int32_t cin, const void* planes, int32_t kvol, int32_t cout,
                             const int32_t* nbr, int64_t m_out, const float* scale, const float* shift,
                             const float* residual, int32_t relu, float* out, void* workspace, int64_t workspace_bytes,
                             void* stream);
/* K9b-XP (round 5): the same convolution with BOTH operands as f16 hi | lo planes (three v_mfma_f32_16x16x32_f16 per fp32-equivalent
 * product instead of six bf16 ones, nothing split inside the kernel) for the layers K9c / K9d do not take — the deep U-Net levels of
 * SimpleSparseUNet [UNVENDORED] (512 / 1024 input channels on a few thousand rows).  feat_planes / feat_inv_scales = the input rows
 * through fsf_rows_to_planes (one power-of-two scale per ROW; the rows a lane gathers change scale from offset to offset, so the
 * accumulators are kept in the unit of the row being multiplied and moved by exact power-of-two ratios); planes from
 * fsf_spconv_prepare_weight_split_f16 (256-byte header: 1 / s_w, s_w, max |w|).  cin % 32 == 0, cout > 64; workspace as
 * fsf_spconv_split_workspace_bytes.  Same epilogue, same k-split + fold, same XCD-aware layout as fsf_spconv_forward_split. */
int64_t fsf_spconv_split_weight_f16_bytes(int32_t kvol, int32_t cin, int32_t cout);
int fsf_spconv_prepare_weight_split_f16(const float* weight, int32_t kvol, int32_t cin, int32_t cout, void* planes, void* stream);
int fsf_spconv_forward_split_planes(const void* feat_planes, const float* feat_inv_scales, int64_t m_in, int32_t cin,
                                    const void* planes, int32_t kvol, int32_t cout, const int32_t* nbr, int64_t m_out,
                                    const float* scale, const float* shift, const float* residual, int32_t relu, float* out,
                                    void* workspace, int64_t workspace_bytes, void* stream);

/* K9c  the submanifold convolution on the f16 matrix cores from PRE-SPLIT FEATURE PLANES (inference; replaces the
 * gather -> GEMM -> scatter-add loop of spconv v1 `indice_conv`, [UNVENDORED] mmdet3d fork, for the SubMConv3d layers
 * SimpleSparseUNet runs on its fine levels: FSF_nuScenes_config.py:58-70).
 *
 * Plane form of a feature tensor f32 [m, c] (c % 8 == 0): planes = [m + 1][c / 8][2][8] f16 (fsf_planes_bytes(m, c) bytes,
 * 16-byte aligned) holding hi = rn_f16(x * s_row) and lo = rn_f16(x * s_row - hi) per 8-channel block, s_row the power of two
 * that puts the largest magnitude of the row's 128-channel chunk into [2^13, 2^14); scales = f32 [m + 1][ceil(c / 128)], the
 * INVERSE scale (x ~ (hi + lo) * scales).  Row m is all zeros (scale 1): the neighbour of a missing pair.  fsf_to_planes
 * converts an fp32 tensor (rows may be strided); fsf_spconv_forward_planes can emit its own output in plane form
 * (out_planes / out_scales non-NULL) so that a chain of convolutions splits every value exactly once.
 *
 * fsf_spconv_forward_planes: up to two sources (the channel concatenation [a | b], each 32..128 channels, a multiple of
 * 32) -> out f32 [m_out, cout] (cout == 64 or cout % 128 == 0), epilogue as K9: out = act(scale * conv + shift + residual).
 * Weights come from fsf_spconv_prepare_weight_planes (weight [kvol, ca + cb, cout] -> per-wave f16 fragment planes with one
 * power-of-two scale per layer; fsf_spconv_planes_weight_bytes bytes).  Product = hi hi + hi lo + lo hi on
 * v_mfma_f32_16x16x32_f16 with fp32 accumulation: relative error <= ~3 * 2^-22 per product, of the order of fp32's own
 * accumulation error over the 27 * cin terms.  (16-row group, offset) cells without a neighbour are skipped.  No workspace,
 * no atomics, fixed summation order: deterministic. */
int64_t fsf_planes_bytes(int64_t m, int32_t c);
int64_t fsf_planes_scale_count(int64_t m, int32_t c);
int fsf_to_planes(const float* feat, int64_t m, int32_t c, int64_t row_stride, void* planes, float* scales, void* stream);
/* The same through a row index (round 6): plane row r = feat row row_index[r] (i64 [m]) — SimpleSparseUNet's neighbour-mask row order
 * (`voxel_features.index_select(0, perm)` in front of conv_input) applied while converting, the permuted fp32 rows never written. */
int fsf_to_planes_rows(const float* feat, int64_t m, int32_t c, int64_t row_stride, const int64_t* row_index, void* planes, float* scales,
                       void* stream);
int64_t fsf_spconv_planes_weight_bytes(int32_t kvol, int32_t cin, int32_t cout);
int fsf_spconv_prepare_weight_planes(const float* weight, int32_t kvol, int32_t cin, int32_t cout, void* planes, void* stream);
int fsf_spconv_forward_planes(const void* planes_a, const float* scales_a, int32_t ca, const void* planes_b,
                              const float* scales_b, int32_t cb, int64_t m_in, const void* weight_planes, int32_t kvol,
                              int32_t cout, const int32_t* nbr, int64_t m_out, const float* scale, const float* shift,
                              const float* residual, int32_t relu, float* out, void* out_planes, float* out_scales,
                              void* stream);

/* ------------------------------------------------------------------------------------------------
 * K10  sparse convolution backward (training)
 * Replaces: spconv v1 indice_conv_backward [UNVENDORED mmdet3d.ops.spconv] = per offset
 *   (gather in, gather grad, mm^T for dW, mm for dX, scatter-add), 27 x 5 launches per layer.
 * Data gradient: no entry point of its own — it IS a forward conv over the transposed table with the
 *   un-transposed weight:  fsf_spconv_forward(grad_out, m_out, cout, weight /[kvol,cin,cout]/, kvol, cin,
 *   nbr_inv /[m_in,kvol]/, m_in, NULL, NULL, NULL, 0, grad_feat, ...).  SubM layers have
 *   nbr_inv[i,k] = nbr[i,kvol-1-k], i.e. pass nbr and the weight flipped along k.
 * Weight gradient: grad_weight[k,ci,co] = sum_p feat[in_k[p],ci] * grad_out[out_k[p],co] over the pair lists
 *   of fsf_rulebook_to_pairs (indice_pairs i32 [kvol,2,cap], indice_num i32 [kvol], both on the device).
 *   cin % 4 == 0, cout % 4 == 0.  fp32 MFMA, partial sums folded in a fixed order (deterministic).
 *   indice_pairs = indice_num = NULL with kvol = 1: the identity pairing of the first `cap` rows, i.e. X^T dY — the weight
 *   gradient of a per-point Linear layer (build_mlp, ops/sst_ops.py:808-833), a [<=256 x 128] result reduced over 5e5
 *   rows, the shape GEMM libraries serve worst.
 */
int64_t fsf_spconv_backward_weight_workspace_bytes(int64_t cap, int32_t cin, int32_t cout, int32_t kvol);
int fsf_spconv_backward_weight(const float* feat, int64_t m_in, int32_t cin, const float* grad_out, int64_t m_out,
                               int32_t cout, const int32_t* indice_pairs, const int32_t* indice_num, int64_t cap,
                               int32_t kvol, float* grad_weight, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * K19  connected components of the graph "xy-distance < dist" (same batch index), labels contiguous 0..K-1
 * numbered in order of each component's smallest member index — the labelling
 * scipy.sparse.csgraph.connected_components(adj, directed=False)[1] produces.
 * Replaces: find_connected_componets_single_batch / find_connected_componets
 *   (projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:45-82): dense n x n distance matrix on the
 *   GPU, .cpu().numpy(), scipy, back to the device — six times per frame — and the optional
 *   torchex.connected_components path (:37-43).
 *   points f32 [n, point_stride] (cols 0,1 = x,y); batch_idx i32 [n] or NULL (single batch);
 *   labels i32 [n]; num_components_dev i64 device scalar or NULL.
 */
int64_t fsf_connected_components_workspace_bytes(int64_t n);
int fsf_connected_components(const float* points, int64_t n, int32_t point_stride, const int32_t* batch_idx,
                             float dist, int32_t* labels, int64_t* num_components_dev, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* All class groups of ClusterAssigner.forward (single_stage_fsd.py:912-934 loops over the classes, each with its own
 * `connected_dist`) in one call: group_idx i32 [n] (points of different groups are never adjacent), dist_table f32
 * [num_groups] (device).  Labels are numbered by first member over ALL points; with group-sorted input the labels of
 * a group form a contiguous range starting at the label of its first point. */
int fsf_connected_components_grouped(const float* points, int64_t n, int32_t point_stride, const int32_t* group_idx,
                                     const float* dist_table, int32_t num_groups, int32_t* labels,
                                     int64_t* num_components_dev, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K21  SIR-layer input: out = cat(points / xyz_normalizer (first 3 cols), feats, extra / extra_div) * rel_mlp(f_cluster / rel_div)
 *   (true fp32 divisions, as the reference's `/`)
 * Replaces (inference): `in_feats = torch.cat([points, out_feats], 1)` of SIR.forward
 *   (projects/mmdet3d_plugin/models/backbones/sir.py:72-74) / FullySparseBboxHead.forward
 *   (models/roi_heads/bbox_heads/fsd_bbox_head.py:127-132), and inside SIRLayer / DynamicClusterVFE [UNVENDORED] the
 *   xyz normalisation, the three Linear(no bias)->LayerNorm->act blocks of `rel_mlp` (build_mlp, ops/sst_ops.py:808-833)
 *   and the `features * rel` product.
 *   points f32 [n, >=p_cols] (row stride in floats), feats f32 [n, f_cols], extra f32 [n, e_cols] or NULL,
 *   f_cluster f32 [n, r_cols]; w1 [h1, r_cols], w2 [h2, h1], w3 [c, h2] in torch Linear layout, g/b = LayerNorm weight /
 *   bias of each block, one eps; act 0 none / 1 ReLU / 2 GELU(erf); out f32 [n, c], c = p_cols + f_cols + e_cols <= 256;
 *   r_cols <= 16, h1 <= 16, h2 <= 32 (the FSF configs: 3|13 -> 16 -> 32 -> C).
 * The three layers run on the fp32 matrix cores in transposed form (weights = A operand, 16 point rows = B operand),
 *   chained through registers; GELU's erf is Abramowitz-Stegun 7.1.26 (absolute error <= 1.5e-7, inside the 1e-4
 *   fp32 budget of the path; exact-erf GELU elsewhere).  Deterministic.
 */
int fsf_sir_input(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                  const float* feats, int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride,
                  int32_t e_cols, float extra_div, const float* f_cluster, int64_t f_cluster_stride, int32_t r_cols,
                  float rel_div, const float* w1, const float* g1, const float* b1, int32_t h1, const float* w2,
                  const float* g2, const float* b2, int32_t h2, const float* w3, const float* g3, const float* b3, float eps,
                  int32_t act, int64_t n, float* out, int64_t out_stride, void* stream);
/* The same with the feature columns taken from up to three tensors standing side by side (feat_parts / feat_strides / feat_cols:
 * HOST arrays of num_parts entries) and, with feats_index (i64 [n]), through an index: row i of the layer input reads row
 * feats_index[i] of every part.  The first SIR layer of the LiDAR-query branch then needs neither the gather of the group-sampled
 * points' features nor the torch.cat([seg_logits, seg_vote_preds, seg_feats], 1) in front of it (FSF.py fsd_forward,
 * single_stage_fsd.py:867-901).  direct_parts_mask (round 6): bit p set = part p is NOT read through the index (its rows are the layer's
 * rows already) — the refine stage's `torch.cat([pts_feat[ext_pts_inds], pts_img_feat], -1)` (FSF.query_feat_refine, FSF.py:961-1010):
 * part 0 = the frame's point features through `ext_pts_inds`, part 1 = the image features of the pooled points as they stand. */
int fsf_sir_input_gather(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3],
                         const float* const* feat_parts, const int64_t* feat_strides, const int32_t* feat_cols, int32_t num_parts,
                         const int64_t* feats_index, int32_t direct_parts_mask, const float* extra, int64_t extra_stride, int32_t e_cols,
                         float extra_div, const float* f_cluster, int64_t f_cluster_stride, int32_t r_cols, float rel_div, const float* w1,
                         const float* g1, const float* b1, int32_t h1, const float* w2, const float* g2, const float* b2, int32_t h2,
                         const float* w3, const float* g3, const float* b3, float eps, int32_t act, int64_t n, float* out,
                         int64_t out_stride, void* stream);

/* K31 (round 6)  a whole SIR stack on rows SORTED by group as one call: SIR.forward (projects/mmdet3d_plugin/models/backbones/sir.py:65-85)
 * and FullySparseBboxHead.forward (models/roi_heads/bbox_heads/fsd_bbox_head.py:96-197) at test time.  Per block: fsf_sir_input_gather (the
 * block's input rows), then per DynamicVFELayer fsf_linear[_f16w]_norm_act_segmax, preceded from the second layer on by the group half of
 * the layer's weight on the previous layer's group maxima (fsf_linear_norm_act / fsf_linear_f16w_norm_act_grouped on [num_groups, c]) —
 * the calls hip_ops.sir_input / sst_ops.sorted_stack_forward issue from Python, with the same arguments, sequenced from C++.
 *   blocks        HOST array [num_blocks]; device pointers inside: the position MLP (w* f32 row-major, g* / b* LayerNorm affine), per layer
 *                 the prepared weights of the point half (`planes_left`, fsf_linear_prepare_weight[_f16]; *_f16 != 0: the f16 format)
 *                 and — layers >= 1 — of the group half (`planes_right`), bias / gamma / beta (or NULL), norm / act codes of
 *                 fsf_linear_norm_act, c = output channels; in_cols = the block's input width (p_cols + feature columns + e_cols);
 *   points / f_cluster / extra / seg_ids: the stack's rows in sorted order (fsf_sorted_rows); block 0 reads its feature columns from
 *                 feat_parts through feats_index exactly as fsf_sir_input_gather does, later blocks the previous block's rows;
 *   groups        f32 [num_groups, groups_stride >= sum of all layers' c], holding -inf: layer by layer, block by block, the group maxima
 *                 side by side (the `cat` SIR.forward returns);
 *   rows_out      f32 [n, c of the last layer] or NULL (the last layer's point rows are then never written);
 *   arena         device, 256-byte aligned, >= fsf_sir_stack_arena_bytes(blocks, num_blocks, n, num_groups).
 * n >= 1, num_groups >= 1 (the host path keeps the degenerate cases). */
#define FSF_SIR_MAX_LAYERS 4
typedef struct {
  const void* planes_left;
  const void* planes_right;
  int32_t left_f16, right_f16;
  const float *bias, *gamma, *beta;
  float eps;
  int32_t norm, act, c;
} FsfSirLayer;
typedef struct {
  const float *w1, *g1, *b1;
  const float *w2, *g2, *b2;
  const float *w3, *g3, *b3;
  int32_t h1, h2;
  float mlp_eps;
  int32_t mlp_act;
  float xyz_normalizer[3];
  float rel_div;
  int32_t in_cols, num_layers;
  FsfSirLayer layer[FSF_SIR_MAX_LAYERS];
} FsfSirBlock;
int64_t fsf_sir_stack_arena_bytes(const FsfSirBlock* blocks, int32_t num_blocks, int64_t n, int64_t num_groups);
int fsf_sir_stack_forward(const FsfSirBlock* blocks, int32_t num_blocks, const float* points, int64_t points_stride, int32_t p_cols,
                          const float* const* feat_parts, const int64_t* feat_strides, const int32_t* feat_cols, int32_t num_parts,
                          const int64_t* feats_index, int32_t direct_parts_mask, const float* extra, int64_t extra_stride, int32_t e_cols,
                          float extra_div, const float* f_cluster, int64_t f_cluster_stride, int32_t r_cols, const int64_t* seg_ids,
                          int64_t n, int64_t num_groups, float* groups, int64_t groups_stride, float* rows_out, void* arena,
                          int64_t arena_bytes, void* stream);

/* K28 (training)  y = cat([points[:, :3] / xyz_normalizer, points[:, 3:], feats, extra / extra_div], 1) * h  — the input side of
 * SIRLayer.forward [UNVENDORED; SURVEY App. C] around the position MLP's output h f32 [n, c] (contiguous), c = p_cols + f_cols + e_cols —
 * and its adjoint: grad_h = grad_out * x (x re-formed from the sources), grad_feats = (grad_out * h)[:, p_cols : p_cols + f_cols],
 * grad_extra = (grad_out * h)[:, p_cols + f_cols :] / extra_div (either may be NULL; contiguous).  points are not differentiated.
 * The same IEEE operations per element as the ATen chain they replace (two cats, a product; two products, a zero-filled slice
 * gradient and its accumulation): bit-identical. */
int fsf_concat_mul(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3], const float* feats,
                   int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride, int32_t e_cols, float extra_div,
                   const float* h, int64_t n, float* out, void* stream);
int fsf_concat_mul_backward(const float* points, int64_t points_stride, int32_t p_cols, const float xyz_normalizer[3], const float* feats,
                            int64_t feats_stride, int32_t f_cols, const float* extra, int64_t extra_stride, int32_t e_cols,
                            float extra_div, const float* h, const float* grad_out, int64_t n, float* grad_h, float* grad_feats,
                            float* grad_extra, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K17  dynamic point pooling: (point, RoI) memberships of the enlarged rotated boxes + box-frame geometry
 * Replaces: TorchEx dynamic_point_pool_ext.forward(rois, pts, extra_wlh, max_inbox_point, out_pts_idx,
 *   out_roi_idx, out_pts_feats) [UNVENDORED], bound at projects/mmdet3d_plugin/ops/dynamic_point_pool_op.py:5,32 and
 *   called per sample by DynamicPointROIExtractor (roi_extractors/dynamic_point_roi_extractor.py:54-59).
 *   rois f32 [n_rois, roi_stride]: box (cx, cy, cz_bottom, w, l, h, rz) at columns box_col..box_col+6, batch index
 *   (as float) at column batch_col or batch_col = -1 (single sample); pts f32 [n_pts, pts_stride] (xyz first),
 *   pts_batch i64 [n_pts] or NULL; extra_wlh {w, l, h} enlargement; caller-allocated outputs of max_all_pts rows:
 *   out_pts_idx i64, out_roi_idx i64, out_pts_feats f32 [.,13] = xyz | local xyz | six face distances | is_in_margin.
 * Same caller-allocates convention as upstream; unlike upstream (two atomic counters) the rows come out in ascending
 * (roi, point) order and the caps keep the FIRST max_inbox_point points of a RoI / first max_all_pts rows —
 * deterministic.  count = rows written (device scalar and/or host copy; the host copy synchronises the stream).
 */
int64_t fsf_dynamic_point_pool_workspace_bytes(int64_t n_pts, int64_t n_rois);
int fsf_dynamic_point_pool(const float* rois, int64_t n_rois, int32_t roi_stride, int32_t box_col, int32_t batch_col,
                           const float* pts, int64_t n_pts, int32_t pts_stride, const int64_t* pts_batch,
                           const float extra_wlh[3], int32_t max_inbox_point, int64_t max_all_pts,
                           int64_t* out_pts_idx, int64_t* out_roi_idx, float* out_pts_feats, int64_t* count_dev,
                           int64_t* count_host, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K20  greedy BEV NMS (rotated or axis-aligned IoU)
 * Replaces: mmdet3d.ops.iou3d nms_gpu / nms_normal_gpu [UNVENDORED] under mmdet3d.core.box3d_multiclass_nms, called
 *   from FrustumClusterHead._get_bboxes_single (dense_heads/frustum_cluster_head.py:636-667).
 *   boxes f32 [n,5] = (x1, y1, x2, y2, yaw) (xywhr2xyxyr of the BEV boxes), ALREADY in descending score order;
 *   keep i64 [n] receives the ascending positions of the survivors; num_keep as in K17's count.
 */
int64_t fsf_nms_bev_workspace_bytes(int64_t n);
int fsf_nms_bev(const float* boxes, int64_t n, float thresh, int32_t rotated, int64_t* keep, int64_t* num_keep_dev,
                int64_t* num_keep_host, void* workspace, int64_t workspace_bytes, void* stream);
/* All classes of one head in three launches (box3d_multiclass_nms runs nms_gpu once per class on the SAME boxes with
 * different score orders): the overlap test runs once in the original box order, each class gets its bits permuted into
 * its own score order, and the greedy scans of all classes run side by side (one workgroup each).
 *   boxes f32 [n,5] in the caller's order; rank i32 [num_classes,n] = position of box i in class c's descending-score
 *   order among the boxes above the score threshold, -1 otherwise; count i32 [num_classes] (device) = boxes per class;
 *   keep i64 [num_classes,n]: kept RANKS of class c, ascending, in keep[c, 0 .. num_keep[c]); num_keep i64
 *   [num_classes] (device).
 */
int64_t fsf_nms_bev_multiclass_workspace_bytes(int64_t n, int32_t num_classes);
int fsf_nms_bev_multiclass(const float* boxes, int64_t n, int32_t num_classes, const int32_t* rank, const int32_t* count,
                           float thresh, int32_t rotated, int64_t* keep, int64_t* num_keep, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* The same with a cap: a class scan stops after its first max_keep kept boxes (max_keep <= 0: no cap).  box3d_multiclass_nms
 * [UNVENDORED mmdet3d.core.post_processing] keeps only the max_num best scores over all classes afterwards
 * (frustum_cluster_head.py:661-663 passes cfg.max_num), and a class's kept boxes come out in descending score order, so
 * its boxes past the first max_num can never be among them: same result, and the scan — a latency chain of one 64-box word
 * per step — ends after max_num keeps instead of walking all n boxes.
 * With `incomplete` (device i32, written 0 / 1) the per-class masks only hold each class's best max(4 max_keep, 2048) boxes
 * (workspace: fsf_nms_bev_multiclass_capped_workspace_bytes — (1 + C w^2 / n^2) n^2 / 8 bytes instead of (1 + C) n^2 / 8); a class
 * that runs out of window before max_keep keeps sets *incomplete = 1 and the caller repeats the call with incomplete = NULL
 * (full masks, fsf_nms_bev_multiclass_workspace_bytes). */
int64_t fsf_nms_bev_multiclass_capped_workspace_bytes(int64_t n, int32_t num_classes, int64_t max_keep);
int fsf_nms_bev_multiclass_capped(const float* boxes, int64_t n, int32_t num_classes, const int32_t* rank, const int32_t* count,
                                  float thresh, int32_t rotated, int64_t max_keep, int64_t* keep, int64_t* num_keep,
                                  int32_t* incomplete, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K24  the box tail of a cluster head around K20 (round 3)
 * Replaces, for one sample and one task of FrustumClusterHead._get_bboxes_single
 *   (projects/mmdet3d_plugin/models/dense_heads/frustum_cluster_head.py:595-698) and what it calls:
 *   - `cls_logits.sigmoid()`, BasePointBBoxCoder.decode (projects/mmdet3d_plugin/core/bbox/coders/base_point_bbox_coder.py:58-82:
 *     xyz = reg[:, :3] + centre, dims = exp(reg[:, 3:6]) - eps, yaw = atan2(reg[:, 6], reg[:, 7]), velocity appended for code size 10),
 *     LiDARInstance3DBoxes(...).bev and xywhr2xyxyr [UNVENDORED mmdet3d.core.bbox]          -> fsf_decode_cluster_boxes
 *       cls_logits f32 [n, C], reg_preds f32 [n, code] (code 8 | 10), cluster_xyz f32 [n, 3], each with its row stride in floats
 *       (the heads hand over column slices of wider outputs) ->
 *       boxes f32 [n, code - 1], boxes_nms f32 [n, 5] = (x1, y1, x2, y2, yaw), scores_t f32 [C, n] (class-major sigmoid scores);
 *   - box3d_multiclass_nms [UNVENDORED mmdet3d.core.post_processing]: the per-class `scores > score_thr` mask and score sort
 *                                                                                           -> fsf_class_rank_desc
 *       order i32 [C, n] = class c's boxes by descending score (ties: ascending index; boxes at or under the threshold last, in
 *       index order), rank i32 [C, n] = position of box i in that order or -1 under the threshold, count i32 [C] — rank / count
 *       are fsf_nms_bev_multiclass[_capped]'s inputs;
 *   - its concatenation of the classes' kept boxes, the `scores.sort(descending=True)[:max_num]` cut, the label table of
 *     frustum_cluster_head.py:680-690 and bbox3d2result's packing                             -> fsf_nms_select
 *       keep / num_keep as K20 returned them (keep_stride = row stride of keep), max_keep = the cap K20 ran with
 *       (num_classes * max_keep <= fsf_nms_select_capacity() = 16 384, else FSF_ERR_UNSUPPORTED) ->
 *       out f32 [max_num, box_dim + 2] rows (box | score | label as float, label = label_lut[c] or c), class-major and
 *       score-descending within a class when at most max_num boxes were kept, otherwise the max_num best by descending
 *       score (ties: class-major order); meta i32 [4] = (rows written, boxes kept over all classes, *incomplete or 0, 0).
 * Nothing here synchronises; the caller reads `out` and `meta` back with one copy.
 */
int fsf_decode_cluster_boxes(const float* cls_logits, int64_t cls_stride, const float* reg_preds, int64_t reg_stride,
                             const float* cluster_xyz, int64_t xyz_stride, int64_t n, int32_t num_classes, int32_t code_size, float eps,
                             float* boxes, float* boxes_nms, float* scores_t, void* stream);
int64_t fsf_class_rank_desc_workspace_bytes(int64_t n, int32_t num_classes);
int fsf_class_rank_desc(const float* scores_t, int64_t n, int32_t num_classes, float score_thr, int32_t* order, int32_t* rank,
                        int32_t* count, void* workspace, int64_t workspace_bytes, void* stream);
int64_t fsf_nms_select_capacity(void);
/* the most classes fsf_class_rank_desc / fsf_nms_select take (32: beyond it they return FSF_ERR_UNSUPPORTED and the caller keeps
 * the per-class path of box3d_multiclass_nms) */
int32_t fsf_box_tail_max_classes(void);
int fsf_nms_select(const float* boxes, int32_t box_dim, const float* scores_t, const int32_t* order, const int64_t* keep,
                   int64_t keep_stride, const int64_t* num_keep, int64_t n, int32_t num_classes, int64_t max_keep, int32_t max_num,
                   const int64_t* label_lut, const int32_t* incomplete, float* out, int32_t* meta, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K18  in-group rank (TorchEx ingroup_indices [UNVENDORED]); sst_ops.py:239-259.
 * Upstream is an atomicAdd counter (a nondeterministic permutation of 0..n_g-1 per group); this returns the
 * stable rank (ascending original index), which satisfies the same contract (sst_ops.py:225-235).
 */
int64_t fsf_ingroup_rank_workspace_bytes(int64_t n);
int fsf_ingroup_rank(const int64_t* group_inds, int64_t n, int64_t* out_inds, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K25  ClusterAssigner's density filter over ALL class groups at once
 * Replaces: in ClusterAssigner.forward_single_class (projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:951-956,
 *   called per class group from :903-935) the `filter_almost_empty(coors, min_points)` mask (`:31-35`: torch.unique + counts +
 *   `cnt[inv] >= min_points`), the `if not valid_mask.any(): valid_mask = ~valid_mask` rule and the boolean compactions of the
 *   points and their voxel keys behind it (two host round trips per group).  With the groups as the leading key column
 *   (key[0] = group * batch_size + sample) and ONE fsf_unique_rows over all (group, point) pairs:
 *   new_keys i64 [m, key_cols] (ascending, as fsf_unique_rows returns them), cnt i64 [m], inv i64 [n] (pair -> key) ->
 *   a key survives iff cnt >= min_points, or no key of its group does;
 *   k_idx i64 [<= m] = surviving keys in ascending order, v_idx i64 [<= n] = surviving pairs in ascending order,
 *   vox_inv i64 [<= n] = position of pair v_idx[j]'s key in k_idx (the inverse a second unique over the survivors would return);
 *   k_group i32 [<= m] (optional) = class group of every surviving key (what the grouped connected components take);
 *   counts_host[0] = surviving keys, counts_host[1] = surviving pairs (one 16-byte read-back, one stream sync).
 *   Stable compactions by single-pass look-back scans: deterministic.  num_groups <= 64.
 * fsf_cluster_point_ids: the tail of the same function (`:971-977`) for all groups at once — the component labels of the cluster
 *   voxels (fsf_connected_components_grouped: numbered by first member over all voxels; the voxels are group-sorted) renumbered
 *   from 0 inside each group and mapped back to the surviving pairs: out i64 [nv, 3] = (group, sample, cluster id), the rows
 *   `combine_classes` concatenates (`:892-901`).  workspace >= 256 bytes.
 */
int64_t fsf_cluster_key_survival_workspace_bytes(int64_t m, int64_t n);
int fsf_cluster_key_survival(const int64_t* new_keys, int32_t key_cols, const int64_t* cnt, int64_t m, const int64_t* inv, int64_t n,
                             int64_t batch_size, int64_t min_points, int32_t num_groups, int64_t* k_idx, int32_t* k_group,
                             int64_t* v_idx, int64_t* vox_inv, int64_t* counts_host, void* workspace, int64_t workspace_bytes,
                             void* stream);
int fsf_cluster_point_ids(const int32_t* labels, const int32_t* vox_group, int64_t m, const int64_t* vox_inv, const int64_t* g_ids,
                          const int64_t* b_pts, int64_t nv, int32_t num_groups, int64_t* out, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* K27  the (group, point) pairs of SingleStageFSD's grouped sampling for ONE sample (single_stage_fsd.py:826-838):
 *   fg = score > thresh[None, :];  keep_one: fg[0] |= ~fg.any(0) (a group nobody passes keeps point 0);  (g_ids, p_ids) = fg.t().nonzero()
 *   score f32 [n, ng] (row stride score_stride floats, ng <= 32), thresh f32 [ng] -> g_ids, p_ids i64 [capacity >= n * ng], the first
 *   *count_host entries valid, group-major, points ascending inside a group.  One read-back.
 *   group_class_masks (HOST u32 [ng], or NULL): `score` is then the CLASS score matrix f32 [n, num_classes] and group g's score the sum
 *   of its member columns (bit c of mask g) — `gather_group_by_names` (:868-872) folded in; one or two members per group (one add has one
 *   result whatever the order: identical to the reference's column sums), else FSF_ERR_UNSUPPORTED. */
int64_t fsf_group_pairs_workspace_bytes(int64_t n, int32_t ng);
int fsf_group_pairs(const float* score, int64_t n, int32_t ng, int64_t score_stride, const float* thresh, int32_t keep_one,
                    const uint32_t* group_class_masks, int32_t num_classes, int64_t* g_ids, int64_t* p_ids, int64_t capacity,
                    int64_t* count_host, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K29  the row-shuffling glue between the query stages (round 6): each entry point is ONE launch for what the reference writes as a
 * chain of 3-18 elementwise / index / cat operations; the arithmetic is that chain's, operation for operation (single IEEE adds,
 * subtracts, multiplies, divides; expf / atan2f as in fsf_decode_cluster_boxes), so the results are bit-identical to it.
 *
 * fsf_sorted_rows — the inputs of a SIR stack on rows sorted by group.  Replaces: in `SIR.forward`
 *   (projects/mmdet3d_plugin/models/backbones/sir.py:65-85) the per-block `torch.cat([points, out_feats], 1)` operands after the
 *   stack's one `torch.unique` (:68), taken through the unique's sort order; with `f_cluster == NULL` also the
 *   `f_cluster = points[:, :3] - cluster_xyz[inv]` of SingleStageFSD.extract_feat (detectors/single_stage_fsd.py:458-474) and of
 *   FSF.get_cluster_delta_weighted (detectors/FSF.py:313-329).
 *   order i32 [n] (sorted position -> source row, fsf_unique_rows' `order`), inv i64 [n] (source row -> group);
 *   seg_ids[i] = inv[order[i]];  pts_sorted[i, :] = points[order[i], :pts_cols];  idx_sorted[i] = index ? index[order[i]] : order[i];
 *   fcl_sorted[i, :] = f_cluster ? f_cluster[order[i], :3] : points[order[i], :3] - centers[inv[order[i]], :3];
 *   fill[0 .. fill_count) = fill_value (the stack's group table starts at -inf) when fill != NULL.
 * fsf_compact_pairs — ClusterAssigner's survivors.  Replaces: the boolean compactions `points[valid_mask]`, `batch_idx[valid_mask]`,
 *   `coors[valid_mask]` of ClusterAssigner.forward_single_class (single_stage_fsd.py:951-960) and update_sample_results_by_mask
 *   (:867-890) for all class groups at once, with the survivor lists of fsf_cluster_key_survival:
 *   vox_centers[j] = means[k_idx[j]] (nk rows), (g_out, p_out, b_out, centers_out)[j] = (g_ids, p_ids, b_pts, centers)[v_idx[j]] (nv rows).
 * fsf_combine_queries — FSF.combine_frustum_and_fsd (FSF.py:657-692) without the feature / head-output concatenations: camera
 *   queries first, LiDAR coors (class, batch, id) -> (batch, class, id + begin_idx), all-zero 2-D predictions for the LiDAR queries.
 * fsf_decode_rois — FSF.decode_stage_bboxes (FSF.py:1085-1094) for a single-task head: BasePointBBoxCoder.decode
 *   (core/bbox/coders/base_point_bbox_coder.py:58-82) + the batch column; rois f32 [m, code_size] = (batch, x, y, z, dx, dy, dz, yaw[, vx, vy]).
 * fsf_refine_rows — after fsf_dynamic_point_pool: `points[ext_pts_inds]` (FSF.query_feat_refine, FSF.py:961-1010) and
 *   FullySparseBboxHead.forward's `f_cluster = cat([local_xyz, boundary_offset, is_in_margin, pts_xyz - roi_centre[roi_inds]], 1)`
 *   (roi_heads/bbox_heads/fsd_bbox_head.py:96-112); info f32 [k, 13] as K17 writes it, roi_xyz = the RoIs' centre columns.
 * fsf_encode_preds_2d — FSF.get_single_cls_preds_2d + encode_preds_2d (FSF.py:476-504, :449-474) for ONE sample's camera queries:
 *   preds_2d[i] = mask_anno[id - 1] (zeros with category = num_classes for id <= 0), encoded[i] = (box / (w, h, w, h), score,
 *   one_hot(category, num_classes + 1)) with the division as the product with the fp32 reciprocal ATen forms for a host scalar.
 * fsf_weighted_xyz / fsf_centroid_divide — FSF.get_cluster_delta_weighted's operands (FSF.py:313-329): out f32 [n, 4] =
 *   (xyz * w, w) with w = max(weight, weight_min) (NaN kept); out f32 [m, 3] = mean[:, :3] / mean[:, 3:4].
 */
int fsf_sorted_rows(const int32_t* order, const int64_t* inv, int64_t n, const float* points, int64_t pts_stride, int32_t pts_cols,
                    const float* f_cluster, int64_t fcl_stride, const float* centers, int64_t centers_stride, const int64_t* index,
                    int64_t* seg_ids, float* pts_sorted, float* fcl_sorted, int64_t* idx_sorted, float* fill, int64_t fill_count,
                    float fill_value, void* stream);
int fsf_compact_pairs(const float* means, int64_t means_stride, const int64_t* k_idx, int64_t nk, float* vox_centers, const int64_t* g_ids,
                      const int64_t* p_ids, const int64_t* b_pts, const float* centers, const int64_t* v_idx, int64_t nv, int64_t* g_out,
                      int64_t* p_out, int64_t* b_out, float* centers_out, void* stream);
int fsf_combine_queries(const float* f_centers, int64_t mf, const float* l_centers, int64_t ml, const int64_t* f_coors,
                        const int64_t* l_coors, const float* f_preds_2d, int32_t d, int64_t begin_idx, float* centers, int64_t* coors,
                        float* preds_2d, void* stream);
int fsf_decode_rois(const float* reg_preds, int64_t reg_stride, int32_t code_size, const float* centers, int64_t centers_stride,
                    const int64_t* batch, int64_t batch_stride, int64_t m, float eps, float* rois, void* stream);
int fsf_refine_rows(const float* info, const float* points, int64_t points_stride, int32_t points_cols, const int64_t* pts_idx,
                    const int64_t* roi_idx, const float* roi_xyz, int64_t roi_stride, int64_t k, float* points_out, float* f_cluster,
                    void* stream);
int fsf_encode_preds_2d(const float* mask_anno, int64_t num_anno, int32_t d, const int64_t* obj_coors, int64_t m, int32_t num_classes,
                        float img_w, float img_h, float* preds_2d, float* encoded, int64_t enc_stride, void* stream);
int fsf_weighted_xyz(const float* points, int64_t points_stride, const float* weights, int64_t n, float weight_min, float* out,
                     void* stream);
int fsf_centroid_divide(const float* mean, int64_t m, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K30  the LiDAR-query branch's clustering front end as ONE call (round 6): a stage-level driver that sequences eleven of the entry points
 * above from C++ on the caller's stream.
 * Replaces: SingleStageFSD.group_sample (projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:802-865), ClusterAssigner.forward
 *   (:903-982), update_sample_results_by_mask (:867-890), combine_classes (:892-901), the `scatter_v2(center_preds, pts_cluster_inds, 'avg')`
 *   of SingleStageFSD.extract_feat (:458-474) and the `torch.unique` of SIR.forward (models/backbones/sir.py:68) — for all class groups of
 *   ONE sample: fsf_group_pairs -> fsf_vote_centers_keys -> fsf_unique_rows -> fsf_cluster_key_survival -> fsf_segment_reduce_short ->
 *   fsf_compact_pairs -> fsf_connected_components_grouped -> fsf_cluster_point_ids -> fsf_gather_rows_strided -> fsf_unique_rows ->
 *   fsf_segment_reduce, same arguments as the Python sequence issues (bit-identical results); the four host waits between them stay (each
 *   count sizes what follows) but the next launch follows a wait by microseconds instead of the interpreter's 30-110 us.
 *   scores f32 [m, num_classes] (row stride score_stride; softmax of the logits without its last column), thresh f32 [ng] (device),
 *   group_class_masks u32 [ng] (HOST: one or two member classes per group), logits / offsets / points: the pre-voxelized fields
 *   (row strides in floats), batch_idx i64 [m] or NULL, group_voxel_size f32 [ng * 3] (HOST), key_min / key_max i64 [4] (HOST: bounds of
 *   the (group, vx, vy, vz) keys; a key outside them sends that unique through its range pass), dist_table f32 [ng] (device).
 *   arena: >= fsf_lidar_cluster_frontend_arena_bytes(m, ng, point_cols) bytes, 256-byte aligned; every result lives inside it at the byte
 *   offset out[FSF_LCF_OFF_*]:  p_ids i64 [rows], centers f32 [rows, 3], cluster_inds i64 [rows, 3] = (group, sample, cluster id), points
 *   f32 [rows, point_cols], new_coors i64 [clusters, 3], inv i64 [rows], cnt i64 [clusters], order i32 [rows], seg_offsets i32
 *   [clusters + 1], cluster_xyz f32 [clusters, 3]; out[FSF_LCF_PAIRS .. FSF_LCF_CLUSTERS] = the five counts.
 */
#define FSF_LCF_PAIRS 0
#define FSF_LCF_KEYS 1
#define FSF_LCF_KEPT_KEYS 2
#define FSF_LCF_ROWS 3
#define FSF_LCF_CLUSTERS 4
#define FSF_LCF_OFF_P_IDS 5
#define FSF_LCF_OFF_CENTERS 6
#define FSF_LCF_OFF_CLUSTER_INDS 7
#define FSF_LCF_OFF_POINTS 8
#define FSF_LCF_OFF_NEW_COORS 9
#define FSF_LCF_OFF_INV 10
#define FSF_LCF_OFF_CNT 11
#define FSF_LCF_OFF_ORDER 12
#define FSF_LCF_OFF_SEG_OFFSETS 13
#define FSF_LCF_OFF_CLUSTER_XYZ 14
#define FSF_LCF_OUT_WORDS 16
int64_t fsf_lidar_cluster_frontend_arena_bytes(int64_t m, int32_t ng, int32_t point_cols);
int fsf_lidar_cluster_frontend(const float* scores, int64_t m, int32_t num_classes, int64_t score_stride, const float* thresh, int32_t ng,
                               const uint32_t* group_class_masks, const float* logits, int32_t logit_stride, const float* offsets,
                               int32_t offset_stride, const float* points, int32_t point_stride, int32_t point_cols,
                               const int64_t* batch_idx, const float* group_voxel_size, const float range_min[3], const int64_t key_min[4],
                               const int64_t key_max[4], int64_t min_points, const float* dist_table, void* arena, int64_t arena_bytes,
                               int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FSF_HIP_H_ */
