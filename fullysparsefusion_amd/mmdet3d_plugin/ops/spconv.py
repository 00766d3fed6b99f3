"""Module-level mirror of `mmdet3d.ops.spconv` (vendored spconv v1.x, [UNVENDORED] in the reference tree; imported
at projects/mmdet3d_plugin/ops/sst_ops.py:5 and used by SimpleSparseUNet): `SparseConvTensor`, `SubMConv3d`,
`SparseConv3d`, `SparseInverseConv3d`, `SparseSequential`, `SparseBasicBlock`, `make_sparse_convmodule`.

Same constructor arguments, parameter names/shapes (weight [kz,ky,kx,Cin,Cout], no bias when a norm follows) and
`indice_key` caching semantics; the rulebook is an output-major neighbour table built through a hash (no dense
grid) and each conv layer is ONE fused gather -> fp32 MFMA -> epilogue launch instead of 27 x (gather, mm,
scatter-add).  In eval mode the BatchNorm affine, the residual add and the ReLU that follow a conv are folded
into that launch.
"""
from ... import switches
import math
import os

import torch
import torch.nn as nn

from ... import hip_ops


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        # `features` may be given as the PARTS of a channel concatenation (a tuple of [m, c_i] tensors: the decoder's
        # `cat([x_bottom.features, x_lateral.features], 1)`): the concatenation is written on first read of `.features` — never, when
        # the consumers read the parts (the plane-form merge convolution, fsf_channel_pair_sum_add2)
        self.feature_parts = tuple(features) if isinstance(features, (tuple, list)) else None
        self._features = None if self.feature_parts is not None else features
        self.indices = indices  # i32 [m,4] (b,z,y,x)
        self.spatial_shape = list(spatial_shape)
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        # K9c plane form of `features` (hip_ops.Planes, one per source of a channel concatenation), when a producer emitted it
        self.plane_sources = None
        # fp32 rows that exist only as a recipe (a producer that emitted `plane_sources` for a plane-form consumer): formed on first read
        self.features_thunk = None

    @property
    def features(self):
        if self._features is None and self.feature_parts is not None:
            self._features = torch.cat(self.feature_parts, dim=1)
        elif self._features is None and self.features_thunk is not None:
            self._features, self.features_thunk = self.features_thunk(), None
        return self._features

    @features.setter
    def features(self, value):
        self._features, self.feature_parts = value, None

    @property
    def num_channels(self):
        if self._features is None and self.feature_parts:
            return sum(int(t.size(1)) for t in self.feature_parts)
        if self._features is None and self.plane_sources:
            return sum(int(p.c) for p in self.plane_sources)
        return int(self._features.size(1))

    @property
    def spatial_size(self):
        return math.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        b, c = self.batch_size, self.features.size(1)
        out = self.features.new_zeros((b, *self.spatial_shape, c))
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out

    def _like(self, features, indices=None, spatial_shape=None):
        t = SparseConvTensor(features, self.indices if indices is None else indices,
                             self.spatial_shape if spatial_shape is None else spatial_shape, self.batch_size, self.grid)
        t.indice_dict = self.indice_dict
        return t


class Rulebook:
    """What spconv v1 keeps per indice_key (outids, indices, indice_pairs, indice_pair_num, out_spatial_shape),
    in output-major form."""

    def __init__(self, kind, nbr, in_indices, in_shape, out_indices, out_shape, nbr_inv=None):
        self.kind = kind
        self.nbr = nbr                  # i32 [m_out, kvol]
        self.nbr_inv = nbr_inv          # i32 [m_in, kvol] (strided only)
        self.in_indices, self.in_shape = in_indices, in_shape
        self.out_indices, self.out_shape = out_indices, out_shape
        self._pairs = {}

    def table(self, inverse):
        return self.nbr_inv if inverse else self.nbr

    def table_transposed(self, inverse):
        """(table, flip_k) for the data gradient: the conv over the transposed rulebook.  A SubM rulebook is its
        own transpose up to reversing the kernel offsets."""
        if self.kind == "subm":
            return self.nbr, True
        return (self.nbr if inverse else self.nbr_inv), False

    def pairs(self, inverse):
        """spconv-v1 pair lists of the table (built once per rulebook, only when a weight gradient is needed)."""
        if inverse not in self._pairs:
            self._pairs[inverse] = hip_ops.rulebook_to_pairs(self.table(inverse))
        return self._pairs[inverse]


def _plane_srcs(feat):
    """fp32 rows -> K9c plane sources (<= 128 channels each; a wider tensor is the concatenation of its two halves)."""
    c = feat.size(1)
    if c <= 128:
        return [hip_ops.to_planes(feat)]
    return [hip_ops.to_planes(feat[:, :c // 2]), hip_ops.to_planes(feat[:, c // 2:])]


def _planes_shape_ok(cin, cout, kvol, rows):
    cins = [cin] if cin <= 128 else [cin // 2, cin - cin // 2]
    return rows >= SparseConvolution.PLANES_MIN_ROWS and hip_ops.spconv_planes_supported(cins, cout, kvol)


class _SparseConvFn(torch.autograd.Function):
    """out = sum_k feat[table[:, k]] @ W[k] with the K10 backward: data gradient = the same fused kernel over the
    transposed table, weight gradient = fsf_spconv_backward_weight over the pair lists.  `split` picks the
    row-stationary split-bf16 kernel (K9b, fp32-accurate) for the forward and the data gradient, as in inference;
    `planes` the f16-plane kernel (K9c) wherever the direction's shape is one it takes (the operand — features forward,
    grad_out backward — is converted by fsf_to_planes, the weights of the step by fsf_spconv_prepare_weight_planes)."""

    @staticmethod
    def forward(ctx, feat, weight, rb, inverse, split, planes=False):
        kvol = rb.nbr.size(1)
        w = weight.detach().reshape(kvol, weight.shape[-2], weight.shape[-1])
        ctx.rb, ctx.inverse, ctx.split, ctx.planes = rb, inverse, split, planes
        ctx.save_for_backward(feat, weight)
        table = rb.table(inverse)
        if planes and feat.size(0) > 0 and _planes_shape_ok(w.size(1), w.size(2), kvol, table.size(0)):
            return hip_ops.spconv_forward_planes(_plane_srcs(feat), hip_ops.spconv_prepare_weight_planes(w), kvol, w.size(2), table)[0]
        if split and feat.size(0) > 0:
            return hip_ops.spconv_forward_split(feat, hip_ops.spconv_prepare_weight_split(w), kvol, w.size(2), table)
        return hip_ops.spconv_forward(feat, hip_ops.spconv_transpose_weight(w), table)

    @staticmethod
    def backward(ctx, grad):
        feat, weight = ctx.saved_tensors
        rb, inverse = ctx.rb, ctx.inverse
        grad = grad.contiguous()
        kvol = rb.nbr.size(1)
        g_feat = g_w = None
        if ctx.needs_input_grad[0]:
            table_t, flip = rb.table_transposed(inverse)
            w = weight.detach().reshape(kvol, weight.shape[-2], weight.shape[-1])
            w = w.flip(0) if flip else w
            if (ctx.planes and grad.size(0) > 0 and feat.size(0) > 0
                    and _planes_shape_ok(w.size(2), w.size(1), kvol, table_t.size(0))):
                wt = hip_ops.spconv_prepare_weight_planes(w.transpose(1, 2).contiguous())
                g_feat = hip_ops.spconv_forward_planes(_plane_srcs(grad), wt, kvol, w.size(1), table_t)[0]
            elif ctx.split and grad.size(0) > 0 and feat.size(0) > 0:
                planes = hip_ops.spconv_prepare_weight_split(w.transpose(1, 2).contiguous())
                g_feat = hip_ops.spconv_forward_split(grad, planes, kvol, w.size(1), table_t)
            else:
                g_feat = hip_ops.spconv_forward(grad, w, table_t)
        if ctx.needs_input_grad[1]:
            pairs, num = rb.pairs(inverse)
            g_w = hip_ops.spconv_backward_weight(feat, grad, pairs, num).reshape(weight.shape)
        return g_feat, g_w, None, None, None, None


def _to3(v):
    return [int(v)] * 3 if not isinstance(v, (list, tuple)) else [int(x) for x in v]


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _to3(kernel_size)
        self.stride, self.padding, self.dilation = _to3(stride), _to3(padding), _to3(dilation)
        self.conv1x1 = math.prod(self.kernel_size) == 1
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._wt_cache = None
        self.reset_parameters()

    def reset_parameters(self):
        # spconv v1: kaiming_uniform(a=sqrt(5)) with fan_in = kvol * Cin
        fan_in = math.prod(self.kernel_size) * self.in_channels
        bound = 1.0 / math.sqrt(fan_in) * math.sqrt(3.0) * math.sqrt(2.0 / (1 + 5.0))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                b = 1.0 / math.sqrt(fan_in)
                self.bias.uniform_(-b, b)

    # -- rulebook ------------------------------------------------------------------------------------
    def _rulebook(self, x):
        rb = x.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and rb.kind == "strided", "SparseInverseConv3d needs the forward conv's indice_key"
            return rb
        if rb is not None:
            return rb
        if self.subm:
            nbr = hip_ops.rulebook_subm(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.dilation)
            rb = Rulebook("subm", nbr, x.indices, x.spatial_shape, x.indices, x.spatial_shape)
        else:
            out_idx, nbr, nbr_inv, out_shape = hip_ops.rulebook_strided(
                x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)
            key = getattr(self, "reorder_output_key", None)
            if (key is not None and x.indice_dict.get("__mask_order__", False) and out_idx.size(0) >= switches.UNET_MASK_ORDER_MIN_ROWS
                    and x.find_indice_pair(key) is None):
                # the coarse level in neighbour-mask order (SimpleSparseUNet.forward): output rows permuted, the inverse table's
                # VALUES (coarse rows) remapped, and the level's submanifold rulebook registered on the permuted coordinates
                perm, inv_perm = hip_ops.order_by_neighbor_mask(out_idx, x.batch_size, out_shape)
                perm64 = perm.long()
                out_idx = out_idx.index_select(0, perm64)
                nbr = nbr.index_select(0, perm64)
                nbr_inv = hip_ops.remap_indices(nbr_inv, inv_perm)
                x.indice_dict[key] = Rulebook("subm", hip_ops.rulebook_subm(out_idx, x.batch_size, out_shape), out_idx, out_shape,
                                              out_idx, out_shape)
            rb = Rulebook("strided", nbr, x.indices, x.spatial_shape, out_idx, out_shape, nbr_inv)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def _weight_t(self):
        w = self.weight
        key = (w._version, w.data_ptr())
        if self._wt_cache is None or self._wt_cache[0] != key:
            kvol = math.prod(self.kernel_size)
            wt = hip_ops.spconv_transpose_weight(w.detach().reshape(kvol, self.in_channels, self.out_channels))
            self._wt_cache = (key, wt)
        return self._wt_cache[1]

    # Which forward kernel (measured per layer on the 10-sweep frame, tools/profiling/scs_layers.py): the row-stationary
    # bf16-split kernel (K9b) wins on every submanifold layer (dense neighbourhoods: 1.15-1.7x on the fine levels,
    # 1.5-2x on the deep ones, where it splits the offset loop over more workgroups) and on the strided convolutions
    # into the deep levels (13+ pairs per output row); inverse convolutions and the strided ones into the fine levels
    # (3-9 pairs per output row) stay on the compacting fp32-pipe kernel.
    SPLIT_STRIDED_MAX_ROWS = 10000

    def _use_split_kernel(self, m_out):
        if self.in_channels % 4 or self.out_channels % 4:
            return False
        return self.subm or (not self.inverse and m_out <= self.SPLIT_STRIDED_MAX_ROWS)

    def _use_split_kernel_training(self):
        """Training uses K9b for the forward AND the data gradient of the submanifold layers (their transposed table
        has the same row count); strided / inverse layers keep the compacting fp32 kernel in both directions."""
        return self.subm and self.in_channels % 4 == 0 and self.out_channels % 4 == 0

    def _weight_split(self):
        w = self.weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_wsplit_cache")
        if cache is None or cache[0] != key:
            kvol = math.prod(self.kernel_size)
            cache = (key, hip_ops.spconv_prepare_weight_split(w.detach().reshape(kvol, self.in_channels, self.out_channels)))
            self.__dict__["_wsplit_cache"] = cache
        return cache[1]

    def _weight_split_f16(self):
        w = self.weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_wsplit16_cache")
        if cache is None or cache[0] != key:
            kvol = math.prod(self.kernel_size)
            cache = (key, hip_ops.spconv_prepare_weight_split_f16(w.detach().reshape(kvol, self.in_channels, self.out_channels)))
            self.__dict__["_wsplit16_cache"] = cache
        return cache[1]

    # K9c (pre-split f16 planes, cell skipping): every layer whose sources are <= 128 channels wide — submanifold, strided and
    # inverse alike (the kernel only sees a neighbour table).  Below ~4 k output rows the launch does not fill the chip and
    # K9b's offset splits win.
    PLANES_MIN_ROWS = switches.PLANES_MIN_ROWS
    emit_planes = True   # plane-form output next to the fp32 one (the consumer is another K9c layer); the U-Net clears it where not

    def _weight_planes(self):
        w = self.weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_wplanes_cache")
        if cache is None or cache[0] != key:
            kvol = math.prod(self.kernel_size)
            cache = (key, hip_ops.spconv_prepare_weight_planes(w.detach().reshape(kvol, self.in_channels, self.out_channels)))
            self.__dict__["_wplanes_cache"] = cache
        return cache[1]

    def _plane_sources(self, x):
        """The input in plane form: what the producer emitted, else a conversion of the fp32 features (<= 128 channels per
        source; a 256-channel input is the concatenation of two halves)."""
        srcs = x.plane_sources
        if srcs is not None and sum(p.c for p in srcs) == self.in_channels and all(p.m == x.indices.size(0) for p in srcs):
            return srcs
        f = x.features
        if self.in_channels <= 128:
            return [hip_ops.to_planes(f)]
        half = self.in_channels // 2
        return [hip_ops.to_planes(f[:, :half]), hip_ops.to_planes(f[:, half:])]

    def _use_planes_kernel(self, x, m_out):
        if not (m_out >= self.PLANES_MIN_ROWS and switches.PLANES):
            return False
        cin = self.in_channels
        cins = [p.c for p in x.plane_sources] if x.plane_sources is not None else ([cin] if cin <= 128 else [cin // 2, cin - cin // 2])
        return sum(cins) == cin and hip_ops.spconv_planes_supported(cins, self.out_channels, math.prod(self.kernel_size))

    def forward(self, x, scale=None, shift=None, residual=None, relu=False):
        """x: SparseConvTensor.  The optional epilogue arguments are the eval-mode BN affine / residual / ReLU
        that SparseSequential and SparseBasicBlock fold into the conv launch."""
        rb = self._rulebook(x)
        if self.inverse:
            nbr, out_indices, out_shape = rb.nbr_inv, rb.in_indices, rb.in_shape
        else:
            nbr, out_indices, out_shape = rb.nbr, rb.out_indices, rb.out_shape
        if shift is None and self.bias is not None:
            shift = self.bias
        elif shift is not None and self.bias is not None:
            shift = shift + (self.bias * scale if scale is not None else self.bias)
        if (not torch.is_grad_enabled() and x.plane_sources is not None and x.indices.size(0) > 0
                and self._use_planes_kernel(x, nbr.size(0))):  # (reads the plane sources only: a lazy concatenation stays unwritten)
            out, planes = hip_ops.spconv_forward_planes(self._plane_sources(x), self._weight_planes(), nbr.size(1), self.out_channels,
                                                        nbr, scale=scale, shift=shift, residual=residual, relu=relu,
                                                        want_planes=self.emit_planes and self.out_channels <= 128)
            y = x._like(out, out_indices, out_shape)
            y.plane_sources = [planes] if planes is not None else None
            return y
        feat = x.features
        needs_grad = torch.is_grad_enabled() and (feat.requires_grad or self.weight.requires_grad)
        if needs_grad:  # training: the epilogue stays in autograd-visible torch ops
            out = _SparseConvFn.apply(feat, self.weight, rb, self.inverse, self._use_split_kernel_training(),
                                      switches.TRAIN_PLANES and switches.PLANES)
            if scale is not None:
                out = out * scale
            if shift is not None:
                out = out + shift
            if residual is not None:
                out = out + residual
            if relu:
                out = torch.relu(out)
        elif feat.size(0) > 0 and self._use_planes_kernel(x, nbr.size(0)):
            out, planes = hip_ops.spconv_forward_planes(self._plane_sources(x), self._weight_planes(), nbr.size(1), self.out_channels,
                                                        nbr, scale=scale, shift=shift, residual=residual, relu=relu,
                                                        want_planes=self.emit_planes and self.out_channels <= 128)
            y = x._like(out, out_indices, out_shape)
            y.plane_sources = [planes] if planes is not None else None
            return y
        elif self._use_split_kernel(nbr.size(0)) and feat.size(0) > 0:
            if (self.in_channels >= 256  # (measured at 256 .. 1024 input channels; below, the conversion is the gain)
                    and hip_ops.spconv_split_planes_supported(self.in_channels, self.out_channels)
                    and hip_ops.rows_to_planes_supported(feat)):
                # K9b-XP: the deep levels (512 / 1024 input channels on a few thousand rows) with both operands as f16 planes
                out = hip_ops.spconv_forward_split_planes(hip_ops.rows_to_planes(feat), self._weight_split_f16(), nbr.size(1),
                                                          self.out_channels, nbr, scale=scale, shift=shift, residual=residual, relu=relu)
            else:
                out = hip_ops.spconv_forward_split(feat, self._weight_split(), nbr.size(1), self.out_channels, nbr, scale=scale,
                                                   shift=shift, residual=residual, relu=relu)
        else:
            out = hip_ops.spconv_forward(feat, self._weight_t(), nbr, scale=scale, shift=shift, residual=residual,
                                         relu=relu)
        return x._like(out, out_indices, out_shape)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


CONV_TYPES = {"SparseConv3d": SparseConv3d, "SubMConv3d": SubMConv3d, "SparseInverseConv3d": SparseInverseConv3d}


def _bn_affine(bn):
    """eval-mode BatchNorm1d as y = x * scale + shift, cached per module until a parameter / buffer changes."""
    tensors = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    cached = getattr(bn, "_fsf_affine", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    with torch.no_grad():
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight * invstd if bn.affine else invstd).contiguous()
        shift = ((bn.bias if bn.affine else 0) - bn.running_mean * scale).contiguous()
    bn._fsf_affine = (key, scale, shift)
    return scale, shift


def _bn_act_training(bn, feats, relu):
    from .sst_ops import batch_norm_act_training

    return batch_norm_act_training(bn, feats, relu)


def _is_eval_bn(m):
    return isinstance(m, nn.BatchNorm1d) and not m.training and m.track_running_stats


class SparseSequential(SparseModule):
    """spconv.SparseSequential: dense modules act on `.features`.  Peephole: conv -> eval BN (-> ReLU) becomes one
    fused launch."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for i, m in enumerate(args):
            self.add_module(str(i), m)
        for name, m in kwargs.items():
            self.add_module(name, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseConvolution) and i + 1 < len(mods) and _is_eval_bn(mods[i + 1]) and not torch.is_grad_enabled():
                scale, shift = _bn_affine(mods[i + 1])
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = m(x, scale=scale, shift=shift, relu=relu)
                i += 3 if relu else 2
                continue
            if (isinstance(m, nn.BatchNorm1d) and isinstance(x, SparseConvTensor) and x.indices.size(0) != 0 and m.training
                    and torch.is_grad_enabled()):  # training: BN (+ the ReLU after it) on K23
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                y = _bn_act_training(m, x.features, relu)
                if y is not None:
                    x = x._like(y)
                    i += 2 if relu else 1
                    continue
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.size(0) != 0:
                    x = x._like(m(x.features))
            else:
                x = m(x)
            i += 1
        return x


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type="SubMConv3d", norm_cfg=None, order=("conv", "norm", "act")):
    """mmdet3d.ops.make_sparse_convmodule: SparseSequential(conv, norm, ReLU) in the given order."""
    from ..registry import build_norm_layer

    assert isinstance(order, tuple) and len(order) <= 3 and set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    layers = []
    for layer in order:
        if layer == "conv":
            if conv_type == "SparseInverseConv3d":
                layers.append(SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False))
            else:
                layers.append(CONV_TYPES[conv_type](in_channels, out_channels, kernel_size, stride=stride,
                                                    padding=padding, bias=False, indice_key=indice_key))
        elif layer == "norm":
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return SparseSequential(*layers)


class SparseBasicBlock(SparseModule):
    """mmdet3d.ops.SparseBasicBlock: conv-bn-relu-conv-bn + identity, relu (both convs SubM on one indice_key)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        from ..registry import build_norm_layer

        conv_cfg = dict(conv_cfg or dict(type="SubMConv3d"))
        ctype = CONV_TYPES[conv_cfg.pop("type")]
        self.conv1 = ctype(inplanes, planes, 3, stride=stride, padding=1, bias=False, **conv_cfg)
        self.norm1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = ctype(planes, planes, 3, padding=1, bias=False, **conv_cfg)
        self.norm2 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        assert downsample is None

    def forward(self, x):
        identity = x.features
        if _is_eval_bn(self.norm1) and _is_eval_bn(self.norm2) and not torch.is_grad_enabled():
            s1, b1 = _bn_affine(self.norm1)
            out = self.conv1(x, scale=s1, shift=b1, relu=True)
            s2, b2 = _bn_affine(self.norm2)
            return self.conv2(out, scale=s2, shift=b2, residual=identity, relu=True)
        out = self.conv1(x)
        y = _bn_act_training(self.norm1, out.features, True)
        out = out._like(y if y is not None else self.relu(self.norm1(out.features)))
        out = self.conv2(out)
        y = _bn_act_training(self.norm2, out.features, False)
        return out._like(self.relu((y if y is not None else self.norm2(out.features)) + identity))
