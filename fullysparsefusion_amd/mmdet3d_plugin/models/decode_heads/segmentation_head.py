"""`VoteSegHead` (HEADS): inference path of
projects/mmdet3d_plugin/models/decode_heads/segmentation_head.py:15-104,265-266 — per-point MLP, seg logits and
vote offsets.  Dense GEMMs on rocBLAS via torch (not a HIP deliverable); target
generation and losses (train-time, host-side label assignment) are out of scope for this round."""
from .... import switches
import os

import torch
from torch import nn

from ...ops.sst_ops import PointLinear, build_mlp
from ...registry import HEADS


@HEADS.register_module()
class VoteSegHead(nn.Module):
    def __init__(self, in_channel, num_classes, hidden_dims=[], dropout_ratio=0.5, conv_cfg=dict(type="Conv1d"),
                 norm_cfg=dict(type="naiveSyncBN1d"), act_cfg=dict(type="ReLU"),
                 loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, class_weight=None, loss_weight=1.0),
                 loss_vote=dict(type="L1Loss"), loss_aux=None, ignore_index=255, logit_scale=1, checkpointing=False,
                 init_bias=None, init_cfg=None):
        super().__init__()
        end_channel = hidden_dims[-1] if len(hidden_dims) > 0 else in_channel
        self.channels = end_channel
        self.num_classes = num_classes
        self.dropout_ratio = dropout_ratio
        self.norm_cfg, self.act_cfg, self.ignore_index = norm_cfg, act_cfg, ignore_index
        self.loss_decode_cfg, self.loss_vote_cfg, self.loss_aux_cfg = loss_decode, loss_vote, loss_aux
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio > 0 else None
        self.pre_seg_conv = None
        if len(hidden_dims) > 0:
            self.pre_seg_conv = build_mlp(in_channel, hidden_dims, norm_cfg, act=act_cfg["type"])
        self.use_sigmoid = loss_decode.get("use_sigmoid", False)
        self.bg_label = self.num_classes
        if not self.use_sigmoid:
            self.num_classes += 1
        self.logit_scale = logit_scale
        self.conv_seg = PointLinear(end_channel, self.num_classes)
        self.voting = PointLinear(end_channel, self.num_classes * 3)
        self.checkpointing = checkpointing
        self.init_bias = init_bias
        self.train_cfg = self.test_cfg = None
        self.init_weights()

    def init_weights(self):
        if self.init_bias is not None:
            self.conv_seg.bias.data.fill_(self.init_bias)
        else:
            nn.init.normal_(self.conv_seg.weight, mean=0, std=0.01)
            nn.init.constant_(self.conv_seg.bias, 0)

    def cls_seg(self, feat):
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def forward(self, voxel_feat):
        output = voxel_feat
        if self.pre_seg_conv is not None:
            output = self.pre_seg_conv(voxel_feat)
        both = self._seg_and_vote(output)
        if both is not None:
            return both
        return self.cls_seg(output), self.voting(output)

    def _seg_and_vote(self, feat):
        """Inference on the GPU: `conv_seg` (-> classes + 1) and `voting` (-> 3 per class) read the same features, so the two
        Linears are ONE K22 launch on the stacked weight ([11 + 33, 128] for nuScenes: the library ran each of the two thin
        GEMMs over 3e5 rows in ~160 us); the results are the column blocks of one buffer."""
        from .... import hip_ops
        from ...ops.sst_ops import _SMALL_N_MIN

        if (self.training or (torch.is_grad_enabled() and (feat.requires_grad or self.conv_seg.weight.requires_grad))
                or not feat.is_cuda or feat.dim() != 2 or feat.size(0) < _SMALL_N_MIN or (self.dropout is not None and self.training)):
            return None
        c1, c2 = self.conv_seg.out_features, self.voting.out_features
        ctot = (c1 + c2 + 3) // 4 * 4
        if not hip_ops.linear_norm_act_supported(feat, ctot) or self.conv_seg.bias is None or self.voting.bias is None:
            return None
        params = (self.conv_seg.weight, self.conv_seg.bias, self.voting.weight, self.voting.bias)
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.get("_fsf_stacked")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.zeros((ctot, feat.size(1)), dtype=torch.float32, device=feat.device)
                b = torch.zeros((ctot,), dtype=torch.float32, device=feat.device)
                w[:c1], w[c1:c1 + c2] = self.conv_seg.weight, self.voting.weight
                b[:c1], b[c1:c1 + c2] = self.conv_seg.bias, self.voting.bias
                cache = (key, hip_ops.linear_prepare_weight(w), b)
            self.__dict__["_fsf_stacked"] = cache
        y = hip_ops.linear_norm_act(feat, cache[1], ctot, bias=cache[2])
        return y[:, :c1], y[:, c1:c1 + c2]

    def forward_test(self, inputs, img_metas, test_cfg):
        return self.forward(inputs)

    def forward_train(self, *args, **kwargs):
        raise NotImplementedError("VoteSegHead losses/targets are train-time host glue outside this round's hot path")

    @staticmethod
    def encode_vote_targets(delta):
        return torch.sign(delta) * (delta.abs() ** 0.5)

    @staticmethod
    def decode_vote_targets(preds):
        if preds.is_cuda and preds.dim() == 2 and preds.dtype == torch.float32 and not (torch.is_grad_enabled() and preds.requires_grad):
            # the result in rows padded to a multiple of four floats: pre_voxelize's mean then reads it with float4 lanes
            c = preds.size(1)
            buf = torch.empty((preds.size(0), (c + 3) // 4 * 4), dtype=torch.float32, device=preds.device)
            return torch.mul(preds, preds.abs(), out=buf[:, :c])
        return preds * preds.abs()
