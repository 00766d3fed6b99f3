"""`PseudoMiddleEncoderForSpconvFSD` (MIDDLE_ENCODERS): mirror of
projects/mmdet3d_plugin/models/middle_encoders/sst_input_layer_v2.py:15-37 — packs the voxel dict."""
from torch import nn

from ...registry import MIDDLE_ENCODERS


@MIDDLE_ENCODERS.register_module()
class PseudoMiddleEncoderForSpconvFSD(nn.Module):
    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        info = {"voxel_feats": voxel_feats, "voxel_coors": voxel_coors}
        if batch_size is not None:
            info["batch_size"] = batch_size
        return info
