from .bbox import (BasePointBBoxCoder, LiDARInstance3DBoxes, bbox3d2result, box3d_multiclass_nms, nms_gpu,  # noqa: F401
                   nms_normal_gpu, xywhr2xyxyr)
