from .pipelines import (Collect3D, DevicePointAssembler, Compose, DefaultFormatBundle3D, GlobalRotScaleTrans, LiDARPoints, LoadMaskFromFiles,  # noqa: F401
                        LoadPointsFromFile, LoadPointsFromMultiSweeps, MultiScaleFlipAug3D, MyLoadPointsFromFile,
                        MyLoadPointsFromMultiSweeps, NormalizePoints, PointsRangeFilter, RandomFlip3D, SaveNoAugPoints,
                        frame_to_device)
