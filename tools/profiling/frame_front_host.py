"""Where the host thread is around the frame boundary of the announced loop (K32): usage python tools/profiling/frame_front_host.py [sweeps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from fullysparsefusion_amd import hip_ops  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
pool = [bench.make_inputs(sweeps, seed=j, device=dev)[1] for j in range(4)]
T = []
now = time.perf_counter


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = now()
        try:
            return f(*a, **k)
        finally:
            T.append((tag, t0, now()))

    setattr(obj, name, g)


wrap(model, "_prefetch_front", "prefetch_front")
wrap(model, "_take_front", "take_front")
wrap(model, "simple_test", "simple_test")
wrap(model.segmentor, "extract_feat_begin", "seg.begin")
wrap(model.segmentor, "extract_feat_finish", "seg.finish")
wrap(model.segmentor.voxel_encoder, "forward", "vfe")
wrap(model.segmentor.backbone, "begin", "unet.begin")
wrap(model, "_prefetch_image_branch", "image_branch")
wrap(hip_ops, "unique_rows", "unique_rows")
wrap(hip_ops, "nms_select", "nms_select")
for ann in (True, False):
    for i in range(6):
        bench.step(model, pool[i % 4], False, pool[(i + 1) % 4] if ann else None)
    torch.cuda.synchronize()
    del T[:]
    t_start = now()
    for i in range(6, 9):
        bench.step(model, pool[i % 4], False, pool[(i + 1) % 4] if ann else None)
    torch.cuda.synchronize()
    print(f"# {'announced' if ann else 'unannounced'}: 3 frames in {(now() - t_start) * 1e3:.2f} ms")
    for tag, a, b in sorted(T, key=lambda r: r[1]):
        if tag == "unique_rows" and (b - a) < 50e-6:
            continue
        print(f"{(a - t_start) * 1e6:9.0f} us  +{(b - a) * 1e6:7.0f}  {tag}")
