"""How long the box tail of a frame takes: FrustumClusterHead._get_bboxes_single (sigmoid -> decode -> per-class NMS -> selection) and
bbox3d2result (device -> host), wall clock with a synchronise on either side, inside the real forward.  (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd.mmdet3d_plugin.core import bbox as bb
from fullysparsefusion_amd.mmdet3d_plugin.models.dense_heads import cluster_heads as ch
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
acc = {}
def timed(owner, name, tag):
    orig = getattr(owner, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(tag, []).append((time.perf_counter() - t0) * 1e3)
        return out
    setattr(owner, name, w)
timed(ch.SparseClusterHeadV2, "_get_bboxes_single", "_get_bboxes_single")
import fullysparsefusion_amd.mmdet3d_plugin.models.detectors.fsf as fsfmod
for mod in (fsfmod, ch):
    if hasattr(mod, "bbox3d2result"): timed(mod, "bbox3d2result", f"bbox3d2result@{mod.__name__.split('.')[-1]}")
N = 8
t0 = time.perf_counter()
for _ in range(N): bench.step(model, inp)
print(f"frame (with the extra synchronises) {(time.perf_counter() - t0) / N * 1e3:.2f} ms")
for k, v in acc.items(): print(f"{k:40s} {len(v) / N:.1f} calls/frame  {sum(v) / N:.3f} ms/frame")
