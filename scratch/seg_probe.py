"""Sub-step timing of the segmentor (sync + wall clock) and the rulebook share of the U-Net."""
import os, sys, time, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
acc = collections.OrderedDict()
def wrapf(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3; return r
    object.__setattr__(obj, name, g)
seg = model.segmentor
with torch.no_grad():
    for _ in range(2): bench.step(model, inp)
    wrapf(seg, 'voxelize', 'voxelize'); wrapf(seg.voxel_encoder, 'forward', 'voxel_encoder (VFE)')
    wrapf(seg.backbone, 'forward', 'backbone (U-Net)'); wrapf(seg.decode_neck if hasattr(seg, 'decode_neck') and seg.decode_neck is not None else seg, 'forward' if hasattr(seg, 'decode_neck') and seg.decode_neck is not None else 'voxelize', 'neck')
    wrapf(seg.segmentation_head, 'forward_test', 'seg head') if hasattr(seg.segmentation_head, 'forward_test') else None
    for nm in ('rulebook_subm', 'rulebook_strided', 'spconv_forward', 'spconv_transpose_weight'):
        if hasattr(hip_ops, nm): wrapf(hip_ops, nm, '  hip_ops.' + nm)
    wrapf(model, 'segmentor_feat_inhance_test', 'image fusion + seg head (FSF)')
    wrapf(seg, 'simple_test', 'segmentor.simple_test')
    n = 5
    for _ in range(n): bench.step(model, inp)
for k, v in acc.items(): print(f'{k:40s} {v / n:8.3f} ms')
