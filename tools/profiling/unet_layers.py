"""Per-launch times of the U-Net's convolutions inside the real forward (events around every hip_ops conv call, lateral side stream off),
plus the forward's total.  usage (GPU box): FSF_UNET_MASK_ORDER=0|1 python tools/profiling/unet_layers.py"""
import os, sys, time, torch
os.environ.setdefault("FSF_UNET_LATERAL_STREAM", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
pts = inp["points"][0][:, :5].contiguous()
recs = []
def wrap(name):
    orig = getattr(hip_ops, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(*a, **k); e1.record()
        recs.append((name, e0, e1, a))
        return out
    setattr(hip_ops, name, w)
    return orig
with torch.no_grad():
    p_dev, coors = seg.voxelize([pts]); vf, vc, _ = seg.voxel_encoder(p_dev, coors, return_inv=True)
    f = lambda: seg.backbone(dict(voxel_feats=vf, voxel_coors=vc, batch_size=1))
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); total = (time.perf_counter() - t0) / 20 * 1e3
    names = ["spconv_forward_planes", "spconv_forward_split", "spconv_forward", "rulebook_subm", "rulebook_strided", "order_by_neighbor_mask",
             "remap_indices", "to_planes", "channel_group_sum_add"]
    for n in names: wrap(n)
    f(); torch.cuda.synchronize(); recs.clear()
    f(); torch.cuda.synchronize()
print(f"FSF_UNET_MASK_ORDER={os.environ.get('FSF_UNET_MASK_ORDER', '1')}  U-Net forward {total:.3f} ms (serial stream)")
agg = {}
for name, e0, e1, a in recs:
    us = e0.elapsed_time(e1) * 1e3
    agg.setdefault(name, [0, 0.0]); agg[name][0] += 1; agg[name][1] += us
    if name.startswith("spconv_forward_planes"):
        nbr = a[4]; cin = sum(p.c for p in a[0]); print(f"   planes m={nbr.shape[0]:7d} {cin:4d}->{a[3]:4d}  {us:8.1f} us")
for k, (c, us) in agg.items(): print(f"{k:28s} calls {c:3d}  {us:9.1f} us")
