"""The segmentor (voxelize -> VFE -> U-Net -> neck) of the 10-sweep frame with the U-Net's index plan on its own stream and in line,
alternating in ONE process; and the U-Net alone.  (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import switches
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
pts = inp["points"][0][:, :5].contiguous()
def t(f, it=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(); torch.cuda.synchronize()   # (a frame ends with a host read-back: synchronise per iteration)
    return (time.perf_counter() - t0) / it * 1e3
with torch.no_grad():
    p_dev, coors = seg.voxelize([pts]); vf, vc, _ = seg.voxel_encoder(p_dev, coors, return_inv=True)
    for rep in range(3):
        for on in (False, True):
            switches.UNET_PLAN_STREAM = on
            a = t(lambda: seg.extract_feat([pts], None))
            b = t(lambda: seg.backbone(dict(voxel_feats=vf, voxel_coors=vc, batch_size=1)))
            print(f"rep {rep} plan stream {'on ' if on else 'off'}: segmentor {a:.3f} ms   U-Net alone {b:.3f} ms")
