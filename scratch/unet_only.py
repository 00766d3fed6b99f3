"""Runs the sparse U-Net forward only (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
with torch.no_grad():
    pts = [inp['points'][0][:, :5].contiguous()]
    bp, coors = seg.voxelize(pts)
    seg.voxel_encoder.max_batch = 1
    vf, vc, inv = seg.voxel_encoder(bp, coors, return_inv=True)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        out = seg.backbone(dict(voxel_feats=vf, voxel_coors=vc, batch_size=1))
torch.cuda.synchronize()
print('ok', out[0]['voxel_feats'].shape)
