"""Where the U-Net's forward spends its time with the index plan in line and on its own stream: timed events on the main / plan streams
(SimpleSparseUNet._trace hook), relative to the forward's entry.  (GPU box)"""
import os, sys, time, torch
os.environ.setdefault("FSF_UNET_LATERAL_STREAM", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import switches
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
pts = inp["points"][0][:, :5].contiguous()
with torch.no_grad():
    for on in (False, True):
        switches.UNET_PLAN_STREAM = on
        for _ in range(4): seg.extract_feat([pts], None)
        torch.cuda.synchronize()
        seg.backbone._trace = []
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        h0 = time.perf_counter()
        seg.extract_feat([pts], None)
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"== plan stream {'on' if on else 'off'}: host returns after {(h1 - h0) * 1e3:.3f} ms")
        for tag, e in seg.backbone._trace: print(f"   {e0.elapsed_time(e) * 1e3:8.1f} us  {tag}")
        seg.backbone._trace = None
