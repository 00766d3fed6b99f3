"""Sub-step timing of the refine stage (sync + wall clock around wrapped methods; 10-sweep frame)."""
import os, sys, time, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(obj, name, g)
with torch.no_grad():
    for _ in range(2): bench.step(model, inp)
    wrap(model, 'combine_frustum_and_fsd'); wrap(model, 'multi_stage_refine_test'); wrap(model, 'decode_stage_bboxes')
    wrap(model.roi_extractor, 'forward', 'roi_extractor'); wrap(model, 'img_cross_attn')
    for i, m in enumerate(model.refine_sir_layers): wrap(model.refine_sir_layers, str(i), f'refine_sir[{i}]') if False else None
    sir0 = model.refine_sir_layers[0]; f0 = sir0.forward
    def sirf(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f0(*a, **k); torch.cuda.synchronize()
        acc['refine_sir'] = acc.get('refine_sir', 0) + (time.perf_counter() - t0) * 1e3; return r
    sir0.forward = sirf
    for nm in ('lidar_img_mlp', 'position_encoder', 'out_proj'):
        mod = getattr(model, nm)[0]; ff = mod.forward
        def mk(ff, nm):
            def h(*a, **k):
                torch.cuda.synchronize(); t0 = time.perf_counter(); r = ff(*a, **k); torch.cuda.synchronize()
                acc[nm] = acc.get(nm, 0) + (time.perf_counter() - t0) * 1e3; return r
            return h
        mod.forward = mk(ff, nm)
    head = model.frustum_refined_head[0]
    hf = head.forward
    def hfw(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = hf(*a, **k); torch.cuda.synchronize()
        acc['refined_head.forward'] = acc.get('refined_head.forward', 0) + (time.perf_counter() - t0) * 1e3; return r
    head.forward = hfw
    wrap(head, 'get_bboxes')
    n = 5
    for _ in range(n): bench.step(model, inp)
for k, v in acc.items(): print(f'{k:32s} {v / n:8.3f} ms')
