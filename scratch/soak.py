"""Soak: N full forwards on the same frame must give bit-identical boxes (threaded branches, in-kernel folds, work queues)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
with torch.no_grad():
    ref = bench.step(model, inp)[0]
    bad = 0
    for i in range(n):
        r = bench.step(model, inp)[0]
        same = (torch.equal(r['boxes_3d'].tensor, ref['boxes_3d'].tensor) and torch.equal(r['scores_3d'], ref['scores_3d'])
                and torch.equal(r['labels_3d'], ref['labels_3d']))
        if not same:
            bad += 1
            print('MISMATCH at iteration', i, float((r['boxes_3d'].tensor - ref['boxes_3d'].tensor).abs().max()) if r['boxes_3d'].tensor.shape == ref['boxes_3d'].tensor.shape else 'shape')
print(f'soak: {n} forwards, {bad} mismatches, boxes {tuple(ref["boxes_3d"].tensor.shape)}')
