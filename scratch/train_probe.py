"""fwd+bwd of the hot path with a dummy scalar loss: which parameters receive gradients, timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else 'train'
model.train(mode == 'train')
frame, inp = bench.make_inputs(sweeps, 0, dev)
def loss_of(out):
    s = out['seg']
    return (s['seg_logits'].sum() + s['seg_vote_preds'].sum() + out['frustum_obj_feats'].sum() + out['fsd_obj_feats'].sum())
for it in range(4):
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.forward_hot_path(inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno'])
    loss = loss_of(out)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'iter {it}: fwd {1e3*(t1-t0):.1f} ms bwd {1e3*(t2-t1):.1f} ms loss {float(loss):.4e}')
nog = [n for n, p in model.named_parameters() if p.grad is None]
bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print('params', len(list(model.parameters())), 'without grad', len(nog), 'non-finite', len(bad))
print('no grad:', nog[:40])
print('bad:', bad[:20])
