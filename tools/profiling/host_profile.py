"""cProfile of the host side of one frame (sequential branches): where does the Python time go?  (GPU box)  usage: host_profile.py [sweeps]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 10, 0, dev)
for _ in range(3): bench.step(model, inp)
pr = cProfile.Profile()
pr.enable()
for _ in range(3): bench.step(model, inp)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime')
import io
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('tottime').print_stats(45)
print(buf.getvalue()[:9000])
