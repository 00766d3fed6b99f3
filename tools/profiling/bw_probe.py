"""What does this box sustain?  Read-only, copy and write-only streams of 0.26 / 1 GiB through torch's own kernels.  (GPU box)"""
import torch
dev = torch.device('cuda:0')
def t(f, it=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
for mb in (261, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    print(f"{mb:5d} MiB: read (sum) {x.numel() * 4 / t(lambda: x.sum()) / 1e12:.2f} TB/s | read (max) {x.numel() * 4 / t(lambda: x.max()) / 1e12:.2f} | "
          f"copy {2 * x.numel() * 4 / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s (r+w) | write (fill) {x.numel() * 4 / t(lambda: y.fill_(1.0)) / 1e12:.2f} TB/s | "
          f"add {3 * x.numel() * 4 / t(lambda: torch.add(x, y, out=y)) / 1e12:.2f} TB/s (2r+w)")
