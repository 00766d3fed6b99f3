"""Known-byte kernels for calibrating FETCH_SIZE / WRITE_SIZE in the same rocprofv3 --pmc pass as the workload:
a 1 GiB device-to-device elementwise copy (reads 1 GiB, writes 1 GiB; larger than the 256 MiB Infinity Cache) and a
row gather of 512-byte rows through a random permutation (reads 1 GiB of rows + 8 MiB of indices, writes 1 GiB)."""
import torch
dev = torch.device('cuda:0')
n = 1 << 28  # fp32 elements = 1 GiB
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)
rows = src.view(-1, 128)
perm = torch.randperm(rows.size(0), device=dev)
out = torch.empty_like(rows)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)                                  # at::native::vectorized copy kernel
for _ in range(3):
    torch.index_select(rows, 0, perm, out=out)      # at::native index/gather kernel, 512 B rows
torch.cuda.synchronize()
print("calibration done: copy = 1 GiB read + 1 GiB written per launch; gather = 1 GiB + 8 MiB read, 1 GiB written")
