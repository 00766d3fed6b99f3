// Does the lane -> address mapping of K9c's row gather cost vector-memory issue time?  (GPU box, L2-resident table)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gprobe tools/profiling/gather_pattern_probe.hip && /tmp/gprobe
// A wave fetches 16 random 512-byte rows (128 channels of hi | lo planes) chunk by chunk (128 B per row and chunk), 2 x dwordx4 per lane:
//   pattern 0 (K9c / K9d): lane (j = lane % 16, q = lane / 16) reads bytes [32 q, 32 q + 16) and [32 q + 16, 32 q + 32) of row j's chunk:
//                          consecutive lanes are on DIFFERENT rows, every instruction touches 16 lines and uses half of each
//   pattern 1 (row-major): lane l reads piece l % 8 of row l / 8 (first load: rows 0-7, second: rows 8-15): 8 consecutive lanes = one
//                          full 128-byte line
//   pattern 2: 1 KB contiguous per instruction (the weight-fragment stream), same bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PAT>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ tab, const int* __restrict__ idx, int nidx, int iters, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned acc = 0;
  int cur = (wave * 977) % nidx;
  for (int it = 0; it < iters; ++it) {
    cur = (cur + 16) % (nidx - 16);
    uint4 a[4], b[4];
    if (PAT == 0) {
      const int row = idx[cur + (lane & 15)];
      const char* p = tab + (size_t)row * 512 + (lane >> 4) * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = *(const uint4*)(p + c * 128); b[c] = *(const uint4*)(p + c * 128 + 16); }
    } else if (PAT == 1) {
      const int r0 = idx[cur + (lane >> 3)], r1 = idx[cur + 8 + (lane >> 3)];
      const char* p0 = tab + (size_t)r0 * 512 + (lane & 7) * 16;
      const char* p1 = tab + (size_t)r1 * 512 + (lane & 7) * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = *(const uint4*)(p0 + c * 128); b[c] = *(const uint4*)(p1 + c * 128); }
    } else if (PAT == 3) {  // 1 KB contiguous, the SAME addresses in all four waves of the workgroup at the same time (L1 hits for three of them)
      const int row = idx[(blockIdx.x * 131 + it * 16) % (nidx - 16)];
      const char* p = tab + (size_t)(row & ~15) * 512 + lane * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = *(const uint4*)(p + c * 2048); b[c] = *(const uint4*)(p + c * 2048 + 1024); }
    } else if (PAT == 4) {  // as 3, but every workgroup of the chip reads the same 8 KB (all L1 hits after the first touch)
      const char* p = tab + (size_t)(it & 7) * 8192 + lane * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = *(const uint4*)(p + c * 2048); b[c] = *(const uint4*)(p + c * 2048 + 1024); }
    } else {
      const int row = idx[cur];
      const char* p = tab + (size_t)(row & ~15) * 512 + lane * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) { a[c] = *(const uint4*)(p + c * 2048); b[c] = *(const uint4*)(p + c * 2048 + 1024); }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc += a[c].x ^ a[c].w ^ b[c].y ^ b[c].z;
  }
  if (acc == 0x12345678u) out[wave] = (float)acc;
}

template <int PAT>
static void run(const char* tab, const int* idx, int nidx, float* out, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;
  for (int grid : {256, 512, 768, 1024}) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<PAT>), dim3(grid), dim3(256), 0, 0, tab, idx, nidx, iters, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<PAT>), dim3(grid), dim3(256), 0, 0, tab, idx, nidx, iters, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * iters * 8192.0;
    printf("%-28s grid %4d: %8.1f us  %7.0f GB/s  (%5.1f B/clk/CU at 2.4 GHz)\n", what, grid, ms * 200, bytes / (ms / 5 * 1e-3) / 1e9,
           bytes / (ms / 5 * 1e-3) / 256 / 2.4e9);
  }
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 8192;  // x 512 B: 4 MB by default
  char* tab; int* idx; float* out;
  hipMalloc(&tab, (size_t)rows * 512); hipMalloc(&out, 1 << 20);
  hipMemset(tab, 1, (size_t)rows * 512);
  const int nidx = 1 << 16;
  std::vector<int> h(nidx);
  srand(1);
  for (int i = 0; i < nidx; ++i) h[i] = rand() % rows;
  hipMalloc(&idx, nidx * 4); hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
  printf("table %d rows x 512 B = %.1f MB\n", rows, rows * 512 / 1e6);
  run<0>(tab, idx, nidx, out, "0: lane (j, q), 32-B pieces");
  run<1>(tab, idx, nidx, out, "1: row-major, 8 lanes / line");
  run<2>(tab, idx, nidx, out, "2: 1 KB contiguous");
  run<3>(tab, idx, nidx, out, "3: 1 KB, 4 waves same addr");
  run<4>(tab, idx, nidx, out, "4: 64 KB hot set, all WGs");
  return 0;
}
