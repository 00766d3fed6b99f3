// How much does the ORDER in which a streaming kernel touches a row-major fp32 matrix cost at the HBM?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xprobe tools/profiling/x_pattern_probe.hip && /tmp/xprobe
// pattern 0: K22's operand walk — a wave owns 16 rows, lane (row = lane % 16, g = lane / 16) reads 32 contiguous bytes per
//            32-wide k chunk, chunk after chunk (every 128-byte piece of a row is requested in a different sweep)
// pattern 1: the same bytes, the wave reading each of its rows contiguously (64 lanes x 16 B = 1 KB per instruction)
// pattern 2: pattern 0 with the whole row of a lane requested back to back (all chunks in flight at once)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int K, int PAT>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ x, long n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  float acc = 0.f;
  for (long r0 = wave * 16; r0 + 16 <= n; r0 += nw * 16) {
    if (PAT == 0 || PAT == 2) {
      const float* p = x + (r0 + (lane & 15)) * K + (lane >> 4) * 8;
      if (PAT == 0) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        for (int c = 1; c < K / 32; ++c) {  // one chunk ahead, as the kernel does
          const float4 na = *(const float4*)(p + 32 * c), nb = *(const float4*)(p + 32 * c + 4);
          acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
          a = na; b = nb;
        }
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
      } else {
        float4 a[K / 32], b[K / 32];
#pragma unroll
        for (int c = 0; c < K / 32; ++c) { a[c] = *(const float4*)(p + 32 * c); b[c] = *(const float4*)(p + 32 * c + 4); }
#pragma unroll
        for (int c = 0; c < K / 32; ++c) acc += a[c].x + a[c].y + a[c].z + a[c].w + b[c].x + b[c].y + b[c].z + b[c].w;
      }
    } else {
      constexpr int PER = 16 * K / 256;  // 16-byte loads per lane for the wave's 16 rows
      float4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = *(const float4*)(x + r0 * K + (long)i * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  if (acc == 12345.678f) out[wave] = acc;  // keep the loads
}

template <int K, int PAT>
static void run(const float* x, long n, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {512, 768, 1024, 2048}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<K, PAT>), dim3(grid), dim3(256), 0, 0, x, n, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<K, PAT>), dim3(grid), dim3(256), 0, 0, x, n, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k=%d pattern %d grid %4d: %7.1f us  %6.0f GB/s\n", K, PAT, grid, ms * 100, (double)n * K * 4 / (ms / 10 * 1e-3) / 1e9);
  }
}

int main() {
  const long n = 510000 / 16 * 16;
  float *x, *out;
  hipMalloc(&x, (size_t)n * 256 * 4); hipMalloc(&out, 1 << 20);
  hipMemset(x, 0, (size_t)n * 256 * 4);
  run<256, 0>(x, n, out); run<256, 2>(x, n, out); run<256, 1>(x, n, out);
  run<128, 0>(x, n, out); run<128, 2>(x, n, out); run<128, 1>(x, n, out);
  return 0;
}
