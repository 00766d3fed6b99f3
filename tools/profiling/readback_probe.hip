// How long does the host take to learn a count the device just produced, and to get the next kernel running?
//   A: hipMemcpyAsync(device -> pageable host) + hipStreamSynchronize   (what the library does)
//   B: the same into pinned host memory
//   C: the kernel posts (value, sequence number) into mapped pinned host memory, the host spins on the sequence number
// Each iteration: kernel K1 (writes the count) -> read-back -> kernel K2 (depends on the host having the count).  The loop time per
// iteration minus the two kernels' own time is the round trip.  build: hipcc --offload-arch=gfx950 -O2 readback_probe.hip -o /tmp/rb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

struct Mailbox { volatile uint64_t seq; volatile int64_t v; };

__global__ void k1(int64_t* out, int64_t x) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = x; }
__global__ void k1_post(Mailbox* mb, int64_t x, uint64_t seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    mb->v = x;
    __threadfence_system();
    __hip_atomic_store(const_cast<uint64_t*>(&mb->seq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void k2(int64_t* sink, int64_t n) { if (threadIdx.x == 0 && blockIdx.x == 0) *sink += n; }

int main() {
  hipStream_t s; hipStreamCreate(&s);
  int64_t *d, *sink; hipMalloc(&d, 8); hipMalloc(&sink, 8); hipMemset(sink, 0, 8);
  int64_t* pinned; hipHostMalloc(&pinned, 8, hipHostMallocDefault);
  Mailbox* mb; hipHostMalloc(&mb, sizeof(Mailbox), hipHostMallocMapped); mb->seq = 0;
  Mailbox* mb_dev; hipHostGetDevicePointer((void**)&mb_dev, mb, 0);
  const int N = 2000;
  auto run = [&](int mode) {
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    int64_t acc = 0;
    for (int i = 1; i <= N; ++i) {
      int64_t h = 0;
      if (mode == 0) { hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, s, d, (int64_t)i); hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
      else if (mode == 1) { hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, s, d, (int64_t)i); hipMemcpyAsync(pinned, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); h = *pinned; }
      else { const uint64_t seq = mb->seq + 1; hipLaunchKernelGGL(k1_post, dim3(1), dim3(64), 0, s, mb_dev, (int64_t)i, seq);
             while (__atomic_load_n(const_cast<uint64_t*>(&mb->seq), __ATOMIC_ACQUIRE) != seq) { } h = mb->v; }
      acc += h;
      hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, s, sink, h);
    }
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
    printf("mode %d: %.1f us per (kernel, read-back, kernel) iteration  (check %lld)\n", mode, us, (long long)acc);
  };
  for (int rep = 0; rep < 2; ++rep) for (int m = 0; m < 3; ++m) run(m);
  // baseline: the two kernels back to back without any read-back
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 1; i <= N; ++i) { hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, s, d, (int64_t)i); hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, s, sink, (int64_t)i); }
  hipStreamSynchronize(s);
  auto t1 = std::chrono::steady_clock::now();
  printf("no read-back: %.1f us per iteration\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  return 0;
}
