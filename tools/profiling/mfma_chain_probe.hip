// How close may two v_mfma_f32_16x16x32_f16 on the SAME accumulator follow each other on gfx950 before the second one stalls?
// DIST independent accumulators are cycled: a dependent pair is DIST - 1 other MFMAs apart.  Reports shader clocks per MFMA for one wave
// per SIMD and for three (K9d's occupancy).  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain_probe tools/profiling/mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIST>
__global__ void __launch_bounds__(256) chain(float* out, unsigned long long* clocks, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(0.5f - e * 0.01f); }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  // 24 MFMAs per trip, written out: the accumulator of MFMA i is c[i % DIST] (hipcc's own loop shuffled the accumulators through
  // AGPR copies and measured those)
#define M(C) "v_mfma_f32_16x16x32_f16 %" #C ", %8, %9, %" #C "\n"
  for (int i = 0; i < iters; ++i) {
    if (DIST == 1) asm volatile(M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 2) asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 3) asm volatile(M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 4) asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 6) asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 8) asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
                                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    if (DIST == 12)  // K9d's pattern: pairs (tile 0, tile 1), each accumulator again one MFMA later: 0 1 0 1 0 1 | 2 3 2 3 2 3 | ...
      asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(2) M(3) M(2) M(3) M(2) M(3) M(4) M(5) M(4) M(5) M(4) M(5) M(6) M(7) M(6) M(7) M(6) M(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  }
#undef M
  const unsigned long long t1 = __builtin_readcyclecounter();
  const float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

template <int DIST>
static void run(int blocks_per_cu, float* out, unsigned long long* clk) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<DIST>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long* h = new unsigned long long[blocks];
  hipMemcpy(h, clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks; ++i) sum += (double)h[i];
  const double per = sum / blocks / ((double)iters * 24);
  printf("dist %d  waves/SIMD %d : %.1f clocks per MFMA per wave  -> %.1f per SIMD\n", DIST, blocks_per_cu, per, per / blocks_per_cu);
  delete[] h;
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
  hipMalloc(&clk, sizeof(unsigned long long) * 1024);
  for (int w : {1}) {  // (one 4-wave workgroup per CU = one wave per SIMD; more blocks are not spread evenly over the CUs)
    run<1>(w, out, clk); run<2>(w, out, clk); run<3>(w, out, clk); run<4>(w, out, clk); run<6>(w, out, clk); run<8>(w, out, clk); run<12>(w, out, clk);
  }
  return 0;
}
