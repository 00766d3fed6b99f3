// The ceiling of K9d's SHAPE ("a micro-benchmark that shows the ceiling of this gather -> LDS -> MFMA shape at this
// occupancy"): the chunk-granular pipeline of csrc/spconv_planes.hip::spconv_fwd_pipe_kernel<2, 4> with everything that is specific to a
// sparse convolution taken out — no neighbour table, no cell skipping (every cell live), no per-row scales, no epilogue, rows gathered
// from a window that moves with the block (the L2 locality a voxel neighbourhood has) — and its five ingredients switchable:
//   iteration i (one 32-cin chunk of one kernel offset, 64 rows x 128 output channels per workgroup, 4 waves):
//     [barrier]  stage X(i+2): 2 x ds_write_b128 per lane;  gather X(i+3): 2 x global_load_dwordx4 per lane (16 rows x 128 B per wave);
//     W(i+1): 4 x global_load_dwordx4 per lane (this wave's 32 channels x 32 cin, hi | lo) into the other register set;
//     4 cells x (6 x v_mfma_f32_16x16x32_f16 + 2 x ds_read_b128 of the cell's fragments for chunk i+1).
// Three workgroups per CU (256 threads, 44 KB LDS each, launch bound 3 waves per SIMD) like the product kernel.
// Reported per variant: time, shader clocks per iteration of one workgroup, issued f16 TFLOP/s and its third (the fp32-equivalent rate the
// product's roofline is priced in) against 2500 / 3 = 833.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/k9d_shape_probe tools/profiling/k9d_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int kgroup(int q) { return (0x9C >> (2 * q)) & 3; }

template <bool G, bool W, bool L, bool M>
__global__ void __launch_bounds__(256, 3)
    shape_kernel(const uint4* __restrict__ x, unsigned m_rows, const uint4* __restrict__ w, int blocks_per_wg, int kvol, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* xring = reinterpret_cast<uint4*>(smem);  // 4 slots x 4 cells x 128 pieces x 16 B = 32 KB (+ 12 KB unused: the table / vectors)
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grow = lane >> 3, gpiece = lane & 7;
  const int wr_a = grow * 8 + (gpiece ^ ((grow >> 1) & 7)), wr_b = (8 + grow) * 8 + (gpiece ^ (((8 + grow) >> 1) & 7));
  const int rd_hi = j * 8 + ((2 * kgroup(q)) ^ ((j >> 1) & 7)), rd_lo = j * 8 + ((2 * kgroup(q) + 1) ^ ((j >> 1) & 7));
  f32x4 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 wA[2][2], wB[2][2], xh[4], xl[4], g_a, g_b;
  g_a = g_b = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int p = 0; p < 2; ++p) wA[t][p] = wB[t][p] = make_uint4(0x38003800u + tid, 0x38003800u, 0x38003800u, 0x38003800u);
#pragma unroll
  for (int g = 0; g < 4; ++g) xh[g] = xl[g] = make_uint4(0x3c003c00u, 0x34003400u + lane, 0x3c003c00u, 0x3c003c00u);
  const int iters = kvol * 4;
  for (int b = 0; b < blocks_per_wg; ++b) {
    const unsigned row0 = ((unsigned)blockIdx.x * (unsigned)blocks_per_wg + (unsigned)b) * 64u;
    auto iteration = [&](auto odd_tag, int i) {  // (the weight set's parity is a compile-time constant, as in the product kernel)
      constexpr bool ODD = decltype(odd_tag)::value;
      const int k = i >> 2, kc = i & 3;
      __syncthreads();
      if (L) {  // stage the chunk gathered during the previous iteration
        uint4* dst = xring + (((i + 2) & 3) * 4 + wave) * 128;
        dst[wr_a] = g_a;
        dst[wr_b] = g_b;
      }
      if (G) {  // gather: piece `gpiece` of rows grow and 8 + grow of this wave's cell, offset k's neighbours = a window that moves with the block
        const unsigned r0 = (row0 + 16u * wave + grow + 97u * k) % m_rows, r1 = (row0 + 16u * wave + 8u + grow + 97u * k + 13u) % m_rows;
        g_a = x[(size_t)r0 * 32 + kc * 8 + gpiece];
        g_b = x[(size_t)r1 * 32 + kc * 8 + gpiece];
      }
      if (W) {  // the next chunk's weight fragments into the set the previous iteration multiplied from
        const uint4* p = w + ((size_t)(k * 4 + ((kc + 1) & 3)) * 4 + wave) * 256 + lane;
        if (ODD) { wA[0][0] = p[0]; wA[0][1] = p[64]; wA[1][0] = p[128]; wA[1][1] = p[192]; }
        else { wB[0][0] = p[0]; wB[0][1] = p[64]; wB[1][0] = p[128]; wB[1][1] = p[192]; }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (M) {
          const f16x8 bh = __builtin_bit_cast(f16x8, xh[g]), bl = __builtin_bit_cast(f16x8, xl[g]);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f16x8 wh = __builtin_bit_cast(f16x8, ODD ? wB[t][0] : wA[t][0]), wl = __builtin_bit_cast(f16x8, ODD ? wB[t][1] : wA[t][1]);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, acc[g][t], 0, 0, 0);
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, acc[g][t], 0, 0, 0);
          }
        } else {
          acc[g][0][0] += __uint_as_float(xh[g].x ^ wA[0][0].x ^ wB[1][1].y ^ xl[g].y);
        }
        if (L) {
          const uint4* xs = xring + (((i + 1) & 3) * 4 + g) * 128;
          xh[g] = xs[rd_hi];
          xl[g] = xs[rd_lo];
        } else if (G) {
          xh[g].x ^= g_a.x; xl[g].y ^= g_b.y;
        }
      }
    };
    for (int i = 0; i < iters; i += 2) {
      iteration(std::false_type{}, i);
      iteration(std::true_type{}, i + 1);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) s += acc[g][0][0] + acc[g][1][1] + acc[g][0][2] + acc[g][1][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <bool G, bool W, bool L, bool M>
static void run(const char* what, const uint4* x, unsigned m_rows, const uint4* w, float* out, double mhz) {
  const int wgs = 768, blocks_per_wg = 2, kvol = 27;
  const size_t lds = 44 * 1024;
  hipFuncSetAttribute((const void*)shape_kernel<G, W, L, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((shape_kernel<G, W, L, M>), dim3(wgs), dim3(256), lds, 0, x, m_rows, w, blocks_per_wg, kvol, out);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((shape_kernel<G, W, L, M>), dim3(wgs), dim3(256), lds, 0, x, m_rows, w, blocks_per_wg, kvol, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double iters_per_wg = (double)blocks_per_wg * kvol * 4;
  const double clk_per_iter = us * mhz / iters_per_wg;           // one workgroup's iteration (three run interleaved on its CU)
  const double f16_flops = (double)wgs * iters_per_wg * 24 * 4 * 16384.0;  // 24 MFMAs per wave and iteration, 4 waves, 16 x 16 x 32 x 2 each
  const double tf = f16_flops / (us * 1e-6) / 1e12;
  printf("%-44s %8.1f us  %7.0f clk / iteration  %7.1f TFLOP/s f16 issued = %6.1f fp32-eq = %.3f of 833\n", what, us, clk_per_iter, M ? tf : 0.0,
         M ? tf / 3 : 0.0, M ? tf / 3 / 833.3 : 0.0);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double mhz = prop.clockRate / 1e3;
  const unsigned m_rows = 101119;  // the 0.4 m level of the 10-sweep frame: 128-channel planes, 512 B per row
  uint4 *x, *w;
  float* out;
  hipMalloc(&x, (size_t)m_rows * 512);
  hipMalloc(&w, (size_t)27 * 4 * 4 * 256 * 16);
  hipMalloc(&out, sizeof(float) * 768 * 256);
  hipMemset(x, 0x3c, (size_t)m_rows * 512);
  hipMemset(w, 0x38, (size_t)27 * 4 * 4 * 256 * 16);
  printf("# %s, %d CUs, shader clock %.0f MHz; 768 workgroups x 2 row blocks x 27 offsets x 4 chunks, 3 workgroups per CU\n", prop.name,
         prop.multiProcessorCount, mhz);
  run<true, true, true, true>("full shape (gather + weights + LDS + MFMA)", x, m_rows, w, out, mhz);
  run<false, true, true, true>("without the gather", x, m_rows, w, out, mhz);
  run<true, false, true, true>("without the weight loads", x, m_rows, w, out, mhz);
  run<false, false, true, true>("without both global streams", x, m_rows, w, out, mhz);
  run<false, false, false, true>("MFMAs + barrier only (operands in registers)", x, m_rows, w, out, mhz);
  run<true, true, true, false>("everything but the MFMAs", x, m_rows, w, out, mhz);
  run<true, true, false, true>("global streams + MFMAs, no LDS round trip", x, m_rows, w, out, mhz);
  return 0;
}
