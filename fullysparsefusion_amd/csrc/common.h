// Shared device/host helpers for libfsf_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/fsf_hip.h"

#define FSF_WAVE 64

#define FSF_HIP_TRY(expr)                      \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return FSF_ERR_HIP;  \
  } while (0)

// Every place the library makes the HOST wait for the stream goes through this: the count is what bench.py / sync_sites.py report
// (fsf_get_option(FSF_OPT_HOST_WAITS)).
namespace fsf {
extern std::atomic<int64_t> g_host_waits;
}
#define FSF_STREAM_WAIT(stream)                                        \
  do {                                                                 \
    fsf::g_host_waits.fetch_add(1, std::memory_order_relaxed);         \
    FSF_HIP_TRY(hipStreamSynchronize(stream));                         \
  } while (0)

// A few words the device just produced -> the host, through the calling thread's pinned mailbox (readback.hip); counts as a host wait.
namespace fsf {
int fsf_read_back(void* host_dst, const void* dev_src, size_t bytes, hipStream_t stream);
}
#define FSF_READ_BACK(host_dst, dev_src, bytes, stream)                              \
  do {                                                                               \
    const int _rb = fsf::fsf_read_back((host_dst), (dev_src), (bytes), (stream));    \
    if (_rb != FSF_OK) return _rb;                                                   \
  } while (0)

#define FSF_LAUNCH_CHECK()                               \
  do {                                                   \
    if (hipPeekAtLastError() != hipSuccess) return FSF_ERR_HIP; \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel, device).
// `done` is a per-call-site bit mask (bit = device ordinal; devices >= 64 set it on every launch).  Two host threads may
// race to set the same bit: both set the same attribute value, which is harmless.
static inline hipError_t fsf_set_max_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// ---- GELU(erf), the one form every kernel uses (K21, K22 / K22s / K22h epilogues, fsf_norm_act, fsf_rows_to_planes) ----------------
// GELU(y) = y Phi(y) = max(y, 0) - t E(t)   with  t = |y| / sqrt 2,  E(t) = erfc(t) / sqrt 2 = 2^P(t).
// P = the degree-8 polynomial fit of log2(erfc t) - 1/2 on [0, 4.5], minimax in the error it causes in GELU (tools/fit_gelu_poly.py:
// |GELU - float64| <= 5e-8 beyond the final rounding over [-8, 8]; the Abramowitz-Stegun 7.1.26 form it replaces: 2.2e-7).  P falls
// monotonically (<= -28.9 from t = 4.2 on, -inf at fp32 overflow), so there is no range clamp: 2^P underflows to 0 and GELU = max(y, 0).
// ONE transcendental per value (v_exp_f32, quarter rate) where the rational form needed two (v_rcp_f32 + v_exp_f32), everything else
// runs on pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the scalar forms' rate, two values): ~11 full-rate slots per value
// against ~15.5.  The scalar and the paired form are the same IEEE operations in the same order: bit-identical.
typedef float fsf_f32x2 __attribute__((ext_vector_type(2)));
#define FSF_GELU_P0 -0.49999886751174927f
#define FSF_GELU_P1 -1.6279340982437134f
#define FSF_GELU_P2 -0.9182308912277222f
#define FSF_GELU_P3 -0.14909829199314117f
#define FSF_GELU_P4 0.029333580285310745f
#define FSF_GELU_P5 -0.0018291344167664647f
#define FSF_GELU_P6 -0.0009156854939647019f
#define FSF_GELU_P7 0.0002837815263774246f
#define FSF_GELU_P8 -2.704608596104663e-05f
__device__ __forceinline__ float fsf_gelu(float y) {
  const float t = fabsf(y) * 0.70710678118654752440f;
  float p = fmaf(FSF_GELU_P8, t, FSF_GELU_P7);
  p = fmaf(p, t, FSF_GELU_P6);
  p = fmaf(p, t, FSF_GELU_P5);
  p = fmaf(p, t, FSF_GELU_P4);
  p = fmaf(p, t, FSF_GELU_P3);
  p = fmaf(p, t, FSF_GELU_P2);
  p = fmaf(p, t, FSF_GELU_P1);
  p = fmaf(p, t, FSF_GELU_P0);
  return fmaxf(y, 0.0f) - t * __builtin_amdgcn_exp2f(p);
}
__device__ __forceinline__ fsf_f32x2 fsf_gelu2(fsf_f32x2 y) {
  const fsf_f32x2 t = fsf_f32x2{fabsf(y.x) * 0.70710678118654752440f, fabsf(y.y) * 0.70710678118654752440f};
  auto k = [](float v) { return fsf_f32x2{v, v}; };
  fsf_f32x2 p = __builtin_elementwise_fma(k(FSF_GELU_P8), t, k(FSF_GELU_P7));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P6));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P5));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P4));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P3));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P2));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P1));
  p = __builtin_elementwise_fma(p, t, k(FSF_GELU_P0));
  const fsf_f32x2 e = fsf_f32x2{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  return fsf_f32x2{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)} - t * e;
}

static inline int64_t fsf_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int fsf_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline int fsf_stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// Bump allocator over the caller's workspace (256-byte aligned slices).
struct FsfArena {
  char* base;
  int64_t size;
  int64_t used;
  __host__ FsfArena(void* p, int64_t n) : base((char*)p), size(n), used(0) {}
  template <typename T>
  __host__ T* take(int64_t count) {
    int64_t bytes = fsf_align_up((int64_t)sizeof(T) * (count > 0 ? count : 1), 256);
    if (base == nullptr || used + bytes > size) {
      used = size + 1;  // poison
      return nullptr;
    }
    T* r = (T*)(base + used);
    used += bytes;
    return r;
  }
  __host__ bool ok() const { return used <= size; }
};

__device__ __forceinline__ int fsf_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float fsf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename T>
__device__ __forceinline__ T fsf_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Inclusive scan across the 64 lanes of a wave.
template <typename T>
__device__ __forceinline__ T fsf_wave_inclusive_scan(T v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
