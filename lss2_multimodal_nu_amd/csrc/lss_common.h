// Shared helpers of the gfx950 kernels (device + host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lss_hip.h"

#define LSS_WAVE 64

#define LSS_CHECK_PTR(p) \
  do {                   \
    if ((p) == nullptr) return LSS_E_NULL; \
  } while (0)

#define LSS_CHECK_POS(v) \
  do {                   \
    if ((v) <= 0) return LSS_E_SHAPE; \
  } while (0)

// Status of the launch just enqueued.  hipGetLastError: a returned failure is also CLEARED, so that one failed LSS
// launch (bad configuration, LDS size) does not stay latched in the thread and get reported again by every later
// LSS entry - or by the host framework's next launch check, attributed to the wrong op.  The host
// framework reads its own launches' status right after each of them, so nothing of its is pending here.
static inline int lss_launch_status() {
  hipError_t e = hipGetLastError();
  return (e == hipSuccess || e == hipErrorNotReady) ? 0 : (int)e;  // NotReady: a pending event query, not a launch error
}

static inline hipStream_t lss_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int lss_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// fp32 -> bf16 round-to-nearest-even via the hardware cast (keeps NaN a NaN,
// MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ unsigned short lss_f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float lss_bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned int lss_pack_bf2(float lo, float hi) {
  return (unsigned int)lss_f2bf(lo) | ((unsigned int)lss_f2bf(hi) << 16);
}

__device__ __forceinline__ float lss_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float lss_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
