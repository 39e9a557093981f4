// Producer / consumer primitives of the loader-consumer kernels (conv_ring.hip, conv_wgrad.hip): LDS-DMA pieces,
// counted waits, and FULL / FREE flag words in LDS with BOUNDED polls.  Include inside the file's anonymous namespace
// after defining RK_TIMEOUT_COUNTER (a __device__ int that counts waits which hit their bound; must stay 0).
#pragma once

constexpr int RK_SPIN_LIMIT = 1 << 16;  // bound of every flag wait (~10 ms): a protocol bug must end the grid, not hang it

__device__ __forceinline__ void rk_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void rk_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void rk_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Flag words live in LDS and are addressed through address_space(3) pointers: a generic `volatile int*` made
// hipcc emit flat_load ... sc0 sc1 followed by vmcnt(0) waits (seen in the ISA of the first build).
typedef __attribute__((address_space(3))) volatile int* rk_flag_t;
// wave-uniform read of a flag word
__device__ __forceinline__ int rk_peek(rk_flag_t f) { return __builtin_amdgcn_readfirstlane(*f); }
// wait until *f >= need (bounded)
template <int SLEEP = 1>
__device__ __forceinline__ int rk_wait_ge(rk_flag_t f, int need) {
  int v = rk_peek(f);
  int n = 0;
  while (v < need) {
    __builtin_amdgcn_s_sleep(SLEEP);  // units of 64 cycles
    v = rk_peek(f);
    if (++n > RK_SPIN_LIMIT) {
      if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&RK_TIMEOUT_COUNTER, 1);
      break;
    }
  }
  asm volatile("" ::: "memory");  // nothing that reads the handed-over buffer may move above the wait
  return n;
}
__device__ __forceinline__ void rk_set(rk_flag_t f, int v, int lane) {
  if (lane == 0) *f = v;
}
// FREE counters: one no-return LDS atomic from lane 0 (inline asm: the compiler's atomic optimiser otherwise wraps
// every add in a wave reduction; an LDS operation it does not know of only makes its counted lgkmcnt waits stricter)
__device__ __forceinline__ void rk_add1(rk_flag_t f, int lane) {
  if (lane == 0) {
    const unsigned int addr = (unsigned int)(__UINTPTR_TYPE__)f;
    asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1) : "memory");
  }
}

