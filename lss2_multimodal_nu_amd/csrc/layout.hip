// NCHW fp32 <-> NHWC (fp32 | bf16) transposes between the reference's tensor
// layout at the module boundary and the conv path's channels-last activations.
// Per (b): a C x HW matrix transposed through a 64x64 LDS tile so both the
// reads and the writes are row-contiguous.
#include "lss_common.h"

namespace {

constexpr int TS = 64;

// src (B, C, HW) f32 -> dst (B, HW, C) T
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           T* __restrict__ dst, int C, int HW) {
  __shared__ float tile[TS][TS + 1];
  const int b = blockIdx.z, c0 = blockIdx.y * TS, p0 = blockIdx.x * TS;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < TS; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < C && p < HW) ? src[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < TS; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) {
      const float v = tile[tx][i];
      if (sizeof(T) == 2) reinterpret_cast<unsigned short*>(dst)[((size_t)b * HW + p) * C + c] = lss_f2bf(v);
      else reinterpret_cast<float*>(dst)[((size_t)b * HW + p) * C + c] = v;
    }
  }
}

// src (B, HW, C) T -> dst (B, C, HW) f32
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src,
                                                           float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[TS][TS + 1];
  const int b = blockIdx.z, c0 = blockIdx.y * TS, p0 = blockIdx.x * TS;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < TS; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    float v = 0.f;
    if (p < HW && c < C) {
      const size_t o = ((size_t)b * HW + p) * C + c;
      v = (sizeof(T) == 2) ? lss_bf2f(reinterpret_cast<const unsigned short*>(src)[o])
                           : reinterpret_cast<const float*>(src)[o];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < TS; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    if (c < C && p < HW) dst[((size_t)b * C + c) * HW + p] = tile[tx][i];
  }
}

}  // namespace

extern "C" int lss_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dt,
                                    void* stream) {
  LSS_CHECK_PTR(src); LSS_CHECK_PTR(dst);
  LSS_CHECK_POS(B); LSS_CHECK_POS(C); LSS_CHECK_POS(H); LSS_CHECK_POS(W);
  if (B > 65535) return LSS_E_SHAPE;
  const int HW = H * W;
  dim3 grid(lss_cdiv(HW, TS), lss_cdiv(C, TS), B);
  if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<unsigned short>, grid, dim3(256), 0, lss_stream(stream),
                       src, reinterpret_cast<unsigned short*>(dst), C, HW);
  else if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, lss_stream(stream), src,
                       reinterpret_cast<float*>(dst), C, HW);
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}

extern "C" int lss_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int dt,
                                    void* stream) {
  LSS_CHECK_PTR(src); LSS_CHECK_PTR(dst);
  LSS_CHECK_POS(B); LSS_CHECK_POS(C); LSS_CHECK_POS(H); LSS_CHECK_POS(W);
  if (B > 65535) return LSS_E_SHAPE;
  const int HW = H * W;
  dim3 grid(lss_cdiv(HW, TS), lss_cdiv(C, TS), B);
  if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<unsigned short>, grid, dim3(256), 0, lss_stream(stream),
                       reinterpret_cast<const unsigned short*>(src), dst, C, HW);
  else if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, lss_stream(stream),
                       reinterpret_cast<const float*>(src), dst, C, HW);
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}
