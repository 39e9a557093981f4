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


// ---------------------------------------------------------------------------
// Gather-pack: every packed weight image of a training step in ONE launch.  A packed image (tile / ring / K-split /
// phase-plane layout, forward or input-gradient form) is a permutation of the layer's fp32 weights with zeros in its
// padding, i.e. out[e] = idx[e] ? bf16(w[idx[e] - 1]) : 0 for a table idx the host derives ONCE per layer by pushing index
// patterns through that layer's own pack routine (ops.WeightPrepack) - so this kernel knows nothing about layouts, and
// the 36 pack launches of a step (the weights change every step) become one.  The job list travels by value in the
// kernel arguments.
namespace {
constexpr int GP_MAX = 96;
struct GatherJob { const float* src; const int* idx; unsigned short* dst; long long n; };
struct GatherTable { GatherJob j[GP_MAX]; int first[GP_MAX + 1]; int count; };
constexpr int GP_CHUNK = 2048;  // elements per workgroup

__global__ __launch_bounds__(256) void gather_pack_kernel(GatherTable tab) {
  int lo = 0, hi = tab.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const GatherJob job = tab.j[lo];
  const long long base = (long long)((int)blockIdx.x - tab.first[lo]) * GP_CHUNK;
  // a thread = 8 consecutive elements of the image: two 16-B index loads, eight 4-B gathers (neighbours of an image are
  // mostly a fixed stride apart in the weights: L2 lines shared by the lanes), one 16-B store
  const long long e = base + (long long)threadIdx.x * 8;
  if (e >= job.n) return;
  if (e + 8 <= job.n && ((reinterpret_cast<uintptr_t>(job.idx + e) | reinterpret_cast<uintptr_t>(job.dst + e)) & 15) == 0) {
    const int4 i0 = *reinterpret_cast<const int4*>(job.idx + e), i1 = *reinterpret_cast<const int4*>(job.idx + e + 4);
    const int ii[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ii[k] ? job.src[ii[k] - 1] : 0.f;
    uint4 o;
    o.x = lss_pack_bf2(v[0], v[1]); o.y = lss_pack_bf2(v[2], v[3]);
    o.z = lss_pack_bf2(v[4], v[5]); o.w = lss_pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(job.dst + e) = o;
  } else {
    for (long long t = e; t < job.n && t < e + 8; ++t) {
      const int i = job.idx[t];
      job.dst[t] = i ? lss_f2bf(job.src[i - 1]) : (unsigned short)0;
    }
  }
}
}  // namespace

// jobs: HOST array of `count` records { const float* src; const int32_t* idx; uint16_t* dst; long long n } (device
// pointers; idx entries are 1-based element numbers of src, 0 = a zero of the image)
extern "C" int lss_gather_pack(const void* jobs, int count, void* stream) {
  LSS_CHECK_PTR(jobs);
  if (count <= 0) return LSS_E_SHAPE;
  const GatherJob* in = static_cast<const GatherJob*>(jobs);
  for (int i = 0; i < count; ++i) {
    if (in[i].src == nullptr || in[i].idx == nullptr || in[i].dst == nullptr) return LSS_E_NULL;
    if (in[i].n <= 0 || in[i].n >= (1LL << 31)) return LSS_E_SHAPE;
  }
  hipStream_t st = lss_stream(stream);
  for (int i0 = 0; i0 < count; i0 += GP_MAX) {
    GatherTable tab;
    tab.count = count - i0 < GP_MAX ? count - i0 : GP_MAX;
    long long b = 0;
    for (int i = 0; i < tab.count; ++i) {
      tab.j[i] = in[i0 + i];
      tab.first[i] = (int)b;
      b += (in[i0 + i].n + GP_CHUNK - 1) / GP_CHUNK;
    }
    if (b >= (1LL << 30)) return LSS_E_SHAPE;
    tab.first[tab.count] = (int)b;
    hipLaunchKernelGGL(gather_pack_kernel, dim3((unsigned)b), dim3(256), 0, st, tab);
  }
  return lss_launch_status();
}
