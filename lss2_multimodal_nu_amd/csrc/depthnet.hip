// K2: CamEncode's 1x1 depthnet conv + depth softmax (ref: src/modules.py:82-83)
// as one MFMA kernel.  out^T[n, pix] = sum_k W[n,k] * x[bn,k,pix] + bias[n].
//
// Work split: one 256-thread workgroup per 16 pixels of one camera image; its 4
// waves split K four ways, each wave accumulating all NT 16-row output tiles
// (NT = ceil((D+C)/16)) for its K quarter on v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains).  Partials meet in LDS, then bias, softmax over the first D rows,
// and the two outputs are written in the layouts the splat kernels want:
//   depth (BN, D, HW)   - same as the reference's `depth` tensor
//   feat  (BN*HW, C)    - channels-last, one 256-B row per pixel at C = 64
// The (B*N, C, D, fH, fW) lifted tensor of ref :84 is never written.
#include "depthnet_body.h"

namespace {

using namespace lss_depthnet;

template <int NT>
__global__ __launch_bounds__(256) void depthnet_softmax_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    int Cin, int HW, int D, int C, float* __restrict__ depth, float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  depthnet_softmax_f32_body<NT>(x, w, bias, Cin, HW, D, C, depth, feat, blockIdx.x, blockIdx.y, lds);
}

// bf16 variant: x and W are rounded to bf16 in registers (inputs stay fp32 in
// HBM), products accumulate in fp32 on v_mfma_f32_16x16x32_bf16.
//   A[row = l&15][k = 8*(l>>4) + e],  B[k = 8*(l>>4) + e][col = l&15],  e = 0..7
template <int NT>
__global__ __launch_bounds__(256) void depthnet_softmax_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    int Cin, int HW, int D, int C, float* __restrict__ depth, float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int bn = blockIdx.y;
  const int pix0 = blockIdx.x * PIX;
  const int NO = D + C;
  const int pix = min(pix0 + col, HW - 1);
  const int kq = Cin >> 2;
  const float* xb = x + ((size_t)bn * Cin + (size_t)wave * kq) * HW + pix;
  const float* wb = w + (size_t)wave * kq + 8 * j;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = 0; kb < kq; kb += 32) {
    bf16x8 bx;
#pragma unroll
    for (int e = 0; e < 8; ++e) bx[e] = (short)lss_f2bf(xb[(size_t)(kb + 8 * j + e) * HW]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + col;
      bf16x8 wa = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (n < NO) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          wa[e] = (short)lss_f2bf(w0[e]);
          wa[4 + e] = (short)lss_f2bf(w1[e]);
        }
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx, acc[t], 0, 0, 0);
    }
  }

  depthnet_epilogue<NT>(acc, lds, bias, bias + D, D, true, bn, pix0, HW, D, C, depth, feat);
}

// ---------------------------------------------------------------------------
// Two-source variant for the vovnet models (ref: src/model_vovnet_transformer.py:73-122):
// the depth logits come from a 1x1 conv over the depth head's hidden activations
// (NHWC, fp32 or bf16, as the MFMA conv kernel leaves them) and the context features
// from `feat_proj`, a 1x1 conv over the trunk's C3 map (NCHW fp32).  Same work split
// and epilogue as above; depth tiles are rows [0, 16*NTD), context tiles follow.
template <typename T>
__device__ __forceinline__ f32x4 load_k4(const T* p);
template <>
__device__ __forceinline__ f32x4 load_k4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load_k4<unsigned short>(const unsigned short* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return (f32x4){lss_bf2f((unsigned short)v[0]), lss_bf2f((unsigned short)v[1]),
                 lss_bf2f((unsigned short)v[2]), lss_bf2f((unsigned short)v[3])};
}

template <int NTD, int NTC, typename T>
__global__ __launch_bounds__(256) void camencode_v2_kernel(
    const T* __restrict__ xd, const float* __restrict__ wd, const float* __restrict__ bd, int Cd,
    const float* __restrict__ xf, const float* __restrict__ wf, const float* __restrict__ bf, int Cf,
    int HW, int D, int C, int softmax, float* __restrict__ depth, float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int bn = blockIdx.y;
  const int pix0 = blockIdx.x * PIX;
  const int pix = min(pix0 + col, HW - 1);
  constexpr int NT = NTD + NTC;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  {  // depth logits: B[k][pix] = xd[pixel row][k], one 4-channel load per lane and K block
    const int kq = Cd >> 2;
    const T* xb = xd + ((size_t)bn * HW + pix) * Cd + (size_t)wave * kq + 4 * j;
    const float* wb = wd + (size_t)wave * kq + 4 * j;
    for (int kb = 0; kb < kq; kb += 16) {
      const f32x4 xs = load_k4<T>(xb + kb);
#pragma unroll
      for (int t = 0; t < NTD; ++t) {
        const int n = 16 * t + col;
        f32x4 wa = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n < D) wa = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cd + kb);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xs[s], acc[t], 0, 0, 0);
      }
    }
  }
  if (NTC > 0) {  // context features from the NCHW trunk map
    const int kq = Cf >> 2;
    const float* xb = xf + ((size_t)bn * Cf + (size_t)wave * kq) * HW + pix;
    const float* wb = wf + (size_t)wave * kq + 4 * j;
    for (int kb = 0; kb < kq; kb += 16) {
      float xs[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) xs[s] = xb[(size_t)(kb + 4 * j + s) * HW];
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        const int n = 16 * t + col;
        f32x4 wa = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n < C) wa = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cf + kb);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[NTD + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xs[s], acc[NTD + t], 0, 0, 0);
      }
    }
  }
  depthnet_epilogue<NT>(acc, lds, bd, bf, 16 * NTD, softmax != 0, bn, pix0, HW, D, NTC > 0 ? C : 0,
                        depth, feat);
}

// bf16-MFMA variant of the two-source kernel (conv path in bf16): operands rounded to bf16 in
// registers (the hidden map usually IS bf16), fp32 accumulation on v_mfma_f32_16x16x32_bf16 -
// 8x fewer MFMA issues than the f32 form, which at these tiny GEMMs is what the time is.
template <typename T>
__device__ __forceinline__ bf16x8 load_k8_bf16(const T* p);
template <>
__device__ __forceinline__ bf16x8 load_k8_bf16<unsigned short>(const unsigned short* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
template <>
__device__ __forceinline__ bf16x8 load_k8_bf16<float>(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r[e] = (short)lss_f2bf(a[e]);
    r[4 + e] = (short)lss_f2bf(b[e]);
  }
  return r;
}
__device__ __forceinline__ bf16x8 w_k8_bf16(const float* p) { return load_k8_bf16<float>(p); }

template <int NTD, int NTC, typename T>
__global__ __launch_bounds__(256) void camencode_v2_bf16_kernel(
    const T* __restrict__ xd, const float* __restrict__ wd, const float* __restrict__ bd, int Cd,
    const float* __restrict__ xf, const float* __restrict__ wf, const float* __restrict__ bf, int Cf,
    int HW, int D, int C, int softmax, float* __restrict__ depth, float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int bn = blockIdx.y;
  const int pix0 = blockIdx.x * PIX;
  const int pix = min(pix0 + col, HW - 1);
  constexpr int NT = NTD + NTC;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  {  // depth logits from the NHWC hidden map: A[n][k = 8j + e], B[k = 8j + e][pix]
    const int kq = Cd >> 2;
    const T* xb = xd + ((size_t)bn * HW + pix) * Cd + (size_t)wave * kq + 8 * j;
    const float* wb = wd + (size_t)wave * kq + 8 * j;
    for (int kb = 0; kb < kq; kb += 32) {
      const bf16x8 bx = load_k8_bf16<T>(xb + kb);
#pragma unroll
      for (int t = 0; t < NTD; ++t) {
        const int n = 16 * t + col;
        const bf16x8 wa = n < D ? w_k8_bf16(wb + (size_t)n * Cd + kb) : zero8;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx, acc[t], 0, 0, 0);
      }
    }
  }
  if (NTC > 0) {  // context features from the NCHW trunk map
    const int kq = Cf >> 2;
    const float* xb = xf + ((size_t)bn * Cf + (size_t)wave * kq) * HW + pix;
    const float* wb = wf + (size_t)wave * kq + 8 * j;
    for (int kb = 0; kb < kq; kb += 32) {
      bf16x8 bx;
#pragma unroll
      for (int e = 0; e < 8; ++e) bx[e] = (short)lss_f2bf(xb[(size_t)(kb + 8 * j + e) * HW]);
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        const int n = 16 * t + col;
        const bf16x8 wa = n < C ? w_k8_bf16(wb + (size_t)n * Cf + kb) : zero8;
        acc[NTD + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx, acc[NTD + t], 0, 0, 0);
      }
    }
  }
  depthnet_epilogue<NT>(acc, lds, bd, bf, 16 * NTD, softmax != 0, bn, pix0, HW, D, NTC > 0 ? C : 0,
                        depth, feat);
}

// MultiScaleDepthNet tail (ref: src/model_vovnet_transformer.py:61-70): bilinear
// (align_corners=False) upsample of the coarse logits, concat, 1x1 fusion conv,
// eval-mode BatchNorm (folded into scale/shift), ReLU, softmax over D.
// One wave per output pixel; lane d owns depth bin d (D <= 64).
__global__ __launch_bounds__(256) void depth_fuse_softmax_kernel(
    const float* __restrict__ d3, const float* __restrict__ d4, const float* __restrict__ w,
    const float* __restrict__ scale, const float* __restrict__ shift, int BN, int D, int H, int W,
    int H4, int W4, float rh, float rw, float* __restrict__ depth) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wt = lds;                 // [2D][D + 1] transposed fusion weights
  float* vec = lds + 2 * D * (D + 1);  // [4 waves][2D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < D * 2 * D; e += 256) {
    const int n = e / (2 * D), k = e % (2 * D);
    wt[k * (D + 1) + n] = w[e];
  }
  const int HW = H * W;
  const long long gp = (long long)blockIdx.x * 4 + wave;  // global pixel
  const bool live = gp < (long long)BN * HW;
  const int bn = live ? (int)(gp / HW) : 0, pix = live ? (int)(gp % HW) : 0;
  const int oh = pix / W, ow = pix % W;
  // source coordinates exactly as ATen's area_pixel_compute_source_index (align_corners=False)
  float sh = rh * ((float)oh + 0.5f) - 0.5f;
  float sw = rw * ((float)ow + 0.5f) - 0.5f;
  sh = sh < 0.f ? 0.f : sh;
  sw = sw < 0.f ? 0.f : sw;
  const int h0 = (int)sh, w0 = (int)sw;
  const int h1 = h0 + (h0 < H4 - 1 ? 1 : 0), w1 = w0 + (w0 < W4 - 1 ? 1 : 0);
  const float lh1 = sh - (float)h0, lh0 = 1.f - lh1;
  const float lw1 = sw - (float)w0, lw0 = 1.f - lw1;
  if (lane < D) {
    const float* c = d4 + ((size_t)bn * D + lane) * (H4 * W4);
    const float up = lh0 * (lw0 * c[h0 * W4 + w0] + lw1 * c[h0 * W4 + w1]) +
                     lh1 * (lw0 * c[h1 * W4 + w0] + lw1 * c[h1 * W4 + w1]);
    vec[wave * 2 * D + lane] = d3[((size_t)bn * D + lane) * HW + pix];
    vec[wave * 2 * D + D + lane] = up;
  }
  __syncthreads();
  float v = -INFINITY;
  if (lane < D) {
    float a = 0.f;
    const float* x = vec + wave * 2 * D;
    for (int k = 0; k < 2 * D; ++k) a = fmaf(wt[k * (D + 1) + lane], x[k], a);
    v = fmaxf(a * scale[lane] + shift[lane], 0.f);
  }
  const float mx = lss_wave_max(v);
  const float e = lane < D ? expf(v - mx) : 0.f;
  const float sum = lss_wave_sum(e);
  if (live && lane < D) depth[((size_t)bn * D + lane) * HW + pix] = e / sum;
}

template <int NT>
int launch_depthnet(const float* x, const float* w, const float* bias, int BN, int Cin, int HW,
                    int D, int C, float* depth, float* feat, int math, hipStream_t st) {
  const size_t lds_bytes = (size_t)5 * NT * 16 * LDS_LD * sizeof(float);
  dim3 grid(lss_cdiv(HW, PIX), BN);
  if (math == LSS_DT_F32)
    hipLaunchKernelGGL(depthnet_softmax_f32_kernel<NT>, grid, dim3(256), lds_bytes, st, x, w, bias,
                       Cin, HW, D, C, depth, feat);
  else
    hipLaunchKernelGGL(depthnet_softmax_bf16_kernel<NT>, grid, dim3(256), lds_bytes, st, x, w,
                       bias, Cin, HW, D, C, depth, feat);
  return lss_launch_status();
}

}  // namespace

extern "C" int lss_depthnet_softmax_fwd(const float* x, const float* w, const float* bias, int BN,
                                        int Cin, int HW, int D, int C, float* depth, float* feat,
                                        int math, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(w); LSS_CHECK_PTR(bias); LSS_CHECK_PTR(depth); LSS_CHECK_PTR(feat);
  LSS_CHECK_POS(BN); LSS_CHECK_POS(Cin); LSS_CHECK_POS(HW); LSS_CHECK_POS(D); LSS_CHECK_POS(C);
  if (math != LSS_DT_F32 && math != LSS_DT_BF16) return LSS_E_LAYOUT;
  // each wave owns Cin/4 input channels in 16- (f32) or 32- (bf16) deep blocks
  if (Cin % (math == LSS_DT_F32 ? 64 : 128) != 0) return LSS_E_SHAPE;
  if (BN > 65535) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) return LSS_E_ALIGN;
  const int NT = (D + C + 15) / 16;
  hipStream_t st = lss_stream(stream);
#define LSS_DN_CASE(n) \
  case n: return launch_depthnet<n>(x, w, bias, BN, Cin, HW, D, C, depth, feat, math, st);
  switch (NT) {
    LSS_DN_CASE(1) LSS_DN_CASE(2) LSS_DN_CASE(3) LSS_DN_CASE(4) LSS_DN_CASE(5) LSS_DN_CASE(6)
    LSS_DN_CASE(7) LSS_DN_CASE(8) LSS_DN_CASE(9) LSS_DN_CASE(10) LSS_DN_CASE(11) LSS_DN_CASE(12)
    default: return LSS_E_SHAPE;  // D + C <= 192
  }
#undef LSS_DN_CASE
}

template <int NTD, int NTC, typename T>
static int launch_camencode_v2(const void* xd, const float* wd, const float* bd, int Cd,
                               const float* xf, const float* wf, const float* bf, int Cf, int BN,
                               int HW, int D, int C, int softmax, float* depth, float* feat, int math,
                               hipStream_t st) {
  const size_t lds_bytes = (size_t)5 * (NTD + NTC) * 16 * LDS_LD * sizeof(float);
  dim3 grid(lss_cdiv(HW, PIX), BN);
  if (math == LSS_DT_BF16)
    hipLaunchKernelGGL((camencode_v2_bf16_kernel<NTD, NTC, T>), grid, dim3(256), lds_bytes, st,
                       static_cast<const T*>(xd), wd, bd, Cd, xf, wf, bf, Cf, HW, D, C, softmax, depth, feat);
  else
    hipLaunchKernelGGL((camencode_v2_kernel<NTD, NTC, T>), grid, dim3(256), lds_bytes, st,
                       static_cast<const T*>(xd), wd, bd, Cd, xf, wf, bf, Cf, HW, D, C, softmax, depth, feat);
  return lss_launch_status();
}

extern "C" int lss_camencode_v2_fwd(const void* x_depth, int dt, const float* w_depth,
                                    const float* b_depth, int Cd, const float* x_feat,
                                    const float* w_feat, const float* b_feat, int Cf, int BN, int HW,
                                    int D, int C, int softmax, int math, float* depth, float* feat,
                                    void* stream) {
  LSS_CHECK_PTR(x_depth); LSS_CHECK_PTR(w_depth); LSS_CHECK_PTR(b_depth); LSS_CHECK_PTR(depth);
  if (math != LSS_DT_F32 && math != LSS_DT_BF16) return LSS_E_LAYOUT;
  // each wave owns a K quarter in 16- (f32 MFMA) or 32-deep (bf16 MFMA) blocks
  if (math == LSS_DT_BF16 && (Cd % 128 != 0 || (C > 0 && Cf % 128 != 0))) return LSS_E_SHAPE;
  LSS_CHECK_POS(BN); LSS_CHECK_POS(HW); LSS_CHECK_POS(D); LSS_CHECK_POS(Cd);
  if (dt != LSS_DT_F32 && dt != LSS_DT_BF16) return LSS_E_LAYOUT;
  if (C < 0 || (C > 0 && (x_feat == nullptr || w_feat == nullptr || b_feat == nullptr || feat == nullptr)))
    return C < 0 ? LSS_E_SHAPE : LSS_E_NULL;
  if (Cd % 64 != 0 || (C > 0 && Cf % 64 != 0) || BN > 65535) return LSS_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(w_depth) | reinterpret_cast<uintptr_t>(x_depth)) & 15) != 0)
    return LSS_E_ALIGN;
  if (C > 0 && (reinterpret_cast<uintptr_t>(w_feat) & 15) != 0) return LSS_E_ALIGN;
  const int ntd = (D + 15) / 16, ntc = (C + 15) / 16;
  hipStream_t st = lss_stream(stream);
#define LSS_V2_CASE(a, b)                                                                         \
  if (ntd == a && ntc == b)                                                                       \
    return dt == LSS_DT_F32                                                                       \
               ? launch_camencode_v2<a, b, float>(x_depth, w_depth, b_depth, Cd, x_feat, w_feat,  \
                                                  b_feat, Cf, BN, HW, D, C, softmax, depth, feat, math, st) \
               : launch_camencode_v2<a, b, unsigned short>(x_depth, w_depth, b_depth, Cd, x_feat, \
                                                           w_feat, b_feat, Cf, BN, HW, D, C,      \
                                                           softmax, depth, feat, math, st);
  LSS_V2_CASE(3, 0) LSS_V2_CASE(3, 4) LSS_V2_CASE(3, 8)
  LSS_V2_CASE(4, 0) LSS_V2_CASE(4, 4) LSS_V2_CASE(4, 8)
  LSS_V2_CASE(1, 0) LSS_V2_CASE(1, 1)
#undef LSS_V2_CASE
  return LSS_E_SHAPE;  // D in (0,16] u (32,64], C in {0, (48,64], (112,128]} (+ the unit-test shape)
}

extern "C" int lss_depth_fuse_softmax_fwd(const float* d3, const float* d4, const float* w_fusion,
                                          const float* scale, const float* shift, int BN, int D,
                                          int H, int W, int H4, int W4, float* depth, void* stream) {
  LSS_CHECK_PTR(d3); LSS_CHECK_PTR(d4); LSS_CHECK_PTR(w_fusion); LSS_CHECK_PTR(scale);
  LSS_CHECK_PTR(shift); LSS_CHECK_PTR(depth);
  LSS_CHECK_POS(BN); LSS_CHECK_POS(D); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(H4);
  LSS_CHECK_POS(W4);
  if (D > 64) return LSS_E_SHAPE;
  const long long npix = (long long)BN * H * W;
  if (npix >= (1LL << 31)) return LSS_E_SHAPE;
  const size_t lds_bytes = ((size_t)2 * D * (D + 1) + 4 * 2 * D) * sizeof(float);
  // ATen: scale = (float)in / out for align_corners=False with an explicit output size
  const float rh = (float)H4 / (float)H, rw = (float)W4 / (float)W;
  hipLaunchKernelGGL(depth_fuse_softmax_kernel, dim3(lss_cdiv(npix, 4)), dim3(256), lds_bytes,
                     lss_stream(stream), d3, d4, w_fusion, scale, shift, BN, D, H, W, H4, W4, rh, rw,
                     depth);
  return lss_launch_status();
}
