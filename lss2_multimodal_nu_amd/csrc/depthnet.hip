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
#include "lss_common.h"

namespace {

constexpr int PIX = 16;         // pixels per workgroup
constexpr int LDS_LD = PIX + 1; // padded row of the [n][pix] logits tile

// Shared tail of both kernels: K-quarter partials -> LDS -> bias -> outputs.
template <int NT>
__device__ __forceinline__ void depthnet_epilogue(const f32x4 (&acc)[NT], float* lds,
                                                  const float* __restrict__ bias, int bn, int pix0,
                                                  int HW, int D, int C, float* __restrict__ depth,
                                                  float* __restrict__ feat) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int NO = D + C;
  // partial[wave][n][pix] -> LDS
  float* part = lds;  // [4][NT*16][LDS_LD]
  const int NR = NT * 16;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      part[((size_t)wave * NR + 16 * t + 4 * j + r) * LDS_LD + col] = acc[t][r];
  __syncthreads();
  // reduce the four K quarters (fixed order -> deterministic), add bias
  float* logit = lds + 4 * NR * LDS_LD;  // [NR][LDS_LD]
  for (int e = tid; e < NO * PIX; e += 256) {
    const int n = e / PIX, p = e % PIX;
    const int o = n * LDS_LD + p;
    float v = part[o] + part[NR * LDS_LD + o];
    v += part[2 * NR * LDS_LD + o];
    v += part[3 * NR * LDS_LD + o];
    logit[o] = v + bias[n];
  }
  __syncthreads();

  // context features: feat[(bn*HW + pix)*C + c] = logit[D + c][pix]
  for (int e = tid; e < PIX * C; e += 256) {
    const int p = e / C, c = e % C;
    if (pix0 + p < HW) feat[((size_t)bn * HW + pix0 + p) * C + c] = logit[(D + c) * LDS_LD + p];
  }
  // softmax over d for each of the 16 pixels: 16 lanes per pixel
  {
    const int p = tid & 15, part_id = tid >> 4;  // 16 parts
    float m = -INFINITY;
    for (int d = part_id; d < D; d += 16) m = fmaxf(m, logit[d * LDS_LD + p]);
    // combine the 16 parts of a pixel: lanes p, p+16, p+32, p+48 of 4 waves -> LDS
    float* red = part;  // reuse: [16 parts][16 pix]
    red[part_id * PIX + p] = m;
    __syncthreads();
    float mx = red[p];
#pragma unroll
    for (int q = 1; q < 16; ++q) mx = fmaxf(mx, red[q * PIX + p]);
    __syncthreads();
    float s = 0.f;
    for (int d = part_id; d < D; d += 16) {
      const float e = expf(logit[d * LDS_LD + p] - mx);
      logit[d * LDS_LD + p] = e;
      s += e;
    }
    red[part_id * PIX + p] = s;
    __syncthreads();
    float sum = red[p];
#pragma unroll
    for (int q = 1; q < 16; ++q) sum += red[q * PIX + p];
    if (pix0 + p < HW)
      for (int d = part_id; d < D; d += 16)
        depth[((size_t)bn * D + d) * HW + pix0 + p] = logit[d * LDS_LD + p] / sum;
  }
}

// MFMA operand maps (cdna_hip_programming.md section 3):
//   16x16x4 f32:  A[row = l&15][k = l>>4],  B[k = l>>4][col = l&15],
//                 D[row = 4*(l>>4) + r][col = l&15], r = 0..3
// Within a 16-deep K block the four k-steps s = 0..3 use k = kb + 4*(l>>4) + s,
// so one 16-B load of W[n][kb + 4*(l>>4) ..+3] feeds four MFMAs.
template <int NT>
__global__ __launch_bounds__(256) void depthnet_softmax_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    int Cin, int HW, int D, int C, float* __restrict__ depth, float* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int bn = blockIdx.y;
  const int pix0 = blockIdx.x * PIX;
  const int NO = D + C;
  const int pix = min(pix0 + col, HW - 1);
  const int kq = Cin >> 2;  // K quarter of this wave
  const float* xb = x + ((size_t)bn * Cin + (size_t)wave * kq) * HW + pix;
  const float* wb = w + (size_t)wave * kq + 4 * j;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = 0; kb < kq; kb += 16) {
    float xs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xs[s] = xb[(size_t)(kb + 4 * j + s) * HW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + col;
      f32x4 wa = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < NO) wa = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xs[s], acc[t], 0, 0, 0);
    }
  }

  depthnet_epilogue<NT>(acc, lds, bias, bn, pix0, HW, D, C, depth, feat);
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

  depthnet_epilogue<NT>(acc, lds, bias, bn, pix0, HW, D, C, depth, feat);
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
