// Weighted cross-entropy over NCHW logits, forward and backward.
// replaces: nn.CrossEntropyLoss(weight)(ypred, ytgt) of SimpleLoss / MultiLoss
//           (src/tools.py:221-231, 234-238: log_softmax + nll_loss2d and their backward).
//   loss = sum_i w[t_i] * (-log softmax(x_i)[t_i]) / sum_i w[t_i]   (pixels with t_i outside
//   [0, C) - e.g. ignore_index -100 - carry weight 0)
// The logits are 4 channels x 160 k pixels: pure bandwidth, so one pass reads them and writes
// per-workgroup partial sums (fixed order -> bit-reproducible), and the backward pass re-reads
// them to write grad = g * w[t] * (softmax - onehot) / sum w without storing the softmax.
#include "lss_common.h"

namespace {

constexpr int CE_MAXC = 16;
constexpr int CE_BLOCKS = 256;

template <bool BWD>
__global__ __launch_bounds__(256) void weighted_ce_kernel(const float* __restrict__ x,
                                                          const long long* __restrict__ tgt,
                                                          const float* __restrict__ w, int B, int C,
                                                          long long HW, const float* __restrict__ sums,
                                                          const float* __restrict__ gout,
                                                          float* __restrict__ part, float* __restrict__ gx) {
  __shared__ float red[2][256];
  const long long n = (long long)B * HW;
  float wc[CE_MAXC];
#pragma unroll
  for (int c = 0; c < CE_MAXC; ++c) wc[c] = c < C ? w[c] : 0.f;
  const float gscale = BWD ? gout[0] / sums[1] : 0.f;
  float s_loss = 0.f, s_w = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long b = i / HW, p = i - b * HW;
    const float* xp = x + (size_t)b * C * HW + p;
    float v[CE_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        v[c] = xp[(size_t)c * HW];
        mx = fmaxf(mx, v[c]);
      }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C) {
        v[c] = expf(v[c] - mx);
        se += v[c];
      }
    const long long t = tgt[i];
    const bool ok = t >= 0 && t < C;
    float wt = 0.f, pt = 1.f;
#pragma unroll
    for (int c = 0; c < CE_MAXC; ++c)
      if (c < C && ok && c == (int)t) {
        wt = wc[c];
        pt = v[c] / se;
      }
    if (!BWD) {
      s_loss += ok ? -wt * logf(pt) : 0.f;
      s_w += wt;
    } else {
      float* gp = gx + (size_t)b * C * HW + p;
      const float k = gscale * wt / se;
#pragma unroll
      for (int c = 0; c < CE_MAXC; ++c)
        if (c < C) gp[(size_t)c * HW] = k * v[c] - ((ok && c == (int)t) ? gscale * wt : 0.f);
    }
  }
  if (!BWD) {
    red[0][threadIdx.x] = s_loss;
    red[1][threadIdx.x] = s_w;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        red[0][threadIdx.x] += red[0][threadIdx.x + o];
        red[1][threadIdx.x] += red[1][threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      part[2 * blockIdx.x] = red[0][0];
      part[2 * blockIdx.x + 1] = red[1][0];
    }
  }
}

// sums[0] = sum w*nll, sums[1] = sum w, loss = sums[0] / sums[1].  One wave: lane i sums partials i, i + 64, ... in
// order, then a fixed shuffle tree (a single thread walking 1024 partials cost 59 us).
__global__ void weighted_ce_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ sums,
                                            float* __restrict__ loss) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  float a = 0.f, b = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    a += part[2 * k];
    b += part[2 * k + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (threadIdx.x == 0) {
    sums[0] = a;
    sums[1] = b;
    loss[0] = a / b;
  }
}

}  // namespace

extern "C" int lss_weighted_ce_fwd(const float* logits, const long long* target, const float* weight, int B, int C,
                                   long long HW, float* workspace, float* sums, float* loss, void* stream) {
  LSS_CHECK_PTR(logits); LSS_CHECK_PTR(target); LSS_CHECK_PTR(weight); LSS_CHECK_PTR(workspace);
  LSS_CHECK_PTR(sums); LSS_CHECK_PTR(loss);
  LSS_CHECK_POS(B); LSS_CHECK_POS(C);
  if (HW <= 0 || C > CE_MAXC) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  hipLaunchKernelGGL(weighted_ce_kernel<false>, dim3(CE_BLOCKS), dim3(256), 0, st, logits, target, weight, B, C, HW,
                     nullptr, nullptr, workspace, nullptr);
  hipLaunchKernelGGL(weighted_ce_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, CE_BLOCKS, sums, loss);
  return lss_launch_status();
}

extern "C" int lss_weighted_ce_bwd(const float* logits, const long long* target, const float* weight, int B, int C,
                                   long long HW, const float* sums, const float* grad_loss, float* grad_logits,
                                   void* stream) {
  LSS_CHECK_PTR(logits); LSS_CHECK_PTR(target); LSS_CHECK_PTR(weight); LSS_CHECK_PTR(sums);
  LSS_CHECK_PTR(grad_loss); LSS_CHECK_PTR(grad_logits);
  LSS_CHECK_POS(B); LSS_CHECK_POS(C);
  if (HW <= 0 || C > CE_MAXC) return LSS_E_SHAPE;
  const long long n = (long long)B * HW;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(weighted_ce_kernel<true>, dim3(grid), dim3(256), 0, lss_stream(stream), logits, target, weight,
                     B, C, HW, sums, grad_loss, nullptr, grad_logits);
  return lss_launch_status();
}

namespace {

// ---------------------------------------------------------------------------
// Fused 1x1 head + log-softmax + weighted NLL, forward and backward.
// replaces: `up2[4]` (nn.Conv2d(128, outC, 1), src/modules.py:115) followed by nn.CrossEntropyLoss(weight) of
//           SimpleLoss / MultiLoss (src/tools.py:221-238), and their autograd.
//   logit[p][k] = b[k] + sum_c y[p][c] * W[k][c];  loss = sum_p w[t_p] * (lse_p - logit[p][t_p]) / sum_p w[t_p]
// y is the NHWC bf16 activation the last conv + BatchNorm + ReLU unit produced; the (B, K, H, W) logits are never
// written (forward) nor read (backward: they are recomputed from y, which autograd keeps anyway).
// Work split: a pixel's Cin channels sit on Cin/8 consecutive lanes (8 channels = one 16-B load per lane), so a wave
// covers 64 / (Cin/8) pixels per pass; the K <= 8 partial dot products of a lane are summed over the pixel's lanes
// with DPP row operations (Cin = 128: a whole DPP row of 16).  Forward: per-workgroup partial sums in a fixed order
// (bit-reproducible), finalize as the plain CE.  Backward: dy[p][c] = sum_k g[p][k] W[k][c] with
// g = grad * w[t] * (softmax - onehot) / sum w, written bf16; dW[k][c] = sum_p g[p][k] y[p][c] and db accumulate in
// registers per lane, meet in LDS per workgroup and leave as per-workgroup partials that a second kernel sums in a
// fixed order.
constexpr int HC_MAXK = 8;
constexpr int HC_BLOCKS = 512;

template <int DPP>
__device__ __forceinline__ float hc_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), DPP, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (every lane gets the total): xor 1, xor 2 (quad_perm), half-mirror, mirror
__device__ __forceinline__ float hc_row_sum16(float v) {
  v += hc_dpp<0xB1>(v);
  v += hc_dpp<0x4E>(v);
  v += hc_dpp<0x141>(v);
  v += hc_dpp<0x140>(v);
  return v;
}

// MODE 0 / 1: head + cross-entropy forward / backward (above).  MODE 2 / 3: the head ALONE, forward (logits written
// NCHW fp32: `logits[(b K + k) HW + pix]`) and backward (g[p][k] read from the NCHW fp32 gradient of the logits): the
// `up2[4]` of a training-mode `model(x)` call whose loss is computed elsewhere.  The library's convolution is not an
// option there: its backward is not safe inside a HIP graph ("What the graph exposed").
template <int K, int MODE>
__global__ __launch_bounds__(256) void head_ce_kernel(const unsigned short* __restrict__ y,
                                                      const float* __restrict__ hw, const float* __restrict__ hb,
                                                      const long long* __restrict__ tgt,
                                                      const float* __restrict__ cw, long long M,
                                                      const float* __restrict__ sums, const float* __restrict__ gout,
                                                      float* __restrict__ part, unsigned short* __restrict__ dy,
                                                      float* __restrict__ dw_part, float* __restrict__ logits_io,
                                                      long long HW) {
  constexpr bool BWD = MODE == 1 || MODE == 3, PLAIN = MODE >= 2;
  constexpr int CIN = 128, LPP = CIN / 8;  // lanes per pixel (16 = one DPP row)
  __shared__ float red[2][256];
  __shared__ float dwred[BWD ? 4 * K * CIN : 1];
  __shared__ float dbred[BWD ? 4 * K : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & (LPP - 1), prow = lane / LPP;  // channel slice, pixel within the wave's pass
  // this lane's 8 x K weights
  float w[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) w[k][j] = hw[k * CIN + sub * 8 + j];
  float bias[K], clsw[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { bias[k] = hb[k]; clsw[k] = PLAIN ? 0.f : cw[k]; }
  const float gscale = (BWD && !PLAIN) ? gout[0] / sums[1] : 0.f;
  float s_loss = 0.f, s_w = 0.f;
  float dwa[BWD ? K : 1][8];
  float dba[BWD ? K : 1];
  if (BWD) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      dba[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dwa[k][j] = 0.f;
    }
  }
  constexpr int PPW = 64 / LPP;  // pixels per wave pass (4)
  const long long stride = (long long)gridDim.x * 4 * PPW;
  for (long long p0 = ((long long)blockIdx.x * 4 + wave) * PPW; p0 < M; p0 += stride) {
    const long long p = p0 + prow;
    const bool live = p < M;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (live) raw = *reinterpret_cast<const uint4*>(y + (size_t)p * CIN + sub * 8);
    const unsigned int ru[4] = {raw.x, raw.y, raw.z, raw.w};
    float yv[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      yv[2 * q] = lss_bf2f((unsigned short)(ru[q] & 0xffff));
      yv[2 * q + 1] = lss_bf2f((unsigned short)(ru[q] >> 16));
    }
    if constexpr (PLAIN) {
      const long long bimg = live ? p / HW : 0, pix = live ? p - bimg * HW : 0;
      float* lp = logits_io + (size_t)bimg * K * HW + pix;
      if constexpr (!BWD) {
        // logits: lane `sub == k` of the pixel's row stores class k
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) a = fmaf(yv[j], w[k][j], a);
          a = hc_row_sum16(a) + bias[k];
          if (sub == k) mine = a;
        }
        if (live && sub < K) lp[(size_t)sub * HW] = mine;
      } else {
        float g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = live ? lp[(size_t)k * HW] : 0.f;
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < K; ++k) a = fmaf(g[k], w[k][j], a);
          dv[j] = a;
        }
        if (live) {
          uint4 o;
          o.x = lss_pack_bf2(dv[0], dv[1]); o.y = lss_pack_bf2(dv[2], dv[3]);
          o.z = lss_pack_bf2(dv[4], dv[5]); o.w = lss_pack_bf2(dv[6], dv[7]);
          *reinterpret_cast<uint4*>(dy + (size_t)p * CIN + sub * 8) = o;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
          for (int j = 0; j < 8; ++j) dwa[k][j] = fmaf(g[k], yv[j], dwa[k][j]);
          if (sub == 0) dba[k] += g[k];
        }
      }
      continue;
    }
    float logit[K];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a = fmaf(yv[j], w[k][j], a);
      logit[k] = hc_row_sum16(a) + bias[k];
      mx = fmaxf(mx, logit[k]);
    }
    float ex[K], se = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ex[k] = expf(logit[k] - mx);
      se += ex[k];
    }
    const long long t = live ? tgt[p] : -1;
    const bool ok = t >= 0 && t < K;
    float wt = 0.f, lt = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (ok && k == (int)t) {
        wt = clsw[k];
        lt = logit[k];
      }
    if (!BWD) {
      if (sub == 0 && ok) {  // one lane per pixel contributes
        s_loss += wt * (mx + logf(se) - lt);
        s_w += wt;
      }
    } else {
      float g[K];
      const float kk = gscale * wt / se;
#pragma unroll
      for (int k = 0; k < K; ++k) g[k] = ok ? kk * ex[k] - ((k == (int)t) ? gscale * wt : 0.f) : 0.f;
      float dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) a = fmaf(g[k], w[k][j], a);
        dv[j] = a;
      }
      if (live) {
        uint4 o;
        o.x = lss_pack_bf2(dv[0], dv[1]); o.y = lss_pack_bf2(dv[2], dv[3]);
        o.z = lss_pack_bf2(dv[4], dv[5]); o.w = lss_pack_bf2(dv[6], dv[7]);
        *reinterpret_cast<uint4*>(dy + (size_t)p * CIN + sub * 8) = o;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dwa[k][j] = fmaf(g[k], yv[j], dwa[k][j]);
        if (sub == 0) dba[k] += g[k];
      }
    }
  }
  if (MODE == 2) return;
  if (!BWD) {
    red[0][tid] = s_loss;
    red[1][tid] = s_w;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        red[0][tid] += red[0][tid + o];
        red[1][tid] += red[1][tid + o];
      }
      __syncthreads();
    }
    if (tid == 0) {
      part[2 * blockIdx.x] = red[0][0];
      part[2 * blockIdx.x + 1] = red[1][0];
    }
  } else {
    // dW / db of this workgroup: the 4 pixel rows of a wave meet through shuffles (xor 16, xor 32), the 4 waves
    // through LDS, in a fixed order
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = dwa[k][j];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (prow == 0) dwred[(wave * K + k) * CIN + sub * 8 + j] = v;
      }
      float b = dba[k];
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if (lane == 0) dbred[wave * K + k] = b;
    }
    __syncthreads();
    float* outp = dw_part + (size_t)blockIdx.x * (K * CIN + K);
    for (int e = tid; e < K * CIN; e += 256) {
      const int k = e / CIN, c = e - k * CIN;
      outp[e] = (dwred[(0 * K + k) * CIN + c] + dwred[(1 * K + k) * CIN + c]) +
                (dwred[(2 * K + k) * CIN + c] + dwred[(3 * K + k) * CIN + c]);
    }
    if (tid < K) outp[K * CIN + tid] = (dbred[tid] + dbred[K + tid]) + (dbred[2 * K + tid] + dbred[3 * K + tid]);
  }
}

// dW[k][c], db[k] = fixed-order sums of the per-workgroup partials: one wave per output element, lane i takes
// partials i, i + 64, ... in order, then a fixed shuffle tree (grid = n / 4 workgroups of 4 waves)
__global__ __launch_bounds__(256) void head_ce_reduce_kernel(const float* __restrict__ dw_part, int nblk, int n,
                                                             float* __restrict__ dw, float* __restrict__ db, int ndw) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= n) return;
  float a = 0.f;
  for (int k = lane; k < nblk; k += 64) a += dw_part[(size_t)k * n + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) {
    if (e < ndw) dw[e] = a;
    else db[e - ndw] = a;
  }
}

}  // namespace

extern "C" size_t lss_head_ce_workspace_bytes(int K) {
  if (K <= 0 || K > HC_MAXK) return 0;
  return (size_t)HC_BLOCKS * (K * 128 + K) * sizeof(float) + 2 * HC_BLOCKS * sizeof(float);
}

extern "C" int lss_head_ce_fwd(const void* y, const float* head_w, const float* head_b, const long long* target,
                               const float* class_w, long long M, int Cin, int K, float* workspace, float* sums,
                               float* loss, void* stream) {
  LSS_CHECK_PTR(y); LSS_CHECK_PTR(head_w); LSS_CHECK_PTR(head_b); LSS_CHECK_PTR(target); LSS_CHECK_PTR(class_w);
  LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(sums); LSS_CHECK_PTR(loss);
  if (M <= 0 || Cin != 128 || (K != 4 && K != 8)) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(y) & 15) != 0) return LSS_E_ALIGN;
  hipStream_t st = lss_stream(stream);
  const int grid = (int)((M + 15) / 16 > HC_BLOCKS ? HC_BLOCKS : (M + 15) / 16);
  const unsigned short* yp = reinterpret_cast<const unsigned short*>(y);
  if (K == 4)
    hipLaunchKernelGGL((head_ce_kernel<4, 0>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, target, class_w,
                       M, nullptr, nullptr, workspace, nullptr, nullptr, nullptr, 0LL);
  else
    hipLaunchKernelGGL((head_ce_kernel<8, 0>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, target, class_w,
                       M, nullptr, nullptr, workspace, nullptr, nullptr, nullptr, 0LL);
  hipLaunchKernelGGL(weighted_ce_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, grid, sums, loss);
  return lss_launch_status();
}

extern "C" int lss_head_ce_bwd(const void* y, const float* head_w, const float* head_b, const long long* target,
                               const float* class_w, long long M, int Cin, int K, const float* sums,
                               const float* grad_loss, float* workspace, void* dy, float* d_head_w, float* d_head_b,
                               void* stream) {
  LSS_CHECK_PTR(y); LSS_CHECK_PTR(head_w); LSS_CHECK_PTR(head_b); LSS_CHECK_PTR(target); LSS_CHECK_PTR(class_w);
  LSS_CHECK_PTR(sums); LSS_CHECK_PTR(grad_loss); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(dy);
  LSS_CHECK_PTR(d_head_w); LSS_CHECK_PTR(d_head_b);
  if (M <= 0 || Cin != 128 || (K != 4 && K != 8)) return LSS_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return LSS_E_ALIGN;
  hipStream_t st = lss_stream(stream);
  const int grid = (int)((M + 15) / 16 > HC_BLOCKS ? HC_BLOCKS : (M + 15) / 16);
  const unsigned short* yp = reinterpret_cast<const unsigned short*>(y);
  unsigned short* dyp = reinterpret_cast<unsigned short*>(dy);
  if (K == 4)
    hipLaunchKernelGGL((head_ce_kernel<4, 1>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, target, class_w, M,
                       sums, grad_loss, nullptr, dyp, workspace, nullptr, 0LL);
  else
    hipLaunchKernelGGL((head_ce_kernel<8, 1>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, target, class_w, M,
                       sums, grad_loss, nullptr, dyp, workspace, nullptr, 0LL);
  const int n = K * 128 + K;
  hipLaunchKernelGGL(head_ce_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, st, workspace, grid, n, d_head_w,
                     d_head_b, K * 128);
  return lss_launch_status();
}

// The 1x1 head alone (ref src/modules.py:115, up2[4] = nn.Conv2d(128, outC, 1)), forward and backward, for a
// training-mode forward whose loss is computed by the caller: y (M, 128) bf16 NHWC rows -> logits (B, K, H, W) fp32
// NCHW (M = B * HW); backward: grad_logits (B, K, H, W) fp32 -> dy (M, 128) bf16, d_head_w (K, 128), d_head_b (K),
// fixed-order partial sums (bit-reproducible).  workspace: lss_head_ce_workspace_bytes(K).
extern "C" int lss_head1x1_fwd(const void* y, const float* head_w, const float* head_b, long long M, long long HW,
                               int Cin, int K, float* logits, void* stream) {
  LSS_CHECK_PTR(y); LSS_CHECK_PTR(head_w); LSS_CHECK_PTR(head_b); LSS_CHECK_PTR(logits);
  if (M <= 0 || HW <= 0 || M % HW != 0 || Cin != 128 || (K != 4 && K != 8)) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(y) & 15) != 0) return LSS_E_ALIGN;
  const int grid = (int)((M + 15) / 16 > HC_BLOCKS ? HC_BLOCKS : (M + 15) / 16);
  const unsigned short* yp = reinterpret_cast<const unsigned short*>(y);
  hipStream_t st = lss_stream(stream);
  if (K == 4)
    hipLaunchKernelGGL((head_ce_kernel<4, 2>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, nullptr, nullptr, M,
                       nullptr, nullptr, nullptr, nullptr, nullptr, logits, HW);
  else
    hipLaunchKernelGGL((head_ce_kernel<8, 2>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, nullptr, nullptr, M,
                       nullptr, nullptr, nullptr, nullptr, nullptr, logits, HW);
  return lss_launch_status();
}

extern "C" int lss_head1x1_bwd(const void* y, const float* head_w, const float* head_b, const float* grad_logits,
                               long long M, long long HW, int Cin, int K, float* workspace, void* dy, float* d_head_w,
                               float* d_head_b, void* stream) {
  LSS_CHECK_PTR(y); LSS_CHECK_PTR(head_w); LSS_CHECK_PTR(head_b); LSS_CHECK_PTR(grad_logits); LSS_CHECK_PTR(workspace);
  LSS_CHECK_PTR(dy); LSS_CHECK_PTR(d_head_w); LSS_CHECK_PTR(d_head_b);
  if (M <= 0 || HW <= 0 || M % HW != 0 || Cin != 128 || (K != 4 && K != 8)) return LSS_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return LSS_E_ALIGN;
  const int grid = (int)((M + 15) / 16 > HC_BLOCKS ? HC_BLOCKS : (M + 15) / 16);
  const unsigned short* yp = reinterpret_cast<const unsigned short*>(y);
  hipStream_t st = lss_stream(stream);
  float* gl = const_cast<float*>(grad_logits);  // read only (the kernel's logits pointer serves both modes)
  if (K == 4)
    hipLaunchKernelGGL((head_ce_kernel<4, 3>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, nullptr, nullptr, M,
                       nullptr, nullptr, nullptr, reinterpret_cast<unsigned short*>(dy), workspace, gl, HW);
  else
    hipLaunchKernelGGL((head_ce_kernel<8, 3>), dim3(grid), dim3(256), 0, st, yp, head_w, head_b, nullptr, nullptr, M,
                       nullptr, nullptr, nullptr, reinterpret_cast<unsigned short*>(dy), workspace, gl, HW);
  const int n = K * 128 + K;
  hipLaunchKernelGGL(head_ce_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, st, workspace, grid, n, d_head_w,
                     d_head_b, K * 128);
  return lss_launch_status();
}
