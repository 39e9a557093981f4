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

// sums[0] = sum w*nll, sums[1] = sum w, loss = sums[0] / sums[1]
__global__ void weighted_ce_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ sums,
                                            float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < nblk; ++k) {
    a += part[2 * k];
    b += part[2 * k + 1];
  }
  sums[0] = a;
  sums[1] = b;
  loss[0] = a / b;
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
