// Training-mode BatchNorm2d (+ residual add + ReLU) on NHWC bf16 rows, forward and backward.
// replaces: the batch_norm / add / relu ATen ops (and their autograd nodes) of
//           src/modules.py:16-21 (Up), torchvision BasicBlock.forward, src/modules.py:100-101,
//           112-113 when the model is in train() mode (train.py:50-61).
//
//   forward   mean, var over the M = B*H*W rows (biased var for normalisation, unbiased for the
//             running estimate, as nn.BatchNorm2d);  y = act(gamma * (z - mean) * invstd + beta (+ res))
//   backward  g = dy * [y > 0];  dbeta = sum g;  dgamma = sum g * xhat;
//             dz = gamma * invstd * (g - dbeta / M - xhat * dgamma / M);  dres = g
//
// All of it is bandwidth work: every tensor is read or written in 16-B pieces, one pass for
// the statistics and one for the elementwise part (z is read twice, bf16).  Channel sums are
// two-stage and fixed-order (per-workgroup partials, then one small kernel), so results are
// bit-reproducible.  Thread layout of the reductions: C/8 channel groups x 256/(C/8) row lanes.
#include "lss_common.h"

namespace {

constexpr int RED_BLOCKS = 512;

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = lss_bf2f((unsigned short)(u[k] & 0xffff));
    f[2 * k + 1] = lss_bf2f((unsigned short)(u[k] >> 16));
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 o;
  o.x = lss_pack_bf2(f[0], f[1]); o.y = lss_pack_bf2(f[2], f[3]);
  o.z = lss_pack_bf2(f[4], f[5]); o.w = lss_pack_bf2(f[6], f[7]);
  return o;
}

// per-workgroup partial sums: part[blk][0][c] = sum a, part[blk][1][c] = sum b over the
// workgroup's rows, where (a, b) = (z - p, (z - p)^2) [MODE 0] or (g, g * xhat) [MODE 1].
// MODE 0 sums are taken about a per-channel PIVOT p: var = E[(z-p)^2] - (E[z-p])^2 cancels relative
// to (mean - p)^2 instead of mean^2, so a channel whose mean dwarfs its spread keeps its variance.  p = the
// tensor's own first row (pivot16; local statistics: |mean - p| is of the order of the spread) or a vector every
// rank shares (pivot32 = running_mean; synchronised statistics: the sums of different ranks must add), else 0.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const unsigned short* __restrict__ z,
                                                        const unsigned short* __restrict__ dy,
                                                        const unsigned short* __restrict__ y,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, long long M, int C,
                                                        int relu, float* __restrict__ part,
                                                        const float* __restrict__ pivot32 = nullptr,
                                                        const unsigned short* __restrict__ pivot16 = nullptr) {
  __shared__ float red[2][256][8];
  const int G = C >> 3;            // channel groups of 8
  const int L = 256 / G;           // row lanes
  const int tid = threadIdx.x, g = tid % G, lane = tid / G;
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
  float mu[8], is[8];
  if (MODE == 1 && lane < L) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = mean[g * 8 + k];
      is[k] = invstd[g * 8 + k];
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      mu[k] = pivot32 ? pivot32[g * 8 + k] : (pivot16 ? lss_bf2f(pivot16[g * 8 + k]) : 0.f);
  }
  if (lane < L) {
    const long long rows_per = (M + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
    for (long long r = r0 + lane; r < r1; r += L) {
      float zv[8];
      unpack8(*reinterpret_cast<const uint4*>(z + r * C + g * 8), zv);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = zv[k] - mu[k];
          a[k] += d;
          b[k] = fmaf(d, d, b[k]);
        }
      } else {
        float gv[8], yv[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + r * C + g * 8), gv);
        if (relu) unpack8(*reinterpret_cast<const uint4*>(y + r * C + g * 8), yv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gk = (relu && !(yv[k] > 0.f)) ? 0.f : gv[k];
          a[k] += gk;
          b[k] = fmaf(gk, (zv[k] - mu[k]) * is[k], b[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[0][tid][k] = a[k];
    red[1][tid][k] = b[k];
  }
  __syncthreads();
  // fixed-order sum over the row lanes
  for (int e = tid; e < 2 * C; e += 256) {
    const int which = e / C, c = e % C;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += red[which][l * G + (c >> 3)][c & 7];
    part[((size_t)blockIdx.x * 2 + which) * C + c] = s;
  }
}

// sum of the nblk per-workgroup partials of channel c: one wave per channel, lane l takes
// blocks l, l+64, ... in order, then a fixed xor tree - deterministic and ~20x shorter than one
// thread walking all partials
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int nblk, int C, int c, float& s1,
                                             float& s2) {
  const int lane = threadIdx.x & 63;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < nblk; k += 64) {
    a += part[((size_t)k * 2) * C + c];
    b += part[((size_t)k * 2 + 1) * C + c];
  }
  s1 = lss_wave_sum(a);
  s2 = lss_wave_sum(b);
}

// forward finalize: batch statistics, running-estimate update, folded scale/shift
__global__ void bn_fwd_finalize_kernel(const float* __restrict__ part, int nblk, long long M, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       float momentum, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, float* __restrict__ scale,
                                       float* __restrict__ shift, float* __restrict__ save_mean,
                                       float* __restrict__ save_invstd, const float* __restrict__ pivot32 = nullptr,
                                       const unsigned short* __restrict__ pivot16 = nullptr) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per channel
  if (c >= C) return;
  float s1, s2;
  sum_partials(part, nblk, C, c, s1, s2);
  if ((threadIdx.x & 63) != 0) return;
  const float inv_m = 1.f / (float)M;
  const float piv = pivot32 ? pivot32[c] : (pivot16 ? lss_bf2f(pivot16[c]) : 0.f);  // the sums are about this pivot
  const float dm = s1 * inv_m;
  const float mean = piv + dm;
  const float var = fmaxf(s2 * inv_m - dm * dm, 0.f);
  const float invstd = rsqrtf(var + eps);
  save_mean[c] = mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (running_mean) {
    const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// y = act(z * scale + shift (+ res))
__global__ __launch_bounds__(256) void bn_apply_kernel(const unsigned short* __restrict__ z,
                                                       const unsigned short* __restrict__ res,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, long long n8, int C,
                                                       int relu, unsigned short* __restrict__ y) {
  const int G = C >> 3;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long long)gridDim.x * 256) {
    const int g = (int)(e % G);
    float v[8], r[8];
    unpack8(*reinterpret_cast<const uint4*>(z + e * 8), v);
    if (res) unpack8(*reinterpret_cast<const uint4*>(res + e * 8), r);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = fmaf(v[k], scale[g * 8 + k], shift[g * 8 + k]);
      if (res) t += r[k];
      v[k] = relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<uint4*>(y + e * 8) = pack8(v);
  }
}

// backward finalize: dgamma, dbeta and the two per-channel coefficients of the apply pass
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, long long M, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per channel
  if (c >= C) return;
  float s1, s2;
  sum_partials(part, nblk, C, c, s1, s2);
  if ((threadIdx.x & 63) != 0) return;
  dbeta[c] = s1;
  dgamma[c] = s2;
  const float inv_m = 1.f / (float)M;
  coef[c] = s1 * inv_m;        // mean of g
  coef[C + c] = s2 * inv_m;    // mean of g * xhat
  coef[2 * C + c] = gamma[c] * invstd[c];
}

// dz = gamma * invstd * (g - mean(g) - xhat * mean(g xhat));  dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short* __restrict__ z,
                                                           const unsigned short* __restrict__ dy,
                                                           const unsigned short* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, long long n8, int C,
                                                           int relu, unsigned short* __restrict__ dz,
                                                           unsigned short* __restrict__ dres) {
  const int G = C >> 3;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long long)gridDim.x * 256) {
    const int g = (int)(e % G);
    float zv[8], gv[8], yv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(z + e * 8), zv);
    unpack8(*reinterpret_cast<const uint4*>(dy + e * 8), gv);
    if (relu) unpack8(*reinterpret_cast<const uint4*>(y + e * 8), yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      const float gk = (relu && !(yv[k] > 0.f)) ? 0.f : gv[k];
      gv[k] = gk;
      const float xhat = (zv[k] - mean[c]) * invstd[c];
      o[k] = coef[2 * C + c] * (gk - coef[c] - xhat * coef[C + c]);
    }
    *reinterpret_cast<uint4*>(dz + e * 8) = pack8(o);
    if (dres) *reinterpret_cast<uint4*>(dres + e * 8) = pack8(gv);
  }
}

inline int red_blocks(long long M) {
  long long b = M / 64;
  if (b < 1) b = 1;
  return (int)(b > RED_BLOCKS ? RED_BLOCKS : b);
}
inline bool bn_shape_ok(long long M, int C) {
  // C/8 channel groups must divide 256 threads: C in {8,16,...,2048} with 256 % (C/8) == 0
  return M > 0 && C >= 8 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0;
}
inline int ew_grid(long long n8) {
  const long long g = (n8 + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" size_t lss_bn_train_workspace_bytes(long long M, int C) {
  if (!bn_shape_ok(M, C)) return 0;
  return ((size_t)red_blocks(M) * 2 * C + 3 * (size_t)C) * sizeof(float);
}

extern "C" int lss_bn_train_fwd(const void* z, const void* residual, long long M, int C, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum,
                                float eps, int relu, void* workspace, void* y, float* save_mean,
                                float* save_invstd, void* stream) {
  LSS_CHECK_PTR(z); LSS_CHECK_PTR(gamma); LSS_CHECK_PTR(beta); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(y);
  LSS_CHECK_PTR(save_mean); LSS_CHECK_PTR(save_invstd);
  if (!bn_shape_ok(M, C)) return LSS_E_SHAPE;
  if ((running_mean == nullptr) != (running_var == nullptr)) return LSS_E_NULL;
  hipStream_t st = lss_stream(stream);
  const int nblk = red_blocks(M);
  float* part = static_cast<float*>(workspace);
  float* scale = part + (size_t)nblk * 2 * C;
  float* shift = scale + C;
  const unsigned short* zz = static_cast<const unsigned short*>(z);
  hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk), dim3(256), 0, st, zz, nullptr, nullptr, nullptr, nullptr, M, C,
                     0, part, nullptr, zz);  // pivot = the tensor's first row
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(lss_cdiv(C, 4)), dim3(256), 0, st, part, nblk, M, C, gamma, beta,
                     eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd, nullptr, zz);
  const long long n8 = M * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n8)), dim3(256), 0, st, zz,
                     static_cast<const unsigned short*>(residual), scale, shift, n8, C, relu,
                     static_cast<unsigned short*>(y));
  return lss_launch_status();
}

extern "C" int lss_bn_train_bwd(const void* dy, const void* y, const void* z, long long M, int C,
                                const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                                void* workspace, void* dz, void* dres, float* dgamma, float* dbeta, void* stream) {
  LSS_CHECK_PTR(dy); LSS_CHECK_PTR(z); LSS_CHECK_PTR(gamma); LSS_CHECK_PTR(save_mean); LSS_CHECK_PTR(save_invstd);
  LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(dz); LSS_CHECK_PTR(dgamma); LSS_CHECK_PTR(dbeta);
  if (relu && y == nullptr) return LSS_E_NULL;
  if (!bn_shape_ok(M, C)) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  const int nblk = red_blocks(M);
  float* part = static_cast<float*>(workspace);
  float* coef = part + (size_t)nblk * 2 * C;
  const unsigned short* zz = static_cast<const unsigned short*>(z);
  const unsigned short* gg = static_cast<const unsigned short*>(dy);
  const unsigned short* yy = static_cast<const unsigned short*>(y);
  hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nblk), dim3(256), 0, st, zz, gg, yy, save_mean, save_invstd, M, C,
                     relu, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(lss_cdiv(C, 4)), dim3(256), 0, st, part, nblk, M, C, gamma,
                     save_invstd, dgamma, dbeta, coef);
  const long long n8 = M * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n8)), dim3(256), 0, st, zz, gg, yy, save_mean, save_invstd,
                     coef, n8, C, relu, static_cast<unsigned short*>(dz), static_cast<unsigned short*>(dres));
  return lss_launch_status();
}

// ---------------------------------------------------------------------------
// One-call training units: conv3x3 (optionally over cat([x2, upsample(x1)])) -> BatchNorm(train)
// -> (+residual) -> ReLU, forward and backward, as ONE host call each.  The training step is
// framework-bound (a Python autograd node costs more than most of these kernels run), so the
// host side hands over every pointer once and the launches are chained here.
// LSS_CONV_RING=0: the tile kernel everywhere (developer A/B, same switch as ops.conv_ring_ok)
static bool train_ring_enabled() {
  const char* e = getenv("LSS_CONV_RING");
  return e == nullptr || atoi(e) != 0;
}

// The weight image a training unit of this shape runs on - forward (dgrad = 0) or input-gradient conv (dgrad = 1):
// ring kernel where the shape qualifies, K-split kernel for the launch-bound layers, tile kernel otherwise.  ONE place
// decides, for the units below and for lss_conv_bn_act_train_pack (which lets a host pack every layer's images ahead of
// the step, ops.WeightPrepack).
static int train_unit_pack(const float* w_oihw, int B, int H, int W, int Cx, int C2, int up, int Cout, int dgrad,
                           void* w_packed, int* flags, void* stream) {
  const int Hh = H * up, Wh = W * up, Ct = Cx + C2;
  bool ring, ks;
  if (!dgrad) {
    ring = train_ring_enabled() && lss_conv2d_ring_ok(B, H, W, Cx, C2, up, Cout, 0);
    ks = !ring && C2 == 0 && up == 1 && lss_conv2d_ks_ok(B, H, W, Cx, Cout);
  } else {  // the gradient conv: Cout -> Ct channels on the (upsampled) grid
    ring = train_ring_enabled() && lss_conv2d_ring_ok(B, Hh, Wh, Cout, 0, 1, Ct, 0);
    ks = !ring && lss_conv2d_ks_ok(B, Hh, Wh, Cout, Ct);
  }
  if (flags != nullptr) *flags = (ring ? LSS_W_RING : 0) | (ks ? LSS_W_KS : 0);
  if (w_oihw == nullptr) return 0;  // the caller packed ahead of the step
  if (!dgrad)
    return ring ? lss_conv2d_pack_weights_ring(w_oihw, Cout, Ct, w_packed, stream)
           : ks ? lss_conv2d_pack_weights_ks(w_oihw, Cout, Cx, w_packed, stream)
                : lss_conv2d_pack_weights(w_oihw, Cout, Ct, 3, 3, LSS_DT_BF16, w_packed, stream);
  return ring ? lss_conv2d_pack_weights_ring_dgrad(w_oihw, Cout, Ct, w_packed, stream)
         : ks ? lss_conv2d_pack_weights_ks_dgrad(w_oihw, Cout, Ct, w_packed, stream)
              : lss_conv2d_pack_weights_dgrad(w_oihw, Cout, Ct, 3, 3, LSS_DT_BF16, w_packed, stream);
}

extern "C" int lss_conv_bn_act_train_pack(const float* w_oihw, int B, int H, int W, int Cx, int C2, int up, int Cout,
                                          int dgrad, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  if (dgrad != 0 && dgrad != 1) return LSS_E_SHAPE;
  return train_unit_pack(w_oihw, B, H, W, Cx, C2, up, Cout, dgrad, w_packed, nullptr, stream);
}

extern "C" int lss_conv_bn_act_train_fwd(const void* x1, const void* x2, const float* w_oihw, const float* gamma,
                                         const float* beta, const void* residual, float* running_mean,
                                         float* running_var, void* w_packed, void* z, void* y, float* save_mean,
                                         float* save_invstd, void* bn_workspace, int B, int H, int W, int Cx, int C2,
                                         int up, int Cout, float momentum, float eps, int relu, void* stream) {
  // the big layers run on the loader / consumer ring kernel (conv_ring.hip), like the inference path, the launch-bound
  // ones (layer1-3) on the K-split one-pass kernel (conv_ks.hip); w_oihw == NULL: w_packed already holds the image
  // (lss_conv_bn_act_train_pack, or a gather-pack of the whole model: lss_gather_pack)
  LSS_CHECK_PTR(w_packed);
  int wflags = 0;
  int rc = train_unit_pack(w_oihw, B, H, W, Cx, C2, up, Cout, 0, w_packed, &wflags, stream);
  if (rc != 0) return rc;
  rc = lss_conv2d_fwd(x1, x2, w_packed, nullptr, nullptr, nullptr, z, nullptr, B, H, W, Cx, C2, up, Cout, 3, 3, 1, 1,
                      LSS_ACT_NONE | wflags, LSS_DT_BF16, stream);
  if (rc != 0) return rc;
  const long long M = (long long)B * (H * up) * (W * up);
  return lss_bn_train_fwd(z, residual, M, Cout, gamma, beta, running_mean, running_var, momentum, eps, relu,
                          bn_workspace, y, save_mean, save_invstd, stream);
}

extern "C" int lss_conv_bn_act_train_bwd(const void* dy, const void* y, const void* z, const void* x1, const void* x2,
                                         const float* w_oihw, const float* gamma, const float* save_mean,
                                         const float* save_invstd, void* bn_workspace, void* wgrad_workspace,
                                         size_t wgrad_workspace_bytes, void* w_dgrad, void* dz, void* dres,
                                         float* dgamma, float* dbeta, void* gcat, void* g1, void* xcat, float* dw,
                                         int B, int H, int W, int Cx, int C2, int up, int Cout, int relu,
                                         void* stream) {
  const int Hh = H * up, Wh = W * up, Ct = Cx + C2;
  const long long M = (long long)B * Hh * Wh;
  int rc = lss_bn_train_bwd(dy, y, z, M, Cout, gamma, save_mean, save_invstd, relu, bn_workspace, dz, dres, dgamma,
                            dbeta, stream);
  if (rc != 0) return rc;
  if (gcat != nullptr) {  // input gradient(s): dgrad conv over the (concatenated, upsampled) input
    LSS_CHECK_PTR(w_dgrad);
    int wflags = 0;   // (w_oihw == NULL: w_dgrad was packed ahead of the step)
    rc = train_unit_pack(w_oihw, B, H, W, Cx, C2, up, Cout, 1, w_dgrad, &wflags, stream);
    if (rc != 0) return rc;
    rc = lss_conv2d_fwd(dz, nullptr, w_dgrad, nullptr, nullptr, nullptr, gcat, nullptr, B, Hh, Wh, Cout, 0, 1, Ct, 3, 3,
                        1, 1, LSS_ACT_NONE | wflags, LSS_DT_BF16, stream);
    if (rc != 0) return rc;
    if (g1 != nullptr) {
      rc = lss_upsample_bwd_nhwc(gcat, B, H, W, Cx, Ct, C2, up, g1, stream);
      if (rc != 0) return rc;
    }
  }
  if (dw != nullptr) {
    const void* xin = x1;
    if (up > 1 || C2 > 0) {
      LSS_CHECK_PTR(xcat);
      rc = lss_upsample_cat_nhwc(x1, x2, B, H, W, Cx, C2, up, xcat, stream);
      if (rc != 0) return rc;
      xin = xcat;
    }
    rc = lss_conv2d_wgrad(xin, dz, B, Hh, Wh, Ct, Cout, wgrad_workspace, wgrad_workspace_bytes, dw, stream);
  }
  return rc;
}

// ---------------------------------------------------------------------------
// Split forms for synchronised BatchNorm under data parallelism: the per-channel sums leave the
// device function as a (2, C) vector, the caller all-reduces them over the ranks (RCCL), and the
// second half normalises with the GLOBAL statistics - so W ranks x B/W samples compute exactly the
// batch statistics of one device holding B samples.
namespace {

// sums[which][c] = fixed-order sum of the per-workgroup partials
__global__ void bn_collapse_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float s1, s2;
  sum_partials(part, nblk, C, c, s1, s2);
  if ((threadIdx.x & 63) == 0) {
    sums[c] = s1;
    sums[C + c] = s2;
  }
}

}  // namespace

// mode 0: sums = (sum (z - p), sum (z - p)^2) about the pivot p = `mean` (a vector shared by all ranks, e.g. the running
// mean; NULL = 0) - pass the SAME vector as running_mean to lss_bn_train_fwd_from_sums;
// mode 1: sums = (sum g, sum g * xhat) with g = dy * [y > 0] (relu) or dy
extern "C" int lss_bn_partial_sums(const void* z, const void* dy, const void* y, const float* mean,
                                   const float* invstd, long long M, int C, int relu, int mode, void* workspace,
                                   float* sums, void* stream) {
  LSS_CHECK_PTR(z); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(sums);
  if (!bn_shape_ok(M, C) || (mode != 0 && mode != 1)) return LSS_E_SHAPE;
  if (mode == 1 && (dy == nullptr || mean == nullptr || invstd == nullptr || (relu && y == nullptr))) return LSS_E_NULL;
  hipStream_t st = lss_stream(stream);
  const int nblk = red_blocks(M);
  float* part = static_cast<float*>(workspace);
  const unsigned short* zz = static_cast<const unsigned short*>(z);
  if (mode == 0)
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk), dim3(256), 0, st, zz, nullptr, nullptr, nullptr, nullptr, M,
                       C, 0, part, mean, nullptr);
  else
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nblk), dim3(256), 0, st, zz, static_cast<const unsigned short*>(dy),
                       static_cast<const unsigned short*>(y), mean, invstd, M, C, relu, part);
  hipLaunchKernelGGL(bn_collapse_kernel, dim3(lss_cdiv(C, 4)), dim3(256), 0, st, part, nblk, C, sums);
  return lss_launch_status();
}

// forward, second half: statistics from `sums` over M_total rows (all ranks), applied to this rank's M rows
extern "C" int lss_bn_train_fwd_from_sums(const void* z, const void* residual, long long M, int C,
                                          const float* sums, long long M_total, const float* gamma,
                                          const float* beta, float* running_mean, float* running_var,
                                          float momentum, float eps, int relu, void* workspace, void* y,
                                          float* save_mean, float* save_invstd, void* stream) {
  LSS_CHECK_PTR(z); LSS_CHECK_PTR(sums); LSS_CHECK_PTR(gamma); LSS_CHECK_PTR(beta); LSS_CHECK_PTR(workspace);
  LSS_CHECK_PTR(y); LSS_CHECK_PTR(save_mean); LSS_CHECK_PTR(save_invstd);
  if (!bn_shape_ok(M, C) || M_total < M) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  float* scale = static_cast<float*>(workspace) + (size_t)red_blocks(M) * 2 * C;
  float* shift = scale + C;
  // `sums` were taken about running_mean (lss_bn_partial_sums mode 0 with mean = running_mean), read before its update
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(lss_cdiv(C, 4)), dim3(256), 0, st, sums, 1, M_total, C, gamma, beta,
                     eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd, running_mean,
                     nullptr);
  const long long n8 = M * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n8)), dim3(256), 0, st, static_cast<const unsigned short*>(z),
                     static_cast<const unsigned short*>(residual), scale, shift, n8, C, relu,
                     static_cast<unsigned short*>(y));
  return lss_launch_status();
}

// backward, second half: `sums` = all-reduced (sum g, sum g xhat) over M_total rows.  dgamma / dbeta
// written here are those GLOBAL sums; a data-parallel caller keeps its LOCAL sums as the parameter
// gradients (the gradient all-reduce averages them) and uses this call for dz only.
extern "C" int lss_bn_train_bwd_from_sums(const void* dy, const void* y, const void* z, long long M, int C,
                                          const float* sums, long long M_total, const float* gamma,
                                          const float* save_mean, const float* save_invstd, int relu,
                                          void* workspace, void* dz, void* dres, float* dgamma, float* dbeta,
                                          void* stream) {
  LSS_CHECK_PTR(dy); LSS_CHECK_PTR(z); LSS_CHECK_PTR(sums); LSS_CHECK_PTR(gamma); LSS_CHECK_PTR(save_mean);
  LSS_CHECK_PTR(save_invstd); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(dz); LSS_CHECK_PTR(dgamma); LSS_CHECK_PTR(dbeta);
  if (relu && y == nullptr) return LSS_E_NULL;
  if (!bn_shape_ok(M, C) || M_total < M) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  float* coef = static_cast<float*>(workspace) + (size_t)red_blocks(M) * 2 * C;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(lss_cdiv(C, 4)), dim3(256), 0, st, sums, 1, M_total, C, gamma,
                     save_invstd, dgamma, dbeta, coef);
  const long long n8 = M * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n8)), dim3(256), 0, st, static_cast<const unsigned short*>(z),
                     static_cast<const unsigned short*>(dy), static_cast<const unsigned short*>(y), save_mean,
                     save_invstd, coef, n8, C, relu, static_cast<unsigned short*>(dz),
                     static_cast<unsigned short*>(dres));
  return lss_launch_status();
}
