// Token-major kernels of the BEV transformer (ref: src/transformer_modules.py):
//   add_pos       q = src + sine position table            (:199-200)
//   deform_attn   softmax over points, sampling locations, bilinear zero-padded
//                 gather of the projected values, weighted sum (:117-156)
//   layernorm     nn.LayerNorm(256) over the channel row   (:204, :208)
// The linear layers around them are 1x1 convs on the MFMA conv kernel
// (conv_mfma.hip).  Activations are (B, H*W, 256) rows = NHWC; all three kernels
// are bandwidth / gather bound: one wave per token, 4 channels per lane, 16-B or
// 8-B accesses, no LDS.
#include "lss_common.h"

namespace {

constexpr int TC = 256;  // d_model of the reference's transformer (8 heads x 32 channels)

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load4<unsigned short>(const unsigned short* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return (f32x4){lss_bf2f((unsigned short)(v.x & 0xffff)), lss_bf2f((unsigned short)(v.x >> 16)),
                 lss_bf2f((unsigned short)(v.y & 0xffff)), lss_bf2f((unsigned short)(v.y >> 16))};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(unsigned short* p, f32x4 v) {
  uint2 o;
  o.x = lss_pack_bf2(v[0], v[1]);
  o.y = lss_pack_bf2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = o;
}

// q[b, t, :] = x[b, t, :] + pos[t, :]
template <typename T>
__global__ __launch_bounds__(256) void add_pos_kernel(const T* __restrict__ x,
                                                      const float* __restrict__ pos, long long rows,
                                                      int T_tok, T* __restrict__ q) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int t = (int)(row % T_tok);
  const f32x4 a = load4<T>(x + row * TC + 4 * lane);
  const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)t * TC + 4 * lane);
  store4(q + row * TC + 4 * lane, (f32x4){a[0] + p[0], a[1] + p[1], a[2] + p[2], a[3] + p[3]});
}

// One wave per query token.  lane = 8*head + s:
//   phase 1: lane (head, s) owns sampling point s of its head: softmax weight over the
//            head's 8 points (3 xor-shuffles), sampling location, 4 tap offsets + weights;
//   phase 2: for each point, the 8 lanes of a head fetch that point's taps (broadcast by
//            shuffle) - lane s reads channels 4s..4s+3 of the head's 32, i.e. the 8 lanes
//            read one contiguous 64-B (bf16) / 128-B (fp32) row segment per tap.
// ol (rows, 192) fp32: [0,128) offsets (head, point, xy), [128,192) attention logits.
template <typename T>
__global__ __launch_bounds__(256) void deform_attn_kernel(
    const T* __restrict__ value, const float* __restrict__ ol, const float* __restrict__ ref_x,
    const float* __restrict__ ref_y, int B, int H, int W, T* __restrict__ out) {
  const int lane = threadIdx.x & 63, head = lane >> 3, s = lane & 7;
  const long long rows = (long long)B * H * W;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // whole waves leave together
  const int t = (int)(row % ((long long)H * W));
  const int b = (int)(row / ((long long)H * W));
  const float* r = ol + row * 192;
  const float2 off = *reinterpret_cast<const float2*>(r + 2 * lane);
  const float logit = r[128 + lane];
  // softmax over the 8 points of this head (ref :121-122)
  float mx = logit;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e = expf(logit - mx);
  float sum = e;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float aw = e / sum;
  // sampling location (ref :124-125: BOTH offsets are divided by H), then grid_sample's
  // own arithmetic (align_corners=False): grid = 2 loc - 1, pixel = ((grid + 1) size - 1) / 2
  const float fH = (float)H;
  float lx = ref_x[t % W] + off.x / fH;
  float ly = ref_y[t / W] + off.y / fH;
  lx = fminf(fmaxf(lx, 0.f), 1.f);
  ly = fminf(fmaxf(ly, 0.f), 1.f);
  const float gx = lx * 2.0f - 1.0f, gy = ly * 2.0f - 1.0f;
  const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  const float py = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  const float x0f = floorf(px), y0f = floorf(py);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float wx1 = px - x0f, wx0 = (x0f + 1.f) - px;
  const float wy1 = py - y0f, wy0 = (y0f + 1.f) - py;
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  // weights of the 4 taps times the attention weight; out-of-image taps contribute 0
  const float w00 = (xin0 && yin0) ? wx0 * wy0 * aw : 0.f;
  const float w01 = (xin1 && yin0) ? wx1 * wy0 * aw : 0.f;
  const float w10 = (xin0 && yin1) ? wx0 * wy1 * aw : 0.f;
  const float w11 = (xin1 && yin1) ? wx1 * wy1 * aw : 0.f;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
  const int i00 = cy0 * W + cx0, i01 = cy0 * W + cx1, i10 = cy1 * W + cx0, i11 = cy1 * W + cx1;

  const T* vb = value + ((size_t)b * H * W) * TC + head * 32 + 4 * s;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int src = (lane & ~7) | p;
    const int j00 = __shfl(i00, src, 64), j01 = __shfl(i01, src, 64);
    const int j10 = __shfl(i10, src, 64), j11 = __shfl(i11, src, 64);
    const float u00 = __shfl(w00, src, 64), u01 = __shfl(w01, src, 64);
    const float u10 = __shfl(w10, src, 64), u11 = __shfl(w11, src, 64);
    const f32x4 a = load4<T>(vb + (size_t)j00 * TC), c = load4<T>(vb + (size_t)j01 * TC);
    const f32x4 d = load4<T>(vb + (size_t)j10 * TC), g = load4<T>(vb + (size_t)j11 * TC);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += a[k] * u00 + c[k] * u01 + d[k] * u10 + g[k] * u11;
  }
  store4(out + row * TC + head * 32 + 4 * s, acc);
}

// y = (x - mean) * rsqrt(var + eps) * gamma + beta over the 256 channels of a row
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long long rows,
                                                        float eps, TO* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 v = load4<TI>(x + row * TC + 4 * lane);
  const float mean = lss_wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / TC);
  const f32x4 d = (f32x4){v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  const float var = lss_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / TC);
  const float inv = rsqrtf(var + eps);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * lane);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
  store4(y + row * TC + 4 * lane, (f32x4){d[0] * inv * g[0] + bt[0], d[1] * inv * g[1] + bt[1],
                                          d[2] * inv * g[2] + bt[2], d[3] * inv * g[3] + bt[3]});
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int lss_add_pos_fwd(const void* x, const float* pos, int B, int T, int C, int dt, void* q,
                               void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(pos); LSS_CHECK_PTR(q);
  LSS_CHECK_POS(B); LSS_CHECK_POS(T);
  if (C != TC) return LSS_E_SHAPE;
  if (!aligned16(x) || !aligned16(pos) || !aligned16(q)) return LSS_E_ALIGN;
  const long long rows = (long long)B * T;
  if (rows >= (1LL << 31)) return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(rows, 4));
  hipStream_t st = lss_stream(stream);
  if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(add_pos_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(x), pos, rows, T,
                       static_cast<float*>(q));
  else if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(add_pos_kernel<unsigned short>, grid, dim3(256), 0, st,
                       static_cast<const unsigned short*>(x), pos, rows, T, static_cast<unsigned short*>(q));
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}

extern "C" int lss_deform_attn_fwd(const void* value, const float* offsets_logits, const float* ref_x,
                                   const float* ref_y, int B, int H, int W, int n_heads, int n_points,
                                   int C, int dt, void* out, void* stream) {
  LSS_CHECK_PTR(value); LSS_CHECK_PTR(offsets_logits); LSS_CHECK_PTR(ref_x); LSS_CHECK_PTR(ref_y);
  LSS_CHECK_PTR(out);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W);
  if (n_heads != 8 || n_points != 8 || C != TC) return LSS_E_SHAPE;  // the reference's configuration
  if (!aligned16(value) || !aligned16(offsets_logits) || !aligned16(out)) return LSS_E_ALIGN;
  const long long rows = (long long)B * H * W;
  if (rows >= (1LL << 31)) return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(rows, 4));
  hipStream_t st = lss_stream(stream);
  if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(deform_attn_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(value),
                       offsets_logits, ref_x, ref_y, B, H, W, static_cast<float*>(out));
  else if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(deform_attn_kernel<unsigned short>, grid, dim3(256), 0, st,
                       static_cast<const unsigned short*>(value), offsets_logits, ref_x, ref_y, B, H, W,
                       static_cast<unsigned short*>(out));
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}

extern "C" int lss_layernorm_fwd(const void* x, int x_dt, const float* gamma, const float* beta,
                                 long long rows, int C, float eps, void* y, int y_dt, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(gamma); LSS_CHECK_PTR(beta); LSS_CHECK_PTR(y);
  if (rows <= 0 || rows >= (1LL << 31)) return LSS_E_SHAPE;
  if (C != TC) return LSS_E_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(y)) return LSS_E_ALIGN;
  dim3 grid(lss_cdiv(rows, 4));
  hipStream_t st = lss_stream(stream);
#define LSS_LN(TI, TO)                                                                              \
  hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, dim3(256), 0, st, static_cast<const TI*>(x), \
                     gamma, beta, rows, eps, static_cast<TO*>(y))
  if (x_dt == LSS_DT_F32 && y_dt == LSS_DT_F32) LSS_LN(float, float);
  else if (x_dt == LSS_DT_F32 && y_dt == LSS_DT_BF16) LSS_LN(float, unsigned short);
  else if (x_dt == LSS_DT_BF16 && y_dt == LSS_DT_BF16) LSS_LN(unsigned short, unsigned short);
  else if (x_dt == LSS_DT_BF16 && y_dt == LSS_DT_F32) LSS_LN(unsigned short, float);
  else return LSS_E_LAYOUT;
#undef LSS_LN
  return lss_launch_status();
}
