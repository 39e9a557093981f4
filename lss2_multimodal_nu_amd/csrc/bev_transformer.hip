// Token-major kernels of the BEV transformer (ref: src/transformer_modules.py):
//   add_pos       q = src + sine position table            (:199-200)
//   deform_attn   softmax over points, sampling locations, bilinear zero-padded
//                 gather of the projected values, weighted sum (:117-156)
//   layernorm     nn.LayerNorm(256) over the channel row   (:204, :208)
// The linear layers around them are 1x1 convs on the MFMA conv kernel
// (conv_mfma.hip).  Activations are (B, H*W, 256) rows = NHWC; all three kernels
// are bandwidth / gather bound: one wave per token, 4 channels per lane, 16-B or
// 8-B accesses, no LDS.
#include <type_traits>

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

// The kernel is VALU-issue bound (address arithmetic, bf16 unpacking, the blend), not cache
// bound, so the mapping minimises instructions per token:
// one wave = 2 query tokens; lane = 32*tok + 4*head + sub, sub owning channels 8*sub..8*sub+7
// of its head (one 16-B load per tap for bf16, two for fp32).
//   phase 1: lane (tok, head, sub) prepares sampling points 2*sub and 2*sub+1 of its
//            (token, head): softmax weight over the head's 8 points (2 xor-shuffles over the 4
//            lanes + the lane's own pair), sampling location, 4 tap offsets + weights;
//   phase 2: for each point, the 4 lanes of a (token, head) fetch its 4 taps (parameters
//            broadcast by shuffle from the lane that prepared the point) and blend them with
//            packed fp32 FMAs.
// ol (rows, 192) fp32: [0,128) offsets (head, point, xy), [128,192) attention logits.
// value strides (elements): pixel pstr, head hstr, sample bstr.
struct TapSet {
  int i00, i01, i10, i11;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ TapSet make_taps(float offx, float offy, float aw, float rx, float ry, int H,
                                            int W) {
  // sampling location (ref :124-125: BOTH offsets are divided by H), then grid_sample's
  // own arithmetic (align_corners=False): grid = 2 loc - 1, pixel = ((grid + 1) size - 1) / 2
  const float fH = (float)H;
  float lx = rx + offx / fH, ly = ry + offy / fH;
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
  TapSet t;
  // weights of the 4 taps times the attention weight; out-of-image taps contribute 0
  t.w00 = (xin0 && yin0) ? wx0 * wy0 * aw : 0.f;
  t.w01 = (xin1 && yin0) ? wx1 * wy0 * aw : 0.f;
  t.w10 = (xin0 && yin1) ? wx0 * wy1 * aw : 0.f;
  t.w11 = (xin1 && yin1) ? wx1 * wy1 * aw : 0.f;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
  t.i00 = cy0 * W + cx0; t.i01 = cy0 * W + cx1; t.i10 = cy1 * W + cx0; t.i11 = cy1 * W + cx1;
  return t;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// acc[0..7] += w * (8 channels at p)
__device__ __forceinline__ void blend8(f32x2 (&acc)[4], const unsigned short* p, float w) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const unsigned int u[4] = {v.x, v.y, v.z, v.w};
  const f32x2 ww = (f32x2){w, w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 c = (f32x2){__builtin_bit_cast(float, u[k] << 16), __builtin_bit_cast(float, u[k] & 0xffff0000u)};
    acc[k] = __builtin_elementwise_fma(c, ww, acc[k]);
  }
}
__device__ __forceinline__ void blend8(f32x2 (&acc)[4], const float* p, float w) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  const f32x2 ww = (f32x2){w, w};
  acc[0] = __builtin_elementwise_fma((f32x2){a[0], a[1]}, ww, acc[0]);
  acc[1] = __builtin_elementwise_fma((f32x2){a[2], a[3]}, ww, acc[1]);
  acc[2] = __builtin_elementwise_fma((f32x2){b[0], b[1]}, ww, acc[2]);
  acc[3] = __builtin_elementwise_fma((f32x2){b[2], b[3]}, ww, acc[3]);
}
__device__ __forceinline__ void store8(unsigned short* p, const f32x2 (&a)[4]) {
  uint4 o;
  o.x = lss_pack_bf2(a[0][0], a[0][1]); o.y = lss_pack_bf2(a[1][0], a[1][1]);
  o.z = lss_pack_bf2(a[2][0], a[2][1]); o.w = lss_pack_bf2(a[3][0], a[3][1]);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const f32x2 (&a)[4]) {
  *reinterpret_cast<f32x4*>(p) = (f32x4){a[0][0], a[0][1], a[1][0], a[1][1]};
  *reinterpret_cast<f32x4*>(p + 4) = (f32x4){a[2][0], a[2][1], a[3][0], a[3][1]};
}

template <typename T>
__global__ __launch_bounds__(256) void deform_attn_kernel(
    const T* __restrict__ value, const float* __restrict__ ol, const float* __restrict__ tok_bias,
    const float* __restrict__ ref_x, const float* __restrict__ ref_y, int B, int H, int W,
    long long pstr, long long hstr, long long bstr, T* __restrict__ out) {
  const int lane = threadIdx.x & 63, tok = lane >> 5, head = (lane >> 2) & 7, sub = lane & 3;
  const long long rows = (long long)B * H * W;
  const long long want = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + tok;
  const bool live = want < rows;
  const long long row = live ? want : rows - 1;  // a tail half-wave redoes the last token
  const int t = (int)(row % ((long long)H * W));
  const int b = (int)(row / ((long long)H * W));
  const float* r = ol + row * 192;
  // points 2*sub, 2*sub+1 of this head: offsets (x0,y0,x1,y1) and the two logits
  f32x4 off = *reinterpret_cast<const f32x4*>(r + 16 * head + 4 * sub);
  float2 lg = *reinterpret_cast<const float2*>(r + 128 + 8 * head + 2 * sub);
  if (tok_bias) {  // the position encoding's share of the two linears: (pos @ W^T)[t, :]
    const float* pb = tok_bias + (size_t)t * 192;
    const f32x4 po = *reinterpret_cast<const f32x4*>(pb + 16 * head + 4 * sub);
    const float2 pl = *reinterpret_cast<const float2*>(pb + 128 + 8 * head + 2 * sub);
    off += po;
    lg.x += pl.x;
    lg.y += pl.y;
  }
  // softmax over the 8 points of this (token, head) (ref :121-122)
  float mx = fmaxf(lg.x, lg.y);
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  const float e0 = expf(lg.x - mx), e1 = expf(lg.y - mx);
  float sum = e0 + e1;
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  const float rx = ref_x[t % W], ry = ref_y[t / W];
  const TapSet ta = make_taps(off[0], off[1], e0 / sum, rx, ry, H, W);
  const TapSet tb = make_taps(off[2], off[3], e1 / sum, rx, ry, H, W);

  const T* vb = value + (size_t)b * bstr + (size_t)head * hstr + 8 * sub;
  f32x2 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x2){0.f, 0.f};
  // one step = the two points prepared by lane q of the (token, head) quad: 8 taps in flight per lane.
  // The tap parameters come from quad lane q by DPP quad_perm broadcasts (VALU moves; as ds_bpermute
  // shuffles they were 64 LDS operations per lane).  The scheduling barrier between the steps keeps the compiler from
  // hoisting all 32 tap loads to the top (occupancy: this kernel lives on many waves hiding gather latency).
  auto step = [&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    constexpr int CTRL = Q | (Q << 2) | (Q << 4) | (Q << 6);  // quad_perm [Q, Q, Q, Q]
    auto bi = [&](int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); };
    auto bf = [&](float v) { return __builtin_bit_cast(float, bi(__builtin_bit_cast(int, v))); };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const TapSet& m = half ? tb : ta;
      const int j00 = bi(m.i00), j01 = bi(m.i01), j10 = bi(m.i10), j11 = bi(m.i11);
      const float u00 = bf(m.w00), u01 = bf(m.w01), u10 = bf(m.w10), u11 = bf(m.w11);
      blend8(acc, vb + (size_t)j00 * pstr, u00);
      blend8(acc, vb + (size_t)j01 * pstr, u01);
      blend8(acc, vb + (size_t)j10 * pstr, u10);
      blend8(acc, vb + (size_t)j11 * pstr, u11);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // each step sits behind its own opaque (always true) scalar condition: control dependence is what
  // actually stops the tap loads of later steps from being scheduled first
  auto gate = []() { int one; asm volatile("s_mov_b32 %0, 1" : "=s"(one)); return one != 0; };
  if (gate()) step(std::integral_constant<int, 0>{});
  if (gate()) step(std::integral_constant<int, 1>{});
  if (gate()) step(std::integral_constant<int, 2>{});
  if (gate()) step(std::integral_constant<int, 3>{});
  if (live) store8(out + row * TC + head * 32 + 8 * sub, acc);
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

extern "C" int lss_deform_attn_fwd(const void* value, int value_layout, const float* offsets_logits,
                                   const float* token_bias, const float* ref_x, const float* ref_y,
                                   int B, int H, int W, int n_heads, int n_points, int C, int dt,
                                   void* out, void* stream) {
  LSS_CHECK_PTR(value); LSS_CHECK_PTR(offsets_logits); LSS_CHECK_PTR(ref_x); LSS_CHECK_PTR(ref_y);
  LSS_CHECK_PTR(out);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W);
  if (n_heads != 8 || n_points != 8 || C != TC) return LSS_E_SHAPE;  // the reference's configuration
  if (value_layout != LSS_VALUE_NHWC && value_layout != LSS_VALUE_HEAD_MAJOR) return LSS_E_LAYOUT;
  if (!aligned16(value) || !aligned16(offsets_logits) || !aligned16(out)) return LSS_E_ALIGN;
  const long long rows = (long long)B * H * W;
  if (rows >= (1LL << 31)) return LSS_E_SHAPE;
  const long long HW = (long long)H * W;
  const long long pstr = value_layout == LSS_VALUE_NHWC ? TC : 32;
  const long long hstr = value_layout == LSS_VALUE_NHWC ? 32 : HW * 32;
  const long long bstr = HW * TC;
  dim3 grid(lss_cdiv(rows, 8));
  hipStream_t st = lss_stream(stream);
  if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(deform_attn_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(value),
                       offsets_logits, token_bias, ref_x, ref_y, B, H, W, pstr, hstr, bstr,
                       static_cast<float*>(out));
  else if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(deform_attn_kernel<unsigned short>, grid, dim3(256), 0, st,
                       static_cast<const unsigned short*>(value), offsets_logits, token_bias, ref_x, ref_y,
                       B, H, W, pstr, hstr, bstr, static_cast<unsigned short*>(out));
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
