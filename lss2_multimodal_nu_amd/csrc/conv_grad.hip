// Gradients of the 3x3 / stride 1 / pad 1 bf16 NHWC convolution (the shape that carries
// 95 % of BevEncode's FLOPs): the backward half of conv_mfma.hip's forward kernels.
// replaces: the autograd nodes ConvolutionBackward of the convs in src/modules.py:22-27,
//           118-130 (+ torchvision BasicBlock) that `loss.backward()` runs in train.py:61.
//
// dgrad  dX = conv3x3(dY, W') with W'[ci][ky][kx][co] = W[co][2-ky][2-kx][ci]: the FORWARD
//        kernel on a differently packed weight (lss_conv2d_pack_weights_dgrad), nothing else.
// wgrad  dW[co][ci][ky][kx] = sum_{b,y,x} dY[b,y,x,co] * X[b,y+ky-1,x+kx-1,ci] is, per tap, a
//        GEMM whose K dimension is the pixel index.  NHWC has K as the SLOW dimension of both
//        operands, so both are first transposed to channel-major rows over a zero-padded
//        image grid (B, H+2, Wp), Wp = W+2 rounded up to 8:
//          dYt[co][flat(b,y+1,x+1)] = dY[b,y,x,co]                       (zeros on the border)
//          Xt_d[ci][flat(b,y',x')]  = Xpad[b][y'][x'+d],  d = -1, 0, +1  (three column-shifted copies)
//        Then tap (ky,kx) is the plain K-contiguous GEMM  dYt . Xt_{kx-1}^T  with the second
//        operand read (ky-1)*Wp elements further on - always 16-B aligned, which is why the
//        column shift is materialised and the row shift is not.  The GEMM is linear_mfma.hip's
//        kernel in split-K mode (grid.y = split x tap, fp32 partial tiles), followed by a
//        fixed-order reduction over the splits (bit-reproducible; no atomics).
#include <stdlib.h>

#include "lss_common.h"

int lss_wgrad_gemm_launch(const void* dyt, const void* const xt[3], float* partial, int M, int N,
                          long long ld, long long split_k, int nsplit, int ntap, long long tap_row,
                          hipStream_t st);  // linear_mfma.hip
// conv_wgrad.hip: the direct kernel (NHWC operands as they lie, transposed reads); 0 splits = not a case for it
int lss_wgrad_direct_splits(int B, int H, int W, int Cin, int Cout);
int lss_wgrad_direct_launch(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, float* partial,
                            hipStream_t st);
int lss_wgrad_taps4x4_launch(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, float* partial,
                             hipStream_t st);

namespace {

// OIHW fp32 -> [tap'][ci][co] with tap' = (KH-1-ky)*KW + (KW-1-kx): the weight of the conv
// that maps dY (Cout channels) to dX (Cin channels)
template <typename T>
__global__ void pack_weights_dgrad_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW,
                                          T* __restrict__ out) {
  const size_t n = (size_t)Cout * Cin * KH * KW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int co = e % Cout;
    const size_t t = e / Cout;
    const int ci = t % Cin;
    const int tap = t / Cin;
    const int ky = KH - 1 - tap / KW, kx = KW - 1 - tap % KW;
    const float v = w[(((size_t)co * Cin + ci) * KH + ky) * KW + kx];
    if (sizeof(T) == 2) reinterpret_cast<unsigned short*>(out)[e] = lss_f2bf(v);
    else reinterpret_cast<float*>(out)[e] = v;
  }
}

// Data gradient of a stride-2 K x K conv (pad p) as a stride-1 conv over dY that produces the input's four PHASE
// PLANES at once: dX[2j + py] = sum_d dY[j - d] * w[ky = 2d + py + p].  d runs over [dmin, dmax] = [ceil((-1 - p) / 2),
// floor((K - 1 - p) / 2)] (3x3 / p1: {-1, 0}; 7x7 / p3: {-2 .. 1}; 1x1 / p0: {0}), tap t of the KT = dmax - dmin + 1 taps
// per dimension reads dY[j - dmax + t], i.e. d = dmax - t.  Packed as [tap][Cout' = (py, px, ci)][Cin' = co], zeros
// where ky or kx falls outside the kernel.
__global__ void pack_weights_s2_dgrad_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int pad, int KT,
                                             int dmax, unsigned short* __restrict__ out) {
  const size_t n = (size_t)KT * KT * 4 * Cin * Cout;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int co = e % Cout;
    size_t r = e / Cout;
    const int ci = r % Cin; r /= Cin;
    const int px = r & 1, py = (r >> 1) & 1; r >>= 2;
    const int tx = r % KT, ty = r / KT;
    const int ky = 2 * (dmax - ty) + py + pad, kx = 2 * (dmax - tx) + px + pad;
    const bool ok = ky >= 0 && ky < K && kx >= 0 && kx < K;
    out[e] = ok ? lss_f2bf(w[(((size_t)co * Cin + ci) * K + ky) * K + kx]) : (unsigned short)0;
  }
}

constexpr int NCLR = 16;  // extra workgroup rows of the transposer that clear guards / K tail

struct WgradGeom {
  int B, H, W, Wp;
  long long ktot;   // B*(H+2)*Wp rounded up to nsplit*32
  long long ld;     // row stride of the channel-major copies: Wp lead + ktot + Wp tail
  int nsplit;
  long long split_k;
};

inline WgradGeom wgrad_geom(int B, int H, int W, int Cin, int Cout) {
  WgradGeom g;
  g.B = B; g.H = H; g.W = W;
  g.Wp = (W + 2 + 7) / 8 * 8;
  const long long k = (long long)B * (H + 2) * g.Wp;
  // enough (tile, tap, split) workgroups for ~2.5 per CU
  const long long tiles = (long long)lss_cdiv(Cout, 128) * lss_cdiv(Cin, 128) * 9;
  int ns = (int)((640 + tiles - 1) / tiles);
  const long long max_ns = k / (32 * 8) > 0 ? k / (32 * 8) : 1;  // at least 8 K-steps per split
  if (ns > max_ns) ns = (int)max_ns;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.split_k = ((k + ns - 1) / ns + 31) / 32 * 32;
  g.ktot = g.split_k * ns;
  g.ld = g.ktot + 2LL * g.Wp;
  return g;
}

// One workgroup = one padded image row (b, y') x 64 channels.  NHWC row -> LDS tile ->
// channel-major rows, `ncopy` column-shifted copies (1: d = 0; 3: d = -1, 0, +1).
__global__ __launch_bounds__(256) void to_channel_major_kernel(const unsigned short* __restrict__ src,
                                                               int B, int H, int W, int C, int Wp,
                                                               long long ld, long long copy_stride,
                                                               int ncopy, unsigned short* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned short tile[];  // [64][Wp + 8]
  const int TS = Wp + 8;  // tile column j holds padded-image column x'' = j - 4
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * 64;
  const int nrows = B * (H + 2);
  if ((int)blockIdx.x >= nrows) {
    // extra workgroups: clear the lead guard row (slice 0) and everything past the image grid
    // (K tail + tail guard row; slices 1..NCLR-1) of this channel block's rows - no memset launches
    const int slice = blockIdx.x - nrows;
    const long long kimg = (long long)nrows * Wp;
    long long lo, hi;
    if (slice == 0) { lo = 0; hi = Wp; }
    else {
      const long long span8 = (ld - Wp - kimg) / 8, per = (span8 + NCLR - 2) / (NCLR - 1);
      lo = Wp + kimg + 8 * min(span8, per * (slice - 1));
      hi = Wp + kimg + 8 * min(span8, per * slice);
    }
    const long long n8 = (hi - lo) / 8;
    for (long long e = tid; e < (long long)ncopy * 64 * n8; e += 256) {
      const long long ch = e % n8;
      const int c = (int)((e / n8) % 64);
      const int copy = (int)(e / (n8 * 64));
      if (c0 + c < C)
        *reinterpret_cast<uint4*>(dst + (size_t)copy * copy_stride + (size_t)(c0 + c) * ld + lo + ch * 8) =
            make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const int yp = blockIdx.x % (H + 2), b = blockIdx.x / (H + 2);
  for (int e = tid; e < 64 * TS; e += 256) tile[e] = 0;
  __syncthreads();
  if (yp >= 1 && yp <= H) {
    const unsigned short* row = src + ((size_t)(b * H + (yp - 1)) * W) * C;
    for (int e = tid; e < W * 8; e += 256) {
      const int px = e >> 3, part = e & 7;
      const int c = c0 + part * 8;
      if (c < C) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + (size_t)px * C + c);
        const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          tile[(part * 8 + 2 * k) * TS + px + 1 + 4] = (unsigned short)(u[k] & 0xffff);
          tile[(part * 8 + 2 * k + 1) * TS + px + 1 + 4] = (unsigned short)(u[k] >> 16);
        }
      }
    }
  }
  __syncthreads();
  const long long flat0 = ((long long)b * (H + 2) + yp) * Wp + Wp;  // + Wp: the lead guard row
  const int chunks = Wp / 8;
  for (int e = tid; e < ncopy * 64 * chunks; e += 256) {
    const int ch = e % chunks;
    const int c = (e / chunks) % 64;
    const int copy = e / (chunks * 64);
    if (c0 + c >= C) continue;
    const int d = ncopy == 3 ? copy - 1 : 0;
    const unsigned short* t = tile + c * TS + ch * 8 + d + 4;
    uint4 o;
    o.x = t[0] | ((unsigned int)t[1] << 16);
    o.y = t[2] | ((unsigned int)t[3] << 16);
    o.z = t[4] | ((unsigned int)t[5] << 16);
    o.w = t[6] | ((unsigned int)t[7] << 16);
    *reinterpret_cast<uint4*>(dst + (size_t)copy * copy_stride + (size_t)(c0 + c) * ld + flat0 + ch * 8) = o;
  }
}

// dW ([Cout][Cin][ntap] fp32: OIHW for the 3x3 conv) = sum over splits, in split order
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int Cout, int Cin,
                                    float* __restrict__ dw, int ntap) {
  const size_t n = (size_t)ntap * Cout * Cin;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int ci = e % Cin;
    const size_t t = e / Cin;
    const int co = t % Cout;
    const int tap = t / Cout;
    // sixteen interleaved partial sums (fixed assignment split k -> accumulator k & 15, fixed final tree): the loads of
    // a thread are independent, so they pipeline instead of serialising.  (Four accumulators until round 4: layer1's
    // 36 864 elements x 136 splits are 144 workgroups at one wave per SIMD - 34 dependent round trips, 11.6 us for 20 MB.)
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
    int k = 0;
    for (; k + 16 <= nsplit; k += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] += partial[(size_t)(k + i) * n + e];
    }
#pragma unroll
    for (int i = 0; i < 15; ++i)
      if (k + i < nsplit) s[i] += partial[(size_t)(k + i) * n + e];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) s[i] += s[i + w];
    dw[((size_t)co * Cin + ci) * ntap + tap] = s[0];
  }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

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

// out[b, Y, X, :] = [ x2[b, Y, X, :C2] | bilinear_align_corners(x)[b, Y, X, :Cx] ]   (ref Up.forward,
// src/modules.py:22-24, materialised: only the weight-gradient GEMM needs it as a tensor)
__global__ __launch_bounds__(256) void upsample_cat_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ x2, int B, int H,
                                                           int W, int Cx, int C2, int up, float ry, float rx,
                                                           unsigned short* __restrict__ out) {
  const int Hh = H * up, Wh = W * up, Ct = C2 + Cx, P8 = Ct / 8;
  const long long n = (long long)B * Hh * Wh * P8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int p8 = (int)(e % P8);
    const long long pix = e / P8;
    const int X = (int)(pix % Wh), Y = (int)((pix / Wh) % Hh), b = (int)(pix / ((long long)Wh * Hh));
    const int c = p8 * 8;
    uint4 o;
    if (c < C2) {
      o = *reinterpret_cast<const uint4*>(x2 + (size_t)pix * C2 + c);
    } else {
      const float sy = ry * (float)Y, sx = rx * (float)X;
      const int y0 = (int)sy, x0 = (int)sx;
      const float ly = sy - (float)y0, lx = sx - (float)x0;
      const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
      const unsigned short* base = x + (size_t)b * H * W * Cx + (c - C2);
      float a[8], bq[8], cq[8], d[8], r[8];
      unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x0) * Cx), a);
      unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * W + x1) * Cx), bq);
      unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x0) * Cx), cq);
      unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * W + x1) * Cx), d);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        r[k] = (1.f - ly) * ((1.f - lx) * a[k] + lx * bq[k]) + ly * ((1.f - lx) * cq[k] + lx * d[k]);
      o = pack8(r);
    }
    *reinterpret_cast<uint4*>(out + (size_t)pix * Ct + c) = o;
  }
}

// Adjoint of the bilinear (align_corners) upsample, gather form: low-res pixel (y, x) collects
// from every high-res pixel whose 2x2 source window contains it.  g = channels [c_off, c_off+Cx)
// of a (B, H*up, W*up, Ct) tensor; dx (B, H, W, Cx).  fp32 accumulation, fixed order.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const unsigned short* __restrict__ g, int B, int H,
                                                           int W, int Cx, int Ct, int c_off, int up, float ry,
                                                           float rx, unsigned short* __restrict__ dx) {
  const int Hh = H * up, Wh = W * up, P8 = Cx / 8;
  const long long n = (long long)B * H * W * P8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int p8 = (int)(e % P8);
    const long long pix = e / P8;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    // candidate high-res rows / columns: source coordinate within (y-1, y+1)
    const int Ylo = max(0, (int)floorf((float)(y - 1) / ry)), Yhi = min(Hh - 1, (int)ceilf((float)(y + 1) / ry));
    const int Xlo = max(0, (int)floorf((float)(x - 1) / rx)), Xhi = min(Wh - 1, (int)ceilf((float)(x + 1) / rx));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      const float sy = ry * (float)Y;
      const int y0 = (int)sy;
      const float ly = sy - (float)y0;
      const float wy = y0 == y ? 1.f - ly : ((y0 + 1 == y && y0 < H - 1) ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int X = Xlo; X <= Xhi; ++X) {
        const float sx = rx * (float)X;
        const int x0 = (int)sx;
        const float lx = sx - (float)x0;
        const float wx = x0 == x ? 1.f - lx : ((x0 + 1 == x && x0 < W - 1) ? lx : 0.f);
        if (wx == 0.f) continue;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(g + (((size_t)b * Hh + Y) * Wh + X) * Ct + c_off + p8 * 8), v);
        const float wgt = wy * wx;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, v[k], acc[k]);
      }
    }
    *reinterpret_cast<uint4*>(dx + (size_t)pix * Cx + p8 * 8) = pack8(acc);
  }
}

// The same sums in the same order with the loads of one high-res ROW of the window requested together: the form above
// walks its (2 up + 1)^2 candidates one dependent 16-B load at a time behind two data-dependent `continue`s - 38.7 us
// per launch at one wave per SIMD for the x4 layer (round 4; the K7 pattern).  NW >= columns of any window (host-checked).
template <int NW>
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(const unsigned short* __restrict__ g, int B, int H,
                                                                int W, int Cx, int Ct, int c_off, int up, float ry,
                                                                float rx, unsigned short* __restrict__ dx) {
  const int Hh = H * up, Wh = W * up, P8 = Cx / 8;
  const long long n = (long long)B * H * W * P8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int p8 = (int)(e % P8);
    const long long pix = e / P8;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    const int Ylo = max(0, (int)floorf((float)(y - 1) / ry)), Yhi = min(Hh - 1, (int)ceilf((float)(y + 1) / ry));
    const int Xlo = max(0, (int)floorf((float)(x - 1) / rx)), Xhi = min(Wh - 1, (int)ceilf((float)(x + 1) / rx));
    float wxv[NW];
    int xoff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int X = Xlo + i;
      const float sx = rx * (float)X;
      const int x0 = (int)sx;
      const float lx = sx - (float)x0;
      const float wx = x0 == x ? 1.f - lx : ((x0 + 1 == x && x0 < W - 1) ? lx : 0.f);
      wxv[i] = X <= Xhi ? wx : 0.f;
      xoff[i] = min(X, Wh - 1) * Ct;
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      const float sy = ry * (float)Y;
      const int y0 = (int)sy;
      const float ly = sy - (float)y0;
      const float wy = y0 == y ? 1.f - ly : ((y0 + 1 == y && y0 < H - 1) ? ly : 0.f);
      if (wy == 0.f) continue;
      const unsigned short* row = g + (((size_t)b * Hh + Y) * Wh) * Ct + c_off + p8 * 8;
      uint4 r[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        r[i] = make_uint4(0, 0, 0, 0);
        if (wxv[i] != 0.f) r[i] = *reinterpret_cast<const uint4*>(row + xoff[i]);  // (predicated, still one batch)
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        if (wxv[i] == 0.f) continue;
        float v[8];
        unpack8(r[i], v);
        const float wgt = wy * wxv[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(wgt, v[k], acc[k]);
      }
    }
    *reinterpret_cast<uint4*>(dx + (size_t)pix * Cx + p8 * 8) = pack8(acc);
  }
}

}  // namespace

extern "C" int lss_conv2d_pack_weights_dgrad(const float* w_oihw, int Cout, int Cin, int KH, int KW, int dt,
                                             void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  LSS_CHECK_POS(Cout); LSS_CHECK_POS(Cin); LSS_CHECK_POS(KH); LSS_CHECK_POS(KW);
  const size_t n = (size_t)Cout * Cin * KH * KW;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(pack_weights_dgrad_kernel<unsigned short>, dim3(grid), dim3(256), 0, lss_stream(stream),
                       w_oihw, Cout, Cin, KH, KW, reinterpret_cast<unsigned short*>(w_packed));
  else if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(pack_weights_dgrad_kernel<float>, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw,
                       Cout, Cin, KH, KW, reinterpret_cast<float*>(w_packed));
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}

// taps per dimension of the phase-plane data-gradient conv of a stride-2 K x K / pad conv (0: not a case)
extern "C" int lss_conv2d_s2_dgrad_taps(int K, int pad) {
  if (!((K == 1 && pad == 0) || (K == 3 && pad == 1) || (K == 7 && pad == 3))) return 0;
  return K == 1 ? 1 : (K == 3 ? 2 : 4);
}

// w_oihw [Cout][Cin][K][K] fp32 -> bf16 [KT*KT][4*Cin][Cout]: use with lss_conv2d_fwd(dy, ..., Cx = Cout,
// Cout = 4 * Cin, KH = KW = KT, stride 1, pad KT / 2); rows / columns [s, s + H/2) of its output (s = 1 for KT = 2, 4;
// 0 for KT = 1) are the phase planes dXs[b, j, i, (py, px, ci)] = dX[b, 2j + py, 2i + px, ci].
extern "C" int lss_conv2d_pack_weights_s2_dgrad(const float* w_oihw, int Cout, int Cin, int K, int pad,
                                                void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  LSS_CHECK_POS(Cout); LSS_CHECK_POS(Cin);
  const int KT = lss_conv2d_s2_dgrad_taps(K, pad);
  if (KT == 0) return LSS_E_SHAPE;
  const int dmax = (K - 1 - pad) / 2;
  const size_t n = (size_t)KT * KT * 4 * Cin * Cout;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_s2_dgrad_kernel, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cout, Cin, K,
                     pad, KT, dmax, reinterpret_cast<unsigned short*>(w_packed));
  return lss_launch_status();
}

extern "C" size_t lss_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int nd = lss_wgrad_direct_splits(B, H, W, Cin, Cout);
  if (nd > 0) return align256((size_t)nd * 9 * Cout * Cin * 4);  // the direct kernel: fp32 partial tiles only
  const WgradGeom g = wgrad_geom(B, H, W, Cin, Cout);
  return align256((size_t)3 * Cin * g.ld * 2) + align256((size_t)Cout * g.ld * 2) +
         align256((size_t)g.nsplit * 9 * Cout * Cin * 4);
}

extern "C" int lss_conv2d_wgrad(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout,
                                void* workspace, size_t workspace_bytes, float* dw_oihw, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(dy); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(dw_oihw);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cin); LSS_CHECK_POS(Cout);
  if (Cin % 8 != 0 || Cout % 8 != 0 || W > 4096) return LSS_E_SHAPE;
  if (workspace_bytes < lss_conv2d_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return LSS_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return LSS_E_ALIGN;
  hipStream_t st = lss_stream(stream);
  const int nd = lss_wgrad_direct_splits(B, H, W, Cin, Cout);
  if (nd > 0) {  // K9w (conv_wgrad.hip), then the same fixed-order reduction over its splits
    float* partial = static_cast<float*>(workspace);
    int rc = lss_wgrad_direct_launch(x, dy, B, H, W, Cin, Cout, partial, st);
    if (rc != 0) return rc;
    const size_t n = (size_t)9 * Cout * Cin;
    const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, partial, nd, Cout, Cin, dw_oihw, 9);
    return lss_launch_status();
  }
  const WgradGeom g = wgrad_geom(B, H, W, Cin, Cout);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  unsigned short* xt = reinterpret_cast<unsigned short*>(ws);
  const size_t xt_bytes = align256((size_t)3 * Cin * g.ld * 2);
  unsigned short* dyt = reinterpret_cast<unsigned short*>(ws + xt_bytes);
  const size_t dyt_bytes = align256((size_t)Cout * g.ld * 2);
  float* partial = reinterpret_cast<float*>(ws + xt_bytes + dyt_bytes);
  const size_t lds = (size_t)64 * (g.Wp + 8) * 2;
  const dim3 gx(B * (H + 2) + NCLR, lss_cdiv(Cin, 64)), gy(B * (H + 2) + NCLR, lss_cdiv(Cout, 64));
  hipLaunchKernelGGL(to_channel_major_kernel, gx, dim3(256), lds, st, static_cast<const unsigned short*>(x), B, H,
                     W, Cin, g.Wp, g.ld, (long long)Cin * g.ld, 3, xt);
  hipLaunchKernelGGL(to_channel_major_kernel, gy, dim3(256), lds, st, static_cast<const unsigned short*>(dy), B,
                     H, W, Cout, g.Wp, g.ld, 0LL, 1, dyt);
  // operand bases skip the lead guard row; tap (ky,kx) then reaches back/forward by one padded row
  const void* xts[3] = {xt + g.Wp, xt + (size_t)Cin * g.ld + g.Wp, xt + (size_t)2 * Cin * g.ld + g.Wp};
  int rc = lss_wgrad_gemm_launch(dyt + g.Wp, xts, partial, Cout, Cin, g.ld, g.split_k, g.nsplit, 9, g.Wp, st);
  if (rc != 0) return rc;
  const size_t n = (size_t)9 * Cout * Cin;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, partial, g.nsplit, Cout, Cin, dw_oihw, 9);
  return lss_launch_status();
}

// Weight gradient of the 4 x 4-tap stride-1 conv with taps (dy, dx) in {-2 .. 1}^2 - what a 7x7 / stride-2 / pad-3
// conv (the BevEncode stem, ref src/modules.py:99) is over the phase planes xs[b, y, x, (py, px, c)] =
// x[b, 2y + py, 2x + px, c]: dw16[co][ci'][(dy + 2) * 4 + dx + 2] = sum dY[b,y,x,co] * xs[b, y + dy, x + dx, ci'].
extern "C" size_t lss_conv2d_wgrad4x4_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const int nd = lss_wgrad_direct_splits(B, H, W, Cin, Cout);
  return nd > 0 ? align256((size_t)nd * 16 * Cout * Cin * 4) : 0;
}

extern "C" int lss_conv2d_wgrad4x4(const void* xs, const void* dy, int B, int H, int W, int Cin, int Cout,
                                   void* workspace, size_t workspace_bytes, float* dw16, void* stream) {
  LSS_CHECK_PTR(xs); LSS_CHECK_PTR(dy); LSS_CHECK_PTR(workspace); LSS_CHECK_PTR(dw16);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cin); LSS_CHECK_POS(Cout);
  const int nd = lss_wgrad_direct_splits(B, H, W, Cin, Cout);
  if (nd <= 0) return LSS_E_SHAPE;
  if (workspace_bytes < lss_conv2d_wgrad4x4_workspace_bytes(B, H, W, Cin, Cout)) return LSS_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return LSS_E_ALIGN;
  hipStream_t st = lss_stream(stream);
  float* partial = static_cast<float*>(workspace);
  int rc = lss_wgrad_taps4x4_launch(xs, dy, B, H, W, Cin, Cout, partial, st);
  if (rc != 0) return rc;
  const size_t n = (size_t)16 * Cout * Cin;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, partial, nd, Cout, Cin, dw16, 16);
  return lss_launch_status();
}

extern "C" int lss_upsample_cat_nhwc(const void* x, const void* x2, int B, int H, int W, int Cx, int C2, int up,
                                     void* out, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(out);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cx); LSS_CHECK_POS(up);
  if (C2 < 0 || Cx % 8 != 0 || C2 % 8 != 0) return LSS_E_SHAPE;
  if (C2 > 0 && x2 == nullptr) return LSS_E_NULL;
  const int Hh = H * up, Wh = W * up;
  const float ry = Hh > 1 ? (float)(H - 1) / (float)(Hh - 1) : 0.f, rx = Wh > 1 ? (float)(W - 1) / (float)(Wh - 1) : 0.f;
  const long long n = (long long)B * Hh * Wh * ((C2 + Cx) / 8);
  const int grid = (int)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(upsample_cat_kernel, dim3(grid), dim3(256), 0, lss_stream(stream),
                     static_cast<const unsigned short*>(x), static_cast<const unsigned short*>(x2), B, H, W, Cx, C2,
                     up, ry, rx, static_cast<unsigned short*>(out));
  return lss_launch_status();
}

extern "C" int lss_upsample_bwd_nhwc(const void* g, int B, int H, int W, int Cx, int Ct, int c_off, int up,
                                     void* dx, void* stream) {
  LSS_CHECK_PTR(g); LSS_CHECK_PTR(dx);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cx); LSS_CHECK_POS(up);
  if (c_off < 0 || Cx % 8 != 0 || c_off % 8 != 0 || c_off + Cx > Ct || Ct % 8 != 0) return LSS_E_SHAPE;
  const int Hh = H * up, Wh = W * up;
  const float ry = Hh > 1 ? (float)(H - 1) / (float)(Hh - 1) : 0.f, rx = Wh > 1 ? (float)(W - 1) / (float)(Wh - 1) : 0.f;
  if (H < 2 || W < 2) return LSS_E_SHAPE;  // ry, rx > 0 below
  const long long n = (long long)B * H * W * (Cx / 8);
  const int grid = (int)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  // columns a window can span: candidates floor((x - 1) / rx) .. ceil((x + 1) / rx)
  const int span = (int)ceilf(2.f / rx) + 3;
  const bool rows_form = getenv("LSS_UPSAMPLE_BWD_ROWS") == nullptr || atoi(getenv("LSS_UPSAMPLE_BWD_ROWS")) != 0;
#define LSS_UB(KERNEL)                                                                                      \
  hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, lss_stream(stream), static_cast<const unsigned short*>(g), B, H, \
                     W, Cx, Ct, c_off, up, ry, rx, static_cast<unsigned short*>(dx))
  if (rows_form && span <= 8) LSS_UB(upsample_bwd_rows_kernel<8>);
  else if (rows_form && span <= 12) LSS_UB(upsample_bwd_rows_kernel<12>);
  else LSS_UB(upsample_bwd_kernel);
#undef LSS_UB
  return lss_launch_status();
}
