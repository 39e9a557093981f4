// K8s: the launch-bound 3x3 / stride-1 convolutions of BevEncode's resnet layers (layer1: 64 -> 64 at 100^2, layer2:
// 128 -> 128 at 50^2, layer3: 256 -> 256 at 25^2; ref src/modules.py:100-102 + torchvision BasicBlock) as an implicit
// GEMM whose K loop is SPLIT OVER THE WAVES of a workgroup, operands straight from global memory.
//
// Why another kernel.  These layers are small GEMMs (M = 2 500 .. 40 000 pixels, N = 64 .. 256, K = 576 .. 2 304) in
// a chain of 11 dependent launches; on the LDS-tiled kernel (conv_mfma.hip) a layer3 conv takes 15 us at batch 4 AND
// at batch 1 - its 14-112 workgroups each walk 18-36 (chunk, tap) steps of LDS-DMA -> barrier -> 8 MFMAs one after
// the other, so the launch lasts as long as ONE workgroup's serial chain.  Here the chain is cut instead of sped up:
//   * workgroup = 64 consecutive output pixels (of the flattened B x H x W index: no 2-D tile waste at 25 x 25) x 64
//     output channels; its KS = Cin / 32 (2 .. 8) waves each take ONE 32-channel chunk of the input (x 9 taps): 144
//     MFMAs (v_mfma_f32_16x16x32_bf16, A = weights, B = pixels) per wave, whatever the layer;
//   * no LDS staging and no barrier in the K loop: a wave loads its fragments with buffer loads (out-of-image taps
//     read zeros through the descriptor's range check), three taps in flight; the weight rows come from the packed
//     [tap][Cout][Cin] layout as they are;
//   * the KS partial tiles meet in LDS (KS x 17 KiB), every thread then finishes 8-channel pieces: sum over the
//     K-slices, scale / shift (folded BatchNorm), residual, ReLU, one 16-B bf16 store.
// Grid: layer3 at batch 4 = 40 x 4 = 160 workgroups of 512 threads, layer2 157 x 2 of 256, layer1 625 of 128.
#include <stdlib.h>

#include "lss_common.h"

namespace {

struct KsArgs {
  const unsigned short* x;         // (B, H, W, Cin) bf16 NHWC
  const unsigned short* w;         // [9][Cout][Cin] bf16 (lss_conv2d_pack_weights)
  const float* scale;
  const float* shift;
  const unsigned short* residual;  // (B, H, W, Cout) bf16 or null
  unsigned short* y;               // (B, H, W, Cout) bf16
  int B, H, W, Cin, Cout, M, relu, wt;
};

constexpr int KS_ROW = 64 * 4 + 16;  // bytes of one pixel row of a partial tile in LDS (64 fp32 + pad: conflict-free b128)

typedef __attribute__((ext_vector_type(4))) unsigned int ks_u32x4;
typedef __attribute__((ext_vector_type(2))) float ks_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ks_bf16x2;
__device__ __forceinline__ unsigned int ks_pack_bf2(float lo, float hi) {
  const ks_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ks_bf16x2));
}

template <int KS>
__global__ __launch_bounds__(64 * KS, 2) void conv_ksplit_kernel(KsArgs a) {  // <= 256 VGPRs: three taps of fragments in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char red[];  // [KS][64 pixels][KS_ROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int nchunk = a.Cin / (32 * KS);  // chunks of this wave (1 for the BevEncode layers)

  // pixels of this lane: p = m0 + 16 pt + n
  int xoff[4];        // element offset of the pixel's channel piece, (pixel * Cin + kq * 8)
  unsigned int tm[4]; // bit t: tap t of this pixel is inside the image (0 for pixels past M)
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int p = m0 + pt * 16 + n;
    const int pc = min(p, a.M - 1);
    const int hw = a.H * a.W;
    const int b = pc / hw, r = pc - b * hw, y = r / a.W, x = r - y * a.W;
    xoff[pt] = pc * a.Cin + kq * 8;
    unsigned int msk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (p < a.M && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) msk |= 1u << t;
    }
    tm[pt] = msk;
  }
  // weight rows of this lane.  Row m of channel tile t is channel 16 (m >> 2) + 4 t + (m & 3): lane (q = kq, n) then
  // ends with the 16 consecutive channels 16 q .. 16 q + 15 of pixel n in its accumulators (4 x 16-B LDS stores)
  int woff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) woff[t] = (n0 + (n >> 2) * 16 + 4 * t + (n & 3)) * a.Cin + kq * 8;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.x), 0, (int)((size_t)a.M * a.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.w), 0, (int)((size_t)9 * a.Cout * a.Cin * 2), 0x00020000);

  f32x4 acc[4][4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[pt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nchunk; ++c) {
    const int cb = (ks * nchunk + c) * 32;  // first input channel of the chunk
    ks_u32x4 fa[3][4], fb[3][4];
    auto load_tap = [&](int tap, int s) {
      const int ky = tap / 3, kx = tap % 3;
      const int xs = ((ky - 1) * a.W + (kx - 1)) * a.Cin + cb;  // pixel shift + chunk, elements
      const int ws = tap * a.Cout * a.Cin + cb;
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[s][t] = __builtin_amdgcn_raw_buffer_load_b128(wr, (woff[t] + ws) * 2, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        // outside the image: an offset past the descriptor's range returns zeros
        const int vo = ((tm[pt] >> tap) & 1u) ? (xoff[pt] + xs) * 2 : 0x7ffffff0;
        fb[s][pt] = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0);
      }
    };
    load_tap(0, 0);
    load_tap(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // the requests of tap + 2 go out BEFORE this tap's MFMAs and stay there (hipcc otherwise sinks every load next to
      // its use - one exposed round trip per fragment in the first build)
      if (tap + 2 < 9) load_tap(tap + 2, (tap + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
      const int s = tap % 3;
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[pt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[s][t]),
                                                               __builtin_bit_cast(bf16x8, fb[s][pt]), acc[pt][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // partial tile -> LDS: lane (q, n), pixel tile pt: channels 16 q .. 16 q + 15 of pixel 16 pt + n
  unsigned char* mine = red + (size_t)ks * (64 * KS_ROW);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<f32x4*>(mine + (pt * 16 + n) * KS_ROW + (kq * 16 + t * 4) * 4) = acc[pt][t];
  __syncthreads();

  // finish: 512 pieces of 8 channels; thread takes pieces tid, tid + 64 KS, ...
  const bool vec_res = a.residual != nullptr;
#pragma unroll
  for (int g = 0; g < 8 / KS; ++g) {
    const int piece = tid + g * (64 * KS);
    const int px = piece >> 3, c8 = piece & 7;
    const int p = m0 + px;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {  // fixed order: bit-reproducible
      const unsigned char* src = red + (size_t)s * (64 * KS_ROW) + px * KS_ROW + c8 * 32;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += lo[k]; v[4 + k] += hi[k]; }
    }
    if (p >= a.M) continue;
    const int co = n0 + c8 * 8;
    if (a.scale) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.scale + co), s1 = *reinterpret_cast<const f32x4*>(a.scale + co + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] *= s0[k]; v[4 + k] *= s1[k]; }
    }
    if (a.shift) {
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(a.shift + co), h1 = *reinterpret_cast<const f32x4*>(a.shift + co + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += h0[k]; v[4 + k] += h1[k]; }
    }
    const size_t o = (size_t)p * a.Cout + co;
    if (vec_res) {
      const uint4 rv = *reinterpret_cast<const uint4*>(a.residual + o);
      const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += lss_bf2f((unsigned short)(ru[k] & 0xffff));
        v[2 * k + 1] += lss_bf2f((unsigned short)(ru[k] >> 16));
      }
    }
    if (a.relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    ks_u32x4 ov;
    ov[0] = ks_pack_bf2(v[0], v[1]); ov[1] = ks_pack_bf2(v[2], v[3]);
    ov[2] = ks_pack_bf2(v[4], v[5]); ov[3] = ks_pack_bf2(v[6], v[7]);
    if (a.wt) {
      const __amdgpu_buffer_rsrc_t yr =
          __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((size_t)a.M * a.Cout * 2), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(ov, yr, (int)(o * 2), 0, 16);  // write-through: nothing dirty at the boundary
    } else {
      *reinterpret_cast<ks_u32x4*>(a.y + o) = ov;
    }
  }
}

template <int KS>
int launch_ks(const KsArgs& a, hipStream_t st) {
  const size_t lds = (size_t)KS * 64 * KS_ROW;
  if (lds > 64 * 1024) {
    static bool big[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !big[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ksplit_kernel<KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return LSS_E_SHAPE;
      big[dev] = true;
    }
  }
  hipLaunchKernelGGL((conv_ksplit_kernel<KS>), dim3(lss_cdiv(a.M, 64), a.Cout / 64), dim3(64 * KS), lds, st, a);
  return lss_launch_status();
}

}  // namespace

// Is this 3x3 / stride-1 / pad-1 bf16 conv (plain input, no statistics) a case for the K-split kernel?  Small
// problems only: at most 1024 workgroup tiles (beyond that the LDS-tiled / ring kernels are throughput-bound and
// better), Cin = 64 .. 512 in 32-channel chunks over 2 .. 8 waves, Cout a multiple of 64.
int lss_conv_ksplit_ok(int B, int H, int W, int Cin, int Cout) {
  if (const char* e = getenv("LSS_CONV_KSPLIT_DIRECT"))
    if (atoi(e) == 0) return 0;
  if (Cin % 64 != 0 || Cin < 64 || Cin > 512 || Cout % 64 != 0) return 0;
  const long long M = (long long)B * H * W;
  if (M * Cin >= (1LL << 30) || M * Cout >= (1LL << 30)) return 0;
  return (M + 63) / 64 * (Cout / 64) <= 1024 ? 1 : 0;
}

int lss_conv_ksplit_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                           void* y, int B, int H, int W, int Cin, int Cout, int relu, int wt, hipStream_t st) {
  KsArgs a;
  a.x = reinterpret_cast<const unsigned short*>(x);
  a.w = reinterpret_cast<const unsigned short*>(w);
  a.scale = scale; a.shift = shift;
  a.residual = reinterpret_cast<const unsigned short*>(residual);
  a.y = reinterpret_cast<unsigned short*>(y);
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.M = B * H * W; a.relu = relu; a.wt = wt;
  const int chunks = Cin / 32;
  if (chunks % 8 == 0) return launch_ks<8>(a, st);
  if (chunks % 4 == 0) return launch_ks<4>(a, st);
  return launch_ks<2>(a, st);
}
