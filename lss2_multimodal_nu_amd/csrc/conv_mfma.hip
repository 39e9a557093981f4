// K8: BevEncode convolutions as implicit GEMM on MFMA, NHWC activations.
//
//   out[m, co] = act( scale[co] * sum_{tap, ci} in[pix(m) + tap, ci] * w[tap, co, ci]
//                     + shift[co] + residual[m, co] )
// m = (b, oy, ox) output pixel.  `in` is x, or - fused into the operand gather -
// cat([x2, bilinear_upsample_align_corners(x, up)], channel) of the reference's
// `Up.forward` (src/modules.py:22-26) and of `up2[0]` (src/modules.py:110-111).
//
// Kernel `conv_direct_kernel` (all shapes): 256 threads = 4 waves; the workgroup
// computes 128 output pixels x 64 output channels, wave w the pixel rows
// [32w, 32w+32) against both 32-wide channel tiles:
//   bf16: v_mfma_f32_32x32x16_bf16, A[row r][k = 8h + j]  B[k = 8h + j][col c]
//   f32 : v_mfma_f32_32x32x2_f32,   A[row r][k = h]       B[k = h][col c]
// (r = c = lane & 31, h = lane >> 5).  Operand fragments are loaded straight
// from global memory in 16-B pieces: within a K block every lane half h owns a
// contiguous run of channels (32 bf16 / 4 f32) and k-step s consumes its s-th
// piece on BOTH operands, so activations (NHWC) and weights ([tap][co][ci]) keep
// their natural layouts - only the order of the K summation is permuted.
#include <stdlib.h>

#include "lss_common.h"

int lss_linear_bf16_launch(const void* x, const void* w, const float* scale, const float* shift,
                           const void* residual, void* y, long long M, int N, int K, int act,
                           int out_f32, int group_hw, hipStream_t st);  // linear_mfma.hip

int lss_conv_ring_launch(const void* x, const void* x2, const void* w_ring, const float* scale, const float* shift,
                         void* y, const float* head_w, const float* head_b, float* head_out, int head_n, int B, int H,
                         int W, int Cx, int C2, int up, int Cout, int relu, int wt, hipStream_t st);  // conv_ring.hip

int lss_conv_ks_launch(const void* x, const void* w_ks, const float* scale, const float* shift, const void* residual,
                       void* y, int B, int H, int W, int Cin, int Cout, int relu, int wt, hipStream_t st);  // conv_ks.hip

namespace {

struct ConvArgs {
  const void* x;
  const void* x2;
  const void* w;
  const float* scale;
  const float* shift;
  const void* residual;
  void* y;
  float* stats;
  int B, H, W;  // spatial size of x (low-res source when up > 1)
  int Cx, C2, up;
  int Hin, Win;  // H*up, W*up: the conv's input plane
  int Cin;       // C2 + Cx
  int Cout, KH, KW, stride, pad;
  int Ho, Wo, M;
  int relu;     // activation: 0 none, 1 ReLU, 2 GELU (erf form)
  int out_f32;  // bf16 kernels: write y as fp32
  int wt;       // epilogue stores write through (sc1): nothing left dirty in L2 at the kernel boundary
  // two convs over the same input in one launch (a BasicBlock's stride-2 conv1 and its 1x1
  // downsample): output channels [0, split) go to y, [split, Cout) to y2, each dense; the
  // activation applies to channels below relu_n only.  split, relu_n are multiples of the 128-wide tile.
  void* y2;
  int split, relu_n;
  float ry, rx;  // (H-1)/(Hin-1), (W-1)/(Win-1) for the align_corners upsample
  // optional fused 1x1 head (ref: src/modules.py:115 up2[4]): out[b,k,oy,ox] =
  // head_b[k] + sum_co act(...)[co] * head_w[k, co]; NCHW fp32; needs Cout == BN
  const float* head_w;
  const float* head_b;
  float* head_out;
  int head_n;
  // diagnostics (tools/conv_phase_stamps.py): 8 x 100-MHz s_memrealtime stamps per workgroup, or null
  unsigned long long* stamps;
  int src_lds;  // MODE 1, KC = 32: the low-res source patch of a chunk is staged in LDS (see SRC below)
};

// LSS_CONV_STAMPS=<hex device address of a u64 buffer, 8 entries per workgroup> switches the phase stamps on
static unsigned long long* conv_stamps_from_env() {
  const char* e = getenv("LSS_CONV_STAMPS");
  return e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16)) : nullptr;
}

__device__ __forceinline__ float conv_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}

template <typename T>
struct Frag;
template <>
struct Frag<unsigned short> {  // bf16: 8 channels per 16-B piece
  typedef bf16x8 type;
  static constexpr int PIECE = 8;    // channels per 16-B piece
  static constexpr int KBLOCK = 64;  // channels per K block (2 halves x 4 pieces)
};
template <>
struct Frag<float> {
  typedef f32x4 type;
  static constexpr int PIECE = 4;
  static constexpr int KBLOCK = 8;  // 2 halves x 1 piece (4 k-steps of 2)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// vmcnt(0), which would expose the latency of the weight / patch prefetch loads
// that are meant to stay in flight across it (cdna_hip_programming.md section 5,
// "Pipelining across barriers").
// x of the lane a DPP control selects (quad_perm / row_mirror / row_half_mirror: all inside a row of 16)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 lerp4(f32x4 a, f32x4 b, float t) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = a[i] + t * (b[i] - a[i]);
  return r;
}

// bilinear blend of four 8-channel bf16 pieces in the four-weight form, on (lo, hi)
// channel pairs: float2 arithmetic maps to v_pk_mul_f32 / v_pk_fma_f32.  One rounding to bf16.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 unpack_bf2(unsigned int u) {
  f32x2 v;
  v[0] = __builtin_bit_cast(float, u << 16);
  v[1] = __builtin_bit_cast(float, u & 0xffff0000u);
  return v;
}
__device__ __forceinline__ uint4 blend_bf16x8(const uint4& q00, const uint4& q01, const uint4& q10,
                                              const uint4& q11, float lx, float ly) {
  const float w11 = lx * ly, w10 = ly - w11, w01 = lx - w11, w00 = 1.f - lx - ly + w11;
  const unsigned int* u00 = reinterpret_cast<const unsigned int*>(&q00);
  const unsigned int* u01 = reinterpret_cast<const unsigned int*>(&q01);
  const unsigned int* u10 = reinterpret_cast<const unsigned int*>(&q10);
  const unsigned int* u11 = reinterpret_cast<const unsigned int*>(&q11);
  uint4 out;
  unsigned int* o = reinterpret_cast<unsigned int*>(&out);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 r = unpack_bf2(u00[i]) * w00;
    r = __builtin_elementwise_fma(unpack_bf2(u01[i]), (f32x2){w01, w01}, r);
    r = __builtin_elementwise_fma(unpack_bf2(u10[i]), (f32x2){w10, w10}, r);
    r = __builtin_elementwise_fma(unpack_bf2(u11[i]), (f32x2){w11, w11}, r);
    o[i] = lss_pack_bf2(r[0], r[1]);
  }
  return out;
}

// 16-B piece (bf16: 8 ch, f32: 4 ch) of pixel (b, iy, ix) of the conv's virtual
// input, channel offset c (multiple of the piece size, never straddling x2 | x).
template <typename T, bool FUSED>
__device__ __forceinline__ uint4 load_in_piece(const ConvArgs& a, int b, int iy, int ix, int c) {
  if (!FUSED) {
    const T* p = reinterpret_cast<const T*>(a.x) + (((size_t)b * a.H + iy) * a.W + ix) * a.Cx + c;
    return *reinterpret_cast<const uint4*>(p);
  }
  if (c < a.C2) {
    const T* p = reinterpret_cast<const T*>(a.x2) + (((size_t)b * a.Hin + iy) * a.Win + ix) * a.C2 + c;
    return *reinterpret_cast<const uint4*>(p);
  }
  c -= a.C2;
  // nn.Upsample(bilinear, align_corners=True): src = dst * (in-1)/(out-1)
  const float sy = a.ry * (float)iy, sx = a.rx * (float)ix;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const T* base = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.Cx + c;
  const uint4 q00 = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * a.W + x0) * a.Cx);
  const uint4 q01 = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * a.W + x1) * a.Cx);
  const uint4 q10 = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * a.W + x0) * a.Cx);
  const uint4 q11 = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * a.W + x1) * a.Cx);
  uint4 out;
  if (sizeof(T) == 4) {
    const f32x4 top = lerp4(__builtin_bit_cast(f32x4, q00), __builtin_bit_cast(f32x4, q01), lx);
    const f32x4 bot = lerp4(__builtin_bit_cast(f32x4, q10), __builtin_bit_cast(f32x4, q11), lx);
    out = __builtin_bit_cast(uint4, lerp4(top, bot, ly));
  } else {
    const unsigned int* u00 = reinterpret_cast<const unsigned int*>(&q00);
    const unsigned int* u01 = reinterpret_cast<const unsigned int*>(&q01);
    const unsigned int* u10 = reinterpret_cast<const unsigned int*>(&q10);
    const unsigned int* u11 = reinterpret_cast<const unsigned int*>(&q11);
    unsigned int* o = reinterpret_cast<unsigned int*>(&out);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float r[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int sh = hh * 16;
        const float v00 = lss_bf2f((unsigned short)(u00[i] >> sh)), v01 = lss_bf2f((unsigned short)(u01[i] >> sh));
        const float v10 = lss_bf2f((unsigned short)(u10[i] >> sh)), v11 = lss_bf2f((unsigned short)(u11[i] >> sh));
        const float top = v00 + lx * (v01 - v00), bot = v10 + lx * (v11 - v10);
        r[hh] = top + ly * (bot - top);
      }
      o[i] = lss_pack_bf2(r[0], r[1]);
    }
  }
  return out;
}

template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  constexpr int PIECE = Frag<T>::PIECE;
  constexpr int KBLOCK = Frag<T>::KBLOCK;
  constexpr int NPIECE = KBLOCK / 2 / PIECE;  // pieces per lane half per K block (4 / 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int n0 = blockIdx.y * 64;
  const int m = m0 + r;
  const bool m_ok = m < a.M;
  int b = 0, oy = 0, ox = 0;
  if (m_ok) {
    b = m / (a.Ho * a.Wo);
    const int rem = m - b * (a.Ho * a.Wo);
    oy = rem / a.Wo;
    ox = rem - oy * a.Wo;
  }
  const T* wbase = reinterpret_cast<const T*>(a.w);
  const int co0 = n0 + r, co1 = n0 + 32 + r;
  const bool c0_ok = co0 < a.Cout, c1_ok = co1 < a.Cout;

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  for (int ky = 0; ky < a.KH; ++ky) {
    const int iy = oy * a.stride - a.pad + ky;
    for (int kx = 0; kx < a.KW; ++kx) {
      const int ix = ox * a.stride - a.pad + kx;
      const bool in_ok = m_ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      const int tap = ky * a.KW + kx;
      const T* w0 = wbase + ((size_t)tap * a.Cout + co0) * a.Cin;
      const T* w1 = wbase + ((size_t)tap * a.Cout + co1) * a.Cin;
      for (int cb = 0; cb < a.Cin; cb += KBLOCK) {
        const int c = cb + h * (KBLOCK / 2);
        uint4 av[NPIECE], b0[NPIECE], b1[NPIECE];
#pragma unroll
        for (int s = 0; s < NPIECE; ++s) {
          av[s] = make_uint4(0, 0, 0, 0);
          b0[s] = make_uint4(0, 0, 0, 0);
          b1[s] = make_uint4(0, 0, 0, 0);
          if (in_ok) av[s] = load_in_piece<T, FUSED>(a, b, iy, ix, c + s * PIECE);
          if (c0_ok) b0[s] = *reinterpret_cast<const uint4*>(w0 + c + s * PIECE);
          if (c1_ok) b1[s] = *reinterpret_cast<const uint4*>(w1 + c + s * PIECE);
        }
        if (sizeof(T) == 2) {
#pragma unroll
          for (int s = 0; s < NPIECE; ++s) {
            const bf16x8 fa = __builtin_bit_cast(bf16x8, av[s]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, b0[s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, b1[s]), acc1, 0, 0, 0);
          }
        } else {
          const f32x4 fa = __builtin_bit_cast(f32x4, av[0]);
          const f32x4 f0 = __builtin_bit_cast(f32x4, b0[0]);
          const f32x4 f1 = __builtin_bit_cast(f32x4, b1[0]);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], f1[s], acc1, 0, 0, 0);
          }
        }
      }
    }
  }

  // epilogue: D[row = (i&3) + 8*(i>>2) + 4*h][col = r]
  T* y = reinterpret_cast<T*>(a.y);
  const T* res = reinterpret_cast<const T*>(a.residual);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int co = nt == 0 ? co0 : co1;
    const bool cok = nt == 0 ? c0_ok : c1_ok;
    const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const int mm = m0 + row;
      const float raw = nt == 0 ? acc0[i] : acc1[i];
      if (mm < a.M && cok) {
        s1 += raw;
        s2 += raw * raw;
        float v = raw * sc + sh;
        const size_t o = (size_t)mm * a.Cout + co;
        if (res) v += (sizeof(T) == 2) ? lss_bf2f(reinterpret_cast<const unsigned short*>(res)[o])
                                       : reinterpret_cast<const float*>(res)[o];
        v = conv_act(v, a.relu);
        if (sizeof(T) == 2 && !a.out_f32) reinterpret_cast<unsigned short*>(y)[o] = lss_f2bf(v);
        else reinterpret_cast<float*>(y)[o] = v;
      }
    }
    if (a.stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0 && cok) {
        atomicAdd(a.stats + co, s1);
        atomicAdd(a.stats + a.Cout + co, s2);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// conv_lds_kernel: the bf16 convolutions of BevEncode (3x3 / stride 1 carries 95 % of its FLOPs; the
// stride-2 3x3 / 7x7 run as stride-1 convs over the four parity phases, MODE 2).  Output-stationary,
// LDS-tiled:
//   * workgroup (256 threads; 2 per CU, 3 with KC = 32) = TH x 16 output pixels x BN output channels
//     of one image; a wave owns RT x 2 tiles of v_mfma_f32_32x32x16_bf16 (RT = 2: 4 pixel rows x 64
//     channels, 64 accumulator registers; RT = 1 for grids that would leave CUs idle);
//   * per KC-channel chunk (64, or 32 for the fused-gather layers) of the input the (TH+2) x 18 halo'd
//     patch is gathered ONCE into LDS (the fused bilinear-upsample / concat gather runs here, once per
//     element instead of once per tap; its low-res source window is itself staged by LDS-DMA, see
//     SRC) and re-used by all taps through shifted ds_read_b128 windows;
//   * per (chunk, tap) step the BN x KC weight slab streams global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR round trip) into a 3-slot ring, two steps ahead of the
//     MFMAs; the step barrier is a raw s_barrier behind a COUNTED vmcnt, so the younger slab stays in
//     flight across it;
//   * KSP = 2 (512 threads): two 4-wave groups split the input channels of one tile (grids of at most
//     256 workgroups); the epilogue stages the fp32 tile in LDS and stores 16-B pieces (write-through),
//     optionally reducing a fused 1x1 head instead.
// LDS images: patch positions are KC*2 + 16 B apart (the 16-B pad) and patch rows are padded to a
// multiple of 256 B, which makes the two pixel rows of a 32-row MFMA tile land on complementary bank
// sets (conflict-free ds_read_b128); weight rows are KC*2 B, XOR-swizzled in 16-B pieces on the DMA's
// SOURCE address (the DMA destination is lane-linear) and on the read address.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: plain NHWC input, stride 1.  MODE 1: fused [x2 | bilinear-upsampled x] input.
// MODE 2: stride-2 conv as a stride-1 conv over the 4 parity phases of the input
// (space-to-depth done by the gather: chunk -> (phase, 64-channel block); weights
// come pre-arranged as [tap'][co][phase*Cx + c] from lss_conv2d_pack_weights_s2d).
// KH x KW = taps of the stride-1 problem, PAD = patch rows/cols before the output pixel.
// RT = 32-pixel MFMA row tiles per wave (2: 4 image rows x 64 channels per wave, the
// throughput shape; 1: 2 image rows, half the work per workgroup - used when the RT = 2
// grid would leave CUs idle, where the per-workgroup critical path is what counts).
// HEAD: the epilogue reduces the fused 1x1 head instead of storing the activation (its own instantiation: the
// head and the store path each keep only their own operands live - the store path's residual prefetch and
// the head's operand fragments together overflowed the 168-register budget of the KC = 32 kernels).
// (Variants built, measured and removed in round 3 - numbers in DESIGN.md section 4: PFB = blend of the next chunk
// under the last taps' MFMAs, PSP = pixel-split pairs on one weight ring, 64-channel tall tiles, mixed full / half
// height launches, s_setprio around the MFMA groups, a 3-slot ring for KC = 32.  The big layers now run on the
// loader / consumer kernel of conv_ring.hip at the sizes it takes; this kernel keeps them at small batch.)
// LDS bytes of one workgroup of the tile body below (same formulas; the body static_asserts the match)
constexpr int LSS_CONV_RING32 = 4;  // ring slots of the KC = 32 kernels
template <int RT, int BN, int MODE, int KH, int KW, int KC, int KSP, int TPS = 1>
constexpr int conv_lds_smem_bytes() {
  constexpr int TH = (4 / (BN / 64)) * RT * 2, IW = 16 + KW - 1, IH = TH + KH - 1, POSB = KC * 2 + 16;
  constexpr int IROWB = (IW * POSB + 255) / 256 * 256, W_BYTES = TPS * BN * KC * 2, PPP = KC / 8;
  constexpr int OUT_BYTES = (TH / (KC == 32 ? 2 : 1)) * 16 * (BN + 4) * 4;
  constexpr int GROUP_BYTES = ((KC == 32 ? LSS_CONV_RING32 : 3) * W_BYTES + IH * IROWB + 1023) / 1024 * 1024;
  constexpr int SRC_PIECES = (MODE == 1 && KC == 32) ? (((IH - 1) / 2 + 3) * ((IW - 1) / 2 + 3) * PPP + 63) / 64 : 0;
  return (KSP * GROUP_BYTES > OUT_BYTES ? KSP * GROUP_BYTES : OUT_BYTES) + SRC_PIECES * 1024;
}

// The tile body: one workgroup's output tile.  bid / nwg: index of the tile in its class and the size of the class
// (XCD-aware order), nblk_y: index of the BN-wide channel block, oy_base: first image row of the class (0).
template <int RT, int BN, int MODE, int KH, int KW, int PAD, int KC = 64, int KSP = 1, bool HEAD = false, int TPS = 1>
__device__ __forceinline__ void conv_lds_tile(const ConvArgs& a, int tilesX, int tilesY, int bid, int nwg, int nblk_y,
                                              int oy_base, unsigned char* smem) {
  // KSP = 2: intra-workgroup split-K for grids that cannot fill the chip (layer2/layer3: 112-208
  // workgroups of 18-36 latency-bound steps on 256 CUs).  512 threads = two 4-wave groups, each
  // with its own weight ring and patch, each taking half of the input-channel chunks; the second
  // group's accumulators meet the first's through LDS before the (unchanged) epilogue.  Halves
  // the main loop at the same workgroup count; barriers stay workgroup-wide (both groups run the
  // same step sequence).
  static_assert(KSP == 1 || (KSP == 2 && KC == 64), "KSP");
  // TPS = taps per step.  2: a step is TWO taps against a double slab - half the steps (barrier + counted wait + slot
  // hand-over each) per chunk.  For the 7x7 / 2 stem, whose 64 steps carry only 8 MFMAs per wave each: phase stamps put
  // a step at 0.43 us against 0.13 us of MFMA issue.  Needs an even tap count and the 24 KB more LDS (BN = 64 only).
  static_assert(TPS == 1 || (TPS == 2 && (KH * KW) % 2 == 0 && KSP == 1 && MODE != 1), "TPS");
  // KC = input channels per (chunk, tap) step.  64: the default.  32: half-depth slabs and patch
  // (39 KB of LDS, <= 168 VGPRs) so THREE workgroups share a CU - used for the big stride-1 layers,
  // whose 728 / 1300 tiles then run in one / two full rounds instead of 1.4 / 2.5 on 512 slots.
  static_assert(KC == 64 || KC == 32, "KC");
  constexpr int PPP = KC / 8;   // 16-B pieces per patch position
  constexpr int KS = KC / 16;   // MFMA k-steps per tap
  constexpr int PSH = KC == 64 ? 3 : 2;  // log2(PPP)
  constexpr int TH = (4 / (BN / 64)) * RT * 2;  // image rows per workgroup
  constexpr bool FUSED = MODE == 1;
  constexpr int NT = KH * KW;
  constexpr int NS = NT / TPS;  // steps per chunk
  constexpr int TW = 16, IW = TW + KW - 1, IH = TH + KH - 1, POSB = KC * 2 + 16;
  constexpr int IROWB = (IW * POSB + 255) / 256 * 256;  // 2816 (KC = 64), 1536 (KC = 32)
  constexpr int IN_BYTES = IH * IROWB;
  constexpr int W1_BYTES = BN * KC * 2;      // slab of one tap
  constexpr int W_BYTES = TPS * W1_BYTES;    // slab of one step
  constexpr int WPT = W_BYTES / 4096;  // LDS-DMA instructions per wave per step (1 KiB each)
  constexpr int NWV = 4;               // waves that share one slab
  constexpr int WCOLS = BN / 64;  // waves along the channel axis
  constexpr int IPT = (IH * IW * PPP + 255) / 256;  // 16-B patch pieces per thread per chunk
  constexpr int OLD = BN + 4;  // fp32 row stride of the epilogue's staged output tile
  // the epilogue stages the fp32 output tile in LDS: whole (KC = 64) or in two halves of TH/2 rows
  constexpr int EPH = KC == 32 ? 2 : 1;
  constexpr int OUT_BYTES = (TH / EPH) * 16 * OLD * 4;  // per pixel group
  // Weight ring: 3 slots = slabs two steps ahead of the MFMAs; the KC = 32 kernels (half-length steps, three
  // workgroups per CU) keep 4 slots = three steps ahead: a timing build that re-read one L1-hot slab ran them 9-13 %
  // faster while the KC = 64 kernel did not move, i.e. their shorter steps exposed the L2 round trip of the slab.
  constexpr int NSL = KC == 32 ? LSS_CONV_RING32 : 3, LA = NSL - 1;
  constexpr int GROUP_BYTES = (NSL * W_BYTES + IN_BYTES + 1023) / 1024 * 1024;  // ring + patch of one K-split group
  // SRC (fused upsample, KC = 32): the low-res pixels a chunk's patch is interpolated from - at most
  // SRC_H x SRC_W source positions for scale factors >= 2 - are copied ONCE per chunk into LDS by
  // LDS-DMA while the previous chunk's taps run, and the 4-corner blend then reads LDS: 2 DMA
  // instructions per wave instead of 12 scattered global loads per thread in the synchronous
  // chunk-boundary phase (measured: that phase was 17 % of the fused convs).
  constexpr bool SRC = FUSED && KC == 32;
  constexpr int SRC_H = (IH - 1) / 2 + 3, SRC_W = (IW - 1) / 2 + 3;  // 7 x 11 for the 10 x 18 patch
  constexpr int SRC_PIECES = SRC ? (SRC_H * SRC_W * PPP + 63) / 64 : 0;  // 1-KiB DMA pieces per chunk (5)
  constexpr int SRC_DMA = (SRC_PIECES + 3) / 4;  // DMA instructions per chunk of the wave that issues most (wave 0)
  constexpr int SRC_BYTES = SRC_PIECES * 1024;
  constexpr int MAIN_BYTES = KSP * GROUP_BYTES > OUT_BYTES ? KSP * GROUP_BYTES : OUT_BYTES;
  constexpr int SMEM_BYTES = MAIN_BYTES + SRC_BYTES;
  static_assert(SMEM_BYTES <= (KSP == 2 ? 160 : (KC == 32 ? 53 : 80)) * 1024, "workgroups per CU vs 160 KiB of LDS");
  static_assert(!SRC || NS > LA, "the source copies retire at step LA");
  static_assert(KC == 64 || (RT == 2 && BN == 128 && MODE != 2), "KC = 32 is built for the RT = 2, BN = 128 stride-1 tiles");
  static_assert(SMEM_BYTES == conv_lds_smem_bytes<RT, BN, MODE, KH, KW, KC, KSP, TPS>(), "conv_lds_smem_bytes out of sync");
  const int grp = KSP == 2 ? (int)(threadIdx.x >> 8) : 0;  // K-split group of this wave
  unsigned char* w_tile = smem + grp * GROUP_BYTES;
  unsigned char* in_tile = w_tile + NSL * W_BYTES;
  unsigned char* src_tile = smem + MAIN_BYTES;  // SRC only

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin to
  // the 8 XCDs, so give XCD k the k-th contiguous eighth of the tile list (= a band of image
  // rows): neighbouring tiles then share halo rows through ONE L2 instead of every L2
  // pulling the whole input from the Infinity Cache.  Bijective for any grid size.
  int t;
  {
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int oy0 = oy_base + ty * TH, ox0 = tx * TW, n0 = nblk_y * BN;
  const int dwave = wave;  // index among the waves that share the weight ring
  const int wc = wave % WCOLS, wr = wave / WCOLS;
  const int prow0 = wr * (2 * RT);
  auto stamp = [&](int k) {
    if (a.stamps != nullptr && threadIdx.x == 0)
      a.stamps[((size_t)nblk_y * nwg + bid) * 8 + k] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);

  int aoff[RT], boff[2][4];
#pragma unroll
  for (int i = 0; i < RT; ++i) aoff[i] = (prow0 + 2 * i + (r >> 4)) * IROWB + (r & 15) * POSB + h * KC;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wc * 64 + i * 32 + r;
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4)
      boff[i][s4] = KC == 64 ? row * 128 + (((4 * h + s4) ^ ((row >> 1) & 7)) << 4)
                             : row * 64 + (((2 * h + s4) ^ ((row >> 2) & 3)) << 4);
  }
  f32x16 acc[RT][2];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const unsigned short* wg = reinterpret_cast<const unsigned short*>(a.w);
  const int nchunks = a.Cin / KC / KSP;  // K chunks of this group
  const int cbase = grp * nchunks;        // first chunk of this group
  const int nsteps = nchunks * NS;
  uint4 ireg[IPT];
  // weight slab of `step` -> ring slot: lane l of DMA block k lands at byte k*1024 + l*16,
  // i.e. row 8k + (l >> 3), 16-B slot l & 7, which must hold channel piece slot ^ swz(row)
  constexpr int BPT = W1_BYTES / 1024;  // 1-KiB DMA blocks per tap
  auto issue_w = [&](int step, int slot) {
    const int tap0 = (step % NS) * TPS, chunk = cbase + step / NS;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int blk = i * NWV + dwave;
      const int tap = tap0 + (TPS == 1 ? 0 : blk / BPT), rblk = TPS == 1 ? blk : blk % BPT;
      const int row = KC == 64 ? rblk * 8 + (lane >> 3) : rblk * 16 + (lane >> 2);
      const int part = KC == 64 ? (lane & 7) ^ ((row >> 1) & 7) : (lane & 3) ^ ((row >> 2) & 3);
      const int co = min(n0 + row, a.Cout - 1);  // rows past Cout: any valid address (never stored)
      glds16(wg + ((size_t)tap * a.Cout + co) * a.Cin + chunk * KC + part * 8,
             w_tile + slot * W_BYTES + blk * 1024);
    }
  };
  // MODE 1: the bilinear source coordinates of a patch piece depend only on its position,
  // not on the chunk, so each thread resolves its IPT pieces ONCE (element offset of the
  // top-left corner, +x / +y corner strides, the two blend weights, the skip-tensor
  // offset) and every chunk boundary is then 4 loads + blend + one LDS store per piece.
  // (Recomputing them per chunk cost ~11k cycles per boundary in 64-bit address math -
  // measured: 35 us of the fused up2 conv.)
  // Two VGPRs per piece: g_off = element offset of the top-left corner (a multiple of 8)
  // with flags in its 3 low bits (1: +x corner exists, 2: +y corner exists, 4: inside
  // the image); g_w = the blend weights as 16-bit fixed point (lx | ly << 16).
  int g_off[FUSED ? IPT : 1];
  unsigned int g_w[FUSED ? IPT : 1];
  // SRC: g_off holds the BYTE offset of the top-left corner inside src_tile instead (a multiple of
  // 16, same flags); src_o[i] = element offset of the source pixel this lane's i-th DMA piece copies
  const bool use_src = SRC && a.src_lds != 0;
  const int sy0 = (int)(a.ry * (float)max(oy0 - PAD, 0)), sx0 = (int)(a.rx * (float)max(ox0 - PAD, 0));
  int src_o[SRC ? SRC_DMA : 1];
  if (SRC) {
#pragma unroll
    for (int i = 0; i < SRC_DMA; ++i) {
      const int q = (i * 4 + wave) * 64 + lane;
      const int pos = min(q >> PSH, SRC_H * SRC_W - 1), part = q & (PPP - 1);  // pieces past the patch: any valid pixel
      const int sr = pos / SRC_W, sc = pos - sr * SRC_W;
      src_o[i] = ((b * a.H + min(sy0 + sr, a.H - 1)) * a.W + min(sx0 + sc, a.W - 1)) * a.Cx + part * 8;
    }
  }
  if (FUSED) {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int q = tid + i * 256;
      g_off[i] = 0; g_w[i] = 0;
      if (q < IH * IW * PPP) {
        const int pos = q >> PSH, part = q & (PPP - 1);
        const int py = pos / IW, px = pos - py * IW;
        const int iy = oy0 - PAD + py, ix = ox0 - PAD + px;
        if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
          const float sy = a.ry * (float)iy, sx = a.rx * (float)ix;
          const int y0 = (int)sy, x0 = (int)sx;
          const unsigned int wy = (unsigned int)((sy - (float)y0) * 65536.f + 0.5f);
          const unsigned int wx = (unsigned int)((sx - (float)x0) * 65536.f + 0.5f);
          g_w[i] = min(wx, 65535u) | (min(wy, 65535u) << 16);
          const int flags = (x0 < a.W - 1 ? 1 : 0) | (y0 < a.H - 1 ? 2 : 0) | 4;
          if (use_src) g_off[i] = ((((y0 - sy0) * SRC_W + (x0 - sx0)) * PPP + part) * 16) | flags;
          else g_off[i] = (((b * a.H + y0) * a.W + x0) * a.Cx + part * 8) | flags;
        }
      }
    }
  }
  auto gather_fused = [&](int chunk) {
    const unsigned short* xp = reinterpret_cast<const unsigned short*>(a.x);
    const unsigned short* x2p = reinterpret_cast<const unsigned short*>(a.x2);
    const int c0 = (cbase + chunk) * KC;
    const bool skip = c0 < a.C2;  // chunk of the skip tensor x2: plain copy
    const int dxs = a.Cx, dys = a.W * a.Cx;
#pragma unroll 2
    for (int i = 0; i < IPT; ++i) {
      const int q = tid + i * 256;
      if (q >= IH * IW * PPP) continue;
      const int pos = q >> PSH, part = q & (PPP - 1);
      const int py = pos / IW, px = pos - py * IW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g_off[i] & 4) {
        if (skip) {
          const int iy = oy0 - PAD + py, ix = ox0 - PAD + px;
          v = *reinterpret_cast<const uint4*>(x2p + ((b * a.Hin + iy) * a.Win + ix) * a.C2 + c0 + part * 8);
        } else if (use_src) {
          const unsigned char* p00 = src_tile + (g_off[i] & ~15);
          const int dx = (g_off[i] & 1) ? PPP * 16 : 0, dy = (g_off[i] & 2) ? SRC_W * PPP * 16 : 0;
          const uint4 q00 = *reinterpret_cast<const uint4*>(p00);
          const uint4 q01 = *reinterpret_cast<const uint4*>(p00 + dx);
          const uint4 q10 = *reinterpret_cast<const uint4*>(p00 + dy);
          const uint4 q11 = *reinterpret_cast<const uint4*>(p00 + dy + dx);
          v = blend_bf16x8(q00, q01, q10, q11, (float)(g_w[i] & 0xffff) * (1.f / 65536.f),
                           (float)(g_w[i] >> 16) * (1.f / 65536.f));
        } else {
          const unsigned short* p00 = xp + (g_off[i] & ~7) + (c0 - a.C2);
          const int dx = (g_off[i] & 1) ? dxs : 0, dy = (g_off[i] & 2) ? dys : 0;
          const uint4 q00 = *reinterpret_cast<const uint4*>(p00);
          const uint4 q01 = *reinterpret_cast<const uint4*>(p00 + dx);
          const uint4 q10 = *reinterpret_cast<const uint4*>(p00 + dy);
          const uint4 q11 = *reinterpret_cast<const uint4*>(p00 + dy + dx);
          v = blend_bf16x8(q00, q01, q10, q11, (float)(g_w[i] & 0xffff) * (1.f / 65536.f),
                           (float)(g_w[i] >> 16) * (1.f / 65536.f));
        }
      }
      *reinterpret_cast<uint4*>(in_tile + py * IROWB + px * POSB + part * 16) = v;
    }
  };
  // SRC: source pixels of `chunk` -> src_tile (lane l of DMA block k lands at byte k*1024 + l*16)
  auto issue_src = [&](int chunk) {
    const unsigned short* xp = reinterpret_cast<const unsigned short*>(a.x);
    const int cx = (cbase + chunk) * KC - a.C2;
#pragma unroll
    for (int i = 0; i < (SRC ? SRC_DMA : 0); ++i)
      if (i * 4 + wave < SRC_PIECES) glds16(xp + src_o[i] + cx, src_tile + (i * 4 + wave) * 1024);
  };
  // does `chunk` read the upsampled tensor through src_tile?
  auto src_chunk = [&](int chunk) { return use_src && chunk < nchunks && (cbase + chunk) * KC >= a.C2; };

  // input patch of one 64-channel chunk.  !FUSED: plain 16-B loads, unrolled so they
  // can be parked in registers (prefetch).  FUSED: the 4-corner bilinear gather goes
  // straight to LDS piece by piece (rolled loop: keeps the live set small).
  auto gather_in = [&](int chunk, bool to_lds) {
    if (FUSED) {
#pragma unroll 1
      for (int q = tid; q < IH * IW * PPP; q += 256) {
        const int pos = q >> PSH, part = q & (PPP - 1);
        const int py = pos / IW, px = pos - py * IW;
        const int iy = oy0 - PAD + py, ix = ox0 - PAD + px;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
          v = load_in_piece<unsigned short, true>(a, b, iy, ix, (cbase + chunk) * KC + part * 8);
        *reinterpret_cast<uint4*>(in_tile + py * IROWB + px * POSB + part * 16) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < IPT; ++i) {
        const int q = tid + i * 256;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < IH * IW * PPP) {
          const int pos = q >> PSH, part = q & (PPP - 1);
          const int py = pos / IW, px = pos - py * IW;
          int iy = oy0 - PAD + py, ix = ox0 - PAD + px, cc = cbase + chunk;
          if (MODE == 2) {  // (iy, ix) are phase-plane coordinates; chunk -> (phase, channel block)
            const int nblk = a.Cx / KC;
            const int ph = cc / nblk;
            cc = cc - ph * nblk;
            iy = 2 * iy + (ph >> 1);
            ix = 2 * ix + (ph & 1);
          }
          if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
            v = load_in_piece<unsigned short, false>(a, b, iy, ix, cc * KC + part * 8);
          if (to_lds) *reinterpret_cast<uint4*>(in_tile + py * IROWB + px * POSB + part * 16) = v;
        }
        if (!to_lds) ireg[i] = v;
      }
    }
  };
  auto store_in = [&]() {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int q = tid + i * 256;
      if (q < IH * IW * PPP) {
        const int pos = q >> PSH, part = q & (PPP - 1);
        const int py = pos / IW, px = pos - py * IW;
        *reinterpret_cast<uint4*>(in_tile + py * IROWB + px * POSB + part * 16) = ireg[i];
      }
    }
  };

  // one 1-KiB LDS-DMA block of the slab of `step` (block index i of this wave)
  auto issue_w1 = [&](int st, int chunk, int slot, int i) {  // st = step inside the chunk
    const int blk = i * NWV + dwave;
    const int tap = st * TPS + (TPS == 1 ? 0 : blk / BPT), rblk = TPS == 1 ? blk : blk % BPT;
    const int row = KC == 64 ? rblk * 8 + (lane >> 3) : rblk * 16 + (lane >> 2);
    const int part = KC == 64 ? (lane & 7) ^ ((row >> 1) & 7) : (lane & 3) ^ ((row >> 2) & 3);
    const int co = min(n0 + row, a.Cout - 1);
    glds16(wg + ((size_t)tap * a.Cout + co) * a.Cin + (cbase + chunk) * KC + part * 8,
           w_tile + slot * W_BYTES + blk * 1024);
  };

  // per-lane epilogue constants, fetched now so their latency is long gone by the epilogue
  float esc[2], esh[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int co = n0 + wc * 64 + ct * 32 + r;
    const bool cok = co < a.Cout;
    esc[ct] = (cok && a.scale) ? a.scale[co] : 1.f;
    esh[ct] = (cok && a.shift) ? a.shift[co] : 0.f;
  }

  // Output pieces of this thread (used by the epilogue; computed here because the residual rows of the launch-bound
  // layers are requested from inside the main loop, see EARLY_RES).  A piece = 8 consecutive channels (16 B of bf16) of
  // one pixel; 256 threads cover PPJ pixels x CG channel groups per pass, so a thread's pixel column, channel group and
  // first row never change: every per-pass address is one base + a compile-time offset.
  constexpr int CG = BN / 8;              // 16-B channel groups per pixel
  constexpr int PPJ = 256 / CG;           // pixels per pass
  constexpr int RPJ = PPJ / 16;           // image rows per pass
  constexpr int HROWS = TH / EPH;         // image rows per staged half (EPH = 2: KC = 32, see OUT_BYTES)
  constexpr int HPIECES = HROWS * 16 * CG;
  constexpr int EPJ = HPIECES / 256;      // passes per half
  static_assert(HPIECES % 256 == 0 && PPJ % 16 == 0, "half tiles split evenly over the threads");
  const int pix0 = tid / CG, c8 = tid % CG;
  const int ox = ox0 + (pix0 & 15), oyb = oy0 + (pix0 >> 4);
  const int co = n0 + c8 * 8;
  const bool col_ok = grp == 0 && ox < a.Wo && co < a.Cout;
  const size_t pixb = ((size_t)b * a.Ho + oyb) * a.Wo + ox;  // pixel index of pass 0 of half 0
  const int rstep = a.Wo * a.Cout;                            // residual elements per image row
  const unsigned short* res = reinterpret_cast<const unsigned short*>(a.residual);
  const bool res_vec = !HEAD && res != nullptr && col_ok && (a.Cout & 7) == 0 && co + 8 <= a.Cout;
  // EARLY_RES: the residual rows of the launch-bound layers (RT = 1: 152-190 registers, room for 16 more) are
  // requested at the top of the LAST chunk's taps instead of at the top of the epilogue: phase stamps put the store
  // phase of the residual layers at 1.6-1.7 us against 1.0-1.1 us for the same tile without a residual - the loads,
  // issued ~0.6 us before their first use, were still in flight when the store loop wanted them.
  constexpr bool EARLY_RES = RT == 1 && EPH == 1 && !HEAD;
  uint4 rres[HEAD ? 1 : EPJ];
  auto load_residual = [&](int half) {
    if (!HEAD) {
      const unsigned short* rb = res + pixb * a.Cout + co;
#pragma unroll
      for (int j = 0; j < EPJ; ++j) {
        const int row = half * HROWS + j * RPJ;
        rres[j] = make_uint4(0, 0, 0, 0);
        if (res_vec && oyb + row < a.Ho) rres[j] = *reinterpret_cast<const uint4*>(rb + row * rstep);
      }
    }
  };

  // prologue: W(0), W(1) on their way; patch of chunk 0
  if (src_chunk(0)) issue_src(0);
  issue_w(0, 0);
#pragma unroll
  for (int q = 1; q < LA; ++q)
    if (nsteps > q) issue_w(q, q);
  if (src_chunk(0)) {
    wait_vmcnt<0>();
    lds_barrier();  // the source pixels of chunk 0 are in LDS for every wave
  }
  if (FUSED) gather_fused(0);
  else gather_in(0, true);
  wait_vmcnt<0>();
  lds_barrier();
  stamp(1);

  // Main loop.  The NT taps of a chunk are fully unrolled (static patch offsets and
  // fragment double buffer); the ring slot advances at run time.  Rolling prefetch:
  // while the 4 MFMAs of one k-step run, the 2 A + 2 B fragments of the NEXT k-step
  // are already in flight (32 fragment VGPRs in total), and one DMA block of the slab
  // two steps ahead is issued per k-step, so LDS latency and DMA issue hide behind
  // MFMAs.
  constexpr int PF_ST = NS >= 5 ? NS - 4 : 0;  // the step at which the next patch's loads are issued
  bf16x8 fa[2][RT], fb[2][2];  // [k-step parity][tile]
  auto read_a = [&](int buf, int tap, int s4) {
    const int toff = (tap / KW) * IROWB + (tap % KW) * POSB + s4 * 16;
#pragma unroll
    for (int i = 0; i < RT; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(in_tile + aoff[i] + toff);
  };
  auto read_b = [&](int buf, int slot, int ks) {  // ks = k-step inside the step: tap ks / KS of the slab
    const unsigned char* wbuf = w_tile + slot * W_BYTES + (ks / KS) * W1_BYTES;
    fb[buf][0] = *reinterpret_cast<const bf16x8*>(wbuf + boff[0][ks % KS]);
    fb[buf][1] = *reinterpret_cast<const bf16x8*>(wbuf + boff[1][ks % KS]);
  };
  int slot = 0;  // ring slot of the current step's slab
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const bool last_chunk = chunk + 1 == nchunks;
    const bool srcq = src_chunk(chunk + 1);  // this chunk's tap 0 also sends the next chunk's source pixels
    if (EARLY_RES && last_chunk) load_residual(0);  // (younger than every DMA piece the counted waits below name)
    read_a(0, 0, 0);
    read_b(0, slot, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      // slab LA steps ahead: step st+LA of this chunk, or st+LA-NS of the next one
      const int t2 = (st + LA) % NS;
      const int c2 = chunk + (st + LA) / NS;
      const bool more = c2 < nchunks;
      const bool prefetch = !FUSED && st == PF_ST && !last_chunk;
      const int slot2 = slot >= 1 ? slot - 1 : NSL - 1;  // (slot + LA) % NSL
#pragma unroll
      for (int ks = 0; ks < KS * TPS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < KS * TPS - 1) {
          read_a(nxt, st * TPS + (ks + 1) / KS, (ks + 1) % KS);
          read_b(nxt, slot, ks + 1);
        } else if (st < NS - 1) {
          read_a(nxt, (st + 1) * TPS, 0);  // next step's weights are only readable after the barrier
        }
        // keep the prefetch reads AHEAD of this k-step's MFMAs (hipcc's scheduler would
        // otherwise sink them next to their use and expose the LDS latency again)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][0], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][1], acc[i][1], 0, 0, 0);
        }
        if (more && ks < WPT) issue_w1(t2, c2, slot2, ks);
      }
      // after this step's weight pieces, so that (like them) the copies have two steps to land: the
      // counted waits of step 0 and step 1 leave them in flight, the wait of step 2 retires them
      if (SRC && st == 0 && srcq) issue_src(chunk + 1);
      if (prefetch) gather_in(chunk + 1, false);  // next patch -> registers
      if (st == NS - 1 && !last_chunk) {
        lds_barrier();  // every wave is done with this chunk's patch
        if (FUSED) gather_fused(chunk + 1);
        else store_in();
      }
      // W(step+1) must have landed in every wave's share before anyone reads it; only the
      // DMA (and patch loads) issued during THIS step may stay in flight
      // (younger than W(step+1): the slabs of steps step+2 .. step+LA, and - in steps 0 .. LA-1 of a chunk - the source
      // copies issued in its step 0: one per wave, two for the waves that carry the pieces past the fourth)
      constexpr int WYOUNG = (LA - 1) * WPT;
      if (prefetch) wait_vmcnt<WYOUNG + IPT>();
      else if (SRC && st < LA && srcq && more) {
        if (SRC_DMA == 2 && wave + 4 < SRC_PIECES) wait_vmcnt<WYOUNG + 2>();
        else wait_vmcnt<WYOUNG + (SRC_DMA >= 1 ? 1 : 0)>();
      }
      else if (more) wait_vmcnt<WYOUNG>();
      else wait_vmcnt<0>();
      lds_barrier();
      slot = slot == NSL - 1 ? 0 : slot + 1;
      if (st < NS - 1) read_b(0, slot, 0);
    }
  }

  // epilogue.  D[row = (i&3) + 8*(i>>2) + 4*h][col = r]: a lane holds ONE output
  // channel of 16 pixels, so storing from registers would be 2-B scattered stores.
  // Instead: scale/shift in registers -> fp32 tile in LDS (the ring + patch area is
  // free now) -> every thread picks up 8 consecutive channels of a pixel, adds the
  // residual (one 16-B load), ReLU, rounds to bf16 once, and stores 16 B.
  float* otile = reinterpret_cast<float*>(smem);
  stamp(2);
  // (the loop's final lds_barrier already guarantees every wave is done reading LDS)
  if (KSP == 2) {
    // split-K: the second group's partial sums meet the first group's through LDS (same lane ->
    // same element in both groups, so only the group hand-over needs barriers)
    if (grp == 1) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            otile[((prow0 + 2 * rt + (row >> 4)) * 16 + (row & 15)) * OLD + wc * 64 + ct * 32 + r] = acc[rt][ct][i];
          }
    }
    lds_barrier();
    if (grp == 0) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            acc[rt][ct][i] += otile[((prow0 + 2 * rt + (row >> 4)) * 16 + (row & 15)) * OLD + wc * 64 + ct * 32 + r];
          }
    }
    lds_barrier();  // the partials are consumed before group 0 restages the tile below
  }
  // (the epilogue of the launch-bound layers is instruction-issue time, not bandwidth)
  // dual-output launches: this workgroup's channel block belongs to y (columns [0, split)) or y2
  const bool second = a.y2 != nullptr && n0 >= a.split;
  const int ycol0 = second ? a.split : 0;                                   // first channel of the tensor
  const int ycw = a.y2 == nullptr ? a.Cout : (second ? a.Cout - a.split : a.split);  // its channel count
  const int eact = n0 < a.relu_n ? a.relu : 0;
  const bool vec_ok = (ycw & 7) == 0 && co + 8 <= a.Cout;  // 16-B aligned, whole channel group
  const int ystep = a.Wo * ycw;                               // output elements per image row
  // residual rows of ONE staged half: requested at the top of the half's iteration, before its accumulators are
  // staged, so the loads fly during the LDS write / barrier / read-back instead of stalling the store loop.  (Per
  // half, not for the whole tile: 16 instead of 32 live registers in the 168-register KC = 32 kernels.)  The RT = 1
  // kernels have requested them inside the main loop already (EARLY_RES).
  unsigned short* y = reinterpret_cast<unsigned short*>(second ? a.y2 : a.y);
  const size_t obase = pixb * ycw + (co - ycol0);
  const __amdgpu_buffer_rsrc_t yrsrc =
      __builtin_amdgcn_make_buffer_rsrc(y, 0, a.wt ? (int)((size_t)a.M * ycw * 2) : 0, 0x00020000);
  // staging address of this lane: D[row = (i&3) + 8*(i>>2) + 4*h][col = r], and (i&3) + 4*h < 16, so
  // pixel = (2*rt + (i>>3))*16 + (i&3) + 8*((i>>2)&1) + 4*h: lane part + compile-time part
  float* const ost = otile + (prow0 * 16 + 4 * h) * OLD + wc * 64 + r;
#pragma unroll
  for (int half = 0; half < EPH; ++half) {
    const bool mine = grp == 0 && (EPH == 1 || (prow0 / HROWS) == half);  // this wave's rows belong to the half
    if (!EARLY_RES) load_residual(half);
    if (mine) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float sc = esc[ct], sh = esh[ct];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int pc = (2 * rt + (i >> 3)) * 16 + (i & 3) + 8 * ((i >> 2) & 1) - half * HROWS * 16;
            ost[pc * OLD + ct * 32] = acc[rt][ct][i] * sc + sh;
          }
      }
      if (!HEAD && a.stats) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int cs = n0 + wc * 64 + ct * 32 + r;
          const bool cok = cs < a.Cout;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
              const int sy = oy0 + prow0 + 2 * rt + (row >> 4), sx = ox0 + (row & 15);
              if (cok && sy < a.Ho && sx < a.Wo) {
                const float raw = acc[rt][ct][i];
                s1 += raw;
                s2 += raw * raw;
              }
            }
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (h == 0 && cok) {
            atomicAdd(a.stats + cs, s1);
            atomicAdd(a.stats + a.Cout + cs, s2);
          }
        }
      }
    }
    lds_barrier();
    if (half == 0) stamp(3);
    if (grp != 0) {
      // second K-split group: its sums were handed over above; it only keeps the barriers company
    } else if (HEAD && a.head_n <= 16) {
      // Fused 1x1 head ON THE MATRIX PIPE (ref src/modules.py:115, up2[4]): per image row of the staged tile one
      // 16 x 16 x BN product  out[pixel][k] = sum_c relu(act[pixel][c]) * head_w[k][c]  on v_mfma_f32_16x16x32_bf16.
      // Both operands are split into bf16 hi + lo parts (x = hi + lo to 16 significant bits) and the three
      // products hi*hi, hi*lo, lo*hi are accumulated in fp32, so the result keeps the accuracy of the fp32
      // head it replaces (the dropped lo*lo term is 2^-16 relative) while the VALU work per pixel falls from
      // ~60 FMA / DPP instructions to ~12 conversions.  A: lane (r = pixel, q) <- 8 consecutive channels of the
      // fp32 tile in LDS (2 ds_read_b128); B: lane (k = lane & 15, q) <- 8 consecutive weights of class k from
      // global (L1-resident: 2 KiB); D: lane (k, q) holds pixels 4q..4q+3 of the row = ONE 16-B NCHW store.
      typedef __attribute__((ext_vector_type(4))) float f32x4_t;
      const int hr = lane & 15, hq = lane >> 4;
      for (int row = wave; row < HROWS; row += 4) {
        const int oy = oy0 + half * HROWS + row;
        f32x4_t hacc = {0.f, 0.f, 0.f, 0.f};
        const float* arow = otile + (row * 16 + hr) * OLD + hq * 8;
        const float* wrow = a.head_w + (size_t)min(hr, a.head_n - 1) * BN + hq * 8;
        // rolled: one k-step's 16 operand floats live at a time (the other half-tile's waves still hold their
        // 64 accumulator registers here, and the 168-register budget of three workgroups per CU is what counts)
#pragma unroll 1
        for (int s = 0; s < BN / 32; ++s) {
          bf16x8 xh, xl, wh, wl;
          {
            f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + s * 32);
            f32x4 w1 = *reinterpret_cast<const f32x4*>(wrow + s * 32 + 4);
            if (hr >= a.head_n) { w0 = (f32x4){0.f, 0.f, 0.f, 0.f}; w1 = w0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float wv = j < 4 ? w0[j & 3] : w1[j & 3];
              const unsigned short whi = lss_f2bf(wv);
              wh[j] = (short)whi;
              wl[j] = (short)lss_f2bf(wv - lss_bf2f(whi));
            }
          }
          {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(arow + s * 32);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(arow + s * 32 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float xv = j < 4 ? x0[j & 3] : x1[j & 3];
              if (a.relu) xv = fmaxf(xv, 0.f);
              const unsigned short xhi = lss_f2bf(xv);
              xh[j] = (short)xhi;
              xl[j] = (short)lss_f2bf(xv - lss_bf2f(xhi));
            }
          }
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, hacc, 0, 0, 0);
        }
        if (hr < a.head_n && oy < a.Ho) {
          const float hb = a.head_b[hr];
          float* op = a.head_out + (((size_t)b * a.head_n + hr) * a.Ho + oy) * a.Wo + ox0 + hq * 4;
          if ((a.Wo & 3) == 0 && ox0 + hq * 4 + 4 <= a.Wo) {
            *reinterpret_cast<f32x4_t*>(op) = (f32x4_t){hacc[0] + hb, hacc[1] + hb, hacc[2] + hb, hacc[3] + hb};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (ox0 + hq * 4 + i < a.Wo) op[i] = hacc[i] + hb;
          }
        }
      }
    } else if (HEAD) {
      // fused 1x1 head, VALU form (more than 16 classes): the CG = BN / 8 lanes
      // that hold the channels of one pixel reduce their partial dot products with DPP row operations; the
      // activation itself is never stored
      {
        for (int e = tid; e < HROWS * 16 * CG; e += 256) {
          const int pl = e / CG, hc8 = e % CG;
          const int oy = oy0 + half * HROWS + (pl >> 4), hx = ox0 + (pl & 15);
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(otile + pl * OLD + hc8 * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(otile + pl * OLD + hc8 * 8 + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if (a.relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          for (int k = 0; k < a.head_n; ++k) {
            const float* hw = a.head_w + k * BN + hc8 * 8;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) part = fmaf(v[j], hw[j], part);
            // sum over the CG lanes of the pixel on DPP row operations (no LDS round trips): lane ^ 1,
            // lane ^ 2, then - every quad now holding its sum - the other quad of the 8 (row_half_mirror)
            // and the other half of the 16 (row_mirror)
            part += dpp_f32<0xB1>(part);
            part += dpp_f32<0x4E>(part);
            part += dpp_f32<0x141>(part);
            if (CG == 16) part += dpp_f32<0x140>(part);
            if (hc8 == 0 && oy < a.Ho && hx < a.Wo)
              a.head_out[(((size_t)b * a.head_n + k) * a.Ho + oy) * a.Wo + hx] = part + a.head_b[k];
          }
        }
      }
    } else if (col_ok) {
      const float* const ord = otile + pix0 * OLD + c8 * 8;
#pragma unroll
      for (int j = 0; j < EPJ; ++j) {
        const int row = half * HROWS + j * RPJ;  // image row of this pass relative to oyb
        if (oyb + row >= a.Ho) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ord + j * PPJ * OLD);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(ord + j * PPJ * OLD + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const size_t o = obase + (size_t)(row * ystep);
        if (vec_ok) {
          if (res_vec) {
            const uint4 rv = rres[j];
            const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[2 * k] += lss_bf2f((unsigned short)(ru[k] & 0xffff));
              v[2 * k + 1] += lss_bf2f((unsigned short)(ru[k] >> 16));
            }
          } else if (res) {
            const unsigned short* rp = res + (pixb * a.Cout + co) + (size_t)(row * rstep);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += lss_bf2f(rp[k]);
          }
          if (eact == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
          } else if (eact == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = conv_act(v[k], 2);
          }
          if (a.out_f32) {
            float* yf = reinterpret_cast<float*>(second ? a.y2 : a.y) + o;
            *reinterpret_cast<f32x4*>(yf) = (f32x4){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(yf + 4) = (f32x4){v[4], v[5], v[6], v[7]};
          } else {
            uint4 ov;
            ov.x = lss_pack_bf2(v[0], v[1]); ov.y = lss_pack_bf2(v[2], v[3]);
            ov.z = lss_pack_bf2(v[4], v[5]); ov.w = lss_pack_bf2(v[6], v[7]);
            if (a.wt) {
              // write-through (sc1) 16-B store: a dependent kernel boundary otherwise pays
              // (dirty bytes / 6 TB/s) for the L2 write-back (MI355X_MICROARCH.md, row 'boundary')
              typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
              __builtin_amdgcn_raw_buffer_store_b128((u32x4){ov.x, ov.y, ov.z, ov.w}, yrsrc, (int)(o * 2), 0, 16);
            } else {
              *reinterpret_cast<uint4*>(y + o) = ov;
            }
          }
        } else {
          // ragged channel counts: element by element
          const size_t ro = (pixb * a.Cout + co) + (size_t)(row * rstep);
          for (int k = 0; k < 8 && co + k < a.Cout; ++k) {
            float t = v[k];
            if (res) t += lss_bf2f(res[ro + k]);
            t = conv_act(t, eact);
            if (a.out_f32) reinterpret_cast<float*>(second ? a.y2 : a.y)[o + k] = t;
            else y[o + k] = lss_f2bf(t);
          }
        }
      }
    }
    if (half + 1 < EPH) lds_barrier();  // everyone has read this half before the next one is staged
  }
  if (a.stamps != nullptr) {
    stamp(4);
    wait_vmcnt<0>();  // stores acknowledged
    stamp(5);
  }
}

template <int RT, int BN, int MODE, int KH, int KW, int PAD, int KC = 64, int KSP = 1, bool HEAD = false, int TPS = 1>
__global__ __launch_bounds__(256 * KSP, KSP == 2 ? 1 : (KC == 32 ? 3 : 2)) void conv_lds_kernel(ConvArgs a, int tilesX,
                                                                                                  int tilesY) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[conv_lds_smem_bytes<RT, BN, MODE, KH, KW, KC, KSP, TPS>()];
  conv_lds_tile<RT, BN, MODE, KH, KW, PAD, KC, KSP, HEAD, TPS>(a, tilesX, tilesY, blockIdx.x, gridDim.x, blockIdx.y, 0, smem);
}

// OIHW fp32 -> [tap][co][ci] in T
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int KHW,
                                    T* __restrict__ out) {
  const size_t n = (size_t)Cout * Cin * KHW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int ci = e % Cin;
    const size_t t = e / Cin;
    const int co = t % Cout;
    const int tap = t / Cout;
    const float v = w[((size_t)co * Cin + ci) * KHW + tap];
    if (sizeof(T) == 2) reinterpret_cast<unsigned short*>(out)[e] = lss_f2bf(v);
    else reinterpret_cast<float*>(out)[e] = v;
  }
}

// OIHW fp32 (k x k, stride 2, pad p) -> [tap'][co][phase*Cin + ci] bf16 of the
// equivalent stride-1 conv over the 4 parity phases: input row 2*oy + ky - p =
// 2*(oy + ty) + py with py = (ky - p) & 1, ty = (ky - p - py) / 2.
__global__ void pack_weights_s2d_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int pad,
                                        int KT, int tmin, unsigned short* __restrict__ out) {
  const size_t n = (size_t)KT * KT * Cout * 4 * Cin;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int ci = e % Cin;
    size_t t = e / Cin;
    const int ph = t % 4; t /= 4;
    const int co = t % Cout; t /= Cout;
    const int tx = t % KT, ty = t / KT;
    const int ky = 2 * (ty + tmin) + (ph >> 1) + pad, kx = 2 * (tx + tmin) + (ph & 1) + pad;
    float v = 0.f;
    if (ky >= 0 && ky < K && kx >= 0 && kx < K) v = w[(((size_t)co * Cin + ci) * K + ky) * K + kx];
    out[e] = lss_f2bf(v);
  }
}

// floor((0 - p) / 2) .. floor((k - 1 - p) / 2): tap range of the phase-plane conv
inline int s2d_tmin(int pad) { return -((pad + 1) / 2); }
inline int s2d_taps(int K, int pad) {
  const int hi = (K - 1 - pad) >= 0 ? (K - 1 - pad) / 2 : -((pad - K + 2) / 2);
  return hi - s2d_tmin(pad) + 1;
}

// May the KC = 32 fused kernel stage its low-res source pixels in LDS?  Its 10 x 18 patch must map
// into the SRC_H x SRC_W = 7 x 11 source window: floor(9 ry) + 2 <= 7 and floor(17 rx) + 2 <= 11.
inline int conv_src_lds_ok(const ConvArgs& a) {
  if (const char* e = getenv("LSS_CONV_SRC"))
    if (atoi(e) == 0) return 0;
  return a.up >= 2 && a.ry * 9.f < 4.99f && a.rx * 17.f < 8.99f;
}

// Tile selection + launch of conv_lds_kernel.  BN = 64 for narrow layers, else 128; RT = 2
// (throughput shape) unless that grid would leave the 256 CUs under-filled, in which case
// half-height workgroups (RT = 1) shorten the per-workgroup critical path instead.
template <int MODE, int KH, int KW, int PAD>
void launch_conv_lds(const ConvArgs& a, hipStream_t st) {
  const int tilesX = lss_cdiv(a.Wo, 16);
  const bool narrow = a.Cout <= 64;
  const int th2 = narrow ? 16 : 8;
  const int nblk = lss_cdiv(a.Cout, narrow ? 64 : 128);
  const long long nwg2 = (long long)tilesX * lss_cdiv(a.Ho, th2) * a.B * nblk;
  int rt = nwg2 < 384 ? 1 : 2;
  if (const char* e = getenv("LSS_CONV_RT")) rt = atoi(e) == 1 ? 1 : 2;
  const int th = rt == 2 ? th2 : th2 / 2;
  const int tilesY = lss_cdiv(a.Ho, th);
  dim3 g(tilesX * tilesY * a.B, nblk);
  if (narrow) {
    if (rt == 2) hipLaunchKernelGGL((conv_lds_kernel<2, 64, MODE, KH, KW, PAD>), g, dim3(256), 0, st, a, tilesX, tilesY);
    else {
      if constexpr (MODE == 2 && KH == 4) {
        // the 7x7 / 2 stem: two taps per step (LSS_CONV_TPS=1 = one, for A/B)
        static const bool tps2 = getenv("LSS_CONV_TPS") == nullptr || atoi(getenv("LSS_CONV_TPS")) != 1;
        if (tps2) {
          hipLaunchKernelGGL((conv_lds_kernel<1, 64, MODE, KH, KW, PAD, 64, 1, false, 2>), g, dim3(256), 0, st, a, tilesX,
                             tilesY);
          return;
        }
      }
      hipLaunchKernelGGL((conv_lds_kernel<1, 64, MODE, KH, KW, PAD>), g, dim3(256), 0, st, a, tilesX, tilesY);
    }
  } else {
    // Half-depth steps, three workgroups per CU.  Measured (same box, B = 4): the fused
    // upsample/concat convs gain (up1.conv0 100 -> 83 us, up2.1 141 -> 133 us: the third workgroup
    // fills the synchronous gather phases), the plain 3x3 loses 5 % (half-depth steps double the
    // barriers per MFMA) - so MODE 1 takes KC = 32 by default, MODE 0 keeps 64 (LSS_CONV_KC overrides).
    bool kc32 = MODE == 1;
    if (const char* e = getenv("LSS_CONV_KC")) kc32 = atoi(e) == 32;
    kc32 = kc32 && MODE != 2 && KH == 3 && rt == 2 && a.Cx % 32 == 0 && a.C2 % 32 == 0;
    if (rt == 2) {
      if constexpr (MODE != 2 && KH == 3) {
        if (kc32) {
          ConvArgs a32 = a;
          a32.src_lds = MODE == 1 ? conv_src_lds_ok(a) : 0;
          hipLaunchKernelGGL((conv_lds_kernel<2, 128, MODE, KH, KW, PAD, 32>), g, dim3(256), 0, st, a32, tilesX, tilesY);
          return;
        }
      }
      hipLaunchKernelGGL((conv_lds_kernel<2, 128, MODE, KH, KW, PAD>), g, dim3(256), 0, st, a, tilesX, tilesY);
    } else {
      // Grids that cannot even give every CU one workgroup (layer2 / layer3 at batch 4): split the
      // K loop over two 4-wave groups inside the workgroup - half the latency-bound steps each.
      const long long nwg1 = (long long)g.x * g.y;
      bool ksplit = MODE != 1 && nwg1 <= 256 && (a.Cin / 64) % 2 == 0 && a.Cin >= 128;
      if (const char* e = getenv("LSS_CONV_KSPLIT")) ksplit = ksplit && atoi(e) != 0;
      if constexpr (MODE != 1) {
        if (ksplit) {
          hipLaunchKernelGGL((conv_lds_kernel<1, 128, MODE, KH, KW, PAD, 64, 2>), g, dim3(512), 0, st, a, tilesX, tilesY);
          return;
        }
      }
      hipLaunchKernelGGL((conv_lds_kernel<1, 128, MODE, KH, KW, PAD>), g, dim3(256), 0, st, a, tilesX, tilesY);
    }
  }
}

}  // namespace

extern "C" size_t lss_conv2d_s2d_packed_weight_bytes(int Cout, int Cin, int K, int pad) {
  if (Cout <= 0 || Cin <= 0 || K <= 0 || pad < 0) return 0;
  const int kt = s2d_taps(K, pad);
  return (size_t)kt * kt * Cout * 4 * Cin * 2;
}

extern "C" int lss_conv2d_pack_weights_s2d(const float* w_oihw, int Cout, int Cin, int K, int pad,
                                           void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  LSS_CHECK_POS(Cout); LSS_CHECK_POS(Cin); LSS_CHECK_POS(K);
  if (pad < 0) return LSS_E_SHAPE;
  const int kt = s2d_taps(K, pad);
  const size_t n = (size_t)kt * kt * Cout * 4 * Cin;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_s2d_kernel, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cout,
                     Cin, K, pad, kt, s2d_tmin(pad), reinterpret_cast<unsigned short*>(w_packed));
  return lss_launch_status();
}

extern "C" size_t lss_conv2d_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int dt) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
  return (size_t)Cout * Cin * KH * KW * (dt == LSS_DT_BF16 ? 2 : 4);
}

extern "C" int lss_conv2d_pack_weights(const float* w_oihw, int Cout, int Cin, int KH, int KW,
                                       int dt, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  LSS_CHECK_POS(Cout); LSS_CHECK_POS(Cin); LSS_CHECK_POS(KH); LSS_CHECK_POS(KW);
  const size_t n = (size_t)Cout * Cin * KH * KW;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (dt == LSS_DT_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<unsigned short>, dim3(grid), dim3(256), 0,
                       lss_stream(stream), w_oihw, Cout, Cin, KH * KW,
                       reinterpret_cast<unsigned short*>(w_packed));
  else if (dt == LSS_DT_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, lss_stream(stream),
                       w_oihw, Cout, Cin, KH * KW, reinterpret_cast<float*>(w_packed));
  else
    return LSS_E_LAYOUT;
  return lss_launch_status();
}

extern "C" int lss_conv2d_fwd(const void* x, const void* x2, const void* w_packed,
                              const float* scale, const float* shift, const void* residual,
                              void* y, float* stats, int B, int H, int W, int Cx, int C2, int up,
                              int Cout, int KH, int KW, int stride, int pad, int relu, int dt,
                              void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(w_packed); LSS_CHECK_PTR(y);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cx); LSS_CHECK_POS(Cout);
  LSS_CHECK_POS(KH); LSS_CHECK_POS(KW); LSS_CHECK_POS(stride); LSS_CHECK_POS(up);
  if (pad < 0 || C2 < 0) return LSS_E_SHAPE;
  if (dt != LSS_DT_F32 && dt != LSS_DT_BF16) return LSS_E_LAYOUT;
  if (C2 > 0 && x2 == nullptr) return LSS_E_NULL;
  if (relu & LSS_W_KS) {  // KS-packed weights: the K-split one-pass kernel of the launch-bound layers, or nothing
    if (dt != LSS_DT_BF16 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || stats != nullptr || C2 != 0 || up != 1 ||
        (relu & ~(LSS_W_KS | 1)) != 0)
      return LSS_E_SHAPE;
    // write-back stores here (LSS_KS_WT=1: write-through like the other kernels): 5 MB of output per launch at most,
    // read back by the next launch - measured in a dependent chain 9.47 / 8.22 / 7.48 -> 9.07 / 8.11 / 7.36 us per launch
    const bool wt_ks = getenv("LSS_KS_WT") != nullptr && atoi(getenv("LSS_KS_WT")) != 0;
    return lss_conv_ks_launch(x, w_packed, scale, shift, residual, y, B, H, W, Cx, Cout, relu & 1, wt_ks ? 1 : 0,
                              lss_stream(stream));
  }
  if (relu & LSS_W_RING) {  // ring-packed weights: the loader / consumer kernel, or nothing
    if (dt != LSS_DT_BF16 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || residual != nullptr || stats != nullptr ||
        (relu & ~(LSS_W_RING | 1)) != 0)
      return LSS_E_SHAPE;
    const bool wt_ring = (getenv("LSS_CONV_WT") == nullptr || atoi(getenv("LSS_CONV_WT")) != 0);
    return lss_conv_ring_launch(x, x2, w_packed, scale, shift, y, nullptr, nullptr, nullptr, 0, B, H, W, Cx, C2, up, Cout,
                                relu & 1, wt_ring ? 1 : 0, lss_stream(stream));
  }
  const int kblock = dt == LSS_DT_BF16 ? 64 : 8;
  // K blocks never straddle the x2 | upsample(x) boundary
  if (Cx % kblock != 0 || C2 % kblock != 0) return LSS_E_SHAPE;
  ConvArgs a;
  a.stamps = conv_stamps_from_env();
  a.src_lds = 0;
  a.x = x; a.x2 = x2; a.w = w_packed; a.scale = scale; a.shift = shift; a.residual = residual;
  a.y = y; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cx = Cx; a.C2 = C2; a.up = up;
  a.Hin = H * up; a.Win = W * up; a.Cin = Cx + C2;
  a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.Ho = (a.Hin + 2 * pad - KH) / stride + 1;
  a.Wo = (a.Win + 2 * pad - KW) / stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return LSS_E_SHAPE;
  const long long M = (long long)B * a.Ho * a.Wo;
  if (M >= (1LL << 31)) return LSS_E_SHAPE;
  a.M = (int)M;
  a.relu = relu & 3;
  a.out_f32 = (relu & LSS_OUT_F32) != 0;
  a.wt = (getenv("LSS_CONV_WT") == nullptr || atoi(getenv("LSS_CONV_WT")) != 0) &&
         (unsigned long long)a.M * a.Cout * 2 < (1ULL << 31);
  a.y2 = nullptr; a.split = 0; a.relu_n = a.Cout;
  const bool head_major = (relu & LSS_OUT_HEAD_MAJOR32) != 0;
  if (a.relu == 3 || (relu & ~(3 | LSS_OUT_F32 | LSS_OUT_HEAD_MAJOR32)) != 0) return LSS_E_LAYOUT;
  a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_n = 0;
  a.ry = a.Hin > 1 ? (float)(H - 1) / (float)(a.Hin - 1) : 0.f;
  a.rx = a.Win > 1 ? (float)(W - 1) / (float)(a.Win - 1) : 0.f;
  const bool fused = (up > 1) || (C2 > 0);
  dim3 grid(lss_cdiv(M, 128), lss_cdiv(Cout, 64));
  if (grid.y > 65535) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  if (dt == LSS_DT_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && a.Cin % 64 == 0 &&
      getenv("LSS_CONV_DIRECT") == nullptr) {
    if (fused) launch_conv_lds<1, 3, 3, 1>(a, st);
    else launch_conv_lds<0, 3, 3, 1>(a, st);
    return lss_launch_status();
  }
  // 2x2 / pad 1 and 4x4 / pad 2, stride 1: the data gradients of the 3x3 / 2 and 7x7 / 2 convs over phase planes
  // (lss_conv2d_pack_weights_s2_dgrad) - same tile kernel, other tap counts
  if (dt == LSS_DT_BF16 && stride == 1 && !fused && KH == KW && a.Cin % 64 == 0 && a.stats == nullptr &&
      ((KH == 2 && pad == 1) || (KH == 4 && pad == 2)) && getenv("LSS_CONV_DIRECT") == nullptr) {
    if (KH == 2) launch_conv_lds<0, 2, 2, 1>(a, st);
    else launch_conv_lds<0, 4, 4, 2>(a, st);
    return lss_launch_status();
  }
  // 1x1 / stride 1: the token-major linear layers of the BEV transformer and the 1x1 convs
  // a 1x1 / stride-1 conv over NHWC is a row-major GEMM over the B*H*W pixel rows
  // (linear_mfma.hip): the token-major linear layers of the BEV transformer and the 1x1
  // convs around it.  Narrow outputs (the 4-class heads: a 128-wide tile would be 97 % padding)
  // and the BN-statistics variant stay on the direct kernel below.
  if (dt == LSS_DT_BF16 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && !fused && a.Cin % 32 == 0 &&
      a.Cout >= 64 && a.stats == nullptr && getenv("LSS_CONV_DIRECT") == nullptr)
    return lss_linear_bf16_launch(x, w_packed, scale, shift, residual, y, M, Cout, a.Cin, a.relu, a.out_f32,
                                  head_major ? a.Ho * a.Wo : 0, st);
  if (head_major) return LSS_E_LAYOUT;  // only the GEMM kernel writes that layout
  if (dt == LSS_DT_BF16) {
    if (fused) hipLaunchKernelGGL((conv_direct_kernel<unsigned short, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_direct_kernel<unsigned short, false>), grid, dim3(256), 0, st, a);
  } else {
    if (fused) hipLaunchKernelGGL((conv_direct_kernel<float, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_direct_kernel<float, false>), grid, dim3(256), 0, st, a);
  }
  return lss_launch_status();
}

// Stride-2 k x k conv (k = 3 pad 1, or k = 7 pad 3) on the LDS-tiled kernel through
// the phase-plane (space-to-depth) form; `w_s2d` from lss_conv2d_pack_weights_s2d.
static int conv2d_s2_impl(const void* x, const void* w_s2d, const float* scale, const float* shift,
                          const void* residual, void* y, float* stats, int B, int H, int W, int Cx, int Cout,
                          int K, int pad, int relu, void* y2, int split, int relu_n, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(w_s2d); LSS_CHECK_PTR(y);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cx); LSS_CHECK_POS(Cout);
  if (!((K == 3 && pad == 1) || (K == 7 && pad == 3) || (K == 1 && pad == 0))) return LSS_E_SHAPE;
  if (Cx % 64 != 0) return LSS_E_SHAPE;
  ConvArgs a;
  a.stamps = conv_stamps_from_env();
  a.src_lds = 0;
  a.x = x; a.x2 = nullptr; a.w = w_s2d; a.scale = scale; a.shift = shift; a.residual = residual;
  a.y = y; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cx = Cx; a.C2 = 0; a.up = 1;
  // K = 1: only parity phase (0,0) is ever read, so the K loop runs over that phase alone
  // and the weights are the plain [1][Cout][Cx] pack
  a.Hin = H; a.Win = W; a.Cin = K == 1 ? Cx : 4 * Cx;
  a.Cout = Cout; a.KH = K; a.KW = K; a.stride = 2; a.pad = pad;
  a.Ho = (H + 2 * pad - K) / 2 + 1;
  a.Wo = (W + 2 * pad - K) / 2 + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return LSS_E_SHAPE;
  const long long M = (long long)B * a.Ho * a.Wo;
  if (M >= (1LL << 31)) return LSS_E_SHAPE;
  a.M = (int)M;
  a.relu = relu & 3;
  a.out_f32 = 0;
  a.wt = (getenv("LSS_CONV_WT") == nullptr || atoi(getenv("LSS_CONV_WT")) != 0) &&
         (unsigned long long)a.M * a.Cout * 2 < (1ULL << 31);
  a.y2 = y2; a.split = split; a.relu_n = y2 ? relu_n : a.Cout;
  a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_n = 0;
  a.ry = a.rx = 0.f;
  hipStream_t st = lss_stream(stream);
  if (K == 7) launch_conv_lds<2, 4, 4, 2>(a, st);
  else if (K == 3) launch_conv_lds<2, 2, 2, 1>(a, st);
  else launch_conv_lds<2, 1, 1, 0>(a, st);
  return lss_launch_status();
}

extern "C" int lss_conv2d_s2_fwd(const void* x, const void* w_s2d, const float* scale,
                                 const float* shift, const void* residual, void* y, float* stats,
                                 int B, int H, int W, int Cx, int Cout, int K, int pad, int relu,
                                 void* stream) {
  return conv2d_s2_impl(x, w_s2d, scale, shift, residual, y, stats, B, H, W, Cx, Cout, K, pad, relu, nullptr, 0, 0,
                        stream);
}

// Two stride-2 convs over the same input in ONE launch: output channels [0, split) -> y (with the
// activation), [split, Cout) -> y2 (no activation).  A BasicBlock's 3x3/2 conv1 and its 1x1/2
// downsample (embedded at the centre tap of a 3x3 frame so both share the phase-plane weight
// layout; ref: torchvision BasicBlock.forward, `out = relu(bn1(conv1(x)))`, `identity =
// downsample(x)`): the downsample's own launch (8.5 us) and its kernel boundary disappear.
extern "C" int lss_conv2d_s2_dual_fwd(const void* x, const void* w_s2d, const float* scale, const float* shift,
                                      void* y, void* y2, int B, int H, int W, int Cx, int Cout, int split, int K,
                                      int pad, int relu, void* stream) {
  LSS_CHECK_PTR(y2);
  if (split <= 0 || split >= Cout || split % 128 != 0 || (Cout - split) % 8 != 0) return LSS_E_SHAPE;
  return conv2d_s2_impl(x, w_s2d, scale, shift, nullptr, y, nullptr, B, H, W, Cx, Cout, K, pad, relu, y2, split,
                        split, stream);
}

// 3x3 / stride 1 / pad 1 conv (optionally with the fused upsample / concat gather)
// + scale/shift + ReLU + a fused 1x1 head, output NCHW fp32 (B, head_n, Ho, Wo).
// BevEncode's last two layers (ref: src/modules.py:110-116, up2[0..4]) in one launch:
// the 128-channel 200x200 activation and the NHWC->NCHW pass never touch HBM.
extern "C" int lss_conv2d_head_fwd(const void* x, const void* x2, const void* w_packed,
                                   const float* scale, const float* shift, const float* head_w,
                                   const float* head_b, float* out, int B, int H, int W, int Cx,
                                   int C2, int up, int Cout, int head_n, int relu, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(w_packed); LSS_CHECK_PTR(head_w); LSS_CHECK_PTR(head_b);
  LSS_CHECK_PTR(out);
  LSS_CHECK_POS(B); LSS_CHECK_POS(H); LSS_CHECK_POS(W); LSS_CHECK_POS(Cx); LSS_CHECK_POS(up);
  LSS_CHECK_POS(head_n);
  if (relu & LSS_W_RING) {
    if ((relu & ~(LSS_W_RING | 1)) != 0) return LSS_E_SHAPE;
    if (C2 > 0 && x2 == nullptr) return LSS_E_NULL;
    return lss_conv_ring_launch(x, x2, w_packed, scale, shift, nullptr, head_w, head_b, out, head_n, B, H, W, Cx, C2, up,
                                Cout, relu & 1, 0, lss_stream(stream));
  }
  if ((Cout != 128 && Cout != 64) || C2 < 0 || Cx % 64 != 0 || C2 % 64 != 0 || head_n > 64) return LSS_E_SHAPE;
  if (Cout == 64 && (up != 1 || C2 != 0)) return LSS_E_SHAPE;  // the 64-wide tile has no fused-gather form
  if (C2 > 0 && x2 == nullptr) return LSS_E_NULL;
  ConvArgs a;
  a.stamps = conv_stamps_from_env();
  a.src_lds = 0;
  a.x = x; a.x2 = x2; a.w = w_packed; a.scale = scale; a.shift = shift; a.residual = nullptr;
  a.y = nullptr; a.stats = nullptr;
  a.B = B; a.H = H; a.W = W; a.Cx = Cx; a.C2 = C2; a.up = up;
  a.Hin = H * up; a.Win = W * up; a.Cin = Cx + C2;
  a.Cout = Cout; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1;
  a.Ho = a.Hin; a.Wo = a.Win;
  const long long M = (long long)B * a.Ho * a.Wo;
  if (M >= (1LL << 31)) return LSS_E_SHAPE;
  a.M = (int)M;
  a.relu = relu & 3;
  a.out_f32 = 0;
  a.wt = (getenv("LSS_CONV_WT") == nullptr || atoi(getenv("LSS_CONV_WT")) != 0) &&
         (unsigned long long)a.M * a.Cout * 2 < (1ULL << 31);
  a.y2 = nullptr; a.split = 0; a.relu_n = a.Cout;
  a.head_w = head_w; a.head_b = head_b; a.head_out = out; a.head_n = head_n;
  a.ry = a.Hin > 1 ? (float)(H - 1) / (float)(a.Hin - 1) : 0.f;
  a.rx = a.Win > 1 ? (float)(W - 1) / (float)(a.Win - 1) : 0.f;
  const bool fused = (up > 1) || (C2 > 0);
  const int tilesX = lss_cdiv(a.Wo, 16), tilesY = lss_cdiv(a.Ho, Cout == 64 ? 16 : 8);
  dim3 g(tilesX * tilesY * B, 1);
  hipStream_t st = lss_stream(stream);
  // the fused-gather form takes the 32-channel steps (three workgroups per CU), as in launch_conv_lds
  bool kc32 = fused && a.Cx % 32 == 0 && a.C2 % 32 == 0;
  if (const char* e = getenv("LSS_CONV_KC")) kc32 = kc32 && atoi(e) == 32;
  if (fused && kc32) {
    a.src_lds = conv_src_lds_ok(a);
    hipLaunchKernelGGL((conv_lds_kernel<2, 128, 1, 3, 3, 1, 32, 1, true>), g, dim3(256), 0, st, a, tilesX, tilesY);
  }
  else if (fused)
    hipLaunchKernelGGL((conv_lds_kernel<2, 128, 1, 3, 3, 1, 64, 1, true>), g, dim3(256), 0, st, a, tilesX, tilesY);
  else if (Cout == 64)
    hipLaunchKernelGGL((conv_lds_kernel<2, 64, 0, 3, 3, 1, 64, 1, true>), g, dim3(256), 0, st, a, tilesX, tilesY);
  else
    hipLaunchKernelGGL((conv_lds_kernel<2, 128, 0, 3, 3, 1, 64, 1, true>), g, dim3(256), 0, st, a, tilesX, tilesY);
  return lss_launch_status();
}

extern "C" int lss_conv2d_sequence(const lss_conv_launch_t* L, int n, void* stream) {
  LSS_CHECK_PTR(L);
  if (n < 0) return LSS_E_SHAPE;
  for (int i = 0; i < n; ++i) {
    const lss_conv_launch_t& c = L[i];
    int rc;
    if (c.kind == 0)
      rc = lss_conv2d_fwd(c.x, c.x2, c.w, c.scale, c.shift, c.residual, c.y, c.stats, c.B, c.H, c.W, c.Cx,
                          c.C2, c.up, c.Cout, c.KH, c.KW, c.stride, c.pad, c.relu, c.dt, stream);
    else if (c.kind == 1)
      rc = lss_conv2d_s2_fwd(c.x, c.w, c.scale, c.shift, c.residual, c.y, c.stats, c.B, c.H, c.W, c.Cx,
                             c.Cout, c.KH, c.pad, c.relu, stream);
    else if (c.kind == 2)
      rc = lss_conv2d_head_fwd(c.x, c.x2, c.w, c.scale, c.shift, c.head_w, c.head_b, c.head_out, c.B, c.H,
                               c.W, c.Cx, c.C2, c.up, c.Cout, c.head_n, c.relu, stream);
    else if (c.kind == 3)
      rc = lss_conv2d_s2_dual_fwd(c.x, c.w, c.scale, c.shift, c.y, c.y2, c.B, c.H, c.W, c.Cx, c.Cout, c.split, c.KH,
                                  c.pad, c.relu, stream);
    else
      rc = LSS_E_LAYOUT;
    if (rc != 0) return rc;
  }
  return 0;
}
