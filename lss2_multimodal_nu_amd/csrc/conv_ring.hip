// K8r: the big 3x3 / stride-1 convolutions of BevEncode (up1.conv0, up1.conv3, up2 + head: 76 % of its FLOPs;
// ref src/modules.py:9-27, 108-116) as an implicit GEMM with SPECIALISED WAVES: consumer waves that only read LDS
// fragments and issue MFMAs, a weight-loader wave that streams the weight slabs by LDS-DMA, and a patch-loader wave
// that gathers the input patch (plain DMA, or the fused bilinear-upsample / concat of `Up.forward`).  No workgroup
// barrier inside the main loop: producers and consumers hand LDS buffers over through FULL / FREE words in LDS
// (MI355X_MICROARCH.md, row 'ring-gemm'), so the four consumer waves drift against each other instead of meeting
// at a barrier every step, and no MFMA wave ever issues a DMA piece or a blend instruction (round 2 measured the
// weight DMA issue at 20-31 % of these kernels when the MFMA waves carry it).
//
// Geometry.  Output pixels are cut into STRIPS of 4 rows x 20 columns = five 4x4-pixel blocks; a block is the 16
// columns of one v_mfma_f32_16x16x32_bf16 (A = 16 output channels x 32 input channels of the weights, B = 32 input
// channels x 16 pixels of the patch).  100 and 200 are multiples of 20 and of 4, so the 100^2 / 200^2 layers tile
// with NO waste (round 2's 8 x 16 tiles covered 104 x 112: +16.5 % MFMAs at 100^2).  A workgroup = 4 strips x 128
// output channels, ONE per CU (250 workgroups at 100^2 x 256 channels x batch 4: 97.6 % of one round; 500 at
// 200^2 x 128 channels: two rounds), 12 waves = three per SIMD:
//   waves 0-7  consumers: strip w >> 1, channel half w & 1 -> 5 blocks x 4 channel tiles = 20 accumulators of 4
//              registers.  The rows of channel tile t are the channels {16 q + 4 t + i}: lane (q = lane >> 4,
//              n = lane & 15) then ends with 16 CONSECUTIVE channels of pixel n in its 16 accumulator registers of a
//              block, and the epilogue stores straight from registers (no LDS staging pass; round 2: 19 % of
//              up1.conv3) - after two row swaps (v_permlane16_swap / v_permlane32_swap) that give the four q lanes of
//              a pixel consecutive 16-B pieces, so one store instruction covers 64 contiguous bytes per pixel
//   wave  8    weight loader: streams the weight slabs by LDS-DMA
//   waves 9-11 patch loaders: plain DMA of the input patch, or source-window DMA + four-corner blend (fused upsample)
// K loop: chunk = 32 input channels, step = (chunk, tap): 20 MFMAs per consumer against one 8-KiB weight slab.
// LDS (140-154 KiB):
//   ring   NSL (7-8) slots x 8 KiB: the weight slab of one step in FRAGMENT ORDER (piece (cg, t, lane) at
//          ((cg*4+t)*64+lane)*16: a consumer reads its A fragment lane-linearly = conflict-free); the packed weights
//          in global memory have exactly this image, so a slab is 8 fully coalesced 1-KiB DMA pieces
//   patch  2 buffers x 4 strips x (6 rows x 24 positions x 64 B): position-major, the four 16-B channel pieces of a
//          position XOR-swizzled by (row & 1) << 1 - conflict-free ds_read_b128 for every tap shift (checked by
//          enumeration over the hardware's b128 lane groups)
//   src    (fused upsample) the low-res source window of a chunk, 4 strips x 5 x 14 positions
//   headx  (fused head) the partial class sums of the second channel half
//   flags  FULL_W[slot] = fill count (weight loader), FREE_W[slot] = releases (consumers, LDS atomic add),
//          FULL_P / FREE_P the same for the two patch buffers, FULL_S / FREE_S for the source window
// Every flag wait is bounded (ring_prims.h); lss_conv2d_ring_timeouts() must read 0.
#include <stdlib.h>

#include <type_traits>

#include "lss_common.h"

namespace {

constexpr int RK_SW = 20, RK_SH = 4;          // strip: 20 x 4 output pixels
constexpr int RK_PW = 24, RK_PH = 6;          // patch positions of a strip (22 x 6 used)
constexpr int RK_KC = 32;                     // input channels per chunk
constexpr int RK_POSB = RK_KC * 2;            // 64 B per patch position
constexpr int RK_STRIP_PATCH = RK_PH * RK_PW * RK_POSB;   // 9216 B = 9 DMA pieces
constexpr int RK_SLAB = 128 * RK_KC * 2;      // 8192 B = 8 DMA pieces
constexpr int RK_SRC_H = 5, RK_SRC_W = 14;    // low-res source window of a strip patch (scale factors >= 2)
constexpr int RK_SRC_STRIP = RK_SRC_H * RK_SRC_W * RK_POSB;  // 4480 B

struct RingArgs {
  const unsigned short* x;    // (B, H, W, Cx) bf16 NHWC; the low-res source when up > 1
  const unsigned short* x2;   // (B, Hin, Win, C2) bf16 NHWC or null
  const unsigned char* w;     // ring-packed weights (lss_conv2d_pack_weights_ring)
  const float* scale;
  const float* shift;
  unsigned short* y;          // (B, Hin, Win, Cout) bf16 NHWC (no head)
  const float* head_w;        // (head_n, 128) fp32
  const float* head_b;
  float* head_out;            // (B, head_n, Hin, Win) fp32 NCHW
  int head_n;
  int B, H, W, Cx, C2, up, Hin, Win, Cin, Cout, relu, wt;
  float ry, rx;
  int SX, SY, nstrips, nblk, nch;
  // diagnostics (LSS_RING_STATS=<hex device address>): per workgroup 12 waves x 6 u64: {polls spent waiting on the
  // first / second kind of flag, 100-MHz ticks from kernel entry to the wave's end, ticks at entry}
  unsigned long long* stats;
};

__device__ __attribute__((aligned(128))) unsigned char lss_ring_zero_page[128];  // source of out-of-image patch pieces
__device__ int lss_ring_timeouts;  // flag waits that hit their bound (must stay 0; read by lss_conv2d_ring_timeouts)

template <int MODE, bool HEAD, int NSL>
struct RingLds {
  static constexpr int RING = 0;
  static constexpr int PATCH = NSL * RK_SLAB;
  static constexpr int SRC = PATCH + 2 * 4 * RK_STRIP_PATCH;           // two buffers x four strips
  static constexpr int HEADX = SRC + (MODE == 1 ? 4 * RK_SRC_STRIP : 0);
  static constexpr int FLAGS = HEADX + (HEAD ? 4 * 80 * 4 * 4 : 0);
  static constexpr int TOTAL = FLAGS + 256;
};
// flag words (ints at FLAGS)
constexpr int F_FULL_W = 0, F_FREE_W = 8, F_FULL_P = 16, F_FREE_P = 18, F_HEAD = 20, F_FULL_S = 24, F_FREE_S = 25;

#define RK_TIMEOUT_COUNTER lss_ring_timeouts
#include "ring_prims.h"

// bilinear blend of four 8-channel bf16 pieces in the four-weight form: r = q00 w00, then fma(q01, w01, r), fma(q10,
// w10, r), fma(q11, w11, r) per channel - the operation order of conv_mfma.hip's blend_bf16x8, so the fused-gather
// layers see the same operand bits on either kernel.  Written with SCALAR v_fma_f32 (the file is compiled with
// -fno-slp-vectorize): packed f32 VALU issued next to MFMAs costs ~20 cycles more per v_pk_fma_f32 than the two plain
// FMAs it replaces (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'), and the blend runs on SIMDs whose other
// two waves issue MFMAs back to back; one v_cvt_pk_bf16_f32 per channel pair.
typedef __attribute__((ext_vector_type(2))) float rk_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 rk_bf16x2;
__device__ __forceinline__ unsigned int rk_pack_bf2(float lo, float hi) {
  const rk_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, rk_bf16x2));
}
__device__ __forceinline__ uint4 rk_blend(const uint4& q00, const uint4& q01, const uint4& q10, const uint4& q11,
                                          float lx, float ly) {
  const float w11 = lx * ly, w10 = ly - w11, w01 = lx - w11, w00 = 1.f - lx - ly + w11;
  const unsigned int* u00 = reinterpret_cast<const unsigned int*>(&q00);
  const unsigned int* u01 = reinterpret_cast<const unsigned int*>(&q01);
  const unsigned int* u10 = reinterpret_cast<const unsigned int*>(&q10);
  const unsigned int* u11 = reinterpret_cast<const unsigned int*>(&q11);
  uint4 out;
  unsigned int* o = reinterpret_cast<unsigned int*>(&out);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __builtin_bit_cast(float, u00[i] << 16) * w00;
    float hi = __builtin_bit_cast(float, u00[i] & 0xffff0000u) * w00;
    lo = __builtin_fmaf(__builtin_bit_cast(float, u01[i] << 16), w01, lo);
    hi = __builtin_fmaf(__builtin_bit_cast(float, u01[i] & 0xffff0000u), w01, hi);
    lo = __builtin_fmaf(__builtin_bit_cast(float, u10[i] << 16), w10, lo);
    hi = __builtin_fmaf(__builtin_bit_cast(float, u10[i] & 0xffff0000u), w10, hi);
    lo = __builtin_fmaf(__builtin_bit_cast(float, u11[i] << 16), w11, lo);
    hi = __builtin_fmaf(__builtin_bit_cast(float, u11[i] & 0xffff0000u), w11, hi);
    o[i] = rk_pack_bf2(lo, hi);
  }
  return out;
}

// Wave priorities (s_setprio; -DRK_PRIO_x=n for A/B builds).  Measured on MI355X, three big convs, same box:
// consumers 0 / patch loaders 3 / weight loader 2: 211.0 us; 2 / 0 / 3: 220.5; 0 / 0 / 0: 220.6; 1 / 0 / 3: 220.1 -
// the blend has to be served promptly (the consumers wait for the patch at every chunk boundary), the MFMAs find
// their slots anyway.
#ifndef RK_PSLEEP
#define RK_PSLEEP 8  // s_sleep argument of the patch loaders' polls for the consumers' release (x 64 cycles)
#endif
#ifndef RK_PRIO_C
#define RK_PRIO_C 0
#endif
#ifndef RK_PRIO_P
#define RK_PRIO_P 3
#endif
#ifndef RK_PRIO_W
#define RK_PRIO_W 2
#endif
constexpr int RK_NS = 4;              // strips per workgroup
constexpr int RK_NCONS = 2 * RK_NS;   // consumer waves: (strip, 64-channel half)
constexpr int RK_NPATCH = 3;          // patch-loader waves
constexpr int RK_NWAVES = RK_NCONS + 1 + RK_NPATCH;  // 12 = three per SIMD: ONE workgroup per CU
constexpr int RK_NPIECE = RK_NS * 9;  // 1-KiB pieces of one patch buffer (all strips)

struct RingTile {
  int nb;            // 128-channel block
  int b[RK_NS], oy0[RK_NS], ox0[RK_NS];
  bool ok[RK_NS];
};

// MODE 0: plain NHWC input.  MODE 1: input = cat([x2, bilinear_upsample_align_corners(x, up)]) (ref Up.forward).
//
// One workgroup per CU: 12 waves = 8 consumers (4 strips x 2 channel halves), 1 weight loader, 3 patch loaders, i.e.
// three waves on every SIMD.  (The first version ran two 6-wave workgroups per CU - on paper: the wait statistics
// showed half of the workgroups STARTING when the other half had finished.  Six waves land on the four SIMDs as
// 2, 2, 1, 1 and a second workgroup only fits next to that when its own 2, 2, 1, 1 happens to fall the other way
// round.)  A weight slab now feeds four strips instead of two: half the weight bytes streamed into LDS per MFMA.
template <int MODE, bool HEAD, int NSL>
__global__ __launch_bounds__(RK_NWAVES * 64, 3) void conv_ring_kernel(RingArgs a) {
  using L = RingLds<MODE, HEAD, NSL>;
  constexpr int LA = NSL - 2;  // weight slabs in flight at most: one slot is being consumed, one is published and waiting
  static_assert(NSL >= 3 && NSL <= 8, "ring depth");
  static_assert(L::TOTAL <= 160 * 1024, "LDS of one CU");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[L::TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const rk_flag_t flags = (rk_flag_t)((__attribute__((address_space(3))) unsigned char*)smem + L::FLAGS);
  if (tid < 64) flags[tid] = 0;
  __syncthreads();
  // wait statistics: a diagnostic build only (-DRK_STATS, tools/ring_stats.py); the shipped kernel carries none of it
  int st0 = 0, st1 = 0;  // polls spent in the two kinds of waits of this wave's role (dead code without RK_STATS)
#ifdef RK_STATS
#define RK_ST(x) x
  const unsigned long long t_entry = a.stats ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto write_stats = [&]() {
    if (a.stats != nullptr && lane == 0) {
      unsigned long long* o = a.stats + ((size_t)blockIdx.x * RK_NWAVES + wave) * 6;
      o[0] = (unsigned long long)st0; o[1] = (unsigned long long)st1;
      o[2] = __builtin_amdgcn_s_memrealtime() - t_entry; o[3] = t_entry;
    }
  };
#else
#define RK_ST(x)
  auto write_stats = [&]() {};
#endif

  // ---- which tile: XCD-aware order (blocks b and b + 8 share an XCD: give each XCD a contiguous range) ----
  RingTile T;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    T.nb = t % a.nblk;
    const int quad = t / a.nblk;
#pragma unroll
    for (int s = 0; s < RK_NS; ++s) {
      int sid = RK_NS * quad + s;
      T.ok[s] = sid < a.nstrips;
      sid = min(sid, a.nstrips - 1);
      const int per = a.SX * a.SY;
      T.b[s] = sid / per;
      const int rem = sid - T.b[s] * per;
      const int sy = rem / a.SX;
      T.oy0[s] = sy * RK_SH;
      T.ox0[s] = (rem - sy * a.SX) * RK_SW;
    }
  }
  const int nsteps = a.nch * 9;

  if (wave == RK_NCONS) {
    // =============================== weight loader ===============================
    __builtin_amdgcn_s_setprio(RK_PRIO_W);
    const unsigned char* wsrc = a.w + (size_t)T.nb * a.nch * 9 * RK_SLAB + lane * 16;
    // Policy: issue a slab whenever its slot is free (at most LA + 1 in flight); while the next slot is NOT free, wait
    // for the OLDEST slab in flight only (a counted vmcnt: the wave blocks for what is left of that slab's latency,
    // not for the whole queue), publish it, look again.  History: build 1 published slab i - LA when it issued slab i -
    // the consumers got slab s + 5 only after releasing slab s + 3, half a step before they needed it (603 TF on
    // up1.conv3); build 2 drained the whole queue (vmcnt(0)) before blocking on FREE.
    int slot = 0, gen = 1, pslot = 0, pgen = 1, inflight = 0;  // inflight = issued, not yet published
    auto publish_oldest = [&]() {
      switch (inflight) {  // all but the inflight - 1 youngest slabs have landed
        case 1: rk_wait_vmcnt<0>(); break;
        case 2: rk_wait_vmcnt<8>(); break;
        case 3: rk_wait_vmcnt<16>(); break;
        case 4: rk_wait_vmcnt<24>(); break;
        case 5: rk_wait_vmcnt<32>(); break;
        case 6: rk_wait_vmcnt<40>(); break;
        default: rk_wait_vmcnt<48>(); break;
      }
      rk_set(flags + F_FULL_W + pslot, pgen, lane);
      if (++pslot == NSL) { pslot = 0; ++pgen; }
      --inflight;
    };
    for (int i = 0; i < nsteps; ++i) {
      if (gen > 1) {
        int spins = 0;
        while (rk_peek(flags + F_FREE_W + slot) < RK_NCONS * (gen - 1)) {
          if (inflight > 0) {
            publish_oldest();
          } else {
            __builtin_amdgcn_s_sleep(1);
            ++st0;
            if (++spins > RK_SPIN_LIMIT) {
              if (lane == 0) atomicAdd(&lss_ring_timeouts, 1);
              break;
            }
          }
        }
        asm volatile("" ::: "memory");
      }
      unsigned char* dst = smem + L::RING + slot * RK_SLAB;
#ifndef RK_DIAG_NOWDMA  // timing-only build: no weight traffic, the hand-off protocol alone
#pragma unroll
      for (int k = 0; k < 8; ++k) rk_glds16(wsrc + (size_t)i * RK_SLAB + k * 1024, dst + k * 1024);
#endif
      ++inflight;
      if (inflight > LA) publish_oldest();
      if (++slot == NSL) { slot = 0; ++gen; }
    }
    while (inflight > 0) publish_oldest();
    rk_wait_vmcnt<0>();  // (see the patch loaders' exit)
    write_stats();
    return;
  }

  if (wave > RK_NCONS) {
    // =============================== patch loaders ===============================
    // Patch loader p takes the 1-KiB pieces it = p, p + 3, ... of a patch buffer (12 of 36).  Piece it = (strip it / 9,
    // block it % 9): lane -> patch position 16 * (it % 9) + (lane >> 2), 16-B slot lane & 3, which holds channel piece
    // slot ^ ((row & 1) << 1).  FULL_P / FREE_S etc. are counters: every patch loader adds its share.
    __builtin_amdgcn_s_setprio(RK_PRIO_P);
    const int pw = wave - RK_NCONS - 1;
    constexpr int NIT = RK_NPIECE / RK_NPATCH;  // 12
    const int nskip = MODE == 1 ? a.C2 / RK_KC : a.nch;  // chunks copied straight from a full-resolution tensor
    const unsigned short* xfull = MODE == 1 ? a.x2 : a.x;
    const int cfull = MODE == 1 ? a.C2 : a.Cx;
    int poff[NIT];  // element offset of the piece's pixel in the full-resolution tensor (+ channel piece), or -1
    int g_off[MODE == 1 ? NIT : 1];
    unsigned int g_w[MODE == 1 ? NIT : 1];
    constexpr int NSRC = MODE == 1 ? (RK_NS * 5 + RK_NPATCH - 1) / RK_NPATCH : 1;  // source-window pieces of this wave (7)
    int src_o[NSRC];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = pw + k * RK_NPATCH;
      const int s = it / 9, pos = (it % 9) * 16 + (lane >> 2);
      const int prow = pos / RK_PW, pcol = pos - prow * RK_PW;
      const int piece = (lane & 3) ^ ((prow & 1) << 1);
      const int sb = s == 0 ? T.b[0] : s == 1 ? T.b[1] : s == 2 ? T.b[2] : T.b[3];
      const int soy = s == 0 ? T.oy0[0] : s == 1 ? T.oy0[1] : s == 2 ? T.oy0[2] : T.oy0[3];
      const int sox = s == 0 ? T.ox0[0] : s == 1 ? T.ox0[1] : s == 2 ? T.ox0[2] : T.ox0[3];
      const bool sok = s == 0 ? T.ok[0] : s == 1 ? T.ok[1] : s == 2 ? T.ok[2] : T.ok[3];
      const int iy = soy - 1 + prow, ix = sox - 1 + pcol;
      const bool in = sok && pcol < RK_SW + 2 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      poff[k] = (in && cfull > 0) ? ((sb * a.Hin + iy) * a.Win + ix) * cfull + piece * 8 : -1;
      if (MODE == 1) {
        g_off[k] = 0;
        g_w[k] = 0;
        if (in) {
          const int wy0 = (int)(a.ry * (float)max(soy - 1, 0)), wx0 = (int)(a.rx * (float)max(sox - 1, 0));
          const float sy = a.ry * (float)iy, sx = a.rx * (float)ix;
          const int y0 = (int)sy, x0 = (int)sx;
          const unsigned int wy = (unsigned int)((sy - (float)y0) * 65536.f + 0.5f);
          const unsigned int wx = (unsigned int)((sx - (float)x0) * 65536.f + 0.5f);
          g_w[k] = min(wx, 65535u) | (min(wy, 65535u) << 16);
          const int fl = (x0 < a.W - 1 ? 1 : 0) | (y0 < a.H - 1 ? 2 : 0) | 4;
          g_off[k] = ((((y0 - wy0) * RK_SRC_W + (x0 - wx0)) * 4 + piece) * 16) | fl;
        }
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        const int it = pw + k * RK_NPATCH;  // source piece it = (strip it / 5, block it % 5); it >= 20: none
        const int s = min(it / 5, RK_NS - 1), q = (it % 5) * 64 + lane;
        const int pos = min(q >> 2, RK_SRC_H * RK_SRC_W - 1), part = q & 3;
        const int sr = pos / RK_SRC_W, sc = pos - sr * RK_SRC_W;
        const int sb = s == 0 ? T.b[0] : s == 1 ? T.b[1] : s == 2 ? T.b[2] : T.b[3];
        const int soy = s == 0 ? T.oy0[0] : s == 1 ? T.oy0[1] : s == 2 ? T.oy0[2] : T.oy0[3];
        const int sox = s == 0 ? T.ox0[0] : s == 1 ? T.ox0[1] : s == 2 ? T.ox0[2] : T.ox0[3];
        const int wy0 = (int)(a.ry * (float)max(soy - 1, 0)), wx0 = (int)(a.rx * (float)max(sox - 1, 0));
        src_o[k] = ((sb * a.H + min(wy0 + sr, a.H - 1)) * a.W + min(wx0 + sc, a.W - 1)) * a.Cx + part * 8;
      }
    }
    // this wave's pieces of full-resolution chunk c -> patch buffer `buf` by DMA
    auto issue_full = [&](int c, int buf) {
      unsigned char* dst = smem + L::PATCH + buf * (RK_NS * RK_STRIP_PATCH);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const void* src = poff[k] >= 0 ? (const void*)(xfull + poff[k] + c * RK_KC) : (const void*)lss_ring_zero_page;
        rk_glds16(src, dst + (pw + k * RK_NPATCH) * 1024);
      }
    };
    if (MODE == 0) {
      // Chunk c + 1 is published BEFORE this wave blocks on the consumers' release of chunk c: they ask for chunk
      // c + 1 in the middle of chunk c's last step and release chunk c at its end (the first build waited for the
      // release first - a circular wait that only the spin bound resolved).
      issue_full(0, 0);
      if (a.nch > 1) {
        issue_full(1, 1);
        rk_wait_vmcnt<NIT>();
      } else {
        rk_wait_vmcnt<0>();
      }
      rk_add1(flags + F_FULL_P + 0, lane);
      for (int c = 0; c < a.nch; ++c) {
        if (c + 1 < a.nch) {
          rk_wait_vmcnt<0>();
          rk_add1(flags + F_FULL_P + ((c + 1) & 1), lane);
        }
        if (c + 2 < a.nch) {
          // (polled every 512 cycles: the wait is a chunk long and the loaders' polls take issue slots from the consumers)
          st0 += rk_wait_ge<RK_PSLEEP>(flags + F_FREE_P + (c & 1), RK_NCONS * ((c >> 1) + 1));  // the consumers are done with chunk c
          issue_full(c + 2, c & 1);
        }
      }
    } else {
      // this wave's pieces of the source window of upsampled chunk c -> src (the last piece of a strip is partial)
      auto issue_src = [&](int c) {
        const int cx = c * RK_KC - a.C2;
#pragma unroll
        for (int k = 0; k < NSRC; ++k) {
          const int it = pw + k * RK_NPATCH;
          if (it < RK_NS * 5 && (it % 5) * 64 + lane < RK_SRC_H * RK_SRC_W * 4)
            rk_glds16(a.x + src_o[k] + cx, smem + L::SRC + (it / 5) * RK_SRC_STRIP + (it % 5) * 1024);
        }
      };
      // start the DMA chunk c needs.  Full-resolution chunk: straight into patch buffer c & 1 once the consumers have
      // released it; upsampled chunk: the source window, once ALL patch loaders have finished blending from it.
      auto prepare = [&](int c) {
        if (c < nskip) {
          st0 += rk_wait_ge<RK_PSLEEP>(flags + F_FREE_P + (c & 1), RK_NCONS * (c >> 1));
          issue_full(c, c & 1);
        } else {
          st1 += rk_wait_ge(flags + F_FREE_S, RK_NPATCH * (c - nskip));
          issue_src(c);
        }
      };
      prepare(0);
      for (int c = 0; c < a.nch; ++c) {
        rk_wait_vmcnt<0>();
        if (c >= nskip) {
          // every loader's share of the source window has landed / the consumers have released the patch buffer
          rk_add1(flags + F_FULL_S, lane);
          st1 += rk_wait_ge(flags + F_FULL_S, RK_NPATCH * (c - nskip + 1));
          st0 += rk_wait_ge<RK_PSLEEP>(flags + F_FREE_P + (c & 1), RK_NCONS * (c >> 1));
          unsigned char* dst = smem + L::PATCH + (c & 1) * (RK_NS * RK_STRIP_PATCH);
          // fully unrolled (static register indices for g_off / g_w), no branches (a piece outside the image blends
          // whatever sits at offset 0 of the window and is replaced by zeros afterwards), and the four corner reads of
          // piece k + 1 are requested before piece k is blended
          uint4 qa[4], qb[4];
          auto corners = [&](int k, uint4* q) {
            const int it = pw + k * RK_NPATCH;
            const unsigned char* p00 = smem + L::SRC + (it / 9) * RK_SRC_STRIP + (g_off[k] & 0xfff0);
            const int dx = (g_off[k] & 1) ? 4 * 16 : 0, dy = (g_off[k] & 2) ? RK_SRC_W * 4 * 16 : 0;
            q[0] = *reinterpret_cast<const uint4*>(p00);
#if defined(RK_DIAG_ONEREAD)  // timing-only: one corner read, the full arithmetic
            q[1] = q[0]; q[2] = q[0]; q[3] = q[0];
            asm volatile("" : "+v"(q[1].x), "+v"(q[2].x), "+v"(q[3].x));
#else
            q[1] = *reinterpret_cast<const uint4*>(p00 + dx);
            q[2] = *reinterpret_cast<const uint4*>(p00 + dy);
            q[3] = *reinterpret_cast<const uint4*>(p00 + dy + dx);
#endif
          };
          auto finish = [&](int k, const uint4* q) {
            uint4 v = make_uint4(0, 0, 0, 0);
#ifndef RK_DIAG_NOBLEND  // timing-only build: the upsampled patch stays zero
#if defined(RK_DIAG_READSONLY)  // timing-only: the four corner reads, no arithmetic
            v.x = q[0].x ^ q[1].x ^ q[2].x ^ q[3].x; v.y = q[0].y ^ q[1].y ^ q[2].y ^ q[3].y;
            v.z = q[0].z ^ q[1].z ^ q[2].z ^ q[3].z; v.w = q[0].w ^ q[1].w ^ q[2].w ^ q[3].w;
#else
            v = rk_blend(q[0], q[1], q[2], q[3], (float)(g_w[k] & 0xffff) * (1.f / 65536.f),
                         (float)(g_w[k] >> 16) * (1.f / 65536.f));
#endif
            const bool in = (g_off[k] & 4) != 0;
            v.x = in ? v.x : 0u; v.y = in ? v.y : 0u; v.z = in ? v.z : 0u; v.w = in ? v.w : 0u;
#endif
            *reinterpret_cast<uint4*>(dst + (pw + k * RK_NPATCH) * 1024 + lane * 16) = v;
          };
          corners(0, qa);
#pragma unroll
          for (int k = 0; k < NIT; k += 2) {
            corners(k + 1, qb);
            finish(k, qa);
            if (k + 2 < NIT) corners(k + 2, qa);
            finish(k + 1, qb);
          }
          rk_wait_lgkm0();  // the blended pieces are in LDS, and this wave's reads of the source window are done
          rk_add1(flags + F_FREE_S, lane);
        }
        if (c + 1 < a.nch && c + 1 >= nskip) prepare(c + 1);  // the next source window, as soon as this one is free
        rk_add1(flags + F_FULL_P + (c & 1), lane);
        if (c + 1 < a.nch && c + 1 < nskip) prepare(c + 1);
      }
    }
    // A loader wave must never reach s_endpgm with LDS-DMA in flight: once the workgroup's last wave is gone its LDS is
    // handed to the next workgroup while the transfer still lands.  Every path above already ends behind a vmcnt(0)
    // (each chunk's pieces are waited for before they are published); this one is free and keeps that true for any
    // timing-only build that compiles waits out (the NOWAITW fault of round 3).
    rk_wait_vmcnt<0>();
    write_stats();
    return;
  }

  // ================================== consumers ==================================
  __builtin_amdgcn_s_setprio(RK_PRIO_C);
  const int sidx = wave >> 1, cg = wave & 1;
  const int n = lane & 15, kq = lane >> 4;
  // this wave's strip (selects, not indexing: a runtime-indexed struct member would live in scratch)
  const int my_b = sidx == 0 ? T.b[0] : sidx == 1 ? T.b[1] : sidx == 2 ? T.b[2] : T.b[3];
  const int my_oy0 = sidx == 0 ? T.oy0[0] : sidx == 1 ? T.oy0[1] : sidx == 2 ? T.oy0[2] : T.oy0[3];
  const int my_ox0 = sidx == 0 ? T.ox0[0] : sidx == 1 ? T.ox0[1] : sidx == 2 ? T.ox0[2] : T.ox0[3];
  const bool my_ok = sidx == 0 ? T.ok[0] : sidx == 1 ? T.ok[1] : sidx == 2 ? T.ok[2] : T.ok[3];
  // LDS byte offsets of this lane's fragments.  Pixel fragment of block j at tap (ky, kx): position (n >> 2) + ky,
  // 4 j + (n & 3) + kx of the strip's patch, 16-B piece kq ^ ((row & 1) << 1).  The swizzle only touches bit 5 of the
  // offset, so tap row 1 is (p ^ 32) + 1536 and tap row 2 is p + 3072: ONE lane-dependent address, recomputed from the
  // lane id where it is needed (a table of per-tap bases hoisted out of the loop was spilled by the first builds).
  f32x4 acc[5][4];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Fragments, all single-buffered (80 accumulator + 16 weight + 20 pixel registers: the 168-register budget of
  // three waves per SIMD, and no buffer parity, so the chunk loop stays rolled).  A step runs in
  // two phases of 10 MFMAs: phase A = channel tiles 0, 1 over the five blocks, phase B = tiles 2, 3.  Weight
  // fragments 2, 3 of THIS slab are requested at the top of the step (used in phase B), fragments 0, 1 of the NEXT
  // slab between the phases (used at the top of the next step), and the pixel fragment of block j for the next step
  // right behind block j's phase-B MFMAs: every request is 10 MFMAs (>= 160 cycles) ahead of its use.
  // The fragment reads and their waits are inline asm: left to itself hipcc put ONE s_waitcnt lgkmcnt(0) in front of
  // phase A (every build tried), i.e. each step waited for reads it had issued two instructions earlier.  The LDS
  // queue of a wave is in order and its content per step is fixed -
  //     fw0' fw1' flag' (add) fp0' .. fp4' (add at tap 8) | fw2 fw3 | phase A | fw0" fw1" flag" | phase B ...
  // - so block j of phase A needs `lgkmcnt(6 - j)` (the 4 - j younger pixel fragments and fw2, fw3 may still be in
  // flight) and phase B `lgkmcnt(3)`.  The waits take the fragments they cover as read-write operands, which is what
  // keeps the MFMAs behind them (cdna_hip_programming.md section 5.4 rule 18).
  bf16x8 fw[4], fp[5];
#define RK_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // The two lane-dependent offsets - weight fragment (cg * 4096 + lane * 16 < 8192) and pixel fragment (< 8192) - live
  // packed in ONE register; an address is (field | or + wave-uniform part): 1-2 VALU where the first builds spent ~25
  // per step recomputing them from the lane id.  The few VALU / SALU instructions of a step are placed BETWEEN the
  // MFMA pairs of phase A (they issue in the shadow of the matrix pipe) instead of in one block between the phases,
  // where neither this wave nor - when the two consumer waves of a SIMD run in step - its partner issues MFMAs.
  const unsigned int lanepk =
      (unsigned int)(cg * 4096 + lane * 16) |
      ((unsigned int)((((n >> 2) * RK_PW + (n & 3)) * RK_POSB + (kq << 4)) ^ (((n >> 2) & 1) << 5)) << 16);
  const unsigned int flag_base = (unsigned int)(__UINTPTR_TYPE__)flags;
  const int pconst = L::PATCH + sidx * RK_STRIP_PATCH;
  // FREE counters: lane 0 adds 1 (exec narrowed inside the asm: no branch around it)
  auto add1 = [&](int word) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(save)
                 : "v"(flag_base + 4 * word), "v"(1)
                 : "memory");
  };

  RK_ST(const unsigned long long tc_loop = __builtin_amdgcn_s_memtime(); const unsigned long long tr_loop = __builtin_amdgcn_s_memrealtime();)
  int slot = 0, gen = 1, buf = 0, pgen = 1;
  rk_wait_ge(flags + F_FULL_P + 0, RK_NPATCH);
  rk_wait_ge(flags + F_FULL_W + 0, 1);
  int fnext;  // FULL word of the next step's slot, read one step ahead
  {
    const int wa = (int)(lanepk & 0xffffu) + L::RING, pa = (int)(lanepk >> 16) + pconst;
    RK_DSR(fw[0], wa, 0);
    RK_DSR(fw[1], wa, 1024);
    asm volatile("ds_read_b32 %0, %1" : "=v"(fnext) : "v"(flag_base + 4 * (F_FULL_W + (1 % NSL))));
    RK_DSR(fp[0], pa, 0 * 4 * RK_POSB);
    RK_DSR(fp[1], pa, 1 * 4 * RK_POSB);
    RK_DSR(fp[2], pa, 2 * 4 * RK_POSB);
    RK_DSR(fp[3], pa, 3 * 4 * RK_POSB);
    RK_DSR(fp[4], pa, 4 * 4 * RK_POSB);
  }

  auto step = [&](auto TAPc, bool last) {
    constexpr int TAP = decltype(TAPc)::value;
    constexpr int NT = (TAP + 1) % 9;
    constexpr int POFF = (NT / 3) * (RK_PW * RK_POSB) + (NT % 3) * RK_POSB;  // tap offset of the next step's pixel fragments
    {
      const int wa = (int)(lanepk & 0xffffu) + (L::RING + slot * RK_SLAB);
      RK_DSR(fw[2], wa, 2048);
      RK_DSR(fw[3], wa, 3072);
    }
    __builtin_amdgcn_sched_barrier(0);
    // phase A; the next step's bookkeeping and addresses ride between its MFMA pairs
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fw[0]), "+v"(fw[1]), "+v"(fp[0]));
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fp[0], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fp[0], acc[0][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int nslot = slot + 1, ngen = gen;
    if (nslot == NSL) { nslot = 0; ++ngen; }
    int nbuf = buf, npgen = pgen;
    if (TAP == 8) {
      nbuf = buf ^ 1;
      if (nbuf == 0) ++npgen;
    }
    const int wan = (int)(lanepk & 0xffffu) + (L::RING + nslot * RK_SLAB);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fp[1]));
    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fp[1], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fp[1], acc[1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int pan = (int)((lanepk >> 16) ^ (NT / 3 == 1 ? 32u : 0u)) + (pconst + nbuf * (RK_NS * RK_STRIP_PATCH));
    const int n2 = nslot + 1 == NSL ? 0 : nslot + 1;
    const unsigned int fa = flag_base + 4 * (F_FULL_W + n2);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fp[2]));
    acc[2][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fp[2], acc[2][0], 0, 0, 0);
    acc[2][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fp[2], acc[2][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fp[3]));
    acc[3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fp[3], acc[3][0], 0, 0, 0);
    acc[3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fp[3], acc[3][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fp[4]));
    acc[4][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fp[4], acc[4][0], 0, 0, 0);
    acc[4][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fp[4], acc[4][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      // the next step's slab (and, at the last tap, the next chunk's patch) must have been published; the FULL word
      // was read a step ago, ahead of fragments phase A has waited for
      if (__builtin_amdgcn_readfirstlane(fnext) < ngen) st0 += 1 + rk_wait_ge(flags + F_FULL_W + nslot, ngen);
      if (TAP == 8) st1 += rk_wait_ge(flags + F_FULL_P + nbuf, RK_NPATCH * npgen);
      asm volatile("" ::: "memory");
      RK_DSR(fw[0], wan, 0);
      RK_DSR(fw[1], wan, 1024);
      asm volatile("ds_read_b32 %0, %1" : "=v"(fnext) : "v"(fa));
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fw[2]), "+v"(fw[3]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw[2]), "+v"(fw[3]));
    }
    // phase B: behind block j's MFMAs the pixel fragment of block j is requested for the next step
    acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2], fp[0], acc[0][2], 0, 0, 0);
    acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fp[0], acc[0][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    add1(F_FREE_W + slot);  // the wave has waited for fragments 2, 3: every read of this slab is done
    if (!last) RK_DSR(fp[0], pan, POFF + 0 * 4 * RK_POSB);
    __builtin_amdgcn_sched_barrier(0);
    acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2], fp[1], acc[1][2], 0, 0, 0);
    acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fp[1], acc[1][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) RK_DSR(fp[1], pan, POFF + 1 * 4 * RK_POSB);
    __builtin_amdgcn_sched_barrier(0);
    acc[2][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2], fp[2], acc[2][2], 0, 0, 0);
    acc[2][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fp[2], acc[2][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) RK_DSR(fp[2], pan, POFF + 2 * 4 * RK_POSB);
    __builtin_amdgcn_sched_barrier(0);
    acc[3][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2], fp[3], acc[3][2], 0, 0, 0);
    acc[3][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fp[3], acc[3][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) RK_DSR(fp[3], pan, POFF + 3 * 4 * RK_POSB);
    __builtin_amdgcn_sched_barrier(0);
    acc[4][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2], fp[4], acc[4][2], 0, 0, 0);
    acc[4][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[3], fp[4], acc[4][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) RK_DSR(fp[4], pan, POFF + 4 * 4 * RK_POSB);
    // behind the last block every pixel fragment of this chunk has arrived (phase A waited for them)
    if (TAP == 8) add1(F_FREE_P + buf);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot; gen = ngen; buf = nbuf; pgen = npgen;
  };
  for (int c = 0; c < a.nch; ++c) {
    step(std::integral_constant<int, 0>{}, false);
    step(std::integral_constant<int, 1>{}, false);
    step(std::integral_constant<int, 2>{}, false);
    step(std::integral_constant<int, 3>{}, false);
    step(std::integral_constant<int, 4>{}, false);
    step(std::integral_constant<int, 5>{}, false);
    step(std::integral_constant<int, 6>{}, false);
    step(std::integral_constant<int, 7>{}, false);
    step(std::integral_constant<int, 8>{}, c + 1 == a.nch);
  }
#undef RK_DSR
  RK_ST(if (a.stats != nullptr && lane == 0) { unsigned long long* o = a.stats + ((size_t)blockIdx.x * RK_NWAVES + wave) * 6; o[2] = __builtin_amdgcn_s_memrealtime() - t_entry; o[4] = __builtin_amdgcn_s_memtime() - tc_loop; o[5] = __builtin_amdgcn_s_memrealtime() - tr_loop; })  // end of the main loop: ticks, shader cycles and 100-MHz ticks of the loop

  // ---- epilogue: lane (q = kq, n) holds channels ch0 .. ch0 + 15 of pixel n of every block ----
  const int ch0 = T.nb * 128 + cg * 64 + kq * 16;
  {
    float sc[16], sh[16];
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) {
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, h4 = {0.f, 0.f, 0.f, 0.f};
      if (a.scale) s4 = *reinterpret_cast<const f32x4*>(a.scale + ch0 + v4 * 4);
      if (a.shift) h4 = *reinterpret_cast<const f32x4*>(a.shift + ch0 + v4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { sc[v4 * 4 + i] = s4[i]; sh[v4 * 4 + i] = h4[i]; }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[j][t][i] * sc[t * 4 + i] + sh[t * 4 + i];
          acc[j][t][i] = a.relu ? fmaxf(v, 0.f) : v;
        }
  }
  const int oy = my_oy0 + (n >> 2);
  if (!HEAD) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.y, 0, a.wt ? (int)((size_t)a.B * a.Hin * a.Win * a.Cout * 2) : 0, 0x00020000);
    // Lane (q, n) holds 32 B of pixel n as two 16-B halves; stored as they lie, one instruction would write 16 B of
    // every 32-B sector and the write-through path sends each partial sector to HBM on its own (WRITE_SIZE = 2 x the
    // output, profiles/r03_hbm_traffic.json before this).  Two row swaps (v_permlane16_swap, v_permlane32_swap) hand
    // the q lanes of a pixel consecutive 16-B pieces: one instruction = 64 contiguous bytes per pixel.
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const int chs = T.nb * 128 + cg * 64 + kq * 8;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int ox = my_ox0 + j * 4 + (n & 3);
      u32x4 ox_[2];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned lo = rk_pack_bf2(acc[j][d >> 1][(d & 1) * 2], acc[j][d >> 1][(d & 1) * 2 + 1]);
        const unsigned hi = rk_pack_bf2(acc[j][2 + (d >> 1)][(d & 1) * 2], acc[j][2 + (d >> 1)][(d & 1) * 2 + 1]);
        const u32x2 r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);      // rows: [0l 0h 2l 2h] [1l 1h 3l 3h]
        const u32x2 t = __builtin_amdgcn_permlane32_swap(r[0], r[1], false, false);  //       [0l 0h 1l 1h] [2l 2h 3l 3h]
        ox_[0][d] = t[0];
        ox_[1][d] = t[1];
      }
      if (!my_ok || oy >= a.Hin || ox >= a.Win) continue;
      const size_t o = (((size_t)my_b * a.Hin + oy) * a.Win + ox) * a.Cout + chs;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (a.wt) __builtin_amdgcn_raw_buffer_store_b128(ox_[hf], yrsrc, (int)((o + hf * 32) * 2), 0, 16);  // write-through
        else *reinterpret_cast<u32x4*>(a.y + o + hf * 32) = ox_[hf];
      }
    }
  } else {
    // fused 1x1 head (ref src/modules.py:115, up2[4]) in fp32 on the VALU: logits[k] = head_b[k] + sum_c act[c] * head_w[k][c].
    // A lane sums its 16 channels, the four q lanes of a pixel and the two channel-half waves add up (shuffles,
    // then one exchange through LDS); lane (q, n) of the cg = 0 wave ends with class q of pixel n.
    float part[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) part[j] = 0.f;
    const int c0 = cg * 64 + kq * 16;  // channel of the 128-wide head input (Cout == 128)
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      float hw[16];
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
        if (k < a.head_n) w4 = *reinterpret_cast<const f32x4*>(a.head_w + (size_t)k * 128 + c0 + v4 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) hw[v4 * 4 + i] = w4[i];
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) s = fmaf(acc[j][t][i], hw[t * 4 + i], s);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (kq == k) part[j] = s;  // every q lane holds the sum over the wave's 64 channels; keep class q
      }
    }
    float* hx = reinterpret_cast<float*>(smem + L::HEADX) + sidx * (80 * 4);
    if (cg == 1) {
#pragma unroll
      for (int j = 0; j < 5; ++j) hx[(j * 16 + n) * 4 + kq] = part[j];
      rk_wait_lgkm0();
      rk_set(flags + F_HEAD + sidx, 1, lane);
    } else {
      rk_wait_ge(flags + F_HEAD + sidx, 1);
      if (kq < a.head_n) {
        const float hb = a.head_b[kq];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int ox = my_ox0 + j * 4 + (n & 3);
          const float v = part[j] + hx[(j * 16 + n) * 4 + kq] + hb;
          if (my_ok && oy < a.Hin && ox < a.Win)
            a.head_out[(((size_t)my_b * a.head_n + kq) * a.Hin + oy) * a.Win + ox] = v;
        }
      }
    }
  }
  RK_ST(if (a.stats != nullptr && lane == 0) {
    unsigned long long* o = a.stats + ((size_t)blockIdx.x * RK_NWAVES + wave) * 6;
    o[0] = (unsigned long long)st0; o[1] = (unsigned long long)st1; o[3] = t_entry;
  })
#undef RK_ST
}

// OIHW fp32 -> ring layout bf16: slab (nb, chunk, tap) = 8 KiB, piece (cg, t, l) at ((cg*4+t)*64+l)*16 B holds
// W[co = nb*128 + cg*64 + ((l&15)>>2)*16 + 4t + (l&3)][ci = chunk*32 + (l>>4)*8 .. +8][tap]
// DGRAD: the conv that maps dY to dX - its output channels are w's INPUT channels and tap (ky, kx) takes w's tap
// (2 - ky, 2 - kx); Cout / Cin below are THAT conv's (w is [Cin][Cout][3][3] then).
template <bool DGRAD>
__global__ void pack_weights_ring_kernel(const float* __restrict__ w, int Cout, int Cin, unsigned short* __restrict__ out) {
  const size_t n = (size_t)Cout * Cin * 9;
  const int nch = Cin / RK_KC;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int j = e & 7;
    size_t r = e >> 3;
    const int l = r & 63; r >>= 6;
    const int t = r & 3; r >>= 2;
    const int cg = r & 1; r >>= 1;
    const int tap = r % 9; r /= 9;
    const int chunk = r % nch;
    const int nb = r / nch;
    const int co = nb * 128 + cg * 64 + ((l & 15) >> 2) * 16 + 4 * t + (l & 3);
    const int ci = chunk * RK_KC + (l >> 4) * 8 + j;
    out[e] = lss_f2bf(DGRAD ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap]);
  }
}

}  // namespace

// Is (shape) a case for the ring kernel?  3x3 / stride 1 / pad 1, bf16, Cout a multiple of 128, channel counts of
// both inputs multiples of 32 (the chunk), scale factor 1 (no skip tensor) or
// >= 2 (source window), head: Cout == 128 and at most 4 classes; and enough pixels for the strips to fill the chip.
extern "C" int lss_conv2d_ring_ok(int B, int H, int W, int Cx, int C2, int up, int Cout, int head_n) {
  if (B <= 0 || H <= 0 || W <= 0 || Cx <= 0 || C2 < 0 || up <= 0 || Cout <= 0 || head_n < 0) return 0;
  if (Cout % 128 != 0 || Cx % 32 != 0 || C2 % 32 != 0) return 0;
  if (head_n > 0 && (Cout != 128 || head_n > 4)) return 0;
  const bool fused = up > 1 || C2 > 0;
  const int Hin = H * up, Win = W * up;
  if (fused) {
    if (up < 2) return 0;
    const float ry = Hin > 1 ? (float)(H - 1) / (float)(Hin - 1) : 0.f, rx = Win > 1 ? (float)(W - 1) / (float)(Win - 1) : 0.f;
    // the 6 x 22 patch of a strip must interpolate from a 5 x 14 source window
    if ((int)(ry * (RK_PH - 1) + 0.999f) + 2 > RK_SRC_H || (int)(rx * (RK_SW + 1) + 0.999f) + 2 > RK_SRC_W) return 0;
  }
  if ((long long)B * Hin * Win * (long long)Cout >= (1LL << 31) / 2) return 0;
  if ((long long)B * H * W * (long long)Cx >= (1LL << 31) || (long long)B * Hin * Win * (long long)(C2 > 0 ? C2 : 1) >= (1LL << 31))
    return 0;
  const long long strips = (long long)B * lss_cdiv(Hin, RK_SH) * lss_cdiv(Win, RK_SW);
  return strips / RK_NS * (Cout / 128) >= 96 ? 1 : 0;
}

// Number of flag waits of the ring kernel that ran into their bound since the module was loaded (synchronises the
// device).  Anything but 0 means a hand-off protocol error: the affected launches produced garbage.
extern "C" int lss_conv2d_ring_timeouts(void) {
  int v = -1;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(lss_ring_timeouts), sizeof(int), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v;
}

extern "C" size_t lss_conv2d_ring_packed_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 128 != 0 || Cin % RK_KC != 0) return 0;
  return (size_t)Cout * Cin * 9 * 2;
}

extern "C" int lss_conv2d_pack_weights_ring(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  if (lss_conv2d_ring_packed_weight_bytes(Cout, Cin) == 0) return LSS_E_SHAPE;
  const size_t n = (size_t)Cout * Cin * 9;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_ring_kernel<false>, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cout, Cin,
                     reinterpret_cast<unsigned short*>(w_packed));
  return lss_launch_status();
}

// Ring-packed weight of the conv that maps dY (Cout channels) to dX (Cin channels) of a 3x3 / s1 / p1 conv with
// weight w_oihw [Cout][Cin][3][3]: use with lss_conv2d_fwd(dy, ..., Cx = Cout, Cout = Cin, LSS_W_RING).
extern "C" int lss_conv2d_pack_weights_ring_dgrad(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  if (lss_conv2d_ring_packed_weight_bytes(Cin, Cout) == 0) return LSS_E_SHAPE;
  const size_t n = (size_t)Cout * Cin * 9;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_ring_kernel<true>, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cin, Cout,
                     reinterpret_cast<unsigned short*>(w_packed));
  return lss_launch_status();
}

// launcher shared by lss_conv2d_fwd / lss_conv2d_head_fwd (conv_mfma.hip) when the weights are ring-packed
int lss_conv_ring_launch(const void* x, const void* x2, const void* w_ring, const float* scale, const float* shift,
                         void* y, const float* head_w, const float* head_b, float* head_out, int head_n, int B, int H,
                         int W, int Cx, int C2, int up, int Cout, int relu, int wt, hipStream_t st) {
  if (!lss_conv2d_ring_ok(B, H, W, Cx, C2, up, Cout, head_n)) return LSS_E_SHAPE;
  if (relu != 0 && relu != 1) return LSS_E_LAYOUT;
  RingArgs a;
  a.x = reinterpret_cast<const unsigned short*>(x);
  a.x2 = reinterpret_cast<const unsigned short*>(x2);
  a.w = reinterpret_cast<const unsigned char*>(w_ring);
  a.scale = scale; a.shift = shift;
  a.y = reinterpret_cast<unsigned short*>(y);
  a.head_w = head_w; a.head_b = head_b; a.head_out = head_out; a.head_n = head_n;
  a.B = B; a.H = H; a.W = W; a.Cx = Cx; a.C2 = C2; a.up = up;
  a.Hin = H * up; a.Win = W * up; a.Cin = Cx + C2; a.Cout = Cout; a.relu = relu; a.wt = wt;
  a.ry = a.Hin > 1 ? (float)(H - 1) / (float)(a.Hin - 1) : 0.f;
  a.rx = a.Win > 1 ? (float)(W - 1) / (float)(a.Win - 1) : 0.f;
  a.SX = lss_cdiv(a.Win, RK_SW); a.SY = lss_cdiv(a.Hin, RK_SH);
  a.nstrips = B * a.SX * a.SY;
  a.nblk = Cout / 128;
  a.nch = a.Cin / RK_KC;
  {
    const char* e = getenv("LSS_RING_STATS");
    a.stats = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16)) : nullptr;
  }
  const bool fused = up > 1 || C2 > 0;
  const dim3 g((a.nstrips + RK_NS - 1) / RK_NS * a.nblk), blk(RK_NWAVES * 64);
  if (head_n > 0) {
    if (fused) hipLaunchKernelGGL((conv_ring_kernel<1, true, 7>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_ring_kernel<0, true, 8>), g, blk, 0, st, a);
  } else {
    if (fused) hipLaunchKernelGGL((conv_ring_kernel<1, false, 7>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_ring_kernel<0, false, 8>), g, blk, 0, st, a);
  }
  return lss_launch_status();
}
