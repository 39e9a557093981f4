// K9w: weight gradient of the 3x3 / stride-1 / pad-1 bf16 NHWC convolutions straight from the NHWC tensors
// (replaces, for training, the ConvolutionBackward nodes of the convs in ref src/modules.py:22-27, 118-130 that
// `loss.backward()` runs in train.py:61):
//     dW[co][ci][ky][kx] = sum_{b,y,x} dY[b,y,x,co] * X[b, y+ky-1, x+kx-1, ci]
// is, per tap, a GEMM whose K dimension is the PIXEL index - the slow dimension of both NHWC operands.  Round 2
// transposed both tensors to channel-major copies (three column-shifted ones of X) and ran a split-K GEMM per tap:
// 0.26 ms of transposes + 0.78 ms of GEMMs per training step, each tap re-reading both operands.  Here the pixel
// dimension is transposed ON THE WAY INTO THE MATRIX CORE by gfx950's `ds_read_b64_tr_b16` (a 4-position x 16-channel
// block of a position-major LDS image arrives channel-major in the lanes), so the kernel reads X and dY once, as they
// lie, and one LDS image of an X row serves all nine taps.
//
// Workgroup = 64 output channels x 64 input channels x all 9 taps, over a range of (padded) image rows; 12 waves:
//   waves 0-8  consumers, ONE TAP EACH: 4 co-tiles x 4 ci-tiles of v_mfma_f32_16x16x32_bf16 (64 accumulator registers);
//              per K block (32 pixel positions of a row) 8 + 8 transposed reads feed 16 MFMAs (0.5 KiB per MFMA)
//   wave  9    dY loader: K blocks of a row (32 positions x 64 co = 4 KiB) into a ring of 10 blocks by LDS-DMA
//   waves 10-11 X loaders: whole rows (W + 2 positions x 64 ci) into a ring of 4-8 row slots; the row above / below an
//              image is a zero row (the zero padding of the conv), as are positions x = -1 and x = W
// Tap (ky, kx) of the tile of row r reads row slot r + ky - 1 at position x + kx: the row shift picks a slot, the column
// shift moves the read by one 128-B position - no shifted copies.  No workgroup barrier after start-up: FULL / FREE
// words in LDS as in conv_ring.hip (bounded polls; lss_conv2d_wgrad_timeouts() must read 0).
// LDS image: position-major, 128 B per position = four 32-B channel tiles, tile index XOR-swizzled by
// f(pos) = bit1(pos) | bit3(pos) << 1: the 8 positions a 32-lane half reads ({s..s+3} and {s+8..s+11} for any shift
// s) then fall in 8 different 8-bank groups - conflict-free transposed reads for every tap.
// Output: fp32 partial tiles [split][tap][Cout][Cin], reduced in split order by conv_grad.hip's wgrad_reduce_kernel
// (fixed order: a training step is bit-reproducible).
#include <stdlib.h>

#include "lss_common.h"

namespace {

__device__ int lss_wgrad_timeouts;  // flag waits that hit their bound (must stay 0)
#define RK_TIMEOUT_COUNTER lss_wgrad_timeouts
#include "ring_prims.h"

constexpr int WK_NDB = 10;          // dY ring: K blocks
constexpr int WK_BLK = 32 * 128;    // 4096 B: 32 positions x 64 channels
constexpr int WK_NXL = 2;            // X loader waves
constexpr int WK_MAXKB = 7;         // K blocks per row (W <= 224)
constexpr int WK_MAXXS = 8;         // X row slots
constexpr int WK_LDS_MAX = 160 * 1024;
// flag words
constexpr int F_FULL_X = 0, F_FREE_X = 8, F_FULL_D = 16, F_FREE_D = 32, WK_NFLAGS = 64;

struct WgradArgs {
  const unsigned short* x;   // (B, H, W, Cin) bf16 NHWC
  const unsigned short* dy;  // (B, H, W, Cout) bf16 NHWC
  float* partial;            // [nsplit][9][Cout][Cin]
  int B, H, W, Cin, Cout;
  int KB;          // K blocks per row = ceil(W / 32)
  int nxs;         // X row slots
  int xs_bytes;    // bytes of a slot: (32 KB + 8) positions x 128
  int rows_total;  // B * (H + 2) padded rows
  int rows_per;    // padded rows per split
  int ntap_total;  // taps of the partial tile array (9, or 16 for the two-launch 4x4 form)
  int tap_base;    // first tap this launch writes
};

__device__ __attribute__((aligned(128))) unsigned char lss_wgrad_zero_page[128];

typedef __attribute__((ext_vector_type(4))) short wk_s16x4;
typedef __attribute__((ext_vector_type(8))) short wk_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 wk_bf16x8;
typedef __attribute__((ext_vector_type(4))) float wk_f32x4;

__device__ __forceinline__ int wk_swz(int pos) { return ((pos >> 1) & 1) | (((pos >> 3) & 1) << 1); }

// 4 positions x 16 channels, transposed: this lane's channel at the 4 positions
__device__ __forceinline__ wk_s16x4 wk_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wk_s16x4*)p);
}
__device__ __forceinline__ wk_bf16x8 wk_frag(const unsigned char* lo, const unsigned char* hi) {
  const wk_s16x4 a = wk_tr(lo), b = wk_tr(hi);
  const wk_s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(wk_bf16x8, v);
}
// Tap grid of one launch: NTY x NTX taps (one consumer wave each, <= 9), tap (ty, tx) = displacement (DY0 + ty,
// tx - LP): the X row image keeps LP zero positions in front of x = 0, so a tap reads position (pixel + tx).
// <3, 3, -1, 1>: the 3x3 / pad-1 conv.  <2, 4, -2, 2> + <2, 4, 0, 2>: the 4x4 taps {-2 .. 1}^2 of a 7x7 / 2 / pad-3
// conv over phase planes, as two launches of eight consumers (two per SIMD).
template <int NTY, int NTX, int DY0, int LP>
__global__ __launch_bounds__((NTY * NTX + 1 + WK_NXL) * 64) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int WK_NCONS = NTY * NTX;
  static_assert(WK_NCONS <= 9 && NTX + LP <= 8, "tap grid");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ntc = a.Cin >> 6, nto = a.Cout >> 6;
  int bid = blockIdx.x;
  const int cit = bid % ntc; bid /= ntc;
  const int cot = bid % nto;
  const int split = bid / nto;
  const int ci0 = cit * 64, co0 = cot * 64;
  const int pi0 = split * a.rows_per;
  const int ntile = min(a.rows_total, pi0 + a.rows_per) - pi0;  // >= 1 by construction of the grid
  const int HP = a.H + 2;
  const int XBASE = 0, DBASE = a.nxs * a.xs_bytes, FBASE = DBASE + WK_NDB * WK_BLK;
  const rk_flag_t flags = (rk_flag_t)((__attribute__((address_space(3))) unsigned char*)smem + FBASE);
  if (tid < WK_NFLAGS) flags[tid] = 0;
  __syncthreads();

  if (wave >= WK_NCONS + 1) {
    // ============================ X loaders: rows pi0 - 1 .. pi0 + ntile ============================
    const int wl = wave - (WK_NCONS + 1);
    const int npieces = 4 * a.KB + 1;  // 8 positions per 1-KiB DMA piece
    constexpr int MAXP = (4 * WK_MAXKB + 1 + WK_NXL - 1) / WK_NXL;
    int xoff[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int i = wl + k * WK_NXL;
      const int idx = 8 * i + (lane >> 3), x = idx - LP;
      const int sb = (lane & 7) >> 1, half = lane & 1;
      const int ch = ci0 + ((sb ^ wk_swz(idx)) << 4) + half * 8;
      xoff[k] = (i < npieces && x >= 0 && x < a.W) ? x * a.Cin + ch : -1;
    }
    const unsigned char* zsrc = lss_wgrad_zero_page + (lane & 7) * 16;
    int slot = 0, uses = 0;
    for (int k = 0; k < ntile + NTY - 1; ++k) {
      // slot `slot` last held row k - nxs: every consumer releases every row once (the taps of row ky use it at tile
      // k - ky; a wave that never uses one of the first rows releases it at start-up), so `uses` x 9 releases free it.
      // A single "tiles done" counter would not do: the consumers drift by up to the dY ring's depth, and eight waves
      // two tiles ahead would add up to the count that was meant to say "all nine have finished".
      if (uses > 0) rk_wait_ge<4>(flags + F_FREE_X + slot, WK_NCONS * uses);
      const int rho = pi0 + DY0 + k;
      const bool inr = rho >= 0 && rho < a.rows_total;
      const int b = inr ? rho / HP : 0, yp = inr ? rho - b * HP : 0;
      const bool real = inr && yp >= 1 && yp <= a.H;
      const unsigned short* rowp = a.x + (real ? (size_t)((b * a.H + (yp - 1)) * a.W) * a.Cin : 0);
      unsigned char* dst = smem + XBASE + slot * a.xs_bytes;
#pragma unroll
      for (int kk = 0; kk < MAXP; ++kk) {
        const int i = wl + kk * WK_NXL;
        if (i < npieces) {
          int xo = xoff[kk];
          asm volatile("" : "+v"(xo));  // keeps the 64-bit addresses out of loop-invariant registers
          const void* src = (real && xo >= 0) ? (const void*)(rowp + xo) : (const void*)zsrc;
          rk_glds16(src, dst + i * 1024);
        }
      }
      rk_wait_vmcnt<0>();
      rk_add1(flags + F_FULL_X + slot, lane);
      if (++slot == a.nxs) { slot = 0; ++uses; }
    }
    rk_wait_vmcnt<0>();  // never end a loader wave with LDS-DMA in flight (every fill above is already drained)
    return;
  }

  if (wave == WK_NCONS) {
    // ============================ dY loader: the K blocks of the real rows ============================
    int doff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pos = 8 * i + (lane >> 3);
      const int sb = (lane & 7) >> 1, half = lane & 1;
      doff[i] = pos * a.Cout + co0 + ((sb ^ wk_swz(pos)) << 4) + half * 8;
    }
    const unsigned char* zsrc = lss_wgrad_zero_page + (lane & 7) * 16;
    int slot = 0, uses = 0;   // uses = earlier fills of `slot`
    int prev = -1;            // slot whose DMA is in flight and unpublished
    for (int t = 0; t < ntile; ++t) {
      const int pi = pi0 + t;
      const int b = pi / HP, yp = pi - b * HP;
      if (yp < 1 || yp > a.H) continue;
      const unsigned short* rowp = a.dy + (size_t)((b * a.H + (yp - 1)) * a.W) * a.Cout;
      for (int j = 0; j < a.KB; ++j) {
        if (uses > 0 && rk_peek(flags + F_FREE_D + slot) < WK_NCONS * uses) {
          // about to block on the consumers: everything issued so far must be visible to them first
          if (prev >= 0) {
            rk_wait_vmcnt<0>();
            rk_add1(flags + F_FULL_D + prev, lane);
            prev = -1;
          }
          rk_wait_ge<2>(flags + F_FREE_D + slot, WK_NCONS * uses);
        }
        unsigned char* dst = smem + DBASE + slot * WK_BLK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = 32 * j + 8 * i + (lane >> 3);
          int dof = doff[i];
          asm volatile("" : "+v"(dof));
          const void* src = x < a.W ? (const void*)(rowp + 32 * j * a.Cout + dof) : (const void*)zsrc;
          rk_glds16(src, dst + i * 1024);
        }
        if (prev >= 0) {  // the block before this one has landed once at most these 4 pieces are outstanding
          rk_wait_vmcnt<4>();
          rk_add1(flags + F_FULL_D + prev, lane);
        }
        prev = slot;
        if (++slot == WK_NDB) { slot = 0; ++uses; }
      }
    }
    if (prev >= 0) {
      rk_wait_vmcnt<0>();
      rk_add1(flags + F_FULL_D + prev, lane);
    }
    return;
  }

  // ======================================= consumers: one tap each =======================================
  const int ky = wave / NTX, kx = wave - NTX * ky;  // (ty, tx) of this wave's tap
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int aoff[2][4], boff[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pa = 8 * g + 4 * h + q, pb = pa + kx;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      aoff[h][ct] = pa * 128 + ((ct ^ wk_swz(pa)) << 5) + p * 8;
      boff[h][ct] = pb * 128 + ((ct ^ wk_swz(pb)) << 5) + p * 8;
    }
  }
  wk_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (wk_f32x4){0.f, 0.f, 0.f, 0.f};

  int dslot = 0, dfill = 1;      // dY ring position and the fill count that makes its block valid
  int xslot = ky, xfill = 1;     // row slot of this tap's row of tile 0 (row k = t + ky), its fill count
  for (int r = 0; r < ky; ++r) rk_add1(flags + F_FREE_X + r, lane);  // rows 0 .. ky - 1: never read by this tap
  for (int t = 0; t < ntile; ++t) {
    const int pi = pi0 + t;
    const int b = pi / HP, yp = pi - b * HP;
    // EVERY tile waits for its row slot's fill before releasing it - pad tiles too, although they read nothing: a
    // release may only ever count for the fill it names.  (Round 3 let pad tiles release without the wait: at an image
    // boundary inside a row range the fast taps ran through the two pad tiles and released the slot's NEXT fill, and
    // FREE_X - one summed counter per slot - then reached NCONS x uses while a slow tap was still reading the current
    // fill: the loader could overwrite a row under a reader (batch >= 2 with nxs <= 5, i.e. W > 128).)
    rk_wait_ge(flags + F_FULL_X + xslot, WK_NXL * xfill);
    if (yp >= 1 && yp <= a.H) {
      const unsigned char* xb = smem + XBASE + xslot * a.xs_bytes;
      for (int j = 0; j < a.KB; ++j) {
        rk_wait_ge(flags + F_FULL_D + dslot, dfill);
        const unsigned char* db = smem + DBASE + dslot * WK_BLK;
        const unsigned char* xj = xb + j * WK_BLK;
        wk_bf16x8 fa[4], fb[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fa[ct] = wk_frag(db + aoff[0][ct], db + aoff[1][ct]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = wk_frag(xj + boff[0][ct], xj + boff[1][ct]);
        // LDS operations of a wave execute in order: the add is performed after the reads above have been
        rk_add1(flags + F_FREE_D + dslot, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
        if (++dslot == WK_NDB) { dslot = 0; ++dfill; }
      }
    }
    rk_add1(flags + F_FREE_X + xslot, lane);  // (after the tile's last reads, in LDS order; pad tiles release too, behind the wait above)
    if (++xslot == a.nxs) { xslot = 0; ++xfill; }
  }

  // partial[split][tap][co][ci]: lane (g, n = lane & 15) holds rows 4 g + i, column n of every 16 x 16 tile
  const int n = lane & 15;
  float* out = a.partial + ((size_t)split * a.ntap_total + a.tap_base + wave) * a.Cout * a.Cin + (size_t)co0 * a.Cin + ci0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)(i * 16 + 4 * g + r) * a.Cin + jj * 16 + n] = acc[i][jj][r];
}

struct WgradPlan {
  int ok, KB, nxs, xs_bytes, rows_total, rows_per, nsplit, lds;
};

WgradPlan wgrad_plan(int B, int H, int W, int Cin, int Cout) {
  WgradPlan p = {};
  if (B <= 0 || H <= 0 || W < 8 || Cin <= 0 || Cout <= 0) return p;
  if (Cin % 64 != 0 || Cout % 64 != 0) return p;
  p.KB = (W + 31) / 32;
  if (p.KB > WK_MAXKB) return p;
  p.xs_bytes = (32 * p.KB + 8) * 128;
  const int room = WK_LDS_MAX - WK_NFLAGS * 4 - WK_NDB * WK_BLK;
  p.nxs = room / p.xs_bytes;
  if (p.nxs > WK_MAXXS) p.nxs = WK_MAXXS;
  if (p.nxs < 4) return p;
  p.lds = p.nxs * p.xs_bytes + WK_NDB * WK_BLK + WK_NFLAGS * 4;
  p.rows_total = B * (H + 2);
  const int tiles = (Cin / 64) * (Cout / 64);
  int ns = 256 / tiles;  // at most one workgroup per CU: a 257th would run alone in a second round
  if (const char* e = getenv("LSS_WGRAD_SPLITS")) ns = atoi(e);  // developer override (tools/bench_wgrad.py)
  if (ns < 1) ns = 1;
  int per = (p.rows_total + ns - 1) / ns;
  if (per < 3) per = 3;
  p.rows_per = per;
  p.nsplit = (p.rows_total + per - 1) / per;
  p.ok = 1;
  return p;
}

}  // namespace

// internal (conv_grad.hip): is (shape) a case for the direct kernel, how many fp32 partial tiles it writes
int lss_wgrad_direct_splits(int B, int H, int W, int Cin, int Cout) {
  const char* e = getenv("LSS_WGRAD_DIRECT");
  if (e != nullptr && atoi(e) == 0) return 0;
  const WgradPlan p = wgrad_plan(B, H, W, Cin, Cout);
  return p.ok ? p.nsplit : 0;
}

namespace {

template <int NTY, int NTX, int DY0, int LP>
int launch_taps(const WgradPlan& p, WgradArgs& a, hipStream_t st) {
  static bool attr_set[64] = {};  // per instantiation and device (the attribute belongs to the device's code object)
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<NTY, NTX, DY0, LP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) attr_set[dev] = true;
  }
  const int grid = (a.Cin / 64) * (a.Cout / 64) * p.nsplit;
  hipLaunchKernelGGL((conv_wgrad_kernel<NTY, NTX, DY0, LP>), dim3(grid), dim3((NTY * NTX + 1 + WK_NXL) * 64), p.lds, st, a);
  return lss_launch_status();
}

WgradArgs wgrad_args(const WgradPlan& p, const void* x, const void* dy, int B, int H, int W, int Cin, int Cout,
                     float* partial) {
  WgradArgs a;
  a.x = static_cast<const unsigned short*>(x);
  a.dy = static_cast<const unsigned short*>(dy);
  a.partial = partial;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.KB = p.KB; a.nxs = p.nxs; a.xs_bytes = p.xs_bytes; a.rows_total = p.rows_total; a.rows_per = p.rows_per;
  a.ntap_total = 9; a.tap_base = 0;
  return a;
}

}  // namespace

int lss_wgrad_direct_launch(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, float* partial,
                            hipStream_t st) {
  const WgradPlan p = wgrad_plan(B, H, W, Cin, Cout);
  if (!p.ok) return LSS_E_SHAPE;
  WgradArgs a = wgrad_args(p, x, dy, B, H, W, Cin, Cout, partial);
  return launch_taps<3, 3, -1, 1>(p, a, st);
}

// The 4 x 4 taps (dy, dx) in {-2 .. 1}^2 (tap index (dy + 2) * 4 + dx + 2): the phase-plane form of a 7x7 / stride-2 /
// pad-3 conv.  Two launches of eight consumer waves (rows dy = -2, -1 and dy = 0, 1) into partial[split][16][Cout][Cin].
int lss_wgrad_taps4x4_launch(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, float* partial,
                             hipStream_t st) {
  const WgradPlan p = wgrad_plan(B, H, W, Cin, Cout);
  if (!p.ok) return LSS_E_SHAPE;
  WgradArgs a = wgrad_args(p, x, dy, B, H, W, Cin, Cout, partial);
  a.ntap_total = 16;
  a.tap_base = 0;
  int rc = launch_taps<2, 4, -2, 2>(p, a, st);
  if (rc != 0) return rc;
  a.tap_base = 8;
  return launch_taps<2, 4, 0, 2>(p, a, st);
}

extern "C" int lss_conv2d_wgrad_timeouts(void) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(lss_wgrad_timeouts), sizeof(int)) != hipSuccess) return -1;
  return v;
}
