// K8k: the launch-bound 3x3 / stride-1 convolutions of BevEncode - resnet18's layer1 / layer2 / layer3 at 100^2 x 64,
// 50^2 x 128, 25^2 x 256 (ref src/modules.py:104-106, 123-125 + torchvision BasicBlock: conv3x3 - BN - ReLU - conv3x3 -
// BN - (+identity) - ReLU) - as ONE PASS per workgroup, without a K loop.
//
// Why another kernel.  On the tile kernel (conv_mfma.hip) these ten launches cost 10.5-14.9 us each for 1.2 us of MFMA
// work (profiles/r03_conv_phase_stamps.txt): a workgroup walks 9-36 (chunk, tap) steps, each a weight slab by LDS-DMA,
// a barrier and 12 ds_read_b128 per wave for 8 MFMAs - the steps are LDS-read bound (96 KiB of fragment reads per step
// and CU against 512 MFMA cycles) and latency bound (0.37-0.46 us per step against 0.11 us of MFMA issue), and the grids
// (112-364 workgroups) leave half of the chip idle.  Deeper weight rings did not move them (profiles/r04_deep_ring_ab.txt).
// Here the work is cut so that EVERYTHING a workgroup needs is on the CU before the first MFMA, and the K dimension is
// split over the WAVES of the workgroup instead of over steps:
//   * workgroup = PB flattened pixels of one image (80 / 160 / 320) x 32 output channels x the whole K = 9 Cin;
//     256 workgroups at batch 4 for all three layers (one per CU, one round);
//   * the input patch - the <= 5 image rows the pixels span plus one halo row above and below, all Cin channels,
//     zero-padded - goes to LDS by LDS-DMA ONCE (91-98 KiB: [32-channel chunk][16-channel half][position][32 B], so that
//     the B fragment of 16 consecutive pixels is one conflict-free 1-KiB read for every tap shift);
//   * the weights never touch LDS: wave w owns a K part (layer3: chunks 2w, 2w + 1 x 9 taps = 18 k-steps of 32; layer2:
//     chunk w x 9 taps; layer1: 2 K parts x 2 pixel halves) and keeps its A fragments - 2 channel tiles x 9-18 k-steps
//     x 4 registers - in REGISTERS for the whole launch, loaded with fully coalesced 16-B-per-lane loads from a pack
//     that has exactly this image (lss_conv2d_pack_weights_ks);
//   * main phase: for every k-step the wave reads the B fragment of each of its 5-10 pixel tiles from LDS (one
//     ds_read_b128 = 16 pixels x 32 channels) and issues two v_mfma_f32_16x16x32_bf16 against the two channel tiles:
//     0.5 KiB of LDS reads per MFMA - the ratio of the ring kernel (section 4c) - no barrier, no flag, no DMA inside;
//   * the K parts meet through LDS once (fp32, fixed order kp = 0, 1, 2, 3: deterministic), each wave finishing a
//     quarter of the pixel tiles: folded BatchNorm scale / shift, residual, ReLU, one 16-B store per lane - the rows of
//     channel tile t are the channels {8 q + 4 t + i}, so lane (q, n) ends with 8 CONSECUTIVE channels of pixel n and
//     the four q lanes of a pixel write 64 contiguous bytes (no LDS staging of the output tile).
// No inter-workgroup communication, no bounded waits: nothing here can hang.
#include <stdlib.h>

#include "lss_common.h"

namespace {

constexpr int KS_ROWS = 7;        // patch rows: the <= 5 image rows a pixel block spans + the halo row above and below
constexpr int KS_POSB = 64;       // bytes per patch position and chunk (32 channels bf16)
constexpr int KS_LDS_MAX = 160 * 1024;

__device__ __attribute__((aligned(128))) unsigned char lss_ks_zero_page[128];  // source of out-of-image patch pieces

__device__ __forceinline__ void ks_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct KsArgs {
  const unsigned short* x;        // (B, H, W, Cin) bf16 NHWC
  const unsigned char* w;         // lss_conv2d_pack_weights_ks
  const float* scale;             // folded BatchNorm (or null: 1)
  const float* shift;             // (or null: 0)
  const unsigned short* residual; // (B, H, W, Cout) bf16 NHWC or null
  unsigned short* y;              // (B, H, W, Cout) bf16 NHWC
  int B, H, W, Cin, Cout, relu, wt;
  int PB;                         // pixels per workgroup
  int npb;                        // pixel blocks per image
  int nposp;                      // patch positions per chunk, padded to a multiple of 32
  int ncb;                        // 32-channel output blocks
  unsigned long long* stamps;     // diagnostics (LSS_KS_STAMPS=<hex device address>, tools/bench_ks.py --stamps): 8 x 100-MHz
                                  // s_memrealtime stamps per workgroup, or null
};

// KSW: k-steps (32 input channels x one tap) per wave; NKW: K parts; PXT: 16-pixel tiles per wave.  NKW * NPW = 4 waves.
template <int KSW, int NKW, int PXT>
__global__ __launch_bounds__(256, 1) void conv_ks_kernel(const KsArgs a) {
  static_assert(KSW % 9 == 0 && (NKW == 4 || NKW == 2), "a wave's K part is whole chunks of nine taps");
  constexpr int NPW = 4 / NKW;          // pixel parts
  constexpr int CPW = KSW / 9;          // 32-channel chunks per wave
  constexpr int NT = PXT * NPW;         // pixel tiles of the workgroup
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kp = wave % NKW, ph = wave / NKW;  // K part, pixel part
  const int n = lane & 15, kq = lane >> 4;
  auto stamp = [&](int k) {
    if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  unsigned long long clk_main = 0;  // shader-clock cycles of the main phase (slot 7 of the stamps)

  // ---- which block: XCD-aware order, channel blocks of one pixel block adjacent (they share the input patch) ----
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int cb = t % a.ncb;
  const int pbg = t / a.ncb;
  const int b = pbg / a.npb, pb = pbg - b * a.npb;
  const int HW = a.H * a.W, WP = a.W + 2;
  const int p0 = pb * a.PB;                       // first pixel of the block (flattened, inside image b)
  const int y_first = p0 / a.W;                   // patch row 0 = image row y_first - 1

  // ---- input patch -> LDS by LDS-DMA: [chunk][channel half][position][32 B]; a piece = 32 positions of one half ----
  // (row / column of a lane's position advance by 64 positions per piece - no per-piece division: the first version
  // spent ~60 VALU instructions of address arithmetic in front of every DMA instruction.  Staging the pieces through
  // registers instead - plain 16-B loads, 24 in flight per wave, one ds_write_b128 each - was built and measured: the
  // same 3-4 us from kernel entry to "patch and weights on the CU" (layer3 7.2 -> 8.2 us per launch): what bounds this
  // phase is the 130-240 KiB every one of the 256 CUs pulls through its 64-B-per-clock L1 path at the same moment,
  // 33-62 MB out of the L2s in ~3 us, not the way the requests are issued.)
  // Layout and bank conflicts.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19,
  // 28-31}, ...: MI355X_MICROARCH.md, LDS) against a 256-B window of banks.  With 64 B per position (the four channel
  // pieces side by side) the window is four positions and the lanes of a group that share a piece index - pixels n and
  // n + 12, n + 4 and n + 8 - fall on the same 16 B: 2-way conflicts on every read.  With the two channel HALVES of a
  // chunk in separate planes of 32 B per position the window is eight positions, a group's lanes of piece pair (0, 1) or
  // (2, 3) cover sixteen different 16-B cells for ANY alignment of the 16 pixels - every tap shift - and the address is
  // plane + position * 32 + (piece & 1) * 16: no arithmetic beyond the tap offset.  What that bought, measured (main
  // phase, layer1 / 2 / 3, tools/bench_ks.py --stamps): 64-B positions with the conflicts 2.60 / 2.56 / 2.36 us; the same
  // with an XOR swizzle of the piece index (conflict-free, four VALU instructions per read next to back-to-back MFMAs)
  // 3.20 / 3.16 / 2.72; this layout 2.54 / 2.48 / 2.32 - and the two timing-only builds of tools/build_diag_libs.sh say
  // why the conflicts never mattered: the 180 MFMAs of a wave alone (KS_NOREAD) take 2.06 / 2.04 / 1.84 us, the 90
  // fragment reads alone (KS_NOMFMA) 1.20 / 1.28 / 1.36: the phase is the matrix pipe plus the quarter of the reads that
  // does not hide behind it, at one wave per SIMD.
  constexpr int NCH = NKW * KSW / 9;           // 32-channel chunks of the input
  {
    const int ppc = a.nposp >> 4;                 // pieces per chunk: nposp / 32 position blocks x 2 halves
    const int q64 = 64 / WP, r64 = 64 - q64 * WP;  // wave-uniform
    const int h = wave & 1;                       // this wave's pieces: position blocks (wave >> 1) + 2 k of half h
    const int pos0 = 32 * (wave >> 1) + (lane >> 1);
    int pr = pos0 / WP, pc = pos0 - pr * WP;
    const unsigned char* zsrc = lss_ks_zero_page + (lane & 7) * 16;
    const unsigned short* xb = a.x + (size_t)b * HW * a.Cin + (2 * h + (lane & 1)) * 8;
    for (int i = wave; i < ppc; i += 4) {
      const int iy = y_first - 1 + pr, ix = pc - 1;
      const bool in = pr < KS_ROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const unsigned short* src = xb + (in ? (iy * a.W + ix) * a.Cin : 0);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        ks_glds16(in ? (const void*)(src + c * 32) : (const void*)zsrc,
                  smem + ((size_t)(c * 2 + h) * a.nposp + 32 * (i >> 1)) * 32);
      pr += q64; pc += r64;
      if (pc >= WP) { pc -= WP; ++pr; }
    }
  }

  // ---- weights: this wave's KSW x 2 A fragments, straight into registers (coalesced 1-KiB loads) ----
  // Only the first WPRE k-steps are requested here; the main phase requests k-step s + WPRE while it computes k-step s
  // (two 1-KiB loads per 10-20 MFMAs: nothing next to the LDS-DMA burst of the prologue).  The prologue is a fill-rate
  // problem - every CU pulls its patch AND its 72-144 KiB of weights at once, ~70 GB/s per CU - so bytes that can
  // arrive during the MFMAs should: with everything requested up front (-DKS_WPRE_ALL, the first form of this kernel)
  // kernel entry -> operands landed took 3.0-3.6 us of the 7-9.
#ifdef KS_WPRE_ALL
  constexpr int WPRE = KSW;
#else
  constexpr int WPRE = KSW >= 18 ? 8 : 5;
#endif
  bf16x8 wf[KSW][2];
  const unsigned char* wp = a.w + ((size_t)(cb * NKW + kp) * KSW * 2) * 1024 + lane * 16;
  auto load_w = [&](int s) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wf[s][ct] = *reinterpret_cast<const bf16x8*>(wp + (s * 2 + ct) * 1024);
  };
#pragma unroll
  for (int s = 0; s < WPRE; ++s) load_w(s);

  // ---- per-lane patch offsets of the wave's pixel tiles (pixel n of tile j; tap (ky, kx) adds (ky WP + kx) 64) ----
  // and the residual pieces of the tiles this wave will finish (requested now: long landed when the epilogue wants them)
  const int npx = min(a.PB, HW - p0);
  int ab[PXT];
  {
    const int x_first = p0 - y_first * a.W;
    const int plast = npx - 1;                                   // pixels past the block's end read its last one
    const int yl = (x_first + plast) / a.W, xl = x_first + plast - yl * a.W;   // wave-uniform
    int pl = ph * PXT * 16 + n;
    int y = (x_first + pl) / a.W, x = x_first + pl - y * a.W;     // relative to y_first
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
      const bool live = pl < npx;
      ab[j] = ((live ? y : yl) * WP + (live ? x : xl)) * 32 + (kq >> 1) * a.nposp * 32 + (kq & 1) * 16;
      pl += 16; x += 16;
      if (x >= a.W) { x -= a.W; ++y; }
    }
  }
  stamp(1);  // every request (weights, patch pieces) has been issued
  constexpr int NOWN = (PXT + NKW - 1) / NKW;   // tiles a wave finishes: j = kp, kp + NKW, ...
  const int ch = cb * 32 + kq * 8;             // this lane's 8 consecutive output channels
  uint4 rres[NOWN];
#pragma unroll
  for (int k = 0; k < NOWN; ++k) {
    const int j = kp + k * NKW;
    const int pl = (ph * PXT + j) * 16 + n;
    rres[k] = make_uint4(0, 0, 0, 0);
    if (a.residual != nullptr && j < PXT && pl < npx)
      rres[k] = *reinterpret_cast<const uint4*>(a.residual + ((size_t)b * HW + p0 + pl) * a.Cout + ch);
  }
  f32x4 acc[PXT][2];
#pragma unroll
  for (int j = 0; j < PXT; ++j) {
    acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the patch pieces (and the weight fragments) have landed
  __syncthreads();
  stamp(2);

  // ---- main phase: KSW k-steps x PXT pixel tiles x 2 channel tiles, nothing but LDS reads and MFMAs ----
  // ONE wave per SIMD: nobody else hides an LDS round trip, so the pixel fragments of k-step s + 1 are requested - all
  // PXT of them - before the 2 PXT MFMAs of k-step s issue (two fragment sets by k-step parity; sched_barrier keeps
  // hipcc from sinking the requests next to their use: left alone it waited with lgkmcnt(0) in front of every MFMA
  // pair, i.e. one exposed LDS latency per 32 matrix cycles).
  bf16x8 fb[2][PXT];
  auto load_frags = [&](int buf, int s) {
    const unsigned char* cbase = smem + (size_t)(kp * CPW + s / 9) * a.nposp * KS_POSB;
    const int tap = s % 9;
    const int toff = ((tap / 3) * WP + (tap % 3)) * 32;
#pragma unroll
    for (int j = 0; j < PXT; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(cbase + ab[j] + toff);
  };
  if (a.stamps != nullptr) clk_main = __builtin_amdgcn_s_memtime();
  load_frags(0, 0);
#pragma unroll
  for (int s = 0; s < KSW; ++s) {
    if (s + WPRE < KSW) load_w(s + WPRE);
#ifndef KS_DIAG_NOREAD                                // timing-only diagnostic builds (tools/build_diag_libs.sh KS_NOREAD ..)
    if (s + 1 < KSW) load_frags((s + 1) & 1, s + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
#ifdef KS_DIAG_NOMFMA
      asm volatile("" :: "v"(fb[s & 1][j]), "v"(wf[s][0]), "v"(wf[s][1]));
#else
      acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][0], fb[s & 1][j], acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][1], fb[s & 1][j], acc[j][1], 0, 0, 0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- the K parts meet: part[kp][tile][ct][lane] (16 B each), summed in the fixed order kp = 0 .. NKW - 1 ----
  if (a.stamps != nullptr) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[PXT - 1][1]));  // the MFMA chain has retired
  stamp(3);
  if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - clk_main;
  __syncthreads();  // every wave is done reading the patch: its LDS is free
  f32x4* part = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int j = 0; j < PXT; ++j) {
    const int tile = ph * PXT + j;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) part[((kp * NT + tile) * 2 + ct) * 64 + lane] = acc[j][ct];
  }
  // epilogue constants of this lane's 8 consecutive channels, requested before the barrier
  float sc[8], sh[8];
  {
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
    if (a.scale) { s0 = *reinterpret_cast<const f32x4*>(a.scale + ch); s1 = *reinterpret_cast<const f32x4*>(a.scale + ch + 4); }
    if (a.shift) { h0 = *reinterpret_cast<const f32x4*>(a.shift + ch); h1 = *reinterpret_cast<const f32x4*>(a.shift + ch + 4); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sc[i] = s0[i]; sc[4 + i] = s1[i]; sh[i] = h0[i]; sh[4 + i] = h1[i]; }
  }
  __syncthreads();
  stamp(4);
  // wave (kp, ph) finishes the tiles ph * PXT + j with j % NKW == kp
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, a.wt ? (int)((size_t)a.B * HW * a.Cout * 2) : 0, 0x00020000);
#pragma unroll
  for (int k = 0; k < NOWN; ++k) {
    const int j = kp + k * NKW;
    if (j >= PXT) break;  // wave-uniform
    const int tile = ph * PXT + j;
    const int pl = tile * 16 + n;
    const bool live = pl < npx;
    const size_t o = ((size_t)b * HW + p0 + (live ? pl : 0)) * a.Cout + ch;
    const uint4 rv = rres[k];
    float v[8];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 sum = part[((0 * NT + tile) * 2 + ct) * 64 + lane];
#pragma unroll
      for (int k = 1; k < NKW; ++k) {
        const f32x4 pv = part[((k * NT + tile) * 2 + ct) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] += pv[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * ct + i] = sum[i];
    }
    const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = v[2 * k] * sc[2 * k] + sh[2 * k] + lss_bf2f((unsigned short)(ru[k] & 0xffff));
      float hi = v[2 * k + 1] * sc[2 * k + 1] + sh[2 * k + 1] + lss_bf2f((unsigned short)(ru[k] >> 16));
      if (a.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      v[2 * k] = lo; v[2 * k + 1] = hi;
    }
    if (live) {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const u32x4 ov = {lss_pack_bf2(v[0], v[1]), lss_pack_bf2(v[2], v[3]), lss_pack_bf2(v[4], v[5]), lss_pack_bf2(v[6], v[7])};
      if (a.wt) __builtin_amdgcn_raw_buffer_store_b128(ov, yrsrc, (int)(o * 2), 0, 16);  // write-through
      else *reinterpret_cast<u32x4*>(a.y + o) = ov;
    }
  }
  if (a.stamps != nullptr) {
    stamp(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores acknowledged
    stamp(6);
  }
}

// OIHW fp32 -> the kernel's register image, bf16: [co block of 32][K part][k-step][channel tile][lane][8], where k-step
// S = kp * KSW + s covers chunk S / 9 (32 input channels) of tap S % 9, A-fragment lane (kq = lane >> 4, m = lane & 15)
// holds W[co = cb * 32 + 8 (m >> 2) + 4 ct + (m & 3)][ci = 32 chunk + 8 kq .. + 8][tap]
// dgrad = 1: the image of the TRANSPOSED, tap-flipped weights (the input-gradient conv: Cout x Cin here are the
// gradient conv's own output / input channels, w is the forward layer's (Cin, Cout, 3, 3))
__global__ void pack_weights_ks_kernel(const float* __restrict__ w, int Cout, int Cin, unsigned short* __restrict__ out,
                                       int dgrad) {
  const size_t ntot = (size_t)Cout * Cin * 9;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < ntot; e += (size_t)gridDim.x * 256) {
    const int j = e & 7;
    size_t r = e >> 3;
    const int l = r & 63; r >>= 6;
    const int ct = r & 1; r >>= 1;
    const int nks = (Cin >> 5) * 9;        // k-steps of the whole K
    const int S = r % nks;
    const int cb = r / nks;
    const int m = l & 15, kq = l >> 4;
    const int co = cb * 32 + 8 * (m >> 2) + 4 * ct + (m & 3);
    const int ci = (S / 9) * 32 + kq * 8 + j;
    out[e] = lss_f2bf(dgrad ? w[((size_t)ci * Cout + co) * 9 + (8 - S % 9)] : w[((size_t)co * Cin + ci) * 9 + (S % 9)]);
  }
}

struct KsPlan {
  int ok, PB, npb, nposp, ncb, lds, grid, variant;  // variant 0: <18, 4, 5> (Cin 256), 1: <9, 4, 10> (128), 2: <9, 2, 10> (64)
};

KsPlan ks_plan(int B, int H, int W, int Cin, int Cout) {
  KsPlan p = {};
  if (B <= 0 || H <= 0 || W < 4 || Cout <= 0 || Cout % 32 != 0) return p;
  if (Cin == 256) { p.variant = 0; p.PB = 80; }
  else if (Cin == 128) { p.variant = 1; p.PB = 160; }
  else if (Cin == 64) { p.variant = 2; p.PB = 320; }
  else return p;
  const long long HW = (long long)H * W;
  if (p.PB > 4 * W + 1) return p;                    // a pixel block spans at most five image rows
  p.npb = (int)((HW + p.PB - 1) / p.PB);
  p.ncb = Cout / 32;
  p.nposp = (KS_ROWS * (W + 2) + 31) / 32 * 32;
  const int patch = (Cin / 32) * p.nposp * KS_POSB;
  if (patch > 112 * 1024) return p;                  // (with the 80-KiB partial tiles of the K parts in the same LDS)
  const int ntile = p.PB / 16;
  const int red = (p.variant == 2 ? 2 : 4) * ntile * 2 * 1024;   // the K parts' partial tiles
  p.lds = patch > red ? patch : red;
  if (p.lds > KS_LDS_MAX) return p;
  const long long grid = (long long)B * p.npb * p.ncb;
  // one workgroup per CU: worth it where the tile kernel's grid leaves the chip under-filled (at most two rounds here)
  if (grid < 64 || grid > 512) return p;
  if ((long long)B * HW * (Cin > Cout ? Cin : Cout) >= (1LL << 30)) return p;
  p.grid = (int)grid;
  p.ok = 1;
  return p;
}

template <int KSW, int NKW, int PXT>
int ks_launch(const KsPlan& p, const KsArgs& a, hipStream_t st) {
  static bool attr_set[64] = {};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ks_kernel<KSW, NKW, PXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, KS_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_ks_kernel<KSW, NKW, PXT>), dim3(p.grid), dim3(256), p.lds, st, a);
  return lss_launch_status();
}

}  // namespace

// Is (shape) a case for the K-split one-pass kernel?  3x3 / stride 1 / pad 1, bf16, Cin in {64, 128, 256}, Cout a
// multiple of 32, an image narrow enough for a pixel block to span five rows, and a grid of 64-512 workgroups.
extern "C" int lss_conv2d_ks_ok(int B, int H, int W, int Cin, int Cout) {
  if (const char* e = getenv("LSS_CONV_KS"))
    if (atoi(e) == 0) return 0;
  return ks_plan(B, H, W, Cin, Cout).ok;
}

extern "C" size_t lss_conv2d_ks_packed_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cout % 32 != 0 || (Cin != 64 && Cin != 128 && Cin != 256)) return 0;
  return (size_t)Cout * Cin * 9 * 2;
}

extern "C" int lss_conv2d_pack_weights_ks(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  if (lss_conv2d_ks_packed_weight_bytes(Cout, Cin) == 0) return LSS_E_SHAPE;
  const size_t n = (size_t)Cout * Cin * 9;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_ks_kernel, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cout, Cin,
                     reinterpret_cast<unsigned short*>(w_packed), 0);
  return lss_launch_status();
}

// weights of the input-gradient conv of a 3x3 / stride-1 layer, K-split image: w_oihw is the FORWARD layer's
// (Cout, Cin, 3, 3); the gradient conv maps Cout -> Cin channels with the taps flipped
extern "C" int lss_conv2d_pack_weights_ks_dgrad(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream) {
  LSS_CHECK_PTR(w_oihw); LSS_CHECK_PTR(w_packed);
  if (lss_conv2d_ks_packed_weight_bytes(Cin, Cout) == 0) return LSS_E_SHAPE;
  const size_t n = (size_t)Cout * Cin * 9;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_weights_ks_kernel, dim3(grid), dim3(256), 0, lss_stream(stream), w_oihw, Cin, Cout,
                     reinterpret_cast<unsigned short*>(w_packed), 1);
  return lss_launch_status();
}

// launcher behind lss_conv2d_fwd when the weights are KS-packed (LSS_W_KS)
int lss_conv_ks_launch(const void* x, const void* w_ks, const float* scale, const float* shift, const void* residual,
                       void* y, int B, int H, int W, int Cin, int Cout, int relu, int wt, hipStream_t st) {
  const KsPlan p = ks_plan(B, H, W, Cin, Cout);
  if (!p.ok) return LSS_E_SHAPE;
  if (relu != 0 && relu != 1) return LSS_E_LAYOUT;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_ks) |
        reinterpret_cast<uintptr_t>(residual)) & 15) != 0)
    return LSS_E_ALIGN;
  KsArgs a;
  a.x = reinterpret_cast<const unsigned short*>(x);
  a.w = reinterpret_cast<const unsigned char*>(w_ks);
  a.scale = scale; a.shift = shift;
  a.residual = reinterpret_cast<const unsigned short*>(residual);
  a.y = reinterpret_cast<unsigned short*>(y);
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu; a.wt = wt;
  a.PB = p.PB; a.npb = p.npb; a.nposp = p.nposp; a.ncb = p.ncb;
  {
    const char* e = getenv("LSS_KS_STAMPS");
    a.stamps = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16)) : nullptr;
  }
  if (p.variant == 0) return ks_launch<18, 4, 5>(p, a, st);
  if (p.variant == 1) return ks_launch<9, 4, 10>(p, a, st);
  return ks_launch<9, 2, 10>(p, a, st);
}
