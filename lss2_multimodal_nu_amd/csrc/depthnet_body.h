// K2's fp32 kernel body and the shared epilogue, as device functions: included by depthnet.hip
// (the stand-alone kernels) and by geom_bucket.hip (the fused K2 || K3 launch).
#pragma once
#include "lss_common.h"

namespace lss_depthnet {

constexpr int PIX = 16;         // pixels per workgroup
constexpr int LDS_LD = PIX + 1; // padded row of the [n][pix] logits tile

// Shared tail of both kernels: K-quarter partials -> LDS -> bias -> outputs.
template <int NT>
__device__ __forceinline__ void depthnet_epilogue(const f32x4 (&acc)[NT], float* lds,
                                                  const float* __restrict__ bias_d,
                                                  const float* __restrict__ bias_c, int feat_row0,
                                                  bool softmax, int bn, int pix0, int HW, int D,
                                                  int C, float* __restrict__ depth,
                                                  float* __restrict__ feat, float* wg_absmax = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  // logit rows: depth bins at [0, D), context channels at [feat_row0, feat_row0 + C)
  const int NO = feat_row0 + C;
  // biases of the logits this thread finishes below: requested now, so the loads fly during the partial
  // stores and the barrier instead of one exposed round trip per pass of the reduce loop
  float bias_v[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = (tid + i * 256) / PIX;
    bias_v[i] = 0.f;
    if (n < D) bias_v[i] = bias_d[n];
    else if (n >= feat_row0 && n < NO) bias_v[i] = bias_c[n - feat_row0];
  }
  // partial[wave][n][pix] -> LDS
  float* part = lds;  // [4][NT*16][LDS_LD]
  const int NR = NT * 16;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      part[((size_t)wave * NR + 16 * t + 4 * j + r) * LDS_LD + col] = acc[t][r];
  __syncthreads();
  // reduce the four K quarters (fixed order -> deterministic), add bias
  float* logit = lds + 4 * NR * LDS_LD;  // [NR][LDS_LD]
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int e = tid + i * 256;
    if (e < NO * PIX) {
      const int n = e / PIX, p = e % PIX;
      const int o = n * LDS_LD + p;
      float v = part[o] + part[NR * LDS_LD + o];
      v += part[2 * NR * LDS_LD + o];
      v += part[3 * NR * LDS_LD + o];
      logit[o] = v + bias_v[i];
    }
  }
  __syncthreads();

  // context features: feat[(bn*HW + pix)*C + c] = logit[D + c][pix]
  float amax = 0.f;
  for (int e = tid; e < PIX * C; e += 256) {
    const int p = e / C, c = e % C;
    if (pix0 + p < HW) {
      const float v = logit[(feat_row0 + c) * LDS_LD + p];
      feat[((size_t)bn * HW + pix0 + p) * C + c] = v;
      // max over the FINITE features only: a NaN or an inf feature must not size the splat's fixed-point scale (an inf
      // maximum used to fall back to scale 2^40, and finite products >= 2^11 of the same call then overflowed the
      // magic-number conversion silently - VERDICT r2); the non-finite products themselves are flagged in the splat
      const float av = fabsf(v);
      amax = fmaxf(amax, av <= 3.0e38f ? av : 0.f);
    }
  }
  if (wg_absmax != nullptr) {
    // max |feature| of this workgroup (the region splat sizes its fixed-point accumulator from the maximum over all
    // workgroups): one plain store per workgroup into its own slot - no atomics, nothing to reset between calls.
    // `part` is free again (every thread passed the barrier above after its last read of it).
    amax = lss_wave_max(amax);
    if (lane == 0) part[wave] = amax;
    __syncthreads();
    if (tid == 0) *wg_absmax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    __syncthreads();
  }
  if (!softmax) {  // raw logits (a later kernel fuses them, ref MultiScaleDepthNet)
    for (int e = tid; e < D * PIX; e += 256) {
      const int d = e / PIX, p = e % PIX;
      if (pix0 + p < HW) depth[((size_t)bn * D + d) * HW + pix0 + p] = logit[d * LDS_LD + p];
    }
    return;
  }
  // softmax over d for each of the 16 pixels: 16 lanes per pixel
  {
    const int p = tid & 15, part_id = tid >> 4;  // 16 parts
    float m = -INFINITY;
    for (int d = part_id; d < D; d += 16) m = fmaxf(m, logit[d * LDS_LD + p]);
    // combine the 16 parts of a pixel: lanes p, p+16, p+32, p+48 of 4 waves -> LDS
    float* red = part;  // reuse: [16 parts][16 pix]
    red[part_id * PIX + p] = m;
    __syncthreads();
    float mx = red[p];
#pragma unroll
    for (int q = 1; q < 16; ++q) mx = fmaxf(mx, red[q * PIX + p]);
    __syncthreads();
    float s = 0.f;
    for (int d = part_id; d < D; d += 16) {
      const float e = expf(logit[d * LDS_LD + p] - mx);
      logit[d * LDS_LD + p] = e;
      s += e;
    }
    red[part_id * PIX + p] = s;
    __syncthreads();
    float sum = red[p];
#pragma unroll
    for (int q = 1; q < 16; ++q) sum += red[q * PIX + p];
    if (pix0 + p < HW)
      for (int d = part_id; d < D; d += 16)
        depth[((size_t)bn * D + d) * HW + pix0 + p] = logit[d * LDS_LD + p] / sum;
  }
}

// MFMA operand maps (cdna_hip_programming.md section 3):
//   16x16x4 f32:  A[row = l&15][k = l>>4],  B[k = l>>4][col = l&15],
//                 D[row = 4*(l>>4) + r][col = l&15], r = 0..3
// Within a 16-deep K block the four k-steps s = 0..3 use k = kb + 4*(l>>4) + s,
// so one 16-B load of W[n][kb + 4*(l>>4) ..+3] feeds four MFMAs.
template <int NT>
__device__ __forceinline__ void depthnet_softmax_f32_body(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    int Cin, int HW, int D, int C, float* __restrict__ depth, float* __restrict__ feat, int tile_x, int bn,
    float* lds, float* wg_absmax = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int pix0 = tile_x * PIX;
  const int NO = D + C;
  const int pix = min(pix0 + col, HW - 1);
  const int kq = Cin >> 2;  // K quarter of this wave
  const float* xb = x + ((size_t)bn * Cin + (size_t)wave * kq) * HW + pix;
  const float* wb = w + (size_t)wave * kq + 4 * j;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Two register sets, software-pipelined: the 4 x loads and NT weight loads of the NEXT 16-deep K
  // block are in flight while the 4*NT MFMAs of this one issue, so block-start round trips are no
  // longer exposed.  In-kernel stamps put what remains of the ~13 us K loop at ~5 us of MFMA issue,
  // ~6 us of weight re-reads through the CU's L1 (every workgroup streams all 215 KB, half a cache
  // line at a time) and ~1.5 us of x loads.  Per accumulator the k order is the plain loop's:
  // bitwise the same sums.
  auto load_block = [&](int kb, float (&xs)[4], f32x4 (&wa)[NT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xs[s] = xb[(size_t)(kb + 4 * j + s) * HW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + col;
      wa[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n < NO) wa[t] = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb);
    }
  };
  auto mma_block = [&](const float (&xs)[4], const f32x4 (&wa)[NT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)  // s outer: consecutive MFMAs hit different accumulators (a dependent
#pragma unroll                   // 16x16x4 f32 pair costs 40 cycles, an independent one 32); per
      for (int t = 0; t < NT; ++t)  // accumulator the k order is unchanged
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][s], xs[s], acc[t], 0, 0, 0);
  };
  const int nblk = kq >> 4;
  float xs0[4], xs1[4];
  f32x4 wa0[NT], wa1[NT];
  load_block(0, xs0, wa0);
  int i = 0;
  for (; i + 2 <= nblk; i += 2) {
    load_block((i + 1) << 4, xs1, wa1);
    mma_block(xs0, wa0);
    if (i + 2 < nblk) load_block((i + 2) << 4, xs0, wa0);
    mma_block(xs1, wa1);
  }
  if (i < nblk) mma_block(xs0, wa0);  // odd number of blocks: the last one is already loaded

  depthnet_epilogue<NT>(acc, lds, bias, bias + D, D, true, bn, pix0, HW, D, C, depth, feat, wg_absmax);
}

// The same product with the OUTPUT ROWS split over two workgroups per pixel tile (the region pipeline's K2):
// `rows` logit rows starting at weight row `row0` - the D depth bins (softmax, -> depth) or the C context
// channels (-> feat, max |feature|).  Against the body above: (i) a workgroup streams only its own rows of W
// (105 or 110 KB instead of 215 KB through one CU's L1), twice as many workgroups share the chip (2 per CU) and
// the softmax runs beside the feature stores instead of after them; (ii) a K block is 32 deep: a lane loads 32
// contiguous bytes of its weight row, the four lane groups of a row cover one whole 128-B line per load, where the
// 16-deep block fetched every line in two halves 7 tile-loads apart (a 57 KB working set per block against the
// 32 KB L1: most lines came from L2 twice).  Per accumulator the k order is (kb + 8 j + s), j = lane group, s = 0..7:
// fixed, so results are reproducible; they differ from the 16-deep body's in the last ulp (different association).
template <int NT>
__device__ __forceinline__ void depthnet_rows_f32_body(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int row0, int rows,
    bool is_depth, int Cin, int HW, int D, int C, float* __restrict__ depth, float* __restrict__ feat, int tile_x,
    int bn, float* lds, float* wg_absmax, unsigned long long* stamps = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, j = lane >> 4;
  const int pix0 = tile_x * PIX;
  const int pix = min(pix0 + col, HW - 1);
  const int kq = Cin >> 2;  // K quarter of this wave
  const float* xb = x + ((size_t)bn * Cin + (size_t)wave * kq + 8 * j) * HW + pix;
  const float* wb = w + (size_t)row0 * Cin + (size_t)wave * kq + 8 * j;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_block = [&](int kb, float (&xs)[8], f32x4 (&wa)[NT][2]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) xs[s] = xb[(size_t)(kb + s) * HW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + col;
      wa[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      wa[t][1] = wa[t][0];
      if (n < rows) {
        wa[t][0] = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb);
        wa[t][1] = *reinterpret_cast<const f32x4*>(wb + (size_t)n * Cin + kb + 4);
      }
    }
  };
  auto mma_block = [&](const float (&xs)[8], const f32x4 (&wa)[NT][2]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][s >> 2][s & 3], xs[s], acc[t], 0, 0, 0);
  };
  const int nblk = kq >> 5;
  // In-kernel stamps (tools/bench_l1.py --stamps): a workgroup is ~2.3 us to its first operands, ~1.5-2 us per further
  // K block (the 528 workgroups together pull ~75 MB through the L1s in ~8 us: the launch sits at the L2 -> L1 rate
  // this access pattern reaches, ~9 TB/s), 2.5 us of epilogue.  Requesting all four K blocks of the LSS depthnet's K
  // quarter before the first MFMA shortens a workgroup by ~2 us but needs 180 registers = 2 waves per SIMD = 512
  // workgroup slots for 528 workgroups: the 16 left over start when a first-round workgroup retires, and the launch
  // ends later than before (level 48.9 -> 51.1 us).
  {
    // two operand sets: the next K block is in flight behind the one being multiplied.  (A third set - two blocks in
    // flight - does not fit the 168-register budget of 3 waves per SIMD: 178 spilled registers.)
    float xs0[8], xs1[8];
    f32x4 wa0[NT][2], wa1[NT][2];
    load_block(0, xs0, wa0);
    if (stamps != nullptr) {  // diagnostic runs only: first operands landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
    int i = 0;
    for (; i + 2 <= nblk; i += 2) {
      load_block((i + 1) << 5, xs1, wa1);
      mma_block(xs0, wa0);
      if (i + 2 < nblk) load_block((i + 2) << 5, xs0, wa0);
      mma_block(xs1, wa1);
    }
    if (i < nblk) mma_block(xs0, wa0);
  }
  if (stamps != nullptr && threadIdx.x == 0) {
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[NT - 1][3]));  // the MFMA chain has retired
    stamps[2] = __builtin_amdgcn_s_memrealtime();
  }

  if (is_depth)
    depthnet_epilogue<NT>(acc, lds, bias, bias, NT * 16, true, bn, pix0, HW, D, 0, depth, feat, nullptr);
  else
    depthnet_epilogue<NT>(acc, lds, bias, bias + D, 0, false, bn, pix0, HW, 0, C, depth, feat, wg_absmax);
}

}  // namespace lss_depthnet
