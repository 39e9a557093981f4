// 1x1 / stride-1 convolution on NHWC bf16 = a row-major GEMM
//     y[m, n] = act(scale[n] * sum_k x[m, k] w[n, k] + shift[n] + residual[m, n])
// over m = B*H*W pixel (token) rows.  This is the shape of every linear layer of
// the BEV transformer (ref: src/transformer_modules.py:77-84, 170-172) and of the
// 1x1 convs around it (ref: src/model_vovnet_transformer.py:131-135, 150).
//
// Both operands are K-contiguous, so both are staged by LDS-DMA (no register
// hop) into a 3-slot ring of 32-channel slabs:
//   workgroup 256 threads = 128 rows x 128 columns, wave = 64 x 64 = 2 x 2 tiles
//   of v_mfma_f32_32x32x16_bf16; one ring slot = 128 x 64 B of x + 128 x 64 B of
//   w = 16 KiB = 16 DMA blocks of 1 KiB, 4 per wave; slabs are fetched three steps
//   ahead and the step barrier is a raw s_barrier behind a COUNTED vmcnt.
//   64-B rows are XOR-swizzled in 16-B pieces by ((row >> 2) & 3) - on the DMA's
//   source address (its destination is lane-linear) and on the read address -
//   which makes every 16-lane group of a ds_read_b128 hit 64 distinct banks.
//   A slot is drained into registers one step before its barrier, so the slab three steps ahead
//   is issued into the slot just consumed: 3 slots, 3 steps of prefetch.  LDS = ring 48 KiB (the fp32
//   output tile is staged in two 64-row halves of 33 KiB) -> 3 workgroups per CU.
// Epilogue as in conv_mfma.hip: scale/shift in registers -> fp32 tile in LDS ->
// 8 consecutive channels per thread, residual / activation / one 16-B (bf16) or
// 32-B (fp32) store.
#include "lss_common.h"

namespace {

struct LinearArgs {
  const unsigned short* x;  // (M, K) bf16
  const unsigned short* w;  // (N, K) bf16
  const float* scale;       // (N) or null
  const float* shift;       // (N) or null
  const unsigned short* residual;  // (M, N) bf16 or null
  void* y;                  // (M, N) bf16 or fp32
  int M, N, K;
  int act;      // 0 none, 1 ReLU, 2 GELU
  int out_f32;
  int group_hw;  // > 0: head-major output (B, N/32, group_hw, 32), rows m = b*group_hw + pixel
  int mtiles, ntiles;
  long long ldx, ldw;  // row strides of x and w in elements (= K for the plain GEMM)
  // weight-gradient mode (conv_grad.hip): blockIdx.y = split * ntap + tap.  x = dY^T (Cout rows),
  // w = one of three column-shifted channel-major copies of the conv input (Cin rows); K = pixels.
  int ntap;            // 0: plain GEMM
  long long split_k;   // K elements per split
  long long tap_row;   // padded image row length: tap (ky, kx) reads copy kx at offset (ky-1)*tap_row
  const unsigned short* wk[3];
};

constexpr int BM = 128, BN = 128, BK = 32, NS = 3;
constexpr int ROWB = BK * 2;                 // bytes per staged row
constexpr int OP_BYTES = BM * ROWB;          // 8 KiB per operand and slot
constexpr int SLOT_BYTES = 2 * OP_BYTES;     // 16 KiB
constexpr int OLD = BN + 4;                  // fp32 output tile row (padded)
constexpr int RING_BYTES = NS * SLOT_BYTES;
constexpr int OUT_BYTES = (BM / 2) * OLD * 4;         // the output tile is staged in two 64-row halves
constexpr int SMEM_BYTES = RING_BYTES > OUT_BYTES ? RING_BYTES : OUT_BYTES;
static_assert(SMEM_BYTES <= 53 * 1024, "three workgroups per CU");

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp): libm's erff
// costs ~40 VALU instructions per element, which on a 1024-wide GELU layer is as much
// time as the GEMM itself.  (The fp32 parity path - conv_direct_kernel - keeps erff.)
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));  // v_rcp_f32 (1 ulp); __frcp_rn is a 12-instruction IEEE division
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = 1.f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 0.5f * v * (1.f + fast_erf(v * 0.70710678118654752f));
  return v;
}

__global__ __launch_bounds__(256, 3) void linear_mfma_kernel(LinearArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware order: XCD k works on the k-th contiguous eighth of the tile list, column
  // tiles of one row tile adjacent, so a row tile's x rows are fetched into ONE L2
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int n0 = (t % a.ntiles) * BN, m0 = (t / a.ntiles) * BM;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned short* xbase = a.x;
  const unsigned short* wbase = a.w;
  float* ybatch = nullptr;
  int nsteps = a.K / BK;
  if (a.ntap > 0) {
    const int z = blockIdx.y, tap = z % a.ntap, split = z / a.ntap;
    xbase = a.x + split * a.split_k;
    wbase = a.wk[tap % 3] + (tap / 3 - 1) * a.tap_row + split * a.split_k;
    ybatch = reinterpret_cast<float*>(a.y) + (size_t)z * a.M * a.N;
    nsteps = (int)(a.split_k / BK);
  }

  // DMA: waves 0,1 stage x (blocks 0..7 of 16 rows), waves 2,3 stage w
  const bool is_w = wave >= 2;
  const unsigned short* gbase = is_w ? wbase : xbase;
  const long long ld = is_w ? a.ldw : a.ldx;
  const int glim = (is_w ? a.N : a.M) - 1;
  const int g0 = is_w ? n0 : m0;
  size_t goff[4];  // element offset of this lane's 16-B piece at k = 0, per block
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = (wave & 1) * 4 + i;
    const int row = blk * 16 + (lane >> 2);
    const int piece = (lane & 3) ^ ((row >> 2) & 3);
    goff[i] = (size_t)min(g0 + row, glim) * ld + piece * 8;  // rows past the edge re-read the last row
  }
  const int dma_off = (is_w ? OP_BYTES : 0) + (wave & 1) * 4 * 1024;
  auto issue = [&](int step, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(gbase + goff[i] + (size_t)step * BK, smem + slot * SLOT_BYTES + dma_off + i * 1024);
  };

  // fragment read offsets within a slot: k-step s of lane half h reads 16-B piece 2h + s
  int aoff[2][2], boff[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + r, rb = wc * 64 + i * 32 + r;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      aoff[i][s] = ra * ROWB + (((2 * h + s) ^ ((ra >> 2) & 3)) << 4);
      boff[i][s] = OP_BYTES + rb * ROWB + (((2 * h + s) ^ ((rb >> 2) & 3)) << 4);
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: slabs 0..2 on their way (the whole ring); slab 0 must have landed
  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  if (nsteps > 2) issue(2, 2);
  if (nsteps > 2) wait_vmcnt<8>();
  else if (nsteps > 1) wait_vmcnt<4>();
  else wait_vmcnt<0>();
  lds_barrier();

  bf16x8 fa[2][2], fb[2][2];  // [k-step][tile]
  auto read = [&](int s, int slot) {
    const unsigned char* base = smem + slot * SLOT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[s][i] = *reinterpret_cast<const bf16x8*>(base + aoff[i][s]);
      fb[s][i] = *reinterpret_cast<const bf16x8*>(base + boff[i][s]);
    }
  };
  int slot = 0;
  read(0, 0);
  for (int step = 0; step < nsteps; ++step) {
    read(1, slot);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
    // slab step+1 must have landed in every wave's share before anyone reads it; the younger slab
    // (step+2) stays in flight across the barrier
    if (step + 2 < nsteps) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    lds_barrier();  // (also: every fragment of THIS slot is in registers now, in every wave)
    // ... so the slot just consumed is free: the slab three steps ahead goes into it (a 3-slot
    // ring that runs 3 steps ahead)
    if (step + 3 < nsteps) issue(step + 3, slot);
    slot = slot == 2 ? 0 : slot + 1;
    if (step + 1 < nsteps) read(0, slot);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
  }
  lds_barrier();  // ring no longer read: reuse it as the fp32 output tile

  // D[row = (e&3) + 8*(e>>2) + 4*h][col = r]: a lane holds one column of 16 rows.  The fp32 tile
  // is staged in two 64-row halves (waves wr = 0, then wr = 1) so it fits the ring's 48 KiB.
  float* otile = reinterpret_cast<float*>(smem);
  unsigned short* yb = reinterpret_cast<unsigned short*>(a.y);
  float* yf = ybatch ? ybatch : reinterpret_cast<float*>(a.y);
  const bool vec_ok = (a.N & 7) == 0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cl = wc * 64 + j * 32 + r;
        const int co = n0 + cl;
        const bool cok = co < a.N;
        const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
        const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // row within the half
            otile[row * OLD + cl] = acc[i][j][e] * sc + sh;
          }
      }
    }
    lds_barrier();
    for (int e = tid; e < (BM / 2) * (BN / 8); e += 256) {
      const int row = e / (BN / 8), c8 = e % (BN / 8);
      const int m = m0 + half * (BM / 2) + row, co = n0 + c8 * 8;
      if (m >= a.M || co >= a.N) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(otile + row * OLD + c8 * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(otile + row * OLD + c8 * 8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const size_t o = (size_t)m * a.N + co;  // residual (and row-major output) index
      size_t oy = o;
      if (a.group_hw > 0) {
        const int bidx = m / a.group_hw, pix = m - bidx * a.group_hw;
        oy = (((size_t)bidx * (a.N >> 5) + (co >> 5)) * a.group_hw + pix) * 32 + (co & 31);
      }
      if (vec_ok && co + 8 <= a.N) {
        if (a.residual) {
          const uint4 rv = *reinterpret_cast<const uint4*>(a.residual + o);
          const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += lss_bf2f((unsigned short)(ru[k] & 0xffff));
            v[2 * k + 1] += lss_bf2f((unsigned short)(ru[k] >> 16));
          }
        }
        if (a.act) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = act_fn(v[k], a.act);
        }
        if (a.out_f32) {
          *reinterpret_cast<f32x4*>(yf + oy) = (f32x4){v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(yf + oy + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
          uint4 ov;
          ov.x = lss_pack_bf2(v[0], v[1]); ov.y = lss_pack_bf2(v[2], v[3]);
          ov.z = lss_pack_bf2(v[4], v[5]); ov.w = lss_pack_bf2(v[6], v[7]);
          *reinterpret_cast<uint4*>(yb + oy) = ov;
        }
      } else {
        for (int k = 0; k < 8 && co + k < a.N; ++k) {
          float tv = v[k];
          if (a.residual) tv += lss_bf2f(a.residual[o + k]);
          tv = act_fn(tv, a.act);
          if (a.out_f32) yf[oy + k] = tv;
          else yb[oy + k] = lss_f2bf(tv);
        }
      }
    }
    if (half == 0) lds_barrier();  // everyone has read half 0 before half 1 is staged
  }
}

}  // namespace

// Internal entry used by lss_conv2d_fwd's 1x1 dispatch (conv_mfma.hip); not part of the C ABI.
int lss_linear_bf16_launch(const void* x, const void* w, const float* scale, const float* shift,
                           const void* residual, void* y, long long M, int N, int K, int act,
                           int out_f32, int group_hw, hipStream_t st) {
  if (M <= 0 || M >= (1LL << 31) || N <= 0 || K <= 0 || K % BK != 0) return LSS_E_SHAPE;
  if (group_hw > 0 && (N % 32 != 0 || M % group_hw != 0)) return LSS_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return LSS_E_ALIGN;
  LinearArgs a;
  a.x = static_cast<const unsigned short*>(x);
  a.w = static_cast<const unsigned short*>(w);
  a.scale = scale; a.shift = shift;
  a.residual = static_cast<const unsigned short*>(residual);
  a.y = y;
  a.M = (int)M; a.N = N; a.K = K; a.act = act; a.out_f32 = out_f32; a.group_hw = group_hw;
  a.ldx = K; a.ldw = K; a.ntap = 0; a.split_k = 0; a.tap_row = 0; a.wk[0] = a.wk[1] = a.wk[2] = nullptr;
  a.mtiles = lss_cdiv(M, BM);
  a.ntiles = lss_cdiv(N, BN);
  const long long nwg = (long long)a.mtiles * a.ntiles;
  if (nwg >= (1LL << 31)) return LSS_E_SHAPE;
  hipLaunchKernelGGL(linear_mfma_kernel, dim3((unsigned)nwg), dim3(256), 0, st, a);
  return lss_launch_status();
}

// Internal entry used by lss_conv2d_wgrad (conv_grad.hip): partial[split][tap][M][N] fp32 =
// sum over the split's K range of dyt[m][k] * xt_kx[n][k + (ky-1)*tap_row].
int lss_wgrad_gemm_launch(const void* dyt, const void* const xt[3], float* partial, int M, int N,
                          long long ld, long long split_k, int nsplit, int ntap, long long tap_row,
                          hipStream_t st) {
  if (M <= 0 || N <= 0 || split_k <= 0 || split_k % BK != 0 || nsplit <= 0 || ntap <= 0) return LSS_E_SHAPE;
  LinearArgs a;
  a.x = static_cast<const unsigned short*>(dyt);
  a.w = nullptr;
  a.scale = nullptr; a.shift = nullptr; a.residual = nullptr;
  a.y = partial;
  a.M = M; a.N = N; a.K = (int)split_k; a.act = 0; a.out_f32 = 1; a.group_hw = 0;
  a.mtiles = lss_cdiv(M, BM);
  a.ntiles = lss_cdiv(N, BN);
  a.ldx = ld; a.ldw = ld; a.ntap = ntap; a.split_k = split_k; a.tap_row = tap_row;
  for (int i = 0; i < 3; ++i) a.wk[i] = static_cast<const unsigned short*>(xt[i]);
  const long long gy = (long long)nsplit * ntap;
  if (gy > 65535) return LSS_E_SHAPE;
  hipLaunchKernelGGL(linear_mfma_kernel, dim3((unsigned)(a.mtiles * a.ntiles), (unsigned)gy), dim3(256), 0, st, a);
  return lss_launch_status();
}
