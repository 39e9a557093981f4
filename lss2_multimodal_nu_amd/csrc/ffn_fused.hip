// Position-wise feed-forward block of the BEV transformer layer as ONE kernel
//     y[m, :] = x[m, :] + b2 + W2 . gelu(W1 . x[m, :] + b1)        (fp32 out, pre-LayerNorm sum)
// ref: src/transformer_modules.py:170-172 (linear1, activation, linear2) and the residual add
// of :208 (`src + dropout2(ff)`); d_model = 256, dim_feedforward <= 1024, bf16 operands.
//
// As two GEMM launches the (M, 1024) hidden activation makes a round trip through HBM: 655 MB
// written and read again per step at 40 000 x 8 tokens, more than everything else the layer
// moves.  Here a workgroup owns 128 tokens and walks the hidden units in chunks of 64:
//   GEMM1 (transposed product)  Ht[64 hidden, 128 tokens] = W1_j[64, 256] . X^T
//       A = W1_j rows from LDS, B = the tokens' x rows, held in REGISTERS for the whole kernel
//       (16 k-steps x 4 VGPRs); D[row = hidden][col = token] puts 4 consecutive hidden units of
//       one token in one lane, so gelu(.) -> bf16 goes to the H tile as 8-B LDS stores;
//   GEMM2                       Y[128 tokens, 256] += H_j[128, 64] . W2_j[256, 64]^T
//       A = H_j from LDS, B = W2_j from LDS, accumulators (4 tiles of 32 x 32 per wave) stay in
//       registers across all chunks.
// 512 threads = 8 waves = 4 token row groups x 2 column groups.  W1_j / W2_j stream by LDS-DMA into
// double buffers one chunk ahead (2 x 64 KiB), H_j is single-buffered (16 KiB): barrier (a) of a
// chunk = "its weights have landed and every wave is done with the previous chunk", barrier (b) =
// "H_j is written".  LDS images: W1 rows are 512 B with their 16-B pieces XOR-swizzled by (row & 15),
// W2 / H rows are 128 B swizzled by ((row >> 1) & 7) - on the DMA source address (the DMA
// destination is lane-linear) and on every read / write address - conflict-free ds_read_b128.
// The bf16 rounding points (x, H, the weights) and the k order of both sums are those of the
// two-launch path, so the two agree to fp32 rounding.
#include "lss_common.h"

namespace {

constexpr int D = 256, HC = 64, BM = 128, FMAX = 1024;
constexpr int W1_BYTES = HC * D * 2;   // 32 KiB: 64 rows x 512 B
constexpr int W2_BYTES = D * HC * 2;   // 32 KiB: 256 rows x 128 B
constexpr int H_BYTES = BM * HC * 2;   // 16 KiB: 128 rows x 128 B
constexpr int OFF_W1 = 0, OFF_W2 = 2 * W1_BYTES, OFF_H = OFF_W2 + 2 * W2_BYTES, OFF_B1 = OFF_H + H_BYTES;
constexpr int SMEM_BYTES = OFF_B1 + FMAX * 4;  // 148 KiB
constexpr int OLD = D + 4;                     // fp32 row of the staged output tile
static_assert(BM * OLD * 4 <= OFF_B1, "the output tile is staged over the (finished) weight buffers");
static_assert(SMEM_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(OFF_W1 == 0 && OFF_W2 == 2 * W1_BYTES && W1_BYTES == W2_BYTES && (D / HC) * W2_BYTES <= OFF_H,
              "PROJ mode lays its four weight chunks over the W1 + W2 buffers");

struct FfnArgs {
  const unsigned short* x;   // (M, 256) bf16
  const unsigned short* w1;  // (F, 256) bf16
  const float* b1;           // (F)
  const unsigned short* w2;  // (256, F) bf16
  const float* b2;           // (256)
  float* y;                  // (M, 256) fp32, or null with the LayerNorm tail
  const unsigned short* res; // (M, 256) bf16 residual rows added in the epilogue (the FFN: x itself)
  int M, F;
  // optional LayerNorm tail (ref: src/transformer_modules.py:208 norm2): y_ln = LN(y) * gamma + beta, bf16
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  unsigned short* y_ln;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), as in linear_mfma.hip's GELU epilogue
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
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.f + fast_erf(v * 0.70710678118654752f)); }

// PROJ = false: the feed-forward block.  PROJ = true: one linear layer with the same register-resident rows,
// y = x . W2^T + b2 + res (W2 (256, F) with F = the layer's input width = 256: chunk c multiplies x's columns
// 64c .. 64c+63 straight from the registers - no GEMM1, no H tile), same epilogue (residual, optional LayerNorm).
template <bool PROJ>
__global__ __launch_bounds__(512, 1) void ffn_fused_kernel(FfnArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int rg = wave >> 1, cg = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int nchunks = a.F / HC;

  // this lane's token row as the B operand of GEMM1: k = 16*ks + 8*h .. +7
  bf16x8 xf[D / 16];
  {
    const unsigned short* xr = a.x + (size_t)min(m0 + rg * 32 + r, a.M - 1) * D + h * 8;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
  }
  // b1 -> LDS (read back 4 floats at a time in the GELU epilogue)
  float* b1s = reinterpret_cast<float*>(smem + OFF_B1);
  if (!PROJ)
    for (int i = tid; i < a.F; i += 512) b1s[i] = a.b1[i];

  // weight DMA: 32 + 32 blocks of 1 KiB per chunk, 4 + 4 per wave.  Lane l of a block lands at byte
  // l*16: W1 block = 2 rows (row = 2*blk + (l >> 5), physical piece l & 31), W2 block = 8 rows
  // (row = 8*blk + (l >> 3), physical piece l & 7); the SOURCE piece is the swizzled one.
  int w1o[4], w2o[4];  // element offsets at chunk 0
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = wave * 4 + i;
    const int row1 = blk * 2 + (lane >> 5);
    w1o[i] = row1 * D + (((lane & 31) ^ (row1 & 15)) << 3);
    const int row2 = blk * 8 + (lane >> 3);
    w2o[i] = row2 * a.F + (((lane & 7) ^ ((row2 >> 1) & 7)) << 3);
  }
  auto issue = [&](int j, int buf) {
    if (!PROJ) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        glds16(a.w1 + (size_t)j * HC * D + w1o[i], smem + OFF_W1 + buf * W1_BYTES + (wave * 4 + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(a.w2 + (size_t)j * HC + w2o[i], smem + OFF_W2 + buf * W2_BYTES + (wave * 4 + i) * 1024);
  };

  f32x16 acc2[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[ct][e] = 0.f;

  // LDS addresses of this lane
  const int a1off = (cg * 32 + r) * 512;            // W1 row (hidden unit) of GEMM1's A operand
  const int a1x = r & 15;                           // its swizzle
  const int trow = rg * 32 + r;                     // token row inside the tile
  const int hrow = trow * 128, hx = (trow >> 1) & 7;  // H row and its swizzle
  int b2off[4], b2x[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = cg * 128 + ct * 32 + r;
    b2off[ct] = n * 128;
    b2x[ct] = (n >> 1) & 7;
  }
  unsigned char* hbuf = smem + OFF_H;

  if (PROJ) {
    // the whole 128 KiB weight matrix fits the (contiguous) W1 + W2 buffer area: all four K chunks are
    // requested at once and the loop below runs without further waits
#pragma unroll
    for (int c = 0; c < D / HC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        glds16(a.w2 + (size_t)c * HC + w2o[i], smem + c * W2_BYTES + (wave * 4 + i) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  } else {
    issue(0, 0);
  }
  for (int j = 0; j < nchunks; ++j) {
    const int buf = j & 1;
    if (!PROJ) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();  // (a): chunk j's weights are in LDS; every wave is done with chunk j-1
      if (j + 1 < nchunks) issue(j + 1, buf ^ 1);
    }

    if (!PROJ) {
      // GEMM1: Ht[hidden cg*32.., tokens rg*32..] over k = 256
      const unsigned char* w1b = smem + OFF_W1 + buf * W1_BYTES + a1off;
      f32x16 acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w1b + (((2 * ks + h) ^ a1x) << 4));
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xf[ks], acc1, 0, 0, 0);
      }
      // D[row = (i&3) + 8*(i>>2) + 4*h][col = r]: hidden units 8g + 4h + {0..3} of token r, g = i >> 2
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + j * HC + cg * 32 + 8 * g + 4 * h);
        uint2 pk;
        pk.x = lss_pack_bf2(gelu(acc1[4 * g] + bb[0]), gelu(acc1[4 * g + 1] + bb[1]));
        pk.y = lss_pack_bf2(gelu(acc1[4 * g + 2] + bb[2]), gelu(acc1[4 * g + 3] + bb[3]));
        *reinterpret_cast<uint2*>(hbuf + hrow + (((cg * 4 + g) ^ hx) << 4) + h * 8) = pk;
      }
      lds_barrier();  // (b): H_j complete
    }

    // GEMM2: Y[tokens rg*32.., n cg*128..] += H_j . W2_j^T over k = 64
    const unsigned char* w2b = PROJ ? smem + j * W2_BYTES : smem + OFF_W2 + buf * W2_BYTES;
#pragma unroll
    for (int s = 0; s < HC / 16; ++s) {
      bf16x8 ha;
      if (PROJ) {
        // columns 64j + 16s .. of the token row: register set 4j + s (j is a run-time index of a fully
        // register-resident array: select with compares, the chunk loop has at most 4 trips)
        ha = xf[s];
#pragma unroll
        for (int q = 1; q < D / HC; ++q)
          if (j == q) ha = xf[4 * q + s];
      } else {
        ha = *reinterpret_cast<const bf16x8*>(hbuf + hrow + (((2 * s + h) ^ hx) << 4));
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(w2b + b2off[ct] + (((2 * s + h) ^ b2x[ct]) << 4));
        acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, wb, acc2[ct], 0, 0, 0);
      }
    }
  }

  // epilogue: (acc + b2) -> fp32 tile in LDS -> 8 consecutive channels per thread + residual x
  // residual pieces of this thread's 8 passes: requested now, so the loads fly during the staging (a load per
  // pass inside the loop below exposed one global round trip per pass: 8 x ~1.5 us per workgroup)
  uint4 rres[BM / 16];
#pragma unroll
  for (int q = 0; q < BM / 16; ++q) {
    const int m = m0 + (tid >> 5) + q * 16;
    rres[q] = make_uint4(0, 0, 0, 0);
    if (m < a.M) rres[q] = *reinterpret_cast<const uint4*>(a.res + (size_t)m * D + (tid & 31) * 8);
  }
  lds_barrier();  // every wave is done reading the last chunk's buffers
  float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = cg * 128 + ct * 32 + r;
    const float bias = a.b2[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = rg * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      otile[row * OLD + n] = acc2[ct][i] + bias;
    }
  }
  lds_barrier();
  {
    const int c8 = tid & 31, row0 = tid >> 5;  // 512 threads = 16 rows x 32 channel groups per pass
    // LayerNorm scale / shift of this thread's 8 channels: the same for every pass (loaded once, not per pass -
    // the stores to y_ln would otherwise force a reload each time)
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (a.ln_g != nullptr) {
      g0 = *reinterpret_cast<const f32x4*>(a.ln_g + c8 * 8);
      g1 = *reinterpret_cast<const f32x4*>(a.ln_g + c8 * 8 + 4);
      t0 = *reinterpret_cast<const f32x4*>(a.ln_b + c8 * 8);
      t1 = *reinterpret_cast<const f32x4*>(a.ln_b + c8 * 8 + 4);
    }
#pragma unroll
    for (int q = 0; q < BM / 16; ++q) {
      const int row = row0 + q * 16, m = m0 + row;
      if (m >= a.M) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(otile + row * OLD + c8 * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(otile + row * OLD + c8 * 8 + 4);
      const uint4 rx = rres[q];
      const unsigned int ru[4] = {rx.x, rx.y, rx.z, rx.w};
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += lss_bf2f((unsigned short)(ru[k] & 0xffff));
        v[2 * k + 1] += lss_bf2f((unsigned short)(ru[k] >> 16));
      }
      if (a.ln_g == nullptr) {
        float* yo = a.y + (size_t)m * D + c8 * 8;
        *reinterpret_cast<f32x4*>(yo) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yo + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
        // the 32 lanes of a half-wave hold one token row: two-pass mean / variance in fp32, as layernorm_kernel
        float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k] -= mean;
          sq = fmaf(v[k], v[k], sq);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float inv = rsqrtf(sq * (1.f / D) + a.ln_eps);
        uint4 ov;
        ov.x = lss_pack_bf2(v[0] * inv * g0[0] + t0[0], v[1] * inv * g0[1] + t0[1]);
        ov.y = lss_pack_bf2(v[2] * inv * g0[2] + t0[2], v[3] * inv * g0[3] + t0[3]);
        ov.z = lss_pack_bf2(v[4] * inv * g1[0] + t1[0], v[5] * inv * g1[1] + t1[1]);
        ov.w = lss_pack_bf2(v[6] * inv * g1[2] + t1[2], v[7] * inv * g1[3] + t1[3]);
        *reinterpret_cast<uint4*>(a.y_ln + (size_t)m * D + c8 * 8) = ov;
      }
    }
  }
}

}  // namespace

static int ffn_launch(bool proj, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                      const void* res, long long M, int d_model, int d_ff, float* y, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, void* y_ln, void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(w2); LSS_CHECK_PTR(b2); LSS_CHECK_PTR(res);
  if (!proj) {
    LSS_CHECK_PTR(w1); LSS_CHECK_PTR(b1);
  }
  const bool ln = ln_gamma != nullptr;
  if (ln) {
    LSS_CHECK_PTR(ln_beta); LSS_CHECK_PTR(y_ln);
  } else {
    LSS_CHECK_PTR(y);
  }
  if (M <= 0 || M >= (1LL << 31) - BM) return LSS_E_SHAPE;
  if (d_model != D || d_ff <= 0 || d_ff % HC != 0 || d_ff > FMAX || (proj && d_ff != D)) return LSS_E_SHAPE;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
        reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y_ln) | reinterpret_cast<uintptr_t>(ln_gamma) |
        reinterpret_cast<uintptr_t>(ln_beta) | reinterpret_cast<uintptr_t>(res)) & 15) != 0)
    return LSS_E_ALIGN;
  // the dynamic-LDS limit is a per-DEVICE function attribute: remember it per (device, kernel)
  static bool attr_set[64][2] = {};
  const void* fn = proj ? reinterpret_cast<const void*>(ffn_fused_kernel<true>)
                        : reinterpret_cast<const void*>(ffn_fused_kernel<false>);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !attr_set[dev][proj]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) attr_set[dev][proj] = true;
  }
  FfnArgs a;
  a.x = reinterpret_cast<const unsigned short*>(x);
  a.w1 = reinterpret_cast<const unsigned short*>(w1);
  a.b1 = b1;
  a.w2 = reinterpret_cast<const unsigned short*>(w2);
  a.b2 = b2;
  a.y = y;
  a.res = reinterpret_cast<const unsigned short*>(res);
  a.M = (int)M;
  a.F = d_ff;
  a.ln_g = ln_gamma;
  a.ln_b = ln_beta;
  a.ln_eps = ln_eps;
  a.y_ln = reinterpret_cast<unsigned short*>(y_ln);
  if (proj)
    hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3(lss_cdiv(M, BM)), dim3(512), SMEM_BYTES, lss_stream(stream), a);
  else
    hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3(lss_cdiv(M, BM)), dim3(512), SMEM_BYTES, lss_stream(stream), a);
  return lss_launch_status();
}

extern "C" int lss_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                 long long M, int d_model, int d_ff, float* y, const float* ln_gamma,
                                 const float* ln_beta, float ln_eps, void* y_ln, void* stream) {
  return ffn_launch(false, x, w1, b1, w2, b2, x, M, d_model, d_ff, y, ln_gamma, ln_beta, ln_eps, y_ln, stream);
}

// y = x . W^T + bias + residual (fp32), or LayerNorm of it (bf16): a 256 -> 256 linear layer whose epilogue
// sees whole token rows.  ref: src/transformer_modules.py:155-156 (output_proj) + :204 (`src + dropout1(.)`, norm1).
extern "C" int lss_linear_res_ln_fwd(const void* x, const void* w, const float* bias, const void* residual,
                                     long long M, int d_model, float* y, const float* ln_gamma,
                                     const float* ln_beta, float ln_eps, void* y_ln, void* stream) {
  return ffn_launch(true, x, nullptr, nullptr, w, bias, residual, M, d_model, d_model, y, ln_gamma, ln_beta, ln_eps,
                    y_ln, stream);
}
