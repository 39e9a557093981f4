// K5/K6: fused lift + splat forward, and K7: its backward.
//
// Forward (output-stationary, atomic-free): a 256-thread workgroup owns a tile of
// 64 consecutive BEV cells of one sample (one z-slab); each of its 4 waves sums 16
// of them with lane = channel:
//     acc[c] = sum_{p in voxel} depth[p] * feat[row(p), c]
// The voxel's point list is loaded 64 ids at a time (one per lane), ordered by
// point id inside the wave (rank sort over readlanes -> the fp32 sum order is
// fixed, results are run-to-run reproducible for lists <= 64 points), each
// lane fetches the depth weight of ITS point, and the sum loop broadcasts
// (row, depth) with v_readlane while the 256-B feature row load is coalesced.
// The 64 x C tile is staged in LDS and written once: zeros for empty voxels
// included, so there is no memset and every BEV byte is stored exactly once, in
// 16-B-per-lane stores (NHWC) or 256-B channel-plane segments (NCHW).
//
// The lifted (B,N,D,fH,fW,C) tensor of the reference (src/modules.py:84,
// src/model_BEV_TXT.py:80,89) exists only as `depth * feat` in registers.
#include "lss_common.h"

namespace {

constexpr int TILE = 64;  // BEV cells per workgroup

template <int CPL /* channels per lane: C = 64*CPL */>
__device__ __forceinline__ void sum_voxel(const float* __restrict__ depth,
                                          const float* __restrict__ feat,
                                          const int32_t* __restrict__ point_id, int start, int len,
                                          int DHW, int HW, int lane, float (&acc)[CPL]) {
  constexpr int C = 64 * CPL;
  for (int base = 0; base < len; base += 64) {
    const int n = min(64, len - base);
    int pid = 0x7fffffff;
    if (lane < n) pid = point_id[start + base + lane];
    if (n > 1) {
      // rank sort: ids are distinct, so ranks are a permutation of 0..n-1
      int rank = 0;
      for (int jj = 0; jj < n; ++jj) rank += (__builtin_amdgcn_readlane(pid, jj) < pid) ? 1 : 0;
      // push each id to the lane of its rank (lanes >= n keep INT_MAX: rank >= n)
      pid = __builtin_amdgcn_ds_permute(min(rank, 63) << 2, pid);
    }
    float dep = 0.f;
    int row = 0;
    if (lane < n) {
      dep = depth[pid];  // depth is (BN, D, HW): its flat index IS the point id
      const int bn = pid / DHW;
      row = bn * HW + (pid % HW);
    }
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      float f[4][CPL];
      float dd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = __builtin_amdgcn_readlane(row, i + u);
        dd[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dep), i + u));
#pragma unroll
        for (int q = 0; q < CPL; ++q) f[u][q] = feat[(size_t)r * C + q * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = fmaf(dd[u], f[u][q], acc[q]);
    }
    for (; i < n; ++i) {
      const int r = __builtin_amdgcn_readlane(row, i);
      const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dep), i));
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[q] = fmaf(dd, feat[(size_t)r * C + q * 64 + lane], acc[q]);
    }
  }
}

// grid = (tiles_per_sample, Z, B).  LAYOUT = LSS_BEV_*.
template <int CPL, int LAYOUT>
__global__ __launch_bounds__(256) void lift_splat_fwd_kernel(
    const float* __restrict__ depth, const float* __restrict__ feat,
    const int32_t* __restrict__ vox_list, const int32_t* __restrict__ point_id, int DHW, int HW,
    int XY, int Z, void* __restrict__ bev_) {
  constexpr int C = 64 * CPL;
  // row stride: +4 keeps 16-B alignment for the NHWC b128 reads; +1 spreads the
  // column reads of the NCHW store over banks
  constexpr int LD = (LAYOUT == LSS_BEV_NCHW_F32) ? C + 1 : C + 4;
  __shared__ __attribute__((aligned(16))) float tile[TILE * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cell0 = blockIdx.x * TILE;
  const int iz = blockIdx.y, b = blockIdx.z;
  const int ncell = min(TILE, XY - cell0);

  // (start, len) of this wave's 16 voxels: one per lane (lanes 0..15)
  int my_start = 0, my_len = 0;
  {
    const int jc = wave * 16 + lane;
    if (lane < 16 && jc < ncell) {
      const size_t v = ((size_t)b * XY + cell0 + jc) * Z + iz;
      my_start = vox_list[2 * v];
      my_len = vox_list[2 * v + 1];
    }
  }
  for (int s = 0; s < 16; ++s) {
    const int len = __builtin_amdgcn_readlane(my_len, s);
    float acc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
    if (len > 0) {
      const int start = __builtin_amdgcn_readlane(my_start, s);
      sum_voxel<CPL>(depth, feat, point_id, start, len, DHW, HW, lane, acc);
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) tile[(wave * 16 + s) * LD + q * 64 + lane] = acc[q];
  }
  __syncthreads();

  if (LAYOUT == LSS_BEV_NHWC_F32) {
    float* bev = reinterpret_cast<float*>(bev_);
    // row of cell j: ((b*XY + cell0 + j)*Z + iz) * C ; 16 B per lane
    for (int e = tid; e < TILE * (C / 4); e += 256) {
      const int j = e / (C / 4), c4 = e % (C / 4);
      if (j < ncell) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[j * LD + c4 * 4]);
        *reinterpret_cast<f32x4*>(bev + (((size_t)b * XY + cell0 + j) * Z + iz) * C + c4 * 4) = v;
      }
    }
  } else if (LAYOUT == LSS_BEV_NHWC_BF16) {
    unsigned short* bev = reinterpret_cast<unsigned short*>(bev_);
    for (int e = tid; e < TILE * (C / 8); e += 256) {
      const int j = e / (C / 8), c8 = e % (C / 8);
      if (j < ncell) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&tile[j * LD + c8 * 8]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&tile[j * LD + c8 * 8 + 4]);
        uint4 o;
        o.x = lss_pack_bf2(v0[0], v0[1]); o.y = lss_pack_bf2(v0[2], v0[3]);
        o.z = lss_pack_bf2(v1[0], v1[1]); o.w = lss_pack_bf2(v1[2], v1[3]);
        *reinterpret_cast<uint4*>(bev + (((size_t)b * XY + cell0 + j) * Z + iz) * C + c8 * 8) = o;
      }
    }
  } else {  // NCHW fp32: bev[b][iz*C + c][cell]
    float* bev = reinterpret_cast<float*>(bev_);
    const bool vec_ok = (ncell == TILE) && ((XY & 3) == 0);
    if (vec_ok) {
      for (int e = tid; e < C * (TILE / 4); e += 256) {
        const int c = e / (TILE / 4), j4 = e % (TILE / 4);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = tile[(j4 * 4 + k) * LD + c];
        *reinterpret_cast<f32x4*>(bev + (((size_t)b * Z + iz) * C + c) * XY + cell0 + j4 * 4) = v;
      }
    } else {
      for (int e = tid; e < C * TILE; e += 256) {
        const int c = e / TILE, j = e % TILE;
        if (j < ncell) bev[(((size_t)b * Z + iz) * C + c) * XY + cell0 + j] = tile[j * LD + c];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Backward.  One wave per camera pixel (bn, pix), lane = channel; a workgroup
// covers 16 consecutive pixels of one image so the (D+C) x 16 block of g_logits
// leaves through LDS in 64-B row segments.
//   g_feat[c]  = sum_d depth[d] * G[voxel(d), c]
//   g_depth[d] = sum_c feat[c]  * G[voxel(d), c]          (wave reduction)
//   g_logit[d] = depth[d] * (g_depth[d] - sum_d' depth[d'] g_depth[d'])   (softmax bwd)
// Dropped points (voxel < 0) contribute nothing (ref: x[kept], src/model_BEV_TXT.py:103).
template <int CPL, int LAYOUT>
__global__ __launch_bounds__(256) void lift_splat_bwd_kernel(
    const float* __restrict__ G, const int32_t* __restrict__ voxel,
    const float* __restrict__ depth, const float* __restrict__ feat, int D, int HW, int XY, int Z,
    float* __restrict__ g_logits) {
  constexpr int C = 64 * CPL;
  constexpr int PIXW = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [(D + C)][PIXW + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bn = blockIdx.y, pix0 = blockIdx.x * PIXW;
  const int ZC = Z * C;
  for (int i = 0; i < 4; ++i) {
    const int pl = wave * 4 + i;  // pixel slot in the workgroup
    const int pix = pix0 + pl;
    if (pix >= HW) break;  // wave-uniform
    float f[CPL], gf[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      f[q] = feat[((size_t)bn * HW + pix) * C + q * 64 + lane];
      gf[q] = 0.f;
    }
    float dot_acc = 0.f;  // sum_d depth[d] * g_depth[d]
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int nd = min(64, D - d0);
      // lane d holds voxel / depth of point (bn, d0 + lane, pix)
      int vv = -1;
      float dv = 0.f;
      if (lane < nd) {
        const size_t p = ((size_t)bn * D + d0 + lane) * HW + pix;
        vv = voxel[p];
        dv = depth[p];
      }
      float gd_mine = 0.f;
      for (int d = 0; d < nd; ++d) {
        const int v = __builtin_amdgcn_readlane(vv, d);
        if (v < 0) continue;  // wave-uniform
        const float dep = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv), d));
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          float g;
          if (LAYOUT == LSS_BEV_NCHW_F32) {
            const int cell = v / Z, iz = v % Z;       // v = (b*XY + c)*Z + iz
            const int bb = cell / XY, cc = cell % XY;
            g = G[((size_t)bb * ZC + iz * C + q * 64 + lane) * XY + cc];
          } else {
            g = G[(size_t)v * C + q * 64 + lane];
          }
          gf[q] = fmaf(dep, g, gf[q]);
          part = fmaf(f[q], g, part);
        }
        part = lss_wave_sum(part);
        if (lane == d) gd_mine = part;
      }
      dot_acc += lss_wave_sum(dv * gd_mine);
      // stash (depth, g_depth) of this chunk for the second softmax-backward pass
      if (lane < nd) {
        lds[(d0 + lane) * (PIXW + 1) + pl] = gd_mine;
      }
    }
    // second pass: g_logit[d] = depth[d] * (g_depth[d] - dot)
    for (int d0 = 0; d0 < D; d0 += 64) {
      if (d0 + lane < D) {
        const size_t p = ((size_t)bn * D + d0 + lane) * HW + pix;
        const float gd = lds[(d0 + lane) * (PIXW + 1) + pl];
        lds[(d0 + lane) * (PIXW + 1) + pl] = depth[p] * (gd - dot_acc);
      }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) lds[(D + q * 64 + lane) * (PIXW + 1) + pl] = gf[q];
  }
  __syncthreads();
  const int NO = D + C;
  for (int e = tid; e < NO * PIXW; e += 256) {
    const int n = e / PIXW, pl = e % PIXW;
    if (pix0 + pl < HW) g_logits[((size_t)bn * NO + n) * HW + pix0 + pl] = lds[n * (PIXW + 1) + pl];
  }
}

}  // namespace

extern "C" int lss_lift_splat_fwd(const float* depth, const float* feat, const int32_t* vox_list,
                                  const int32_t* point_id, int B, int N, int D, int fH, int fW,
                                  int C, int X, int Y, int Z, void* bev, int layout, void* stream) {
  LSS_CHECK_PTR(depth); LSS_CHECK_PTR(feat); LSS_CHECK_PTR(vox_list); LSS_CHECK_PTR(point_id);
  LSS_CHECK_PTR(bev);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  if (C != 64 && C != 128) return LSS_E_SHAPE;
  if (layout < 0 || layout > 2) return LSS_E_LAYOUT;
  if (B > 65535 || Z > 65535) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(bev) & 15) != 0) return LSS_E_ALIGN;
  const int HW = fH * fW, DHW = D * HW, XY = X * Y;
  dim3 grid(lss_cdiv(XY, TILE), Z, B);
  hipStream_t st = lss_stream(stream);
#define LSS_FWD(CPL, LAY)                                                                      \
  hipLaunchKernelGGL((lift_splat_fwd_kernel<CPL, LAY>), grid, dim3(256), 0, st, depth, feat,   \
                     vox_list, point_id, DHW, HW, XY, Z, bev)
  if (C == 64) {
    if (layout == LSS_BEV_NCHW_F32) LSS_FWD(1, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_FWD(1, LSS_BEV_NHWC_F32);
    else LSS_FWD(1, LSS_BEV_NHWC_BF16);
  } else {
    if (layout == LSS_BEV_NCHW_F32) LSS_FWD(2, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_FWD(2, LSS_BEV_NHWC_F32);
    else LSS_FWD(2, LSS_BEV_NHWC_BF16);
  }
#undef LSS_FWD
  return lss_launch_status();
}

extern "C" int lss_lift_splat_bwd(const void* grad_bev, int layout, const int32_t* voxel,
                                  const float* depth, const float* feat, int B, int N, int D,
                                  int fH, int fW, int C, int X, int Y, int Z, float* g_logits,
                                  void* stream) {
  LSS_CHECK_PTR(grad_bev); LSS_CHECK_PTR(voxel); LSS_CHECK_PTR(depth); LSS_CHECK_PTR(feat);
  LSS_CHECK_PTR(g_logits);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  if (C != 64 && C != 128) return LSS_E_SHAPE;
  if (layout != LSS_BEV_NCHW_F32 && layout != LSS_BEV_NHWC_F32) return LSS_E_LAYOUT;
  if ((long long)B * N > 65535) return LSS_E_SHAPE;
  const int HW = fH * fW, XY = X * Y;
  const size_t lds_bytes = (size_t)(D + C) * 17 * sizeof(float);
  if (lds_bytes > 64 * 1024) return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(HW, 16), B * N);
  hipStream_t st = lss_stream(stream);
  const float* G = reinterpret_cast<const float*>(grad_bev);
#define LSS_BWD(CPL, LAY)                                                                       \
  hipLaunchKernelGGL((lift_splat_bwd_kernel<CPL, LAY>), grid, dim3(256), lds_bytes, st, G,      \
                     voxel, depth, feat, D, HW, XY, Z, g_logits)
  if (C == 64) {
    if (layout == LSS_BEV_NCHW_F32) LSS_BWD(1, LSS_BEV_NCHW_F32);
    else LSS_BWD(1, LSS_BEV_NHWC_F32);
  } else {
    if (layout == LSS_BEV_NCHW_F32) LSS_BWD(2, LSS_BEV_NCHW_F32);
    else LSS_BWD(2, LSS_BEV_NHWC_F32);
  }
#undef LSS_BWD
  return lss_launch_status();
}
