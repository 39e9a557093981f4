// K5/K6: fused lift + splat forward, and K7: its backward.
//
// Forward (output-stationary, atomic-free): a 512-thread workgroup owns a tile of
// 64 consecutive BEV cells of one sample (one z-slab); each of its 8 waves sums 8
// of them with lane = channel:
//     acc[c] = sum_{p in voxel} w[p] * feat[row(p), c]
// K4 lays the entry lists of consecutive voxels out back to back, so a wave
// streams its 8 voxels as ONE contiguous range of {point id, depth weight}
// entries, 64 at a time (one per lane, whole voxels per chunk): the entries of a
// chunk are ordered by point id inside each voxel (rank sort over ds_bpermute ->
// the fp32 summation order is fixed, results are run-to-run reproducible), then
// the sum loop broadcasts (row, weight) with v_readlane while the 256-B feature
// row loads - 16 in flight per lane - are coalesced; voxel boundaries come from one
// wave-wide ballot (a bit test per entry).  The entry key is (feature row << 7) | depth
// bin, so no integer division is needed here.  The 64 x C tile is staged in
// LDS and written once, zeros for empty voxels included: no memset, no atomics,
// every BEV byte stored exactly once, in 16-B-per-lane stores (NHWC) or 256-B
// channel-plane segments (NCHW).
//
// The lifted (B,N,D,fH,fW,C) tensor of the reference (src/modules.py:84,
// src/model_BEV_TXT.py:80,89) exists only as `w * feat` in registers.
#include <stdlib.h>

#include "lss_common.h"

namespace {

constexpr int TILE = 64;  // BEV cells per workgroup
constexpr int NV = 8;     // voxels per wave
constexpr int NWAVE = TILE / NV;

__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Generic per-voxel path (lists longer than one chunk, or slices that are not
// back to back): 64 entries at a time, ordered inside the chunk only.
template <int CPL>
__device__ __forceinline__ void sum_voxel(const float* __restrict__ feat,
                                          const int2* __restrict__ entries, int start, int len,
                                          int DHW, int HW, int lane, float (&acc)[CPL]) {
  constexpr int C = 64 * CPL;
  for (int base = 0; base < len; base += 64) {
    const int n = min(64, len - base);
    int pid = 0x7fffffff;
    float dep = 0.f;
    if (lane < n) {
      const int2 en = entries[start + base + lane];
      pid = en.x;
      dep = __builtin_bit_cast(float, en.y);
    }
    if (n > 1) {
      int rank = 0;
      for (int jj = 0; jj < n; ++jj) rank += (__builtin_amdgcn_readlane(pid, jj) < pid) ? 1 : 0;
      const int dest = (lane < n ? rank : lane) << 2;
      pid = __builtin_amdgcn_ds_permute(dest, pid);
      dep = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dest, __builtin_bit_cast(int, dep)));
    }
    int row = 0;
    if (lane < n) row = pid >> 7;  // entry key = (feature row << 7) | depth bin
    for (int i = 0; i < n; ++i) {
      const int r = __builtin_amdgcn_readlane(row, i);
      const float dd = rl_f(dep, i);
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[q] = fmaf(dd, feat[(size_t)r * C + q * 64 + lane], acc[q]);
    }
  }
}

// grid = (tiles_per_sample, Z, B).  LAYOUT = LSS_BEV_*.
template <int CPL, int LAYOUT>
__global__ __launch_bounds__(512) void lift_splat_fwd_kernel(
    const float* __restrict__ feat, const int32_t* __restrict__ vox_list,
    const int2* __restrict__ entries, int DHW, int HW, int XY, int Z, void* __restrict__ bev_) {
  constexpr int C = 64 * CPL;
  // row stride: +4 keeps 16-B alignment for the NHWC b128 reads; +1 spreads the
  // column reads of the NCHW store over banks
  constexpr int LD = (LAYOUT == LSS_BEV_NCHW_F32) ? C + 1 : C + 4;
  __shared__ __attribute__((aligned(16))) float tile[TILE * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cell0 = blockIdx.x * TILE;
  const int iz = blockIdx.y, b = blockIdx.z;
  const int ncell = min(TILE, XY - cell0);
  float* my_rows = tile + wave * NV * LD;

  // {start, len} of this wave's NV voxels: one per lane (lanes 0..NV-1)
  int my_start = 0, my_len = 0;
  {
    const int jc = wave * NV + lane;
    if (lane < NV && jc < ncell) {
      const size_t v = ((size_t)b * XY + cell0 + jc) * Z + iz;
      my_start = vox_list[2 * v];
      my_len = vox_list[2 * v + 1];
    }
  }
  int pre_end = my_len;  // inclusive scan over lanes 0..NV-1
#pragma unroll
  for (int o = 1; o < NV; o <<= 1) {
    const int t = __shfl_up(pre_end, o, 64);
    if (lane >= o) pre_end += t;
  }
  const int pre = pre_end - my_len;
  const int total = __builtin_amdgcn_readlane(pre_end, NV - 1);

  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
  auto flush = [&](int slot) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      my_rows[slot * LD + q * 64 + lane] = acc[q];
      acc[q] = 0.f;
    }
  };

  if (total == 0) {
    for (int k = 0; k < NV; ++k) flush(k);
  } else {
    // are the non-empty slices back to back (same K4 group)?
    const unsigned long long nonempty = __ballot(lane < NV && my_len > 0);
    const int f0 = __builtin_ctzll(nonempty);
    const int base = __builtin_amdgcn_readlane(my_start - pre, f0);
    const bool ok = !(lane < NV && my_len > 0) || (my_start - pre == base);
    if (__ballot(ok) != ~0ull) {
      for (int k = 0; k < NV; ++k) {
        const int len = __builtin_amdgcn_readlane(my_len, k);
        if (len > 0) sum_voxel<CPL>(feat, entries, __builtin_amdgcn_readlane(my_start, k), len, DHW, HW, lane, acc);
        flush(k);
      }
    } else {
      int s = 0;
      while (s < NV) {
        const int off = __builtin_amdgcn_readlane(pre, s);
        // whole voxels s..e-1 that fit one 64-entry chunk
        const int cnt = __builtin_popcountll(__ballot(lane >= s && lane < NV && (pre_end - off) <= 64));
        if (cnt == 0) {  // voxel s alone is longer than a chunk
          sum_voxel<CPL>(feat, entries, base + off, __builtin_amdgcn_readlane(my_len, s), DHW, HW, lane, acc);
          flush(s);
          ++s;
          continue;
        }
        const int e = s + cnt;
        const int n = __builtin_amdgcn_readlane(pre_end, e - 1) - off;
        int cur = s;
        if (n > 0) {
          int pid = 0x7fffffff;
          float dep = 0.f;
          if (lane < n) {
            const int2 en = entries[base + off + lane];
            pid = en.x;
            dep = __builtin_bit_cast(float, en.y);
          }
          // voxel slot of every entry (entries are grouped by voxel, in voxel order)
          int slot = s, maxlen = 0;
          for (int i = s; i < e; ++i) {
            if (i < e - 1) slot += (lane >= __builtin_amdgcn_readlane(pre_end, i) - off) ? 1 : 0;
            maxlen = max(maxlen, __builtin_amdgcn_readlane(my_len, i));
          }
          if (maxlen > 1) {
            const int first = __shfl(pre, slot, 64) - off;
            const int vlen = __shfl(my_len, slot, 64);
            int rank = 0;
            for (int t = 0; t < maxlen; ++t) {
              const int other = __shfl(pid, min(first + t, 63), 64);
              rank += (t < vlen && other < pid) ? 1 : 0;
            }
            const int dest = (lane < n ? first + rank : lane) << 2;
            pid = __builtin_amdgcn_ds_permute(dest, pid);
            dep = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dest, __builtin_bit_cast(int, dep)));
          }
          int row = 0;  // lanes >= n: row 0 with weight 0 -> harmless loads
          if (lane < n) row = pid >> 7;  // entry key = (feature row << 7) | depth bin
          // bit i of lastmask: entry i closes its voxel's run -> the only per-entry scalar
          // work left in the sum loop is one bit test
          const int nslot = __shfl_down(slot, 1, 64);
          const unsigned long long lastmask = __ballot(lane < n && (lane == n - 1 || nslot != slot));
          for (int k = s; k < e; ++k) flush(k);  // zero rows; occupied voxels overwrite theirs below
          for (int i0 = 0; i0 < n; i0 += 16) {
            float f[16][CPL];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int r = __builtin_amdgcn_readlane(row, min(i0 + u, 63));
#pragma unroll
              for (int q = 0; q < CPL; ++q) f[u][q] = feat[(size_t)r * C + q * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int i = i0 + u;
              if (i < n) {
                const float dd = rl_f(dep, i);
#pragma unroll
                for (int q = 0; q < CPL; ++q) acc[q] = fmaf(dd, f[u][q], acc[q]);
                if ((lastmask >> i) & 1) flush(__builtin_amdgcn_readlane(slot, i));
              }
            }
          }
          cur = e;
        }
        while (cur < e) flush(cur++);
        s = e;
      }
    }
  }
  __syncthreads();

  if (LAYOUT == LSS_BEV_NHWC_F32) {
    float* bev = reinterpret_cast<float*>(bev_);
    // row of cell j: ((b*XY + cell0 + j)*Z + iz) * C ; 16 B per lane
    for (int e = tid; e < TILE * (C / 4); e += 512) {
      const int j = e / (C / 4), c4 = e % (C / 4);
      if (j < ncell) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[j * LD + c4 * 4]);
        *reinterpret_cast<f32x4*>(bev + (((size_t)b * XY + cell0 + j) * Z + iz) * C + c4 * 4) = v;
      }
    }
  } else if (LAYOUT == LSS_BEV_NHWC_BF16) {
    unsigned short* bev = reinterpret_cast<unsigned short*>(bev_);
    for (int e = tid; e < TILE * (C / 8); e += 512) {
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
      for (int e = tid; e < C * (TILE / 4); e += 512) {
        const int c = e / (TILE / 4), j4 = e % (TILE / 4);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = tile[(j4 * 4 + k) * LD + c];
        *reinterpret_cast<f32x4*>(bev + (((size_t)b * Z + iz) * C + c) * XY + cell0 + j4 * 4) = v;
      }
    } else {
      for (int e = tid; e < C * TILE; e += 512) {
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

extern "C" int lss_lift_splat_fwd(const float* feat, const int32_t* vox_list, const int32_t* entries,
                                  int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z,
                                  void* bev, int layout, void* stream) {
  LSS_CHECK_PTR(feat); LSS_CHECK_PTR(vox_list); LSS_CHECK_PTR(entries); LSS_CHECK_PTR(bev);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  if (C != 64 && C != 128) return LSS_E_SHAPE;
  if (layout < 0 || layout > 2) return LSS_E_LAYOUT;
  if (B > 65535 || Z > 65535) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(bev) & 15) != 0 || (reinterpret_cast<uintptr_t>(entries) & 7) != 0)
    return LSS_E_ALIGN;
  const int HW = fH * fW, DHW = D * HW, XY = X * Y;
  dim3 grid(lss_cdiv(XY, TILE), Z, B);
  hipStream_t st = lss_stream(stream);
  const int2* en = reinterpret_cast<const int2*>(entries);
#define LSS_FWD(CPL, LAY)                                                                      \
  hipLaunchKernelGGL((lift_splat_fwd_kernel<CPL, LAY>), grid, dim3(512), 0, st, feat, vox_list, \
                     en, DHW, HW, XY, Z, bev)
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

static int lift_splat_forward_impl(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                   const float* combine, const float* trans, const float* calib_host, const float* dx,
                                   const float* bx, const float* x, const float* w, const float* bias, int B, int N,
                                   int D, int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                                   int32_t* vox_count, int32_t* vox_list, int32_t* entries, int32_t* cursor,
                                   float* depth, float* feat, void* bev, int layout, int math, void* stream) {
  int rc;
  if (calib_host != nullptr) {
    if (math != LSS_DT_F32) return LSS_E_LAYOUT;
    rc = lss_depthnet_voxels_hostcal_fwd(frustum, calib_host, dx, bx, x, w, bias, B, N, D, fH, fW, Cin, C, X, Y, Z,
                                         voxel, vox_count, depth, feat, stream);
    if (rc) return rc;
  } else if (math == LSS_DT_F32 && getenv("LSS_NO_K2K3") == nullptr) {
    // K2 || K3 as one launch (independent, both latency-bound)
    rc = lss_depthnet_voxels_fwd(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, x, w, bias, B, N, D, fH,
                                 fW, Cin, C, X, Y, Z, voxel, vox_count, depth, feat, stream);
    if (rc) return rc;
  } else {
    rc = lss_points_to_voxels(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, B, N, D, fH, fW, X, Y, Z,
                              voxel, vox_count, nullptr, stream);
    if (rc) return rc;
    rc = lss_depthnet_softmax_fwd(x, w, bias, B * N, Cin, fH * fW, D, C, depth, feat, math, stream);
    if (rc) return rc;
  }
  rc = lss_bucket_points(voxel, depth, B * N * D * fH * fW, D, fH * fW, B * X * Y * Z, vox_count, vox_list,
                         entries, cursor, stream);
  if (rc) return rc;
  return lss_lift_splat_fwd(feat, vox_list, entries, B, N, D, fH, fW, C, X, Y, Z, bev, layout, stream);
}

extern "C" int lss_lift_splat_forward(const float* frustum, const float* inv_post_rots,
                                      const float* post_trans, const float* combine, const float* trans,
                                      const float* dx, const float* bx, const float* x, const float* w,
                                      const float* bias, int B, int N, int D, int fH, int fW, int Cin,
                                      int C, int X, int Y, int Z, int32_t* voxel, int32_t* vox_count,
                                      int32_t* vox_list, int32_t* entries, int32_t* cursor, float* depth,
                                      float* feat, void* bev, int layout, int math, void* stream) {
  return lift_splat_forward_impl(frustum, inv_post_rots, post_trans, combine, trans, nullptr, dx, bx, x, w, bias, B, N,
                                 D, fH, fW, Cin, C, X, Y, Z, voxel, vox_count, vox_list, entries, cursor, depth, feat,
                                 bev, layout, math, stream);
}

// The same with the calibration handed over as ONE HOST buffer (layout of lss_depthnet_voxels_hostcal_fwd,
// B*N <= 36, f32 depthnet math): it travels inside the kernel arguments - no H2D copy, no staging.
extern "C" int lss_lift_splat_forward_hostcal(const float* frustum, const float* calib_host, const float* dx,
                                              const float* bx, const float* x, const float* w, const float* bias,
                                              int B, int N, int D, int fH, int fW, int Cin, int C, int X, int Y,
                                              int Z, int32_t* voxel, int32_t* vox_count, int32_t* vox_list,
                                              int32_t* entries, int32_t* cursor, float* depth, float* feat,
                                              void* bev, int layout, void* stream) {
  LSS_CHECK_PTR(calib_host);
  return lift_splat_forward_impl(frustum, nullptr, nullptr, nullptr, nullptr, calib_host, dx, bx, x, w, bias, B, N, D,
                                 fH, fW, Cin, C, X, Y, Z, voxel, vox_count, vox_list, entries, cursor, depth, feat,
                                 bev, layout, LSS_DT_F32, stream);
}
