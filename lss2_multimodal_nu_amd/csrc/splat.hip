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
#include "region_plan.h"

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
// Region splat (launch 3 of the fused inference path; bucketing: geom_bucket.hip).
// One 256-thread workgroup per REGION of 8 x 8 BEV cells (all z) of one sample.  Its points arrive as one
// contiguous run of {(feature row << 8) | cell, depth weight} entries in ARBITRARY order; each wave takes 64 of
// them at a time (one coalesced 512-B load), broadcasts (row, cell, weight) lane by lane through SGPRs and -
// lane = channel - adds  w * feat[row][c]  into the region's LDS tile with an LDS atomic.  The sums are FIXED
// POINT (int64, scale 2^(40 - e) with max|feat| < 2^e taken from K2): integer addition is associative, so the
// result does not depend on the order the bucketing produced - run-to-run bit-reproducible without sorting
// anything - and carries 40 bits below the largest feature (fp32 itself has 24).
// The inner loop is instruction-issue bound (one wave-iteration per point), so it is kept to 12 instructions:
// 3 v_readlane, 1 buffer_load (SGPR row offset), v_mul, 2 for the non-finite watch, cvt + fma + sub for the
// fixed-point conversion (magic-number rounding: bits(x*scale + 1.5*2^52) - bits(1.5*2^52)), 1 address add,
// ds_add_u64.  Non-finite products are only WATCHED there (running max of the magnitude bits); if one shows up the
// region is redone by the careful loop, which flags the (cell, channel) elements it touches - those are written
// as NaN.
// The tile is then converted once and leaves in coalesced stores (8 x Z cells of an ix row are contiguous in
// the NHWC layouts), zeros for empty cells included: every BEV byte is written exactly once, no memset, no
// global atomics.  Empty regions skip LDS altogether.  The workgroup also returns its workspace counters to
// zero for the next call.
constexpr double FX_MAGIC = 6755399441055744.0;  // 1.5 * 2^52: bit pattern 0x4338000000000000
constexpr int FX_MAGIC_HI = 0x43380000;

template <int CPL>
__device__ __forceinline__ void region_accumulate_careful(const float* __restrict__ feat,
                                                          const int2* __restrict__ entries, int start, int n,
                                                          double scale, float limit, unsigned long long* tile,
                                                          unsigned int* flags, int wave, int lane) {
  constexpr int C = 64 * CPL;
  for (int c0 = wave * 64; c0 < n; c0 += 256) {
    int key = 0;
    float w = 0.f;
    if (c0 + lane < n) {
      const int2 en = entries[start + c0 + lane];
      key = en.x;
      w = __builtin_bit_cast(float, en.y);
    }
    const int cnt = min(64, n - c0);
    for (int i = 0; i < cnt; ++i) {
      const int k = __builtin_amdgcn_readlane(key, i);
      const float ww = rl_f(w, i);
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const float x = ww * feat[(size_t)(k >> 8) * C + q * 64 + lane];
        const int o = (k & 255) * C + q * 64 + lane;
        if (!(fabsf(x) < limit)) {  // non-finite, or beyond what the fixed-point accumulator holds: the cell reads NaN
          atomicOr(&flags[o >> 5], 1u << (o & 31));
        } else {
          const double t = __builtin_fma((double)x, scale, FX_MAGIC);
          atomicAdd(&tile[o], (unsigned long long)(__builtin_bit_cast(long long, t) - ((long long)FX_MAGIC_HI << 32)));
        }
      }
    }
  }
}

// (An fp32 tile updated with ds_add_f32 - 7 instructions per point instead of 12 - was built and measured: 92.8 us
// against 20.1 us for this kernel on the bench workload.  LDS float atomics are an order of magnitude slower than
// the 64-bit integer ones on gfx950, so the fixed-point form is the fast one as well as the reproducible one.)
// DIRECT (region_plan.h): the entries of region r sit at r * cap as {key, point id} - written by launch 1 itself, no
// fill launch, no region_start - and the depth weight is gathered from depth[point id] beside the entry load; the
// maximum over K2's per-workgroup |feature| maxima (what the fill kernel's first workgroup used to reduce) is taken
// by every workgroup from the n2 slots (1 KB, L2-resident) while its other loads fly.  A region whose count exceeds
// `cap` takes its full bucket plus its records of the overflow list (region_plan.h; hi-res rigs do this every frame);
// only if the list itself overflowed is it rebuilt from the voxel ids of its sample (region_accumulate_scan): exact,
// slow, and only reachable with degenerate calibrations.
struct DirectArgs {
  const float* depth;      // (B*N, D, HW): flat index = point id
  const int32_t* voxel;    // point id -> voxel id or -1
  int Ncam, DHW, HW;
};

// every point of sample b whose voxel lies in region (rx, ry): the careful (flagging) accumulation, element by element
template <int CPL>
__device__ __forceinline__ void region_accumulate_scan(const float* __restrict__ feat, const DirectArgs& da, int b,
                                                       int rx, int ry, int X, int Y, int Z, double scale, float limit,
                                                       unsigned long long* tile, unsigned int* flags, int wave,
                                                       int lane) {
  constexpr int C = 64 * CPL;
  const int np = da.Ncam * da.DHW;  // points of one sample
  for (int p0 = wave * 64; p0 < np; p0 += 256) {
    const int pl = p0 + lane;
    int key = -1;
    float w = 0.f;
    if (pl < np) {
      const int p = b * np + pl;
      const int v = da.voxel[p];
      if (v >= 0) {
        const int iz = v % Z, cxy = v / Z - b * X * Y;
        const int ix = cxy / Y, iy = cxy - ix * Y;
        if ((ix >> 3) == rx && (iy >> 3) == ry) {
          const int bn = p / da.DHW, f = p - bn * da.DHW, pix = f % da.HW;
          key = ((bn * da.HW + pix) << 8) | ((((ix & 7) << 3) | (iy & 7)) * Z + iz);
          w = da.depth[p];
        }
      }
    }
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
      const int i = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int k = __builtin_amdgcn_readlane(key, i);
      const float ww = rl_f(w, i);
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const float x = ww * feat[(size_t)(k >> 8) * C + q * 64 + lane];
        const int o = (k & 255) * C + q * 64 + lane;
        if (!(fabsf(x) < limit)) {
          atomicOr(&flags[o >> 5], 1u << (o & 31));
        } else {
          const double t = __builtin_fma((double)x, scale, FX_MAGIC);
          atomicAdd(&tile[o], (unsigned long long)(__builtin_bit_cast(long long, t) - ((long long)FX_MAGIC_HI << 32)));
        }
      }
    }
  }
}

// the records of the overflow list that belong to region r (region_plan.h), element by element like the scan above:
// finite products in fixed point, anything else flagged
template <int CPL>
__device__ __forceinline__ void region_accumulate_overflow(const float* __restrict__ feat, const DirectArgs& da,
                                                           const int4* __restrict__ ovf, int novf, int r, double scale,
                                                           float limit, unsigned long long* tile, unsigned int* flags,
                                                           int wave, int lane) {
  constexpr int C = 64 * CPL;
  for (int i0 = wave * 64; i0 < novf; i0 += 256) {
    int key = -1;
    float w = 0.f;
    if (i0 + lane < novf) {
      const int4 rec = ovf[i0 + lane];
      if (rec.x == r) {
        key = rec.y;
        w = da.depth[rec.z];
      }
    }
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
      const int i = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int k = __builtin_amdgcn_readlane(key, i);
      const float ww = rl_f(w, i);
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const float x = ww * feat[(size_t)(k >> 8) * C + q * 64 + lane];
        const int o = (k & 255) * C + q * 64 + lane;
        if (!(fabsf(x) < limit)) {
          atomicOr(&flags[o >> 5], 1u << (o & 31));
        } else {
          const double t = __builtin_fma((double)x, scale, FX_MAGIC);
          atomicAdd(&tile[o], (unsigned long long)(__builtin_bit_cast(long long, t) - ((long long)FX_MAGIC_HI << 32)));
        }
      }
    }
  }
}

template <int CPL, int LAYOUT, bool DIRECT = false>
__global__ __launch_bounds__(256) void region_splat_kernel(const float* __restrict__ feat,
                                                           const int2* __restrict__ entries, LssRegionPlan rp,
                                                           int X, int Y, int Z, void* __restrict__ bev_,
                                                           unsigned long long* stamps, int centre_out, DirectArgs da) {
  constexpr int C = 64 * CPL;
  // LSS_L1_STAMPS diagnostic: s_memrealtime at entry / tile cleared / sums complete / stores issued; word 4 = points
  unsigned long long* const stp = stamps ? stamps + (size_t)blockIdx.x * 8 : nullptr;
  if (stp != nullptr && threadIdx.x == 0) stp[0] = __builtin_amdgcn_s_memrealtime();
  constexpr int RSIDE = LSS_REGION_SIDE;
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];  // [64*Z][C] sums, then flag words
  __shared__ unsigned int wg_watch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Block -> region: samples interleaved, and inside a sample the region rows - and the regions of a row - taken from
  // the CENTRE of the grid outwards (centre, +1, -1, +2, ...).  The frustum points crowd around the ego vehicle, so the
  // busiest regions (7-8 us of accumulation against a median of 2) are dispatched in the first round instead of
  // starting 8 us into the launch as the last of 2.4 rounds: stamps put the launch's end at 18.8 us with the row-major
  // order against ~12 us of work per slot.  A pure relabelling: every region is still visited exactly once.
  const int nb = gridDim.x / rp.rps;  // samples
  const int b = blockIdx.x % nb, kk = blockIdx.x / nb;
  const int zi = kk / rp.nRy, zj = kk - zi * rp.nRy;
  const int rx = centre_out ? (rp.nRx - 1) / 2 + ((zi & 1) ? (zi + 1) / 2 : -(zi / 2)) : zi;
  const int ry = centre_out ? (rp.nRy - 1) / 2 + ((zj & 1) ? (zj + 1) / 2 : -(zj / 2)) : zj;
  const int r = b * rp.rps + rx * rp.nRy + ry;
  const int ncell = RSIDE * RSIDE * Z;
  unsigned int* flags = reinterpret_cast<unsigned int*>(tile + (size_t)ncell * C);  // [ncell*C/32]
  __shared__ float wave_absmax[4];
  // DIRECT: a wave's first 64 entries are requested BEFORE anything is known about the region - the bucket sits at a
  // fixed offset - together with their depth weights (point id clamped: slots past the region's count hold stale
  // bytes), so that the count, the feature maximum and entries -> depth are ONE overlapped round trip pair instead of
  // count -> entries -> depth (measured +2.5 us on the launch with the dependent form).  Entries are dealt to the waves
  // in groups of eight, round robin: entry ((lane >> 3) * 4 + wave) * 8 + (lane & 7) of every 256 - whatever the
  // count, all four waves get a quarter of the region.
  const int e0 = ((lane >> 3) * 4 + wave) * 8 + (lane & 7);
  int skey = 0;
  float sw = 0.f;
  if (DIRECT) {
    const int2 en = entries[(size_t)r * rp.cap + e0];
    skey = en.x;
    sw = da.depth[min((unsigned int)en.y, (unsigned int)(gridDim.x / rp.rps) * (unsigned int)(da.Ncam * da.DHW) - 1u)];
  }
  const int ntot = rp.region_count[r];                       // points of the region
  const bool over = DIRECT && ntot > rp.cap;                 // more than its bucket holds
  // ... then: the full bucket + this region's records of the overflow list, unless the LIST overflowed (or there is
  // none): only then is the region rebuilt from the voxel ids of its sample
  const int novf = over ? rp.ovf_ctl[0] : 0;
  const bool list_ok = over && rp.ovf != nullptr && novf <= rp.ovf_cap;
  const int n = DIRECT ? (over ? (list_ok ? rp.cap : 0) : ntot) : ntot;   // entries to stream
  const int start = DIRECT ? r * rp.cap : rp.region_start[r];
  float m;  // max FINITE |feature| over K2's workgroups
  if (DIRECT) {
    float mm = 0.f;
    for (int i = tid; i < rp.n2; i += 256) mm = fmaxf(mm, rp.wg_absmax[i]);
    mm = lss_wave_max(mm);
    if (lane == 0) wave_absmax[wave] = mm;
  } else {
    m = rp.wg_absmax[rp.n2];  // (reduced by the fill kernel)
  }
  if (tid == 0) wg_watch = 0;
  __syncthreads();  // every thread has read region_count[r]
  if (DIRECT) m = fmaxf(fmaxf(wave_absmax[0], wave_absmax[1]), fmaxf(wave_absmax[2], wave_absmax[3]));
  if (tid == 0) {   // the counters go back to zero (workspace contract)
    rp.region_count[r] = 0;
    if (!DIRECT) rp.region_cursor[r] = 0;
  }
  // Every finite product is |depth weight (<= 1) x feature| <= m < 2^e, so |x * 2^(40-e)| < 2^40: far inside the
  // 2^51 the magic-number conversion holds.  A product at or above 2^(e+10) can only come from a non-finite operand
  // or from weights above 1 (no caller has them); `watch` sends the region to the careful path in either case.
  int e = 0;
  if (m > 0.f && m <= 3.0e38f) (void)frexpf(m, &e);  // m < 2^e
  e = max(e, -80);  // all-tiny features: keep 2^(40-e) and its reciprocals finite in fp32
  const double scale = ldexp(1.0, 40 - e);
  const unsigned int watch_limit = __builtin_bit_cast(unsigned int, ldexpf(1.0f, min(e + 10, 127)));
  const float inv_lo = ldexpf(1.0f, e - 40), inv_hi = ldexpf(1.0f, e - 8);  // 2^-(40-e), and x 2^32
  const int zero_n = ncell * C / 2 + ncell * C / 128;  // 16-B stores that clear the sums and the flag words

  if (n > 0) {
    // The region's entries are dealt to the 4 waves in slices of `chunk` <= 64 (a multiple of 8, about n / 4): all four
    // waves work on any region of >= 32 points.  The first slice is requested BEFORE the tile is cleared, so the
    // round trip of that load overlaps the LDS stores and the barrier.  (DIRECT: groups of eight round robin, see e0.)
    const int chunk = DIRECT ? 64 : min(64, max(8, ((n + 3) / 4 + 7) & ~7));
    int key = 0;
    float w = 0.f;  // lanes past the end of a slice: weight 0 on row 0 / cell 0 - adds nothing
    int c0 = DIRECT ? 0 : wave * chunk;  // DIRECT: base of the current block of 256 entries
    if (DIRECT) {
      if (e0 < n) { key = skey; w = sw; }
    } else if (lane < chunk && c0 + lane < n) {
      const int2 en = entries[start + c0 + lane];
      key = en.x;
      w = __builtin_bit_cast(float, en.y);
    }
    for (int i = tid; i < zero_n; i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (stp != nullptr && tid == 0) stp[1] = __builtin_amdgcn_s_memrealtime();
    const __amdgpu_buffer_rsrc_t frs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, 0x7fffffff, 0x00020000);
    unsigned long long* const my_col = tile + lane;  // lane = channel
    unsigned int watch = 0;  // running max of |x| bit patterns: >= 0x7f800000 <=> a non-finite product went by
    while (c0 < n) {
      const int rowoff = (key >> 8) * (C * 4);             // byte offset of the feature row
      const int celloff = (key & 255) * C;                 // element offset of the cell's tile row
      // DIRECT: this wave's entries of the block are a prefix of its lanes (entry index grows with the lane)
      const int cnt = DIRECT ? (int)__builtin_popcountll(__ballot(c0 + e0 < n)) : min(chunk, n - c0);
      for (int i0 = 0; i0 < cnt; i0 += 8) {  // i0 + 7 <= 63
        float f[8][CPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ro = __builtin_amdgcn_readlane(rowoff, i0 + u);
#pragma unroll
          for (int q = 0; q < CPL; ++q)
            f[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, lane * 4 + q * 256, ro, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int co = __builtin_amdgcn_readlane(celloff, i0 + u);
          const float ww = rl_f(w, i0 + u);
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const float x = ww * f[u][q];
            watch = max(watch, __builtin_bit_cast(unsigned int, x) & 0x7fffffffu);
            const double t = __builtin_fma((double)x, scale, FX_MAGIC);
            unsigned long long qv = __builtin_bit_cast(unsigned long long, t);
            qv -= (unsigned long long)FX_MAGIC_HI << 32;  // one 32-bit subtract: the magic's low word is zero
            __hip_atomic_fetch_add(my_col + co + q * 64, qv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u64
          }
        }
      }
      c0 += 4 * chunk;
      key = 0;
      w = 0.f;
      if (DIRECT) {
        if (c0 + e0 < n) {
          const int2 en = entries[start + c0 + e0];
          key = en.x;
          w = da.depth[en.y];
        }
      } else if (c0 < n && lane < chunk && c0 + lane < n) {
        const int2 en = entries[start + c0 + lane];
        key = en.x;
        w = __builtin_bit_cast(float, en.y);
      }
    }
    if (watch >= watch_limit) atomicOr(&wg_watch, 1u);  // a non-finite (or impossibly large) product went by
    __syncthreads();
    if (stp != nullptr && tid == 0) stp[2] = __builtin_amdgcn_s_memrealtime();
    if (wg_watch != 0) {  // rare: a non-finite product - redo the region element by element, with flags
      __syncthreads();
      for (int i = tid; i < zero_n; i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
      if (DIRECT)
        region_accumulate_scan<CPL>(feat, da, b, rx, ry, X, Y, Z, scale, ldexpf(1.0f, min(e + 10, 127)), tile, flags, wave, lane);
      else
        region_accumulate_careful<CPL>(feat, entries, start, n, scale, ldexpf(1.0f, min(e + 10, 127)), tile, flags, wave, lane);
      __syncthreads();
    }
  } else if (over) {  // DIRECT, bucket AND list overflow: the whole region from the voxel ids of its sample
    for (int i = tid; i < zero_n; i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    region_accumulate_scan<CPL>(feat, da, b, rx, ry, X, Y, Z, scale, ldexpf(1.0f, min(e + 10, 127)), tile, flags, wave, lane);
    __syncthreads();
  }
  if (DIRECT && over) {
    if (list_ok && wg_watch == 0) {  // (a careful re-run above has already taken every point of the region)
      region_accumulate_overflow<CPL>(feat, da, reinterpret_cast<const int4*>(rp.ovf), novf, r, scale,
                                      ldexpf(1.0f, min(e + 10, 127)), tile, flags, wave, lane);
      __syncthreads();
    }
    // the last over-capacity region to get here puts the three control words back to zero (workspace contract);
    // every such workgroup has read ovf_ctl[0] and the list before it counts itself in
    if (tid == 0) {
      __threadfence();
      const int done = atomicAdd(rp.ovf_ctl + 2, 1) + 1;
      if (done == __hip_atomic_load(rp.ovf_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        rp.ovf_ctl[0] = 0; rp.ovf_ctl[1] = 0; rp.ovf_ctl[2] = 0;
      }
    }
  }
  const bool any = n > 0 || over;  // the tile holds sums

  auto value = [&](int o) -> float {  // (cell, channel) element o of the tile as fp32
    // int64 -> fp32 on the fp32 pipe: hi * 2^32 * 2^-s + lo * 2^-s (hi, lo the two words).  |sum| < 2^56 (terms below
    // 2^40, fewer than 2^16 of them in a cell) keeps the conversion of hi exact and the result within one rounding of
    // the exact quotient; a cell with more terms than that (hi-res rigs can put > 65536 points in ONE cell only with
    // degenerate calibrations) pays one more fp32 rounding, nothing worse: hi stays far inside int32.
    const unsigned long long sv = tile[o];
    const float v = __builtin_fmaf((float)(int)(sv >> 32), inv_hi, (float)(unsigned int)sv * inv_lo);
    return ((flags[o >> 5] >> (o & 31)) & 1u) ? __builtin_nanf("") : v;
  };
  const int ix0 = rx * RSIDE, iy0 = ry * RSIDE;
  if (LAYOUT == LSS_BEV_NCHW_F32) {
    // bev[((b*Z + iz)*C + c)][ix][iy]: a piece = the 8 iy-consecutive cells of (c, lx, iz)
    float* bev = reinterpret_cast<float*>(bev_);
    const bool vec_ok = (Y & 3) == 0 && iy0 + RSIDE <= Y;
    for (int pc = tid; pc < C * RSIDE * Z; pc += 256) {
      const int c = pc % C, t = pc / C, lx = t % RSIDE, iz = t / RSIDE;
      const int ix = ix0 + lx;
      if (ix >= X) continue;
      float v[RSIDE];
#pragma unroll
      for (int ly = 0; ly < RSIDE; ++ly) v[ly] = any ? value(((lx * RSIDE + ly) * Z + iz) * C + c) : 0.f;
      float* op = bev + (((size_t)b * Z + iz) * C + c) * X * Y + (size_t)ix * Y + iy0;
      if (vec_ok) {
        *reinterpret_cast<f32x4*>(op) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
#pragma unroll
        for (int ly = 0; ly < RSIDE; ++ly)
          if (iy0 + ly < Y) op[ly] = v[ly];
      }
    }
  } else {
    // NHWC: row of cell (ix, iy, iz) = (((b*X + ix)*Y + iy)*Z + iz) * C; a piece = 8 channels of one cell
    for (int pc = tid; pc < ncell * (C / 8); pc += 256) {
      const int cell = pc / (C / 8), c8 = pc % (C / 8);
      const int iz = cell % Z, t = cell / Z, ly = t % RSIDE, lx = t / RSIDE;
      const int ix = ix0 + lx, iy = iy0 + ly;
      if (ix >= X || iy >= Y) continue;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = any ? value(cell * C + c8 * 8 + k) : 0.f;
      const size_t o = ((((size_t)b * X + ix) * Y + iy) * Z + iz) * C + c8 * 8;
      if (LAYOUT == LSS_BEV_NHWC_F32) {
        float* bev = reinterpret_cast<float*>(bev_);
        *reinterpret_cast<f32x4*>(bev + o) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(bev + o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
        uint4 ov;
        ov.x = lss_pack_bf2(v[0], v[1]); ov.y = lss_pack_bf2(v[2], v[3]);
        ov.z = lss_pack_bf2(v[4], v[5]); ov.w = lss_pack_bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(bev_) + o) = ov;
      }
    }
  }
  if (stp != nullptr) {
    __syncthreads();
    if (tid == 0) {
      stp[3] = __builtin_amdgcn_s_memrealtime();
      stp[4] = (unsigned long long)ntot;
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
__global__ __launch_bounds__(1024) void lift_splat_bwd_kernel(
    const float* __restrict__ G, const int32_t* __restrict__ voxel,
    const float* __restrict__ depth, const float* __restrict__ feat, int D, int HW, int XY, int Z,
    float* __restrict__ g_logits) {
  constexpr int C = 64 * CPL;
  constexpr int PIXW = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [(D + C)][PIXW + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bn = blockIdx.y, pix0 = blockIdx.x * PIXW;
  const int ZC = Z * C;
  // one pixel per wave, sixteen waves per workgroup (four per SIMD: three more to hide each gather's round trip than the
  // four-pixels-per-wave form had)
  for (int i = 0; i < 1; ++i) {
    const int pl = wave;  // pixel slot in the workgroup
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
      // The gradient rows of EIGHT points are requested before any of them is used (dropped points read row 0 and are
      // skipped afterwards): with one load in flight per wave and one wave per SIMD - 264 workgroups - the kernel was a
      // chain of 164 dependent L2 round trips per wave: 87 us of the training step (round 4: see DESIGN.md section 9).
      for (int d = 0; d < nd; d += 8) {
        int vs[8];
        float g[8][CPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int v = __builtin_amdgcn_readlane(vv, min(d + u, nd - 1));
          vs[u] = d + u < nd ? v : -1;
          const int vr = v < 0 ? 0 : v;  // a valid row for the loads of points that do not count
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            if (LAYOUT == LSS_BEV_NCHW_F32) {
              const int cell = vr / Z, iz = vr % Z;       // v = (b*XY + c)*Z + iz
              const int bb = cell / XY, cc = cell % XY;
              g[u][q] = G[((size_t)bb * ZC + iz * C + q * 64 + lane) * XY + cc];
            } else if (LAYOUT == LSS_BEV_NHWC_BF16) {
              g[u][q] = lss_bf2f(reinterpret_cast<const unsigned short*>(G)[(size_t)vr * C + q * 64 + lane]);
            } else {
              g[u][q] = G[(size_t)vr * C + q * 64 + lane];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (vs[u] < 0) continue;  // wave-uniform
          const float dep = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv), d + u));
          float part = 0.f;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            gf[q] = fmaf(dep, g[u][q], gf[q]);
            part = fmaf(f[q], g[u][q], part);
          }
          part = lss_wave_sum(part);
          if (lane == d + u) gd_mine = part;
        }
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
  for (int e = tid; e < NO * PIXW; e += 1024) {
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
  if (layout != LSS_BEV_NCHW_F32 && layout != LSS_BEV_NHWC_F32 && layout != LSS_BEV_NHWC_BF16) return LSS_E_LAYOUT;
  if ((long long)B * N > 65535) return LSS_E_SHAPE;
  const int HW = fH * fW, XY = X * Y;
  const size_t lds_bytes = (size_t)(D + C) * 17 * sizeof(float);
  if (lds_bytes > 64 * 1024) return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(HW, 16), B * N);
  hipStream_t st = lss_stream(stream);
  const float* G = reinterpret_cast<const float*>(grad_bev);
#define LSS_BWD(CPL, LAY)                                                                       \
  hipLaunchKernelGGL((lift_splat_bwd_kernel<CPL, LAY>), grid, dim3(1024), lds_bytes, st, G,     \
                     voxel, depth, feat, D, HW, XY, Z, g_logits)
  if (C == 64) {
    if (layout == LSS_BEV_NCHW_F32) LSS_BWD(1, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_BWD(1, LSS_BEV_NHWC_F32);
    else LSS_BWD(1, LSS_BEV_NHWC_BF16);
  } else {
    if (layout == LSS_BEV_NCHW_F32) LSS_BWD(2, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_BWD(2, LSS_BEV_NHWC_F32);
    else LSS_BWD(2, LSS_BEV_NHWC_BF16);
  }
#undef LSS_BWD
  return lss_launch_status();
}

// Can the region-bucketed pipeline run this problem out of the ABI's workspace?  (vox_count: the zero-between-
// calls words; vox_list: plain scratch.)  Otherwise the voxel-list pipeline below is used.
constexpr int LSS_DIRECT_CAP = 1024;     // slots per region of the direct form (the busiest region of the benched rigs
constexpr int LSS_DIRECT_OVF = 65536;     // records of the overflow list (16 B each: 1 MiB)
constexpr int LSS_DIRECT_MIN_CAP = 256;  // holds ~410 points); below this capacity the three-launch form is used instead

static bool region_plan_for(int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z, int32_t* vox_count,
                            int32_t* vox_list, LssRegionPlan* rp, void* direct_entries = nullptr,
                            unsigned long long direct_bytes = 0) {
  rp->dentries = nullptr;
  rp->cap = 0;
  rp->ovf = nullptr;
  rp->ovf_ctl = nullptr;
  rp->ovf_cap = 0;
  if (const char* e = getenv("LSS_SPLAT_LEGACY"))
    if (atoi(e) != 0) return false;
  const long long nvox = (long long)B * X * Y * Z, P = (long long)B * N * D * fH * fW;
  rp->nRx = lss_cdiv(X, LSS_REGION_SIDE);
  rp->nRy = lss_cdiv(Y, LSS_REGION_SIDE);
  rp->rps = rp->nRx * rp->nRy;
  rp->n2 = lss_region_k2_blocks(B, N, fH, fW);
  const long long nreg = (long long)B * rp->rps;
  const size_t tile_bytes = (size_t)64 * Z * C * 8 + (size_t)64 * Z * C / 8;
  // (cell index of an entry: 8 bits -> 64 * Z <= 256; the int64 tile: up to 128 KiB of the CU's 160 KiB, opted into
  // per instantiation with hipFuncSetAttribute - C = 128 and Z = 2 used to be turned away at 64 KiB and ran on the
  // voxel-list pipeline without any test noticing)
  if (rp->rps > 4096 || tile_bytes > 130 * 1024 || 64 * Z > 256 || P >= (1LL << 22) || nreg >= (1LL << 24) ||
      (long long)B * N * fH * fW >= (1LL << 23))
    return false;
  if (2 * nreg > nvox || nreg + rp->n2 + 1 > 2 * nvox) return false;
  rp->region_count = vox_count;
  rp->region_cursor = vox_count + nreg;
  rp->region_start = vox_list;
  rp->wg_absmax = reinterpret_cast<float*>(vox_list + nreg);
  // the direct form: as many slots per region as the caller's workspace holds, at most LSS_DIRECT_CAP
  if (direct_entries != nullptr && (reinterpret_cast<uintptr_t>(direct_entries) & 7) == 0 &&
      !(getenv("LSS_SPLAT_DIRECT") != nullptr && atoi(getenv("LSS_SPLAT_DIRECT")) == 0)) {
    // the overflow list takes the buffer's tail when the buckets leave room for it
    const unsigned long long ovf_bytes = (unsigned long long)LSS_DIRECT_OVF * 16;
    const bool with_list = direct_bytes >= (unsigned long long)nreg * LSS_DIRECT_MIN_CAP * 8 + ovf_bytes;
    long long cap = (long long)((direct_bytes - (with_list ? ovf_bytes : 0)) / 8) / nreg;
    if (cap > LSS_DIRECT_CAP) cap = LSS_DIRECT_CAP;
    cap &= ~7LL;
    if (cap >= LSS_DIRECT_MIN_CAP && nreg * cap < (1LL << 31) && nreg >= 3) {
      rp->dentries = reinterpret_cast<int32_t*>(direct_entries);
      rp->cap = (int)cap;
      rp->ovf_ctl = rp->region_cursor;   // (three of the zero-between-calls words the direct form does not use)
      if (with_list) {
        rp->ovf = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(direct_entries) + (size_t)nreg * cap * 8);
        rp->ovf_cap = LSS_DIRECT_OVF;
      }
    }
  }
  return true;
}

// Bytes of the direct form's entry workspace for this problem (0: the region pipeline does not take it at all).
extern "C" size_t lss_lift_splat_direct_bytes(int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z) {
  if (B <= 0 || N <= 0 || D <= 0 || fH <= 0 || fW <= 0 || X <= 0 || Y <= 0 || Z <= 0 || (C != 64 && C != 128)) return 0;
  static int32_t dummy[4];
  LssRegionPlan rp;
  if (!region_plan_for(B, N, D, fH, fW, C, X, Y, Z, dummy, dummy, &rp)) return 0;
  return (size_t)B * rp.rps * LSS_DIRECT_CAP * 8 + (size_t)LSS_DIRECT_OVF * 16;
}

// Does lss_lift_splat_forward run this problem on the region-bucketed pipeline (f32 depthnet math assumed)?  The same
// limits as region_plan_for, without touching a workspace: lets a test assert which pipeline produced its result.
extern "C" int lss_region_pipeline_ok(int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z) {
  if (B <= 0 || N <= 0 || D <= 0 || fH <= 0 || fW <= 0 || X <= 0 || Y <= 0 || Z <= 0 || (C != 64 && C != 128)) return 0;
  static int32_t dummy[4];
  LssRegionPlan rp;
  return region_plan_for(B, N, D, fH, fW, C, X, Y, Z, dummy, dummy, &rp) ? 1 : 0;
}

static int region_splat_launch(const float* feat, const int32_t* entries, const LssRegionPlan& rp, int B, int C, int X,
                               int Y, int Z, void* bev, int layout, hipStream_t st, const DirectArgs* dap = nullptr) {
  const bool direct = rp.dentries != nullptr;
  if (direct && dap == nullptr) return LSS_E_NULL;
  const DirectArgs da = direct ? *dap : DirectArgs{nullptr, nullptr, 0, 0, 0};
  if (direct) entries = rp.dentries;
  const size_t lds = (size_t)64 * Z * C * 8 + (size_t)64 * Z * C / 8;
  dim3 grid(B * rp.rps);
  const int2* en = reinterpret_cast<const int2*>(entries);
  // diagnostic runs: the splat's stamps follow the 16384 slots of the K2 || K3 launch in the same buffer
  unsigned long long* stamps = getenv("LSS_L1_STAMPS")
      ? reinterpret_cast<unsigned long long*>(strtoull(getenv("LSS_L1_STAMPS"), nullptr, 16)) + (size_t)16384 * 8 : nullptr;
  static const int centre_out = getenv("LSS_SPLAT_ORDER") == nullptr || atoi(getenv("LSS_SPLAT_ORDER")) != 0;
#define LSS_RS(CPL, LAY)                                                                                          \
  do {                                                                                                            \
    if (lds > 64 * 1024) {                                                                                        \
      static bool big[16] = {};                                                                                   \
      int dev = 0;                                                                                                \
      (void)hipGetDevice(&dev);                                                                                   \
      if (dev >= 0 && dev < 16 && !big[dev]) {                                                                    \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&region_splat_kernel<CPL, LAY, false>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024) != hipSuccess ||          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&region_splat_kernel<CPL, LAY, true>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024) != hipSuccess)            \
          return LSS_E_SHAPE;                                                                                     \
        big[dev] = true;                                                                                          \
      }                                                                                                           \
    }                                                                                                             \
    if (direct)                                                                                                   \
      hipLaunchKernelGGL((region_splat_kernel<CPL, LAY, true>), grid, dim3(256), lds, st, feat, en, rp, X, Y, Z, bev, \
                         stamps, centre_out, da);                                                                 \
    else                                                                                                          \
      hipLaunchKernelGGL((region_splat_kernel<CPL, LAY, false>), grid, dim3(256), lds, st, feat, en, rp, X, Y, Z, bev, \
                         stamps, centre_out, da);                                                                 \
  } while (0)
  if (C == 64) {
    if (layout == LSS_BEV_NCHW_F32) LSS_RS(1, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_RS(1, LSS_BEV_NHWC_F32);
    else LSS_RS(1, LSS_BEV_NHWC_BF16);
  } else {
    if (layout == LSS_BEV_NCHW_F32) LSS_RS(2, LSS_BEV_NCHW_F32);
    else if (layout == LSS_BEV_NHWC_F32) LSS_RS(2, LSS_BEV_NHWC_F32);
    else LSS_RS(2, LSS_BEV_NHWC_BF16);
  }
#undef LSS_RS
  return lss_launch_status();
}

static int lift_splat_forward_impl(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                   const float* combine, const float* trans, const float* calib_host, const float* dx,
                                   const float* bx, const float* x, const float* w, const float* bias, int B, int N,
                                   int D, int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                                   int32_t* vox_count, int32_t* vox_list, int32_t* entries, int32_t* cursor,
                                   float* depth, float* feat, void* bev, int layout, int math, void* stream,
                                   void* direct_entries = nullptr, unsigned long long direct_bytes = 0) {
  int rc;
  // Region-bucketed pipeline (3 launches: K2 || K3 + LDS region histograms, fill, region splat) whenever the
  // depthnet runs in f32 and the problem fits its limits; LSS_SPLAT_LEGACY=1 forces the voxel-list pipeline.
  LssRegionPlan rp;
  if (math == LSS_DT_F32 && (C == 64 || C == 128) && layout >= 0 && layout <= 2 && vox_count != nullptr &&
      vox_list != nullptr && entries != nullptr && bev != nullptr && B > 0 && N > 0 && D > 0 && fH > 0 && fW > 0 &&
      X > 0 && Y > 0 && Z > 0 && (reinterpret_cast<uintptr_t>(bev) & 15) == 0 &&
      region_plan_for(B, N, D, fH, fW, C, X, Y, Z, vox_count, vox_list, &rp, direct_entries, direct_bytes)) {
    rc = lss_region_depthnet_voxels(frustum, inv_post_rots, post_trans, combine, trans, calib_host, dx, bx, x, w, bias,
                                    B, N, D, fH, fW, Cin, C, X, Y, Z, voxel, depth, feat, rp, stream);
    if (rc) return rc;
    if (rp.dentries != nullptr) {  // direct form: launch 1 wrote the entries, no fill launch
      const DirectArgs da = {depth, voxel, N, D * fH * fW, fH * fW};
      return region_splat_launch(feat, entries, rp, B, C, X, Y, Z, bev, layout, lss_stream(stream), &da);
    }
    rc = lss_region_fill(voxel, depth, B, N, D, fH * fW, X, Y, Z, rp, entries, stream);
    if (rc) return rc;
    return region_splat_launch(feat, entries, rp, B, C, X, Y, Z, bev, layout, lss_stream(stream));
  }
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

// Lift-splat of depth / context tensors produced elsewhere (vovnet depth heads, CamEncodeV2): geometry + bucketing +
// splat behind one call.  Region pipeline when the problem fits it (C = 128 included), else K3 -> K4 -> K5.
static int lift_splat_from_heads_impl(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                      const float* combine, const float* trans, const float* dx, const float* bx,
                                      const float* depth, const float* feat, int B, int N, int D, int fH, int fW,
                                      int C, int X, int Y, int Z, int32_t* voxel, int32_t* vox_count,
                                      int32_t* vox_list, int32_t* entries, int32_t* cursor, void* bev, int layout,
                                      void* stream, void* direct_entries, unsigned long long direct_bytes) {
  LSS_CHECK_PTR(depth); LSS_CHECK_PTR(feat); LSS_CHECK_PTR(voxel); LSS_CHECK_PTR(vox_count); LSS_CHECK_PTR(vox_list);
  LSS_CHECK_PTR(entries); LSS_CHECK_PTR(cursor); LSS_CHECK_PTR(bev);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  if (C != 64 && C != 128) return LSS_E_SHAPE;
  if (layout < 0 || layout > 2) return LSS_E_LAYOUT;
  int rc;
  LssRegionPlan rp;
  if ((reinterpret_cast<uintptr_t>(bev) & 15) == 0 &&
      region_plan_for(B, N, D, fH, fW, C, X, Y, Z, vox_count, vox_list, &rp, direct_entries, direct_bytes)) {
    rc = lss_region_voxels_absmax(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, feat, B, N, D, fH, fW, C, X,
                                  Y, Z, voxel, rp, stream);
    if (rc) return rc;
    if (rp.dentries != nullptr) {
      const DirectArgs da = {depth, voxel, N, D * fH * fW, fH * fW};
      return region_splat_launch(feat, entries, rp, B, C, X, Y, Z, bev, layout, lss_stream(stream), &da);
    }
    rc = lss_region_fill(voxel, depth, B, N, D, fH * fW, X, Y, Z, rp, entries, stream);
    if (rc) return rc;
    return region_splat_launch(feat, entries, rp, B, C, X, Y, Z, bev, layout, lss_stream(stream));
  }
  rc = lss_points_to_voxels(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, B, N, D, fH, fW, X, Y, Z, voxel,
                            vox_count, nullptr, stream);
  if (rc) return rc;
  rc = lss_bucket_points(voxel, depth, B * N * D * fH * fW, D, fH * fW, B * X * Y * Z, vox_count, vox_list, entries,
                         cursor, stream);
  if (rc) return rc;
  return lss_lift_splat_fwd(feat, vox_list, entries, B, N, D, fH, fW, C, X, Y, Z, bev, layout, stream);
}

extern "C" int lss_lift_splat_from_heads(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                         const float* combine, const float* trans, const float* dx, const float* bx,
                                         const float* depth, const float* feat, int B, int N, int D, int fH, int fW,
                                         int C, int X, int Y, int Z, int32_t* voxel, int32_t* vox_count,
                                         int32_t* vox_list, int32_t* entries, int32_t* cursor, void* bev, int layout,
                                         void* stream) {
  return lift_splat_from_heads_impl(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, depth, feat, B, N, D, fH,
                                    fW, C, X, Y, Z, voxel, vox_count, vox_list, entries, cursor, bev, layout, stream,
                                    nullptr, 0);
}

// Descriptor form of the three entries above / below (lss_lift_splat_forward, _hostcal, _from_heads): the same
// arguments in one struct, plus the DIRECT entry workspace (lss_lift_splat_direct_bytes) that lets the region pipeline
// run in two launches instead of three (region_plan.h).  calib_host != NULL: host calibration (the four device
// calibration pointers are ignored); x == NULL: depth / feat are INPUTS (the _from_heads form).
extern "C" int lss_lift_splat_forward_desc(const lss_lift_splat_desc_t* d, void* stream) {
  LSS_CHECK_PTR(d);
  if (d->x == nullptr) {
    return lift_splat_from_heads_impl(d->frustum, d->inv_post_rots, d->post_trans, d->combine, d->trans, d->dx, d->bx,
                                      d->depth, d->feat, d->B, d->N, d->D, d->fH, d->fW, d->C, d->X, d->Y, d->Z, d->voxel,
                                      d->vox_count, d->vox_list, d->entries, d->cursor, d->bev, d->layout, stream,
                                      d->direct_entries, d->direct_bytes);
  }
  if (d->calib_host != nullptr && d->math != LSS_DT_F32) return LSS_E_LAYOUT;
  return lift_splat_forward_impl(d->frustum, d->calib_host ? nullptr : d->inv_post_rots,
                                 d->calib_host ? nullptr : d->post_trans, d->calib_host ? nullptr : d->combine,
                                 d->calib_host ? nullptr : d->trans, d->calib_host, d->dx, d->bx, d->x, d->w, d->bias, d->B,
                                 d->N, d->D, d->fH, d->fW, d->Cin, d->C, d->X, d->Y, d->Z, d->voxel, d->vox_count,
                                 d->vox_list, d->entries, d->cursor, d->depth, d->feat, d->bev, d->layout, d->math, stream,
                                 d->direct_entries, d->direct_bytes);
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
