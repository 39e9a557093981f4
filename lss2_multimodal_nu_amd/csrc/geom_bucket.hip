// K3 (frustum point -> voxel id, exact) and K4 (sort-free bucketing).
//
// This translation unit MUST be compiled with -ffp-contract=off: the voxel index
// of a point has to equal the reference's bit for bit, and torch-CPU evaluates
// every 3x3 * 3 product of get_geometry (ref: src/model_BEV_TXT.py:60,67) as
// ((m0*p0 + m1*p1) + m2*p2) with each mul/add rounded to fp32.
// tests/test_build_abi.py compiles this file to ISA and checks that the two geometry -> voxel-id kernels hold
// no v_fma / v_fmac / v_mac / v_mad.
#include "depthnet_body.h"
#include "region_plan.h"

namespace {

struct Mat3 {
  float m[9];
};

// ((m0*p0 + m1*p1) + m2*p2), every operation rounded (no contraction in this TU)
__device__ __forceinline__ float row_dot(const float* m, float p0, float p1, float p2) {
  float t = __fmul_rn(m[0], p0);
  t = __fadd_rn(t, __fmul_rn(m[1], p1));
  return __fadd_rn(t, __fmul_rn(m[2], p2));
}

// ref :92  ((geom - (bx - dx/2)) / dx).long()   - fp32 sub, true division,
// truncation toward zero.  x86 maps NaN/inf/huge to INT64_MIN (dropped); here the
// range test is made on the float quotient so those never reach a conversion:
//   trunc(q) in [0, n)  <=>  q > -1 && q < n        (q in (-1,0) truncates to 0)
// ref :99-103 in-box filter, :106-109 rank -> flat voxel id, -1 when dropped.
__device__ __forceinline__ int quantise_point(float g0, float g1, float g2,
                                              const float* __restrict__ dx,
                                              const float* __restrict__ bx, int b, int X, int Y,
                                              int Z, int* region_xy = nullptr, int nRy = 0,
                                              int* cell_in_region = nullptr) {
  const float d0 = dx[0], d1 = dx[1], d2 = dx[2];
  const float lo0 = __fsub_rn(bx[0], __fmul_rn(d0, 0.5f));
  const float lo1 = __fsub_rn(bx[1], __fmul_rn(d1, 0.5f));
  const float lo2 = __fsub_rn(bx[2], __fmul_rn(d2, 0.5f));
  const float u0 = __fdiv_rn(__fsub_rn(g0, lo0), d0);
  const float u1 = __fdiv_rn(__fsub_rn(g1, lo1), d1);
  const float u2 = __fdiv_rn(__fsub_rn(g2, lo2), d2);
  const bool kept = (u0 > -1.0f) & (u0 < (float)X) & (u1 > -1.0f) & (u1 < (float)Y) &
                    (u2 > -1.0f) & (u2 < (float)Z);
  if (!kept) return -1;
  const int ix = (int)u0, iy = (int)u1, iz = (int)u2;  // v_cvt_i32_f32 truncates
  if (region_xy) *region_xy = (ix >> 3) * nRy + (iy >> 3);  // RS = 8 cells per region side
  if (cell_in_region) *cell_in_region = (((ix & 7) << 3) | (iy & 7)) * Z + iz;
  return ((b * X + ix) * Y + iy) * Z + iz;
}

// the four per-camera calibration arrays, behind device pointers ...
struct CalPtr {
  const float* ipr; const float* pt; const float* cmb; const float* tr;
  __device__ __forceinline__ float inv_post_rot(int bn, int i) const { return ipr[bn * 9 + i]; }
  __device__ __forceinline__ float combine(int bn, int i) const { return cmb[bn * 9 + i]; }
  __device__ __forceinline__ float post_tran(int bn, int i) const { return pt[bn * 3 + i]; }
  __device__ __forceinline__ float tran(int bn, int i) const { return tr[bn * 3 + i]; }
};
// ... or carried INSIDE the kernel arguments (host calibration, <= CAL_MAX cameras): no H2D copy, no
// staging buffer, one dependent launch boundary less.  Layout [inv_post_rots | combine | post_trans | trans].
constexpr int CAL_MAX = 36;
struct CalInline {
  float v[CAL_MAX * 24];
  int n;  // cameras
  __device__ __forceinline__ float inv_post_rot(int bn, int i) const { return v[bn * 9 + i]; }
  __device__ __forceinline__ float combine(int bn, int i) const { return v[n * 9 + bn * 9 + i]; }
  __device__ __forceinline__ float post_tran(int bn, int i) const { return v[n * 18 + bn * 3 + i]; }
  __device__ __forceinline__ float tran(int bn, int i) const { return v[n * 21 + bn * 3 + i]; }
};

// ---- region bucketing (the fused inference path) -------------------------------------------------------------
// The BEV plane of a sample is cut into REGIONS of RS x RS cells (all z of a cell belong to it).  Points are
// bucketed by region, not by voxel: a K3 workgroup first counts its 256 points per region in LDS (they all
// belong to ONE sample, so rps = ceil(X/RS)*ceil(Y/RS) counters) and then issues ONE global atomic per non-empty
// region - ~50 per workgroup instead of one per point.  The region splat kernel (splat.hip) then sums a region's
// points into an LDS tile.  Workspace words (all zero between calls, like vox_count in the voxel path):
//   region_count[B*rps]   points per region            (K3 counts, the splat kernel clears)
//   region_cursor[B*rps]  slots handed out by the fill (the splat kernel clears)
constexpr int RS = 8, RS_SHIFT = 3;
struct RegionArgs {
  int32_t* region_count;
  int32_t* region_cursor;
  int32_t* region_start;  // [B*rps] exclusive scan over (sample, region), written by the fill kernel
  float* wg_absmax;       // [n2 + 1] max |feature| per K2 workgroup (plain stores); [n2] = their maximum (fill)
  int nRy, rps;
  int2* dentries;         // DIRECT form (region_plan.h): the geometry workgroups write the entries; else nullptr
  int cap, HW;            // slots per region; pixels per camera image
  int4* ovf;              // DIRECT form: the overflow list (points beyond their region's bucket), or nullptr
  int32_t* ovf_ctl;       // {records appended, regions over capacity, -}
  int ovf_cap;
};

// one thread per frustum point of camera image bn; tile_x = 256-point block within the image.
// hist != nullptr: LDS histogram of rps counters (region bucketing); every thread of the block must call.
template <class Cal>
__device__ __forceinline__ void points_to_voxels_body(
    const float* __restrict__ frustum, const Cal& cal, const float* __restrict__ dx,
    const float* __restrict__ bx, int Ncam, int DHW, int X, int Y, int Z,
    int32_t* __restrict__ voxel, int32_t* __restrict__ vox_count, float* __restrict__ geom, int tile_x,
    int bn, int* hist = nullptr, const RegionArgs* rg = nullptr) {
  const int f = tile_x * 256 + threadIdx.x;
  if (hist != nullptr) {
    for (int i = threadIdx.x; i < rg->rps; i += 256) hist[i] = 0;
    __syncthreads();
  }
  int v = -2, region = 0, cell = 0;  // -2: no point for this thread
  if (f < DHW) {
  Mat3 ipr, cmb;  // block-uniform -> scalar loads
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    ipr.m[i] = cal.inv_post_rot(bn, i);
    cmb.m[i] = cal.combine(bn, i);
  }
  const float pt0 = cal.post_tran(bn, 0), pt1 = cal.post_tran(bn, 1), pt2 = cal.post_tran(bn, 2);
  const float t0 = cal.tran(bn, 0), t1 = cal.tran(bn, 1), t2 = cal.tran(bn, 2);

  // ref :59  points = frustum - post_trans
  const float p0 = __fsub_rn(frustum[f * 3 + 0], pt0);
  const float p1 = __fsub_rn(frustum[f * 3 + 1], pt1);
  const float p2 = __fsub_rn(frustum[f * 3 + 2], pt2);
  // ref :60  inv(post_rots) @ points
  const float q0 = row_dot(ipr.m + 0, p0, p1, p2);
  const float q1 = row_dot(ipr.m + 3, p0, p1, p2);
  const float q2 = row_dot(ipr.m + 6, p0, p1, p2);
  // ref :63-65  (x*z, y*z, z)
  const float r0 = __fmul_rn(q0, q2);
  const float r1 = __fmul_rn(q1, q2);
  // ref :67-68  combine @ points + trans
  const float g0 = __fadd_rn(row_dot(cmb.m + 0, r0, r1, q2), t0);
  const float g1 = __fadd_rn(row_dot(cmb.m + 3, r0, r1, q2), t1);
  const float g2 = __fadd_rn(row_dot(cmb.m + 6, r0, r1, q2), t2);

  if (geom) {
    float* gp = geom + ((size_t)bn * DHW + f) * 3;
    gp[0] = g0; gp[1] = g1; gp[2] = g2;
  }
  v = quantise_point(g0, g1, g2, dx, bx, bn / Ncam, X, Y, Z, &region, rg ? rg->nRy : 0, &cell);
  if (v >= 0 && vox_count) atomicAdd(vox_count + v, 1);
  voxel[(size_t)bn * DHW + f] = v;
  }
  if (hist != nullptr) {
    const int b = bn / Ncam;
    const bool direct = rg->dentries != nullptr;
    int rank = 0;
    if (v >= 0) rank = atomicAdd(&hist[region], 1);  // ds_add(_rtn)_u32: count, and this point's rank in its group
    __syncthreads();
    // one global atomic per non-empty (workgroup, region); nothing per sample: ~700 workgroups adding to the
    // same word serialise at ~11 ns each (measured: +10 us on this launch), the fill kernel sums the counts instead.
    // DIRECT form: the same atomic RETURNS the group's base slot in the region's bucket.
    for (int i = threadIdx.x; i < rg->rps; i += 256) {
      const int c = hist[i];
      if (c > 0) {
        const int base = atomicAdd(rg->region_count + b * rg->rps + i, c);
        if (direct) {
          hist[i] = base;
          // exactly one group of an over-capacity region holds slot number `cap`: it counts the region in
          if (base <= rg->cap && rg->cap < base + c) atomicAdd(rg->ovf_ctl + 1, 1);
        }
      }
    }
    if (direct) {
      __syncthreads();
      bool spill = false;
      int key = 0;
      if (v >= 0) {
        const int slot = hist[region] + rank;
        const int d = f / rg->HW, pix = f - d * rg->HW;
        key = ((bn * rg->HW + pix) << 8) | cell;
        if (slot < rg->cap)
          rg->dentries[(size_t)(b * rg->rps + region) * rg->cap + slot] = make_int2(key, bn * DHW + f);
        else
          spill = true;  // the bucket is full: this point goes to the overflow list
      }
      // one global atomic per wave that has spilling points (a wave-wide ballot: every lane is here)
      const unsigned long long mask = __ballot(spill);
      if (mask != 0 && rg->ovf != nullptr) {
        const int lane = threadIdx.x & 63, leader = __builtin_ctzll(mask);
        int ob = 0;
        if (lane == leader) ob = atomicAdd(rg->ovf_ctl, (int)__builtin_popcountll(mask));
        ob = __builtin_amdgcn_readlane(ob, leader);
        if (spill) {
          const int o = ob + (int)__builtin_popcountll(mask & ((1ULL << lane) - 1ULL));
          if (o < rg->ovf_cap) rg->ovf[o] = make_int4(b * rg->rps + region, key, bn * DHW + f, 0);
        }
      }
    }
  }
}

// grid = (ceil(D*fH*fW / 256), B*N)
__global__ __launch_bounds__(256) void points_to_voxels_kernel(
    const float* __restrict__ frustum, const float* __restrict__ inv_post_rots,
    const float* __restrict__ post_trans, const float* __restrict__ combine,
    const float* __restrict__ trans, const float* __restrict__ dx,
    const float* __restrict__ bx, int Ncam, int DHW, int X, int Y, int Z,
    int32_t* __restrict__ voxel, int32_t* __restrict__ vox_count, float* __restrict__ geom) {
  const CalPtr cal = {inv_post_rots, post_trans, combine, trans};
  points_to_voxels_body(frustum, cal, dx, bx, Ncam, DHW, X, Y, Z, voxel, vox_count, geom, blockIdx.x, blockIdx.y);
}

// K2 || K3 in ONE launch: the two kernels are independent (K2 reads the trunk features, K3 the
// calibration), both are latency-bound and neither fills the chip, so their workgroups share
// it: blocks [0, n2) run the depthnet body (they are the longer ones and start first), the rest
// the geometry body.  One kernel boundary less per step and K3 hides under K2.
struct FusedK2K3Args {
  // K2
  const float* x; const float* w; const float* bias; int Cin, HW, D, C; float* depth; float* feat;
  int gx2, n2;  // pixel tiles per image, number of K2 blocks
  // K3
  const float* frustum; const float* inv_post_rots; const float* post_trans; const float* combine;
  const float* trans; const float* dx; const float* bx; int Ncam, DHW, X, Y, Z;
  int32_t* voxel; int32_t* vox_count; int gx3;
  // region bucketing (use_regions != 0): K3 blocks count per region in LDS instead of per voxel in global
  // memory, K2 blocks leave their max |feature|
  int use_regions;
  unsigned long long* stamps;  // LSS_L1_STAMPS=<hex device address, 8 u64 per workgroup>: s_memrealtime phase stamps
  int diag;  // timing-only builds (LSS_K2K3_DIAG): 1 = the K2 blocks return at once, 2 = the K3 blocks do
  int k2_xcd;  // XCD-aware order of the row-split depthnet workgroups (LSS_K2_XCD=0: id order, for A/B)
  RegionArgs rg;
};

template <int NT>
__global__ __launch_bounds__(256) void depthnet_and_voxels_kernel(FusedK2K3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int id = blockIdx.x;
  if (id < a.n2) {
    lss_depthnet::depthnet_softmax_f32_body<NT>(a.x, a.w, a.bias, a.Cin, a.HW, a.D, a.C, a.depth, a.feat, id % a.gx2,
                                                id / a.gx2, lds, a.use_regions ? a.rg.wg_absmax + id : nullptr);
  } else {
    const int k = id - a.n2;
    const CalPtr cal = {a.inv_post_rots, a.post_trans, a.combine, a.trans};
    points_to_voxels_body(a.frustum, cal, a.dx, a.bx, a.Ncam, a.DHW, a.X, a.Y, a.Z, a.voxel, a.vox_count, nullptr,
                          k % a.gx3, k / a.gx3, a.use_regions ? reinterpret_cast<int*>(lds) : nullptr, &a.rg);
  }
}

// the same with the calibration inside the kernel arguments
template <int NT>
__global__ __launch_bounds__(256) void depthnet_and_voxels_hostcal_kernel(FusedK2K3Args a, CalInline cal) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int id = blockIdx.x;
  if (id < a.n2) {
    lss_depthnet::depthnet_softmax_f32_body<NT>(a.x, a.w, a.bias, a.Cin, a.HW, a.D, a.C, a.depth, a.feat, id % a.gx2,
                                                id / a.gx2, lds, a.use_regions ? a.rg.wg_absmax + id : nullptr);
  } else {
    const int k = id - a.n2;
    points_to_voxels_body(a.frustum, cal, a.dx, a.bx, a.Ncam, a.DHW, a.X, a.Y, a.Z, a.voxel, a.vox_count, nullptr,
                          k % a.gx3, k / a.gx3, a.use_regions ? reinterpret_cast<int*>(lds) : nullptr, &a.rg);
  }
}

// Region pipeline: K2 with its output rows split over two workgroups per pixel tile (depthnet_rows_f32_body).  Blocks
// [0, n2x) are K2's ((tile, half), half 0 = the D depth bins + softmax, half 1 = the C context channels): the long
// ones (~10-13 us) go first, ALL of them resident at once - 3 waves per SIMD = 768 workgroup slots for the 528 - and
// the short geometry workgroups (K3, ~2 us each) flow through the slots that are left.  In-kernel stamps
// (tools/bench_l1.py --stamps) showed what the other arrangements cost: with 2 waves per SIMD (512 slots) 16 of the
// K2 workgroups started only when a first-round one retired (launch end 20 us instead of 14) and the geometry
// started 14 us in; with the geometry first every K2 workgroup started 3 us late.
template <int ND, int NC, bool HOSTCAL>
__global__ __launch_bounds__(256, 3) void depthnet_rows_and_voxels_kernel(FusedK2K3Args a, CalInline cal, int n2x) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int id = blockIdx.x;
  if (a.diag != 0 && (id < n2x) == (a.diag == 1)) return;
  unsigned long long* st = a.stamps ? a.stamps + (size_t)id * 8 : nullptr;
  if (st != nullptr && threadIdx.x == 0) st[0] = __builtin_amdgcn_s_memrealtime();
  if (id < n2x) {
    // XCD-aware order of the depthnet workgroups (cdna_hip_programming.md T1): workgroups are dealt round-robin to the 8
    // XCDs, so XCD k takes the k-th contiguous eighth of the (tile, row half) list = whole camera images.  The two row
    // halves of a pixel tile read the same x lines, and so do neighbouring tiles (a 128-B line of x holds 32 pixels of
    // one channel and a tile is 16): in id order those four workgroups sat on four XCDs and each L2 fetched the lines
    // for itself - FETCH_SIZE 37 MB for 8.9 MB of trunk features (profiles/r03_hbm_traffic.json).
    int k2 = id;
    if (a.k2_xcd) {
      const int xcd = id & 7, q8 = n2x >> 3, r8 = n2x & 7;
      k2 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    }
    const int tile = k2 >> 1;
    if ((k2 & 1) == 0)
      lss_depthnet::depthnet_rows_f32_body<ND>(a.x, a.w, a.bias, 0, a.D, true, a.Cin, a.HW, a.D, a.C, a.depth, a.feat,
                                               tile % a.gx2, tile / a.gx2, lds, nullptr, st);
    else
      lss_depthnet::depthnet_rows_f32_body<NC>(a.x, a.w, a.bias, a.D, a.C, false, a.Cin, a.HW, a.D, a.C, a.depth,
                                               a.feat, tile % a.gx2, tile / a.gx2, lds, a.rg.wg_absmax + tile, st);
  } else {
    const int k = id - n2x;
    if (HOSTCAL) {
      points_to_voxels_body(a.frustum, cal, a.dx, a.bx, a.Ncam, a.DHW, a.X, a.Y, a.Z, a.voxel, a.vox_count, nullptr,
                            k % a.gx3, k / a.gx3, reinterpret_cast<int*>(lds), &a.rg);
    } else {
      const CalPtr calp = {a.inv_post_rots, a.post_trans, a.combine, a.trans};
      points_to_voxels_body(a.frustum, calp, a.dx, a.bx, a.Ncam, a.DHW, a.X, a.Y, a.Z, a.voxel, a.vox_count, nullptr,
                            k % a.gx3, k / a.gx3, reinterpret_cast<int*>(lds), &a.rg);
    }
  }
  if (st != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) st[3] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ __launch_bounds__(256) void geom_to_voxels_kernel(const float* __restrict__ geom,
                                                             const float* __restrict__ dx,
                                                             const float* __restrict__ bx, int P,
                                                             int pps, int X, int Y, int Z,
                                                             int32_t* __restrict__ voxel,
                                                             int32_t* __restrict__ vox_count) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int v = quantise_point(geom[(size_t)p * 3], geom[(size_t)p * 3 + 1], geom[(size_t)p * 3 + 2],
                               dx, bx, p / pps, X, Y, Z);
  if (v >= 0 && vox_count) atomicAdd(vox_count + v, 1);
  voxel[p] = v;
}

// One 1024-thread workgroup per 1024 voxels: scan the counts (wave scan + LDS over
// the 16 wave totals) and reserve the group's slice of the entry list with ONE
// atomic on the cursor.  Group order in the list is arbitrary; inside a group
// the voxels' slices are contiguous and in voxel order, which is what lets K5
// stream a wave's 8 consecutive voxels as one contiguous range.
__global__ __launch_bounds__(1024) void bucket_alloc_kernel(const int32_t* __restrict__ vox_count,
                                                            int nvox, int32_t* __restrict__ vox_list,
                                                            int32_t* __restrict__ cursor) {
  __shared__ int wave_tot[16];
  __shared__ int group_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = blockIdx.x * 1024 + threadIdx.x;
  const int c = (v < nvox) ? vox_count[v] : 0;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = wave_tot[i];
      wave_tot[i] = run;
      run += t;
    }
    group_base = run > 0 ? atomicAdd(cursor, run) : 0;
  }
  __syncthreads();
  if (v < nvox) {
    vox_list[2 * v + 0] = group_base + wave_tot[wave] + incl - c;
    vox_list[2 * v + 1] = c;
  }
}

// One thread per point: take a slot of the voxel's slice by counting the voxel's
// counter back down to zero (so vox_count is all-zero again for the next call) and
// write the list entry {point id, depth weight}: K5 then needs no dependent
// depth load.
__global__ __launch_bounds__(256) void bucket_fill_kernel(const int32_t* __restrict__ voxel, int P,
                                                          int D, int HW,
                                                          const float* __restrict__ depth,
                                                          int32_t* __restrict__ vox_count,
                                                          const int32_t* __restrict__ vox_list,
                                                          int2* __restrict__ entries,
                                                          int32_t* __restrict__ cursor) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0) *cursor = 0;  // alloc (previous kernel on the stream) is done with it
  if (p >= P) return;
  const int v = voxel[p];
  if (v < 0) return;
  const float w = depth ? depth[p] : 1.0f;  // depth is (BN, D, HW): flat index == point id
  const int slot = atomicSub(vox_count + v, 1) - 1;
  // key = (feature row << 7) | depth bin: unique per point, gives K5 the row without a
  // division and a fixed order to sum in.  p = (bn*D + d)*HW + pix, row = bn*HW + pix
  const int bd = p / HW, pix = p - bd * HW;
  const int bn = bd / D, d = bd - bn * D;
  entries[vox_list[2 * v] + slot] = make_int2(((bn * HW + pix) << 7) | d, __builtin_bit_cast(int, w));
}

// Region fill (second launch of the fused inference path).  grid = (ceil(DHW/256), B*N), one thread per point:
//   1. the workgroup scans the rps region counts of ITS sample in LDS (exclusive) and adds the totals of the
//      samples before it: region_start; workgroup (0, first camera) of each sample also writes it out for the splat;
//   2. every kept point takes a rank inside its (workgroup, region) group with an LDS atomic;
//   3. one GLOBAL atomic per non-empty (workgroup, region) reserves the group's slots in the region's bucket;
//   4. the entry {(feature row << 8) | cell in region, depth weight} goes to start + base + rank.
// Entries of a region end up contiguous, in arbitrary order - the splat's fixed-point sums do not depend on it.
__global__ __launch_bounds__(256) void region_fill_kernel(const int32_t* __restrict__ voxel,
                                                          const float* __restrict__ depth, int Ncam, int D,
                                                          int HW, int X, int Y, int Z, RegionArgs rg, int n2,
                                                          int2* __restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) int fl[];  // [rps] start | [rps] count -> base
  int* rstart = fl;
  int* rcnt = fl + rg.rps;
  __shared__ int wave_tot[4], wave_before[4];
  __shared__ float wave_max[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bn = blockIdx.y, b = bn / Ncam, DHW = D * HW;
  const int f = blockIdx.x * 256 + tid;
  // -- 1. exclusive scan of this sample's region counts (each thread owns a contiguous run of regions)
  const int per = (rg.rps + 255) / 256;
  const int i0 = tid * per;
  int run = 0;
  for (int i = i0; i < min(i0 + per, rg.rps); ++i) run += rg.region_count[b * rg.rps + i];
  int incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  // points of the samples before this one (their buckets come first): sum of their region counts
  int before = 0;
  for (int i = tid; i < b * rg.rps; i += 256) before += rg.region_count[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) wave_before[wave] = before;
  __syncthreads();
  int base = wave_before[0] + wave_before[1] + wave_before[2] + wave_before[3];
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  int acc = base + incl - run;
  const bool writer = blockIdx.x == 0 && bn == b * Ncam;
  for (int i = i0; i < min(i0 + per, rg.rps); ++i) {
    rstart[i] = acc;
    if (writer) rg.region_start[b * rg.rps + i] = acc;
    acc += rg.region_count[b * rg.rps + i];
  }
  for (int i = tid; i < rg.rps; i += 256) rcnt[i] = 0;
  __syncthreads();
  // -- 2. rank inside the (workgroup, region) group
  int v = -1, region = 0, cell = 0, rank = 0;
  if (f < DHW) v = voxel[(size_t)bn * DHW + f];
  if (v >= 0) {
    const int iz = v % Z, cxy = v / Z - b * X * Y;  // v = ((b*X + ix)*Y + iy)*Z + iz
    const int ix = cxy / Y, iy = cxy - ix * Y;
    region = (ix >> RS_SHIFT) * rg.nRy + (iy >> RS_SHIFT);
    cell = (((ix & (RS - 1)) << RS_SHIFT) | (iy & (RS - 1))) * Z + iz;
    rank = atomicAdd(&rcnt[region], 1);  // ds_add_rtn_u32
  }
  __syncthreads();
  // -- 3. reserve the groups' slots: count -> base offset inside the region's bucket
  for (int i = tid; i < rg.rps; i += 256) {
    const int c = rcnt[i];
    if (c > 0) rcnt[i] = atomicAdd(rg.region_cursor + b * rg.rps + i, c);
  }
  __syncthreads();
  // -- the first workgroup also reduces K2's per-workgroup max |feature| to ONE word for the splat (slot n2)
  if (blockIdx.x == 0 && bn == 0) {
    float m = 0.f;
    for (int i = tid; i < n2; i += 256) m = fmaxf(m, rg.wg_absmax[i]);
    m = lss_wave_max(m);
    if (lane == 0) wave_max[wave] = m;
    __syncthreads();
    if (tid == 0) rg.wg_absmax[n2] = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
  }
  // -- 4. the entry
  if (v >= 0) {
    const int d = f / HW, pix = f - d * HW;
    const float w = depth[(size_t)bn * DHW + f];  // depth is (BN, D, HW): flat index == point id
    entries[rstart[region] + rcnt[region] + rank] = make_int2(((bn * HW + pix) << 8) | cell, __builtin_bit_cast(int, w));
  }
}

// API-compat segmented sum (QuickCumsum.forward): one wave per run, lane = channel.
__global__ __launch_bounds__(256) void segmented_sum_kernel(const float* __restrict__ x,
                                                            const int32_t* __restrict__ seg_start,
                                                            int M, int C, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int m = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (m >= M) return;
  const int s = seg_start[m], e = seg_start[m + 1];
  for (int c = lane; c < C; c += 64) {
    float acc = 0.f;
    for (int r = s; r < e; ++r) acc += x[(size_t)r * C + c];
    y[(size_t)m * C + c] = acc;
  }
}

}  // namespace

extern "C" int lss_points_to_voxels(const float* frustum, const float* inv_post_rots,
                                    const float* post_trans, const float* combine,
                                    const float* trans, const float* dx, const float* bx, int B,
                                    int N, int D, int fH, int fW, int X, int Y, int Z,
                                    int32_t* voxel, int32_t* vox_count, float* geom, void* stream) {
  LSS_CHECK_PTR(frustum); LSS_CHECK_PTR(inv_post_rots); LSS_CHECK_PTR(post_trans);
  LSS_CHECK_PTR(combine); LSS_CHECK_PTR(trans); LSS_CHECK_PTR(dx); LSS_CHECK_PTR(bx);
  LSS_CHECK_PTR(voxel);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  const long long DHW = (long long)D * fH * fW;
  const long long nvox = (long long)B * X * Y * Z;
  if (DHW * B * N >= (1LL << 31) || nvox >= (1LL << 31) || (long long)B * N > 65535 ||
      X >= (1 << 24) || Y >= (1 << 24) || Z >= (1 << 24))
    return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(DHW, 256), B * N);
  hipLaunchKernelGGL(points_to_voxels_kernel, grid, dim3(256), 0, lss_stream(stream), frustum,
                     inv_post_rots, post_trans, combine, trans, dx, bx, N, (int)DHW, X, Y, Z,
                     voxel, vox_count, geom);
  return lss_launch_status();
}

extern "C" int lss_geom_to_voxels(const float* geom, const float* dx, const float* bx, int B,
                                  int pts_per_sample, int X, int Y, int Z, int32_t* voxel,
                                  int32_t* vox_count, void* stream) {
  LSS_CHECK_PTR(geom); LSS_CHECK_PTR(dx); LSS_CHECK_PTR(bx); LSS_CHECK_PTR(voxel);
  LSS_CHECK_POS(B); LSS_CHECK_POS(pts_per_sample); LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z);
  const long long P = (long long)B * pts_per_sample;
  if (P >= (1LL << 31) || (long long)B * X * Y * Z >= (1LL << 31) || X >= (1 << 24) ||
      Y >= (1 << 24) || Z >= (1 << 24))
    return LSS_E_SHAPE;
  hipLaunchKernelGGL(geom_to_voxels_kernel, dim3(lss_cdiv(P, 256)), dim3(256), 0,
                     lss_stream(stream), geom, dx, bx, (int)P, pts_per_sample, X, Y, Z, voxel,
                     vox_count);
  return lss_launch_status();
}

extern "C" int lss_bucket_points(const int32_t* voxel, const float* depth, int P, int D, int HW,
                                 int nvox, int32_t* vox_count, int32_t* vox_list, int32_t* entries,
                                 int32_t* cursor, void* stream) {
  LSS_CHECK_PTR(voxel); LSS_CHECK_PTR(vox_count); LSS_CHECK_PTR(vox_list);
  LSS_CHECK_PTR(entries); LSS_CHECK_PTR(cursor);
  LSS_CHECK_POS(P); LSS_CHECK_POS(nvox); LSS_CHECK_POS(D); LSS_CHECK_POS(HW);
  // key packing: depth bin in 7 bits, feature row in the 24 bits above
  if (D > 128 || P % (D * HW) != 0 || (long long)(P / D) >= (1LL << 24)) return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(entries) & 7) != 0) return LSS_E_ALIGN;
  hipLaunchKernelGGL(bucket_alloc_kernel, dim3(lss_cdiv(nvox, 1024)), dim3(1024), 0,
                     lss_stream(stream), vox_count, nvox, vox_list, cursor);
  hipLaunchKernelGGL(bucket_fill_kernel, dim3(lss_cdiv(P, 256)), dim3(256), 0,
                     lss_stream(stream), voxel, P, D, HW, depth, vox_count, vox_list,
                     reinterpret_cast<int2*>(entries), cursor);
  return lss_launch_status();
}

extern "C" int lss_segmented_sum(const float* x, const int32_t* seg_start, int M, int C, float* y,
                                 void* stream) {
  LSS_CHECK_PTR(x); LSS_CHECK_PTR(seg_start); LSS_CHECK_PTR(y);
  LSS_CHECK_POS(C);
  if (M < 0) return LSS_E_SHAPE;
  if (M == 0) return 0;
  hipLaunchKernelGGL(segmented_sum_kernel, dim3(lss_cdiv((long long)M * 64, 256)), dim3(256), 0,
                     lss_stream(stream), x, seg_start, M, C, y);
  return lss_launch_status();
}

extern "C" int lss_abi_version(void) { return LSS_ABI_VERSION; }

extern "C" const char* lss_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case LSS_E_NULL: return "lss: required pointer is NULL";
    case LSS_E_SHAPE: return "lss: size out of range for this kernel";
    case LSS_E_LAYOUT: return "lss: unknown layout or dtype";
    case LSS_E_ALIGN: return "lss: pointer not sufficiently aligned";
    case LSS_E_WORKSPACE: return "lss: workspace too small";
    case LSS_E_RCCL_BASE: return "lss: librccl could not be resolved in this process";
    default: break;
  }
  if (code < LSS_E_RCCL_BASE && code > LSS_E_RCCL_BASE - 64) return "lss: RCCL call failed (ncclResult_t = LSS_E_RCCL_BASE - code)";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "lss: unknown error";
}

// K3 (voxel ids + histogram) and K2 (depthnet + softmax, f32 MFMA) as one launch; arguments as
// lss_points_to_voxels (without geom) and lss_depthnet_softmax_fwd (math = LSS_DT_F32).
static int depthnet_voxels_impl(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                const float* combine, const float* trans, const float* calib_host, const float* dx,
                                const float* bx, const float* x, const float* w, const float* bias, int B, int N,
                                int D, int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                                int32_t* vox_count, float* depth, float* feat, void* stream,
                                const LssRegionPlan* plan = nullptr) {
  LSS_CHECK_PTR(frustum);
  if (calib_host == nullptr) {
    LSS_CHECK_PTR(inv_post_rots); LSS_CHECK_PTR(post_trans); LSS_CHECK_PTR(combine); LSS_CHECK_PTR(trans);
  } else if (B * N > CAL_MAX) {
    return LSS_E_SHAPE;
  }
  LSS_CHECK_PTR(dx); LSS_CHECK_PTR(bx); LSS_CHECK_PTR(voxel); LSS_CHECK_PTR(x);
  LSS_CHECK_PTR(w); LSS_CHECK_PTR(bias); LSS_CHECK_PTR(depth); LSS_CHECK_PTR(feat);
  LSS_CHECK_POS(B); LSS_CHECK_POS(N); LSS_CHECK_POS(D); LSS_CHECK_POS(fH); LSS_CHECK_POS(fW);
  LSS_CHECK_POS(X); LSS_CHECK_POS(Y); LSS_CHECK_POS(Z); LSS_CHECK_POS(Cin); LSS_CHECK_POS(C);
  const long long DHW = (long long)D * fH * fW, P = DHW * B * N, nvox = (long long)B * X * Y * Z;
  if (P >= (1LL << 31) || nvox >= (1LL << 31) || B * N > 65535 || Cin % 64 != 0 || X >= (1 << 24) ||
      Y >= (1 << 24) || Z >= (1 << 24))
    return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) return LSS_E_ALIGN;
  FusedK2K3Args a;
  a.x = x; a.w = w; a.bias = bias; a.Cin = Cin; a.HW = fH * fW; a.D = D; a.C = C; a.depth = depth; a.feat = feat;
  a.gx2 = lss_cdiv(a.HW, lss_depthnet::PIX);
  a.n2 = a.gx2 * B * N;
  a.frustum = frustum; a.inv_post_rots = inv_post_rots; a.post_trans = post_trans; a.combine = combine;
  a.trans = trans; a.dx = dx; a.bx = bx; a.Ncam = N; a.DHW = (int)DHW; a.X = X; a.Y = Y; a.Z = Z;
  a.voxel = voxel; a.vox_count = vox_count;
  a.use_regions = plan != nullptr;
  a.diag = getenv("LSS_K2K3_DIAG") ? atoi(getenv("LSS_K2K3_DIAG")) : 0;
  a.k2_xcd = getenv("LSS_K2_XCD") == nullptr || atoi(getenv("LSS_K2_XCD")) != 0;
  a.stamps = getenv("LSS_L1_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("LSS_L1_STAMPS"), nullptr, 16))
                                     : nullptr;
  if (plan != nullptr) {
    if (plan->n2 != a.n2 || plan->rps != plan->nRx * plan->nRy) return LSS_E_WORKSPACE;
    a.rg.region_count = plan->region_count; a.rg.region_cursor = plan->region_cursor;
    a.rg.region_start = plan->region_start;
    a.rg.wg_absmax = plan->wg_absmax; a.rg.nRy = plan->nRy; a.rg.rps = plan->rps;
    a.rg.dentries = reinterpret_cast<int2*>(plan->dentries); a.rg.cap = plan->cap; a.rg.HW = fH * fW;
    a.rg.ovf = reinterpret_cast<int4*>(plan->ovf); a.rg.ovf_ctl = plan->ovf_ctl; a.rg.ovf_cap = plan->ovf_cap;
  } else {
    a.rg = RegionArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0};
  }
  a.gx3 = lss_cdiv(DHW, 256);
  const long long nblk = (long long)a.n2 + (long long)a.gx3 * B * N;
  if (nblk >= (1LL << 31)) return LSS_E_SHAPE;
  const int NT = (D + C + 15) / 16;
  size_t lds_bytes = (size_t)5 * NT * 16 * lss_depthnet::LDS_LD * sizeof(float);
  if (plan != nullptr && (size_t)plan->rps * sizeof(int) > lds_bytes) lds_bytes = (size_t)plan->rps * sizeof(int);
  if (lds_bytes > 64 * 1024) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  CalInline cal;
  if (calib_host != nullptr) {
    cal.n = B * N;
    for (int i = 0; i < B * N * 24; ++i) cal.v[i] = calib_host[i];
  }
  // the row-split K2 (region pipeline, the shapes of the LSS depthnet: D <= 48, C <= 64, Cin a multiple of 128)
  static const bool rows_off = getenv("LSS_K2_ROWS") != nullptr && atoi(getenv("LSS_K2_ROWS")) == 0;
  const int nd16 = (D + 15) / 16;
  if (plan != nullptr && !rows_off && (nd16 == 3 || nd16 == 4) && (C + 15) / 16 == 4 && Cin % 128 == 0 &&
      ((size_t)D * Cin * sizeof(float)) % 16 == 0) {
    const long long nb = nblk + a.n2;
    if (nb >= (1LL << 31)) return LSS_E_SHAPE;
    size_t lb = (size_t)5 * 4 * 16 * lss_depthnet::LDS_LD * sizeof(float);
    if ((size_t)plan->rps * sizeof(int) > lb) lb = (size_t)plan->rps * sizeof(int);
#define LSS_ROWS(ND, HC)                                                                                          \
    hipLaunchKernelGGL((depthnet_rows_and_voxels_kernel<ND, 4, HC>), dim3((unsigned)nb), dim3(256), lb, st, a, cal, \
                       2 * a.n2)
    if (nd16 == 3) {  // D <= 48: the 41 bins of the 352 x 128 configurations
      if (calib_host != nullptr) LSS_ROWS(3, true); else LSS_ROWS(3, false);
    } else {          // D <= 64: the 60 bins of the 704 x 256 configuration
      if (calib_host != nullptr) LSS_ROWS(4, true); else LSS_ROWS(4, false);
    }
#undef LSS_ROWS
    return lss_launch_status();
  }
#define LSS_F_CASE(n)                                                                                          \
  case n:                                                                                                      \
    if (calib_host != nullptr)                                                                                 \
      hipLaunchKernelGGL(depthnet_and_voxels_hostcal_kernel<n>, dim3((unsigned)nblk), dim3(256), lds_bytes, st, a, \
                         cal);                                                                                 \
    else                                                                                                       \
      hipLaunchKernelGGL(depthnet_and_voxels_kernel<n>, dim3((unsigned)nblk), dim3(256), lds_bytes, st, a);    \
    break;
  switch (NT) {
    LSS_F_CASE(1) LSS_F_CASE(2) LSS_F_CASE(3) LSS_F_CASE(4) LSS_F_CASE(5) LSS_F_CASE(6) LSS_F_CASE(7) LSS_F_CASE(8)
    LSS_F_CASE(9) LSS_F_CASE(10) LSS_F_CASE(11) LSS_F_CASE(12)
    default: return LSS_E_SHAPE;
  }
#undef LSS_F_CASE
  return lss_launch_status();
}

extern "C" int lss_depthnet_voxels_fwd(const float* frustum, const float* inv_post_rots, const float* post_trans,
                                       const float* combine, const float* trans, const float* dx, const float* bx,
                                       const float* x, const float* w, const float* bias, int B, int N, int D, int fH,
                                       int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                                       int32_t* vox_count, float* depth, float* feat, void* stream) {
  return depthnet_voxels_impl(frustum, inv_post_rots, post_trans, combine, trans, nullptr, dx, bx, x, w, bias, B, N, D,
                              fH, fW, Cin, C, X, Y, Z, voxel, vox_count, depth, feat, stream);
}

// calib_host: HOST pointer to B*N*24 floats laid out [inv_post_rots (B*N*9) | combine (B*N*9) | post_trans (B*N*3) |
// trans (B*N*3)] (= data.CalibrationPack.buffer); read during this call and shipped inside the kernel arguments.
extern "C" int lss_depthnet_voxels_hostcal_fwd(const float* frustum, const float* calib_host, const float* dx,
                                               const float* bx, const float* x, const float* w, const float* bias,
                                               int B, int N, int D, int fH, int fW, int Cin, int C, int X, int Y,
                                               int Z, int32_t* voxel, int32_t* vox_count, float* depth, float* feat,
                                               void* stream) {
  LSS_CHECK_PTR(calib_host);
  return depthnet_voxels_impl(frustum, nullptr, nullptr, nullptr, nullptr, calib_host, dx, bx, x, w, bias, B, N, D, fH,
                              fW, Cin, C, X, Y, Z, voxel, vox_count, depth, feat, stream);
}

int lss_region_k2_blocks(int B, int N, int fH, int fW) { return lss_cdiv(fH * fW, lss_depthnet::PIX) * B * N; }

int lss_region_depthnet_voxels(const float* frustum, const float* inv_post_rots, const float* post_trans,
                               const float* combine, const float* trans, const float* calib_host, const float* dx,
                               const float* bx, const float* x, const float* w, const float* bias, int B, int N, int D,
                               int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel, float* depth,
                               float* feat, const LssRegionPlan& plan, void* stream) {
  // vox_count = nullptr: no per-voxel histogram, no per-point global atomics on this path
  return depthnet_voxels_impl(frustum, inv_post_rots, post_trans, combine, trans, calib_host, dx, bx, x, w, bias, B, N,
                              D, fH, fW, Cin, C, X, Y, Z, voxel, nullptr, depth, feat, stream, &plan);
}

// Region pipeline for depth / context tensors that OTHER kernels produced (the vovnet depth heads + CamEncodeV2, ref
// src/model_vovnet_transformer.py:22-122): launch 1 without the depthnet.  Blocks [0, n2): max |finite feature| of one
// 16-pixel tile of a camera image (the slots the depthnet workgroups fill on the fused path: the fill kernel reduces
// them to the scale of the fixed-point splat); blocks [n2, ...): the geometry with its LDS region histograms.
__global__ __launch_bounds__(256) void absmax_and_voxels_kernel(FusedK2K3Args a, int C) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int id = blockIdx.x;
  if (id < a.n2) {
    const int tile = id % a.gx2, bn = id / a.gx2, pix0 = tile * lss_depthnet::PIX;
    const int npix = min(lss_depthnet::PIX, a.HW - pix0);
    const float* f = a.feat + ((size_t)bn * a.HW + pix0) * C;
    float amax = 0.f;
    for (int e = threadIdx.x; e < npix * C; e += 256) {
      const float av = fabsf(f[e]);
      amax = fmaxf(amax, av <= 3.0e38f ? av : 0.f);
    }
    amax = lss_wave_max(amax);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) a.rg.wg_absmax[id] = fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
  } else {
    const int k = id - a.n2;
    const CalPtr calp = {a.inv_post_rots, a.post_trans, a.combine, a.trans};
    points_to_voxels_body(a.frustum, calp, a.dx, a.bx, a.Ncam, a.DHW, a.X, a.Y, a.Z, a.voxel, nullptr, nullptr, k % a.gx3,
                          k / a.gx3, reinterpret_cast<int*>(lds), &a.rg);
  }
}

int lss_region_voxels_absmax(const float* frustum, const float* inv_post_rots, const float* post_trans,
                             const float* combine, const float* trans, const float* dx, const float* bx,
                             const float* feat, int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z,
                             int32_t* voxel, const LssRegionPlan& plan, void* stream) {
  LSS_CHECK_PTR(frustum); LSS_CHECK_PTR(inv_post_rots); LSS_CHECK_PTR(post_trans); LSS_CHECK_PTR(combine);
  LSS_CHECK_PTR(trans); LSS_CHECK_PTR(dx); LSS_CHECK_PTR(bx); LSS_CHECK_PTR(feat); LSS_CHECK_PTR(voxel);
  const long long DHW = (long long)D * fH * fW;
  if (DHW * B * N >= (1LL << 31) || B * N > 65535) return LSS_E_SHAPE;
  FusedK2K3Args a = {};
  a.feat = const_cast<float*>(feat); a.HW = fH * fW; a.D = D; a.C = C;
  a.gx2 = lss_cdiv(a.HW, lss_depthnet::PIX);
  a.n2 = a.gx2 * B * N;
  if (plan.n2 != a.n2 || plan.rps != plan.nRx * plan.nRy) return LSS_E_WORKSPACE;
  a.frustum = frustum; a.inv_post_rots = inv_post_rots; a.post_trans = post_trans; a.combine = combine;
  a.trans = trans; a.dx = dx; a.bx = bx; a.Ncam = N; a.DHW = (int)DHW; a.X = X; a.Y = Y; a.Z = Z;
  a.voxel = voxel; a.vox_count = nullptr; a.use_regions = 1;
  a.rg.region_count = plan.region_count; a.rg.region_cursor = plan.region_cursor; a.rg.region_start = plan.region_start;
  a.rg.wg_absmax = plan.wg_absmax; a.rg.nRy = plan.nRy; a.rg.rps = plan.rps;
  a.rg.dentries = reinterpret_cast<int2*>(plan.dentries); a.rg.cap = plan.cap; a.rg.HW = fH * fW;
  a.rg.ovf = reinterpret_cast<int4*>(plan.ovf); a.rg.ovf_ctl = plan.ovf_ctl; a.rg.ovf_cap = plan.ovf_cap;
  a.gx3 = lss_cdiv(DHW, 256);
  const long long nblk = (long long)a.n2 + (long long)a.gx3 * B * N;
  if (nblk >= (1LL << 31)) return LSS_E_SHAPE;
  size_t lds_bytes = (size_t)plan.rps * sizeof(int);
  if (lds_bytes < 64) lds_bytes = 64;
  if (lds_bytes > 64 * 1024) return LSS_E_SHAPE;
  hipLaunchKernelGGL(absmax_and_voxels_kernel, dim3((unsigned)nblk), dim3(256), lds_bytes, lss_stream(stream), a, C);
  return lss_launch_status();
}

int lss_region_fill(const int32_t* voxel, const float* depth, int B, int N, int D, int HW, int X, int Y, int Z,
                    const LssRegionPlan& plan, int32_t* entries, void* stream) {
  LSS_CHECK_PTR(voxel); LSS_CHECK_PTR(depth); LSS_CHECK_PTR(entries);
  const long long DHW = (long long)D * HW;
  // key packing: cell-in-region in 8 bits (RS*RS*Z <= 256), feature row in the 23 bits above
  if (LSS_REGION_SIDE * LSS_REGION_SIDE * Z > 256 || (long long)B * N * HW >= (1LL << 23) || B * N > 65535)
    return LSS_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(entries) & 7) != 0) return LSS_E_ALIGN;
  RegionArgs rg{plan.region_count, plan.region_cursor, plan.region_start, plan.wg_absmax, plan.nRy, plan.rps, nullptr, 0, HW};
  const size_t lds_bytes = (size_t)2 * plan.rps * sizeof(int);
  if (lds_bytes > 64 * 1024) return LSS_E_SHAPE;
  dim3 grid(lss_cdiv(DHW, 256), B * N);
  hipLaunchKernelGGL(region_fill_kernel, grid, dim3(256), lds_bytes, lss_stream(stream), voxel, depth, N, D, HW, X, Y,
                     Z, rg, plan.n2, reinterpret_cast<int2*>(entries));
  return lss_launch_status();
}
