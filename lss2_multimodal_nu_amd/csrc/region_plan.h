// Internal (not part of the C ABI): the region-bucketed form of the fused lift-splat pipeline.
//   launch 1  K2 || K3   depthnet + softmax  ||  frustum points -> voxel ids + per-region LDS histograms
//   launch 2  fill       per-(workgroup, region) slot reservation, entries grouped by region
//   launch 3  splat      one workgroup per region: LDS fixed-point segmented sums -> coalesced BEV stores
// DIRECT form (two launches, when the caller hands over the larger entry workspace): launch 1's geometry workgroups
// reserve their slots in the regions' fixed-capacity buckets with the one global atomic per (workgroup, region) they
// issue anyway and write the entries; the fill launch and its kernel boundary (6.6 + ~1.5 us of a 45-us level) go.
// geom_bucket.hip implements launches 1-2, splat.hip launch 3 and the dispatch.
#pragma once
#include <stdint.h>

struct LssRegionPlan {
  int32_t* region_count;   // [B*rps]  zero between calls
  int32_t* region_cursor;  // [B*rps]  zero between calls
  int32_t* region_start;   // [B*rps]
  float* wg_absmax;        // [n2 + 1]: per K2 workgroup, and their maximum (written by the fill kernel)
  int nRx, nRy, rps, n2;
  // DIRECT form (round 4; dentries != nullptr): launch 1 writes the entries itself, at FIXED per-region offsets -
  // region r owns dentries[r * cap .. (r + 1) * cap) as {(feature row << 8) | cell in region, point id} - and launch 2
  // (the fill) does not exist: the splat reads region_count[r], its slots, and depth[point id].  A region with more
  // than `cap` points keeps its first `cap` there; the rest goes to the overflow list below.
  int32_t* dentries;
  int cap;
  // ... and the points that did not fit go to ONE overflow list behind the buckets: {region, key, point id, -} records
  // appended with one global atomic per WAVE that has any.  ovf_ctl (three of the zero-between-calls words: the
  // region_cursor words, which the direct form does not use) = {records appended, regions over capacity, such regions
  // the splat has finished}; the last of those splat workgroups clears all three.  An over-capacity region = its full
  // bucket + its records of the list (hi-res rigs put 1 000-2 000 points into the regions next to the ego vehicle: the
  // list holds a few thousand records); only when the LIST overflows (ovf_cap records) is such a region rebuilt from
  // the voxel ids of its sample.
  int32_t* ovf;
  int32_t* ovf_ctl;
  int ovf_cap;
};

constexpr int LSS_REGION_SIDE = 8;  // cells per region side

// launch 1 (calib_host: HOST calibration buffer or nullptr, as lss_depthnet_voxels_hostcal_fwd)
int lss_region_depthnet_voxels(const float* frustum, const float* inv_post_rots, const float* post_trans,
                               const float* combine, const float* trans, const float* calib_host, const float* dx,
                               const float* bx, const float* x, const float* w, const float* bias, int B, int N, int D,
                               int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel, float* depth,
                               float* feat, const LssRegionPlan& plan, void* stream);
// launch 1 without the depthnet: depth / context come from other kernels (vovnet heads); feat (B*N*fH*fW, C) fp32
int lss_region_voxels_absmax(const float* frustum, const float* inv_post_rots, const float* post_trans,
                             const float* combine, const float* trans, const float* dx, const float* bx,
                             const float* feat, int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z,
                             int32_t* voxel, const LssRegionPlan& plan, void* stream);
// launch 2
int lss_region_fill(const int32_t* voxel, const float* depth, int B, int N, int D, int HW, int X, int Y, int Z,
                    const LssRegionPlan& plan, int32_t* entries, void* stream);
// K2 workgroups of launch 1 (= slots of wg_absmax)
int lss_region_k2_blocks(int B, int N, int fH, int fW);
