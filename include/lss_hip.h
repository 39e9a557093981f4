/*
 * lss_hip.h - C ABI of the MI355X (gfx950) Lift-Splat-Shoot camera->BEV path.
 *
 * The reference (fircarpediem/LSS2_Multimodal_nu) is pure Python/PyTorch and
 * has no FFI layer; its boundary for this path is the nn.Module API of
 * src/model_BEV_TXT.py / src/modules.py / src/tools.py.  This library sits
 * directly below that boundary: every entry point replaces a run of ATen ops
 * of the reference (cited per function as "replaces: file:lines", paths
 * relative to the reference root) and is what a maintainer binds from Python
 * with ctypes - plain pointers and sizes, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it, nothing synchronises, nothing allocates;
 *   - return value: 0 = ok; < 0 = argument check failed (LSS_E_*);
 *     > 0 = a hipError_t from the launch;
 *   - functions are re-entrant and keep no global state.
 *
 * Index conventions
 *   point  p    = (((b*N + n)*D + d)*fH + h)*fW + w          0 <= p < P
 *   pixel  row  = (b*N + n)*fH*fW + h*fW + w
 *   voxel  v    = ((b*X + ix)*Y + iy)*Z + iz                 0 <= v < B*X*Y*Z
 *   BEV tensor  = logical (B, Z*C, X, Y), channel = iz*C + c   (ref cat(unbind(2),1))
 */
#ifndef LSS_HIP_H_
#define LSS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSS_ABI_VERSION 1

/* argument-check codes */
#define LSS_E_NULL (-1)      /* a required pointer is NULL            */
#define LSS_E_SHAPE (-2)     /* a size is <= 0 or violates a kernel limit */
#define LSS_E_LAYOUT (-3)    /* unknown layout / dtype enum           */
#define LSS_E_ALIGN (-4)     /* pointer not aligned as required       */
#define LSS_E_WORKSPACE (-5) /* workspace too small                   */
/* RCCL entry points: LSS_E_RCCL_BASE = librccl could not be resolved; LSS_E_RCCL_BASE - r = ncclResult_t r */
#define LSS_E_RCCL_BASE (-100)

/* BEV tensor memory layouts */
#define LSS_BEV_NCHW_F32 0  /* (B, Z*C, X, Y) contiguous fp32 - the reference's layout */
#define LSS_BEV_NHWC_F32 1  /* (B, X, Y, Z*C) fp32 = torch channels_last of the same logical tensor */
#define LSS_BEV_NHWC_BF16 2 /* (B, X, Y, Z*C) bf16 - feeds the MFMA BevEncode convs directly */

/* activation dtypes of the conv path */
#define LSS_DT_F32 0
#define LSS_DT_BF16 1

/* `relu` argument of the conv entry points: activation code, optionally OR-ed with
 * LSS_OUT_F32 (lss_conv2d_fwd only: y is written as fp32 although dt is bf16) */
#define LSS_ACT_NONE 0
#define LSS_ACT_RELU 1
#define LSS_ACT_GELU 2 /* 0.5 x (1 + erf(x / sqrt 2)), torch.nn.GELU() */
#define LSS_OUT_F32 16
/* lss_conv2d_fwd / lss_conv2d_head_fwd (3x3, stride 1, pad 1, bf16): `w_packed` is in the layout of
 * lss_conv2d_pack_weights_ring and the launch runs on the loader / consumer ring kernel (csrc/conv_ring.hip);
 * LSS_E_SHAPE when lss_conv2d_ring_ok() says the shape is not one of its cases. */
#define LSS_W_RING 64
/* lss_conv2d_fwd (3x3, stride 1, pad 1, bf16, no skip tensor): `w_packed` is in the layout of lss_conv2d_pack_weights_ks
 * and the launch runs on the K-split one-pass kernel of the launch-bound layers (csrc/conv_ks.hip); LSS_E_SHAPE when
 * lss_conv2d_ks_ok() says the shape is not one of its cases. */
#define LSS_W_KS 128
/* lss_conv2d_fwd, 1x1 bf16 only: y is written head-major, (B, Cout/32, Ho*Wo, 32), the
 * layout the deformable-attention gather reads with the fewest cache lines */
#define LSS_OUT_HEAD_MAJOR32 32

int lss_abi_version(void);
/* Static string for a return code of this library (never NULL). */
const char* lss_error_string(int code);

/* ---------------------------------------------------------------------------
 * K3  frustum points -> voxel ids (exact integer parity with the reference).
 * replaces: src/model_BEV_TXT.py:59-68 (get_geometry, given the two per-camera
 *           matrices) + :92 (quantise) + :99-103 (in-box filter) + :106-109.
 *   frustum        (D,fH,fW,3) fp32  - the state_dict tensor, consumed as is
 *   inv_post_rots  (B*N,3,3)   fp32  = torch.inverse(post_rots)      [host LAPACK]
 *   post_trans     (B*N,3)
 *   combine        (B*N,3,3)   fp32  = rots @ torch.inverse(intrins) [host LAPACK]
 *   trans          (B*N,3)
 *   dx, bx         (3) fp32 device   - the module's Parameters
 *   X,Y,Z                            - nx
 *   voxel          (P) int32 out: voxel id, or -1 where the reference drops the point
 *   vox_count      (B*X*Y*Z) int32 in/out, may be NULL: if given, the kernel also
 *                  histograms (atomicAdd 1 per kept point); caller provides zeros.
 *   geom           (P,3) fp32 out, may be NULL: the ego-frame points themselves
 *                  (what `get_geometry` returns, bit-identical to the reference
 *                  given the same two matrices)
 * fp32 arithmetic is issued un-contracted in the reference's exact order.
 */
int lss_points_to_voxels(const float* frustum, const float* inv_post_rots,
                         const float* post_trans, const float* combine,
                         const float* trans, const float* dx, const float* bx,
                         int B, int N, int D, int fH, int fW, int X, int Y, int Z,
                         int32_t* voxel, int32_t* vox_count, float* geom, void* stream);

/* API-compat half of K3 for callers that already hold a geometry tensor
 * (`voxel_pooling(geom_feats, x)`, ref: src/model_BEV_TXT.py:84-109):
 *   geom (P,3) fp32, P = B * pts_per_sample, sample b owns points [b*pps, (b+1)*pps)
 */
int lss_geom_to_voxels(const float* geom, const float* dx, const float* bx, int B,
                       int pts_per_sample, int X, int Y, int Z, int32_t* voxel,
                       int32_t* vox_count, void* stream);

/* ---------------------------------------------------------------------------
 * K4  sort-free bucketing of points by voxel (replaces the argsort + gathers of
 *     src/model_BEV_TXT.py:110-111).
 *   depth     (P) fp32 or NULL: per-point depth weight = K2's `depth` tensor, whose
 *             flat index is the point id (NULL: weight 1, for pre-lifted inputs)
 *   vox_count (nvox) int32 in/out: per-voxel point counts on entry (from K3),
 *             all zero again on return (ready for the next call)
 *   vox_list  (nvox) int2 out: {start, len} of each voxel's slice of `entries`
 *   D, HW     depth bins and pixels per camera image (point p = (bn*D + d)*HW + pix);
 *             D <= 128, P/D < 2^24
 *   entries   (P) int2 out: {key, depth weight bits} grouped by voxel, key =
 *             (feature row << 7) | depth bin with feature row = bn*HW + pix; the
 *             order inside one voxel's slice is unspecified (K5 orders each
 *             slice by key before summing, so BEV sums are reproducible)
 *   cursor    (1) int32 in/out scratch, zero on entry, zero again on return
 */
int lss_bucket_points(const int32_t* voxel, const float* depth, int P, int D, int HW, int nvox,
                      int32_t* vox_count, int32_t* vox_list /* nvox*2 */,
                      int32_t* entries /* P*2 */, int32_t* cursor, void* stream);

/* ---------------------------------------------------------------------------
 * K2  CamEncode: 1x1 depthnet conv + softmax over the D depth logits.
 * replaces: src/modules.py:82-83 (depthnet, softmax).  The outer product of
 *           :84 is NOT materialised; the splat kernels form it in registers.
 *   x      (BN, Cin, HW) fp32 NCHW trunk features (Cin % 16 == 0)
 *   w      (D+C, Cin) fp32, bias (D+C) fp32 - depthnet.weight / .bias as stored
 *   depth  (BN, D, HW) fp32 out - softmax probabilities (ref `depth`)
 *   feat   (BN*HW, C)  fp32 out - context features, pixel-major / channels-last
 *   math   LSS_DT_F32: f32 MFMA (exact fp32 FMA chains); LSS_DT_BF16: bf16 MFMA
 */
int lss_depthnet_softmax_fwd(const float* x, const float* w, const float* bias,
                             int BN, int Cin, int HW, int D, int C,
                             float* depth, float* feat, int math, void* stream);

/* ---------------------------------------------------------------------------
 * K2v  two-source CamEncode of the vovnet models: depth logits from the depth
 *      head's hidden map, context from `feat_proj` over the trunk's C3 map.
 * replaces: src/model_vovnet_transformer.py:82,86-87 (StandardDepthNet's last 1x1
 *           conv + softmax), :31,39 (MultiScaleDepthNet's 1x1 convs, raw logits),
 *           :97,108 (CamEncodeV2.feat_proj); the outer product of :116-120 is
 *           NOT materialised (K5 forms it in registers).
 *   x_depth (BN*HW, Cd) NHWC activations of the depth head's 3x3 conv, dtype `dt`
 *   w_depth (D, Cd) fp32, b_depth (D) fp32
 *   x_feat  (BN, Cf, HW) fp32 NCHW trunk map, w_feat (C, Cf), b_feat (C); all three
 *           may be NULL with C = 0 (logits only)
 *   softmax 1: depth = softmax over D; 0: depth = raw logits
 *   depth   (BN, D, HW) fp32 out;  feat (BN*HW, C) fp32 out (channels-last rows)
 *   Cd % 64 == 0, Cf % 64 == 0; D <= 64; C in {0, 49..64, 113..128}
 *   math    LSS_DT_F32: f32 MFMA (exact fp32 FMA chains, bf16 inputs widened exactly);
 *           LSS_DT_BF16: operands rounded to bf16, fp32 accumulation (Cd, Cf % 128 == 0)
 */
int lss_camencode_v2_fwd(const void* x_depth, int dt, const float* w_depth, const float* b_depth,
                         int Cd, const float* x_feat, const float* w_feat, const float* b_feat,
                         int Cf, int BN, int HW, int D, int C, int softmax, int math, float* depth,
                         float* feat, void* stream);

/* MultiScaleDepthNet tail.
 * replaces: src/model_vovnet_transformer.py:61-70 (F.interpolate bilinear
 *           align_corners=False, cat, fusion 1x1 conv + BatchNorm(eval) + ReLU, softmax)
 *   d3 (BN, D, H, W), d4 (BN, D, H4, W4) fp32 raw logits; w_fusion (D, 2D) fp32;
 *   scale, shift (D) fp32 = eval BatchNorm folded with the conv bias; D <= 64
 *   depth (BN, D, H, W) fp32 out
 */
int lss_depth_fuse_softmax_fwd(const float* d3, const float* d4, const float* w_fusion,
                               const float* scale, const float* shift, int BN, int D, int H, int W,
                               int H4, int W4, float* depth, void* stream);

/* K3 (voxel ids + histogram, no geom output) and K2 (f32-MFMA depthnet + softmax) in ONE launch: the
 * two are independent and neither fills the chip, so their workgroups share it.  Same arguments and
 * results as lss_points_to_voxels + lss_depthnet_softmax_fwd(math = LSS_DT_F32). */
int lss_depthnet_voxels_fwd(const float* frustum, const float* inv_post_rots, const float* post_trans,
                            const float* combine, const float* trans, const float* dx, const float* bx,
                            const float* x, const float* w, const float* bias, int B, int N, int D, int fH,
                            int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                            int32_t* vox_count, float* depth, float* feat, void* stream);

/* ---------------------------------------------------------------------------
 * K5/K6  fused lift + splat: bev[v, c] = sum_{p in voxel v} w[p] * feat[row(p), c]
 *        (w = the depth weight K4 stored next to the point id)
 * replaces: src/modules.py:84 (outer product), src/model_BEV_TXT.py:80,89
 *           (permute/reshape copy), :110-111 (gathers), src/tools.py:195-200
 *           (cumsum trick), src/model_BEV_TXT.py:120-124 (zeros, index_put, cat).
 * Every BEV element is written exactly once (zeros for empty voxels).
 *   C in {64, 128}; layout = LSS_BEV_*
 */
int lss_lift_splat_fwd(const float* feat, const int32_t* vox_list, const int32_t* entries,
                       int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z,
                       void* bev, int layout, void* stream);

/* ---------------------------------------------------------------------------
 * K7  backward of K5/K6 (+ softmax backward), point-stationary gather, no atomics.
 * replaces: src/tools.py:211-218 (QuickCumsum.backward) + the autograd nodes of
 *           src/model_BEV_TXT.py:80-121 and src/modules.py:83-84.
 *   grad_bev   BEV gradient in `layout` (NCHW_F32, NHWC_F32, or NHWC_BF16: the gradient a bf16 stem hands back)
 *   voxel      (P) int32 from K3
 *   g_logits   (BN, D+C, HW) fp32 out: gradient w.r.t. the depthnet output
 *              (softmax backward applied to the first D channels)
 */
int lss_lift_splat_bwd(const void* grad_bev, int layout, const int32_t* voxel,
                       const float* depth, const float* feat, int B, int N, int D,
                       int fH, int fW, int C, int X, int Y, int Z, float* g_logits,
                       void* stream);

/* ---------------------------------------------------------------------------
 * API-compat: segmented sum over rows pre-sorted by rank
 * replaces: src/tools.py:181-189 / :194-208 (cumsum_trick / QuickCumsum.forward)
 *   x (K,C) fp32; seg_start (M+1) int32 row offsets of the M equal-rank runs
 *   y (M,C) fp32 out
 */
int lss_segmented_sum(const float* x, const int32_t* seg_start, int M, int C, float* y,
                      void* stream);

/* ---------------------------------------------------------------------------
 * BEV transformer, token-major (B, H*W, 256) rows = NHWC activations.
 * The linear layers are 1x1 convs (lss_conv2d_fwd); these three kernels are the
 * rest of the encoder layer.  C must be 256, n_heads = n_points = 8 (the
 * reference's only configuration).
 */
/* q = x + pos.  replaces: src/transformer_modules.py:199-200 (pos_flat, q = src + pos)
 *   x, q (B, T, C) in `dt`; pos (T, C) fp32 = PositionEmbeddingSine table (:25-59) */
int lss_add_pos_fwd(const void* x, const float* pos, int B, int T, int C, int dt, void* q,
                    void* stream);

/* Deformable attention core.  replaces: src/transformer_modules.py:117-156 (views,
 *  softmax over points, sampling locations, the per-head grid_sample loop, weighting).
 *   value          value_proj(src) in `dt`, (B, H, W, C) or head-major (see value_layout)
 *   offsets_logits (B*H*W, 192) fp32: [0,128) = sampling_offsets(q) as (head, point, xy),
 *                  [128,192) = attention_weights(q) as (head, point)
 *   token_bias     (H*W, 192) fp32 or NULL: added to every sample's offsets_logits row of the
 *                  same token.  With token_bias = pos @ [W_off; W_attn]^T the two linears can
 *                  run on src instead of q = src + pos (linearity), so q is never formed
 *   ref_x (W), ref_y (H) fp32 = torch.linspace(0, 1, n)  (:243-244)
 *   out            (B, H*W, C) in `dt`, ready for output_proj
 * Sampling is bilinear with zero padding at align_corners=False pixel coordinates, both
 * offset axes divided by H as the reference does (:124). */
#define LSS_VALUE_NHWC 0       /* value (B, H*W, heads*32) */
#define LSS_VALUE_HEAD_MAJOR 1 /* value (B, heads, H*W, 32): what LSS_OUT_HEAD_MAJOR32 writes */
int lss_deform_attn_fwd(const void* value, int value_layout, const float* offsets_logits,
                        const float* token_bias, const float* ref_x, const float* ref_y, int B, int H,
                        int W, int n_heads, int n_points, int C, int dt, void* out, void* stream);

/* nn.LayerNorm(C) over rows.  replaces: src/transformer_modules.py:204,208 (norm1, norm2)
 *   x (rows, C) in x_dt; y (rows, C) in y_dt; gamma, beta (C) fp32 */
int lss_layernorm_fwd(const void* x, int x_dt, const float* gamma, const float* beta,
                      long long rows, int C, float eps, void* y, int y_dt, void* stream);

/* ---------------------------------------------------------------------------
 * K8  BevEncode convolutions: implicit-GEMM on MFMA, NHWC activations.
 * replaces: the conv2d / batch_norm / relu / add / interpolate / cat ATen ops of
 *           src/modules.py:22-27, 118-130 (and torchvision BasicBlock.forward).
 *
 * One launch computes
 *     y = act( scale[co] * conv(in, w)[.., co] + shift[co] + residual )
 * where `in` is either x itself, or (fused gather) the channel concat
 *     [ x2 , bilinear_align_corners_upsample(x, up) ]          (ref Up.forward)
 *   x        (B, H, W, Cx)  NHWC, dtype `dt`
 *   x2       (B, H*up, W*up, C2) NHWC or NULL (C2 = 0)
 *   w        packed weights from lss_conv2d_pack_weights
 *   scale, shift (Cout) fp32 or NULL (=1 / =0): folded eval-mode BatchNorm or bias
 *   residual (B, Ho, Wo, Cout) dtype `dt` or NULL
 *   y        (B, Ho, Wo, Cout) dtype `dt`;  Ho = (H*up + 2*pad - KH)/stride + 1
 *   stats    (2*Cout) fp32 or NULL: if given, per-channel sum and sum of squares
 *            of the raw conv output are atomically accumulated (training-mode BN)
 */
size_t lss_conv2d_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int dt);
int lss_conv2d_pack_weights(const float* w_oihw, int Cout, int Cin, int KH, int KW,
                            int dt, void* w_packed, void* stream);
int lss_conv2d_fwd(const void* x, const void* x2, const void* w_packed,
                   const float* scale, const float* shift, const void* residual,
                   void* y, float* stats, int B, int H, int W, int Cx, int C2, int up,
                   int Cout, int KH, int KW, int stride, int pad, int relu, int dt,
                   void* stream);

/* K8r  the ring kernel's weight layout for the big 3x3 / stride-1 layers (replaces the same lines as K8:
 * src/modules.py:22-27 `Up.conv`, :110-116 `up2`): per (128-channel block, 32-channel chunk, tap) one 8-KiB slab in
 * the consumer waves' MFMA fragment order, so that a slab is 8 coalesced 1-KiB LDS-DMA pieces.
 * lss_conv2d_ring_ok: 1 when (B, H, W, Cx, C2, up, Cout, head_n) is a case for that kernel (head_n = 0: no head). */
int lss_conv2d_ring_ok(int B, int H, int W, int Cx, int C2, int up, int Cout, int head_n);
size_t lss_conv2d_ring_packed_weight_bytes(int Cout, int Cin);
int lss_conv2d_pack_weights_ring(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream);
/* The same layout for the conv that maps dY (Cout channels) to dX (Cin channels) of a 3x3 / s1 / p1 conv with weight
 * w_oihw [Cout][Cin][3][3] (the autograd node ConvolutionBackward of ref src/modules.py:22-27 run by train.py:61):
 * lss_conv2d_fwd(dy, ..., Cx = Cout, Cout = Cin, relu = LSS_W_RING) then computes the input gradient. */
int lss_conv2d_pack_weights_ring_dgrad(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream);
/* flag waits of the ring kernel that hit their bound since load (must be 0; synchronises the device) */
int lss_conv2d_ring_timeouts(void);

/* K8k  the launch-bound 3x3 / stride-1 convs of resnet18's layer1-3 inside BevEncode (replaces the conv + BN + ReLU
 * (+ identity) of torchvision's BasicBlock as called from src/modules.py:104-106, 123-125) as ONE pass per workgroup:
 * the whole input patch of a pixel block in LDS, the K dimension split over the waves, the weights in registers.
 * lss_conv2d_ks_ok: 1 when (B, H, W, Cin, Cout) is a case for it (Cin in {64, 128, 256}, Cout % 32 == 0, 64-512
 * workgroups).  Weight layout: [32-channel output block][k-step = 32 input channels of one tap][2 channel tiles][lane][8]
 * bf16 - each wave's A fragments as they lie in its registers; use with lss_conv2d_fwd(..., relu | LSS_W_KS). */
int lss_conv2d_ks_ok(int B, int H, int W, int Cin, int Cout);
size_t lss_conv2d_ks_packed_weight_bytes(int Cout, int Cin);
int lss_conv2d_pack_weights_ks(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream);
/* the same image for the INPUT-GRADIENT conv of that layer (training): w_oihw is the forward layer's (Cout, Cin, 3, 3);
 * the gradient conv maps Cout -> Cin channels with transposed, tap-flipped weights (needs Cout in {64, 128, 256},
 * Cin % 32 == 0) */
int lss_conv2d_pack_weights_ks_dgrad(const float* w_oihw, int Cout, int Cin, void* w_packed, void* stream);

/* ---------------------------------------------------------------------------
 * K8b  gradients of the convolutions (training; replaces the ConvolutionBackward autograd
 *      nodes behind `loss.backward()`, train.py:61, for the convs of src/modules.py:22-27,
 *      118-130 and torchvision's BasicBlock).
 *
 * Input gradient: dX = conv(dY, W') with W'[ci][tap'][co] = W[co][KH-1-ky][KW-1-kx][ci].
 *   lss_conv2d_pack_weights_dgrad arranges W' as [tap'][Cin][Cout]; then call lss_conv2d_fwd
 *   with x = dY (Cx = Cout), Cout = Cin, the same KH, KW, stride 1 and pad' = KH-1-pad.
 * Weight gradient (3x3 / stride 1 / pad 1, bf16 NHWC operands, fp32 OIHW result):
 *   x  (B,H,W,Cin) bf16 - the conv's input;  dy (B,H,W,Cout) bf16 - gradient of its raw output
 *   workspace: lss_conv2d_wgrad_workspace_bytes bytes, 256-B aligned (channel-major copies of
 *   both operands + fp32 split-K partials); dw_oihw (Cout,Cin,3,3) fp32, fully overwritten.
 *   Cin % 8 == 0, Cout % 8 == 0.  Fixed summation order: bit-reproducible.
 *   Shapes with Cin % 64 == 0, Cout % 64 == 0 and 8 <= W <= 224 run on K9w (csrc/conv_wgrad.hip): one kernel that
 *   reads both NHWC tensors as they lie and transposes the pixel dimension in its LDS reads (no channel-major
 *   copies; the workspace then holds the fp32 partial tiles only).  lss_conv2d_wgrad_timeouts: flag waits of that
 *   kernel that hit their bound since load (must be 0; synchronises the device).
 */
int lss_conv2d_pack_weights_dgrad(const float* w_oihw, int Cout, int Cin, int KH, int KW, int dt,
                                  void* w_packed, void* stream);
size_t lss_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int lss_conv2d_wgrad(const void* x, const void* dy, int B, int H, int W, int Cin, int Cout,
                     void* workspace, size_t workspace_bytes, float* dw_oihw, void* stream);
int lss_conv2d_wgrad_timeouts(void);
/* Weight gradient of the 4 x 4-tap stride-1 conv with taps (dy, dx) in {-2 .. 1}^2 on K9w (two launches of eight
 * consumer waves): what the 7x7 / stride-2 / pad-3 stem conv (ref src/modules.py:99, its ConvolutionBackward under
 * train.py:61) is over the phase planes xs[b, y, x, (py, px, c)] = x[b, 2y + py, 2x + px, c].
 *   xs (B,H,W,Cin) bf16 NHWC (Cin = 4 x the conv's input channels), dy (B,H,W,Cout) bf16 NHWC;
 *   dw16 (Cout, Cin, 16) fp32, tap index (dy + 2) * 4 + dx + 2, fully overwritten; fixed summation order.
 *   Same shape rules as K9w (Cin % 64 == 0, Cout % 64 == 0, 8 <= W <= 224); LSS_E_SHAPE otherwise. */
/* Data gradient of the stride-2 convs (7x7 / pad 3 stem, 3x3 / pad 1, 1x1 / pad 0 shortcut; ref src/modules.py:99 and
 * torchvision BasicBlock, their ConvolutionBackward under train.py:61) as ONE stride-1 conv over dY that yields the
 * input's four phase planes: lss_conv2d_s2_dgrad_taps(K, pad) = taps per dimension KT (4 / 2 / 1; 0: not a case);
 * lss_conv2d_pack_weights_s2_dgrad arranges w [Cout][Cin][K][K] as bf16 [KT*KT][4*Cin][Cout]; then
 * lss_conv2d_fwd(dy, ..., Cx = Cout, Cout = 4*Cin, KH = KW = KT, stride 1, pad KT/2) and rows / columns [s, s + H/2)
 * of its output (s = 1 for KT = 2, 4; 0 for KT = 1) are dXs[b, j, i, (py, px, ci)] = dX[b, 2j + py, 2i + px, ci]. */
int lss_conv2d_s2_dgrad_taps(int K, int pad);
int lss_conv2d_pack_weights_s2_dgrad(const float* w_oihw, int Cout, int Cin, int K, int pad, void* w_packed, void* stream);
size_t lss_conv2d_wgrad4x4_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int lss_conv2d_wgrad4x4(const void* xs, const void* dy, int B, int H, int W, int Cin, int Cout,
                        void* workspace, size_t workspace_bytes, float* dw16, void* stream);

/* Backward helpers of the fused upsample + concat conv input (ref Up.forward, src/modules.py:22-24;
 * bf16 NHWC).  lss_upsample_cat_nhwc materialises [x2 | bilinear_align_corners(x, up)] as
 * (B, H*up, W*up, C2+Cx) - only the weight-gradient GEMM needs it as a tensor.
 * lss_upsample_bwd_nhwc is the adjoint of the upsample: g = channels [c_off, c_off+Cx) of a
 * (B, H*up, W*up, Ct) gradient -> dx (B, H, W, Cx); replaces upsample_bilinear2d_backward. */
int lss_upsample_cat_nhwc(const void* x, const void* x2, int B, int H, int W, int Cx, int C2, int up,
                          void* out, void* stream);
int lss_upsample_bwd_nhwc(const void* g, int B, int H, int W, int Cx, int Ct, int c_off, int up,
                          void* dx, void* stream);

/* ---------------------------------------------------------------------------
 * Training-mode BatchNorm2d (+ residual + ReLU) over NHWC bf16 rows, forward and backward.
 * replaces: batch_norm / add / relu (and their autograd nodes) of src/modules.py:16-21,
 *           100-101, 112-113 and torchvision BasicBlock.forward under model.train().
 *   z (M, C) bf16 raw conv output, M = B*H*W; residual (M, C) bf16 or NULL
 *   y = act(gamma * (z - mean) * invstd + beta (+ residual)), bf16
 *   running_mean/var (C) fp32 updated in place with `momentum` (both NULL: not tracked);
 *   save_mean, save_invstd (C) fp32 out - the batch statistics the backward needs
 *   workspace: lss_bn_train_workspace_bytes(M, C) bytes (partial sums; reusable between calls)
 * Backward: dz (M, C) bf16 = gradient w.r.t. z; dres (M, C) bf16 or NULL = gradient w.r.t.
 *   residual (= dy masked by the ReLU); dgamma, dbeta (C) fp32, overwritten.
 * C % 8 == 0, 256 % (C/8) == 0.  Fixed summation order: bit-reproducible.
 */
size_t lss_bn_train_workspace_bytes(long long M, int C);
int lss_bn_train_fwd(const void* z, const void* residual, long long M, int C, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int relu, void* workspace, void* y, float* save_mean,
                     float* save_invstd, void* stream);
int lss_bn_train_bwd(const void* dy, const void* y, const void* z, long long M, int C,
                     const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                     void* workspace, void* dz, void* dres, float* dgamma, float* dbeta, void* stream);

/* Split forms for SYNCHRONISED BatchNorm under data parallelism: the per-channel sums
 * come back as a (2, C) fp32 vector, the caller all-reduces it over the ranks, and the second half
 * normalises this rank's rows with the GLOBAL statistics (M_total = rows of all ranks).
 *   lss_bn_partial_sums  mode 0: (sum (z-p), sum (z-p)^2) about the pivot vector p = `mean` (shared by all ranks: the
 *                        running mean; NULL = 0; pass the same vector as running_mean to _fwd_from_sums - it keeps a
 *                        large-mean channel's variance from cancelling);  mode 1: (sum g, sum g*xhat), g = dy masked by ReLU
 *   lss_bn_train_fwd_from_sums / _bwd_from_sums: as lss_bn_train_fwd / _bwd with the statistics given.
 *   (_bwd_from_sums writes the GLOBAL sums to dgamma / dbeta; a DP caller keeps its local sums instead.) */
int lss_bn_partial_sums(const void* z, const void* dy, const void* y, const float* mean, const float* invstd,
                        long long M, int C, int relu, int mode, void* workspace, float* sums, void* stream);
int lss_bn_train_fwd_from_sums(const void* z, const void* residual, long long M, int C, const float* sums,
                               long long M_total, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, int relu, void* workspace,
                               void* y, float* save_mean, float* save_invstd, void* stream);
int lss_bn_train_bwd_from_sums(const void* dy, const void* y, const void* z, long long M, int C,
                               const float* sums, long long M_total, const float* gamma,
                               const float* save_mean, const float* save_invstd, int relu, void* workspace,
                               void* dz, void* dres, float* dgamma, float* dbeta, void* stream);

/* One-call training units (the training step is framework-bound: chaining the launches here costs
 * two host calls per conv+BN unit instead of sixteen).
 * forward:  z = conv3x3(x) with x = x1 (up = 1, C2 = 0) or cat([x2, bilinear_align_corners(x1, up)]),
 *           y = act(BN_train(z) (+ residual));  w_packed: scratch of 9*Cout*(Cx+C2) bf16 - or, with w_oihw = NULL,
 *           the image itself, packed ahead of the step (lss_conv_bn_act_train_pack / lss_gather_pack); the same
 *           convention for w_dgrad in the backward call.
 * backward: BN backward -> dz (and dres); if gcat != NULL: gcat = dgrad conv (B,H*up,W*up,Cx+C2) bf16
 *           (w_dgrad: scratch like w_packed), and if g1 != NULL: g1 = upsample adjoint of gcat's
 *           channels [C2, C2+Cx) -> (B,H,W,Cx); if dw != NULL: dw (Cout,Cx+C2,3,3) fp32 via
 *           lss_conv2d_wgrad (xcat: scratch (B,H*up,W*up,Cx+C2) bf16 when up > 1 or C2 > 0).
 * All tensors bf16 NHWC unless noted; shapes/limits as the individual entry points. */
int lss_conv_bn_act_train_fwd(const void* x1, const void* x2, const float* w_oihw, const float* gamma,
                              const float* beta, const void* residual, float* running_mean,
                              float* running_var, void* w_packed, void* z, void* y, float* save_mean,
                              float* save_invstd, void* bn_workspace, int B, int H, int W, int Cx, int C2,
                              int up, int Cout, float momentum, float eps, int relu, void* stream);
/* The weight image the unit of this shape runs on (forward: dgrad = 0, input-gradient conv: dgrad = 1): the same
 * ring / K-split / tile decision the two calls below make.  With it a host packs a layer's images itself and passes
 * w_oihw = NULL to the units (w_packed / w_dgrad then hold the images) - e.g. all layers of a model in one launch per
 * step through lss_gather_pack. */
int lss_conv_bn_act_train_pack(const float* w_oihw, int B, int H, int W, int Cx, int C2, int up, int Cout, int dgrad,
                               void* w_packed, void* stream);
int lss_conv_bn_act_train_bwd(const void* dy, const void* y, const void* z, const void* x1, const void* x2,
                              const float* w_oihw, const float* gamma, const float* save_mean,
                              const float* save_invstd, void* bn_workspace, void* wgrad_workspace,
                              size_t wgrad_workspace_bytes, void* w_dgrad, void* dz, void* dres,
                              float* dgamma, float* dbeta, void* gcat, void* g1, void* xcat, float* dw,
                              int B, int H, int W, int Cx, int C2, int up, int Cout, int relu,
                              void* stream);

/* ---------------------------------------------------------------------------
 * Gather-pack (training): every packed weight image of a step in ONE launch.  A packed image is a permutation of the
 * layer's fp32 weights (+ zeros): out[e] = idx[e] ? bf16(src[idx[e] - 1]) : 0.  `jobs`: HOST array of `count` records
 *     { const float* src; const int32_t* idx; uint16_t* dst; long long n; }          (device pointers)
 * passed on by value in the kernel arguments.  The host derives idx once per layer by pushing index patterns through
 * that layer's own lss_*_pack_weights* routine (lss2_multimodal_nu_amd/ops.py: WeightPrepack); replaces the 36 pack
 * launches a training step otherwise needs (the weights change every step; no reference counterpart). */
int lss_gather_pack(const void* jobs, int count, void* stream);

/* ---------------------------------------------------------------------------
 * K10  gradient-norm clip + Adam over a list of fp32 tensors (training; replaces
 *      torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm) + torch.optim.Adam.step(), train.py:41, 62-63)
 *      in three launches: chunk sums of squares, one-workgroup finalize (norm, clip coefficient, step counter, bias
 *      corrections), clipped Adam update.  `tensors`: HOST array of `count` records
 *          { float* p; float* g; float* m; float* v; long long n; }      (device pointers, contiguous fp32)
 *      passed on by value in the kernel arguments.  state: 8 device floats, zero before the first step: [0] step
 *      count, [1] total gradient norm before the clip, [2] clip coefficient.  partials: lss_clip_adam_partials()
 *      device floats.  max_norm <= 0: no clip.  g is overwritten with the clipped gradient. */
long long lss_clip_adam_partials(const long long* numel, int count);
int lss_clip_adam_step(const void* tensors, int count, float* state, float* partials, long long n_partials, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float max_norm, void* stream);

/* ---------------------------------------------------------------------------
 * Weighted cross-entropy over NCHW logits.
 * replaces: nn.CrossEntropyLoss(weight)(ypred, ytgt) of SimpleLoss / MultiLoss, src/tools.py:221-238
 *           (log_softmax + nll_loss2d, forward and backward).
 *   logits (B, C, H*W) fp32, C <= 16; target (B, H*W) int64 (entries outside [0, C) are ignored);
 *   weight (C) fp32;  loss = sum w[t] * nll / sum w[t].
 *   workspace: 512 floats.  sums (2) fp32 out: {sum w*nll, sum w} - kept for the backward.
 *   backward: grad_logits (B, C, H*W) fp32 = grad_loss[0] * w[t] * (softmax - onehot) / sums[1].
 */
int lss_weighted_ce_fwd(const float* logits, const long long* target, const float* weight, int B, int C,
                        long long HW, float* workspace, float* sums, float* loss, void* stream);
int lss_weighted_ce_bwd(const float* logits, const long long* target, const float* weight, int B, int C,
                        long long HW, const float* sums, const float* grad_loss, float* grad_logits,
                        void* stream);

/* Stride-2 convs (3x3 pad 1, 7x7 pad 3, and 1x1 pad 0 with the plain weight pack;
 * bf16) on the LDS-tiled MFMA kernel: the
 * conv is evaluated as a stride-1 conv over the 4 parity phases of the input
 * (space-to-depth folded into the operand gather).  Weights are arranged
 * [tap'][Cout][phase*Cin + ci] by lss_conv2d_pack_weights_s2d.  Same epilogue as
 * lss_conv2d_fwd.  replaces: conv1 (src/modules.py:99) and the stride-2 convs of
 * torchvision's layer2.0 / layer3.0. */
size_t lss_conv2d_s2d_packed_weight_bytes(int Cout, int Cin, int K, int pad);
int lss_conv2d_pack_weights_s2d(const float* w_oihw, int Cout, int Cin, int K, int pad,
                                void* w_packed, void* stream);
int lss_conv2d_s2_fwd(const void* x, const void* w_s2d, const float* scale, const float* shift,
                      const void* residual, void* y, float* stats, int B, int H, int W, int Cx,
                      int Cout, int K, int pad, int relu, void* stream);

/* Two stride-2 convs over the same input in one launch (K = 3, pad 1): w_s2d / scale / shift hold
 * Cout = split + rest output channels; channels [0, split) -> y (B,Ho,Wo,split) with the activation,
 * [split, Cout) -> y2 (B,Ho,Wo,Cout-split) without.  split % 128 == 0.
 * replaces: torchvision BasicBlock.forward's `relu(bn1(conv1(x)))` and `downsample(x)` of layer2.0 /
 * layer3.0 (the 1x1/2 downsample weight sits at the centre tap of a 3x3 frame in w_s2d). */
int lss_conv2d_s2_dual_fwd(const void* x, const void* w_s2d, const float* scale, const float* shift,
                           void* y, void* y2, int B, int H, int W, int Cx, int Cout, int split, int K,
                           int pad, int relu, void* stream);

/* 3x3/s1/p1 conv (+ fused upsample/concat gather) + scale/shift + ReLU + fused 1x1
 * head, bf16 in, NCHW fp32 out (B, head_n, H*up, W*up).  Cout = 128, or 64 for the plain 3x3 (up = 1, C2 = 0).
 * replaces: src/modules.py:110-116 (up2: upsample, conv3x3, BN, ReLU, conv1x1+bias) and
 *   src/model_vovnet_transformer.py:141-143 (seg_head: conv3x3 128->64, BN, ReLU, conv1x1)
 *   head_w (head_n, Cout) fp32, head_b (head_n) fp32 */
int lss_conv2d_head_fwd(const void* x, const void* x2, const void* w_packed, const float* scale,
                        const float* shift, const float* head_w, const float* head_b, float* out,
                        int B, int H, int W, int Cx, int C2, int up, int Cout, int head_n,
                        int relu, void* stream);

/* Lift-splat of depth / context tensors that other kernels produced - the vovnet model's depth heads and
 * CamEncodeV2 (replaces src/model_vovnet_transformer.py:513-554 `get_voxels`' geometry + voxel_pooling for those
 * tensors): K3 geometry, bucketing and splat behind one call, on the region-bucketed pipeline when the problem fits
 * it (lss_region_pipeline_ok; C = 64 or 128), else K3 -> K4 -> K5.
 *   depth (B*N, D, fH, fW) fp32 softmax weights; feat (B*N*fH*fW, C) fp32 channels-last context
 *   workspace words as lss_lift_splat_forward; bev / layout as lss_lift_splat_fwd */
int lss_lift_splat_from_heads(const float* frustum, const float* inv_post_rots, const float* post_trans,
                              const float* combine, const float* trans, const float* dx, const float* bx,
                              const float* depth, const float* feat, int B, int N, int D, int fH, int fW, int C,
                              int X, int Y, int Z, int32_t* voxel, int32_t* vox_count, int32_t* vox_list,
                              int32_t* entries, int32_t* cursor, void* bev, int layout, void* stream);

/* Descriptor form of lss_lift_splat_forward / _hostcal / _from_heads (same replaced reference lines:
 * src/model_BEV_TXT.py:50-126 get_geometry + get_cam_feats + voxel_pooling, src/modules.py:82-84, src/tools.py:181-218),
 * with one more workspace: `direct_entries` (lss_lift_splat_direct_bytes(...) bytes, 8-byte aligned, contents
 * irrelevant).  With it the region pipeline runs as TWO launches - the geometry workgroups write their points straight
 * into fixed-capacity per-region buckets, the splat gathers the depth weights itself - instead of three (no fill
 * launch).  A region that overflows its bucket (1024 points; hi-res rigs do, next to the ego vehicle) sends the rest
 * to one overflow list in the same workspace (65 536 records); if that overflows too (degenerate calibrations only) the
 * region is rebuilt from the voxel ids: results are the exact, order-independent fixed-point sums every way.  NULL / too
 * small: the three-launch form.
 *   calib_host != NULL : host calibration, B*N <= 36 (inv_post_rots .. trans ignored), f32 depthnet math only
 *   x == NULL          : depth (B*N, D, fH, fW) and feat (B*N*fH*fW, C) are INPUTS (the vovnet heads' form) */
typedef struct lss_lift_splat_desc {
  const float *frustum, *inv_post_rots, *post_trans, *combine, *trans, *calib_host, *dx, *bx;
  const float *x, *w, *bias;
  int32_t *voxel, *vox_count, *vox_list, *entries, *cursor;
  void* direct_entries;
  unsigned long long direct_bytes;
  float *depth, *feat;
  void* bev;
  int32_t B, N, D, fH, fW, Cin, C, X, Y, Z, layout, math;
} lss_lift_splat_desc_t;
size_t lss_lift_splat_direct_bytes(int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z);
int lss_lift_splat_forward_desc(const lss_lift_splat_desc_t* desc, void* stream);

/* 1 when lss_lift_splat_forward (f32 depthnet math) runs (B,N,D,fH,fW,C | X,Y,Z) on the region-bucketed pipeline
 * (K2 || K3 with LDS region histograms -> region fill -> fixed-point region splat), 0 when the problem exceeds its
 * limits and the voxel-list pipeline (K3, K4, K2, K5) is used.  Honours LSS_SPLAT_LEGACY. */
int lss_region_pipeline_ok(int B, int N, int D, int fH, int fW, int C, int X, int Y, int Z);

/* ---------------------------------------------------------------------------
 * Host-overhead reducers: the same kernels, several launches per call.  A caller
 * that binds these issues ONE FFI call for the whole lift-splat level and ONE for
 * the whole BevEncode instead of ~23 (each Python->C hop costs ~8-10 us of host
 * time; on a slow host the step would otherwise become launch-bound).
 */
typedef struct lss_conv_launch {
  const void* x; const void* x2; const void* w; const float* scale; const float* shift;
  const void* residual; void* y; float* stats;
  const float* head_w; const float* head_b; float* head_out;   /* kind 2 only */
  void* y2;                                                     /* kind 3 only */
  int32_t B, H, W, Cx, C2, up, Cout, KH, KW, stride, pad, relu, dt, head_n;
  int32_t kind;   /* 0 = lss_conv2d_fwd, 1 = lss_conv2d_s2_fwd, 2 = lss_conv2d_head_fwd, 3 = lss_conv2d_s2_dual_fwd */
  int32_t split;  /* kind 3 only */
} lss_conv_launch_t;
/* Enqueue `n` conv launches in order on `stream`; returns the first non-zero code. */
int lss_conv2d_sequence(const lss_conv_launch_t* launches, int n, void* stream);

/* The whole lift-splat level in one call (same arguments as the individual entries).
 * With math = LSS_DT_F32 (and 64*Z <= 256, B*X*Y*Z large enough to hold the region words below) it runs the
 * REGION-BUCKETED pipeline, three launches:
 *   1. K2 || K3: depthnet + softmax (for D <= 64, C <= 64 two workgroups per 16-pixel tile: the depth rows + softmax
 *      and the context rows; the K sum is associated in 32-deep blocks, so depth / feat may differ from
 *      lss_depthnet_softmax_fwd's in the last ulp)  ||  points -> voxel ids, each workgroup counting its 256 points per
 *      8 x 8-cell region in LDS and issuing one global atomic per non-empty region (no per-point atomics);
 *   2. fill: per-workgroup LDS ranks + one global atomic per (workgroup, region) -> entries grouped by region;
 *   3. region splat: one workgroup per region, int64 fixed-point sums in an LDS tile (associative, so the
 *      result is bit-reproducible whatever order the atomics produced), coalesced BEV stores incl. zeros.
 * It lays its words out inside the same workspace: vox_count = [region_count | region_cursor] (zero on entry, zero
 * on return, like the voxel histogram), vox_list = [region_start | per-workgroup max|feature|], entries as below,
 * voxel as below (exact ids; the backward needs them).  Otherwise (bf16 depthnet math, Z > 4, LSS_SPLAT_LEGACY=1)
 * the voxel-list pipeline K3 -> K2 -> K4 -> K5 of the individual entries runs. */
int lss_lift_splat_forward(const float* frustum, const float* inv_post_rots, const float* post_trans,
                           const float* combine, const float* trans, const float* dx, const float* bx,
                           const float* x, const float* w, const float* bias, int B, int N, int D,
                           int fH, int fW, int Cin, int C, int X, int Y, int Z, int32_t* voxel,
                           int32_t* vox_count, int32_t* vox_list, int32_t* entries, int32_t* cursor,
                           float* depth, float* feat, void* bev, int layout, int math, void* stream);

/* Position-wise feed-forward block of the BEV transformer layer in one launch (the hidden activation never
 * leaves the CU): y = x + b2 + W2 . gelu(W1 . x + b1), erf GELU.
 * ref: src/transformer_modules.py:170-172 (linear1 / activation / linear2) + the residual add of :208.
 * x (M, d_model) bf16, w1 (d_ff, d_model) bf16, b1 (d_ff) fp32, w2 (d_model, d_ff) bf16, b2 (d_model) fp32,
 * y (M, d_model) fp32 (the pre-LayerNorm sum).  d_model = 256, d_ff a multiple of 64, <= 1024.
 * With ln_gamma / ln_beta (d_model, fp32) the layer's second LayerNorm (:208 `norm2`) runs in the epilogue:
 * y_ln (M, d_model) bf16 = LN(y) * gamma + beta is written instead and y may be NULL. */
int lss_ffn_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                      long long M, int d_model, int d_ff, float* y, const float* ln_gamma, const float* ln_beta,
                      float ln_eps, void* y_ln, void* stream);

/* A d_model -> d_model linear layer whose epilogue sees whole token rows: y = x . W^T + bias + residual (fp32), or -
 * with ln_gamma / ln_beta - y_ln = LayerNorm(y) * gamma + beta (bf16) with y never written.
 * ref: src/transformer_modules.py:155-156 (DeformableAttention.output_proj) + :204 (`src + dropout1(.)`, norm1).
 * x, residual (M, 256) bf16; w (256, 256) bf16 row-major [out][in]; bias (256) fp32. */
int lss_linear_res_ln_fwd(const void* x, const void* w, const float* bias, const void* residual, long long M,
                          int d_model, float* y, const float* ln_gamma, const float* ln_beta, float ln_eps,
                          void* y_ln, void* stream);

/* Host-calibration forms: the four per-camera arrays arrive as ONE HOST buffer of B*N*24 floats,
 * [inv_post_rots (B*N*9) | combine (B*N*9) | post_trans (B*N*3) | trans (B*N*3)], are read during the call
 * and travel inside the kernel arguments (B*N <= 36): no H2D copy, no staging buffer, one launch boundary
 * less.  Otherwise identical to lss_depthnet_voxels_fwd / lss_lift_splat_forward (f32 depthnet math). */
int lss_depthnet_voxels_hostcal_fwd(const float* frustum, const float* calib_host, const float* dx,
                                    const float* bx, const float* x, const float* w, const float* bias,
                                    int B, int N, int D, int fH, int fW, int Cin, int C, int X, int Y,
                                    int Z, int32_t* voxel, int32_t* vox_count, float* depth, float* feat,
                                    void* stream);
int lss_lift_splat_forward_hostcal(const float* frustum, const float* calib_host, const float* dx,
                                   const float* bx, const float* x, const float* w, const float* bias,
                                   int B, int N, int D, int fH, int fW, int Cin, int C, int X, int Y,
                                   int Z, int32_t* voxel, int32_t* vox_count, int32_t* vox_list,
                                   int32_t* entries, int32_t* cursor, float* depth, float* feat,
                                   void* bev, int layout, void* stream);

/* ---------------------------------------------------------------------------
 * Fused 1x1 head + log-softmax + weighted NLL, forward and backward.
 * replaces: `up2[4]` = nn.Conv2d(128, outC, 1) (src/modules.py:115) followed by nn.CrossEntropyLoss(weight) of
 *           SimpleLoss / MultiLoss (src/tools.py:221-238), and their autograd: the (B, outC, H, W) logits are
 *           neither written nor read - the loss comes straight from the last activation.
 *   y (M, Cin) bf16 NHWC rows of the last conv + BatchNorm + ReLU unit, M = B*H*W, Cin = 128;
 *   head_w (K, Cin) fp32, head_b (K) fp32, K = 4 or 8; target (M) int64 (entries outside [0, K) are ignored);
 *   class_w (K) fp32.  loss = sum_p class_w[t_p] * (logsumexp(logit_p) - logit_p[t_p]) / sum_p class_w[t_p].
 *   workspace: lss_head_ce_workspace_bytes(K) bytes.  sums (2) fp32 out: {sum w*nll, sum w}, kept for the backward.
 *   backward: dy (M, Cin) bf16 = head_w^T . g with g = grad_loss[0] * class_w[t] * (softmax - onehot) / sums[1];
 *             d_head_w (K, Cin) fp32 = g^T . y;  d_head_b (K) fp32 = sum_p g.  Fixed summation order throughout. */
size_t lss_head_ce_workspace_bytes(int K);
int lss_head_ce_fwd(const void* y, const float* head_w, const float* head_b, const long long* target,
                    const float* class_w, long long M, int Cin, int K, float* workspace, float* sums, float* loss,
                    void* stream);
int lss_head_ce_bwd(const void* y, const float* head_w, const float* head_b, const long long* target,
                    const float* class_w, long long M, int Cin, int K, const float* sums, const float* grad_loss,
                    float* workspace, void* dy, float* d_head_w, float* d_head_b, void* stream);

/* The 1x1 head ALONE, forward and backward, for a training-mode `model(x)` whose loss the caller computes
 * (ref src/modules.py:115 `up2[4] = nn.Conv2d(128, outC, kernel_size=1)` and its autograd behind train.py:61 when the
 * loss is `MultiLoss` over returned logits, src/tools.py:232-251).  y: (M, 128) bf16 NHWC rows, M = B * HW;
 * logits / grad_logits: (B, K, H, W) fp32 NCHW; K in {4, 8}.  The backward leaves dy (M, 128) bf16 and the
 * fixed-order sums d_head_w (K, 128), d_head_b (K); workspace: lss_head_ce_workspace_bytes(K).  Exists because the
 * library convolution's backward is not replay-safe inside a HIP graph. */
int lss_head1x1_fwd(const void* y, const float* head_w, const float* head_b, long long M, long long HW, int Cin, int K,
                    float* logits, void* stream);
int lss_head1x1_bwd(const void* y, const float* head_w, const float* head_b, const float* grad_logits, long long M,
                    long long HW, int Cin, int K, float* workspace, void* dy, float* d_head_w, float* d_head_b,
                    void* stream);

/* ---- data-parallel gradient step (the reference has no collective on this path: its loop is
 * `loss.backward(); clip_grad_norm_(5.0); opt.step()`, train.py:63-65, on one device) --------------------------
 * One process per GPU; the flat fp32 gradient buffer (every p.grad is a view of it) is summed in place over
 * the ranks of an RCCL communicator (xGMI), enqueued on `stream` behind the backward kernels that wrote it.
 * RCCL is resolved at run time from the librccl already loaded in the process (PyTorch-ROCm's own copy), else
 * from the system's librccl.so: this library has no link-time dependency on it.
 *   lss_rccl_get_unique_id : rank 0 fills `id_host` (lss_rccl_unique_id_bytes() = 128 bytes) and hands it to the
 *                            other ranks by any out-of-band channel (dp.py: torch.distributed's store);
 *   lss_rccl_comm_init     : collective over all `nranks` processes, the calling process's current HIP device;
 *   lss_allreduce_bucket   : buf[0..n) <- sum over ranks, in place, asynchronous on `stream`;
 *   lss_rccl_comm_destroy  : releases the communicator. */
size_t lss_rccl_unique_id_bytes(void);
int lss_rccl_version(int* version);
int lss_rccl_get_unique_id(void* id_host);
int lss_rccl_comm_init(const void* id_host, int nranks, int rank, void** comm);
int lss_allreduce_bucket(void* comm, float* buf, long long n, void* stream);
int lss_rccl_comm_destroy(void* comm);

/* Layout / dtype conversion helpers between the reference's NCHW fp32 tensors
 * and the conv path's NHWC tensors. */
int lss_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dt,
                         void* stream);
int lss_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int dt,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSS_HIP_H_ */
