"""Drop-in for the reference's `src/model_vovnet_transformer.py`: the depth heads
(`MultiScaleDepthNet` :22-70, `StandardDepthNet` :73-87), `CamEncodeV2` :90-122,
`BEVEncoderTransformer` :125-173, the TXT-branch blocks :176-351 and
`VoVNetBEVTransformer` :354-639 with its factory :642-688.

Same constructor arguments, `forward()` signatures, public methods
(`create_frustum`, `get_geometry`, `voxel_pooling`) and `state_dict` keys.  The
camera->BEV half runs on the HIP kernels of csrc/ (C = 128 context channels):

    K3 points->voxels   ||  3x3 depth-head convs (K8, MFMA)
                            K2v 1x1 depth logits + feat_proj (+ softmax for v1)
                            [v2: fusion of the two scales + softmax]
                        ->  K4 bucket  ->  K5 fused lift-splat

What is NOT here: the VoVNet trunk (`timm`, third-party weights, a by-name
network fetch in the reference, src/vovnet_timm.py:48-53).  `backbone=` accepts
any module returning {'c3': (B*N,768,H/16,W/16), 'c4': (B*N,1024,H/32,W/32)};
the default passes such feature maps straight through.  The TXT branch is stock
PyTorch (outside the hot path).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .heads import SceneUnder
from .model_BEV_TXT import _LiftSplatMixin, _histogram_guard
from .modules import _FoldedConv, _PRECISIONS, _needs_autograd, _to_nhwc, default_precision
from .tools import QuickCumsum, gen_dx_bx  # noqa: F401  (reference's import surface)
from .transformer_modules import LightweightBEVTransformer


def _conv_bn_relu_head(cin, D):
    return nn.Sequential(nn.Conv2d(cin, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                         nn.Conv2d(256, D, 1))


def _dt(precision):
    return _PRECISIONS[precision or default_precision()]


class StandardDepthNet(nn.Module):
    """Single-scale depth head: 3x3 conv-BN-ReLU, 1x1 conv, softmax over depth bins."""

    def __init__(self, c3_channels=768, depth_channels=41, precision=None):
        super().__init__()
        self.depth_head = _conv_bn_relu_head(c3_channels, depth_channels)
        self.precision = precision
        self._f = _FoldedConv(self.depth_head[0], self.depth_head[1])

    def hidden(self, c3):
        """(BN, fH, fW, 256) NHWC activations of the 3x3 conv-BN-ReLU (K8)."""
        dt = _dt(self.precision)
        return self._f.run(_to_nhwc(c3, dt), dt, relu=True)

    def depth_and_context(self, c3, c4, cam_encode):
        """One K2v launch: softmax depth (BN,D,fH,fW) + context rows (BN,fH,fW,C)."""
        last = self.depth_head[3]
        return ops.camencode_v2(self.hidden(c3), last.weight.detach(), last.bias.detach(), last.out_channels,
                                c3.float().contiguous(), cam_encode.feat_proj.weight.detach(),
                                cam_encode.feat_proj.bias.detach())

    def forward(self, c3, c4=None):
        if _needs_autograd(self, c3):
            return F.softmax(self.depth_head(c3), dim=1)
        last = self.depth_head[3]
        depth, _ = ops.camencode_v2(self.hidden(c3), last.weight.detach(), last.bias.detach(), last.out_channels)
        return depth


class MultiScaleDepthNet(nn.Module):
    """Depth logits from C3 (1/16) and C4 (1/32); the coarse logits are bilinearly
    upsampled (align_corners=False), concatenated, fused by 1x1 conv-BN-ReLU, softmax."""

    def __init__(self, c3_channels=768, c4_channels=1024, depth_channels=41, precision=None):
        super().__init__()
        D = depth_channels
        self.depth_c3 = _conv_bn_relu_head(c3_channels, D)
        self.depth_c4 = _conv_bn_relu_head(c4_channels, D)
        self.fusion = nn.Sequential(nn.Conv2d(D * 2, D, 1), nn.BatchNorm2d(D), nn.ReLU(inplace=True))
        self.precision = precision
        self._f3 = _FoldedConv(self.depth_c3[0], self.depth_c3[1])
        self._f4 = _FoldedConv(self.depth_c4[0], self.depth_c4[1])
        self._ff = _FoldedConv(self.fusion[0], self.fusion[1], pack=False)

    def _fuse(self, d3, d4):
        _, scale, shift = self._ff.get(ops.DT_F32)
        return ops.depth_fuse_softmax(d3, d4, self.fusion[0].weight.detach(), scale, shift)

    def depth_and_context(self, c3, c4, cam_encode):
        dt = _dt(self.precision)
        D = self.fusion[0].out_channels
        h3 = self._f3.run(_to_nhwc(c3, dt), dt, relu=True)
        h4 = self._f4.run(_to_nhwc(c4, dt), dt, relu=True)
        l3, l4 = self.depth_c3[3], self.depth_c4[3]
        fp = None if cam_encode is None else cam_encode.feat_proj
        d3, feat = ops.camencode_v2(h3, l3.weight.detach(), l3.bias.detach(), D,
                                    None if fp is None else c3.float().contiguous(),
                                    None if fp is None else fp.weight.detach(),
                                    None if fp is None else fp.bias.detach(), softmax=False)
        d4, _ = ops.camencode_v2(h4, l4.weight.detach(), l4.bias.detach(), D, softmax=False)
        return self._fuse(d3, d4), feat

    def forward(self, c3, c4):
        if _needs_autograd(self, c3, c4):
            d3 = self.depth_c3(c3)
            d4 = F.interpolate(self.depth_c4(c4), size=d3.shape[2:], mode="bilinear", align_corners=False)
            return F.softmax(self.fusion(torch.cat([d3, d4], dim=1)), dim=1)
        return self.depth_and_context(c3, c4, None)[0]


class CamEncodeV2(nn.Module):
    """feat_proj 1x1 conv, then depth (x) context outer product.  `forward`
    materialises the (B*N, C_out, D, H, W) lifted tensor only because that is its
    return value; the fused model path never forms it."""

    def __init__(self, D, C_in, C_out):
        super().__init__()
        self.D = D
        self.C_out = C_out
        self.feat_proj = nn.Conv2d(C_in, C_out, 1)
        self._f = _FoldedConv(self.feat_proj)

    def forward(self, features, depth):
        if _needs_autograd(self, features, depth):
            return self.feat_proj(features).unsqueeze(2) * depth.unsqueeze(1)
        feat = self._f.run(_to_nhwc(features, ops.DT_F32), ops.DT_F32, relu=False)  # (BN,H,W,C) fp32
        return feat.permute(0, 3, 1, 2).unsqueeze(2) * depth.unsqueeze(1)


class BEVEncoderTransformer(nn.Module):
    """1x1 compress conv-BN-ReLU -> deformable-attention transformer layer -> 3x3/3x3/1x1 seg head."""

    def __init__(self, in_channels, out_channels=4, precision=None):
        super().__init__()
        self.compress = nn.Sequential(nn.Conv2d(in_channels, 256, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True))
        self.transformer = LightweightBEVTransformer(d_model=256, n_heads=8, dim_feedforward=1024, dropout=0.1,
                                                     precision=precision)
        self.seg_head = nn.Sequential(
            nn.Conv2d(256, 128, 3, padding=1), nn.BatchNorm2d(128), nn.ReLU(inplace=True),
            nn.Conv2d(128, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.Conv2d(64, out_channels, 1))
        self.precision = precision
        self._fc = _FoldedConv(self.compress[0], self.compress[1])
        self._fs0 = _FoldedConv(self.seg_head[0], self.seg_head[1])
        self._fs1 = _FoldedConv(self.seg_head[3], self.seg_head[4])
        self._fs2 = _FoldedConv(self.seg_head[6])

    def forward_nhwc(self, x, dt):
        """HIP path.  x (B,H,W,C_in) NHWC in dt -> seg (B,out_C,H,W) fp32 NCHW, refined (B,H,W,256) NHWC in dt."""
        h = self._fc.run(x, dt, relu=True)
        refined = self.transformer.forward_nhwc(h, dt)
        s = self._fs0.run(refined, dt, relu=True)
        head = self.seg_head[6]
        if dt == ops.DT_BF16 and self.seg_head[3].out_channels == 64 and head.out_channels <= 64:
            # second 3x3 + BN + ReLU + the 1x1 classifier in ONE launch, NCHW fp32 out: the 64-channel
            # activation is never stored (ref: model_vovnet_transformer.py:139-143)
            w1, sc1, sh1 = self._fs1.get(dt)
            hw = head.weight.detach().float().reshape(head.out_channels, -1).contiguous()
            return ops.conv3x3_head_nchw(s, w1, sc1, sh1, hw, head.bias.detach().float().contiguous(), up=1), refined
        s = self._fs1.run(s, dt, relu=True)
        w, _, shift = self._fs2.get(dt)
        seg = ops.conv2d_nhwc(s, w, (1, 1), 1, 0, None, shift, None, False, dt=dt, out_f32=True)
        return seg.permute(0, 3, 1, 2).contiguous(), refined

    def forward(self, x):
        """(B, C_in, H, W) -> seg (B, out_C, H, W), refined (B, 256, H, W)."""
        if _needs_autograd(self, x):
            refined = self.transformer(self.compress(x))
            return self.seg_head(refined), refined
        dt = _dt(self.precision)
        seg, refined = self.forward_nhwc(_to_nhwc(x, dt), dt)
        return seg, ops.nhwc_to_nchw(refined, dt)


# ----------------------------------------------------------------------------
# TXT branch (stock PyTorch; outside the hot path)
# ----------------------------------------------------------------------------
class AdaptiveFeaturePyramid(nn.Module):
    """Two dilated 3x3 branches (d=1, d=2) fused by a 1x1 conv."""

    def __init__(self, in_channels=768, out_channels=256):
        super().__init__()

        def branch(d):
            return nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=d, dilation=d),
                                 nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

        self.scale1 = branch(1)
        self.scale2 = branch(2)
        self.fusion = nn.Sequential(nn.Conv2d(out_channels * 2, out_channels, 1), nn.BatchNorm2d(out_channels),
                                    nn.ReLU(inplace=True))

    def forward(self, x):
        return self.fusion(torch.cat([self.scale1(x), self.scale2(x)], dim=1))


class LightweightCameraTransformer(nn.Module):
    """One post-norm transformer layer over the N camera tokens of a sample."""

    def __init__(self, d_model=256, n_heads=4, dropout=0.1, n_cameras=6):
        super().__init__()
        self.cam_embed = nn.Embedding(n_cameras, d_model)
        self.self_attn = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout, batch_first=True)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(nn.Linear(d_model, d_model * 2), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(d_model * 2, d_model))

    def forward(self, x, camera_ids):
        x = x + self.cam_embed(camera_ids)
        x = self.norm1(x + self.self_attn(x, x, x)[0])
        return self.norm2(x + self.ffn(x))


class BEVCameraFusion(nn.Module):
    """Camera tokens attend to the globally pooled BEV token."""

    def __init__(self, camera_dim=256, bev_dim=256, n_heads=4):
        super().__init__()
        self.cross_attn = nn.MultiheadAttention(embed_dim=camera_dim, num_heads=n_heads, dropout=0.1, batch_first=True)
        self.norm = nn.LayerNorm(camera_dim)

    def forward(self, camera_feat, bev_feat):
        """bev_feat: (B, C, H, W) map, or its global average (B, C) if the caller already pooled it."""
        tok = (bev_feat.mean(dim=(2, 3)) if bev_feat.dim() == 4 else bev_feat).unsqueeze(1)
        return self.norm(camera_feat + self.cross_attn(camera_feat, tok, tok)[0])


class UnifiedPredictor(nn.Module):
    """Softmax-weighted mean over cameras -> shared MLP -> action / description logits."""

    def __init__(self, input_dim=256, num_action_classes=4, num_desc_classes=8, n_cameras=6):
        super().__init__()
        self.camera_weights = nn.Parameter(torch.ones(n_cameras) / n_cameras)
        self.encoder = nn.Sequential(nn.Linear(input_dim, 512), nn.LayerNorm(512), nn.GELU(), nn.Dropout(0.1),
                                     nn.Linear(512, 256), nn.LayerNorm(256), nn.GELU())
        self.action_head = nn.Linear(256, num_action_classes)
        self.desc_head = nn.Linear(256, num_desc_classes)

    def forward(self, camera_features):
        w = F.softmax(self.camera_weights, dim=0).view(1, -1, 1)
        z = self.encoder((camera_features * w).sum(dim=1))
        return self.action_head(z), self.desc_head(z)


class TrunkC3C4(nn.Module):
    """Stand-in for the reference's `VoVNetV2` slot: accepts {'c3','c4'} (or a
    (c3, c4) pair) of trunk feature maps and hands them on."""
    c3_channels = 768
    c4_channels = 1024

    def forward(self, x):
        if isinstance(x, dict):
            return x
        if isinstance(x, (tuple, list)) and len(x) == 2:
            return {"c3": x[0], "c4": x[1]}
        raise RuntimeError(
            "got camera images: the VoVNet trunk of the reference (timm ese_vovnet, third-party weights) is "
            "not bundled; pass backbone=<your trunk module> or feed {'c3': ..., 'c4': ...} feature maps")


class VoVNetBEVTransformer(_LiftSplatMixin, nn.Module):
    def __init__(self, bsize, grid_conf, data_aug_conf, outC=4, vovnet_type="vovnet57", pretrained=True,
                 lss_version="v2", use_camera_attn=True, use_cross_attn=True, backbone=None, precision=None):
        nn.Module.__init__(self)
        self.vovnet_type = vovnet_type
        self.lss_version = lss_version.lower()
        self.use_camera_attn = use_camera_attn
        self.use_cross_attn = use_cross_attn
        if self.lss_version not in {"v1", "v2"}:
            raise ValueError(f"Unsupported lss_version: {lss_version}. Use 'v1' or 'v2'.")
        self._init_grid(bsize, grid_conf, data_aug_conf)
        self.D = 41
        self.C = 128
        self.precision = precision
        self.backbone = backbone if backbone is not None else TrunkC3C4()
        c3c, c4c = self.backbone.c3_channels, self.backbone.c4_channels
        if self.lss_version == "v2":
            self.depth_net = MultiScaleDepthNet(c3_channels=c3c, c4_channels=c4c, depth_channels=self.D,
                                                precision=precision)
        else:
            self.depth_net = StandardDepthNet(c3_channels=c3c, depth_channels=self.D, precision=precision)
        self.cam_encode = CamEncodeV2(D=self.D, C_in=c3c, C_out=self.C)
        self.bev_encoder = BEVEncoderTransformer(in_channels=self.C * int(self.nx[2].item()), out_channels=outC,
                                                 precision=precision)
        self.feature_pyramid = AdaptiveFeaturePyramid(in_channels=c3c, out_channels=256)
        self.sceneunder = SceneUnder(in_channels=256)
        self.camera_names = data_aug_conf["cams"]
        self.n_cameras = len(self.camera_names)
        self.register_buffer("camera_ids", torch.arange(self.n_cameras, dtype=torch.long))
        self.camera_transformer = LightweightCameraTransformer(256, 4, 0.1, self.n_cameras) if use_camera_attn else None
        self.bev_fusion = BEVCameraFusion(256, 256, 4) if use_cross_attn else None
        self.unified_predictor = UnifiedPredictor(256, 4, 8, self.n_cameras)

    # get_geometry / voxel_pooling / create_frustum: _LiftSplatMixin (same arithmetic as
    # ref :483-554, which repeats src/model_BEV_TXT.py:37-126)

    def get_voxels(self, c3, c4, rots, trans, intrins, post_rots, post_trans, layout=ops.BEV_NCHW_F32):
        """Trunk maps + calibration -> BEV grid (B, C*nz, nx, ny) [logical shape]."""
        BN, _, fH, fW = c3.shape
        B = rots.shape[0]  # (a data.CalibrationPack has .shape = (B, N) too)
        Ncam = BN // B
        cshape = tuple(rots.shape[:2]) if trans is None else tuple(trans.shape[:2])
        if BN % B != 0 or cshape != (B, Ncam):
            raise RuntimeError("features for %d images do not match %s calibrations" % (BN, cshape))
        if (self.D, fH, fW) != tuple(self.frustum.shape[:3]):
            raise RuntimeError("feature map %dx%d / D=%d does not match the frustum %s"
                               % (fH, fW, self.D, tuple(self.frustum.shape[:3])))
        if _needs_autograd(self.depth_net, c3, c4) or _needs_autograd(self.cam_encode, c3):
            # training: library ops for the heads, then the differentiable voxel pooling
            depth = self.depth_net(c3, c4)
            cam = self.cam_encode(c3, depth).view(B, Ncam, self.C, self.D, fH, fW).permute(0, 1, 3, 4, 5, 2)
            return self.voxel_pooling(self.get_geometry(rots, trans, intrins, post_rots, post_trans), cam)
        with ops.region("lift_splat_level"):
            # depth heads + CamEncodeV2 (K2v), then geometry + bucketing + splat behind one native call: the
            # region-bucketed pipeline at C = 128 (LDS region histograms, fixed-point region splat)
            dev = self.frustum.device
            nx = self._nx_ints()
            inv_pr, comb, ptr, trn = self._device_calib(dev, rots, trans, intrins, post_rots, post_trans)
            ws = self._workspace(B * Ncam * self.D * fH * fW, B * nx[0] * nx[1] * nx[2], dev)
            depth, feat = self.depth_net.depth_and_context(c3, c4, self.cam_encode)
            with _histogram_guard(ws):  # a failed call must not leave the zero-between-calls words dirty
                return ops.lift_splat_from_heads(self.frustum.detach(), inv_pr, ptr, comb, trn, self.dx.detach(),
                                                 self.bx.detach(), depth, feat, ws, (B, Ncam, self.D, fH, fW, self.C),
                                                 nx, layout)

    def forward(self, imgs, rots, trans, intrins, post_rots, post_trans):
        """imgs: (B*N,3,H,W) / (B,N,3,H,W) camera images for a real trunk, or the
        trunk's {'c3','c4'} maps for the default pass-through.  Returns
        (bev_seg (B,outC,X,Y), action (B,4), description (B,8))."""
        if torch.is_tensor(imgs) and imgs.dim() == 5:
            imgs = imgs.reshape(-1, *imgs.shape[2:])
        B = rots.shape[0]
        feats = self.backbone(imgs)
        c3, c4 = feats["c3"], feats["c4"]
        N = c3.shape[0] // B

        be = self.bev_encoder
        if _needs_autograd(be, c3, c4) or _needs_autograd(self.depth_net) or _needs_autograd(self.cam_encode):
            bev_seg, bev_refined = be(self.get_voxels(c3, c4, rots, trans, intrins, post_rots, post_trans))
            bev_pooled = bev_refined.mean(dim=(2, 3))
        else:  # the BEV grid goes to the encoder channels-last in the conv dtype: no NCHW fp32 round trip
            dt = _dt(be.precision)
            layout = ops.BEV_NHWC_BF16 if dt == ops.DT_BF16 else ops.BEV_NHWC_F32
            grid = self.get_voxels(c3, c4, rots, trans, intrins, post_rots, post_trans, layout)
            bev_seg, refined = be.forward_nhwc(grid.permute(0, 2, 3, 1), dt)
            bev_pooled = refined.float().mean(dim=(1, 2))

        scene = self.sceneunder(self.feature_pyramid(c3))
        tokens = scene.mean(dim=(2, 3)).view(B, N, -1)
        if self.camera_transformer is not None:
            tokens = self.camera_transformer(tokens, self.camera_ids.unsqueeze(0).expand(B, -1))
        if self.bev_fusion is not None:
            tokens = self.bev_fusion(tokens, bev_pooled)
        action, description = self.unified_predictor(tokens)
        return bev_seg, action, description


def compile_model_vovnet_transformer(bsize, grid_conf, data_aug_conf, outC, vovnet_type="vovnet39", pretrained=True,
                                     lss_version="v2", use_camera_attn=True, use_cross_attn=True, **kw):
    return VoVNetBEVTransformer(bsize, grid_conf, data_aug_conf, outC, vovnet_type=vovnet_type,
                                pretrained=pretrained, lss_version=lss_version, use_camera_attn=use_camera_attn,
                                use_cross_attn=use_cross_attn, **kw)
