"""Drop-in `LSS` / `BEV_TXT` for the reference's `src/model_BEV_TXT.py`
(`LSS` :11-140, `BEV_TXT` :143-334, factories :337-340).

Same constructor arguments, same `forward(x, rots, trans, intrins, post_rots,
post_trans)` signature, same public methods (`create_frustum`, `get_geometry`,
`get_cam_feats`, `voxel_pooling`, `get_voxels`), same `state_dict` entries
(`dx bx nx frustum camencode.* bevencode.*` ...).  Underneath, the inference
camera->BEV path is three HIP launches behind ONE native call (csrc/, DESIGN.md
section 3):

    [K2 depthnet + softmax  ||  K3 points -> voxels, region histograms in LDS]   one launch
        -> K4r region fill (entries grouped by 8 x 8-cell region, no sort)
        -> K5r region splat (fixed-point LDS sums, coalesced BEV stores incl. zeros)

and the lifted (B,N,D,fH,fW,C) tensor, the sort, the cumsum and the zero-filled
scatter target of the reference never exist.  The autograd path and the API-compat
entries (`voxel_pooling`, `get_geometry`) use the voxel-list form K3 -> K4 -> K5 / K7.

What is NOT here: the EfficientNet-B4 trunk (`Encoder`, third-party weights, a
network fetch).  `encoder=` accepts any module producing the (B*N, 512, fH, fW)
trunk features; the default passes such features straight through.
"""
import os

import torch
from torch import nn

from . import ops
from .data import CalibrationPack, calib_matrices
from .heads import (BevPost, Embedder_f1, Embedder_f2, Embedder_lr1, Embedder_lr2, Predictor,
                    SceneUnder)
from .modules import BevEncode, CamEncode, _PRECISIONS, _needs_autograd, default_precision
from .tools import QuickCumsum, cumsum_trick, gen_dx_bx  # noqa: F401  (reference's import surface)
from .tools import head_weighted_cross_entropy


class TrunkFeatures(nn.Module):
    """Stand-in for the reference's `Encoder` slot: accepts trunk features
    (B*N, 512, fH, fW) or (B, N, 512, fH, fW) and returns them as (B*N, 512, fH, fW)."""

    def forward(self, x):
        if x.dim() == 5:
            if x.shape[2] == 3:
                raise RuntimeError(
                    "got raw camera images (B,N,3,H,W): the EfficientNet trunk of the reference's "
                    "Encoder is not bundled (third-party weights); pass encoder=<your trunk module> "
                    "to the model or feed (B*N,512,H/16,W/16) trunk features")
            x = x.reshape(-1, *x.shape[2:])
        return x

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """A reference checkpoint carries `encoder.trunk.*` / `encoder.up1.*` entries (ref `Encoder`,
        src/modules.py:30-66); this slot has no parameters, so `strict=True` loads (predict.py:37,78)
        would reject them as unexpected.  They are skipped here - with a warning, because the trunk
        they belong to is then NOT what produces this model's input features."""
        n = sum(1 for k in state_dict if k.startswith(prefix))
        if n:
            import warnings
            warnings.warn("%d checkpoint entries under '%s' belong to the reference's Encoder (EfficientNet trunk + up1), "
                          "which this model was built without (encoder=None: trunk features are the input); they "
                          "were ignored.  Pass encoder=lss2_multimodal_nu_amd.Encoder(trunk=...) to load them."
                          % (n, prefix), stacklevel=3)


class _LiftSplatFn(torch.autograd.Function):
    """geometry -> depthnet -> softmax -> lift -> splat as ONE differentiable op.  The forward is the inference
    pipeline (`lss_lift_splat_forward`: K2 || K3, region fill, region splat - LDS-privatised histograms, no per-point
    global atomics); it leaves the voxel ids, depth and context tensors K7 needs for the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, calib, consts, ws, dims, nx, math, layout):
        inv_pr, comb, ptr, trn = calib
        frustum, dx, bx = consts
        with _histogram_guard(ws):  # a failed call must not leave the zero-between-calls words of the workspace dirty
            bev, depth, feat = ops.lift_splat_forward(frustum, inv_pr, ptr, comb, trn, dx, bx, x.contiguous(), weight,
                                                      bias, ws, dims, nx, layout, math)
        ctx.save_for_backward(x, weight, depth, feat, ws.voxel.clone())
        ctx.dims, ctx.nx = dims, nx
        return bev

    @staticmethod
    def backward(ctx, grad_bev):
        x, weight, depth, feat, voxel = ctx.saved_tensors
        B, N, D, fH, fW, C = ctx.dims
        g_logits = ops.lift_splat_bwd(grad_bev, voxel, depth, feat, ctx.dims, ctx.nx)   # (casts only what it must)
        gl = g_logits.view(B * N, D + C, fH * fW)
        xf = x.reshape(B * N, -1, fH * fW)
        w2 = weight.reshape(D + C, -1)
        # the two plain GEMMs of the 1x1-conv backward go to the BLAS library
        gx = torch.matmul(w2.t().unsqueeze(0), gl).view_as(x) if ctx.needs_input_grad[0] else None
        # (one batched GEMM per image + a sum over the images, and two one-axis sums: the einsum / two-axis forms of the
        # same contractions cost two transposing copies and a strided reduction more - 38 + 9.5 against 21 + 6.4 us)
        gw = torch.bmm(gl, xf.transpose(1, 2)).sum(0).view_as(weight) if ctx.needs_input_grad[1] else None
        gb = gl.sum(2).sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None, None, None, None


class _VoxelPoolFn(torch.autograd.Function):
    """API-compat `voxel_pooling(geom, x)` on an already-lifted x: sums by voxel
    (K5 with unit depth weights), backward = gather of the voxel's gradient row."""

    @staticmethod
    def forward(ctx, xflat, ws, B, nx, C):
        P = xflat.shape[0]
        bev = ops.lift_splat_fwd(xflat, ws, (B, P // B, 1, 1, 1, C), nx, ops.BEV_NCHW_F32)
        ctx.save_for_backward(ws.voxel.clone())
        ctx.shape = (B, nx, C)
        return bev

    @staticmethod
    def backward(ctx, g):
        voxel, = ctx.saved_tensors
        B, (X, Y, Z), C = ctx.shape
        rows = g.view(B, Z, C, X, Y).permute(0, 3, 4, 1, 2).reshape(B * X * Y * Z, C)
        gx = rows[voxel.clamp(min=0).long()] * (voxel >= 0).unsqueeze(1)
        return gx, None, None, None, None


class _histogram_guard:
    """K3 leaves a point histogram in the cached workspace and K4 counts it back to zero (`vox_count` and
    `cursor` are zero between calls - no memset launch per step).  If anything raises between the two
    (a shape check of the depthnet operands, an allocation failure ...) the counters would stay dirty and
    every later call on that workspace would build overlapping voxel slices; this guard re-zeroes them."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        return self.ws

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None:
            self.ws.vox_count.zero_()
            self.ws.cursor.zero_()
        return False


class _LiftSplatMixin:
    """Everything `LSS` and `BEV_TXT` share on the camera->BEV path."""

    def _init_grid(self, bsize, grid_conf, data_aug_conf):
        """Grid Parameters, frustum and the host-side caches of the index path."""
        self.grid_conf = grid_conf
        self.data_aug_conf = data_aug_conf
        self.bsize = bsize
        dx, bx, nx = gen_dx_bx(grid_conf["xbound"], grid_conf["ybound"], grid_conf["zbound"])
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        self.downsample = 16
        self.frustum = self.create_frustum()
        self.use_quickcumsum = True  # kept for API compatibility; both settings take the HIP path
        self._nx_cache = None
        self._ws = {}
        self._stage = None
        self._stage_i = 0

    def _init_lift_splat(self, bsize, grid_conf, data_aug_conf, outC, encoder, precision, heads=None):
        """Sub-modules in the reference's registration order (`state_dict` key order, ref :160-176):
        encoder, [TXT heads via `heads()`], camencode, bevencode."""
        self._init_grid(bsize, grid_conf, data_aug_conf)
        self.camC = 64
        self.D = self.frustum.shape[0]
        self.encoder = encoder if encoder is not None else TrunkFeatures()
        if heads is not None:
            heads()
        self.camencode = CamEncode(self.D, self.camC, self.downsample)
        self.bevencode = BevEncode(inC=self.camC, outC=outC, precision=precision)
        self.precision = precision

    # -- grid bookkeeping ---------------------------------------------------
    def create_frustum(self):
        """(D, fH, fW, 3) image-plane sample grid (x_pix, y_pix, depth)."""
        ogfH, ogfW = self.data_aug_conf["final_dim"]
        fH, fW = ogfH // self.downsample, ogfW // self.downsample
        ds = torch.arange(*self.grid_conf["dbound"], dtype=torch.float)
        xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float)
        ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float)
        frustum = torch.stack(torch.broadcast_tensors(xs.view(1, 1, fW), ys.view(1, fH, 1), ds.view(-1, 1, 1)), -1)
        return nn.Parameter(frustum.contiguous(), requires_grad=False)

    def _nx_ints(self):
        """(X, Y, Z) as python ints; one D2H read per change of the nx Parameter."""
        key = (self.nx.data_ptr(), self.nx._version)
        if self._nx_cache is None or self._nx_cache[0] != key:
            self._nx_cache = (key, tuple(int(v) for v in self.nx.detach().cpu()))
        return self._nx_cache[1]

    def _workspace(self, P, nvox, device):
        key = (P, nvox, str(device))
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = ops.SplatWorkspace(P, nvox, device)
        return ws

    # -- calibration ----------------------------------------------------------
    def _calib_matrices(self, rots, intrins, post_rots):
        """inv(post_rots) and rots @ inv(intrins), computed with torch ON THE HOST:
        the voxel index of a point must equal the reference CPU path's bit for
        bit, and only the host LAPACK reproduces its 3x3 inverses.
        CPU calibration tensors (what a DataLoader hands over) cost no sync.
        Both inverses go through ONE flattened (2*B*N, 3, 3) call: bitwise the same
        matrices as the reference's two 4-D calls (tests/test_modules_cpu.py), ~8x less
        host time.  `data.prepare_calibration` does the same in the loader instead."""
        return calib_matrices(rots, intrins, post_rots)

    def _host_calib(self, rots, trans, intrins, post_rots, post_trans, ncams):
        """The B*N*24-float host buffer [inv_post_rots | combine | post_trans | trans] when the calibration
        lives on the CPU (DataLoader tensors or a CalibrationPack) and is small enough to travel in the kernel
        arguments; None otherwise (GPU-resident calibration keeps the device-pointer path)."""
        if ncams > ops.HOSTCAL_MAX_CAMS or os.environ.get("LSS_NO_HOSTCAL"):
            return None
        if isinstance(rots, CalibrationPack):
            return rots.buffer
        if any(t.is_cuda for t in (rots, trans, intrins, post_rots, post_trans)):
            return None
        inv_pr, comb = self._calib_matrices(rots, intrins, post_rots)
        return torch.cat([inv_pr.reshape(-1), comb.reshape(-1), post_trans.detach().float().reshape(-1),
                          trans.detach().float().reshape(-1)])

    def _device_calib(self, dev, rots, trans, intrins, post_rots, post_trans):
        """(inv_post_rots, combine, post_trans, trans) on the device.  A `CalibrationPack` passed as
        `rots` (built by the DataLoader, data.prepare_calibration) skips the host linear algebra: its
        buffer goes up in one copy; otherwise the matrices are computed here and staged."""
        if isinstance(rots, CalibrationPack):
            d = rots.buffer.to(dev, non_blocking=True)
            self._packs = (getattr(self, "_packs", ()) + (rots,))[-8:]  # keep pinned sources alive until copied
            return rots.views(d)
        inv_pr, comb = self._calib_matrices(rots, intrins, post_rots)
        inv_pr, comb, ptr, trn = self._upload(dev, inv_pr, comb, post_trans, trans)
        return inv_pr, comb, ptr, trn

    def _upload(self, device, *ts):
        """Host tensors -> one pinned staging block -> one async H2D copy -> device views.
        The staging blocks form a small ring guarded by events, so a host that runs many
        steps ahead of the GPU never overwrites a block whose copy has not executed yet."""
        total = sum(t.numel() for t in ts)
        ring = self._stage
        if ring is None or ring[0][0].numel() != total:
            ring = self._stage = [[torch.empty(total, dtype=torch.float32).pin_memory(), None] for _ in range(8)]
            self._stage_i = 0
        self._stage_i = (self._stage_i + 1) % len(ring)
        buf, ev = ring[self._stage_i]
        if ev is not None:
            ev.synchronize()  # normally long done
        o = 0
        for t in ts:
            n = t.numel()
            buf[o:o + n].copy_(t.detach().reshape(-1))
            o += n
        d = buf.to(device, non_blocking=True)
        if ring[self._stage_i][1] is None:
            ring[self._stage_i][1] = torch.cuda.Event()
        ring[self._stage_i][1].record()
        out, o = [], 0
        for t in ts:
            out.append(d[o:o + t.numel()].view(t.shape))
            o += t.numel()
        return out

    def _index_points(self, rots, trans, intrins, post_rots, post_trans, want_geom=False):
        """K3 (+K4): fills the workspace; returns (workspace, geom or None)."""
        dev = self.frustum.device
        inv_pr, comb, ptr, trn = self._device_calib(dev, rots, trans, intrins, post_rots, post_trans)
        B, Ncam = trn.shape[:2]
        D, fH, fW, _ = self.frustum.shape
        nx = self._nx_ints()
        ws = self._workspace(B * Ncam * D * fH * fW, B * nx[0] * nx[1] * nx[2], dev)
        geom = ops.points_to_voxels(self.frustum.detach(), inv_pr, ptr, comb, trn, self.dx.detach(),
                                    self.bx.detach(), nx, ws, want_geom=want_geom)
        return ws, geom

    # -- reference API ----------------------------------------------------------
    def get_geometry(self, rots, trans, intrins, post_rots, post_trans):
        """(x,y,z) ego-frame location of every frustum point: B x N x D x fH x fW x 3."""
        ws, geom = self._index_points(rots, trans, intrins, post_rots, post_trans, want_geom=True)
        with _histogram_guard(ws):
            ops.bucket_points(ws)  # leave the histogram counters at zero (K4 contract)
        return geom

    def get_cam_feats(self, x):
        """B x N x D x fH x fW x C lifted features (materialised: API compatibility only)."""
        BN, C, imH, imW = x.shape
        B = self.bsize
        N = BN // B
        x = self.camencode(x)
        x = x.view(B, N, self.camC, self.D, imH, imW)
        return x.permute(0, 1, 3, 4, 5, 2)

    def voxel_pooling(self, geom_feats, x):
        """Sum lifted features into the BEV grid: (B,N,D,H,W,3), (B,N,D,H,W,C) -> (B, C*nz, nx, ny)."""
        B, N, D, H, W, C = x.shape
        nx = self._nx_ints()
        P = B * N * D * H * W
        dev = x.device
        ws = self._workspace(P, B * nx[0] * nx[1] * nx[2], dev)
        with _histogram_guard(ws):
            ops.geom_to_voxels(geom_feats.detach().float().contiguous(), self.dx.detach(), self.bx.detach(), nx, B, ws)
            ops.bucket_points(ws)
        return _VoxelPoolFn.apply(x.reshape(P, C).float().contiguous(), ws, B, nx, C)

    def _lift_splat(self, x, rots, trans, intrins, post_rots, post_trans, layout):
        """Fused path: trunk features + calibration -> BEV (logical (B, C*nz, nx, ny))."""
        BN, _, fH, fW = x.shape
        B = self.bsize
        cshape = rots.shape if isinstance(rots, CalibrationPack) else tuple(trans.shape[:2])
        if BN % B != 0 or tuple(cshape) != (B, BN // B):
            raise RuntimeError("features for %d images do not match bsize=%d x %s cameras" % (BN, B, tuple(cshape)))
        if (self.D, fH, fW) != tuple(self.frustum.shape[:3]):
            raise RuntimeError("feature map %dx%d does not match the frustum %s"
                               % (fH, fW, tuple(self.frustum.shape[:3])))
        dims = (B, BN // B, self.D, fH, fW, self.camC)
        ce = self.camencode
        if _needs_autograd(ce, x):
            dev = self.frustum.device
            nx = self._nx_ints()
            ws = self._workspace(B * dims[1] * self.D * fH * fW, B * nx[0] * nx[1] * nx[2], dev)
            calib = self._device_calib(dev, rots, trans, intrins, post_rots, post_trans)
            return _LiftSplatFn.apply(x.float(), ce.depthnet.weight, ce.depthnet.bias, calib,
                                      (self.frustum.detach(), self.dx.detach(), self.bx.detach()), ws, dims, nx,
                                      _PRECISIONS[ce.math], layout)
        # inference: K3 -> K2 -> K4 -> K5 through one native call
        dev = self.frustum.device
        nx = self._nx_ints()
        ws = self._workspace(B * dims[1] * self.D * fH * fW, B * nx[0] * nx[1] * nx[2], dev)
        host = self._host_calib(rots, trans, intrins, post_rots, post_trans, BN) if ce.math == "fp32" else None
        if host is not None:
            # host calibration rides in the kernel arguments: no H2D copy, no staging buffer
            with ops.region("lift_splat_level"), _histogram_guard(ws):
                bev, _, _ = ops.lift_splat_forward_hostcal(self.frustum.detach(), host, self.dx.detach(), self.bx.detach(),
                                                           x.float().contiguous(), ce.depthnet.weight.detach(),
                                                           ce.depthnet.bias.detach(), ws, dims, nx, layout)
            return bev
        inv_pr, comb, ptr, trn = self._device_calib(dev, rots, trans, intrins, post_rots, post_trans)
        with ops.region("lift_splat_level"), _histogram_guard(ws):
            bev, _, _ = ops.lift_splat_forward(self.frustum.detach(), inv_pr, ptr, comb, trn, self.dx.detach(),
                                               self.bx.detach(), x.float().contiguous(), ce.depthnet.weight.detach(),
                                               ce.depthnet.bias.detach(), ws, dims, nx, layout, _PRECISIONS[ce.math])
        return bev

    def get_voxels(self, x, rots, trans, intrins, post_rots, post_trans):
        return self._lift_splat(x, rots, trans, intrins, post_rots, post_trans, ops.BEV_NCHW_F32)

    def _train_voxels(self, x, rots, trans, intrins, post_rots, post_trans):
        """`get_voxels` for the autograd path.  Under bf16 autocast BevEncode's stem takes its input as bf16 NHWC
        rows anyway (modules._ConvS2Fn), so the splat writes exactly that (layout NHWC bf16, the inference layout: the
        fp32 sums rounded once, the same bits `.to(bfloat16)` would give) and hands over the logical NCHW view - the
        (B, 64, 200, 200) fp32 NCHW grid, its permute copy and its cast (41 MB, two launches, 40 us per step) never
        exist.  fp32 training and LSS_TRAIN_NATIVE=0: `get_voxels` as it is."""
        from . import modules
        if modules._native_training() and self.bevencode.training:
            return self._lift_splat(x, rots, trans, intrins, post_rots, post_trans, ops.BEV_NHWC_BF16)
        return self.get_voxels(x, rots, trans, intrins, post_rots, post_trans)

    def _bev(self, x, rots, trans, intrins, post_rots, post_trans):
        """get_voxels + bevencode with the BEV grid handed over channels-last (bf16
        when the conv path computes in bf16) - no NCHW fp32 round trip."""
        be = self.bevencode
        if _needs_autograd(be, x) or _needs_autograd(self.camencode, x):
            return be(self._train_voxels(x, rots, trans, intrins, post_rots, post_trans))
        dt = _PRECISIONS[be.precision or default_precision()]
        layout = ops.BEV_NHWC_BF16 if dt == ops.DT_BF16 else ops.BEV_NHWC_F32
        grid = self._lift_splat(x, rots, trans, intrins, post_rots, post_trans, layout)
        return be.forward_nhwc(grid.permute(0, 2, 3, 1), dt)


_class_weight_cache = {}


def _bev_class_weights(device, class_weights=None):
    """ref src/tools.py:224,236: CrossEntropyLoss(weight=[1, 10, 5, 10]).  Cached per (device, values): no
    host-to-device copy per training step."""
    vals = (1.0, 10.0, 5.0, 10.0) if class_weights is None else tuple(float(v) for v in class_weights)
    key = (str(device), vals)
    t = _class_weight_cache.get(key)
    if t is None:
        t = _class_weight_cache[key] = torch.tensor(vals, dtype=torch.float32, device=device)
    return t


class LSS(_LiftSplatMixin, nn.Module):
    def __init__(self, bsize, grid_conf, data_aug_conf, outC, encoder=None, precision=None):
        nn.Module.__init__(self)
        self._init_lift_splat(bsize, grid_conf, data_aug_conf, outC, encoder, precision)

    def forward(self, x, rots, trans, intrins, post_rots, post_trans):
        x = self.encoder(x)
        return self._bev(x, rots, trans, intrins, post_rots, post_trans)

    def forward_loss(self, x, rots, trans, intrins, post_rots, post_trans, binimgs, class_weights=None):
        """`SimpleLoss()(self(x, ...), binimgs)` - the training loop body of the reference's pre_train.py:54-58 (loss:
        src/tools.py:221-231) - as ONE differentiable scalar with the 1x1 head and the weighted cross-entropy fused
        (SURVEY.md 8f-3): the (B, outC, X, Y) logits are never materialised.  Not part of the reference's API: an
        opt-in entry for training loops that only need the loss; `forward` + `SimpleLoss` give the same value."""
        x = self.encoder(x)
        y = self.bevencode.features(self._train_voxels(x, rots, trans, intrins, post_rots, post_trans))
        return head_weighted_cross_entropy(y, self.bevencode.up2[4], binimgs, _bev_class_weights(y.device, class_weights))


class BEV_TXT(_LiftSplatMixin, nn.Module):
    """BEV segmentation + driving action / description heads.  The BEV half is the
    HIP path above; the TXT half (ASPP scene features, per-camera embedders, linear
    predictors) is stock PyTorch, as SURVEY.md section 2 scopes it."""

    def __init__(self, bsize, grid_conf, data_aug_conf, outC, encoder=None, precision=None):
        nn.Module.__init__(self)

        def heads():
            self.sceneunder = SceneUnder()
            self.embeder_f1 = Embedder_f1(in_channels=256, out_channels=32)
            self.embeder_f2 = Embedder_f2(out_channels=40)
            self.embeder_lr1 = Embedder_lr1(in_channels=256, out_channels=32)
            self.embeder_lr2 = Embedder_lr2(out_channels=40)
            self.predictorf1 = Predictor(num_in=40, classes=4)
            self.predictorf2 = Predictor(num_in=40, classes=4)
            self.predictorlr = Predictor(num_in=40, classes=1)

        self._init_lift_splat(bsize, grid_conf, data_aug_conf, outC, encoder, precision, heads)
        self.bevpost = BevPost()

    def forward(self, x, rots, trans, intrins, post_rots, post_trans):
        x = self.encoder(x)
        bev = self._bev(x, rots, trans, intrins, post_rots, post_trans)
        act_f, desc = self._txt_heads(x, bev.detach()[:, :, 60:140, 56:144])
        return bev, act_f, desc

    def forward_loss(self, x, rots, trans, intrins, post_rots, post_trans, binimgs, act_gt, desc_gt):
        """`MultiLoss(*self(x, ...), binimgs, act_gt, desc_gt)` (ref train.py:52-62, src/tools.py:234-251) as one
        differentiable scalar with the BEV head + weighted cross-entropy fused; the TXT heads get the logits of
        their 80 x 88 crop only (detached, as in `forward`).  Opt-in, like `LSS.forward_loss`."""
        x = self.encoder(x)
        y = self.bevencode.features(self._train_voxels(x, rots, trans, intrins, post_rots, post_trans))
        head = self.bevencode.up2[4]
        loss_bev = head_weighted_cross_entropy(y, head, binimgs, _bev_class_weights(y.device))
        with torch.no_grad():
            crop = head(y[:, :, 60:140, 56:144].float())
        act_f, desc = self._txt_heads(x, crop)
        F = torch.nn.functional
        w1 = torch.tensor([1.0, 5.0, 5.0, 5.0], device=y.device)
        w2 = torch.tensor([1.0, 5.0, 5.0, 5.0, 1.0, 1.0, 1.0, 1.0], device=y.device)
        return (loss_bev + F.binary_cross_entropy_with_logits(act_f, act_gt, weight=w1)
                + F.binary_cross_entropy_with_logits(desc, desc_gt, weight=w2))

    def _txt_heads(self, x, bev_crop):
        """TXT half: the BEV crop around the ego vehicle joins every camera's features (ref :285-334)."""
        bev_post = self.bevpost(bev_crop)
        scene = self.sceneunder(x)
        ncams = self.data_aug_conf["Ncams"]
        cam = lambda i: scene[i::ncams]  # noqa: E731

        def side(i):
            y = torch.cat([self.embeder_lr1(cam(i)), bev_post], dim=1)
            return self.predictorlr(self.embeder_lr2(y))

        y_f = self.embeder_f2(torch.cat([self.embeder_f1(cam(1)), bev_post], dim=1))
        desc_f, act_f = self.predictorf1(y_f), self.predictorf2(y_f)
        desc = torch.cat([desc_f, side(0), side(3), side(2), side(5)], dim=1)
        return act_f, desc


def compile_model_lss(bsize, grid_conf, data_aug_conf, outC, **kw):
    return LSS(bsize, grid_conf, data_aug_conf, outC, **kw)


def compile_model_bevtxt(bsize, grid_conf, data_aug_conf, outC, **kw):
    return BEV_TXT(bsize, grid_conf, data_aug_conf, outC, **kw)
