"""Host-side mirror of the hot-path building blocks of the reference's
`src/modules.py`: `Up` (:9-27), `CamEncode` (:69-91), `BevEncode` (:94-130).

Constructor signatures, `forward()` signatures, parameter names / shapes and the
default initialisation follow the reference (and torchvision 0.13.1's resnet18
for `layer1..3`), so a reference `state_dict` loads with `strict=True` into the
corresponding module here.  torch.nn modules are used as PARAMETER CONTAINERS;
the inference arithmetic runs in the HIP kernels of csrc/ (conv_mfma.hip and, for
the three big 3x3 layers at the sizes it takes, conv_ring.hip; depthnet.hip) through
the C ABI - there is no eager fallback for it.

Training mode under bf16 autocast: every `conv3x3/s1 -> BatchNorm(train) -> (+residual)
-> ReLU` unit (95 % of BevEncode's FLOPs, the fused upsample + concat input of `Up`
included) is ONE autograd node on the HIP kernels (conv forward / dgrad / wgrad,
batch-statistics BN forward / backward: conv_grad.hip, bn_train.hip); the 7x7/2 stem,
the stride-2 and 1x1 convs and plain-fp32 training run the same parameters through
torch's GPU ops.  See DESIGN.md section 9.
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from . import ops

_PRECISIONS = {"bf16": ops.DT_BF16, "fp32": ops.DT_F32}


def default_precision():
    """Conv-path math: 'bf16' (bf16 MFMA, fp32 accumulate - BASELINE config 2) or
    'fp32' (f32 MFMA, exact fp32 FMA chains - the parity mode)."""
    p = os.environ.get("LSS_PRECISION", "bf16")
    if p not in _PRECISIONS:
        raise ValueError("LSS_PRECISION must be one of %s" % sorted(_PRECISIONS))
    return p


_warned_eval_autograd = set()


def _needs_autograd(module, *inputs):
    """Does this call have to be differentiable?  (Then it takes the autograd nodes - HIP units where they
    exist, torch ops otherwise - instead of the fused inference kernels.)  An eval-mode module whose
    parameters still require grad, called outside `torch.no_grad()`, lands here too: correct, but far
    slower than the inference path, so that case warns once per module class."""
    if not torch.is_grad_enabled():
        return False
    in_grad = any(t is not None and torch.is_tensor(t) and t.requires_grad for t in inputs)
    if module.training or in_grad:
        return True
    if any(p.requires_grad for p in module.parameters()):
        name = type(module).__name__
        if name not in _warned_eval_autograd:
            _warned_eval_autograd.add(name)
            import warnings
            warnings.warn("%s is in eval mode but was called with autograd enabled and trainable parameters: this "
                          "call takes the differentiable path, not the fused HIP inference kernels.  Wrap inference "
                          "in torch.no_grad() (as predict.py / get_val_info do)." % name, stacklevel=3)
        return True
    return False


class _Conv3x3Fn(torch.autograd.Function):
    """Bias-free 3x3 / stride 1 / pad 1 conv with BOTH directions on the HIP kernels: forward = K8,
    input gradient = K8 on the flipped/transposed weight pack, weight gradient = split-K MFMA GEMM
    over channel-major copies (csrc/conv_grad.hip).  Takes / returns logical (B,C,H,W) tensors; the
    arithmetic is bf16 NHWC with fp32 accumulation, the weight gradient fp32."""

    @staticmethod
    def forward(ctx, x, weight):
        xn = x.permute(0, 2, 3, 1)
        if xn.dtype != torch.bfloat16:
            xn = xn.to(torch.bfloat16)
        xn = xn.contiguous()  # no copy when x is already channels_last
        wp = ops.pack_conv_weight(weight.detach().float().contiguous(), ops.DT_BF16)
        y = ops.conv2d_nhwc(xn, wp, (3, 3), 1, 1, tag="conv2d_train_fwd")
        ctx.save_for_backward(xn, weight)
        ctx.x_dtype = x.dtype
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, weight = ctx.saved_tensors
        gyn = gy.permute(0, 2, 3, 1)
        if gyn.dtype != torch.bfloat16:
            gyn = gyn.to(torch.bfloat16)
        gyn = gyn.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wd = ops.pack_conv_weight_dgrad(weight.detach().float().contiguous(), ops.DT_BF16)
            gx = ops.conv2d_nhwc(gyn, wd, (3, 3), 1, 1, tag="conv2d_dgrad").permute(0, 3, 1, 2).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            gw = ops.conv3x3_wgrad(xn, gyn).to(weight.dtype)
        return gx, gw


def conv_s2_backward_gemm(g, x, weight, pad, need_x=True, need_w=True):
    """Both gradients of a bias-free stride-2 conv as im2col + GEMM + col2im (ATen `unfold` / `matmul` / `fold`):
    g (B,Co,Ho,Wo), x (B,C,H,W), weight (Co,C,K,K) -> (gx like x or None, gw (Co,C,K,K) fp32 or None).
    Used instead of the library's `convolution_backward` because MIOpen's weight-gradient solvers are not safe
    inside a HIP graph on this stack: the captured step replayed garbage (1e30) into `layer3.0.conv1.weight.grad`
    from the second replay on (an accumulation buffer that is only cleared outside the captured stream), and with
    `cudnn.deterministic` every gradient - found by replaying the forward + backward half alone and looking at the
    gradients.  GEMMs and the two layout kernels replay exactly."""
    B, C, H, W = x.shape
    Co, _, K, _ = weight.shape
    L = g.shape[2] * g.shape[3]
    g2 = g.reshape(B, Co, L)
    gx = gw = None
    if need_w:
        cols = F.unfold(x, K, padding=pad, stride=2)                      # (B, C*K*K, L)
        # one batched GEMM over the samples, summed in fp32 (no (C*K*K, B*L) copy of the columns: 250 MB at the stem)
        gw = torch.bmm(g2, cols.transpose(1, 2)).float().sum(0).view(Co, C, K, K)
    if need_x:
        colsg = torch.matmul(weight.reshape(Co, C * K * K).t().to(g.dtype), g2)   # (B, C*K*K, L)
        gx = F.fold(colsg, (H, W), K, padding=pad, stride=2)
    return gx, gw


_s2_tap_index = {}


def _s2_tables(device, K):
    """(phase, tap) index tensors of a K x K / stride-2 kernel on `device`: kernel row k reads input row
    2 oy + k - K // 2 = 2 (oy + d) + p, i.e. phase p and tap index t = d - d_min.  Building them is a pageable
    host-to-device copy, which a stream capture refuses: `warm_s2_tables` builds them ahead of any capture
    (dp.GraphedTrainStep calls it; BevEncode does when it is moved to a device)."""
    key = (str(device), K)
    idx = _s2_tap_index.get(key)
    if idx is None:
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("stride-2 tap tables for %s are not built yet and a stream capture is active: call "
                               "modules.warm_s2_tables(device) before capturing" % (device,))
        ds = [(k - K // 2) // 2 for k in range(K)]
        idx = _s2_tap_index[key] = (torch.tensor([(k - K // 2) % 2 for k in range(K)], device=device),
                                    torch.tensor([d - min(ds) for d in ds], device=device))
    return idx


_s2_gather_index = {}


def _s2_wgrad_gather(g6, K):
    """(Co, C, K, K) weight gradient out of K9w's phase-plane result g6 = [co][py][px][ci][ty][tx] (T x T taps): kernel
    row ky lives at phase ph[ky], tap tp[ky].  One `index_select` over a cached flat index (the advanced-indexing
    form g6[:, ph[:, None], ph[None, :], :, tp[:, None], tp[None, :]] costs six launches per call: four index
    preparations, the gather, the permute copy - 15 of a training step's launches over its three K > 1 stride-2 convs)."""
    Co, _, _, C, T, _ = g6.shape
    key = (str(g6.device), Co, C, K, T)
    idx = _s2_gather_index.get(key)
    if idx is None:
        if g6.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("stride-2 gather index %s is not built yet and a stream capture is active: run one eager "
                               "training step (or modules.warm_s2_tables) before capturing" % (key,))
        ph = torch.tensor([(k - K // 2) % 2 for k in range(K)])
        ds = [(k - K // 2) // 2 for k in range(K)]
        tp = torch.tensor([d - min(ds) for d in ds])
        co = torch.arange(Co).view(Co, 1, 1, 1)
        ci = torch.arange(C).view(1, C, 1, 1)
        ky, kx = torch.arange(K).view(1, 1, K, 1), torch.arange(K).view(1, 1, 1, K)
        flat = ((((co * 2 + ph[ky]) * 2 + ph[kx]) * C + ci) * T + tp[ky]) * T + tp[kx]
        idx = _s2_gather_index[key] = flat.reshape(-1).to(g6.device)
    return g6.reshape(-1).index_select(0, idx).view(Co, C, K, K)


def warm_s2_tables(device):
    for K in (1, 3, 7):
        _s2_tables(device, K)


def conv_s2_wgrad_phase_planes(xn, gyn, K):
    """Weight gradient of a bias-free stride-2 conv with K in {1, 3, 7} (pad K // 2) on K9w (csrc/conv_wgrad.hip): in
    phase planes xs[b, y, x, (py, px, c)] = x[b, 2y + py, 2x + px, c] the conv is a stride-1 conv with taps
    (dy, dx) in {-1, 0}^2, i.e. a sub-set of the 3x3 / pad 1 weight gradient of (xs, dy) that K9w computes:
    input row 2 oy + ky - 1 = 2 (oy + dy) + py gives ky = 0 -> (dy -1, py 1), ky = 1 -> (0, 0), ky = 2 -> (0, 1);
    the 1x1 / 2 conv is the centre tap of phase (0, 0).  (5 of the 9 taps K9w computes are not used: these are the
    two smallest weight gradients of the step.)  xn (B,H,W,C), gyn (B,H/2,W/2,Co) bf16 NHWC -> (Co,C,K,K) fp32."""
    B, H, W, C = xn.shape
    Co = gyn.shape[3]
    xs = xn.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C)
    ph, tp = _s2_tables(xn.device, K)
    if K == 7:  # the stem: taps d in {-2 .. 1} -> the 4x4-tap form of K9w (two launches of eight consumer waves)
        return _s2_wgrad_gather(ops.conv4x4_wgrad(xs, gyn).view(Co, 2, 2, C, 4, 4), K)
    g6 = ops.conv3x3_wgrad(xs, gyn).view(Co, 2, 2, C, 3, 3)   # [co][py][px][ci][ty][tx], tap t = d + 1
    if K == 1:
        return g6[:, 0, 0, :, 1, 1].reshape(Co, C, 1, 1).contiguous()
    return _s2_wgrad_gather(g6, K)


def conv_s2_dgrad_phase_planes(gyn, weight, pad, H, W):
    """Data gradient of a bias-free stride-2 conv on K8: ONE stride-1 conv over dY (2x2 taps for the 3x3 / 2 conv,
    4x4 for the 7x7 / 2 stem, 1x1 for the shortcut) that produces the four phase planes of dX, then the depth-to-space
    copy.  gyn (B,H/2,W/2,Co) bf16 NHWC, weight (Co,C,K,K) fp32 -> dX (B,H,W,C) bf16 NHWC.  Deterministic (the
    library's transposed-conv kernels are not: the loss of the captured step differed run to run with them)."""
    B, Ho, Wo, Co = gyn.shape
    C = weight.shape[1]
    KT = ops.N.lib().lss_conv2d_s2_dgrad_taps(weight.shape[2], pad)
    wp = ops.prepacked(weight.detach().float().contiguous(), ("s2_dgrad", pad),
                       lambda w: ops.pack_conv_weight_s2_dgrad(w, pad)[0])
    ys = ops.conv2d_nhwc(gyn, wp, (KT, KT), 1, KT // 2, tag="conv2d_dgrad")   # (B, Ho + 1, Wo + 1, 4 C) for KT = 2, 4
    s = 1 if KT > 1 else 0
    ys = ys[:, s:s + Ho, s:s + Wo].reshape(B, Ho, Wo, 2, 2, C)
    return ys.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def _s2_dgrad_native_ok(xn, Co, K):
    B, H, W, C = xn.shape
    return (xn.is_cuda and K in (1, 3, 7) and H % 2 == 0 and W % 2 == 0 and Co % 64 == 0 and C % 16 == 0
            and os.environ.get("LSS_S2_DGRAD_LIB") != "1" and os.environ.get("LSS_S2_DGRAD_GEMM") != "1")


def _s2_wgrad_native_ok(xn, Co, K):
    B, H, W, C = xn.shape
    return (xn.is_cuda and K in (1, 3, 7) and H % 2 == 0 and W % 2 == 0 and (4 * C) % 64 == 0 and Co % 64 == 0
            and 8 <= W // 2 <= 224 and os.environ.get("LSS_S2_WGRAD_GEMM") != "1")


class _ConvS2Fn(torch.autograd.Function):
    """Bias-free stride-2 conv (the 7x7 / 2 stem, the 3x3 / 2 first convs and the 1x1 / 2 shortcuts of layer2 / layer3;
    ref src/modules.py:99 + torchvision BasicBlock) with all three directions on the HIP kernels over PHASE PLANES
    (bf16 NHWC, fp32 accumulation): forward = the K8 phase-plane kernel of inference; data gradient = one stride-1 K8
    conv over dY that yields the four phase planes of dX (`conv_s2_dgrad_phase_planes`); weight gradient = K9w on the
    phase planes of x (`conv_s2_wgrad_phase_planes`).  The library's `convolution_backward` is kept as an A/B switch
    only (LSS_S2_DGRAD_LIB=1): its weight-gradient solvers are not safe inside a HIP graph and its data-gradient
    kernels are not deterministic."""

    @staticmethod
    def forward(ctx, x, weight, pad):
        xn = x.permute(0, 2, 3, 1)
        if xn.dtype != torch.bfloat16:
            xn = xn.to(torch.bfloat16)
        xn = xn.contiguous()  # no copy when x is already channels_last bf16
        w32 = weight.detach().float().contiguous()
        K = weight.shape[2]
        wp = (ops.prepacked(w32, ("s2d", pad), lambda w: ops.pack_conv_weight_s2d(w, pad)) if K > 1 else
              ops.prepacked(w32, ("tile",), lambda w: ops.pack_conv_weight(w, ops.DT_BF16)))
        y = ops.conv2d_s2_nhwc(xn, wp, K, pad, tag="conv2d_train_fwd")
        ctx.save_for_backward(xn, weight)
        ctx.cfg = (x.dtype, pad)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, weight = ctx.saved_tensors
        x_dtype, pad = ctx.cfg
        g = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
        x = xn.permute(0, 3, 1, 2)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            # the WEIGHT gradient never goes to the library: its solvers are the ones that broke graph replays.
            # 3x3 / 2 and 1x1 / 2: K9w over phase planes; the 7x7 / 2 stem (taps dy in {-2 .. 1}): im2col + GEMM
            K = weight.shape[2]
            if _s2_wgrad_native_ok(xn, weight.shape[0], K):
                gw = conv_s2_wgrad_phase_planes(xn, g.permute(0, 2, 3, 1).contiguous(), K)
            else:
                _, gw = conv_s2_backward_gemm(g.contiguous(), x, weight.detach(), pad, need_x=False)
        if ctx.needs_input_grad[0]:
            if _s2_dgrad_native_ok(xn, weight.shape[0], weight.shape[2]):
                gx = conv_s2_dgrad_phase_planes(g.permute(0, 2, 3, 1).contiguous(), weight, pad, xn.shape[1],
                                                xn.shape[2]).permute(0, 3, 1, 2)
            elif x.is_cuda and os.environ.get("LSS_S2_DGRAD_GEMM") != "1":
                # data gradient: the library's transposed-conv kernels (no accumulation buffers; replay-checked by
                # dp.GraphedTrainStep._self_check and tests/test_dp_gpu.py on every replay); 4x cheaper than col2im
                gx = torch.ops.aten.convolution_backward(
                    g.contiguous(memory_format=torch.channels_last), x, weight.detach().to(torch.bfloat16), None,
                    [2, 2], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            else:
                gx, _ = conv_s2_backward_gemm(g.contiguous(), x, weight.detach(), pad, need_w=False)
        return (None if gx is None else gx.to(x_dtype), None if gw is None else gw.to(weight.dtype), None)


class _UpConv3x3Fn(torch.autograd.Function):
    """conv3x3(cat([x2, bilinear_align_corners(x1, up)])) with the upsample and the concat fused into
    the conv's operand gather in the forward (neither tensor exists), and in the backward: one dgrad
    conv for the concatenated input, the upsample's adjoint as a gather kernel, and the concatenated
    input re-materialised in bf16 only for the weight-gradient GEMM.  x2 may be None (plain upsample)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, up):
        def nhwc(t):
            t = t.permute(0, 2, 3, 1)
            return (t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)).contiguous()

        x1n = nhwc(x1)
        x2n = None if x2 is None else nhwc(x2)
        wp = ops.pack_conv_weight(weight.detach().float().contiguous(), ops.DT_BF16)
        y = ops.conv2d_nhwc(x1n, wp, (3, 3), 1, 1, x2=x2n, up=up, tag="conv2d_train_fwd")
        ctx.save_for_backward(x1n, x2n, weight)
        ctx.up = up
        ctx.dtypes = (x1.dtype, None if x2 is None else x2.dtype)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x1n, x2n, weight = ctx.saved_tensors
        up = ctx.up
        C2 = 0 if x2n is None else x2n.shape[3]
        Cx = x1n.shape[3]
        gyn = gy.permute(0, 2, 3, 1)
        gyn = (gyn if gyn.dtype == torch.bfloat16 else gyn.to(torch.bfloat16)).contiguous()
        g1 = g2 = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wd = ops.pack_conv_weight_dgrad(weight.detach().float().contiguous(), ops.DT_BF16)
            gcat = ops.conv2d_nhwc(gyn, wd, (3, 3), 1, 1, tag="conv2d_dgrad")  # (B, H*up, W*up, C2+Cx)
            if ctx.needs_input_grad[0]:
                g1 = ops.upsample_bwd_nhwc(gcat, C2, Cx, up).permute(0, 3, 1, 2).to(ctx.dtypes[0])
            if x2n is not None and ctx.needs_input_grad[1]:
                g2 = gcat[..., :C2].permute(0, 3, 1, 2).to(ctx.dtypes[1])
        if ctx.needs_input_grad[2]:
            gw = ops.conv3x3_wgrad(ops.upsample_cat_nhwc(x1n, x2n, up), gyn).to(weight.dtype)
        return g1, g2, gw, None


class _BNActFn(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ residual add, + ReLU) on the HIP kernels (csrc/bn_train.hip),
    forward and backward; logical (B,C,H,W) tensors in, computed on bf16 NHWC rows.  Running
    statistics are updated in place like nn.BatchNorm2d (momentum must be a number)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, residual, running_mean, running_var, momentum, eps, relu):
        def nhwc(t):
            t = t.permute(0, 2, 3, 1)
            return (t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)).contiguous()

        zn = nhwc(z)
        rn = None if residual is None else nhwc(residual)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, mean, invstd = ops.bn_train_fwd(zn, g32, b32, running_mean, running_var, momentum, eps, relu, rn)
        ctx.save_for_backward(zn, y, g32, mean, invstd)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.dtypes = (z.dtype, None if residual is None else residual.dtype, gamma.dtype)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        zn, y, g32, mean, invstd = ctx.saved_tensors
        gyn = gy.permute(0, 2, 3, 1)
        gyn = (gyn if gyn.dtype == torch.bfloat16 else gyn.to(torch.bfloat16)).contiguous()
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        dz, dres, dgamma, dbeta = ops.bn_train_bwd(gyn, y, zn, g32, mean, invstd, ctx.relu, want_res)
        gz = dz.permute(0, 3, 1, 2).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        gres = dres.permute(0, 3, 1, 2).to(ctx.dtypes[1]) if want_res else None
        return (gz, dgamma.to(ctx.dtypes[2]) if ctx.needs_input_grad[1] else None,
                dbeta.to(ctx.dtypes[2]) if ctx.needs_input_grad[2] else None, gres, None, None, None, None, None)


def _nhwc_bf16(t):
    t = t.permute(0, 2, 3, 1)
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    return t if t.is_contiguous() else t.contiguous()


class _ConvBNActFn(torch.autograd.Function):
    """One autograd node for a whole `conv3x3 -> BatchNorm(train) -> (+residual) -> ReLU` unit, with the
    optional fused `cat([x2, upsample(x1)])` input: conv forward / dgrad / wgrad, BN forward /
    backward and the upsample adjoint all on the HIP kernels.  One node instead of three keeps the
    Python/autograd overhead per layer below the kernels' own time (the training step is otherwise
    host-bound: ~340 us of framework time per conv+BN unit)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, up):
        x1n = _nhwc_bf16(x1)
        x2n = None if x2 is None else _nhwc_bf16(x2)
        rn = None if residual is None else _nhwc_bf16(residual)
        w, g32 = weight.detach(), gamma.detach()
        z, y, stat = ops.conv_bn_act_train_fwd(x1n, x2n, w, g32, beta.detach(), rn, running_mean, running_var,
                                               momentum, eps, relu, up)
        ctx.save_for_backward(x1n, x2n, z, y, w, g32, stat)
        ctx.cfg = (relu, up, residual is not None, x1.dtype, None if x2 is None else x2.dtype,
                   None if residual is None else residual.dtype)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x1n, x2n, z, y, w, g32, stat = ctx.saved_tensors
        relu, up, has_res, dt1, dt2, dtr = ctx.cfg
        need = ctx.needs_input_grad
        g1, g2, gw, dgamma, dbeta, dres = ops.conv_bn_act_train_bwd(
            _nhwc_bf16(gy), y, z, x1n, x2n, w, g32, stat, relu, up, has_res and need[5], need[0],
            x2n is not None and need[1], need[2])

        def out(t, dt):
            if t is None:
                return None
            t = t.permute(0, 3, 1, 2)
            return t if t.dtype == dt else t.to(dt)

        return (out(g1, dt1), out(g2, dt2), gw, dgamma if need[3] else None, dbeta if need[4] else None,
                out(dres, dtr), None, None, None, None, None, None)


class _SyncBNActFn(torch.autograd.Function):
    """BatchNorm(train) + residual + ReLU with statistics over ALL data-parallel ranks: local
    per-channel sums on the HIP kernels, ONE small all-reduce (2*C floats) per direction over the
    process group, then the normalisation with the global statistics.  Parameter gradients are the
    local sums (the gradient all-reduce of the DP step averages them, like torch's SyncBatchNorm)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, group):
        import torch.distributed as dist
        zn = _nhwc_bf16(z)
        rn = None if residual is None else _nhwc_bf16(residual)
        g32 = gamma.detach()
        sums = ops.bn_partial_sums(zn, 0, mean=running_mean)  # sums about the running mean (shared by all ranks)
        world = dist.get_world_size(group)
        dist.all_reduce(sums, group=group)
        m_total = (zn.numel() // zn.shape[-1]) * world  # equal shards (drop_last loaders, ref src/data.py:294)
        y, mean, invstd = ops.bn_train_fwd_from_sums(zn, sums, m_total, g32, beta.detach(), running_mean, running_var,
                                                     momentum, eps, relu, rn)
        ctx.save_for_backward(zn, y, g32, mean, invstd)
        ctx.cfg = (relu, residual is not None, group, m_total, z.dtype, None if residual is None else residual.dtype)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        zn, y, g32, mean, invstd = ctx.saved_tensors
        relu, has_res, group, m_total, dtz, dtr = ctx.cfg
        gyn = _nhwc_bf16(gy)
        local = ops.bn_partial_sums(zn, 1, dy=gyn, y=y, mean=mean, invstd=invstd, relu=relu)
        glob = local.clone()
        dist.all_reduce(glob, group=group)
        want_res = has_res and ctx.needs_input_grad[3]
        dz, dres = ops.bn_train_bwd_from_sums(gyn, y, zn, glob, m_total, g32, mean, invstd, relu, want_res)

        def out(t, dt):
            if t is None:
                return None
            t = t.permute(0, 3, 1, 2)
            return t if t.dtype == dt else t.to(dt)

        return (out(dz, dtz), local[1] if ctx.needs_input_grad[1] else None, local[0] if ctx.needs_input_grad[2] else None,
                out(dres, dtr), None, None, None, None, None, None)


def enable_sync_bn(module, group=None):
    """Make every BatchNorm2d under `module` that runs on the HIP training units use statistics over
    all ranks of `group` (default: the world group).  Call after torch.distributed is initialised;
    `enable_sync_bn(module, False)` turns it off again."""
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            if group is False:
                m.__dict__.pop("_lss_sync", None)
            else:
                m.__dict__["_lss_sync"] = (group,)
    return module


def x_is_capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


import threading  # noqa: E402

_tls = threading.local()  # .counters: a list while BevEncode.features (of THIS thread) collects its BatchNorms' counters


def _count_batch(bn):
    """`num_batches_tracked += 1` of a BatchNorm that ran on a native unit (torch's own forward does it itself).
    Inside BevEncode.features the 18 counters are bumped by ONE foreach launch at the end instead of 18 kernels."""
    pending = getattr(_tls, "counters", None)
    if pending is not None:
        pending.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def _bn_native_ok(bn, z_is_cuda):
    C = bn.num_features
    return (z_is_cuda and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
            and C % 8 == 0 and 256 % (C // 8) == 0 and bn.weight.dtype == torch.float32)


def _train_conv_bn_act(conv, bn, x1, relu, residual=None, up=None, x2=None):
    """act(bn(conv(x)) (+ residual)) on the autograd path, x = x1 or cat([x2, up(x1)]).  One fused
    HIP node when the shapes allow and the caller asked for bf16 math; otherwise composed from
    the separate (native or library) pieces."""
    c2 = 0 if x2 is None else x2.shape[1]
    scale = 1 if up is None else int(up.scale_factor)
    if (x1.is_cuda and _native_training() and _bn_native_ok(bn, True) and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.bias is None and conv.weight.dtype == torch.float32 and x1.shape[1] % 64 == 0 and c2 % 64 == 0
            and conv.out_channels % 8 == 0 and (scale == 1 or (x1.shape[2] > 1 and x1.shape[3] > 1))):
        if "_lss_sync" not in bn.__dict__:
            y = _ConvBNActFn.apply(x1, x2, conv.weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                                   float(bn.momentum), float(bn.eps), bool(relu), scale)
            _count_batch(bn)
            return y
    z = _train_conv(conv, x1) if up is None else _train_up_conv(conv, up, x1, x2)
    return _train_bn_act(bn, z, relu, residual)


def _train_bn_act(bn, z, relu, residual=None):
    """act(bn(z) (+ residual)) on the autograd path: one fused HIP forward/backward pair when the
    caller asked for bf16 math and the module is in training mode, the library ops otherwise."""
    if _native_training() and _bn_native_ok(bn, z.is_cuda):
        if "_lss_sync" in bn.__dict__:
            y = _SyncBNActFn.apply(z, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                                   float(bn.momentum), float(bn.eps), bool(relu), bn.__dict__["_lss_sync"][0])
        else:
            y = _BNActFn.apply(z, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, float(bn.momentum),
                               float(bn.eps), bool(relu))
        _count_batch(bn)
        return y
    y = bn(z)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _native_training():
    """Training convs go to the HIP kernels when the caller asked for bf16 math (bf16 autocast);
    plain fp32 training keeps torch's fp32 convolutions.  LSS_TRAIN_NATIVE=0 disables."""
    return (os.environ.get("LSS_TRAIN_NATIVE", "1") != "0" and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def _train_up_conv(conv, up, x1, x2):
    """conv(cat([x2, upsample(x1)])) on the autograd path (x2 may be None)."""
    c2 = 0 if x2 is None else x2.shape[1]
    if (x1.is_cuda and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.bias is None and x1.shape[1] % 64 == 0 and c2 % 64 == 0 and conv.out_channels % 8 == 0
            and x1.shape[2] > 1 and x1.shape[3] > 1 and _native_training()):
        return _UpConv3x3Fn.apply(x1, x2, conv.weight, int(up.scale_factor))
    x1 = up(x1)
    return _train_conv(conv, x1 if x2 is None else torch.cat([x2, x1], dim=1))


def _train_conv(conv, x):
    """conv(x) on the autograd path: the 3x3/s1/p1 shapes (95 % of BevEncode's FLOPs) run
    forward and backward on the HIP kernels, everything else on the library."""
    if (x.is_cuda and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.in_channels % 64 == 0 and conv.out_channels % 8 == 0 and _native_training()):
        return _Conv3x3Fn.apply(x, conv.weight)
    k, p = conv.kernel_size[0], conv.padding[0]
    if (x.is_cuda and conv.stride == (2, 2) and conv.kernel_size in ((1, 1), (3, 3), (7, 7)) and conv.padding == (k // 2, k // 2)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and _native_training()):
        return _ConvS2Fn.apply(x, conv.weight, p)
    return conv(x)


class _FoldedConv:
    """Packed weights + eval-mode BatchNorm folded into (scale, shift) for one
    conv of the HIP path; rebuilt only when a source tensor changes."""

    def __init__(self, conv, bn=None, pack=True, pad_in=None):
        self.conv, self.bn = conv, bn
        self.pack = pack  # False: only the folded (scale, shift) are wanted
        # (at, n): insert n zero input channels at channel `at` of the packed weight - lets a
        # skip tensor whose width is not a multiple of the kernels' 64-channel K block be
        # zero-padded (its padding channels then meet zero weights)
        self.pad_in = pad_in
        self.key = None
        self.w = self.scale = self.shift = None
        self.w_ring, self.ring_key = None, None
        self.w_ks, self.ks_key = None, None

    def ks(self, dt):
        """The same weights in the K-split one-pass kernel's layout (csrc/conv_ks.hip), packed on first use and
        whenever a source tensor changes; `get(dt)` must have been called for the current key."""
        if self.ks_key != self.key:
            with torch.no_grad():
                self.w_ks = ops.pack_conv_weight_ks(self.conv.weight.detach().float().contiguous())
            self.ks_key = self.key
        return self.w_ks

    def ks_case(self, x, dt, x2=None, up=1):
        """Does this call go to the K-split one-pass kernel?  3x3 / stride 1 / pad 1, bf16, no fused gather, and a shape
        `lss_conv2d_ks_ok` accepts (layer1-3 of BevEncode at the benchmark sizes)."""
        c = self.conv
        if (dt != ops.DT_BF16 or x2 is not None or up != 1 or c.kernel_size != (3, 3) or c.stride != (1, 1)
                or c.padding != (1, 1) or not self.pack or self.pad_in is not None):
            return False
        B, H, W, Cx = x.shape
        return ops.conv_ks_ok(B, H, W, Cx, c.out_channels)

    def ring(self, dt):
        """The same weights in the ring kernel's layout (csrc/conv_ring.hip), packed on first use and whenever a
        source tensor changes; `get(dt)` must have been called for the current key."""
        if self.ring_key != self.key:
            with torch.no_grad():
                w32 = self.conv.weight.detach().float().contiguous()
                if self.pad_in is not None:
                    at, n = self.pad_in
                    w32 = torch.cat([w32[:, :at], w32.new_zeros(w32.shape[0], n, *w32.shape[2:]), w32[:, at:]],
                                    1).contiguous()
                self.w_ring = ops.pack_conv_weight_ring(w32)
            self.ring_key = self.key
        return self.w_ring

    def ring_case(self, x, dt, residual=None, x2=None, up=1, head_n=0):
        """Does this call go to the loader / consumer ring kernel?  3x3 / stride 1 / pad 1, bf16, no residual, and a
        shape `lss_conv2d_ring_ok` accepts (the three big layers of BevEncode at the benchmark sizes)."""
        c = self.conv
        if (dt != ops.DT_BF16 or residual is not None or c.kernel_size != (3, 3) or c.stride != (1, 1)
                or c.padding != (1, 1) or not self.pack):
            return False
        B, H, W, Cx = x.shape
        return ops.conv_ring_ok(B, H, W, Cx, x2.shape[3] if x2 is not None else 0, up, c.out_channels, head_n)

    def _key(self, dt):
        # (storage pointer, in-place version) of every tensor the folded form depends on
        c, bn = self.conv, self.bn
        k = (dt, c.weight.data_ptr(), c.weight._version)
        if c.bias is not None:
            k += (c.bias._version,)
        if bn is not None:
            k += (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
                  bn.running_mean.data_ptr())
        return k

    def _s2d(self, dt):
        """Stride-2 3x3/pad1 and 7x7/pad3 convs run on the LDS-tiled kernel through the
        phase-plane (space-to-depth) weight arrangement (bf16 path only)."""
        c = self.conv
        return (dt == ops.DT_BF16 and c.stride[0] == 2 and c.kernel_size in ((1, 1), (3, 3), (7, 7))
                and c.padding[0] == c.kernel_size[0] // 2 and c.in_channels % 64 == 0)

    def get(self, dt):
        key = self._key(dt)
        if key != self.key:
            with torch.no_grad():
                w32 = self.conv.weight.detach().float().contiguous()
                if self.pad_in is not None:
                    at, n = self.pad_in
                    w32 = torch.cat([w32[:, :at], w32.new_zeros(w32.shape[0], n, *w32.shape[2:]), w32[:, at:]],
                                    1).contiguous()
                # (the 1x1/2 conv reads parity phase (0,0) only: plain pack)
                if not self.pack:
                    self.w = None
                elif self._s2d(dt) and self.conv.kernel_size[0] > 1:
                    self.w = ops.pack_conv_weight_s2d(w32, self.conv.padding[0])
                else:
                    self.w = ops.pack_conv_weight(w32, dt)
                if self.bn is not None:
                    inv = torch.rsqrt(self.bn.running_var.float() + self.bn.eps)
                    self.scale = (self.bn.weight.float() * inv).contiguous()
                    mean = self.bn.running_mean.float()
                    if self.conv.bias is not None:  # conv bias in front of the BN (vovnet heads)
                        mean = mean - self.conv.bias.detach().float()
                    self.shift = (self.bn.bias.float() - mean * self.scale).contiguous()
                else:
                    self.scale = None
                    self.shift = self.conv.bias.detach().float().contiguous() if self.conv.bias is not None else None
            self.key = key
        return self.w, self.scale, self.shift

    def run(self, x, dt, relu, residual=None, x2=None, up=1):
        w, scale, shift = self.get(dt)
        c = self.conv
        if self._s2d(dt) and x2 is None and up == 1:
            return ops.conv2d_s2_nhwc(x, w, c.kernel_size[0], c.padding[0], scale, shift, residual, relu)
        if relu in (False, True) and self.ks_case(x, dt, x2, up):
            w = self.ks(dt)
        elif relu in (False, True) and self.ring_case(x, dt, residual, x2, up):
            w = self.ring(dt)
        return ops.conv2d_nhwc(x, w, c.kernel_size, c.stride[0], c.padding[0], scale, shift, residual, relu,
                               x2=x2, up=up, dt=dt)


def _to_nhwc(x, dt):
    """(B,C,H,W) fp32 (contiguous or channels_last) -> (B,H,W,C) activations in dt."""
    if x.dtype == torch.bfloat16 and dt == ops.DT_BF16 and x.is_contiguous(memory_format=torch.channels_last):
        return x.permute(0, 2, 3, 1)  # already NHWC bf16 (the fused splat hand-off)
    x = x.float()
    if dt == ops.DT_F32 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return x.permute(0, 2, 3, 1)
    return ops.nchw_to_nhwc(x.contiguous(), dt)


class Up(nn.Module):
    """Bilinear (align_corners=True) upsample of x1, concat [x2, x1], two 3x3
    conv-BN-ReLU.  On the HIP path the upsample and the concat are fused into the
    first conv's operand gather: neither tensor is materialised."""

    def __init__(self, in_channels, out_channels, scale_factor=2, precision=None):
        super().__init__()
        self.up = nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=True)
        self.conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))
        self.scale_factor = int(scale_factor)
        self.precision = precision
        self._f0 = _FoldedConv(self.conv[0], self.conv[1])
        self._f1 = _FoldedConv(self.conv[3], self.conv[4])
        self._f0_c2 = None  # skip-tensor width the padded form of conv[0] was built for

    def _nhwc(self, x1, x2, dt):
        kb = 64 if dt == ops.DT_BF16 else 8  # K block of the conv kernels
        c2 = x2.shape[3]
        if c2 % kb != 0:
            # e.g. Encoder.up1: 160 skip channels (ref: src/modules.py:32).  Pad the skip tensor with
            # zero channels up to the K block; the packed weight gets matching zero columns.
            padc = kb - c2 % kb
            if self._f0_c2 != c2:
                self._f0 = _FoldedConv(self.conv[0], self.conv[1], pad_in=(c2, padc))
                self._f0_c2 = c2
            x2 = F.pad(x2, (0, padc))
        y = self._f0.run(x1, dt, relu=True, x2=x2, up=self.scale_factor)
        return self._f1.run(y, dt, relu=True)

    def forward(self, x1, x2):
        if _needs_autograd(self, x1, x2):
            c = self.conv
            y = _train_conv_bn_act(c[0], c[1], x1, relu=True, up=self.up, x2=x2)
            return _train_conv_bn_act(c[3], c[4], y, relu=True)
        dt = _PRECISIONS[self.precision or default_precision()]
        y = self._nhwc(_to_nhwc(x1, dt), _to_nhwc(x2, dt), dt)
        return ops.nhwc_to_nchw(y, dt)


class CamEncode(nn.Module):
    """depthnet 1x1 conv -> softmax over D depth bins -> depth (x) context outer
    product.  `forward` / `get_depth_feat` materialise the (B*N, C, D, fH, fW)
    lifted tensor only because that is their return value; the fused LSS path
    calls `depth_and_context` and never forms it."""

    def __init__(self, D, C, downsample, math="fp32"):
        super().__init__()
        self.D = D
        self.C = C
        self.depthnet = nn.Conv2d(512, self.D + self.C, kernel_size=1, padding=0)
        self.math = math

    def get_depth_dist(self, x, eps=1e-20):
        return x.softmax(dim=1)

    def depth_and_context(self, x):
        """HIP K2: x (BN,Cin,fH,fW) -> depth (BN,D,fH,fW), context (BN,fH,fW,C)."""
        return ops.depthnet_softmax(x.float().contiguous(), self.depthnet.weight.detach(),
                                    self.depthnet.bias.detach(), self.D, self.C, _PRECISIONS[self.math])

    def get_depth_feat(self, x):
        if _needs_autograd(self, x):
            y = self.depthnet(x)
            depth = self.get_depth_dist(y[:, :self.D])
            return depth, depth.unsqueeze(1) * y[:, self.D:(self.D + self.C)].unsqueeze(2)
        depth, ctx = self.depth_and_context(x)
        return depth, depth.unsqueeze(1) * ctx.permute(0, 3, 1, 2).unsqueeze(2)

    def forward(self, x):
        depth, x = self.get_depth_feat(x)
        return x


class Encoder(nn.Module):
    """Mirror of the reference's `Encoder` (src/modules.py:28-66) from the trunk's endpoints on:
    `up1 = Up(448 + 160, 512)` fuses EfficientNet-B4's reduction_5 (1/32, 448 ch, upsampled x2)
    with reduction_4 (1/16, 160 ch) into the (B*N, 512, H/16, W/16) map CamEncode consumes.

    The EfficientNet itself is third-party (`efficientnet_pytorch`, a by-name weight fetch in the
    reference) and is NOT bundled: `trunk` is any module whose call returns the endpoints
    {'reduction_4': (B*N,160,H/16,W/16), 'reduction_5': (B*N,448,H/32,W/32)} for (B*N,3,H,W)
    images (efficientnet_pytorch's `extract_endpoints` has exactly this contract); without a
    trunk, `forward` takes that dict (or a (reduction_5, reduction_4) pair) directly.
    `up1.*` state_dict keys equal the reference's."""

    def __init__(self, trunk=None, c5=448, c4=160, precision=None):
        super().__init__()
        if trunk is not None:
            self.trunk = trunk
        self.up1 = Up(c5 + c4, 512, precision=precision)

    def get_eff_depth(self, x):
        if torch.is_tensor(x):
            if getattr(self, "trunk", None) is None:
                raise RuntimeError("got camera images but no trunk: pass trunk=<EfficientNet endpoints module> or "
                                   "feed {'reduction_4': ..., 'reduction_5': ...}")
            if x.dim() == 5:
                x = x.reshape(-1, *x.shape[2:])
            fn = getattr(self.trunk, "extract_endpoints", self.trunk)
            x = fn(x)
        if isinstance(x, dict):
            x = (x["reduction_5"], x["reduction_4"])
        r5, r4 = x
        return self.up1(r5, r4)

    def forward(self, x):
        return self.get_eff_depth(x)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """Without a trunk, the `trunk.*` entries of a reference checkpoint (the EfficientNet weights) have
        no home here; they are dropped with a warning instead of failing a `strict=True` load of the
        whole model (`up1.*` still loads strictly)."""
        if getattr(self, "trunk", None) is None:
            drop = [k for k in state_dict if k.startswith(prefix + "trunk.")]
            if drop:
                import warnings
                warnings.warn("%d '%strunk.*' checkpoint entries ignored: this Encoder was built without a trunk"
                              % (len(drop), prefix), stacklevel=3)
                for k in drop:
                    del state_dict[k]  # load_state_dict works on its own shallow copy of the caller's dict
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class BasicBlock(nn.Module):
    """torchvision resnet BasicBlock (expansion 1): conv3x3-BN-ReLU-conv3x3-BN,
    + identity or [conv1x1(stride)-BN], ReLU."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))
        self._f1 = _FoldedConv(self.conv1, self.bn1)
        self._f2 = _FoldedConv(self.conv2, self.bn2)
        self._fd = _FoldedConv(self.downsample[0], self.downsample[1]) if self.downsample is not None else None

    def forward(self, x):
        idt = x if self.downsample is None else _train_bn_act(self.downsample[1], _train_conv(self.downsample[0], x), relu=False)
        out = _train_conv_bn_act(self.conv1, self.bn1, x, relu=True)
        return _train_conv_bn_act(self.conv2, self.bn2, out, relu=True, residual=idt)

    def _dual(self, dt):
        """conv1 (3x3/2) and the 1x1/2 downsample as ONE launch: both weight sets in the phase-plane layout
        (the 1x1 weight sits at the centre tap of a 3x3 frame), stacked along Cout; rebuilt when a source changes."""
        key = self._f1._key(dt) + self._fd._key(dt)
        if getattr(self, "_dual_key", None) != key:
            with torch.no_grad():
                w1, s1, b1 = self._f1.get(dt)
                _, sd, bd = self._fd.get(dt)
                wd = self.downsample[0].weight.detach().float()
                frame = wd.new_zeros(wd.shape[0], wd.shape[1], 3, 3)
                frame[:, :, 1, 1] = wd[:, :, 0, 0]
                self._dual_pack = (torch.cat([w1, ops.pack_conv_weight_s2d(frame.contiguous(), 1)], 1).contiguous(),
                                   torch.cat([s1, sd]).contiguous(), torch.cat([b1, bd]).contiguous())
            self._dual_key = key
        return self._dual_pack

    def _nhwc(self, x, dt):
        c1 = self.conv1
        if (self._fd is not None and dt == ops.DT_BF16 and c1.stride == (2, 2) and self._f1._s2d(dt)
                and self.downsample[0].stride == (2, 2) and c1.out_channels % 128 == 0
                and os.environ.get("LSS_NO_DUAL") is None):
            w, scale, shift = self._dual(dt)
            t, idt = ops.conv2d_s2_dual_nhwc(x, w, scale, shift, c1.out_channels, relu=True)
            return self._f2.run(t, dt, relu=True, residual=idt)
        idt = x if self._fd is None else self._fd.run(x, dt, relu=False)
        t = self._f1.run(x, dt, relu=True)
        return self._f2.run(t, dt, relu=True, residual=idt)


def _resnet18_layer(inplanes, planes, stride):
    return nn.Sequential(BasicBlock(inplanes, planes, stride), BasicBlock(planes, planes, 1))


class BevEncode(nn.Module):
    """BEV decoder: 7x7/2 stem, resnet18 layer1-3, Up(x4) with the layer1 skip,
    x2 upsample + 3x3 conv + 1x1 head.  (B, inC, X, Y) -> (B, outC, X, Y)."""

    def __init__(self, inC, outC, precision=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inC, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = _resnet18_layer(64, 64, 1)
        self.layer2 = _resnet18_layer(64, 128, 2)
        self.layer3 = _resnet18_layer(128, 256, 2)
        # resnet18(pretrained=False, zero_init_residual=True) initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, BasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        # conv1 is a plain nn.Conv2d in the reference (default init), as are up1 / up2
        self.conv1.reset_parameters()
        self.up1 = Up(64 + 256, 256, scale_factor=4, precision=precision)
        self.up2 = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(256, 128, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(128),
            nn.ReLU(inplace=True),
            nn.Conv2d(128, outC, kernel_size=1, padding=0),
        )
        self.precision = precision
        self._plans = {}
        self._stem = _FoldedConv(self.conv1, self.bn1)
        self._up2a = _FoldedConv(self.up2[1], self.up2[2])
        self._up2b = _FoldedConv(self.up2[4], None)

    def features(self, x):
        """Differentiable path up to (not including) the 1x1 head `up2[4]`: (B, inC, X, Y) -> (B, 128, X, Y), the
        activation `tools.head_weighted_cross_entropy` fuses the head and the loss on."""
        if _native_training() and x.is_cuda:
            # the whole chain stays NHWC; a bf16 NHWC grid (the training splat's, model_BEV_TXT._train_voxels) is what the
            # stem reads anyway: taken as it is, no fp32 round trip
            if x.dtype not in (torch.float32, torch.bfloat16):
                x = x.float()
            x = x.contiguous(memory_format=torch.channels_last)
        elif x.dtype != torch.float32:
            x = x.float()
        outer, _tls.counters = getattr(_tls, "counters", None), []
        try:
            x = _train_bn_act(self.bn1, _train_conv(self.conv1, x), relu=True)
            x1 = self.layer1(x)
            x = self.layer3(self.layer2(x1))
            x = self.up1(x, x1)
            u = self.up2
            return _train_conv_bn_act(u[1], u[2], x, relu=True, up=u[0])
        finally:
            counters, _tls.counters = _tls.counters, outer
            if counters:
                torch._foreach_add_(counters, 1)

    def _forward_autograd(self, x):
        from .tools import head_1x1  # (tools imports nothing from this module)
        return head_1x1(self.features(x), self.up2[4]).float()

    def invalidate_plan(self):
        """Drop the cached launch lists (called whenever parameters may have changed)."""
        self._plans = {}
        self.__dict__.pop("_stamp_mods", None)

    def train(self, mode=True):
        self.invalidate_plan()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.invalidate_plan()
        out = super()._apply(fn, *a, **k)
        dev = self.conv1.weight.device
        if dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            warm_s2_tables(dev)  # index tensors of the stride-2 weight gradients: never built inside a capture
        return out

    def load_state_dict(self, *a, **k):
        self.invalidate_plan()
        return super().load_state_dict(*a, **k)

    def _plan_stamp(self):
        """(in-place version, storage pointer) of EVERY tensor a recorded plan holds a packed / folded copy
        of: all conv weights (the 1x1 downsamples included), every BatchNorm's affine pair and running
        statistics, the head's weight and bias.  Compared on every replay (~25 us of host time), so an
        in-place edit, an optimizer step, a re-assigned Parameter or a `load_state_dict` on ANY ancestor
        module (which never calls this module's own `load_state_dict`) can not be served from a stale plan."""
        mods = self.__dict__.get("_stamp_mods")
        if mods is None:
            mods = self.__dict__["_stamp_mods"] = [m for m in self.modules()
                                                   if isinstance(m, (nn.Conv2d, nn.BatchNorm2d))]
        out = []
        for m in mods:
            for t in m._parameters.values():
                if t is not None:
                    out.append(t._version)
                    out.append(t.data_ptr())
            for t in m._buffers.values():
                if t is not None:
                    out.append(t._version)
        return tuple(out)

    def forward_nhwc(self, x, dt):
        """HIP path: x (B,X,Y,inC) channels-last activations in dt -> (B,outC,X,Y) fp32 NCHW.
        bf16: the 18 launches are recorded once per input shape and replayed with one native
        call (same kernels, same order; ~150 us less host time per step)."""
        with ops.region("bevencode"):
            if dt != ops.DT_BF16 or ops._recorder is not None or os.environ.get("LSS_NO_PLAN"):
                return self._forward_nhwc(x, dt)
            key = (tuple(x.shape), x.device)
            ent = self._plans.get(key)
            if ent is not None and ent[1] != self._plan_stamp():
                ent = None  # parameters were modified since the plan was built (checked on every call)
            if ent is None:
                rec = ops.ConvRecorder()
                ops.set_recorder(rec)
                try:
                    out = self._forward_nhwc(x, dt)
                finally:
                    ops.set_recorder(None)
                self._plans[key] = (ops.ConvPlan(rec, x, out), self._plan_stamp())
                self._guard("BevEncode (launch plan recorded)")
                return out.clone()  # the recorded output buffer stays with the plan
            out = torch.empty(x.shape[0], self.up2[4].out_channels, x.shape[1], x.shape[2],
                              dtype=torch.float32, device=x.device)
            ent[0].run(x, out)
            every = ops.guard_every()
            if every:
                n = self.__dict__["_guard_calls"] = self.__dict__.get("_guard_calls", 0) + 1
                if n % every == 0:
                    self._guard("BevEncode (call %d)" % n)
            return out

    @staticmethod
    def _guard(where):
        """The ring kernels' flag waits are bounded; a hit leaves garbage in the output (ops.assert_no_timeouts).
        Checked whenever a plan is (re-)recorded and every LSS_GUARD_EVERY-th replay; a device synchronisation each
        time, so never inside a stream capture."""
        if x_is_capturing():
            return
        ops.assert_no_timeouts(where)

    def _forward_nhwc(self, x, dt):
        x = self._stem.run(x, dt, relu=True)
        for blk in self.layer1:
            x = blk._nhwc(x, dt)
        x1 = x
        for blk in self.layer2:
            x = blk._nhwc(x, dt)
        for blk in self.layer3:
            x = blk._nhwc(x, dt)
        x = self.up1._nhwc(x, x1, dt)
        if dt == ops.DT_BF16 and self.up2[1].out_channels == 128 and self.up2[4].out_channels <= 64:
            # up2: x2 upsample + 3x3 conv + BN + ReLU + 1x1 head in ONE launch, NCHW fp32 out
            w, scale, shift = self._up2a.get(dt)
            head = self.up2[4]
            if self._up2a.ring_case(x, dt, up=2, head_n=head.out_channels):
                w = self._up2a.ring(dt)
            hw = head.weight.detach().float().reshape(head.out_channels, -1).contiguous()
            return ops.conv3x3_head_nchw(x, w, scale, shift, hw, head.bias.detach().float().contiguous(), up=2)
        x = self._up2a.run(x, dt, relu=True, up=2)
        return ops.nhwc_to_nchw(self._up2b.run(x, dt, relu=False), dt)

    def forward(self, x):
        """(B, inC, X, Y) -> (B, outC, X, Y) fp32 NCHW, like the reference."""
        if _needs_autograd(self, x):
            return self._forward_autograd(x.float() if x.dtype != torch.float32 else x)
        dt = _PRECISIONS[self.precision or default_precision()]
        return self.forward_nhwc(_to_nhwc(x, dt), dt)
