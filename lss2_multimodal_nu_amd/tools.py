"""Host-side mirror of the hot-path helpers of the reference's `src/tools.py`
(`gen_dx_bx` :172-178, `cumsum_trick` :181-189, `QuickCumsum` :192-218).

`gen_dx_bx` is init-time host arithmetic.  `cumsum_trick` / `QuickCumsum` keep
the reference's call signature for callers that still drive the splat through
them; the sums run in the HIP segmented-sum kernel (direct per-run sums, not a
cumsum followed by differences - ~1000x less rounding noise).
The fused `LSS.forward` path does not go through them at all.
"""
import os

import torch

from . import ops


def gen_dx_bx(xbound, ybound, zbound):
    """Voxel size `dx`, first voxel centre `bx` (fp32) and voxel counts `nx`
    (int64) of a grid given as three [lo, hi, step] triples."""
    bounds = (xbound, ybound, zbound)
    dx = torch.tensor([b[2] for b in bounds], dtype=torch.float32)
    bx = torch.tensor([b[0] + b[2] / 2.0 for b in bounds], dtype=torch.float32)
    # float division then truncation, as LongTensor(list of floats) does
    nx = torch.tensor([int((b[1] - b[0]) / b[2]) for b in bounds], dtype=torch.int64)
    return dx, bx, nx


def _runs(ranks):
    """Boundaries of the equal-rank runs of a sorted rank vector: `last` marks the
    final row of every run, `seg_start` (M+1 int32) the row offsets."""
    K = ranks.shape[0]
    last = torch.ones(K, device=ranks.device, dtype=torch.bool)
    if K > 1:
        last[:-1] = ranks[1:] != ranks[:-1]
    ends = torch.nonzero(last).flatten()
    seg_start = torch.cat([ends.new_zeros(1), ends + 1]).to(torch.int32)
    return last, seg_start


def _segmented(x, seg_start):
    if not x.is_cuda:
        raise RuntimeError("cumsum_trick / QuickCumsum run on the GPU (HIP segmented-sum kernel); "
                           "got a %s tensor" % x.device)
    return ops.segmented_sum(x.contiguous().float(), seg_start)


def cumsum_trick(x, geom_feats, ranks):
    """Per-voxel sums of rows pre-sorted by `ranks`; returns (sums, geom of each run)."""
    last, seg_start = _runs(ranks)
    return _segmented(x, seg_start), geom_feats[last]


class QuickCumsum(torch.autograd.Function):
    """Same contract as the reference's autograd.Function: forward = per-run sums,
    backward = every row receives its run's gradient."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        last, seg_start = _runs(ranks)
        y = _segmented(x, seg_start)
        geom_kept = geom_feats[last]
        ctx.save_for_backward(last)
        ctx.mark_non_differentiable(geom_kept)
        return y, geom_kept

    @staticmethod
    def backward(ctx, gradx, gradgeom):
        last, = ctx.saved_tensors
        run_of_row = torch.cumsum(last, 0) - last.to(torch.int64)
        return gradx[run_of_row], None, None


class _WeightedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight):
        x = logits.float().contiguous()
        t = target.contiguous()
        w = weight.float().contiguous()
        loss, sums = ops.weighted_ce_fwd(x, t, w)
        ctx.save_for_backward(x, t, w, sums)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, w, sums = ctx.saved_tensors
        gx = ops.weighted_ce_bwd(x, t, w, sums, g)
        return gx.to(ctx.in_dtype), None, None


def weighted_cross_entropy(ypred, ytgt, weight):
    """nn.CrossEntropyLoss(weight=weight)(ypred, ytgt) for (B,C,H,W) logits / (B,H,W) int64 targets:
    one HIP pass forward, one backward on the GPU (csrc/loss.hip); torch's own op elsewhere."""
    if ypred.is_cuda and ypred.dim() >= 3 and ypred.shape[1] <= 16 and ytgt.dtype == torch.int64:
        return _WeightedCEFn.apply(ypred, ytgt, weight)
    return torch.nn.functional.cross_entropy(ypred, ytgt, weight=weight)


class _HeadCEFn(torch.autograd.Function):
    """loss = CrossEntropy(weight)(conv1x1(y; W, b), target) with the head and the loss in ONE HIP pass each way
    (csrc/loss.hip: lss_head_ce_fwd / _bwd).  y: logical (B, Cin, H, W) tensor whose memory is NHWC bf16 (what the
    conv + BatchNorm + ReLU training unit hands over); W (K, Cin, 1, 1), b (K)."""

    @staticmethod
    def forward(ctx, y, weight, bias, target, class_w):
        yn = y.permute(0, 2, 3, 1)
        if yn.dtype != torch.bfloat16:
            yn = yn.to(torch.bfloat16)
        yn = yn.contiguous()
        w2 = weight.detach().float().reshape(weight.shape[0], -1).contiguous()
        b1 = bias.detach().float().contiguous()
        cw = class_w.detach().float().contiguous()
        t = target.contiguous()
        loss, sums = ops.head_ce_fwd(yn, w2, b1, t, cw)
        ctx.save_for_backward(yn, w2, b1, t, cw, sums)
        ctx.meta = (y.dtype, weight.shape, weight.dtype, bias.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        yn, w2, b1, t, cw, sums = ctx.saved_tensors
        ydt, wshape, wdt, bdt = ctx.meta
        dy, dw, db = ops.head_ce_bwd(yn, w2, b1, t, cw, sums, g)
        dy = dy.permute(0, 3, 1, 2)
        return (dy if dy.dtype == ydt else dy.to(ydt)), dw.view(wshape).to(wdt), db.to(bdt), None, None


class _Head1x1Fn(torch.autograd.Function):
    """`head(y)` for the 1x1 head `nn.Conv2d(128, K, 1)` (ref src/modules.py:115) on the HIP kernels, both ways: the
    logits of a training-mode `model(x)` whose loss the caller computes (ref train.py:52, 62).  y: logical
    (B, 128, H, W) tensor whose memory is NHWC bf16; returns (B, K, H, W) fp32.  Replaces torch's convolution there
    because the library's backward is not safe inside a HIP graph: round 3's data-parallel graph step went wrong in
    exactly this op."""

    @staticmethod
    def forward(ctx, y, weight, bias):
        yn = y.permute(0, 2, 3, 1)
        if yn.dtype != torch.bfloat16:
            yn = yn.to(torch.bfloat16)
        yn = yn.contiguous()
        w2 = weight.detach().float().reshape(weight.shape[0], -1).contiguous()
        b1 = bias.detach().float().contiguous()
        out = ops.head1x1_fwd(yn, w2, b1)
        ctx.save_for_backward(yn, w2, b1)
        ctx.meta = (y.dtype, weight.shape, weight.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        yn, w2, b1 = ctx.saved_tensors
        ydt, wshape, wdt, bdt = ctx.meta
        dy, dw, db = ops.head1x1_bwd(yn, w2, b1, g.float().contiguous())
        dy = dy.permute(0, 3, 1, 2)
        return (dy if dy.dtype == ydt else dy.to(ydt)), dw.view(wshape).to(wdt), db.to(bdt)


def head_1x1(y, head):
    """head(y) on the autograd path: the HIP head kernels when the shapes fit (128 input channels, 4 or 8 classes, a
    bf16 activation on the GPU), torch's convolution otherwise (fp32 parity mode, or LSS_TRAIN_NATIVE=0 - the
    library comparison leg of the tests; NOT safe inside a HIP graph)."""
    if (os.environ.get("LSS_TRAIN_NATIVE", "1") != "0" and y.is_cuda and y.dtype == torch.bfloat16 and y.dim() == 4 and y.shape[1] == 128 and head.kernel_size == (1, 1)
            and head.stride == (1, 1) and head.bias is not None and head.out_channels in (4, 8)):
        return _Head1x1Fn.apply(y, head.weight, head.bias)
    return head(y)


def head_weighted_cross_entropy(y, head, ytgt, weight):
    """nn.CrossEntropyLoss(weight=weight)(head(y), ytgt) for a 1x1 `head` = nn.Conv2d(128, K, 1) applied to the
    (B, 128, H, W) activation y: fused into one HIP kernel per direction on the GPU when the shapes
    fit (128 input channels, 4 or 8 classes) AND y is already bf16 (autocast / bf16 precision: the kernel reads y
    as bf16 and returns dy rounded to bf16, which for a bf16 y loses nothing).  An fp32 y (parity mode) takes the two
    separate ops, so that `forward_loss` equals `forward` + `SimpleLoss` to fp32 accuracy, loss and gradients."""
    if (y.is_cuda and y.dtype == torch.bfloat16 and y.dim() == 4 and y.shape[1] == 128 and head.kernel_size == (1, 1)
            and head.bias is not None
            and head.out_channels in (4, 8) and ytgt.dtype == torch.int64
            and tuple(ytgt.shape) == (y.shape[0], y.shape[2], y.shape[3])):
        return _HeadCEFn.apply(y, head.weight, head.bias, ytgt, weight)
    return weighted_cross_entropy(head(y).float(), ytgt, weight)


class SimpleLoss(torch.nn.Module):
    """ref: src/tools.py:221-231 - weighted 4-class BEV cross-entropy, class weights [1, 10, 5, 10]."""

    def __init__(self, class_weights=(1.0, 10.0, 5.0, 10.0)):
        super().__init__()
        self.register_buffer("weight", torch.tensor(class_weights, dtype=torch.float32), persistent=False)

    def forward(self, ypred, ytgt):
        return weighted_cross_entropy(ypred, ytgt, self.weight.to(ypred.device))


def MultiLoss(bev_pre, act_pre, desc_pre, bev_gt, act_gt, desc_gt, args=None):
    """ref: src/tools.py:234-252 - BEV cross-entropy [1,10,5,10] + weighted BCE-with-logits on the
    action [1,5,5,5] and description [1,5,5,5,1,1,1,1] heads (those two are 4- and 8-element rows:
    library ops)."""
    dev = bev_pre.device
    F = torch.nn.functional
    loss_bev = weighted_cross_entropy(bev_pre, bev_gt, torch.tensor([1.0, 10.0, 5.0, 10.0], device=dev))
    w1 = torch.tensor([1.0, 5.0, 5.0, 5.0], device=dev)
    w2 = torch.tensor([1.0, 5.0, 5.0, 5.0, 1.0, 1.0, 1.0, 1.0], device=dev)
    loss_act = F.binary_cross_entropy_with_logits(act_pre.to(dev), act_gt, weight=w1)
    loss_desc = F.binary_cross_entropy_with_logits(desc_pre.to(dev), desc_gt, weight=w2)
    return loss_bev + loss_act + loss_desc
