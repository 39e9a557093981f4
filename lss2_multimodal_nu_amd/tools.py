"""Host-side mirror of the hot-path helpers of the reference's `src/tools.py`
(`gen_dx_bx` :172-178, `cumsum_trick` :181-189, `QuickCumsum` :192-218).

`gen_dx_bx` is init-time host arithmetic.  `cumsum_trick` / `QuickCumsum` keep
the reference's call signature for callers that still drive the splat through
them; the sums run in the HIP segmented-sum kernel (direct per-run sums, not a
cumsum followed by differences - ~1000x less rounding noise).
The fused `LSS.forward` path does not go through them at all.
"""
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
