"""Gradients of the 3x3/s1/p1 conv through the C ABI (lss_conv2d_pack_weights_dgrad +
lss_conv2d_fwd, lss_conv2d_wgrad) against torch's own gradient formulas evaluated on the CPU in
fp32 from the same bf16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lss2_multimodal_nu_amd import ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout
    (2, 12, 20, 64, 64),     # one tile column, 64-wide tiles mostly padding
    (1, 25, 25, 256, 256),   # layer3-like, odd size
    (2, 50, 50, 128, 128),   # layer2-like
    (1, 9, 7, 64, 192),      # Cin != Cout, tiny image (more guard than image)
    (4, 100, 100, 64, 64),   # layer1 at the bench batch
]


def _operands(B, H, W, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g).bfloat16()
    dy = torch.randn(B, H, W, Cout, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    return x, dy, w


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_wgrad_vs_torch(shape):
    B, H, W, Cin, Cout = shape
    x, dy, w = _operands(*shape, seed=sum(shape))
    dw = ops.conv3x3_wgrad(x.cuda(), dy.cuda())
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, 3, 3),
                                      dy.float().permute(0, 3, 1, 2), padding=1)
    assert dw.shape == ref.shape and dw.dtype == torch.float32
    err = float((dw.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-4, err
    # fixed summation order: bitwise reproducible, also with a dirty workspace from another call
    ops.conv3x3_wgrad(dy.cuda()[..., :Cin].contiguous() if Cout >= Cin else x.cuda(), dy.cuda())
    assert torch.equal(dw, ops.conv3x3_wgrad(x.cuda(), dy.cuda()))


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_conv3x3_dgrad_vs_torch(shape):
    B, H, W, Cin, Cout = shape
    x, dy, w = _operands(*shape, seed=1 + sum(shape))
    wb = w.bfloat16().float()
    wd = ops.pack_conv_weight_dgrad(wb.cuda(), ops.DT_BF16)
    assert wd.shape == (9, Cin, Cout)
    dx = ops.conv2d_nhwc(dy.cuda(), wd, (3, 3), 1, 1, None, None, None, False, out_f32=True)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), wb, dy.float().permute(0, 3, 1, 2), padding=1)
    err = float((dx.cpu().permute(0, 3, 1, 2) - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-4, err


def test_wgrad_argument_checks():
    x = torch.zeros(1, 4, 4, 64).bfloat16().cuda()
    with pytest.raises(ValueError):
        ops.conv3x3_wgrad(x.float(), x)
    with pytest.raises(ValueError):
        ops.conv3x3_wgrad(x, torch.zeros(1, 4, 5, 64).bfloat16().cuda())
