"""K8k, the K-split one-pass kernel of the launch-bound 3x3 / stride-1 layers (csrc/conv_ks.hip: resnet18 layer1-3 of
BevEncode, ref src/modules.py:104-106, 123-125 + torchvision BasicBlock): against torch's CPU conv on the same
bf16-rounded operands, against the tile kernel it replaces, at the benchmark shapes and at ragged ones (pixel blocks
that end inside an image row, a last block shorter than the others, odd widths), with and without the residual /
ReLU / folded BatchNorm of the epilogue."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_OUT_TOL = 6e-3   # tests/test_kernels_gpu.py: output rounded once to bf16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from lss2_multimodal_nu_amd import ops as _ops
    return _ops


def _q(t):
    return t.bfloat16().float()


SHAPES = [
    # B, H, W, Cin, Cout, residual, relu, folded BN
    (4, 25, 25, 256, 256, True, True, True),     # layer3 conv2 at the benched batch (80-pixel blocks, 18 k-steps per wave)
    (4, 25, 25, 256, 256, False, True, True),    # layer3 conv1
    (4, 50, 50, 128, 128, True, True, True),     # layer2 (160-pixel blocks)
    (4, 100, 100, 64, 64, True, True, True),     # layer1 (320-pixel blocks, two K parts x two pixel halves)
    (4, 100, 100, 64, 64, False, False, False),  # plain conv: no BatchNorm, no activation
    (4, 27, 29, 256, 64, True, True, True),      # ragged: 783 pixels = 9 blocks of 80 + one of 63, odd width, Cout != Cin
    (2, 50, 44, 128, 128, False, True, False),   # 2200 pixels: last block of 120
    (1, 100, 100, 64, 64, True, False, True),    # batch 1: 64 workgroups, the smallest grid the kernel takes
    (5, 41, 83, 64, 96, True, True, True),       # odd everything; 3 channel blocks
    (2, 96, 21, 256, 128, True, True, True),     # a narrow image: an 80-pixel block spans five rows (4 W + 1 = 85)
]


@pytest.mark.parametrize("cfg", SHAPES)
def test_ks_conv_vs_torch_and_tile_kernel(ops, report, cfg):
    B, H, W, Cin, Cout, res, relu, bn = cfg
    assert ops.conv_ks_ok(B, H, W, Cin, Cout), "test shape must be a case for the K-split kernel"
    gen = torch.Generator().manual_seed(sum(int(c) for c in cfg))
    x = _q(torch.randn(B, Cin, H, W, generator=gen))
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=gen) * (Cin * 9) ** -0.5)
    r = _q(torch.randn(B, Cout, H, W, generator=gen)) if res else None
    scale = (torch.rand(Cout, generator=gen) + 0.5) if bn else None
    shift = (torch.randn(Cout, generator=gen) * 0.1) if bn else None
    ref = torch.nn.functional.conv2d(x, w, None, padding=1)
    if bn:
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if res:
        ref = ref + r
    if relu:
        ref = ref.relu()
    xg = ops.nchw_to_nhwc(x.cuda(), 1)
    rg = ops.nchw_to_nhwc(r.cuda(), 1) if res else None
    sg, hg = (scale.cuda(), shift.cuda()) if bn else (None, None)
    wk = ops.pack_conv_weight_ks(w.cuda())
    wt = ops.pack_conv_weight(w.cuda(), 1)
    y = ops.conv2d_nhwc(xg, wk, (3, 3), 1, 1, sg, hg, rg, relu, None, 1, None, 1)
    y0 = ops.conv2d_nhwc(xg, wt, (3, 3), 1, 1, sg, hg, rg, relu, None, 1, None, 1)
    y2 = ops.conv2d_nhwc(xg, wk, (3, 3), 1, 1, sg, hg, rg, relu, None, 1, None, 1)
    out, old, out2 = ops.nhwc_to_nchw(y, 1).cpu(), ops.nhwc_to_nchw(y0, 1).cpu(), ops.nhwc_to_nchw(y2, 1).cpu()
    assert out.shape == ref.shape
    tag = "x".join(str(int(c)) for c in cfg)
    assert report("k8k_max_rel_" + tag, (out - ref).abs().max() / ref.abs().max()) <= BF16_OUT_TOL
    assert report("k8k_rel_l2_" + tag, (out - ref).norm() / ref.norm()) <= BF16_OUT_TOL / 3
    # the tile kernel on the same operands: the same fp32 products in another order (here: four K parts summed at the
    # end), one bf16 ulp apart where the two sums round to different neighbours
    assert report("k8k_vs_tile_" + tag, (out - old).abs().max() / ref.abs().max()) <= 8e-3
    assert torch.equal(out, out2)  # fixed summation order: bit-reproducible


def test_ks_weights_are_rejected_where_the_kernel_has_no_case(ops):
    """LSS_W_KS on a shape the kernel does not take is an argument error, not a silent fallback."""
    wk = ops.pack_conv_weight_ks(torch.randn(64, 64, 3, 3).cuda())
    x = torch.randn(1, 8, 8, 64, device="cuda").bfloat16()   # one pixel block: far too few workgroups
    assert not ops.conv_ks_ok(1, 8, 8, 64, 64)
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(x, wk, (3, 3), 1, 1, None, None, None, True, None, 1, None, 1)
    with pytest.raises(ValueError):
        ops.pack_conv_weight_ks(torch.randn(64, 96, 3, 3).cuda())   # Cin must be 64, 128 or 256
    assert not ops.conv_ks_ok(4, 400, 400, 64, 64)                  # 2000 workgroups: the tile kernel's territory
    assert not ops.conv_ks_ok(4, 100, 60, 64, 64)                   # a 320-pixel block would span six rows of 60
    assert not ops.conv_ks_ok(4, 40, 40, 256, 256)                  # 7 x 42 positions x 256 channels: more than the patch LDS


@pytest.mark.parametrize("cfg", [(4, 25, 25, 256, 256), (4, 50, 50, 128, 128), (3, 50, 44, 64, 128), (4, 100, 100, 64, 64)])
def test_ks_input_gradient_conv_vs_autograd(ops, report, cfg):
    """The dgrad image of the weights (lss_conv2d_pack_weights_ks_dgrad: transposed, taps flipped) run through the same
    kernel = d/dx of conv2d(x, w, padding=1), against torch's CPU autograd on bf16-rounded operands."""
    B, H, W, Cin, Cout = cfg
    assert ops.conv_ks_ok(B, H, W, Cout, Cin), "the gradient conv (Cout -> Cin) must be a case for the kernel"
    gen = torch.Generator().manual_seed(sum(cfg))
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=gen) * (Cin * 9) ** -0.5)
    dz = _q(torch.randn(B, Cout, H, W, generator=gen))
    x = torch.zeros(B, Cin, H, W, requires_grad=True)
    torch.nn.functional.conv2d(x, w, None, padding=1).backward(dz)
    ref = x.grad
    wd = ops.pack_conv_weight_ks(w.cuda(), dgrad=True)
    assert (wd.Cout, wd.Cin) == (Cin, Cout)
    dx = ops.conv2d_nhwc(ops.nchw_to_nhwc(dz.cuda(), 1), wd, (3, 3), 1, 1, None, None, None, False, None, 1, None, 1)
    out = ops.nhwc_to_nchw(dx, 1).cpu()
    tag = "x".join(str(c) for c in cfg)
    assert report("k8k_dgrad_max_rel_" + tag, (out - ref).abs().max() / ref.abs().max()) <= BF16_OUT_TOL
    assert report("k8k_dgrad_rel_l2_" + tag, (out - ref).norm() / ref.norm()) <= BF16_OUT_TOL / 3
