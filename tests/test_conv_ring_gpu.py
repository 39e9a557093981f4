"""K8r, the loader / consumer ring kernel of the big 3x3 convs (csrc/conv_ring.hip): against torch's CPU conv on
the same bf16-rounded operands (ref: src/modules.py:9-27 `Up`, :108-116 `up2`), against the tile kernel it replaces on
those layers, at the benchmark shapes and at ragged ones (image sizes that are not multiples of the 4 x 20 strip, an
odd strip count), with and without the fused upsample / concat gather and the fused 1x1 head."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bev_oracle as bo  # noqa: E402

BF16_OUT_TOL = 6e-3   # tests/test_kernels_gpu.py: output rounded once to bf16
BF16_UP_TOL = 1.2e-2  # fused bilinear upsample: the interpolated operand is rounded to bf16 as well


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from lss2_multimodal_nu_amd import ops as _ops
    return _ops


def _q(t):
    return t.bfloat16().float()


SHAPES = [
    # B, H, W, Cx, Cout, C2, up, head_n, relu
    (4, 100, 100, 256, 256, 0, 1, 0, True),    # BevEncode.up1.conv[3] at batch 4
    (4, 25, 25, 256, 256, 64, 4, 0, True),     # BevEncode.up1.conv[0]: cat([x1, up4(x3)])
    (4, 100, 100, 256, 128, 0, 2, 4, True),    # BevEncode.up2 + head
    (3, 54, 67, 64, 128, 0, 2, 0, False),      # 108 x 134 output: ragged columns, odd strip count, no ReLU
    (4, 91, 113, 64, 128, 0, 1, 3, True),      # ragged rows and columns, 3-class head on the plain conv
    (2, 30, 35, 128, 256, 32, 4, 0, True),     # 32-channel skip tensor: one full-resolution chunk + four upsampled
    (12, 48, 80, 32, 128, 0, 1, 0, True),      # one 32-channel chunk (a single patch buffer generation), many images
    (1, 200, 200, 64, 128, 0, 1, 0, False),    # batch 1, two chunks, 500 workgroups
    (10, 13, 27, 96, 128, 0, 4, 0, True),      # 52 x 108 output from a x4 upsample of an odd-sized source, three chunks
    (10, 7, 100, 64, 128, 0, 2, 2, True),      # 14 x 200: a wide, flat image (rows ragged, columns exact), 2-class head
]


@pytest.mark.parametrize("cfg", SHAPES)
def test_ring_conv_vs_torch_and_tile_kernel(ops, report, cfg, monkeypatch):
    B, H, W, Cx, Cout, C2, up, head_n, relu = cfg
    assert ops.conv_ring_ok(B, H, W, Cx, C2, up, Cout, head_n), "test shape must be a ring-kernel case"
    gen = torch.Generator().manual_seed(sum(int(c) for c in cfg))
    x = _q(torch.randn(B, Cx, H, W, generator=gen))
    x2 = _q(torch.randn(B, C2, H * up, W * up, generator=gen)) if C2 else None
    w = _q(torch.randn(Cout, Cx + C2, 3, 3, generator=gen) * ((Cx + C2) * 9) ** -0.5)
    scale, shift = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen) * 0.1
    xin = bo.upsample_bilinear_ac(x, up) if up > 1 else x
    if C2:
        xin = torch.cat([x2, xin], 1)
    ref = torch.nn.functional.conv2d(xin, w, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        ref = ref.relu()
    xg = ops.nchw_to_nhwc(x.cuda(), 1)
    x2g = ops.nchw_to_nhwc(x2.cuda(), 1) if C2 else None
    wr = ops.pack_conv_weight_ring(w.cuda())
    wt = ops.pack_conv_weight(w.cuda(), 1)
    tile_ok = (Cx % 64 == 0 and C2 % 64 == 0)  # the tile kernel's K block is 64 channels
    before = ops.N.lib().lss_conv2d_ring_timeouts()
    if head_n:
        hw, hb = torch.randn(head_n, Cout, generator=gen) * Cout ** -0.5, torch.randn(head_n, generator=gen)
        ref = torch.nn.functional.conv2d(ref, hw.view(head_n, Cout, 1, 1), hb)
        out = ops.conv3x3_head_nchw(xg, wr, scale.cuda(), shift.cuda(), hw.cuda(), hb.cuda(), x2=x2g, up=up, relu=relu)
        old = ops.conv3x3_head_nchw(xg, wt, scale.cuda(), shift.cuda(), hw.cuda(), hb.cuda(), x2=x2g, up=up,
                                    relu=relu) if tile_ok else out
        out2 = ops.conv3x3_head_nchw(xg, wr, scale.cuda(), shift.cuda(), hw.cuda(), hb.cuda(), x2=x2g, up=up, relu=relu)
        out, old, out2 = out.cpu(), old.cpu(), out2.cpu()
    else:
        y = ops.conv2d_nhwc(xg, wr, (3, 3), 1, 1, scale.cuda(), shift.cuda(), None, relu, x2g, up, None, 1)
        y0 = ops.conv2d_nhwc(xg, wt, (3, 3), 1, 1, scale.cuda(), shift.cuda(), None, relu, x2g, up, None, 1) if tile_ok else y
        y2 = ops.conv2d_nhwc(xg, wr, (3, 3), 1, 1, scale.cuda(), shift.cuda(), None, relu, x2g, up, None, 1)
        out, old, out2 = ops.nhwc_to_nchw(y, 1).cpu(), ops.nhwc_to_nchw(y0, 1).cpu(), ops.nhwc_to_nchw(y2, 1).cpu()
    assert ops.N.lib().lss_conv2d_ring_timeouts() == before, "a flag wait of the ring kernel hit its bound"
    assert out.shape == ref.shape
    tag = "x".join(str(int(c)) for c in cfg)
    tol = BF16_UP_TOL if up > 1 else BF16_OUT_TOL
    err = report("k8r_max_rel_" + tag, (out - ref).abs().max() / ref.abs().max())
    assert err <= tol
    assert report("k8r_rel_l2_" + tag, (out - ref).norm() / ref.norm()) <= tol / 3
    # the tile kernel on the same operands: the same fp32 products summed in another order, one bf16 ulp apart where
    # the two sums round to different neighbours (heads: fp32 VALU head here, split-bf16 MFMA head there)
    assert report("k8r_vs_tile_" + tag, (out - old).abs().max() / ref.abs().max()) <= 8e-3
    assert torch.equal(out, out2)  # no atomics, fixed summation order: bit-reproducible


def test_ring_weights_are_rejected_where_the_kernel_has_no_case(ops):
    """LSS_W_RING on a shape the ring kernel does not take is an argument error, not a silent fallback."""
    w = torch.randn(128, 64, 3, 3).cuda()
    wr = ops.pack_conv_weight_ring(w)
    x = torch.randn(1, 8, 8, 64, device="cuda").bfloat16()   # 8 strips: far too few workgroups
    assert not ops.conv_ring_ok(1, 8, 8, 64, 0, 1, 128)
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(x, wr, (3, 3), 1, 1, None, None, None, True, None, 1, None, 1)
    with pytest.raises(ValueError):
        ops.pack_conv_weight_ring(torch.randn(96, 64, 3, 3).cuda())


@pytest.mark.parametrize("cfg", [(4, 100, 100, 256, 256), (2, 96, 120, 128, 256), (6, 61, 83, 64, 128)])
def test_ring_dgrad_vs_autograd(ops, report, cfg):
    """Input gradient of a 3x3 / s1 / p1 conv = the ring kernel on dY with the dgrad-packed weight
    (lss_conv2d_pack_weights_ring_dgrad), against torch's CPU autograd on the same bf16-rounded operands
    (the ConvolutionBackward nodes of ref src/modules.py:22-27 under train.py:61)."""
    B, H, W, Cout, Cin = cfg     # forward conv: Cin -> Cout; dgrad: Cout channels in, Cin channels out
    assert ops.conv_ring_ok(B, H, W, Cout, 0, 1, Cin, 0)
    gen = torch.Generator().manual_seed(B + H + W + Cout + Cin)
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=gen) * (Cin * 9) ** -0.5)
    dy = _q(torch.randn(B, Cout, H, W, generator=gen))
    x = torch.zeros(B, Cin, H, W, requires_grad=True)
    torch.nn.functional.conv2d(x, w, None, padding=1).backward(dy)
    ref = x.grad
    wd = ops.pack_conv_weight_ring_dgrad(w.cuda())
    before = ops.N.lib().lss_conv2d_ring_timeouts()
    g = ops.conv2d_nhwc(ops.nchw_to_nhwc(dy.cuda(), 1), wd, (3, 3), 1, 1, None, None, None, False, None, 1, None, 1)
    out = ops.nhwc_to_nchw(g, 1).cpu()
    assert ops.N.lib().lss_conv2d_ring_timeouts() == before
    tag = "x".join(str(c) for c in cfg)
    assert report("k8r_dgrad_max_rel_" + tag, (out - ref).abs().max() / ref.abs().max()) <= BF16_OUT_TOL
    assert report("k8r_dgrad_rel_l2_" + tag, (out - ref).norm() / ref.norm()) <= BF16_OUT_TOL / 3
