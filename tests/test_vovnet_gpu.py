"""vovnet-model half of the hot path on the GPU: depth heads,
CamEncodeV2 and the C=128 lift-splat through the HIP kernels, against the
fixtures the reference's own classes produced and against the CPU oracle at
full size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lss2_multimodal_nu_amd as L  # noqa: E402
from lss2_multimodal_nu_amd import model_vovnet_transformer as mv  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402
from oracle import vovnet_oracle as vo  # noqa: E402

GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
            dbound=[4.0, 45.0, 1.0])
GRID_COARSE = dict(GRID, xbound=[-50.0, 50.0, 2.0], ybound=[-50.0, 50.0, 2.0])


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30), np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def load(module, shapes, seed):
    module.load_state_dict(vo.seeded_state(shapes, seed), strict=True)
    return module.cuda().eval()


# tolerance per conv-path precision: fp32 = parity mode (f32 MFMA), bf16 = config-2 style math
TOL = {"fp32": 2e-5, "bf16": 2e-2}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_standard_depthnet_native(golden, prec):
    g = golden("g10_standard_depthnet")
    m = load(mv.StandardDepthNet(64, 41, precision=prec), vo.standard_depthnet_shapes(64, 41), int(g["seed"]))
    with torch.no_grad():
        depth = m(t(g["c3"]).cuda())
    assert depth.shape == g["depth"].shape
    assert rel(depth, g["depth"])[0] <= TOL[prec]
    assert float((depth.sum(1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_multiscale_depthnet_native(golden, prec):
    g = golden("g10_multiscale_depthnet")
    m = load(mv.MultiScaleDepthNet(64, 128, 41, precision=prec), vo.multiscale_depthnet_shapes(64, 128, 41),
             int(g["seed"]))
    with torch.no_grad():
        da = m(t(g["c3a"]).cuda(), t(g["c4a"]).cuda())
        db = m(t(g["c3b"]).cuda(), t(g["c4b"]).cuda())  # non-integer upsample ratio, odd sizes
    assert rel(da, g["depth_a"])[0] <= TOL[prec]
    assert rel(db, g["depth_b"])[0] <= TOL[prec]


def test_camencode_v2_native(golden):
    g = golden("g10_camencode_v2")
    m = load(mv.CamEncodeV2(41, 64, 8), vo.camencode_v2_shapes(64, 8), int(g["seed"]))
    with torch.no_grad():
        out = m(t(g["features"]).cuda(), t(g["depth"]).cuda())
    assert rel(out, g["cam_feats"])[0] <= 1e-5


class _Trunk(mv.TrunkC3C4):
    def __init__(self, c3, c4):
        super().__init__()
        self.c3_channels, self.c4_channels = c3, c4


def small_model(ver, seed, prec):
    conf = dict(final_dim=(64, 96), Ncams=2, cams=["A", "B"])
    m = L.compile_model_vovnet_transformer(1, GRID_COARSE, conf, 4, lss_version=ver, backbone=_Trunk(64, 128),
                                           precision=prec)
    m.depth_net.load_state_dict(vo.seeded_state(vo.multiscale_depthnet_shapes(64, 128, 41) if ver == "v2"
                                                else vo.standard_depthnet_shapes(64, 41), seed))
    m.cam_encode.load_state_dict(vo.seeded_state(vo.camencode_v2_shapes(64, 128), seed + 100))
    return m.cuda().eval()


@pytest.mark.parametrize("ver", ["v1", "v2"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_vovnet_get_voxels_vs_reference(golden, ver, prec):
    g = golden("g10_vovnet_liftsplat_" + ver)
    m = small_model(ver, int(g["seed"]), prec)
    assert np.array_equal(m.frustum.cpu().numpy(), g["frustum"])
    calib = [t(g[k]) for k in ("rots", "trans", "intrins", "post_rots", "post_trans")]
    with torch.no_grad():
        bev = m.get_voxels(t(g["c3"]).cuda(), t(g["c4"]).cuda(), *calib)
        bev_cl = m.get_voxels(t(g["c3"]).cuda(), t(g["c4"]).cuda(), *calib, layout=L.ops.BEV_NHWC_F32)
    assert bev.shape == (1, 128, 50, 50) and bev.is_contiguous()
    assert torch.equal(bev, bev_cl.contiguous())
    # the occupied cells are an index property: exact in either precision
    assert np.array_equal(bev.cpu().numpy() != 0, g["bev"] != 0)
    mx, l2 = rel(bev, g["bev"])
    assert l2 <= (1e-3 if prec == "fp32" else 2e-2) and mx <= (1e-3 if prec == "fp32" else 4e-2), (mx, l2)


def test_vovnet_training_path_matches_and_differentiates(golden):
    g = golden("g10_vovnet_liftsplat_v2")
    m = small_model("v2", int(g["seed"]), "fp32")
    calib = [t(g[k]) for k in ("rots", "trans", "intrins", "post_rots", "post_trans")]
    c3 = t(g["c3"]).cuda().requires_grad_(True)
    bev = m.get_voxels(c3, t(g["c4"]).cuda(), *calib)  # eval-mode BN, autograd on: library heads + K5/K7
    assert rel(bev, g["bev"])[1] <= 1e-3
    bev.square().sum().backward()
    assert c3.grad is not None and float(c3.grad.abs().sum()) > 0
    assert m.cam_encode.feat_proj.weight.grad is not None


@pytest.mark.parametrize("ver,prec", [("v2", "fp32"), ("v2", "bf16"), ("v1", "bf16")])
def test_vovnet_full_size_vs_oracle(ver, prec):
    """BASELINE config-4 shapes (768/1024-channel trunk maps, 6 cameras 8x22, C=128, 200x200)."""
    B = 2
    conf = dict(final_dim=(128, 352), Ncams=6, cams=list("abcdef"))
    torch.manual_seed(1)
    m = L.compile_model_vovnet_transformer(B, GRID, conf, 4, lss_version=ver, precision=prec)
    dshapes = vo.multiscale_depthnet_shapes() if ver == "v2" else vo.standard_depthnet_shapes()
    dsd, csd = vo.seeded_state(dshapes, 21), vo.seeded_state(vo.camencode_v2_shapes(), 22)
    m.depth_net.load_state_dict(dsd)
    m.cam_encode.load_state_dict(csd)
    m = m.cuda().eval()
    gen = np.random.RandomState(5)
    c3 = t(gen.randn(B * 6, 768, 8, 22).astype(np.float32))
    c4 = t(gen.randn(B * 6, 1024, 4, 11).astype(np.float32))
    calib = lo.synthetic_rig(B, 6, train_aug=True, seed=7)
    with torch.no_grad():
        bev = m.get_voxels(c3.cuda(), c4.cuda(), *calib)
    dx, bx, nx = lo.gen_dx_bx(GRID["xbound"], GRID["ybound"], GRID["zbound"])
    ref, _ = vo.vovnet_lift_splat(c3, c4, dsd, csd, ver, m.frustum.cpu(), *calib, dx, bx, nx, B)
    assert bev.shape == (B, 128, 200, 200)
    # SURVEY 8 a-7 parity rule: norm-wise AND element-wise relative to max|ref| (the reference's
    # cumsum trick itself carries ~1e-4 of cancellation noise, so cells it rounds to 0 may hold ~1e-5 here)
    mx, l2 = rel(bev, ref)
    assert l2 <= (1e-3 if prec == "fp32" else 2e-2) and mx <= (1e-3 if prec == "fp32" else 4e-2), (mx, l2)
    # a cell no frustum point falls into is exactly zero
    vox = lo.voxel_indices_np(lo.get_geometry_torch(m.frustum.cpu(), *calib).numpy(), dx.numpy(), bx.numpy(), nx.numpy())
    occ = np.zeros((B, 200, 200), dtype=bool)
    idx, kept = vox
    bidx = np.broadcast_to(np.arange(B).reshape(B, 1, 1, 1, 1), kept.shape)
    occ[bidx[kept], idx[..., 0][kept], idx[..., 1][kept]] = True
    assert float(bev.cpu().numpy()[~np.broadcast_to(occ[:, None], bev.shape)].__abs__().max()) == 0.0
