"""Host-side mirror of the reference interface: state_dict layout, init-time
tensors, and the (autograd / library) training path - all without a GPU."""
import numpy as np
import pytest
import torch

import lss2_multimodal_nu_amd as L
from oracle import bev_oracle as bo
from oracle import lss_oracle as lo

GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
            dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


@pytest.fixture(scope="module")
def lss():
    torch.manual_seed(0)
    return L.compile_model_lss(2, GRID, AUG, 4)


def test_state_dict_layout_matches_reference(lss):
    sd = lss.state_dict()
    head = [(k, tuple(v.shape), v.dtype) for k, v in sd.items() if not k.startswith("bevencode.")]
    assert head == [("dx", (3,), torch.float32), ("bx", (3,), torch.float32), ("nx", (3,), torch.int64),
                    ("frustum", (41, 8, 22, 3), torch.float32),
                    ("camencode.depthnet.weight", (105, 512, 1, 1), torch.float32),
                    ("camencode.depthnet.bias", (105,), torch.float32)]
    got = [(k[len("bevencode."):], tuple(v.shape)) for k, v in sd.items() if k.startswith("bevencode.")]
    assert got == [(k, tuple(s)) for k, s in bo.bev_encode_state_shapes(64, 4)]
    assert not any(p.requires_grad for p in (lss.dx, lss.bx, lss.nx, lss.frustum))
    n_hot = sum(p.numel() for n, p in lss.named_parameters() if n.split(".")[0] in ("camencode", "bevencode"))
    assert n_hot == 53865 + sum(int(np.prod(s)) for k, s in bo.bev_encode_state_shapes(64, 4)
                                if "running" not in k and "tracked" not in k)


def test_init_tensors_match_reference_goldens(lss, golden):
    g1, g2 = golden("g1_gen_dx_bx"), golden("g2_frustum")
    assert np.array_equal(lss.dx.numpy(), g1["default_dx"]) and np.array_equal(lss.bx.numpy(), g1["default_bx"])
    assert np.array_equal(lss.nx.numpy(), g1["default_nx"])
    assert np.array_equal(lss.frustum.numpy(), g2["default"])
    for tag in ("hires", "coarse", "small_z2"):
        dx, bx, nx = L.gen_dx_bx(*g1[tag + "_bounds"].tolist())
        assert np.array_equal(dx.numpy(), g1[tag + "_dx"]) and np.array_equal(bx.numpy(), g1[tag + "_bx"])
        assert np.array_equal(nx.numpy(), g1[tag + "_nx"])


def test_strict_state_dict_round_trip(lss):
    other = L.compile_model_lss(2, GRID, AUG, 4)
    other.load_state_dict(lss.state_dict(), strict=True)
    for (ka, va), (kb, vb) in zip(lss.state_dict().items(), other.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_resnet18_style_init(lss):
    be = lss.bevencode
    for blk in list(be.layer1) + list(be.layer2) + list(be.layer3):
        assert float(blk.bn2.weight.detach().abs().sum()) == 0.0  # zero_init_residual=True
        assert float(blk.bn1.weight.detach().min()) == 1.0
    assert be.layer2[0].downsample is not None and be.layer1[0].downsample is None


def test_bevencode_library_path_equals_oracle(lss):
    """Training-mode / autograd path (torch ops) against the functional oracle."""
    be = lss.bevencode
    torch.manual_seed(1)
    with torch.no_grad():
        for blk in list(be.layer1) + list(be.layer2) + list(be.layer3):
            blk.bn2.weight.uniform_(0.5, 1.5)
    x = torch.randn(2, 64, 40, 40)
    be.eval()
    sd = {k: v.clone() for k, v in be.state_dict().items()}
    ref = bo.bev_encode(x, sd, training=False)
    with torch.enable_grad():
        out = be(x)
    np.testing.assert_allclose(out.detach().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    be.train()
    stats = {}
    ref_t = bo.bev_encode(x, sd, training=True, stats_out=stats)
    out_t = be(x)
    np.testing.assert_allclose(out_t.detach().numpy(), ref_t.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(be.bn1.running_mean.numpy(), stats["bn1.running_mean"].numpy(), rtol=1e-5, atol=1e-6)
    be.eval()


@pytest.mark.parametrize("name", ["g9_up_x2_eval", "g9_up_x4_train"])
def test_up_container_loads_reference_state(golden, name):
    g = golden(name)
    up = L.Up(12, 6, scale_factor=int(g["scale"]))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_") and not k.startswith("sd_after_")}
    up.load_state_dict(sd, strict=True)
    up.train(bool(g["training"]))
    with torch.enable_grad():
        y = up(torch.from_numpy(g["x1"]), torch.from_numpy(g["x2"]))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-5)


def test_host_calibration_matrices_equal_reference_expressions(lss):
    """The flattened single-call form must give bitwise the matrices of the reference's
    `torch.inverse(post_rots)` and `rots.matmul(torch.inverse(intrins))` (ref :60,:66)."""
    for seed, aug in ((0, False), (1, True), (2, True)):
        r, t, i, pr, pt = lo.synthetic_rig(3, train_aug=aug, seed=seed)
        inv_pr, comb = lss._calib_matrices(r, i, pr)
        assert torch.equal(inv_pr, torch.inverse(pr))
        assert torch.equal(comb, r.matmul(torch.inverse(i)))
    g = torch.Generator().manual_seed(5)
    r, i, pr = (torch.randn(2, 6, 3, 3, generator=g) for _ in range(3))
    inv_pr, comb = lss._calib_matrices(r, i, pr)
    assert torch.equal(inv_pr, torch.inverse(pr)) and torch.equal(comb, r.matmul(torch.inverse(i)))


def test_trunk_slot_rejects_raw_images(lss):
    with pytest.raises(RuntimeError):
        lss.encoder(torch.zeros(2, 6, 3, 128, 352))
    assert lss.encoder(torch.zeros(2, 6, 512, 8, 22)).shape == (12, 512, 8, 22)


def test_product_path_has_no_cpu_fallback(lss):
    """On a box without a GPU the inference path must raise, not silently compute."""
    lss.eval()
    r, t, i, pr, pt = lo.synthetic_rig(2)
    with torch.no_grad(), pytest.raises(Exception):
        lss(torch.zeros(12, 512, 8, 22), r, t, i, pr, pt)


def test_bev_txt_constructs_with_reference_heads():
    m = L.compile_model_bevtxt(2, GRID, AUG, 4)
    keys = set(k.split(".")[0] for k in m.state_dict())
    assert {"sceneunder", "embeder_f1", "embeder_f2", "embeder_lr1", "embeder_lr2", "predictorf1", "predictorf2",
            "predictorlr", "bevpost", "camencode", "bevencode", "frustum"} <= keys
