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


# ---- BEV_TXT / only-BEV variants against the reference's own forward (fixture g13, tools/gen_golden_heads.py) ----
def _heads_fixture(golden, tag):
    from oracle import vovnet_oracle as vo  # seeded parameter generator shared with the fixture script
    g = golden("g13_bevtxt_heads")
    keys = [str(k) for k in g[tag + "_keys"]]
    shapes = [tuple(int(d) for d in str(s).split(",")) if str(s) else () for s in g[tag + "_shapes"]]
    dtypes = [str(d) for d in g[tag + "_dtypes"]]
    head = [(k, s) for k, s in zip(keys, shapes) if k.split(".")[0] not in ("dx", "bx", "nx", "frustum", "camencode")]
    return g, keys, shapes, dtypes, vo.seeded_state(head, int(g[tag + "_seed"]))


@pytest.mark.parametrize("tag", ["txt", "onlybev"])
def test_bevtxt_variants_state_dict_layout_is_the_references(golden, tag):
    """Keys, ORDER, shapes and dtypes of every entry outside `encoder.*` / `bevencode.*` equal the list the
    reference's own modules produced (the reference registers bevencode between camencode and bevpost)."""
    g, keys, shapes, dtypes, _ = _heads_fixture(golden, tag)
    make = L.compile_model_bevtxt if tag == "txt" else L.compile_model_onlybev
    m = make(2, GRID, AUG, 4)
    sd = m.state_dict()
    mine = [(k, tuple(v.shape), str(v.dtype)) for k, v in sd.items() if not k.startswith(("encoder.", "bevencode."))]
    assert mine == list(zip(keys, shapes, dtypes))
    order = [k.split(".")[0] for k in sd]
    assert order.index("camencode") < order.index("bevencode") < order.index("bevpost")  # ref __init__ order


@pytest.mark.parametrize("tag", ["txt", "onlybev"])
def test_bevtxt_variants_txt_half_equals_reference_forward(golden, tag):
    """The TXT half (crop [:, :, 60:140, 56:144], BevPost, camera selection, embedders, predictors, concat
    order) against the outputs of the reference's forward code on the same seeded weights and inputs; the
    BEV map is the fixture's (the camera->BEV half needs the GPU and is tested in test_modules_gpu.py)."""
    g, _, _, _, state = _heads_fixture(golden, tag)
    make = L.compile_model_bevtxt if tag == "txt" else L.compile_model_onlybev
    m = make(int(g["B"]), GRID, AUG, 4).eval()
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected and all(k.split(".")[0] in ("dx", "bx", "nx", "frustum", "camencode", "bevencode")
                                  for k in missing)
    rs = np.random.RandomState
    x = torch.from_numpy(rs(int(g["seed_x"])).randn(int(g["B"]) * 6, 512, 8, 22).astype(np.float32))
    bev = torch.from_numpy(rs(int(g["seed_bev"])).randn(int(g["B"]), 4, 200, 200).astype(np.float32))
    m._bev = lambda *a: bev
    with torch.no_grad():
        bev_o, act, desc = m(x, None, None, None, None, None)
    assert bev_o is bev
    np.testing.assert_allclose(act.numpy(), g[tag + "_act"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(desc.numpy(), g[tag + "_desc"], rtol=1e-4, atol=1e-5)


def test_onlybev_heads_backpropagate_into_the_bev_map():
    """ref src/model_baseline.py:283: the only-BEV variant does NOT detach the crop (model_BEV_TXT.py:285 does)."""
    for make, expect in ((L.compile_model_onlybev, True), (L.compile_model_bevtxt, False)):
        m = make(1, GRID, AUG, 4).eval()
        bev = torch.randn(1, 4, 200, 200, requires_grad=True)
        m._bev = lambda *a: bev
        _, act, desc = m(torch.randn(6, 512, 8, 22), None, None, None, None, None)
        (act.sum() + desc.sum()).backward()
        got = bev.grad is not None and float(bev.grad.abs().sum()) > 0
        assert got == expect
        if expect:  # only the crop receives gradient
            mask = torch.ones_like(bev, dtype=torch.bool)
            mask[:, :, 60:140, 56:144] = False
            assert float(bev.grad[mask].abs().sum()) == 0.0


def test_strict_load_of_a_reference_checkpoint_with_trunk_entries(lss):
    """predict.py:37,78 load with strict=True; a reference checkpoint also carries `encoder.*` entries."""
    sd = dict(lss.state_dict())
    sd["encoder.trunk._conv_stem.weight"] = torch.zeros(48, 3, 3, 3)
    sd["encoder.up1.conv.0.weight"] = torch.zeros(512, 608, 3, 3)
    m = L.compile_model_lss(2, GRID, AUG, 4)
    with pytest.warns(UserWarning, match="Encoder"):
        m.load_state_dict(sd, strict=True)
    # with the Encoder mirror in the slot, up1.* loads strictly and only the (absent) trunk's entries are dropped
    m2 = L.compile_model_lss(2, GRID, AUG, 4, encoder=L.Encoder())
    sd2 = dict(m2.state_dict())
    sd2["encoder.trunk._conv_stem.weight"] = torch.zeros(48, 3, 3, 3)
    with pytest.warns(UserWarning, match="trunk"):
        m2.load_state_dict(sd2, strict=True)
    del sd2["encoder.up1.conv.0.weight"]
    with pytest.raises(RuntimeError), pytest.warns(UserWarning):
        m2.load_state_dict(sd2, strict=True)


def test_plan_stamp_sees_every_folded_tensor(lss):
    """the launch-plan stamp must change for an in-place edit of ANY conv / BatchNorm tensor,
    the 1x1 downsamples included, and for a load_state_dict on the PARENT model."""
    be = lss.bevencode
    base = be._plan_stamp()
    with torch.no_grad():
        for name, t in list(be.named_parameters()) + list(be.named_buffers()):
            if name.endswith("num_batches_tracked"):
                continue
            before = be._plan_stamp()
            t.add_(0)
            assert be._plan_stamp() != before, name
    assert be._plan_stamp() != base
    before = be._plan_stamp()
    lss.load_state_dict({k: v.clone() for k, v in lss.state_dict().items()})
    assert be._plan_stamp() != before


def test_eval_mode_call_with_grad_enabled_warns(lss):
    from lss2_multimodal_nu_amd import modules as M
    M._warned_eval_autograd.discard("BevEncode")
    lss.bevencode.eval()
    with pytest.warns(UserWarning, match="no_grad"), torch.enable_grad():
        lss.bevencode(torch.randn(1, 64, 16, 16))


def test_batch_counters_are_collected_and_flushed_once():
    """modules._count_batch: inside BevEncode.features the BatchNorm counters are collected and bumped by one foreach
    call at the end; outside (stand-alone units) each is bumped at once; a failing forward leaves no collector behind."""
    import torch

    from lss2_multimodal_nu_amd import modules as M
    bns = [torch.nn.BatchNorm2d(8) for _ in range(3)]
    M._count_batch(bns[0])
    assert int(bns[0].num_batches_tracked) == 1
    outer, M._tls.counters = getattr(M._tls, "counters", None), []
    try:
        for bn in bns:
            M._count_batch(bn)
        assert [int(b.num_batches_tracked) for b in bns] == [1, 0, 0]   # nothing bumped yet
        # the collector belongs to THIS thread: another thread's BatchNorm is bumped at once
        import threading
        other = torch.nn.BatchNorm2d(8)
        th = threading.Thread(target=M._count_batch, args=(other,))
        th.start(); th.join()
        assert int(other.num_batches_tracked) == 1 and len(M._tls.counters) == 3
        torch._foreach_add_(M._tls.counters, 1)
    finally:
        M._tls.counters = outer
    assert [int(b.num_batches_tracked) for b in bns] == [2, 1, 1]
    be = M.BevEncode(64, 4)
    be.train()
    try:
        be.features(torch.zeros(1, 3, 8, 8))   # wrong channel count: raises inside the collecting region
    except Exception:
        pass
    assert getattr(M._tls, "counters", None) is None


def test_conv_s2_backward_gemm_equals_autograd():
    """modules.conv_s2_backward_gemm (im2col + GEMM + col2im) against torch's own autograd for the three stride-2
    conv shapes of BevEncode (7x7 / pad 3, 3x3 / pad 1, 1x1 / pad 0), fp32 on the CPU."""
    import torch

    from lss2_multimodal_nu_amd.modules import conv_s2_backward_gemm
    g = torch.Generator().manual_seed(3)
    for K, pad, C, Co, H, W in ((7, 3, 8, 6, 12, 16), (3, 1, 8, 16, 10, 14), (1, 0, 8, 16, 10, 14)):
        x = torch.randn(2, C, H, W, generator=g, requires_grad=True)
        w = torch.randn(Co, C, K, K, generator=g, requires_grad=True)
        y = torch.nn.functional.conv2d(x, w, None, stride=2, padding=pad)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        gx, gw = conv_s2_backward_gemm(gy, x.detach(), w.detach(), pad)
        assert torch.allclose(gx, x.grad, rtol=1e-4, atol=1e-4), K
        assert torch.allclose(gw, w.grad, rtol=1e-4, atol=1e-4), K


def test_s2_wgrad_gather_equals_advanced_indexing():
    """modules._s2_wgrad_gather (one index_select over a cached flat index) against the advanced-indexing form it
    replaced, for the 3x3 / 2 convs (3 taps per axis) and the 7x7 / 2 stem (4 taps)."""
    import torch
    from lss2_multimodal_nu_amd import modules as M
    for K, T in ((3, 3), (7, 4)):
        Co, C = 8, 6
        g6 = torch.randn(Co, 2, 2, C, T, T)
        ph = torch.tensor([(k - K // 2) % 2 for k in range(K)])
        ds = [(k - K // 2) // 2 for k in range(K)]
        tp = torch.tensor([d - min(ds) for d in ds])
        ref = g6[:, ph[:, None], ph[None, :], :, tp[:, None], tp[None, :]].permute(2, 3, 0, 1).contiguous()
        assert torch.equal(M._s2_wgrad_gather(g6, K), ref)
