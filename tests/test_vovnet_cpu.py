"""Host-side mirror of src/model_vovnet_transformer.py / src/transformer_modules.py
without a GPU: state_dict layout, and the library (autograd) path of every module
against the fixtures the reference's own classes produced."""
import numpy as np
import pytest
import torch

import lss2_multimodal_nu_amd as L
from lss2_multimodal_nu_amd import model_vovnet_transformer as mv
from lss2_multimodal_nu_amd import transformer_modules as tm
from oracle import vovnet_oracle as vo

GRID_COARSE = dict(xbound=[-50.0, 50.0, 2.0], ybound=[-50.0, 50.0, 2.0],
                   zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    a = a.detach().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= tol, err


def load(module, shapes, seed, wide=None, scale=1.0):
    sd = vo.seeded_state(shapes, seed)
    if wide:
        sd[wide] = sd[wide] * float(scale)
    assert [(k, tuple(v.shape)) for k, v in module.state_dict().items()] == [(k, tuple(s)) for k, s in shapes]
    module.load_state_dict(sd, strict=True)
    return module.eval()


def test_standard_depthnet_library_path(golden):
    g = golden("g10_standard_depthnet")
    m = load(mv.StandardDepthNet(64, 41), vo.standard_depthnet_shapes(64, 41), int(g["seed"]))
    close(m(t(g["c3"])), g["depth"], 1e-5)


def test_multiscale_depthnet_library_path(golden):
    g = golden("g10_multiscale_depthnet")
    m = load(mv.MultiScaleDepthNet(64, 128, 41), vo.multiscale_depthnet_shapes(64, 128, 41), int(g["seed"]))
    close(m(t(g["c3a"]), t(g["c4a"])), g["depth_a"], 1e-5)
    close(m(t(g["c3b"]), t(g["c4b"])), g["depth_b"], 1e-5)


def test_camencode_v2_library_path(golden):
    g = golden("g10_camencode_v2")
    m = load(mv.CamEncodeV2(41, 64, 8), vo.camencode_v2_shapes(64, 8), int(g["seed"]))
    close(m(t(g["features"]), t(g["depth"])), g["cam_feats"], 1e-6)


def test_pos_embed(golden):
    g = golden("g11_pos_embed")
    pe = tm.PositionEmbeddingSine(128, normalize=True)
    close(pe(torch.zeros(3, 256, 10, 14))[1], g["pos"], 1e-6)
    with pytest.raises(ValueError):
        tm.PositionEmbeddingSine(128, normalize=False, scale=1.0)


def test_deformable_attention_default_init():
    da = tm.DeformableAttention(256, 8, 8)
    b = da.sampling_offsets.bias.detach().view(8, 8, 2)
    # head 0 looks along +x, point p at distance p+1; head 2 along +y
    assert torch.allclose(b[0, :, 0], torch.arange(1.0, 9.0)) and float(b[0, :, 1].abs().max()) < 1e-6
    assert torch.allclose(b[2, :, 1], torch.arange(1.0, 9.0)) and float(b[2, :, 0].abs().max()) < 1e-5
    assert float(da.sampling_offsets.weight.abs().sum()) == 0.0
    assert float(da.attention_weights.weight.abs().sum()) == 0.0 and float(da.attention_weights.bias.abs().sum()) == 0.0


def test_deformable_attention_library_path(golden):
    g = golden("g11_deform_attn")
    shapes = [(k[len("encoder.self_attn."):], v) for k, v in vo.transformer_shapes() if "self_attn" in k]
    da = load(tm.DeformableAttention(256, 8, 8), shapes, int(g["seed"]), "sampling_offsets.bias", g["bias_scale"])
    ref = vo.reference_points(12, 12)[None]
    close(da(t(g["query"]), t(g["value"]), ref), g["out"], 1e-5)


def test_bev_transformer_library_path(golden):
    g = golden("g11_bev_transformer")
    m = load(tm.LightweightBEVTransformer(256, 8, 1024, 0.1), vo.transformer_shapes(), int(g["seed"]),
             "encoder.self_attn.sampling_offsets.bias", g["bias_scale"])
    close(m(t(g["x"])), g["out"], 1e-5)


def test_bev_encoder_transformer_library_path(golden):
    g = golden("g11_bev_encoder_transformer")
    m = load(mv.BEVEncoderTransformer(128, 4), vo.bev_encoder_transformer_shapes(128, 4), int(g["seed"]),
             "transformer.encoder.self_attn.sampling_offsets.bias", g["bias_scale"])
    seg, refined = m(t(g["x"]))
    close(refined, g["refined"], 1e-5)
    close(seg, g["seg"], 1e-5)


def test_model_layout_and_errors():
    conf = dict(final_dim=(128, 352), Ncams=6, cams=["a", "b", "c", "d", "e", "f"])
    grid = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
                dbound=[4.0, 45.0, 1.0])
    m = L.compile_model_vovnet_transformer(2, grid, conf, 4, lss_version="v2")
    sd = m.state_dict()
    assert list(sd)[:4] == ["dx", "bx", "nx", "frustum"] and tuple(sd["frustum"].shape) == (41, 8, 22, 3)
    for pre, shapes in (("depth_net.", vo.multiscale_depthnet_shapes()), ("cam_encode.", vo.camencode_v2_shapes()),
                        ("bev_encoder.", vo.bev_encoder_transformer_shapes(128, 4))):
        got = [(k[len(pre):], tuple(v.shape)) for k, v in sd.items() if k.startswith(pre)]
        assert got == [(k, tuple(s)) for k, s in shapes]
    assert tuple(sd["camera_ids"].shape) == (6,)
    tops = {k.split(".")[0] for k in sd}
    assert {"feature_pyramid", "sceneunder", "camera_transformer", "bev_fusion", "unified_predictor"} <= tops
    v1 = L.compile_model_vovnet_transformer(2, grid, conf, 4, lss_version="V1", use_camera_attn=False,
                                            use_cross_attn=False)
    assert isinstance(v1.depth_net, mv.StandardDepthNet) and v1.camera_transformer is None and v1.bev_fusion is None
    with pytest.raises(ValueError):
        L.compile_model_vovnet_transformer(2, grid, conf, 4, lss_version="v3")
    with pytest.raises(RuntimeError):  # raw images, no trunk bundled
        m(torch.zeros(12, 3, 128, 352), *[torch.zeros(2, 6, 3, 3)] * 1, torch.zeros(2, 6, 3),
          torch.zeros(2, 6, 3, 3), torch.zeros(2, 6, 3, 3), torch.zeros(2, 6, 3))
