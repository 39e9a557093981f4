"""oracle/vovnet_oracle.py against the fixtures the reference's own classes
produced (tools/gen_golden_vovnet.py): depth heads, CamEncodeV2, the vovnet BEV
branch up to voxel_pooling, and the BEV transformer."""
import os

import numpy as np
import pytest
import torch

from oracle import lss_oracle as lo
from oracle import vovnet_oracle as vo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
GRID_COARSE = dict(xbound=[-50.0, 50.0, 2.0], ybound=[-50.0, 50.0, 2.0],
                   zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])


def gold(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= tol, err


def test_standard_depthnet():
    g = gold("g10_standard_depthnet")
    sd = vo.seeded_state(vo.standard_depthnet_shapes(64, 41), int(g["seed"]))
    close(vo.standard_depthnet(t(g["c3"]), sd), g["depth"], 1e-5)


def test_multiscale_depthnet_both_ratios():
    g = gold("g10_multiscale_depthnet")
    sd = vo.seeded_state(vo.multiscale_depthnet_shapes(64, 128, 41), int(g["seed"]))
    close(vo.multiscale_depthnet(t(g["c3a"]), t(g["c4a"]), sd), g["depth_a"], 1e-5)
    close(vo.multiscale_depthnet(t(g["c3b"]), t(g["c4b"]), sd), g["depth_b"], 1e-5)


def test_upsample_half_pixel_is_interpolate():
    x = torch.randn(2, 3, 3, 5)
    for size in ((6, 10), (5, 7), (3, 5), (9, 6)):
        ref = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
        close(vo.upsample_bilinear_half_pixel(x, size), ref, 1e-6)


def test_camencode_v2():
    g = gold("g10_camencode_v2")
    sd = vo.seeded_state(vo.camencode_v2_shapes(64, 8), int(g["seed"]))
    close(vo.camencode_v2(t(g["features"]), t(g["depth"]), sd), g["cam_feats"], 1e-6)


@pytest.mark.parametrize("ver", ["v1", "v2"])
def test_vovnet_lift_splat(ver):
    g = gold("g10_vovnet_liftsplat_" + ver)
    seed = int(g["seed"])
    dsd = vo.seeded_state(vo.multiscale_depthnet_shapes(64, 128, 41) if ver == "v2"
                          else vo.standard_depthnet_shapes(64, 41), seed)
    csd = vo.seeded_state(vo.camencode_v2_shapes(64, 128), seed + 100)
    dx, bx, nx = lo.gen_dx_bx(GRID_COARSE["xbound"], GRID_COARSE["ybound"], GRID_COARSE["zbound"])
    frustum = lo.create_frustum((64, 96), 16, GRID_COARSE["dbound"])
    assert np.array_equal(frustum.numpy(), g["frustum"])
    bev, depth = vo.vovnet_lift_splat(t(g["c3"]), t(g["c4"]), dsd, csd, ver, frustum, t(g["rots"]), t(g["trans"]),
                                      t(g["intrins"]), t(g["post_rots"]), t(g["post_trans"]), dx, bx, nx, 1)
    close(depth, g["depth"], 1e-5)
    assert bev.shape == (1, 128, 50, 50)
    # same occupied cells, values within accumulated fp32 rounding of the depth head
    assert np.array_equal(bev.numpy() != 0, g["bev"] != 0)
    close(bev, g["bev"], 1e-4)


def test_pos_embed():
    g = gold("g11_pos_embed")
    close(vo.position_embedding_sine(10, 14, 128), g["pos"], 1e-6)


def _wide(sd, key, scale):
    sd[key] = sd[key] * float(scale)
    return sd


def test_deformable_attention():
    g = gold("g11_deform_attn")
    shapes = [(k[len("encoder.self_attn."):], v) for k, v in vo.transformer_shapes() if "self_attn" in k]
    sd = _wide(vo.seeded_state(shapes, int(g["seed"])), "sampling_offsets.bias", g["bias_scale"])
    out = vo.deformable_attention(t(g["query"]), t(g["value"]), vo.reference_points(12, 12), sd, "")
    close(out, g["out"], 1e-5)


def test_bev_transformer():
    g = gold("g11_bev_transformer")
    sd = _wide(vo.seeded_state(vo.transformer_shapes(), int(g["seed"])),
               "encoder.self_attn.sampling_offsets.bias", g["bias_scale"])
    close(vo.lightweight_bev_transformer(t(g["x"]), sd), g["out"], 1e-5)


def test_bev_encoder_transformer():
    g = gold("g11_bev_encoder_transformer")
    sd = _wide(vo.seeded_state(vo.bev_encoder_transformer_shapes(128, 4), int(g["seed"])),
               "transformer.encoder.self_attn.sampling_offsets.bias", g["bias_scale"])
    seg, refined = vo.bev_encoder_transformer(t(g["x"]), sd)
    close(refined, g["refined"], 1e-5)
    close(seg, g["seg"], 1e-5)
