#!/usr/bin/env python3
"""Generate tests/golden/g10_*.npz / g11_*.npz by RUNNING THE REFERENCE'S OWN
vovnet-model classes (src/model_vovnet_transformer.py, src/transformer_modules.py)
on CPU.  Build container only (needs /root/reference); the fixtures are data
(inputs + the reference's outputs).  Weights are NOT stored: both this script and
the tests regenerate them from `oracle.vovnet_oracle.seeded_state` (numpy
RandomState, platform independent) and load them by state_dict key.

    python tools/gen_golden_vovnet.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_loader import load_reference  # noqa: E402
from oracle import vovnet_oracle as vo  # noqa: E402  (parameter/input generators only)
from oracle.lss_oracle import synthetic_rig  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
GRID_COARSE = dict(xbound=[-50.0, 50.0, 2.0], ybound=[-50.0, 50.0, 2.0],
                   zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-36s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def randn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape).astype(np.float32))


def load_seeded(module, shapes, seed, tweak=None):
    sd = vo.seeded_state(shapes, seed)
    if tweak:
        tweak(sd)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


def main():
    rtools, rmodules, rmodel = load_reference()
    import src.model_vovnet_transformer as rv
    import src.transformer_modules as rt

    torch.set_grad_enabled(False)
    # --- G10a StandardDepthNet (reduced C3 width: the ctor argument of the reference)
    m = rv.StandardDepthNet(c3_channels=64, depth_channels=41)
    load_seeded(m, vo.standard_depthnet_shapes(64, 41), 10)
    c3 = randn(100, 2, 64, 4, 6)
    save("g10_standard_depthnet", c3=c3.numpy(), depth=m(c3).numpy(), seed=10)

    # --- G10b MultiScaleDepthNet: exact x2 and a non-integer upsample ratio
    m = rv.MultiScaleDepthNet(c3_channels=64, c4_channels=128, depth_channels=41)
    load_seeded(m, vo.multiscale_depthnet_shapes(64, 128, 41), 11)
    c3a, c4a = randn(101, 2, 64, 4, 6), randn(102, 2, 128, 2, 3)
    c3b, c4b = randn(103, 1, 64, 5, 7), randn(104, 1, 128, 3, 4)
    save("g10_multiscale_depthnet", c3a=c3a.numpy(), c4a=c4a.numpy(), depth_a=m(c3a, c4a).numpy(),
         c3b=c3b.numpy(), c4b=c4b.numpy(), depth_b=m(c3b, c4b).numpy(), seed=11)

    # --- G10c CamEncodeV2
    ce = rv.CamEncodeV2(D=41, C_in=64, C_out=8)
    load_seeded(ce, vo.camencode_v2_shapes(64, 8), 12)
    feats = randn(105, 2, 64, 4, 6)
    depth = torch.softmax(randn(106, 2, 41, 4, 6), dim=1)
    save("g10_camencode_v2", features=feats.numpy(), depth=depth.numpy(), cam_feats=ce(feats, depth).numpy(), seed=12)

    # --- G10d BEV branch of VoVNetBEVTransformer up to voxel_pooling (v1 and v2 heads).
    # The model is assembled WITHOUT its __init__ (which builds the timm trunk = a
    # by-name network fetch); the methods run are the reference's own.
    conf = dict(final_dim=(64, 96), Ncams=2, cams=["A", "B"])
    for ver, seed in (("v2", 13), ("v1", 14)):
        s = rv.VoVNetBEVTransformer.__new__(rv.VoVNetBEVTransformer)
        nn.Module.__init__(s)
        s.bsize, s.grid_conf, s.data_aug_conf = 1, GRID_COARSE, conf
        dx, bx, nx = rtools.gen_dx_bx(GRID_COARSE["xbound"], GRID_COARSE["ybound"], GRID_COARSE["zbound"])
        s.dx = nn.Parameter(dx, requires_grad=False)
        s.bx = nn.Parameter(bx, requires_grad=False)
        s.nx = nn.Parameter(nx, requires_grad=False)
        s.downsample, s.D, s.C = 16, 41, 128
        s.frustum = s.create_frustum()
        if ver == "v2":
            s.depth_net = rv.MultiScaleDepthNet(64, 128, 41)
            load_seeded(s.depth_net, vo.multiscale_depthnet_shapes(64, 128, 41), seed)
        else:
            s.depth_net = rv.StandardDepthNet(64, 41)
            load_seeded(s.depth_net, vo.standard_depthnet_shapes(64, 41), seed)
        s.cam_encode = rv.CamEncodeV2(D=41, C_in=64, C_out=128)
        load_seeded(s.cam_encode, vo.camencode_v2_shapes(64, 128), seed + 100)
        s.use_quickcumsum = True
        c3, c4 = randn(107, 2, 64, 4, 6), randn(108, 2, 128, 2, 3)
        rots, trans, intrins, post_rots, post_trans = synthetic_rig(1, 2, final_dim=(64, 96), train_aug=True, seed=3)
        # the lines of VoVNetBEVTransformer.forward :586-602
        depth = s.depth_net(c3, c4)
        cam = s.cam_encode(c3, depth)
        _, C, D, Hf, Wf = cam.shape
        cam = cam.view(1, 2, C, D, Hf, Wf).permute(0, 1, 3, 4, 5, 2)
        geom = s.get_geometry(rots, trans, intrins, post_rots, post_trans)
        bev = s.voxel_pooling(geom, cam)
        save("g10_vovnet_liftsplat_" + ver, c3=c3.numpy(), c4=c4.numpy(), rots=rots.numpy(), trans=trans.numpy(),
             intrins=intrins.numpy(), post_rots=post_rots.numpy(), post_trans=post_trans.numpy(),
             depth=depth.numpy(), bev=bev.numpy(), frustum=s.frustum.detach().numpy(), seed=seed)

    # --- G11a positional encoding
    pe = rt.PositionEmbeddingSine(128, normalize=True)
    save("g11_pos_embed", pos=pe(torch.zeros(1, 256, 10, 14))[0].numpy())

    # --- G11b DeformableAttention with offsets large enough to leave the grid
    da = rt.DeformableAttention(d_model=256, n_heads=8, n_points=8)
    shapes = [(k[len("encoder.self_attn."):], v) for k, v in vo.transformer_shapes() if "self_attn" in k]

    def wide(sd):
        sd["sampling_offsets.bias"] *= 30.0

    load_seeded(da, shapes, 15, wide)
    H = W = 12
    q, v = randn(109, 1, H * W, 256), randn(110, 1, H * W, 256)
    ref_pts = vo.reference_points(H, W)[None]
    save("g11_deform_attn", query=q.numpy(), value=v.numpy(), out=da(q, v, ref_pts).numpy(), seed=15, bias_scale=30.0)

    # --- G11c LightweightBEVTransformer, G11d BEVEncoderTransformer
    tr = rt.LightweightBEVTransformer(d_model=256, n_heads=8, dim_feedforward=1024, dropout=0.1)

    def wide_t(sd):
        sd["encoder.self_attn.sampling_offsets.bias"] *= 30.0

    load_seeded(tr, vo.transformer_shapes(), 16, wide_t)
    x = randn(111, 1, 256, 12, 12)
    save("g11_bev_transformer", x=x.numpy(), out=tr(x).numpy(), seed=16, bias_scale=30.0)

    be = rv.BEVEncoderTransformer(in_channels=128, out_channels=4)

    def wide_b(sd):
        sd["transformer.encoder.self_attn.sampling_offsets.bias"] *= 30.0

    load_seeded(be, vo.bev_encoder_transformer_shapes(128, 4), 17, wide_b)
    x = randn(112, 2, 128, 12, 12)
    seg, refined = be(x)
    save("g11_bev_encoder_transformer", x=x.numpy(), seg=seg.numpy(), refined=refined.numpy(), seed=17, bias_scale=30.0)


if __name__ == "__main__":
    main()
