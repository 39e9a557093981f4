#!/usr/bin/env python3
"""Generate tests/golden/g13_bevtxt_heads.npz by RUNNING THE REFERENCE'S OWN `BEV_TXT.forward`
code (src/model_BEV_TXT.py:278-334 and src/model_baseline.py:274-290) on CPU.

Build container only (needs /root/reference).  The two `BEV_TXT` classes cannot be constructed
through their `__init__` here (`Encoder()` = EfficientNet by-name fetch, `BevEncode` = torchvision
resnet18, both absent offline - SURVEY.md 8c), so each model is assembled WITHOUT `__init__`:
the heads are the reference's own classes (`BevPost`, `SceneUnder`, `Embedder_*`, `Predictor`),
`encoder` is the identity on trunk features and `get_voxels` / `bevencode` hand back a FIXED BEV
map.  What runs - the crop, `BevPost`, the camera selection, the embedders, the predictors and the
concatenation order - is the reference's forward, line for line.

The fixture stores: the (key, shape) list of every non-trunk, non-BevEncode `state_dict` entry in
the reference's order (the layout pin for the product's modules), and act / desc outputs.  Weights
and inputs are regenerated on both sides from numpy RandomState seeds (`oracle.vovnet_oracle.seeded_state`).

    python tools/gen_golden_heads.py
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
from oracle import vovnet_oracle as vo  # noqa: E402  (parameter generator only)

OUT = os.path.join(ROOT, "tests", "golden")
GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
            dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


def randn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape).astype(np.float32))


def shell(cls, rtools, rmodules, heads):
    s = cls.__new__(cls)
    nn.Module.__init__(s)
    s.grid_conf, s.data_aug_conf, s.bsize = GRID, AUG, 2
    dx, bx, nx = rtools.gen_dx_bx(GRID["xbound"], GRID["ybound"], GRID["zbound"])
    s.dx = nn.Parameter(dx, requires_grad=False)
    s.bx = nn.Parameter(bx, requires_grad=False)
    s.nx = nn.Parameter(nx, requires_grad=False)
    s.downsample, s.camC = 16, 64
    s.frustum = s.create_frustum()
    s.D = s.frustum.shape[0]
    s.encoder = nn.Identity()
    for name, mod in heads:
        setattr(s, name, mod)
    s.camencode = rmodules.CamEncode(s.D, s.camC, s.downsample)
    return s


def main():
    rtools, rmodules, rmodel = load_reference()
    import src.model_baseline as rbase
    torch.set_grad_enabled(False)
    B, N = 2, 6
    x = randn(201, B * N, 512, 8, 22)
    bev = randn(202, B, 4, 200, 200)
    out = {"seed_x": 201, "seed_bev": 202, "B": B}

    R = rmodules
    variants = (
        ("txt", rmodel.BEV_TXT, 21, [
            ("sceneunder", R.SceneUnder()),
            ("embeder_f1", R.Embedder_f1(in_channels=256, out_channels=32)),
            ("embeder_f2", R.Embedder_f2(out_channels=40)),
            ("embeder_lr1", R.Embedder_lr1(in_channels=256, out_channels=32)),
            ("embeder_lr2", R.Embedder_lr2(out_channels=40)),
            ("predictorf1", R.Predictor(num_in=40, classes=4)),
            ("predictorf2", R.Predictor(num_in=40, classes=4)),
            ("predictorlr", R.Predictor(num_in=40, classes=1))]),
        ("onlybev", rbase.BEV_TXT, 22, [
            ("sceneunder", R.SceneUnder()),
            ("embeder_bev", R.Embedder_f2(out_channels=8)),
            ("predictor_bev1", R.Predictor(num_in=8, classes=4)),
            ("predictor_bev2", R.Predictor(num_in=8, classes=8))]),
    )
    for tag, cls, seed, heads in variants:
        s = shell(cls, rtools, rmodules, heads)
        s.bevpost = R.BevPost()  # constructed after camencode / bevencode in the reference's __init__
        s.eval()
        sd = s.state_dict()
        keys = [k for k in sd if not k.startswith("encoder.")]
        shapes = [(k, tuple(sd[k].shape)) for k in keys]
        head_shapes = [(k, shp) for k, shp in shapes if k.split(".")[0] not in ("dx", "bx", "nx", "frustum", "camencode")]
        s.load_state_dict(vo.seeded_state(head_shapes, seed), strict=False)
        s.get_voxels = lambda *a: None
        s.bevencode = lambda y2: bev
        bev_o, act, desc = s.forward(x, None, None, None, None, None)
        assert bev_o is bev
        out[tag + "_keys"] = np.array(keys)
        out[tag + "_shapes"] = np.array([",".join(str(d) for d in shp) for _, shp in shapes])
        out[tag + "_dtypes"] = np.array([str(sd[k].dtype) for k in keys])
        out[tag + "_seed"] = seed
        out[tag + "_act"] = act.numpy()
        out[tag + "_desc"] = desc.numpy()
        print(tag, len(keys), "state_dict entries; act", tuple(act.shape), "desc", tuple(desc.shape))
    path = os.path.join(OUT, "g13_bevtxt_heads.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
