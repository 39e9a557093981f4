"""Parity at the BENCHMARKED configuration (BASELINE configs[1]: batch 4, bf16, no_grad, CPU calibration
tensors - exactly `bench.py`'s `step()`): the launch set the driver times (region pipeline at B = 4, the
recorded conv plan whose tile shapes / split-K / grid rounds are chosen from the B = 4 grids) against the oracle.

Reference path: /root/reference/src/model_BEV_TXT.py:128-140 (`get_voxels` -> `bevencode`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lss2_multimodal_nu_amd as L  # noqa: E402
from lss2_multimodal_nu_amd import ops  # noqa: E402
from oracle import bev_oracle as bo  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402

import bench  # noqa: E402  (GRID / AUG of the benched workload; importing runs nothing)

BF16_TOL = 2e-2  # tests/test_modules_gpu.py: 3x the largest rel-L2 observed for the bf16 conv path vs the fp32 oracle


def randomize_bn(m, seed=3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.bias.shape, generator=g) + 0.5)


def _bench_inputs(B=4, rank=0):
    """The tensors bench.py builds (same seeds): trunk features on the GPU, calibration on the CPU."""
    g = torch.Generator().manual_seed(1234 + rank)
    feats = torch.randn(B * 6, 512, 8, 22, generator=g)
    calib = lo.synthetic_rig(B, final_dim=bench.AUG["final_dim"], train_aug=True, seed=rank)
    return feats, calib


@pytest.mark.parametrize("bn", ["init", "random"])
def test_benched_step_b4_bf16_vs_oracle(report, bn):
    B = 4
    torch.manual_seed(0)
    m = L.compile_model_lss(B, bench.GRID, bench.AUG, 4, precision="bf16")
    if bn == "random":  # BatchNorm statistics / affine away from (0, 1, 1, 0): the folded scale / shift matter
        randomize_bn(m)
    m = m.cuda().eval()
    feats, calib = _bench_inputs(B)
    x = feats.cuda()
    with torch.no_grad():
        out1 = m(x, *calib).clone()      # call 1: eager launches, records the conv plan
        out2 = m(x, *calib).clone()      # call 2: plan replay = what the timed steps run
        out3 = m(x, *calib).clone()
        # the hand-off tensor of the timed step (NHWC bf16 grid) and the public fp32 layout, same pipeline
        grid_bf = m._lift_splat(x, *calib, ops.BEV_NHWC_BF16).float().cpu()
        ws = next(iter(m._ws.values()))
        vox = ws.voxel.cpu().numpy()
        grid32 = m.get_voxels(x, *calib).cpu()
    assert out1.shape == (B, 4, 200, 200) and out1.dtype == torch.float32
    assert torch.equal(out1, out2) and torch.equal(out2, out3)   # replay == eager, and reproducible

    # voxel ids: exact against the oracle (geometry on the CPU in the reference's op order)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    geom = lo.get_geometry_torch(sd["frustum"], *calib).numpy()
    idx, kept = lo.voxel_indices_np(geom, sd["dx"].numpy(), sd["bx"].numpy(), sd["nx"].numpy())
    bi = np.broadcast_to(np.arange(B).reshape(B, 1, 1, 1, 1), kept.shape)
    vid = np.where(kept, ((bi * 200 + idx[..., 0]) * 200 + idx[..., 1]) * 1 + idx[..., 2], -1)
    assert np.array_equal(vox.reshape(vid.shape), vid)

    # BEV grid vs the reference-order CPU lift-splat: the 1e-3 rule
    ref_grid = lo.lift_splat_torch(feats, sd["camencode.depthnet.weight"], sd["camencode.depthnet.bias"], sd["frustum"],
                                   *calib, sd["dx"], sd["bx"], sd["nx"], B, 41, 64)
    e_grid = report("bench_b4_grid_rel_l2_" + bn, (grid32 - ref_grid).norm() / ref_grid.norm())
    assert e_grid < 1e-3
    assert (grid32 - ref_grid).abs().max() < 1e-3 * ref_grid.abs().max()
    # same occupied cells (element-wise zeros differ: the reference's cumsum-difference leaves exact zeros / tiny
    # residues where the direct sum does not)
    assert torch.equal(grid32.abs().sum(1) > 0, ref_grid.abs().sum(1) > 0)
    # the bf16 hand-off is the same sums rounded once to bf16 (2^-9 relative)
    assert (grid_bf - grid32).abs().max() <= 2.0 ** -8 * grid32.abs().max()

    # whole step vs the fp32 oracle
    ref = bo.bev_encode(ref_grid, {k[len("bevencode."):]: v for k, v in sd.items() if k.startswith("bevencode.")})
    err = report("bench_b4_out_rel_l2_bf16_" + bn, (out2.cpu() - ref).norm() / ref.norm())
    emax = report("bench_b4_out_max_rel_bf16_" + bn, (out2.cpu() - ref).abs().max() / ref.abs().max())
    assert err < BF16_TOL, err
    assert emax < 2 * BF16_TOL, emax


def test_region_pipeline_b4_vs_reference_rows(golden):
    """The region pipeline (what the benched step runs) against the REFERENCE's own voxel_pooling rows of the
    batch-4 fixture - until now only the voxel-list K5 saw `g4_full_b4_train`."""
    g = golden("g4_full_b4_train")
    B, N, D, fH, fW, C = [int(v) for v in g["dims"]]
    assert B == 4
    torch.manual_seed(int(g["seed"]))
    feat_in = torch.randn(B * N, 512, fH, fW)
    fr = lo.create_frustum((128, 352), 16, bench.GRID["dbound"])
    dx, bx, nx = lo.gen_dx_bx(bench.GRID["xbound"], bench.GRID["ybound"], bench.GRID["zbound"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    ws = ops.SplatWorkspace(B * N * D * fH * fW, B * 200 * 200, "cuda")
    outs = {}
    for layout in (ops.BEV_NCHW_F32, ops.BEV_NHWC_F32):
        bev, depth, feat = ops.lift_splat_forward(fr.cuda(), dev(g["inv_post_rots"]), dev(g["post_trans"]),
                                                  dev(g["combine"]), dev(g["trans"]), dx.cuda(), bx.cuda(),
                                                  feat_in.cuda(), dev(g["depthnet_weight"]), dev(g["depthnet_bias"]),
                                                  ws, (B, N, D, fH, fW, C), (200, 200, 1), layout)
        outs[layout] = bev.float().cpu().numpy()
    out = outs[ops.BEV_NCHW_F32]
    assert np.array_equal(out, outs[ops.BEV_NHWC_F32])   # fixed-point sums: the same bits in either layout
    assert int((np.abs(out).sum(1) > 0).sum()) == int(g["n_occupied"])
    pick, ref = g["pick"], g["rows"]
    rows = out[pick[:, 0], :, pick[:, 1], pick[:, 2]]
    assert np.linalg.norm(rows - ref) <= 1e-3 * np.linalg.norm(ref)
    assert np.abs(rows - ref).max() <= 1e-3 * np.abs(ref).max()
    assert np.array_equal(np.abs(rows).sum(1) == 0, np.abs(ref).sum(1) == 0)
    np.testing.assert_allclose(np.sqrt((out.astype(np.float64) ** 2).sum((0, 2, 3))), g["chan_l2"], rtol=1e-3)
    np.testing.assert_allclose(out.astype(np.float64).sum((0, 2, 3)), g["chan_sum"], rtol=1e-3,
                               atol=1e-3 * np.abs(g["chan_sum"]).max())
    assert int(ws.vox_count.abs().sum()) == 0 and int(ws.cursor.abs().sum()) == 0
