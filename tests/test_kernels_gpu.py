"""Parity of every HIP kernel (through the C ABI) against the CPU oracle and the
golden fixtures captured from the reference.  Run with `-m gpu` on an MI355X."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bev_oracle as bo  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402

GRID_DEFAULT = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5],
                    zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
GRID_HIRES = dict(xbound=[-50.0, 50.0, 0.25], ybound=[-50.0, 50.0, 0.25],
                  zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 1.0])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from lss2_multimodal_nu_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def run_k3(ops, g, frustum, dxbxnx, want_geom=True):
    dx, bx, nx = dxbxnx
    B, Ncam = g["trans"].shape[:2]
    D, fH, fW, _ = frustum.shape
    X, Y, Z = [int(v) for v in nx]
    ws = ops.SplatWorkspace(B * Ncam * D * fH * fW, B * X * Y * Z, "cuda")
    geom = ops.points_to_voxels(frustum.cuda(), dev(g["inv_post_rots"]), dev(g["post_trans"]),
                                dev(g["combine"]), dev(g["trans"]), dx.cuda(), bx.cuda(), (X, Y, Z), ws,
                                want_geom=want_geom)
    return ws, geom


G3 = ["g3_val_b1_s0", "g3_val_b1_s1", "g3_val_b1_s2", "g3_train_b1_s0", "g3_train_b1_s1",
      "g3_train_b1_s2", "g3_randn_b1_s0", "g3_randn_b1_s1", "g3_train_b4_s0", "g3_hires_b2_s0"]


@pytest.mark.parametrize("name", G3)
def test_k3_voxel_ids_exact(ops, golden, name):
    """Voxel ids and ego-frame points are BIT-EXACT vs the reference (fixtures
    hold the reference's own outputs; Z = 1 so voxel id == cell id)."""
    g = golden(name)
    hires = "hires" in name
    gc = GRID_HIRES if hires else GRID_DEFAULT
    fr = lo.create_frustum((256, 704) if hires else (128, 352), 16, gc["dbound"])
    dxbxnx = lo.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
    ws, geom = run_k3(ops, g, fr, dxbxnx)
    voxel = ws.voxel.cpu().numpy()
    assert sha(geom.cpu().numpy()) == str(g["geom_sha256"])
    assert sha(voxel) == str(g["cell_sha256"])
    assert int((voxel >= 0).sum()) == int(g["n_kept"])
    if "cell" in g:
        assert np.array_equal(voxel, g["cell"])
    # histogram == bincount of kept ids
    cnt = ws.vox_count.cpu().numpy()
    ref = np.bincount(voxel[voxel >= 0], minlength=cnt.size)
    assert np.array_equal(cnt, ref)


@pytest.mark.parametrize("name", ["g3_train_b1_s0", "g3_train_b4_s0", "g3_randn_b1_s0"])
def test_k4_bucketing(ops, golden, name):
    g = golden(name)
    fr = lo.create_frustum((128, 352), 16, GRID_DEFAULT["dbound"])
    dxbxnx = lo.gen_dx_bx(GRID_DEFAULT["xbound"], GRID_DEFAULT["ybound"], GRID_DEFAULT["zbound"])
    ws, _ = run_k3(ops, g, fr, dxbxnx, want_geom=False)
    for rep in range(2):  # second pass checks the zero-on-return contract
        if rep == 1:
            ops.points_to_voxels(fr.cuda(), dev(g["inv_post_rots"]), dev(g["post_trans"]), dev(g["combine"]),
                                 dev(g["trans"]), dxbxnx[0].cuda(), dxbxnx[1].cuda(),
                                 tuple(int(v) for v in dxbxnx[2]), ws)
        ops.bucket_points(ws)
        torch.cuda.synchronize()
        assert int(ws.vox_count.abs().sum()) == 0 and int(ws.cursor[0]) == 0
        voxel = ws.voxel.cpu().numpy()
        vl = ws.vox_list.cpu().numpy()
        pid = ws.entries.cpu().numpy()[:, 0] >> 7  # no depth given: D = HW = 1, key = point id << 7
        kept = np.flatnonzero(voxel >= 0)
        assert np.all(ws.entries.cpu().numpy()[:kept.size, 1].copy().view(np.float32) == 1.0)  # no depth given
        cnt = np.bincount(voxel[kept], minlength=ws.nvox)
        assert np.array_equal(vl[:, 1], cnt)
        occ = np.flatnonzero(cnt)
        # slices are disjoint and tile [0, K)
        order = np.argsort(vl[occ, 0])
        starts, lens = vl[occ, 0][order], vl[occ, 1][order]
        assert starts[0] == 0 and np.array_equal(starts[1:], np.cumsum(lens)[:-1])
        assert starts[-1] + lens[-1] == kept.size
        # every slice holds exactly the voxel's points
        got_vox = np.repeat(occ[order], lens)
        assert np.array_equal(voxel[pid[:kept.size]], got_vox)
        assert np.array_equal(np.sort(pid[:kept.size]), kept)


def _depthnet_ref(x, w, b, D, C):
    depth, _ = lo.cam_encode_torch(x, w, b, D, C)
    y = torch.nn.functional.conv2d(x, w, b)
    return depth, y[:, D:D + C].permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("shape", [(2, 512, 2, 3, 41, 64), (24, 512, 8, 22, 41, 64), (3, 512, 16, 44, 60, 64),
                                   (2, 768, 5, 7, 41, 128), (1, 128, 3, 3, 5, 64)])
def test_k2_depthnet_softmax(ops, golden, shape):
    BN, Cin, fH, fW, D, C = shape
    if shape[:4] == (2, 512, 2, 3):
        g = golden("g6_camencode")
        x, w, b = torch.from_numpy(g["x"]), torch.from_numpy(g["weight"]), torch.from_numpy(g["bias"])
    else:
        gen = torch.Generator().manual_seed(BN + fH)
        x = torch.randn(BN, Cin, fH, fW, generator=gen)
        w = torch.randn(D + C, Cin, 1, 1, generator=gen) * Cin ** -0.5
        b = torch.randn(D + C, generator=gen) * 0.1
    dref, fref = _depthnet_ref(x, w, b, D, C)
    depth, feat = ops.depthnet_softmax(x.cuda(), w.cuda(), b.cuda(), D, C, ops.DT_F32)
    # fp32 MFMA = exact fp32 FMA chains: only the summation order differs from oneDNN
    np.testing.assert_allclose(depth.cpu().numpy(), dref.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(feat.cpu().numpy(), fref.numpy(), rtol=1e-4, atol=1e-5)
    if shape[:4] == (2, 512, 2, 3):
        np.testing.assert_allclose(depth.cpu().numpy(), g["depth"], rtol=2e-5, atol=1e-7)
        lifted = depth.unsqueeze(1).cpu() * feat.permute(0, 3, 1, 2).unsqueeze(2).cpu()
        np.testing.assert_allclose(lifted.numpy(), g["lifted"], rtol=1e-4, atol=1e-6)
    if Cin % 128 == 0:
        d16, f16 = ops.depthnet_softmax(x.cuda(), w.cuda(), b.cuda(), D, C, ops.DT_BF16)
        # bf16 operands: ~2^-9 relative per product, K = Cin
        assert float((d16.cpu() - dref).abs().max()) < 2e-2
        assert float((f16.cpu() - fref).abs().max()) < 6e-2 * float(fref.abs().max())
        np.testing.assert_allclose(d16.sum(1).cpu().numpy(), 1.0, rtol=1e-5)


def _small(golden, name):
    g = golden(name)
    B, N, D, fH, fW, C = [int(v) for v in g["dims"]]
    xb, yb, zb, db = g["bounds"].tolist()
    return g, (B, N, D, fH, fW, C), lo.gen_dx_bx(xb, yb, zb), db


LAYOUTS = [0, 1, 2]


@pytest.mark.parametrize("name", ["g4_small_c64", "g4_small_z2"])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_k5_lift_splat_small_vs_reference(ops, golden, name, layout):
    """depthnet -> K3 -> K4 -> K5 on the small fixtures: BEV vs the REFERENCE's
    voxel_pooling output (1e-3 rule of SURVEY 8a-7) and vs the fp64 direct sum."""
    g, dims, (dx, bx, nx), db = _small(golden, name)
    B, N, D, fH, fW, C = dims
    X, Y, Z = [int(v) for v in nx]
    fr = torch.from_numpy(g["frustum"])
    ws, geom = run_k3(ops, g, fr, (dx, bx, nx))
    assert np.array_equal(geom.cpu().numpy(), g["geom"])
    vox_ref = np.where(g["cell"] >= 0, g["cell"] * Z + g["iz"], -1)
    assert np.array_equal(ws.voxel.cpu().numpy(), vox_ref)
    depth, feat = ops.depthnet_softmax(dev(g["feat_in"]), dev(g["depthnet_weight"]), dev(g["depthnet_bias"]), D, C)
    np.testing.assert_allclose(depth.cpu().numpy(), g["depth"], rtol=2e-5, atol=1e-7)
    ops.bucket_points(ws, depth)
    bev = ops.lift_splat_fwd(feat, ws, dims, (X, Y, Z), layout)
    assert tuple(bev.shape) == (B, Z * C, X, Y)
    out = bev.float().cpu().numpy()
    ref = g["out"]
    tol = 1e-3 if layout != 2 else 8e-3  # bf16 storage: 2^-9 relative
    assert np.linalg.norm(out - ref) <= tol * np.linalg.norm(ref)
    assert np.abs(out - ref).max() <= tol * np.abs(ref).max()
    assert np.array_equal(out == 0, ref == 0)  # same occupancy, empty voxels exactly 0
    direct = lo.splat_direct_np(g["cell"], g["iz"], depth.cpu().numpy(), feat.cpu().numpy().reshape(B * N, fH * fW, C).transpose(0, 2, 1),
                                B, N, D, fH, fW, C, X, Y, Z)
    if layout != 2:
        np.testing.assert_allclose(out, direct, rtol=2e-5, atol=2e-6 * np.abs(direct).max())


@pytest.mark.parametrize("name,bsz", [("g4_full_b1_val", 1), ("g4_full_b4_train", 4)])
def test_k5_full_size_vs_reference_stats(ops, golden, name, bsz):
    g = golden(name)
    B, N, D, fH, fW, C = [int(v) for v in g["dims"]]
    torch.manual_seed(int(g["seed"]))
    feat_in = torch.randn(B * N, 512, fH, fW)
    assert sha(feat_in.numpy()) == str(g["feat_sha256"])
    fr = lo.create_frustum((128, 352), 16, GRID_DEFAULT["dbound"])
    dxbxnx = lo.gen_dx_bx(GRID_DEFAULT["xbound"], GRID_DEFAULT["ybound"], GRID_DEFAULT["zbound"])
    ws, _ = run_k3(ops, g, fr, dxbxnx, want_geom=False)
    depth, feat = ops.depthnet_softmax(feat_in.cuda(), dev(g["depthnet_weight"]), dev(g["depthnet_bias"]), D, C)
    ops.bucket_points(ws, depth)
    outs = {}
    for layout in LAYOUTS:
        outs[layout] = ops.lift_splat_fwd(feat, ws, (B, N, D, fH, fW, C), (200, 200, 1), layout).float().cpu().numpy()
    assert np.array_equal(outs[0], outs[1])  # same sums, two layouts
    out = outs[0]
    occ = np.abs(out).sum(1) > 0
    assert int(occ.sum()) == int(g["n_occupied"])
    pick = g["pick"]
    rows = out[pick[:, 0], :, pick[:, 1], pick[:, 2]]
    ref = g["rows"]
    assert np.linalg.norm(rows - ref) <= 1e-3 * np.linalg.norm(ref)
    assert np.abs(rows - ref).max() <= 1e-3 * np.abs(ref).max()
    assert np.array_equal(np.abs(rows).sum(1) == 0, np.abs(ref).sum(1) == 0)
    np.testing.assert_allclose(np.sqrt((out.astype(np.float64) ** 2).sum((0, 2, 3))), g["chan_l2"], rtol=1e-3)
    np.testing.assert_allclose(out.astype(np.float64).sum((0, 2, 3)), g["chan_sum"], rtol=1e-3,
                               atol=1e-3 * np.abs(g["chan_sum"]).max())
    assert np.abs(outs[2] - out).max() <= 8e-3 * np.abs(out).max()
    # run-to-run reproducible (per-voxel sums are ordered by point id)
    ops.points_to_voxels(fr.cuda(), dev(g["inv_post_rots"]), dev(g["post_trans"]), dev(g["combine"]),
                         dev(g["trans"]), dxbxnx[0].cuda(), dxbxnx[1].cuda(), (200, 200, 1), ws)
    ops.bucket_points(ws, depth)
    again = ops.lift_splat_fwd(feat, ws, (B, N, D, fH, fW, C), (200, 200, 1), 0).cpu().numpy()
    assert np.array_equal(again, out)


@pytest.mark.parametrize("name", ["g4_small_c64", "g4_small_z2"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_k7_backward_vs_reference(ops, golden, name, channels_last):
    g, dims, (dx, bx, nx), db = _small(golden, name)
    B, N, D, fH, fW, C = dims
    X, Y, Z = [int(v) for v in nx]
    ws, _ = run_k3(ops, g, torch.from_numpy(g["frustum"]), (dx, bx, nx), want_geom=False)
    depth, feat = ops.depthnet_softmax(dev(g["feat_in"]), dev(g["depthnet_weight"]), dev(g["depthnet_bias"]), D, C)
    G = dev(g["grad_out"])
    if channels_last:
        G = G.contiguous(memory_format=torch.channels_last)
    g_logits = ops.lift_splat_bwd(G, ws.voxel, depth, feat, dims, (X, Y, Z)).cpu()
    # oracle: push the reference's grad_lifted through the outer product + softmax by autograd
    x_in = torch.from_numpy(g["feat_in"])
    w, b = torch.from_numpy(g["depthnet_weight"]), torch.from_numpy(g["depthnet_bias"])
    logits = torch.nn.functional.conv2d(x_in, w, b).requires_grad_(True)
    dep = logits[:, :D].softmax(1)
    lifted = dep.unsqueeze(1) * logits[:, D:D + C].unsqueeze(2)
    lifted.backward(torch.from_numpy(g["grad_lifted"]))
    ref = logits.grad.numpy()
    np.testing.assert_allclose(g_logits.numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    # and all the way down to the reference's parameter / input gradients
    gl = g_logits.reshape(B * N, D + C, fH * fW)
    gw = torch.einsum("bnp,bkp->nk", gl, x_in.reshape(B * N, 512, fH * fW))
    np.testing.assert_allclose(gw.numpy(), g["grad_weight"][:, :, 0, 0], rtol=1e-3, atol=1e-4 * np.abs(g["grad_weight"]).max())
    np.testing.assert_allclose(gl.sum((0, 2)).numpy(), g["grad_bias"], rtol=1e-3, atol=1e-4 * np.abs(g["grad_bias"]).max())
    gx = torch.einsum("bnp,nk->bkp", gl, w[:, :, 0, 0]).reshape(x_in.shape)
    np.testing.assert_allclose(gx.numpy(), g["grad_feat_in"], rtol=1e-3, atol=1e-4 * np.abs(g["grad_feat_in"]).max())


@pytest.mark.parametrize("cfg", [("c128_default", 2, 6, 41, 8, 22, 128, (200, 200, 1), False),
                                 ("hires_c64", 2, 6, 60, 16, 44, 64, (400, 400, 1), True),
                                 ("odd_grid_z3", 1, 3, 7, 5, 9, 64, (37, 53, 3), False)])
def test_k5_k7_other_configs_vs_fp64(ops, cfg):
    """C = 128 (VoVNet row of SURVEY 8a-10), the hi-res config 5 shapes, and a grid whose
    X*Y is not a multiple of the 64-cell tile with Z > 1: forward vs the fp64 direct sum,
    backward vs the fp64 adjoint."""
    name, B, N, D, fH, fW, C, (X, Y, Z), hires = cfg
    gen = torch.Generator().manual_seed(len(name))
    if hires:
        gc = GRID_HIRES
        fr = lo.create_frustum((256, 704), 16, gc["dbound"])
        dx, bx, nx = lo.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
        rig = lo.synthetic_rig(B, final_dim=(256, 704), train_aug=True, seed=3)
    else:
        span = 50.0 if X == 200 else 12.0
        dx, bx, nx = lo.gen_dx_bx([-span, span, 2 * span / X], [-span, span, 2 * span / Y], [-3.0, 3.0, 6.0 / Z])
        fr = lo.create_frustum((fH * 16, fW * 16), 16, [4.0, 4.0 + D, 1.0])
        rig = lo.synthetic_rig(B, N=N, final_dim=(fH * 16, fW * 16), train_aug=True, seed=2)
        if X != 200:  # squeeze the rig into the small grid
            rig = list(rig)
            rig[2] = rig[2].clone()
            rig[2][..., 0, 0] = 90.0; rig[2][..., 1, 1] = 90.0; rig[2][..., 0, 2] = fW * 8.0; rig[2][..., 1, 2] = fH * 8.0
            rig[3] = torch.eye(3).repeat(B, N, 1, 1); rig[4] = torch.zeros(B, N, 3)
    assert [int(v) for v in nx] == [X, Y, Z]
    rots, trans, intr, prot, ptr = rig
    inv_pr, comb = lo.calib_matrices(rots, intr, prot)
    g = {"inv_post_rots": inv_pr.numpy(), "post_trans": ptr.numpy(), "combine": comb.numpy(), "trans": trans.numpy()}
    ws, geom = run_k3(ops, g, fr, (dx, bx, nx))
    geom_ref = lo.geometry_points_np(fr.numpy(), g["inv_post_rots"], g["post_trans"], g["combine"], g["trans"])
    assert np.array_equal(geom.cpu().numpy(), geom_ref, equal_nan=True)
    cell, iz = lo.cell_ids_np(geom_ref, dx.numpy(), bx.numpy(), nx.numpy())
    vox_ref = np.where(cell >= 0, cell.astype(np.int64) * Z + iz, -1).astype(np.int32)
    assert np.array_equal(ws.voxel.cpu().numpy(), vox_ref)
    assert (vox_ref >= 0).mean() > 0.3
    x = torch.randn(B * N, 512, fH, fW, generator=gen)
    w = torch.randn(D + C, 512, 1, 1, generator=gen) * 512 ** -0.5
    b = torch.randn(D + C, generator=gen) * 0.1
    depth, feat = ops.depthnet_softmax(x.cuda(), w.cuda(), b.cuda(), D, C)
    ops.bucket_points(ws, depth)
    dims = (B, N, D, fH, fW, C)
    outs = [ops.lift_splat_fwd(feat, ws, dims, (X, Y, Z), lay).float().cpu().numpy() for lay in (0, 1)]
    assert np.array_equal(outs[0], outs[1])
    direct = lo.splat_direct_np(cell, iz, depth.cpu().numpy(), feat.cpu().numpy().reshape(B * N, fH * fW, C).transpose(0, 2, 1),
                                B, N, D, fH, fW, C, X, Y, Z)
    np.testing.assert_allclose(outs[0], direct, rtol=3e-5, atol=3e-6 * np.abs(direct).max())
    # backward: adjoint of the same linear map in fp64
    G = torch.randn(B, Z * C, X, Y, generator=gen)
    g_logits = ops.lift_splat_bwd(G.cuda().contiguous(memory_format=torch.channels_last), ws.voxel, depth, feat, dims, (X, Y, Z)).cpu()
    Gr = G.double().view(B, Z, C, X, Y).permute(0, 3, 4, 1, 2).reshape(B * X * Y * Z, C).numpy()
    rows = np.where(vox_ref[:, None] >= 0, Gr[np.clip(vox_ref, 0, None)], 0.0)          # (P, C)
    dep = depth.cpu().double().numpy().reshape(B * N, D, fH * fW)
    ft = feat.cpu().double().numpy().reshape(B * N, fH * fW, C)
    rows = rows.reshape(B * N, D, fH * fW, C)
    g_feat = np.einsum("bdp,bdpc->bpc", dep, rows)
    g_dep = np.einsum("bpc,bdpc->bdp", ft, rows)
    g_logit_d = dep * (g_dep - (dep * g_dep).sum(1, keepdims=True))
    ref = np.concatenate([g_logit_d, g_feat.transpose(0, 2, 1)], 1).reshape(B * N, D + C, fH, fW)
    np.testing.assert_allclose(g_logits.numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    # the bf16 channels-last gradient a bf16 stem hands back is read as it is: the same numbers as its fp32 copy
    Gb = G.cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    g16 = ops.lift_splat_bwd(Gb, ws.voxel, depth, feat, dims, (X, Y, Z))
    g32 = ops.lift_splat_bwd(Gb.float(), ws.voxel, depth, feat, dims, (X, Y, Z))
    assert torch.equal(g16, g32)


def test_segmented_sum(ops, golden):
    g = golden("g5_quickcumsum")
    ranks = g["ranks"]
    starts = np.flatnonzero(np.r_[True, ranks[1:] != ranks[:-1], True]).astype(np.int32)
    x = np.tile(g["x"], (1, 40))[:, :70]
    y = ops.segmented_sum(dev(x), dev(starts)).cpu().numpy()
    ref = np.add.reduceat(x.astype(np.float64), starts[:-1], axis=0)
    np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(y[:, :2], g["y"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt", [0, 1])
def test_layout_round_trip(ops, dt):
    x = torch.randn(3, 70, 9, 13)
    y = ops.nchw_to_nhwc(x.cuda(), dt)
    ref = x.permute(0, 2, 3, 1)
    tol = 0 if dt == 0 else 8e-3
    assert float((y.float().cpu() - ref).abs().max()) <= tol * 5
    back = ops.nhwc_to_nchw(y, dt).cpu()
    assert float((back - x).abs().max()) <= tol * 5


CONVS = [
    # B, H, W, Cx, Cout, k, stride, pad, relu, residual, C2, up
    (2, 20, 24, 64, 64, 3, 1, 1, True, True, 0, 1),
    (1, 21, 19, 64, 128, 3, 2, 1, True, False, 0, 1),
    (2, 20, 20, 64, 64, 7, 2, 3, True, False, 0, 1),
    (1, 17, 9, 64, 128, 1, 2, 0, False, False, 0, 1),
    (2, 12, 10, 128, 4, 1, 1, 0, False, False, 0, 1),
    (1, 5, 7, 256, 256, 3, 1, 1, True, False, 64, 4),
    (2, 6, 5, 128, 64, 3, 1, 1, True, False, 0, 2),
    (1, 9, 9, 64, 96, 3, 1, 1, False, False, 64, 1),
]


def _q(t, dt):
    """What the bf16 kernels actually consume: operands rounded to bf16 once (exactly representable in fp32).
    The fp32 torch reference is evaluated on THESE values, so the remaining difference is fp32 summation
    order + the single bf16 rounding of the output (<= 2^-8 relative) - and the tolerance can be that tight."""
    return t if (dt == 0 or t is None) else t.bfloat16().float()


# bf16 kernels vs the fp32 reference on bf16-rounded operands: half an ulp of the output rounding is 2^-8 =
# 3.9e-3 of |value|; measured maxima (r02, gpurun_out/test_errors.txt) sit at 2.0e-3 .. 3.9e-3 of max|ref|.
BF16_OUT_TOL = 6e-3
# fused bilinear upsample: the interpolated operand is itself rounded to bf16 before the MFMA (one extra rounding
# per input element); measured <= 4.5e-3
BF16_UP_TOL = 1.2e-2


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("dt", [0, 1])
def test_k8_conv_vs_torch(ops, report, cfg, dt):
    B, H, W, Cx, Cout, k, stride, pad, relu, use_res, C2, up = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    x = _q(torch.randn(B, Cx, H, W, generator=gen), dt)
    x2 = _q(torch.randn(B, C2, H * up, W * up, generator=gen), dt) if C2 else None
    w = _q(torch.randn(Cout, Cx + C2, k, k, generator=gen) * ((Cx + C2) * k * k) ** -0.5, dt)
    scale = torch.rand(Cout, generator=gen) + 0.5
    shift = torch.randn(Cout, generator=gen) * 0.1
    xin = x
    if up > 1:
        xin = bo.upsample_bilinear_ac(x, up)
    if C2:
        xin = torch.cat([x2, xin], 1)
    raw = torch.nn.functional.conv2d(xin, w, None, stride=stride, padding=pad)
    ref = raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _q(torch.randn(ref.shape, generator=gen), dt) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = ref.relu()
    xg = ops.nchw_to_nhwc(x.cuda(), dt)
    x2g = ops.nchw_to_nhwc(x2.cuda(), dt) if C2 else None
    resg = ops.nchw_to_nhwc(res.cuda(), dt) if use_res else None
    wp = ops.pack_conv_weight(w.cuda(), dt)
    stats = torch.zeros(2 * Cout, device="cuda")
    y = ops.conv2d_nhwc(xg, wp, (k, k), stride, pad, scale.cuda(), shift.cuda(), resg, relu, x2g, up, stats, dt)
    out = ops.nhwc_to_nchw(y, dt).cpu()
    assert out.shape == ref.shape
    tol = 2e-5 if dt == 0 else (BF16_UP_TOL if up > 1 else BF16_OUT_TOL)
    err = report("k8_conv_max_rel_dt%d_%s" % (dt, "x".join(str(int(c)) for c in cfg)),
                 (out - ref).abs().max() / ref.abs().max())
    assert err <= tol
    s = stats.cpu()
    np.testing.assert_allclose(s[:Cout].numpy(), raw.sum((0, 2, 3)).numpy(), rtol=tol * 4, atol=tol * 4 * float(raw.abs().sum((0, 2, 3)).max()))
    np.testing.assert_allclose(s[Cout:].numpy(), (raw ** 2).sum((0, 2, 3)).numpy(), rtol=max(tol * 4, 1e-4))


S2_CONVS = [  # B, H, W, Cin, Cout, k, relu, residual
    (2, 20, 24, 64, 64, 7, True, False),
    (1, 21, 19, 64, 128, 3, True, False),
    (2, 50, 50, 128, 256, 3, False, True),
    (1, 33, 40, 64, 64, 3, True, False),
    (1, 37, 18, 128, 200, 7, False, False),
    (2, 50, 50, 128, 256, 1, False, False),
    (1, 21, 33, 64, 128, 1, False, False),
]


@pytest.mark.parametrize("cfg", S2_CONVS)
def test_k8_stride2_conv_phase_plane_path(ops, report, cfg):
    """Stride-2 convs through the space-to-depth form of the LDS-tiled kernel."""
    B, H, W, Cin, Cout, k, relu, use_res = cfg
    pad = k // 2
    gen = torch.Generator().manual_seed(sum(cfg[:6]))
    x = _q(torch.randn(B, Cin, H, W, generator=gen), 1)
    w = _q(torch.randn(Cout, Cin, k, k, generator=gen) * (Cin * k * k) ** -0.5, 1)
    scale = torch.rand(Cout, generator=gen) + 0.5
    shift = torch.randn(Cout, generator=gen) * 0.1
    raw = torch.nn.functional.conv2d(x, w, None, stride=2, padding=pad)
    ref = raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _q(torch.randn(ref.shape, generator=gen), 1) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = ref.relu()
    xg = ops.nchw_to_nhwc(x.cuda(), 1)
    resg = ops.nchw_to_nhwc(res.cuda(), 1) if use_res else None
    wp = ops.pack_conv_weight_s2d(w.cuda(), pad) if k > 1 else ops.pack_conv_weight(w.cuda(), 1)
    stats = torch.zeros(2 * Cout, device="cuda")
    y = ops.conv2d_s2_nhwc(xg, wp, k, pad, scale.cuda(), shift.cuda(), resg, relu, stats)
    out = ops.nhwc_to_nchw(y, 1).cpu()
    assert out.shape == ref.shape
    assert report("k8_s2_max_rel_" + "x".join(str(int(c)) for c in cfg),
                  (out - ref).abs().max() / ref.abs().max()) <= BF16_OUT_TOL
    # and it agrees with the generic direct kernel on the same bf16 inputs to fp32-sum noise (one bf16 ulp
    # where the two fp32 sums round to different neighbours)
    y0 = ops.conv2d_nhwc(xg, ops.pack_conv_weight(w.cuda(), 1), (k, k), 2, pad, scale.cuda(), shift.cuda(), resg, relu, dt=1)
    assert float((y.float() - y0.float()).abs().max()) <= 8e-3 * float(ref.abs().max())
    np.testing.assert_allclose(stats[:Cout].cpu().numpy(), raw.sum((0, 2, 3)).numpy(), rtol=0.1,
                               atol=0.1 * float(raw.abs().sum((0, 2, 3)).max()))


@pytest.mark.parametrize("cfg", [(2, 12, 10, 256, 0, 2, 4), (1, 9, 16, 64, 0, 1, 3), (1, 5, 6, 128, 64, 4, 4)])
def test_k8_conv_with_fused_head(ops, report, cfg):
    """up2 of BevEncode in one launch: (upsample) + 3x3 conv + scale/shift + ReLU + 1x1 head -> NCHW fp32."""
    B, H, W, Cx, C2, up, n = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    x = _q(torch.randn(B, Cx, H, W, generator=gen), 1)
    x2 = _q(torch.randn(B, C2, H * up, W * up, generator=gen), 1) if C2 else None
    w = _q(torch.randn(128, Cx + C2, 3, 3, generator=gen) * ((Cx + C2) * 9) ** -0.5, 1)
    scale, shift = torch.rand(128, generator=gen) + 0.5, torch.randn(128, generator=gen) * 0.1
    hw, hb = torch.randn(n, 128, generator=gen) * 128 ** -0.5, torch.randn(n, generator=gen)
    xin = bo.upsample_bilinear_ac(x, up) if up > 1 else x
    if C2:
        xin = torch.cat([x2, xin], 1)
    act = (torch.nn.functional.conv2d(xin, w, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).relu()
    ref = torch.nn.functional.conv2d(act, hw.view(n, 128, 1, 1), hb)
    out = ops.conv3x3_head_nchw(ops.nchw_to_nhwc(x.cuda(), 1), ops.pack_conv_weight(w.cuda(), 1), scale.cuda(),
                                shift.cuda(), hw.cuda(), hb.cuda(), x2=ops.nchw_to_nhwc(x2.cuda(), 1) if C2 else None,
                                up=up).cpu()
    assert out.shape == ref.shape and out.is_contiguous()
    # the head sums 128 activations: with an MFMA head the activation is rounded to bf16 first (as the unfused
    # two-launch path does), so the bound is the output-rounding one, not fp32 noise
    assert report("k8_head_max_rel_" + "x".join(str(int(c)) for c in cfg),
                  (out - ref).abs().max() / ref.abs().max()) <= (BF16_UP_TOL if up > 1 else BF16_OUT_TOL)


@pytest.mark.parametrize("rt", ["1", "2"])
def test_k8_conv_tile_variants_agree(ops, rt, monkeypatch):
    """Both workgroup shapes (RT = 1 / 2 row tiles per wave) of the LDS-tiled kernel."""
    monkeypatch.setenv("LSS_CONV_RT", rt)
    gen = torch.Generator().manual_seed(11)
    for (B, H, W, Cin, Cout) in ((2, 25, 25, 128, 256), (1, 37, 20, 64, 64), (1, 9, 50, 192, 130)):
        x = _q(torch.randn(B, Cin, H, W, generator=gen), 1)
        w = _q(torch.randn(Cout, Cin, 3, 3, generator=gen) * (Cin * 9) ** -0.5, 1)
        ref = torch.nn.functional.conv2d(x, w, None, padding=1)
        y = ops.conv2d_nhwc(ops.nchw_to_nhwc(x.cuda(), 1), ops.pack_conv_weight(w.cuda(), 1), (3, 3), 1, 1, dt=1)
        out = ops.nhwc_to_nchw(y, 1).cpu()
        assert float((out - ref).abs().max()) <= BF16_OUT_TOL * float(ref.abs().max())


def test_k8_stem_two_taps_per_step(ops, monkeypatch):
    """7x7 / 2 stem: steps of two taps against a double slab (the default) = steps of one tap, bitwise."""
    gen = torch.Generator().manual_seed(5)
    x = _q(torch.randn(2, 64, 200, 200, generator=gen), 1)
    w = _q(torch.randn(64, 64, 7, 7, generator=gen) * (64 * 49) ** -0.5, 1)
    scale, shift = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen) * 0.1
    xg, wp = ops.nchw_to_nhwc(x.cuda(), 1), ops.pack_conv_weight_s2d(w.cuda(), 3)
    y2 = ops.conv2d_s2_nhwc(xg, wp, 7, 3, scale.cuda(), shift.cuda(), None, True)
    ref = (torch.nn.functional.conv2d(x, w, None, stride=2, padding=3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).relu()
    out = ops.nhwc_to_nchw(y2, 1).cpu()
    assert float((out - ref).abs().max()) <= BF16_OUT_TOL * float(ref.abs().max())


def test_bad_arguments_raise(ops):
    ws = ops.SplatWorkspace(10, 10, "cuda")
    with pytest.raises(ValueError):
        ops.lift_splat_fwd(torch.zeros(10, device="cuda"), ws, (1, 1, 1, 1, 10, 7), (10, 1, 1))
    with pytest.raises(ValueError):
        ops.depthnet_softmax(torch.zeros(1, 100, 2, 2, device="cuda"), torch.zeros(8, 100, device="cuda"),
                             torch.zeros(8, device="cuda"), 4, 4)


# ---------------------------------------------------------------------------
# Edge cases of the splat: empty grids, one voxel holding every point (collisions far beyond the
# 64-entry chunk), ragged voxel populations, non-finite geometry.
# ---------------------------------------------------------------------------
def _pool(ops, geom, x, B, nx, dx, bx):
    """voxel_pooling-style call: geom (P,3), x (P,C) pre-lifted rows -> BEV (B, Z*C, X, Y)."""
    P, C = x.shape
    X, Y, Z = nx
    ws = ops.SplatWorkspace(P, B * X * Y * Z, "cuda")
    ops.geom_to_voxels(geom.cuda().contiguous(), dx.cuda(), bx.cuda(), nx, B, ws)
    cnt = ws.vox_count.clone()
    ops.bucket_points(ws)
    assert int(ws.vox_count.abs().sum()) == 0 and int(ws.cursor.abs().sum()) == 0  # counters back at zero
    return ops.lift_splat_fwd(x.cuda().contiguous(), ws, (B, P // B, 1, 1, 1, C), nx, 0), cnt, ws


def _centre(ix, iy, dx, bx):
    lo0 = (bx - dx / 2)
    return (lo0[0] + (ix + 0.5) * dx[0], lo0[1] + (iy + 0.5) * dx[1], lo0[2] + 0.5 * dx[2])


def test_splat_empty_grid(ops):
    dx, bx, nx = lo.gen_dx_bx([-10.0, 10.0, 1.0], [-10.0, 10.0, 1.0], [-10.0, 10.0, 20.0])
    B, P, C = 2, 4096, 64
    geom = torch.full((P, 3), 1000.0)  # every point far outside
    geom[::7] = float("nan")
    geom[1::7] = float("inf")
    geom[2::7, 0] = -1e30
    bev, cnt, ws = _pool(ops, geom, torch.randn(P, C), B, (20, 20, 1), dx, bx)
    assert int(cnt.sum()) == 0 and bool((ws.voxel == -1).all())
    assert bev.shape == (B, C, 20, 20) and float(bev.abs().max()) == 0.0


@pytest.mark.parametrize("C", [64, 128])
def test_splat_every_point_in_one_voxel(ops, C):
    """5000 points collide in one cell (78 chunks of 64): fp64 sum parity and bitwise reproducibility."""
    dx, bx, nx = lo.gen_dx_bx([-10.0, 10.0, 1.0], [-10.0, 10.0, 1.0], [-10.0, 10.0, 20.0])
    B, P = 1, 5000
    g = torch.Generator().manual_seed(C)
    cx, cy, cz = _centre(3, 17, dx, bx)
    geom = torch.stack([torch.full((P,), float(cx)), torch.full((P,), float(cy)), torch.full((P,), float(cz))], 1)
    geom += (torch.rand(P, 3, generator=g) - 0.5) * 0.4  # jitter inside the cell
    x = torch.randn(P, C, generator=g)
    bev, cnt, _ = _pool(ops, geom, x, B, (20, 20, 1), dx, bx)
    assert int(cnt.sum()) == P and int(cnt.max()) == P
    ref = x.double().sum(0)
    got = bev[0, :, 3, 17].double().cpu()
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-4
    other = bev.clone()
    other[0, :, 3, 17] = 0
    assert float(other.abs().max()) == 0.0
    # beyond 64 points a voxel is summed in 64-entry chunks whose membership follows the (atomic) fill
    # order: equal up to fp32 re-association, not bitwise (voxels of <= 64 points are: next test)
    bev2, _, _ = _pool(ops, geom, x, B, (20, 20, 1), dx, bx)
    assert float((bev - bev2).abs().max()) <= 1e-5 * float(bev.abs().max())


def test_splat_ragged_voxel_populations(ops):
    """Voxels holding 1, 63, 64, 65, 129 and 1000 points next to empty ones, two samples."""
    dx, bx, nx = lo.gen_dx_bx([-16.0, 16.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    g = torch.Generator().manual_seed(0)
    pops = [1, 63, 64, 65, 129, 1000]
    B, C = 2, 64
    per_sample = 1536  # >= sum(pops) = 1322, rest outside the grid
    geoms, want = [], torch.zeros(B, C, 32, 32, dtype=torch.float64)
    xs = torch.randn(B * per_sample, C, generator=g)
    for b in range(B):
        rows = []
        for k, n in enumerate(pops):
            ix, iy = (5 * k + 3 * b) % 32, (7 * k + 11 * b) % 32
            cx, cy, cz = _centre(ix, iy, dx, bx)
            rows.append(torch.tensor([[float(cx), float(cy), float(cz)]]).repeat(n, 1))
        rows.append(torch.full((per_sample - sum(pops), 3), 500.0))
        gb = torch.cat(rows)
        perm = torch.randperm(per_sample, generator=g)  # points of a voxel are not contiguous in the input
        geoms.append(gb[perm])
        xb = xs[b * per_sample:(b + 1) * per_sample]
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(per_sample)
        o = 0
        for k, n in enumerate(pops):
            ix, iy = (5 * k + 3 * b) % 32, (7 * k + 11 * b) % 32
            want[b, :, ix, iy] += xb[inv[o:o + n]].double().sum(0)
            o += n
    bev, cnt, _ = _pool(ops, torch.cat(geoms), xs, B, (32, 32, 1), dx, bx)
    assert int(cnt.sum()) == B * sum(pops) and sorted(cnt[cnt > 0].tolist()) == sorted(pops * B)
    assert float((bev.double().cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-4
    assert int((bev != 0).flatten(2).any(1).sum()) == B * len(pops)
    # voxels of <= 64 points are summed in key order: bitwise reproducible from run to run
    bev2, _, _ = _pool(ops, torch.cat(geoms), xs, B, (32, 32, 1), dx, bx)
    for b in range(B):
        for k, n in enumerate(pops):
            if n <= 64:
                ix, iy = (5 * k + 3 * b) % 32, (7 * k + 11 * b) % 32
                assert torch.equal(bev[b, :, ix, iy], bev2[b, :, ix, iy])


@pytest.mark.parametrize("shape", [(2, 100, 100, 64, 128), (1, 50, 50, 128, 256), (2, 37, 41, 64, 128)])
def test_k8_dual_output_stride2_conv(ops, shape):
    """conv1 (3x3/2) + 1x1/2 downsample of a BasicBlock in one launch == the two separate launches."""
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, generator=g).bfloat16().cuda()
    w1 = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    wd = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).cuda()
    sc = (torch.rand(2 * Cout, generator=g) + 0.5).cuda()
    sh = torch.randn(2 * Cout, generator=g).cuda()
    frame = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    frame[:, :, 1, 1] = wd[:, :, 0, 0]
    wcat = torch.cat([ops.pack_conv_weight_s2d(w1, 1), ops.pack_conv_weight_s2d(frame, 1)], 1).contiguous()
    y, y2 = ops.conv2d_s2_dual_nhwc(x, wcat, sc, sh, Cout, relu=True)
    ref1 = ops.conv2d_s2_nhwc(x, ops.pack_conv_weight_s2d(w1, 1), 3, 1, sc[:Cout].contiguous(), sh[:Cout].contiguous(), None, True)
    ref2 = ops.conv2d_s2_nhwc(x, ops.pack_conv_weight(wd, ops.DT_BF16), 1, 0, sc[Cout:].contiguous(), sh[Cout:].contiguous(),
                              None, False)
    assert y.shape == ref1.shape and y2.shape == ref2.shape
    assert torch.equal(y, ref1)
    assert float((y2.float() - ref2.float()).abs().max()) <= 2e-2 * float(ref2.float().abs().max())
    assert float(y2.min()) < 0  # no activation on the second output
