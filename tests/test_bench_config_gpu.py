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


# ---- the TRAINING configuration that bench.py times (configs[2]'s per-GPU shape: batch 4, bf16 autocast, one HIP graph
# per step).  Until round 4 every full-step training test ran at batch 1, where `lss_conv2d_ring_ok` refuses both up1
# layers (62 workgroups) - the up1 ring forward / dgrad, K9w at the 240-256-workgroup batch-4 geometry and the batch-4
# graph replay had never run together under a checker.  Reference loop: /root/reference/train.py:49-66.
CLASS_W = [1.0, 10.0, 5.0, 10.0]   # ref src/tools.py:234 (the model's default in forward_loss)


def _train_model(sd0, lr_zero=False):
    m = L.compile_model_lss(4, bench.GRID, bench.AUG, 4, precision="bf16")
    m.load_state_dict(sd0)
    m = m.cuda().train()
    if lr_zero:  # frozen statistics: nothing may move between an eager step and the replays it is compared with
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 0.0
    return m


class _Amp(torch.nn.Module):
    def __init__(self, inner, tgt):
        super().__init__()
        self.inner, self.tgt = inner, tgt

    def forward(self, *a):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return self.inner.forward_loss(*a, self.tgt)


def _b4_problem():
    B = 4
    torch.manual_seed(0)
    m = L.compile_model_lss(B, bench.GRID, bench.AUG, 4, precision="bf16")
    randomize_bn(m)
    with torch.no_grad():  # zero_init_residual would hide half the net from the gradient check
        be = m.bevencode
        for blk in list(be.layer1) + list(be.layer2) + list(be.layer3):
            blk.bn2.weight.fill_(0.7)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    feats, calib = _bench_inputs(B)
    tgt = torch.randint(0, 4, (B, 200, 200), generator=torch.Generator().manual_seed(77))
    return B, sd0, feats, calib, tgt


# per-tensor gradient bounds by CLASS of tensor, set from what the step measures at batch 4 (every tensor's cosine is
# written to gpurun_out/test_errors.txt by `report`): bf16 activations and gradients through 19 layers against fp32
# (measured minima, round 4: conv 0.9729 [layer1.0.conv1], stem 0.9706, BatchNorm vectors 0.9675 [bn1.weight], depthnet
# 0.9706, head 1.0000 - the stem, bn1 and the depthnet sit at the end of the longest backward chain; the step is
# bit-reproducible, so these do not move from run to run)
GRAD_COS = {"conv": 0.97, "stem": 0.96, "bn": 0.955, "depthnet": 0.96, "head": 0.999}


def _tensor_class(name):
    if name.startswith("camencode."):
        return "depthnet"
    if name.startswith("bevencode.up2.4."):
        return "head"
    if name == "bevencode.conv1.weight":
        return "stem"
    if ".bn" in name or "downsample.1." in name or name.endswith("bn1.weight") or name.endswith("bn1.bias") \
            or ".conv.1." in name or ".conv.4." in name or ".up2.2." in name:
        return "bn"
    return "conv"


def test_benched_training_step_b4_vs_oracle(report):
    """One `forward_loss` step of the benched training shape (batch 4, bf16 autocast) against the CPU oracle's autograd
    of the whole path - lift-splat (`lo.lift_splat_torch`, ref src/model_BEV_TXT.py:128-133) + `bo.bev_encode(training=
    True)` (ref src/modules.py:94-130) + weighted cross-entropy (ref src/tools.py:221-231): the loss, EVERY parameter
    gradient, the updated running statistics; and no flag wait of the ring / K9w kernels at its bound."""
    B, sd0, feats, calib, tgt = _b4_problem()
    # the batch-4 launch set: both up1 layers and up2 take the ring kernel here (they do not at batch 1)
    assert ops.conv_ring_ok(B, 25, 25, 256, 64, 4, 256) and ops.conv_ring_ok(B, 100, 100, 256, 0, 1, 256)
    assert ops.conv_ring_ok(B, 100, 100, 256, 0, 2, 128)
    m = _train_model(sd0)
    x = feats.cuda()
    spans = ops.KernelTimer(fine=True)
    ops.set_timer(spans)
    loss = _Amp(m, tgt.cuda())(x, *calib)
    loss.backward()
    torch.cuda.synchronize()
    ops.set_timer(None)
    assert {"conv_bn_act_train_fwd", "conv_bn_act_train_bwd"} <= set(spans.spans)
    ops.assert_no_timeouts("batch-4 training step")

    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k
                                      and k not in ("dx", "bx", "frustum"))
          for k, v in sd0.items()}
    grid = lo.lift_splat_torch(feats, sd["camencode.depthnet.weight"], sd["camencode.depthnet.bias"], sd["frustum"],
                               *calib, sd["dx"], sd["bx"], sd["nx"], B, 41, 64)
    bsd = {k[len("bevencode."):]: v for k, v in sd.items() if k.startswith("bevencode.")}
    stats = {}
    ref_logits = bo.bev_encode(grid, bsd, training=True, stats_out=stats)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, tgt, weight=torch.tensor(CLASS_W))
    ref_loss.backward()
    e_loss = report("train_b4 loss rel err (bf16 GPU vs fp32 CPU oracle)", abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    assert e_loss < 1e-2
    n, dots, worst = 0, [0.0, 0.0, 0.0], {}
    for name, p in m.named_parameters():
        g_ref = sd[name].grad
        if not sd[name].requires_grad:
            continue
        assert (p.grad is None) == (g_ref is None), name
        if g_ref is None or float(g_ref.norm()) == 0:
            continue
        a, b = p.grad.float().cpu().flatten(), g_ref.flatten()
        cos = report("train_b4 grad cos " + name, float(torch.nn.functional.cosine_similarity(a, b, dim=0)))
        ratio = float(a.norm() / b.norm())
        cls = _tensor_class(name)
        worst[cls] = min(worst.get(cls, 1.0), cos)
        dots = [dots[0] + float(a @ b), dots[1] + float(a @ a), dots[2] + float(b @ b)]
        assert cos > GRAD_COS[cls] and abs(ratio - 1) < 0.1, (name, cls, cos, ratio)
        n += 1
    for cls, v in worst.items():
        report("train_b4 min grad cosine, class " + cls, v)
    whole = report("train_b4 cosine of the whole gradient vs oracle", dots[0] / (dots[1] * dots[2]) ** 0.5)
    assert whole > 0.99 and n >= 52
    for name, buf in m.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            ref = stats.get(name[len("bevencode."):])
            if ref is not None:
                assert torch.allclose(buf.float().cpu(), ref, rtol=3e-2, atol=3e-2), name
    # the same step again from the same state: the same loss and the same gradients, bit for bit
    m2 = _train_model(sd0)
    loss2 = _Amp(m2, tgt.cuda())(x, *calib)
    loss2.backward()
    assert float(loss) == float(loss2)
    for (nm, p), q in zip(m.named_parameters(), m2.parameters()):
        if p.grad is not None and nm.startswith("bevencode."):
            assert torch.equal(p.grad, q.grad), nm
    ops.assert_no_timeouts("batch-4 training step, repeated")


def test_benched_training_graph_replays_equal_eager_b4(report):
    """The batch-4 step as bench.py's train leg runs it - dp.GraphedTrainStep, one HIP graph - with a zero learning rate
    and frozen BatchNorm statistics: replay 1 and replay 3 must leave the gradients of the eager step of the same state,
    bit for bit."""
    from lss2_multimodal_nu_amd import dp
    B, sd0, feats, calib, tgt = _b4_problem()
    x, t = feats.cuda(), tgt.cuda()

    def build():
        m = _train_model(sd0, lr_zero=True)
        return m, torch.optim.Adam(m.parameters(), lr=0.0, capturable=True), _Amp(m, t)

    m1, o1, w1 = build()
    dp.train_step(w1, None, o1, lambda l: l, (x,) + tuple(calib), clip=1e9)
    ref = {n: p.grad.detach().clone() for n, p in m1.named_parameters() if p.grad is not None}
    m2, o2, w2 = build()
    gs = dp.GraphedTrainStep(w2, None, o2, lambda l: l, x, tuple(calib), clip=1e9, warmup=2)
    for rep in (1, 2, 3):
        gs(x, tuple(calib))
        torch.cuda.synchronize()
        if rep == 2:
            continue
        for n, p in m2.named_parameters():
            if p.grad is None:
                continue
            # every gradient, the two depthnet tensors behind the BLAS library's GEMMs included (measured: 0.0)
            assert torch.equal(p.grad, ref[n]), (rep, n, float((p.grad - ref[n]).abs().max()))
    ops.assert_no_timeouts("batch-4 graph replays")
