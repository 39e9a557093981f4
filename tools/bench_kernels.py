#!/usr/bin/env python3
"""Per-kernel micro-benchmark on one GPU: every BevEncode conv shape at batch B
(bf16) and the L1 kernels, timed with HIP events on the launch stream.
    python tools/bench_kernels.py [--batch 4] [--iters 30] [--only conv|l1|vovnet|gemm|grad]
`--only vovnet` times BASELINE configs[3] (vovnet shapes, C=128, BEV transformer) stage by stage."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lss2_multimodal_nu_amd import ops  # noqa: E402

# name, H, W (of x), Cx, Cout, k, stride, pad, C2, up, residual
CONVS = [
    ("conv1 7x7/2", 200, 200, 64, 64, 7, 2, 3, 0, 1, False),
    ("layer1 3x3", 100, 100, 64, 64, 3, 1, 1, 0, 1, True),
    ("layer2.0.c1 3x3/2", 100, 100, 64, 128, 3, 2, 1, 0, 1, False),
    ("layer2.0.ds 1x1/2", 100, 100, 64, 128, 1, 2, 0, 0, 1, False),
    ("layer2 3x3", 50, 50, 128, 128, 3, 1, 1, 0, 1, True),
    ("layer3.0.c1 3x3/2", 50, 50, 128, 256, 3, 2, 1, 0, 1, False),
    ("layer3.0.ds 1x1/2", 50, 50, 128, 256, 1, 2, 0, 0, 1, False),
    ("layer3 3x3", 25, 25, 256, 256, 3, 1, 1, 0, 1, True),
    ("up1.conv0 up4+cat", 25, 25, 256, 256, 3, 1, 1, 64, 4, False),
    ("up1.conv3 3x3", 100, 100, 256, 256, 3, 1, 1, 0, 1, False),
    ("up2.1 up2", 100, 100, 256, 128, 3, 1, 1, 0, 2, False),
    ("up2.4 1x1 head", 200, 200, 128, 4, 1, 1, 0, 0, 1, False),
]
COUNT = {"layer1 3x3": 4, "layer2 3x3": 3, "layer3 3x3": 3}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    B, dt = args.batch, ops.DT_BF16
    tot_us = tot_fl = 0.0
    if args.only in ("", "conv"):
        print("%-22s %10s %10s %8s" % ("conv (B=%d, bf16)" % B, "us", "GFLOP", "TFLOP/s"))
        for name, H, W, Cx, Cout, k, st, pad, C2, up, res in CONVS:
            x = torch.randn(B, H, W, Cx, device="cuda").to(torch.bfloat16)
            x2 = torch.randn(B, H * up, W * up, C2, device="cuda").to(torch.bfloat16) if C2 else None
            s2 = st == 2 and k in (1, 3, 7)
            wraw = torch.randn(Cout, Cx + C2, k, k, device="cuda") * 0.05
            w = ops.pack_conv_weight_s2d(wraw, pad) if (s2 and k > 1) else ops.pack_conv_weight(wraw, dt)
            sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
            Ho = (H * up + 2 * pad - k) // st + 1
            Wo = (W * up + 2 * pad - k) // st + 1
            r = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16) if res else None
            if s2:
                us = timeit(lambda: ops.conv2d_s2_nhwc(x, w, k, pad, sc, sh, r, True), args.iters)
            else:
                us = timeit(lambda: ops.conv2d_nhwc(x, w, (k, k), st, pad, sc, sh, r, True, x2, up, None, dt), args.iters)
            fl = 2.0 * B * Ho * Wo * Cout * (Cx + C2) * k * k
            n = COUNT.get(name, 1)
            tot_us += us * n
            tot_fl += fl * n
            print("%-22s %10.1f %10.2f %8.1f   x%d" % (name, us, fl / 1e9, fl / us / 1e6, n))
        print("%-22s %10.1f %10.2f %8.1f" % ("BevEncode total", tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))
    if args.only == "stamps":
        # in-kernel phase stamps of the LDS-tiled conv (100-MHz s_memrealtime, 8 per workgroup):
        # where a launch's microseconds go - start ramp, prologue, main loop, epilogue, store drain
        import numpy as np
        print("%-22s %5s %7s | %s" % ("conv (B=%d)" % B, "WGs", "evt us",
                                     "start-spread  prologue  loop  stage  stores  drain | first->last us (p50 / max over WGs)"))
        for name, H, W, Cx, Cout, k, st, pad, C2, up, res in CONVS:
            if k == 1:
                continue
            x = torch.randn(B, H, W, Cx, device="cuda").to(torch.bfloat16)
            x2 = torch.randn(B, H * up, W * up, C2, device="cuda").to(torch.bfloat16) if C2 else None
            s2 = st == 2 and k in (3, 7)
            wraw = torch.randn(Cout, Cx + C2, k, k, device="cuda") * 0.05
            w = ops.pack_conv_weight_s2d(wraw, pad) if s2 else ops.pack_conv_weight(wraw, dt)
            sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
            Ho = (H * up + 2 * pad - k) // st + 1
            Wo = (W * up + 2 * pad - k) // st + 1
            r = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16) if res else None
            if s2:
                fn = lambda: ops.conv2d_s2_nhwc(x, w, k, pad, sc, sh, r, True)
            else:
                fn = lambda: ops.conv2d_nhwc(x, w, (k, k), st, pad, sc, sh, r, True, x2, up, None, dt)
            us = timeit(fn, args.iters)
            stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")  # >= any grid here
            os.environ["LSS_CONV_STAMPS"] = "%x" % stamps.data_ptr()
            for _ in range(3):
                stamps.zero_()
                torch.cuda.synchronize()
                fn()
                torch.cuda.synchronize()
            del os.environ["LSS_CONV_STAMPS"]
            t = stamps.view(-1, 8).cpu().numpy()
            t = t[t[:, 0] != 0][:, :6].astype(np.float64) * 0.01  # us
            t0 = t[:, 0].min()
            d = np.diff(t, axis=1)
            f = lambda v: "%5.2f/%5.2f" % (np.median(v), v.max())
            print("%-22s %5d %7.1f | %s  %s  %s  %s  %s  %s | %5.2f / %5.2f" % (
                name, len(t), us, f(t[:, 0] - t0), f(d[:, 0]), f(d[:, 1]), f(d[:, 2]), f(d[:, 3]), f(d[:, 4]),
                np.median(t[:, 5] - t0), (t[:, 5] - t0).max()))
        return
    if args.only in ("", "l1"):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import lss_oracle as lo
        D, C, fH, fW, N = 41, 64, 8, 22, 6
        dx, bx, nx = lo.gen_dx_bx([-50.0, 50.0, 0.5], [-50.0, 50.0, 0.5], [-10.0, 10.0, 20.0])
        fr = lo.create_frustum((128, 352), 16, [4.0, 45.0, 1.0]).cuda()
        rots, trans, intr, prot, ptr = lo.synthetic_rig(B, train_aug=True, seed=0)
        inv_pr, comb = lo.calib_matrices(rots, intr, prot)
        g = [t.contiguous().cuda() for t in (inv_pr, ptr, comb, trans, dx, bx)]
        x = torch.randn(B * N, 512, fH, fW, device="cuda")
        w = torch.randn(D + C, 512, device="cuda") * 512 ** -0.5
        b = torch.randn(D + C, device="cuda") * 0.1
        ws = ops.SplatWorkspace(B * N * D * fH * fW, B * 200 * 200, "cuda")
        depth, feat = ops.depthnet_softmax(x, w, b, D, C)

        def k3():
            ops.points_to_voxels(fr, g[0], g[1], g[2], g[3], g[4], g[5], (200, 200, 1), ws)

        def k34():
            k3()
            ops.bucket_points(ws, depth)

        k34()
        print("%-28s %10s" % ("L1 kernel (B=%d)" % B, "us"))
        t3 = timeit(k34, args.iters)
        print("%-28s %10.1f" % ("K3+K4 (voxels,alloc,fill)", t3))
        for math, nm in ((ops.DT_F32, "f32"), (ops.DT_BF16, "bf16")):
            print("%-28s %10.1f" % ("K2 depthnet+softmax " + nm, timeit(lambda: ops.depthnet_softmax(x, w, b, D, C, math), args.iters)))
        k34()
        for lay, nm, nbytes in ((0, "NCHW f32", 4), (1, "NHWC f32", 4), (2, "NHWC bf16", 2)):
            us = timeit(lambda: ops.lift_splat_fwd(feat, ws, (B, N, D, fH, fW, C), (200, 200, 1), lay), args.iters)
            print("%-28s %10.1f   %7.0f GB/s written" % ("K5 lift-splat " + nm, us, B * 64 * 200 * 200 * nbytes / us / 1e3))
        G = torch.randn(B, 200, 200, 64, device="cuda").permute(0, 3, 1, 2)
        print("%-28s %10.1f" % ("K7 lift-splat bwd (NHWC G)", timeit(lambda: ops.lift_splat_bwd(G, ws.voxel, depth, feat, (B, N, D, fH, fW, C), (200, 200, 1)), args.iters)))
    if args.only == "vovnet":
        vovnet(args)
    if args.only == "gemm":
        gemm(args)
    if args.only == "grad":
        grad(args)


def grad(args):
    """dgrad / wgrad of the 3x3 convs at the bench batch (bf16), against the forward kernel."""
    B = args.batch
    print("%-18s %10s %10s %10s   (us; TF/s in brackets)" % ("conv 3x3 (B=%d)" % B, "fwd", "dgrad", "wgrad"))
    for name, H, W, Cin, Cout in [("layer1", 100, 100, 64, 64), ("layer2", 50, 50, 128, 128), ("layer3", 25, 25, 256, 256),
                                  ("up1.conv0", 100, 100, 320, 256), ("up1.conv3", 100, 100, 256, 256),
                                  ("up2.1", 200, 200, 256, 128)]:
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        dy = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wf, wd = ops.pack_conv_weight(w, ops.DT_BF16), ops.pack_conv_weight_dgrad(w, ops.DT_BF16)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        tf = timeit(lambda: ops.conv2d_nhwc(x, wf, (3, 3), 1, 1), args.iters)
        td = timeit(lambda: ops.conv2d_nhwc(dy, wd, (3, 3), 1, 1), args.iters)
        tw = timeit(lambda: ops.conv3x3_wgrad(x, dy), args.iters)
        print("%-18s %6.1f [%4.0f] %6.1f [%4.0f] %6.1f [%4.0f]" % (name, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6))


def gemm(args):
    """1x1 convs = token-major GEMMs of the BEV transformer (linear_mfma.hip), bf16, no epilogue extras."""
    for (B, H, W, K, N) in [(8, 200, 200, 256, 1024), (8, 200, 200, 1024, 256), (8, 200, 200, 256, 256),
                            (8, 200, 200, 256, 192), (8, 200, 200, 128, 256), (1, 128, 128, 256, 1024)]:
        x = torch.randn(B, H, W, K, device="cuda").to(torch.bfloat16)
        w = ops.pack_conv_weight(torch.randn(N, K, 1, 1, device="cuda") * 0.05, ops.DT_BF16)
        b = torch.randn(N, device="cuda")
        us = timeit(lambda: ops.conv2d_nhwc(x, w, (1, 1), 1, 0, None, b, None, False), args.iters)
        fl = 2.0 * B * H * W * K * N
        print("M=%7d K=%4d N=%4d %9.1f us %8.1f TF/s   in %4.0f MB  out %4.0f MB" %
              (B * H * W, K, N, us, fl / us / 1e6, B * H * W * K * 2 / 1e6, B * H * W * N * 2 / 1e6))


def vovnet(args):
    """BASELINE configs[3]: trunk maps (B*6,768,8,22)/(B*6,1024,4,11) -> C=128 lift-splat ->
    BEVEncoderTransformer, bf16 conv math.  Whole-model time with plain events, then one pass
    with a bracket around every native launch (each bracket adds a few us of idle)."""
    import numpy as np

    import lss2_multimodal_nu_amd as L
    from oracle import lss_oracle as lo
    B = args.batch
    grid = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
                dbound=[4.0, 45.0, 1.0])
    conf = dict(final_dim=(128, 352), Ncams=6, cams=list("abcdef"))
    torch.manual_seed(0)
    m = L.compile_model_vovnet_transformer(B, grid, conf, 4, lss_version="v2", precision="bf16").cuda().eval()
    gen = np.random.RandomState(0)
    c3 = torch.from_numpy(gen.randn(B * 6, 768, 8, 22).astype(np.float32)).cuda()
    c4 = torch.from_numpy(gen.randn(B * 6, 1024, 4, 11).astype(np.float32)).cuda()
    calib = lo.synthetic_rig(B, 6, train_aug=True, seed=0)
    dt = ops.DT_BF16

    def bev_branch():
        g = m.get_voxels(c3, c4, *calib, layout=ops.BEV_NHWC_BF16)
        return m.bev_encoder.forward_nhwc(g.permute(0, 2, 3, 1), dt)

    with torch.no_grad():
        us = timeit(bev_branch, args.iters)
        us_l1 = timeit(lambda: m.get_voxels(c3, c4, *calib, layout=ops.BEV_NHWC_BF16), args.iters)
        print("vovnet BEV branch (B=%d, bf16): %.1f us/step = %.0f frames/s; lift-splat level %.1f us"
              % (B, us, B / us * 1e6, us_l1))
        tm = ops.KernelTimer(fine=True)
        ops.set_timer(tm)
        for _ in range(5):
            bev_branch()
        torch.cuda.synchronize()
        ops.set_timer(None)
    for tag, (n, ms) in sorted(tm.totals_ms().items(), key=lambda kv: -kv[1][1]):
        print("  %-24s %3d launches/step %9.1f us/step" % (tag, n // 5, ms / 5 * 1e3))
    for tag in ("linear", "conv2d_fwd", "layernorm"):  # per launch, in issue order (mean of the 5 passes)
        sp = tm.spans.get(tag, [])
        n = len(sp) // 5
        print("  %s per launch: %s" % (tag, "  ".join(
            "%.0f" % (sum(sp[p * n + i][0].elapsed_time(sp[p * n + i][1]) for p in range(5)) / 5 * 1e3) for i in range(n))))


if __name__ == "__main__":
    main()
