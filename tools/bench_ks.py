#!/usr/bin/env python3
"""The launch-bound 3x3 / stride-1 layers of BevEncode (layer1-3, batch 4) on the tile kernel (conv_mfma.hip) and on the
K-split one-pass kernel (conv_ks.hip), back to back inside ONE recorded launch list per kernel so that the host is out
of the picture (HIP-event timing of 40 launches per bracket).   python tools/bench_ks.py [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lss2_multimodal_nu_amd import ops  # noqa: E402

LAYERS = [("layer1 64->64 @100", 4, 100, 100, 64), ("layer2 128->128 @50", 4, 50, 50, 128), ("layer3 256->256 @25", 4, 25, 25, 256)]


def stamps():
    import numpy as np
    torch.manual_seed(0)
    print("%-22s %4s | %s" % ("K-split kernel (B=4)", "WGs", "start-spread  issue  landed  main  reduce  epilogue  drain | first->last us (p50 / max)"))
    for name, B, H, W, C in LAYERS:
        x = torch.randn(B, H, W, C, device="cuda").bfloat16()
        r = torch.randn(B, H, W, C, device="cuda").bfloat16()
        w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        wk = ops.pack_conv_weight_ks(w)
        for _ in range(5):
            ops.conv2d_nhwc(x, wk, (3, 3), 1, 1, sc, sh, r, True, None, 1, None, 1)
        buf = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
        os.environ["LSS_KS_STAMPS"] = "%x" % buf.data_ptr()
        torch.cuda.synchronize()
        ops.conv2d_nhwc(x, wk, (3, 3), 1, 1, sc, sh, r, True, None, 1, None, 1)
        torch.cuda.synchronize()
        del os.environ["LSS_KS_STAMPS"]
        t = buf.view(-1, 8).cpu().numpy().astype(np.float64) * 0.01
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        f = lambda v: "%5.2f/%5.2f" % (np.median(v), v.max())  # noqa: E731
        print("%-22s %4d | %s  %s  %s  %s  %s  %s  %s | %s" % (
            name, len(t), f(t[:, 0] - t0), f(t[:, 1] - t[:, 0]), f(t[:, 2] - t[:, 1]), f(t[:, 3] - t[:, 2]),
            f(t[:, 4] - t[:, 3]), f(t[:, 5] - t[:, 4]), f(t[:, 6] - t[:, 5]), f(t[:, 6] - t[:, 0])))
        clk = t[:, 7] * 100.0  # slot 7: s_memtime ticks over the main phase
        print("%-22s        main phase: %.0f s_memtime ticks (median) = %.1f per MFMA of a wave; ticks per us of s_memrealtime %.0f"
              % ("", np.median(clk), np.median(clk) / 180.0, np.median(clk / np.maximum(t[:, 3] - t[:, 2], 1e-3))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--chain", type=int, default=40)
    ap.add_argument("--stamps", action="store_true", help="in-kernel phase stamps of the K-split kernel (one launch per layer)")
    a = ap.parse_args()
    if a.stamps:
        return stamps()
    torch.manual_seed(0)
    res = {}
    plans = []
    for name, B, H, W, C in LAYERS:
        x = torch.randn(B, H, W, C, device="cuda").bfloat16()
        r = torch.randn(B, H, W, C, device="cuda").bfloat16()
        w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        for tag, wp in (("tile", ops.pack_conv_weight(w, 1)), ("ks", ops.pack_conv_weight_ks(w))):
            rec = ops.ConvRecorder()
            ops.set_recorder(rec)
            y = x
            for _ in range(a.chain):   # a dependent chain, like the network: each launch reads the previous one's output
                y = ops.conv2d_nhwc(y, wp, (3, 3), 1, 1, sc, sh, r, True, None, 1, None, 1)
            ops.set_recorder(None)
            plans.append((name, tag, ops.ConvPlan(rec, x, y), x, y, 2.0 * B * H * W * C * C * 9))
    for _ in range(3):
        for name, tag, plan, x, y, fl in plans:
            plan.run(x, y)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for name, tag, plan, x, y, fl in plans:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            plan.run(x, y)
            e.record()
            torch.cuda.synchronize()
            res.setdefault((name, tag), []).append(s.elapsed_time(e) * 1e3 / a.chain)
    for name, B, H, W, C in LAYERS:
        t, k = sorted(res[(name, "tile")]), sorted(res[(name, "ks")])
        print("%-22s tile %6.2f us (min %6.2f)   ks %6.2f us (min %6.2f)   per launch incl. its boundary"
              % (name, t[len(t) // 2], t[0], k[len(k) // 2], k[0]))


if __name__ == "__main__":
    main()
