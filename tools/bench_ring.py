#!/usr/bin/env python3
"""The three big BevEncode convs (batch 4) on the tile kernel (conv_mfma.hip) and on the loader / consumer ring kernel
(conv_ring.hip), interleaved rounds in ONE process (cdna_hip_programming.md section 5.4 rule 24), HIP-event timing.

    python tools/bench_ring.py [--rounds 7] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lss2_multimodal_nu_amd import ops  # noqa: E402

LAYERS = [  # name, B, H, W, Cx, Cout, C2, up, head_n
    ("up1.conv0 320->256 @100 (x4 + cat)", 4, 25, 25, 256, 256, 64, 4, 0),
    ("up1.conv3 256->256 @100", 4, 100, 100, 256, 256, 0, 1, 0),
    ("up2 256->128 @200 (x2) + head", 4, 100, 100, 256, 128, 0, 2, 4),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(0)
    cases = []
    for name, B, H, W, Cx, Cout, C2, up, hn in LAYERS:
        x = torch.randn(B, H, W, Cx, device="cuda").bfloat16()
        x2 = torch.randn(B, H * up, W * up, C2, device="cuda").bfloat16() if C2 else None
        w = torch.randn(Cout, Cx + C2, 3, 3, device="cuda") * ((Cx + C2) * 9) ** -0.5
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
        wt, wr = ops.pack_conv_weight(w, 1), ops.pack_conv_weight_ring(w)
        hw, hb = torch.randn(max(hn, 1), Cout, device="cuda") * Cout ** -0.5, torch.randn(max(hn, 1), device="cuda")
        flops = 2.0 * B * (H * up) * (W * up) * Cout * (Cx + C2) * 9

        def run(wp, x=x, x2=x2, sc=sc, sh=sh, hw=hw, hb=hb, hn=hn, up=up):
            if hn:
                return ops.conv3x3_head_nchw(x, wp, sc, sh, hw, hb, x2=x2, up=up)
            return ops.conv2d_nhwc(x, wp, (3, 3), 1, 1, sc, sh, None, True, x2, up, None, 1)
        cases.append((name, flops, run, wt, wr))
    for _ in range(30):  # clocks up
        for _, _, run, wt, wr in cases:
            run(wt); run(wr)
    torch.cuda.synchronize()
    res = {}
    for r in range(a.rounds):
        for name, flops, run, wt, wr in cases:
            for tag, wp in (("tile", wt), ("ring", wr)):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    run(wp)
                e.record()
                torch.cuda.synchronize()
                res.setdefault((name, tag), []).append(s.elapsed_time(e) * 1e3 / a.iters)
    tot = {"tile": 0.0, "ring": 0.0}
    for name, flops, *_ in cases:
        line = "%-40s" % name
        for tag in ("tile", "ring"):
            v = sorted(res[(name, tag)])
            med = v[len(v) // 2]
            tot[tag] += med
            line += "  %s %7.1f us (min %7.1f) %6.0f TF" % (tag, med, v[0], flops / med / 1e6)
        print(line)
    print("sum of medians: tile %.1f us, ring %.1f us; ring timeouts: %d"
          % (tot["tile"], tot["ring"], ops.N.lib().lss_conv2d_ring_timeouts()))


if __name__ == "__main__":
    main()
