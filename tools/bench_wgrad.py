#!/usr/bin/env python3
"""Weight gradients of BevEncode's 3x3 / s1 convs at batch 4: the direct kernel (csrc/conv_wgrad.hip) against the
channel-major copies + split-K GEMM path (LSS_WGRAD_DIRECT=0), interleaved rounds in one process, HIP-event timing.

    python tools/bench_wgrad.py [--rounds 5] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lss2_multimodal_nu_amd import ops  # noqa: E402

LAYERS = [  # name, B, H, W, Cin, Cout, count in BevEncode
    ("layer1 64->64 @100", 4, 100, 100, 64, 64, 4),
    ("layer2 128->128 @50", 4, 50, 50, 128, 128, 3),
    ("layer3 256->256 @25", 4, 25, 25, 256, 256, 3),
    ("up1.conv0 320->256 @100", 4, 100, 100, 320, 256, 1),
    ("up1.conv3 256->256 @100", 4, 100, 100, 256, 256, 1),
    ("up2 256->128 @200", 4, 200, 200, 256, 128, 1),
]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--direct-only", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    tot = {"direct": 0.0, "gemm": 0.0}
    for name, B, H, W, Cin, Cout, cnt in LAYERS:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        dy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
        res = {"direct": [], "gemm": []}
        for _ in range(a.rounds):
            for mode in (("direct",) if a.direct_only else ("direct", "gemm")):
                if mode == "gemm":
                    os.environ["LSS_WGRAD_DIRECT"] = "0"
                else:
                    os.environ.pop("LSS_WGRAD_DIRECT", None)
                ops._wgrad_ws.clear()
                res[mode].append(timed(lambda: ops.conv3x3_wgrad(x, dy), a.iters))
        os.environ.pop("LSS_WGRAD_DIRECT", None)
        gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
        if a.direct_only:
            res["gemm"] = res["direct"]
        d, g = sorted(res["direct"])[a.rounds // 2], sorted(res["gemm"])[a.rounds // 2]
        tot["direct"] += d * cnt
        tot["gemm"] += g * cnt
        print("%-26s direct %7.1f us %6.0f TF   copies + GEMM %7.1f us %6.0f TF   (x%d per step)"
              % (name, d, gf / d * 1e3, g, gf / g * 1e3, cnt), flush=True)
    print("per training step (13 layers): direct %.0f us, copies + GEMM %.0f us; wgrad timeouts: %d"
          % (tot["direct"], tot["gemm"], ops.N.lib().lss_conv2d_wgrad_timeouts()))


if __name__ == "__main__":
    main()
