#!/usr/bin/env python3
"""What does one kernel cost as a node of a replayed HIP graph, against the same kernel launched into a stream from a
C loop?  A dependent chain of N launches of the K-split convolution kernel (layer3 geometry, ~7 us of work) and of a
trivial kernel (a 1-KiB fill) is timed three ways with HIP events:
  stream   the recorded launch list (ops.ConvPlan: one C loop of hipLaunchKernel calls), no graph
  graph    the same launches captured once in a torch.cuda.CUDAGraph, replayed
  eager    (fill only) N torch calls from Python: host-bound, for scale
python tools/graph_node_cost.py [--chain 40] [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lss2_multimodal_nu_amd import ops  # noqa: E402


def timed(fn, rounds, n):
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) * 1e3 / n)
    return min(out), sorted(out)[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    torch.manual_seed(0)
    B, H, W, C = 4, 25, 25, 256
    x = torch.randn(B, H, W, C, device="cuda").bfloat16()
    r = torch.randn(B, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    wk = ops.pack_conv_weight_ks(w)
    rec = ops.ConvRecorder()
    ops.set_recorder(rec)
    y = x
    for _ in range(a.chain):
        y = ops.conv2d_nhwc(y, wk, (3, 3), 1, 1, sc, sh, r, True, None, 1, None, 1)
    ops.set_recorder(None)
    plan = ops.ConvPlan(rec, x, y)
    for _ in range(3):
        plan.run(x, y)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        plan.run(x, y)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            plan.run(x, y)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("K-split conv, layer3 geometry, chain of %d:  stream %.2f us / launch (median %.2f)   graph node %.2f us (median %.2f)"
          % ((a.chain,) + timed(lambda: plan.run(x, y), a.rounds, a.chain) + timed(g.replay, a.rounds, a.chain)))

    t = torch.zeros(256, device="cuda")
    n = 200
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(3):
            t.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=side):
            for _ in range(n):
                t.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()

    def eager():
        for _ in range(n):
            t.add_(1.0)
    print("1-KiB add_, chain of %d:  graph node %.2f us (median %.2f)   eager from Python %.2f us (median %.2f)"
          % ((n,) + timed(g2.replay, a.rounds, n) + timed(eager, a.rounds, n)))


if __name__ == "__main__":
    main()
