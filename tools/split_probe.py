#!/usr/bin/env python3
"""Probe: one batch of B frames as ONE forward vs as two half-batch forwards on two HIP streams (fork / join events
per step included).   python tools/split_probe.py [--batch 4] [--steps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lss2_multimodal_nu_amd as L  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402  (input rig only)

GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--parts", type=int, default=2)
    args = ap.parse_args()
    B, P = args.batch, args.parts
    torch.manual_seed(0)
    full = L.compile_model_lss(B, GRID, AUG, 4).cuda().eval()
    parts = [L.compile_model_lss(B // P, GRID, AUG, 4).cuda().eval() for _ in range(P)]
    for m in parts:
        m.load_state_dict(full.state_dict())
    x = torch.randn(B * 6, 512, 8, 22, device="cuda")
    calib = lo.synthetic_rig(B, final_dim=AUG["final_dim"], train_aug=True, seed=0)
    h = B // P
    xs = [x[i * h * 6:(i + 1) * h * 6].contiguous() for i in range(P)]
    cs = [[t[i * h:(i + 1) * h].contiguous() for t in calib] for i in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]

    def one():
        return full(x, *calib)

    def split():
        main_s = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main_s)
        outs = []
        for i in range(P):
            streams[i].wait_event(fork)
            with torch.cuda.stream(streams[i]):
                outs.append(parts[i](xs[i], *cs[i]))
            e = torch.cuda.Event()
            e.record(streams[i])
            main_s.wait_event(e)
        return outs

    with torch.no_grad():
        ref = one()
        got = torch.cat(split(), 0)
        torch.cuda.synchronize()
        print("max |split - one| = %.3e (max |one| %.3e)" % (float((got - ref).abs().max()), float(ref.abs().max())))
        for name, fn in (("one forward", one), ("%d part streams" % P, split), ("one forward", one), ("%d part streams" % P, split)):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%-16s %.4f ms / step  %.0f frames/s" % (name, dt / args.steps * 1e3, args.steps * B / dt))


if __name__ == "__main__":
    main()
