#!/usr/bin/env python3
"""Lift-splat level (fused inference entry: K2 || K3 -> fill -> region splat) timed with HIP events, per
pipeline / diagnostic mode.   python tools/bench_l1.py [--batch 4] [--iters 50] [--hires]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lss2_multimodal_nu_amd as L  # noqa: E402
from lss2_multimodal_nu_amd import ops  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402  (input rig only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--hires", action="store_true")
    ap.add_argument("--modes", default="legacy,0")
    args = ap.parse_args()
    B = args.batch
    grid = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
    aug, fH, fW = {"final_dim": (128, 352), "Ncams": 6}, 8, 22
    if args.hires:
        grid = dict(xbound=[-50.0, 50.0, 0.25], ybound=[-50.0, 50.0, 0.25], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 61.0, 1.0])
        aug, fH, fW = {"final_dim": (256, 704), "Ncams": 6}, 16, 44
    torch.manual_seed(0)
    m = L.compile_model_lss(B, grid, aug, 4).cuda().eval()
    x = torch.randn(B * 6, 512, fH, fW, device="cuda")
    calib = lo.synthetic_rig(B, final_dim=aug["final_dim"], train_aug=True, seed=0)

    def step():
        return m._lift_splat(x, *calib, ops.BEV_NHWC_BF16)

    for mode in args.modes.split(","):
        os.environ.pop("LSS_SPLAT_LEGACY", None)
        os.environ.pop("LSS_RS_DBG", None)
        if mode == "legacy":
            os.environ["LSS_SPLAT_LEGACY"] = "1"
        else:
            os.environ["LSS_RS_DBG"] = mode
        with torch.no_grad():
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                step()
            e.record()
            torch.cuda.synchronize()
        print("mode %-7s %8.1f us / call" % (mode, s.elapsed_time(e) / args.iters * 1e3))


if __name__ == "__main__":
    main()
