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
    ap.add_argument("--stamps", action="store_true", help="s_memrealtime phase stamps of the fused K2 || K3 launch")
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

    if args.stamps:
        import numpy as np
        n2 = -(-fH * fW // 16) * B * 6
        with torch.no_grad():
            for _ in range(5):
                step()
            buf = torch.zeros(2 * 16384 * 8, dtype=torch.int64, device="cuda")
            os.environ["LSS_L1_STAMPS"] = "%x" % buf.data_ptr()
            for _ in range(3):
                buf.zero_()
                torch.cuda.synchronize()
                step()
                torch.cuda.synchronize()
            del os.environ["LSS_L1_STAMPS"]
        raw = buf.view(-1, 8).cpu().numpy()
        sp = raw[16384:]
        sp = sp[sp[:, 0] != 0]
        t = raw[:16384].astype(np.float64) * 0.01  # us
        live = t[:, 0] != 0
        t0 = t[live, 0].min()
        ids = np.arange(len(t))
        f = lambda v: "p50 %5.2f  p90 %5.2f  max %5.2f" % (np.median(v), np.percentile(v, 90), v.max())
        for name, sel in (("K2 depth rows", live & (ids < 2 * n2) & (ids % 2 == 0)),
                          ("K2 context rows", live & (ids < 2 * n2) & (ids % 2 == 1)), ("K3 geometry", live & (ids >= 2 * n2))):
            q = t[sel]
            print("%-16s %5d workgroups" % (name, len(q)))
            print("   start after launch   %s" % f(q[:, 0] - t0))
            if name != "K3 geometry":
                print("   first K block landed %s" % f(q[:, 1] - q[:, 0]))
                print("   K loop               %s" % f(q[:, 2] - q[:, 1]))
                print("   epilogue             %s" % f(q[:, 3] - q[:, 2]))
            print("   whole workgroup      %s" % f(q[:, 3] - q[:, 0]))
            print("   end after launch     %s" % f(q[:, 3] - t0))
            if name != "K3 geometry":   # who the stragglers are: launch ids (id & 7 = XCD) of the slowest first operands
                w = np.nonzero(sel)[0]
                order = np.argsort(-(q[:, 1] - q[:, 0]))[:8]
                print("   slowest first blocks: " + ", ".join("id %d (xcd %d) %.1f us" % (w[o], w[o] & 7, q[o, 1] - q[o, 0]) for o in order))
        # region splat: one workgroup per region
        q = sp[:, :4].astype(np.float64) * 0.01
        npts = sp[:, 4]
        s0 = q[:, 0].min()
        print("region splat     %5d workgroups, %d points, busiest region %d points, %d empty regions" % (
            len(q), npts.sum(), npts.max(), (npts == 0).sum()))
        print("   start after launch   %s" % f(q[:, 0] - s0))
        ne = npts > 0
        print("   clear tile (n > 0)   %s" % f(q[ne, 1] - q[ne, 0]))
        print("   accumulate (n > 0)   %s" % f(q[ne, 2] - q[ne, 1]))
        print("   convert + store      %s" % f(np.where(ne, q[:, 3] - q[:, 2], q[:, 3] - q[:, 0])))
        print("   whole workgroup      %s" % f(q[:, 3] - q[:, 0]))
        print("   end after launch     %s" % f(q[:, 3] - s0))
        order = np.argsort(-npts)[:5]
        print("   five busiest regions: points %s, accumulate us %s, start us %s" % (
            npts[order].tolist(), np.round(q[order, 2] - q[order, 1], 2).tolist(), np.round(q[order, 0] - s0, 2).tolist()))
        return
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
