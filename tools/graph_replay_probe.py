#!/usr/bin/env python3
"""Stress the captured training step (dp.GraphedTrainStep): build, capture, then replay N times and look at every
parameter gradient after every replay - reports the first replay (and tensor) whose gradient is non-finite or wild.

    python tools/graph_replay_probe.py [bucket|local] [replays]      (LSS_S2_DGRAD_GEMM=1 etc. select variants)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lss2_multimodal_nu_amd as L  # noqa: E402
from lss2_multimodal_nu_amd import dp, ops  # noqa: E402
from oracle import lss_oracle as lo  # noqa: E402

GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


def main():
    bucketed = len(sys.argv) > 1 and sys.argv[1] == "bucket"
    nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    B = int(os.environ.get("PROBE_BATCH", "1"))
    tgt = torch.randint(0, 4, (B, 200, 200), generator=torch.Generator().manual_seed(9)).to(dev)
    g = torch.Generator().manual_seed(40)
    x = torch.randn(B * 6, 512, 8, 22, generator=g).to(dev)
    calib = lo.synthetic_rig(B, 6, train_aug=True, seed=0)
    torch.manual_seed(0)
    m = L.compile_model_lss(B, GRID, AUG, 4, precision="bf16").to(dev).train()
    # an eager model first, like a test session that ran other things before (fills the per-shape workspace caches)
    bucket = dp.make_bucket(m) if bucketed else None
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True, fused=not bucketed)

    class Amp(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.inner.forward_loss(*a, tgt)

    dp.GraphedTrainStep._self_check = lambda self: None
    gs = dp.GraphedTrainStep(Amp(m), bucket, opt, lambda l: l, x, tuple(calib), warmup=3)
    for i in range(nrep):
        gs.graph.replay()
        torch.cuda.synchronize()
        bad = [(n, float(p.grad.abs().max())) for n, p in m.named_parameters()
               if p.grad is not None and not (float(p.grad.abs().max()) < 1e3)]
        if bad:
            print("BAD at replay", i, bad[:6], len(bad), "loss", float(gs.loss), flush=True)
            return 1
    print("clean:", nrep, "replays, loss %.5f" % float(gs.loss), "wgrad / ring timeouts",
          ops.N.lib().lss_conv2d_wgrad_timeouts(), ops.N.lib().lss_conv2d_ring_timeouts(), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
