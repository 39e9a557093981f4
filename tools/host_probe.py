#!/usr/bin/env python3
"""Host enqueue time per step vs GPU completion over a 20-step region that starts from a synchronised (idle) queue:
what a short timed region pays at its two ends.   python tools/host_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lss2_multimodal_nu_amd as L
from lss2_multimodal_nu_amd import ops
from oracle import lss_oracle as lo
GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}
m = L.compile_model_lss(4, GRID, AUG, 4).cuda().eval()
x = torch.randn(24, 512, 8, 22, device="cuda")
calib = lo.synthetic_rig(4, final_dim=AUG["final_dim"], train_aug=True, seed=0)
with torch.no_grad():
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end: m(x, *calib)
    for timed in (False, True, True):
        torch.cuda.synchronize()
        if timed:
            ops.set_timer(ops.KernelTimer())
        ts = [time.perf_counter()]
        for _ in range(20):
            m(x, *calib); ts.append(time.perf_counter())
        t_enq = ts[-1]
        torch.cuda.synchronize(); t_done = time.perf_counter()
        ops.set_timer(None)
        d = [(b - a) * 1e6 for a, b in zip(ts, ts[1:])]
        print("timer" if timed else "plain", "host us/step first5", [round(v) for v in d[:5]], "mean rest", round(sum(d[5:]) / 15),
              "enqueue done at %.0f us, gpu done at %.0f us -> %.1f us/step" % ((t_enq - ts[0]) * 1e6, (t_done - ts[0]) * 1e6, (t_done - ts[0]) * 1e6 / 20))
