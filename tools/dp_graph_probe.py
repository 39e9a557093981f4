#!/usr/bin/env python3
"""Bisect probe for the data-parallel graph step (dp.GraphedTrainStep with a bucket over 2 ranks on one GPU, gloo):
per-step losses of variants against the eager step.  python tools/dp_graph_probe.py [batch]"""
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


def worker(rank, world, port, batch, variant, q):
    import torch.distributed as dist

    import lss2_multimodal_nu_amd as L
    from lss2_multimodal_nu_amd import dp, ops
    from lss2_multimodal_nu_amd.tools import weighted_cross_entropy
    from oracle import lss_oracle as lo
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    m = L.compile_model_lss(batch, GRID, AUG, 4).cuda().train()
    bucket = dp.GradBucket(m.parameters())
    opt = torch.optim.Adam(bucket.params, lr=1e-3, capturable=True)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(6 * batch, 512, 8, 22, generator=g).cuda()
    tgt = torch.randint(0, 4, (batch, 200, 200), generator=g).cuda()
    w = torch.tensor([1.0, 10.0, 5.0, 10.0]).cuda()
    calib = lo.synthetic_rig(batch, 6, train_aug=True, seed=rank)

    class Amp(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if os.environ.get("PROBE_FUSED_LOSS") == "1":   # the fused head + cross-entropy entry (no library conv)
                    return self.inner.forward_loss(*a, tgt)
                return self.inner(*a)

    wrapped = Amp(m)
    if os.environ.get("PROBE_FUSED_LOSS") == "1":
        loss_fn = lambda l: l  # noqa: E731
    else:
        loss_fn = lambda y: weighted_cross_entropy(y.float(), tgt, w)  # noqa: E731
    losses, norms = [], []
    if variant == "eager":
        for _ in range(11):
            losses.append(float(dp.train_step(wrapped, bucket, opt, loss_fn, (x,) + tuple(calib))))
            norms.append(float(dp._last_norm[0]))
    else:
        if variant == "graph_diag":
            dp.GraphedTrainStep.CHECK_REPLAYS = 7
            os.environ["LSS_GRAPH_DEBUG"] = "1"
        gs = dp.GraphedTrainStep(wrapped, bucket, opt, loss_fn, x, tuple(calib), warmup=2)
        for i in range(6):
            if variant == "graph_inspect":
                # graph A alone, then a look at the gradients it left (pre-reduce, pre-clip), then the rest of the step
                gs.graph.replay()
                torch.cuda.synchronize()
                rows = []
                for n, p in m.named_parameters():
                    if p.grad is not None:
                        g_ = p.grad.float()
                        rows.append((float(g_.abs().max()) if bool(torch.isfinite(g_).all()) else float("inf"), n))
                rows.sort(reverse=True)
                if rank == 0:
                    print("  replay %d: largest |grad| by parameter: %s" % (i, ", ".join("%s %.3g" % (n, v) for v, n in rows[:4])), flush=True)
                if gs.graph_b is not None:
                    bucket.all_reduce_all()
                    gs.graph_b.replay()
                l = gs.loss
            elif variant == "graph_norefresh":
                gs._replay()
                l = gs.loss
            elif variant == "graph_replaysync":   # exactly what _self_check does, after __init__ has returned
                gs._replay()
                torch.cuda.synchronize()
                l = gs.loss
            elif variant == "graph_diag":
                gs._replay()
                torch.cuda.synchronize()
                print("  replay %d: static norm %.5g  norm of the flat buffer now %.5g  loss %.5f  adam step %s" % (
                    i, float(gs.grad_norm), float(torch.linalg.vector_norm(bucket.flat)), float(gs.loss),
                    float(opt.state[bucket.params[0]]["step"])), flush=True)
                l = gs.loss
            elif variant == "graph_presync":      # a device synchronisation BEFORE the replay instead
                torch.cuda.synchronize()
                gs._replay()
                l = gs.loss
            else:
                l = gs(x, tuple(calib))
            if variant == "graph_sync":
                torch.cuda.synchronize()
            losses.append(float(l))
            norms.append(float(gs.grad_norm))
    torch.cuda.synchronize()
    c = ops.timeout_counters()
    if rank == 0:
        q.put((variant, losses, norms, c))
    if world > 1:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


if __name__ == "__main__":
    import torch.multiprocessing as mp
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    worlds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 1]
    ctx = mp.get_context("spawn")
    for world in worlds:
        for variant in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("eager", "graph", "graph_sync", "graph_norefresh")):
            q = ctx.SimpleQueue()
            mp.spawn(worker, args=(world, free_port(), batch, variant, q), nprocs=world, join=True)
            v, losses, norms, c = q.get()
            print("world %d batch %d %-16s losses %s\n%40s norms %s  %s" % (
                world, batch, v, " ".join("%.4f" % l for l in losses), "", " ".join("%.3f" % n for n in norms), c), flush=True)
