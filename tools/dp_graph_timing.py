#!/usr/bin/env python3
"""Where a data-parallel graphed training step spends its time: two gloo ranks on ONE GPU (the rehearsal setup of
bench.py --gpus 2 under LSS_BENCH_REHEARSE=1), dp.GraphedTrainStep at batch 4 per rank; per phase - input refresh,
graph A replay, bucket all-reduces, graph B replay - the host time to the next synchronisation.
    python tools/dp_graph_timing.py [--steps 5]"""
import argparse
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0], dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}


def worker(rank, world, port, steps):
    import torch.distributed as dist

    import lss2_multimodal_nu_amd as L
    from lss2_multimodal_nu_amd import dp
    from oracle import lss_oracle as lo
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    B = 4
    m = L.compile_model_lss(B, GRID, AUG, 4).cuda().train()
    bucket = dp.make_bucket(m)
    opt = L.ClipAdam(bucket.params, lr=1e-4)
    x = torch.randn(6 * B, 512, 8, 22).cuda()
    tgt = torch.randint(0, 4, (B, 200, 200)).cuda()
    calib = lo.synthetic_rig(B, 6, train_aug=True, seed=rank)

    class Amp(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.inner.forward_loss(*a, tgt)

    gs = dp.GraphedTrainStep(Amp(m), bucket, opt, lambda l: l, x, tuple(calib), warmup=3)
    sync = torch.cuda.synchronize
    for it in range(steps):
        t = [time.perf_counter()]
        gs.feats.copy_(x, non_blocking=True); sync(); t.append(time.perf_counter())
        gs.graph.replay(); sync(); t.append(time.perf_counter())
        bucket.all_reduce_all(); sync(); t.append(time.perf_counter())
        gs.graph_b.replay(); sync(); t.append(time.perf_counter())
        if rank == 0:
            print("step %d: refresh %.2f ms | graph A %.2f ms | all-reduce (3 buckets, gloo) %.2f ms | graph B %.2f ms"
                  % ((it,) + tuple((b - a) * 1e3 for a, b in zip(t, t[1:]))), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, port, a.steps), nprocs=2, join=True)
