#!/usr/bin/env python3
"""bench.py - BEV frames/sec of the camera->BEV hot path on N MI355X.

One "step" = one forward pass of the whole hot path (K3 points->voxels, K4
bucketing, K2 depthnet+softmax, K5 fused lift-splat, K8 BevEncode convs) over
one batch of synthetic trunk features + calibrations; one frame = one sample =
6 cameras.  Workload = BASELINE.json configs[1]: batch 4, 6 cams 352x128
(trunk features 8x22x512), D=41, 200x200x64 BEV, bf16 conv path.
Multi-GPU: one process per GPU, samples sharded across ranks (data parallel,
no data-path collective on the inference path) -> weak scaling.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...      # bare: spawns N ranks itself (torch.distributed.run child, before any GPU call)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # or launched by a driver

    --workload config1|config2|hires  = BASELINE configs[0] shapes (B=1) | configs[1] (default, the metric's
                                        workload; also the per-GPU shape of configs[2]) | configs[4] per-GPU shapes

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel =
the MFMA convs), `roofline_l1` (the HBM-bound lift-splat kernel), `levels`
(L1 = lift-splat only, L2 = full hot path), `cpu_baseline` (the CPU oracle
timed on this host), `train` (fwd+bwd+Adam step with RCCL gradient all-reduce).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRID = dict(xbound=[-50.0, 50.0, 0.5], ybound=[-50.0, 50.0, 0.5], zbound=[-10.0, 10.0, 20.0],
            dbound=[4.0, 45.0, 1.0])
AUG = {"final_dim": (128, 352), "Ncams": 6}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_TFLOPS = 2500.0   # dense bf16
MFMA_F32_TFLOPS = 157.3


def bevencode_flops(X, Y, inC=64, outC=4):
    """Forward FLOPs/frame of BevEncode (2*MACs), SURVEY.md 8a-9: 62.38 G at 200x200."""
    def conv(h, w, cin, cout, k):
        return 2.0 * h * w * cin * cout * k * k
    h1, w1 = X // 2, Y // 2
    h2, w2 = (h1 + 1) // 2, (w1 + 1) // 2
    h3, w3 = (h2 + 1) // 2, (w2 + 1) // 2
    f = conv(h1, w1, inC, 64, 7)
    f += 4 * conv(h1, w1, 64, 64, 3)
    f += conv(h2, w2, 64, 128, 3) + 3 * conv(h2, w2, 128, 128, 3) + conv(h2, w2, 64, 128, 1)
    f += conv(h3, w3, 128, 256, 3) + 3 * conv(h3, w3, 256, 256, 3) + conv(h3, w3, 128, 256, 1)
    f += conv(h1, w1, 320, 256, 3) + conv(h1, w1, 256, 256, 3)
    f += conv(X, Y, 256, 128, 3) + conv(X, Y, 128, outC, 1)
    return f


def l1_bytes_per_frame(N, fH, fW, C, X, Y, Z, D, out_bytes):
    """Algorithmic HBM bytes/frame of the lift-splat level: read the
    trunk features once + write the BEV grid once (+ depthnet W, frustum, calib)."""
    return N * 512 * fH * fW * 4 + C * Z * X * Y * out_bytes


def pmc_traffic(kernel_prefix, grid=None):
    """(HBM bytes per launch, source file) from the committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic.json, made by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this same command); (None, None) when no such file exists.  bench.py itself cannot run
    the profiler."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for k, v in d.items():
        name, g = k.split("|")
        if name.startswith(kernel_prefix) and (grid is None or g == str(grid)):
            tot += (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"]
    return (tot / n, os.path.basename(files[-1])) if n else (None, None)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None):
    """The command a bare `python bench.py --gpus N` turns itself into: one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """Bare multi-GPU invocation: this process has NOT touched the GPU (torch is imported, nothing else), so it
    may start the ranks as children and relay their output; it never re-executes itself in place."""
    import subprocess
    cmd = launch_command(args.gpus, sys.argv[1:])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for RCCL on this driver stack
    env.setdefault("OMP_NUM_THREADS", "2")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def dry_launch(world, rank):
    """`--dry-launch`: rendezvous + one all-reduce over gloo on the CPU, then the JSON line - the launcher and
    the rank plumbing without a GPU (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": int(dist.get_world_size()),
                          "rank_sum": float(t[0]), "backend": dist.get_backend()}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "hires"],
                    help="config1 = BASELINE configs[0] shapes (model_baseline LSS, 1 sample) on the GPU; config2 = "
                         "configs[1] (default, the metric's workload, = the per-GPU shape of configs[2]); hires = "
                         "configs[4] per-GPU shapes (6 x 704x256, D=60, 400x400 BEV, batch 2/GPU)")
    ap.add_argument("--dry-launch", action="store_true", help="launcher / rendezvous check on the CPU (gloo), no GPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--two-streams", action=argparse.BooleanOptionalAction, default=True,
                    help="extra informational leg (levels.two_batches_in_flight_fps): the same steps with two batches in "
                         "flight on two HIP streams; `value` stays the single-stream loop")
    ap.add_argument("--train-steps", type=int, default=16)
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="keep the GPU under load this long before the warm-up steps (DVFS ramp; 0 = off)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_launch:
        return dry_launch(world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # The per-step host work of the GPU legs is a handful of 3x3 inverses and one pinned copy: it
    # gains nothing from an intra-op pool, and with one rank per GPU on a shared node 8 x 16
    # spinning OpenMP threads would contend for the same cores.  The cpu_baseline leg sets its own
    # (full) thread count.
    torch.set_num_threads(min(2, host_cores()))
    # rehearsal hook for a 1-GPU box: LSS_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo
    rehearse = os.environ.get("LSS_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import lss2_multimodal_nu_amd as L
    from lss2_multimodal_nu_amd import ops
    from oracle import lss_oracle as lo  # input generator + cpu_baseline leg only

    B = args.batch
    grid, aug, fH, fW, X, Y = GRID, AUG, 8, 22, 200, 200
    compile_lss = L.compile_model_lss
    if args.workload == "config1":
        from lss2_multimodal_nu_amd import model_baseline  # BASELINE configs[0] names model_baseline's LSS
        compile_lss = model_baseline.compile_model_lss
        B = 1 if args.batch == 4 else args.batch
    if args.workload == "hires":
        grid = dict(xbound=[-50.0, 50.0, 0.25], ybound=[-50.0, 50.0, 0.25], zbound=[-10.0, 10.0, 20.0],
                    dbound=[1.0, 61.0, 1.0])
        aug, fH, fW, X, Y = {"final_dim": (256, 704), "Ncams": 6}, 16, 44, 400, 400
        B = 2 if args.batch == 4 else args.batch
    torch.manual_seed(0)
    model = compile_lss(B, grid, aug, 4, precision=args.precision).to(dev).eval()
    D, C = model.D, model.camC
    Z = 1
    # per-rank shard of the global batch: different samples (seeded by rank), same shapes
    g = torch.Generator().manual_seed(1234 + rank)
    feats = torch.randn(B * 6, 512, fH, fW, generator=g).to(dev)
    calib = lo.synthetic_rig(B, final_dim=aug["final_dim"], train_aug=True, seed=rank)  # CPU tensors, as a DataLoader delivers them

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return model(feats, *calib)

    def step_l1():
        return model._lift_splat(feats, *calib, ops.BEV_NHWC_BF16 if args.precision == "bf16" else ops.BEV_NHWC_F32)

    with torch.no_grad():
        # Clock spin-up (stated in the line as config.spinup_ms): the GPU's power management needs some tens of
        # milliseconds under load to reach the clock it then sustains; measured on MI355X, the same kernels run 4 %
        # slower in a 20-step (10 ms) run that starts from an idle device than in a 400-step one (conv group 458 vs
        # 435 us).  The W warm-up steps and the K timed steps below are untouched.
        # The driver's nominal protocol first, from a device that has only seen the W warm-up steps: reported as
        # levels.cold_start_fps (both numbers in the line).
        for _ in range(args.warmup):
            step()
        barrier()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_cold = time.perf_counter() - tc
        t_spin = time.perf_counter() + args.spinup_ms * 1e-3
        while time.perf_counter() < t_spin:
            step()
        for _ in range(args.warmup):
            step()
        # ---- timed region: EXACTLY args.steps steps, NO instrumentation inside ------------------
        # (round 2 kept two HIP-event brackets per step in here: the trace showed 10-17 us of idle at each, ~4 % of
        # the step - it was timing the probe.  The roofline brackets now run in their own pass below.)
        ops.set_timer(None)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        # ---- roofline pass: the same K steps with ONE HIP-event bracket around the lift-splat level and one around
        # the BevEncode launch group (outside the timed region) -------------------------------------------------
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        for _ in range(args.steps):
            step()
        barrier()
        ops.set_timer(None)
        spans = timer.totals_ms()
        # ---- the dominant kernel on its own (its own pass: a bracket costs ~10 us of idle) ----
        # Same steps again with BevEncode's recorded launch list replayed in two calls, so that its LAST launch - up2:
        # x2 upsample + 3x3 conv 256 -> 128 + BN + ReLU + fused 1x1 head, 38 % of the step's FLOPs - gets a HIP-event
        # bracket of its own; profiles/r02_bench_kernel_summary.txt holds the same kernel's rocprofv3 average.
        timer_k = ops.KernelTimer(last_conv=True)
        ops.set_timer(timer_k)
        for _ in range(args.steps):
            step()
        barrier()
        ops.set_timer(None)
        n_dom, ms_dom = timer_k.totals_ms().get("conv_plan_last", (0, 0.0))
        # ---- L1 only (lift-splat level), same protocol ---------------------------------
        for _ in range(3):
            step_l1()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_l1()
        barrier()
        dt_l1 = time.perf_counter() - t1

    guards = {"inference": read_guards()}
    tmax = torch.tensor([dt, dt_l1, dt_cold], dtype=torch.float64, device=dev)
    per_rank = [args.steps * B / dt]
    comm = {"backend": "none", "ranks": 1}
    if dist is not None:
        mine = torch.tensor([args.steps * B / dt], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t[0]) for t in every]
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)  # a real collective over the communicator: counts the ranks that answered
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        comm = {"backend": dist.get_backend(), "ranks": int(ones[0]),
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if not rehearse else None,
                "devices": "all ranks on cuda:0 (LSS_BENCH_REHEARSE)" if rehearse else "one GPU per rank (LOCAL_RANK)"}
    dt, dt_l1, dt_cold = float(tmax[0]), float(tmax[1]), float(tmax[2])
    frames = args.steps * B * world
    fps = frames / dt

    # ---- roofline of the dominant kernel(s), from the HIP-event brackets -----------------
    n_conv, ms_conv = spans.get("bevencode", (0, 0.0))   # one bracket per step around the 16 conv launches
    n_spl, ms_spl = spans.get("lift_splat_level", (0, 0.0))   # one bracket around K2 || K3, region_fill, region_splat
    conv_flops_step = bevencode_flops(X, Y) * B
    peak_tf = MFMA_BF16_TFLOPS if args.precision == "bf16" else MFMA_F32_TFLOPS
    conv_tf = conv_flops_step * args.steps / (ms_conv * 1e-3) / 1e12 if ms_conv else 0.0
    out_bytes = 2 if args.precision == "bf16" else 4
    # algorithmic bytes of the lift-splat level: trunk features in, BEV grid out
    l1_bytes_step = B * l1_bytes_per_frame(6, fH, fW, C, X, Y, Z, D, out_bytes)
    splat_bytes_step = l1_bytes_step
    spl_gbs = splat_bytes_step * args.steps / (ms_spl * 1e-3) / 1e9 if ms_spl else 0.0

    conv_traffic, splat_traffic = pmc_traffic("conv_"), pmc_traffic("region_splat_kernel")
    l1_launches = 3 if os.environ.get("LSS_SPLAT_DIRECT") == "0" else 2   # (splat.hip: the direct form is the default)
    # the dominant kernel by itself: algorithmic FLOPs of ONE launch / its own average duration
    dom_flops = 2.0 * B * X * Y * 128 * (256 * 9 + 4)   # 3x3 256 -> 128 on the upsampled grid + the 1x1 head (4 classes)
    dom_us = ms_dom * 1e3 / n_dom if n_dom else 0.0
    dom_tf = dom_flops / (dom_us * 1e-6) / 1e12 if dom_us else 0.0
    ring = os.environ.get("LSS_CONV_RING") != "0" and args.precision == "bf16"
    dom_name = "conv_ring_kernel<1, true" if ring else "conv_lds_kernel<2, 128, 1, 3, 3, 1, 32, 1, true"
    dom_traffic = pmc_traffic(dom_name)
    dominant = {"kernel": ("conv_ring_kernel<MODE 1, HEAD> (loader / consumer ring kernel, csrc/conv_ring.hip)" if ring else
                           "conv_lds_kernel<2,128,1,3,3,1,32,1,HEAD>") + " = BevEncode.up2: x2 bilinear upsample (fused gather) + 3x3 "
                          "conv 256->128 + BN + ReLU + 1x1 head, one launch; own HIP-event bracket in a separate pass of the "
                          "same K steps (outside the timed region)",
                "bound": "mfma", "flops_per_launch": dom_flops, "avg_us": dom_us, "launches": n_dom, "achieved": dom_tf,
                "peak": peak_tf, "unit": "TFLOP/s", "frac": dom_tf / peak_tf if peak_tf else 0.0,
                "traffic": dom_traffic[0], "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": dom_traffic[1]}
    out = {
        "metric": "BEV frames/sec (6-cam 352x128 -> 200x200x64), full hot path: CamEncode lift + splat + BevEncode",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
        "data": "synthetic (seeded N(0,1) trunk features 6x512x8x22 per frame, nuScenes-like 6-camera rig "
                "with train-time augmentation, random-init weights)",
        "config": {"workload": {"config2": "BASELINE configs[1]: model_BEV_TXT LSS hot path, batch=%d/GPU, 6 cams 352x128, "
                                           "D=41, 200x200x64 BEV -> BevEncode -> 200x200x4, trunk (EfficientNet) not included: "
                                           "features are the input" % B,
                                "config1": "BASELINE configs[0] shapes on the GPU: model_baseline LSS, batch=%d, 6 cams 352x128, "
                                           "D=41, 200x200x64 BEV -> BevEncode -> 200x200x4 (the CPU leg of that config is "
                                           "`cpu_baseline`)" % B,
                                "hires": "BASELINE configs[4] per-GPU shapes: batch=%d/GPU, 6 cams 704x256, D=60, 400x400x64 BEV "
                                         "-> BevEncode -> 400x400x4" % B}[args.workload],
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d (sample-sharded, no collective)" % world,
                   "precision": args.precision, "spinup_ms": args.spinup_ms, "calibration": "CPU tensors per step; 3x3 inverses on host (exact-index contract), 576 floats passed in the kernel arguments (no H2D copy)"},
        "roofline": {"kernel": "conv_lds_kernel / conv_ks_kernel / conv_ring_kernel: the 16 BevEncode launches of a step (stride-2 convs carry their 1x1 downsample, "
                               "up2 its head; the ten 3x3 / stride-1 convs of layer1-3 run on the K-split one-pass kernel, the three big 3x3 layers on the ring kernel), one HIP-event "
                               "bracket around the group (per-launch brackets cost ~10 us of idle each)",
                     "bound": "mfma", "achieved": conv_tf, "peak": peak_tf, "unit": "TFLOP/s",
                     "frac": conv_tf / peak_tf, "traffic": conv_traffic[0], "traffic_unit": "HBM bytes per launch (PMC)",
                     "traffic_source": conv_traffic[1], "launches": n_conv * 16,
                     "avg_us": ms_conv * 1e3 / max(n_conv * 16, 1), "flops_per_step": conv_flops_step,
                     "dominant_kernel": dominant},
        "roofline_l1": {"kernel": ("lift-splat level = depthnet_rows_and_voxels (K2 || K3, writes the region entries itself) + "
                                   "region_splat (direct form: 2 launches, one HIP-event bracket); `traffic` is region_splat_kernel's own"
                                   if l1_launches == 2 else
                                   "lift-splat level = depthnet_rows_and_voxels (K2 || K3) + region_fill + region_splat "
                                   "(3 launches, one HIP-event bracket); `traffic` is region_splat_kernel's own"),
                        "bound": "hbm", "achieved": spl_gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": spl_gbs / HBM_PEAK_GBS, "traffic": splat_traffic[0],
                        "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": splat_traffic[1],
                        "launches": n_spl * l1_launches,
                        "avg_us": ms_spl * 1e3 / max(n_spl * l1_launches, 1), "level_us": ms_spl * 1e3 / max(n_spl, 1),
                        "bytes_per_step": splat_bytes_step},
        "per_rank_fps": per_rank, "comm": comm,
        "levels": {"L2_hot_path_fps": fps,
                   "cold_start_fps": frames / dt_cold,
                   "cold_start_note": "the same K steps timed right after the first W warm-up steps, before the "
                                      "config.spinup_ms spin-up (device still ramping its clock)",
                   "L1_lift_splat_fps": frames / dt_l1, "L1_ms_per_step": dt_l1 / args.steps * 1e3,
                   "L1_algorithmic_GBs": l1_bytes_step * args.steps * world / dt_l1 / 1e9,
                   "L1_frac_of_hbm_peak": l1_bytes_step * args.steps / dt_l1 / 1e9 / HBM_PEAK_GBS},
    }

    # ---- training step (fwd + bwd + Adam, RCCL all-reduce of one flat gradient bucket) ----
    if not args.no_train and args.workload == "config2":
        # An extra, informational leg: a failure in it (every rank sees the same exception class) must not cost the
        # main metric's line, so it is reported instead of raised.
        # ... a HANG in it (a collective some rank never joins) is different: every rank arms a watchdog that prints
        # the line measured so far, naming the leg (rank 0), and leaves with rc = WATCHDOG_RC - never a clean exit.
        wd = _arm_watchdog(float(os.environ.get("LSS_BENCH_TRAIN_TIMEOUT", "300")),
                           dict(out, train={"error": "watchdog: the train leg did not finish in time"}), rank, "train")
        try:
            out["train"] = train_leg(args, model, feats, calib, dev, dist, world, B)
        except Exception as e:  # noqa: BLE001
            out["train"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        wd.cancel()
        guards["train"] = read_guards()

    # ---- informational: two batches in flight (two HIP streams, one model instance each) ------
    # The single-stream number above stays `value`.  This leg shows how much of the step is
    # latency (the launch-bound BevEncode layers and the lift-splat level leave CUs idle that a
    # second, independent batch can use) - what a serving loop that double-buffers batches gets.
    dt_2s = 0.0
    if args.two_streams:
        with torch.no_grad():
            model_b = L.compile_model_lss(B, grid, aug, 4, precision=args.precision).to(dev).eval()
            model_b.load_state_dict(model.state_dict())
            models = (model, model_b)
            streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            torch.cuda.synchronize()
            for i in range(6):
                with torch.cuda.stream(streams[i & 1]):
                    models[i & 1](feats, *calib)
            barrier()
            t2 = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(streams[i & 1]):
                    models[i & 1](feats, *calib)
            barrier()
            dt_2s = time.perf_counter() - t2
    if dt_2s > 0:
        if dist is not None:
            t2s = torch.tensor([dt_2s], dtype=torch.float64, device=dev)
            dist.all_reduce(t2s, op=dist.ReduceOp.MAX)
            dt_2s = float(t2s[0])
        out["levels"]["two_batches_in_flight_fps"] = frames / dt_2s
        out["levels"]["two_batches_in_flight_note"] = ("same K steps alternated over 2 HIP streams / 2 model instances "
                                                       "(informational; `value` is the single-stream loop)")

    # ---- CPU baseline: the oracle (op-for-op torch port of the reference) on this host ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("config2", "config1"):
        out["cpu_baseline"] = cpu_baseline(model, feats, calib, B)

    # ---- hang guards: every leg's counters in the line; any non-zero one makes the run fail (rc GUARD_RC) ----
    guards["end"] = read_guards()
    rc = guards_rc(guards)
    if dist is not None:
        worst = torch.tensor([rc], dtype=torch.int32, device=dev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        rc = int(worst[0])
    out["guards"] = dict(guards, ok=rc == 0,
                         note="flag-wait timeout counters of the ring / K9w / chained kernels after each leg (0 = every "
                              "hand-off completed); non-zero anywhere -> exit code %d" % GUARD_RC)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        wd = _arm_watchdog(60.0, None, rank, "destroy_process_group")  # the line is out; a hung teardown is still rc != 0
        dist.destroy_process_group()
        wd.cancel()
    if rc:
        sys.stderr.write("bench.py: loader / consumer flag waits hit their bound: %s\n" % json.dumps(guards))
        raise SystemExit(rc)


WATCHDOG_RC = 3
GUARD_RC = 4   # a bounded flag wait of a loader / consumer kernel hit its limit during the run: outputs were garbage


def read_guards(lib=None):
    """Timeout counters of the loader / consumer kernels (ops.timeout_counters): read after every leg, OUTSIDE the
    timed regions (each read synchronises the device)."""
    from lss2_multimodal_nu_amd import ops
    return ops.timeout_counters(lib)


def guards_rc(guards):
    """Exit code the counters demand: 0 when every one reads 0."""
    return GUARD_RC if any(v != 0 for leg in guards.values() for v in leg.values()) else 0


def _arm_watchdog(seconds, line, rank, leg="?"):
    """After `seconds`: rank 0 prints `line` (the JSON object of everything measured so far, with the leg that hung
    named in it) and EVERY rank leaves with a non-zero code (WATCHDOG_RC): a hang in a process that has touched the
    GPU must not reach the launcher as a clean exit.  No retry, no re-exec."""
    import threading

    def fire():
        sys.stderr.write("bench.py watchdog: leg '%s' did not finish in %.0f s (rank %d)\n" % (leg, seconds, rank))
        sys.stderr.flush()
        if rank == 0 and line is not None:
            print(json.dumps(dict(line, watchdog={"hung_leg": leg, "timeout_s": seconds})), flush=True)
        os._exit(WATCHDOG_RC)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def train_leg(args, model, feats, calib, dev, dist, world, B):
    """fwd + bwd + one flat-bucket gradient all-reduce (RCCL) + clip + Adam: the
    reference's train.py:49-66 loop body on synthetic tensors."""
    import lss2_multimodal_nu_amd as L
    from lss2_multimodal_nu_amd import dp
    torch.manual_seed(0)
    m = L.compile_model_lss(B, GRID, AUG, 4, precision=args.precision).to(dev).train()
    # N > 1: every p.grad is a view of one flat buffer and the bucket all-reduces start inside backward.
    # N = 1: the reference's loop as it stands (dp.train_step_local) - no flat buffer, no per-parameter `grad += g`
    # One HIP graph per step when this is the only rank (dp.GraphedTrainStep; LSS_TRAIN_GRAPH=0: eager launches); under
    # data parallelism two graphs around the eager bucket all-reduces (graph A = zero / forward / backward, graph B = clip / Adam).
    graph_env = os.environ.get("LSS_TRAIN_GRAPH", "1") != "0"
    want_graph = graph_env
    bucket = dp.make_bucket(m) if world > 1 else None
    params = bucket.params if bucket is not None else [p for p in m.parameters() if p.requires_grad]
    if os.environ.get("LSS_TRAIN_OPT", "clipadam") == "clipadam":
        # clip_grad_norm_ + Adam (train.py:41, 62-63: same update rule) as three HIP launches (csrc/optim.hip)
        opt = L.ClipAdam(params, lr=1e-4, weight_decay=1e-8)
        adam = "ClipAdam (csrc/optim.hip: norm partials, finalize, clipped Adam update; LSS_TRAIN_OPT=torch = torch.optim.Adam)"
    else:
        try:  # torch's fused multi-tensor Adam where the build has it (A/B)
            opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-8, capturable=want_graph, fused=True)
            adam = "torch fused"
        except Exception:
            opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-8, capturable=want_graph)
            adam = "torch foreach"
    tgt = torch.randint(0, 4, (B, 200, 200), device=dev)
    weight = torch.tensor([1.0, 10.0, 5.0, 10.0], device=dev)         # ref: src/tools.py:234

    amp = args.precision == "bf16"  # bf16 autocast: native conv + BatchNorm units; fp32 master weights

    class _Amp(torch.nn.Module):
        """loss = SimpleLoss()(model(x, calib...), tgt) (ref: pre_train.py:54-58, src/tools.py:221-231) through the
        model's fused entry: 1x1 head + log-softmax + weighted NLL in one HIP kernel per direction, no logits tensor."""

        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, *a):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                return self.inner.forward_loss(*a, tgt)  # class weights [1, 10, 5, 10]: the model's default

    wrapped = _Amp(m)

    def loss_fn(loss):
        return loss

    def one():  # ref: train.py:49-66 (zero_grad, forward, loss, backward, clip 5.0, Adam) + the DP all-reduce
        dp.train_step(wrapped, bucket, opt, loss_fn, (feats,) + tuple(calib), clip=5.0)

    graph_note = "eager launches (LSS_TRAIN_GRAPH=0)"
    if want_graph:
        try:
            graphed = dp.GraphedTrainStep(wrapped, bucket, opt, loss_fn, feats, tuple(calib), clip=5.0, warmup=5)

            def one():  # noqa: F811  (the same step, replayed: features and calibration refreshed every step)
                graphed(feats, tuple(calib))
            graph_note = ("one HIP graph per step (dp.GraphedTrainStep): features + calibration refreshed, then one replay"
                          if graphed.graph_b is None else
                          "graph under world > 1 (dp.GraphedTrainStep): graph A [zero, forward, backward] -> eager bucket "
                          "all-reduces -> graph B [clip, Adam]")
        except Exception as e:  # capture refused by a library call: the eager step is still valid
            graph_note = "eager launches (graph capture failed: %s)" % (str(e).splitlines()[0][:160],)
            graphed = None
            torch.cuda.synchronize()
        if dist is not None:  # every rank must run the same sequence of collectives: all graphed, or none
            ok = torch.tensor([0 if graphed is None else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0 and graphed is not None:
                graphed = None
                graph_note = "eager launches (graph capture failed on another rank)"

                def one():  # noqa: F811
                    dp.train_step(wrapped, bucket, opt, loss_fn, (feats,) + tuple(calib), clip=5.0)
    for _ in range(5):  # MIOpen's first-call kernel selection, Adam state, allocator growth: all outside the timed steps
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt[0])
    return {"samples_per_s": args.train_steps * B * world / dt, "ms_per_step": dt / args.train_steps * 1e3,
            "steps": args.train_steps, "grad_bucket_MB": sum(p.numel() for p in params) * 4 / 1e6,
            "grad_buckets": [(hi - lo) * 4 / 1e6 for lo, hi, _ in bucket.buckets] if bucket is not None else [],
            "allreduce": ("none (1 rank)" if bucket is None else "direct RCCL (lss_allreduce_bucket)"
                          if bucket._direct is not None else "torch.distributed %s" % dist.get_backend()),
            "adam": adam,
            "amp_bf16": amp, "launch": graph_note,
            "note": "lift-splat fwd/bwd native HIP (fp32); under bf16 autocast every 3x3/s1 conv + BatchNorm(train) + "
                    "residual + ReLU unit of BevEncode (95 % of its FLOPs) is one HIP autograd node: conv fwd / dgrad / "
                    "wgrad, BN fwd / bwd, fused upsample+concat and its adjoint (LSS_TRAIN_NATIVE=0 = library path for "
                    "A/B); the 7x7/2 stem, the 3x3/2 convs and the 1x1/2 shortcuts run forward, dgrad and wgrad on the same HIP kernels "
                    "over phase planes; 1x1 head + weighted cross-entropy = one HIP kernel per direction; clip + Adam = three HIP "
                    "launches (optim.ClipAdam); library ops left: the two fp32 depthnet GEMMs of the lift-splat backward"}


def host_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota (a 1-GPU box exposes 256 logical
    CPUs but grants ~16).  LSS_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("LSS_CPU_THREADS", n))


def cpu_baseline(model, feats, calib, B):
    """The CPU oracle (= op-for-op port of the reference's torch code) on ALL the host cores this process is granted,
    same tensors, bounded sample: 1 warm-up + 5 timed passes of L1 and L2, median."""
    from oracle import bev_oracle as bo
    from oracle import lss_oracle as lo
    ncores = host_cores()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(ncores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    bsd = {k[len("bevencode."):]: v for k, v in sd.items() if k.startswith("bevencode.")}
    x = feats.cpu()
    NPASS = 5

    def l1():
        return lo.lift_splat_torch(x, sd["camencode.depthnet.weight"], sd["camencode.depthnet.bias"], sd["frustum"],
                                   *calib, sd["dx"], sd["bx"], sd["nx"], B, 41, 64)

    with torch.no_grad():
        grid = l1()
        bo.bev_encode(grid, bsd)
        t_l1, t_l2 = [], []
        for _ in range(NPASS):
            t0 = time.perf_counter(); grid = l1(); t1 = time.perf_counter()
            bo.bev_encode(grid, bsd); t2 = time.perf_counter()
            t_l1.append(t1 - t0); t_l2.append(t2 - t0)
    t_l1.sort(); t_l2.sort()
    torch.set_num_threads(threads_before)
    return {"value": B / t_l2[NPASS // 2], "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": "%d passes of batch %d (median; 1 warm-up pass before), fp32, full hot path, torch intra-op threads = "
                      "the %d cores granted to this process; L1 (lift-splat only) = %.1f frames/s"
                      % (NPASS, B, ncores, B / t_l1[NPASS // 2]),
            "L1_value": B / t_l1[NPASS // 2]}


if __name__ == "__main__":
    main()
