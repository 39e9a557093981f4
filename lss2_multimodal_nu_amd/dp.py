"""Data parallelism for the hot path: one process per GPU, samples sharded across
ranks, gradients averaged through ONE flat fp32 buffer.

The reference has no distributed code at all; this mirrors what
its single-GPU loop does after `loss.backward()` (`train.py:63-65`) - all-reduce,
`clip_grad_norm_(5.0)`, `opt.step()`.

Design for MI355X / RCCL over xGMI (point-to-point links, per-link-bound rings; the 18.6 MB of
CamEncode + BevEncode gradients sit in the latency-bound small-message regime, so few large
messages beat one per parameter):

* every `p.grad` IS a view into the flat buffer (no pack / unpack copy kernels: autograd
  accumulates straight into it; the buffer is cleared by one memset per step);
* the buffer is cut into a few contiguous buckets in REVERSE registration order (= the order in
  which backward produces gradients: BevEncode's up2/up1 first, CamEncode last); a
  post-accumulate hook counts a bucket's parameters and starts its all-reduce (async, on RCCL's
  own stream) the moment the last one is written, so the collective of the BevEncode tail
  overlaps the rest of backward;
* gradient clipping is two kernels on the flat buffer (norm, scale) instead of a foreach over
  every parameter.

Works with any torch.distributed backend (tests run it on gloo/CPU with world_size 2).  With
`LSS_DP_DIRECT_RCCL=1` the collective goes through this package's own RCCL communicator and the
C-ABI entry `lss_allreduce_bucket` (include/lss_hip.h) instead of torch's process group - on a
SIDE stream fenced with events (the bucket's gradients are complete on the compute stream ->
event -> the side stream waits, reduces, records -> `all_reduce_mean` makes the compute stream
wait), so it overlaps the rest of backward exactly like the torch.distributed path.
"""
import os

import torch

from . import ops


def shard_range(global_batch, rank, world):
    """Samples [lo, hi) owned by `rank`: the path shards by sample, all cameras of a
    sample stay on one rank."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d (the models fix bsize "
                         "at construction, ref src/model_BEV_TXT.py:76)" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradBucket:
    """Flat fp32 buffer that every trainable parameter's `.grad` is a view of."""

    def __init__(self, params, n_buckets=3, group=None, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        if any(p.dtype != torch.float32 for p in self.params):
            raise ValueError("GradBucket holds fp32 master gradients")
        dev = self.params[0].device
        self.group = group
        self.overlap = overlap
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        # layout: REVERSE registration order, so that bucket 0 (the first to complete during
        # backward) is one contiguous slice
        order = list(reversed(self.params))
        self.views, o = {}, 0
        target = -(-self.numel // max(1, n_buckets))
        self.buckets = []  # [start, end, n_params]
        self._bucket_of = {}
        cur = [0, 0, 0]
        for p in order:
            n = p.numel()
            self.views[p] = self.flat[o:o + n].view_as(p)
            o += n
            cur[1], cur[2] = o, cur[2] + 1
            self._bucket_of[p] = len(self.buckets)
            if o - cur[0] >= target and len(self.buckets) < n_buckets - 1:
                self.buckets.append(cur)
                cur = [o, o, 0]
        if cur[2]:
            self.buckets.append(cur)
        self._ready = [0] * len(self.buckets)
        self._work = []
        self._fired = set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._direct = None
        self.capturing = False  # True while GraphedTrainStep captures backward: the hooks must not start collectives
        self.attach()

    # -- the views --------------------------------------------------------------
    def attach(self):
        """(Re-)point every p.grad at its slice (after something replaced or dropped it)."""
        for p in self.params:
            if p.grad is not self.views[p]:
                p.grad = self.views[p]

    def zero(self):
        """Start of a step: one memset instead of `opt.zero_grad()` (which would drop the views)."""
        self.flat.zero_()
        self._ready = [0] * len(self.buckets)
        self._work = []
        self._fired = set()
        self.attach()

    # -- collective -------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _launch(self, b):
        import torch.distributed as dist
        lo, hi, _ = self.buckets[b]
        seg = self.flat[lo:hi]
        if self._direct is not None:
            self._work.append(self._direct.all_reduce_sum_async(seg))  # side stream, event-fenced
            return
        self._work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        if p.grad is not self.views[p]:
            # autograd installed its own tensor (the view had been dropped): fold it in and re-attach
            self.views[p].add_(p.grad)
            p.grad = self.views[p]
        self._fired.add(p)
        b = self._bucket_of[p]
        self._ready[b] += 1
        if self.overlap and not self.capturing and self._ready[b] == self.buckets[b][2] and self._world() > 1:
            self._launch(b)
            self._ready[b] = -1  # launched

    def use_direct_rccl(self, comm):
        """Route the collective through `lss_allreduce_bucket` on this package's own communicator."""
        self._direct = comm

    def all_reduce_mean(self):
        """grads <- mean over ranks (the loss is a per-rank mean, so the global-batch gradient is
        the mean of the per-rank gradients).  Buckets whose all-reduce the backward hooks already
        started are only waited for."""
        world = self._world()
        if world > 1:
            for b in range(len(self.buckets)):
                if self._ready[b] != -1:  # unused parameters in it, or overlap off
                    self._launch(b)
                    self._ready[b] = -1
            for w in self._work:
                w.wait()
            self._work = []
            self.flat.div_(world)

    def all_reduce_all(self):
        """The collective of a REPLAYED backward (GraphedTrainStep under data parallelism): no hook ran, so every
        bucket is launched here, in bucket order, then waited for; grads <- mean over ranks."""
        world = self._world()
        if world > 1:
            import torch.distributed as dist
            if self._direct is None and self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
                # a host-driven collective (the one-GPU rehearsal: bench.py under LSS_BENCH_REHEARSE, the gloo tests)
                # reads the bucket through the host anyway; handing it a buffer whose producing graph is still in the
                # queue made each step stall 0.8-2.9 s at random with two ranks on one GPU (14 ms with this wait).  RCCL
                # collectives are stream-ordered on the device and are launched without any host wait.
                torch.cuda.current_stream().synchronize()
            self._work = []
            for b in range(len(self.buckets)):
                self._launch(b)
            for w in self._work:
                w.wait()
            self._work = []
            self.flat.div_(world)

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(params, max_norm) (ref train.py:64) on the flat buffer: the
        same total 2-norm (the buffer holds exactly the gradients) and the same `max_norm / (norm +
        1e-6)` scale clamped to 1, in two kernels and without a host sync."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total

    def hide_unused(self):
        """Parameters no gradient reached this step get `.grad = None` for the optimizer call, as in
        the reference's loop (Adam skips them); returns them for `attach()` afterwards."""
        unused = [p for p in self.params if p not in self._fired]
        for p in unused:
            p.grad = None
        return unused


class _StreamWork:
    """Handle of a collective enqueued on a side stream: `wait()` orders the caller's current stream behind it."""

    def __init__(self, done_event, keep):
        self.done, self.keep = done_event, keep

    def wait(self):
        _streams().current_stream().wait_event(self.done)
        self.keep = None


def _streams():
    """The stream / event provider (torch.cuda); a test substitutes a recording fake to check the fence order
    without a GPU."""
    return _stream_api[0]


_stream_api = [torch.cuda]


def _fenced_launch(comm, t):
    """compute stream --event--> side stream: collective --event--> (wait) compute stream."""
    api = _streams()
    if getattr(comm, "_side", None) is None:
        comm._side = api.Stream()
    ready = api.Event()
    ready.record(api.current_stream())          # the bucket's gradients are complete here
    comm._side.wait_event(ready)
    with api.stream(comm._side):
        comm.all_reduce_sum(t)                  # enqueued on the side stream (N.stream() = torch's current)
    done = api.Event()
    done.record(comm._side)
    return _StreamWork(done, t)


class DirectRccl:
    """This package's own RCCL communicator over the ranks of a torch.distributed group, driven
    through the C ABI (`lss_rccl_*`, `lss_allreduce_bucket` in include/lss_hip.h).  The unique id is
    created on rank 0 and broadcast through the (already initialised) torch process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        from . import _native as N
        self.N = N
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = [N.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.comm = N.rccl_comm_init(uid[0], world, rank)
        self.world = world

    def _check(self, t):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("all_reduce_sum: contiguous fp32 GPU tensor")

    def all_reduce_sum(self, t):
        """In-stream form: the collective is enqueued on the caller's current stream."""
        self._check(t)
        N = self.N
        N.check(N.lib().lss_allreduce_bucket(self.comm, N.ptr(t), t.numel(), N.stream()), "lss_allreduce_bucket")

    def all_reduce_sum_async(self, t):
        """Overlapped form: everything enqueued so far on the current (compute) stream happens before the
        collective, which runs on this communicator's own side stream; the returned work's `wait()` makes the
        then-current stream wait for it (no host blocking anywhere)."""
        self._check(t)
        return _fenced_launch(self, t)

    def close(self):
        if self.comm is not None:
            self.N.lib().lss_rccl_comm_destroy(self.comm)
            self.comm = None


def make_bucket(model_or_params, group=None, n_buckets=3, overlap=True):
    params = model_or_params.parameters() if isinstance(model_or_params, torch.nn.Module) else model_or_params
    bucket = GradBucket(params, n_buckets=n_buckets, group=group, overlap=overlap)
    if os.environ.get("LSS_DP_DIRECT_RCCL") == "1" and bucket.flat.is_cuda and bucket._world() > 1:
        bucket.use_direct_rccl(DirectRccl(group))
    return bucket


def train_step(model, bucket, opt, loss_fn, inputs, clip=5.0):
    """One data-parallel step: fwd, bwd (bucket all-reduces start inside it), mean, clip, optimizer.
    `bucket=None`: the single-process form (`train_step_local`)."""
    if bucket is None:
        return train_step_local(model, opt, loss_fn, inputs, clip)
    bucket.zero()
    ops.prepack.run(bucket.params)   # every registered bf16 weight image of the step in one launch (ops.WeightPrepack)
    try:
        loss = loss_fn(model(*inputs))
        loss.backward()
    finally:
        ops.prepack.invalidate()
    bucket.all_reduce_mean()
    _clip_and_step(opt, bucket, None, clip)
    return loss.detach()


def _clip_and_step(opt, bucket, params, clip):
    """clip_grad_norm_(clip) + opt.step() (train.py:62-63).  optim.ClipAdam does both in three launches; any other
    optimizer gets the clip on the flat buffer (bucket) or torch's foreach form (no host sync), then its own step.
    Leaves the total norm before the clip in `_last_norm[0]` (a device tensor)."""
    from .optim import ClipAdam
    unused = bucket.hide_unused() if bucket is not None else None
    if isinstance(opt, ClipAdam):
        opt.step(max_grad_norm=clip)
        _last_norm[0] = opt.grad_norm
    else:
        if bucket is not None:
            _last_norm[0] = bucket.clip_grad_norm_(clip)
        else:
            _last_norm[0] = torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], clip, foreach=True)
        opt.step()
    if unused:
        bucket.attach()


_last_norm = [None]  # total gradient 2-norm BEFORE the clip of the latest step (a device tensor: no host sync)


def train_step_local(model, opt, loss_fn, inputs, clip=5.0):
    """The reference's loop body as it stands (train.py:49-66: zero_grad, forward, loss, backward,
    clip_grad_norm_(5.0), step) for ONE process: no flat buffer.  With `.grad = None` at the start autograd hands
    each parameter its gradient tensor as produced (no `grad += g` kernel per parameter: 62 launches a step on
    the bucket path, whose flat buffer only the all-reduce needs); clipping is torch's foreach form (no host
    sync)."""
    params = [p for g in opt.param_groups for p in g["params"]]
    for p in params:
        p.grad = None
    ops.prepack.run(params)   # every registered bf16 weight image of the step in one launch (ops.WeightPrepack)
    try:
        loss = loss_fn(model(*inputs))
        loss.backward()
    finally:
        ops.prepack.invalidate()
    _clip_and_step(opt, None, params, clip)
    return loss.detach()


class GraphedTrainStep:
    """`train_step` captured once in HIP graphs and replayed: the ~300 launches of a step (19 conv + BatchNorm
    units forward and backward, the lift-splat pair, loss, clip, Adam) leave the host as ONE launch (two under data
    parallelism), so the step runs at the speed of its kernels instead of the speed of the Python / autograd dispatch
    around them (measured: the eager step is host-bound once the weight-gradient kernels are fast).

    Everything a step reads from outside lives in STATIC device buffers that `__call__` refreshes before the
    replay: the feature tensor, the targets the loss function closes over (pass them as `(static_device_tensor,
    new_value)` pairs through `__call__(..., refresh=...)`), and the calibration as a device-resident
    `CalibrationPack` (host inverses as always, data.prepare_calibration; one H2D copy per step).  The optimizer
    must be optim.ClipAdam (clip + Adam in three launches, device-resident step counter) or a torch optimizer built
    with `capturable=True`.

    One process: ONE graph [zero / drop grads, forward, loss, backward, clip, Adam].
    Data parallel (`bucket` over a group of more than one rank): graph A = [zero the flat buffer, forward, loss,
    backward] - the backward hooks count their buckets in but start nothing while capturing -, then the bucket
    all-reduces EAGERLY on the flat buffer (`GradBucket.all_reduce_all`: the communication library's launches stay
    outside any capture), then graph B = [clip, Adam] from the same memory pool.  Every rank replays the same sequence,
    so the collectives match up.

    After capture the step is replayed three times on the capture inputs and checked (`_self_check`): the total
    gradient norm BEFORE the clip (the clip would hide any finite garbage: it rescales whatever it is handed to the
    clip value) must be finite, positive and stable, and the loader / consumer kernels' timeout counters must read 0
    (`ops.assert_no_timeouts`).  A library kernel that is not safe inside a graph shows up there (MIOpen's
    weight-gradient solvers were: modules.conv_s2_backward_gemm)."""

    CHECK_REPLAYS = 3  # replays of the self-check after capture (ordinary steps on the capture inputs)

    def __init__(self, model, bucket, opt, loss_fn, feats, calib, clip=5.0, warmup=3):
        from . import modules
        from .data import CalibrationPack, prepare_calibration
        if warmup < 1:
            raise ValueError("GraphedTrainStep needs at least one eager warm-up step before the capture (first-call "
                             "work - kernel attributes, index tables, optimizer state - must not happen inside it)")
        if not feats.is_cuda:
            raise ValueError("GraphedTrainStep: features must be on the GPU")
        modules.warm_s2_tables(feats.device)  # (pageable H2D copies: refused inside a capture)
        self._prepare = prepare_calibration
        self.feats = feats.detach().clone()
        host = calib if isinstance(calib, CalibrationPack) else prepare_calibration(*calib)
        self.pack = CalibrationPack(host.buffer.to(feats.device), host.shape)
        self._inputs = (self.feats, self.pack, None, None, None, None)
        self._params = [p for g in opt.param_groups for p in g["params"]]
        self.bucket = bucket
        self.distributed = bucket is not None and bucket._world() > 1

        def step():
            return train_step(model, bucket, opt, loss_fn, self._inputs, clip=clip)

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # first-call work (kernel selection, optimizer state, allocator growth)
            for _ in range(warmup):
                step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_b = None
        if not self.distributed:
            with torch.cuda.graph(self.graph):
                self.loss = step()
            self.grad_norm = _last_norm[0]
        else:
            bucket.capturing = True
            try:
                # (thread-local capture mode: the process group's watchdog thread keeps querying its events meanwhile)
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    bucket.zero()
                    ops.prepack.run(bucket.params)
                    loss = loss_fn(model(*self._inputs))
                    loss.backward()
                    self.loss = loss.detach()
            finally:
                bucket.capturing = False
                ops.prepack.invalidate()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode="thread_local"):
                _clip_and_step(opt, bucket, None, clip)
                self.grad_norm = _last_norm[0]
        self._self_check()

    def _replay(self):
        if os.environ.get("LSS_GRAPH_DEBUG") and self.graph_b is not None:   # per-phase host times (synchronising)
            import time
            t = [time.perf_counter()]
            self.graph.replay(); torch.cuda.synchronize(); t.append(time.perf_counter())
            self.bucket.all_reduce_all(); torch.cuda.synchronize(); t.append(time.perf_counter())
            self.graph_b.replay(); torch.cuda.synchronize(); t.append(time.perf_counter())
            print("    graph A %.2f ms | all-reduce %.2f ms | graph B %.2f ms" % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])),
                  flush=True)
            return
        self.graph.replay()
        if self.graph_b is not None:
            self.bucket.all_reduce_all()
            self.graph_b.replay()

    def _self_check(self):
        """Replay the captured step on the inputs it was captured with and look at what it leaves: the PRE-clip total
        gradient norm (a static tensor of the captured step) and the loader / consumer kernels' timeout counters.  A
        kernel that accumulates into a buffer which is only cleared outside the captured stream shows up as a norm
        that explodes, vanishes or goes non-finite from the second replay on - seen with MIOpen's weight-gradient
        solvers: 1e30 in one tensor, which the clip then turned into an all-zero gradient while Adam's momentum kept
        the loss moving.  (Round 3 looked at the gradients AFTER the clip: any finite garbage above the clip value
        came out as norm ~ clip and passed - ADVICE r3.)  The replays are ordinary training steps: extra warm-up."""
        from . import ops
        norms = []
        for _ in range(self.CHECK_REPLAYS):
            self._replay()
            torch.cuda.synchronize()
            norms.append(float(self.grad_norm))
            if os.environ.get("LSS_GRAPH_DEBUG"):
                print("  self-check replay: pre-clip norm %.5g loss %.5f" % (norms[-1], float(self.loss)), flush=True)
        ok = all(n == n and 0.0 < n < float("inf") for n in norms) and max(norms) <= 20.0 * min(norms)
        if not ok:
            raise RuntimeError("replayed gradients are not stable (pre-clip norms %s): something in the step is not "
                               "graph-safe" % (", ".join("%.3g" % n for n in norms),))
        ops.assert_no_timeouts("GraphedTrainStep self-check")

    def __call__(self, feats, calib, refresh=()):
        """One step on new inputs; returns the (static) loss tensor.  `refresh`: (static_tensor, new_value) pairs
        of anything else the captured step reads (e.g. the target tensor of the loss function)."""
        from .data import CalibrationPack
        dbg = os.environ.get("LSS_GRAPH_DEBUG")
        if dbg:
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if feats is not self.feats:
            self.feats.copy_(feats, non_blocking=True)
        if calib is not None:
            host = calib if isinstance(calib, CalibrationPack) else self._prepare(*calib)
            if host.buffer is not self.pack.buffer:
                self.pack.buffer.copy_(host.buffer, non_blocking=True)
        for dst, src in refresh:
            dst.copy_(src, non_blocking=True)
        if dbg:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        self._replay()
        if dbg:
            torch.cuda.synchronize()
            print("  graphed step: refresh %.2f ms, replay %.2f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3),
                  flush=True)
        return self.loss
