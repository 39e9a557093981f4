"""Tensor-level wrappers over the C ABI (include/lss_hip.h).

Each function validates shapes/dtypes on the host (a mis-shaped operand must
never reach a kernel), allocates outputs with torch, and enqueues the kernel on
torch's current HIP stream.  No CPU / eager fallbacks exist here.
"""
import ctypes
import os
import threading

import torch

from . import _native as N
from ._native import (ACT_GELU, ACT_NONE, ACT_RELU, BEV_NCHW_F32, BEV_NHWC_BF16, BEV_NHWC_F32, DT_BF16,  # noqa: F401
                      DT_F32, OUT_F32, OUT_HEAD_MAJOR32, VALUE_HEAD_MAJOR, VALUE_NHWC, W_KS, W_RING)


def _f32c(t, name, shape=None):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError("%s must be a contiguous fp32 GPU tensor (got %s %s contiguous=%s)"
                         % (name, t.dtype, t.device, t.is_contiguous()))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


class KernelTimer:
    """Optional HIP-event brackets around native launches (bench.py uses it for
    the roofline numbers).  Events are recorded on torch's current stream - the
    stream the kernels are enqueued on."""

    def __init__(self, fine=False, last_conv=False):
        # fine=True brackets every native launch (each bracket costs ~5-10 us of GPU idle
        # time: diagnostics only); fine=False only the regions opened with `region()`.
        # last_conv=True: a recorded conv plan is replayed in two native calls and its LAST launch
        # (BevEncode: up2 3x3 + BN + ReLU + fused 1x1 head, the dominant kernel) gets a bracket of
        # its own, tag "conv_plan_last" - bench.py's separate per-kernel roofline pass.
        self.spans = {}
        self.fine = fine
        self.last_conv = last_conv

    def bracket(self, tag):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.setdefault(tag, []).append((s, e))
        return s, e

    def totals_ms(self):
        """tag -> (launches, total ms); call after a device synchronize."""
        return {t: (len(v), sum(s.elapsed_time(e) for s, e in v)) for t, v in self.spans.items()}


_timer = None


def set_timer(t):
    global _timer
    _timer = t


class _timed:
    def __init__(self, tag, coarse=False):
        self.ev = _timer.bracket(tag) if (_timer is not None and tag and (coarse or _timer.fine)) else None

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev:
            self.ev[1].record()
        return False


def region(tag):
    """Coarse HIP-event bracket around a group of launches (e.g. all of BevEncode)."""
    return _timed(tag, coarse=True)


# --- hang guards of the loader / consumer kernels (conv_ring.hip, conv_wgrad.hip, chained convs) ---------------------
# Every FULL / FREE flag wait of those kernels is bounded; a wait that hits its bound ends the grid instead of hanging
# the GPU, bumps a device counter and leaves GARBAGE in that launch's output.  The counters are therefore part of the
# product's contract, not a test detail: bench.py reads them after every leg, BevEncode whenever it records a launch
# plan (and every LSS_GUARD_EVERY-th call), dp.GraphedTrainStep in its self-check.
GUARD_COUNTERS = ("lss_conv2d_ring_timeouts", "lss_conv2d_wgrad_timeouts")


def timeout_counters(lib=None):
    """{counter name: value} of the flag-wait timeout counters (0 = every hand-off completed; -1 = the counter could
    not be read).  Synchronises the device (a device-to-host copy of one int each): never call inside a stream capture."""
    L = lib if lib is not None else N.lib()
    return {name[len("lss_conv2d_"):]: int(getattr(L, name)()) for name in GUARD_COUNTERS}


def assert_no_timeouts(where, lib=None):
    """Raise LssNativeError when a loader / consumer hand-off of any launch so far ran into its bound."""
    bad = {k: v for k, v in timeout_counters(lib).items() if v != 0}
    if bad:
        raise N.LssNativeError("%s: flag waits of the loader / consumer kernels hit their bound (%s): the affected "
                               "launches produced garbage" % (where, ", ".join("%s=%d" % kv for kv in sorted(bad.items()))))


def guard_every():
    """LSS_GUARD_EVERY=N: BevEncode checks the counters on every N-th inference call (0 / unset: only when it records a
    launch plan; each check is a device synchronisation)."""
    try:
        return max(0, int(os.environ.get("LSS_GUARD_EVERY", "0")))
    except ValueError:
        return 0


class ConvRecorder:
    """Collects the conv launches of one eager pass (pointers + shapes) together with the
    tensors that must outlive them; `ConvPlan` replays the list with ONE native call."""

    def __init__(self):
        self.launches = []
        self.keep = []

    def add(self, kind, keep, **f):
        self.launches.append((kind, f))
        self.keep.extend(t for t in keep if t is not None)


_recorder = None


def set_recorder(r):
    global _recorder
    _recorder = r


class ConvPlan:
    def __init__(self, rec, x_in, out):
        n = len(rec.launches)
        self.n = n
        self.arr = (N.ConvLaunch * n)()
        self.keep = rec.keep
        self.in_slots, self.out_slots = [], []
        pin, pout = x_in.data_ptr(), out.data_ptr()
        for i, (kind, f) in enumerate(rec.launches):
            c = self.arr[i]
            c.kind = kind
            for k, v in f.items():
                setattr(c, k, v.data_ptr() if isinstance(v, torch.Tensor) else (v if v is not None else None))
            for name in ("x", "x2", "residual"):
                if getattr(c, name) == pin:
                    self.in_slots.append((i, name))
            for name in ("y", "head_out"):
                if getattr(c, name) == pout:
                    self.out_slots.append((i, name))
        if not self.in_slots or not self.out_slots:
            raise RuntimeError("conv plan: input / output tensor not found among the recorded launches")

    def run(self, x_in, out):
        pin, pout = x_in.data_ptr(), out.data_ptr()
        for i, name in self.in_slots:
            setattr(self.arr[i], name, pin)
        for i, name in self.out_slots:
            setattr(self.arr[i], name, pout)
        if _timer is not None and _timer.last_conv and self.n > 1:
            import ctypes
            N.check(N.lib().lss_conv2d_sequence(self.arr, self.n - 1, N.stream()), "lss_conv2d_sequence")
            last = ctypes.byref(self.arr, (self.n - 1) * ctypes.sizeof(self.arr[0]))  # &arr[n - 1]
            with _timed("conv_plan_last", coarse=True):
                N.check(N.lib().lss_conv2d_sequence(last, 1, N.stream()), "lss_conv2d_sequence")
            return
        N.check(N.lib().lss_conv2d_sequence(self.arr, self.n, N.stream()), "lss_conv2d_sequence")


class SplatWorkspace:
    """Index buffers of one (B,N,D,fH,fW | X,Y,Z) problem, reused across steps.

    vox_count and cursor obey the K4 contract: zero on entry, zero on return, so
    they are zero-filled exactly once, here."""

    def __init__(self, P, nvox, device):
        self.P, self.nvox = P, nvox
        self.voxel = torch.empty(P, dtype=torch.int32, device=device)
        self.vox_count = torch.zeros(nvox, dtype=torch.int32, device=device)
        self.vox_list = torch.empty(nvox, 2, dtype=torch.int32, device=device)
        self.entries = torch.empty(P, 2, dtype=torch.int32, device=device)  # {point id, depth weight}
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        self._direct = {}

    def direct_buffer(self, dims, nx):
        """The larger entry workspace of the two-launch (direct) region pipeline (lss_lift_splat_direct_bytes): fixed-
        capacity per-region buckets, 8 KiB per region (20 MB at batch 4, 200 x 200); None where the region pipeline
        does not apply or with LSS_SPLAT_DIRECT=0.  Allocated on first use (a warm-up step, never inside a capture)."""
        key = (tuple(dims), tuple(nx))
        if key not in self._direct:
            B, Ncam, D, fH, fW, C = dims
            X, Y, Z = nx
            n = 0 if os.environ.get("LSS_SPLAT_DIRECT") == "0" else \
                int(N.lib().lss_lift_splat_direct_bytes(B, Ncam, D, fH, fW, C, X, Y, Z))
            self._direct[key] = torch.empty(n, dtype=torch.uint8, device=self.voxel.device) if n else None
        return self._direct[key]


def _lift_splat_desc(ws, dims, nx, layout, math, bev, depth, feat, frustum, dx, bx, x=None, w=None, bias=None, Cin=0,
                     calib_dev=None, calib_host=None):
    """One lss_lift_splat_desc_t for the three forms of the fused lift-splat call (include/lss_hip.h)."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    d = N.LiftSplatDesc()
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    d.frustum, d.dx, d.bx = P(frustum), P(dx), P(bx)
    if calib_host is not None:
        d.calib_host = calib_host.data_ptr()
    else:
        d.inv_post_rots, d.post_trans, d.combine, d.trans = (P(t) for t in calib_dev)
    d.x, d.w, d.bias = P(x), P(w), P(bias)
    d.voxel, d.vox_count, d.vox_list = P(ws.voxel), P(ws.vox_count), P(ws.vox_list)
    d.entries, d.cursor = P(ws.entries), P(ws.cursor)
    direct = ws.direct_buffer(dims, nx)
    d.direct_entries = P(direct)
    d.direct_bytes = 0 if direct is None else direct.numel()
    d.depth, d.feat, d.bev = P(depth), P(feat), P(bev)
    d.B, d.N, d.D, d.fH, d.fW, d.Cin, d.C, d.X, d.Y, d.Z = B, Ncam, D, fH, fW, Cin, C, X, Y, Z
    d.layout, d.math = layout, math
    return d


def _run_desc(d, what):
    import ctypes
    N.check(N.lib().lss_lift_splat_forward_desc(ctypes.byref(d), N.stream()), what)


def points_to_voxels(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, nx, ws,
                     want_geom=False, histogram=True):
    """K3.  frustum (D,fH,fW,3); per-camera tensors (B,N,3,3)/(B,N,3); nx = (X,Y,Z) ints.
    Fills ws.voxel (+ ws.vox_count histogram); returns geom (B,N,D,fH,fW,3) or None."""
    D, fH, fW, _ = frustum.shape
    B, Ncam = inv_post_rots.shape[:2]
    X, Y, Z = nx
    _f32c(frustum, "frustum", (D, fH, fW, 3))
    _f32c(inv_post_rots, "inv_post_rots", (B, Ncam, 3, 3))
    _f32c(combine, "combine", (B, Ncam, 3, 3))
    _f32c(post_trans, "post_trans", (B, Ncam, 3))
    _f32c(trans, "trans", (B, Ncam, 3))
    _f32c(dx, "dx", (3,))
    _f32c(bx, "bx", (3,))
    P = B * Ncam * D * fH * fW
    if ws.P != P or ws.nvox != B * X * Y * Z:
        raise ValueError("workspace sized for P=%d nvox=%d, problem has P=%d nvox=%d"
                         % (ws.P, ws.nvox, P, B * X * Y * Z))
    geom = torch.empty(B, Ncam, D, fH, fW, 3, dtype=torch.float32, device=frustum.device) if want_geom else None
    N.check(N.lib().lss_points_to_voxels(
        N.ptr(frustum), N.ptr(inv_post_rots), N.ptr(post_trans), N.ptr(combine), N.ptr(trans),
        N.ptr(dx), N.ptr(bx), B, Ncam, D, fH, fW, X, Y, Z, N.ptr(ws.voxel),
        N.ptr(ws.vox_count) if histogram else None, N.ptr(geom), N.stream()), "lss_points_to_voxels")
    return geom


def geom_to_voxels(geom, dx, bx, nx, B, ws, histogram=True):
    """API-compat half of K3: geom (..., 3) fp32 with B samples of equal point count."""
    _f32c(geom, "geom")
    P = geom.numel() // 3
    X, Y, Z = nx
    if geom.shape[-1] != 3 or P % B != 0 or ws.P != P or ws.nvox != B * X * Y * Z:
        raise ValueError("geom / workspace shape mismatch")
    N.check(N.lib().lss_geom_to_voxels(N.ptr(geom), N.ptr(_f32c(dx, "dx", (3,))), N.ptr(_f32c(bx, "bx", (3,))),
                                       B, P // B, X, Y, Z, N.ptr(ws.voxel),
                                       N.ptr(ws.vox_count) if histogram else None, N.stream()),
            "lss_geom_to_voxels")


def bucket_points(ws, depth=None, D=1, HW=1):
    """K4 on a workspace whose voxel/vox_count were filled by K3.  `depth` = K2's
    (BN,D,fH,fW) tensor (its flat index is the point id); None = unit weights with the
    points taken as D = HW = 1 rows (pre-lifted inputs)."""
    if depth is not None:
        _f32c(depth, "depth")
        if depth.numel() != ws.P or depth.dim() != 4:
            raise ValueError("depth must be (BN,D,fH,fW) with %d elements" % ws.P)
        D, HW = depth.shape[1], depth.shape[2] * depth.shape[3]
    N.check(N.lib().lss_bucket_points(N.ptr(ws.voxel), N.ptr(depth), ws.P, D, HW, ws.nvox, N.ptr(ws.vox_count),
                                      N.ptr(ws.vox_list), N.ptr(ws.entries), N.ptr(ws.cursor),
                                      N.stream()), "lss_bucket_points")


def depthnet_softmax(x, weight, bias, D, C, math=DT_F32):
    """K2.  x (BN,Cin,fH,fW) fp32 NCHW; weight (D+C,Cin,1,1) or (D+C,Cin); bias (D+C).
    Returns depth (BN,D,fH,fW) and feat (BN,fH,fW,C) (channels-last rows)."""
    BN, Cin, fH, fW = x.shape
    _f32c(x, "x")
    w2 = weight.reshape(weight.shape[0], -1)
    _f32c(w2, "depthnet.weight", (D + C, Cin))
    _f32c(bias, "depthnet.bias", (D + C,))
    depth = torch.empty(BN, D, fH, fW, dtype=torch.float32, device=x.device)
    feat = torch.empty(BN, fH, fW, C, dtype=torch.float32, device=x.device)
    N.check(N.lib().lss_depthnet_softmax_fwd(N.ptr(x), N.ptr(w2), N.ptr(bias), BN, Cin, fH * fW, D, C,
                                             N.ptr(depth), N.ptr(feat), math, N.stream()),
            "lss_depthnet_softmax_fwd")
    return depth, feat


def camencode_v2(hidden, w_depth, b_depth, D, c3=None, w_feat=None, b_feat=None, softmax=True, math=None):
    """K2v.  hidden (BN,fH,fW,Cd) NHWC fp32|bf16; w_depth (D,Cd[,1,1]); optional c3
    (BN,Cf,fH,fW) fp32 NCHW with w_feat (C,Cf[,1,1]) / b_feat.  Returns depth
    (BN,D,fH,fW) (probabilities, or raw logits with softmax=False) and feat
    (BN,fH,fW,C) or None.  math: DT_F32 (exact fp32 FMA chains) or DT_BF16 (bf16 MFMA, fp32
    accumulate); default = the hidden map's own precision when the shapes allow."""
    BN, fH, fW, Cd = hidden.shape
    if not hidden.is_contiguous() or hidden.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("hidden must be contiguous NHWC fp32/bf16")
    dt = DT_F32 if hidden.dtype == torch.float32 else DT_BF16
    wd = w_depth.reshape(w_depth.shape[0], -1)
    _f32c(wd, "w_depth", (D, Cd))
    _f32c(b_depth, "b_depth", (D,))
    dev = hidden.device
    depth = torch.empty(BN, D, fH, fW, dtype=torch.float32, device=dev)
    if c3 is not None:
        _f32c(c3, "c3")
        if c3.shape[0] != BN or tuple(c3.shape[2:]) != (fH, fW):
            raise ValueError("c3 %s does not match hidden %s" % (tuple(c3.shape), tuple(hidden.shape)))
        Cf = c3.shape[1]
        wf = w_feat.reshape(w_feat.shape[0], -1)
        C = wf.shape[0]
        _f32c(wf, "w_feat", (C, Cf))
        _f32c(b_feat, "b_feat", (C,))
        feat = torch.empty(BN, fH, fW, C, dtype=torch.float32, device=dev)
        args = (N.ptr(c3), N.ptr(wf), N.ptr(b_feat), Cf)
    else:
        C, feat, args = 0, None, (None, None, None, 0)
    if math is None:
        math = DT_BF16 if (dt == DT_BF16 and Cd % 128 == 0 and (C == 0 or args[3] % 128 == 0)) else DT_F32
    with _timed("camencode_v2"):
        N.check(N.lib().lss_camencode_v2_fwd(N.ptr(hidden), dt, N.ptr(wd), N.ptr(b_depth), Cd, *args, BN,
                                             fH * fW, D, C, 1 if softmax else 0, math, N.ptr(depth),
                                             N.ptr(feat) if feat is not None else None, N.stream()),
                "lss_camencode_v2_fwd")
    return depth, feat


def depth_fuse_softmax(d3, d4, w_fusion, scale, shift):
    """MultiScaleDepthNet tail: d3 (BN,D,H,W), d4 (BN,D,H4,W4) raw logits ->
    softmax(relu(scale * fusion([d3, up(d4)]) + shift)) (BN,D,H,W)."""
    BN, D, H, W = d3.shape
    _f32c(d3, "d3")
    _f32c(d4, "d4")
    if d4.shape[0] != BN or d4.shape[1] != D:
        raise ValueError("d4 %s does not match d3 %s" % (tuple(d4.shape), tuple(d3.shape)))
    w2 = w_fusion.reshape(w_fusion.shape[0], -1)
    _f32c(w2, "w_fusion", (D, 2 * D))
    _f32c(scale, "scale", (D,))
    _f32c(shift, "shift", (D,))
    depth = torch.empty_like(d3)
    with _timed("depth_fuse_softmax"):
        N.check(N.lib().lss_depth_fuse_softmax_fwd(N.ptr(d3), N.ptr(d4), N.ptr(w2), N.ptr(scale), N.ptr(shift),
                                                   BN, D, H, W, d4.shape[2], d4.shape[3], N.ptr(depth),
                                                   N.stream()), "lss_depth_fuse_softmax_fwd")
    return depth


def lift_splat_fwd(feat, ws, dims, nx, layout=BEV_NCHW_F32, tag="lift_splat_fwd"):
    """K5/K6 on a bucketed workspace (depth weights already sit in ws.entries).
    dims = (B,N,D,fH,fW,C).  Returns the BEV tensor with LOGICAL shape
    (B, Z*C, X, Y): contiguous for NCHW_F32, channels_last strides for NHWC_*."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    _f32c(feat, "feat")
    if feat.numel() != B * Ncam * fH * fW * C:
        raise ValueError("feat size does not match dims %s" % (dims,))
    if ws.P != B * Ncam * D * fH * fW or ws.nvox != B * X * Y * Z:
        raise ValueError("workspace does not match dims")
    dev = feat.device
    if layout == BEV_NCHW_F32:
        bev = torch.empty(B, Z * C, X, Y, dtype=torch.float32, device=dev)
        out = bev
    else:
        dt = torch.float32 if layout == BEV_NHWC_F32 else torch.bfloat16
        bev = torch.empty(B, X, Y, Z * C, dtype=dt, device=dev)
        out = bev.permute(0, 3, 1, 2)
    with _timed(tag):
        N.check(N.lib().lss_lift_splat_fwd(N.ptr(feat), N.ptr(ws.vox_list), N.ptr(ws.entries),
                                           B, Ncam, D, fH, fW, C, X, Y, Z, N.ptr(bev), layout, N.stream()),
                "lss_lift_splat_fwd")
    return out


def lift_splat_forward(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, x, weight, bias, ws, dims, nx,
                       layout=BEV_NCHW_F32, math=DT_F32):
    """K3 -> K2 -> K4 -> K5 with ONE native call (inference path).  Returns (bev, depth, feat)."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    for t, name, shp in ((frustum, "frustum", (D, fH, fW, 3)), (inv_post_rots, "inv_post_rots", (B, Ncam, 3, 3)),
                         (combine, "combine", (B, Ncam, 3, 3)), (post_trans, "post_trans", (B, Ncam, 3)),
                         (trans, "trans", (B, Ncam, 3)), (dx, "dx", (3,)), (bx, "bx", (3,))):
        _f32c(t, name, shp)
    Cin = x.shape[1]
    _f32c(x, "x", (B * Ncam, Cin, fH, fW))
    w2 = weight.reshape(weight.shape[0], -1)
    _f32c(w2, "depthnet.weight", (D + C, Cin))
    _f32c(bias, "depthnet.bias", (D + C,))
    if ws.P != B * Ncam * D * fH * fW or ws.nvox != B * X * Y * Z:
        raise ValueError("workspace does not match dims")
    dev = x.device
    depth = torch.empty(B * Ncam, D, fH, fW, dtype=torch.float32, device=dev)
    feat = torch.empty(B * Ncam, fH, fW, C, dtype=torch.float32, device=dev)
    if layout == BEV_NCHW_F32:
        bev = torch.empty(B, Z * C, X, Y, dtype=torch.float32, device=dev)
        out = bev
    else:
        bev = torch.empty(B, X, Y, Z * C, dtype=torch.float32 if layout == BEV_NHWC_F32 else torch.bfloat16, device=dev)
        out = bev.permute(0, 3, 1, 2)
    _run_desc(_lift_splat_desc(ws, dims, nx, layout, math, bev, depth, feat, frustum, dx, bx, x, w2, bias, Cin,
                               calib_dev=(inv_post_rots, post_trans, combine, trans)), "lss_lift_splat_forward_desc")
    return out, depth, feat


def ffn_fused(x, w1, b1, w2, b2, ln=None):
    """y = x + b2 + W2 . gelu(W1 . x + b1) in one launch (transformer FFN, erf GELU).
    x (..., 256) bf16 contiguous; w1 (F, 256) / w2 (256, F) bf16 (packed 1x1 weights are accepted as
    (1, N, K)); b1 (F), b2 (256) fp32.  Returns fp32 of x's shape (the pre-LayerNorm sum), or - with
    ln = (gamma, beta, eps) - LayerNorm(y) * gamma + beta in bf16, normalised in the kernel's epilogue."""
    w1 = w1.reshape(w1.shape[-2], w1.shape[-1])
    w2 = w2.reshape(w2.shape[-2], w2.shape[-1])
    F, Dm = w1.shape
    if x.dtype != torch.bfloat16 or w1.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16:
        raise ValueError("ffn_fused: bf16 operands")
    if not (x.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()) or x.shape[-1] != Dm \
            or tuple(w2.shape) != (Dm, F):
        raise ValueError("ffn_fused: x (...,%d), w1 (F,%d), w2 (%d,F) contiguous" % (Dm, Dm, Dm))
    _f32c(b1, "b1", (F,))
    _f32c(b2, "b2", (Dm,))
    M = x.numel() // Dm
    gamma = beta = y = y_ln = None
    eps = 0.0
    if ln is None:
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    else:
        gamma, beta, eps = ln
        _f32c(gamma, "ln.gamma", (Dm,))
        _f32c(beta, "ln.beta", (Dm,))
        y_ln = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with _timed("ffn_fused"):
        N.check(N.lib().lss_ffn_fused_fwd(N.ptr(x), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), M, Dm, F, N.ptr(y),
                                          N.ptr(gamma), N.ptr(beta), float(eps), N.ptr(y_ln), N.stream()),
                "lss_ffn_fused_fwd")
    return y if ln is None else y_ln


def linear_res_ln(x, w, bias, residual, ln=None):
    """x . W^T + bias + residual for a 256 -> 256 layer, fp32 - or, with ln = (gamma, beta, eps), its LayerNorm
    in bf16 without the fp32 sum ever being written.  x, residual (..., 256) bf16; w (256, 256) bf16 (packed
    1x1 weights (1, N, K) accepted); bias (256) fp32."""
    w = w.reshape(w.shape[-2], w.shape[-1])
    Dm = x.shape[-1]
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or residual.dtype != torch.bfloat16:
        raise ValueError("linear_res_ln: bf16 operands")
    if not (x.is_contiguous() and w.is_contiguous() and residual.is_contiguous()) or tuple(w.shape) != (Dm, Dm) \
            or residual.shape != x.shape:
        raise ValueError("linear_res_ln: x, residual (...,%d) and w (%d,%d) contiguous" % (Dm, Dm, Dm))
    _f32c(bias, "bias", (Dm,))
    M = x.numel() // Dm
    gamma = beta = y = y_ln = None
    eps = 0.0
    if ln is None:
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    else:
        gamma, beta, eps = ln
        _f32c(gamma, "ln.gamma", (Dm,))
        _f32c(beta, "ln.beta", (Dm,))
        y_ln = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with _timed("linear_res_ln"):
        N.check(N.lib().lss_linear_res_ln_fwd(N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(residual), M, Dm, N.ptr(y),
                                              N.ptr(gamma), N.ptr(beta), float(eps), N.ptr(y_ln), N.stream()),
                "lss_linear_res_ln_fwd")
    return y if ln is None else y_ln


HOSTCAL_MAX_CAMS = 36


def lift_splat_from_heads(frustum, inv_post_rots, post_trans, combine, trans, dx, bx, depth, feat, ws, dims, nx,
                          layout=BEV_NCHW_F32):
    """Geometry + bucketing + splat of depth (B*N, D, fH, fW) / context (B*N, fH, fW, C) tensors that other kernels
    produced (the vovnet depth heads + CamEncodeV2) with ONE native call; region-bucketed pipeline when the problem
    fits it.  Returns the BEV grid (logical (B, Z*C, X, Y))."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    for t, name, shp in ((frustum, "frustum", (D, fH, fW, 3)), (inv_post_rots, "inv_post_rots", (B, Ncam, 3, 3)),
                         (combine, "combine", (B, Ncam, 3, 3)), (post_trans, "post_trans", (B, Ncam, 3)),
                         (trans, "trans", (B, Ncam, 3)), (dx, "dx", (3,)), (bx, "bx", (3,)),
                         (depth, "depth", (B * Ncam, D, fH, fW))):
        _f32c(t, name, shp)
    _f32c(feat, "feat")
    if feat.numel() != B * Ncam * fH * fW * C or C not in (64, 128):
        raise ValueError("feat must hold (B*N*fH*fW, C) context rows with C in (64, 128)")
    if ws.P != B * Ncam * D * fH * fW or ws.nvox != B * X * Y * Z:
        raise ValueError("workspace does not match dims")
    dev = feat.device
    if layout == BEV_NCHW_F32:
        bev = torch.empty(B, Z * C, X, Y, dtype=torch.float32, device=dev)
        out = bev
    else:
        bev = torch.empty(B, X, Y, Z * C, dtype=torch.float32 if layout == BEV_NHWC_F32 else torch.bfloat16, device=dev)
        out = bev.permute(0, 3, 1, 2)
    _run_desc(_lift_splat_desc(ws, dims, nx, layout, DT_F32, bev, depth, feat, frustum, dx, bx,
                               calib_dev=(inv_post_rots, post_trans, combine, trans)),
              "lss_lift_splat_forward_desc (from heads)")
    return out


def lift_splat_forward_hostcal(frustum, calib_host, dx, bx, x, weight, bias, ws, dims, nx, layout=BEV_NCHW_F32):
    """lift_splat_forward with the calibration as ONE CPU fp32 buffer of B*N*24 floats
    ([inv_post_rots | combine | post_trans | trans] = data.CalibrationPack.buffer): it is read during the
    call and rides in the kernel arguments - no H2D copy.  B*N <= HOSTCAL_MAX_CAMS; f32 depthnet math."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    if calib_host.is_cuda or calib_host.dtype != torch.float32 or not calib_host.is_contiguous() \
            or calib_host.numel() != B * Ncam * 24 or B * Ncam > HOSTCAL_MAX_CAMS:
        raise ValueError("calib_host must be a contiguous CPU fp32 buffer of B*N*24 floats (B*N <= %d)" % HOSTCAL_MAX_CAMS)
    for t, name, shp in ((frustum, "frustum", (D, fH, fW, 3)), (dx, "dx", (3,)), (bx, "bx", (3,))):
        _f32c(t, name, shp)
    Cin = x.shape[1]
    _f32c(x, "x", (B * Ncam, Cin, fH, fW))
    w2 = weight.reshape(weight.shape[0], -1)
    _f32c(w2, "depthnet.weight", (D + C, Cin))
    _f32c(bias, "depthnet.bias", (D + C,))
    if ws.P != B * Ncam * D * fH * fW or ws.nvox != B * X * Y * Z:
        raise ValueError("workspace does not match dims")
    dev = x.device
    depth = torch.empty(B * Ncam, D, fH, fW, dtype=torch.float32, device=dev)
    feat = torch.empty(B * Ncam, fH, fW, C, dtype=torch.float32, device=dev)
    if layout == BEV_NCHW_F32:
        bev = torch.empty(B, Z * C, X, Y, dtype=torch.float32, device=dev)
        out = bev
    else:
        bev = torch.empty(B, X, Y, Z * C, dtype=torch.float32 if layout == BEV_NHWC_F32 else torch.bfloat16, device=dev)
        out = bev.permute(0, 3, 1, 2)
    _run_desc(_lift_splat_desc(ws, dims, nx, layout, DT_F32, bev, depth, feat, frustum, dx, bx, x, w2, bias, Cin,
                               calib_host=calib_host), "lss_lift_splat_forward_desc (host calibration)")
    return out, depth, feat


def lift_splat_bwd(grad_bev, voxel, depth, feat, dims, nx):
    """K7.  grad_bev logical (B, Z*C, X, Y): fp32 contiguous or channels_last, or bf16 channels_last (the gradient a
    bf16 stem hands back: read as it is, no fp32 copy).  Returns g_logits (BN, D+C, fH, fW)."""
    B, Ncam, D, fH, fW, C = dims
    X, Y, Z = nx
    if tuple(grad_bev.shape) != (B, Z * C, X, Y):
        raise ValueError("grad_bev must be (B, Z*C, X, Y)")
    if grad_bev.dtype == torch.bfloat16 and grad_bev.is_contiguous(memory_format=torch.channels_last):
        layout = BEV_NHWC_BF16
    else:
        if grad_bev.dtype != torch.float32:
            grad_bev = grad_bev.float()
        if grad_bev.is_contiguous():
            layout = BEV_NCHW_F32
        elif grad_bev.is_contiguous(memory_format=torch.channels_last):
            layout = BEV_NHWC_F32
        else:
            grad_bev = grad_bev.contiguous()
            layout = BEV_NCHW_F32
    g_logits = torch.empty(B * Ncam, D + C, fH, fW, dtype=torch.float32, device=grad_bev.device)
    N.check(N.lib().lss_lift_splat_bwd(N.ptr(grad_bev), layout, N.ptr(voxel), N.ptr(depth), N.ptr(feat),
                                       B, Ncam, D, fH, fW, C, X, Y, Z, N.ptr(g_logits), N.stream()),
            "lss_lift_splat_bwd")
    return g_logits


def segmented_sum(x, seg_start):
    """x (K,C) fp32, seg_start (M+1) int32 -> (M,C)."""
    _f32c(x, "x")
    M = seg_start.numel() - 1
    y = torch.empty(M, x.shape[1], dtype=torch.float32, device=x.device)
    if M > 0:
        N.check(N.lib().lss_segmented_sum(N.ptr(x), N.ptr(seg_start), M, x.shape[1], N.ptr(y), N.stream()),
                "lss_segmented_sum")
    return y


_TORCH_DT = {DT_F32: torch.float32, DT_BF16: torch.bfloat16}


def add_pos(x, pos):
    """q = x + pos: x (B, ..., 256) token rows in fp32|bf16, pos (T, 256) fp32 with T = tokens per sample."""
    if not x.is_contiguous() or x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("x must be contiguous fp32/bf16")
    _f32c(pos, "pos")
    B, C = x.shape[0], x.shape[-1]
    T = x.numel() // (B * C)
    if tuple(pos.shape) != (T, C):
        raise ValueError("pos %s does not match %d tokens x %d channels" % (tuple(pos.shape), T, C))
    q = torch.empty_like(x)
    with _timed("add_pos"):
        N.check(N.lib().lss_add_pos_fwd(N.ptr(x), N.ptr(pos), B, T, C, DT_F32 if x.dtype == torch.float32 else DT_BF16,
                                        N.ptr(q), N.stream()), "lss_add_pos_fwd")
    return q


def deform_attn(value, offsets_logits, ref_x, ref_y, n_heads=8, n_points=8, token_bias=None):
    """value fp32|bf16: (B,H,W,256) NHWC, or head-major (B,8,H*W,32) (conv2d_nhwc(head_major=True));
    offsets_logits (B,H,W,192) fp32; token_bias (H*W,192) fp32 or None (per-token addend shared by
    all samples) -> (B,H,W,256) NHWC in value's dtype."""
    B, H, W = offsets_logits.shape[:3]
    C = n_heads * 32
    if tuple(value.shape) == (B, H, W, C):
        layout = VALUE_NHWC
    elif tuple(value.shape) == (B, n_heads, H * W, 32):
        layout = VALUE_HEAD_MAJOR
    else:
        raise ValueError("value %s is neither (B,H,W,%d) nor (B,%d,H*W,32)" % (tuple(value.shape), C, n_heads))
    if not value.is_contiguous() or value.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("value must be contiguous fp32/bf16")
    _f32c(offsets_logits, "offsets_logits", (B, H, W, n_heads * n_points * 3))
    _f32c(ref_x, "ref_x", (W,))
    _f32c(ref_y, "ref_y", (H,))
    if token_bias is not None:
        _f32c(token_bias, "token_bias", (H * W, n_heads * n_points * 3))
    out = torch.empty(B, H, W, C, dtype=value.dtype, device=value.device)
    with _timed("deform_attn"):
        N.check(N.lib().lss_deform_attn_fwd(N.ptr(value), layout, N.ptr(offsets_logits), N.ptr(token_bias), N.ptr(ref_x),
                                            N.ptr(ref_y), B, H, W,
                                            n_heads, n_points, C, DT_F32 if value.dtype == torch.float32 else DT_BF16,
                                            N.ptr(out), N.stream()), "lss_deform_attn_fwd")
    return out


def layernorm(x, gamma, beta, eps, out_dtype):
    """nn.LayerNorm over the last (256-wide) dim of contiguous fp32|bf16 rows."""
    if not x.is_contiguous() or x.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("x must be contiguous fp32/bf16")
    C = x.shape[-1]
    _f32c(gamma, "gamma", (C,))
    _f32c(beta, "beta", (C,))
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    code = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}
    with _timed("layernorm"):
        N.check(N.lib().lss_layernorm_fwd(N.ptr(x), code[x.dtype], N.ptr(gamma), N.ptr(beta), x.numel() // C, C,
                                          float(eps), N.ptr(y), code[out_dtype], N.stream()), "lss_layernorm_fwd")
    return y


def pack_conv_weight(w_oihw, dt):
    """OIHW fp32 -> the conv kernels' [tap][Cout][Cin] layout in `dt`."""
    Cout, Cin, KH, KW = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    out = torch.empty(KH * KW, Cout, Cin, dtype=_TORCH_DT[dt], device=w_oihw.device)
    nbytes = N.lib().lss_conv2d_packed_weight_bytes(Cout, Cin, KH, KW, dt)
    assert nbytes == out.numel() * out.element_size()
    N.check(N.lib().lss_conv2d_pack_weights(N.ptr(w_oihw), Cout, Cin, KH, KW, dt, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights")
    return out


class RingWeight:
    """3x3 weights in the loader / consumer ring kernel's layout (csrc/conv_ring.hip): `data` is the flat bf16
    image, `Cout`, `Cin` the logical shape."""

    def __init__(self, data, Cout, Cin):
        self.data, self.Cout, self.Cin = data, Cout, Cin


class KsWeight(RingWeight):
    """3x3 weights in the layout of the K-split one-pass kernel (csrc/conv_ks.hip)."""


def conv_ks_ok(B, H, W, Cin, Cout):
    """Is this 3x3 / stride-1 / pad-1 bf16 conv a case for the K-split one-pass kernel (lss_conv2d_ks_ok)?"""
    return bool(N.lib().lss_conv2d_ks_ok(B, H, W, Cin, Cout))


def pack_conv_weight_ks(w_oihw, dgrad=False):
    """OIHW fp32 (3x3) -> KsWeight; dgrad=True: the weights of the layer's input-gradient conv (Cout -> Cin channels,
    transposed and tap-flipped), as the training units pack them."""
    Cout, Cin, KH, KW = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    co, ci = (Cin, Cout) if dgrad else (Cout, Cin)   # channels of the conv the image is for
    nbytes = N.lib().lss_conv2d_ks_packed_weight_bytes(co, ci)
    if (KH, KW) != (3, 3) or nbytes == 0:
        raise ValueError("KS weights: 3x3, Cout %% 32 == 0, Cin in (64, 128, 256) (got %s%s)"
                         % (tuple(w_oihw.shape), ", dgrad" if dgrad else ""))
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_oihw.device)
    fn = N.lib().lss_conv2d_pack_weights_ks_dgrad if dgrad else N.lib().lss_conv2d_pack_weights_ks
    N.check(fn(N.ptr(w_oihw), Cout, Cin, N.ptr(out), N.stream()), "lss_conv2d_pack_weights_ks")
    return KsWeight(out, co, ci)


def conv_ring_ok(B, H, W, Cx, C2, up, Cout, head_n=0):
    """Is this 3x3 / stride-1 / pad-1 bf16 conv a case for the ring kernel (lss_conv2d_ring_ok)?"""
    if os.environ.get("LSS_CONV_RING") == "0":
        return False
    return bool(N.lib().lss_conv2d_ring_ok(B, H, W, Cx, C2, up, Cout, head_n))


def pack_conv_weight_ring(w_oihw):
    """OIHW fp32 (3x3) -> RingWeight."""
    Cout, Cin, KH, KW = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    nbytes = N.lib().lss_conv2d_ring_packed_weight_bytes(Cout, Cin)
    if (KH, KW) != (3, 3) or nbytes == 0:
        raise ValueError("ring weights: 3x3, Cout % 128 == 0, Cin % 32 == 0 (got %s)" % (tuple(w_oihw.shape),))
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_oihw.device)
    N.check(N.lib().lss_conv2d_pack_weights_ring(N.ptr(w_oihw), Cout, Cin, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights_ring")
    return RingWeight(out, Cout, Cin)


def pack_conv_weight_ring_dgrad(w_oihw):
    """OIHW fp32 (3x3) -> RingWeight of the conv that maps dY (Cout channels) to dX (Cin channels)."""
    Cout, Cin, KH, KW = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    nbytes = N.lib().lss_conv2d_ring_packed_weight_bytes(Cin, Cout)
    if (KH, KW) != (3, 3) or nbytes == 0:
        raise ValueError("ring dgrad weights: 3x3, Cin % 128 == 0, Cout % 32 == 0 (got %s)" % (tuple(w_oihw.shape),))
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_oihw.device)
    N.check(N.lib().lss_conv2d_pack_weights_ring_dgrad(N.ptr(w_oihw), Cout, Cin, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights_ring_dgrad")
    return RingWeight(out, Cin, Cout)


def conv2d_nhwc(x, w_packed, ksize, stride, pad, scale=None, shift=None, residual=None, relu=False,
                x2=None, up=1, stats=None, dt=DT_BF16, tag="conv2d_fwd", out_f32=False, head_major=False):
    """K8.  x (B,H,W,Cx) NHWC in `dt`; x2 (B,H*up,W*up,C2) optional skip tensor
    (conv input = cat([x2, upsample(x, up)])).  Returns y (B,Ho,Wo,Cout) in `dt`
    (fp32 with out_f32).  relu: False/True or an ACT_* code (ACT_GELU = erf GELU).
    head_major (1x1 bf16 convs): y is laid out (B, Cout/32, Ho*Wo, 32) and returned with that shape."""
    tdt = _TORCH_DT[dt]
    act = int(relu) | (OUT_F32 if (out_f32 and dt == DT_BF16) else 0) | (OUT_HEAD_MAJOR32 if head_major else 0)
    B, H, W, Cx = x.shape
    KH, KW = ksize
    if isinstance(w_packed, RingWeight):  # the ring / KS kernels' layouts (3x3 / s1 / p1 bf16 only; the C side checks)
        act |= W_KS if isinstance(w_packed, KsWeight) else W_RING
        taps, Cout, Cin, w_packed = 9, w_packed.Cout, w_packed.Cin, w_packed.data
    else:
        taps, Cout, Cin = w_packed.shape
    C2 = 0
    if x.dtype != tdt or not x.is_contiguous() or w_packed.dtype != tdt or not w_packed.is_contiguous():
        raise ValueError("conv operands must be contiguous %s" % tdt)
    if x2 is not None:
        if x2.dtype != tdt or not x2.is_contiguous() or tuple(x2.shape[:3]) != (B, H * up, W * up):
            raise ValueError("x2 must be contiguous %s of shape (B,H*up,W*up,C2)" % tdt)
        C2 = x2.shape[3]
    if taps != KH * KW or Cin != Cx + C2:
        raise ValueError("packed weight %s does not match Cin=%d taps=%d" % (tuple(w_packed.shape), Cx + C2, KH * KW))
    Ho = (H * up + 2 * pad - KH) // stride + 1
    Wo = (W * up + 2 * pad - KW) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32 if out_f32 else tdt, device=x.device)
    for name, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _f32c(t, name, (Cout,))
    if residual is not None and (residual.dtype != tdt or tuple(residual.shape) != (B, Ho, Wo, Cout)
                                 or not residual.is_contiguous()):
        raise ValueError("residual must match the output")
    if stats is not None:
        _f32c(stats, "stats", (2 * Cout,))
    if _recorder is not None:
        _recorder.add(0, (x, x2, w_packed, scale, shift, residual, y, stats), x=x, x2=x2, w=w_packed, scale=scale,
                      shift=shift, residual=residual, y=y, stats=stats, B=B, H=H, W=W, Cx=Cx, C2=C2, up=up, Cout=Cout,
                      KH=KH, KW=KW, stride=stride, pad=pad, relu=act, dt=dt)
    with _timed(tag):
        N.check(N.lib().lss_conv2d_fwd(N.ptr(x), N.ptr(x2), N.ptr(w_packed), N.ptr(scale), N.ptr(shift),
                                       N.ptr(residual), N.ptr(y), N.ptr(stats), B, H, W, Cx, C2, up, Cout,
                                       KH, KW, stride, pad, act, dt, N.stream()), "lss_conv2d_fwd")
    return y.view(B, Cout // 32, Ho * Wo, 32) if head_major else y


def pack_conv_weight_dgrad(w_oihw, dt):
    """OIHW fp32 -> [tap'][Cin][Cout] in dt: the weight of the conv that maps dY to dX
    (use with conv2d_nhwc(dy, w, (KH,KW), 1, KH-1-pad))."""
    Cout, Cin, KH, KW = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    out = torch.empty(KH * KW, Cin, Cout, dtype=_TORCH_DT[dt], device=w_oihw.device)
    N.check(N.lib().lss_conv2d_pack_weights_dgrad(N.ptr(w_oihw), Cout, Cin, KH, KW, dt, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights_dgrad")
    return out


_wgrad_ws = {}


def conv3x3_wgrad(x, dy):
    """Weight gradient of a 3x3/s1/p1 conv.  x (B,H,W,Cin), dy (B,H,W,Cout) contiguous bf16 NHWC
    -> dW (Cout,Cin,3,3) fp32.  The workspace is cached per shape and device."""
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    for t, nm in ((x, "x"), (dy, "dy")):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("%s must be a contiguous bf16 NHWC GPU tensor" % nm)
    if tuple(dy.shape[:3]) != (B, H, W):
        raise ValueError("dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    nbytes = N.lib().lss_conv2d_wgrad_workspace_bytes(B, H, W, Cin, Cout)
    key = (B, H, W, Cin, Cout, str(x.device))
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=x.device)
    with _timed("conv2d_wgrad"):
        N.check(N.lib().lss_conv2d_wgrad(N.ptr(x), N.ptr(dy), B, H, W, Cin, Cout, N.ptr(ws), nbytes, N.ptr(dw),
                                         N.stream()), "lss_conv2d_wgrad")
    return dw


def pack_conv_weight_s2_dgrad(w_oihw, pad):
    """OIHW fp32 of a stride-2 K x K conv -> (KT*KT, 4*Cin, Cout) bf16: the weight of the stride-1 conv that maps dY to
    the four phase planes of dX (lss_conv2d_pack_weights_s2_dgrad).  Returns (packed, KT)."""
    Cout, Cin, K, _ = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    KT = N.lib().lss_conv2d_s2_dgrad_taps(K, pad)
    if KT == 0:
        raise ValueError("stride-2 data gradient: K / pad in (1, 0), (3, 1), (7, 3)")
    out = torch.empty(KT * KT, 4 * Cin, Cout, dtype=torch.bfloat16, device=w_oihw.device)
    N.check(N.lib().lss_conv2d_pack_weights_s2_dgrad(N.ptr(w_oihw), Cout, Cin, K, pad, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights_s2_dgrad")
    return out, KT


def conv4x4_wgrad(xs, dy):
    """Weight gradient of the 4x4-tap stride-1 conv with taps (dy, dx) in {-2 .. 1}^2 (a 7x7 / 2 / pad-3 conv over
    phase planes): xs (B,H,W,Cin), dy (B,H,W,Cout) contiguous bf16 NHWC -> (Cout, Cin, 4, 4) fp32, tap [dy + 2][dx + 2]."""
    B, H, W, Cin = xs.shape
    Cout = dy.shape[3]
    for t, nm in ((xs, "xs"), (dy, "dy")):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("%s must be a contiguous bf16 NHWC GPU tensor" % nm)
    if tuple(dy.shape[:3]) != (B, H, W):
        raise ValueError("dy %s does not match xs %s" % (tuple(dy.shape), tuple(xs.shape)))
    nbytes = N.lib().lss_conv2d_wgrad4x4_workspace_bytes(B, H, W, Cin, Cout)
    if nbytes == 0:
        raise ValueError("conv4x4_wgrad: channel counts must be multiples of 64 and 8 <= W <= 224")
    key = ("4x4", B, H, W, Cin, Cout, str(xs.device))
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=xs.device)
    dw = torch.empty(Cout, Cin, 4, 4, dtype=torch.float32, device=xs.device)
    with _timed("conv2d_wgrad"):
        N.check(N.lib().lss_conv2d_wgrad4x4(N.ptr(xs), N.ptr(dy), B, H, W, Cin, Cout, N.ptr(ws), nbytes, N.ptr(dw),
                                            N.stream()), "lss_conv2d_wgrad4x4")
    return dw


def upsample_cat_nhwc(x, x2, up):
    """[x2 | bilinear_align_corners(x, up)] as a contiguous bf16 (B, H*up, W*up, C2+Cx) tensor."""
    B, H, W, Cx = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    for t in (x, x2):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous()):
            raise ValueError("upsample_cat_nhwc operands must be contiguous bf16 NHWC")
    if x2 is not None and tuple(x2.shape[:3]) != (B, H * up, W * up):
        raise ValueError("x2 %s does not match x %s upsampled x%d" % (tuple(x2.shape), tuple(x.shape), up))
    out = torch.empty(B, H * up, W * up, C2 + Cx, dtype=torch.bfloat16, device=x.device)
    with _timed("upsample_cat"):
        N.check(N.lib().lss_upsample_cat_nhwc(N.ptr(x), N.ptr(x2), B, H, W, Cx, C2, up, N.ptr(out), N.stream()),
                "lss_upsample_cat_nhwc")
    return out


def upsample_bwd_nhwc(g, c_off, Cx, up):
    """Adjoint of the bilinear (align_corners) x`up` upsample applied to channels [c_off, c_off+Cx)
    of g (B, H*up, W*up, Ct) bf16 -> (B, H, W, Cx) bf16."""
    B, Hh, Wh, Ct = g.shape
    if g.dtype != torch.bfloat16 or not g.is_contiguous() or Hh % up or Wh % up:
        raise ValueError("g must be contiguous bf16 NHWC with sizes divisible by the scale")
    dx = torch.empty(B, Hh // up, Wh // up, Cx, dtype=torch.bfloat16, device=g.device)
    with _timed("upsample_bwd"):
        N.check(N.lib().lss_upsample_bwd_nhwc(N.ptr(g), B, Hh // up, Wh // up, Cx, Ct, c_off, up, N.ptr(dx),
                                              N.stream()), "lss_upsample_bwd_nhwc")
    return dx


_bn_ws = {}


def _bn_workspace(M, C, device):
    key = (M, C, str(device))
    ws = _bn_ws.get(key)
    if ws is None:
        nbytes = N.lib().lss_bn_train_workspace_bytes(M, C)
        if nbytes == 0:
            raise ValueError("BatchNorm over %d rows x %d channels is outside the kernels' range" % (M, C))
        ws = _bn_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def bn_train_fwd(z, gamma, beta, running_mean, running_var, momentum, eps, relu, residual=None):
    """Training-mode BatchNorm (+residual, +ReLU) on contiguous bf16 NHWC z (..., C).
    Returns y (bf16, like z), save_mean, save_invstd (fp32); running stats updated in place."""
    C = z.shape[-1]
    M = z.numel() // C
    for t, nm in ((z, "z"), (residual, "residual")):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or tuple(t.shape) != tuple(z.shape)):
            raise ValueError("%s must be contiguous bf16 shaped like z" % nm)
    _f32c(gamma, "gamma", (C,))
    _f32c(beta, "beta", (C,))
    if running_mean is not None:
        _f32c(running_mean, "running_mean", (C,))
        _f32c(running_var, "running_var", (C,))
    y = torch.empty_like(z)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    invstd = torch.empty(C, dtype=torch.float32, device=z.device)
    with _timed("bn_train_fwd"):
        N.check(N.lib().lss_bn_train_fwd(N.ptr(z), N.ptr(residual), M, C, N.ptr(gamma), N.ptr(beta), N.ptr(running_mean),
                                         N.ptr(running_var), float(momentum), float(eps), 1 if relu else 0,
                                         N.ptr(_bn_workspace(M, C, z.device)), N.ptr(y), N.ptr(mean), N.ptr(invstd),
                                         N.stream()), "lss_bn_train_fwd")
    return y, mean, invstd


def bn_train_bwd(dy, y, z, gamma, mean, invstd, relu, want_dres):
    """Backward of bn_train_fwd: returns dz (bf16), dres (bf16 or None), dgamma, dbeta (fp32)."""
    C = z.shape[-1]
    M = z.numel() // C
    for t, nm in ((dy, "dy"), (z, "z"), (y, "y")):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or tuple(t.shape) != tuple(z.shape)):
            raise ValueError("%s must be contiguous bf16 shaped like z" % nm)
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_dres else None
    dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    with _timed("bn_train_bwd"):
        N.check(N.lib().lss_bn_train_bwd(N.ptr(dy), N.ptr(y), N.ptr(z), M, C, N.ptr(gamma), N.ptr(mean), N.ptr(invstd),
                                         1 if relu else 0, N.ptr(_bn_workspace(M, C, z.device)), N.ptr(dz), N.ptr(dres),
                                         N.ptr(dgamma), N.ptr(dbeta), N.stream()), "lss_bn_train_bwd")
    return dz, dres, dgamma, dbeta


def bn_partial_sums(z, mode, dy=None, y=None, mean=None, invstd=None, relu=False):
    """(2, C) fp32 per-channel sums of this rank's rows: mode 0 = (sum z, sum z^2); mode 1 = (sum g, sum g*xhat)."""
    C = z.shape[-1]
    M = z.numel() // C
    sums = torch.empty(2, C, dtype=torch.float32, device=z.device)
    N.check(N.lib().lss_bn_partial_sums(N.ptr(z), N.ptr(dy), N.ptr(y), N.ptr(mean), N.ptr(invstd), M, C,
                                        1 if relu else 0, mode, N.ptr(_bn_workspace(M, C, z.device)), N.ptr(sums),
                                        N.stream()), "lss_bn_partial_sums")
    return sums


def bn_train_fwd_from_sums(z, sums, m_total, gamma, beta, running_mean, running_var, momentum, eps, relu, residual=None):
    C = z.shape[-1]
    M = z.numel() // C
    y = torch.empty_like(z)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    invstd = torch.empty(C, dtype=torch.float32, device=z.device)
    N.check(N.lib().lss_bn_train_fwd_from_sums(N.ptr(z), N.ptr(residual), M, C, N.ptr(sums), int(m_total), N.ptr(gamma),
                                               N.ptr(beta), N.ptr(running_mean), N.ptr(running_var), float(momentum),
                                               float(eps), 1 if relu else 0, N.ptr(_bn_workspace(M, C, z.device)),
                                               N.ptr(y), N.ptr(mean), N.ptr(invstd), N.stream()),
            "lss_bn_train_fwd_from_sums")
    return y, mean, invstd


def bn_train_bwd_from_sums(dy, y, z, sums, m_total, gamma, mean, invstd, relu, want_dres):
    C = z.shape[-1]
    M = z.numel() // C
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_dres else None
    scratch = torch.empty(2, C, dtype=torch.float32, device=z.device)  # the global dgamma / dbeta (unused by DP callers)
    N.check(N.lib().lss_bn_train_bwd_from_sums(N.ptr(dy), N.ptr(y), N.ptr(z), M, C, N.ptr(sums), int(m_total),
                                               N.ptr(gamma), N.ptr(mean), N.ptr(invstd), 1 if relu else 0,
                                               N.ptr(_bn_workspace(M, C, z.device)), N.ptr(dz), N.ptr(dres),
                                               scratch.data_ptr(), scratch.data_ptr() + 4 * C, N.stream()),
            "lss_bn_train_bwd_from_sums")
    return dz, dres


def _p(t):
    return None if t is None else t.data_ptr()


class _GatherJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_longlong)]


class WeightPrepack:
    """Every packed weight image of a training step in ONE launch (csrc/layout.hip: lss_gather_pack).

    The weights change every step, so every conv unit packs its bf16 image(s) - tile / ring / K-split / phase-plane
    layout, forward and input-gradient form - on every call: 36 small launches per step of the benched model.  A packed
    image is a permutation of the layer's fp32 weights plus zeros, so it can be produced by a table-driven gather that
    knows nothing about layouts.  A unit REGISTERS its image the first time it packs one (eager warm-up steps): the
    table comes from pushing the three byte planes of the element numbers 1..n through the unit's own pack routine
    (values <= 255 are exact in bf16), and is checked once against that routine on the real weights.  From then on
    `run(params)` - called by dp.train_step* at the top of a step with the step's trainable tensors - fills their
    registered images with one launch (images of tensors that are gone are dropped, not read), and the units find
    theirs through `lookup` instead of packing; `invalidate()` after the optimizer step makes them pack
    themselves again unless another `run()` comes first, so a caller with its own loop is never handed a stale image.
    Keys are (weight address, shape, kind): whatever lives at that address is what gets packed.  LSS_PREPACK=0 disables."""

    def __init__(self):
        self.jobs = {}
        self._table = None
        self._tls = threading.local()   # `fresh` belongs to the thread that ran the step's `run()`

    @property
    def fresh(self):
        return getattr(self._tls, "fresh", False)

    @fresh.setter
    def fresh(self, v):
        self._tls.fresh = bool(v)

    @staticmethod
    def enabled():
        return os.environ.get("LSS_PREPACK", "1") != "0"

    @staticmethod
    def _key(w, kind):
        return (w.data_ptr(), tuple(w.shape), str(w.device)) + tuple(kind)

    def lookup(self, w, kind):
        if not self.fresh:
            return None
        j = self.jobs.get(self._key(w, kind))
        return None if j is None else j[2]

    def register(self, w, kind, pack_fn):
        """pack_fn(fp32 tensor shaped like w) -> the bf16 image (any shape); no-op while a stream capture is active."""
        if not self.enabled() or not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous():
            return
        if torch.cuda.is_current_stream_capturing() or w.numel() >= (1 << 24):
            return
        key = self._key(w, kind)
        if key in self.jobs:
            return
        n = w.numel()
        num = torch.arange(1, n + 1, device=w.device, dtype=torch.int64).view(w.shape)
        idx = None
        for b in range(3):
            img = pack_fn(((num >> (8 * b)) & 255).float())
            v = img.reshape(-1).float().to(torch.int64) << (8 * b)
            idx = v if idx is None else idx | v
        ref = pack_fn(w.detach())
        dst = torch.empty_like(ref)
        got = torch.where(idx > 0, w.detach().reshape(-1)[(idx - 1).clamp(min=0)], torch.zeros((), device=w.device))
        if not torch.equal(got.to(torch.bfloat16), ref.reshape(-1)):
            raise N.LssNativeError("WeightPrepack: the image of kind %s is not a permutation of its weights" % (kind,))
        self.jobs[key] = (w.data_ptr(), idx.to(torch.int32).contiguous(), dst)
        self._table = None

    def run(self, params):
        """Fill the registered images of `params` (the step's trainable tensors) from their current values: one launch
        per 96 images.  Images registered for tensors that are not among `params` any more - a model that has been
        deleted: its memory may be unmapped by now - are dropped, never read."""
        if not self.jobs or not self.enabled():
            return
        live = {p.data_ptr() for p in params}
        if any(j[0] not in live for j in self.jobs.values()):
            self.jobs = {k: j for k, j in self.jobs.items() if j[0] in live}
            self._table = None
            if not self.jobs:
                return
        if self._table is None:
            tab = (_GatherJob * len(self.jobs))()
            for i, (src, idx, dst) in enumerate(self.jobs.values()):
                tab[i].src, tab[i].idx, tab[i].dst, tab[i].n = src, idx.data_ptr(), dst.data_ptr(), idx.numel()
            self._table = tab
        N.check(N.lib().lss_gather_pack(ctypes.cast(self._table, ctypes.c_void_p), len(self.jobs), N.stream()),
                "lss_gather_pack")
        self.fresh = True

    def invalidate(self):
        self.fresh = False


prepack = WeightPrepack()


def prepacked(w, kind, pack_fn):
    """The bf16 image of weights `w` a unit needs: the step's pre-packed one when `prepack.run()` has filled it,
    else packed now (and registered for the next steps)."""
    pk = prepack.lookup(w, kind)
    if pk is not None:
        return pk
    out = pack_fn(w)
    prepack.register(w, kind, pack_fn)
    return out


def _unit_pack_fn(B, H, W, Cx, C2, up, Cout, dgrad):
    def fn(w):
        out = torch.empty(9 * Cout * (Cx + C2), dtype=torch.bfloat16, device=w.device)
        N.check(N.lib().lss_conv_bn_act_train_pack(w.data_ptr(), B, H, W, Cx, C2, up, Cout, dgrad, out.data_ptr(),
                                                   N.stream()), "lss_conv_bn_act_train_pack")
        return out
    return fn


def conv_bn_act_train_fwd(x1n, x2n, weight, gamma, beta, resn, running_mean, running_var, momentum, eps, relu, up):
    """One host call: conv3x3 (fused upsample/concat input) -> BN(train) -> (+res) -> ReLU.  Operands are
    trusted (contiguous bf16 NHWC activations, fp32 parameters): the caller is modules._ConvBNActFn.
    Returns z, y (bf16 NHWC), save_mean, save_invstd."""
    B, H, W, Cx = x1n.shape
    C2 = 0 if x2n is None else x2n.shape[3]
    Cout = weight.shape[0]
    dev = x1n.device
    z = torch.empty(B, H * up, W * up, Cout, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(z)
    kind = ("unit", 0, B, H, W, Cx, C2, up)
    wp = prepack.lookup(weight, kind)          # filled by prepack.run() at the top of the step, or ...
    wsrc = None
    if wp is None:                             # ... packed inside the call (and registered for the next steps)
        wp = torch.empty(9 * Cout * (Cx + C2), dtype=torch.bfloat16, device=dev)
        wsrc = weight
        prepack.register(weight, kind, _unit_pack_fn(B, H, W, Cx, C2, up, Cout, 0))
    stat = torch.empty(2, Cout, dtype=torch.float32, device=dev)
    ws = _bn_workspace(B * H * up * W * up, Cout, dev)
    with _timed("conv_bn_act_train_fwd"):
        N.check(N.lib().lss_conv_bn_act_train_fwd(
            x1n.data_ptr(), _p(x2n), _p(wsrc), gamma.data_ptr(), beta.data_ptr(), _p(resn), _p(running_mean),
            _p(running_var), wp.data_ptr(), z.data_ptr(), y.data_ptr(), stat.data_ptr(), stat.data_ptr() + 4 * Cout,
            ws.data_ptr(), B, H, W, Cx, C2, up, Cout, momentum, eps, 1 if relu else 0, N.stream()),
            "lss_conv_bn_act_train_fwd")
    return z, y, stat


def conv_bn_act_train_bwd(gyn, y, z, x1n, x2n, weight, gamma, stat, relu, up, want_res, want_x1, want_x2, want_w):
    """Backward of conv_bn_act_train_fwd in one host call.  Returns (g1, g2, gw, dgamma, dbeta, dres)."""
    B, H, W, Cx = x1n.shape
    C2 = 0 if x2n is None else x2n.shape[3]
    Cout, Ct = weight.shape[0], Cx + C2
    Hh, Wh = H * up, W * up
    dev = x1n.device
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_res else None
    dgb = torch.empty(2, Cout, dtype=torch.float32, device=dev)
    plain = up == 1 and C2 == 0
    gcat = g1 = xcat = gw = wd = wws = wsrc = None   # wsrc: the fp32 weights, when the call has to pack the dgrad image
    nws = 0
    if want_x1 or want_x2:
        gcat = torch.empty(B, Hh, Wh, Ct, dtype=torch.bfloat16, device=dev)
        kind = ("unit", 1, B, H, W, Cx, C2, up)
        wd = prepack.lookup(weight, kind)
        if wd is None:
            wd = torch.empty(9 * Cout * Ct, dtype=torch.bfloat16, device=dev)
            wsrc = weight
            prepack.register(weight, kind, _unit_pack_fn(B, H, W, Cx, C2, up, Cout, 1))
        if want_x1 and not plain:
            g1 = torch.empty(B, H, W, Cx, dtype=torch.bfloat16, device=dev)
    if want_w:
        gw = torch.empty(Cout, Ct, 3, 3, dtype=torch.float32, device=dev)
        if not plain:
            xcat = torch.empty(B, Hh, Wh, Ct, dtype=torch.bfloat16, device=dev)
        nws = N.lib().lss_conv2d_wgrad_workspace_bytes(B, Hh, Wh, Ct, Cout)
        key = (B, Hh, Wh, Ct, Cout, str(dev))
        wws = _wgrad_ws.get(key)
        if wws is None:
            wws = _wgrad_ws[key] = torch.empty(nws, dtype=torch.uint8, device=dev)
    ws = _bn_workspace(B * Hh * Wh, Cout, dev)
    with _timed("conv_bn_act_train_bwd"):
        N.check(N.lib().lss_conv_bn_act_train_bwd(
            gyn.data_ptr(), y.data_ptr(), z.data_ptr(), x1n.data_ptr(), _p(x2n), _p(wsrc), gamma.data_ptr(),
            stat.data_ptr(), stat.data_ptr() + 4 * Cout, ws.data_ptr(), _p(wws), nws, _p(wd), dz.data_ptr(), _p(dres),
            dgb.data_ptr(), dgb.data_ptr() + 4 * Cout, _p(gcat), _p(g1), _p(xcat), _p(gw), B, H, W, Cx, C2, up, Cout,
            1 if relu else 0, N.stream()), "lss_conv_bn_act_train_bwd")
    if want_x1 and plain:
        g1 = gcat
    g2 = gcat[..., :C2] if (want_x2 and C2 > 0) else None
    return g1, g2, gw, dgb[0], dgb[1], dres


def weighted_ce_fwd(logits, target, weight):
    """loss (1-element fp32 tensor) and the saved sums for weighted_ce_bwd.  logits (B,C,...) fp32
    contiguous, target (B,...) int64 contiguous, weight (C) fp32."""
    B, C = logits.shape[:2]
    _f32c(logits, "logits")
    _f32c(weight, "weight", (C,))
    if target.dtype != torch.int64 or not target.is_contiguous() or target.numel() * C != logits.numel():
        raise ValueError("target must be contiguous int64 of shape (B, ...) matching the logits")
    HW = logits.numel() // (B * C)
    ws = torch.empty(512 + 3, dtype=torch.float32, device=logits.device)
    with _timed("weighted_ce_fwd"):
        N.check(N.lib().lss_weighted_ce_fwd(N.ptr(logits), N.ptr(target), N.ptr(weight), B, C, HW, ws.data_ptr(),
                                            ws.data_ptr() + 4 * 512, ws.data_ptr() + 4 * 514, N.stream()),
                "lss_weighted_ce_fwd")
    return ws[514:515].view(()), ws[512:514]


def weighted_ce_bwd(logits, target, weight, sums, grad_loss):
    B, C = logits.shape[:2]
    HW = logits.numel() // (B * C)
    g = torch.empty_like(logits)
    gl = grad_loss.reshape(1).float().contiguous()
    with _timed("weighted_ce_bwd"):
        N.check(N.lib().lss_weighted_ce_bwd(N.ptr(logits), N.ptr(target), N.ptr(weight), B, C, HW, N.ptr(sums),
                                            N.ptr(gl), N.ptr(g), N.stream()), "lss_weighted_ce_bwd")
    return g


_head_ce_ws = {}


def _head_ce_workspace(K, device):
    key = (K, str(device))
    ws = _head_ce_ws.get(key)
    if ws is None:
        nbytes = N.lib().lss_head_ce_workspace_bytes(K)
        if nbytes == 0:
            raise ValueError("fused head + cross-entropy supports 4 or 8 classes")
        ws = _head_ce_ws[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    return ws


def head_ce_fwd(y, head_w, head_b, target, class_w):
    """Fused 1x1 head + weighted cross-entropy.  y (..., 128) contiguous bf16 NHWC rows; head_w (K, 128) fp32;
    head_b (K); target int64 with y's leading shape; class_w (K).  Returns (loss 0-d tensor, sums (2,))."""
    K, Cin = head_w.shape
    if y.dtype != torch.bfloat16 or not y.is_contiguous() or y.shape[-1] != Cin:
        raise ValueError("y must be contiguous bf16 rows of %d channels" % Cin)
    _f32c(head_w, "head_w", (K, Cin))
    _f32c(head_b, "head_b", (K,))
    _f32c(class_w, "class_w", (K,))
    M = y.numel() // Cin
    if target.dtype != torch.int64 or not target.is_contiguous() or target.numel() != M or not target.is_cuda:
        raise ValueError("target must be a contiguous int64 GPU tensor with one entry per pixel")
    out = torch.empty(3, dtype=torch.float32, device=y.device)
    with _timed("head_ce_fwd"):
        N.check(N.lib().lss_head_ce_fwd(N.ptr(y), N.ptr(head_w), N.ptr(head_b), N.ptr(target), N.ptr(class_w), M, Cin, K,
                                        N.ptr(_head_ce_workspace(K, y.device)), out.data_ptr(), out.data_ptr() + 8,
                                        N.stream()), "lss_head_ce_fwd")
    return out[2].view(()), out[:2]


def head_ce_bwd(y, head_w, head_b, target, class_w, sums, grad_loss):
    """Backward of head_ce_fwd: returns (dy bf16 like y, d_head_w (K, Cin) fp32, d_head_b (K) fp32)."""
    K, Cin = head_w.shape
    M = y.numel() // Cin
    dy = torch.empty_like(y)
    dw = torch.empty(K, Cin, dtype=torch.float32, device=y.device)
    db = torch.empty(K, dtype=torch.float32, device=y.device)
    gl = grad_loss.reshape(1).float().contiguous()
    with _timed("head_ce_bwd"):
        N.check(N.lib().lss_head_ce_bwd(N.ptr(y), N.ptr(head_w), N.ptr(head_b), N.ptr(target), N.ptr(class_w), M, Cin, K,
                                        N.ptr(sums), N.ptr(gl), N.ptr(_head_ce_workspace(K, y.device)), N.ptr(dy),
                                        N.ptr(dw), N.ptr(db), N.stream()), "lss_head_ce_bwd")
    return dy, dw, db


def head1x1_fwd(y, head_w, head_b):
    """The 1x1 head alone: y (B, H, W, 128) contiguous bf16 -> logits (B, K, H, W) fp32 NCHW (lss_head1x1_fwd)."""
    K, Cin = head_w.shape
    if y.dtype != torch.bfloat16 or not y.is_contiguous() or y.dim() != 4 or y.shape[-1] != Cin:
        raise ValueError("y must be a contiguous (B, H, W, %d) bf16 tensor" % Cin)
    _f32c(head_w, "head_w", (K, Cin))
    _f32c(head_b, "head_b", (K,))
    B, H, W, _ = y.shape
    out = torch.empty(B, K, H, W, dtype=torch.float32, device=y.device)
    with _timed("head1x1_fwd"):
        N.check(N.lib().lss_head1x1_fwd(N.ptr(y), N.ptr(head_w), N.ptr(head_b), B * H * W, H * W, Cin, K, N.ptr(out),
                                        N.stream()), "lss_head1x1_fwd")
    return out


def head1x1_bwd(y, head_w, head_b, grad_logits):
    """Backward of head1x1_fwd: grad_logits (B, K, H, W) fp32 contiguous -> (dy bf16 like y, d_head_w (K, Cin), d_head_b (K))."""
    K, Cin = head_w.shape
    B, H, W, _ = y.shape
    _f32c(grad_logits, "grad_logits", (B, K, H, W))
    dy = torch.empty_like(y)
    dw = torch.empty(K, Cin, dtype=torch.float32, device=y.device)
    db = torch.empty(K, dtype=torch.float32, device=y.device)
    with _timed("head1x1_bwd"):
        N.check(N.lib().lss_head1x1_bwd(N.ptr(y), N.ptr(head_w), N.ptr(head_b), N.ptr(grad_logits), B * H * W, H * W, Cin,
                                        K, N.ptr(_head_ce_workspace(K, y.device)), N.ptr(dy), N.ptr(dw), N.ptr(db),
                                        N.stream()), "lss_head1x1_bwd")
    return dy, dw, db


def pack_conv_weight_s2d(w_oihw, pad):
    """OIHW fp32 of a stride-2 k x k conv -> bf16 [tap'][Cout][4*Cin] for conv2d_s2_nhwc."""
    Cout, Cin, K, K2 = w_oihw.shape
    _f32c(w_oihw, "conv weight")
    nbytes = N.lib().lss_conv2d_s2d_packed_weight_bytes(Cout, Cin, K, pad)
    taps = nbytes // (Cout * 4 * Cin * 2)
    out = torch.empty(taps, Cout, 4 * Cin, dtype=torch.bfloat16, device=w_oihw.device)
    N.check(N.lib().lss_conv2d_pack_weights_s2d(N.ptr(w_oihw), Cout, Cin, K, pad, N.ptr(out), N.stream()),
            "lss_conv2d_pack_weights_s2d")
    return out


def conv2d_s2_nhwc(x, w_s2d, K, pad, scale=None, shift=None, residual=None, relu=False, stats=None,
                   tag="conv2d_fwd"):
    """Stride-2 K x K conv (3/pad 1, 7/pad 3 with s2d-packed weights; 1/pad 0 with the plain
    pack), bf16 NHWC, on the LDS-tiled kernel."""
    B, H, W, Cx = x.shape
    taps, Cout, C4 = w_s2d.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or w_s2d.dtype != torch.bfloat16 \
            or C4 != (Cx if K == 1 else 4 * Cx):
        raise ValueError("conv2d_s2_nhwc operands must be contiguous bf16 with s2d-packed weights")
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=x.device)
    for name, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _f32c(t, name, (Cout,))
    if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != tuple(y.shape)
                                 or not residual.is_contiguous()):
        raise ValueError("residual must match the output")
    if stats is not None:
        _f32c(stats, "stats", (2 * Cout,))
    if _recorder is not None:
        _recorder.add(1, (x, w_s2d, scale, shift, residual, y, stats), x=x, w=w_s2d, scale=scale, shift=shift,
                      residual=residual, y=y, stats=stats, B=B, H=H, W=W, Cx=Cx, Cout=Cout, KH=K, KW=K, stride=2,
                      pad=pad, relu=1 if relu else 0, dt=DT_BF16)
    with _timed(tag):
        N.check(N.lib().lss_conv2d_s2_fwd(N.ptr(x), N.ptr(w_s2d), N.ptr(scale), N.ptr(shift), N.ptr(residual),
                                          N.ptr(y), N.ptr(stats), B, H, W, Cx, Cout, K, pad, 1 if relu else 0,
                                          N.stream()), "lss_conv2d_s2_fwd")
    return y


def conv2d_s2_dual_nhwc(x, w_s2d, scale, shift, split, relu=True, tag="conv2d_fwd"):
    """Two 3x3/2 (pad 1) convs over the same bf16 NHWC input in one launch: w_s2d (taps, Cout, 4*Cx) holds
    both weight sets stacked along Cout; channels [0, split) -> y (activation applied), the rest -> y2."""
    B, H, W, Cx = x.shape
    taps, Cout, C4 = w_s2d.shape
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or w_s2d.dtype != torch.bfloat16 or C4 != 4 * Cx:
        raise ValueError("conv2d_s2_dual_nhwc operands must be contiguous bf16 with s2d-packed weights")
    _f32c(scale, "scale", (Cout,))
    _f32c(shift, "shift", (Cout,))
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(B, Ho, Wo, split, dtype=torch.bfloat16, device=x.device)
    y2 = torch.empty(B, Ho, Wo, Cout - split, dtype=torch.bfloat16, device=x.device)
    if _recorder is not None:
        _recorder.add(3, (x, w_s2d, scale, shift, y, y2), x=x, w=w_s2d, scale=scale, shift=shift, y=y, y2=y2, B=B, H=H,
                      W=W, Cx=Cx, Cout=Cout, KH=3, KW=3, stride=2, pad=1, relu=1 if relu else 0, dt=DT_BF16,
                      split=split)
    with _timed(tag):
        N.check(N.lib().lss_conv2d_s2_dual_fwd(N.ptr(x), N.ptr(w_s2d), N.ptr(scale), N.ptr(shift), N.ptr(y), N.ptr(y2),
                                               B, H, W, Cx, Cout, split, 3, 1, 1 if relu else 0, N.stream()),
                "lss_conv2d_s2_dual_fwd")
    return y, y2


def conv3x3_head_nchw(x, w_packed, scale, shift, head_w, head_b, x2=None, up=1, relu=True, tag="conv2d_fwd"):
    """3x3/s1/p1 conv (+fused upsample/concat) + scale/shift + ReLU + 1x1 head in one launch.
    x (B,H,W,Cx) bf16 NHWC; head_w (n,Cout) fp32, Cout = 128 (64: plain 3x3 only); returns (B, n, H*up, W*up)
    fp32 NCHW."""
    B, H, W, Cx = x.shape
    ring = isinstance(w_packed, RingWeight)
    if ring:
        taps, Cout, Cin, w_packed = 9, w_packed.Cout, w_packed.Cin, w_packed.data
    else:
        taps, Cout, Cin = w_packed.shape
    C2 = x2.shape[3] if x2 is not None else 0
    if x.dtype != torch.bfloat16 or not x.is_contiguous() or taps != 9 or Cin != Cx + C2 or Cout not in (64, 128) \
            or (Cout == 64 and (up != 1 or C2 != 0)):
        raise ValueError("conv3x3_head_nchw: bf16 NHWC input, 3x3 weights with Cout == 128 (or 64 without "
                         "upsample / concat) required")
    n = head_w.shape[0]
    _f32c(head_w, "head_w", (n, Cout))
    _f32c(head_b, "head_b", (n,))
    _f32c(scale, "scale", (Cout,))
    _f32c(shift, "shift", (Cout,))
    out = torch.empty(B, n, H * up, W * up, dtype=torch.float32, device=x.device)
    if _recorder is not None:
        _recorder.add(2, (x, x2, w_packed, scale, shift, head_w, head_b, out), x=x, x2=x2, w=w_packed, scale=scale,
                      shift=shift, head_w=head_w, head_b=head_b, head_out=out, B=B, H=H, W=W, Cx=Cx, C2=C2, up=up,
                      Cout=Cout, KH=3, KW=3, stride=1, pad=1, relu=(1 if relu else 0) | (W_RING if ring else 0),
                      dt=DT_BF16, head_n=n)
    with _timed(tag):
        N.check(N.lib().lss_conv2d_head_fwd(N.ptr(x), N.ptr(x2), N.ptr(w_packed), N.ptr(scale), N.ptr(shift),
                                            N.ptr(head_w), N.ptr(head_b), N.ptr(out), B, H, W, Cx, C2, up, Cout, n,
                                            (1 if relu else 0) | (W_RING if ring else 0), N.stream()),
                "lss_conv2d_head_fwd")
    return out


def nchw_to_nhwc(x, dt):
    """(B,C,H,W) fp32 contiguous -> (B,H,W,C) in dt."""
    _f32c(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, H, W, C, dtype=_TORCH_DT[dt], device=x.device)
    N.check(N.lib().lss_nchw_f32_to_nhwc(N.ptr(x), N.ptr(y), B, C, H, W, dt, N.stream()), "lss_nchw_f32_to_nhwc")
    return y


def nhwc_to_nchw(x, dt):
    """(B,H,W,C) in dt -> (B,C,H,W) fp32 contiguous."""
    if x.dtype != _TORCH_DT[dt] or not x.is_contiguous():
        raise ValueError("x must be contiguous %s" % _TORCH_DT[dt])
    B, H, W, C = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    N.check(N.lib().lss_nhwc_to_nchw_f32(N.ptr(x), N.ptr(y), B, C, H, W, dt, N.stream()), "lss_nhwc_to_nchw_f32")
    return y
