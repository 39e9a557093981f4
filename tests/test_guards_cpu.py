"""The hang guards are load-bearing (VERDICT r3 item 6 / ADVICE r3): the bounded flag waits of the loader / consumer
kernels (conv_ring.hip, conv_wgrad.hip, conv_chain.hip) leave garbage behind when they hit their bound, so the product -
not just the tests - reads the timeout counters: bench.py after every leg (non-zero -> exit code GUARD_RC), BevEncode
when it records a launch plan, dp.GraphedTrainStep in its self-check.  A fake library that reports one timed-out wait
stands in for the GPU here."""
import pytest
import torch

import bench
from lss2_multimodal_nu_amd import _native as N
from lss2_multimodal_nu_amd import ops


class _Lib:
    def __init__(self, ring=0, wgrad=0, chain=0):
        self.v = {"ring": ring, "wgrad": wgrad, "chain": chain}

    def __getattr__(self, name):
        for k, val in self.v.items():
            if name == "lss_conv2d_%s_timeouts" % k:
                return lambda: val
        raise AttributeError(name)


def test_counters_are_read_by_name_from_the_abi():
    got = ops.timeout_counters(_Lib(ring=0, wgrad=3))
    assert got["wgrad_timeouts"] == 3 and got["ring_timeouts"] == 0
    assert set("lss_conv2d_" + k for k in got) == set(ops.GUARD_COUNTERS)
    for name in ops.GUARD_COUNTERS:  # every guard counter is an export of include/lss_hip.h
        assert name in N.SIGNATURES


def test_a_timed_out_wait_raises():
    ops.assert_no_timeouts("clean", _Lib())
    with pytest.raises(N.LssNativeError, match="ring_timeouts=1"):
        ops.assert_no_timeouts("BevEncode (launch plan recorded)", _Lib(ring=1))
    with pytest.raises(N.LssNativeError, match="wgrad_timeouts=2"):
        ops.assert_no_timeouts("GraphedTrainStep self-check", _Lib(wgrad=2))
    with pytest.raises(N.LssNativeError):  # a counter that cannot be read (-1) is a failure too
        ops.assert_no_timeouts("x", _Lib(ring=-1))


def test_bench_exit_code_follows_the_counters():
    clean = {"inference": ops.timeout_counters(_Lib()), "train": ops.timeout_counters(_Lib())}
    assert bench.guards_rc(clean) == 0
    dirty = dict(clean, end=ops.timeout_counters(_Lib(wgrad=1)))
    assert bench.guards_rc(dirty) == bench.GUARD_RC != 0
    assert bench.GUARD_RC != bench.WATCHDOG_RC


def test_bench_reads_the_counters_through_ops(monkeypatch):
    seen = []
    monkeypatch.setattr(ops, "timeout_counters", lambda lib=None: seen.append(lib) or {"ring_timeouts": 1})
    assert bench.read_guards("fake") == {"ring_timeouts": 1} and seen == ["fake"]


def test_guard_every_knob(monkeypatch):
    monkeypatch.delenv("LSS_GUARD_EVERY", raising=False)
    assert ops.guard_every() == 0
    monkeypatch.setenv("LSS_GUARD_EVERY", "64")
    assert ops.guard_every() == 64
    monkeypatch.setenv("LSS_GUARD_EVERY", "junk")
    assert ops.guard_every() == 0


def test_graphed_step_refuses_a_capture_without_warm_up():
    from lss2_multimodal_nu_amd import dp
    with pytest.raises(ValueError, match="warm-up"):
        dp.GraphedTrainStep(None, None, None, None, torch.zeros(1), None, warmup=0)


def test_profile_collector_refuses_tracebacks():
    """tools/make_profiles.py never writes a tool's error text under profiles/ (round 3 committed five tracebacks)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_profiles", os.path.join(root, "tools", "make_profiles.py"))
    mp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp)
    assert mp.refuse_traceback("up2 98.0 us 945 TF\n", "x").startswith("up2")
    with pytest.raises(mp.BadEvidence):
        mp.refuse_traceback("Traceback (most recent call last):\n  File ...\nAttributeError: liblss_STATS.so: undefined "
                            "symbol: lss_conv2d_wgrad4x4_workspace_bytes\n", "ring_wait_stats.txt")
    assert not [f for f in os.listdir(os.path.join(root, "profiles"))
                if "Traceback" in open(os.path.join(root, "profiles", f), errors="ignore").read()]


def test_prepack_freshness_is_per_thread_and_stale_by_default():
    """ops.WeightPrepack: a unit is only handed a pre-packed image between the `run()` of its own thread's step and the
    `invalidate()` after the backward pass - never by default, never because another thread's step is in flight."""
    import threading

    from lss2_multimodal_nu_amd import ops
    pp = ops.WeightPrepack()
    assert pp.fresh is False and pp.lookup(__import__("torch").zeros(4), ("tile",)) is None
    pp.fresh = True
    seen = []
    t = threading.Thread(target=lambda: seen.append(pp.fresh))
    t.start()
    t.join()
    assert seen == [False] and pp.fresh is True
    pp.invalidate()
    assert pp.fresh is False
    pp.run([])   # nothing registered: a no-op that must not need the library
    assert pp.fresh is False
