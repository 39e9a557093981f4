"""bench.py as the driver invokes it: a BARE `python bench.py --gpus N` must start its own N ranks (children of a
process that has not touched the GPU), rendezvous on 127.0.0.1 and print ONE JSON line from rank 0.  `--dry-launch`
runs exactly that plumbing over gloo on the CPU, so it is testable here; the GPU legs are the same code path with
backend nccl (= RCCL) and one device per LOCAL_RANK."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


@pytest.mark.timeout(300)
def test_bare_multi_gpu_invocation_spawns_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rank_sum"] == 3.0 and out["backend"] == "gloo"


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_fired_watchdog_exits_non_zero_and_names_the_leg():
    """a hang in a process that has touched the GPU must not be reported as success.  The watchdog prints
    the partial line with the hung leg (rank 0) and leaves with rc != 0."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench._arm_watchdog(0.2, {'metric': 'm', 'value': 1.0}, 0, 'train'); time.sleep(30)" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-300:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["watchdog"]["hung_leg"] == "train" and line["value"] == 1.0
    assert "watchdog" in p.stderr
    # a non-zero rank prints no line but leaves with the same code
    code1 = code.replace("0, 'train'", "1, 'train'")
    p1 = subprocess.run([sys.executable, "-c", code1], capture_output=True, text=True, timeout=120)
    assert p1.returncode == 3 and p1.stdout.strip() == ""
