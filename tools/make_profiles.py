#!/usr/bin/env python3
"""gpurun_out/prof_<round>/ (collected by tools/refresh_profiles.sh on the GPU box) -> profiles/<round>_*.
    python tools/make_profiles.py r02"""
import glob
import io
import json
import os
import shutil
import subprocess
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)


def find(d, pat):
    # newest match: gpurun MERGES a call's files into gpurun_out/, so an earlier collection's files (other process
    # ids in their names) may still lie next to the current ones
    g = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None


class BadEvidence(RuntimeError):
    pass


def refuse_traceback(text, what):
    """A tool that died leaves a Python traceback where its numbers should be: such text never becomes a file under
    profiles/ (round 3 committed five of them after an ABI change left diag_libs/ stale)."""
    if "Traceback (most recent call last)" in text or "AttributeError:" in text or "undefined symbol" in text:
        raise BadEvidence("%s: the tool's output is an error, not a measurement:\n%s" % (what, text[-600:]))
    return text


def run_tool(script, *args):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + list(args), capture_output=True, text=True)
    if r.returncode != 0:
        raise BadEvidence("%s %s exited with %d:\n%s" % (script, " ".join(args), r.returncode, r.stderr[-600:]))
    return refuse_traceback(r.stdout, script)


def last_json_line(path):
    for ln in reversed(open(path).read().strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise ValueError("no JSON line in " + path)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    dst = os.path.join(ROOT, "profiles")
    w = lambda name, text: open(os.path.join(dst, "%s_%s" % (rnd, name)), "w").write(text)
    bad = []
    for tag in ("bench_line", "bench_hires", "bench_fp32", "bench_config1"):
        p = os.path.join(src, tag + ".json")
        if os.path.exists(p) and os.path.getsize(p):
            w(tag + ".json", json.dumps(last_json_line(p)) + "\n")
    p = os.path.join(src, "bench_rehearse2.txt")
    if os.path.exists(p) and os.path.getsize(p):
        w("bench_rehearse2.json", json.dumps(last_json_line(p)) + "\n")
    for tag, out in (("kt", "bench_kernel_summary.txt"), ("kt_hires", "hires_kernel_summary.txt"),
                     ("kt_fp32", "fp32_kernel_summary.txt"), ("kt_train", "train_kernel_summary.txt")):
        kt = find(os.path.join(src, tag), "*kernel_trace.csv")
        if kt:
            w(out, run_tool("prof_summary.py", kt))
    st = find(os.path.join(src, "kt"), "*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(dst, rnd + "_bench_kernel_stats.csv"))
    f, wr = find(os.path.join(src, "fetch"), "*counter_collection.csv"), find(os.path.join(src, "write"), "*counter_collection.csv")
    if f and wr:
        js = os.path.join(dst, rnd + "_hbm_traffic.json")
        w("hbm_traffic.txt", run_tool("pmc_traffic.py", f, wr, js))
    sq, sqkt = find(os.path.join(src, "sq"), "*counter_collection.csv"), find(os.path.join(src, "sq"), "*kernel_trace.csv")
    if sq and sqkt:
        head = ("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
                "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -- python bench.py --steps 10 --warmup 3 "
                "--no-cpu-baseline --no-train\n# mean per dispatch (tools/pmc_sq_summary.py); MFMA_util = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)\n")
        w("conv_pmc_sq.txt", head + run_tool("pmc_sq_summary.py", sq, sqkt))
    sq, sqkt = find(os.path.join(src, "sq_ring"), "*counter_collection.csv"), find(os.path.join(src, "sq_ring"), "*kernel_trace.csv")
    if sq and sqkt:
        head = ("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python tools/bench_ring.py --rounds 2 --iters 3\n"
                "# the three big convs on the tile kernel (conv_lds_kernel) and on the ring kernel (conv_ring_kernel), mean per "
                "dispatch; MFMA_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)\n")
        w("ring_pmc_sq.txt", head + "\n".join(ln for ln in run_tool("pmc_sq_summary.py", sq, sqkt).splitlines()
                                               if "conv_" in ln or ln.startswith("kernel")) + "\n")
    for tag, out in (("bk_conv.txt", "conv_microbench.txt"), ("bk_stamps.txt", "conv_phase_stamps.txt"),
                     ("bk_l1.txt", "l1_microbench.txt"), ("l1_stamps.txt", "l1_stamps.txt"),
                     ("ring_microbench.txt", "ring_microbench.txt"), ("ring_wait_stats.txt", "ring_wait_stats.txt"),
                     ("ring_microbench_noblend.txt", "ring_microbench_noblend.txt"),
                     ("ring_microbench_nowdma.txt", "ring_microbench_nowdma.txt"),
                     ("ring_microbench_readsonly.txt", "ring_microbench_readsonly.txt"),
                     ("ring_microbench_oneread.txt", "ring_microbench_oneread.txt"),
                     ("wgrad_microbench.txt", "wgrad_microbench.txt"), ("ks_microbench.txt", "ks_microbench.txt"),
                     ("ks_stamps.txt", "ks_stamps.txt"), ("ks_stamps_diag.txt", "ks_stamps_diag.txt"),
                     ("graph_node_cost.txt", "graph_node_cost.txt"), ("train_step_trace.txt", "train_step_trace.txt")):
        p = os.path.join(src, tag)
        if os.path.exists(p):
            try:
                w(out, refuse_traceback("".join(ln for ln in open(p) if "amdgpu.ids" not in ln), tag))
            except BadEvidence as e:
                bad.append(str(e))
    print("\n".join(sorted(os.listdir(dst))))
    if bad:
        sys.stderr.write("\n".join(["REFUSED (not written to profiles/):"] + bad) + "\n")
        raise SystemExit(2)


if __name__ == "__main__":
    main()
