"""Build liblss_hip.so (gfx950) in-tree with hipcc.

    python -m lss2_multimodal_nu_amd.build_native [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "liblss_hip.so")
ARCH = "gfx950"

# per-source extra flags.  geom_bucket.hip carries the exact-index arithmetic:
# no FMA contraction allowed there.
SOURCES = {
    "geom_bucket.hip": ["-ffp-contract=off"],
    "depthnet.hip": [],
    "splat.hip": [],
    "conv_mfma.hip": [],
    "conv_ring.hip": ["-fno-slp-vectorize"],  # no v_pk_*_f32 next to MFMAs (MI355X_MICROARCH.md)
    "conv_ks.hip": [],
    "layout.hip": [],
    "bev_transformer.hip": [],
    "linear_mfma.hip": [],
    "ffn_fused.hip": [],
    "conv_grad.hip": [],
    "conv_wgrad.hip": [],
    "bn_train.hip": [],
    "loss.hip": [],
    "optim.hip": [],
    "rccl_bucket.hip": [],  # host code only: RCCL resolved with dlsym at run time (no -lrccl)
}
COMMON = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-variable",
          "-Wno-unused-but-set-variable"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, save_temps=False):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "lss_hip.h"))
    objs, procs = [], []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        stale = force or _newer(sp, op) or any(_newer(h, op) for h in headers)
        if stale:
            cmd = [hipcc, "-c", sp, "-o", op] + COMMON + extra
            if save_temps:
                cmd += ["-save-temps=obj"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, cwd=OBJ)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if procs or force or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs if os.path.exists(o)):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
