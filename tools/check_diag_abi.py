#!/usr/bin/env python3
"""Do the diagnostic builds under diag_libs/ export the ABI of the shipped library?  (tools/build_diag_libs.sh links
them from the objects of ONE build; when include/lss_hip.h grows afterwards they go stale and every tool run through
LSS_HIP_LIB=diag_libs/... dies in `_native.lib()` - round 3 committed those tracebacks as profiles.)
    python tools/check_diag_abi.py            exit 0: every diag lib resolves every symbol of _native.SIGNATURES
Needs no GPU (dlopen + dlsym only)."""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def missing_symbols(path, names):
    lib = ctypes.CDLL(path)
    return [n for n in names if not hasattr(lib, n)]


def main():
    from lss2_multimodal_nu_amd import _native as N
    libs = sorted(glob.glob(os.path.join(ROOT, "diag_libs", "*.so")))
    if not libs:
        print("no diag_libs/*.so (run tools/build_diag_libs.sh)")
        return 1
    main_lib = N.LIB_PATH
    rc = 0
    for p in libs:
        miss = missing_symbols(p, list(N.SIGNATURES))
        stale = os.path.getmtime(p) < os.path.getmtime(main_lib)
        print("%-40s %s%s" % (os.path.relpath(p, ROOT), "ok" if not miss else "MISSING " + ", ".join(miss[:4]),
                              "  (older than liblss_hip.so: rebuild)" if stale else ""))
        if miss or stale:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
