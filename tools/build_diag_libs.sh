#!/bin/bash
# Diagnostic builds of csrc/conv_ring.hip linked against the other objects of the shipped library:
#   diag_libs/liblss_STATS.so   (-DRK_STATS: wait statistics, tools/ring_stats.py)
#   diag_libs/liblss_NOBLEND.so (-DRK_DIAG_NOBLEND: timing-only, the upsampled patch stays zero)
# Use through LSS_HIP_LIB=<path>.  Run after `python -m lss2_multimodal_nu_amd.build_native`.
set -e
# always from the CURRENT objects of the shipped library: rebuild that first (no-op when nothing changed)
python -m lss2_multimodal_nu_amd.build_native > /dev/null
cd "$(dirname "$0")/../lss2_multimodal_nu_amd/csrc"
mkdir -p ../../diag_libs /tmp/lss_diag
for v in STATS NOBLEND NOWDMA "$@"; do
  case "$v" in KS_*)   # timing-only builds of the K-split kernel: KS_NOREAD (fragments read once), KS_NOMFMA (reads only)
    /opt/rocm/bin/hipcc -c conv_ks.hip -o /tmp/lss_diag/conv_ks_$v.o -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I../../include $([ "$v" = KS_WPRE_ALL ] && echo -DKS_WPRE_ALL || echo -DKS_DIAG_${v:3})
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../diag_libs/liblss_$v.so $(ls build/*.o | grep -v conv_ks.o) /tmp/lss_diag/conv_ks_$v.o -ldl
    continue;;
  esac
  def=-DRK_DIAG_$v; [ "$v" = STATS ] && def=-DRK_STATS
  case "$v" in PSLEEP*) def="-DRK_PSLEEP=${v:6}";; esac
  case "$v" in PRIO*) def="-DRK_PRIO_C=${v:4:1} -DRK_PRIO_P=${v:5:1} -DRK_PRIO_W=${v:6:1}";; esac
  /opt/rocm/bin/hipcc -c conv_ring.hip -o /tmp/lss_diag/conv_ring_$v.o -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $def
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../diag_libs/liblss_$v.so $(ls build/*.o | grep -v conv_ring.o) /tmp/lss_diag/conv_ring_$v.o -ldl
done
ls -la ../../diag_libs
python ../../tools/check_diag_abi.py
