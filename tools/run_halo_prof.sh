#!/bin/bash
# Builds a private library with -DTAG_HALO_PROF, runs tools/conv_halo_prof.py (or, with the argument "timeline",
# tools/halo_wg_timeline.py) on it and restores the product library.
set -e
cd texttoaudiogrounding_amd/csrc
L="tag_lib.o logmel.o bn_pool.o conv_x3.o conv_rows.o conv_wgrad_dma.o gemm.o gru.o heads.o text_tower.o cross.o mha.o"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off"
[ -f ../libtag_hprof.so ] || { /opt/rocm/bin/hipcc $F -DTAG_HALO_PROF -c conv.hip -o /tmp/conv_hprof.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_hprof.so $L /tmp/conv_hprof.o; }
cd ..
cp libtag_hip.so /tmp/lib_orig.so
cp libtag_hprof.so libtag_hip.so
mkdir -p ../gpurun_out
if [ "${1:-}" = timeline ]; then python ../tools/halo_wg_timeline.py > ../gpurun_out/halo_timeline.log 2>&1 || true
else python ../tools/conv_halo_prof.py > ../gpurun_out/haloprof.log 2>&1 || true; fi
cp /tmp/lib_orig.so libtag_hip.so
