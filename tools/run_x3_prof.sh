#!/bin/bash
# Builds a private library with -DTAG_X3_PROF, runs tools/conv_x3_prof.py on it and restores the product library.
set -e
cd texttoaudiogrounding_amd/csrc
L="tag_lib.o logmel.o bn_pool.o conv.o gemm.o gru.o heads.o text_tower.o cross.o mha.o"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off"
[ -f ../libtag_prof.so ] || { /opt/rocm/bin/hipcc $F -DTAG_X3_PROF -c conv_x3.hip -o /tmp/conv_x3_prof.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_prof.so $L /tmp/conv_x3_prof.o; }
cd ..
cp libtag_hip.so /tmp/lib_orig.so
cp libtag_prof.so libtag_hip.so
mkdir -p ../gpurun_out
python ../tools/conv_x3_prof.py > ../gpurun_out/x3prof.log 2>&1 || true
cp /tmp/lib_orig.so libtag_hip.so
