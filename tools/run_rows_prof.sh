#!/bin/bash
# Phase clocks of the row-streaming conv kernel (conv_rows.hip): builds a private library with -DTAG_ROWS_PROF HERE (no GPU
# needed), then on the GPU box:   TAG_HIP_LIB=texttoaudiogrounding_amd/libtag_rowsprof.so python tools/conv_rows_prof.py
set -e
cd "$(dirname "$0")/../texttoaudiogrounding_amd/csrc"
L="tag_lib.o logmel.o bn_pool.o conv.o conv_x3.o gemm.o gru.o heads.o text_tower.o cross.o mha.o"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off"
/opt/rocm/bin/hipcc $F -DTAG_ROWS_PROF $TAG_ROWS_EXTRA -c conv_rows.hip -o /tmp/conv_rows_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_rowsprof.so $L /tmp/conv_rows_prof.o
echo built ../libtag_rowsprof.so
