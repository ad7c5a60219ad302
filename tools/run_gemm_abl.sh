#!/bin/bash
# GEMM ablation builds (TAG_GEMM_ABL in gemm.hip): rebuild gemm.o with the macro, time the step's GEMMs, restore.
# Results are WRONG by construction (staging skipped) -- timing only.   bash tools/run_gemm_abl.sh "1 2"
cd "$(dirname "$0")/../texttoaudiogrounding_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-unused-function"
cp ../libtag_hip.so /tmp/libtag_hip.keep.so; cp gemm.o /tmp/gemm.keep.o
for abl in ${1:-1 2}; do
    /opt/rocm/bin/hipcc $FLAGS -DTAG_GEMM_ABL=$abl -c gemm.hip -o gemm.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_hip.so tag_lib.o logmel.o bn_pool.o conv.o conv_x3.o gemm.o gru.o heads.o text_tower.o cross.o mha.o
    echo "== TAG_GEMM_ABL=$abl"
    (cd ../.. && timeout 300 python tools/gemm_bench.py | tail -10)
done
cp /tmp/libtag_hip.keep.so ../libtag_hip.so; cp /tmp/gemm.keep.o gemm.o
