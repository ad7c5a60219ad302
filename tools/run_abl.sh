#!/bin/bash
# Ablation sweep of the one-product bf16 conv kernel (GPU box; run from the repo root through gpurun):
#   bash tools/run_abl.sh            -> gpurun_out/abl{0..7}.log  (0 = the product build)
# Builds private libraries with -DTAG_X3_ABL=n (see conv_x3.hip) next to the product one and times the 14 layer shapes.
set -e
cd texttoaudiogrounding_amd/csrc
L="tag_lib.o logmel.o bn_pool.o conv.o gemm.o gru.o heads.o text_tower.o cross.o mha.o"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off"
for a in 1 2 3 4 5 6 7; do
  [ -f ../libtag_abl$a.so ] || { /opt/rocm/bin/hipcc $F -DTAG_X3_ABL=$a -c conv_x3.hip -o /tmp/conv_x3_abl$a.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_abl$a.so $L /tmp/conv_x3_abl$a.o; }
done
cd ..
cp libtag_hip.so /tmp/lib_orig.so
mkdir -p ../gpurun_out
python ../tools/conv_bf16_bench.py > ../gpurun_out/abl0.log 2>&1
for a in 1 2 3 4 5 6 7; do cp libtag_abl$a.so libtag_hip.so; python ../tools/conv_bf16_bench.py > ../gpurun_out/abl$a.log 2>&1; done
cp /tmp/lib_orig.so libtag_hip.so
