#!/bin/bash
# Ablation builds of the fused Winograd kernel (CPU: cross-compiles; the .so files travel to the GPU box):
#   bash tools/wino_fused_abl.sh build "1 2 12 3 4"   -> texttoaudiogrounding_amd/libtag_wf<n>.so  (-DTAG_WF_ABL=n, results wrong by construction)
#   bash tools/wino_fused_abl.sh run "1 2 12 3 4" [bench args]   (GPU box) -> one line per variant
set -e
cd "$(dirname "$0")/../texttoaudiogrounding_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -fno-slp-vectorize"   # = csrc/Makefile for this file
if [ "$1" = build ]; then
  OBJS=$(ls *.o | grep -v conv_wino_fused.o)
  for a in $2; do
    /opt/rocm/bin/hipcc $F -DTAG_WF_ABL=$a $WF_EXTRA -c conv_wino_fused.hip -o /tmp/wf_abl$a.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtag_wf$a.so $OBJS /tmp/wf_abl$a.o
  done
else
  cd ../..
  python tools/wino_fused_bench.py ${@:3}
  for a in $2; do TAG_HIP_LIB=$PWD/texttoaudiogrounding_amd/libtag_wf$a.so TAG_ALLOW_STALE_LIB=1 python tools/wino_fused_bench.py ${@:3}; done
fi
