#!/bin/bash
# timing-only diagnostic build of the LDS-DMA bf16 kernel: what a k-tile costs without its requests / without its math
out=gpurun_out/${1:-ringdiag}
mkdir -p $out
export A3D_LIB=$GRAFT_REPO_ROOT/tools/ab/liba3d_diag.so A3D_TUNING=1
for cfg in ${CFGS:-2 3}; do
  for dbg in 0 1 2 3 4 7; do
    echo "cfg $cfg dbg $dbg"
    A3D_RING_CFG=$cfg A3D_DBG=$dbg timeout -k 10 200 python tools/bench_layers_bf16.py ${LAYERS:-conv2d_1 conv2d_3} 2>&1 | grep -v "amdgpu\|bwd_f\|total"
  done
done > $out/diag.txt 2>&1
cat $out/diag.txt
