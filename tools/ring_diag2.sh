#!/bin/bash
# diagnostic build: which synchronisation a k-tile of the LDS-DMA kernel waits on (timings only; results are wrong)
# A3D_DBG bits: 8 no barrier, 16 no wait for the requests, 32 no fragment reads, 3 requests fetch nothing
out=gpurun_out/${1:-ringdiag2}
mkdir -p $out
export A3D_LIB=$GRAFT_REPO_ROOT/tools/ab/liba3d_diag.so A3D_TUNING=1
for cfg in ${CFGS:-2}; do
  for dbg in ${DBGS:-0 8 16 24 32 40 56 59}; do
    echo "cfg $cfg dbg $dbg"
    A3D_RING_CFG=$cfg A3D_DBG=$dbg timeout -k 10 200 python tools/bench_layers_bf16.py ${LAYERS:-conv2d_1} 2>&1 | grep -v "amdgpu\|bwd_f\|total"
  done
done > $out/diag.txt 2>&1
cat $out/diag.txt
