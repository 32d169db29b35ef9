#!/bin/bash
# GPU box: parity tests of the LDS-DMA bf16 kernel, then its per-layer timings against igemm_bf16 and over its tile configs.
out=gpurun_out/${1:-ring}
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "lds_dma_kernel or bf16" > $out/tests.log 2>&1
rc=$?
echo "tests rc $rc" >> $out/tests.log
tail -15 $out/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_layers_bf16.py > $out/layers_ring.txt 2>&1 || exit 1
A3D_RING=0 timeout -k 10 300 python tools/bench_layers_bf16.py > $out/layers_old.txt 2>&1 || exit 1
for cfg in 0 2 3; do
  A3D_TUNING=1 A3D_RING_CFG=$cfg timeout -k 10 300 python tools/bench_layers_bf16.py conv2d_1 conv2d_2 conv2d_3 > $out/layers_cfg$cfg.txt 2>&1 || exit 1
done
paste $out/layers_ring.txt $out/layers_old.txt | cut -c1-150
