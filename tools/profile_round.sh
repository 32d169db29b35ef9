#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/prof_<tag>/ and writes the summaries to be judged into
# gpurun_out/prof_<tag>/profiles/ (only gpurun_out/ travels back from the box: copy them into profiles/ afterwards).
#     bash tools/profile_round.sh r03
# Counter passes are separate runs (one --pmc group each, kernel trace only), as MI355X_MICROARCH.md prescribes.
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
P=$out/profiles
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-fine --no-dp-rank --also \"\""
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
cp $out/bench_default.json $P/${tag}_bench_default.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/bench_under_rocprof.log 2>&1
cp $out/kt/kt_kernel_stats.csv $P/${tag}_bench_coarse_kernel_stats.csv
grep '^{' $out/bench_under_rocprof.log | tail -1 > $P/${tag}_bench_coarse_under_rocprof.json
python3 tools/timeline.py "$out/kt/kt_kernel_trace.csv" adam_frozen -v > $P/${tag}_step_timeline.txt
echo "kernel trace + timeline done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv $P/${tag}_pmc_traffic.json > $out/pmc_traffic.txt
echo "traffic done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/pmc_mfma -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_mfma.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma/m_counter_collection.csv $out/pmc_mfma/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy.json > $out/pmc_mfma.txt
echo "mfma done"
# the same counters over the fine phase (fine/first's filter gradient: fewch.hip) and over DCNF's step (its first conv's)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/pmc_mfma_fine -o m -- python3 bench.py --phase fine --steps 5 --warmup 2 --no-cpu-baseline --no-dp-rank --also "" > $out/pmc_mfma_fine.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma_fine/m_counter_collection.csv $out/pmc_mfma_fine/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy_fine.json > $out/pmc_mfma_fine.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/pmc_mfma_dcnf -o m -- python3 bench.py --model dcnf --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_mfma_dcnf.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma_dcnf/m_counter_collection.csv $out/pmc_mfma_dcnf/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy_dcnf.json > $out/pmc_mfma_dcnf.txt
rm -rf $out/pmc_mfma_fine $out/pmc_mfma_dcnf
echo "mfma fine + dcnf done"
python3 bench.py --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --precision bf16s --also bf16,bf16x3,fp32 > $P/${tag}_bench_batch64_bf16_storage.json 2> $out/bench_b64.err
rocprofv3 --kernel-trace --output-format csv -d $out/kt16 -o kt -- python3 bench.py --batch 64 --precision bf16s --steps 6 --warmup 3 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/kt16.log 2>&1
python3 tools/timeline.py "$out/kt16/kt_kernel_trace.csv" adam_frozen -v > $P/${tag}_step_timeline_bf16_storage.txt
rm -rf $out/kt16
echo "b64 done"
python3 bench.py --model dcnf --no-cpu-baseline > $P/${tag}_bench_dcnf.json 2> $out/bench_dcnf.err
echo "dcnf done"
python3 tools/bench_input.py 256 32 100 2> $out/input.err | tail -1 > $P/${tag}_input_pipeline.json
echo "input done"
python3 tools/bench_layers.py > $P/${tag}_bench_layers.txt 2> $out/layers.err
echo "layers done"
# round 5: the few-channel filter gradient (fewch.hip) against the generic window-run GEMM, fine/third's kernels
A3D_TUNING=1 python3 tools/bench_fewch.py 2> /dev/null | grep -v amdgpu > $out/fewch_on.txt
A3D_TUNING=1 A3D_FEWCH=0 python3 tools/bench_fewch.py 2> /dev/null | grep -v amdgpu > $out/fewch_off.txt
{ echo "# LDS-staged kernel (fewch.hip):"; cat $out/fewch_on.txt; echo "# generic window-run implicit GEMM (A3D_TUNING=1 A3D_FEWCH=0):"; cat $out/fewch_off.txt; } > $P/${tag}_bench_fewch.txt
{ python3 tools/bench_fine3.py 32; python3 tools/bench_fine3.py 64; } 2> /dev/null | grep -v amdgpu > $P/${tag}_bench_fine_third.txt
echo "fewch + fine/third done"
# bf16 storage (config 5) per layer: the LDS-DMA kernel, and igemm_bf16 alone beside it
python3 tools/bench_layers_bf16.py 2> $out/layers_bf16.err | grep -v amdgpu > $out/l16_ring.txt
A3D_TUNING=1 A3D_RING=0 python3 tools/bench_layers_bf16.py 2> /dev/null | grep -v amdgpu > $out/l16_old.txt
paste -d'|' $out/l16_ring.txt $out/l16_old.txt | awk -F'|' '{printf "%-52s | igemm_bf16 only (A3D_TUNING=1 A3D_RING=0): %s\n", $1, substr($2, 18)}' > $P/${tag}_bench_layers_bf16.txt
echo "bf16 layers done"
rm -rf $out/kt $out/pmc_fetch $out/pmc_write $out/pmc_mfma
bash tools/profile_fine.sh $tag
ls $P
