#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/).
#   bash tools/profile_round.sh r02
# Counter passes are separate runs (one --pmc group each, kernel trace only), as MI355X_MICROARCH.md prescribes.
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 20 --warmup 5 --no-cpu-baseline --no-fine --also \"\""
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/bench_under_rocprof.log 2>&1
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out/pmc_mfma -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_mfma.log 2>&1
echo "mfma done"
python3 bench.py --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --precision bf16s --also bf16,bf16x3,fp32 > $out/bench_b64.json 2> $out/bench_b64.err
echo "b64 done"
python3 bench.py --model dcnf --no-cpu-baseline > $out/bench_dcnf.json 2> $out/bench_dcnf.err
echo "dcnf done"
ls $out
