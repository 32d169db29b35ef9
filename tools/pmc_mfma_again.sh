set -e -o pipefail
tag=r05; out=gpurun_out/prof_$tag; P=$out/profiles; mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/pmc_mfma -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also "" > $out/pmc_mfma.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma/m_counter_collection.csv $out/pmc_mfma/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy.json > $out/pmc_mfma.txt
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/pmc_mfma_fine -o m -- python3 bench.py --phase fine --steps 5 --warmup 2 --no-cpu-baseline --no-dp-rank --also "" > $out/pmc_mfma_fine.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma_fine/m_counter_collection.csv $out/pmc_mfma_fine/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy_fine.json > $out/pmc_mfma_fine.txt
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/pmc_mfma_dcnf -o m -- python3 bench.py --model dcnf --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_mfma_dcnf.log 2>&1
python3 tools/pmc_mfma.py $out/pmc_mfma_dcnf/m_counter_collection.csv $out/pmc_mfma_dcnf/m_kernel_trace.csv $P/${tag}_pmc_mfma_busy_dcnf.json > $out/pmc_mfma_dcnf.txt
rm -rf $out/pmc_mfma $out/pmc_mfma_fine $out/pmc_mfma_dcnf
grep -h fewch $out/pmc_mfma.txt $out/pmc_mfma_fine.txt $out/pmc_mfma_dcnf.txt || true
