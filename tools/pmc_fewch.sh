#!/bin/bash
# SQ counters of fewch_bwdf_kernel alone (fp32, B = 32):  bash tools/pmc_fewch.sh [tag]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-pmc_few32}
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_$n -o p -- python3 tools/bench_fewch.py B=32 > gpurun_out/${tag}_$n.log 2>&1
  python3 tools/pmc_table.py gpurun_out/${tag}_$n/p_counter_collection.csv gpurun_out/${tag}_$n/p_kernel_trace.csv fewch_bwdf
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA &&
run c SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_c
