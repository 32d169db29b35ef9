#!/bin/bash
# SQ counters of fine/third's kernels alone (B = 32):  bash tools/pmc_fine3.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=pmc_fine3
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_$n -o p -- python3 tools/bench_fine3.py 32 > gpurun_out/${tag}_$n.log 2>&1
  python3 tools/pmc_table.py gpurun_out/${tag}_$n/p_counter_collection.csv gpurun_out/${tag}_$n/p_kernel_trace.csv stencil1_
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES &&
run c SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_c
