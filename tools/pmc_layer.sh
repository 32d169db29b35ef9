set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$n -- python3 tools/run_layer.py conv2d_1 fwd -1 > gpurun_out/pmc_$n.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA &&
run c SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT &&
run d GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum
ls gpurun_out/pmc_*
