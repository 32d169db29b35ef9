cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc0_$n -o p -- python3 tools/run_layer.py ${LAYER:-conv2d_0} ${MODE:-bwd_f} -1 > gpurun_out/pmc0_$n.log 2>&1
  python3 tools/pmc_table.py gpurun_out/pmc0_$n/p_counter_collection.csv gpurun_out/pmc0_$n/p_kernel_trace.csv igemm_kernel
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA &&
run c SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
rm -rf gpurun_out/pmc0_*
