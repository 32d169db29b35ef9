# Counters of the first- and second-generation fp32 kernels on one layer / direction, each with its plan pinned:
#   LAYER=conv2d_1 MODE=fwd bash tools/pmc_gen2.sh        -> gpurun_out/pmc_gen2_<layer>_<mode>.txt
# (effective clock = GRBM_GUI_ACTIVE / 8 / duration; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x clock x duration))
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=${LAYER:-conv2d_1}; M=${MODE:-fwd}
out=gpurun_out/pmc_gen2_${L}_${M}.txt
: > $out
run() { tag=$1; cfg=$2; shift 2
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pg2_$tag -o p -- python3 tools/run_layer.py $L $M $cfg ${SPLITK:-1} 8 > gpurun_out/pg2_$tag.log 2>&1
  echo "== cfg $cfg ${STREAMK:+streamk $STREAMK} : $*" >> $out
  python3 tools/pmc_table.py gpurun_out/pg2_$tag/p_counter_collection.csv gpurun_out/pg2_$tag/p_kernel_trace.csv igemm >> $out
  rm -rf gpurun_out/pg2_$tag
}
for cfg in ${CFGS:-7 11}; do
  [ -n "$STREAMK" ] && export A3D_FORCE_STREAMK=$STREAMK
  run a$cfg $cfg GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA &&
  run b$cfg $cfg SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
done
cat $out
