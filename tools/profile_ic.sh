#!/bin/bash
# The Infinity-Cache evidence of a round (profiles/<tag>_ic_evidence.txt), split from profile_round.sh because its batch-256
# counter passes take minutes:   bash tools/profile_ic.sh r04
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
P=$out/profiles
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# HBM or Infinity Cache? (tools/ic_evidence.py: the dominant kernel at batch 32 and 256, tile-major and K-sliced shares)
: > $P/${tag}_ic_evidence.txt
for b in 32 256; do for s in 0 1; do
  A3D_SK_SLICED=$s python3 tools/ic_evidence.py $b 2> /dev/null | grep "^{" >> $P/${tag}_ic_evidence.txt
done; done
export A3D_SK_SLICED
for b in 32 256; do for s in 0 1; do
  A3D_SK_SLICED=$s
  echo "batch $b sliced $s" >> $P/${tag}_ic_evidence.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/ic_f -o p -- python3 tools/ic_evidence.py $b > /dev/null 2>&1
  python3 tools/pmc_table.py $out/ic_f/p_counter_collection.csv $out/ic_f/p_kernel_trace.csv "igemm_kernel<2" >> $P/${tag}_ic_evidence.txt
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/ic_r -o p -- python3 tools/ic_evidence.py $b > /dev/null 2>&1
  python3 tools/pmc_table.py $out/ic_r/p_counter_collection.csv $out/ic_r/p_kernel_trace.csv "igemm_kernel<2" >> $P/${tag}_ic_evidence.txt
  rm -rf $out/ic_f $out/ic_r
done; done
unset A3D_SK_SLICED
echo "ic evidence done"
ls $P
