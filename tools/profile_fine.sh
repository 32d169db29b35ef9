#!/bin/bash
# Fine-phase evidence: kernel stats + one step's timeline, fp32 at B = 32 and bf16 storage at B = 64.
#     bash tools/profile_fine.sh r05   ->  gpurun_out/prof_<tag>/profiles/<tag>_*fine*
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
P=$out/profiles
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktf -o kt -- python3 bench.py --phase fine --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rank --also "" > $out/bench_fine_under_rocprof.log 2>&1
cp $out/ktf/kt_kernel_stats.csv $P/${tag}_bench_fine_kernel_stats.csv
grep '^{' $out/bench_fine_under_rocprof.log | tail -1 > $P/${tag}_bench_fine_under_rocprof.json
python3 tools/timeline.py "$out/ktf/kt_kernel_trace.csv" resize_kernel -v > $P/${tag}_step_timeline_fine.txt
rm -rf $out/ktf
echo "fp32 fine done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktf16 -o kt -- python3 bench.py --phase fine --batch 64 --precision bf16s --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rank --also "" > $out/bench_fine16_under_rocprof.log 2>&1
cp $out/ktf16/kt_kernel_stats.csv $P/${tag}_bench_fine_bf16_storage_kernel_stats.csv
grep '^{' $out/bench_fine16_under_rocprof.log | tail -1 > $P/${tag}_bench_fine_bf16_storage_under_rocprof.json
python3 tools/timeline.py "$out/ktf16/kt_kernel_trace.csv" resize_kernel -v > $P/${tag}_step_timeline_fine_bf16_storage.txt
rm -rf $out/ktf16
echo "bf16s fine done"
ls $P
