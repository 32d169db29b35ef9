#!/bin/bash
# One step's timeline of a bench configuration:  bash tools/timeline_run.sh <outfile> <anchor kernel> <bench args...>
set -e -o pipefail
out=$1; anchor=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/tl_$$
rocprofv3 --kernel-trace --output-format csv -d $d -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rank --also "" "$@" > $d.log 2>&1
python3 tools/timeline.py "$d/kt_kernel_trace.csv" "$anchor" -v > "$out"
rm -rf $d $d.log
