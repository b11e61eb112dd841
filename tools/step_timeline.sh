#!/bin/bash
# tools/step_timeline.sh <outdir> [bench flags]  (GPU box, repo root): timeline of one replayed train step with queues and overlaps
set -e -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_train -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-native-line --no-sub-lines --steps 4 --warmup 2 "$@" > $out/kt_train.log 2>&1
f=$(find $out/kt_train -name "*kernel_trace.csv" | head -1)
head -1 $f > $out/trace_header.txt
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 6 > $out/step_timeline_train.txt
rm -rf $out/kt_train
tail -1 $out/step_timeline_train.txt
