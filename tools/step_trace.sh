#!/bin/bash
# One replayed train step kernel by kernel (run on the GPU box from the repo root): tools/step_trace.sh <outdir> [bench flags]
set -e -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_train -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-native-line --steps 4 --warmup 2 "$@" > $out/kt_train.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py $(find $out/kt_train -name "*kernel_trace.csv" | head -1) 6 > $out/step_sequence_train.txt
rm -rf $out/kt_train
tail -1 $out/step_sequence_train.txt
