#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <mode> <outdir-under-gpurun_out>
mode=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$2
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --mode $mode --steps 3 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
