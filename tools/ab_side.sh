#!/bin/bash
# Same-box A/B of the train step with the weight-gradient kernels on the main stream (A: DCS_WGRAD_SIDE=0) and on the side stream (B).
# usage (GPU box, repo root): tools/ab_side.sh <tag>
set -e -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; mkdir -p $out
F="--no-cpu-baseline --no-native-line --no-sub-lines --steps 20 --warmup 5"
for i in 1 2; do
  DCS_WGRAD_SIDE=0 timeout -k 10 300 python3 $R/bench.py $F > $out/A$i.json 2> $out/A$i.log
  DCS_WGRAD_SIDE=1 timeout -k 10 300 python3 $R/bench.py $F > $out/B$i.json 2> $out/B$i.log
done
python3 - <<PY
import json
for n in ('A1','B1','A2','B2'):
    d=json.loads(open('$out/'+n+'.json').read().strip().splitlines()[-1])
    print(n, round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],3), round(d['roofline']['frac'],3))
PY
