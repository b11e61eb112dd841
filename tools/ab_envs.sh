#!/bin/bash
# Same-box A/B/C... of the train step under environment variants.  usage (GPU box, repo root):
#   tools/ab_envs.sh <tag> "<bench flags>" "<env A>" "<env B>" ...   (each env a string of assignments, may be empty)
set -e -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; mkdir -p $out; shift
F="--no-cpu-baseline --no-native-line --no-sub-lines --steps 20 --warmup 5 $1"; shift
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for e in "$@"; do
    env $e timeout -k 10 300 python3 $R/bench.py $F > $out/v${i}_$rep.json 2> $out/v${i}_$rep.log
    python3 -c "
import json,sys
d=json.loads(open('$out/v${i}_$rep.json').read().strip().splitlines()[-1])
print('v$i rep$rep [$e]', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],3), round(d['roofline']['frac'],3))"
    i=$((i+1))
  done
done
