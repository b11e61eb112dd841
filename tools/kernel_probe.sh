#!/bin/bash
# Average in-step duration of the kernels whose names match a pattern (rocprofv3 kernel stats over a short train run).
# usage (GPU box): tools/kernel_probe.sh <tag> <python-regex> [bench flags]
set -e -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; pat=$2; shift; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-cpu-baseline --no-native-line --no-sub-lines --steps 10 --warmup 3 "$@" > $out/stats.log 2>&1
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print(f"{float(r['AverageNs'])/1e3:8.2f} us (min {float(r['MinNs'])/1e3:6.2f}) x {r['Calls']:>5}  {r['Name'][:100]}")
PY
rm -rf $out/stats
