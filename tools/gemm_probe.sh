#!/bin/bash
# Per-kernel times of the LSTM projection GEMMs inside the train step (rocprofv3 kernel stats).  usage (GPU box): tools/gemm_probe.sh <tag>
set -e -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-cpu-baseline --no-native-line --no-sub-lines --steps 10 --warmup 3 > $out/stats.log 2>&1
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('gemm_f32','Cijk','atb_chunks','elementwise_kernel_manual','lstm_')):
        print(f"{float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>5}  {n[:100]}")
PY
