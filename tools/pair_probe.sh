#!/bin/bash
# per-launch durations of one kernel name inside a replayed step (kernel trace), in launch order
set -e -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; pat=$2; shift; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 $R/bench.py --no-cpu-baseline --no-native-line --no-sub-lines --steps 6 --warmup 3 "$@" > $out/kt.log 2>&1
f=$(find $out/kt -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$pat" <<'PY'
import csv,re,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
prev_end=None; out=[]
for i,r in enumerate(rows):
    if re.search(sys.argv[2], r['Kernel_Name']):
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        gap=(int(r['Start_Timestamp'])-int(rows[i-1]['End_Timestamp']))/1e3 if i else 0
        out.append((d,gap,rows[i-1]['Kernel_Name'][:30] if i else ''))
for d,g,p in out[-16:]: print(f'{d:6.2f} us  gap-before {g:6.2f}  after {p}')
PY
rm -rf $out/kt
