# usage: kpmc.sh <out-tag> "<counters>" <kernel-substring> <python script + args...> — one rocprofv3 --pmc pass, per-kernel averages
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/$1; shift; ctr=$1; shift; pat=$1; shift; mkdir -p $out; rm -rf $out/pmc
timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc -o p -- python3 "$@" > $out/pmc.log 2>&1
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if sys.argv[2] not in k: continue
    k = k.replace('void (anonymous namespace)::', '').split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, c in acc.items():
    print(k, 'x%d' % len(n[k]))
    for a, v in sorted(c.items()): print(f'   {a:32s} {v/len(n[k]):14.4g}')
PY
rm -rf $out/pmc
