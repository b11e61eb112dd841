# usage: ring_pmc.sh <out-tag> "<B> <T> <layers...>" "<counters>" <variant> [...] — one rocprofv3 --pmc pass of tools/ring_check.py per library
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/$1; shift; args=$1; shift; ctr=$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  rm -rf $out/pmc_$v
  timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$v -o p -- python3 tools/ring_check.py $args > $out/pmc_$v.log 2>&1
  f=$(find $out/pmc_$v -name "*counter_collection.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'cconv_' not in k: continue
    k = k.replace('void (anonymous namespace)::', '').split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, c in acc.items():
    print(k, 'x%d' % len(n[k]), ' '.join(f'{a}={v/len(n[k]):.4g}' for a, v in sorted(c.items())))
    if 'SQ_BUSY_CU_CYCLES' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        print('   mfma util %.3f' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES'])))
PY
  find $out/pmc_$v -name "*.csv" -delete; find $out/pmc_$v -name "*.db" -delete
done
