# usage: ring_prof.sh <out-tag> "<B> <T> <layers...>" <variant> [...] — rocprofv3 kernel durations of tools/ring_check.py under diagnostic libraries
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
out=gpurun_out/$1; shift; args=$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  rm -rf $out/prof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o p -- python3 tools/ring_check.py $args > $out/prof_$v.log 2>&1
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'cconv_ring' in n or 'cconv_mfma_kernel' in n:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us avg  {float(r['MinNs'])/1e3:8.1f} min  x{r['Calls']:>4}  {n[:90]}")
PY
done
