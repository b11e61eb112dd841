cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for v in ${VARIANTS:-default m16_1 m16_3}; do
  if [ $v = default ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  rm -rf gpurun_out/m16p; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m16p -o p -- python3 tools/conv_layers_bench.py 32 256 > /dev/null 2>&1
  f=$(find gpurun_out/m16p -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'cconv_mfma16' in r['Name']: print(f"{float(r['AverageNs'])/1e3:8.1f} us avg x{r['Calls']}  {r['Name'][:80]}")
PY
done; rm -rf gpurun_out/m16p
