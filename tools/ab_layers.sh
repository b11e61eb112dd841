# usage: ab_layers.sh <outdir-tag> <variant> [<variant> ...]  — per-layer conv bench under diagnostic libraries (tools/exp_build.py),
# "default" = the shipped library; prints forward / data-gradient / weight-gradient us per layer and the totals
set -e
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  python tools/conv_layers_bench.py 32 256 > $out/layers_$v.txt 2>&1
done
for v in "$@"; do echo "== $v"; grep -E "^(enc|dec)" $out/layers_$v.txt | awk '{printf "%s %s/%s/%s; ", $1, $6, $11, $16}'; echo; grep total $out/layers_$v.txt; done
