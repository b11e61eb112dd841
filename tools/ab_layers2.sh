# usage: ab_layers2.sh "<B> <T>" <variantA> <variantB> ... : conv_layers_bench per library, alternating, 2 rounds
args=$1; shift
for i in 1 2; do for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/conv_layers_bench.py $args 2>&1 | grep -E "^(enc|dec|total)"
done; done
