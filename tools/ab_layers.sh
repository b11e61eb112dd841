set -e
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  python tools/conv_layers_bench.py 32 256 > $out/layers_$v.txt 2>&1
done
for v in "$@"; do echo "== $v"; grep -E "^(enc|dec|total)" $out/layers_$v.txt | awk '{print $1, $6, $11, $16}' | tr '\n' ';'; echo; done
