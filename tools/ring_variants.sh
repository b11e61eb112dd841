# usage: ring_variants.sh <out-tag> "<B> <T> <layers...>" <variant> [<variant> ...] — tools/ring_check.py under diagnostic libraries
# (tools/exp_build.py <variant> conv_ring.hip -D...), "default" = the shipped library; same box, one process each
out=gpurun_out/$1; shift; args=$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset DCS_LIB_PATH; else export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so; fi
  timeout -k 10 120 python tools/ring_check.py $args > $out/ring_$v.txt 2>&1
  echo "== $v"; grep -E "^(enc|dec|total)" $out/ring_$v.txt
done
